"""How far is each compute mode from the reference's golden vectors at BASELINE configs[1]'s real size?  (G10: tests/golden/full_size.npz, vectors the imported
reference produced for the DDPM-CIFAR10-32 topology at batch 128.)  Prints one JSON object: relative L2 error of the prediction rows, relative loss error,
relative clip-norm error per mode.  usage (GPU): python scripts/golden_error_report.py > gpurun_out/r06_golden_error.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sched_ref                      # noqa: E402  (test infrastructure: this script is a checker, not product)
from oracle import unet_ref as U                  # noqa: E402
from tests.golden import cases as C               # noqa: E402
import baddiffusion_amd.unet as unet              # noqa: E402
from baddiffusion_amd import ops                  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "full_size.npz"))
tag = "cifar128"
cfg = U.CIFAR10_32
_, a, ac = sched_ref.make_tables()
out = {}
for mode in ("f32", "bf16x3"):
    m = unet.unet_from_config(cfg).cuda()
    m.load_state_dict(U.gen_params(cfg, 0))
    m.set_compute_mode(mode)
    x0, R, t, eps = C.train_inputs(cfg, 128)
    xn, tg = ops.qsample(x0.cuda(), R.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda())
    pred = m(xn.permute(0, 3, 1, 2), t.cuda(), return_dict=False)[0]
    pd = pred.detach()
    want = torch.as_tensor(g[f"{tag}_pred_rows"]).double()
    got = pd[list(C.FULL_ROWS)].cpu().double()
    loss, dp = ops.loss_fwd_bwd(pred.permute(0, 2, 3, 1), tg, "l2")
    pred.backward(dp.reshape(pred.permute(0, 2, 3, 1).shape).permute(0, 3, 1, 2))
    norm = float(ops.sumsq(m.flat.grad).sqrt()) if hasattr(ops.sumsq(m.flat.grad), "sqrt") else None
    out[mode] = {"pred_rel_l2": float((got - want).norm() / want.norm()), "loss_rel": abs(float(loss) - float(g[f"{tag}_loss"])) / abs(float(g[f"{tag}_loss"])),
                 "clip_norm_rel": abs(norm - float(g[f"{tag}_total_norm"])) / float(g[f"{tag}_total_norm"]) if norm is not None else None}
print(json.dumps(out, indent=1))
