"""Winograd F(2x2,3x3) prototype (bd_conv3x3_wino): correctness against the fp64 convolution and timing against conv_ps3 on the same shapes.
usage: python scripts/wino/check.py [iters]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from baddiffusion_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wino_probe as wp

torch.manual_seed(0)
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SMALL = [(2, 16, 32, 64), (3, 32, 64, 128), (2, 16, 128, 64)]
BIG = [(128, 32, 128, 128), (128, 16, 256, 256), (128, 16, 512, 256), (128, 32, 256, 128)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def ref64(x, w, bias):
    y = F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().permute(0, 3, 1, 2).cpu(), bias.double().cpu(), padding=1)
    return y.permute(0, 2, 3, 1)


out = []
for (B, S, Cin, Cout) in SMALL + BIG:
    x = torch.randn(B, S, S, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    bias = torch.randn(Cout, device=dev)
    u = wp.wino_weights(w, 1)
    y = wp.conv3x3_wino(x, u, bias=bias)
    torch.cuda.synchronize()
    rec = {"B": B, "S": S, "Cin": Cin, "Cout": Cout}
    nb = min(B, 4)
    r = ref64(x[:nb], w, bias).to(dev)
    e = (y[:nb].double() - r)
    rec["wino_rel_l2_vs_fp64"] = float(e.norm() / r.norm())
    rec["wino_max_over_max"] = float(e.abs().max() / r.abs().max())
    if Cin % 32 == 0 and Cout % 128 == 0:
        xs = ops.split_rows(x); ws = ops.split_bf16(w)
        yd = ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias)
        ed = (yd[:nb].double() - r)
        rec["direct_rel_l2_vs_fp64"] = float(ed.norm() / r.norm())
        # extras: rowbias + residual epilogue
        rb = torch.randn(B, Cout, device=dev); res = torch.randn(B, S, S, Cout, device=dev)
        y2 = wp.conv3x3_wino(x, u, bias=bias, rowbias=rb, residual=res, out_scale=0.7)
        want = (y + rb[:, None, None, :] + res) * 0.7
        rec["epilogue_max_abs_diff"] = float((y2 - want).abs().max())
        if B >= 64:
            fl = 2.0 * B * S * S * Cin * Cout * 9
            t_w = timeit(lambda: wp.conv3x3_wino(x, u, bias=bias))
            t_d = timeit(lambda: ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias))
            t_u = timeit(lambda: wp.wino_weights(w, 1))
            rec.update({"wino_us": t_w, "direct_ps_us": t_d, "wino_weights_us": t_u, "wino_alg_tflops": fl / t_w / 1e6, "direct_alg_tflops": fl / t_d / 1e6,
                        "speedup": t_d / t_w})
    # data gradient through the same kernel (rotated, transposed weights): compare with autograd-free fp64 transposed conv on a few samples
    if Cout % 16 == 0 and Cin % 64 == 0:
        dy = torch.randn(nb, S, S, Cout, device=dev)
        ut = wp.wino_weights(w, -1)
        dx = wp.conv3x3_wino(dy, ut)
        rdx = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2).cpu(), w.double().permute(0, 3, 1, 2).cpu(), padding=1).permute(0, 2, 3, 1).to(dev)
        rec["dgrad_rel_l2_vs_fp64"] = float((dx.double() - rdx).norm() / rdx.norm())
    print(json.dumps(rec), flush=True)
    out.append(rec)
json.dump(out, open("gpurun_out/r06_wino_probe_check.json", "w"), indent=1)
