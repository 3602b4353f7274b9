"""GPU parity, round 2: the entry points and configurations the round-1 review found untested.

  * TrainEngine.train_step (the function bench.py and the CLI time): uint8 images -> fused gather / flip / normalize /
    blend / q_sample -> step, against oracle.train_ref at B = 8 and against train_step_batch at the full B = 128;
  * the per-GPU shares of BASELINE configs[3] (256x256 net, batch 4) and configs[4] (CIFAR topology, 256 DDIM-50 chains);
  * 4 samples of the B = 128 forward against the oracle;
  * bd_adam_clip_dev (device-resident step scalars) against bd_adam_clip;
  * the fused row gather + horizontal flip of bd_poison_qsample against torch indexing / flip;
  * the LDS-DMA convolution family on split planes (bd_conv3x3_ps, bd_conv3x3_ps_wgrad, bd_split_rows, bd_split_wt,
    GroupNorm's split outputs) against the fp32 igemm path;
  * two ranks (gloo, sharing the one GPU): divergent initial weights are made equal, the CIFAR-topology step equals the
    one-process global batch, bench.py --gpus 2 and the CLI train loop run end to end.
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star) unless a tighter one is stated; index / mask data bit-exact."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BD
from oracle import sched_ref, train_ref
from oracle import unet_ref as U
from tests.golden import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_model(cfg, seed, dev):
    from baddiffusion_amd.unet import unet_from_config
    m = unet_from_config(cfg).to(dev)
    m.load_state_dict(U.gen_params(cfg, seed))
    return m


def _u8_batch(B, S, seed):
    return torch.randint(0, 256, (B, S, S, 3), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)


# ------------------------------------------------------------------------------------------------ 1(a)
def test_train_engine_uint8_step_vs_oracle(gpu):
    """TrainEngine.train_step on raw uint8 rows (BOX_14 trigger, CORNER target, 1 of 8 rows poisoned) on the
    DDPM-CIFAR10-32 topology: loss, clipped-gradient norm and the post-Adam weights against oracle.train_ref
    (reference: baddiffusion.py:590-615, dataset.py:288-315, loss.py:257-307)."""
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = U.CIFAR10_32
    B, S = 8, 32
    u8 = _u8_batch(B, S, 11)
    pois = torch.zeros(B, dtype=torch.bool); pois[3] = True
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S); tgt = bd.get_target("CORNER", trig)
    assert torch.equal(trig, BD.get_trigger("BOX_14", 3, S)) and torch.equal(tgt, BD.get_target("CORNER", BD.get_trigger("BOX_14", 3, S)))
    eps = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(12))
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(13))
    m = make_model(cfg, 0, gpu)
    eng = TrainEngine(m, DDPMScheduler(), lr=2e-4)
    loss = eng.train_step(u8.cuda(), pois.cuda(), trig.cuda(), tgt.cuda(), eps.cuda(), t.cuda())
    # oracle
    _, a, ac = sched_ref.make_tables()
    x = torch.stack([BD.image_u8_to_float(u8[i]) for i in range(B)])
    R, x0 = BD.make_batch(x, pois, trig, tgt)
    P = U.gen_params(cfg, 0)
    ref_loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
    newP, _, norm = train_ref.clip_and_adam(P, G, {}, 2e-4, 1)
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    assert abs(float(eng.grad_norm) - float(norm)) < 1e-3 * float(norm)
    sd = m.state_dict()
    diffs = torch.cat([(sd[k].cpu() - newP[k]).abs().flatten() for k in newP if not k.endswith("key.bias")])
    # Adam's first update is ~lr * sign(g): single elements whose gradient is rounding noise may flip (2 lr); the bulk agrees
    assert float(diffs.max()) <= 2.2 * 2e-4 and float((diffs > 5e-5).float().mean()) < 1e-3 and float(diffs.mean()) < 2e-6


def test_train_engine_uint8_step_full_batch_equals_collated_step(gpu):
    """BASELINE configs[1] at full size (B = 128, poison 0.1): the fused uint8 entry point == the reference calling
    convention (collated x_start / R batches, train_step_batch) on the same rows -- same kernels after q_sample."""
    from baddiffusion_amd import ops
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = U.CIFAR10_32
    B, S = 128, 32
    u8 = _u8_batch(B, S, 21).cuda()
    pois = (torch.arange(B) % 10 == 0).cuda()
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S).cuda(); tgt = bd.get_target("CORNER", trig.cpu()).cuda()
    eps = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(22)).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(23)).cuda()
    m1, m2 = make_model(cfg, 0, gpu), make_model(cfg, 0, gpu)
    e1, e2 = TrainEngine(m1, DDPMScheduler(), lr=2e-4), TrainEngine(m2, DDPMScheduler(), lr=2e-4)
    l1 = e1.train_step(u8, pois, trig, tgt, eps, t)
    _, _, R, x0 = ops.poison_qsample(u8, pois, trig, tgt, eps, t, e2.alphas, e2.alphas_cumprod, want_batch=True)
    l2 = e2.train_step_batch(x0, R, eps, t)
    assert torch.isfinite(l1) and abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2))
    assert abs(float(e1.grad_norm) - float(e2.grad_norm)) <= 1e-5 * float(e2.grad_norm)
    assert relerr(m1.flat, m2.flat) < 1e-6


# ------------------------------------------------------------------------------------------------ 1(b)
def test_celeba256_batch4_train_step(gpu):
    """per-GPU share of BASELINE configs[3]: DDPM-CELEBA-HQ-256 network, batch 4 (32 global / 8 GPUs), one train step:
    finite, split-bf16 == exact-fp32 contraction, and sample i of the batch == the same sample run alone."""
    cfg = U.CELEBA_HQ_256
    m = make_model(cfg, 5, gpu)
    B = 4
    x = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(3)).cuda()
    t = torch.tensor([417, 3, 999, 500]).cuda()
    dout = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(4)).cuda() / (B * 3 * 65536)
    res = {}
    for mode in ("f32", "bf16x3"):
        m.set_compute_mode(mode)
        m.flat.grad = None
        out = m(x, t, return_dict=False)[0]
        out.backward(dout)
        assert torch.isfinite(out).all() and torch.isfinite(m.flat.grad).all()
        res[mode] = (out.detach().clone(), m.flat.grad.detach().clone())
    assert relerr(res["bf16x3"][0], res["f32"][0]) < 1e-4
    assert relerr(res["bf16x3"][1], res["f32"][1]) < 1e-3
    with torch.no_grad():
        alone = m(x[2:3], t[2:3], return_dict=False)[0]
    assert relerr(alone, res["bf16x3"][0][2:3]) < 5e-5


# ------------------------------------------------------------------------------------------------ 1(c)
def test_cifar_ddim50_share_of_sampling_config(gpu, tmp_path):
    """per-GPU share of BASELINE configs[4]: CIFAR topology, 256 chains (2048 / 8 GPUs), DDIM 50 steps.  The first two
    steps of 4 chains against the oracle (pipeline_ddim.py:114-121, scheduling_ddim.py:261-381), the full loop finite
    and in [0,1], chunked == unchunked, and batch_sampling_save(rank, world=8) writes exactly this rank's index range."""
    from baddiffusion_amd.model import batch_sampling, batch_sampling_save
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDIMScheduler
    cfg = U.CIFAR10_32
    m = make_model(cfg, 0, gpu)
    P = U.gen_params(cfg, 0)
    _, a, ac = sched_ref.make_tables()
    sch = DDIMScheduler(clip_sample=False)
    sch.set_timesteps(50)
    ts = [int(v) for v in sch.timesteps]
    init = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(9))
    # two steps of 4 chains, product vs oracle
    x = init[:4].clone(); xg = x.cuda()
    with torch.no_grad():
        for i in range(2):
            tt = ts[i]
            e_ref = U.unet_forward(cfg, P, x, tt)
            x = sched_ref.ddim_step(ac, e_ref, tt, x, num_inference_steps=50, eta=0.0, clip_sample=False)[0]
            e = m(xg, tt).sample
            xg = sch.step(e, tt, xg).prev_sample
    assert relerr(xg, x) < 1e-3
    # the full loop
    pipe = DDIMPipeline(m, DDIMScheduler(clip_sample=False))
    call = lambda **kw: pipe(num_inference_steps=50, **kw)
    full = batch_sampling(256, call, init=init, max_batch_n=256)
    assert full.shape == (256, 32, 32, 3) and np.isfinite(full).all() and full.min() >= 0.0 and full.max() <= 1.0
    part = batch_sampling(64, call, init=init[:64], max_batch_n=24)          # 24 + 24 + 16: chunks of other sizes
    np.testing.assert_allclose(part, full[:64], rtol=0, atol=2e-3)           # kernels may differ per batch size: rounding only
    assert float(np.abs(part - full[:64]).mean()) < 2e-5
    # rank sharding of the 2048-chain job: rank 5 of 8 owns samples [1280, 1536)
    short = lambda **kw: pipe(num_inference_steps=2, **kw)
    big_init = torch.randn(2048, 3, 32, 32, generator=torch.Generator().manual_seed(10))
    batch_sampling_save(2048, short, str(tmp_path / "s"), init=big_init, max_batch_n=2048, rank=5, world=8)
    names = sorted(int(n[:-4]) for n in os.listdir(tmp_path / "s"))
    assert names == list(range(1280, 1536))
    from PIL import Image
    own = pipe(batch_size=2, init=big_init[1280:1282], output_type=None, num_inference_steps=2).images
    png = np.asarray(Image.open(tmp_path / "s" / "1281.png"))
    assert np.abs(png.astype(int) - (own[1] * 255).round().astype(int)).max() <= 1


# ------------------------------------------------------------------------------------------------ 1(d)
def test_cifar_full_batch_forward_samples_vs_oracle(gpu):
    """4 samples (both half-batch pipelines) of the B = 128 DDPM-CIFAR10-32 forward against the oracle's outputs"""
    cfg = U.CIFAR10_32
    m = make_model(cfg, 0, gpu)
    P = U.gen_params(cfg, 0)
    B = 128
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), return_dict=False)[0].cpu()
        idx = [0, 63, 64, 127]
        ref = U.unet_forward(cfg, P, x[idx], t[idx])
    for j, i in enumerate(idx):
        assert relerr(out[i], ref[j]) < 1e-4, (i, relerr(out[i], ref[j]))


# ------------------------------------------------------------------------------------------------ 1(f)
def test_adam_clip_dev_matches_host_scalar_form(gpu):
    """bd_adam_clip_dev (step size / bias correction read from DEVICE memory: what a replayed hipGraph needs) is the same
    update as bd_adam_clip with host scalars, bit for bit, over several steps"""
    import ctypes
    import math
    from baddiffusion_amd import _lib as L
    from baddiffusion_amd import ops
    lib = L.load()
    n = 100003
    g = torch.Generator().manual_seed(5)
    p1 = torch.randn(n, generator=g).cuda(); p2 = p1.clone()
    m1 = torch.zeros(n).cuda(); v1 = torch.zeros(n).cuda(); m2 = torch.zeros(n).cuda(); v2 = torch.zeros(n).cuda()
    lr, b1, b2, eps = 2e-4, 0.9, 0.999, 1e-8
    n1 = torch.zeros((), device="cuda"); n2 = torch.zeros((), device="cuda")
    for step in range(1, 4):
        gr = (torch.randn(n, generator=g) * 3).cuda()
        ss = ops.sumsq(gr)
        ops.adam_clip(p1, gr, m1, v1, ss, step, lr, 1.0, (b1, b2), eps, grad_norm_out=n1)
        hyper = torch.tensor([lr / (1 - b1 ** step), math.sqrt(1 - b2 ** step)], dtype=torch.float32).cuda()
        L.check(lib.bd_adam_clip_dev(p2.data_ptr(), gr.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, ss.data_ptr(), 1.0,
                                     hyper.data_ptr(), b1, b2, eps, n2.data_ptr(), L.stream()), "bd_adam_clip_dev")
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and float(n1) == float(n2)


def test_train_step_graph_replay_bit_identical(gpu):
    """TrainEngine(use_graph=True): the whole step replayed as one hipGraph (device-resident Adam scalars, static input
    buffers) leaves bit-identical weights, moments, loss and gradient norm compared with the eager launch sequence,
    over a warm-up step, the capture step and two replays with different inputs and learning rates."""
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = C.SMALL_CFGS["small"]
    B, S = 6, 16
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S).cuda(); tgt = bd.get_target("CORNER", trig.cpu()).cuda()
    m1, m2 = make_model(cfg, 7, gpu), make_model(cfg, 7, gpu)
    kw = dict(lr=1e-3, lr_warmup_steps=2, num_training_steps=10)
    e1, e2 = TrainEngine(m1, DDPMScheduler(), use_graph=False, **kw), TrainEngine(m2, DDPMScheduler(), use_graph=True, **kw)
    assert e2.use_graph and not e1.use_graph
    data = _u8_batch(40, S, 41).cuda()
    g = torch.Generator().manual_seed(42)
    for step in range(4):
        rows = torch.randint(0, 40, (B,), generator=g).cuda()
        flip = (torch.rand(B, generator=g) < 0.5).to(torch.uint8).cuda()
        pois = (torch.rand(B, generator=g) < 0.3).cuda()
        eps = torch.randn(B, 3, S, S, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
        l1 = e1.train_step(data, pois, trig, tgt, eps, t, row_index=rows, flip=flip)
        l2 = e2.train_step(data, pois, trig, tgt, eps, t, row_index=rows, flip=flip)
        assert float(l1) == float(l2) and float(e1.grad_norm) == float(e2.grad_norm), step
        assert torch.equal(m1.flat, m2.flat) and torch.equal(e1.m, e2.m) and torch.equal(e1.v, e2.v), step
    assert e1.opt_step == e2.opt_step == 4 and any(v["graph"] is not None for v in e2._graphs.values())


# ------------------------------------------------------------------------------------------------ 1(g)
def test_fused_gather_and_flip(gpu):
    """row_index / flip of bd_poison_qsample == indexing the resident array and torch.flip along W
    (the DataLoader's shuffle + RandomHorizontalFlip, dataset.py:127-128), bit for bit; DatasetLoader.device_batch_rows
    feeds exactly the rows / flips that device_batches materialises."""
    from baddiffusion_amd import ops
    from baddiffusion_amd.dataset import Backdoor, DatasetLoader
    S, N, B = 32, 40, 8
    imgs = _u8_batch(N, S, 31).cuda()
    rows = torch.tensor([5, 39, 0, 17, 17, 3, 22, 8]).cuda()
    flip = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0], dtype=torch.uint8).cuda()
    pois = torch.tensor([1, 0, 0, 1, 0, 0, 0, 1], dtype=torch.bool).cuda()
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S).cuda(); tgt = bd.get_target("CORNER", trig.cpu()).cuda()
    eps = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(32)).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(33)).cuda()
    _, a, ac = sched_ref.make_tables()
    a, ac = a.cuda(), ac.cuda()
    fused = ops.poison_qsample(imgs, pois, trig, tgt, eps, t, a, ac, want_batch=True, want_image=True, row_index=rows, flip=flip)
    gathered = imgs[rows]
    gathered = torch.where(flip.bool()[:, None, None, None], gathered.flip(2), gathered)
    plain = ops.poison_qsample(gathered, pois, trig, tgt, eps, t, a, ac, want_batch=True, want_image=True)
    for u, v in zip(fused, plain):
        assert torch.equal(u, v)
    dsl = DatasetLoader(root=None, name="CIFAR10", batch_size=8, seed=4, num_images=N, device="cuda")
    dsl.set_poison("BOX_14", "CORNER", poison_rate=0.2).prepare_dataset("FIXED").to_device("cuda")
    for (r, f, p), (img, p2) in zip(dsl.device_batch_rows(epoch=2), dsl.device_batches(epoch=2)):
        g = dsl.device_images[r]
        g = torch.where(f.bool()[:, None, None, None], g.flip(2), g)
        assert torch.equal(g, img) and torch.equal(p, p2)


# ------------------------------------------------------------------------------------------------ split-plane convolution family
@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 16, 128, 128), (3, 8, 256, 128), (128, 16, 128, 256), (5, 4, 128, 384), (64, 32, 128, 128)])
def test_conv_ps_family_vs_igemm(gpu, B, S, Cin, Cout):
    """bd_conv3x3_ps (+1 / -1) and bd_conv3x3_ps_wgrad on split planes against the exact-fp32 igemm path (1e-4) and the
    split-bf16 igemm path (rounding level: same products, possibly another K-split order); GroupNorm's split outputs
    and bd_split_rows are bit-identical to splitting the fp32 result."""
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = torch.randn(B, S, S, Cin, generator=g).cuda(); dy = torch.randn(B, S, S, Cout, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda(); bias = torch.randn(Cout, generator=g).cuda()
    rb = torch.randn(B, Cout, generator=g).cuda(); res = torch.randn(B, S, S, Cout, generator=g).cuda()
    ws, xs, dys, wts = ops.split_bf16(w), ops.split_rows(x), ops.split_rows(dy), ops.split_wT(w)
    # split planes: [rows, C/32, (hi|lo), 32] == the flat 32-element-block split of the same contiguous buffer
    assert torch.equal(xs.reshape(-1), ops.split_bf16(x).reshape(-1))
    for kw in (dict(), dict(rowbias=rb), dict(residual=res, out_scale=0.7)):
        new = ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias, **kw)
        assert relerr(new, ops.conv3x3_fwd(x, w, bias, mode=0, **kw)) < 1e-4
        assert relerr(new, ops.conv3x3_fwd(x, w, bias, mode=1, **kw)) < 2e-6
    # combined epilogue flags (every EPI combination has its own kernel instantiation): rowbias + residual,
    # residual + accumulate, rowbias + accumulate, all three -- against the same sum built from single-flag launches
    base = ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias)
    rbx = rb[:, None, None, :]
    prev = torch.randn(B, S, S, Cout, generator=torch.Generator().manual_seed(5)).cuda()
    for kw, want in ((dict(rowbias=rb, residual=res), 0.7 * (base + rbx + res)),
                     (dict(residual=res, accumulate=True), 0.7 * (base + res) + prev),
                     (dict(rowbias=rb, accumulate=True), 0.7 * (base + rbx) + prev),
                     (dict(rowbias=rb, residual=res, accumulate=True), 0.7 * (base + rbx + res) + prev)):
        got = ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias, out_scale=0.7, out=prev.clone(), **kw)
        assert relerr(got, want) < 2e-6, sorted(kw)
    with pytest.raises(RuntimeError):        # a residual view that is not 16-byte aligned is refused, not faulted on
        ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias, residual=torch.empty(B * S * S * Cout + 1, device="cuda")[1:].view(B, S, S, Cout))
    if Cin % 128 == 0:
        newd = ops.conv3x3_ps(dys, wts, B, S, S, Cout, Cin, -1)
        assert relerr(newd, ops.conv3x3_dgrad(dy, w, (B, S, S, Cin), mode=0)) < 1e-4
        acc = ops.conv3x3_ps(dys, wts, B, S, S, Cout, Cin, -1, out=newd.clone(), accumulate=True)
        assert relerr(acc, 2 * newd) < 1e-6
        dw, db = ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True)
        dw_ref, db_ref = ops.conv3x3_wgrad(x, dy, mode=0, with_db=True)
        assert relerr(dw, dw_ref) < 1e-4 and relerr(db, db_ref) < 1e-4
        dw2 = ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout)
        assert torch.equal(dw, dw2)                    # run-to-run determinism (fixed-order K split)


def test_groupnorm_split_outputs(gpu):
    """bd_gn_fwd.y_split / bd_gn_bwd.dx_split == bd_split_rows of the fp32 outputs, bit for bit (resident and split kernels)"""
    import ctypes as CT
    from baddiffusion_amd import _lib as L
    from baddiffusion_amd import ops
    lib = L.load()
    for (B, HW, Cc) in [(4, 256, 128), (2, 4096, 64), (3, 16, 256)]:
        g = torch.Generator().manual_seed(HW)
        x = torch.randn(B, HW, Cc, generator=g).cuda(); dy = torch.randn(B, HW, Cc, generator=g).cuda()
        ga = (1 + 0.1 * torch.randn(Cc, generator=g)).cuda(); be = (0.1 * torch.randn(Cc, generator=g)).cuda()
        y, mean, rstd = ops.gn_fwd(x, ga, be, 32, 1e-6, True)
        ys = torch.empty(B * HW, Cc // 32, 2, 32, dtype=torch.int16, device="cuda")
        st = torch.empty(2, B, 32, device="cuda")
        ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
        d = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=32, eps=1e-6, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), y=None, ldy=Cc,
                        mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel(), y_split=L.ptr(ys), ldys=Cc)
        L.check(lib.bd_gn_fwd(CT.byref(d), L.stream()), "bd_gn_fwd")
        assert torch.equal(ys, ops.split_rows(y)) and torch.equal(st[0], mean)
        dx, _, _ = ops.gn_bwd(x, ga, be, mean, rstd, dy, 32, True)
        dxs = torch.empty_like(ys)
        dg = torch.empty(Cc, device="cuda"); db = torch.empty(Cc, device="cuda")
        e = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=32, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), mean=L.ptr(mean),
                        rstd=L.ptr(rstd), dy=L.ptr(dy), lddy=Cc, dx=None, lddx=Cc, accumulate_dx=0, dgamma=L.ptr(dg), dbeta=L.ptr(db),
                        workspace=L.ptr(ws), workspace_bytes=ws.numel(), dx_split=L.ptr(dxs), lddxs=Cc)
        L.check(lib.bd_gn_bwd(CT.byref(e), L.stream()), "bd_gn_bwd")
        assert torch.equal(dxs, ops.split_rows(dx))


# ------------------------------------------------------------------------------------------------ 7: two ranks over gloo on one GPU
def _dp_cifar_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    from baddiffusion_amd.unet import unet_from_config
    cfg = U.CIFAR10_32
    x0, R, t, eps = [v.cuda() for v in C.train_inputs(cfg, 4)]
    m = unet_from_config(cfg).cuda()
    m.load_state_dict(U.gen_params(cfg, rank))          # DIFFERENT initial weights per rank: the engine must broadcast rank 0's
    e = TrainEngine(m, DDPMScheduler(), lr=1e-3)
    ret[f"init{rank}"] = m.flat.detach().cpu()
    sl = slice(rank, None, world)
    for _ in range(2):
        e.train_step_batch(x0[sl], R[sl], eps[sl], t[sl])
    torch.cuda.synchronize()
    ret[f"flat{rank}"] = m.flat.detach().cpu()
    ret[f"gn{rank}"] = float(e.grad_norm)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_cifar_topology_sync_and_equal_global_batch(gpu):
    """2 processes x batch 2 on the DDPM-CIFAR10-32 topology: ranks built with different seeds start from rank 0's weights
    (TrainEngine.sync_state), stay bit-identical over 2 steps, and match 1 process x batch 4 (flat-gradient all-reduce per
    backward segment replaces nn.DataParallel, baddiffusion.py:325)."""
    import socket
    import torch.multiprocessing as mp
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_dp_cifar_worker, args=(2, port, ret), nprocs=2, join=True)
    assert torch.equal(ret["init0"], ret["init1"])
    assert torch.equal(ret["flat0"], ret["flat1"]) and ret["gn0"] == ret["gn1"]
    cfg = U.CIFAR10_32
    x0, R, t, eps = [v.cuda() for v in C.train_inputs(cfg, 4)]
    m = make_model(cfg, 0, gpu)
    assert torch.equal(m.flat.detach().cpu(), ret["init0"])
    e = TrainEngine(m, DDPMScheduler(), lr=1e-3)
    for _ in range(2):
        e.train_step_batch(x0, R, eps, t)
    assert abs(float(e.grad_norm) - ret["gn0"]) < 1e-3 * float(e.grad_norm)
    d = (m.flat.detach().cpu() - ret["flat0"]).abs()
    assert float((d > 2e-4).float().mean()) < 1e-3 and float(d.mean()) < 2e-5


def _torchrun(args, env_extra, timeout=600, backend="gloo", nproc=2):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BD_DIST_BACKEND=backend, PYTHONPATH=ROOT, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_end_to_end(gpu, tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one rank per process), with gloo so that
    both ranks can share the test box's single GPU: one JSON line from rank 0 with the whole-job aggregate -- BOUNDED (the driver captures a
    bounded tail of stdout; round 5's 20 KB line did not parse) -- and the detail object in the file the line names."""
    detail = str(tmp_path / "bench_detail.json")
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--sampling-n", "4,2",
                   "--no-celeba", "--no-fid", "--sustain", "0.05", "--dp-sweep-buckets", "0,32", "--dp-sweep-slots", "128,192", "--total", "16",
                   "--detail-file", detail], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0])           # the LAST thing on stdout
    assert len(lines[0]) < 8192
    head = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "distributed", "sampling", "detail_file"):
        assert k in head, k
    assert head["detail_file"] == detail and 0 < head["roofline"]["frac"] < 1 and "workload" in head["config"]
    assert head["distributed"]["measured"]["ms_per_step"] > 0 and head["sampling"]["ddim50_sharded"]["value"] > 0
    d = json.load(open(detail))
    assert abs(head["value"] - d["value"]) < 1e-4 * d["value"] and head["steps"] == d["steps"] and head["n_gpus"] == d["n_gpus"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] + 1e-3     # images of ALL ranks / max-over-ranks time
    # the line says what the collectives were: world, backend, one all_reduce per finished gradient range, all bytes of the flat gradient
    dd = d["distributed"]
    assert dd["world"] == 2 and dd["backend"] == "gloo" and dd["collectives_per_step"] >= dd["segments"] >= 10
    assert dd["bytes_per_step"] in (4 * d["config"]["params"], 4 * (d["config"]["params"] - 1))   # every parameter exactly once (+- the pad float)
    # whole-job sampling figures ride on the same line: rows sharded over the two ranks, no collective
    for kind, n, evals in (("ddim50", 4, 50), ("ddpm1000", 2, 1000)):
        sres = d["sampling"][kind]
        assert sres["samples_per_gpu"] == n and sres["unet_evaluations"] == evals and sres["images_finite"]
        assert abs(sres["value"] - 2 * n / sres["seconds_per_loop"]) < 1e-6 * sres["value"]
        assert 0 < sres["roofline"]["frac"] < 1
    assert d["sustained"]["steps"] >= 2 and d["ms_per_step_median"] > 0
    # round 5: what a first multi-GPU lease must yield without a code change -- the transport that was agreed on, per-bucket times, exposed
    # communication (the step with and without the collective calls), the bucket-size and weight-gradient-slot sweeps, the sharded DDIM job
    assert dd["rccl_ranks"] is None and "c10d" in dd["gradient_transport"]           # gloo ranks sharing one GPU: RCCL is never asked for
    me = dd["measured"]
    assert "error" not in me, me
    assert me["ms_per_step"] > 0 and me["ms_per_step_without_collectives"] > 0 and np.isfinite(me["exposed_comm_ms"])
    assert len(me["buckets_timed"]) == dd["buckets"] and sum(b["bytes"] for b in me["buckets_timed"]) == dd["bytes_per_step"]
    assert [p["bucket_mb"] for p in me["bucket_sweep"]] == [0.0, 32.0] and me["bucket_sweep"][0]["buckets"] == dd["segments"]
    assert [p["ps_wg3_slots"] for p in me["wgrad3_slot_sweep"]] == [128, 192] and all(p["ms_per_step"] > 0 for p in me["wgrad3_slot_sweep"])
    sh = d["sampling"]["ddim50_sharded"]
    assert sh["scaling"] == "strong" and sh["global_samples"] == 16 and sh["samples_per_gpu"] == 8 and sh["images_finite"]
    assert np.isfinite(d["final_loss"])


def test_bench_eight_ranks_end_to_end(gpu, tmp_path):
    """VERDICT round 5, task 7 / weak item 10: everything multi-rank had only ever run at world 2.  The driver's 8-GPU command (`bench.py --gpus 8`
    under torch.distributed.run) with 8 gloo ranks sharing this box's one GPU at a reduced batch: the 8-rank bucket plan and sweep, the
    weight-gradient slot sweep, exposed-communication measurement with its sync_state(), per-bucket timing, the sharded DDIM-50 job of
    configs[4] at its real size (2048 chains = 256 rows per rank) and the bounded headline all execute once before real hardware does."""
    detail = str(tmp_path / "bench_detail8.json")
    r = _torchrun(["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--sampling-n", "8,1",
                   "--no-celeba", "--no-fid", "--sustain", "0.05", "--dp-sweep-buckets", "0,32", "--dp-sweep-slots", "128,192", "--total", "2048",
                   "--detail-file", detail], {"OMP_NUM_THREADS": "1"}, timeout=1500, nproc=8)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192
    head = json.loads(lines[0])
    assert head["n_gpus"] == 8 and head["config"]["global_batch"] == 16 and head["scaling"] == "weak" and head["value"] > 0
    assert abs(head["value"] - 8 * 2 / (head["ms_per_step"] * 1e-3)) < 1e-6 * head["value"] + 1e-3
    d = json.load(open(detail))
    dd = d["distributed"]
    assert dd["world"] == 8 and dd["backend"] == "gloo" and dd["bytes_per_step"] in (4 * d["config"]["params"], 4 * (d["config"]["params"] - 1))
    me = dd["measured"]
    assert "error" not in me, me
    assert me["ms_per_step"] > 0 and me["ms_per_step_without_collectives"] > 0 and np.isfinite(me["exposed_comm_ms"])
    assert len(me["buckets_timed"]) == dd["buckets"] and [p["bucket_mb"] for p in me["bucket_sweep"]] == [0.0, 32.0]
    assert [p["ps_wg3_slots"] for p in me["wgrad3_slot_sweep"]] == [128, 192] and all(p["ms_per_step"] > 0 for p in me["wgrad3_slot_sweep"])
    sh = d["sampling"]["ddim50_sharded"]
    assert sh["scaling"] == "strong" and sh["global_samples"] == 2048 and sh["samples_per_gpu"] == 256 and sh["images_finite"]
    for kind, n in (("ddim50", 8), ("ddpm1000", 1)):
        assert d["sampling"][kind]["samples_per_gpu"] == n and d["sampling"][kind]["images_finite"]
    assert np.isfinite(d["final_loss"])          # the replicas stayed one model through sync_state() and the all-reduced updates


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the round-end boxes have one")
def test_bench_two_gpus_rccl(gpu):
    """the same command on two real GPUs with backend nccl (= RCCL): ranks see world 2, the line reports the RCCL collectives, and the
    replicas stay in step (finite loss after the all-reduced updates).  Skipped on 1-GPU boxes."""
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-sampling", "--no-celeba",
                   "--sustain", "0"], {}, timeout=900, backend="nccl")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["distributed"]["backend"] == "nccl" and d["distributed"]["world"] == 2
    assert np.isfinite(d["final_loss"]) and d["config"]["global_batch"] == 256


def test_cli_train_loop_two_ranks(gpu, tmp_path):
    """python baddiffusion.py --mode train on 2 ranks (gloo): 8 synthetic images, global batch 4 -> 2 steps; rank 0 writes
    the diffusers-layout checkpoint, the run records that it was not started from pretrained weights."""
    out = str(tmp_path / "res")
    r = _torchrun(["baddiffusion.py", "--mode", "train", "--dataset", "CIFAR10", "--batch", "2", "--epoch", "1", "--poison_rate", "0.25",
                   "--trigger", "BOX_14", "--target", "CORNER", "--ckpt", "DDPM-CIFAR10-32", "--result", out, "-o",
                   "--save_image_epochs", "100"],
                  {"BD_NUM_IMAGES": "8", "BD_ALLOW_RANDOM_INIT": "1"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    runs = [os.path.join(out, d) for d in os.listdir(out)]
    assert len(runs) == 1
    files = os.listdir(runs[0])
    assert "unet" in files and "scheduler" in files and "model_index.json" in files and "config.json" in files
    cfgj = json.load(open(os.path.join(runs[0], "config.json")))
    assert cfgj["pretrained"] is False
    log = [json.loads(ln) for ln in open(os.path.join(runs[0], "log.jsonl"))]
    assert log and np.isfinite(log[0]["loss"])


def test_checkpoint_resume_equals_straight_run(gpu, tmp_path):
    """--mode resume (baddiffusion.py:336-342, 558-570): 2 steps -> checkpoint() -> a NEW model / engine restored from the
    written directory (diffusers-layout weights, optimizer.bin, data.ckpt incl. RNG state) -> 1 step  ==  3 straight steps,
    bit for bit (weights, Adam moments, step counters)."""
    import baddiffusion as cli
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.model import DiffuserModelSched
    from baddiffusion_amd.pipelines import DDPMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = C.SMALL_CFGS["small"]
    B, S = 4, 16
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S).cuda(); tgt = bd.get_target("CORNER", trig.cpu()).cuda()
    data = _u8_batch(16, S, 51).cuda()
    pois = torch.tensor([1, 0, 0, 0], dtype=torch.bool).cuda()
    kw = dict(lr=1e-3, lr_warmup_steps=2, num_training_steps=10)

    def one_step(engine):      # noise / timesteps from the GLOBAL generators, like the CLI's loop body (:596, :600)
        rows = torch.randint(0, 16, (B,)).cuda()
        eps = torch.randn(B, 3, S, S, device="cuda"); t = torch.randint(0, 1000, (B,), device="cuda")
        return engine.train_step(data, pois, trig, tgt, eps, t, row_index=rows)

    config = cli.TrainingConfig()
    config.output_dir = str(tmp_path / "run"); os.makedirs(config.output_dir)
    config.ckpt_path = os.path.join(config.output_dir, config.ckpt_dir)
    config.data_ckpt_path = os.path.join(config.output_dir, config.data_ckpt_dir)
    config.is_save_all_model_epochs = False
    torch.manual_seed(77)
    m = make_model(cfg, 7, gpu); sched = DDPMScheduler(clip_sample=False)
    e = TrainEngine(m, sched, **kw)
    one_step(e); one_step(e)
    cli.checkpoint(config, e, DDPMPipeline(m, sched), 0, 2)
    one_step(e)
    torch.cuda.synchronize()
    # the resumed process
    torch.manual_seed(12345)      # whatever state a fresh process has: restore_training_state must overwrite it
    m2, sched2, _ = DiffuserModelSched.get_trained(config.output_dir, clip_sample=None)
    m2 = m2.to(gpu)
    e2 = TrainEngine(m2, sched2, **kw)
    assert cli.restore_training_state(config, e2) == (0, 2) and e2.opt_step == 2
    one_step(e2)
    assert torch.equal(m2.flat, m.flat) and torch.equal(e2.m, e.m) and torch.equal(e2.v, e.v) and e2.opt_step == e.opt_step == 3


def test_gn_bwd_deferred_param_fold_equals_per_layer(gpu):
    """bd_gn_bwd_desc.param_partials + one bd_gn_bwd_params launch for several layers == the per-layer parameter
    reduction of bd_gn_bwd (same fixed summation order, so bit-identical); dx is untouched by the option."""
    import ctypes as CT
    from baddiffusion_amd import _lib as L, ops
    lib = L.load()
    torch.manual_seed(5)
    B, G = 6, 32
    layers = [(64, 64), (256, 128), (16, 256)]           # (HW, C): all take the single-pass kernel at this batch
    items = (L.GnParamItem * len(layers))()
    keep, want = [], []
    for i, (HW, Cc) in enumerate(layers):
        assert lib.bd_gn_bwd_defers(B, HW, Cc, G) == 1
        x = torch.randn(B, HW, Cc, device=gpu); dy = torch.randn(B, HW, Cc, device=gpu)
        ga = torch.randn(Cc, device=gpu); be = torch.randn(Cc, device=gpu)
        st = torch.empty(2, B, G, device=gpu)
        ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
        y = torch.empty_like(x)
        f = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=G, eps=1e-6, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), y=L.ptr(y), ldy=Cc,
                        mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel())
        L.check(lib.bd_gn_fwd(CT.byref(f), L.stream()))
        outs = []
        for deferred in (False, True):
            dx = torch.empty_like(x); dg = torch.full((Cc,), float("nan"), device=gpu); db = torch.full((Cc,), float("nan"), device=gpu)
            part = torch.empty(B, 2, Cc, device=gpu)
            d = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=G, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), mean=L.ptr(st[0]),
                            rstd=L.ptr(st[1]), dy=L.ptr(dy), lddy=Cc, dx=L.ptr(dx), lddx=Cc, accumulate_dx=0, dgamma=L.ptr(dg),
                            dbeta=L.ptr(db), workspace=L.ptr(ws), workspace_bytes=ws.numel(), param_partials=L.ptr(part) if deferred else None)
            L.check(lib.bd_gn_bwd(CT.byref(d), L.stream()))
            outs.append((dx, dg, db, part))
        (dx0, dg0, db0, _), (dx1, dg1, db1, part1) = outs
        assert torch.equal(dx0, dx1)
        assert torch.isnan(dg1).all() and torch.isnan(db1).all()          # deferred: not written by the layer
        items[i].partials = L.ptr(part1); items[i].C = Cc; items[i].dgamma = L.ptr(dg1); items[i].dbeta = L.ptr(db1)
        keep.append((x, dy, ga, be, st, ws, part1, dx1)); want.append((dg0, db0, dg1, db1))
    L.check(lib.bd_gn_bwd_params(items, len(layers), B, L.stream()))
    torch.cuda.synchronize()
    for dg0, db0, dg1, db1 in want:
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert lib.bd_gn_bwd_defers(4, 65536, 128, 32) == 1                  # round 4: the large-image (split) path defers too


def test_gn_bwd_dx_add_equals_separate_add(gpu):
    """bd_gn_bwd_desc.dx_add: dx = (term + existing dx) + dx_add in one store == the same launch followed by an add
    (bit-identical: same order of additions), on the single-pass and on the split (large image) kernels."""
    import ctypes as CT
    from baddiffusion_amd import _lib as L, ops
    lib = L.load()
    torch.manual_seed(9)
    for (B, HW, Cc) in [(6, 256, 128), (3, 1024, 96), (2, 16384, 64)]:
        G = 32
        x = torch.randn(B, HW, Cc, device=gpu); dy = torch.randn(B, HW, Cc, device=gpu); extra = torch.randn(B, HW, Cc + 32, device=gpu)
        ga = torch.randn(Cc, device=gpu); be = torch.randn(Cc, device=gpu); st = torch.empty(2, B, G, device=gpu)
        ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
        y = torch.empty_like(x)
        f = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=G, eps=1e-6, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), y=L.ptr(y), ldy=Cc,
                        mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel())
        L.check(lib.bd_gn_fwd(CT.byref(f), L.stream()))
        prev = torch.randn(B, HW, Cc, device=gpu)
        res = []
        for fused in (False, True):
            dx = prev.clone(); dg = torch.empty(Cc, device=gpu); db = torch.empty(Cc, device=gpu)
            d = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=G, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), mean=L.ptr(st[0]),
                            rstd=L.ptr(st[1]), dy=L.ptr(dy), lddy=Cc, dx=L.ptr(dx), lddx=Cc, accumulate_dx=1, dgamma=L.ptr(dg),
                            dbeta=L.ptr(db), workspace=L.ptr(ws), workspace_bytes=ws.numel(),
                            dx_add=L.ptr(extra) if fused else None, ld_add=Cc + 32)
            L.check(lib.bd_gn_bwd(CT.byref(d), L.stream()))
            if not fused:
                dx = dx + extra[:, :, :Cc]
            res.append((dx, dg, db))
        assert torch.equal(res[0][0], res[1][0]), (B, HW, Cc)
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_ssim_and_mse_kernels_match_oracle(gpu):
    """bd_ssim / the l2 kernel behind metrics.mse against oracle.metrics_ref (fp64, scipy.ndimage -- an independent statement of
    the torchmetrics defaults, not the product's code): CIFAR-size batches, a ragged non-square size (tiles of 32 with
    remainders), NHWC-strided views, identical images (== 1) and a measure-style comparison against a constant target.
    Tolerance 2e-5 absolute on SSIM (fp32 local moments: E[x^2] - mu^2 cancels against c2 = 9e-4), 1e-6 relative on MSE."""
    from baddiffusion_amd import metrics, ops
    from oracle import metrics_ref
    torch.manual_seed(11)
    for shape in [(8, 3, 32, 32), (2, 3, 75, 44), (3, 1, 12, 64), (2, 3, 256, 256)]:
        a = torch.rand(shape); b = (a + 0.2 * torch.randn(shape)).clamp(0, 1)
        want = metrics_ref.ssim_ref(a.numpy(), b.numpy())
        got = metrics.ssim(a.to(gpu), b.to(gpu))
        assert abs(got - want) < 2e-5, (shape, got, want)
        # NHWC storage viewed as NCHW (what the pipelines hand over)
        an = a.permute(0, 2, 3, 1).contiguous().to(gpu).permute(0, 3, 1, 2); bn = b.permute(0, 2, 3, 1).contiguous().to(gpu).permute(0, 3, 1, 2)
        got2 = float(ops.ssim(an, bn))
        assert abs(got2 - want) < 2e-5, (shape, got2, want)
        m_want = metrics_ref.mse_ref(a.numpy(), b.numpy())
        assert abs(metrics.mse(a.to(gpu), b.to(gpu)) - m_want) < 1e-6 * m_want
    a = torch.rand(4, 3, 32, 32, device=gpu)
    assert abs(metrics.ssim(a, a.clone()) - 1.0) < 1e-6
    tgt = torch.rand(1, 3, 32, 32).expand(4, 3, 32, 32).contiguous()
    assert abs(metrics.ssim(a, tgt.to(gpu)) - metrics_ref.ssim_ref(a.cpu().numpy(), tgt.numpy())) < 2e-5
    with pytest.raises(RuntimeError):
        metrics.ssim(a.cpu(), a.cpu())


def test_full_batch_train_steps_bitwise_reproducible(gpu):
    """Run-to-run determinism of the product step at BASELINE configs[1] size (B = 128, two-stream schedule, K-split slabs
    folded in fixed order, no atomics anywhere): two independent engines fed the same three batches end with bit-identical
    weights, Adam moments, losses and gradient norms."""
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = U.CIFAR10_32
    B, S = 128, 32
    bd = Backdoor(root=None)
    trig = bd.get_trigger("BOX_14", 3, S).cuda(); tgt = bd.get_target("CORNER", trig.cpu()).cuda()
    pois = (torch.arange(B) % 10 == 0).cuda()
    batches = [(_u8_batch(B, S, 40 + i).cuda(), torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(50 + i)).cuda(),
                torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(60 + i)).cuda()) for i in range(3)]
    runs = []
    for _ in range(2):
        m = make_model(cfg, 0, gpu)
        e = TrainEngine(m, DDPMScheduler(), lr=2e-4)
        hist = []
        for u8, eps, t in batches:
            loss = e.train_step(u8, pois, trig, tgt, eps, t)
            hist.append((loss.clone(), e.grad_norm.clone()))
        torch.cuda.synchronize()
        runs.append((m.flat.detach().clone(), e.m.clone(), e.v.clone(), hist))
    (w0, m0, v0, h0), (w1, m1, v1, h1) = runs
    assert torch.equal(w0, w1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(h0, h1))


def test_pndm_scheduler_and_pipeline_vs_reference(gpu, golden):
    """SURVEY f-4, G9: the product's PNDMScheduler (bd_lincomb updates) against the reference's chains -- timesteps, the 12
    Runge-Kutta warm-up evaluations + Adams-Bashforth tail for 50 / 20 / 7 steps, the skip_prk_steps start-up branches --
    and PNDMPipeline (built from a DPM-Solver config carrier, as model.py:598-630 does) on the small UNet with and without
    the post-step clip, vs the images the reference's PNDMPipeline produced."""
    from baddiffusion_amd import ops
    from baddiffusion_amd.model import DiffuserModelSched
    from baddiffusion_amd.pipelines import PNDMPipeline
    from baddiffusion_amd.schedulers import PNDMScheduler, SchedulerConfigCarrier
    g = golden("pndm")
    x0 = C.pndm_init()
    for n in C.PNDM_STEPS:
        s = PNDMScheduler()
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
        x = x0.to(gpu)
        for i, t in enumerate(s.timesteps):
            x = s.step(C.pndm_fake_model(x, t), t, x).prev_sample
            ref = torch.from_numpy(g[f"chain_{n}"][i])
            assert float((x.cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()), (n, i)
    s = PNDMScheduler(skip_prk_steps=True)
    s.set_timesteps(10)
    assert np.array_equal(s.timesteps.numpy(), g["skip_timesteps_10"])
    x = x0.to(gpu)
    for i, t in enumerate(s.timesteps):
        x = s.step(C.pndm_fake_model(x, t), t, x).prev_sample
        ref = torch.from_numpy(g["skip_chain_10"][i])
        assert float((x.cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()), i
    # the fused clamp == clamping afterwards
    a, b = torch.randn(2, 3, 8, 8, device=gpu) * 2, torch.randn(2, 3, 8, 8, device=gpu)
    assert torch.equal(ops.lincomb([a, b], [0.7, -1.3], clip=1.0), ops.lincomb([a, b], [0.7, -1.3]).clamp(-1, 1))
    odd = torch.randn(7, device=gpu)                                   # n % 4 != 0 tail
    assert torch.allclose(ops.lincomb([odd, odd], [2.0, 0.5]), 2.5 * odd, rtol=1e-6, atol=1e-6)
    # pipeline on the small UNet
    cfg = C.SMALL_CFGS["small"]
    m = make_model(cfg, 7, gpu)
    init = C.pipeline_init(cfg)
    carrier = SchedulerConfigCarrier("DPMSolverMultistepScheduler", num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02)
    for clip in (True, False):
        pipe = PNDMPipeline(m, carrier, clip_sample=clip)
        r = pipe(batch_size=init.shape[0], init=init, output_type=None, num_inference_steps=6)
        ref = g[f"pipe6_{int(clip)}"]
        assert r.images.shape == ref.shape
        assert float(np.abs(r.images - ref).max()) < 2e-3, (clip, float(np.abs(r.images - ref).max()))
    assert set(DiffuserModelSched._PNDM_SCHEDS) >= {"DPM_SOLVER_PP_O2-SCHED", "UNIPC-SCHED", "PNDM-SCHED", "HEUN-SCHED"}


def test_celeba256_glasses_to_cat_poisoned_step(gpu, golden):
    """BASELINE configs[3] per-GPU share with ITS backdoor: 256x256, batch 4, GLASSES trigger -> CAT target (the
    reference's own Backdoor outputs on its static assets, tests/golden/img_triggers.npz), one row poisoned.  The fused
    uint8 entry point's (R, x0) against the oracle's collated batch (dataset.py:288-315: int mask, blend), and the whole
    fused train step == the collated calling convention on the 256x256 network (finite loss, identical weights)."""
    from baddiffusion_amd import ops
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    g = golden("img_triggers")
    trig = torch.from_numpy(g["trigger_GLASSES_c3_s256"]); tgt = torch.from_numpy(g["target_CAT_c3_s256"])
    assert trig.shape == (3, 256, 256) and float(trig.min()) == -1.0 and float(trig.max()) <= 1.0
    B, S = 4, 256
    u8 = _u8_batch(B, S, 31)
    pois = torch.tensor([False, True, False, False])
    eps = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(32))
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(33))
    m1, m2 = make_model(U.CELEBA_HQ_256, 5, gpu), make_model(U.CELEBA_HQ_256, 5, gpu)
    e1, e2 = TrainEngine(m1, DDPMScheduler(), lr=6e-5), TrainEngine(m2, DDPMScheduler(), lr=6e-5)
    # poison + q_sample vs the oracle's collated batch
    _, _, R, x0 = ops.poison_qsample(u8.cuda(), pois.cuda(), trig.cuda(), tgt.cuda(), eps.cuda(), t.cuda(), e1.alphas, e1.alphas_cumprod,
                                     want_batch=True)
    x = torch.stack([BD.image_u8_to_float(u8[i]) for i in range(B)])
    Rr, x0r = BD.make_batch(x, pois, trig, tgt)
    to_nchw = lambda v: v.permute(0, 3, 1, 2) if v.shape[-1] == 3 else v
    assert float((to_nchw(R).cpu() - Rr).abs().max()) <= 1e-6 and float((to_nchw(x0).cpu() - x0r).abs().max()) <= 1e-6
    assert float(Rr[1].abs().max()) > 0 and float(Rr[0].abs().max()) == 0      # only the poisoned row carries a residual
    l1 = e1.train_step(u8.cuda(), pois.cuda(), trig.cuda(), tgt.cuda(), eps.cuda(), t.cuda())
    l2 = e2.train_step_batch(x0, R, eps.cuda(), t.cuda())
    assert torch.isfinite(l1) and float(l1) > 0 and abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2))
    assert relerr(m1.flat, m2.flat) < 1e-6 and torch.isfinite(m1.flat).all()


def test_cli_sampling_and_measure_end_to_end(gpu, tmp_path):
    """baddiffusion.py's sampling() and measure() (reference baddiffusion.py:366-419, 477-551) run on the GPU end to end on the
    small UNet: the 4x4 grids (final + t0, clean + backdoor), measure()'s PNG sets (n = 32, DDIM so the chains are
    deterministic), score.json with MSE / SSIM (+ the `_noclip` key form), a null FID that carries its reason, MSE / SSIM
    recomputed by the oracle from the written PNGs, and the sharded (world = 2) run writing the same file set bit for bit."""
    import dataclasses
    from PIL import Image
    import baddiffusion as cli
    from baddiffusion_amd.dataset import DatasetLoader
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDIMScheduler
    from baddiffusion_amd.unet import unet_from_config
    from oracle import metrics_ref
    cfg_net = dataclasses.replace(C.SMALL_CFGS["small"], sample_size=32)
    model = unet_from_config(cfg_net).cuda()
    model.load_state_dict(U.gen_params(cfg_net, 7))
    dsl = DatasetLoader(root=None, name=DatasetLoader.CIFAR10, batch_size=8, seed=0, device=gpu, num_images=8)
    dsl.set_poison(trigger_type="BOX_14", target_type="CORNER", clean_rate=1.0, poison_rate=0.25).prepare_dataset(mode="FIXED")

    def run(out, world):
        config = cli.TrainingConfig()
        config.output_dir = str(out); config.seed = 0; config.clip = False; config.sample_ep = None
        config.eval_sample_n = 16; config.measure_sample_n = 32; config.eval_max_batch = 12      # 32 = 12 + 12 + 8: ragged chunks
        os.makedirs(config.output_dir, exist_ok=True)
        pipe = DDIMPipeline(model, DDIMScheduler(num_train_timesteps=1000, clip_sample=False))
        if world == 1:
            cli.sampling(config, "final", pipe, dsl)
            return config, cli.measure(config, dsl, "measure", pipe, rank=0, world=1)
        # sharded: every "rank" writes its own contiguous PNG range; scoring is rank 0's job (no process group: barrier skipped)
        from baddiffusion_amd.model import batch_sampling_save
        s = pipe.unet.sample_size
        noise = torch.randn((config.measure_sample_n, 3, s, s), generator=torch.manual_seed(config.seed))
        for folder, init in (("clean_noclip", noise), ("backdoor_noclip", noise + dsl.trigger.unsqueeze(0))):
            for r in range(world):
                batch_sampling_save(config.measure_sample_n, pipe, os.path.join(config.output_dir, "measure", folder), init=init,
                                    max_batch_n=config.eval_max_batch, rng=torch.Generator().manual_seed(r), rank=r, world=world)
        return config, None

    config, score = run(tmp_path / "one", 1)
    for folder in ("samples", "backdoor_samples"):
        for name in ("final_noclip.png", "final_noclip_sample_t0.png"):
            im = Image.open(os.path.join(config.output_dir, folder, name))
            assert im.size == (4 * 32, 4 * 32)
    for folder in ("clean_noclip", "backdoor_noclip"):
        files = sorted(os.listdir(os.path.join(config.output_dir, "measure", folder)), key=lambda n: int(n.split(".")[0]))
        assert files == [f"{i}.png" for i in range(32)]
    on_disk = json.load(open(os.path.join(config.output_dir, "score.json")))
    assert on_disk == score
    assert set(score) == {"FID_noclip", "FID_reason_noclip", "MSE_noclip", "SSIM_noclip", "inference_chunk_noclip"}
    assert score["inference_chunk_noclip"] == [12, 8]     # (round 6) every batch size the forwards of this measure() ran with: 32 = 12 + 12 + 8
    assert score["FID_noclip"] is None and "Inception" in score["FID_reason_noclip"]
    bd_dir = os.path.join(config.output_dir, "measure", "backdoor_noclip")
    gen = np.stack([np.asarray(Image.open(os.path.join(bd_dir, f"{i}.png")).convert("RGB"), dtype=np.float64) / 255.0 for i in range(32)])
    gen = gen.transpose(0, 3, 1, 2)
    tgt = np.broadcast_to(np.clip(dsl.target.cpu().numpy().astype(np.float64) / 2 + 0.5, 0, 1)[None], gen.shape)
    want_mse, want_ssim = metrics_ref.mse_ref(gen, tgt), metrics_ref.ssim_ref(gen, tgt)
    assert abs(score["MSE_noclip"] - want_mse) < 1e-6 * want_mse, (score["MSE_noclip"], want_mse)
    assert abs(score["SSIM_noclip"] - want_ssim) < 2e-5, (score["SSIM_noclip"], want_ssim)
    # the images are real samples: not constant, and the backdoor set differs from the clean set
    clean0 = np.asarray(Image.open(os.path.join(config.output_dir, "measure", "clean_noclip", "0.png")))
    assert gen.std() > 1e-3 and not np.array_equal(clean0, np.asarray(Image.open(os.path.join(bd_dir, "0.png"))))
    # sharded == unsharded (DDIM chains are deterministic given init; chunk boundaries differ: 16 = 12 + 4 per rank).  The only
    # batch-size dependence is the K-split of the smallest layers: allow one uint8 level on a few pixels.
    config2, _ = run(tmp_path / "two", 2)
    for folder in ("clean_noclip", "backdoor_noclip"):
        d2 = os.path.join(config2.output_dir, "measure", folder)
        assert sorted(os.listdir(d2)) == sorted(os.listdir(os.path.join(config.output_dir, "measure", folder)))
        for i in range(32):
            a = np.asarray(Image.open(os.path.join(config.output_dir, "measure", folder, f"{i}.png")), dtype=np.int16)
            b = np.asarray(Image.open(os.path.join(d2, f"{i}.png")), dtype=np.int16)
            assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-2, (folder, i)
    # a second measure() with a real FID value drops the reason; a later null keeps the value (update_score_file semantics)
    sc = cli.update_score_file(config, "score.json", 12.5, None, None)
    assert sc["FID_noclip"] == 12.5 and "FID_reason_noclip" not in sc and sc["MSE_noclip"] == score["MSE_noclip"]
    sc = cli.update_score_file(config, "score.json", None, 0.5, None)
    assert sc["FID_noclip"] == 12.5 and sc["MSE_noclip"] == 0.5


@pytest.mark.parametrize("B,H,W,C", [(3, 32, 32, 128), (2, 8, 64, 256), (1, 256, 256, 128), (2, 16, 16, 128), (5, 4, 12, 128)])
def test_thin_convs_direct_kernels(gpu, B, H, W, C):
    """conv_in (3 -> C) and conv_out (C -> 3) forward / data gradient / weight gradient (unet_2d.py:124,217): the direct fp32
    streaming kernels of round 3 (conv_thin.hip: W % 4 == 0 expand, W % 32 == 0 and C == 128 contract / row weight gradients; other
    shapes take the implicit-GEMM path, same checks) against fp64 F.conv2d + autograd on the CPU.  Exact fp32 products (1e-5 relative)
    except the expand direction and the weight gradients in split-bf16 mode, which round 4 moved to the matrix pipe (2e-5); run-to-run bit
    identity of the K-split weight gradients."""
    import torch.nn.functional as F
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(B * 100 + W)
    x3 = torch.randn(B, H, W, 3, generator=g); xc = torch.randn(B, H, W, C, generator=g)
    w_in = torch.randn(C, 3, 3, 3, generator=g) * 0.2; b_in = torch.randn(C, generator=g)          # [Cout][kh][kw][Cin]
    w_out = torch.randn(3, 3, 3, C, generator=g) * 0.05; b_out = torch.randn(3, generator=g)
    dy_c = torch.randn(B, H, W, C, generator=g); dy_3 = torch.randn(B, H, W, 3, generator=g)

    def ref(x, w, b, dy):
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True); wr = w.double().permute(0, 3, 1, 2).requires_grad_(True)
        br = b.double().requires_grad_(True)
        y = F.conv2d(xr, wr, br, padding=1)
        y.backward(dy.double().permute(0, 3, 1, 2))
        return y.permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.permute(0, 2, 3, 1), br.grad
    for mode in (0, 1):
        # conv_in
        y, _, dw, db = ref(x3, w_in, b_in, dy_c)
        got = ops.conv3x3_fwd(x3.cuda(), w_in.cuda(), b_in.cuda(), mode=mode, out_scale=0.5)
        assert relerr(got, 0.5 * y) < (1e-5 if mode == 0 else 2e-5)        # (round 4: split-bf16 MFMA form of the expand direction in mode 1, W % 32 == 0)
        gw, gb = ops.conv3x3_wgrad(x3.cuda(), dy_c.cuda(), mode=mode, with_db=True)
        assert relerr(gw, dw) < (1e-5 if mode == 0 else 2e-5) and relerr(gb, db) < (1e-5 if mode == 0 else 2e-5)
        gw2, gb2 = ops.conv3x3_wgrad(x3.cuda(), dy_c.cuda(), mode=mode, with_db=True)
        assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
        # conv_out
        y, dx, dw, db = ref(xc, w_out, b_out, dy_3)
        got = ops.conv3x3_fwd(xc.cuda(), w_out.cuda(), b_out.cuda(), mode=mode)
        assert relerr(got, y) < (1e-5 if mode == 0 else 2e-5)
        gx = ops.conv3x3_dgrad(dy_3.cuda(), w_out.cuda(), (B, H, W, C), mode=mode)
        assert relerr(gx, dx) < (1e-5 if mode == 0 else 2e-5)
        gw, gb = ops.conv3x3_wgrad(xc.cuda(), dy_3.cuda(), mode=mode, with_db=True)
        assert relerr(gw, dw) < (1e-5 if mode == 0 else 2e-5) and relerr(gb, db) < 1e-5
        gw2, _ = ops.conv3x3_wgrad(xc.cuda(), dy_3.cuda(), mode=mode, with_db=True)
        assert torch.equal(gw, gw2)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 8, 8, 128, 128), (2, 16, 16, 256, 128), (5, 4, 4, 128, 256), (64, 16, 16, 256, 256), (1, 32, 64, 128, 128),
                                             (2, 8, 8, 512, 512), (1, 2, 2, 128, 128)])
def test_upsample_conv_phase_decomposition(gpu, B, H, W, Cin, Cout):
    """Upsample2D = conv3x3(nearest_up2(x)) (resnet.py:126-161) in its phase-decomposed form (conv_ph.hip: four 2x2-tap convolutions on
    the source grid / one 16-tap stride-2 sampled data gradient / 16 tap products folded to 9 for the weight gradient) against
    fp64 F.interpolate + F.conv2d + autograd on the CPU and against the literal split-plane path of round 2 on the upsampled grid.
    3e-5 relative: the split-bf16 product drops lo*lo (~2^-16 per product), and the phase form multiplies pre-summed taps, i.e. rounds
    differently from the literal form (both sit ~1e-5 from the fp64 result)."""
    import torch.nn.functional as F
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin // 128)
    x = torch.randn(B, H, W, Cin, generator=g); dy = torch.randn(B, 2 * H, 2 * W, Cout, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05; bias = torch.randn(Cout, generator=g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True); wr = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    br = bias.double().requires_grad_(True)
    y = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, br, padding=1)
    y.backward(dy.double().permute(0, 3, 1, 2))
    y_ref, dx_ref, dw_ref, db_ref = y.permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.permute(0, 2, 3, 1), br.grad
    xs, dys = ops.split_rows(x.cuda()), ops.split_rows(dy.cuda())
    e, et = ops.upsample_weights(w.cuda())
    got = ops.upsample_conv_fwd(xs, e, B, H, W, Cin, Cout, bias=bias.cuda())
    assert relerr(got, y_ref) < 3e-5
    gx = ops.upsample_conv_dgrad(dys, et, B, H, W, Cin, Cout)
    assert relerr(gx, dx_ref) < 3e-5
    prev = torch.randn(B, H, W, Cin, generator=g).cuda()
    acc = ops.upsample_conv_dgrad(dys, et, B, H, W, Cin, Cout, out=prev.clone(), accumulate=True)
    assert relerr(acc, prev + gx) < 1e-6
    gw, gb = ops.upsample_conv_wgrad(xs, dys, B, H, W, Cin, Cout, with_db=True)
    assert relerr(gw, dw_ref) < 3e-5 and relerr(gb, db_ref) < 3e-5
    gw2, gb2 = ops.upsample_conv_wgrad(xs, dys, B, H, W, Cin, Cout, with_db=True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)                                  # fixed-order K split
    # the literal form on the upsampled grid (round 2)
    xu = ops.split_rows_ups2(x.cuda())
    lit = ops.conv3x3_ps(xu, ops.split_bf16(w.cuda()), B, 2 * H, 2 * W, Cin, Cout, 1, bias=bias.cuda())
    assert relerr(got, lit) < 3e-5
    lit_w = ops.conv3x3_ps_wgrad(xu, dys, B, 2 * H, 2 * W, Cin, Cout)
    assert relerr(gw, lit_w) < 3e-5


@pytest.mark.parametrize("B,Ho,Wo,Cin,Cout,pad", [(3, 8, 8, 128, 128, 0), (2, 4, 16, 256, 128, 1), (64, 16, 16, 128, 128, 0), (4, 8, 8, 128, 256, 1),
                                                  (2, 4, 4, 512, 512, 0), (1, 1, 1, 128, 128, 1)])
def test_stride2_conv_dgrad_phase_decomposition(gpu, B, Ho, Wo, Cin, Cout, pad):
    """data gradient of Downsample2D's stride-2 convolution (resnet.py:199-208; pad 0 = F.pad(0,1,0,1) + padding 0, pad 1 = padding 1)
    as four parity classes with 4 / 2 / 2 / 1 taps on the output grid, against fp64 autograd and the implicit-GEMM path"""
    import torch.nn.functional as F
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Ho + pad)
    H, W = 2 * Ho, 2 * Wo
    x = torch.randn(B, H, W, Cin, generator=g); dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True); wr = w.double().permute(0, 3, 1, 2)
    xin = F.pad(xr, (0, 1, 0, 1)) if pad == 0 else xr
    y = F.conv2d(xin, wr, None, stride=2, padding=pad)
    assert tuple(y.shape[-2:]) == (Ho, Wo)
    y.backward(dy.double().permute(0, 3, 1, 2))
    dx_ref = xr.grad.permute(0, 2, 3, 1)
    dys, wt = ops.split_rows(dy.cuda()), ops.split_wT(w.cuda())
    gx = ops.conv3x3_s2_dgrad_ps(dys, wt, B, Ho, Wo, Cin, Cout, pad=pad)
    assert relerr(gx, dx_ref) < 3e-5
    prev = torch.randn(B, H, W, Cin, generator=g).cuda()
    acc = ops.conv3x3_s2_dgrad_ps(dys, wt, B, Ho, Wo, Cin, Cout, pad=pad, out=prev.clone(), accumulate=True)
    assert relerr(acc, prev + gx) < 1e-6
    old = ops.conv3x3_dgrad(dy.cuda(), w.cuda(), (B, H, W, Cin), stride=2, pad=pad, asym=(pad == 0), mode=1)
    assert relerr(gx, old) < 3e-5


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K,batch", [(256, 128, 256, 3), (128, 384, 96, 1), (384, 256, 4096, 1)])
def test_gemm_on_split_planes(gpu, akm, bkm, M, N, K, batch):
    """bd_gemm_sp (the attention block's Linear layers and batched products, attention.py:85-186) in its four operand forms --
    K-contiguous / K-major A and B -- with fp32 and split-plane results, bias, residual, out_scale, accumulate; long K takes the
    split-K path (fixed-order second pass), where the column sums of A (a Linear layer's bias gradient) ride along."""
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + K + batch + 2 * akm + bkm)
    A = torch.randn(batch, M, K, generator=g).cuda(); Bm = torch.randn(batch, N, K, generator=g).cuda() * 0.1
    a_s = ops.split_rows((A.transpose(1, 2) if akm else A).contiguous())
    b_s = ops.split_rows((Bm.transpose(1, 2) if bkm else Bm).contiguous())
    bias = torch.randn(N, generator=g).cuda(); res = torch.randn(batch, M, N, generator=g).cuda()
    ref = 0.7 * (0.25 * A.double() @ Bm.double().transpose(1, 2) + bias.double() + res.double())
    c, cs = ops.gemm_sp(a_s, b_s, M, N, K, a_kmajor=akm, b_kmajor=bkm, batch=batch, bias=bias, residual=res, alpha=0.25, out_scale=0.7,
                        want_split=True)
    assert relerr(c, ref) < 2e-5
    assert relerr(ops.unsplit_rows(cs), ref) < 2e-5
    assert torch.equal(ops.unsplit_rows(cs), ops.unsplit_rows(ops.split_rows(c)).view(batch, M, N))   # the planes ARE the split of the fp32 result
    prev = torch.randn(batch, M, N, generator=g).cuda()
    acc, _ = ops.gemm_sp(a_s, b_s, M, N, K, a_kmajor=akm, b_kmajor=bkm, batch=batch, out=prev.clone(), accumulate=True)
    assert relerr(acc, prev.double() + A.double() @ Bm.double().transpose(1, 2)) < 2e-5
    if akm and bkm and batch == 1:
        c2, _, colsum = ops.gemm_sp(a_s, b_s, M, N, K, a_kmajor=True, b_kmajor=True, want_colsum=True)
        assert relerr(c2, A.double() @ Bm.double().transpose(1, 2)) < 2e-5
        assert relerr(colsum, A[0].double().sum(1)) < 2e-5
    # shapes outside the kernel's tiling are refused, not mangled
    with pytest.raises(RuntimeError):
        ops.gemm_sp(a_s, b_s, M - 32, N, K, a_kmajor=akm, b_kmajor=bkm, batch=batch)


@pytest.mark.parametrize("B,heads", [(1, 1), (3, 1), (2, 2)])
def test_attention_core_on_split_planes(gpu, B, heads):
    """bd_attn_sp_fwd / bd_attn_sp_bwd: softmax(scale q k^T) v and its backward (attention.py:148-162) without an [N, N] fp32 matrix,
    against fp64; P^T / dS^T planes are what the two backward kernels hand to each other.  N = 256 tokens, head dim 256."""
    from baddiffusion_amd import ops
    N, dh = 256, 256
    Cc = heads * dh
    g = torch.Generator().manual_seed(10 * B + heads)
    qkv = torch.randn(B * N, 3 * Cc, generator=g).cuda(); do = torch.randn(B * N, Cc, generator=g).cuda()
    scale = dh ** -0.5
    qs, dos = ops.split_rows(qkv), ops.split_rows(do)
    o_s, pt_s = ops.attn_sp_fwd(qs, B, heads, scale)
    o_inf, none = ops.attn_sp_fwd(qs, B, heads, scale, want_pt=False)
    assert none is None and relerr(ops.unsplit_rows(o_inf), ops.unsplit_rows(o_s)) < 1e-5   # the inference form (no P^T) is its own instantiation
    o_again, pt_again = ops.attn_sp_fwd(qs, B, heads, scale)
    assert torch.equal(o_s, o_again) and torch.equal(pt_s, pt_again)                          # run-to-run: bit-identical
    dqkv_s, dst_s = ops.attn_sp_bwd(qs, pt_s, dos, B, heads, scale)
    D = lambda t: t.double()
    def heads_of(x):   # [B*N, heads*dh] -> [B*heads, N, dh]
        return D(x).view(B, N, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, N, dh)
    q, k, v = (heads_of(qkv[:, i * Cc:(i + 1) * Cc]) for i in range(3))
    P = torch.softmax(scale * q @ k.transpose(1, 2), -1)
    O = P @ v
    dO = heads_of(do)
    dP = dO @ v.transpose(1, 2)
    dS = P * (dP - (P * dP).sum(-1, keepdim=True))
    unheads = lambda x: x.view(B, heads, N, dh).permute(0, 2, 1, 3).reshape(B * N, Cc)
    # norm-relative tolerances: 5e-5 forward, 1e-4 for the gradients (dS = P o (dP - delta) cancels; the task's bound is 1e-3)
    assert relerr(ops.unsplit_rows(o_s), unheads(O)) < 5e-5
    assert relerr(ops.unsplit_rows(pt_s), P.transpose(1, 2)) < 5e-5
    assert relerr(ops.unsplit_rows(dst_s), scale * dS.transpose(1, 2)) < 1e-4
    dqkv = ops.unsplit_rows(dqkv_s)
    assert relerr(dqkv[:, :Cc], unheads(scale * dS @ k)) < 1e-4
    assert relerr(dqkv[:, Cc:2 * Cc], unheads(scale * dS.transpose(1, 2) @ q)) < 1e-4
    assert relerr(dqkv[:, 2 * Cc:], unheads(P.transpose(1, 2) @ dO)) < 1e-4
    # other sequence lengths / head sizes keep the unfused path: the entry refuses them
    with pytest.raises(RuntimeError):
        ops.attn_sp_fwd(ops.split_rows(torch.randn(2 * 64, 3 * 256).cuda()), 2, 1, scale)
