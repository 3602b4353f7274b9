"""GPU parity: every C-ABI op of libbd_hip.so against the CPU oracle / plain fp32 PyTorch-CPU ops on the
same seeded inputs (run with -m gpu on an MI355X).  Integer masks bit-exact; fp32 within the stated tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BD
from oracle import loss_ref, sched_ref, train_ref
from oracle import unet_ref as U
from tests.golden import cases as C


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from baddiffusion_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def close(a, b, rtol, atol):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def R(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------------------------------ a-1 / a-2
@pytest.mark.parametrize("S,trig,tgt", [(32, "BOX_14", "CORNER"), (32, "BOX_8", "SHIFT"), (64, "BOX_18", "TRIGGER")])
def test_poison_qsample(ops, S, trig, tgt, golden):
    B = 6
    _, a, ac = sched_ref.make_tables()
    u8 = torch.randint(0, 256, (B, S, S, 3), generator=torch.Generator().manual_seed(0), dtype=torch.uint8)
    img = torch.stack([BD.image_u8_to_float(u) for u in u8])
    g = BD.get_trigger(trig, 3, S); y = BD.get_target(tgt, g)
    pois = torch.tensor([True, False, True, False, False, True])
    eps = R(1, B, 3, S, S); t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(2))
    Rr, x0 = BD.make_batch(img, pois, g, y)
    xn_ref, tg_ref = loss_ref.q_sample(a, ac, x0, Rr, t, eps)
    for images in (dev(img), dev(u8)):
        xn, tg, Rg, x0g, mask = ops.poison_qsample(images, dev(pois), dev(g), dev(y), dev(eps), dev(t), dev(a), dev(ac),
                                                  want_batch=True, want_mask=True)
        assert torch.equal(mask.cpu(), BD.get_mask(g))                                  # bit-exact int64 mask
        if S == 32 and trig in ("BOX_14", "BOX_8"):
            assert np.array_equal(mask.cpu().numpy(), golden("backdoor")[f"mask_{trig}_32"])
        close(Rg, Rr, 0, 1e-6); close(x0g, x0, 0, 1e-6)
        close(xn.permute(0, 3, 1, 2), xn_ref, 1e-6, 1e-6)
        # rho_t = (1 - sqrt(alpha_t)) * s / (1 - alpha_t) (loss.py:270) cancels catastrophically, so its last
        # ~2 decimal digits depend on how the HOST's torch build rounds pow(x, 0.5) (observed: 7e-6 relative
        # between two x86 hosts).  The kernel evaluates it with correctly-rounded IEEE sqrt/div (bit-equal to the
        # golden vectors, test_qsample_golden / test_rho_ieee); against a live CPU oracle allow 1e-5.
        close(tg.permute(0, 3, 1, 2), tg_ref, 1e-5, 1e-5)


def test_qsample_golden(ops, golden):
    g = golden("qsample")
    _, a, ac = sched_ref.make_tables()
    x0, Rr, eps, t = C.qsample_inputs()
    xn, tg = ops.qsample(dev(x0), dev(Rr), dev(eps), dev(t), dev(a), dev(ac))
    close(xn.permute(0, 3, 1, 2), g["x_noisy"], 1e-6, 1e-7)
    close(tg.permute(0, 3, 1, 2), g["target"], 1e-6, 1e-7)


def test_rho_ieee(ops):
    # target = rho_t * 1 + 0 for every t: bit-equal to the correctly-rounded fp32 evaluation of loss.py:268-270
    _, a, ac = sched_ref.make_tables()
    t = torch.arange(1000)
    one = torch.ones(1000, 1, 1, 1); zero = torch.zeros(1000, 1, 1, 1)
    xn, tg = ops.qsample(dev(zero), dev(one), dev(zero), dev(t), dev(a), dev(ac))
    an, acn, f = a.numpy(), ac.numpy(), np.float32
    rho = (f(1) - np.sqrt(an)) * np.sqrt(f(1) - acn) / (f(1) - an)
    assert np.array_equal(tg.flatten().cpu().numpy(), rho)
    assert np.array_equal(xn.flatten().cpu().numpy(), f(1) - np.sqrt(acn))


def test_layout_roundtrip(ops):
    x = R(3, 5, 3, 16, 16)
    y = ops.nchw_to_nhwc(dev(x))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), x)


# ------------------------------------------------------------------------------------------ a-5 / a-6 / a-7
def test_ddpm_ddim_steps_golden(ops, golden):
    g = golden("sched")
    _, _, ac = sched_ref.make_tables()
    x, eps, z = C.sched_inputs()
    for vt in ("fixed_small", "fixed_large"):
        for clip in (True, False):
            for t in C.DDPM_TS:
                prev, x0 = ops.ddpm_step(dev(eps), dev(x), dev(z), dev(ac), t, t - 1, vt, clip, want_x0=True)
                close(prev, g[f"ddpm_{vt}_{int(clip)}_{t}_prev"], 2e-6, 2e-6)
                close(x0, g[f"ddpm_{vt}_{int(clip)}_{t}_x0"], 2e-6, 2e-6)
    prev = ops.ddpm_step(dev(eps), dev(x), dev(z), dev(ac), 500, 499, clip_sample=False, clip_defense=True, clip_defense_range=0.5)
    close(prev, g["ddpm_clipdef_500_prev"], 2e-6, 2e-6)
    for clip in (True, False):
        for t in C.DDIM_TS:
            prev = ops.ddim_step(dev(eps), dev(x), dev(ac), t, t - 20, clip_sample=clip)
            close(prev, g[f"ddim_{int(clip)}_{t}_prev"], 2e-6, 2e-6)
        prev = ops.ddim_step(dev(eps), dev(x), dev(ac), 500, 480, eta=0.5, noise=dev(z), clip_sample=clip)
        close(prev, g[f"ddim_{int(clip)}_500_eta_prev"], 2e-6, 2e-6)


def test_ddpm_full_loop_kat(ops):
    # diffusers/tests/schedulers/test_scheduler_ddpm.py:71-100 with the device step kernel
    _, _, ac = sched_ref.make_tables()
    n = 4 * 3 * 8 * 8
    sample = (torch.arange(n).reshape(3, 8, 8, 4) / n).permute(3, 0, 1, 2).contiguous()
    gen = torch.manual_seed(0)
    s = dev(sample); acd = dev(ac)
    for t in reversed(range(1000)):
        residual = s * (t / (t + 1))
        noise = dev(torch.randn(sample.shape, generator=gen)) if t > 0 else None
        s = ops.ddpm_step(residual.contiguous(), s, noise, acd, t, t - 1)
    assert abs(float(s.abs().sum()) - 258.9606) < 5e-2
    assert abs(float(s.abs().mean()) - 0.3372) < 1e-3


def test_to_image(ops):
    x = R(5, 3, 3, 8, 8) * 1.5
    f, u = ops.to_image(dev(x), False, (3, 3, 8, 8), want_u8=True)
    ref = sched_ref.to_image(x)
    close(f, ref, 0, 1e-7)
    assert torch.equal(u.cpu(), (ref * 255).round().to(torch.uint8))
    f2 = ops.to_image(dev(x.permute(0, 2, 3, 1).contiguous()), True, (3, 3, 8, 8))
    close(f2, ref, 0, 1e-7)


def test_timestep_embedding(ops, golden):
    g = golden("temb")
    t = torch.tensor(C.TEMB_TS)
    # sin/cos arguments reach t*f ~ 1e3, where ONE ulp of expf() in f moves the argument by 6e-5: atol 2e-4
    close(ops.timestep_embedding(dev(t), 128, False, 1), g["cifar"], 0, 2e-4)
    close(ops.timestep_embedding(dev(t), 128, True, 0), g["default"], 0, 2e-4)
    t = torch.arange(1000)
    close(ops.timestep_embedding(dev(t), 128, False, 1), U.timestep_embedding(t, 128, False, 1), 0, 2e-4)


# ------------------------------------------------------------------------------------------ GroupNorm
# resident single-pass kernels (slab in registers): all but the 4096-pixel cases, which take the split two-stage path
@pytest.mark.parametrize("B,HW,Cc,silu", [(3, 64, 128, True), (2, 256, 384, True), (2, 16, 512, False), (5, 1024, 256, True),
                                          (130, 16, 128, True), (2, 1024, 128, True), (3, 900, 128, False), (2, 4096, 128, True),
                                          (2, 4096, 64, False), (2, 1024, 384, True)])
def test_groupnorm_fwd_bwd(ops, B, HW, Cc, silu):
    x = R(1, B, HW, Cc) * 1.7 + 0.3
    gam = 1 + 0.1 * R(2, Cc); bet = 0.1 * R(3, Cc); dy = R(4, B, HW, Cc)
    xr = x.clone().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    y_ref = F.group_norm(xr.permute(0, 2, 1), 32, gr, br, 1e-6)
    if silu:
        y_ref = F.silu(y_ref)
    y_ref = y_ref.permute(0, 2, 1)
    y_ref.backward(dy)
    y, mean, rstd = ops.gn_fwd(dev(x), dev(gam), dev(bet), 32, 1e-6, silu)
    close(y, y_ref, 1e-4, 1e-5)
    dx, dg, db, cs = ops.gn_bwd(dev(x), dev(gam), dev(bet), mean, rstd, dev(dy), 32, silu, with_colsum=True)
    close(dx, xr.grad, 1e-3, 2e-5)
    ref_cs = xr.grad.double().sum(1)            # per-sample column sums of dx (time-embedding gradient)
    close(cs, ref_cs.float(), 1e-3, 1e-5 * HW ** 0.5 * max(1.0, float(xr.grad.abs().max())))
    close(dg, gr.grad, 1e-3, 1e-3 * float(gr.grad.abs().max()))
    close(db, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()))
    # strided (channel-slice) input + accumulate
    wide = torch.zeros(B, HW, Cc + 64); wide[:, :, 32:32 + Cc] = x
    wd = dev(wide)
    y2, m2, r2 = ops.gn_fwd(wd[:, :, 32:32 + Cc], dev(gam), dev(bet), 32, 1e-6, silu)
    close(y2, y_ref, 1e-4, 1e-5)
    base = dev(R(9, B, HW, Cc))
    dx2, _, _ = ops.gn_bwd(wd[:, :, 32:32 + Cc], dev(gam), dev(bet), m2, r2, dev(dy), 32, silu, dx=base.clone(), accumulate=True)
    close(dx2, xr.grad + base.cpu(), 1e-3, 3e-5)


# ------------------------------------------------------------------------------------------ igemm: dense
@pytest.mark.parametrize("M,N,K,ta,tb", [(128, 512, 128, False, True), (300, 200, 72, False, True), (256, 256, 256, False, False),
                                         (4992, 512, 128, True, False), (77, 3, 1152, False, True), (64, 8, 64, False, False),
                                         (128, 128, 8, False, True), (130, 131, 33, False, True), (130, 131, 33, True, False)])
def test_gemm_dense(ops, M, N, K, ta, tb):
    a = R(1, K, M) if ta else R(1, M, K)
    b = R(2, N, K) if tb else R(2, K, N)
    bias = R(3, N)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() * 0.5 + bias.double()
    for tile in (0, 64, 128):
        c = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), alpha=0.5, tile=tile)
        close(c, ref.float(), 1e-4, 1e-4)
    c = ops.gemm(dev(a), dev(b), ta, tb, bias=dev(bias), alpha=0.5, ksplit=3)
    close(c, ref.float(), 1e-4, 1e-4)


def test_gemm_transpose_detect(ops):
    # A = I with an ASYMMETRIC B catches a swapped row/col in the MFMA C-write (cdna guide 5.4 rule 16)
    n = 128
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    c = ops.gemm(dev(torch.eye(n)), dev(b), False, False)
    assert torch.equal(c.cpu(), b)


def test_gemm_batched(ops):
    a = R(1, 6, 64, 40); b = R(2, 6, 48, 40)
    close(ops.gemm(dev(a), dev(b)), torch.bmm(a, b.transpose(1, 2)), 1e-4, 1e-4)
    b2 = R(3, 6, 40, 48)
    close(ops.gemm(dev(a), dev(b2), False, False), torch.bmm(a, b2), 1e-4, 1e-4)
    a2 = R(4, 6, 40, 64)
    close(ops.gemm(dev(a2), dev(b2), True, False), torch.bmm(a2.transpose(1, 2), b2), 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------ igemm: conv
CONV_CASES = [  # B, H, Cin, Cout, stride, pad, ups, asym
    (2, 8, 128, 128, 1, 1, 0, False), (3, 16, 32, 64, 1, 1, 0, False), (2, 8, 128, 128, 2, 0, 0, True),
    (2, 8, 128, 128, 2, 1, 0, False), (2, 4, 128, 128, 1, 1, 1, False), (2, 16, 3, 128, 1, 1, 0, False),
    (2, 16, 128, 3, 1, 1, 0, False), (1, 4, 512, 256, 1, 1, 0, False), (2, 9, 36, 20, 1, 1, 0, False),
    (130, 4, 64, 64, 1, 1, 0, False),
    # thin convs (3 or 1 channels on one side, 128-multiples on the other): direct wgrad kernels (conv_thin.hip)
    (3, 10, 3, 256, 1, 1, 0, False), (1, 40, 256, 3, 1, 1, 0, False), (2, 12, 1, 128, 1, 1, 0, False), (5, 32, 3, 128, 1, 1, 0, False),
]


@pytest.mark.parametrize("B,H,Cin,Cout,stride,pad,ups,asym", CONV_CASES)
def test_conv3x3(ops, B, H, Cin, Cout, stride, pad, ups, asym):
    x = R(1, B, Cin, H, H); w = R(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5); bias = R(3, Cout)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    xi = F.interpolate(xr, scale_factor=2.0, mode="nearest") if ups else xr
    if asym:
        xi = F.pad(xi, (0, 1, 0, 1))
    y_ref = F.conv2d(xi, wr, bias, stride=stride, padding=0 if asym else pad)
    dy = R(4, *y_ref.shape)
    y_ref.backward(dy)
    xn = dev(x.permute(0, 2, 3, 1).contiguous()); wn = dev(w.permute(0, 2, 3, 1).contiguous())
    dyn = dev(dy.permute(0, 2, 3, 1).contiguous())
    y = ops.conv3x3_fwd(xn, wn, dev(bias), stride, pad, ups, asym)
    close(y.permute(0, 3, 1, 2), y_ref, 2e-4, 2e-4)
    dx = ops.conv3x3_dgrad(dyn, wn, (B, H, H, Cin), stride, pad, ups, asym)
    if ups:
        dx = ops.sum2x2(dx)
    close(dx.permute(0, 3, 1, 2), xr.grad, 2e-4, 2e-4)
    fused = Cout % 4 == 0     # bias gradient in the same launch (sum of dy over pixels)
    dw = ops.conv3x3_wgrad(xn, dyn, stride, pad, ups, asym, with_db=fused)
    if fused:
        dw, db = dw
        ref_db = dy.double().sum((0, 2, 3)).float()
        close(db, ref_db, 2e-5, 2e-5 * max(1.0, float(ref_db.abs().max())))
    close(dw.permute(0, 3, 1, 2), wr.grad, 2e-4, 2e-4 * max(1.0, float(wr.grad.abs().max())))


def test_conv_epilogue_and_views(ops):
    B, H, Cin, Cout = 2, 8, 128, 128
    x = R(1, B, Cin, H, H); w = R(2, Cout, Cin, 3, 3) / 30; bias = R(3, Cout); rb = R(4, B, 300); res = R(5, B, Cout, H, H)
    ref = (F.conv2d(x, w, bias, padding=1) + rb[:, 100:100 + Cout, None, None] + res) * 0.5
    wide = torch.zeros(B, H, H, Cin + 64); wide[..., 64:] = x.permute(0, 2, 3, 1)
    resw = torch.zeros(B, H, H, Cout + 32); resw[..., :Cout] = res.permute(0, 2, 3, 1)
    y = ops.conv3x3_fwd(dev(wide)[..., 64:], dev(w.permute(0, 2, 3, 1).contiguous()), dev(bias), rowbias=dev(rb)[:, 100:100 + Cout],
                        residual=dev(resw)[..., :Cout], out_scale=0.5)
    close(y.permute(0, 3, 1, 2), ref, 2e-4, 2e-4)


# ------------------------------------------------------------------------------------------ small ops
def test_softmax(ops):
    for n in (16, 64, 256, 100):
        s = R(1, 3, 20, n) * 3; dp = R(2, 3, 20, n)
        sr = s.clone().requires_grad_(True)
        p_ref = torch.softmax(sr, -1); p_ref.backward(dp)
        p = ops.softmax_fwd(dev(s)); close(p, p_ref, 1e-5, 1e-6)
        close(ops.softmax_bwd(p, dev(dp)), sr.grad, 1e-4, 1e-6)


def test_colsum_sum2x2_silu(ops):
    x = R(1, 4 * 64, 130)
    close(ops.colsum(dev(x), 64), x.reshape(4, 64, 130).sum(1), 1e-5, 1e-5)
    close(ops.colsum(dev(x), 256), x.sum(0, keepdim=True), 1e-5, 1e-4)
    du = R(2, 2, 8, 8, 64)
    close(ops.sum2x2(dev(du)), du.reshape(2, 4, 2, 4, 2, 64).sum((2, 4)), 1e-6, 1e-6)
    z = R(3, 1000) * 3; dy = R(4, 1000)
    zr = z.clone().requires_grad_(True); F.silu(zr).backward(dy)
    close(ops.silu_fwd(dev(z)), F.silu(z), 1e-6, 1e-6); close(ops.silu_bwd(dev(z), dev(dy)), zr.grad, 1e-5, 1e-6)


@pytest.mark.parametrize("lt", ["l2", "l1", "huber"])
def test_loss(ops, lt):
    pred = R(1, 6, 16, 16, 3) * 2; tgt = R(2, 6, 16, 16, 3)
    pr = pred.clone().requires_grad_(True)
    fn = {"l2": F.mse_loss, "l1": F.l1_loss, "huber": F.smooth_l1_loss}[lt]
    ref = fn(tgt, pr); ref.backward()
    loss, dp = ops.loss_fwd_bwd(dev(pred), dev(tgt), lt)
    close(loss, ref, 1e-5, 1e-7); close(dp.reshape(pred.shape), pr.grad, 1e-5, 1e-9)


def test_adam_clip_vs_torch(ops):
    n = 100003
    p = R(1, n); g = R(2, n) * 0.01
    pr = torch.nn.Parameter(p.clone()); opt = torch.optim.Adam([pr], lr=2e-4)
    pd = dev(p.clone()); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    for step in (1, 2, 3):
        gs = g * step
        pr.grad = gs.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = dev(gs)
        ss = ops.sumsq(gd)
        gn = torch.empty((), device="cuda")
        ops.adam_clip(pd, gd, m, v, ss, step, 2e-4, grad_norm_out=gn)
        close(gn, norm_ref, 1e-5, 0)
        close(pd, pr.detach(), 1e-6, 1e-7)


# ------------------------------------------------------------------------------------------ split-bf16 mode
@pytest.mark.parametrize("M,N,K,ta,tb", [(256, 256, 1152, False, True), (384, 128, 256, False, False), (1152, 128, 4096, True, False),
                                         (130, 131, 36, False, True)])
def test_gemm_bf16x3(ops, M, N, K, ta, tb):
    """hi/lo bf16 split, 3 MFMAs: error ~2^-16 relative per product, far inside the 1e-3 parity budget"""
    a = R(1, K, M) if ta else R(1, M, K)
    b = R(2, N, K) if tb else R(2, K, N)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    c = ops.gemm(dev(a), dev(b), ta, tb, mode=1).cpu().double()
    scale = float(ref.abs().mean())
    err = float((c - ref).abs().max()) / scale
    assert err < 2e-4, err
    c32 = ops.gemm(dev(a), dev(b), ta, tb, mode=0).cpu().double()
    assert float((c32 - ref).abs().max()) / scale < 2e-5


@pytest.mark.parametrize("B,H,Cin,Cout,stride,pad,ups,asym", [c for c in CONV_CASES if c[2] % 32 == 0 or c[2] == 3][:8])
def test_conv3x3_bf16x3(ops, B, H, Cin, Cout, stride, pad, ups, asym):
    x = R(1, B, Cin, H, H); w = R(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5); bias = R(3, Cout)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    xi = F.interpolate(xr, scale_factor=2.0, mode="nearest") if ups else xr
    if asym:
        xi = F.pad(xi, (0, 1, 0, 1))
    y_ref = F.conv2d(xi, wr, bias, stride=stride, padding=0 if asym else pad)
    dy = R(4, *y_ref.shape)
    y_ref.backward(dy)
    xn = dev(x.permute(0, 2, 3, 1).contiguous()); wn = dev(w.permute(0, 2, 3, 1).contiguous())
    dyn = dev(dy.permute(0, 2, 3, 1).contiguous())
    y = ops.conv3x3_fwd(xn, wn, dev(bias), stride, pad, ups, asym, mode=1)
    close(y.permute(0, 3, 1, 2), y_ref, 3e-4, 3e-4)
    dx = ops.conv3x3_dgrad(dyn, wn, (B, H, H, Cin), stride, pad, ups, asym, mode=1)
    if ups:
        dx = ops.sum2x2(dx)
    close(dx.permute(0, 3, 1, 2), xr.grad, 3e-4, 3e-4)
    fused = Cout % 4 == 0
    dw = ops.conv3x3_wgrad(xn, dyn, stride, pad, ups, asym, mode=1, with_db=fused)
    if fused:   # the fused bias gradient is summed in fp32 from the unsplit values in both modes
        dw, db = dw
        ref_db = dy.double().sum((0, 2, 3)).float()
        close(db, ref_db, 2e-5, 2e-5 * max(1.0, float(ref_db.abs().max())))
    close(dw.permute(0, 3, 1, 2), wr.grad, 3e-4, 3e-4 * max(1.0, float(wr.grad.abs().max())))


def test_split_bf16_planes(ops):
    x = R(1, 4096) * torch.logspace(-6, 6, 4096)
    sp = ops.split_bf16(dev(x))
    hi, lo = sp[:, 0].reshape(-1), sp[:, 1].reshape(-1)     # blocked layout [block, hi|lo, 32]
    hif = ((hi.cpu().int() & 0xFFFF) << 16).view(torch.float32); lof = ((lo.cpu().int() & 0xFFFF) << 16).view(torch.float32)
    assert torch.equal(hif, x.to(torch.bfloat16).float())                         # hi = RNE bf16 (round 6; truncation before: one bit less)
    assert float(((hif.double() + lof.double() - x.double()).abs() / x.double().abs()).max()) < 2.0 ** -17
    assert torch.equal(lof, (x - hif).to(torch.bfloat16).float())                 # lo = RNE bf16 of the (exact) remainder


@pytest.mark.parametrize("B,H,Cin,Cout,stride,pad,ups,asym", [(2, 16, 128, 256, 1, 1, 0, False), (2, 16, 256, 128, 2, 0, 0, True),
                                                              (2, 8, 128, 128, 1, 1, 1, False), (3, 8, 64, 64, 1, 1, 0, False),
                                                              (130, 4, 256, 256, 1, 1, 0, False)])
def test_conv3x3_presplit_weights_bit_identical(ops, B, H, Cin, Cout, stride, pad, ups, asym):
    """weights handed over as bf16 hi/lo planes (bd_split_bf16, once per step) give bit-identical results to the
    on-the-fly split of the same kernel family"""
    x = dev(R(1, B, H, H, Cin)); w = dev(R(2, Cout, 3, 3, Cin) / (3 * Cin ** 0.5)); bias = dev(R(3, Cout))
    ws = ops.split_bf16(w)
    y0 = ops.conv3x3_fwd(x, w, bias, stride, pad, ups, asym, mode=1)
    y1 = ops.conv3x3_fwd(x, w, bias, stride, pad, ups, asym, mode=1, w_split=ws)
    assert torch.equal(y0, y1)
    dy = dev(R(4, *y0.shape))
    d0 = ops.conv3x3_dgrad(dy, w, (B, H, H, Cin), stride, pad, ups, asym, mode=1)
    d1 = ops.conv3x3_dgrad(dy, w, (B, H, H, Cin), stride, pad, ups, asym, mode=1, w_split=ws)
    assert torch.equal(d0, d1)
    # exact mode ignores the planes
    assert torch.equal(ops.conv3x3_fwd(x, w, bias, stride, pad, ups, asym, mode=0, w_split=ws), ops.conv3x3_fwd(x, w, bias, stride, pad, ups, asym, mode=0))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K,ksplit", [(128, 256, 4096, 0), (384, 128, 512, 1), (64, 64, 96, 1), (132, 72, 1000, 0), (256, 1152, 131072, 0)])
def test_gemm_fused_colsum(ops, M, N, K, ksplit, mode):
    """dW = dY^T X with db = column sums of dY out of the same launch (with and without split-K)"""
    a = R(1, K, M); b = R(2, K, N)
    cs = torch.full((M,), float("nan"), device="cuda")
    c = ops.gemm(dev(a), dev(b), True, False, ksplit=ksplit, mode=mode, a_colsum=cs).cpu().double()
    ref = a.double().t() @ b.double()
    assert float((c - ref).abs().max()) / float(ref.abs().mean()) < 2e-4
    ref_cs = a.double().sum(0)
    # fp32 summation: a few ulps of the column's L1 norm
    assert float((cs.cpu().double() - ref_cs).abs().max()) < 3e-7 * float(a.double().abs().sum(0).max())


def test_gemm_fused_colsum_rejects_kc(ops):
    a = R(1, 64, 128); b = R(2, 64, 128)
    with pytest.raises(RuntimeError, match="a_colsum"):
        ops.gemm(dev(a), dev(b), False, True, a_colsum=torch.empty(64, device="cuda"))
