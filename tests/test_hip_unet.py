"""GPU parity: the whole UNet2DModel plan (forward, backward, clip + Adam) against the CPU oracle and the
golden vectors captured from the reference (tests/golden/unet_small.npz, unet_cifar.npz).
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star); observed errors are ~1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss_ref, sched_ref, train_ref
from oracle import unet_ref as U
from tests.golden import cases as C


@pytest.fixture(scope="module")
def bd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import baddiffusion_amd.unet as unet
    import baddiffusion_amd.ops as ops
    return unet, ops


def relerr(a, b):
    a = a.detach().cpu().double(); b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_model(unet, cfg, seed):
    m = unet.unet_from_config(cfg).cuda()
    m.load_state_dict(U.gen_params(cfg, seed))
    return m


def test_state_dict_roundtrip(bd):
    unet, _ = bd
    cfg = C.SMALL_CFGS["small"]
    P = U.gen_params(cfg, 3)
    m = make_model(unet, cfg, 3)
    sd = m.state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k


@pytest.mark.parametrize("tag", ["small", "small_default"])
def test_unet_forward_vs_oracle(bd, tag):
    unet, _ = bd
    cfg = C.SMALL_CFGS[tag]
    m = make_model(unet, cfg, 7)
    P = U.gen_params(cfg, 7)
    x = torch.randn(3, 3, 16, 16, generator=torch.Generator().manual_seed(5))
    for t in (torch.tensor([3, 500, 999]), torch.tensor(17), 998):
        with torch.no_grad():
            ref = U.unet_forward(cfg, P, x, t)
            out = m(x.cuda(), t.cuda() if torch.is_tensor(t) else t).sample
        assert out.shape == ref.shape
        assert relerr(out, ref) < 1e-4, relerr(out, ref)
    # chunked inference == unchunked (samples are independent)
    m.max_chunk = 2
    with torch.no_grad():
        out2 = m(x.cuda(), torch.tensor([3, 500, 999]).cuda(), return_dict=False)[0]
        ref = U.unet_forward(cfg, P, x, torch.tensor([3, 500, 999]))
    assert relerr(out2, ref) < 1e-4


def _train_step(bd, cfg, seed, B, tag, g, lr=2e-4, mode="f32"):
    unet, ops = bd
    _, a, ac = sched_ref.make_tables()
    m = make_model(unet, cfg, seed).set_compute_mode(mode)
    x0, R, t, eps = C.train_inputs(cfg, B)
    xn, tg = ops.qsample(x0.cuda(), R.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda())
    # forward (inference mode) vs golden prediction
    with torch.no_grad():
        pred = m(xn.permute(0, 3, 1, 2), t.cuda()).sample
    assert relerr(pred, g[f"{tag}_pred"]) < 1e-4
    # training step through autograd: loss + grads
    pred = m(xn.permute(0, 3, 1, 2), t.cuda(), return_dict=False)[0]
    loss, dp = ops.loss_fwd_bwd(pred.permute(0, 2, 3, 1), tg, "l2")
    pred.backward(dp.reshape(pred.permute(0, 2, 3, 1).shape).permute(0, 3, 1, 2))
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-4 * abs(float(g[f"{tag}_loss"]))
    names = [str(s) for s in g[f"{tag}_names"]]
    grads = m.logical_grads()
    gn = np.array([float(grads[k].double().norm()) for k in names])
    ref = g[f"{tag}_gradnorms"]
    # floor: gradients that are mathematically zero (key.bias) are rounding noise, judged against the largest norm
    bad = [(k, a_, b_) for k, a_, b_ in zip(names, gn, ref) if abs(a_ - b_) > 1e-3 * max(b_, 1e-4 * ref.max())]
    assert not bad, bad[:10]
    g8 = np.stack([np.pad(grads[k].contiguous().flatten()[:8].cpu().numpy(), (0, max(0, 8 - grads[k].numel()))) for k in names])
    np.testing.assert_allclose(g8, g[f"{tag}_grad8"], rtol=2e-3, atol=2e-3 * float(np.abs(g[f"{tag}_grad8"]).max()))
    # full-tensor gradient check against the oracle's autograd
    _, G = train_ref.loss_and_grads(cfg, U.gen_params(cfg, seed), a, ac, x0, R, t, eps)
    # (key.bias gradients are mathematically zero -- softmax is shift-invariant -- so they are pure rounding
    #  noise on both sides: judge every tensor against 1e-3 of its own norm OR 1e-6 of the global grad norm)
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in G.values())))
    def err(k):
        d = float((grads[k].detach().cpu().double() - G[k].double()).norm())
        return d / max(float(G[k].double().norm()), 1e-3 * total)
    worst = max((err(k), k) for k in names)
    assert worst[0] < 1e-3, worst
    # clip + Adam on the flat buffers
    flat = m.flat.data; gflat = m.flat.grad
    mom = torch.zeros_like(flat); var = torch.zeros_like(flat)
    ss = ops.sumsq(gflat)
    norm = torch.empty((), device="cuda")
    ops.adam_clip(flat, gflat, mom, var, ss, 1, lr, grad_norm_out=norm)
    assert abs(float(norm) - float(g[f"{tag}_total_norm"])) < 1e-3 * float(g[f"{tag}_total_norm"])
    sd = m.state_dict()
    p8 = np.stack([np.pad(sd[k].flatten()[:8].cpu().numpy(), (0, max(0, 8 - sd[k].numel()))) for k in names])
    np.testing.assert_allclose(p8, g[f"{tag}_p8_after"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("tag", ["small", "small_default"])
def test_small_unet_train_step(bd, golden, tag):
    _train_step(bd, C.SMALL_CFGS[tag], 7, 2, tag, golden("unet_small"))


def test_cifar_unet_train_step(bd, golden):
    _train_step(bd, U.CIFAR10_32, 0, 2, "cifar", golden("unet_cifar"))


def test_cifar_unet_train_step_bf16x3(bd, golden):
    """same golden vectors, same 1e-3 tolerances, split-bf16 contraction"""
    _train_step(bd, U.CIFAR10_32, 0, 2, "cifar", golden("unet_cifar"), mode="bf16x3")


def test_small_unet_train_step_bf16x3(bd, golden):
    _train_step(bd, C.SMALL_CFGS["small"], 7, 2, "small", golden("unet_small"), mode="bf16x3")


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_celeba_topology_train_step_vs_oracle(bd, mode):
    """BASELINE configs[3] topology (6 levels, (128,128,256,256,512,512), attention on level 4) at 64x64 so that the
    CPU oracle finishes in seconds: loss and every gradient tensor against the oracle's autograd.  64x64 feature
    maps also take the split (non-resident) GroupNorm kernels and split-K / non-split conv paths the 32x32 net
    never reaches."""
    import dataclasses
    unet, ops = bd
    cfg = dataclasses.replace(U.CELEBA_HQ_256, sample_size=64)
    _, a, ac = sched_ref.make_tables()
    m = make_model(unet, cfg, 11).set_compute_mode(mode)
    x0, R, t, eps = C.train_inputs(cfg, 2)
    x0, R, t, eps = x0[:1], R[:1], t[:1], eps[:1]
    xn, tg = ops.qsample(x0.cuda(), R.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda())
    pred = m(xn.permute(0, 3, 1, 2), t.cuda(), return_dict=False)[0]
    loss, dp = ops.loss_fwd_bwd(pred.permute(0, 2, 3, 1), tg, "l2")
    pred.backward(dp.reshape(pred.permute(0, 2, 3, 1).shape).permute(0, 3, 1, 2))
    ref_loss, G = train_ref.loss_and_grads(cfg, U.gen_params(cfg, 11), a, ac, x0, R, t, eps)
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    grads = m.logical_grads()
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in G.values())))
    def err(k):
        d = float((grads[k].detach().cpu().double() - G[k].double()).norm())
        return d / max(float(G[k].double().norm()), 1e-3 * total)
    worst = max((err(k), k) for k in G)
    assert worst[0] < 1e-3, worst


def _golden_grad_check(m, g, tag, rtol=1e-3):
    """per-tensor gradient norms and leading 8 elements against what the imported reference computed (G10)"""
    names = [str(s) for s in g[f"{tag}_names"]]
    grads = m.logical_grads()
    gn = np.array([float(grads[k].double().norm()) for k in names])
    ref = g[f"{tag}_gradnorms"]
    bad = [(k, a_, b_) for k, a_, b_ in zip(names, gn, ref) if abs(a_ - b_) > rtol * max(b_, 1e-4 * ref.max())]
    assert not bad, bad[:10]
    g8 = np.stack([np.pad(grads[k].contiguous().flatten()[:8].cpu().numpy(), (0, max(0, 8 - grads[k].numel()))) for k in names])
    np.testing.assert_allclose(g8, g[f"{tag}_grad8"], rtol=2e-3, atol=2e-3 * float(np.abs(g[f"{tag}_grad8"]).max()))
    return names


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_cifar_full_batch_train_step_vs_reference(bd, golden, mode):
    """BASELINE configs[1] at its real size (DDPM-CIFAR10-32 topology, batch 128) against vectors the imported reference
    produced for exactly this batch (tests/golden/make_golden.py g10 -> full_size.npz; unet_2d.py:229-326,
    baddiffusion.py:590-615): prediction rows of both half-batch pipelines, per-sample checksums of all 128 predictions, loss,
    every gradient tensor's norm + leading elements, the clip norm and the post-Adam weights.  1e-3 relative (north_star)."""
    unet, ops = bd
    g = golden("full_size"); tag = "cifar128"
    cfg = U.CIFAR10_32
    _, a, ac = sched_ref.make_tables()
    m = make_model(unet, cfg, 0).set_compute_mode(mode)
    x0, R, t, eps = C.train_inputs(cfg, 128)
    xn, tg = ops.qsample(x0.cuda(), R.cuda(), eps.cuda(), t.cuda(), a.cuda(), ac.cuda())
    pred = m(xn.permute(0, 3, 1, 2), t.cuda(), return_dict=False)[0]
    pd = pred.detach()
    assert relerr(pd[list(C.FULL_ROWS)], g[f"{tag}_pred_rows"]) < 1e-4
    s1 = pd.double().sum(dim=(1, 2, 3)).cpu().numpy(); s2 = (pd.double() ** 2).sum(dim=(1, 2, 3)).cpu().numpy()
    np.testing.assert_allclose(s2, g[f"{tag}_pred_sumsq"], rtol=1e-4)
    np.testing.assert_allclose(s1, g[f"{tag}_pred_sum"], rtol=1e-3, atol=1e-3 * float(np.sqrt(g[f"{tag}_pred_sumsq"].max() * 3072)))
    loss, dp = ops.loss_fwd_bwd(pred.permute(0, 2, 3, 1), tg, "l2")
    pred.backward(dp.reshape(pred.permute(0, 2, 3, 1).shape).permute(0, 3, 1, 2))
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-4 * abs(float(g[f"{tag}_loss"]))
    names = _golden_grad_check(m, g, tag)
    flat = m.flat.data; gflat = m.flat.grad
    mom = torch.zeros_like(flat); var = torch.zeros_like(flat)
    norm = torch.empty((), device="cuda")
    ops.adam_clip(flat, gflat, mom, var, ops.sumsq(gflat), 1, 2e-4, grad_norm_out=norm)
    assert abs(float(norm) - float(g[f"{tag}_total_norm"])) < 1e-3 * float(g[f"{tag}_total_norm"])
    sd = m.state_dict()
    p8 = np.stack([np.pad(sd[k].flatten()[:8].cpu().numpy(), (0, max(0, 8 - sd[k].numel()))) for k in names])
    # first Adam update is +-lr wherever |g| is above rounding noise: elements whose gradient is noise may flip sign
    d = np.abs(p8 - g[f"{tag}_p8_after"])
    assert d.max() <= 2.2 * 2e-4 and (d > 5e-5).mean() < 5e-3, (d.max(), (d > 5e-5).mean())


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_celeba_full_resolution_vs_reference(bd, golden, mode):
    """the real 256x256 DDPM-CELEBA-HQ-256 network (113.7 M parameters), batch 1, forward + backward against the imported
    reference's output (strided slices + checksums) and every gradient tensor's norm + leading elements (G10)."""
    unet, ops = bd
    g = golden("full_size"); tag = "celeba256"
    cfg = U.CELEBA_HQ_256
    m = make_model(unet, cfg, 5).set_compute_mode(mode)
    x, t, dout = C.celeba_full_inputs()
    out = m(x.cuda(), t.cuda(), return_dict=False)[0]
    od = out.detach()
    assert relerr(od[0, :, ::16, ::16], g[f"{tag}_out_slices"]) < 1e-4
    assert abs(float((od.double() ** 2).sum()) - float(g[f"{tag}_out_sumsq"])) < 1e-4 * float(g[f"{tag}_out_sumsq"])
    assert abs(float(od.double().sum()) - float(g[f"{tag}_out_sum"])) < 1e-3 * float(np.sqrt(float(g[f"{tag}_out_sumsq"]) * od.numel()))
    out.backward(dout.cuda())
    _golden_grad_check(m, g, tag)


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_celeba_batch4_vs_reference(bd, golden, mode):
    """BASELINE configs[3]'s per-GPU share at its real size: the 256x256 DDPM-CELEBA-HQ-256 network at B = 4 (four different
    timesteps), forward + backward against what the imported reference computed for this batch (G12, make_golden.py g12 ->
    full_size_b4.npz; unet_2d.py:229-326): per-sample output slices + checksums, every gradient tensor's norm + leading elements.
    At this batch the wide layers take the strip-order kernels (conv_ps3 / wgrad3 at W = 64 .. 256) and the large-image GroupNorm path."""
    unet, ops = bd
    g = golden("full_size_b4"); tag = "celeba256b4"
    cfg = U.CELEBA_HQ_256
    m = make_model(unet, cfg, 5).set_compute_mode(mode)
    x, t, dout = C.celeba_b4_inputs()
    out = m(x.cuda(), t.cuda(), return_dict=False)[0]
    od = out.detach()
    assert relerr(od[:, :, ::16, ::16], g[f"{tag}_out_slices"]) < 1e-4
    for i in range(4):
        sq = float(g[f"{tag}_out_sumsq"][i])
        assert abs(float((od[i].double() ** 2).sum()) - sq) < 1e-4 * sq, i
        assert abs(float(od[i].double().sum()) - float(g[f"{tag}_out_sum"][i])) < 1e-3 * float(np.sqrt(sq * od[i].numel())), i
    out.backward(dout.cuda())
    _golden_grad_check(m, g, tag)


def test_cifar_full_batch_modes_and_schedules_agree(bd):
    """BASELINE configs[1] at its full size (DDPM-CIFAR10-32, batch 128): the oracle would need a minute, so the check
    is self-consistency across the independent code paths -- split-bf16 vs exact-fp32 contraction, two-stream vs
    single-stream schedule -- plus per-sample independence (sample i of the batch == the same sample run alone)."""
    unet, ops = bd
    cfg = U.CIFAR10_32
    m = make_model(unet, cfg, 0)
    B = 128
    x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(4)).cuda()
    dout = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(5)).cuda() / (B * 3072)
    res = {}
    for mode, aux in (("f32", True), ("bf16x3", True), ("bf16x3", False)):
        m.set_compute_mode(mode); m.set_aux_stream(aux)
        m.flat.grad = None
        out = m(x, t, return_dict=False)[0]
        out.backward(dout)
        assert torch.isfinite(out).all() and torch.isfinite(m.flat.grad).all()
        res[(mode, aux)] = (out.detach().clone(), m.flat.grad.detach().clone())
    m.set_aux_stream(True)
    assert relerr(res[("bf16x3", True)][0], res[("f32", True)][0]) < 1e-4
    assert relerr(res[("bf16x3", True)][1], res[("f32", True)][1]) < 1e-3
    assert relerr(res[("bf16x3", True)][0], res[("bf16x3", False)][0]) < 5e-5
    assert relerr(res[("bf16x3", True)][1], res[("bf16x3", False)][1]) < 2e-4
    with torch.no_grad():
        for i in (0, 63, 64, 127):     # both half-batch pipelines
            alone = m(x[i:i + 1], t[i:i + 1], return_dict=False)[0]
            assert relerr(alone, res[("bf16x3", True)][0][i:i + 1]) < 5e-5


def test_celeba_full_resolution_modes_agree(bd):
    """the real 256x256 DDPM-CELEBA-HQ-256 network (113.7 M parameters), batch 1: forward + backward run, are finite,
    and the split-bf16 contraction agrees with the exact-fp32 one (size-independent self-consistency; the oracle
    would need minutes here)."""
    unet, ops = bd
    cfg = U.CELEBA_HQ_256
    m = make_model(unet, cfg, 5)
    assert m.num_flat >= 113673219
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(3)).cuda()
    t = torch.tensor([417]).cuda()
    dout = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(4)).cuda()
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


@pytest.mark.parametrize("B", [33, 64])
def test_two_stream_schedule_matches_single_stream(bd, B):
    """B >= 32: forward runs as two half-batch pipelines (caller's stream + the plan's side stream) and backward puts the
    weight gradients on the side stream; the single-stream order (bd_unet_set_aux_stream(0)) must give the same numbers
    up to the summation order of split-K (tile counts differ between a half and the full batch)."""
    unet, ops = bd
    cfg = C.SMALL_CFGS["small"]
    m = make_model(unet, cfg, 7)
    x = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(1)).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(2)).cuda()
    dout = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(3)).cuda()
    res = []
    for aux in (1, 0):
        m.set_aux_stream(bool(aux))
        m.flat.grad = None
        out = m(x, t, return_dict=False)[0]
        out.backward(dout)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), m.flat.grad.detach().clone()))
    m.set_aux_stream(True)
    # fp32 accumulation order differs (split-K counts, GroupNorm slab widths): rounding-level differences, amplified by
    # the depth of the network to ~1e-5 -- an order of magnitude below either schedule's distance to the oracle
    d_out, d_grad = relerr(res[0][0], res[1][0]), relerr(res[0][1], res[1][1])
    assert d_out < 5e-5 and d_grad < 2e-4, (d_out, d_grad)
    # and against the oracle on a few samples of both halves
    P = U.gen_params(cfg, 7)
    idx = [0, B // 2 - 1, B // 2, B - 1]
    with torch.no_grad():
        ref = U.unet_forward(cfg, P, x[idx].cpu(), t[idx].cpu())
    assert relerr(res[0][0][idx], ref) < 1e-4


@pytest.mark.parametrize("B", [2, 40])     # 40: the forward also forks into its two half-batch pipelines
def test_forward_backward_hipgraph_capture(bd, B):
    """the C ABI only enqueues on the given stream (and, in backward, on the plan's side stream forked / joined with
    events): a whole forward + backward is capturable in a HIP graph and replays to the same bits as the eager run"""
    unet, ops = bd
    cfg = C.SMALL_CFGS["small"]
    m = make_model(unet, cfg, 7)
    x = torch.randn(B, 16, 16, 3, generator=torch.Generator().manual_seed(1)).cuda()
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(5)).cuda()
    dout = torch.randn(B, 16, 16, 3, generator=torch.Generator().manual_seed(2)).cuda()
    ws = torch.empty(m.workspace_bytes(B, True), dtype=torch.uint8, device="cuda")
    out_e = torch.empty(B, 16, 16, 3, device="cuda"); g_e = torch.zeros(m.num_flat, device="cuda")
    out_g = torch.empty_like(out_e); g_g = torch.zeros_like(g_e)
    from baddiffusion_amd import _lib as L
    lib = L.load()

    def run(out, grads):
        L.check(lib.bd_unet_forward(m._plan, B, 1, m.flat.data_ptr(), x.data_ptr(), 3, t.data_ptr(), 1, out.data_ptr(), 3,
                                    ws.data_ptr(), ws.numel(), L.stream()), "fwd")
        L.check(lib.bd_unet_backward(m._plan, B, m.flat.data_ptr(), x.data_ptr(), 3, dout.data_ptr(), 3, grads.data_ptr(),
                                     ws.data_ptr(), ws.numel(), L.stream()), "bwd")

    run(out_e, g_e)                      # eager (also creates the plan's side stream outside of the capture)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run(out_g, g_g)
    out_g.zero_(); g_g.fill_(123.0)   # padding elements between tensors are never written
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, out_e)
    written = g_g != 123.0
    assert int(written.sum()) >= m.num_flat - 64 * len(m.state_dict())   # everything but alignment padding was rewritten
    assert torch.equal(g_g[written], g_e[written])


def test_backward_segments_equal_whole(bd):
    """bd_unet_backward_segment over all segments == bd_unet_backward (the DP overlap path)."""
    unet, ops = bd
    import ctypes
    from baddiffusion_amd import _lib as L
    cfg = C.SMALL_CFGS["small"]
    m = make_model(unet, cfg, 7)
    x = torch.randn(2, 16, 16, 3, generator=torch.Generator().manual_seed(1)).cuda()
    t = torch.tensor([10, 900]).cuda()
    dout = torch.randn(2, 16, 16, 3, generator=torch.Generator().manual_seed(2)).cuda()
    out, ws = m._run_forward(m.flat.data, x, t, True)
    g1 = m._run_backward(m.flat.data, x, dout, ws)
    out, ws2 = m._run_forward(m.flat.data, x, t, True)
    g2 = torch.zeros_like(g1)
    lib = L.load()
    nseg = lib.bd_unet_num_segments(m._plan)
    covered = torch.zeros(m.num_flat, dtype=torch.bool)
    for s in range(nseg):
        lo = ctypes.c_int64(); hi = ctypes.c_int64()
        L.check(lib.bd_unet_backward_segment(m._plan, s, 2, m.flat.data_ptr(), x.data_ptr(), 3, dout.data_ptr(), 3, g2.data_ptr(),
                                             ws2.data_ptr(), ws2.numel(), L.stream(), ctypes.byref(lo), ctypes.byref(hi)))
        assert 0 <= lo.value < hi.value <= m.num_flat
        # every range reported ready (the segment's own parameters + its resnets' rows of the batched time_emb_proj)
        # must already be final, and the ranges partition the flat gradient
        torch.cuda.synchronize()
        for k in range(lib.bd_unet_segment_num_ranges(m._plan, s)):
            L.check(lib.bd_unet_segment_range_k(m._plan, s, k, ctypes.byref(lo), ctypes.byref(hi)))
            assert not covered[lo.value:hi.value].any()
            assert torch.equal(g2[lo.value:hi.value], g1[lo.value:hi.value]), (s, k)
            covered[lo.value:hi.value] = True
    assert torch.equal(g1, g2)
    assert int((~covered).sum()) < 64      # only alignment padding is uncovered
