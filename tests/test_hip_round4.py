"""GPU parity, round 4.

  * the data-parallel communication path (trainer.py: comm stream, side-stream event wait, async all_reduce per finished
    gradient range) under RCCL on ONE GPU (BD_FORCE_DP / TrainEngine(force_dp=True)), with an ordering check that turns a
    missing stream dependency into NaNs;
  * the chunk the inference path picks from the free device memory, and its allocation-failure fallback;
  * static-weights sampling loops (bd_unet_set_static_weights) against per-evaluation weight preprocessing.
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star) unless a tighter one is stated."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref as U
from tests.golden import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def relerr(a, b):
    a = torch.as_tensor(a) if not torch.is_tensor(a) else a
    b = torch.as_tensor(b) if not torch.is_tensor(b) else b
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("transport", ["rccl", "c10d"])
def test_dp_path_under_rccl_at_world_1(gpu, transport):
    """VERDICT round 3, task 3: trainer.py's DP branch (third stream ordered by bd_unet_stream_wait_aux, async all_reduce with
    RCCL called directly on our communication stream -- the product transport -- or through torch.distributed's nccl backend,
    each on a 1-rank communicator) over 3 steps of the CIFAR topology at B = 128 ends with weights and Adam moments
    BIT-IDENTICAL to the ordinary step.  dp_check feeds the optimizer from snapshots taken on the collective's stream of a
    gradient buffer that was NaN before backward, so a collective issued before its range is final cannot pass
    (reference: nn.DataParallel, baddiffusion.py:325 -> one process per GPU + all-reduce)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BD_DP_TRANSPORT=transport)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_force_dp_worker.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["transport"] == ("rccl-direct" if transport == "rccl" else "c10d:nccl") and d["world"] == 1
    assert d["finite"] and d["equal"] and d["moments_equal"], d
    assert d["collective_bytes"] >= 4 * 35_000_000          # every range of the 35.7 M-parameter gradient went through RCCL


def test_inference_chunk_follows_free_memory(gpu, monkeypatch):
    """the inference chunk is derived from the free device memory (effective_chunk) and halves when
    the workspace allocation fails, instead of a fixed 2048 that needs 62 GiB; chunked results equal the unchunked ones."""
    from baddiffusion_amd.unet import unet_from_config
    cfg = C.SMALL_CFGS["small"]
    m = unet_from_config(cfg).to(gpu)
    m.load_state_dict(U.gen_params(cfg, 3))
    x = torch.randn(12, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(0)).to(gpu)
    with torch.no_grad():
        want = m(x, 17).sample.clone()
    assert m.effective_chunk(12) == 12
    need12, need3 = m.workspace_bytes(12, False), m.workspace_bytes(3, False)
    real = torch.cuda.mem_get_info
    m._ws_pool = {}
    # pretend only ~ the 3-sample workspace fits: the chunk comes from the fixed ladder max_chunk / 2^k (2048 ... 8, 4, 2), so 12 -> 2
    # (round 5, ADVICE: the path the plan picks depends on the chunk; a ladder makes a reduced chunk reproducible for a given B)
    slack = torch.cuda.memory_reserved(gpu) - torch.cuda.memory_allocated(gpu)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(need3 / 0.85) + 4096 - slack, real()[1]))
    with pytest.warns(UserWarning, match="inference chunk reduced"):
        assert need12 > need3 and m.effective_chunk(12) == 2 and m.last_chunk == 2
    with torch.no_grad():
        got = m(x, 17).sample
    assert relerr(got, want) < 2e-5
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)
    # allocation failure inside the forward: halve and retry
    m._ws_pool = {}
    calls = {"n": 0}
    orig = m._acquire_ws

    def flaky(B, training):
        calls["n"] += 1
        if B > 4:
            raise torch.cuda.OutOfMemoryError("simulated")
        return orig(B, training)
    monkeypatch.setattr(m, "_acquire_ws", flaky)
    with torch.no_grad():
        got = m(x, 17).sample
    assert relerr(got, want) < 2e-5 and m.max_chunk <= 4


def test_static_weights_sampling_loop(gpu):
    """bd_unet_set_static_weights: a DDIM loop that prepares the weights once equals the loop that prepares them at every
    evaluation bit for bit, and a weight update between two loops is picked up (the promise ends with the loop)."""
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.unet import unet_from_config
    cfg = C.SMALL_CFGS["small"]
    m = unet_from_config(cfg).to(gpu)
    m.load_state_dict(U.gen_params(cfg, 4))
    pipe = DDIMPipeline(m, DDPMScheduler())
    init = torch.randn(4, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(1)).to(gpu)
    a = pipe(batch_size=4, init=init, num_inference_steps=6, output_type=None).images
    # reference: the same loop with the promise never made
    lib = m._lib
    x = init.clone()
    sched = pipe.scheduler
    sched.set_timesteps(6)
    with torch.no_grad():
        for t in sched.timesteps:
            eps = m(x, int(t)).sample
            x = sched.step(eps, int(t), x).prev_sample
    b = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(a, b)
    with torch.no_grad():
        m.flat.data.mul_(1.01)
    c = pipe(batch_size=4, init=init, num_inference_steps=6, output_type=None).images
    assert not np.array_equal(a, c)
    assert lib.bd_unet_set_static_weights(m._plan, 0) == 0


# ------------------------------------------------------------------------------------------------ f-3: FID feature extractor
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [
    (2, 19, 19, 3, 32, (3, 3), 2, (0, 0)),        # the stem: 3 channels (scalar loader), stride 2
    (3, 17, 17, 80, 192, (3, 3), 1, (0, 0)),      # Cin % 16 != 0 (ragged K chunk)
    (2, 12, 12, 48, 64, (5, 5), 1, (2, 2)),
    (2, 17, 17, 160, 160, (1, 7), 1, (0, 3)),
    (2, 17, 17, 160, 192, (7, 1), 1, (3, 0)),
    (2, 8, 8, 448, 384, (3, 3), 1, (1, 1)),
    (5, 9, 9, 96, 96, (3, 3), 2, (0, 0)),         # ragged pixel tiles: M = 5 * 16 = 80
    (1, 8, 8, 2048, 320, (1, 1), 1, (0, 0)),
])
def test_conv2d_nhwc_vs_torch(gpu, B, H, W, Cin, Cout, k, stride, pad):
    """bd_conv2d_nhwc (BasicConv2d of the FID Inception network with BatchNorm folded: conv + bias + ReLU, output at a channel
    offset of a wider buffer) against F.conv2d in fp64 on the CPU.  Exact fp32 products: 1e-5 relative."""
    import ctypes as CT
    import torch.nn.functional as F
    from baddiffusion_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / (Cin * k[0] * k[1]) ** 0.5
    b = torch.randn(Cout, generator=g)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride, pad)).permute(0, 2, 3, 1)
    Ho, Wo = want.shape[1:3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    wd = w.permute(2, 3, 1, 0).contiguous().to(gpu)
    bd = b.to(gpu)
    out = torch.full((B, Ho, Wo, Cout + 8), -7.0, device=gpu)
    view = out[..., 4:4 + Cout]
    d = L.Conv2dDesc(x=xd.data_ptr(), ldx=Cin, w=wd.data_ptr(), bias=bd.data_ptr(), y=view.data_ptr(), ldy=Cout + 8, B=B, H=H, W=W, Cin=Cin,
                     Cout=Cout, KH=k[0], KW=k[1], stride_h=stride, stride_w=stride, pad_h=pad[0], pad_w=pad[1], relu=1)
    L.check(lib.bd_conv2d_nhwc(CT.byref(d), L.stream()), "bd_conv2d_nhwc")
    assert relerr(view, want) < 1e-5
    assert float(out[..., :4].min()) == -7.0 and float(out[..., 4 + Cout:].max()) == -7.0        # neighbours of the slice untouched
    # round 5: the K-contiguous weight layout [KH][KW][Cout][Cin] -> the double-buffered 128 x 64 kernel (what inception.py launches)
    wk = w.permute(2, 3, 0, 1).contiguous().to(gpu)
    out.fill_(-7.0)
    d.w = wk.data_ptr(); d.w_kc = 1
    L.check(lib.bd_conv2d_nhwc(CT.byref(d), L.stream()), "bd_conv2d_nhwc")
    assert relerr(view, want) < 1e-5
    assert float(out[..., :4].min()) == -7.0 and float(out[..., 4 + Cout:].max()) == -7.0


def test_inception_pools_and_resize_vs_torch(gpu):
    """bd_pool2d_nhwc (max 3/2/0 and 3/1/1, average 3/1/1 with count_include_pad False and True), bd_global_avgpool_nhwc and
    bd_resize_bilinear_nhwc (uint8 and float sources, up- and down-scaling, then 2v - 1) against torch.nn.functional on the CPU."""
    import torch.nn.functional as F
    from baddiffusion_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 24, 17, 15, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    for (k, s, p, mode, cip), want in (((3, 2, 0, 0, 0), F.max_pool2d(x, 3, 2)), ((3, 1, 1, 0, 0), F.max_pool2d(x, 3, 1, 1)),
                                       ((3, 1, 1, 1, 0), F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)),
                                       ((3, 1, 1, 1, 1), F.avg_pool2d(x, 3, 1, 1, count_include_pad=True))):
        Ho, Wo = want.shape[2:]
        y = torch.empty(3, Ho, Wo, 24, device=gpu)
        L.check(lib.bd_pool2d_nhwc(xd.data_ptr(), 24, y.data_ptr(), 24, 3, 17, 15, 24, k, s, p, mode, cip, L.stream()), "bd_pool2d_nhwc")
        assert float((y.cpu() - want.permute(0, 2, 3, 1)).abs().max()) < 1e-6, (k, s, p, mode, cip)
    y = torch.empty(3, 24, device=gpu)
    L.check(lib.bd_global_avgpool_nhwc(xd.data_ptr(), 24, y.data_ptr(), 3, 17 * 15, 24, L.stream()), "bd_global_avgpool_nhwc")
    assert float((y.cpu() - F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)).abs().max()) < 1e-6
    for H, W in ((32, 32), (256, 256), (299, 299), (400, 301)):
        u8 = torch.randint(0, 256, (2, H, W, 3), generator=g, dtype=torch.uint8)
        want = 2 * F.interpolate((u8.float() / 255).permute(0, 3, 1, 2), size=(299, 299), mode="bilinear", align_corners=False) - 1
        for src, is_u8 in ((u8.to(gpu), 1), ((u8.float() / 255).to(gpu), 0)):
            y = torch.empty(2, 299, 299, 3, device=gpu)
            L.check(lib.bd_resize_bilinear_nhwc(src.data_ptr(), is_u8, y.data_ptr(), 2, H, W, 3, 299, 299, 2.0, -1.0, L.stream()), "resize")
            assert float((y.cpu() - want.permute(0, 2, 3, 1)).abs().max()) < 1e-5, (H, W, is_u8)     # a few ulps of the [-1, 1] range: ATen's CPU kernel orders the lerps differently


def test_fid_inception_pool3_vs_oracle(gpu):
    """f-3: FIDInceptionV3 (94 bd_conv2d_nhwc launches with folded BatchNorm, pools, resize) against the CPU restatement of
    pytorch_fid's network (oracle/inception_ref.py) on seeded random weights: pool3 features of 32 x 32 uint8 images (the CIFAR
    measure path) and of float 64 x 48 images, 1e-4 relative (fp32 both sides; 1e-3 is north_star's bar).  PARITY UNPINNED against
    pytorch_fid itself (package and weights absent)."""
    from baddiffusion_amd.inception import FIDInceptionV3
    from oracle import inception_ref as I
    P = I.gen_params(11)
    net = FIDInceptionV3(P, device=gpu, batch_size=3)
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (5, 32, 32, 3), generator=g, dtype=torch.uint8)
    got = net(u8.to(gpu))
    want = I.pool3_features(P, (u8.float() / 255).permute(0, 3, 1, 2))
    assert got.shape == (5, 2048) and relerr(got, want) < 1e-4, relerr(got, want)
    fl = torch.rand(2, 3, 64, 48, generator=g)
    assert relerr(net(fl.to(gpu)), I.pool3_features(P, fl)) < 1e-4
    with pytest.raises(RuntimeError, match="device tensors"):
        net(fl)
    bad = dict(P); bad["Mixed_5b.branch1x1.conv.weight"] = torch.zeros(64, 192, 3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        FIDInceptionV3(bad, device=gpu)


def test_measure_writes_fid_when_weights_are_mounted(gpu, tmp_path, monkeypatch):
    """baddiffusion.py measure() (reference :477-551, FID at :533): with BD_FID_WEIGHTS pointing at a state-dict file score.json
    carries a finite FID and no FID_reason; the features of both image sets (seed-shuffled dataset subset, written clean PNGs)
    match the ORACLE network, and the statistics + Frechet distance (pinned by G8) over them reproduce the number.  (The distance
    itself is not compared across networks: with n = 16 rows the 2048 x 2048 covariance product is rank-deficient and its matrix
    square root is ill-conditioned.)"""
    import dataclasses
    import baddiffusion as cli
    from baddiffusion_amd.dataset import DatasetLoader
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDIMScheduler
    from baddiffusion_amd.unet import unet_from_config
    from oracle import inception_ref as I
    cfg_net = dataclasses.replace(C.SMALL_CFGS["small"], sample_size=32)
    model = unet_from_config(cfg_net).cuda()
    model.load_state_dict(U.gen_params(cfg_net, 7))
    dsl = DatasetLoader(root=None, name=DatasetLoader.CIFAR10, batch_size=8, seed=0, device=gpu, num_images=24)
    dsl.set_poison(trigger_type="BOX_14", target_type="CORNER", clean_rate=1.0, poison_rate=0.25).prepare_dataset(mode="FIXED")
    P = I.gen_params(5)
    wfile = str(tmp_path / "pt_inception.pth")
    torch.save(P, wfile)
    monkeypatch.setenv("BD_FID_WEIGHTS", wfile)
    config = cli.TrainingConfig()
    config.output_dir = str(tmp_path / "out"); config.seed = 0; config.clip = False; config.sample_ep = None
    config.measure_sample_n = 16; config.eval_max_batch = 16
    os.makedirs(config.output_dir, exist_ok=True)
    pipe = DDIMPipeline(model, DDIMScheduler(num_train_timesteps=1000, clip_sample=False))
    score = cli.measure(config, dsl, "measure", pipe, rank=0, world=1)
    assert "FID_reason_noclip" not in score and np.isfinite(score["FID_noclip"])
    # the same two image sets through the oracle network
    # (round 5, ADVICE: the real set follows HF datasets' shuffle rule, np.random.default_rng(seed).permutation -- baddiffusion.py measure())
    order = torch.from_numpy(np.random.default_rng(config.seed).permutation(len(dsl))[:16].astype(np.int64))
    assert "default_rng" in score["FID_real_set_noclip"]
    real = dsl.device_images[dsl._rows()[order].to(gpu)].cpu()
    clean = torch.from_numpy(cli._png_dir_u8(os.path.join(config.output_dir, "measure", "clean_noclip"), 3))
    from baddiffusion_amd.inception import load_fid_weights
    net = load_fid_weights(device=gpu)
    fa, fb = net(real.to(gpu)), net(clean.to(gpu))
    assert relerr(fa, I.pool3_features(P, (real.float() / 255).permute(0, 3, 1, 2))) < 1e-4
    assert relerr(fb, I.pool3_features(P, (clean.float() / 255).permute(0, 3, 1, 2))) < 1e-4
    # statistics + Frechet distance (pinned by G8) over exactly these features give the number in score.json
    again = cli.fid_of_dirs(net, real.to(gpu), clean.to(gpu))
    assert abs(again - score["FID_noclip"]) <= 1e-6 * max(1.0, abs(again)), (again, score["FID_noclip"])


# ------------------------------------------------------------------------------------------------ 256 x 256 network: wide layers
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 256, 256, 128, 128), (3, 128, 64, 128, 128), (3, 64, 128, 256, 128), (4, 64, 64, 128, 256),
                                             (1, 8, 64, 128, 128)])
def test_conv_ps_wide_images_virtual_pixel_order(gpu, B, H, W, Cin, Cout):
    """Round 4: the vertical-tap-sharing kernels (conv_ps3_kernel forward / data gradient, conv_ps_wgrad3_kernel) on images WIDER than 32
    pixels -- walked strip by strip in virtual pixel order (conv_ps.hip: ps_v2r) -- against the exact-fp32 igemm path (1e-4) and the
    split-bf16 igemm path (same products, another summation order: 2e-6), square and non-square grids, every epilogue form.
    Reference ops: resnet.py:493,514 (the 3x3 convolutions of ResnetBlock2D at the 64 .. 256 pixel levels of DDPM-CELEBA-HQ-256)."""
    from baddiffusion_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.randn(B, H, W, Cin, generator=g).cuda(); dy = torch.randn(B, H, W, Cout, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda(); bias = torch.randn(Cout, generator=g).cuda()
    rb = torch.randn(B, Cout, generator=g).cuda(); res = torch.randn(B, H, W, Cout, generator=g).cuda()
    ws, xs, dys, wts = ops.split_bf16(w), ops.split_rows(x), ops.split_rows(dy), ops.split_wT(w)
    for kw in (dict(), dict(rowbias=rb), dict(residual=res, out_scale=0.7)):
        new = ops.conv3x3_ps(xs, ws, B, H, W, Cin, Cout, 1, bias=bias, **kw)
        assert relerr(new, ops.conv3x3_fwd(x, w, bias, mode=0, **kw)) < 1e-4, sorted(kw)
        assert relerr(new, ops.conv3x3_fwd(x, w, bias, mode=1, **kw)) < 2e-6, sorted(kw)
    newd = ops.conv3x3_ps(dys, wts, B, H, W, Cout, Cin, -1)
    assert relerr(newd, ops.conv3x3_dgrad(dy, w, (B, H, W, Cin), mode=0)) < 1e-4
    acc = ops.conv3x3_ps(dys, wts, B, H, W, Cout, Cin, -1, out=newd.clone(), accumulate=True)
    assert relerr(acc, 2 * newd) < 1e-6
    dw, db = ops.conv3x3_ps_wgrad(xs, dys, B, H, W, Cin, Cout, with_db=True)
    dw_ref, db_ref = ops.conv3x3_wgrad(x, dy, mode=0, with_db=True)
    assert relerr(dw, dw_ref) < 1e-4 and relerr(db, db_ref) < 1e-4
    assert torch.equal(dw, ops.conv3x3_ps_wgrad(xs, dys, B, H, W, Cin, Cout))          # fixed-order K split: run-to-run identical
    # per-tap check of the weight gradient (a wrong vertical neighbour would hide in a relative norm less well than here)
    for t in range(9):
        assert relerr(dw[:, t // 3, t % 3], dw_ref[:, t // 3, t % 3]) < 2e-4, t


@pytest.mark.parametrize("B,H,W,Cin,Cout,G", [(2, 128, 128, 128, 128, 32), (1, 256, 256, 128, 128, 32), (3, 64, 128, 128, 256, 32),
                                               (2, 64, 64, 256, 512, 32), (4, 128, 64, 128, 128, 4)])
def test_conv_ps_epilogue_groupnorm_statistics(gpu, B, H, W, Cin, Cout, G):
    """Round 4: the forward convolution's epilogue leaves the (sum, sum of squares) partials of the GroupNorm that reads its output next
    (resnet.py:591 norm2 behind conv1 + temb, :559 norm1 of the next block behind conv2 + shortcut) -- per sample, 256-pixel tile and group,
    fp64 [B, S, G, 2] -- and bd_gn_fwd, handed those, skips its statistics pass: y bit-identical to the call without them, statistics equal
    to the two-pass ones (1e-6) and to fp64 torch (1e-5); group sizes 4 .. 32 channels, both epilogue forms, wide (virtual pixel order) grids."""
    import ctypes as CT
    from baddiffusion_amd import _lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(B + H + Cout + G)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda(); bias = torch.randn(Cout, generator=g).cuda()
    rb = torch.randn(B, Cout, generator=g).cuda() * 3; res = torch.randn(B, H, W, Cout, generator=g).cuda() + 1.5
    ws_, xs = ops.split_bf16(w), ops.split_rows(x)
    HW = H * W
    assert lib.bd_gn_fwd_takes_stats(B, HW, Cout, G) == 1
    for kw in (dict(rowbias=rb), dict(residual=res, out_scale=0.7)):
        y0 = ops.conv3x3_ps(xs, ws_, B, H, W, Cin, Cout, 1, bias=bias, **kw)
        y, part = ops.conv3x3_ps(xs, ws_, B, H, W, Cin, Cout, 1, bias=bias, gn_groups=G, **kw)
        assert torch.equal(y, y0)
        assert part.shape[1] == HW // 256 and not torch.isnan(part).any()
        yg = y.double().reshape(B, HW, G, Cout // G)
        tot = part.sum(1)
        assert relerr(tot[..., 0], yg.sum((1, 3))) < 1e-6 and relerr(tot[..., 1], (yg * yg).sum((1, 3))) < 1e-6
        ga = torch.randn(Cout, generator=g).cuda(); be = torch.randn(Cout, generator=g).cuda()
        outs = []
        for stats in (None, part):
            st = torch.empty(2, B, G, device=gpu)
            wsb = ops.workspace(lib.bd_gn_workspace_bytes(B, Cout), x.device)
            z = torch.empty(B, HW, Cout, device=gpu)
            f = L.GnFwdDesc(B=B, HW=HW, C=Cout, G=G, eps=1e-6, silu=1, x=L.ptr(y), ldx=Cout, gamma=L.ptr(ga), beta=L.ptr(be), y=L.ptr(z),
                            ldy=Cout, mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(wsb), workspace_bytes=wsb.numel())
            if stats is not None:
                f.stats = L.ptr(stats); f.stats_splits = stats.shape[1]
            L.check(lib.bd_gn_fwd(CT.byref(f), L.stream()))
            outs.append((z, st))
        (z0, st0), (z1, st1) = outs
        assert relerr(st1, st0) < 1e-6 and relerr(z1, z0) < 1e-5
        xg = y.double().reshape(B, HW, G, Cout // G).permute(0, 2, 1, 3).reshape(B, G, -1)
        assert relerr(st1[0], xg.mean(-1)) < 1e-5 and relerr(st1[1], 1 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6)) < 1e-5
    # calls that cannot write them say so
    assert lib.bd_conv3x3_ps_gn_splits(2, 8, 8, 128, 128, 32) == 0
    with pytest.raises(ValueError):
        ops.conv3x3_ps(ops.split_rows(x[:, :8, :8].contiguous()), ws_, B, 8, 8, Cin, Cout, 1, bias=bias, rowbias=rb, gn_groups=G)


def test_groupnorm_large_image_path_round4(gpu):
    """Round 4, the GroupNorm kernels of the 256 x 256 network (grids that do not fit the single-pass form): the backward can leave
    per-sample (dgamma, dbeta) partials -- written by its group-finalize launch -- to ONE bd_gn_bwd_params launch per backward segment
    (1e-6 against the per-layer reduction, dx bit-identical).  Forward, statistics and dx against fp64 torch group_norm + silu
    (resnet.py:559,591): 1e-5."""
    import ctypes as CT
    import torch.nn.functional as F
    from baddiffusion_amd import _lib as L, ops
    lib = L.load()
    torch.manual_seed(9)
    G = 32
    for (B, HW, Cc) in [(2, 16384, 128), (1, 65536, 64), (3, 4096, 256)]:
        x = torch.randn(B, HW, Cc, device=gpu) * 2 + 0.5; dy = torch.randn(B, HW, Cc, device=gpu)
        ga = torch.randn(Cc, device=gpu); be = torch.randn(Cc, device=gpu)
        st = torch.empty(2, B, G, device=gpu)
        ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
        y = torch.empty_like(x)
        f = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=G, eps=1e-6, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), y=L.ptr(y), ldy=Cc,
                        mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel())
        L.check(lib.bd_gn_fwd(CT.byref(f), L.stream()))
        x64 = x.double().cpu().permute(0, 2, 1).reshape(B, Cc, HW, 1).requires_grad_(True)
        y64 = F.silu(F.group_norm(x64, G, ga.double().cpu(), be.double().cpu(), 1e-6))
        assert relerr(y, y64.reshape(B, Cc, HW).permute(0, 2, 1)) < 1e-5
        xg = x64.detach().reshape(B, G, -1)
        assert relerr(st[0], xg.mean(-1)) < 1e-5 and relerr(st[1], 1 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6)) < 1e-5
        y64.backward(dy.double().cpu().permute(0, 2, 1).reshape(B, Cc, HW, 1))
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
        assert torch.equal(dx0, dx1) and relerr(dx0, x64.grad.reshape(B, Cc, HW).permute(0, 2, 1)) < 1e-5
        assert torch.isnan(dg1).all() and torch.isnan(db1).all()
        items = (L.GnParamItem * 1)()
        items[0].partials = L.ptr(part1); items[0].C = Cc; items[0].dgamma = L.ptr(dg1); items[0].dbeta = L.ptr(db1)
        L.check(lib.bd_gn_bwd_params(items, 1, B, L.stream()))
        assert relerr(dg1, dg0) < 1e-6 and relerr(db1, db0) < 1e-6


@pytest.mark.parametrize("B,HW,Cc,G", [(2, 4096, 256, 2), (3, 4096, 96, 8), (2, 8192, 128, 32), (1, 4096, 1024, 32)])
def test_groupnorm_large_image_group_finalize_shapes(gpu, B, HW, Cc, G):
    """The one-pass group finalize of the large-image backward (grid = (B, G); round 4): group sizes from 4 to 128 channels -- 3 x 128 (plane,
    channel) sums exceed the 256 threads of a block, the looped form -- with the closed-form per-sample column sum of dx and the per-sample
    weight / bias gradient partials taken from the same totals.  dx, dgamma, dbeta against fp64 torch (normalization.py / resnet.py:559): 1e-5;
    the column sum against the sum of the dx it describes: 1e-4 of its scale."""
    import ctypes as CT
    import torch.nn.functional as F
    from baddiffusion_amd import _lib as L, ops
    lib = L.load()
    torch.manual_seed(B * 1000 + Cc + G)
    x = torch.randn(B, HW, Cc, device=gpu) * 1.5 - 0.3; dy = torch.randn(B, HW, Cc, device=gpu)
    ga = torch.randn(Cc, device=gpu); be = torch.randn(Cc, device=gpu)
    y, mean, rstd = ops.gn_fwd(x, ga, be, G, 1e-6, True)
    x64 = x.double().cpu().permute(0, 2, 1).reshape(B, Cc, HW, 1).requires_grad_(True)
    g64 = ga.double().cpu().requires_grad_(True); b64 = be.double().cpu().requires_grad_(True)
    F.silu(F.group_norm(x64, G, g64, b64, 1e-6)).backward(dy.double().cpu().permute(0, 2, 1).reshape(B, Cc, HW, 1))
    dx, dg, db, cs = ops.gn_bwd(x, ga, be, mean, rstd, dy, G, True, with_colsum=True)
    ref_dx = x64.grad.reshape(B, Cc, HW).permute(0, 2, 1)
    assert relerr(dx, ref_dx) < 1e-5 and relerr(dg, g64.grad) < 1e-5 and relerr(db, b64.grad) < 1e-5
    ref_cs = ref_dx.sum(1)
    assert float((cs.double().cpu() - ref_cs).abs().max()) < 1e-4 * float(ref_dx.abs().sum(1).max())
    # deferred form: the per-sample partials fold to the same gradients
    part = torch.empty(B, 2, Cc, device=gpu); dx2 = torch.empty_like(x)
    dg2 = torch.empty(Cc, device=gpu); db2 = torch.empty(Cc, device=gpu)
    ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
    d = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=G, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), mean=L.ptr(mean), rstd=L.ptr(rstd),
                    dy=L.ptr(dy), lddy=Cc, dx=L.ptr(dx2), lddx=Cc, accumulate_dx=0, dgamma=L.ptr(dg2), dbeta=L.ptr(db2), workspace=L.ptr(ws),
                    workspace_bytes=ws.numel(), param_partials=L.ptr(part))
    L.check(lib.bd_gn_bwd(CT.byref(d), L.stream()))
    items = (L.GnParamItem * 1)()
    items[0].partials = L.ptr(part); items[0].C = Cc; items[0].dgamma = L.ptr(dg2); items[0].dbeta = L.ptr(db2)
    L.check(lib.bd_gn_bwd_params(items, 1, B, L.stream()))
    assert torch.equal(dx, dx2) and relerr(dg2, dg) < 1e-6 and relerr(db2, db) < 1e-6


@pytest.mark.parametrize("topology,batch", [("google/ddpm-cifar10-32", 40), ("google/ddpm-ema-celebahq-256", 1)])
def test_round4_paths_equal_their_fallbacks(gpu, tmp_path, topology, batch):
    """Every plan / kernel path added in round 4 against the path it replaces, on whole networks (each knob is read once per process, hence
    the child processes): ready-split gradients (BD_GSPLIT), the shortcut data gradient on planes (BD_SHORTCUT_SP), the 3-channel convolutions on
    the matrix pipe (BD_THIN_MFMA), strip-order convolutions on wide images (BD_PS_V3_MAXW / BD_PS_WG3_MAXW = 32) and -- B = 40 holds >= 32 K
    pixels -- the two forward pipelines (BD_FWD_PIPES_MINPX).  Output and the full flat gradient agree to rounding (2e-5 / 1e-4 relative; the paths
    differ in summation order and, for the thin convolutions, in exact-fp32 vs split-bf16 products)."""
    base = dict(os.environ, BD_T_TOPOLOGY=topology, BD_T_BATCH=str(batch))
    off = dict(base, BD_GSPLIT="0", BD_SHORTCUT_SP="0", BD_THIN_MFMA="0", BD_PS_V3_MAXW="32", BD_PS_WG3_MAXW="32", BD_FWD_PIPES_MINPX=str(1 << 40))
    res = {}
    for tag, env in (("new", base), ("old", off)):
        f = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_knob_worker.py"), f], capture_output=True, text=True, timeout=600,
                           cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(f)
    assert relerr(res["new"]["out"], res["old"]["out"]) < 2e-5
    assert relerr(res["new"]["grad"], res["old"]["grad"]) < 1e-4
    assert torch.isfinite(res["new"]["grad"]).all()


# ------------------------------------------------------------------------------------------------ f-4: Adversarial Neuron Pruning
def test_anp_perturbed_unet_and_train_step_vs_reference(gpu, golden):
    """The ANP defense on the HIP path (baddiffusion_amd/anp.py: effective weights through bd_anp_apply, perturbation gradient through bd_anp_grad,
    clip + Adam + clamp as bd_sumsq / bd_adam_clip / bd_lincomb) against G13 -- vectors made by the reference's own PerturbConv2d /
    convert_model on its UNet2DModel (anp_model.py:490-514, anp_util.py:60-88, anp_defense.py:47-75, 136-160): the wrapped model with bn = (1, 0) and
    the disabled one ARE the model, perturbed forward, -loss, every bn gradient, the clip norm, bn parameters after Adam + clip_weight, backdoor MSE,
    forward after the step (no-grad path, and inside a sampling loop's static-weights block).  1e-3 as everywhere; both compute modes."""
    from oracle import unet_ref as U
    from tests.golden import cases as C
    from baddiffusion_amd import anp, unet as unet_mod
    from baddiffusion_amd.schedulers import DDPMScheduler
    g = golden("anp")
    cfg = C.SMALL_CFGS["small"]
    clean, trig, targ, t, eps = [v.cuda() for v in C.anp_inputs(cfg)]
    sched = DDPMScheduler(num_train_timesteps=1000)
    from baddiffusion_amd.loss import q_sample_diffuser
    xn = q_sample_diffuser(sched, clean, torch.zeros_like(clean), t, eps)[0]
    for mode in ("f32", "bf16x3"):
        m = unet_mod.unet_from_config(cfg).cuda()
        m.load_state_dict(U.gen_params(cfg, 7))
        m.set_compute_mode(mode)
        with torch.no_grad():
            plain = m(xn, t).sample
        assert relerr(plain, g["pred_plain"]) < 1e-3
        pm = anp.convert_model(m)
        bn_names = [str(n) for n in g["bn_names"]]             # the reference's named_parameters() order
        assert sorted(n for n, _ in pm.named_bn_parameters()) == sorted(bn_names)
        assert [p.requires_grad for p in pm.parameters()] == [False, True] or [n for n, p in pm.named_parameters() if p.requires_grad] == ["perturb"]
        with torch.no_grad():
            assert torch.equal(pm(xn, t).sample, plain)                      # bn = (1, 0): w * W is W, 1 * b + 0 is b -- bit-identical
            names = [str(n) for n in g["conv_names"]]
            pm.load_bn_state(C.anp_bn_init(list(zip(names, [int(c) for c in g["conv_couts"]]))))
            assert relerr(pm(xn, t).sample, g["pred_perturbed"]) < 1e-3
            anp.disable_perturb(pm)
            assert torch.equal(pm(xn, t).sample, plain)
            anp.enable_perturb(pm)
        tr = anp.AnpTrainer(pm, sched, anp.AnpConfig(learning_rate=C.ANP_LR, perturb_budget=C.ANP_BUDGET))
        logs = tr.step(clean, trig, targ, t, eps)
        assert abs(float(logs["loss"]) - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
        gref = torch.from_numpy(g["bn_grads"])
        bg = pm.bn_grads()
        gflat = torch.cat([bg[n].flatten() for n in bn_names])
        assert relerr(gflat, gref) < 1e-3, relerr(gflat, gref)
        assert abs(float(logs["grad_norm"]) - float(g["total_norm"])) < 1e-3 * float(g["total_norm"])
        bp = dict(pm.named_bn_parameters())
        after = torch.cat([bp[n].detach().flatten() for n in bn_names]).cpu()
        big = gref.abs() > 1e-2 * gref.abs().max()
        np.testing.assert_allclose(after[big].numpy(), g["bn_after"][big.numpy()], rtol=1e-4, atol=1e-5)
        assert float(after.abs().max()) <= C.ANP_BUDGET + 1e-7
        assert abs(float(logs["backdoor_mse"]) - float(g["backdoor_mse"])) < 1e-3 * float(g["backdoor_mse"])
        with torch.no_grad():
            o1 = pm(xn, t).sample
            with pm.static_weights():
                o2 = pm(xn, t).sample; o3 = pm(xn, t).sample
        assert relerr(o1, g["pred_after"]) < 2e-3 and torch.equal(o1, o2) and torch.equal(o2, o3)
        sd = pm.state_dict()
        assert all(k in sd for k in (names[0] + ".bn.weight", names[-1] + ".bn.bias", "conv_in.weight"))
    # the kernels themselves: a non-conv tensor is copied untouched, a conv row is scaled, the gradient contraction is the row dot product
    flat = m.flat.detach()
    T_ = pm.total_rows
    w = torch.randn(T_, device=gpu); b = torch.randn(T_, device=gpu)
    from baddiffusion_amd import ops
    eff = ops.anp_apply(flat, w, b, pm._items, T_)
    off, shape, _ = m._table["time_embedding.linear_1.weight"]
    assert torch.equal(eff[off: off + int(np.prod(shape))], flat[off: off + int(np.prod(shape))])
    it = pm._items.cpu().tolist()
    for (wo, bo, co, ln, po) in (it[0], it[3], it[-1]):
        ref = flat[wo: wo + co * ln].view(co, ln) * w[po: po + co, None]
        assert torch.equal(eff[wo: wo + co * ln].view(co, ln), ref)
        assert torch.allclose(eff[bo: bo + co], w[po: po + co] * flat[bo: bo + co] + b[po: po + co], rtol=1e-6, atol=1e-7)
    ge = torch.randn_like(flat)
    gw = torch.empty(T_, device=gpu); gb = torch.empty(T_, device=gpu)
    rn = torch.empty(T_, device=gpu)
    ops.anp_grad(flat, ge, pm._items, T_, gw, gb, pert_w=w, row_norm=rn)
    for (wo, bo, co, ln, po) in (it[0], it[3], it[-1]):
        ref = (ge[wo: wo + co * ln].view(co, ln).double() * flat[wo: wo + co * ln].view(co, ln).double()).sum(1) + ge[bo: bo + co].double() * flat[bo: bo + co].double()
        assert relerr(gw[po: po + co], ref) < 1e-5 and torch.equal(gb[po: po + co], ge[bo: bo + co])
        nref = w[po: po + co].double().abs() * torch.sqrt((ge[wo: wo + co * ln].view(co, ln).double() ** 2).sum(1) + ge[bo: bo + co].double() ** 2)
        assert relerr(rn[po: po + co], nref) < 1e-5


def test_anp_defense_cli_loop_end_to_end(gpu, tmp_path):
    """anp_defense.py (the reference's anp_defense.py:113-186 + anp_config.py + anp_util.py:149-260): flags / output-dir naming / args.json of the
    backdoor run, the all-poisoned data loader, two epochs of the loop on the small UNet (32 x 32), per-epoch sample grids, score.json with one
    (MSE, SSIM) per measure() call, the saved bn parameters inside the budget -- and the loop does what ANP is for: the clean loss it MAXIMISES goes up."""
    import dataclasses, json, os
    from PIL import Image
    import anp_defense as cli
    from oracle import unet_ref as U
    from tests.golden import cases as C
    from baddiffusion_amd import anp
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.unet import unet_from_config
    ck = tmp_path / "res_backdoored"
    os.makedirs(ck)
    json.dump({"trigger": "BOX_14", "target": "CORNER", "poison_rate": 0.1, "dataset": "CIFAR10"}, open(ck / "args.json", "w"))
    config = cli.get_config(["--ckpt", str(ck), "--epoch", "2", "-lr", "0.05", "-pb", "1.5", "--output_dir", str(tmp_path / "out"), "--batch", "8",
                             "--tag", "t"])
    assert os.path.basename(config.output_dir) == "res_anp_2_lr0.05_pb1.5_t_res_backdoored" and os.path.exists(os.path.join(config.output_dir, "config.json"))
    assert (config.trigger, config.target, config.dataset) == ("BOX_14", "CORNER", "CIFAR10")
    config.num_images = 16; config.eval_sample_n = 4; config.measure_sample_n = 6; config.eval_max_batch = 4; config.dataset_path = None
    dsl = cli.get_data_loader(config, device=gpu)
    b = next(iter(dsl.get_dataloader()))
    assert not bool(b["is_clean"].any()) and b["image"].shape == (8, 3, 32, 32)      # clean_rate 0, poison_rate 1 (anp_util.py:152)
    cfg_net = dataclasses.replace(C.SMALL_CFGS["small"], sample_size=32)
    model = unet_from_config(cfg_net).cuda()
    model.load_state_dict(U.gen_params(cfg_net, 7))
    pm = anp.convert_model(model)
    sched = DDPMScheduler(num_train_timesteps=1000)
    sched.set_timesteps(1000)
    import baddiffusion_amd.pipelines as P_
    keep = P_.DDPMPipeline.__call__
    # (1000-step chains on 10 images would dominate the test: the loop's pipelines sample with 5 steps here)
    cli.DDPMPipeline = type("FastDDPM", (P_.DDPMPipeline,), {"__call__": lambda self, *a, **k: keep(self, *a, **{**k, "num_inference_steps": 5})})
    try:
        pipe, hist = cli.train_loop(config, pm, sched, dsl, log=lambda *_: None)
    finally:
        cli.DDPMPipeline = P_.DDPMPipeline
    assert len(hist) == 4 and all(np.isfinite(h["loss"]) and np.isfinite(h["backdoor_mse"]) for h in hist)
    assert hist[-1]["clean_mse"] > hist[0]["clean_mse"]                              # gradient ASCENT on the clean loss
    sc = json.load(open(os.path.join(config.output_dir, "score.json")))
    assert sc["epoch"] == [1, 2, 2] and len(sc["MSE"]) == 3 and all(0 <= v <= 1 for v in sc["MSE"]) and all(-1 <= v <= 1 for v in sc["SSIM"])
    for name in ("0000.png", "0001.png", "final.png"):
        assert Image.open(os.path.join(config.output_dir, "samples", name)).size == (2 * 32, 2 * 32)
    bn = torch.load(os.path.join(config.output_dir, "anp_bn.pt"))
    assert len(bn) == 2 * len(pm.conv_names) and max(float(v.abs().max()) for v in bn.values()) <= 1.5 + 1e-6
    assert any(float((v - 1).abs().max()) > 0.05 for k, v in bn.items() if k.endswith("bn.weight"))


def test_bench_anp_workload_line(gpu):
    """`python bench.py --workload anp`: the side-measurement line of the ANP defense batch (SURVEY f-4) -- one JSON line, the contract's keys,
    a finite loss that is the NEGATIVE clean MSE (anp_defense.py:147), the CPU baseline object when asked for."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "anp", "--steps", "3", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and abs(d["value"] - 8 * 1000 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["final_loss"] < 0 and np.isfinite(d["final_backdoor_mse"]) and d["config"]["bn_parameters"] > 0


def test_anp_step_cifar_topology_vs_oracle(gpu):
    """The ANP batch on the full DDPM-CIFAR10-32 topology (35.7 M parameters, 65 wrapped convolutions, every kernel path of the CIFAR step: LDS-DMA
    convolutions, phase-decomposed up / down sampling, split-plane attention) against oracle/anp_ref.py (itself pinned on G13) on the same seeded
    inputs, two consecutive steps (the second one sees Adam moments and a clamped state): loss, bn gradients, clip norm, bn parameters, backdoor MSE."""
    from oracle import anp_ref as A, sched_ref
    from oracle import unet_ref as U
    from tests.golden import cases as C
    from baddiffusion_amd import anp, unet as unet_mod
    from baddiffusion_amd.schedulers import DDPMScheduler
    cfg = U.CIFAR10_32
    P = U.gen_params(cfg, 3)
    B = 4
    g = torch.Generator().manual_seed(11)
    clean = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1; trig = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    targ = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1; eps = torch.randn(B, 3, 32, 32, generator=g)
    t = torch.tensor([7, 450, 820, 999])
    names = A.conv_names(cfg)
    bn = C.anp_bn_init([(n, P[n + ".weight"].shape[0]) for n in names])
    _, a, ac = sched_ref.make_tables()
    m = unet_mod.unet_from_config(cfg).cuda()
    m.load_state_dict(P)
    pm = anp.convert_model(m)
    assert sorted(n for n, _, _ in pm.conv_names) == sorted(names) and len(names) == 65
    pm.load_bn_state(bn)
    tr = anp.AnpTrainer(pm, DDPMScheduler(num_train_timesteps=1000), anp.AnpConfig(learning_rate=C.ANP_LR, perturb_budget=C.ANP_BUDGET))
    state = {}
    for step in (1, 2):
        loss, G, norm, bn, state, bm = A.anp_step(cfg, P, bn, state, a, ac, clean, trig, targ, t, eps, C.ANP_LR, step, C.ANP_BUDGET)
        logs = tr.step(clean.cuda(), trig.cuda(), targ.cuda(), t.cuda(), eps.cuda())
        assert abs(float(logs["loss"]) - float(loss)) < 1e-3 * abs(float(loss)), step
        bg = pm.bn_grads()
        keys = [n + sfx for n in names for sfx in (".bn.weight", ".bn.bias")]
        gref = torch.cat([G[k].flatten() for k in keys])
        assert relerr(torch.cat([bg[k].flatten() for k in keys]), gref) < 1e-3, step
        assert abs(float(logs["grad_norm"]) - float(norm)) < 1e-3 * float(norm), step
        bp = dict(pm.named_bn_parameters())
        big = gref.abs() > 1e-2 * gref.abs().max()
        got = torch.cat([bp[k].detach().flatten() for k in keys]).cpu()
        want = torch.cat([bn[k].flatten() for k in keys])
        np.testing.assert_allclose(got[big].numpy(), want[big].numpy(), rtol=2e-4, atol=2e-5)
        assert abs(float(logs["backdoor_mse"]) - float(bm)) < 1e-3 * float(bm), step
        pm.load_bn_state(bn)          # continue both sides from the oracle's state (sign flips of noise-level gradients do not accumulate)
