"""Pin the CPU oracle: (1) against the vectors captured from the imported reference
(tests/golden/*.npz, made by tests/golden/make_golden.py), (2) against the known answers the
reference's own diffusers tests hold (file:line cited per test).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import backdoor_ref as BD
from oracle import loss_ref, metrics_ref, sched_ref, train_ref
from oracle import unet_ref as U
from tests.golden import cases as C

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


# ------------------------------------------------------------------ G1 + upstream scheduler KATs
def test_tables_bit_exact(golden):
    g = golden("sched")
    b, a, ac = sched_ref.make_tables()
    assert np.array_equal(b.numpy(), g["betas"]) and np.array_equal(a.numpy(), g["alphas"])
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])


def test_ddpm_variance_kat():
    # diffusers/tests/schedulers/test_scheduler_ddpm.py:62-69
    _, _, ac = sched_ref.make_tables()
    assert abs(float(sched_ref.ddpm_variance(ac, 0, -1)) - 0.0) < 1e-5
    assert abs(float(sched_ref.ddpm_variance(ac, 487, 486)) - 0.00979) < 1e-5
    assert abs(float(sched_ref.ddpm_variance(ac, 999, 998)) - 0.02) < 1e-5


def _dummy_sample_deter():
    # diffusers/tests/schedulers/test_schedulers.py:222-234
    n = 4 * 3 * 8 * 8
    s = torch.arange(n).reshape(3, 8, 8, 4) / n
    return s.permute(3, 0, 1, 2)


def test_ddpm_full_loop_kat():
    # test_scheduler_ddpm.py:71-100 -> sum 258.9606, mean 0.3372 (fixed_small, clip_sample=True)
    _, _, ac = sched_ref.make_tables()
    sample = _dummy_sample_deter()
    gen = torch.manual_seed(0)
    for t in reversed(range(1000)):
        residual = sample * t / (t + 1)                                  # test_schedulers.py:239-243
        noise = torch.randn(sample.shape, generator=gen) if t > 0 else None
        sample, _ = sched_ref.ddpm_step(ac, residual, t, sample, noise)
    assert abs(float(sample.abs().sum()) - 258.9606) < 1e-2
    assert abs(float(sample.abs().mean()) - 0.3372) < 1e-3


def test_ddim_variance_and_loop_kat():
    # test_scheduler_ddim.py:94-113
    _, _, ac = sched_ref.make_tables()

    def var(t, p):
        return float(((1 - ac[p]) / (1 - ac[t])) * (1 - ac[t] / ac[p]))
    assert abs(var(420, 400) - 0.14771) < 1e-5 and abs(var(980, 960) - 0.32460) < 1e-5
    sample = _dummy_sample_deter()
    for t in sched_ref.ddim_timesteps(10):
        residual = sample * int(t) / (int(t) + 1)
        sample, _ = sched_ref.ddim_step(ac, residual, int(t), sample, 10)
    assert abs(float(sample.abs().sum()) - 172.0067) < 1e-2
    assert abs(float(sample.abs().mean()) - 0.223967) < 1e-3


def test_ddim_steps_offset_kat():
    # test_scheduler_ddim.py:46-54
    assert list(sched_ref.ddim_timesteps(5, steps_offset=1)) == [801, 601, 401, 201, 1]


def test_sched_steps_vs_reference(golden):
    g = golden("sched")
    _, _, ac = sched_ref.make_tables()
    x, eps, z = C.sched_inputs()
    for vt in ("fixed_small", "fixed_large"):
        for clip in (True, False):
            for t in C.DDPM_TS:
                prev, x0 = sched_ref.ddpm_step(ac, eps, t, x, z, variance_type=vt, clip_sample=clip)
                close(prev, g[f"ddpm_{vt}_{int(clip)}_{t}_prev"], 1e-6, 1e-6)
                close(x0, g[f"ddpm_{vt}_{int(clip)}_{t}_x0"], 1e-6, 1e-6)
    prev, _ = sched_ref.ddpm_step(ac, eps, 500, x, z, clip_sample=False, clip_defense=True, clip_defense_range=0.5)
    close(prev, g["ddpm_clipdef_500_prev"], 1e-6, 1e-6)
    assert np.array_equal(sched_ref.ddpm_timesteps(50), g["ddpm_ts50"])
    assert np.array_equal(sched_ref.ddim_timesteps(50), g["ddim_ts50"])
    for clip in (True, False):
        for t in C.DDIM_TS:
            prev, _ = sched_ref.ddim_step(ac, eps, t, x, 50, clip_sample=clip)
            close(prev, g[f"ddim_{int(clip)}_{t}_prev"], 1e-6, 1e-6)
        prev, _ = sched_ref.ddim_step(ac, eps, 500, x, 50, eta=0.5, noise=z, clip_sample=clip)
        close(prev, g[f"ddim_{int(clip)}_500_eta_prev"], 1e-6, 1e-6)
    close(sched_ref.add_noise(ac, x, eps, torch.tensor([3, 977])), g["add_noise"], 1e-6, 1e-7)


# ------------------------------------------------------------------ G2
def test_qsample_vs_reference(golden):
    g = golden("qsample")
    _, a, ac = sched_ref.make_tables()
    x0, R, eps, t = C.qsample_inputs()
    xn, tgt = loss_ref.q_sample(a, ac, x0, R, t, eps)
    assert np.array_equal(xn.numpy(), g["x_noisy"]) and np.array_equal(tgt.numpy(), g["target"])
    # rho(t) for t in {0,10,500,999}
    rho = (1 - a[t] ** 0.5) * (1 - ac[t]) ** 0.5 / (1 - a[t])
    close(rho, [0.0050004, 0.0234175, 0.4813690, 0.5025141], 1e-5, 1e-7)
    model = lambda x, tt: 0.5 * x - 0.01 * tt.reshape(-1, 1, 1, 1).float() / 1000
    for lt in ("l2", "l1", "huber"):
        close(loss_ref.p_losses(a, ac, model, x0, R, t, eps, lt), g[lt], 1e-6, 1e-7)


# ------------------------------------------------------------------ G3
def test_backdoor_vs_reference(golden):
    g = golden("backdoor")
    for S in (32, 256):
        for trig in ("BOX_4", "BOX_8", "BOX_11", "BOX_14", "BOX_18", "SM_BOX", "NONE"):
            t = BD.get_trigger(trig, 3, S)
            assert np.array_equal(t.numpy(), g[f"trig_{trig}_{S}"])
            m = BD.get_mask(t)
            assert m.dtype == torch.int64 and np.array_equal(m.numpy(), g[f"mask_{trig}_{S}"])     # bit-exact
            if S == 32 or trig == "BOX_14":
                for tg in ("CORNER", "TRIGGER", "SHIFT"):
                    assert np.array_equal(BD.get_target(tg, t).numpy(), g[f"tgt_{tg}_{trig}_{S}"])
    assert int((BD.get_trigger("BOX_14", 3, 32) > -1).sum()) == 588                              # SURVEY a-1
    img = C.backdoor_images()
    trig = BD.get_trigger("BOX_14", 3, 32)
    R, x0 = BD.make_batch(img, torch.ones(4, dtype=torch.bool), trig, BD.get_target("CORNER", trig))
    assert np.array_equal(R.numpy(), g["blend_BOX_14_32"])
    R, x0 = BD.make_batch(img, torch.tensor([True, False, False, True]), trig, BD.get_target("CORNER", trig))
    assert float(R[1].abs().max()) == 0.0 and torch.equal(x0[1], img[1])
    assert np.array_equal(BD.normalize(torch.arange(256, dtype=torch.float32) / 255.0).numpy(), g["normalize_u8"])


# ------------------------------------------------------------------ G4 + upstream embedding KAT
def test_timestep_embedding(golden):
    g = golden("temb")
    t = torch.tensor(C.TEMB_TS)
    close(U.timestep_embedding(t, 128, False, 1), g["cifar"], 1e-6, 1e-6)
    close(U.timestep_embedding(t, 128, True, 0), g["default"], 1e-6, 1e-6)


# ------------------------------------------------------------------ G5 modules (fwd + bwd)
def _grads_ok(P, g, name, rtol=2e-4):
    for k, p in P.items():
        gn = float(p.grad.double().norm())
        assert abs(gn - float(g[f"{name}_gn_{k}"])) <= rtol * max(1.0, gn), (name, k)
        close(p.grad.flatten()[:8], g[f"{name}_g8_{k}"], 1e-3, 1e-5)


def test_modules_vs_reference(golden):
    g = golden("modules")
    for name in C.RESNET_CASES:
        P = {k: v.requires_grad_(True) for k, v in C.module_params(name).items()}
        x, temb, dy = C.resnet_inputs(name)
        x.requires_grad_(True); temb.requires_grad_(True)
        y = U.resnet_block(P, "", x, temb, 32, 1e-6)
        y.backward(dy)
        close(y, g[f"{name}_y"], 1e-5, 1e-5); close(x.grad, g[f"{name}_dx"], 1e-4, 1e-5)
        close(temb.grad, g[f"{name}_dtemb"], 1e-4, 1e-5)
        _grads_ok(P, g, name)
    for name, (Cc, hw, hd) in C.ATTN_CASES.items():
        P = {k: v.requires_grad_(True) for k, v in C.module_params(name).items()}
        x, dy = C.attn_inputs(name); x.requires_grad_(True)
        y = U.attention_block(P, "", x, 32, 1e-6, hd)
        y.backward(dy)
        close(y, g[f"{name}_y"], 1e-5, 1e-5); close(x.grad, g[f"{name}_dx"], 1e-4, 1e-5)
        _grads_ok(P, g, name)
    for name, (Cc, hw, pad) in C.DOWN_CASES.items():
        P = {k: v.requires_grad_(True) for k, v in C.module_params(name).items()}
        x, dy = C.down_inputs(name); x.requires_grad_(True)
        y = U.downsample(P, "", x, pad); y.backward(dy)
        close(y, g[f"{name}_y"], 1e-5, 1e-5); close(x.grad, g[f"{name}_dx"], 1e-4, 1e-5)
        _grads_ok(P, g, name)
    for name in C.UP_CASES:
        P = {k: v.requires_grad_(True) for k, v in C.module_params(name).items()}
        x, dy = C.up_inputs(name); x.requires_grad_(True)
        y = U.upsample(P, "", x); y.backward(dy)
        close(y, g[f"{name}_y"], 1e-5, 1e-5); close(x.grad, g[f"{name}_dx"], 1e-4, 1e-5)
        _grads_ok(P, g, name)


def test_oracle_attention_at_16x16_vs_reference(golden):
    """G11: the AttentionBlock shapes the split-plane GPU path takes (256 tokens, head dim 256; one and two heads) -- the oracle's block
    against the reference's module, so the GPU test's comparison target is pinned on the CPU side too"""
    g = golden("attn_planes")
    for name, (Cc, hw, hd) in C.ATTN_SP_CASES.items():
        P = {k: v.requires_grad_(True) for k, v in C.module_params(name).items()}
        x, dy = C.attn_inputs(name); x.requires_grad_(True)
        y = U.attention_block(P, "", x, 32, 1e-6, hd)
        y.backward(dy)
        close(y, g[f"{name}_y"], 1e-5, 1e-5); close(x.grad, g[f"{name}_dx"], 1e-4, 1e-5)
        _grads_ok(P, g, name)


# ------------------------------------------------------------------ G6 / G7 whole UNet + one train step
def _train_step_check(cfg, seed, B, tag, g, lr=2e-4):
    _, a, ac = sched_ref.make_tables()
    P = U.gen_params(cfg, seed)
    x0, R, t, eps = C.train_inputs(cfg, B)
    xn, _ = loss_ref.q_sample(a, ac, x0, R, t, eps)
    with torch.no_grad():
        pred = U.unet_forward(cfg, P, xn, t)
    close(pred, g[f"{tag}_pred"], 1e-4, 2e-5)
    loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
    close(loss, g[f"{tag}_loss"], 1e-5, 1e-6)
    names = [str(s) for s in g[f"{tag}_names"]]
    assert set(names) == set(P.keys())
    gn = np.array([float(G[k].double().norm()) for k in names])
    np.testing.assert_allclose(gn, g[f"{tag}_gradnorms"], rtol=2e-3, atol=1e-7)
    g8 = np.stack([np.pad(G[k].flatten()[:8].numpy(), (0, max(0, 8 - G[k].numel()))) for k in names])
    np.testing.assert_allclose(g8, g[f"{tag}_grad8"], rtol=5e-3, atol=2e-6)
    newP, _, norm = train_ref.clip_and_adam(P, G, {}, lr, 1)
    close(norm, g[f"{tag}_total_norm"], 1e-4, 1e-6)
    p8 = np.stack([np.pad(newP[k].flatten()[:8].numpy(), (0, max(0, 8 - newP[k].numel()))) for k in names])
    np.testing.assert_allclose(p8, g[f"{tag}_p8_after"], rtol=1e-4, atol=2e-6)


def test_small_unet_train_step(golden):
    for tag, cfg in C.SMALL_CFGS.items():
        _train_step_check(cfg, 7, 2, tag, golden("unet_small"))


def test_cifar_unet_train_step(golden):
    torch.set_num_threads(8)
    _train_step_check(U.CIFAR10_32, 0, 2, "cifar", golden("unet_cifar"))


def test_pipelines_vs_reference(golden):
    g = golden("unet_small")
    cfg = C.SMALL_CFGS["small"]
    P = U.gen_params(cfg, 7)
    _, _, ac = sched_ref.make_tables()
    init = C.pipeline_init(cfg)
    with torch.no_grad():
        for clip in (True, False):
            for vt in ("fixed_small", "fixed_large"):
                gen = torch.Generator().manual_seed(C.PIPE_SEED)
                x = init.clone()
                for t in sched_ref.ddpm_timesteps(3):
                    e = U.unet_forward(cfg, P, x, int(t))
                    z = torch.randn(x.shape, generator=gen) if t > 0 else None      # randn_tensor order
                    x, _ = sched_ref.ddpm_step(ac, e, int(t), x, z, num_inference_steps=3,
                                               variance_type=vt, clip_sample=clip)
                close(sched_ref.to_image(x), g[f"ddpm3_{int(clip)}_{vt}"], 1e-4, 2e-5)
            x = init.clone()
            for t in sched_ref.ddim_timesteps(4):
                e = U.unet_forward(cfg, P, x, int(t))
                x, _ = sched_ref.ddim_step(ac, e, int(t), x, 4, clip_sample=clip)
            close(sched_ref.to_image(x), g[f"ddim4_{int(clip)}"], 1e-4, 2e-5)


def test_cosine_lr_closed_form():
    # optimization.py:134-138 ; baddiffusion.py:327-331 (warmup 500)
    T_ = 469 * 50
    f = lambda s: train_ref.cosine_lr_lambda(s, 500, T_)
    assert f(0) == 0.0 and abs(f(1) - 1 / 500) < 1e-12 and abs(f(499) - 499 / 500) < 1e-12
    assert f(500) == 1.0 and abs(f(T_)) < 1e-12
    mid = 500 + (T_ - 500) // 2
    assert abs(f(mid) - 0.5 * (1 + math.cos(math.pi * (mid - 500) / (T_ - 500)))) < 1e-12


def test_g8_frechet_distance_and_statistics(golden):
    """fid_score.py:150-230 restatement (oracle) and the product's device-side accumulation against the reference's
    own outputs, incl. the rank-deficient case (fewer samples than dimensions -> singular product -> eps retry)."""
    from oracle import metrics_ref
    from baddiffusion_amd import metrics
    g = golden("fid")
    for tag in ("d64", "d16", "rank_deficient"):
        for a, mu, sg in ((g[f"{tag}_a1"], g[f"{tag}_mu1"], g[f"{tag}_sigma1"]), (g[f"{tag}_a2"], g[f"{tag}_mu2"], g[f"{tag}_sigma2"])):
            m, s = metrics_ref.activation_statistics(a)
            np.testing.assert_allclose(m, mu, rtol=0, atol=1e-13); np.testing.assert_allclose(s, sg, rtol=1e-12, atol=1e-13)
            acc = metrics.ActivationStats(a.shape[1])
            for lo in range(0, a.shape[0], 37):            # batch by batch, like the measure loop
                acc.update(torch.from_numpy(a[lo:lo + 37]))
            m2, s2 = acc.finalize()
            np.testing.assert_allclose(m2, mu, rtol=0, atol=1e-12); np.testing.assert_allclose(s2, sg, rtol=1e-9, atol=1e-11)
        ref = float(g[f"{tag}_fid"])
        args = (g[f"{tag}_mu1"], g[f"{tag}_sigma1"], g[f"{tag}_mu2"], g[f"{tag}_sigma2"])
        assert abs(metrics_ref.frechet_distance(*args) - ref) <= 1e-9 * abs(ref)
        assert abs(metrics.frechet_distance(*args) - ref) <= 1e-9 * abs(ref)
    with pytest.raises(ValueError):
        metrics.frechet_distance(np.zeros(3), np.eye(3), np.zeros(4), np.eye(4))


def test_ssim_oracle_properties():
    """oracle.metrics_ref.ssim_ref (torchmetrics defaults restated with scipy, parity unpinned against torchmetrics itself):
    identity = 1, symmetric, drops with noise, closed form for two constant images; mse_ref = mean squared difference"""
    gen = torch.Generator().manual_seed(0)
    a = torch.rand(4, 3, 32, 32, generator=gen)
    b = (a + 0.1 * torch.randn(a.shape, generator=gen)).clamp(0, 1)
    c = (a + 0.3 * torch.randn(a.shape, generator=gen)).clamp(0, 1)
    S = lambda x, y: metrics_ref.ssim_ref(x.numpy(), y.numpy())
    assert abs(S(a, a) - 1.0) < 1e-12
    sab, sba, sac = S(a, b), S(b, a), S(a, c)
    assert abs(sab - sba) < 1e-12 and 0 < sac < sab < 1
    u, v = 0.2, 0.7     # two constant images: variances vanish, SSIM = (2uv + c1) / (u^2 + v^2 + c1)
    s = S(torch.full((1, 3, 32, 32), u), torch.full((1, 3, 32, 32), v))
    assert abs(s - (2 * u * v + 1e-4) / (u * u + v * v + 1e-4)) < 1e-6       # inputs are fp32 roundings of u, v
    assert abs(metrics_ref.mse_ref(a.numpy(), b.numpy()) - float(((a.double() - b.double()) ** 2).mean())) < 1e-15


def test_ssim_oracle_against_direct_window_sum():
    """ssim_ref's scipy.ndimage blur against a literal evaluation of the definition on a small image: reflect-pad by 5, slide the
    normalised 11x11 Gaussian (sigma 1.5) as explicit loops, crop the 5-pixel border, mean."""
    rng = np.random.default_rng(3)
    a = rng.random((1, 1, 14, 13)); b = np.clip(a + 0.15 * rng.standard_normal(a.shape), 0, 1)
    x = np.arange(11) - 5.0
    g = np.exp(-(x / 1.5) ** 2 / 2); g /= g.sum()
    w = np.outer(g, g)
    pa, pb = (np.pad(v[0, 0], 5, mode="reflect") for v in (a, b))
    H, W = a.shape[-2:]
    vals = []
    for i in range(5, H - 5):
        for j in range(5, W - 5):
            wa, wb = pa[i:i + 11, j:j + 11], pb[i:i + 11, j:j + 11]
            ma, mb = (w * wa).sum(), (w * wb).sum()
            va, vb, cab = (w * wa * wa).sum() - ma * ma, (w * wb * wb).sum() - mb * mb, (w * wa * wb).sum() - ma * mb
            vals.append(((2 * ma * mb + 1e-4) * (2 * cab + 9e-4)) / ((ma * ma + mb * mb + 1e-4) * (va + vb + 9e-4)))
    assert abs(metrics_ref.ssim_ref(a, b) - float(np.mean(vals))) < 1e-9


def test_product_metrics_have_no_cpu_path():
    """metrics.mse / metrics.ssim are HIP kernels: CPU tensors must raise, not fall back"""
    from baddiffusion_amd import metrics
    a = torch.rand(2, 3, 32, 32)
    with pytest.raises(RuntimeError):
        metrics.ssim(a, a)
    with pytest.raises(RuntimeError):
        metrics.mse(a, a)


def test_oracle_matches_full_size_reference_vectors(golden):
    """G10 on the CPU side: the oracle's forward on the four recorded rows of the B = 128 CIFAR batch (samples are independent)
    and on the real 256x256 CelebA-HQ network, against what the imported reference produced."""
    g = golden("full_size")
    _, a, ac = sched_ref.make_tables()
    cfg = U.CIFAR10_32
    x0, R, t, eps = C.train_inputs(cfg, 128)
    rows = list(C.FULL_ROWS)
    xn, _ = loss_ref.q_sample(a, ac, x0[rows], R[rows], t[rows], eps[rows])
    with torch.no_grad():
        pred = U.unet_forward(cfg, U.gen_params(cfg, 0), xn, t[rows])
    ref = torch.from_numpy(g["cifar128_pred_rows"])
    assert float((pred - ref).norm() / ref.norm()) < 1e-5
    cfg = U.CELEBA_HQ_256
    x, tt, _ = C.celeba_full_inputs()
    with torch.no_grad():
        out = U.unet_forward(cfg, U.gen_params(cfg, 5), x, tt)
    ref = torch.from_numpy(g["celeba256_out_slices"])
    assert float((out[0, :, ::16, ::16] - ref).norm() / ref.norm()) < 1e-5
    assert abs(float((out.double() ** 2).sum()) - float(g["celeba256_out_sumsq"])) < 1e-5 * float(g["celeba256_out_sumsq"])
    # G12: one sample of the B = 4 batch (samples are independent through the network)
    g4 = golden("full_size_b4")
    x4, t4, _ = C.celeba_b4_inputs()
    with torch.no_grad():
        out = U.unet_forward(cfg, U.gen_params(cfg, 5), x4[2:3], t4[2:3])
    ref = torch.from_numpy(g4["celeba256b4_out_slices"][2])
    assert float((out[0, :, ::16, ::16] - ref).norm() / ref.norm()) < 1e-5
    assert abs(float((out.double() ** 2).sum()) - float(g4["celeba256b4_out_sumsq"][2])) < 1e-5 * float(g4["celeba256b4_out_sumsq"][2])


def test_pndm_oracle_matches_reference_chains(golden):
    """G9: PNDMScheduler timesteps (also after from_config of a DPM-Solver++ / UniPC scheduler, which is what model.py:598-630
    effectively selects) and full chains with the stand-in model, incl. the skip_prk_steps start-up branches."""
    g = golden("pndm")
    _, _, ac = sched_ref.make_tables()
    x0 = C.pndm_init()
    for n in C.PNDM_STEPS:
        r = sched_ref.PNDMRef(ac, n)
        assert np.array_equal(r.timesteps, g[f"timesteps_{n}"])
        x = x0.clone()
        for i, t in enumerate(r.timesteps):
            x = r.step(C.pndm_fake_model(x, t), t, x)
            ref = torch.from_numpy(g[f"chain_{n}"][i])
            assert float((x - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (n, i)   # fp32 rounding over up to 59 steps
    for name in ("dpmpp2", "unipc"):
        assert np.array_equal(sched_ref.PNDMRef(ac, 20).timesteps, g[f"converted_{name}_timesteps_20"])
        assert int(g[f"converted_{name}_skip_prk"]) == 0 and float(g[f"converted_{name}_final_alpha"]) == float(ac[0])
    r = sched_ref.PNDMRef(ac, 10, skip_prk_steps=True)
    assert np.array_equal(r.timesteps, g["skip_timesteps_10"])
    x = x0.clone()
    for i, t in enumerate(r.timesteps):
        x = r.step(C.pndm_fake_model(x, t), t, x)
        ref = torch.from_numpy(g["skip_chain_10"][i])
        assert float((x - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), i


# ------------------------------------------------------------------ G13: Adversarial Neuron Pruning (f-4)
def test_anp_oracle_matches_reference_vectors(golden):
    """oracle/anp_ref.py against what the reference's own PerturbConv2d / convert_model computed (tests/golden/anp.npz): wrapped model with
    bn = (1, 0) IS the model, perturbed / disabled forward, -loss, every bn gradient, the clip norm, bn parameters after Adam + clip_weight,
    the backdoor MSE.  Also: parameter naming and order ('bn' parameters as anp_util.py:133 selects them)."""
    from oracle import anp_ref as A
    g = golden("anp")
    cfg = C.SMALL_CFGS["small"]
    P = U.gen_params(cfg, 7)
    names = [str(n) for n in g["conv_names"]]              # the reference's named_parameters() order (up_blocks before mid_block)
    assert sorted(names) == sorted(A.conv_names(cfg))      # the same layers: every nn.Conv2d, nothing else
    assert [P[n + ".weight"].shape[0] for n in names] == [int(c) for c in g["conv_couts"]]
    bn_names = [str(n) for n in g["bn_names"]]
    assert bn_names == [n + s for n in names for s in (".bn.weight", ".bn.bias")]
    _, a, ac = sched_ref.make_tables()
    clean, trig, targ, t, eps = C.anp_inputs(cfg)
    xn, _ = loss_ref.q_sample(a, ac, clean, torch.zeros_like(clean), t, eps)
    with torch.no_grad():
        plain = U.unet_forward(cfg, P, xn, t)
        close(plain, g["pred_plain"], rtol=1e-4, atol=1e-5)
        ident = dict(P); ident.update(A.init_bn(cfg, P))
        close(U.unet_forward(cfg, ident, xn, t), g["pred_identity"], rtol=1e-4, atol=1e-5)
        assert np.array_equal(g["pred_identity"], g["pred_plain"]) and np.array_equal(g["pred_disabled"], g["pred_plain"])
        bn = C.anp_bn_init(list(zip(names, [int(c) for c in g["conv_couts"]])))
        full = dict(P); full.update(bn)
        close(U.unet_forward(cfg, full, xn, t), g["pred_perturbed"], rtol=1e-4, atol=1e-5)
    loss, G, norm, bn2, _, bm = A.anp_step(cfg, P, bn, {}, a, ac, clean, trig, targ, t, eps, C.ANP_LR, 1, C.ANP_BUDGET)
    close(loss, g["loss"], rtol=1e-5)
    gflat = torch.cat([G[n].flatten() for n in bn_names])
    assert float((gflat - T(g["bn_grads"])).norm() / T(g["bn_grads"]).norm()) < 1e-4
    close(norm, g["total_norm"], rtol=1e-5)
    after = torch.cat([bn2[n].flatten() for n in bn_names])
    big = T(g["bn_grads"]).abs() > 1e-3 * T(g["bn_grads"]).abs().max()        # (Adam's first step is lr * sign(g): exclude gradients at noise level)
    close(after[big], g["bn_after"][big.numpy()], rtol=1e-5, atol=1e-6)
    assert float(after.abs().max()) <= C.ANP_BUDGET + 1e-7 and float((after.abs() >= C.ANP_BUDGET - 1e-7).float().mean()) > 0.05
    close(bm, g["backdoor_mse"], rtol=1e-4)


# ------------------------------------------------------------------ G14: a 32-step trajectory of the reference's loop body
def traj_batch(step, trigger, target):
    """(R, x0, noise, t) of optimisation step `step`, built with the oracle's blend (pinned by G3) from the seeded pool of cases.py"""
    u8, flags = C.traj_pool()
    rows = C.traj_rows(step)
    x = torch.stack([BD.image_u8_to_float(u8[i]) for i in rows.tolist()])
    R, x0 = BD.make_batch(x, flags[rows], trigger, target)
    noise, t = C.traj_noise(step)
    return R, x0, noise, t


def test_oracle_replays_reference_training_trajectory(golden):
    """G14 (tests/golden/trajectory.npz): 32 steps of baddiffusion.py:590-615 run by the reference's own modules, optimizer and LR schedule.  The
    oracle (loss_ref + unet_ref + train_ref.clip_and_adam + cosine_lr_lambda) replays it: per-step loss, pre-clip norm and LR, the final weights, and
    the 5-step DDPM images from noise + trigger (baddiffusion.py:497-499) with those weights."""
    g = golden("trajectory")
    cfg = C.SMALL_CFGS["small"]
    trigger, target = BD.get_trigger(C.TRAJ_TRIGGER, 3, cfg.sample_size), None
    target = BD.get_target(C.TRAJ_TARGET, trigger)
    assert torch.equal(trigger, T(g["trigger"])) and torch.equal(target, T(g["target"]))
    _, a, ac = sched_ref.make_tables()
    P, state = U.gen_params(cfg, 7), {}
    for step in range(C.TRAJ_STEPS):
        R, x0, noise, t = traj_batch(step, trigger, target)
        loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, noise)
        lr = C.TRAJ_LR * train_ref.cosine_lr_lambda(step, C.TRAJ_WARMUP, C.TRAJ_TOTAL)
        P, state, norm = train_ref.clip_and_adam(P, G, state, lr, step + 1)
        assert abs(lr - g["lr"][step]) <= 1e-12 + 1e-9 * g["lr"][step], (step, lr, g["lr"][step])
        # fp32 on the same host, different op order (functional oracle vs nn.Module autograd): 1e-4 holds over the whole trajectory
        assert abs(float(loss) - g["loss"][step]) <= 1e-4 * g["loss"][step], (step, float(loss), g["loss"][step])
        assert abs(float(norm) - g["grad_norm"][step]) <= 5e-4 * g["grad_norm"][step], (step, float(norm), g["grad_norm"][step])
    names = [str(n) for n in g["names"]]
    move = g["pmove_final"]
    for i, k in enumerate(names):
        if k.endswith("key.bias"):        # gradient mathematically zero: Adam turns rounding noise into +-lr steps on both sides
            continue
        got = P[k].flatten()[:8].numpy()
        want = g["p8_final"][i][: got.size]
        # measured: 2.4e-7 worst element against a typical per-element displacement of 7.7e-4 over the 32 steps (pmove_final / sqrt(numel));
        # held to 1e-5 = 1.3 % of that displacement
        assert abs(float(P[k].double().norm()) - g["pnorm_final"][i]) <= 1e-5 * g["pnorm_final"][i] + 1e-7, k
        assert np.abs(got - want).max() <= 1e-5, (k, got, want, move[i])
    with torch.no_grad():
        x, R0, t, eps = C.train_inputs(cfg, 2)
        pred = U.unet_forward(cfg, P, loss_ref.q_sample(a, ac, x, R0, t, eps)[0], t)
        close(pred, g["pred_final"], 2e-3, 2e-4)
        for clip in (True, False):
            gen = torch.Generator().manual_seed(C.PIPE_SEED)
            x = C.traj_sample_init() + trigger.unsqueeze(0)
            for tt in sched_ref.ddpm_timesteps(C.TRAJ_SAMPLE_STEPS):
                e = U.unet_forward(cfg, P, x, int(tt))
                z = torch.randn(x.shape, generator=gen) if tt > 0 else None
                x, _ = sched_ref.ddpm_step(ac, e, int(tt), x, z, num_inference_steps=C.TRAJ_SAMPLE_STEPS, clip_sample=clip)
            close(sched_ref.to_image(x), g[f"ddpm{C.TRAJ_SAMPLE_STEPS}_trigger_init_{int(clip)}"], 2e-3, 5e-4)
