"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE in the build container.

Run once, here (CPU), from the repo root:   python tests/golden/make_golden.py
The reference (/root/reference, read-only) never travels to the GPU box; the vectors do.
Import recipe = SURVEY.md Appendix C.  Every fixture is data only: seeded inputs are
re-created by the tests from the same seeds, weights come from oracle.unet_ref.gen_params
(loaded into the reference's own nn.Modules with load_state_dict), outputs are what the
reference computed.

Fixtures:
  sched.npz     G1  tables, DDPM step (t x variance_type x clip), DDIM timesteps + step
  qsample.npz   G2  loss.q_sample_diffuser in/out + p_losses_diffuser value with a linear model
  backdoor.npz  G3  Backdoor box triggers, CORNER/TRIGGER/SHIFT targets, int masks, blend
  temb.npz      G4  get_timestep_embedding
  modules.npz   G5  ResnetBlock2D / AttentionBlock / Downsample2D / Upsample2D fwd + bwd
  unet_small.npz G6 two-level UNet fwd/bwd + 2/3-step DDPM and DDIM pipeline images
  unet_cifar.npz G7 full DDPM-CIFAR10-32 topology, B=2: output, loss, grad norms, one Adam step
  full_size.npz G10 BASELINE configs[1] at its real size (DDPM-CIFAR10-32 topology, B=128 train step) and the real
                    DDPM-CELEBA-HQ-256 network at 256x256, B=1 (forward + backward): what the reference computes, so the
                    full-size GPU tests compare with the reference and not with the product's other arithmetic mode
  attn_planes.npz G11 AttentionBlock forward + backward at 16 x 16 (256 tokens, head dim 256; one and two heads): the shapes the
                    split-plane attention path takes (modules.npz has 8 x 8 / 4 x 4 blocks, which stay on the unfused path)
  full_size_b4.npz G12 the same 256x256 network at B = 4, the per-GPU batch of BASELINE configs[3] (four timesteps, per-sample output
                    slices + checksums, every gradient tensor's norm + leading elements)
  trajectory.npz G14 32 optimisation steps of baddiffusion.py:590-615 on the small UNet (per-step loss, clip norm, LR; final weights;
                    DDPM images from noise + trigger with the trained weights)
  pndm.npz      G9  PNDMScheduler timesteps + full chains with a stand-in model, the scheduler every `--sched` other than
                    DDPM / DDIM ends up as (pipeline_pndm.py:46 converts whatever it is given), PNDMPipeline images
"""
import os, sys, importlib.util, math
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.set_num_threads(8)

# ---- import shim -------------------------------------------------------
import huggingface_hub, huggingface_hub.constants as c
if not hasattr(c, "hf_cache_home"):
    c.hf_cache_home = os.path.expanduser("~/.cache/huggingface")
class _Stub:
    def __init__(self, *a, **k): pass
    @staticmethod
    def get_token(): return None
for n in ("HfFolder", "cached_download"):
    if not hasattr(huggingface_hub, n):
        setattr(huggingface_hub, n, _Stub)
_orig = importlib.util.find_spec
importlib.util.find_spec = lambda name, *a, **k: None if name == "transformers" else _orig(name, *a, **k)
sys.path.insert(0, "/root/reference/diffusers/src")
import diffusers
importlib.util.find_spec = _orig
import datasets  # noqa: real HF datasets must be imported before the stubs
from unittest.mock import MagicMock
for n in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets",
          "comet_ml", "wandb", "pytorch_fid", "pytorch_fid.inception", "torchmetrics"):
    sys.modules[n] = MagicMock()
sys.path.insert(0, "/root/reference")
_cwd = os.getcwd(); os.chdir("/tmp")
import loss as ref_loss, dataset as ref_dataset
os.chdir(_cwd)
from diffusers import UNet2DModel, DDPMScheduler, DDIMScheduler, DDPMPipeline, DDIMPipeline
from diffusers.models.resnet import ResnetBlock2D, Downsample2D, Upsample2D
from diffusers.models.attention import AttentionBlock
from diffusers.models.embeddings import get_timestep_embedding

from oracle import unet_ref as U          # only for gen_params / configs (no math used here)
from tests.golden.cases import *          # shared seeds / shapes


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in kw.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


# ---- G1 schedulers ----------------------------------------------------------------------------
def g1():
    out = {}
    s = DDPMScheduler(num_train_timesteps=1000)
    out["betas"], out["alphas"], out["alphas_cumprod"] = s.betas, s.alphas, s.alphas_cumprod
    x, eps, z = sched_inputs()
    for vt in ("fixed_small", "fixed_large"):
        for clip in (True, False):
            s = DDPMScheduler(num_train_timesteps=1000, variance_type=vt, clip_sample=clip)
            for t in DDPM_TS:
                class G:  # hand the SAME noise to the reference through its randn_tensor call
                    pass
                import diffusers.schedulers.scheduling_ddpm as M
                keep = M.randn_tensor
                M.randn_tensor = lambda *a, **k: z.clone()
                r = s.step(eps, t, x)
                M.randn_tensor = keep
                out[f"ddpm_{vt}_{int(clip)}_{t}_prev"] = r.prev_sample
                out[f"ddpm_{vt}_{int(clip)}_{t}_x0"] = r.pred_original_sample
    s = DDPMScheduler(num_train_timesteps=1000, clip_sample=False, clip_defense=True, clip_defense_range=0.5)
    import diffusers.schedulers.scheduling_ddpm as M
    keep = M.randn_tensor; M.randn_tensor = lambda *a, **k: z.clone()
    out["ddpm_clipdef_500_prev"] = s.step(eps, 500, x).prev_sample
    M.randn_tensor = keep
    s = DDPMScheduler(num_train_timesteps=1000); s.set_timesteps(50)
    out["ddpm_ts50"] = s.timesteps
    for clip in (True, False):
        d = DDIMScheduler(num_train_timesteps=1000, clip_sample=clip); d.set_timesteps(50)
        out["ddim_ts50"] = d.timesteps
        for t in DDIM_TS:
            r = d.step(eps, t, x)
            out[f"ddim_{int(clip)}_{t}_prev"] = r.prev_sample
        r = d.step(eps, 500, x, eta=0.5, variance_noise=z)
        out[f"ddim_{int(clip)}_500_eta_prev"] = r.prev_sample
    out["add_noise"] = DDPMScheduler().add_noise(x, eps, torch.tensor([3, 977]))
    save("sched.npz", **out)


# ---- G2 q_sample / loss -----------------------------------------------------------------------
def g2():
    s = DDPMScheduler(num_train_timesteps=1000)
    x0, R, eps, t = qsample_inputs()
    xn, tgt = ref_loss.q_sample_diffuser(s, x0, R, t, eps)
    lin = torch.nn.Module(); lin.forward = None
    class Lin(torch.nn.Module):
        def forward(self, x, t, return_dict=False):
            return (0.5 * x - 0.01 * t.reshape(-1, 1, 1, 1).float() / 1000,)
    l2 = ref_loss.p_losses_diffuser(s, Lin(), x0, R, t, eps, "l2")
    l1 = ref_loss.p_losses_diffuser(s, Lin(), x0, R, t, eps, "l1")
    hub = ref_loss.p_losses_diffuser(s, Lin(), x0, R, t, eps, "huber")
    save("qsample.npz", x_noisy=xn, target=tgt, l2=l2, l1=l1, huber=hub)


# ---- G3 backdoor --------------------------------------------------------------------------------
def g3():
    bd = ref_dataset.Backdoor(root="/tmp")
    out = {}
    for S in (32, 256):
        for trig in ("BOX_4", "BOX_8", "BOX_11", "BOX_14", "BOX_18", "SM_BOX", "NONE"):
            g = bd.get_trigger(type=trig, channel=3, image_size=S)
            out[f"trig_{trig}_{S}"] = g
            out[f"mask_{trig}_{S}"] = torch.where(g > -1.0, 0, 1)          # dataset.py:275-276
            if S == 32 or trig == "BOX_14":
                for tg in ("CORNER", "TRIGGER", "SHIFT"):
                    out[f"tgt_{tg}_{trig}_{S}"] = bd.get_target(type=tg, trigger=g)
    # blend of dataset.py:306-315 on a seeded batch, BOX_14 / CORNER @32
    img = backdoor_images()
    g = bd.get_trigger(type="BOX_14", channel=3, image_size=32)
    m = torch.where(g > -1.0, 0, 1).repeat(img.shape[0], 1, 1, 1)
    out["blend_BOX_14_32"] = m * img + (1 - m) * g.repeat(img.shape[0], 1, 1, 1)
    import util as ref_util
    out["normalize_u8"] = ref_util.normalize(vmin_in=0.0, vmax_in=1.0, vmin_out=-1.0, vmax_out=1.0,
                                             x=torch.arange(256, dtype=torch.float32) / 255.0)
    save("backdoor.npz", **out)


# ---- G4 timestep embedding ------------------------------------------------------------------------
def g4():
    t = torch.tensor(TEMB_TS)
    save("temb.npz",
         cifar=get_timestep_embedding(t, 128, flip_sin_to_cos=False, downscale_freq_shift=1),
         default=get_timestep_embedding(t, 128, flip_sin_to_cos=True, downscale_freq_shift=0))


# ---- G5 modules --------------------------------------------------------------------------------------
def load_sub(mod, P, prefix):
    sd = {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}
    mod.load_state_dict(sd)
    return mod


def grads_summary(mod, prefix=""):
    out = {}
    for k, p in mod.named_parameters():
        out[f"{prefix}gn_{k}"] = p.grad.double().norm().float()
        out[f"{prefix}g8_{k}"] = p.grad.flatten()[:8]
    return out


def g5():
    out = {}
    for name, (cin, cout, hw) in RESNET_CASES.items():
        P = module_params(name)
        m = load_sub(ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=512, eps=1e-6, groups=32), P, "")
        x, temb, dy = resnet_inputs(name)
        x.requires_grad_(True); temb.requires_grad_(True)
        y = m(x, temb); y.backward(dy)
        out[f"{name}_y"] = y; out[f"{name}_dx"] = x.grad; out[f"{name}_dtemb"] = temb.grad
        out.update(grads_summary(m, name + "_"))
    for name, (C, hw, hd) in ATTN_CASES.items():
        P = module_params(name)
        m = load_sub(AttentionBlock(C, num_head_channels=hd, norm_num_groups=32, eps=1e-6), P, "")
        x, dy = attn_inputs(name); x.requires_grad_(True)
        y = m(x); y.backward(dy)
        out[f"{name}_y"] = y; out[f"{name}_dx"] = x.grad
        out.update(grads_summary(m, name + "_"))
    for name, (C, hw, pad) in DOWN_CASES.items():
        P = module_params(name)
        m = load_sub(Downsample2D(C, use_conv=True, out_channels=C, padding=pad, name="op"), P, "")
        x, dy = down_inputs(name); x.requires_grad_(True)
        y = m(x); y.backward(dy)
        out[f"{name}_y"] = y; out[f"{name}_dx"] = x.grad
        out.update(grads_summary(m, name + "_"))
    for name, (C, hw) in UP_CASES.items():
        P = module_params(name)
        m = load_sub(Upsample2D(C, use_conv=True, out_channels=C), P, "")
        x, dy = up_inputs(name); x.requires_grad_(True)
        y = m(x); y.backward(dy)
        out[f"{name}_y"] = y; out[f"{name}_dx"] = x.grad
        out.update(grads_summary(m, name + "_"))
    save("modules.npz", **out)


def g11():
    from tests.golden.cases import ATTN_SP_CASES
    out = {}
    for name, (C, hw, hd) in ATTN_SP_CASES.items():
        P = module_params(name)
        m = load_sub(AttentionBlock(C, num_head_channels=hd, norm_num_groups=32, eps=1e-6), P, "")
        x, dy = attn_inputs(name); x.requires_grad_(True)
        y = m(x); y.backward(dy)
        out[f"{name}_y"] = y; out[f"{name}_dx"] = x.grad
        out.update(grads_summary(m, name + "_"))
    save("attn_planes.npz", **out)


def ref_unet(cfg, P):
    m = UNet2DModel(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                    block_out_channels=cfg.block_out_channels, down_block_types=cfg.down_block_types,
                    up_block_types=cfg.up_block_types, layers_per_block=cfg.layers_per_block,
                    downsample_padding=cfg.downsample_padding, flip_sin_to_cos=cfg.flip_sin_to_cos,
                    freq_shift=cfg.freq_shift, norm_eps=cfg.norm_eps, norm_num_groups=cfg.norm_num_groups,
                    attention_head_dim=cfg.attention_head_dim)
    missing = m.load_state_dict(P, strict=True)
    return m


def train_step_fixture(cfg, seed, B, tag, lr=2e-4):
    out = {}
    P = U.gen_params(cfg, seed)
    m = ref_unet(cfg, P); m.train()
    sched = DDPMScheduler(num_train_timesteps=1000)
    x0, R, t, eps = train_inputs(cfg, B)
    y = m(q := ref_loss.q_sample_diffuser(sched, x0, R, t, eps)[0], t, return_dict=False)[0]
    out[f"{tag}_pred"] = y.detach()
    opt = torch.optim.Adam(m.parameters(), lr=lr)                      # baddiffusion.py:320
    loss = ref_loss.p_losses_diffuser(sched, m, x_start=x0, R=R, timesteps=t, noise=eps, loss_type="l2")
    loss.backward()                                                     # baddiffusion.py:607-608
    out[f"{tag}_loss"] = loss.detach()
    names = [k for k, _ in m.named_parameters()]
    out[f"{tag}_gradnorms"] = torch.stack([p.grad.double().norm().float() for _, p in m.named_parameters()])
    out[f"{tag}_grad8"] = torch.stack([torch.nn.functional.pad(p.grad.flatten()[:8], (0, max(0, 8 - p.numel())))
                                       for _, p in m.named_parameters()])
    out[f"{tag}_names"] = np.array(names)
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)           # baddiffusion.py:612
    out[f"{tag}_total_norm"] = gn
    opt.step()
    out[f"{tag}_p8_after"] = torch.stack([torch.nn.functional.pad(p.detach().flatten()[:8], (0, max(0, 8 - p.numel())))
                                          for _, p in m.named_parameters()])
    return out, m


# ---- G6 small UNet + pipelines ---------------------------------------------------------------------------
def g6():
    out = {}
    for tag, cfg in SMALL_CFGS.items():
        o, m = train_step_fixture(cfg, 7, 2, tag)
        out.update(o)
    cfg = SMALL_CFGS["small"]
    m = ref_unet(cfg, U.gen_params(cfg, 7)).eval()
    init = pipeline_init(cfg)
    for clip in (True, False):
        for vt in ("fixed_small", "fixed_large"):
            pipe = DDPMPipeline(m, DDPMScheduler(num_train_timesteps=1000, clip_sample=clip, variance_type=vt))
            pipe.set_progress_bar_config(disable=True)
            g = torch.Generator().manual_seed(PIPE_SEED)
            r = pipe(batch_size=init.shape[0], generator=g, init=init, output_type=None, num_inference_steps=3)
            out[f"ddpm3_{int(clip)}_{vt}"] = r.images
        pipe = DDIMPipeline(m, DDIMScheduler(num_train_timesteps=1000, clip_sample=clip))
        pipe.set_progress_bar_config(disable=True)
        r = pipe(batch_size=init.shape[0], init=init, output_type=None, num_inference_steps=4)
        out[f"ddim4_{int(clip)}"] = r.images
    save("unet_small.npz", **out)


# ---- G7 full CIFAR topology ---------------------------------------------------------------------------------
def g7():
    out, _ = train_step_fixture(U.CIFAR10_32, 0, 2, "cifar")
    save("unet_cifar.npz", **out)


# ---- G8 Frechet distance / activation statistics (fid_score.py:150-204, 207-230) ---------------------------
def g8():
    import fid_score as ref_fid          # pytorch_fid is a MagicMock; the two functions below are pure numpy / scipy
    out = {}
    for tag, d, n1, n2, seed in (("d64", 64, 300, 200, 0), ("d16", 16, 50, 40, 1), ("rank_deficient", 32, 20, 24, 2)):
        rng = np.random.RandomState(seed)
        a1 = rng.randn(n1, d) * (1 + rng.rand(d)) + rng.randn(d) * 0.3
        a2 = rng.randn(n2, d) @ (np.eye(d) + 0.2 * rng.randn(d, d)) + 0.5
        mu1, s1 = np.mean(a1, axis=0), np.cov(a1, rowvar=False)     # calculate_activation_statistics (fid_score.py:227-229)
        mu2, s2 = np.mean(a2, axis=0), np.cov(a2, rowvar=False)
        out[f"{tag}_a1"] = a1; out[f"{tag}_a2"] = a2
        out[f"{tag}_mu1"] = mu1; out[f"{tag}_sigma1"] = s1; out[f"{tag}_mu2"] = mu2; out[f"{tag}_sigma2"] = s2
        out[f"{tag}_fid"] = np.float64(ref_fid.calculate_frechet_distance(mu1, s1, mu2, s2))
    save("fid.npz", **out)


# ---- G9 PNDM: scheduling_pndm.py, pipelines/pndm/pipeline_pndm.py (SURVEY f-4) ----------------------------------
def g9():
    from diffusers import PNDMScheduler, PNDMPipeline, DPMSolverMultistepScheduler, UniPCMultistepScheduler
    out = {}
    x0 = pndm_init()
    for n in PNDM_STEPS:
        s = PNDMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02)
        s.set_timesteps(n)
        out[f"timesteps_{n}"] = s.timesteps
        x = x0.clone()
        xs = []
        for t in s.timesteps:
            x = s.step(pndm_fake_model(x, t), t, x).prev_sample
            xs.append(x.clone())
        out[f"chain_{n}"] = torch.stack(xs)
    # what model.py:598-630 hands to PNDMPipeline is converted with PNDMScheduler.from_config(other.config)
    for name, other in (("dpmpp2", DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                                                               solver_order=2, algorithm_type="dpmsolver++")),
                        ("unipc", UniPCMultistepScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02))):
        s = PNDMScheduler.from_config(other.config)
        s.set_timesteps(20)
        out[f"converted_{name}_timesteps_20"] = s.timesteps
        out[f"converted_{name}_final_alpha"] = s.final_alpha_cumprod
        out[f"converted_{name}_skip_prk"] = np.int64(bool(s.config.skip_prk_steps))
    # skip_prk_steps variant (PLMS start-up branches of step_plms)
    s = PNDMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, skip_prk_steps=True)
    s.set_timesteps(10)
    out["skip_timesteps_10"] = s.timesteps
    x = x0.clone(); xs = []
    for t in s.timesteps:
        x = s.step(pndm_fake_model(x, t), t, x).prev_sample
        xs.append(x.clone())
    out["skip_chain_10"] = torch.stack(xs)
    # the pipeline over the small UNet (same weights / init as G6), with and without the post-step clip
    cfg = SMALL_CFGS["small"]
    m = ref_unet(cfg, U.gen_params(cfg, 7)).eval()
    init = pipeline_init(cfg)
    for clip in (True, False):
        pipe = PNDMPipeline(m, DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02),
                            clip_sample=clip)
        pipe.set_progress_bar_config(disable=True)
        r = pipe(batch_size=init.shape[0], init=init, output_type=None, num_inference_steps=6)
        out[f"pipe6_{int(clip)}"] = r.images
    save("pndm.npz", **out)


# ---- G10 full-size anchors (unet_2d.py:229-326, baddiffusion.py:590-615) -----------------------------------------
def g10():
    import time
    t0 = time.time()
    out, _ = train_step_fixture(U.CIFAR10_32, 0, 128, "cifar128")
    pred = out.pop("cifar128_pred")                                       # [128,3,32,32]: keep 4 samples + per-sample sums
    out["cifar128_pred_rows"] = pred[list(FULL_ROWS)]
    out["cifar128_pred_sum"] = pred.double().sum(dim=(1, 2, 3))
    out["cifar128_pred_sumsq"] = (pred.double() ** 2).sum(dim=(1, 2, 3))
    print(f"cifar128 {time.time() - t0:.1f}s"); t0 = time.time()
    cfg = U.CELEBA_HQ_256
    m = ref_unet(cfg, U.gen_params(cfg, 5)); m.train()
    x, t, dout = celeba_full_inputs()
    y = m(x, t, return_dict=False)[0]
    y.backward(dout)
    out["celeba256_out_slices"] = y.detach()[0, :, ::16, ::16]
    out["celeba256_out_sum"] = y.detach().double().sum()
    out["celeba256_out_sumsq"] = (y.detach().double() ** 2).sum()
    out["celeba256_names"] = np.array([k for k, _ in m.named_parameters()])
    out["celeba256_gradnorms"] = torch.stack([p.grad.double().norm().float() for _, p in m.named_parameters()])
    out["celeba256_grad8"] = torch.stack([torch.nn.functional.pad(p.grad.flatten()[:8], (0, max(0, 8 - p.numel())))
                                          for _, p in m.named_parameters()])
    print(f"celeba256 {time.time() - t0:.1f}s")
    save("full_size.npz", **out)


# ---- G12: the 256 x 256 network at the per-GPU batch of BASELINE configs[3] (B = 4) -------------------------------------------------
def g12():
    import time
    from tests.golden.cases import celeba_b4_inputs
    t0 = time.time()
    cfg = U.CELEBA_HQ_256
    m = ref_unet(cfg, U.gen_params(cfg, 5)); m.train()
    x, t, dout = celeba_b4_inputs()
    y = m(x, t, return_dict=False)[0]
    y.backward(dout)
    out = {}
    out["celeba256b4_out_slices"] = y.detach()[:, :, ::16, ::16]
    out["celeba256b4_out_sum"] = y.detach().double().sum(dim=(1, 2, 3))
    out["celeba256b4_out_sumsq"] = (y.detach().double() ** 2).sum(dim=(1, 2, 3))
    out["celeba256b4_names"] = np.array([k for k, _ in m.named_parameters()])
    out["celeba256b4_gradnorms"] = torch.stack([p.grad.double().norm().float() for _, p in m.named_parameters()])
    out["celeba256b4_grad8"] = torch.stack([torch.nn.functional.pad(p.grad.flatten()[:8], (0, max(0, 8 - p.numel())))
                                            for _, p in m.named_parameters()])
    print(f"celeba256 B=4 {time.time() - t0:.1f}s")
    save("full_size_b4.npz", **out)


# ---- G13: Adversarial Neuron Pruning with the reference's own PerturbConv2d (anp_model.py) on the reference's UNet2DModel ----------
def g13():
    import anp_model as ref_anp                     # pure torch: imports as is
    # torch >= 2.x's Python wrapper F.batch_norm refuses eps = 0.0, which PerturbBatchNorm2d passes on purpose (anp_model.py:187-205, written for
    # torch 1.x); the aten op behind it takes it.  Shim of the WRAPPER's argument check only -- the reference class still decides every argument.
    class _F:
        def __getattr__(self, k): return getattr(torch.nn.functional, k)
        @staticmethod
        def batch_norm(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
            return torch.batch_norm(input, weight, bias, running_mean, running_var, training, momentum, eps, torch.backends.cudnn.enabled)
    ref_anp.F = _F()
    try:                                             # anp_util.convert_model / freeze (anp_util.py:60-101); its imports need the same stubs as dataset.py
        _c = os.getcwd(); os.chdir("/tmp")
        import anp_util as ref_anp_util
        os.chdir(_c)
        convert_model, freeze = ref_anp_util.convert_model, ref_anp_util.freeze
        how = "anp_util.convert_model"
    except Exception as e:                           # fall back to the same walk over the reference's PerturbConv2d class
        os.chdir(_c)
        print("anp_util not importable here (%s): wrapping with anp_model.PerturbConv2d directly" % type(e).__name__)
        how = "anp_model.PerturbConv2d applied by attribute type (anp_util.py:60-88 not importable)"

        def convert_model(model):
            def rec(module):
                for a in dir(module):
                    tgt = getattr(module, a)
                    if type(tgt) == torch.nn.Conv2d:
                        setattr(module, a, ref_anp.PerturbConv2d(layer=tgt))
                for _, ch in module.named_children():
                    rec(ch)
            rec(model)
            return model

        def freeze(model):
            for ch in model.children():
                for p in ch.parameters():
                    p.requires_grad = False
            return model
    cfg = SMALL_CFGS["small"]
    m = ref_unet(cfg, U.gen_params(cfg, 7)); m.eval()
    clean, trig, targ, t, eps = anp_inputs(cfg)
    sched = DDPMScheduler(num_train_timesteps=1000)
    with torch.no_grad():
        plain = m(ref_loss.q_sample_diffuser(sched, clean, torch.zeros_like(clean), t, eps)[0], t, return_dict=False)[0]
    pm = convert_model(freeze(m))
    named = [(n, p) for n, p in pm.named_parameters() if "bn" in n]           # anp_util.py:133
    convs = [(n[: -len(".bn.weight")], p.numel()) for n, p in named if n.endswith(".bn.weight")]
    out = {"how": np.array(how), "bn_names": np.array([n for n, _ in named]), "conv_names": np.array([n for n, _ in convs]),
           "conv_couts": np.array([c for _, c in convs])}
    xn = ref_loss.q_sample_diffuser(sched, clean, torch.zeros_like(clean), t, eps)[0]
    with torch.no_grad():
        out["pred_identity"] = pm(xn, t, return_dict=False)[0]                   # bn = (1, 0): the wrapped model IS the model (diff_output, anp_util.py:103-121)
        out["pred_plain"] = plain
    init = anp_bn_init(convs)
    with torch.no_grad():
        for n, p in named:
            p.copy_(init[n])
        out["pred_perturbed"] = pm(xn, t, return_dict=False)[0]
        ref_anp.disable_perturb(pm)
        out["pred_disabled"] = pm(xn, t, return_dict=False)[0]
        ref_anp.enable_perturb(pm)
    opt = torch.optim.Adam([p for _, p in named], lr=ANP_LR)                   # anp_util.py:135
    loss = -ref_loss.p_losses_diffuser(sched, model=pm, x_start=clean, R=torch.full_like(trig, 0), timesteps=t, noise=eps, loss_type="l2")
    loss.backward()                                                             # anp_defense.py:147-148
    out["loss"] = loss.detach()
    out["bn_grads"] = torch.cat([p.grad.flatten() for _, p in named])
    out["total_norm"] = torch.nn.utils.clip_grad_norm_(pm.parameters(), 1.0)   # anp_defense.py:152
    opt.step()
    with torch.no_grad():                                                       # clip_weight, anp_defense.py:68-75
        for n, p in named:
            p.clamp_(-ANP_BUDGET, ANP_BUDGET)
        out["bn_after"] = torch.cat([p.detach().flatten() for _, p in named])
        x_noisy, _ = ref_loss.q_sample_diffuser(sched, clean, torch.full_like(trig, 0), t, eps)      # backdoor_mse_fn, anp_defense.py:47-66
        _, btarget = ref_loss.q_sample_diffuser(sched, targ, trig, t, eps)
        out["backdoor_mse"] = torch.nn.functional.mse_loss(btarget, pm(x_noisy.contiguous(), t.contiguous(), return_dict=False)[0])
        out["pred_after"] = pm(xn, t, return_dict=False)[0]
    print("G13 via", how, "| loss", float(loss), "| norm", float(out["total_norm"]), "| backdoor mse", float(out["backdoor_mse"]))
    save("anp.npz", **out)


# ---- G14: a 32-step training trajectory of the reference's loop body + trigger-initialised sampling from the trained weights -------------
def g14():
    """baddiffusion.py:590-615 step by step on SMALL_CFGS["small"]: batches built by the reference's own Backdoor class and blend
    (dataset.py:275-276, 306-315), loss.p_losses_diffuser, clip_grad_norm_(1.0), torch.optim.Adam, diffusers' get_cosine_schedule_with_warmup;
    then DDPMPipeline from noise + trigger (baddiffusion.py:497-499) with the weights the trajectory ends with."""
    import util as ref_util
    from diffusers.optimization import get_cosine_schedule_with_warmup
    cfg = SMALL_CFGS["small"]
    S = cfg.sample_size
    bd = ref_dataset.Backdoor(root="/tmp")
    trigger = bd.get_trigger(type=TRAJ_TRIGGER, channel=3, image_size=S)
    target = bd.get_target(type=TRAJ_TARGET, trigger=trigger)
    mask = torch.where(trigger > -1.0, 0, 1)                                              # dataset.py:275-276
    u8, flags = traj_pool()
    img = ref_util.normalize(vmin_in=0.0, vmax_in=1.0, vmin_out=-1.0, vmax_out=1.0, x=u8.permute(0, 3, 1, 2).float() / 255.0)
    m = ref_unet(cfg, U.gen_params(cfg, 7)); m.train()
    sched = DDPMScheduler(num_train_timesteps=1000)
    opt = torch.optim.Adam(m.parameters(), lr=TRAJ_LR)                                     # baddiffusion.py:320
    lr_sched = get_cosine_schedule_with_warmup(optimizer=opt, num_warmup_steps=TRAJ_WARMUP, num_training_steps=TRAJ_TOTAL)   # :327-331
    names = [k for k, _ in m.named_parameters()]
    losses, norms, lrs, wnorm = [], [], [], []
    for step in range(TRAJ_STEPS):
        rows = traj_rows(step)
        x, f = img[rows], flags[rows]
        # clean rows: pixel_values = 0, target = image (dataset.py:288-304); backdoor rows: blended image, fixed target (:306-315)
        R = torch.where(f[:, None, None, None], mask * x + (1 - mask) * trigger, torch.zeros_like(x))
        x0 = torch.where(f[:, None, None, None], target.expand_as(x), x)
        noise, t = traj_noise(step)
        loss = ref_loss.p_losses_diffuser(sched, model=m, x_start=x0, R=R, timesteps=t, noise=noise, loss_type="l2")     # :607
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)                          # :612
        lrs.append(lr_sched.get_last_lr()[0])                                             # the LR this optimizer step uses
        opt.step(); lr_sched.step(); opt.zero_grad()                                      # :613-615
        losses.append(float(loss)); norms.append(float(gn))
        wnorm.append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in m.parameters()))))
    out = {"loss": np.array(losses, np.float64), "grad_norm": np.array(norms, np.float64), "lr": np.array(lrs, np.float64),
           "weight_norm": np.array(wnorm, np.float64), "names": np.array(names),
           "p8_final": torch.stack([torch.nn.functional.pad(p.detach().flatten()[:8], (0, max(0, 8 - p.numel()))) for p in m.parameters()]),
           "pnorm_final": torch.stack([p.detach().double().norm() for p in m.parameters()]),
           "trigger": trigger, "target": target}
    # distance of every tensor from its start: what 32 Adam steps moved (the tolerance of the test is stated against it)
    P0 = U.gen_params(cfg, 7)
    out["pmove_final"] = torch.stack([(p.detach().double() - P0[k].double()).norm() for k, p in m.named_parameters()])
    m.eval()
    with torch.no_grad():
        x, R0, t, eps = train_inputs(cfg, 2)
        out["pred_final"] = m(ref_loss.q_sample_diffuser(sched, x, R0, t, eps)[0], t, return_dict=False)[0]
    init = traj_sample_init() + trigger.unsqueeze(0)                                      # :497-499
    for clip in (True, False):
        pipe = DDPMPipeline(m, DDPMScheduler(num_train_timesteps=1000, clip_sample=clip))
        pipe.set_progress_bar_config(disable=True)
        r = pipe(batch_size=init.shape[0], generator=torch.Generator().manual_seed(PIPE_SEED), init=init, output_type=None,
                 num_inference_steps=TRAJ_SAMPLE_STEPS)
        out[f"ddpm{TRAJ_SAMPLE_STEPS}_trigger_init_{int(clip)}"] = r.images
    print("G14 loss", losses[0], "->", losses[-1], "| lr", lrs[0], lrs[TRAJ_WARMUP], lrs[-1], "| clip norm", norms[0], norms[-1])
    save("trajectory.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14"]
    for w in which:
        globals()[w]()
