"""Seeded inputs / shapes shared by make_golden.py (reference side) and the parity tests.
Pure torch CPU generators; no reference import, no oracle math."""
import math
import torch

from oracle.unet_ref import UNetConfig

DDPM_TS = (999, 500, 1, 0)
DDIM_TS = (980, 500, 0)
TEMB_TS = [0, 1, 2, 10, 100, 500, 998, 999]
PIPE_SEED = 1234


def _r(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def sched_inputs():
    return _r(11, 2, 3, 8, 8), _r(12, 2, 3, 8, 8), _r(13, 2, 3, 8, 8)       # x_t, eps_hat, z


def qsample_inputs():
    x0, R, eps = _r(21, 4, 3, 8, 8), _r(22, 4, 3, 8, 8), _r(23, 4, 3, 8, 8)
    R[1] = 0                                                                  # a clean row
    return x0, R, eps, torch.tensor([0, 10, 500, 999])


def backdoor_images():
    u8 = torch.randint(0, 256, (4, 32, 32, 3), generator=torch.Generator().manual_seed(31), dtype=torch.uint8)
    x = u8.permute(0, 3, 1, 2).float() / 255.0
    return ((x - 0.0) / (1.0 - 0.0 + 1e-5)) * 2.0 - 1.0


# name -> (cin, cout, hw)
RESNET_CASES = {"res_128_128": (128, 128, 8), "res_128_256": (128, 256, 8), "res_512_256": (512, 256, 4)}
# name -> (C, hw, head_dim)
ATTN_CASES = {"attn_256_h1": (256, 8, None), "attn_256_hd8": (256, 4, 8), "attn_128_h1": (128, 4, None)}
# the attention blocks the split-plane path takes (256 tokens = 16 x 16, head dim 256): one head as in the CIFAR topology, and two heads
ATTN_SP_CASES = {"attn_256_n256": (256, 16, None), "attn_512_n256_hd256": (512, 16, 256)}
# name -> (C, hw, padding)
DOWN_CASES = {"down_128_p0": (128, 8, 0), "down_128_p1": (128, 8, 1)}
UP_CASES = {"up_128": (128, 4)}
MOD_B = 2


def _module_shapes(name):
    if name in RESNET_CASES:
        cin, cout, _ = RESNET_CASES[name]
        s = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,),
             "time_emb_proj.weight": (cout, 512), "time_emb_proj.bias": (cout,),
             "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
        if cin != cout:
            s["conv_shortcut.weight"] = (cout, cin, 1, 1); s["conv_shortcut.bias"] = (cout,)
        return s
    if name in ATTN_CASES or name in ATTN_SP_CASES:
        C = (ATTN_CASES.get(name) or ATTN_SP_CASES[name])[0]
        s = {"group_norm.weight": (C,), "group_norm.bias": (C,)}
        for n in ("query", "key", "value", "proj_attn"):
            s[n + ".weight"] = (C, C); s[n + ".bias"] = (C,)
        return s
    C = (DOWN_CASES.get(name) or UP_CASES.get(name))[0]
    return {"conv.weight": (C, C, 3, 3), "conv.bias": (C,)}


def module_params(name):
    """Seeded parameters for a stand-alone module case (same recipe as oracle.unet_ref.gen_params)."""
    out = {}
    shapes = _module_shapes(name)
    base = sum(ord(ch) for ch in name) * 7919
    for i, (k, shp) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(base + i)
        u = torch.rand(shp, generator=g) * 2 - 1
        if "norm" in k:
            out[k] = 1 + 0.1 * u if k.endswith("weight") else 0.1 * u
        else:
            w = shapes[k[:-4] + "weight"] if k.endswith("bias") else shp
            out[k] = u / math.sqrt(math.prod(w[1:]))
    return out


def resnet_inputs(name):
    cin, cout, hw = RESNET_CASES[name]
    return _r(41, MOD_B, cin, hw, hw), _r(42, MOD_B, 512), _r(43, MOD_B, cout, hw, hw)


def attn_inputs(name):
    C, hw, _ = ATTN_CASES.get(name) or ATTN_SP_CASES[name]
    return _r(51, MOD_B, C, hw, hw), _r(52, MOD_B, C, hw, hw)


def down_inputs(name):
    C, hw, _ = DOWN_CASES[name]
    return _r(61, MOD_B, C, hw, hw), _r(62, MOD_B, C, hw // 2, hw // 2)


def up_inputs(name):
    C, hw = UP_CASES[name]
    return _r(71, MOD_B, C, hw, hw), _r(72, MOD_B, C, 2 * hw, 2 * hw)


SMALL_CFGS = {
    # two levels, every layer kind of the CIFAR topology (attn down/up, 1x1 shortcut, pad-0 downsample, upsample)
    "small": UNetConfig(sample_size=16, block_out_channels=(128, 256),
                        down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                        up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1),
    # MODEL_DEFAULT-style knobs (model.py:654-680): pad 1, flip_sin_to_cos, freq_shift 0, eps 1e-5, head dim 8
    "small_default": UNetConfig(sample_size=16, block_out_channels=(128, 256),
                                down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                                up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1,
                                downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, norm_eps=1e-5,
                                attention_head_dim=8),
}


def train_inputs(cfg, B):
    """(x0, R, t, eps) as in SURVEY 8d: uint8-derived images, every other row poisoned with BOX/CORNER-like R."""
    S = cfg.sample_size
    u8 = torch.randint(0, 256, (B, S, S, 3), generator=torch.Generator().manual_seed(0), dtype=torch.uint8)
    x = (u8.permute(0, 3, 1, 2).float() / 255.0) / (1.0 + 1e-5) * 2.0 - 1.0
    R = torch.zeros_like(x)
    x0 = x.clone()
    k = max(2, S // 2 - 2)
    for b in range(0, B, 2):                                   # poisoned rows: grey box trigger, corner target
        R[b] = x[b]
        R[b, :, -(k + 2):-2, -(k + 2):-2] = 0.0
        tgt = torch.full((3, S, S), -0.4)
        tgt[:, : S // 3, : S // 3] = 0.0
        x0[b] = tgt
    eps = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(1))
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(2))
    return x0, R, t, eps


# ---- G10: full-size anchors -------------------------------------------------------------------------------------
FULL_ROWS = (0, 63, 64, 127)          # both half-batch pipelines of the product's forward


def celeba_full_inputs():
    """(x, t, dout) for the real 256x256 network, batch 1"""
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(3))
    dout = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(4))
    return x, torch.tensor([417]), dout


def celeba_b4_inputs():
    """(x, t, dout) for the real 256x256 network at BASELINE configs[3]'s per-GPU batch of 4 (distinct timesteps per sample)"""
    x = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(13))
    dout = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(14)) / 4
    return x, torch.tensor([417, 12, 988, 500]), dout


def pipeline_init(cfg, n=2):
    return torch.randn(n, 3, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(0))


# ---- G9: PNDM (f-4) ------------------------------------------------------------------------------------------
PNDM_STEPS = (50, 20, 7)


def pndm_fake_model(x, t):
    """deterministic stand-in for the UNet in the scheduler-level PNDM chains: elementwise, so the reference run (CPU)
    and the test run (any device) evaluate exactly the same function"""
    import torch
    tt = float(t) / 1000.0
    return 0.6 * x + 0.25 * torch.sin(3.0 * x + tt) - 0.1 * tt


def pndm_init():
    import torch
    return torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(77))


# ---- G13: Adversarial Neuron Pruning (f-4; anp_model.py / anp_util.py / anp_defense.py) ------------------------------------------
ANP_LR, ANP_BUDGET, ANP_B = 1e-3, 1.25, 2


def anp_bn_init(conv_couts):
    """seeded bn.weight ~ 1 + 0.3 N(0,1) (a fifth of them beyond the budget), bn.bias ~ 0.2 N(0,1); conv_couts: [(conv name, Cout)] in
    state-dict order.  The reference initialises (1, 0); a trained state exercises the effective-weight algebra."""
    g = torch.Generator().manual_seed(1234)
    out = {}
    for name, c in conv_couts:
        out[name + ".bn.weight"] = 1.0 + 0.3 * torch.randn(c, generator=g)
        out[name + ".bn.bias"] = 0.2 * torch.randn(c, generator=g)
    return out


def anp_inputs(cfg, B=ANP_B):
    """(clean, trigger_images, target_images, t, noise): batch['image'], batch['pixel_values'], batch['target'] of anp_defense.py:122-124."""
    x0, R, t, eps = train_inputs(cfg, B)
    S = cfg.sample_size
    clean = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(5)) * 2 - 1
    return clean, R, x0, t, eps


# ---- G14: a K-step training trajectory (baddiffusion.py:590-615: p_losses_diffuser -> backward -> clip 1.0 -> Adam -> cosine LR) ----------
TRAJ_STEPS, TRAJ_B, TRAJ_POOL = 32, 8, 64
TRAJ_LR, TRAJ_WARMUP, TRAJ_TOTAL = 2e-4, 8, 48          # warm-up ends inside the trajectory; the cosine part is exercised as well
TRAJ_TRIGGER, TRAJ_TARGET = "BOX_8", "CORNER"          # 16 x 16 images: an 8 x 8 grey box two pixels from the corner, the 10 x 10 corner target
TRAJ_SAMPLE_STEPS, TRAJ_SAMPLE_N = 5, 4


def traj_pool():
    """(uint8 images [POOL,16,16,3], poison flags [POOL]): a seeded pool; every fourth row is a backdoor row (poison_rate 0.25)"""
    S = SMALL_CFGS["small"].sample_size
    u8 = torch.randint(0, 256, (TRAJ_POOL, S, S, 3), generator=torch.Generator().manual_seed(140), dtype=torch.uint8)
    return u8, (torch.arange(TRAJ_POOL) % 4 == 0)


def traj_rows(step):
    """rows of the pool that make up the batch of optimisation step `step` (a stride walk: every batch mixes clean and backdoor rows)"""
    return (torch.arange(TRAJ_B) * 5 + step * 3) % TRAJ_POOL


def traj_noise(step):
    """(noise [B,3,S,S], timesteps [B]) of step `step` (baddiffusion.py:596, 600 draw them from the global RNG; here they are seeded per step)"""
    S = SMALL_CFGS["small"].sample_size
    g = torch.Generator().manual_seed(14000 + step)
    return torch.randn(TRAJ_B, 3, S, S, generator=g), torch.randint(0, 1000, (TRAJ_B,), generator=g)


def traj_sample_init():
    """clean noise for the trigger-initialised DDPM chains of baddiffusion.py:497-499 (the trigger is added by the caller)"""
    S = SMALL_CFGS["small"].sample_size
    return torch.randn(TRAJ_SAMPLE_N, 3, S, S, generator=torch.Generator().manual_seed(141))
