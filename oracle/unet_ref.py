"""Oracle: UNet2DModel forward as functional fp32 PyTorch-CPU ops (autograd gives backward).

Follows /root/reference/diffusers/src/diffusers/models/unet_2d.py:82-217 (topology),
:229-326 (forward); resnet.py:551-601 (ResnetBlock2D), :95-161 (Upsample2D),
:164-208 (Downsample2D); attention.py:121-174 (AttentionBlock);
embeddings.py:22-62, 155-229 (timestep embedding); unet_2d_blocks.py:884-960, 674-750,
390-466, 1871-1942, 1663-1735 (block wrappers, skip bookkeeping).
Parameters are a flat dict keyed by the reference's state_dict names.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    sample_size: int = 32
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 256, 256)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    downsample_padding: int = 0
    flip_sin_to_cos: bool = False
    freq_shift: int = 1
    norm_eps: float = 1e-6
    norm_num_groups: int = 32
    attention_head_dim: Optional[int] = None
    mid_block_scale_factor: float = 1.0


CIFAR10_32 = UNetConfig()                     # google/ddpm-cifar10-32 topology
CELEBA_HQ_256 = UNetConfig(
    sample_size=256, block_out_channels=(128, 128, 256, 256, 512, 512),
    down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)


def _resnet_shapes(p, prefix, cin, cout, temb):
    p[prefix + "norm1.weight"] = (cin,); p[prefix + "norm1.bias"] = (cin,)
    p[prefix + "conv1.weight"] = (cout, cin, 3, 3); p[prefix + "conv1.bias"] = (cout,)
    p[prefix + "time_emb_proj.weight"] = (cout, temb); p[prefix + "time_emb_proj.bias"] = (cout,)
    p[prefix + "norm2.weight"] = (cout,); p[prefix + "norm2.bias"] = (cout,)
    p[prefix + "conv2.weight"] = (cout, cout, 3, 3); p[prefix + "conv2.bias"] = (cout,)
    if cin != cout:
        p[prefix + "conv_shortcut.weight"] = (cout, cin, 1, 1); p[prefix + "conv_shortcut.bias"] = (cout,)


def _attn_shapes(p, prefix, c):
    p[prefix + "group_norm.weight"] = (c,); p[prefix + "group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        p[prefix + n + ".weight"] = (c, c); p[prefix + n + ".bias"] = (c,)


def up_block_channels(cfg, i):
    """(resnet_in, skip) channel pairs of up block i (unet_2d.py:186-211, unet_2d_blocks.py:1690-1692)."""
    rev = list(reversed(cfg.block_out_channels))
    n = len(rev)
    prev_out = rev[0] if i == 0 else rev[i - 1]
    out = rev[i]
    inp = rev[min(i + 1, n - 1)]
    L = cfg.layers_per_block + 1
    return [((prev_out if j == 0 else out), (inp if j == L - 1 else out), out) for j in range(L)]


def param_shapes(cfg: UNetConfig):
    """Ordered {state_dict key: logical shape}."""
    p = {}
    c0 = cfg.block_out_channels[0]
    temb = 4 * c0
    p["conv_in.weight"] = (c0, cfg.in_channels, 3, 3); p["conv_in.bias"] = (c0,)
    p["time_embedding.linear_1.weight"] = (temb, c0); p["time_embedding.linear_1.bias"] = (temb,)
    p["time_embedding.linear_2.weight"] = (temb, temb); p["time_embedding.linear_2.bias"] = (temb,)
    out = c0
    n = len(cfg.block_out_channels)
    for i, bt in enumerate(cfg.down_block_types):
        inp, out = out, cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            if bt == "AttnDownBlock2D":
                _attn_shapes(p, f"down_blocks.{i}.attentions.{j}.", out)
        for j in range(cfg.layers_per_block):
            _resnet_shapes(p, f"down_blocks.{i}.resnets.{j}.", inp if j == 0 else out, out, temb)
        if i != n - 1:
            p[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out, out, 3, 3)
            p[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out,)
    cm = cfg.block_out_channels[-1]
    _attn_shapes(p, "mid_block.attentions.0.", cm)
    _resnet_shapes(p, "mid_block.resnets.0.", cm, cm, temb)
    _resnet_shapes(p, "mid_block.resnets.1.", cm, cm, temb)
    for i, bt in enumerate(cfg.up_block_types):
        chans = up_block_channels(cfg, i)
        for j, (_, _, o) in enumerate(chans):
            if bt == "AttnUpBlock2D":
                _attn_shapes(p, f"up_blocks.{i}.attentions.{j}.", o)
        for j, (rin, skip, o) in enumerate(chans):
            _resnet_shapes(p, f"up_blocks.{i}.resnets.{j}.", rin + skip, o, temb)
        if i != n - 1:
            o = chans[0][2]
            p[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (o, o, 3, 3)
            p[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (o,)
    p["conv_norm_out.weight"] = (c0,); p["conv_norm_out.bias"] = (c0,)
    p["conv_out.weight"] = (cfg.out_channels, c0, 3, 3); p["conv_out.bias"] = (cfg.out_channels,)
    return p


def gen_params(cfg: UNetConfig, seed: int = 0, dtype=torch.float32):
    """Deterministic, construction-order-independent parameters (one generator per key):
    weights U(-b, b) with b = 1/sqrt(fan_in) (the nn.Conv2d / nn.Linear default bound),
    norm weights 1 + 0.1*U(-1,1), norm biases 0.1*U(-1,1), other biases U(-b, b).
    Used by the golden script (loaded into the reference modules) and by every parity test."""
    out = {}
    for idx, (k, shp) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        u = torch.rand(shp, generator=g, dtype=torch.float32) * 2 - 1
        is_norm = ("norm" in k.split(".")[-2])
        if is_norm:
            v = 1 + 0.1 * u if k.endswith("weight") else 0.1 * u
        else:
            wkey = k[: -len("bias")] + "weight" if k.endswith("bias") else k
            wshape = param_shapes_cached(cfg)[wkey]
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            v = u / math.sqrt(fan_in)
        out[k] = v.to(dtype)
    return out


_shape_cache = {}


def param_shapes_cached(cfg):
    key = id(cfg)
    if key not in _shape_cache:
        _shape_cache[key] = param_shapes(cfg)
    return _shape_cache[key]


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift, max_period=10000):
    # embeddings.py:40-57
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def conv2d(P, name, x, **kw):
    """nn.Conv2d `name`; when P holds `<name>.bn.weight / .bn.bias` the layer is the ANP-wrapped PerturbConv2d (anp_model.py:490-514):
    the convolution followed by F.batch_norm(..., mean 0, var 1, weight, bias, training=False, eps=0.0) = a per-channel affine map."""
    y = F.conv2d(x, P[name + ".weight"], P[name + ".bias"], **kw)
    if name + ".bn.weight" in P:
        y = y * P[name + ".bn.weight"][None, :, None, None] + P[name + ".bn.bias"][None, :, None, None]
    return y


def resnet_block(P, pre, x, emb, groups, eps, scale=1.0):
    # resnet.py:551-601 (time_embedding_norm="default", dropout p=0)
    h = F.silu(F.group_norm(x, groups, P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps))
    h = conv2d(P, pre + "conv1", h, padding=1)
    t = F.linear(F.silu(emb), P[pre + "time_emb_proj.weight"], P[pre + "time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps))
    h = conv2d(P, pre + "conv2", h, padding=1)
    if pre + "conv_shortcut.weight" in P:
        x = conv2d(P, pre + "conv_shortcut", x)
    return (x + h) / scale


def attention_block(P, pre, x, groups, eps, head_dim=None, scale_out=1.0):
    # attention.py:121-174
    B, C, H, W = x.shape
    heads = C // head_dim if head_dim is not None else 1
    h = F.group_norm(x, groups, P[pre + "group_norm.weight"], P[pre + "group_norm.bias"], eps)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, P[pre + "query.weight"], P[pre + "query.bias"])
    k = F.linear(h, P[pre + "key.weight"], P[pre + "key.bias"])
    v = F.linear(h, P[pre + "value.weight"], P[pre + "value.bias"])

    def split(z):
        return z.reshape(B, H * W, heads, C // heads).permute(0, 2, 1, 3).reshape(B * heads, H * W, C // heads)

    q, k, v = split(q), split(k), split(v)
    s = torch.bmm(q, k.transpose(-1, -2)) * (1 / math.sqrt(C / heads))
    p = torch.softmax(s.float(), dim=-1)
    o = torch.bmm(p, v)
    o = o.reshape(B, heads, H * W, C // heads).permute(0, 2, 1, 3).reshape(B, H * W, C)
    o = F.linear(o, P[pre + "proj_attn.weight"], P[pre + "proj_attn.bias"])
    o = o.transpose(-1, -2).reshape(B, C, H, W)
    return (o + x) / scale_out


def downsample(P, pre, x, padding):
    # resnet.py:199-208
    if padding == 0:
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return conv2d(P, pre + "conv", x, stride=2, padding=padding)


def upsample(P, pre, x):
    # resnet.py:126-161
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return conv2d(P, pre + "conv", x, padding=1)


def unet_forward(cfg: UNetConfig, P, sample, timestep):
    """sample [B,C,H,W] fp32, timestep int / 0-dim / [B]  ->  [B,out,H,W]   (unet_2d.py:229-326)."""
    B = sample.shape[0]
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.long)
    elif timestep.dim() == 0:
        timestep = timestep[None]
    timestep = timestep * torch.ones(B, dtype=timestep.dtype)
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    c0 = cfg.block_out_channels[0]
    t_emb = timestep_embedding(timestep, c0, cfg.flip_sin_to_cos, cfg.freq_shift)
    emb = F.linear(t_emb, P["time_embedding.linear_1.weight"], P["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), P["time_embedding.linear_2.weight"], P["time_embedding.linear_2.bias"])

    h = conv2d(P, "conv_in", sample, padding=1)
    skips = [h]
    n = len(cfg.block_out_channels)
    for i, bt in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            h = resnet_block(P, f"down_blocks.{i}.resnets.{j}.", h, emb, G, eps)
            if bt == "AttnDownBlock2D":
                h = attention_block(P, f"down_blocks.{i}.attentions.{j}.", h, G, eps, cfg.attention_head_dim)
            skips.append(h)
        if i != n - 1:
            h = downsample(P, f"down_blocks.{i}.downsamplers.0.", h, cfg.downsample_padding)
            skips.append(h)
    s = cfg.mid_block_scale_factor
    h = resnet_block(P, "mid_block.resnets.0.", h, emb, G, eps, s)
    h = attention_block(P, "mid_block.attentions.0.", h, G, eps, cfg.attention_head_dim, s)
    h = resnet_block(P, "mid_block.resnets.1.", h, emb, G, eps, s)
    for i, bt in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(P, f"up_blocks.{i}.resnets.{j}.", h, emb, G, eps)
            if bt == "AttnUpBlock2D":
                h = attention_block(P, f"up_blocks.{i}.attentions.{j}.", h, G, eps, cfg.attention_head_dim)
        if i != n - 1:
            h = upsample(P, f"up_blocks.{i}.upsamplers.0.", h)
    h = F.silu(F.group_norm(h, G, P["conv_norm_out.weight"], P["conv_norm_out.bias"], eps))
    return conv2d(P, "conv_out", h, padding=1)
