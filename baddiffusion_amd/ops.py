"""Thin Python wrappers over the C ABI (include/bd_hip.h): torch tensors in, torch tensors out.

Tensors are only carriers of device memory here (allocation, streams); every FLOP on the hot path
happens in libbd_hip.so.  All wrappers enqueue on torch's current stream and never synchronise.
Activations are NHWC: a [B,H,W,C] contiguous tensor, or any 2-D [rows, C] view with row stride `ld`.
"""
import ctypes as C

import torch

from . import _lib as L

_ws_cache = {}


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("baddiffusion_amd.ops: tensors must live on the GPU (no CPU fallback on the hot path)")


def workspace(nbytes, device, tag="default"):
    """A cached scratch buffer of at least nbytes (per device / tag)."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _ld(t):
    """leading dimension (elements between rows) of a [..., C] tensor whose last dim is contiguous."""
    assert t.size(-1) == 1 or t.stride(-1) == 1     # (the stride of a size-1 dimension is arbitrary)
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


# ------------------------------------------------------------------------------------------------ a-1/a-2
def poison_qsample(images, is_poison, trigger, target_img, noise, timesteps, alphas, alphas_cumprod,
                   vmin=-1.0, want_batch=False, want_mask=False, want_image=False, row_index=None, flip=None):
    """Fused blend + q_sample.  images: float [B,C,H,W] (normalised) or uint8 [B,H,W,C].
    row_index (int64 [B]): batch row b reads images[row_index[b]] (images is then the whole resident array);
    flip (bool/uint8 [B]): mirror that image along W.  Both optional: the shuffled gather and
    RandomHorizontalFlip of the reference's DataLoader, fused into the kernel.
    Returns (x_noisy NHWC [B,H,W,C], target NHWC [B,H,W,C][, R NCHW, x0 NCHW][, mask int64 [C,H,W]])."""
    lib = L.load()
    _need_cuda(images, is_poison, trigger, target_img, noise, timesteps, alphas, alphas_cumprod, row_index, flip)
    u8 = images.dtype == torch.uint8
    if u8:
        B, H, W, Cc = images.shape
    else:
        B, Cc, H, W = images.shape
    if row_index is not None:
        row_index = row_index.to(torch.int64).contiguous()
        B = row_index.numel()
    if flip is not None:
        flip = flip.to(torch.uint8).contiguous()
        assert flip.numel() == B
    dev = images.device
    images = images.contiguous(); noise = noise.contiguous()
    is_poison = is_poison.to(torch.uint8).contiguous()
    trigger = trigger.contiguous().float(); target_img = target_img.contiguous().float()
    timesteps = timesteps.to(torch.int64).contiguous()
    xn = torch.empty(B, H, W, Cc, device=dev); tg = torch.empty(B, H, W, Cc, device=dev)
    R = torch.empty(B, Cc, H, W, device=dev) if want_batch else None
    x0 = torch.empty(B, Cc, H, W, device=dev) if want_batch else None
    mask = torch.empty(Cc, H, W, dtype=torch.int64, device=dev) if want_mask else None
    image = torch.empty(B, Cc, H, W, device=dev) if want_image else None
    d = L.PoisonQsampleDesc(B=B, C=Cc, H=H, W=W, images_f32=None if u8 else L.ptr(images),
                            images_u8=L.ptr(images) if u8 else None, is_poison=L.ptr(is_poison),
                            trigger=L.ptr(trigger), target_img=L.ptr(target_img), noise=L.ptr(noise),
                            timesteps=L.ptr(timesteps), alphas=L.ptr(alphas), alphas_cumprod=L.ptr(alphas_cumprod),
                            vmin=vmin, x_noisy=L.ptr(xn), ld_noisy=Cc, target=L.ptr(tg), ld_target=Cc,
                            R_out=L.ptr(R), x0_out=L.ptr(x0), mask_out=L.ptr(mask), image_out=L.ptr(image),
                            row_index=L.ptr(row_index), flip=L.ptr(flip))
    L.check(lib.bd_poison_qsample(C.byref(d), L.stream()), "bd_poison_qsample")
    out = [xn, tg]
    if want_batch:
        out += [R, x0]
    if want_mask:
        out += [mask]
    if want_image:
        out += [image]
    return tuple(out)


def qsample(x0, R, noise, timesteps, alphas, alphas_cumprod):
    """loss.py:257-285 on NCHW inputs -> (x_noisy, target) NHWC [B,H,W,C]."""
    lib = L.load()
    _need_cuda(x0, R, noise, timesteps, alphas, alphas_cumprod)
    B, Cc, H, W = x0.shape
    x0 = x0.contiguous().float(); R = R.contiguous().float(); noise = noise.contiguous().float()
    timesteps = timesteps.to(torch.int64).contiguous()
    xn = torch.empty(B, H, W, Cc, device=x0.device); tg = torch.empty(B, H, W, Cc, device=x0.device)
    d = L.QsampleDesc(B=B, C=Cc, H=H, W=W, x0=L.ptr(x0), R=L.ptr(R), noise=L.ptr(noise), timesteps=L.ptr(timesteps),
                      alphas=L.ptr(alphas), alphas_cumprod=L.ptr(alphas_cumprod), x_noisy=L.ptr(xn), ld_noisy=Cc,
                      target=L.ptr(tg), ld_target=Cc)
    L.check(lib.bd_qsample(C.byref(d), L.stream()), "bd_qsample")
    return xn, tg


def nchw_to_nhwc(x):
    lib = L.load(); _need_cuda(x)
    B, Cc, H, W = x.shape
    x = x.contiguous().float()
    y = torch.empty(B, H, W, Cc, device=x.device)
    L.check(lib.bd_nchw_to_nhwc(L.ptr(x), L.ptr(y), B, Cc, H, W, Cc, L.stream()), "bd_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x):
    lib = L.load(); _need_cuda(x)
    B, H, W, Cc = x.shape
    x = x.contiguous()
    y = torch.empty(B, Cc, H, W, device=x.device)
    L.check(lib.bd_nhwc_to_nchw(L.ptr(x), Cc, L.ptr(y), B, Cc, H, W, L.stream()), "bd_nhwc_to_nchw")
    return y


# ------------------------------------------------------------------------------------------------ a-5/a-6/a-7
def ddpm_step(model_output, sample, noise, alphas_cumprod, t, prev_t, variance_type="fixed_small", clip_sample=True,
              clip_sample_range=1.0, clip_defense=False, clip_defense_range=1.0, out=None, want_x0=False):
    lib = L.load(); _need_cuda(model_output, sample, noise, alphas_cumprod)
    vt = {"fixed_small": 0, "fixed_large": 1}.get(variance_type)
    if vt is None:
        raise NotImplementedError(f"variance_type {variance_type} is not supported by the HIP DDPM step "
                                  "(BadDiffusion uses fixed_small / fixed_large)")
    assert model_output.is_contiguous() and sample.is_contiguous()
    prev = torch.empty_like(sample) if out is None else out
    x0 = torch.empty_like(sample) if want_x0 else None
    d = L.DdpmStepDesc(n=sample.numel(), model_output=L.ptr(model_output), sample=L.ptr(sample),
                       noise=L.ptr(noise.contiguous()) if noise is not None else None, prev_sample=L.ptr(prev),
                       pred_original=L.ptr(x0), alphas_cumprod=L.ptr(alphas_cumprod), t=int(t), prev_t=int(prev_t),
                       variance_type=vt, clip_sample=int(bool(clip_sample)), clip_sample_range=clip_sample_range,
                       clip_defense=int(bool(clip_defense)), clip_defense_range=clip_defense_range)
    L.check(lib.bd_ddpm_step(C.byref(d), L.stream()), "bd_ddpm_step")
    return (prev, x0) if want_x0 else prev


def ddim_step(model_output, sample, alphas_cumprod, t, prev_t, eta=0.0, noise=None, clip_sample=True,
              clip_sample_range=1.0, final_alpha_cumprod=1.0, out=None, want_x0=False):
    lib = L.load(); _need_cuda(model_output, sample, noise, alphas_cumprod)
    assert model_output.is_contiguous() and sample.is_contiguous()
    prev = torch.empty_like(sample) if out is None else out
    x0 = torch.empty_like(sample) if want_x0 else None
    d = L.DdimStepDesc(n=sample.numel(), model_output=L.ptr(model_output), sample=L.ptr(sample),
                       noise=L.ptr(noise.contiguous()) if noise is not None else None, prev_sample=L.ptr(prev),
                       pred_original=L.ptr(x0), alphas_cumprod=L.ptr(alphas_cumprod), t=int(t), prev_t=int(prev_t),
                       final_alpha_cumprod=final_alpha_cumprod, eta=eta, clip_sample=int(bool(clip_sample)),
                       clip_sample_range=clip_sample_range)
    L.check(lib.bd_ddim_step(C.byref(d), L.stream()), "bd_ddim_step")
    return (prev, x0) if want_x0 else prev


def to_image(x, nhwc, shape_bchw, want_u8=False):
    """(x/2+0.5).clamp(0,1) -> NHWC float32 [B,H,W,C] (and uint8 round(255 x))."""
    lib = L.load(); _need_cuda(x)
    B, Cc, H, W = shape_bchw
    x = x.contiguous()
    of = torch.empty(B, H, W, Cc, device=x.device)
    ou = torch.empty(B, H, W, Cc, dtype=torch.uint8, device=x.device) if want_u8 else None
    L.check(lib.bd_to_image(L.ptr(x), int(nhwc), Cc, B, Cc, H, W, L.ptr(of), L.ptr(ou), L.stream()), "bd_to_image")
    return (of, ou) if want_u8 else of


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift):
    lib = L.load(); _need_cuda(t)
    t = t.to(torch.int64).contiguous()
    out = torch.empty(t.numel(), dim, device=t.device)
    L.check(lib.bd_timestep_embedding(L.ptr(t), 1, t.numel(), dim, int(flip_sin_to_cos), float(freq_shift), L.ptr(out),
                                      L.stream()), "bd_timestep_embedding")
    return out


# ------------------------------------------------------------------------------------------------ GroupNorm
def gn_fwd(x, gamma, beta, G, eps, silu):
    """x [B,HW,C] (last dim contiguous, row stride = ld).  Returns (y [B,HW,C], mean [B,G], rstd [B,G])."""
    lib = L.load(); _need_cuda(x, gamma, beta)
    B, HW, Cc = x.shape
    y = torch.empty(B, HW, Cc, device=x.device)
    stats = torch.empty(2, B, G, device=x.device)
    ws = workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
    d = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=G, eps=eps, silu=int(silu), x=L.ptr(x), ldx=_ld(x), gamma=L.ptr(gamma),
                    beta=L.ptr(beta), y=L.ptr(y), ldy=Cc, mean=L.ptr(stats[0]), rstd=L.ptr(stats[1]),
                    workspace=L.ptr(ws), workspace_bytes=ws.numel())
    L.check(lib.bd_gn_fwd(C.byref(d), L.stream()), "bd_gn_fwd")
    return y, stats[0], stats[1]


def gn_bwd(x, gamma, beta, mean, rstd, dy, G, silu, dx=None, accumulate=False, with_colsum=False):
    """Returns (dx, dgamma, dbeta) and, with_colsum, the per-sample sum over pixels of dx [B, C]."""
    lib = L.load(); _need_cuda(x, gamma, beta, mean, rstd, dy)
    B, HW, Cc = x.shape
    if dx is None:
        dx = torch.empty(B, HW, Cc, device=x.device)
    dg = torch.empty(Cc, device=x.device); db = torch.empty(Cc, device=x.device)
    ws = workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
    d = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=G, silu=int(silu), x=L.ptr(x), ldx=_ld(x), gamma=L.ptr(gamma), beta=L.ptr(beta),
                    mean=L.ptr(mean), rstd=L.ptr(rstd), dy=L.ptr(dy), lddy=_ld(dy), dx=L.ptr(dx), lddx=_ld(dx),
                    accumulate_dx=int(accumulate), dgamma=L.ptr(dg), dbeta=L.ptr(db), workspace=L.ptr(ws),
                    workspace_bytes=ws.numel())
    cs = torch.empty(B, Cc, device=x.device) if with_colsum else None
    d.dx_colsum = L.ptr(cs); d.ld_colsum = Cc
    L.check(lib.bd_gn_bwd(C.byref(d), L.stream()), "bd_gn_bwd")
    return (dx, dg, db, cs) if with_colsum else (dx, dg, db)


# ------------------------------------------------------------------------------------------------ igemm family
def _conv_out_hw(Hs, Ws, stride, pad_t, pad_l, ups, pad_b=None, pad_r=None):
    Hi, Wi = Hs << ups, Ws << ups
    pad_b = pad_t if pad_b is None else pad_b
    pad_r = pad_l if pad_r is None else pad_r
    return (Hi + pad_t + pad_b - 3) // stride + 1, (Wi + pad_l + pad_r - 3) // stride + 1


def conv3x3_fwd(x, w, bias=None, stride=1, pad=1, ups=0, asym=False, rowbias=None, residual=None, out_scale=1.0, mode=0,
                w_split=None):
    """x [B,Hs,Ws,Cin] NHWC ; w [Cout,3,3,Cin].  asym => F.pad(0,1,0,1) + stride-2 conv with padding 0."""
    lib = L.load(); _need_cuda(x, w, bias, rowbias, residual)
    B, Hs, Ws, Cin = x.shape
    Cout = w.shape[0]
    pt = pl = 0 if asym else pad
    Ho, Wo = _conv_out_hw(Hs, Ws, stride, pt, pl, ups, 1 if asym else None, 1 if asym else None)
    y = torch.empty(B, Ho, Wo, Cout, device=x.device)
    ws = workspace(lib.bd_conv3x3_workspace_bytes(B, Ho, Wo, Hs, Ws, Cin, Cout, ups), x.device)
    d = L.ConvFwdDesc(B=B, Hs=Hs, Ws=Ws, Cin=Cin, Cout=Cout, stride=stride, pad_t=pt, pad_l=pl, ups=ups, Ho=Ho, Wo=Wo,
                      x=L.ptr(x), ldx=_ld(x), w=L.ptr(w), bias=L.ptr(bias), rowbias=L.ptr(rowbias),
                      ld_rowbias=rowbias.stride(0) if rowbias is not None else 0, residual=L.ptr(residual),
                      ldr=_ld(residual) if residual is not None else 0, out_scale=out_scale, y=L.ptr(y), ldy=Cout,
                      workspace=L.ptr(ws), workspace_bytes=ws.numel(), mode=mode)
    if w_split is not None:   # w already split (split_bf16): the weight operand is staged without VALU work
        d.w_split = L.ptr(w_split)
    L.check(lib.bd_conv3x3_fwd(C.byref(d), L.stream()), "bd_conv3x3_fwd")
    return y


def split_bf16(x):
    """The bf16 hi/lo split the bf16x3 kernels apply on the fly, as one int16 tensor [numel/32, 2, 32]
    (block, hi|lo, element): include/bd_hip.h bd_split_bf16."""
    lib = L.load(); _need_cuda(x)
    x = x.contiguous()
    out = torch.empty(x.numel() // 32, 2, 32, dtype=torch.int16, device=x.device)
    L.check(lib.bd_split_bf16(L.ptr(x), x.numel(), L.ptr(out), L.stream()), "bd_split_bf16")
    return out


def split_rows(x, out=None):
    """fp32 [..., C] (last dim contiguous, uniform row stride) -> split planes int16 [rows, C/32, 2, 32]
    (row, 32-channel block, hi|lo, element): include/bd_hip.h bd_split_rows."""
    lib = L.load(); _need_cuda(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if out is None:
        out = torch.empty(rows, Cc // 32, 2, 32, dtype=torch.int16, device=x.device)
    L.check(lib.bd_split_rows(L.ptr(x), _ld(x), rows, Cc, L.ptr(out), Cc, L.stream()), "bd_split_rows")
    return out


def split_rows_ups2(x):
    """x [B,H,W,C] fp32 -> split planes of the nearest x2 upsampling, int16 [B*2H*2W, C/32, 2, 32]."""
    lib = L.load(); _need_cuda(x)
    B, H, W, Cc = x.shape
    out = torch.empty(B * 4 * H * W, Cc // 32, 2, 32, dtype=torch.int16, device=x.device)
    L.check(lib.bd_split_rows_ups2(L.ptr(x), _ld(x), B, H, W, Cc, L.ptr(out), Cc, L.stream()), "bd_split_rows_ups2")
    return out


def split_wT(w):
    """conv weight [Cout,3,3,Cin] fp32 -> split planes of the transpose [Cin,3,3,Cout] (for the data gradient)."""
    lib = L.load(); _need_cuda(w)
    Cout, _, _, Cin = w.shape
    out = torch.empty(Cin, 9 * Cout // 32, 2, 32, dtype=torch.int16, device=w.device)
    L.check(lib.bd_split_wt(L.ptr(w.contiguous()), Cin, Cout, L.ptr(out), L.stream()), "bd_split_wt")
    return out


def conv3x3_ps(x_split, w_split, B, H, W, K, N, direction=1, bias=None, rowbias=None, residual=None, out_scale=1.0,
               out=None, accumulate=False, gn_groups=0):
    """stride-1 pad-1 3x3 convolution (direction +1) / data gradient (-1) on pre-split operands -> fp32 [B,H,W,N].
    gn_groups > 0: also the GroupNorm partials of y from the epilogue -> (y, partials [B, S, gn_groups, 2] fp64), S =
    bd_conv3x3_ps_gn_splits() (raises when the call cannot produce them)."""
    lib = L.load(); _need_cuda(x_split, w_split, bias, rowbias, residual)
    y = torch.empty(B, H, W, N, device=x_split.device) if out is None else out
    part = None
    if gn_groups:
        S = lib.bd_conv3x3_ps_gn_splits(B, H, W, K, N, gn_groups)
        if S <= 0:
            raise ValueError("conv3x3_ps: this call cannot write GroupNorm partials (bd_conv3x3_ps_gn_splits() == 0)")
        part = torch.full((B, S, gn_groups, 2), float("nan"), dtype=torch.float64, device=x_split.device)
    d = L.ConvPsDesc(B=B, H=H, W=W, K=K, N=N, direction=direction, x_split=L.ptr(x_split), ldx=K, w_split=L.ptr(w_split),
                     bias=L.ptr(bias), rowbias=L.ptr(rowbias), ld_rowbias=rowbias.stride(0) if rowbias is not None else 0,
                     residual=L.ptr(residual), ldr=_ld(residual) if residual is not None else 0, out_scale=out_scale,
                     y=L.ptr(y), ldy=_ld(y), accumulate=int(accumulate))
    ws = workspace(lib.bd_conv3x3_ps_workspace_bytes(C.byref(d)), x_split.device, "ps")
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    if part is not None:
        d.gn_part = L.ptr(part); d.gn_groups = gn_groups
    L.check(lib.bd_conv3x3_ps(C.byref(d), L.stream()), "bd_conv3x3_ps")
    return y if part is None else (y, part)


def conv3x3_ps_wgrad(x_split, dy_split, B, H, W, Cin, Cout, with_db=False):
    """dw [Cout,3,3,Cin] (and db [Cout]) of the stride-1 pad-1 conv from split-plane operands."""
    lib = L.load(); _need_cuda(x_split, dy_split)
    dw = torch.empty(Cout, 3, 3, Cin, device=x_split.device)
    db = torch.empty(Cout, device=x_split.device) if with_db else None
    d = L.ConvPsWgradDesc(B=B, H=H, W=W, Cin=Cin, Cout=Cout, x_split=L.ptr(x_split), ldx=Cin, dy_split=L.ptr(dy_split),
                          lddy=Cout, dw=L.ptr(dw), db=L.ptr(db))
    ws = workspace(lib.bd_conv3x3_ps_wgrad_workspace_bytes(C.byref(d)), x_split.device, "ps_wgrad")
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    L.check(lib.bd_conv3x3_ps_wgrad(C.byref(d), L.stream()), "bd_conv3x3_ps_wgrad")
    return (dw, db) if with_db else dw


def upsample_weights(w):
    """conv weight [Cout,3,3,Cin] of an Upsample2D layer -> (E planes [Cout,16,Cin], E^T planes [Cin,16,Cout]) as split planes:
    E[oy][ox] = sum of the taps that read the same source pixel (include/bd_hip.h, phase-decomposed convolutions)."""
    lib = L.load(); _need_cuda(w)
    Cout, _, _, Cin = w.shape
    e = torch.empty(Cout, 16 * Cin // 32, 2, 32, dtype=torch.int16, device=w.device)
    et = torch.empty(Cin, 16 * Cout // 32, 2, 32, dtype=torch.int16, device=w.device)
    L.check(lib.bd_upsample_weights(L.ptr(w.contiguous()), Cin, Cout, L.ptr(e), L.ptr(et), L.stream()), "bd_upsample_weights")
    return e, et


def upsample_conv_fwd(x_split, e_split, B, H, W, Cin, Cout, bias=None):
    """y [B,2H,2W,Cout] = conv3x3(nearest_up2(x)) + bias from the SOURCE-grid split planes of x and the E planes."""
    lib = L.load(); _need_cuda(x_split, e_split, bias)
    y = torch.empty(B, 2 * H, 2 * W, Cout, device=x_split.device)
    d = L.UpsampleConvDesc(B=B, H=H, W=W, Cin=Cin, Cout=Cout, x_split=L.ptr(x_split), ldx=Cin, e_split=L.ptr(e_split), bias=L.ptr(bias),
                           y=L.ptr(y), ldy=Cout)
    L.check(lib.bd_upsample_conv_fwd(C.byref(d), L.stream()), "bd_upsample_conv_fwd")
    return y


def upsample_conv_dgrad(dy_split, et_split, B, H, W, Cin, Cout, out=None, accumulate=False):
    """dx [B,H,W,Cin] of the same layer from the fine-grid split planes of dY [B,2H,2W,Cout] and the E^T planes."""
    lib = L.load(); _need_cuda(dy_split, et_split)
    dx = torch.empty(B, H, W, Cin, device=dy_split.device) if out is None else out
    d = L.UpsampleConvDesc(B=B, H=H, W=W, Cin=Cin, Cout=Cout, dy_split=L.ptr(dy_split), lddy=Cout, et_split=L.ptr(et_split),
                           dx=L.ptr(dx), lddx=_ld(dx), accumulate=int(accumulate))
    ws = workspace(lib.bd_upsample_conv_dgrad_workspace_bytes(C.byref(d)), dy_split.device, "ups_dgrad")
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    L.check(lib.bd_upsample_conv_dgrad(C.byref(d), L.stream()), "bd_upsample_conv_dgrad")
    return dx


def upsample_conv_wgrad(x_split, dy_split, B, H, W, Cin, Cout, with_db=False):
    """dw [Cout,3,3,Cin] (and db) of the same layer from x (source grid) and dY (fine grid), both split planes."""
    lib = L.load(); _need_cuda(x_split, dy_split)
    dw = torch.empty(Cout, 3, 3, Cin, device=x_split.device)
    db = torch.empty(Cout, device=x_split.device) if with_db else None
    d = L.UpsampleConvDesc(B=B, H=H, W=W, Cin=Cin, Cout=Cout, x_split=L.ptr(x_split), ldx=Cin, dy_split=L.ptr(dy_split), lddy=Cout,
                           dw=L.ptr(dw), db=L.ptr(db))
    ws = workspace(lib.bd_upsample_conv_wgrad_workspace_bytes(C.byref(d)), x_split.device, "ups_wgrad")
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    L.check(lib.bd_upsample_conv_wgrad(C.byref(d), L.stream()), "bd_upsample_conv_wgrad")
    return (dw, db) if with_db else dw


def conv3x3_s2_dgrad_ps(dy_split, wT_split, B, Ho, Wo, Cin, Cout, pad=0, out=None, accumulate=False):
    """dx [B,2Ho,2Wo,Cin] of a stride-2 3x3 convolution (pad 0 = F.pad(0,1,0,1) + padding 0; 1 = padding 1) from the split planes
    of dy [B,Ho,Wo,Cout] and the transposed weight planes (split_wT)."""
    lib = L.load(); _need_cuda(dy_split, wT_split)
    dx = torch.empty(B, 2 * Ho, 2 * Wo, Cin, device=dy_split.device) if out is None else out
    d = L.ConvS2DgradDesc(B=B, Ho=Ho, Wo=Wo, Cin=Cin, Cout=Cout, pad=pad, dy_split=L.ptr(dy_split), lddy=Cout, wT_split=L.ptr(wT_split),
                          dx=L.ptr(dx), lddx=_ld(dx), accumulate=int(accumulate))
    L.check(lib.bd_conv3x3_s2_dgrad_ps(C.byref(d), L.stream()), "bd_conv3x3_s2_dgrad_ps")
    return dx


def conv3x3_dgrad(dy, w, x_shape, stride=1, pad=1, ups=0, asym=False, mode=0, w_split=None):
    """Returns dx over the conv's own input grid [B, Hs<<ups, Ws<<ups, Cin]."""
    lib = L.load(); _need_cuda(dy, w)
    B, Hs, Ws, Cin = x_shape
    Cout = w.shape[0]
    pt = pl = 0 if asym else pad
    Ho, Wo = dy.shape[1], dy.shape[2]
    dx = torch.empty(B, Hs << ups, Ws << ups, Cin, device=dy.device)
    ws = workspace(lib.bd_conv3x3_workspace_bytes(B, Ho, Wo, Hs, Ws, Cin, Cout, ups), dy.device)
    d = L.ConvDgradDesc(B=B, Hs=Hs, Ws=Ws, Cin=Cin, Cout=Cout, stride=stride, pad_t=pt, pad_l=pl, ups=ups, Ho=Ho, Wo=Wo,
                        dy=L.ptr(dy), lddy=_ld(dy), w=L.ptr(w), dx=L.ptr(dx), lddx=Cin, accumulate=0,
                        workspace=L.ptr(ws), workspace_bytes=ws.numel(), mode=mode)
    if w_split is not None:
        d.w_split = L.ptr(w_split)
    L.check(lib.bd_conv3x3_dgrad(C.byref(d), L.stream()), "bd_conv3x3_dgrad")
    return dx


def conv3x3_wgrad(x, dy, stride=1, pad=1, ups=0, asym=False, mode=0, with_db=False):
    """dw [Cout,3,3,Cin]; with_db also returns the bias gradient (sum of dy over pixels) fused into the same launch."""
    lib = L.load(); _need_cuda(x, dy)
    B, Hs, Ws, Cin = x.shape
    Cout = dy.shape[-1]
    pt = pl = 0 if asym else pad
    Ho, Wo = dy.shape[1], dy.shape[2]
    dw = torch.empty(Cout, 3, 3, Cin, device=x.device)
    ws = workspace(lib.bd_conv3x3_workspace_bytes(B, Ho, Wo, Hs, Ws, Cin, Cout, ups), x.device)
    d = L.ConvWgradDesc(B=B, Hs=Hs, Ws=Ws, Cin=Cin, Cout=Cout, stride=stride, pad_t=pt, pad_l=pl, ups=ups, Ho=Ho, Wo=Wo,
                        x=L.ptr(x), ldx=_ld(x), dy=L.ptr(dy), lddy=_ld(dy), dw=L.ptr(dw), workspace=L.ptr(ws),
                        workspace_bytes=ws.numel(), mode=mode)
    db = torch.empty(Cout, device=x.device) if with_db else None
    d.db = L.ptr(db)
    L.check(lib.bd_conv3x3_wgrad(C.byref(d), L.stream()), "bd_conv3x3_wgrad")
    return (dw, db) if with_db else dw


def gemm(a, b, trans_a=False, trans_b=True, bias=None, alpha=1.0, tile=0, ksplit=0, mode=0, a_colsum=None):
    """C = alpha * op(a) @ op(b)^T-style product on the igemm engine.
    a: [M,K] (trans_a False) or [K,M] (trans_a True); b: [N,K] (trans_b True, 'weights') or [K,N] (False).
    Batched when a/b are 3-D (same leading batch)."""
    lib = L.load(); _need_cuda(a, b, bias)
    batched = a.dim() == 3
    if not batched:
        a, b = a[None], b[None]
    nb = a.shape[0]
    M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    N = b.shape[1] if trans_b else b.shape[2]
    a = a.contiguous(); b = b.contiguous()
    c = torch.empty(nb, M, N, device=a.device)
    d = L.IgemmDesc()
    d.A.kind = 0; d.A.kc = 0 if trans_a else 1; d.A.p = L.ptr(a); d.A.ld = a.shape[2]; d.A.bs_outer = a.stride(0)
    d.B.kind = 0; d.B.kc = 1 if trans_b else 0; d.B.p = L.ptr(b); d.B.ld = b.shape[2]; d.B.bs_outer = b.stride(0)
    d.M, d.N, d.K = M, N, K
    d.batch_outer, d.batch_inner = nb, 1
    d.C = L.ptr(c); d.ldc = N; d.c_bs_outer = M * N
    d.alpha = alpha; d.out_scale = 1.0; d.bias = L.ptr(bias)
    d.tile = tile; d.ksplit = ksplit; d.mode = mode; d.a_colsum = L.ptr(a_colsum)
    need = lib.bd_igemm_workspace_bytes(C.byref(d))
    ws = workspace(need, a.device)
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    L.check(lib.bd_igemm(C.byref(d), L.stream()), "bd_igemm")
    return c if batched else c[0]


def unsplit_rows(s):
    """Split planes int16 [..., C/32, 2, 32] -> fp32 [..., C] (hi + lo, exact): test / inspection helper, plain torch."""
    hi = (s[..., 0, :].to(torch.int32) << 16).view(torch.float32)
    lo = (s[..., 1, :].to(torch.int32) << 16).view(torch.float32)
    return (hi + lo).reshape(*s.shape[:-3], s.shape[-3] * 32)


def gemm_sp(a_split, b_split, M, N, K, a_kmajor=False, b_kmajor=False, batch=1, bias=None, residual=None, alpha=1.0, out_scale=1.0,
            want_f32=True, want_split=False, out=None, accumulate=False, want_colsum=False):
    """C = out_scale * (alpha * A B^T + bias + residual) on split planes (include/bd_hip.h bd_gemm_sp).  a_split: planes of A [batch, M, K]
    (or [batch, K, M] when a_kmajor); b_split: planes of B [batch, N, K] (or [batch, K, N] when b_kmajor), or un-batched (shared by the
    batch).  Returns (c fp32 or None, c_split or None[, colsum])."""
    lib = L.load(); _need_cuda(a_split, b_split, bias, residual)
    dev = a_split.device
    d = L.GemmSpDesc()
    d.M, d.N, d.K, d.batch = M, N, K, batch
    ar, ac = (K, M) if a_kmajor else (M, K)
    br, bc = (K, N) if b_kmajor else (N, K)
    d.a = L.ptr(a_split); d.lda = ac; d.a_bs = ar * ac if a_split.numel() * 2 == batch * ar * ac * 4 else 0; d.a_kmajor = int(a_kmajor)
    d.b = L.ptr(b_split); d.ldb = bc; d.b_bs = br * bc if b_split.numel() * 2 == batch * br * bc * 4 else 0; d.b_kmajor = int(b_kmajor)
    c = out if out is not None else (torch.empty(batch, M, N, device=dev) if want_f32 else None)
    cs = torch.empty(batch, M, N // 32, 2, 32, dtype=torch.int16, device=dev) if want_split else None
    d.c = L.ptr(c); d.ldc = N; d.c_bs = M * N
    d.c_split = L.ptr(cs); d.ldcs = N; d.cs_bs = M * N
    d.bias = L.ptr(bias)
    if residual is not None:
        residual = residual.contiguous()
        d.residual = L.ptr(residual); d.ldr = N; d.r_bs = M * N
    d.alpha = alpha; d.out_scale = out_scale; d.accumulate = int(accumulate)
    colsum_t = torch.empty(M, device=dev) if want_colsum else None
    d.a_colsum = L.ptr(colsum_t)
    need = lib.bd_gemm_sp_workspace_bytes(C.byref(d))
    ws = workspace(need, dev)
    d.workspace = L.ptr(ws); d.workspace_bytes = ws.numel()
    L.check(lib.bd_gemm_sp(C.byref(d), L.stream()), "bd_gemm_sp")
    return (c, cs, colsum_t) if want_colsum else (c, cs)


# ------------------------------------------------------------------------------------------------ small ops
def colsum(x, rows_per_group):
    lib = L.load(); _need_cuda(x)
    rows, N = x.shape
    groups = (rows + rows_per_group - 1) // rows_per_group
    out = torch.empty(groups, N, device=x.device)
    L.check(lib.bd_colsum(L.ptr(x), _ld(x), rows, N, rows_per_group, L.ptr(out), N, 0, L.stream()), "bd_colsum")
    return out


def sum2x2(du):
    lib = L.load(); _need_cuda(du)
    B, H2, W2, Cc = du.shape
    dx = torch.empty(B, H2 // 2, W2 // 2, Cc, device=du.device)
    L.check(lib.bd_sum2x2(L.ptr(du), Cc, L.ptr(dx), Cc, B, H2 // 2, W2 // 2, Cc, 0, L.stream()), "bd_sum2x2")
    return dx


def attn_sp_fwd(qkv_split, B, heads, scale, want_pt=True, C_=None):
    """Attention core forward on the planes of the QKV projection's output [B*N, 3C/32, 2, 32] (include/bd_hip.h bd_attn_sp_fwd):
    returns (o_split [B*N, C/32, 2, 32], pt_split [B*heads, N, N/32, 2, 32] or None)."""
    lib = L.load(); _need_cuda(qkv_split)
    rows, blocks = qkv_split.shape[0], qkv_split.shape[1]
    N = rows // B
    Cc = C_ or blocks * 32 // 3            # C_: the planes' rows are padded beyond 3C (row stride = blocks * 32)
    o = torch.empty(rows, Cc // 32, 2, 32, dtype=torch.int16, device=qkv_split.device)
    pt = torch.empty(B * heads, N, N // 32, 2, 32, dtype=torch.int16, device=qkv_split.device) if want_pt else None
    d = L.AttnSpDesc(B=B, heads=heads, N=N, dh=Cc // heads, qkv_split=L.ptr(qkv_split), ld=blocks * 32, scale=scale, o_split=L.ptr(o), ldo=Cc,
                     pt_split=L.ptr(pt))
    L.check(lib.bd_attn_sp_fwd(C.byref(d), L.stream()), "bd_attn_sp_fwd")
    return o, pt


def attn_sp_bwd(qkv_split, pt_split, do_split, B, heads, scale):
    """Attention core backward: returns (dqkv_split [B*N, 3C/32, 2, 32], dst_split = planes of scale * dS^T)."""
    lib = L.load(); _need_cuda(qkv_split, pt_split, do_split)
    rows, blocks = qkv_split.shape[0], qkv_split.shape[1]
    N = rows // B
    Cc = blocks * 32 // 3
    dqkv = torch.empty_like(qkv_split)
    dst = torch.empty_like(pt_split)
    d = L.AttnSpDesc(B=B, heads=heads, N=N, dh=Cc // heads, qkv_split=L.ptr(qkv_split), ld=3 * Cc, scale=scale, pt_split=L.ptr(pt_split),
                     do_split=L.ptr(do_split), lddo=Cc, dst_split=L.ptr(dst), dqkv_split=L.ptr(dqkv), lddqkv=3 * Cc)
    L.check(lib.bd_attn_sp_bwd(C.byref(d), L.stream()), "bd_attn_sp_bwd")
    return dqkv, dst


def softmax_fwd(s):
    lib = L.load(); _need_cuda(s)
    s = s.contiguous()
    p = torch.empty_like(s)
    L.check(lib.bd_softmax_fwd(L.ptr(s), L.ptr(p), s.numel() // s.shape[-1], s.shape[-1], L.stream()), "bd_softmax_fwd")
    return p


def softmax_bwd(p, dp):
    lib = L.load(); _need_cuda(p, dp)
    ds = torch.empty_like(p)
    L.check(lib.bd_softmax_bwd(L.ptr(p.contiguous()), L.ptr(dp.contiguous()), L.ptr(ds), p.numel() // p.shape[-1],
                               p.shape[-1], L.stream()), "bd_softmax_bwd")
    return ds


def silu_fwd(x):
    lib = L.load(); _need_cuda(x)
    y = torch.empty_like(x)
    L.check(lib.bd_silu_fwd(L.ptr(x.contiguous()), L.ptr(y), x.numel(), L.stream()), "bd_silu_fwd")
    return y


def silu_bwd(x, dy):
    lib = L.load(); _need_cuda(x, dy)
    dx = torch.empty_like(x)
    L.check(lib.bd_silu_bwd(L.ptr(x.contiguous()), L.ptr(dy.contiguous()), L.ptr(dx), x.numel(), 0, L.stream()), "bd_silu_bwd")
    return dx


LOSS_TYPES = {"l2": 0, "l1": 1, "huber": 2}


def loss_fwd_bwd(pred, target, loss_type="l2", grad_scale=1.0, want_grad=True):
    """pred / target: [..., C] views with the same logical shape (last dim contiguous, uniform row stride).
    Returns (loss 0-dim tensor, dpred contiguous [rows, C] or None)."""
    lib = L.load(); _need_cuda(pred, target)
    if loss_type not in LOSS_TYPES:
        raise NotImplementedError()
    Cc = pred.shape[-1]
    rows = pred.numel() // Cc
    p2 = pred.reshape(rows, Cc) if pred.is_contiguous() else pred
    t2 = target.reshape(rows, Cc) if target.is_contiguous() else target
    loss = torch.empty((), device=pred.device)
    dp = torch.empty(rows, Cc, device=pred.device) if want_grad else None
    ws = workspace(lib.bd_reduce_workspace_bytes(), pred.device, "reduce")
    L.check(lib.bd_loss_fwd_bwd(L.ptr(p2), _ld(p2), L.ptr(t2), _ld(t2), rows, Cc, LOSS_TYPES[loss_type], grad_scale,
                                L.ptr(loss), L.ptr(dp), Cc, L.ptr(ws), L.stream()), "bd_loss_fwd_bwd")
    return loss, dp


def lincomb(terms, coeffs, clip=None, out=None):
    """out = clamp?(sum_j coeffs[j] * terms[j]) (bd_lincomb): up to 6 dense float32 GPU tensors of one shape and layout."""
    lib = L.load(); _need_cuda(*terms)
    k = len(terms)
    if k < 1 or k > 6 or len(coeffs) != k:
        raise ValueError("lincomb: 1..6 terms with one coefficient each")
    t0 = terms[0]
    for t in terms:
        if t.shape != t0.shape or t.stride() != t0.stride() or t.dtype != torch.float32:
            raise ValueError("lincomb: terms must share shape, strides and dtype float32")
    if not t0.permute(*sorted(range(t0.dim()), key=lambda d: -t0.stride(d))).is_contiguous():
        raise ValueError("lincomb: terms must be dense")
    if out is None:
        out = torch.empty_like(t0)
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in terms])
    cs = (C.c_float * k)(*[float(c) for c in coeffs])
    L.check(lib.bd_lincomb(k, ptrs, cs, t0.numel(), 0 if clip is None else 1, 0.0 if clip is None else float(clip), L.ptr(out),
                           L.stream()), "bd_lincomb")
    return out


def anp_apply(params, pert_w, pert_b, items, total_rows, out=None):
    """effective parameters of the ANP-perturbed network (bd_anp_apply): a copy of the flat parameter buffer with every conv weight row scaled
    by its channel's pert_w and every conv bias mapped to pert_w * b + pert_b.  items: int64 device tensor [n, 5] (include/bd_hip.h)."""
    lib = L.load(); _need_cuda(params, pert_w, pert_b, items)
    if out is None:
        out = torch.empty_like(params)
    L.check(lib.bd_anp_apply(L.ptr(params), params.numel(), L.ptr(pert_w), L.ptr(pert_b), L.ptr(items), items.shape[0], int(total_rows),
                             L.ptr(out), L.stream()), "bd_anp_apply")
    return out


def anp_grad(params, grad_eff, items, total_rows, grad_w, grad_b, pert_w=None, row_norm=None):
    """gradient of the ANP perturbation from the ordinary flat weight gradient (bd_anp_grad); row_norm: per-channel gradient norm of the
    layer's own weights in the perturbed network (for the reference's clip norm), needs pert_w."""
    lib = L.load(); _need_cuda(params, grad_eff, items, grad_w, grad_b, pert_w, row_norm)
    L.check(lib.bd_anp_grad(L.ptr(params), L.ptr(grad_eff), L.ptr(items), items.shape[0], int(total_rows), L.ptr(pert_w), L.ptr(grad_w),
                            L.ptr(grad_b), L.ptr(row_norm), L.stream()), "bd_anp_grad")
    return grad_w, grad_b


def ssim(preds, target, data_range=1.0):
    """Mean SSIM (torchmetrics defaults, see bd_hip.h) of two [N,C,H,W] float32 GPU tensors with identical strides (any
    layout: a permuted NHWC buffer is fine).  Returns a 0-dim device tensor; batches with N*C > 65535 are chunked."""
    lib = L.load(); _need_cuda(preds, target)
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError(f"expected two [N,C,H,W] tensors of the same shape, got {tuple(preds.shape)} and {tuple(target.shape)}")
    if preds.dtype != torch.float32 or target.dtype != torch.float32:
        raise TypeError("ssim: float32 tensors required")
    if preds.stride() != target.stride():
        target = target.contiguous(); preds = preds.contiguous()
    N, Cc, H, W = preds.shape
    per = max(1, 65535 // Cc)
    total = torch.zeros((), dtype=torch.float64, device=preds.device)
    for s0 in range(0, N, per):
        pc, tc = preds[s0:s0 + per], target[s0:s0 + per]
        n = pc.shape[0]
        nbytes = lib.bd_ssim_workspace_bytes(n, Cc, H, W)
        ws = workspace(nbytes, preds.device, "ssim")
        out = torch.empty((), device=preds.device)
        sn, sc, sh, sw = pc.stride()
        L.check(lib.bd_ssim(L.ptr(pc), L.ptr(tc), n, Cc, H, W, sn, sc, sh, sw, float(data_range), L.ptr(out), L.ptr(ws), nbytes, L.stream()),
                "bd_ssim")
        if n == N:
            return out
        total += out.double() * n
    return (total / N).float()


def sumsq(g, out=None):
    lib = L.load(); _need_cuda(g)
    if out is None:
        out = torch.empty((), dtype=torch.float64, device=g.device)
    ws = workspace(lib.bd_reduce_workspace_bytes(), g.device, "reduce")
    L.check(lib.bd_sumsq(L.ptr(g), g.numel(), L.ptr(out), L.ptr(ws), L.stream()), "bd_sumsq")
    return out


def adam_clip(p, g, m, v, sumsq_t, step, lr, max_norm=1.0, betas=(0.9, 0.999), eps=1e-8, grad_norm_out=None):
    lib = L.load(); _need_cuda(p, g, m, v, sumsq_t)
    L.check(lib.bd_adam_clip(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(sumsq_t), float(max_norm), float(lr),
                             float(betas[0]), float(betas[1]), float(eps), int(step), L.ptr(grad_norm_out), L.stream()),
            "bd_adam_clip")


def tune_set(name, value, check=True):
    """bd_tune_set (run-time tuning knobs of the kernels, e.g. "ps_wg3_slots"; 0 restores the default) + what the C call cannot do: every plan re-lays
    out its workspace on its next call, so the workspaces the live models pool by size are dropped here (callers used to reach into
    model._ws_pool).  Not to be called between a training forward and its backward: the C side refuses a re-layout while saved activations are live."""
    from . import unet
    rc = L.load().bd_tune_set(name.encode() if isinstance(name, str) else name, int(value))
    for m in list(unet.live_models()):
        m._ws_pool = {}
    if check:
        L.check(rc, "bd_tune_set")
    return rc
