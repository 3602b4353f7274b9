"""UNet2DModel -- drop-in for the class BadDiffusion gets from its vendored diffusers
(/root/reference/diffusers/src/diffusers/models/unet_2d.py:36-326), running on libbd_hip.so.

Same constructor arguments, `.config`, `forward(sample, timestep, return_dict)` -> `.sample` / tuple,
deprecated `.in_channels` / `.sample_size` attributes (used at baddiffusion.py:410,512) and the same
`state_dict()` keys / logical shapes, so `google/ddpm-*` checkpoints load.

Physically the parameters are ONE flat fp32 nn.Parameter (`self.flat`) laid out by the C plan
(conv weights [O][kh][kw][I]); autograd sees the whole network as a single Function whose backward
is `bd_unet_backward`.  There is no PyTorch implementation of the network here: without the HIP
library or without a GPU tensor, forward raises.
"""
import os
import weakref

import numpy as np
import ctypes as C
import math
from collections import OrderedDict
from contextlib import contextmanager

import torch
from torch import nn

from . import _lib as L
from . import ops


class FrozenConfig(dict):
    """Attribute-style config (diffusers' FrozenDict allows `config.x = v` through name mangling --
    model.py:639-641 relies on it -- so plain attribute assignment is allowed here too)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class UNet2DOutput:
    def __init__(self, sample):
        self.sample = sample


COMPUTE_MODES = {"f32": 0, "bf16x3": 1}   # include/bd_hip.h: bd_compute_mode

_SUPPORTED_DOWN = ("DownBlock2D", "AttnDownBlock2D")
_SUPPORTED_UP = ("UpBlock2D", "AttnUpBlock2D")


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: forward = bd_unet_forward(training), backward = bd_unet_backward."""

    @staticmethod
    def forward(ctx, flat, x_nhwc, t, model):
        out, ws = model._run_forward(flat, x_nhwc, t, training=True)
        ctx.model = model
        ctx.ws = ws
        ctx.save_for_backward(flat, x_nhwc)
        return out

    @staticmethod
    def backward(ctx, dout):
        flat, x_nhwc = ctx.saved_tensors
        model = ctx.model
        grads = model._run_backward(flat, x_nhwc, dout.contiguous(), ctx.ws)
        model._release_ws(ctx.ws)
        ctx.ws = None
        return grads, None, None, None


_LIVE = weakref.WeakSet()     # every model with a plan: ops.tune_set drops their pooled workspaces (sizes follow the knobs)


def live_models():
    return list(_LIVE)


class UNet2DModel(nn.Module):
    config_name = "config.json"

    def __init__(self, sample_size=None, in_channels=3, out_channels=3, center_input_sample=False,
                 time_embedding_type="positional", freq_shift=0, flip_sin_to_cos=True,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1,
                 downsample_padding=1, act_fn="silu", attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 resnet_time_scale_shift="default", add_attention=True, class_embed_type=None, num_class_embeds=None,
                 max_chunk=2048, compute_mode=None, **unused):
        super().__init__()
        # ---- loud failures for what the reference class supports but BadDiffusion never uses -------------
        if len(down_block_types) != len(up_block_types):
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        for bt in down_block_types:
            if bt not in _SUPPORTED_DOWN:
                raise NotImplementedError(f"{bt} is not on the BadDiffusion hot path (supported: {_SUPPORTED_DOWN})")
        for bt in up_block_types:
            if bt not in _SUPPORTED_UP:
                raise NotImplementedError(f"{bt} is not on the BadDiffusion hot path (supported: {_SUPPORTED_UP})")
        if time_embedding_type != "positional":
            raise NotImplementedError("only time_embedding_type='positional' is supported")
        if act_fn not in ("silu", "swish"):
            raise NotImplementedError(f"act_fn={act_fn} is not supported (silu only)")
        if resnet_time_scale_shift != "default":
            raise NotImplementedError("resnet_time_scale_shift must be 'default'")
        if not add_attention:
            raise NotImplementedError("add_attention=False is not supported")
        if class_embed_type is not None or num_class_embeds is not None:
            raise NotImplementedError("class conditioning is not on the BadDiffusion hot path")
        if isinstance(sample_size, (tuple, list)):
            if sample_size[0] != sample_size[1]:
                raise NotImplementedError("non-square sample_size is not supported")
            sample_size = sample_size[0]
        if sample_size is None:
            raise ValueError("sample_size is required")

        self.config = FrozenConfig(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            center_input_sample=center_input_sample, time_embedding_type=time_embedding_type, freq_shift=freq_shift,
            flip_sin_to_cos=flip_sin_to_cos, down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            mid_block_scale_factor=mid_block_scale_factor, downsample_padding=downsample_padding, act_fn=act_fn,
            attention_head_dim=attention_head_dim, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
            resnet_time_scale_shift=resnet_time_scale_shift, add_attention=add_attention,
            class_embed_type=class_embed_type, num_class_embeds=num_class_embeds)
        # inference chunk (samples are independent).  `max_chunk` is the upper bound; the chunk actually used is the largest
        # one whose workspace fits the device memory that is free at the time (effective_chunk), halved again if the allocation
        # fails -- B = 2048 wants 62 GiB on the CIFAR topology.  BD_MAX_CHUNK = A/B knob.
        self.max_chunk = int(os.environ.get("BD_MAX_CHUNK", max_chunk))

        lib = L.load()
        c = L.UnetConfig()
        c.sample_size, c.in_channels, c.out_channels = sample_size, in_channels, out_channels
        n = len(block_out_channels)
        c.num_blocks = n
        for i in range(n):
            c.block_out_channels[i] = block_out_channels[i]
            c.down_attn[i] = int(down_block_types[i] == "AttnDownBlock2D")
            c.up_attn[i] = int(up_block_types[i] == "AttnUpBlock2D")
        c.layers_per_block = layers_per_block
        c.downsample_padding = int(downsample_padding)
        if downsample_padding not in (0, 1):
            raise NotImplementedError("downsample_padding must be 0 or 1")
        c.flip_sin_to_cos = int(bool(flip_sin_to_cos))
        c.freq_shift = float(freq_shift)
        c.norm_eps = float(norm_eps)
        c.norm_num_groups = int(norm_num_groups)
        c.attention_head_dim = int(attention_head_dim or 0)
        c.mid_block_scale_factor = float(mid_block_scale_factor)
        if compute_mode is None:   # matrix-product arithmetic: split-bf16 (default) or bit-exact fp32 (include/bd_hip.h)
            compute_mode = os.environ.get("BD_COMPUTE_MODE", "bf16x3")
        if compute_mode not in COMPUTE_MODES:
            raise ValueError(f"compute_mode must be one of {sorted(COMPUTE_MODES)}, got {compute_mode!r}")
        c.compute_mode = COMPUTE_MODES[compute_mode]
        self.compute_mode = compute_mode
        h = C.c_void_p()
        L.check(lib.bd_unet_create(C.byref(c), C.byref(h)), "bd_unet_create")
        self._plan = h
        self._lib = lib

        # parameter table: key -> (offset, logical shape, layout)
        self._table = OrderedDict()
        for i in range(lib.bd_unet_num_tensors(h)):
            name = C.c_char_p(); off = C.c_int64(); rank = C.c_int(); shp = (C.c_int64 * 4)(); lay = C.c_int()
            L.check(lib.bd_unet_param_info(h, i, C.byref(name), C.byref(off), C.byref(rank), shp, C.byref(lay)))
            self._table[name.value.decode()] = (off.value, tuple(shp[: rank.value]), lay.value)
        self.num_flat = lib.bd_unet_num_params(h)
        # alignment pads between / behind the tensors: the only elements of a gradient buffer that backward does not write
        ivals = sorted((off, off + int(np.prod(shp))) for off, shp, _ in self._table.values())
        self._pads, end = [], 0
        for lo, hi in ivals:
            if lo > end:
                self._pads.append((end, lo))
            end = max(end, hi)
        if end < self.num_flat:
            self._pads.append((end, self.num_flat))
        if os.environ.get("BD_AUX_STREAM", "1") == "0":
            lib.bd_unet_set_aux_stream(h, 0)
        self.flat = nn.Parameter(torch.zeros(self.num_flat))
        self._segments = None
        self._ws_pool = {}
        self.chunks_used = set()      # batch sizes the inference forwards actually ran with since the caller last cleared it (measure(): score.json)
        _LIVE.add(self)
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters / state dict
    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self._lib.bd_unet_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def set_compute_mode(self, mode):
        """'f32' (exact fp32 MFMA) or 'bf16x3' (split-bf16, ~2^-16 relative per product, 3 MFMAs on the bf16 pipe)."""
        L.check(self._lib.bd_unet_set_compute_mode(self._plan, COMPUTE_MODES[mode]), "bd_unet_set_compute_mode")
        if getattr(self, "compute_mode", mode) != mode:
            self._ws_pool = {}        # the workspace bound depends on which kernels the mode selects: pooled buffers may be too small
        self.compute_mode = mode
        return self

    def set_aux_stream(self, enabled=True):
        """weight-gradient GEMMs of backward on the plan's second stream (default on); BD_AUX_STREAM=0 turns it off."""
        L.check(self._lib.bd_unet_set_aux_stream(self._plan, int(bool(enabled))), "bd_unet_set_aux_stream")
        return self

    def _logical_view(self, flat, key):
        off, shape, layout = self._table[key]
        n = math.prod(shape)
        v = flat[off: off + n]
        if layout == 1:  # stored [O][kh][kw][I] -> logical OIHW
            O, I, kh, kw = shape
            return v.view(O, kh, kw, I).permute(0, 3, 1, 2)
        return v.view(shape)

    @torch.no_grad()
    def reset_parameters(self):
        """nn.Conv2d / nn.Linear / nn.GroupNorm default initialisation (what the reference's
        `weight_reset`, model.py:647-652, re-applies): kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)."""
        flat = self.flat.data
        for key, (off, shape, layout) in self._table.items():
            n = math.prod(shape)
            mod = key.rsplit(".", 1)[0].rsplit(".", 1)[-1]
            is_norm = "norm" in mod
            if is_norm:
                flat[off: off + n].fill_(1.0 if key.endswith("weight") else 0.0)
            else:
                wkey = key[: -len("bias")] + "weight" if key.endswith("bias") else key
                fan_in = math.prod(self._table[wkey][1][1:])
                bound = 1.0 / math.sqrt(fan_in)
                flat[off: off + n].uniform_(-bound, bound)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kw):
        sd = OrderedDict() if destination is None else destination
        flat = self.flat if keep_vars else self.flat.detach()
        for key in self._table:
            sd[prefix + key] = self._logical_view(flat, key).contiguous()
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._table if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._table]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UNet2DModel: missing keys {missing[:5]}..., "
                               f"unexpected keys {unexpected[:5]}...")
        for key, (off, shape, layout) in self._table.items():
            if key not in state_dict:
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {key}: checkpoint {tuple(src.shape)} vs model {tuple(shape)}")
            self._logical_view(self.flat.data, key).copy_(src.to(self.flat.device, torch.float32))
        self._reset_static_cache()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def named_logical_parameters(self):
        """(key, logical-view) pairs in state-dict order (views of the flat parameter)."""
        for key in self._table:
            yield key, self._logical_view(self.flat, key)

    def logical_grads(self, flat_grad=None):
        g = self.flat.grad if flat_grad is None else flat_grad
        return OrderedDict((k, self._logical_view(g, k)) for k in self._table)

    @property
    def in_channels(self):
        return self.config.in_channels

    @property
    def sample_size(self):
        return self.config.sample_size

    @property
    def device(self):
        return self.flat.device

    @property
    def dtype(self):
        return self.flat.dtype

    # ------------------------------------------------------------------ execution
    def workspace_bytes(self, B, training):
        return self._lib.bd_unet_workspace_bytes(self._plan, int(B), int(bool(training)))

    def effective_chunk(self, B):
        """Inference chunk for a batch of B: min(B, max_chunk) when its workspace fits in 85 % of the currently free device memory (a
        pooled workspace of that size counts as free: it is reused); otherwise the largest rung of the FIXED ladder max_chunk / 2^k that
        does.  The plan picks kernels by batch (two-pipeline threshold, small-layer K-splits), so the chunk decides the fp32 summation
        order: a ladder makes a reduced chunk reproducible for a given B instead of depending on how many bytes happened to be free, the
        reduction is logged once, and `last_chunk` records what ran (score.json / the bench line carry it)."""
        want = max(1, min(int(B), self.max_chunk))
        dev = self.flat.device
        chunk = want
        if dev.type == "cuda" and not self._ws_pool.get((chunk, False, str(dev))):
            # (a pooled workspace of this size exists: nothing to allocate, no driver query on the sampling loop's path)
            free, _ = torch.cuda.mem_get_info(dev)
            free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # the caching allocator's idle blocks
            fits = lambda c: self._ws_pool.get((c, False, str(dev))) or self.workspace_bytes(c, False) <= 0.85 * free
            if not fits(chunk):
                rung = self.max_chunk
                while rung > 1 and (rung >= want or not fits(rung)):
                    rung = (rung + 1) // 2
                chunk = max(1, rung)
        if chunk < want and not getattr(self, "_chunk_warned", False):
            import warnings
            self._chunk_warned = True
            warnings.warn(f"UNet2DModel: inference chunk reduced from {want} to {chunk} samples (workspace does not fit the free device "
                          f"memory); kernel selection and fp32 summation order follow the chunk, so samples / scores may differ in the last digits")
        self.last_chunk = chunk
        return chunk

    def _reset_static_cache(self):
        """forget the prepared (split / transposed) weight planes of a static_weights() block: the parameters changed in place, or the
        workspace they were built in is gone (the C side keys its cache on pointers + batch only)"""
        if getattr(self, "_plan", None):
            self._lib.bd_unet_reset_static_cache(self._plan)

    @contextmanager
    def static_weights(self):
        """Inside this block the parameters are promised constant: inference forwards skip the per-forward weight preprocessing
        after the first one (bd_unet_set_static_weights; the sampling loops of pipelines.py run 50-1000 evaluations)."""
        L.check(self._lib.bd_unet_set_static_weights(self._plan, 1), "bd_unet_set_static_weights")
        try:
            yield self
        finally:
            L.check(self._lib.bd_unet_set_static_weights(self._plan, 0), "bd_unet_set_static_weights")

    def _acquire_ws(self, B, training):
        key = (B, bool(training), str(self.flat.device))
        pool = self._ws_pool.setdefault(key, [])
        if pool:
            return pool.pop()
        nbytes = self.workspace_bytes(B, training)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=self.flat.device)
        buf._bd_key = key
        return buf

    def _release_ws(self, buf):
        self._ws_pool.setdefault(buf._bd_key, []).append(buf)

    def _run_forward(self, flat, x_nhwc, t, training):
        if not flat.is_cuda:
            raise RuntimeError("UNet2DModel runs on the GPU only (libbd_hip.so); move the model with .to('cuda')")
        B, H, W, Cin = x_nhwc.shape
        S = self.config.sample_size
        if (H, W) != (S, S) or Cin != self.config.in_channels:
            raise ValueError(f"expected input [B,{self.config.in_channels},{S},{S}], got [B,{Cin},{H},{W}]")
        out = torch.empty(B, S, S, self.config.out_channels, device=flat.device)
        ws = self._acquire_ws(B, training)
        t_stride = 1 if t.numel() == B and B > 1 else (1 if t.numel() == B else 0)
        if t.numel() not in (1, B):
            raise ValueError(f"timestep must have 1 or {B} elements")
        if t.numel() == 1:
            t_stride = 0
        L.check(self._lib.bd_unet_forward(self._plan, B, int(training), flat.data_ptr(), x_nhwc.data_ptr(), Cin,
                                          t.data_ptr(), t_stride, out.data_ptr(), self.config.out_channels,
                                          ws.data_ptr(), ws.numel(), L.stream()), "bd_unet_forward")
        if not training:
            # stream-ordered reuse is safe: later launches on this stream run after these kernels
            self._release_ws(ws)
            ws = None
        out._bd_t = t  # keep the timestep tensor alive until the kernels ran
        return out, ws

    def _run_backward(self, flat, x_nhwc, dout, ws, grads=None):
        B = x_nhwc.shape[0]
        if grads is None:   # backward writes every parameter's gradient exactly once: only the alignment pads need zeros
            grads = torch.empty(self.num_flat, device=flat.device)
            for lo, hi in self._pads:
                grads[lo:hi].zero_()
        L.check(self._lib.bd_unet_backward(self._plan, B, flat.data_ptr(), x_nhwc.data_ptr(), x_nhwc.shape[-1],
                                           dout.data_ptr(), dout.shape[-1], grads.data_ptr(), ws.data_ptr(), ws.numel(),
                                           L.stream()), "bd_unet_backward")
        return grads

    def segments(self):
        """[(lo, hi)] flat-gradient ranges, in the order backward finalises them (for DP bucketing)."""
        if self._segments is None:
            n = self._lib.bd_unet_num_segments(self._plan)
            self._segments = n
        return self._segments

    def _prep_inputs(self, sample, timestep):
        if not torch.is_tensor(sample) or not sample.is_cuda:
            raise RuntimeError("UNet2DModel.forward: `sample` must be a GPU tensor (no CPU fallback on the hot path)")
        if self.config.center_input_sample:
            sample = 2 * sample - 1.0
        B = sample.shape[0]
        if sample.dim() != 4:
            raise ValueError("sample must be [B,C,H,W]")
        # logical NCHW -> physical NHWC.  A channels_last tensor (or a permuted NHWC buffer) is already NHWC.
        perm = sample.permute(0, 2, 3, 1)
        x_nhwc = perm if perm.is_contiguous() else ops.nchw_to_nhwc(sample.float())
        if x_nhwc.dtype != torch.float32:
            x_nhwc = x_nhwc.float()
        if not torch.is_tensor(timestep):
            t = torch.tensor([timestep], dtype=torch.int64, device=sample.device)
        else:
            t = timestep.reshape(-1).to(device=sample.device, dtype=torch.int64)
        if t.numel() == 1 and B > 1:
            pass  # broadcast in-kernel (t_stride 0)
        return x_nhwc, t.contiguous()

    def forward(self, sample, timestep, class_labels=None, return_dict=True):
        if class_labels is not None:
            raise NotImplementedError("class conditioning is not supported")
        x_nhwc, t = self._prep_inputs(sample, timestep)
        B = x_nhwc.shape[0]
        need_grad = torch.is_grad_enabled() and (self.flat.requires_grad or x_nhwc.requires_grad)
        if need_grad:
            if x_nhwc.requires_grad:
                raise NotImplementedError("gradient w.r.t. the UNet input is not provided (the reference never needs it)")
            out = _UNetFn.apply(self.flat, x_nhwc, t, self)
        else:
            out = self._forward_chunked(x_nhwc, t, self.effective_chunk(B))
        sample_out = out.permute(0, 3, 1, 2)     # logical NCHW view of the NHWC buffer (channels_last strides)
        if not return_dict:
            return (sample_out,)
        return UNet2DOutput(sample=sample_out)

    def _forward_chunked(self, x_nhwc, t, chunk, flat=None):
        """Inference over `chunk`-sample pieces (samples are independent: chunking is exact and bounds the workspace).  A failed
        workspace allocation halves the chunk and starts over instead of failing the call.  `flat`: another parameter buffer of this
        topology (anp.py: the effective weights of the perturbed network)."""
        B = x_nhwc.shape[0]
        flat = self.flat.detach() if flat is None else flat
        while True:
            try:
                if B <= chunk:
                    self.chunks_used.add(int(B))
                    return self._run_forward(flat, x_nhwc, t, False)[0]
                outs = []
                for s in range(0, B, chunk):
                    tt = t if t.numel() == 1 else t[s: s + chunk]
                    self.chunks_used.add(int(min(chunk, B - s)))
                    outs.append(self._run_forward(flat, x_nhwc[s: s + chunk], tt, False)[0])
                return torch.cat(outs, 0)
            except torch.cuda.OutOfMemoryError:
                if chunk <= 1:
                    raise
                self._ws_pool = {k: v for k, v in self._ws_pool.items() if k[1]}     # drop pooled inference workspaces
                torch.cuda.empty_cache()
                self._reset_static_cache()       # the allocator may hand the same address back: never match a stale prepared workspace
                rung = self.max_chunk
                while rung > 1 and rung >= min(chunk, B):
                    rung = (rung + 1) // 2
                import warnings
                warnings.warn(f"UNet2DModel: workspace allocation for {min(chunk, B)} samples failed; continuing with chunks of {rung}")
                chunk = self.last_chunk = rung
                self.max_chunk = min(self.max_chunk, chunk)

    # ------------------------------------------------------------------ diffusers-layout I/O (SURVEY f-2)
    def config_dict(self):
        d = dict(self.config)
        d["_class_name"] = "UNet2DModel"
        d["_diffusers_version"] = "0.16.0.dev0"
        return d


def unet_from_config(cfg, **kw):
    """Build a UNet2DModel from any object carrying the UNet2DModel config fields as attributes."""
    return UNet2DModel(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                       block_out_channels=cfg.block_out_channels, down_block_types=cfg.down_block_types,
                       up_block_types=cfg.up_block_types, layers_per_block=cfg.layers_per_block,
                       downsample_padding=cfg.downsample_padding, flip_sin_to_cos=cfg.flip_sin_to_cos,
                       freq_shift=cfg.freq_shift, norm_eps=cfg.norm_eps, norm_num_groups=cfg.norm_num_groups,
                       attention_head_dim=cfg.attention_head_dim, mid_block_scale_factor=cfg.mid_block_scale_factor, **kw)

