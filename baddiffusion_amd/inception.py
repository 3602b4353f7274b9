"""FID feature extractor on libbd_hip.so (SURVEY f-3): pytorch_fid's InceptionV3 up to pool3.

Replaces `from pytorch_fid.inception import InceptionV3` + `model(batch)[0]` of /root/reference/fid_score.py:53, 91-148, 255
(pytorch-fid==0.2.1, requirements.txt; the package is not part of the reference tree).  The published graph -- torchvision's
inception_v3 with pytorch_fid's FIDInceptionA / C / E_1 / E_2 blocks, bilinear resize to 299 x 299 and x -> 2x - 1 in front, global
average pool behind -- runs as a fixed launch sequence of four kernels (include/bd_hip.h: bd_conv2d_nhwc, bd_pool2d_nhwc,
bd_resize_bilinear_nhwc, bd_global_avgpool_nhwc), NHWC fp32 with exact fp32 products:
  * BatchNorm (inference statistics, eps 1e-3) is folded into each convolution's weight and bias once, at load time;
  * every branch writes its output at its channel offset of the block's output buffer: no torch.cat;
  * images stay on the device: uint8 NHWC batches (what the sampler produces) or float [N, 3, H, W] in [0, 1].
Weights: a state dict with pytorch_fid's key names (pt_inception-2015-12-05-6726825d.pth); `load_fid_weights()` reads the file
named by BD_FID_WEIGHTS.  The file is a third-party asset that does not travel with the reference or this repo: without it
`measure()` reports FID null together with the reason.  PARITY UNPINNED against pytorch_fid itself (absent here): the GPU path is
compared with oracle/inception_ref.py on seeded random weights (tests/test_hip_round4.py).  No CPU path.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _lib as L

BN_EPS = 0.001
POOL3_DIM = 2048


def _block_a(pre, cin, pf):
    return [(pre + ".branch1x1", cin, 64, (1, 1)), (pre + ".branch5x5_1", cin, 48, (1, 1)), (pre + ".branch5x5_2", 48, 64, (5, 5)),
            (pre + ".branch3x3dbl_1", cin, 64, (1, 1)), (pre + ".branch3x3dbl_2", 64, 96, (3, 3)), (pre + ".branch3x3dbl_3", 96, 96, (3, 3)),
            (pre + ".branch_pool", cin, pf, (1, 1))]


def _block_c(pre, cin, c7):
    return [(pre + ".branch1x1", cin, 192, (1, 1)), (pre + ".branch7x7_1", cin, c7, (1, 1)), (pre + ".branch7x7_2", c7, c7, (1, 7)),
            (pre + ".branch7x7_3", c7, 192, (7, 1)), (pre + ".branch7x7dbl_1", cin, c7, (1, 1)), (pre + ".branch7x7dbl_2", c7, c7, (7, 1)),
            (pre + ".branch7x7dbl_3", c7, c7, (1, 7)), (pre + ".branch7x7dbl_4", c7, c7, (7, 1)), (pre + ".branch7x7dbl_5", c7, 192, (1, 7)),
            (pre + ".branch_pool", cin, 192, (1, 1))]


def _block_e(pre, cin):
    return [(pre + ".branch1x1", cin, 320, (1, 1)), (pre + ".branch3x3_1", cin, 384, (1, 1)), (pre + ".branch3x3_2a", 384, 384, (1, 3)),
            (pre + ".branch3x3_2b", 384, 384, (3, 1)), (pre + ".branch3x3dbl_1", cin, 448, (1, 1)), (pre + ".branch3x3dbl_2", 448, 384, (3, 3)),
            (pre + ".branch3x3dbl_3a", 384, 384, (1, 3)), (pre + ".branch3x3dbl_3b", 384, 384, (3, 1)), (pre + ".branch_pool", cin, 192, (1, 1))]


def conv_layers():
    """(name, Cin, Cout, (kh, kw)) of the 94 BasicConv2d layers in state-dict order."""
    t = [("Conv2d_1a_3x3", 3, 32, (3, 3)), ("Conv2d_2a_3x3", 32, 32, (3, 3)), ("Conv2d_2b_3x3", 32, 64, (3, 3)),
         ("Conv2d_3b_1x1", 64, 80, (1, 1)), ("Conv2d_4a_3x3", 80, 192, (3, 3))]
    t += _block_a("Mixed_5b", 192, 32) + _block_a("Mixed_5c", 256, 64) + _block_a("Mixed_5d", 288, 64)
    t += [("Mixed_6a.branch3x3", 288, 384, (3, 3)), ("Mixed_6a.branch3x3dbl_1", 288, 64, (1, 1)), ("Mixed_6a.branch3x3dbl_2", 64, 96, (3, 3)),
          ("Mixed_6a.branch3x3dbl_3", 96, 96, (3, 3))]
    for pre, c7 in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):
        t += _block_c(pre, 768, c7)
    t += [("Mixed_7a.branch3x3_1", 768, 192, (1, 1)), ("Mixed_7a.branch3x3_2", 192, 320, (3, 3)), ("Mixed_7a.branch7x7x3_1", 768, 192, (1, 1)),
          ("Mixed_7a.branch7x7x3_2", 192, 192, (1, 7)), ("Mixed_7a.branch7x7x3_3", 192, 192, (7, 1)), ("Mixed_7a.branch7x7x3_4", 192, 192, (3, 3))]
    t += _block_e("Mixed_7b", 1280) + _block_e("Mixed_7c", 2048)
    return t


def state_dict_manifest():
    """key -> shape of the state dict this network loads (pytorch_fid / torchvision names; `num_batches_tracked` entries and
    the unused 1008-way `fc` head are accepted and ignored)."""
    m = OrderedDict()
    for name, cin, cout, (kh, kw) in conv_layers():
        m[name + ".conv.weight"] = (cout, cin, kh, kw)
        for k in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.bn.{k}"] = (cout,)
    m["fc.weight"] = (1008, POOL3_DIM)
    m["fc.bias"] = (1008,)
    return m


class FIDInceptionV3:
    """`features = net(images)`: images [N, 3, H, W] float in [0, 1] or uint8 [N, H, W, 3] on the GPU -> [N, 2048] float32."""

    def __init__(self, state_dict=None, device="cuda", batch_size=200):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FIDInceptionV3 runs on the GPU only (libbd_hip.so); there is no CPU path")
        self._lib = L.load()
        # fid_score.py:55 feeds 50 images at a time; samples are independent, so the chunk is a throughput knob only: at 50 the 8 x 8 / 17 x 17
        # stages launch too few tiles for 256 CUs (59.7 TFLOP/s over the 94 convolutions), at 200 they fill the chip (83.8; scripts/bench_fid_layers.py)
        self.batch_size = int(batch_size)
        self._w = {}
        self._kc = os.environ.get("BD_FID_CONV", "kc") != "old"
        self.flops = 0.0                        # algorithmic flops of the convolutions launched so far (2 * MACs; bench.py reads and resets it)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        want = state_dict_manifest()
        missing = [k for k in want if k not in sd and not k.startswith("fc.")]
        extra = [k for k in sd if k not in want and not k.endswith("num_batches_tracked") and not k.startswith("AuxLogits.")]
        if strict and (missing or extra):
            raise RuntimeError(f"FIDInceptionV3: state dict mismatch: missing {missing[:4]}... unexpected {extra[:4]}...")
        for k, shp in want.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"FIDInceptionV3: size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shp)}")
        w = {}
        for name, cin, cout, _ in conv_layers():
            W = sd[name + ".conv.weight"].detach().to(torch.float64)
            g, b = sd[name + ".bn.weight"].detach().double(), sd[name + ".bn.bias"].detach().double()
            mu, var = sd[name + ".bn.running_mean"].detach().double(), sd[name + ".bn.running_var"].detach().double()
            s = g / torch.sqrt(var + BN_EPS)
            # fold in fp64, store fp32: [Cout, Cin, KH, KW] -> [KH, KW, Cout, Cin] (K contiguous: bd_conv2d_desc.w_kc = 1, the double-buffered
            # 128 x 64 kernel of round 5; BD_FID_CONV=old keeps the [KH, KW, Cin, Cout] layout and the 64 x 64 kernel for A/B)
            Wf = (W * s.view(-1, 1, 1, 1)).permute(2, 3, 0, 1) if self._kc else (W * s.view(-1, 1, 1, 1)).permute(2, 3, 1, 0)
            Wf = Wf.contiguous().to(torch.float32)
            w[name] = (Wf.to(self.device), (b - mu * s).to(torch.float32).to(self.device))
        self._w = w
        return self

    # ---- layer launches -----------------------------------------------------------------------------------------------------
    def _conv(self, x, name, k, stride=1, pad=(0, 0), out=None):
        Wf, bias = self._w[name]
        KH, KW, Cin, Cout = (Wf.shape[0], Wf.shape[1], Wf.shape[3], Wf.shape[2]) if self._kc else Wf.shape
        assert (KH, KW) == tuple(k) and x.shape[-1] == Cin, (name, tuple(Wf.shape), tuple(x.shape))
        B, H, W_, _ = x.shape
        Ho, Wo = (H + 2 * pad[0] - KH) // stride + 1, (W_ + 2 * pad[1] - KW) // stride + 1
        if out is None:
            out = torch.empty(B, Ho, Wo, Cout, device=x.device)
        assert tuple(out.shape) == (B, Ho, Wo, Cout), (name, tuple(out.shape), (B, Ho, Wo, Cout))
        d = L.Conv2dDesc(x=x.data_ptr(), ldx=x.stride(2), w=Wf.data_ptr(), bias=bias.data_ptr(), y=out.data_ptr(), ldy=out.stride(2),
                         B=B, H=H, W=W_, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride_h=stride, stride_w=stride, pad_h=pad[0], pad_w=pad[1], relu=1, w_kc=int(self._kc))
        L.check(self._lib.bd_conv2d_nhwc(C.byref(d), L.stream()), "bd_conv2d_nhwc")
        self.flops += 2.0 * B * Ho * Wo * Cout * KH * KW * Cin
        return out

    def _pool(self, x, kernel, stride, pad, mode, out=None, count_include_pad=False):
        B, H, W_, Cc = x.shape
        Ho, Wo = (H + 2 * pad - kernel) // stride + 1, (W_ + 2 * pad - kernel) // stride + 1
        if out is None:
            out = torch.empty(B, Ho, Wo, Cc, device=x.device)
        L.check(self._lib.bd_pool2d_nhwc(x.data_ptr(), x.stride(2), out.data_ptr(), out.stride(2), B, H, W_, Cc, kernel, stride, pad,
                                         0 if mode == "max" else 1, int(count_include_pad), L.stream()), "bd_pool2d_nhwc")
        return out

    def _a(self, x, pre, pf):
        B, H, W_, _ = x.shape
        out = torch.empty(B, H, W_, 224 + pf, device=x.device)
        self._conv(x, pre + ".branch1x1", (1, 1), out=out[..., 0:64])
        t = self._conv(x, pre + ".branch5x5_1", (1, 1))
        self._conv(t, pre + ".branch5x5_2", (5, 5), pad=(2, 2), out=out[..., 64:128])
        t = self._conv(x, pre + ".branch3x3dbl_1", (1, 1))
        t = self._conv(t, pre + ".branch3x3dbl_2", (3, 3), pad=(1, 1))
        self._conv(t, pre + ".branch3x3dbl_3", (3, 3), pad=(1, 1), out=out[..., 128:224])
        self._conv(self._pool(x, 3, 1, 1, "avg"), pre + ".branch_pool", (1, 1), out=out[..., 224:])
        return out

    def _b(self, x, pre):
        B, H, W_, Cc = x.shape
        Ho, Wo = (H - 3) // 2 + 1, (W_ - 3) // 2 + 1
        out = torch.empty(B, Ho, Wo, 384 + 96 + Cc, device=x.device)
        self._conv(x, pre + ".branch3x3", (3, 3), stride=2, out=out[..., 0:384])
        t = self._conv(x, pre + ".branch3x3dbl_1", (1, 1))
        t = self._conv(t, pre + ".branch3x3dbl_2", (3, 3), pad=(1, 1))
        self._conv(t, pre + ".branch3x3dbl_3", (3, 3), stride=2, out=out[..., 384:480])
        self._pool(x, 3, 2, 0, "max", out=out[..., 480:])
        return out

    def _c(self, x, pre):
        B, H, W_, _ = x.shape
        out = torch.empty(B, H, W_, 768, device=x.device)
        self._conv(x, pre + ".branch1x1", (1, 1), out=out[..., 0:192])
        t = self._conv(x, pre + ".branch7x7_1", (1, 1))
        t = self._conv(t, pre + ".branch7x7_2", (1, 7), pad=(0, 3))
        self._conv(t, pre + ".branch7x7_3", (7, 1), pad=(3, 0), out=out[..., 192:384])
        t = self._conv(x, pre + ".branch7x7dbl_1", (1, 1))
        t = self._conv(t, pre + ".branch7x7dbl_2", (7, 1), pad=(3, 0))
        t = self._conv(t, pre + ".branch7x7dbl_3", (1, 7), pad=(0, 3))
        t = self._conv(t, pre + ".branch7x7dbl_4", (7, 1), pad=(3, 0))
        self._conv(t, pre + ".branch7x7dbl_5", (1, 7), pad=(0, 3), out=out[..., 384:576])
        self._conv(self._pool(x, 3, 1, 1, "avg"), pre + ".branch_pool", (1, 1), out=out[..., 576:])
        return out

    def _d(self, x, pre):
        B, H, W_, Cc = x.shape
        Ho, Wo = (H - 3) // 2 + 1, (W_ - 3) // 2 + 1
        out = torch.empty(B, Ho, Wo, 320 + 192 + Cc, device=x.device)
        t = self._conv(x, pre + ".branch3x3_1", (1, 1))
        self._conv(t, pre + ".branch3x3_2", (3, 3), stride=2, out=out[..., 0:320])
        t = self._conv(x, pre + ".branch7x7x3_1", (1, 1))
        t = self._conv(t, pre + ".branch7x7x3_2", (1, 7), pad=(0, 3))
        t = self._conv(t, pre + ".branch7x7x3_3", (7, 1), pad=(3, 0))
        self._conv(t, pre + ".branch7x7x3_4", (3, 3), stride=2, out=out[..., 320:512])
        self._pool(x, 3, 2, 0, "max", out=out[..., 512:])
        return out

    def _e(self, x, pre, max_pool):
        B, H, W_, _ = x.shape
        out = torch.empty(B, H, W_, 2048, device=x.device)
        self._conv(x, pre + ".branch1x1", (1, 1), out=out[..., 0:320])
        t = self._conv(x, pre + ".branch3x3_1", (1, 1))
        self._conv(t, pre + ".branch3x3_2a", (1, 3), pad=(0, 1), out=out[..., 320:704])
        self._conv(t, pre + ".branch3x3_2b", (3, 1), pad=(1, 0), out=out[..., 704:1088])
        t = self._conv(x, pre + ".branch3x3dbl_1", (1, 1))
        t = self._conv(t, pre + ".branch3x3dbl_2", (3, 3), pad=(1, 1))
        self._conv(t, pre + ".branch3x3dbl_3a", (1, 3), pad=(0, 1), out=out[..., 1088:1472])
        self._conv(t, pre + ".branch3x3dbl_3b", (3, 1), pad=(1, 0), out=out[..., 1472:1856])
        self._conv(self._pool(x, 3, 1, 1, "max" if max_pool else "avg"), pre + ".branch_pool", (1, 1), out=out[..., 1856:])
        return out

    # ---- forward ------------------------------------------------------------------------------------------------------------
    def _forward_chunk(self, imgs):
        """imgs: uint8 [n, H, W, 3] or float32 [n, H, W, 3] (NHWC storage) on the device -> [n, 2048]."""
        n, H, W_, Cc = imgs.shape
        assert Cc == 3, "FID Inception takes RGB images"
        x = torch.empty(n, 299, 299, 3, device=imgs.device)
        L.check(self._lib.bd_resize_bilinear_nhwc(imgs.data_ptr(), int(imgs.dtype == torch.uint8), x.data_ptr(), n, H, W_, 3, 299, 299,
                                                  2.0, -1.0, L.stream()), "bd_resize_bilinear_nhwc")
        x = self._conv(x, "Conv2d_1a_3x3", (3, 3), stride=2)
        x = self._conv(x, "Conv2d_2a_3x3", (3, 3))
        x = self._conv(x, "Conv2d_2b_3x3", (3, 3), pad=(1, 1))
        x = self._pool(x, 3, 2, 0, "max")
        x = self._conv(x, "Conv2d_3b_1x1", (1, 1))
        x = self._conv(x, "Conv2d_4a_3x3", (3, 3))
        x = self._pool(x, 3, 2, 0, "max")
        x = self._a(x, "Mixed_5b", 32)
        x = self._a(x, "Mixed_5c", 64)
        x = self._a(x, "Mixed_5d", 64)
        x = self._b(x, "Mixed_6a")
        for pre in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = self._c(x, pre)
        x = self._d(x, "Mixed_7a")
        x = self._e(x, "Mixed_7b", False)
        x = self._e(x, "Mixed_7c", True)
        feats = torch.empty(n, POOL3_DIM, device=imgs.device)
        L.check(self._lib.bd_global_avgpool_nhwc(x.data_ptr(), x.stride(2), feats.data_ptr(), n, x.shape[1] * x.shape[2], POOL3_DIM, L.stream()),
                "bd_global_avgpool_nhwc")
        return feats

    @torch.no_grad()
    def __call__(self, images):
        if not self._w:
            raise RuntimeError("FIDInceptionV3: no weights loaded (load_state_dict / load_fid_weights)")
        if not torch.is_tensor(images) or not images.is_cuda:
            raise RuntimeError("FIDInceptionV3: device tensors required (no CPU fallback)")
        if images.dtype == torch.uint8:
            if images.dim() != 4 or images.shape[-1] != 3:
                raise ValueError("uint8 images must be [N, H, W, 3]")
            nhwc = images.contiguous()
        else:
            if images.dim() != 4 or images.shape[1] != 3:
                raise ValueError("float images must be [N, 3, H, W] in [0, 1]")
            perm = images.float().permute(0, 2, 3, 1)
            nhwc = perm if perm.is_contiguous() else perm.contiguous()       # layout change only; the arithmetic is in the kernels
        outs = [self._forward_chunk(nhwc[s:s + self.batch_size]) for s in range(0, nhwc.shape[0], self.batch_size)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def load_fid_weights(path=None, device="cuda"):
    """FIDInceptionV3 with the weights of the file `path` / $BD_FID_WEIGHTS (pytorch_fid's pt_inception-2015-12-05-6726825d.pth:
    a plain state dict).  Returns None when no file is configured or it does not exist -- callers report FID as unavailable."""
    path = path or os.environ.get("BD_FID_WEIGHTS")
    if not path or not os.path.exists(path):
        return None
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return FIDInceptionV3(sd, device=device)
