"""ctypes binding of the stand-alone Winograd probe (scripts/wino/conv_wino.hip -> libbd_wino_probe.so).  NOT part of the product library: the op was
prototyped in round 5, measured slower than conv_ps3 (docs/EXPERIMENTS.md 8.2) and taken out of libbd_hip.so / include/bd_hip.h in round 6.
    python scripts/wino/wino_probe.py          # build
    python scripts/wino/check.py               # fp64 check + timing against conv_ps3 (needs a GPU)"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, "libbd_wino_probe.so")
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvWinoDesc(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("C", i32), ("N", i32), ("x", vp), ("ldx", i64), ("u_planes", vp), ("bias", vp),
                ("rowbias", vp), ("ld_rowbias", i64), ("residual", vp), ("ldr", i64), ("out_scale", f32), ("y", vp), ("ldy", i64)]


def build(force=False):
    src = os.path.join(HERE, "conv_wino.hip")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        from baddiffusion_amd import build as B
        cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", LIB, "-L" + B._torch_lib_dir(), "-lamdhip64"]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (maps libamdhip64)
        C.CDLL(os.path.join(ROOT, "baddiffusion_amd", "libbd_hip.so"), mode=C.RTLD_GLOBAL)   # bd::prof_on / bd::set_error
        _lib = C.CDLL(build())
        _lib.bd_conv3x3_wino_supported.argtypes = [i32] * 5
        _lib.bd_conv3x3_wino.argtypes = [C.POINTER(ConvWinoDesc), vp]
        _lib.bd_wino_weights.argtypes = [vp, i32, i32, i32, vp, vp]
    return _lib


def wino_weights(w, direction=1):
    """conv weight [Cout,3,3,Cin] fp32 -> Winograd planes U = G g G^T: int16 [C/16, 16, N, 32] (16 bf16 hi | 16 bf16 lo)"""
    import torch
    from baddiffusion_amd import _lib as L
    Cout, _, _, Cin = w.shape
    N, K = (Cout, Cin) if direction > 0 else (Cin, Cout)
    u = torch.empty(K // 16, 16, N, 32, dtype=torch.int16, device=w.device)
    L.check(load().bd_wino_weights(L.ptr(w.contiguous()), Cin, Cout, direction, L.ptr(u), L.stream()), "bd_wino_weights")
    return u


def conv3x3_wino(x, u, bias=None, rowbias=None, residual=None, out_scale=1.0, out=None):
    """Winograd F(2x2,3x3) stride-1 pad-1 convolution: x fp32 NHWC [B,H,W,C], u = wino_weights(w) -> fp32 [B,H,W,N]"""
    import torch
    from baddiffusion_amd import _lib as L
    B, H, W, K = x.shape
    N = u.shape[2]
    y = torch.empty(B, H, W, N, device=x.device) if out is None else out
    ld = lambda t: t.stride(-2)
    d = ConvWinoDesc(B=B, H=H, W=W, C=K, N=N, x=L.ptr(x), ldx=ld(x), u_planes=L.ptr(u), bias=L.ptr(bias), rowbias=L.ptr(rowbias),
                     ld_rowbias=rowbias.stride(0) if rowbias is not None else 0, residual=L.ptr(residual),
                     ldr=ld(residual) if residual is not None else 0, out_scale=out_scale, y=L.ptr(y), ldy=ld(y))
    L.check(load().bd_conv3x3_wino(C.byref(d), L.stream()), "bd_conv3x3_wino")
    return y


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
