"""Build libbd_hip.so (gfx950) in-tree with hipcc.

    python -m baddiffusion_amd.build [--force]

The library is linked against the libamdhip64.so that PyTorch-ROCm already maps into the process
(torch/lib), so the C-ABI entry points run on the same HIP runtime, streams and allocations as the
torch tensors whose data_ptr() they are handed.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libbd_hip.so")
ARCH = "gfx950"
SOURCES = [  # (file, extra flags)
    ("elementwise.hip", ["-ffp-contract=off"]),
    ("groupnorm.hip", []),
    ("igemm.hip", []),
    ("conv_ps.hip", ["-DBD_PS_ABLATION"] if os.environ.get("BD_BUILD_ABLATION") == "1" else []),
    ("conv_ph.hip", []),
    ("gemm_sp.hip", []),
    ("attn_sp.hip", ["-DBD_AS_ABLATION"] if os.environ.get("BD_BUILD_ABLATION") == "1" else []),
    ("metrics.hip", ["-ffp-contract=off"]),
    ("inception.hip", ["-ffp-contract=off"]),
    ("conv.cpp", ["-x", "hip"]),
    ("conv_thin.hip", ["-fno-slp-vectorize"]),   # SLP pairs the fp32 FMAs into v_pk_fma_f32 with splatted coefficients: 2x the registers, spills
    ("unet_plan.cpp", ["-x", "hip"]),
    ("prof.cpp", ["-x", "hip"]),
]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "bd_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-L" + _torch_lib_dir(), "-lamdhip64"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
