"""Effective bandwidth of the GroupNorm entry points at the shapes of the two train steps and of the sampling chunk.

Per shape: microseconds per call (HIP events over `reps` back-to-back calls on the current stream, a 300 MB scratch fill between repetitions is NOT
done: the working set of every shape listed is either far above the 256 MiB Infinity Cache or is what the step itself would find warm) and the
algorithmic bytes moved (forward: x read twice on the large-image path / once on the resident path, planes written; backward: x and dy, dx written, planes
when asked) over that time.  Run on the GPU box:  python scripts/probes/gn_bw.py [cifar|celeba|sample]
"""
import ctypes as C
import sys

import torch

from baddiffusion_amd import _lib as L
from baddiffusion_amd import ops


def run(B, HW, Cc, G=32, planes=True, reps=20):
    dev = torch.device("cuda:0")
    lib = L.load()
    x = torch.randn(B, HW, Cc, device=dev)
    dy = torch.randn(B, HW, Cc, device=dev)
    gamma = torch.randn(Cc, device=dev); beta = torch.randn(Cc, device=dev)
    stats = torch.empty(2, B, G, device=dev)
    ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), dev)
    ys = torch.empty(B * HW * Cc * 2, dtype=torch.int16, device=dev)
    y = None if planes else torch.empty(B, HW, Cc, device=dev)
    f = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=G, eps=1e-5, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(gamma), beta=L.ptr(beta),
                    y=L.ptr(y), ldy=Cc, mean=L.ptr(stats[0]), rstd=L.ptr(stats[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel(),
                    y_split=L.ptr(ys) if planes else None, ldys=Cc)
    dx = torch.empty(B, HW, Cc, device=dev)
    dxs = torch.empty(B * HW * Cc * 2, dtype=torch.int16, device=dev)
    dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
    pp = torch.empty(B, 2, Cc, device=dev)
    b = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=G, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(gamma), beta=L.ptr(beta), mean=L.ptr(stats[0]),
                    rstd=L.ptr(stats[1]), dy=L.ptr(dy), lddy=Cc, dx=L.ptr(dx), lddx=Cc, accumulate_dx=0, dgamma=L.ptr(dg), dbeta=L.ptr(db),
                    workspace=L.ptr(ws), workspace_bytes=ws.numel(), dx_split=L.ptr(dxs) if planes else None, lddxs=Cc,
                    param_partials=L.ptr(pp))
    st = L.stream()

    def timed(fn):
        for _ in range(3):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    tf = timed(lambda: L.check(lib.bd_gn_fwd(C.byref(f), st), "fwd"))
    tb = timed(lambda: L.check(lib.bd_gn_bwd(C.byref(b), st), "bwd"))
    n = B * HW * Cc * 4
    resident = bool(lib.bd_gn_resident(B, HW, Cc, G)) if hasattr(lib, "bd_gn_resident") else None
    return tf, tb, n, resident


SHAPES = {
    # (B, HW, C): GroupNorm inputs of the CIFAR step (B = 128, forward runs as two half batches of 64)
    "cifar": [(64, 1024, 128), (128, 1024, 128), (128, 1024, 256), (128, 1024, 384), (128, 256, 256), (128, 256, 512), (128, 64, 256),
              (128, 64, 512), (128, 16, 512)],
    "sample": [(512, 1024, 128), (512, 1024, 256), (512, 1024, 384), (512, 256, 256), (512, 256, 512), (512, 64, 512)],
    # 256 x 256 step, B = 4 (forward as two half batches of 2)
    "pmc": [(128, 1024, 128), (128, 1024, 256), (128, 1024, 384), (128, 256, 384), (512, 1024, 128), (4, 65536, 128)],      # scripts/probes/gn_pmc.sh
    "c384": [(64, 1024, 384), (128, 1024, 384), (512, 1024, 384), (128, 256, 384)],
    "celeba": [(2, 65536, 128), (4, 65536, 128), (4, 65536, 256), (4, 16384, 128), (4, 16384, 256), (4, 4096, 256), (4, 4096, 512),
               (4, 1024, 256), (4, 1024, 512), (4, 1024, 768), (4, 256, 512), (4, 256, 1024), (4, 64, 512), (4, 64, 1024)],
}

if __name__ == "__main__":
    which = sys.argv[1:] or ["cifar", "sample", "celeba"]
    print(f"{'set':8s} {'B':>4s} {'HW':>6s} {'C':>5s} {'MB':>7s} | {'fwd us':>8s} {'GB/s(2x)':>9s} {'GB/s(3x)':>9s} | {'bwd us':>8s} {'GB/s(4x)':>9s} {'GB/s(6x)':>9s}")
    for w in which:
        for (B, HW, Cc) in SHAPES[w]:
            tf, tb, n, res = run(B, HW, Cc, reps=2 if w == "pmc" else 20)
            # forward: 2x = x once + planes; 3x = x twice + planes (large-image path).  backward: 4x = x, dy, dx, planes; 6x = x, dy twice
            print(f"{w:8s} {B:4d} {HW:6d} {Cc:5d} {n / 1e6:7.1f} | {tf:8.1f} {2 * n / tf / 1e3:9.0f} {3 * n / tf / 1e3:9.0f} | "
                  f"{tb:8.1f} {4 * n / tb / 1e3:9.0f} {6 * n / tb / 1e3:9.0f}", flush=True)
