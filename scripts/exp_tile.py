import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baddiffusion_amd import ops
dev = torch.device("cuda")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K, ta, tb) in ((32768, 1024, 1024, False, True), (131072, 128, 1152, False, True), (32768, 1024, 1024, False, False), (1024, 1024, 32768, True, False)):
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    for tile in (128, 64):
        us = t(lambda: ops.gemm(a, b, ta, tb, mode=1, tile=tile))
        print(M, N, K, ta, tb, "tile", tile, round(us, 1), "us", round(2.0 * M * N * K / us / 1e6, 1), "TF")
