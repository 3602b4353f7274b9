"""Time the attention / 1x1 GEMM shapes of the CIFAR UNet step on the igemm engine (bf16x3 unless BD_MODE=0)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baddiffusion_amd import ops
mode = int(os.environ.get("BD_MODE", "1"))
dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cases = [("qkv fwd NT", 1, 32768, 768, 256, False, True), ("QK^T NT b128", 128, 256, 256, 256, False, True),
         ("PV NN b128", 128, 256, 256, 256, False, False), ("proj NT", 1, 32768, 256, 256, False, True),
         ("qkv dgrad NN", 1, 32768, 256, 768, False, False), ("qkv wgrad TN", 1, 768, 256, 32768, True, False),
         ("dV TN b128", 128, 256, 256, 256, True, False),
         ("sc32 fwd NT", 1, 131072, 128, 256, False, True), ("sc32 dgrad NN", 1, 131072, 256, 128, False, False),
         ("sc32 wgrad TN", 1, 128, 256, 131072, True, False), ("sc16 fwd NT", 1, 32768, 256, 512, False, True),
         ("big NT", 1, 32768, 1024, 1024, False, True)]
for name, nb, M, N, K, ta, tb in cases:
    a = torch.randn((nb, K, M) if ta else (nb, M, K), device=dev)
    b = torch.randn((nb, N, K) if tb else (nb, K, N), device=dev)
    if nb == 1: a, b = a[0], b[0]
    us = t(lambda: ops.gemm(a, b, ta, tb, mode=mode))
    fl = 2.0 * nb * M * N * K
    by = 4.0 * nb * (M * K + N * K + M * N)
    print(f"{name:16s} {us:8.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e3:7.1f} GB/s")
