import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
x = torch.randn(1024, device="cuda")
print("tiny silu        %.2f us" % timeit(lambda: ops.silu_fwd(x)))
for M, N, K in [(128, 128, 32), (128, 128, 1024), (32768, 128, 32), (131072, 128, 32), (131072, 128, 128), (131072, 128, 1152)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    for mode in (0, 1):
        print(f"gemm_nt {M}x{N}x{K} mode{mode} %.2f us" % timeit(lambda: ops.gemm(a, b, mode=mode)))
