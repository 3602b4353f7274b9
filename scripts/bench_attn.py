"""fused attention forward vs the unfused GEMM + softmax + GEMM chain"""
import torch
from baddiffusion_amd import ops
dev = "cuda"
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (B, N, Cc, heads) in [(128, 256, 256, 1), (64, 256, 256, 1), (8, 256, 512, 1)]:
    qkv = torch.randn(B, N, 3 * Cc, device=dev)
    dh = Cc // heads; scale = dh ** -0.5
    q, k, v = (qkv[..., i * Cc:(i + 1) * Cc].contiguous() for i in range(3))
    def unfused():
        S = ops.gemm(q, k, alpha=scale, mode=1)
        P = ops.softmax_fwd(S)
        return ops.gemm(P, v, trans_b=False, mode=1)
    t_f = timeit(lambda: ops.attn_fwd(qkv, heads, scale))
    t_fp = timeit(lambda: ops.attn_fwd(qkv, heads, scale, want_p=True))
    t_u = timeit(unfused)
    fl = 4.0 * B * N * N * Cc
    print(f"B{B} N{N} C{Cc}: fused {t_f:.1f} us ({fl/t_f/1e6:.0f} TF)  fused+P {t_fp:.1f} us  unfused {t_u:.1f} us", flush=True)
