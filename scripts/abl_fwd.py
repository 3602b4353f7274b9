"""Ablation timings: needs a library built with -DBD_PS_ABLATION on conv_ps.hip, BD_PS_ABLATE=<bits> selects what is
removed (1 steady-state DMA, 2 MFMAs, 4 LDS fragment reads).  Timing only, results are wrong by design: split-plane forward conv at the step's main shapes"""
import torch
from baddiffusion_amd import ops
dev = "cuda"
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
out = []
for (B, S, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 32, 256, 128), (128, 16, 512, 256)]:
    x = torch.randn(B, S, S, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) / 30
    xs, ws = ops.split_rows(x), ops.split_bf16(w)
    y = torch.empty(B, S, S, Cout, device=dev)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    t = timeit(lambda: ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, out=y))
    out.append(f"{S}x{S} {Cin}->{Cout}: {t:.1f}us {fl/t/1e6:.0f}TF")
print(" | ".join(out), flush=True)
