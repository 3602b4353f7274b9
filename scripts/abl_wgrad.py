"""Ablation timings: needs a library built with -DBD_PS_ABLATION on conv_ps.hip, BD_PS_ABLATE=<bits> selects what is
removed (1 steady-state DMA, 2 MFMAs, 4 LDS fragment reads).  Timing only, results are wrong by design: split-plane wgrad at the step's main shapes"""
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
out = []
for (B, S, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 32, 256, 128), (128, 16, 512, 256), (128, 8, 256, 256)]:
    x = torch.randn(B, S, S, Cin, device=dev); dy = torch.randn(B, S, S, Cout, device=dev)
    xs = ops.split_rows(x); dys = ops.split_rows(dy)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    t = timeit(lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True))
    out.append(f"{S}x{S} {Cin}->{Cout}: {t:.1f}us {fl/t/1e6:.0f}TF")
print(" | ".join(out), flush=True)
