"""timing only (ablation runs): python scripts/bench_conv_ps2.py"""
import torch
from baddiffusion_amd import ops
dev = "cuda"
for (B, S, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 16, 512, 256)]:
    x = torch.randn(B, S, S, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
    ws = ops.split_bf16(w); xs = ops.split_rows(x)
    y = torch.empty(B, S, S, Cout, device=dev)
    f = lambda: ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, out=y)
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    print(f"  B{B} {S}x{S} {Cin}->{Cout}: {t:.1f} us ({2.0*B*S*S*Cin*Cout*9/t/1e6:.0f} TF)", flush=True)
