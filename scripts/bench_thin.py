"""Stand-alone timing of the 3-channel convolutions (conv_in / conv_out: forward, data gradient, weight gradient) with events.
usage: python scripts/bench_thin.py [B] [S] [reps]      (BD_THIN_DIRECT=0 -> the implicit-GEMM / old thin kernels, for A/B)"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
C = 128
dev = "cuda"


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x3 = torch.randn(B, S, S, 3, device=dev); xc = torch.randn(B, S, S, C, device=dev)
w_in = torch.randn(C, 3, 3, 3, device=dev); b_in = torch.randn(C, device=dev)
w_out = torch.randn(3, 3, 3, C, device=dev); b_out = torch.randn(3, device=dev)
dy_c = torch.randn(B, S, S, C, device=dev); dy_3 = torch.randn(B, S, S, 3, device=dev)
mb = B * S * S * C * 4 / 1e6
for name, fn in (("conv_in fwd", lambda: ops.conv3x3_fwd(x3, w_in, b_in, mode=1)),
                 ("conv_in wgrad", lambda: ops.conv3x3_wgrad(x3, dy_c, mode=1, with_db=True)),
                 ("conv_out fwd", lambda: ops.conv3x3_fwd(xc, w_out, b_out, mode=1)),
                 ("conv_out dgrad", lambda: ops.conv3x3_dgrad(dy_3, w_out, (B, S, S, C), mode=1)),
                 ("conv_out wgrad", lambda: ops.conv3x3_wgrad(xc, dy_3, mode=1, with_db=True))):
    t = timeit(fn)
    print(f"{name:16s} {t:7.1f} us   {mb / t * 1e6 / 1e6:6.2f} TB/s of the wide tensor ({mb:.0f} MB)", flush=True)
