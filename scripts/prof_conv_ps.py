"""one conv shape on the LDS-DMA kernels, a few launches, for rocprofv3 --pmc.  usage: prof_conv_ps.py which H Cin Cout [B]
which = fwd | dgrad | wgrad"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
which = sys.argv[1]; H, Cin, Cout = (int(v) for v in sys.argv[2:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 30
dy = torch.randn(B, H, H, Cout, device="cuda"); bias = torch.randn(Cout, device="cuda")
xs, dys, ws, wts = ops.split_rows(x), ops.split_rows(dy), ops.split_bf16(w), ops.split_wT(w)
y = torch.empty(B, H, H, Cout if which == "fwd" else Cin, device="cuda")
for _ in range(3):
    if which == "fwd": ops.conv3x3_ps(xs, ws, B, H, H, Cin, Cout, 1, bias=bias, out=y)
    elif which == "dgrad": ops.conv3x3_ps(dys, wts, B, H, H, Cout, Cin, -1, out=y)
    else: ops.conv3x3_ps_wgrad(xs, dys, B, H, H, Cin, Cout, with_db=True)
torch.cuda.synchronize()
