"""one conv shape, a few launches, for rocprofv3 --pmc.  usage: prof_conv.py mode which H Cin Cout"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
mode = int(sys.argv[1]); which = sys.argv[2]; H, Cin, Cout = (int(v) for v in sys.argv[3:6])
B = 128
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 30
dy = torch.randn(B, H, H, Cout, device="cuda"); bias = torch.randn(Cout, device="cuda")
for _ in range(3):
    if which == "fwd": ops.conv3x3_fwd(x, w, bias, mode=mode)
    elif which == "dgrad": ops.conv3x3_dgrad(dy, w, (B, H, H, Cin), mode=mode)
    else: ops.conv3x3_wgrad(x, dy, mode=mode)
torch.cuda.synchronize()
