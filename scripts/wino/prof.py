"""one shape on bd_conv3x3_wino (BD_WINO_V selects the kernel version), a few launches, for rocprofv3 --pmc.  usage: prof.py S Cin Cout [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from baddiffusion_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wino_probe as wp
S, Cin, Cout = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
x = torch.randn(B, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 30
bias = torch.randn(Cout, device="cuda")
u = wp.wino_weights(w, 1)
y = torch.empty(B, S, S, Cout, device="cuda")
for _ in range(3):
    wp.conv3x3_wino(x, u, bias=bias, out=y)
torch.cuda.synchronize()
