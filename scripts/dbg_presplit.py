import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baddiffusion_amd import ops
B, H, Cin, Cout = 1, 16, 128, 128
for (co0, tap0, ci0) in ((0, 4, 0), (0, 4, 5), (3, 4, 8), (70, 4, 33), (1, 4, 1)):
    w = torch.zeros(Cout, 3, 3, Cin, device="cuda"); w[co0, tap0 // 3, tap0 % 3, ci0] = 1.0
    s = ops.split_bf16(w)
    hits = []
    for c in range(Cin):
        x = torch.zeros(B, H, H, Cin, device="cuda"); x[..., c] = 1.0
        y = ops.conv3x3_fwd(x, w, None, mode=1, w_split=s)
        nz = (y[0, 8, 8].abs() > 1e-3).nonzero().flatten().tolist()
        if nz: hits.append((c, nz, [round(float(y[0, 8, 8, n]), 3) for n in nz]))
    print((co0, tap0, ci0), "->", hits)
