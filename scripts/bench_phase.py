"""Launch the phase-decomposed convolutions (conv_ph.hip) and their literal counterparts at the CIFAR step's shapes, a few times each;
run under `rocprofv3 --kernel-trace --stats` and read the kernel durations (the Python wrappers are host-bound).
usage: python scripts/bench_phase.py [B] [reps]"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda"
for H, C in ((16, 256), (8, 256), (4, 256)):
    x = torch.randn(B, H, H, C, device=dev); dy = torch.randn(B, 2 * H, 2 * H, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05; bias = torch.randn(C, device=dev)
    xs, dys, xu = ops.split_rows(x), ops.split_rows(dy), ops.split_rows_ups2(x)
    e, et = ops.upsample_weights(w)
    ws, wts = ops.split_bf16(w), ops.split_wT(w)
    for _ in range(reps):
        ops.upsample_conv_fwd(xs, e, B, H, H, C, C, bias=bias)
        ops.upsample_conv_dgrad(dys, et, B, H, H, C, C)
        ops.upsample_conv_wgrad(xs, dys, B, H, H, C, C, with_db=True)
        ops.conv3x3_ps(xu, ws, B, 2 * H, 2 * H, C, C, 1, bias=bias)          # literal forms on the upsampled grid
        ops.conv3x3_ps(dys, wts, B, 2 * H, 2 * H, C, C, -1)
        ops.conv3x3_ps_wgrad(xu, dys, B, 2 * H, 2 * H, C, C, with_db=True)
    torch.cuda.synchronize()
    print(f"upsample {H}->{2*H} C={C}: GFLOP literal {2*B*4*H*H*C*9*C/1e9:.1f}, phase {2*B*H*H*C*16*C/1e9:.1f}", flush=True)
for Ho, C in ((16, 128), (8, 256), (4, 256)):
    dy = torch.randn(B, Ho, Ho, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    dys, wt = ops.split_rows(dy), ops.split_wT(w)
    for _ in range(reps):
        ops.conv3x3_s2_dgrad_ps(dys, wt, B, Ho, Ho, C, C, pad=0)
        ops.conv3x3_dgrad(dy, w, (B, 2 * Ho, 2 * Ho, C), stride=2, pad=0, asym=True, mode=1)
    torch.cuda.synchronize()
    print(f"stride-2 dgrad {2*Ho}->{Ho} C={C}: GFLOP {2*B*Ho*Ho*C*9*C/1e9:.1f}", flush=True)
