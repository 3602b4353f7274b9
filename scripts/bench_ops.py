"""Micro-benchmark single igemm shapes of the CIFAR UNet step (B=128) with hipEvents.
usage: python scripts/bench_ops.py [mode 0|1] [reps] [which: fwd,dgrad,wgrad,...]"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fwd", "dgrad", "wgrad"]
B = 128
SHAPES = [(32, 128, 128), (16, 256, 256), (8, 256, 256), (4, 256, 256), (32, 256, 128), (16, 512, 256), (4, 512, 256)]
dev = "cuda"


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for H, Cin, Cout in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) / 30
    dy = torch.randn(B, H, H, Cout, device=dev); bias = torch.randn(Cout, device=dev)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    out = [f"{H:3d}^2 {Cin:4d}->{Cout:4d}"]
    if "fwd" in which:
        t = timeit(lambda: ops.conv3x3_fwd(x, w, bias, mode=mode)); out.append(f"fwd {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF")
    if "dgrad" in which:
        t = timeit(lambda: ops.conv3x3_dgrad(dy, w, (B, H, H, Cin), mode=mode)); out.append(f"dgrad {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF")
    if "wgrad" in which:
        t = timeit(lambda: ops.conv3x3_wgrad(x, dy, mode=mode)); out.append(f"wgrad {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF")
    print("  ".join(out), flush=True)
