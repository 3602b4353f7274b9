"""Timing of the split-plane weight gradient (conv_ps_wgrad3_kernel + its slab reduce) at the three main shapes of the CIFAR step, for
scripts/abl_wg3.sh: the library loaded is one of the compile-time ablation variants (BD_WG3_ABL, conv_ps.hip), so results are wrong by design."""
import os

import torch

from baddiffusion_amd import ops

dev = "cuda"


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


out = []
for (B, S, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 16, 512, 256)]:
    x = torch.randn(B, S, S, Cin, device=dev); dy = torch.randn(B, S, S, Cout, device=dev)
    xs = ops.split_rows(x); dys = ops.split_rows(dy)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    t = timeit(lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True))
    out.append(f"{S}x{S} {Cin}->{Cout}: {t:.1f}us {fl / t / 1e6:.0f}TF")
print("ablate", os.environ.get("BD_WG3_ABL", "0"), " | ".join(out), flush=True)
