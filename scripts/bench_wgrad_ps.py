"""check + timing of the split-plane weight gradient against the register-staged igemm wgrad"""
import sys
import torch
from baddiffusion_amd import ops
torch.manual_seed(0)
dev = "cuda"
iters = 20
def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (B, S, Cin, Cout) in [(128, 32, 128, 128), (128, 16, 256, 256), (128, 32, 256, 128), (128, 16, 512, 256), (128, 8, 256, 256), (128, 4, 256, 256), (2, 16, 128, 128)]:
    x = torch.randn(B, S, S, Cin, device=dev); dy = torch.randn(B, S, S, Cout, device=dev)
    xs = ops.split_rows(x); dys = ops.split_rows(dy)
    ref, refb = ops.conv3x3_wgrad(x, dy, mode=1, with_db=True)
    new, newb = ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True)
    # exact fp32 reference (slow mode) for error scale
    ex = ops.conv3x3_wgrad(x, dy, mode=0)
    e_new = float((new - ex).abs().max() / ex.abs().max()); e_old = float((ref - ex).abs().max() / ex.abs().max())
    eb = float((newb - dy.sum((0, 1, 2))).abs().max() / dy.sum((0, 1, 2)).abs().max())
    new2 = ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    t_old = timeit(lambda: ops.conv3x3_wgrad(x, dy, mode=1, with_db=True))
    t_new = timeit(lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True))
    print(f"B{B} {S}x{S} {Cin}->{Cout}: rel err vs exact new {e_new:.2e} old {e_old:.2e} db {eb:.2e} deterministic={bool(torch.equal(new, new2))} | "
          f"old {t_old:.1f}us ({fl/t_old/1e6:.0f} TF) new {t_new:.1f}us ({fl/t_new/1e6:.0f} TF)", flush=True)
