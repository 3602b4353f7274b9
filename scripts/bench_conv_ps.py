"""Stand-alone check + timing of the pre-split / LDS-DMA convolution against the register-staged split-bf16 igemm.
usage: python scripts/bench_conv_ps.py [iters]"""
import sys
import torch
from baddiffusion_amd import ops

torch.manual_seed(0)
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(128, 32, 128, 128), (128, 16, 256, 256), (128, 16, 512, 256), (128, 8, 256, 256), (128, 8, 512, 256), (128, 4, 256, 256), (128, 4, 512, 256), (2, 16, 64, 128), (3, 4, 128, 128)]


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (B, S, Cin, Cout) in SHAPES:
    x = torch.randn(B, S, S, Cin, device=dev)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    rb = torch.randn(B, Cout, device=dev)
    res = torch.randn(B, S, S, Cout, device=dev)
    ws = ops.split_bf16(w)
    xs = ops.split_rows(x)
    # forward, three epilogue variants
    ok = True
    for kw in (dict(), dict(rowbias=rb), dict(residual=res, out_scale=0.7)):
        ref = ops.conv3x3_fwd(x, w, bias, mode=1, w_split=ws, **kw)
        new = ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias, **kw)
        ok &= bool(torch.equal(ref, new))
        if not torch.equal(ref, new):
            rel = float((ref - new).abs().max() / ref.abs().max())
            ok = rel < 2e-6 if not ok else ok   # K-split summation orders differ: rounding-level differences only
            if rel >= 2e-6:
                print("  fwd mismatch", list(kw), rel)
    # data gradient
    dy = torch.randn(B, S, S, Cout, device=dev)
    dys = ops.split_rows(dy)
    wts = ops.split_wT(w)
    refd = ops.conv3x3_dgrad(dy, w, (B, S, S, Cin), mode=1, w_split=ws)
    newd = ops.conv3x3_ps(dys, wts, B, S, S, Cout, Cin, -1) if Cin % 128 == 0 else None
    okd = newd is None or bool(torch.equal(refd, newd))
    if newd is not None and not okd:
        rel = float((refd - newd).abs().max() / refd.abs().max())
        okd = rel < 2e-6
        if not okd:
            print("  dgrad mismatch", rel)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    t_ref = timeit(lambda: ops.conv3x3_fwd(x, w, bias, mode=1, w_split=ws))
    t_new = timeit(lambda: ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, bias=bias))
    t_refd = timeit(lambda: ops.conv3x3_dgrad(dy, w, (B, S, S, Cin), mode=1, w_split=ws))
    t_newd = timeit(lambda: ops.conv3x3_ps(dys, wts, B, S, S, Cout, Cin, -1)) if newd is not None else float("nan")
    t_split = timeit(lambda: ops.split_rows(x, xs))
    print(f"B{B} {S}x{S} {Cin}->{Cout}: fwd equal={ok} old {t_ref:.1f}us ({fl/t_ref/1e6:.0f} TF) new {t_new:.1f}us ({fl/t_new/1e6:.0f} TF) | "
          f"dgrad equal={okd} old {t_refd:.1f}us ({fl/t_refd/1e6:.0f} TF) new {t_newd:.1f}us ({fl/t_newd/1e6:.0f} TF) | split_rows {t_split:.1f}us", flush=True)
