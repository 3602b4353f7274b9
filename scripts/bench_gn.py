"""GroupNorm resident kernels, stand-alone timings at the CIFAR step's shapes"""
import ctypes as CT
import torch
from baddiffusion_amd import _lib as L, ops
lib = L.load()
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
import sys
SHAPES = [(128, 1024, 128), (128, 1024, 256), (128, 256, 256), (128, 256, 512), (128, 1024, 384), (128, 64, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "celeba":   # the split (non-resident) kernels: B = 4 at 256^2 .. 64^2
    SHAPES = [(4, 65536, 128), (4, 65536, 256), (4, 16384, 128), (4, 16384, 256), (4, 4096, 256), (4, 4096, 512), (4, 1024, 512)]
for (B, HW, Cc) in SHAPES:
    x = torch.randn(B, HW, Cc, device="cuda"); dy = torch.randn(B, HW, Cc, device="cuda")
    ga = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda")
    ys = torch.empty(B * HW, Cc // 32, 2, 32, dtype=torch.int16, device="cuda")
    st = torch.empty(2, B, 32, device="cuda")
    ws = ops.workspace(lib.bd_gn_workspace_bytes(B, Cc), x.device)
    d = L.GnFwdDesc(B=B, HW=HW, C=Cc, G=32, eps=1e-6, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), y=None, ldy=Cc,
                    mean=L.ptr(st[0]), rstd=L.ptr(st[1]), workspace=L.ptr(ws), workspace_bytes=ws.numel(), y_split=L.ptr(ys), ldys=Cc)
    tf = timeit(lambda: L.check(lib.bd_gn_fwd(CT.byref(d), L.stream())))
    dx = torch.empty_like(x); dg = torch.empty(Cc, device="cuda"); db = torch.empty(Cc, device="cuda")
    e = L.GnBwdDesc(B=B, HW=HW, C=Cc, G=32, silu=1, x=L.ptr(x), ldx=Cc, gamma=L.ptr(ga), beta=L.ptr(be), mean=L.ptr(st[0]), rstd=L.ptr(st[1]),
                    dy=L.ptr(dy), lddy=Cc, dx=L.ptr(dx), lddx=Cc, accumulate_dx=0, dgamma=L.ptr(dg), dbeta=L.ptr(db), workspace=L.ptr(ws),
                    workspace_bytes=ws.numel())
    tb = timeit(lambda: L.check(lib.bd_gn_bwd(CT.byref(e), L.stream())))
    e.accumulate_dx = 1
    tba = timeit(lambda: L.check(lib.bd_gn_bwd(CT.byref(e), L.stream())))
    e.accumulate_dx = 0; e.dx = None; e.dx_split = L.ptr(ys); e.lddxs = Cc
    tbs = timeit(lambda: L.check(lib.bd_gn_bwd(CT.byref(e), L.stream())))
    n = B * HW * Cc * 4
    print(f"B{B} HW{HW} C{Cc}: fwd(split out) {tf:.1f} us ({2*n/tf/1e6:.2f} TB/s)  bwd {tb:.1f} us ({3*n/tb/1e6:.2f} TB/s)  bwd+acc {tba:.1f} us ({4*n/tba/1e6:.2f} TB/s)  bwd(split out) {tbs:.1f} us ({3*n/tbs/1e6:.2f} TB/s)", flush=True)
