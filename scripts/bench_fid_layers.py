"""Per-layer timing of the FID Inception convolutions (bd_conv2d_nhwc): both kernels (64 x 64 single-buffered, BD_FID_CONV=old; 128 x 64 double-buffered
with K-contiguous weights, the default) on every distinct layer shape of one forward at batch `B`, TFLOP/s against the 157.3 fp32-MFMA peak.
usage: python scripts/bench_fid_layers.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from baddiffusion_amd import _lib as L
from baddiffusion_amd.inception import FIDInceptionV3, state_dict_manifest

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
sd = {}
for k, shp in state_dict_manifest().items():
    if k.startswith("fc."):
        continue
    sd[k] = (torch.randn(shp, generator=g) * 0.05) if k.endswith("conv.weight") else (0.5 + torch.rand(shp, generator=g))
net = FIDInceptionV3(sd, device=dev, batch_size=B)
calls = []
orig = net._conv


def rec(x, name, k, stride=1, pad=(0, 0), out=None):
    calls.append((name, tuple(x.shape), tuple(k), stride, tuple(pad)))
    return orig(x, name, k, stride, pad, out)


net._conv = rec
net(torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8, device=dev))
torch.cuda.synchronize()
lib = L.load()
shapes = {}
for name, xs, k, st, pad in calls:
    Wf, _ = net._w[name]
    cout = Wf.shape[2]
    key = (xs[1], xs[2], xs[3], cout, k, st, pad)
    shapes.setdefault(key, []).append(name)


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = {"old": 0.0, "new": 0.0}
flt = 0.0
print(f"# batch {B}; us per launch (old 64x64 | new 128x64), TFLOP/s, count of layers with this shape")
for (H, W, cin, cout, k, st, pad), names in sorted(shapes.items(), key=lambda kv: -kv[0][0]):
    Ho, Wo = (H + 2 * pad[0] - k[0]) // st + 1, (W + 2 * pad[1] - k[1]) // st + 1
    x = torch.randn(B, H, W, cin, device=dev)
    w_old = torch.randn(k[0], k[1], cin, cout, device=dev) * 0.05
    w_new = w_old.permute(0, 1, 3, 2).contiguous()
    bias = torch.randn(cout, device=dev)
    y = torch.empty(B, Ho, Wo, cout, device=dev)
    fl = 2.0 * B * Ho * Wo * cout * k[0] * k[1] * cin
    res = {}
    for tag, w, kc in (("old", w_old, 0), ("new", w_new, 1)):
        d = L.Conv2dDesc(x=x.data_ptr(), ldx=cin, w=w.data_ptr(), bias=bias.data_ptr(), y=y.data_ptr(), ldy=cout, B=B, H=H, W=W, Cin=cin, Cout=cout,
                         KH=k[0], KW=k[1], stride_h=st, stride_w=st, pad_h=pad[0], pad_w=pad[1], relu=1, w_kc=kc)
        res[tag] = timeit(lambda: L.check(lib.bd_conv2d_nhwc(C.byref(d), L.stream()), "conv"))
        tot[tag] += res[tag] * len(names)
    flt += fl * len(names)
    print(f"{H:3d}x{W:<3d} {cin:4d}->{cout:<4d} k{k[0]}x{k[1]} s{st}: {res['old']:8.1f} | {res['new']:8.1f} us   {fl/res['old']/1e6:6.1f} | {fl/res['new']/1e6:6.1f} TF   x{len(names)}")
print(f"# sum over the 94 layers: old {tot['old']/1e3:.2f} ms ({flt/tot['old']/1e6:.1f} TF), new {tot['new']/1e3:.2f} ms ({flt/tot['new']/1e6:.1f} TF); best-of per layer would need a dispatch rule")
