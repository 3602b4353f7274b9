"""experiment: fwd+bwd of B=128 in one pipeline vs two concurrent half-batch pipelines on two streams"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baddiffusion_amd.model import KNOWN_TOPOLOGIES
from baddiffusion_amd.unet import UNet2DModel
from baddiffusion_amd import _lib as L
dev = torch.device("cuda")
m = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"]).to(dev)
lib = L.load()
B = 128
x = torch.randn(B, 32, 32, 3, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
out = torch.empty(B, 32, 32, 3, device=dev); dout = torch.randn(B, 32, 32, 3, device=dev)
def ws_for(b): return torch.empty(m.workspace_bytes(b, True), dtype=torch.uint8, device=dev)
w_full = ws_for(B); w_a = ws_for(B // 2); w_b = ws_for(B // 2)
g_full = torch.zeros(m.num_flat, device=dev); g_a = torch.zeros_like(g_full); g_b = torch.zeros_like(g_full)
s2 = torch.cuda.Stream()
def fwd(b, xs, ts, os_, ws, stream):
    L.check(lib.bd_unet_forward(m._plan, b, 1, m.flat.data_ptr(), xs.data_ptr(), 3, ts.data_ptr(), 1, os_.data_ptr(), 3,
                                ws.data_ptr(), ws.numel(), stream), "fwd")
def bwd(b, xs, ds, g, ws, stream):
    L.check(lib.bd_unet_backward(m._plan, b, m.flat.data_ptr(), xs.data_ptr(), 3, ds.data_ptr(), 3, g.data_ptr(), ws.data_ptr(),
                                 ws.numel(), stream), "bwd")
def one():
    s = torch.cuda.current_stream().cuda_stream
    fwd(B, x, t, out, w_full, s); bwd(B, x, dout, g_full, w_full, s)
def two():
    h = B // 2
    s1 = torch.cuda.current_stream().cuda_stream
    ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev)
    fwd(h, x[:h], t[:h], out[:h], w_a, s1)
    fwd(h, x[h:], t[h:], out[h:], w_b, s2.cuda_stream)
    bwd(h, x[:h], dout[:h], g_a, w_a, s1)
    bwd(h, x[h:], dout[h:], g_b, w_b, s2.cuda_stream)
    ev2 = torch.cuda.Event(); ev2.record(s2); torch.cuda.current_stream().wait_event(ev2)
    g_a.add_(g_b)
for name, fn in (("one B=128", one), ("two B=64", two), ("one B=128", one), ("two B=64", two)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms")
one(); two(); torch.cuda.synchronize()
print("grad rel diff", float((g_a - g_full).norm() / g_full.norm()))
