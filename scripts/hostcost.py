"""Host enqueue time vs wall time of one UNet forward (is the sampling loop launch-bound?).  usage: python scripts/hostcost.py"""
import time, torch
from baddiffusion_amd import model as M, unet
from baddiffusion_amd.schedulers import DDPMScheduler
m = unet.UNet2DModel(**M.KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"]).cuda()
s = DDPMScheduler()
for B in (16, 256):
    x = torch.randn(B, 3, 32, 32, device="cuda")
    t = torch.tensor([10], device="cuda")
    with torch.no_grad():
        for _ in range(3): m(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): m(x, t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3*(t1-t0)/20:.2f} ms per forward, wall {1e3*(t2-t0)/20:.2f} ms per forward")
