"""Does the chip clock down under the conv kernels?  Loops one kernel for ~2.5 s per case while a thread samples
rocm-smi (sclk, socket power).  usage: python scripts/power_probe.py"""
import json, os, subprocess, sys, threading, time
import torch
from baddiffusion_amd import ops
dev = "cuda"
samples = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(o)
            c = d[sorted(d)[0]]
            samples.append((time.time(), {k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()}))
        except Exception as e:
            samples.append((time.time(), {"err": str(e)[:80]}))
        time.sleep(0.15)
th = threading.Thread(target=poll); th.start()
def run(name, fn, secs=2.5):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    t1 = time.time()
    ss = [s for (t, s) in samples if t0 + 0.8 < t < t1]
    print(name, f"{a.elapsed_time(b) / n * 1e3:.1f} us/call", ss[-1] if ss else None, flush=True)
time.sleep(1.0)
print("idle", samples[-1][1] if samples else None, flush=True)
B, S, Cin, Cout = 128, 16, 512, 256
x = torch.randn(B, S, S, Cin, device=dev); dy = torch.randn(B, S, S, Cout, device=dev)
xs = ops.split_rows(x); dys = ops.split_rows(dy)
run("wgrad_ps 16x16 512->256", lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True))
a = torch.randn(1 << 26, device=dev); b2 = torch.empty_like(a)
run("copy 256MB", lambda: b2.copy_(a))
run("wgrad_ps again", lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True))
stop = True; th.join()
