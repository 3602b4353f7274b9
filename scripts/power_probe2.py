"""clock / power under the forward conv kernel, the wgrad kernel, GroupNorm and the whole train step (rocm-smi samples)"""
import json, subprocess, threading, time, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
dev = "cuda"
samples = []; stop = False
def poll():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = d[sorted(d)[0]]
            samples.append((time.time(), c.get("sclk clock speed:"), c.get("Current Socket Graphics Package Power (W)")))
        except Exception as e:
            pass
        time.sleep(0.1)
th = threading.Thread(target=poll); th.start()
def run(name, fn, secs=2.0, flops=0.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        n += 20
        torch.cuda.synchronize()
    t1 = time.time()
    ss = [(c, w) for (t, c, w) in samples if t0 + 0.7 < t < t1]
    us = (t1 - t0) / n * 1e6
    print(f"{name:34s} {us:8.1f} us/call {flops/us/1e6:6.0f} TF  sclk {[c for c, w in ss][-3:]}  W {[w for c, w in ss][-3:]}", flush=True)
B = 128
for (S, Cin, Cout) in [(32, 128, 128), (16, 256, 256), (16, 512, 256)]:
    x = torch.randn(B, S, S, Cin, device=dev); dy = torch.randn(B, S, S, Cout, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) / 30
    xs, dys, ws = ops.split_rows(x), ops.split_rows(dy), ops.split_bf16(w)
    y = torch.empty(B, S, S, Cout, device=dev)
    fl = 2.0 * B * S * S * Cin * Cout * 9
    run(f"fwd {S}x{S} {Cin}->{Cout}", lambda: ops.conv3x3_ps(xs, ws, B, S, S, Cin, Cout, 1, out=y), flops=fl)
    run(f"wgrad {S}x{S} {Cin}->{Cout}", lambda: ops.conv3x3_ps_wgrad(xs, dys, B, S, S, Cin, Cout, with_db=True), flops=fl)
a = torch.randn(1 << 26, device=dev); b2 = torch.empty_like(a)
run("copy 256MB", lambda: b2.copy_(a))
stop = True; th.join()
