"""Host and stream cost of one ncclAllReduce on a 1-rank communicator (in place, fp32 sum), idle GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baddiffusion_amd.rccl import RcclComm  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
comm = RcclComm(dev)
cs = torch.cuda.Stream(device=dev)
buf = torch.zeros(8 << 20, device=dev)
for n in (256, 1 << 16, 1 << 20, 7 << 20):
    t = buf[:n]
    for _ in range(5):
        comm.all_reduce_(t, cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(cs)
    for _ in range(50):
        comm.all_reduce_(t, cs)
    e1.record(cs)
    host = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    print(f"{4 * n:>10d} B: host {host:8.1f} us/call, stream {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us/call", flush=True)
# the same while a long kernel runs on another stream
x = torch.randn(8192, 8192, device=dev)
main = torch.cuda.current_stream()
t = buf[: 1 << 20]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y = x @ x
for _ in range(30):
    comm.all_reduce_(t, cs)
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"20 matmuls + 30 all-reduces: host enqueue {h * 1e3:.1f} ms, total {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
t0 = time.perf_counter()
for _ in range(20):
    y = x @ x
torch.cuda.synchronize()
print(f"20 matmuls alone: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
# hypothesis: the 1-rank path touches the legacy NULL stream (implicit sync with it); put the compute on a non-blocking stream
ms = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(ms):
    for _ in range(20):
        y = x @ x
for _ in range(30):
    comm.all_reduce_(t, cs)
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"matmuls on a non-default stream + 30 all-reduces: host enqueue {h * 1e3:.1f} ms, total {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
# and interleaved, as the trainer issues them
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(ms):
    for _ in range(20):
        y = x @ x
        cs.wait_stream(ms)
        comm.all_reduce_(t, cs)
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"interleaved, compute on a non-default stream: host enqueue {h * 1e3:.1f} ms, total {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y = x @ x
    cs.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(t, cs)
h = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"interleaved, compute on the default stream: host enqueue {h * 1e3:.1f} ms, total {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
