"""Per-queue view of one steady-state train step from a rocprofv3 --kernel-trace CSV: when each HIP stream (queue) is busy, in 0.5-ms bins, and
what runs in the last milliseconds before the optimizer kernel -- is the side stream (weight gradients) or the main stream (data gradients)
the one the step waits for?   usage: python scripts/timeline_queues.py <kernel_trace.csv> [step_index_from_end]"""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
ends = [e for (s, e, n, q) in ev if "adam_clip" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0, t1 = ends[-k - 1], ends[-k]
win = [(s, e, n, q) for (s, e, n, q) in ev if e > t0 and s < t1]
qs = sorted(set(q for *_, q in win))
print(f"step {(t1 - t0) / 1e6:.2f} ms, {len(win)} kernels, queues {qs}")
BIN = 500_000
nb = (t1 - t0 + BIN - 1) // BIN
busy = {q: [0] * nb for q in qs}
for s, e, n, q in win:
    s, e = max(s, t0), min(e, t1)
    b = (s - t0) // BIN
    while s < e:
        be = t0 + (b + 1) * BIN
        busy[q][b] += min(e, be) - s
        s = min(e, be); b += 1
print("busy fraction per 0.5 ms bin (one column per bin):")
for q in qs:
    print(f"  queue {q:>3s}: " + " ".join(f"{int(100 * v / BIN):3d}" for v in busy[q]) + f"   total {sum(busy[q]) / 1e6:.2f} ms")
# the last kernels of each queue before the optimizer
adam_start = [s for (s, e, n, q) in win if "adam_clip" in n][-1]
print("last kernels before adam_clip (start offset from the step's end in us, duration us, queue, name):")
tail = [(s, e, n, q) for (s, e, n, q) in win if e <= adam_start + 1 and e > adam_start - 1_500_000]
for s, e, n, q in sorted(tail)[-28:]:
    print(f"  {-(t1 - s) / 1e3:8.0f} {(e - s) / 1e3:7.1f}  q{q}  {n.split('(')[0].replace('void bd::', '').replace('bd::', '')[:60]}")
lastq = collections.defaultdict(int)
for s, e, n, q in win:
    if e <= adam_start + 1: lastq[q] = max(lastq[q], e)
print("each queue's last kernel ends", {q: f"{(adam_start - v) / 1e3:.0f} us before adam_clip starts" for q, v in lastq.items()})
