"""Every kernel of one steady-state train step in start order, per queue, with the idle gap in front of it on its queue.
usage: python scripts/timeline_dump.py <kernel_trace.csv> [step_index_from_end] [from_ms] [to_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
ends = [e for (s, e, n, q) in ev if "adam_clip" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
t0, t1 = ends[-k - 1], ends[-k]
win = [(s, e, n, q) for (s, e, n, q) in ev if e > t0 and s < t1]
last = {}
print(f"step {(t1 - t0) / 1e6:.3f} ms; columns: start_us dur_us gap_us queue kernel")
for s, e, n, q in win:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    off = (s - t0) / 1e6
    if lo <= off <= hi:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  q{q}  {n.split('(')[0].replace('void bd::', '').replace('bd::', '')[:70]}")
