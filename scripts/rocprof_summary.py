"""Turn a rocprofv3 (rocpd sqlite) kernel trace into a text summary for profiles/.
usage: python scripts/rocprof_summary.py <results.db> <steps> > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
rows = [r for r in rows if "mfma_probe" not in r[0]]     # bench.py's MFMA issue-rate probe is not part of any step
tot = sum(r[2] for r in rows)
rows = [(n, c, t, a, 100.0 * t / tot) for n, c, t, a, _ in rows]
print(f"# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]}); durations in microseconds; {steps:g} steps traced")
print(f"# total kernel time {tot/1e3:.2f} ms = {tot/1e3/steps:.2f} ms/step")
print(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
for name, calls, total, avg, pct in rows:
    if len(name) > 110:
        name = name[:107] + "..."
    print(f"{calls:8d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name}")
