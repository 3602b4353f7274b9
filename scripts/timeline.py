"""Overlap analysis of one steady-state train step from a rocprofv3 --kernel-trace CSV: how much of the step has an
MFMA-class kernel in flight, how much only memory-bound kernels, how much nothing.
usage: python scripts/timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
# steps are delimited by the adam_clip kernel
ends = [e for (s, e, n, q) in ev if "adam_clip" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0, t1 = ends[-k - 1], ends[-k]
win = [(max(s, t0), min(e, t1), n, q) for (s, e, n, q) in ev if e > t0 and s < t1]
mf = lambda n: any(x in n for x in ("igemm", "conv_ps_kernel", "conv_ps3_kernel", "conv_ps128_kernel", "conv_ps_wgrad_kernel", "conv_ps_wgrad3_kernel",
                                    "conv_ph_kernel", "gemm_sp_kernel", "attn_sp_", "wgrad_thin"))
pts = []
for s, e, n, q in win:
    c = 0 if mf(n) else 1
    pts.append((s, +1, c)); pts.append((e, -1, c))
pts.sort()
cnt = [0, 0]
last = t0
acc = {"mfma": 0, "mfma+mem": 0, "mem_only": 0, "idle": 0}
for t, d, c in pts:
    dt = t - last
    if cnt[0] and cnt[1]: acc["mfma+mem"] += dt
    elif cnt[0]: acc["mfma"] += dt
    elif cnt[1]: acc["mem_only"] += dt
    else: acc["idle"] += dt
    cnt[c] += d
    last = t
acc["idle"] += t1 - last
tot = t1 - t0
print(f"step window {tot/1e6:.2f} ms, {len(win)} kernels, queues {sorted(set(q for *_, q in win))}")
for k_, v in acc.items():
    print(f"  {k_:9s} {v/1e6:7.2f} ms  {100.0*v/tot:5.1f} %")
busy = sum(e - s for s, e, n, q in win)
print(f"  sum of kernel durations {busy/1e6:.2f} ms (overlap factor {busy/tot:.2f})")
# the biggest idle / mem-only stretches
segs = []
cnt = [0, 0]; last = t0
for t, d, c in pts:
    if not cnt[0] and t > last:
        segs.append((t - last, last - t0, "mem_only" if cnt[1] else "idle"))
    cnt[c] += d; last = t
segs.sort(reverse=True)
print("  longest stretches without an MFMA kernel:", [(round(a / 1e3), round(b / 1e6, 2), c) for a, b, c in segs[:12]], "(us, at ms, kind)")
# merge adjacent non-MFMA stretches (gaps < 3 us) and name the kernels inside the longest ones
segs2 = []
cnt = [0, 0]; last = t0; cur = None
for t, d, c in pts:
    if not cnt[0] and t > last:
        if cur and last - cur[1] < 3000: cur[1] = t
        else:
            if cur: segs2.append(tuple(cur))
            cur = [last, t]
    cnt[c] += d; last = t
if cur: segs2.append(tuple(cur))
segs2.sort(key=lambda ab: ab[0] - ab[1])
print("  merged stretches without an MFMA kernel in flight (total %.2f ms):" % (sum(b - a for a, b in segs2) / 1e6))
for a, b in segs2[:10]:
    names = [n.split("(")[0].replace("void bd::", "").replace("bd::", "")[:28] for s_, e_, n, q in win if s_ < b and e_ > a and not mf(n)]
    print(f"    {(b - a) / 1e3:6.0f} us at {(a - t0) / 1e6:6.2f} ms: {names[:8]}")
