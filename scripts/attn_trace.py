"""Per-kind average kernel time from a rocprofv3 kernel_trace.csv of scripts/bench_attn.py (kinds are separated by silu marker launches).
usage: python scripts/attn_trace.py <kernel_trace.csv> <reps> "<KINDS line>" """
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]); kinds = sys.argv[3].replace("KINDS ", "").split("|")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "silu_fwd" in r["Kernel_Name"]]
marks = marks[-(len(kinds) + 1):]
tot = 0.0
for kn, a, b in zip(kinds, marks[:-1], marks[1:]):
    seg = rows[a + 1:b]
    us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3 / reps
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / reps
    names = sorted({r["Kernel_Name"].split("(")[0][-60:] for r in seg})
    tot += us
    print(f"{kn}  {us:8.1f} us busy  {span:8.1f} us span  {len(seg)//reps} launches  {names}")
print(f"sum {tot:.1f} us")
