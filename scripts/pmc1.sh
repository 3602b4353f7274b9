#!/bin/bash
# one --pmc pass (the SQ cycle counters) of a command; usage: pmc1.sh <cmd...>  (prints per-kernel averages)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  rm -rf /tmp/pmc_1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_1 -o p -- "$@" > /tmp/pmc_1.log 2>&1
  f=$(find /tmp/pmc_1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -z "$f" ] && { tail -3 /tmp/pmc_1.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if "conv_ps" in k and "reduce" not in k and "split" not in k:
        n = len(disp[k])
        print(k, "dispatches", n, {c: round(x / n / 1e3, 1) for c, x in v.items()}, "(thousands)")
PY
done
