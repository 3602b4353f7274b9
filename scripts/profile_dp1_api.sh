#!/bin/bash
# HIP API + kernel + memory-copy trace of the forced data-parallel step at world 1 (what does RCCL's 1-rank all-reduce call?)
out=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rd2 && BD_FORCE_DP=1 BD_DP_BUCKET_MB=100000 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --stats --output-format csv -d /tmp/rd2 -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/rd2.log 2>&1
ls /tmp/rd2 /tmp/rd2/* | head -20
f=$(find /tmp/rd2 -name "*hip_api_stats.csv" | head -1); echo "== $f"; head -25 "$f"
f=$(find /tmp/rd2 -name "*memory_copy_stats.csv" | head -1); echo "== $f"; head -10 "$f"
f=$(find /tmp/rd2 -name "*hip_api_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
# longest individual API calls
rows.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
for r in rows[:25]:
    print(r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
