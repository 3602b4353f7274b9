#!/bin/bash
# HBM-side bytes per launch of the GroupNorm entry points by shape (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only): is the
# 64 / 96-byte row piece of a resident slab fetched once or by both workgroups that share its 128-byte line?
#   on the box: scripts/probes/gn_pmc.sh > gpurun_out/gn_pmc.txt
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/gp_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/gp_$c -o p -- python $GRAFT_REPO_ROOT/scripts/probes/gn_bw.py pmc > /tmp/gp_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
SH = [(128, 1024, 128), (128, 1024, 256), (128, 1024, 384), (128, 256, 384), (512, 1024, 128), (4, 65536, 128)]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/gp_{c}/**/*counter_collection.csv", recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c and "gn_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    res[c] = [(r["Kernel_Name"].split("(")[0][:40], float(r["Counter_Value"])) for r in rows]
# the probe runs, per shape, (3 warm + reps) forward calls then the same for backward: print every dispatch group compactly
import itertools
for c in res:
    print(c)
    for k, g in itertools.groupby(res[c], key=lambda t: t[0]):
        g = list(g); print(f"  {k:40s} x{len(g):3d}  KB/launch {sum(v for _, v in g) / len(g):12.1f}")
PY
