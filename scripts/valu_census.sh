#!/bin/bash
# Where do the VALU (and MFMA / LDS / SALU) instructions of a train step go?  One --pmc pass over the whole bench step, summed per kernel.
# usage: scripts/valu_census.sh [cifar|celeba]  -> gpurun_out/valu_census_<workload>.txt  (counts are wave-instructions per STEP)
wl=${1:-cifar}
case $wl in
  cifar)  WLARGS="--steps 3 --warmup 1 --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"; steps=4 ;;
  celeba) WLARGS="--workload celeba --steps 3 --warmup 1 --sustain 0"; steps=4 ;;
esac
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vc && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/vc -o p -- python $GRAFT_REPO_ROOT/bench.py $WLARGS --no-cpu-baseline --no-prof > /tmp/vc.log 2>&1
python3 - $steps $(find /tmp/vc -name "*counter_collection.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/valu_census_$wl.txt <<'PY'
import csv, sys, collections
steps = float(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
tot = collections.defaultdict(float)
for k, v in agg.items():
    for c, x in v.items(): tot[c] += x
print(f"# wave-instructions per step ({steps:g} steps traced); totals: " + ", ".join(f"{c} {x / steps / 1e6:.1f} M" for c, x in sorted(tot.items())))
print(f"{'launches':>8} {'VALU M':>9} {'%VALU':>6} {'MFMA M':>8} {'LDS M':>8} {'SALU M':>8} {'VALU/MFMA':>9}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    va, mf = v.get("SQ_INSTS_VALU", 0) / steps, v.get("SQ_INSTS_MFMA", 0) / steps
    print(f"{len(disp[k]) / steps:8.1f} {va / 1e6:9.2f} {100 * v.get('SQ_INSTS_VALU', 0) / max(tot['SQ_INSTS_VALU'], 1):6.1f} {mf / 1e6:8.2f} "
          f"{v.get('SQ_INSTS_LDS', 0) / steps / 1e6:8.2f} {v.get('SQ_INSTS_SALU', 0) / steps / 1e6:8.2f} {va / mf if mf else float('nan'):9.1f}  {k[:110]}")
PY
head -30 $GRAFT_REPO_ROOT/gpurun_out/valu_census_$wl.txt | cut -c1-190
