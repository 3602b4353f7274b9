#!/bin/bash
# HBM traffic of every kernel of the bench step from PMC counters, one counter per pass (guide: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; kernel-trace only).  usage: pmc_bench.sh <mode> [cifar|celeba|ddim50|ddpm1000] -> gpurun_out/pmc_bench_[<workload>_]<mode>.json
mode=${1:-bf16x3}
wl=${2:-cifar}
case $wl in
  cifar)  WLARGS="--steps 3 --warmup 1 --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"; tag="" ;;
  celeba) WLARGS="--workload celeba --steps 3 --warmup 1 --sustain 0"; tag="celeba_" ;;
  ddim50) WLARGS="--workload ddim50 --batch 2048"; tag="ddim50_" ;;       # the chunk shape of the default line's DDIM-50 x 2048 loop
  ddpm1000) WLARGS="--workload ddim50 --batch 256"; tag="ddpm1000_" ;;    # same kernels and batch as the DDPM-1000 x 256 loop, 50 evaluations
esac
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pb_$c -o p -- python $GRAFT_REPO_ROOT/bench.py $WLARGS --no-cpu-baseline --no-prof --mode $mode > /tmp/pb_$c.log 2>&1
done
python3 - $tag$mode <<'PY'
import csv, sys, glob, json, collections, os
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pb_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        agg[r["Kernel_Name"]] += float(r["Counter_Value"]); disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    for k, v in agg.items():
        out[k][c + "_KB_per_launch"] = v / len(disp[k]); out[k]["launches"] = len(disp[k])
root = os.environ["GRAFT_REPO_ROOT"]
json.dump({"note": "rocprofv3 --pmc, KB per launch (raw counter values: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, "
                   "MI355X_MICROARCH.md; bench.py applies the x2)", "kernels": out}, open(f"{root}/gpurun_out/pmc_bench_{sys.argv[1]}.json", "w"), indent=1)
print({k[:60]: v for k, v in list(out.items())[:6]})
PY
