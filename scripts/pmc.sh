#!/bin/bash
# usage: pmc.sh <outname> <cmd...>   -> gpurun_out/pmc_<outname>.txt  (separate --pmc passes, kernel-trace only)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE TCC_HIT TCC_MISS TCC_EA0_RDREQ" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" 2>/dev/null | head -1); [ -z "$f" ] && { echo "no counter csv (see /tmp/pmc_$i.log)" >> $out; tail -3 /tmp/pmc_$i.log >> $out; continue; }
  echo "## pass $i: $set" >> $out
  python3 - "$f" >> $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.defaultdict(set)
for r in rows: disp[r["Kernel_Name"][:70]].add(r["Dispatch_Id"])
for k, v in agg.items():
    if "igemm" in k or "gn_" in k or "conv_ps" in k or "thin" in k or "attn" in k or "wino" in k:
        n = len(disp[k])
        print(k, "dispatches", n, {c: round(x / n, 1) for c, x in v.items()})
PY
done
cat $out
