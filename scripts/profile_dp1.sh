#!/bin/bash
# The data-parallel path at world = 1 (BD_FORCE_DP=1: RCCL on a 1-rank group) under rocprofv3: which RCCL kernels run, how long,
# and what they do to the backward kernels beside them.  usage: scripts/profile_dp1.sh <tag> [extra env]  -> gpurun_out/<tag>_dp1_*
tag=${1:-r04}; shift
out=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rd1 && env BD_FORCE_DP=1 "$@" rocprofv3 --kernel-trace --stats -d /tmp/rd1 -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/rd1.log 2>&1
grep "^{\"metric\"" /tmp/rd1.log | tail -1 | cut -c1-300 > $out/${tag}_dp1_bench.txt
tail -3 /tmp/rd1.log >> $out/${tag}_dp1_bench.txt
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rd1 -name "*.db" | head -1) 8 > $out/${tag}_dp1_kernel_stats.txt
head -40 $out/${tag}_dp1_kernel_stats.txt | cut -c1-160
