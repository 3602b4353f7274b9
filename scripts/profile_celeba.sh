#!/bin/bash
# 256x256 workload artifacts (run on the GPU box through gpurun): rocprofv3 kernel tables of the DDPM-CELEBA-HQ-256 train step at B = 4
# (two-stream schedule and side stream off; rocprof_summary.py drops the in-process MFMA probe) + PMC traffic passes.
# usage: scripts/profile_celeba.sh <tag>   -> gpurun_out/<tag>_celeba256_*
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--workload celeba --steps 6 --warmup 2 --no-cpu-baseline --no-prof --sustain 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rc1 && rocprofv3 --kernel-trace --stats -d /tmp/rc1 -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/rc1.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rc1 -name "*.db" | head -1) 8 > $out/${tag}_celeba256_b4_kernel_stats.txt
rm -rf /tmp/rc2 && BD_AUX_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rc2 -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/rc2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rc2 -name "*.db" | head -1) 8 > $out/${tag}_celeba256_b4_single_stream_kernel_stats.txt
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 celeba > /dev/null 2>&1
cp $out/pmc_bench_celeba_bf16x3.json $out/${tag}_pmc_bench_celeba_bf16x3.json
head -30 $out/${tag}_celeba256_b4_single_stream_kernel_stats.txt | cut -c1-150
