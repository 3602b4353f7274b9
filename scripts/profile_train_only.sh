#!/bin/bash
# the train-step part of profile_round.sh alone (kernel tables in both schedules + the PMC traffic pass of the CIFAR step)
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out
TRAIN="--no-cpu-baseline --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp
rocprofv3 --kernel-trace --stats -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-prof $TRAIN > /tmp/rp.log 2>&1
grep "^{\"metric\"" /tmp/rp.log | tail -1 > $out/${tag}_bench_under_rocprof.json
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp -name "*.db" | head -1) 10 > $out/${tag}_train_step_b128_kernel_stats.txt
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 > /dev/null 2>&1
cp $out/pmc_bench_bf16x3.json $out/${tag}_pmc_bench_bf16x3.json
cd /tmp && rm -rf /tmp/rp2 && BD_AUX_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-prof $TRAIN > /tmp/rp2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp2 -name "*.db" | head -1) 10 > $out/${tag}_train_step_b128_single_stream_kernel_stats.txt
head -8 $out/${tag}_train_step_b128_kernel_stats.txt | cut -c1-140
