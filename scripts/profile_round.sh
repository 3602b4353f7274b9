#!/bin/bash
# Round profile artifacts (run on the GPU box through gpurun): rocprofv3 kernel-trace summary + bench JSON of the
# same command, then the PMC traffic passes.  usage: scripts/profile_round.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp
rocprofv3 --kernel-trace --stats -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /tmp/rp.log 2>&1
grep "^{\"metric\"" /tmp/rp.log | tail -1 > $out/${tag}_bench_under_rocprof.json
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp -name "*.db" | head -1) 12 > $out/${tag}_train_step_b128_kernel_stats.txt
cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/${tag}_bench.json
cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --mode f32 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_f32.json
cd $GRAFT_REPO_ROOT && python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_celeba256.json
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 > /dev/null 2>&1
head -12 $out/${tag}_train_step_b128_kernel_stats.txt | cut -c1-140
cut -c1-400 $out/${tag}_bench.json
cd $GRAFT_REPO_ROOT && python bench.py --workload ddim50 --batch 2048 2>/dev/null | tail -1 > $out/${tag}_bench_ddim50.json
cd $GRAFT_REPO_ROOT && python bench.py --workload ddpm1000 --batch 256 2>/dev/null | tail -1 > $out/${tag}_bench_ddpm1000.json
cd $GRAFT_REPO_ROOT && python bench.py --workload pndm50 --batch 2048 2>/dev/null | tail -1 > $out/${tag}_bench_pndm50.json
# stand-alone kernel durations (side stream off) for the kernel table in DESIGN.md
cd /tmp && rm -rf /tmp/rp2 && BD_AUX_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-prof > /tmp/rp2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp2 -name "*.db" | head -1) 10 > $out/${tag}_train_step_b128_single_stream_kernel_stats.txt
