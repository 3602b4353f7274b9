#!/bin/bash
# Round profile artifacts (run on the GPU box through gpurun): rocprofv3 kernel-trace summaries + bench JSON lines, then the
# PMC traffic passes.  usage: scripts/profile_round.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out
TRAIN="--no-cpu-baseline --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0"     # the headline train step alone (what the kernel tables describe)
cd /tmp && export TMPDIR=/tmp
# 1. the train step of the default command under rocprofv3 (two-stream schedule = the timed region of the bench line)
rm -rf /tmp/rp
rocprofv3 --kernel-trace --stats -d /tmp/rp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-prof $TRAIN > /tmp/rp.log 2>&1
grep "^{\"metric\"" /tmp/rp.log | tail -1 > $out/${tag}_bench_under_rocprof.json
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp -name "*.db" | head -1) 10 > $out/${tag}_train_step_b128_kernel_stats.txt
# 2. the driver's command, unprofiled: the whole default line (train + sustained + sampling + celeba + CPU baselines)
# (round 6: stdout carries the bounded headline, the full object goes to --detail-file)
cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --detail-file $out/${tag}_bench_detail.json 2>/dev/null | tail -1 > $out/${tag}_bench.json
cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --mode f32 $TRAIN --detail-file $out/${tag}_bench_f32.json 2>/dev/null | tail -1 > /dev/null
cd $GRAFT_REPO_ROOT && python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --detail-file $out/${tag}_bench_celeba256.json 2>/dev/null | tail -1 > /dev/null
cd $GRAFT_REPO_ROOT && python bench.py --workload pndm50 --batch 2048 --no-cpu-baseline --detail-file $out/${tag}_bench_pndm50.json 2>/dev/null | tail -1 > /dev/null
# 3. HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate --pmc passes)
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 > /dev/null 2>&1
cp $out/pmc_bench_bf16x3.json $out/${tag}_pmc_bench_bf16x3.json
# (bench.py reads roofline.traffic from profiles/<tag>_pmc_bench_*.json: re-run step 2's first line after copying this file there for a line
#  whose traffic is this build's)
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 ddim50 > /dev/null 2>&1
cp $out/pmc_bench_ddim50_bf16x3.json $out/${tag}_pmc_bench_ddim50_bf16x3.json
$GRAFT_REPO_ROOT/scripts/pmc_bench.sh bf16x3 ddpm1000 > /dev/null 2>&1
cp $out/pmc_bench_ddpm1000_bf16x3.json $out/${tag}_pmc_bench_ddpm1000_bf16x3.json
# 4. stand-alone kernel durations (side stream off) for the kernel table in DESIGN.md
cd /tmp && rm -rf /tmp/rp2 && BD_AUX_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-prof $TRAIN > /tmp/rp2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp2 -name "*.db" | head -1) 10 > $out/${tag}_train_step_b128_single_stream_kernel_stats.txt
# 5. the sampling loops of the default line under rocprofv3 (DDIM-50 x 512: same chunk size as the x 2048 loop)
cd /tmp && rm -rf /tmp/rp3 && rocprofv3 --kernel-trace --stats -d /tmp/rp3 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload ddim50 --batch 512 --no-cpu-baseline --no-prof > /tmp/rp3.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp3 -name "*.db" | head -1) 57 > $out/${tag}_ddim50_b512_kernel_stats.txt
# 6. the 256 x 256 workload: kernel tables (two-stream, single-stream) + PMC traffic
$GRAFT_REPO_ROOT/scripts/profile_celeba.sh $tag > /dev/null 2>&1
# 7. the measure path's feature extractor (SURVEY f-3): bench.py --workload fid under rocprofv3 (2048 CIFAR-size + 256 full-size images, + 1 warm-up chunk each)
cd /tmp && rm -rf /tmp/rp4 && rocprofv3 --kernel-trace --stats -d /tmp/rp4 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload fid > /tmp/rp4.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/rp4 -name "*.db" | head -1) 1 > $out/${tag}_fid_features_kernel_stats.txt
head -12 $out/${tag}_train_step_b128_kernel_stats.txt | cut -c1-140
cut -c1-400 $out/${tag}_bench.json
