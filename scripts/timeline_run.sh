#!/bin/bash
# kernel trace (CSV) of the CIFAR train step + the per-queue views; usage: scripts/timeline_run.sh <tag>
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-prof --no-cpu-baseline --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/timeline_queues.py $f > $out/${tag}_timeline_queues.txt
python $GRAFT_REPO_ROOT/scripts/timeline_dump.py $f 3 > $out/${tag}_timeline_dump.txt
head -6 $out/${tag}_timeline_queues.txt | cut -c1-260
