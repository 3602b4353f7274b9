#!/bin/bash
# A/B of environment knobs on ONE box: alternates the variants round-robin (box-to-box variance is 2-5 %, in-box repeatability ~0.5 %).
# usage: scripts/abenv.sh <rounds> "<ENV=.. ENV=..>" "<ENV=..>" ...   (an empty string "" = defaults)   -> one line per run
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    ms=$(env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sampling --no-celeba --no-fid --no-dp-probe --sustain 0 --no-prof 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r variant $i [$v] $ms ms/step"
    i=$((i+1))
  done
done
