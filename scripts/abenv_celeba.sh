#!/bin/bash
# scripts/abenv.sh for the 256 x 256 step (B = 4): alternates environment variants round-robin on ONE box.
# usage: scripts/abenv_celeba.sh <rounds> "<ENV=.. ENV=..>" "<ENV=..>" ...   ("" = defaults)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    ms=$(env $v python bench.py --workload celeba --steps 20 --warmup 5 --no-cpu-baseline --no-sampling --sustain 0 --no-prof 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r variant $i [$v] $ms ms/step"
    i=$((i+1))
  done
done
