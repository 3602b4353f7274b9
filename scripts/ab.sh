#!/bin/bash
# A/B on ONE box: .ab/ holds an exported copy of a previous commit (git archive HEAD | tar -x -C .ab; built in place).
# Alternates the two trees so that box-to-box variance (~3 %) cancels.  usage: scripts/ab.sh [rounds] [bench args...]
rounds=${1:-2}; shift
for r in $(seq $rounds); do
  for tree in .ab .; do
    (cd $GRAFT_REPO_ROOT/$tree && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tree', round(d['ms_per_step'],3), round(d['value'],1), ' '.join(c['kernel'].replace('igemm_','')+'='+str(round(c['flops']/c['ms']/1e9)) for c in d.get('kernel_classes',[])[:6]))")
  done
done
