#!/bin/bash
# ablation of conv_wino_kernel (BD_WINO_ABL bits: 1 no U DMA, 2 no MFMA, 4 no transform, 8 no raw DMA, 16 no epilogue stores; compiled: 0 2 4 9 13 31);
# timing only (results are wrong by construction)
for a in ${ABLS:-0 2 4 9 13 31}; do
  echo "ABL=$a"; BD_WINO_ABL=$a timeout 200 python scripts/wino/check.py 30 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        if 'wino_us' in d: print('  B%d %dx%d %d->%d: wino %.1f us  direct %.1f us' % (d['B'], d['S'], d['S'], d['Cin'], d['Cout'], d['wino_us'], d['direct_ps_us']))
"
done
