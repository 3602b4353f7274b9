"""bench.py JSON line on stdin -> one row per kernel class (with BD_PROF_SHAPES=1: one per GEMM shape).
usage: BD_PROF_SHAPES=1 python bench.py ... | python scripts/classes.py [kernel_classes|kernel_classes_standalone]"""
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'])
tot=0
for c in sorted(d[sys.argv[1] if len(sys.argv) > 1 else 'kernel_classes'], key=lambda c:-c['ms']):
    tot+=c['ms']
    print(f"{c['kernel']:70s} n={c['launches']:4d} ms={c['ms']:8.3f} us/call={1e3*c['ms']/c['launches']:7.1f} TF={c['flops']/c['ms']/1e9:6.1f}")
