"""One row per profiled kernel class of the CIFAR train step: launches per step, time per launch in the timed two-stream schedule and with the side stream
off, algorithmic (GroupNorm: required) bytes and flops per launch, the rates they imply, and the HBM-side bytes rocprofv3's PMC passes measured for the
class's kernels.  Inputs: a bench detail object (bench.py --detail-file) and a PMC file (scripts/pmc_bench.sh).
usage: python scripts/roofline_table.py profiles/r06_bench_detail.json profiles/r06_pmc_bench_bf16x3.json > profiles/r06_roofline_table.txt"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
_argv, sys.argv = sys.argv, [sys.argv[0]]
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
sys.argv = _argv
d = json.load(open(sys.argv[1]))
pmc = json.load(open(sys.argv[2]))["kernels"]
steps = d["roofline"]["sampled_steps"]
iso = {c["kernel"]: c for c in d["kernel_classes_standalone"]}


def pmc_of(cls):
    """HBM-side bytes per call of class `cls` from the GIVEN PMC file: the class's own kernels (launch-weighted) + the mean of each second-pass
    kernel -- bench.py's class -> kernel-name patterns (_PMC_PS, _PMC_OPERANDS), applied to this file"""
    import re
    by = lambda v: (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0

    def mean(pat):
        rx = re.compile(pat)
        tot = n = 0.0
        for name, v in pmc.items():
            if rx.search(name):
                tot += by(v) * v["launches"]; n += v["launches"]
        return tot / n if n else None

    if cls.startswith("gn_"):
        return mean(cls + "_kernel")
    if cls in bench._PMC_PS:
        pats = bench._PMC_PS[cls]
        main = mean(pats[0])
        return None if main is None else main + sum(mean(p) or 0.0 for p in pats[1:])
    m = re.match(r"igemm_(\w+?)_(\d+)(_bf16x3)?$", cls)
    if not m or m.group(1) not in bench._PMC_OPERANDS:
        return None
    la, lbs = bench._PMC_OPERANDS[m.group(1)]
    kname = "igemm_bf16x3_kernel" if m.group(3) else "igemm_kernel"
    rx = re.compile(rf"{kname}<{m.group(2)}, {m.group(2)}, bd::(\w+)<[^>]*>, bd::(\w+)<[^>]*>")
    tot = n = 0.0
    for name, v in pmc.items():
        mm = rx.search(name)
        if mm and mm.group(1) == la and mm.group(2) in lbs:
            tot += by(v) * v["launches"]; n += v["launches"]
    return tot / n if n else None


print(f"# {sys.argv[1]} + {sys.argv[2]}: CIFAR-32 UNet train step, B = 128, {d['ms_per_step']:.2f} ms/step, MFMA probe "
      f"{d['roofline'].get('mfma_power_limited_peak_measured', 0):.0f} TFLOP/s; peaks: HBM 8 TB/s, split-bf16 833 TFLOP/s (algorithmic)")
print(f"{'class':34s} {'n/step':>6s} {'us sched':>9s} {'us alone':>9s} {'alg MB':>8s} {'PMC MB':>8s} {'PMC/alg':>7s} {'TB/s alone':>10s} {'TF/s sched':>10s} {'TF/s alone':>10s} {'frac alone':>10s}")
for c in d["kernel_classes"]:
    n = c["launches"] / steps
    us = c["ms"] * 1e3 / c["launches"]
    i = iso.get(c["kernel"])
    us_i = i["ms"] * 1e3 / i["launches"] if i else float("nan")
    mb = c["bytes"] / c["launches"] / 1e6
    p = pmc_of(c["kernel"])
    tf = c["flops"] / c["ms"] / 1e9 if c["flops"] else 0.0
    tf_i = (i["flops"] / i["ms"] / 1e9) if (i and i["flops"]) else 0.0
    tbs = mb / us_i if us_i == us_i else float("nan")
    frac = max(tf_i / 833.3, tbs / 8.0) if us_i == us_i else float("nan")
    print(f"{c['kernel'][:34]:34s} {n:6.1f} {us:9.1f} {us_i:9.1f} {mb:8.1f} {(p / 1e6 if p else float('nan')):8.1f} {(p / 1e6 / mb if p and mb else float('nan')):7.2f} "
          f"{tbs:10.2f} {tf:10.1f} {tf_i:10.1f} {frac:10.2f}")
