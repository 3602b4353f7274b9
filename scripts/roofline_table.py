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
    import re
    if cls.startswith("gn_"):
        pat = re.compile(cls + "_kernel")
        tot = n = 0.0
        for name, v in pmc.items():
            if pat.search(name):
                tot += (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0 * v["launches"]; n += v["launches"]
        return tot / n if n else None
    # bench.py's own mapping (class -> kernel-name patterns, second passes included), pointed at the given file
    bench.PMC_FILE = os.path.relpath(sys.argv[2], ROOT).replace("r06_", "{rnd}_").replace("bench_bf16x3", "bench_{wl}{mode}") if False else bench.PMC_FILE
    b, _ = bench._pmc_traffic(cls, "")
    return b


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
