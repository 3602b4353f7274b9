"""bench.py -- headline benchmark of the BadDiffusion hot path on MI355X.

metric (BASELINE.json): train images/sec of the 32x32 UNet (DDPM-CIFAR10-32 topology), batch 128 per GPU,
poison_rate 0.1, BOX_14 trigger, fp32, synthetic data.  One "step" = poison-blend + q_sample -> UNet forward
-> MSE -> UNet backward (+ RCCL all-reduce of the 143 MB flat gradient when N > 1) -> clip + Adam.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Inputs (uint8 images, noise, timesteps) are resident in HBM before the timed
region, which carries no instrumentation.  `roofline` is measured live with hipEvent pairs around every GEMM-class
launch (bd_prof_*) of extra untimed steps behind it, `cpu_baseline` is the CPU oracle (pure PyTorch fp32 restatement
of the reference path) timed on this box's host cores.  The line also carries `sampling` (DDIM-50 x 2048, DDPM-1000 x
256), `celeba` (the 256x256 DDPM-CELEBA-HQ-256 step at B = 4) -- each with its own `roofline` and `cpu_baseline` --
`sustained`, and `dp_path_at_world_1` (trainer.py's data-parallel branch under RCCL on a 1-rank group).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / CUDA-tensor sharing across ranks fail with hipIpcGetMemHandle: invalid
# argument otherwise); the driver's environment exports it, a bare shell may not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# kernel arguments in device memory: the runtime's default on this stack; with it OFF the ~770 launches of a step cost +0.7 ms (profiles/r06_ab_runtime_env.txt)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_IMG = 37.324          # SURVEY 8d (fwd+bwd, 2*MAC of conv/GEMM/BMM), CIFAR-32 UNet
FP32_MFMA_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA (split-bf16 mode issues 3 MFMA flops per algorithmic flop)
HBM_PEAK_GBS = 8000.0
_JSON_OUT = sys.stdout


_PMC_OPERANDS = {"conv_fwd": ("ConvKC", ("WgtKC", "WgtKCs")), "conv_dgrad": ("TConvKC", ("WgtRC", "WgtRCs")),
                 "conv_wgrad": ("DenseRC", ("ConvRC", "ConvRCs")), "gemm_nt": ("DenseKC", ("DenseKC",)),
                 "gemm_nn": ("DenseKC", ("DenseRC",)), "gemm_tn": ("DenseRC", ("DenseRC",))}
# kernel classes of the LDS-DMA family (conv_ps.hip) -> kernel-name patterns whose HBM traffic makes up one call
_PMC_PS = {"conv_ps_wgrad3": (r"conv_ps_wgrad3_kernel", r"conv_ps_wgrad_reduce(<true>|\()"),      # (<true>: round 6 names the large layers' second pass)
           "conv_ps_wgrad": (r"conv_ps_wgrad_kernel<2, 8, false>", r"conv_ps_wgrad_reduce(<false>|\()"), "conv_ps_fwd": (r"conv_ps3?_kernel<[12]>",),
           "conv_ps_dgrad": (r"conv_ps3?_kernel<0>",), "conv_ps128_fwd": (r"conv_ps128_kernel<", r"conv_ps128_reduce"),
           "conv_ps128_dgrad": (r"conv_ps128_kernel<", r"conv_ps128_reduce"),
           "conv_ph_ups_fwd": (r"conv_ph_kernel",), "conv_ph_ups_dgrad": (r"conv_ph_kernel",), "conv_ph_s2_dgrad": (r"conv_ph_kernel",),
           "conv_ph_ups_wgrad": (r"conv_ps_wgrad_kernel<2, 8, true>", r"conv_ps_wgrad_reduce"),
           "attn_sp_fwd": (r"attn_sp_fwd_kernel",), "attn_sp_bwd_a": (r"attn_sp_bwd_a_kernel",), "attn_sp_bwd_b": (r"attn_sp_bwd_b_kernel",),
           "gemm_sp_nt": (r"gemm_sp_kernel<false, false>",), "gemm_sp_nn": (r"gemm_sp_kernel<false, true>",),
           "gemm_sp_tn": (r"gemm_sp_kernel<true, true>", r"gemm_sp_reduce")}
PMC_ROUNDS = ("r06", "r05", "r04", "r03", "r02")                     # this round's file first; an older one is used only when it is absent, and flagged
PMC_FILE = "profiles/{rnd}_pmc_bench_{wl}{mode}.json"   # wl = "" (CIFAR train step), "celeba_" (256x256 train step), "ddim50_" / "ddpm1000_" (sampling)


def _pmc_traffic(cls, workload=""):
    """HBM-side bytes per call of kernel class `cls` from the committed rocprofv3 PMC passes of this same command
    (scripts/pmc_bench.sh -> profiles/r02_pmc_bench_<mode>.json; FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950
    correction for 16 B/lane reads, + WRITE_SIZE; KB -> bytes; launch-weighted over the kernel instantiations of the
    class, split-K second passes included).  Returns (bytes, source) -- (None, reason) when that file or kernel is
    absent: counters cannot be collected from inside the timed process, so this is read from a committed file."""
    import re
    mode = "bf16x3" if ("bf16x3" in cls or cls.startswith("conv_ps") or cls.startswith("conv_ph") or "_sp_" in cls) else "f32"
    wl = workload + "_" if workload else ""
    rel = path = None
    for rnd in PMC_ROUNDS:
        rel = PMC_FILE.format(rnd=rnd, wl=wl, mode=mode)
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            break
    else:
        return None, f"{PMC_FILE.format(rnd=PMC_ROUNDS[0], wl=wl, mode=mode)} not found"
    if not rel.startswith(f"profiles/{PMC_ROUNDS[0]}_"):
        rel += f" (FALLBACK: this round's {PMC_ROUNDS[0]} file is absent)"
    kernels = json.load(open(path))["kernels"]
    by = lambda v: (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0
    if cls in _PMC_PS:
        # (inference runs no data gradient: every conv_ps3 / conv_ps instantiation of a sampling loop is a forward launch)
        main_pat = re.compile(r"conv_ps3?_kernel<" if (workload in ("ddim50", "ddpm1000") and cls == "conv_ps_fwd") else _PMC_PS[cls][0])
        tot = n = 0.0
        for name, v in kernels.items():
            if main_pat.search(name):
                tot += by(v) * v["launches"]; n += v["launches"]
        if not n:
            return None, f"no {cls} kernel in {rel}"
        extra = 0.0
        for pat in _PMC_PS[cls][1:]:     # second-pass kernels: one launch per call of the class
            pp = re.compile(pat)
            t2 = n2 = 0.0
            for name, v in kernels.items():
                if pp.search(name):
                    t2 += by(v) * v["launches"]; n2 += v["launches"]
            if n2:
                extra += t2 / n2
        return tot / n + extra, rel
    m = re.match(r"igemm_(\w+?)_(\d+)(_bf16x3)?$", cls)
    if not m or m.group(1) not in _PMC_OPERANDS:
        return None, "class not mapped to a kernel"
    la, lbs = _PMC_OPERANDS[m.group(1)]
    kname = "igemm_bf16x3_kernel" if m.group(3) else "igemm_kernel"
    pat = re.compile(rf"{kname}<{m.group(2)}, {m.group(2)}, bd::(\w+)<[^>]*>, bd::(\w+)<[^>]*>")
    tot = n = 0.0
    for name, v in kernels.items():
        mm = pat.search(name)
        if mm and mm.group(1) == la and mm.group(2) in lbs and "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
            tot += by(v) * v["launches"]
            n += v["launches"]
    return (tot / n, rel) if n else (None, f"no {cls} kernel in {rel}")


def _host_cores():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 hardware
    threads but a 16-CPU quota; running 256 OpenMP threads there is ~50x slower than running 16)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("BD_CPU_THREADS", cores))))


def _cpu_model():
    """CPU model string of this host (print it next to every CPU number)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _cpu_baseline_worker(kind="train"):
    """Child process: time oracle train steps (config 1: batch 16, poison 0.0) -- or, kind == "sample", oracle DDPM sampling
    steps (UNet forward + scheduler step, batch 16) -- and print one JSON line per step."""
    from oracle import sched_ref, train_ref
    from oracle import unet_ref as U
    celeba = kind == "celeba"       # the 256x256 DDPM-CELEBA-HQ-256 network (113.7 M parameters), batch 1: one step is ~10-30 s of CPU
    cfg = U.CELEBA_HQ_256 if celeba else U.CIFAR10_32
    cores = _host_cores()
    torch.set_num_threads(cores)
    P = U.gen_params(cfg, 0)
    _, a, ac = sched_ref.make_tables()
    B = 1 if celeba else 16
    S = 256 if celeba else 32
    if kind == "sample":      # 10 DDPM sampling steps at batch 16, extrapolated to 1000 / 50
        x = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(0))
        gz = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for step in range(64):
                t = 999 - step
                t0 = time.time()
                e = U.unet_forward(cfg, P, x, t)
                x, _ = sched_ref.ddpm_step(ac, e, t, x, torch.randn(x.shape, generator=gz))
                print(json.dumps({"step": step, "s": time.time() - t0, "cores": torch.get_num_threads(), "B": B}), flush=True)
        return
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    R = torch.zeros_like(x0)
    eps = torch.randn(B, 3, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    state = {}
    if kind == "anp":         # SURVEY f-4: one batch of the ANP defense loop (anp_defense.py:136-160) on the CIFAR network, batch 16
        from oracle import anp_ref
        bn = anp_ref.init_bn(cfg, P)
        trig = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        for step in range(64):
            t0 = time.time()
            _, _, _, bn, state, _ = anp_ref.anp_step(cfg, P, bn, state, a, ac, x0, trig, x0.flip(0), t, eps, 1e-4, step + 1, 4.0)
            print(json.dumps({"step": step, "s": time.time() - t0, "cores": torch.get_num_threads(), "B": B}), flush=True)
        return
    for step in range(64):
        t0 = time.time()
        loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
        P, state, _ = train_ref.clip_and_adam(P, G, state, 2e-4, step + 1)
        print(json.dumps({"step": step, "s": time.time() - t0, "cores": torch.get_num_threads(), "B": B}), flush=True)


def cpu_baseline(seconds_budget=40.0, kind="train"):
    """The reference CPU path = the oracle's fp32 restatement (oracle/), BASELINE configs[0]: batch 16, poison 0.0,
    run in a child process that is killed after `seconds_budget` so the bench always finishes in minutes.
    kind == "sample": per-step cost of the oracle's DDPM sampling loop (returns seconds per step at batch 16)."""
    import subprocess
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-baseline-kind", kind],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    recs = []
    t_start = time.time()
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(p.stdout, selectors.EVENT_READ)
    while time.time() - t_start < seconds_budget and len(recs) < 12:
        if not sel.select(timeout=1.0):
            if p.poll() is not None:
                break
            continue
        line = p.stdout.readline()
        if not line:
            break
        try:
            recs.append(json.loads(line))
        except ValueError:
            pass
    p.kill()
    if not recs:
        return {"value": None, "unit": "images/s", "cores": None, "cpu_model": _cpu_model(), "kind": "port",
                "sample": f"no oracle {kind} step finished within {seconds_budget:.0f} s on this host"}
    timed = [r["s"] for r in recs[2:]] or [r["s"] for r in recs[-1:]]
    timed.sort()
    med = timed[len(timed) // 2]
    B = recs[0]["B"]
    if kind == "celeba":
        return {"value": B / med, "unit": "images/s", "cores": recs[0]["cores"], "cpu_model": _cpu_model(), "kind": "port",
                "sample": f"{len(timed)} timed train step(s) ({len(recs) - len(timed)} warm-up) of the 256x256 DDPM-CELEBA-HQ-256 UNet, batch {B}, "
                          f"poison_rate 0.0, fp32, oracle/train_ref.py on the host CPU (median {med:.2f} s/step, budget {seconds_budget:.0f} s)"}
    if kind == "sample":
        return {"seconds_per_step": med, "batch": B, "cores": recs[0]["cores"], "cpu_model": _cpu_model(), "kind": "port",
                "timed_steps": len(timed)}
    if kind == "anp":
        return {"value": B / med, "unit": "images/s", "cores": recs[0]["cores"], "cpu_model": _cpu_model(), "kind": "port",
                "sample": f"{len(timed)} timed ANP batches ({len(recs) - len(timed)} warm-up) of the CIFAR-32 UNet, batch {B}: perturbed forward + backward, clip + "
                          f"Adam on the bn parameters, clamp, backdoor-MSE forward; fp32, oracle/anp_ref.py on the host CPU (median {med:.3f} s/batch, "
                          f"budget {seconds_budget:.0f} s)"}
    return {"value": B / med, "unit": "images/s", "cores": recs[0]["cores"], "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{len(timed)} timed train steps ({len(recs) - len(timed)} warm-up) of the CIFAR-32 UNet, batch {B}, "
                      f"poison_rate 0.0, fp32, oracle/train_ref.py on the host CPU (median {med:.3f} s/step, "
                      f"budget {seconds_budget:.0f} s)"}


def _read_classes(lib):
    cl = []
    for c in range(lib.bd_prof_num_classes()):
        name = ctypes.c_char_p(); n = ctypes.c_int64(); tms = ctypes.c_double(); fl = ctypes.c_double(); by = ctypes.c_double()
        lib.bd_prof_get(c, ctypes.byref(name), ctypes.byref(n), ctypes.byref(tms), ctypes.byref(fl), ctypes.byref(by))
        cl.append({"kernel": name.value.decode(), "launches": n.value, "ms": tms.value, "flops": fl.value, "bytes": by.value})
    cl.sort(key=lambda c: -c["ms"])
    return cl


def run_sampling(model, kind, n, mode, world, rank, dev, lib, want_roofline=True):
    """SURVEY 8(d) sampling metric: the full DDIM-50 / DDPM-1000 / PNDM-50 loop (UNet forward + scheduler step per timestep,
    pipeline_ddpm.py:106-111, pipeline_ddim.py:114-121) over `n` initial noises per GPU; chains are independent, ranks shard
    the rows, no collective (8e).  Returns the result dict (rank 0) -- whole-job samples/s, the dominant kernel's roofline
    fraction from hipEvent pairs around every GEMM-class launch of three extra, untimed UNet evaluations of the same chunk size."""
    from baddiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline, PNDMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    ddim, pndm = kind == "ddim50", kind == "pndm50"   # pndm50: what every other --sched resolves to (SURVEY f-4): 59 UNet evaluations
    steps = 50 if (ddim or pndm) else 1000
    pipe = (PNDMPipeline if pndm else DDIMPipeline if ddim else DDPMPipeline)(model, DDPMScheduler(num_train_timesteps=1000))
    pipe.set_progress_bar_config(disable=True)
    init = torch.randn(n, 3, 32, 32, generator=torch.Generator().manual_seed(rank)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    pipe(batch_size=n, init=init[:min(n, 64)], generator=gen, num_inference_steps=min(steps, 5), output_type=None)   # warm-up
    pipe(batch_size=n, init=init, generator=gen, num_inference_steps=5 if pndm else 2, output_type=None)            # + the full chunk shape (PNDM: >= 4 steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(batch_size=n, init=init, generator=gen, num_inference_steps=steps, output_type=None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    evals = len(pipe.scheduler.timesteps) if pndm else steps
    gf = 12.444 * evals                               # GFLOP per sample
    res = {"metric": f"{'PNDM 50' if pndm else 'DDIM 50' if ddim else 'DDPM 1000'}-step samples/sec (32x32 UNet, DDPM-CIFAR10-32 topology)",
           "value": world * n / dt, "unit": "samples/s", "samples_per_gpu": n, "unet_evaluations": evals,
           "seconds_per_loop": dt, "ms_per_unet_step": dt / evals * 1e3, "step_tflops": gf * n * world / dt / 1e3,
           "frac_of_fp32_mfma_peak": gf * n / dt / 1e3 / FP32_MFMA_PEAK_TFLOPS,
           "images_finite": bool(np.isfinite(np.asarray(out.images)).all()),
           "inference_chunk": getattr(model, "last_chunk", None)}       # the chunk decides which kernels the plan picks
    if want_roofline:
        lib.bd_prof_reset(); lib.bd_prof_enable(1)
        pipe(batch_size=n, init=init[:min(n, model.max_chunk)], generator=gen, num_inference_steps=5 if pndm else 3, output_type=None)
        lib.bd_prof_enable(0)
        torch.cuda.synchronize()
        cl = _read_classes(lib)
        cl_mm = [c for c in cl if c["flops"] > 0]         # (round 6: the resident GroupNorm launches are classes too -- bytes only)
        if cl_mm:
            d = cl_mm[0]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            peak = FP32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS / 3
            traffic, tsrc = _pmc_traffic(d["kernel"], "ddpm1000" if kind == "ddpm1000" else "ddim50")     # PMC run of the same chunk shape
            res["roofline"] = {"bound": "mfma", "kernel": d["kernel"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                               "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                               "share_of_gemm_class_time": d["ms"] / sum(c["ms"] for c in cl_mm),
                               "note": "hipEvent pairs on the launch stream, 3 untimed UNet evaluations of one chunk; the two half-batch "
                                       "forward pipelines share the chip, so this is an in-schedule figure"}
    return res


def sampling_cpu_baseline(seconds_budget=25.0):
    """10 oracle DDPM sampling steps at batch 16 on the host cores, extrapolated to the 1000-step
    (DDPM) and 50-step (DDIM: same UNet evaluation per step, a cheaper update) loops."""
    r = cpu_baseline(seconds_budget, kind="sample")
    if "seconds_per_step" not in r:
        return {"ddpm1000": r, "ddim50": r}
    sps, B = r["seconds_per_step"], r["batch"]
    mk = lambda nsteps: {"value": B / (sps * nsteps), "unit": "samples/s", "cores": r["cores"], "cpu_model": r["cpu_model"], "kind": "port",
                         "sample": f"{r['timed_steps']} timed oracle DDPM sampling steps (UNet forward + scheduler step, batch {B}, fp32, "
                                   f"oracle/unet_ref.py + sched_ref.py; median {sps:.3f} s/step) extrapolated x{nsteps}"}
    return {"ddpm1000": mk(1000), "ddim50": mk(50)}


def bench_sampling(args, world, rank, dev):
    """--workload ddim50 / ddpm1000 / pndm50: the sampling loop alone, one JSON line (side measurement)."""
    from baddiffusion_amd import _lib as L
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.unet import UNet2DModel
    model = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"], compute_mode=args.mode).to(dev)
    # BASELINE configs[4]: ONE job of eval_max_batch 2048 chains sharded over the GPUs (rows contiguous per rank, no collective): strong scaling.
    # --batch n keeps n chains per GPU instead (weak scaling).
    strong = args.batch <= 0
    total = args.total if args.total > 0 else 2048
    n = args.batch if args.batch > 0 else max(1, total // world)
    res = run_sampling(model, args.workload, n, args.mode, world, rank, dev, L.load(), want_roofline=not args.no_prof)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        steps = 1000 if args.workload == "ddpm1000" else 50
        line = {"metric": res["metric"], "value": res["value"], "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": 5,
                "ms_per_step": res["seconds_per_loop"] / steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
                "vs_baseline": None,
                "dtype": "f32" if args.mode == "f32" else "f32 (split-bf16 hi+lo products, fp32 accumulate)",
                "data": "synthetic (seeded N(0,1) init, seeded default-init weights)",
                "config": {"workload": f"BASELINE configs[4]-style sampling: {args.workload} loop, {n} samples per GPU, images to float NHWC "
                                       f"at the end, no PNG I/O", "global_batch": world * n,
                           "parallelism": f"replicas x{world} (rows sharded, no collective)",
                           "inference_chunk": getattr(model, "last_chunk", None)}}
        line.update({k: v for k, v in res.items() if k not in line})
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = sampling_cpu_baseline()["ddpm1000" if args.workload == "ddpm1000" else "ddim50"]
        emit(line, args)
    return 0


def run_fid_features(dev, n32=2048, n299=256, batch=None):
    """SURVEY f-3 measure path (fid_score.py:91-148): pool3 features of pytorch_fid's InceptionV3 on the device (baddiffusion_amd/inception.py:
    bilinear resize to 299 x 299 + 94 bd_conv2d_nhwc launches with folded BatchNorm + pools), seeded random weights of the published shapes (the
    real weights file is a third-party asset).  images/s for the CIFAR measure set (n32 uint8 32 x 32 images, what measure() feeds it) and for
    n299 uint8 299 x 299 images (no up-sampling); TFLOP/s = algorithmic flops of the convolutions / wall time, against the exact-fp32 MFMA peak the
    kernel computes on (v_mfma_f32_32x32x2_f32)."""
    from baddiffusion_amd.inception import FIDInceptionV3, state_dict_manifest
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in state_dict_manifest().items():
        if k.startswith("fc."):
            continue
        if k.endswith("conv.weight"):
            fan = shp[1] * shp[2] * shp[3]
            sd[k] = torch.randn(shp, generator=g) * (2.0 / fan) ** 0.5
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("bn.weight"):
            sd[k] = 0.8 + 0.4 * torch.rand(shp, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(shp, generator=g)
    net = FIDInceptionV3(sd, device=dev, **({"batch_size": batch or int(os.environ["BD_FID_BATCH"])} if (batch or "BD_FID_BATCH" in os.environ) else {}))
    out = {"weights": "seeded random, pytorch_fid InceptionV3 shapes (23.9 M parameters)", "batch_size": net.batch_size,
           "peak_tflops": FP32_MFMA_PEAK_TFLOPS, "kernel": "ic_conv2_kernel (bd_conv2d_nhwc, 128 x 64 double-buffered tile, K-contiguous weights), exact fp32 products"}
    for tag, n, S in (("cifar32", n32, 32), ("direct299", n299, 299)):
        imgs = torch.randint(0, 256, (n, S, S, 3), generator=g, dtype=torch.uint8).to(dev)
        f = net(imgs[: net.batch_size])
        torch.cuda.synchronize()
        net.flops = 0.0
        t0 = time.perf_counter()
        f = net(imgs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[tag] = {"images": n, "input": f"uint8 {S}x{S}x3 -> bilinear 299x299", "seconds": dt, "images_per_s": n / dt,
                    "gflop_per_image": net.flops / n / 1e9, "tflops": net.flops / dt / 1e12,
                    "frac_of_fp32_mfma_peak": net.flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, "features_finite": bool(torch.isfinite(f).all())}
    return out


def bench_anp(args, world, rank, dev):
    """--workload anp: one batch of the ANP defense loop (SURVEY f-4; anp_defense.py:136-160) on the DDPM-CIFAR10-32 topology: perturbed forward +
    backward (bd_anp_apply, the plan, bd_anp_grad), clip + Adam on the bn parameters, clamp, the backdoor-MSE forward.  One JSON line (side
    measurement); ranks run replicas (the reference's loop has no gradient exchange of its own beyond DataParallel; not wired here)."""
    from baddiffusion_amd import anp
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.unet import UNet2DModel
    B = args.batch if args.batch > 0 else 128
    model = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"], compute_mode=args.mode).to(dev)
    pm = anp.convert_model(model)
    sched = DDPMScheduler(num_train_timesteps=1000)
    tr = anp.AnpTrainer(pm, sched, anp.AnpConfig())
    g = torch.Generator(device=dev); g.manual_seed(rank)
    clean = torch.rand(B, 3, 32, 32, device=dev, generator=g) * 2 - 1
    trig = torch.rand(B, 3, 32, 32, device=dev, generator=g) * 2 - 1
    targ = torch.rand(B, 3, 32, 32, device=dev, generator=g) * 2 - 1
    noise = torch.randn(B, 3, 32, 32, device=dev, generator=g)
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    for _ in range(args.warmup):
        logs = tr.step(clean, trig, targ, t, noise)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = tr.step(clean, trig, targ, t, noise)
    torch.cuda.synchronize(dev)
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    if rank == 0:
        sec = float(dt)
        line = {"metric": "ANP defense images/sec (32x32 UNet, DDPM-CIFAR10-32 topology, bs%d/GPU)" % B, "value": world * B * args.steps / sec,
                "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if args.mode == "f32" else "f32 (split-bf16 hi+lo products, fp32 accumulate)",
                "data": "synthetic (seeded uniform images, seeded default-init weights, bn = (1, 0) at the start)",
                "config": {"workload": "SURVEY f-4: one batch of anp_defense.train_loop -- perturbed forward + backward, clip + Adam on the bn "
                                       "parameters, clamp to the budget, backdoor-MSE forward", "global_batch": world * B,
                           "parallelism": f"replicas x{world}", "bn_parameters": int(pm.perturb.numel())},
                "final_loss": float(logs["loss"]), "final_backdoor_mse": float(logs["backdoor_mse"])}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(30.0, kind="anp")
        emit(line, args)
    return 0


def setup_train(celeba, B, mode, dev, rank, use_graph=False):
    """model + engine + resident synthetic data for one train workload; returns (model, engine, step(i), names)"""
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    from baddiffusion_amd.unet import UNet2DModel
    topo = KNOWN_TOPOLOGIES["google/ddpm-ema-celebahq-256" if celeba else "google/ddpm-cifar10-32"]
    S_IMG = topo["sample_size"]
    model = UNet2DModel(**topo, compute_mode=mode).to(dev)
    eng = TrainEngine(model, DDPMScheduler(num_train_timesteps=1000), lr=2e-4, lr_warmup_steps=500, num_training_steps=469 * 50,
                      use_graph=use_graph)
    # synthetic data resident in HBM: uint8 images, poison flags i % 10 == 0.  BASELINE configs[1] names BOX_14 -> HAT and
    # configs[3] GLASSES -> CAT: static/*.png are reference assets and do not travel, but what the reference's Backdoor returns
    # for them does, as a reference-derived golden vector (tests/golden/img_triggers.npz, made by make_trigger_fixture.py with a
    # torchvision stand-in for the resize).  CORNER / BOX_14 only if that file is missing; the arithmetic is the same either way.
    bd = Backdoor(root=None)
    trigger = bd.get_trigger("BOX_14", 3, S_IMG).to(dev)
    trigger_name, target_name, src = "BOX_14", "CORNER", "built-in constructors"
    target = bd.get_target("CORNER", trigger.cpu())
    hat = os.path.join(ROOT, "tests", "golden", "img_triggers.npz")
    if os.path.exists(hat):
        gv = np.load(hat)
        src = "reference-derived golden vector tests/golden/img_triggers.npz (torchvision stand-in for the resize)"
        if not celeba:
            target, target_name = torch.from_numpy(gv["target_HAT_c3_s32"]), "HAT"
        else:
            trigger, trigger_name = torch.from_numpy(gv["trigger_GLASSES_c3_s256"]).to(dev), "GLASSES"
            target, target_name = torch.from_numpy(gv["target_CAT_c3_s256"]), "CAT"
    target = target.to(dev)
    NIMG = 256 if celeba else 8192
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.randint(0, 256, (NIMG, S_IMG, S_IMG, 3), generator=g, dtype=torch.uint8).to(dev)
    flags = (torch.arange(NIMG) % 10 == 0).to(dev)
    NPOOL = 2 if celeba else 8
    noise = torch.randn(NPOOL, B, 3, S_IMG, S_IMG, generator=g).to(dev)
    ts = torch.randint(0, 1000, (NPOOL, B), generator=g).to(dev)

    def inputs(i):
        s0 = (i * B) % (NIMG - B + 1)
        return images[s0:s0 + B], flags[s0:s0 + B], trigger, target, noise[i % NPOOL], ts[i % NPOOL]

    def step(i):
        return eng.train_step(*inputs(i))
    step.inputs = inputs
    return model, eng, step, {"trigger": trigger_name, "target": target_name, "target_source": src, "S": S_IMG}


def profile_steps(lib, model, step, first, n, barrier):
    """(classes in the product's two-stream schedule, classes with the side stream off, n): per-launch hipEvent pairs on the
    launch stream over `n` extra untimed steps each.  In the two-stream schedule the weight-gradient GEMMs share the chip with
    the dgrad / GroupNorm chain, so their durations there include that sharing; every rank runs these steps (DP collectives
    must stay matched)."""
    lib.bd_prof_reset(); lib.bd_prof_enable(1)
    for i in range(n):
        step(first + i)
    lib.bd_prof_enable(0)
    barrier()
    classes = _read_classes(lib)
    lib.bd_unet_set_aux_stream(model._plan, 0)
    lib.bd_prof_reset(); lib.bd_prof_enable(1)
    for i in range(2):
        step(first + n + i)
    lib.bd_prof_enable(0)
    barrier()
    classes_iso = _read_classes(lib)
    lib.bd_unet_set_aux_stream(model._plan, 1)
    lib.bd_prof_reset()
    return classes, classes_iso, n


def roofline_object(classes, classes_iso, prof_steps, mode, ms, probe, workload=""):
    """`roofline` of the dominant kernel class (most time in the profiled steps) + the per-class tables."""
    mm = [c for c in classes if c["flops"] > 0]       # matrix classes; the GroupNorm classes (round 6) carry required bytes only
    if not mm:
        return None, [], []
    d = mm[0]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    # split-bf16 issues 3 bf16 MFMA products per algorithmic multiply: its ceiling for ALGORITHMIC flops is
    # the dense bf16 peak / 3 (833 TF), above the exact-fp32 MFMA peak (157 TF) the f32 mode is bound by
    peak = FP32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS / 3
    traffic, traffic_source = _pmc_traffic(d["kernel"], workload)
    plim = probe["random_operands_tflops"] if probe else None
    r = {"bound": "mfma", "kernel": d["kernel"], "achieved": ach, "peak": peak,
         "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
         "mfma_flops_per_algorithmic_flop": 1 if mode == "f32" else 3,
         "mfma_dense_peak": FP32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS,
         "mfma_power_limited_peak_measured": plim,
         "mfma_power_limited_peak_source": probe["source"] if probe else None,
         "mfma_constant_operand_peak_measured": probe["constant_operands_tflops"] if probe else None,
         "frac_of_power_limited_peak": ach / (plim / 3) if plim else None,
         "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
         "launches_per_step": d["launches"] / prof_steps, "sampled_steps": prof_steps,
         "sampled_steps_note": "extra untimed steps behind the timed region (same two-stream schedule); the timed steps carry no events",
         "avg_launch_us": d["ms"] * 1e3 / d["launches"],
         "gflop_per_launch": d["flops"] / d["launches"] / 1e9,
         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
         "alg_gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9,
         "share_of_step": d["ms"] / prof_steps / ms,
         "gflop_per_step_all_classes": sum(c["flops"] for c in classes) / prof_steps / 1e9}
    iso = next((c for c in classes_iso if c["kernel"] == d["kernel"]), None)
    if iso:   # the same kernel class with the side stream off (stand-alone launch durations, untimed steps)
        ai = iso["flops"] / (iso["ms"] * 1e-3) / 1e12
        r["standalone"] = {"achieved": ai, "frac": ai / peak, "avg_launch_us": iso["ms"] * 1e3 / iso["launches"],
                           "frac_of_power_limited_peak": ai / (plim / 3) if plim else None,
                           "note": "2 extra untimed steps with bd_unet_set_aux_stream(0): no overlap with other kernels"}
    rnd = lambda cl: [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items()} for c in cl]
    return r, rnd(classes), rnd(classes_iso)


def timed_steps(step, first, n, barrier, dev, world, prof=None):
    """EXACTLY n steps bracketed by barrier + synchronize on both sides; returns (wall seconds max over ranks, last loss,
    per-step milliseconds from events on the launch stream)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    barrier()
    t0 = time.perf_counter()
    loss = None
    for i in range(n):
        ev[i].record()
        if prof:
            prof(i, True)
        loss = step(first + i)
        if prof:
            prof(i, False)
    ev[n].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    return dt, loss, per


def measure_dp(eng, model, step, lib, barrier, dev, world, args, log, first):
    """N > 1, behind the contract region (every rank runs the same sequence, so collectives stay matched): (1) per-bucket collective time from
    events on the communication stream; (2) EXPOSED communication = the step with the collectives minus the same step with every part of the
    data-parallel path except the collective calls (TrainEngine.set_collectives(False); the replicas are re-synchronised from rank 0 afterwards);
    (3) sweeps of the bucket size and of the large weight-gradient kernel's K-split slot count (collectives take CUs from kernels sized for all
    256) -- one multi-GPU lease yields the scaling curve AND the two settings that matter for it.  Each point: 2 untimed + 6 timed steps under the
    same barrier rule as the contract region."""
    state = {"i": first}

    def timed(n=6, warm=2):
        for _ in range(warm):
            step(state["i"]); state["i"] += 1
        dt, _, _ = timed_steps(step, state["i"], n, barrier, dev, world)
        state["i"] += n
        return dt / n * 1e3

    from baddiffusion_amd import ops
    default_mb = float(os.environ.get("BD_DP_BUCKET_MB", "32"))
    out = {}
    try:
        direct = getattr(eng, "_rccl", None) not in (None, "none")
        out.update({"transport": "rccl-direct" if direct else "c10d", "ms_per_step": timed(8)})
        eng.time_buckets = True
        for _ in range(4):
            step(state["i"]); state["i"] += 1
        eng.time_buckets = False
        out["buckets_timed"] = [{"closes_with_segment": s, "bytes": b, "ms_on_comm_stream": ms, "launches": n,
                                 "ring_bus_GBs": 2.0 * (world - 1) / world * b / (ms * 1e-3) / 1e9 if ms > 0 else None}
                                for s, b, ms, n in eng.bucket_times()]
        out["buckets_timed_note"] = ("events on the engine's communication stream around each bucket's grouped ncclAllReduce, 4 steps, in schedule (beside the "
                                     "backward kernels)" if direct else "c10d transport: the collectives run on the process group's own stream / threads, "
                                     "these events do not bracket them")
        barrier()
        eng.set_collectives(False)
        out["ms_per_step_without_collectives"] = timed(8)
        eng.set_collectives(True)
        eng.sync_state()           # the replicas applied their own gradients for those steps: start again from rank 0's state
        barrier()
        out["exposed_comm_ms"] = out["ms_per_step"] - out["ms_per_step_without_collectives"]
        if not args.no_dp_sweep:
            sw = []
            for mb in [float(v) for v in args.dp_sweep_buckets.split(",") if v.strip()]:
                eng.set_buckets(mb)
                sw.append({"bucket_mb": mb, "buckets": len(eng._buckets), "collectives_per_step": sum(len(rs) for _, rs in eng._buckets),
                           "ms_per_step": timed()})
                log(f"dp sweep: bucket {mb:g} MB -> {sw[-1]['buckets']} buckets, {sw[-1]['ms_per_step']:.2f} ms/step")
            eng.set_buckets(default_mb)
            out["bucket_sweep"] = sw
            ss = []
            for sl in [int(v) for v in args.dp_sweep_slots.split(",") if v.strip()]:
                L_check = ops.tune_set("ps_wg3_slots", sl, check=False)     # clears every model's pooled workspaces: sizes follow the slot count
                if L_check != 0:
                    ss.append({"ps_wg3_slots": sl, "error": "bd_tune_set refused"}); continue
                ss.append({"ps_wg3_slots": sl, "ms_per_step": timed()})
                log(f"dp sweep: wgrad3 slots {sl} -> {ss[-1]['ms_per_step']:.2f} ms/step")
            ops.tune_set("ps_wg3_slots", 0)
            out["wgrad3_slot_sweep"] = ss
            out["sweep_note"] = ("defaults: BD_DP_BUCKET_MB=%g, ps_wg3_slots = 3/4 of the CUs; each point 2 untimed + 6 timed steps behind the contract region; "
                                 "set BD_DP_BUCKET_MB / BD_PS_WG3_SLOTS in the environment to make a value the default of a run" % default_mb)
    finally:
        # whatever happened above, leave the engine as the later measurements (celeba, sampling) and the other ranks expect it: collectives on,
        # default buckets and slot count; main() agrees on the outcome collectively before anyone continues
        eng.time_buckets = False
        eng.set_collectives(True)
        eng.set_buckets(default_mb)
        ops.tune_set("ps_wg3_slots", 0)
    return out


HEADLINE_MAX_BYTES = 8192          # the driver captures a bounded tail of stdout: the ONE line must fit it (round 5's 20 KB line did not parse)


def _rnd(v, nd=4):
    """floats to `nd` significant digits (the detail file keeps full precision)"""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}") if np.isfinite(v) else None
    return v


def _pick(d, keys, nd=4):
    return {k: _rnd(d[k], nd) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _roofline_headline(r, full=True):
    if not isinstance(r, dict):
        return None
    keys = (("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "launches_per_step",
             "frac_of_power_limited_peak", "mfma_power_limited_peak_measured") if full else ("kernel", "frac", "traffic"))
    o = _pick(r, keys, 5)
    if full and isinstance(r.get("standalone"), dict):
        o["standalone"] = _pick(r["standalone"], ("frac", "avg_launch_us"))
    return o


def headline_line(out, detail_file=None):
    """The ONE JSON line of the contract, bounded (< HEADLINE_MAX_BYTES): the contract's keys, `roofline` and `cpu_baseline` of the timed
    workload, and for the riders (`sampling.*`, `celeba`, `fid_features`) value / ms_per_step / roofline.frac / roofline.traffic /
    cpu_baseline.value only.  Everything else -- per-class kernel tables, sweeps, per-bucket times, the notes saying how each figure was
    taken -- is the DETAIL object `out`, written to `detail_file` (gpurun_out/bench_detail.json) and copied under profiles/ per round."""
    h = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data") if k in out}
    h["config"] = dict(out["config"])
    for k in ("final_loss", "ms_per_step_median", "step_tflops", "step_frac_of_hbm_roofline", "images_finite", "final_backdoor_mse"):
        if k in out:
            h[k] = _rnd(out[k], 5)
    if isinstance(out.get("sustained"), dict):
        h["sustained"] = _pick(out["sustained"], ("steps", "seconds", "ms_per_step"))
    if "roofline" in out:
        h["roofline"] = _roofline_headline(out["roofline"])
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        h["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "cpu_model", "kind"))
        h["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:110]

    def rider(d):
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": str(d["error"])[:120]}
        o = _pick(d, ("value", "unit", "ms_per_step", "ms_per_unet_step", "samples_per_gpu", "global_samples", "scaling"))
        if "roofline" in d:
            o["roofline"] = _roofline_headline(d["roofline"], full=False)
        if isinstance(d.get("cpu_baseline"), dict):
            o["cpu_baseline"] = _pick(d["cpu_baseline"], ("value", "cores"))
        return o
    if isinstance(out.get("sampling"), dict):
        h["sampling"] = {k: rider(v) for k, v in out["sampling"].items()}
    if "celeba" in out:
        h["celeba"] = rider(out["celeba"])
        if isinstance(out["celeba"], dict) and "step_frac_of_hbm_roofline" in out["celeba"]:
            h["celeba"]["step_frac_of_hbm_roofline"] = _rnd(out["celeba"]["step_frac_of_hbm_roofline"])
    ff = out.get("fid_features")
    if isinstance(ff, dict):
        c = ff.get("cifar32", {})
        h["fid_features"] = ({"error": str(ff["error"])[:120]} if "error" in ff else
                             {"value": _rnd(c.get("images_per_s")), "unit": "images/s", "tflops": _rnd(c.get("tflops")),
                              "frac_of_fp32_mfma_peak": _rnd(c.get("frac_of_fp32_mfma_peak"))})
    dd = out.get("distributed")
    if isinstance(dd, dict):
        hd = _pick(dd, ("world", "backend", "rccl_ranks", "collectives_per_step", "segments", "buckets", "bytes_per_step"))
        hd["transport"] = ("rccl-direct" if dd.get("rccl_ranks") else "c10d" if dd.get("world", 1) > 1 else "none")
        if dd.get("transport_fallback_reason"):
            hd["transport_fallback_reason"] = str(dd["transport_fallback_reason"])[:120]
        me = dd.get("measured")
        if isinstance(me, dict):
            hd["measured"] = ({"error": str(me["error"])[:120]} if "error" in me else
                              _pick(me, ("ms_per_step", "ms_per_step_without_collectives", "exposed_comm_ms")))
        h["distributed"] = hd
    dp = out.get("dp_path_at_world_1")
    if isinstance(dp, dict):
        h["dp_path_at_world_1"] = ({"error": str(dp["error"])[:120]} if "error" in dp else
                                   _pick(dp, ("ms_per_step", "ms_per_step_without_the_rccl_call", "host_enqueue_ms_plain")))
    if detail_file:
        h["detail_file"] = detail_file
    line = json.dumps(h)
    if len(line) >= HEADLINE_MAX_BYTES:       # never print a line the driver cannot capture: drop the riders, keep the contract
        for k in ("dp_path_at_world_1", "fid_features", "celeba", "sampling", "distributed", "sustained"):
            h.pop(k, None)
            line = json.dumps(h)
            if len(line) < HEADLINE_MAX_BYTES:
                break
    assert len(line) < HEADLINE_MAX_BYTES, len(line)
    return line


def emit(out, args):
    """write the detail object next to the run (and say where), print the bounded headline as the LAST stdout write"""
    path = args.detail_file
    rel = None
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            rel = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
        except OSError as e:
            rel = f"(not written: {e})"
    print(headline_line(out, rel), file=_JSON_OUT, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 128 for cifar, 4 for celeba)")
    ap.add_argument("--workload", default="cifar", choices=["cifar", "celeba", "ddim50", "ddpm1000", "pndm50", "anp", "fid"],
                    help="cifar = BASELINE configs[1] (the metric; its line also carries the sampling loops and the 256x256 step as "
                         "`sampling` / `celeba` objects); celeba = the 256x256 DDPM-CELEBA-HQ-256 topology alone; "
                         "ddim50 / ddpm1000 / pndm50 = one CIFAR sampling loop alone (samples/s, --batch samples per GPU, --steps ignored); "
                         "anp = one batch of the ANP defense loop on the CIFAR network (SURVEY f-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-sampling", action="store_true", help="skip the DDIM-50 x 2048 and DDPM-1000 x 256 loops of the default line")
    ap.add_argument("--sampling-n", default="2048,256", help="samples per GPU of the DDIM-50 and DDPM-1000 loops (BASELINE configs[4]'s "
                                                             "eval_max_batch 2048; the reference's default eval batch 256)")
    ap.add_argument("--no-fid", action="store_true", help="skip the FID feature-extractor measurement of the default line")
    ap.add_argument("--no-celeba", action="store_true", help="skip the 256x256 batch-4 train step of the default line")
    ap.add_argument("--no-dp-probe", action="store_true", help="skip the N = 1 measurement of the data-parallel communication path "
                                                               "(TrainEngine(force_dp=True): RCCL on a 1-rank group)")
    ap.add_argument("--no-dp-sweep", action="store_true", help="N > 1: skip the bucket-size / weight-gradient-slot sweep and the exposed-communication "
                                                               "measurement behind the timed region")
    ap.add_argument("--dp-sweep-buckets", default=os.environ.get("BD_BENCH_SWEEP_BUCKETS", "0,8,32,64,1000000"),
                    help="N > 1: BD_DP_BUCKET_MB values timed behind the contract region (0 = one bucket per backward segment, 1000000 = ONE bucket)")
    ap.add_argument("--dp-sweep-slots", default=os.environ.get("BD_BENCH_SWEEP_SLOTS", "128,160,192,224"),
                    help="N > 1: K-split slot counts of the large weight-gradient kernel (bd_tune_set ps_wg3_slots) timed behind the contract region: "
                         "collectives take CUs from kernels sized for all 256")
    ap.add_argument("--total", type=int, default=0, help="sampling workloads: GLOBAL number of chains, sharded over the ranks (strong scaling; "
                                                         "default 2048 = BASELINE configs[4]'s eval_max_batch); --batch gives a per-GPU count instead")
    ap.add_argument("--sustain", type=float, default=10.0, help="seconds of back-to-back train steps after the timed region "
                                                                "(a sustained figure on a power-limited part); 0 = skip")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("BD_TRAIN_GRAPH", "0")),
                    help="1: replay the train step as one hipGraph (TrainEngine(use_graph=True)); sampled roofline steps stay eager")
    ap.add_argument("--mode", default=os.environ.get("BD_COMPUTE_MODE", "bf16x3"), choices=["f32", "bf16x3"],
                    help="contraction arithmetic: exact fp32 MFMA, or split-bf16 (hi+lo, 3 MFMAs, ~2^-16 rel. error)")
    ap.add_argument("--detail-file", default=os.environ.get("BD_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json")),
                    help="where the DETAIL object goes (per-class kernel tables, sweeps, per-bucket times, notes); the stdout line stays < 8 KB "
                         "and names this file as `detail_file`; '' = do not write")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-kind", default="train", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return _cpu_baseline_worker(args.cpu_baseline_kind)

    # ONE JSON line on stdout: native libraries print there too (RCCL: "Librccl path : ..."), so file descriptor 1 is pointed
    # at stderr for the life of the process and the line goes to a private duplicate of the original stdout
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(1, ndev)      # BD_DIST_BACKEND=gloo lets several ranks share one GPU (functional test)
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("BD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from baddiffusion_amd import _lib as L

    # DDPM-CIFAR10-32 topology, torch default init with seed 0 (no hub weights offline)
    torch.manual_seed(0)
    if args.workload in ("ddim50", "ddpm1000", "pndm50"):
        return bench_sampling(args, world, rank, dev)
    if args.workload == "anp":
        return bench_anp(args, world, rank, dev)
    if args.workload == "fid":         # the measure path's feature extractor alone (side measurement; also rides in the default line)
        res = run_fid_features(dev, batch=args.batch if args.batch > 0 else None)
        if rank == 0:
            emit({"metric": "FID InceptionV3 pool3 features, images/s (2048 CIFAR-size images)", "value": res["cifar32"]["images_per_s"],
                  "unit": "images/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": res["cifar32"]["seconds"] * 1e3,
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded uint8 images)",
                  "config": {"workload": "SURVEY f-3: fid_score.py:91-148 feature extraction on the device"}, "fid_features": res}, args)
        return 0
    celeba = args.workload == "celeba"      # BASELINE configs[3] topology (256x256, 113.7 M params)
    B = args.batch if args.batch > 0 else (4 if celeba else 128)
    model, eng, step, names = setup_train(celeba, B, args.mode, dev, rank, bool(args.graph))
    S_IMG = names["S"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    lib = L.load()
    log(f"setup done; workspace {model.workspace_bytes(B, True) / 2**30:.2f} GiB; warmup {args.warmup} steps")
    for i in range(args.warmup):
        loss = step(i)
        if i == 0:
            torch.cuda.synchronize(); log(f"first step done, loss {float(loss):.5f}")
    log("warmup done")
    # The timed region carries NO instrumentation (round 3 bracketed 2 of the 20 timed steps with hipEvent pairs, ~5 % each).
    dt, loss, per_step = timed_steps(step, args.warmup, args.steps, barrier, dev, world)
    final_loss = float(loss)
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")
    # roofline: hipEvent pairs around every GEMM-class launch (bd_prof_*) of PROF_STEPS extra, UNTIMED steps of the same
    # two-stream schedule, then of two more with the side stream off (stand-alone launch durations)
    PROF_STEPS = 3
    classes, classes_iso, prof_steps = profile_steps(lib, model, step, args.warmup + args.steps, PROF_STEPS, barrier) if not args.no_prof else ([], [], 0)

    sustained = None
    if args.sustain > 0:
        # a power-limited part: how the step holds up over >= `sustain` seconds of back-to-back steps (same barrier rule)
        n_sus = max(args.steps, int(args.sustain / (dt / args.steps)) + 1)
        sdt, _, sper = timed_steps(step, args.warmup + args.steps + 2, n_sus, barrier, dev, world)
        sper.sort()
        sustained = {"seconds": sdt, "steps": n_sus, "ms_per_step": sdt / n_sus * 1e3, "value": world * B * n_sus / sdt,
                     "ms_per_step_median": sper[len(sper) // 2], "ms_per_step_p90": sper[int(len(sper) * 0.9)]}
        log(f"sustained: {n_sus} steps in {sdt:.1f} s = {sdt / n_sus * 1e3:.2f} ms/step")

    dp_meas = None
    if world > 1 and not celeba:
        try:
            dp_meas = measure_dp(eng, model, step, lib, barrier, dev, world, args, log, args.warmup + 4 * args.steps + 100000)
            log(f"dp: {dp_meas['ms_per_step']:.2f} ms/step, without collectives {dp_meas['ms_per_step_without_collectives']:.2f}")
        except Exception as e:        # never lose the headline line to a side measurement
            dp_meas = {"error": f"{type(e).__name__}: {e}"}
        # a rank that failed has left the sequence of collectives the others are in: agree on the outcome, and when any rank
        # failed, every rank re-synchronises its replica from rank 0 before the later measurements
        okf = torch.tensor([0 if (dp_meas is None or "error" in dp_meas) else 1], device=dev, dtype=torch.int32)
        try:
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0:
                if "error" not in (dp_meas or {}):
                    dp_meas = dict(dp_meas or {}, error="the data-parallel side measurement failed on another rank")
                eng.sync_state()
                barrier()
        except Exception as e:
            dp_meas = dict(dp_meas or {}, error=f"agreement after a failed side measurement failed too: {type(e).__name__}: {e}")

    dp_probe = None
    if world == 1 and not celeba and not args.no_dp_probe:
        # the data-parallel communication path at N = 1 (task 3): a second engine over the same model with
        # force_dp -- comm stream, side-stream event wait, one grouped ncclAllReduce per bucket on a 1-rank communicator.  Measured
        # twice: with the RCCL call, and with everything but that call (BD_DP_TRANSPORT=none) -- RCCL's 1-rank path performs a
        # host-synchronous operation per call, which stalls the host's run-ahead (host_enqueue_ms); its multi-rank path launches a
        # kernel instead.
        try:
            from baddiffusion_amd.trainer import TrainEngine
            from baddiffusion_amd.schedulers import DDPMScheduler

            def host_and_gpu(fn, n=10):
                for i in range(3):
                    fn(i)
                barrier()
                t0 = time.perf_counter()
                for i in range(n):
                    fn(3 + i)
                th = time.perf_counter() - t0
                barrier()
                return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3

            res = {}
            for transport in ("rccl", "none"):
                os.environ["BD_DP_TRANSPORT"] = transport
                e2 = TrainEngine(model, DDPMScheduler(num_train_timesteps=1000), lr=2e-4, lr_warmup_steps=500, num_training_steps=469 * 50,
                                 force_dp=True)
                ms_, host_ = host_and_gpu(lambda i, _e=e2: _e.train_step(*step.inputs(i)))
                res[transport] = (ms_, host_, e2)
            os.environ.pop("BD_DP_TRANSPORT", None)
            pms, phost = host_and_gpu(step)      # the ordinary step again, back to back on the same clocks
            e2 = res["rccl"][2]
            dp_probe = {"ms_per_step": res["rccl"][0], "host_enqueue_ms": res["rccl"][1],
                        "ms_per_step_without_the_rccl_call": res["none"][0], "host_enqueue_ms_without_the_rccl_call": res["none"][1],
                        "ms_per_step_plain_back_to_back": pms, "host_enqueue_ms_plain": phost,
                        "overhead_ms": res["rccl"][0] - pms, "overhead_ms_without_the_rccl_call": res["none"][0] - pms,
                        "transport": "RCCL called directly (baddiffusion_amd/rccl.py), 1-rank communicator", "world": e2.world,
                        "collectives_per_step": sum(len(rs) for _, rs in e2._buckets), "buckets": len(e2._buckets),
                        "bytes_per_step": int(e2.collective_bytes),
                        "note": "TrainEngine(force_dp=True): trainer.py's DP branch on one GPU; tests/test_hip_round4.py checks its stream ordering"}
            log(f"dp path at world 1: {res['rccl'][0]:.2f} ms/step, without the RCCL call {res['none'][0]:.2f} (plain {pms:.2f})")
            torch.cuda.synchronize()
            for _t in ("rccl", "none"):
                _c = getattr(res[_t][2], "_rccl", None)
                if _c not in (None, "none"):
                    _c.destroy()
            del e2, res
        except Exception as e:      # a box without a working RCCL init must not lose the headline line
            dp_probe = {"error": f"{type(e).__name__}: {e}"}

    probe = None
    if args.mode != "f32":
        # what the matrix pipe alone sustains on this board right now (register-operand MFMA loop, random bf16 operands)
        tf_r, tf_c = ctypes.c_double(), ctypes.c_double()
        L.check(lib.bd_mfma_probe(1, 40000, 12, ctypes.byref(tf_r), L.stream()), "bd_mfma_probe")
        L.check(lib.bd_mfma_probe(0, 40000, 4, ctypes.byref(tf_c), L.stream()), "bd_mfma_probe")
        probe = {"random_operands_tflops": tf_r.value, "constant_operands_tflops": tf_c.value,
                 "source": "bd_mfma_probe, measured in this process after the timed region (v_mfma_f32_32x32x16_bf16 on register operands, "
                           "2 x 512-thread workgroups per CU)"}
        log(f"mfma probe: {tf_r.value:.0f} TFLOP/s random operands, {tf_c.value:.0f} constant")

    sampling = None
    if not celeba and not args.no_sampling:
        sampling = {}
        from baddiffusion_amd.unet import UNet2DModel
        from baddiffusion_amd.model import KNOWN_TOPOLOGIES
        smodel = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"], compute_mode=args.mode).to(dev)
        n_ddim, n_ddpm = (int(v) for v in args.sampling_n.split(","))
        for kind, n in (("ddim50", n_ddim), ("ddpm1000", n_ddpm)):
            sampling[kind] = run_sampling(smodel, kind, n, args.mode, world, rank, dev, lib, want_roofline=not args.no_prof)
            log(f"{kind}: {sampling[kind]['value']:.1f} samples/s ({sampling[kind]['seconds_per_loop']:.1f} s)")
        if world > 1:
            # BASELINE configs[4] itself: ONE job of eval_max_batch 2048 DDIM-50 chains sharded over the ranks (strong scaling, no collective)
            tot = args.total if args.total > 0 else 2048
            sh = run_sampling(smodel, "ddim50", max(1, tot // world), args.mode, world, rank, dev, lib, want_roofline=False)
            sh["scaling"] = "strong"; sh["global_samples"] = max(1, tot // world) * world
            sampling["ddim50_sharded"] = sh
            log(f"ddim50 sharded ({sh['global_samples']} chains over {world} ranks): {sh['value']:.1f} samples/s")
        del smodel
        torch.cuda.empty_cache()

    fid_feat = None
    if not celeba and not args.no_fid and rank == 0:
        try:
            fid_feat = run_fid_features(dev)
            log(f"fid features: {fid_feat['cifar32']['images_per_s']:.0f} images/s ({fid_feat['cifar32']['tflops']:.1f} TFLOP/s)")
        except Exception as e:
            fid_feat = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()

    side = None
    if not celeba and not args.no_celeba:
        cmodel, ceng, cstep, cnames = setup_train(True, 4, args.mode, dev, rank)
        for i in range(3):
            cstep(i)
        cdt, closs, cper = timed_steps(cstep, 3, 10, barrier, dev, world)
        cper.sort()
        side = {"metric": "train images/sec (256x256 UNet, DDPM-CELEBA-HQ-256 topology, bs4/GPU, poison_rate 0.1)",
                "value": world * 4 * 10 / cdt, "unit": "images/s", "steps": 10, "warmup": 3, "ms_per_step": cdt / 10 * 1e3,
                "ms_per_step_median": cper[5], "global_batch": 4 * world, "params": int(cmodel.num_flat), "final_loss": float(closs),
                "step_tflops": 1490.63 * 4 * world / cdt * 10 / 1e3,
                "step_frac_of_hbm_roofline": (16688e6 * 4 + 6.65e9) / 8e12 / (cdt / 10),
                "workload": f"BASELINE configs[3] topology: {cnames['trigger']} trigger, {cnames['target']} target, clip 1.0 + Adam"}
        log(f"celeba256 B=4: {cdt / 10 * 1e3:.2f} ms/step")
        if not args.no_prof:
            ccl, ccl_iso, cn = profile_steps(lib, cmodel, cstep, 13, 2, barrier)
            side["roofline"], side["kernel_classes"], side["kernel_classes_standalone"] = \
                roofline_object(ccl, ccl_iso, cn, args.mode, cdt / 10 * 1e3, probe, "celeba")
        if getattr(ceng, "_rccl", None) not in (None, "none"):
            torch.cuda.synchronize()
            ceng._rccl.destroy()
        del cmodel, ceng, cstep
        torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        gflop_img = 1490.63 if celeba else TRAIN_GFLOP_PER_IMG
        hbm_floor_ms = (16688e6 * B + 6.65e9 if celeba else 452.3e6 * B + 2.25e9) / 8e12 * 1e3   # eager-level bytes / 8 TB/s
        target_name, trigger_name = names["target"], names["trigger"]
        if celeba:
            metric = f"train images/sec (256x256 UNet, DDPM-CELEBA-HQ-256 topology, bs{B}/GPU, poison_rate 0.1)"
            workload = (f"BASELINE configs[3] topology: DDPM-CELEBA-HQ-256 train step, batch {B}/GPU, poison_rate 0.1, {trigger_name} trigger, "
                        f"{target_name} target, clip 1.0 + Adam (side measurement, not the headline metric)")
        else:
            metric = "train images/sec (32x32 UNet, DDPM-CIFAR10-32 topology, bs128/GPU, poison_rate 0.1)"
            workload = ("BASELINE configs[1]: CIFAR10 DDPM-CIFAR10-32 train step, batch 128/GPU, poison_rate 0.1, "
                        f"BOX_14 trigger, {target_name} target" + ("" if target_name == "HAT" else " (HAT stand-in)") + ", clip 1.0 + Adam, fp32 storage")
        srt = sorted(per_step)
        nseg = len(eng._seg_ranges)
        coll = [sum(hi - lo for lo, hi in rs) * 4 for _, rs in eng._buckets]
        out = {"metric": metric,
               "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.mode == "f32" else "f32 (split-bf16 hi+lo products, fp32 accumulate)",
               "data": f"synthetic (uint8 {S_IMG}x{S_IMG}x3 images resident in HBM, seeded default-init weights)",
               "trigger_target_source": names["target_source"],
               "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}",
                          "params": int(model.num_flat)},
               "final_loss": final_loss,
               "ms_per_step_median": srt[len(srt) // 2], "ms_per_step_min": srt[0], "ms_per_step_max": srt[-1],
               "step_tflops": gflop_img * B * world / (ms * 1e-3) / 1e3,
               "step_frac_of_fp32_mfma_peak": gflop_img * B / (ms * 1e-3) / 1e3 / FP32_MFMA_PEAK_TFLOPS,
               "step_frac_of_hbm_roofline": hbm_floor_ms / ms,
               "distributed": {"world": world, "backend": backend if world > 1 else "none (single process)",
                               "gradient_transport": ("RCCL called directly on the engine's communication stream (baddiffusion_amd/rccl.py)"
                                                      if eng._rccl is not None else ("torch.distributed (c10d)" if world > 1 else "none")),
                               "rccl_ranks": (eng._rccl.world if getattr(eng, "_rccl", None) not in (None, "none") else None),
                               "rccl_self_test": ("passed: ncclAllReduce of a rank-dependent vector equals its closed form on every rank (collective MIN)"
                                                  if getattr(eng, "_rccl", None) not in (None, "none") else None),
                               "transport_fallback_reason": getattr(eng, "transport_note", None),
                               "collectives_per_step": sum(len(rs) for _, rs in eng._buckets) if world > 1 else 0,
                               "segments": nseg, "buckets": len(eng._buckets), "bytes_per_bucket": coll, "bytes_per_step": sum(coll),
                               "note": "backward segments are fused into buckets of >= BD_DP_BUCKET_MB (32); when a bucket's last segment has run, "
                                       "its ranges of the flat fp32 gradient are all-reduced (sum; loss gradient pre-scaled by 1/world) from the "
                                       "communication stream while the next segments compute"}}
        if sustained:
            out["sustained"] = sustained
        if not args.no_prof and classes:
            out["roofline"], out["kernel_classes"], out["kernel_classes_standalone"] = \
                roofline_object(classes, classes_iso, prof_steps, args.mode, ms, probe, "celeba" if celeba else "")
        if dp_probe:
            out["dp_path_at_world_1"] = dp_probe
        if dp_meas:
            out["distributed"]["measured"] = dp_meas
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(kind="celeba") if celeba else cpu_baseline()
            if side:
                log("cpu baseline of the 256x256 step ...")
                side["cpu_baseline"] = cpu_baseline(seconds_budget=60.0, kind="celeba")
            if sampling:
                sb = sampling_cpu_baseline()
                for k in sampling:
                    if k in sb:
                        sampling[k]["cpu_baseline"] = sb[k]
        if sampling:
            out["sampling"] = sampling
        if side:
            out["celeba"] = side
        if fid_feat:
            out["fid_features"] = fid_feat
        final_out = out
    else:
        final_out = None
    if getattr(eng, "_rccl", None) not in (None, "none"):      # tear the gradient communicator down while the HIP runtime is still up
        torch.cuda.synchronize()
        eng._rccl.destroy()
    if dist.is_available() and dist.is_initialized():      # world > 1 (or a c10d 1-rank group of an A/B run)
        dist.destroy_process_group()
    if final_out is not None:      # the LAST thing this process writes: nothing a library prints at teardown can follow it
        emit(final_out, args)


if __name__ == "__main__":
    main()
