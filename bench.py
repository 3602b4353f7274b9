"""bench.py -- headline benchmark of the BadDiffusion hot path on MI355X.

metric (BASELINE.json): train images/sec of the 32x32 UNet (DDPM-CIFAR10-32 topology), batch 128 per GPU,
poison_rate 0.1, BOX_14 trigger, fp32, synthetic data.  One "step" = poison-blend + q_sample -> UNet forward
-> MSE -> UNet backward (+ RCCL all-reduce of the 143 MB flat gradient when N > 1) -> clip + Adam.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Inputs (uint8 images, noise, timesteps) are resident in HBM before the timed
region.  `roofline` is measured live with hipEvent pairs around every igemm launch (bd_prof_*), `cpu_baseline`
is the CPU oracle (pure PyTorch fp32 restatement of the reference path) timed on this box's host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_IMG = 37.324          # SURVEY 8d (fwd+bwd, 2*MAC of conv/GEMM/BMM), CIFAR-32 UNet
FP32_MFMA_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA (split-bf16 mode issues 3 MFMA flops per algorithmic flop)
HBM_PEAK_GBS = 8000.0


_PMC_OPERANDS = {"conv_fwd": ("ConvKC", ("WgtKC", "WgtKCs")), "conv_dgrad": ("TConvKC", ("WgtRC", "WgtRCs")),
                 "conv_wgrad": ("DenseRC", ("ConvRC", "ConvRCs")), "gemm_nt": ("DenseKC", ("DenseKC",)),
                 "gemm_nn": ("DenseKC", ("DenseRC",)), "gemm_tn": ("DenseRC", ("DenseRC",))}
# kernel classes of the LDS-DMA family (conv_ps.hip) -> kernel-name patterns whose HBM traffic makes up one call
_PMC_PS = {"conv_ps_wgrad": (r"conv_ps_wgrad_kernel<", r"conv_ps_wgrad_reduce"), "conv_ps_fwd": (r"conv_ps_kernel<[12]>",),
           "conv_ps_dgrad": (r"conv_ps_kernel<0>",), "conv_ps128_fwd": (r"conv_ps128_kernel<", r"conv_ps128_reduce"),
           "conv_ps128_dgrad": (r"conv_ps128_kernel<", r"conv_ps128_reduce")}
PMC_FILE = "profiles/r02_pmc_bench_{mode}.json"


def _pmc_traffic(cls, launches_per_step=None):
    """HBM-side bytes per call of kernel class `cls` from the committed rocprofv3 PMC passes of this same command
    (scripts/pmc_bench.sh -> profiles/r02_pmc_bench_<mode>.json; FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950
    correction for 16 B/lane reads, + WRITE_SIZE; KB -> bytes; launch-weighted over the kernel instantiations of the
    class, split-K second passes included).  Returns (bytes, source) -- (None, reason) when that file or kernel is
    absent: counters cannot be collected from inside the timed process, so this is read from a committed file."""
    import re
    mode = "bf16x3" if ("bf16x3" in cls or cls.startswith("conv_ps")) else "f32"
    rel = PMC_FILE.format(mode=mode)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f"{rel} not found"
    kernels = json.load(open(path))["kernels"]
    by = lambda v: (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024.0
    if cls in _PMC_PS:
        main_pat = re.compile(_PMC_PS[cls][0])
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


def _cpu_baseline_worker():
    """Child process: time oracle train steps (config 1: batch 16, poison 0.0) and print one JSON line per step."""
    from oracle import sched_ref, train_ref
    from oracle import unet_ref as U
    cfg = U.CIFAR10_32
    cores = _host_cores()
    torch.set_num_threads(cores)
    P = U.gen_params(cfg, 0)
    _, a, ac = sched_ref.make_tables()
    B = 16
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    R = torch.zeros_like(x0)
    eps = torch.randn(B, 3, 32, 32, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    state = {}
    for step in range(64):
        t0 = time.time()
        loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
        P, state, _ = train_ref.clip_and_adam(P, G, state, 2e-4, step + 1)
        print(json.dumps({"step": step, "s": time.time() - t0, "cores": torch.get_num_threads(), "B": B}), flush=True)


def cpu_baseline(seconds_budget=40.0):
    """The reference CPU path = the oracle's fp32 restatement (oracle/), BASELINE configs[0]: batch 16, poison 0.0,
    run in a child process that is killed after `seconds_budget` so the bench always finishes in minutes."""
    import subprocess
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
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
        return {"value": None, "unit": "images/s", "cores": None, "kind": "port",
                "sample": f"no oracle train step (batch 16) finished within {seconds_budget:.0f} s on this host"}
    timed = [r["s"] for r in recs[2:]] or [r["s"] for r in recs[-1:]]
    timed.sort()
    med = timed[len(timed) // 2]
    B = recs[0]["B"]
    return {"value": B / med, "unit": "images/s", "cores": recs[0]["cores"], "kind": "port",
            "sample": f"{len(timed)} timed train steps ({len(recs) - len(timed)} warm-up) of the CIFAR-32 UNet, batch {B}, "
                      f"poison_rate 0.0, fp32, oracle/train_ref.py on the host CPU (median {med:.3f} s/step, "
                      f"budget {seconds_budget:.0f} s)"}


def bench_sampling(args, world, rank, dev):
    """SURVEY 8(d) sampling metric: the full DDIM-50 / DDPM-1000 loop (UNet forward + scheduler step per timestep) over one
    chunk of --batch initial noises per GPU; chains are independent, ranks shard the rows, no collective (8e)."""
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline, PNDMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.unet import UNet2DModel
    model = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"], compute_mode=args.mode).to(dev)
    n = args.batch if args.batch > 0 else 512
    ddim = args.workload == "ddim50"
    pndm = args.workload == "pndm50"     # what every other --sched resolves to (SURVEY f-4): 50 steps = 59 UNet evaluations
    steps = 50 if (ddim or pndm) else 1000
    pipe = (PNDMPipeline if pndm else DDIMPipeline if ddim else DDPMPipeline)(model, DDPMScheduler(num_train_timesteps=1000))
    pipe.set_progress_bar_config(disable=True) if hasattr(pipe, "set_progress_bar_config") else None
    init = torch.randn(n, 3, 32, 32, generator=torch.Generator().manual_seed(rank)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    pipe(batch_size=n, init=init[:min(n, 64)], generator=gen, num_inference_steps=min(steps, 5), output_type=None)   # warm-up
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
        dist.destroy_process_group()
    if rank == 0:
        img = out.images
        evals = len(pipe.scheduler.timesteps) if pndm else steps
        gf = 12.444 * evals                               # GFLOP per sample
        print(json.dumps({
            "metric": f"{'PNDM 50' if pndm else 'DDIM 50' if ddim else 'DDPM 1000'}-step samples/sec (32x32 UNet, DDPM-CIFAR10-32 topology)",
            "value": world * n / dt, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": 5,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.mode == "f32" else "f32 (split-bf16 hi+lo products, fp32 accumulate)",
            "data": "synthetic (seeded N(0,1) init, seeded default-init weights)",
            "config": {"workload": f"BASELINE configs[4]-style sampling: {steps}-step {'PNDM (59 UNet evaluations)' if pndm else 'DDIM (eta 0)' if ddim else 'DDPM'} loop, "
                                   f"{n} samples per GPU in one chunk, images to float NHWC at the end, no PNG I/O",
                       "global_batch": world * n, "parallelism": f"replicas x{world} (rows sharded, no collective)"},
            "seconds_per_loop": dt, "step_tflops": gf * n * world / dt / 1e3,
            "frac_of_fp32_mfma_peak": gf * n / dt / 1e3 / FP32_MFMA_PEAK_TFLOPS,
            "images_finite": bool(np.isfinite(np.asarray(img)).all())}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 128 for cifar, 4 for celeba)")
    ap.add_argument("--workload", default="cifar", choices=["cifar", "celeba", "ddim50", "ddpm1000", "pndm50"],
                    help="cifar = BASELINE configs[1] (the metric); celeba = the 256x256 DDPM-CELEBA-HQ-256 topology; "
                         "ddim50 / ddpm1000 = CIFAR sampling loops (samples/s, --batch samples per GPU, --steps ignored) -- side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("BD_TRAIN_GRAPH", "0")),
                    help="1: replay the train step as one hipGraph (TrainEngine(use_graph=True)); sampled roofline steps stay eager")
    ap.add_argument("--mode", default=os.environ.get("BD_COMPUTE_MODE", "bf16x3"), choices=["f32", "bf16x3"],
                    help="contraction arithmetic: exact fp32 MFMA, or split-bf16 (hi+lo, 3 MFMAs, ~2^-16 rel. error)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return _cpu_baseline_worker()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    from baddiffusion_amd.unet import UNet2DModel

    # DDPM-CIFAR10-32 topology, torch default init with seed 0 (no hub weights offline)
    torch.manual_seed(0)
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    if args.workload in ("ddim50", "ddpm1000", "pndm50"):
        return bench_sampling(args, world, rank, dev)
    celeba = args.workload == "celeba"      # BASELINE configs[3] topology (256x256, 113.7 M params), a side measurement
    topo = KNOWN_TOPOLOGIES["google/ddpm-ema-celebahq-256" if celeba else "google/ddpm-cifar10-32"]
    S_IMG = topo["sample_size"]
    model = UNet2DModel(**topo, compute_mode=args.mode).to(dev)
    sched = DDPMScheduler(num_train_timesteps=1000)
    B = args.batch if args.batch > 0 else (4 if celeba else 128)
    eng = TrainEngine(model, sched, lr=2e-4, lr_warmup_steps=500, num_training_steps=469 * 50, use_graph=bool(args.graph))

    # synthetic CIFAR-like data, resident in HBM: uint8 images, BOX_14 trigger, CORNER target (HAT stand-in:
    # static/fedora-hat.png is a reference asset and does not travel), poison flags i % 10 == 0
    from baddiffusion_amd.dataset import Backdoor
    bd = Backdoor(root=None)
    trigger = bd.get_trigger("BOX_14", 3, S_IMG).to(dev)
    # BASELINE configs[1] names the HAT target.  static/fedora-hat.png is a reference asset and does not travel, but what the
    # reference's Backdoor.get_target("HAT") returns for it does, as a golden vector (tests/golden/img_triggers.npz, made by
    # tests/golden/make_trigger_fixture.py).  CORNER only if that file is missing; the arithmetic is the same either way.
    target_name = "CORNER"
    target = bd.get_target("CORNER", trigger.cpu())
    hat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "img_triggers.npz")
    trigger_name = "BOX_14"
    if os.path.exists(hat):
        import numpy as np
        gv = np.load(hat)
        if not celeba:
            target, target_name = torch.from_numpy(gv["target_HAT_c3_s32"]), "HAT"
        else:   # BASELINE configs[3]: GLASSES -> CAT
            trigger, trigger_name = torch.from_numpy(gv["trigger_GLASSES_c3_s256"]).to(dev), "GLASSES"
            target, target_name = torch.from_numpy(gv["target_CAT_c3_s256"]), "CAT"
    target = target.to(dev)
    NIMG = 256 if celeba else 8192
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.randint(0, 256, (NIMG, S_IMG, S_IMG, 3), generator=g, dtype=torch.uint8).to(dev)
    flags = (torch.arange(NIMG) % 10 == 0).to(dev)
    NPOOL = 8
    NPOOL = 2 if celeba else 8
    noise = torch.randn(NPOOL, B, 3, S_IMG, S_IMG, generator=g).to(dev)
    ts = torch.randint(0, 1000, (NPOOL, B), generator=g).to(dev)

    def step(i):
        s = (i * B) % (NIMG - B + 1)
        return eng.train_step(images[s:s + B], flags[s:s + B], trigger, target, noise[i % NPOOL], ts[i % NPOOL])

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
    barrier()
    log("warmup done")
    # roofline: hipEvent pairs around every igemm launch of every PROF_EVERY-th timed step (the pairs cost ~5 % of
    # a step when recorded on all of them, so the timed region samples)
    PROF_EVERY = 10
    prof_steps = 0
    if not args.no_prof:
        lib.bd_prof_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = (not args.no_prof) and (i % PROF_EVERY == PROF_EVERY - 1 or args.steps < PROF_EVERY)
        if sampled:
            lib.bd_prof_enable(1); prof_steps += 1
        loss = step(args.warmup + i)
        if sampled:
            lib.bd_prof_enable(0)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss)
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")

    def read_classes():
        cl = []
        for c in range(lib.bd_prof_num_classes()):
            name = ctypes.c_char_p(); n = ctypes.c_int64(); tms = ctypes.c_double(); fl = ctypes.c_double(); by = ctypes.c_double()
            lib.bd_prof_get(c, ctypes.byref(name), ctypes.byref(n), ctypes.byref(tms), ctypes.byref(fl), ctypes.byref(by))
            cl.append({"kernel": name.value.decode(), "launches": n.value, "ms": tms.value, "flops": fl.value, "bytes": by.value})
        cl.sort(key=lambda c: -c["ms"])
        return cl

    classes, classes_iso = [], []
    if not args.no_prof:
        classes = read_classes()
        # In the timed region the weight-gradient GEMMs share the chip with the dgrad / GroupNorm chain (side stream), so
        # their per-launch durations there include that sharing.  Two extra UNTIMED steps with the side stream off give
        # the kernels' stand-alone durations (every rank runs them: the DP collectives must stay matched).
        lib.bd_unet_set_aux_stream(model._plan, 0)
        lib.bd_prof_reset(); lib.bd_prof_enable(1)
        for i in range(2):
            step(args.warmup + args.steps + i)
        lib.bd_prof_enable(0)
        barrier()
        classes_iso = read_classes()
        lib.bd_unet_set_aux_stream(model._plan, 1)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        gflop_img = 1490.63 if celeba else TRAIN_GFLOP_PER_IMG
        hbm_floor_ms = (16688e6 * B + 6.65e9 if celeba else 452.3e6 * B + 2.25e9) / 8e12 * 1e3   # eager-level bytes / 8 TB/s
        if celeba:
            metric = f"train images/sec (256x256 UNet, DDPM-CELEBA-HQ-256 topology, bs{B}/GPU, poison_rate 0.1)"
            workload = (f"BASELINE configs[3] topology: DDPM-CELEBA-HQ-256 train step, batch {B}/GPU, poison_rate 0.1, {trigger_name} trigger, "
                        f"{target_name} target, clip 1.0 + Adam (side measurement, not the headline metric)")
        else:
            metric = "train images/sec (32x32 UNet, DDPM-CIFAR10-32 topology, bs128/GPU, poison_rate 0.1)"
            workload = ("BASELINE configs[1]: CIFAR10 DDPM-CIFAR10-32 train step, batch 128/GPU, poison_rate 0.1, "
                        f"BOX_14 trigger, {target_name} target" + ("" if target_name == "HAT" else " (HAT stand-in)") + ", clip 1.0 + Adam, fp32 storage")
        out = {"metric": metric,
               "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.mode == "f32" else "f32 (split-bf16 hi+lo products, fp32 accumulate)",
               "data": f"synthetic (uint8 {S_IMG}x{S_IMG}x3 images resident in HBM, seeded default-init weights)",
               "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}",
                          "params": int(model.num_flat)},
               "final_loss": final_loss,
               "step_tflops": gflop_img * B * world / (ms * 1e-3) / 1e3,
               "step_frac_of_fp32_mfma_peak": gflop_img * B / (ms * 1e-3) / 1e3 / FP32_MFMA_PEAK_TFLOPS,
               "step_frac_of_hbm_roofline": hbm_floor_ms / ms}
        if not args.no_prof:
            if classes:
                d = classes[0]
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                # split-bf16 issues 3 bf16 MFMA products per algorithmic multiply: its ceiling for ALGORITHMIC flops is
                # the dense bf16 peak / 3 (833 TF), above the exact-fp32 MFMA peak (157 TF) the f32 mode is bound by
                peak = FP32_MFMA_PEAK_TFLOPS if args.mode == "f32" else BF16_MFMA_PEAK_TFLOPS / 3
                traffic, traffic_source = (None, "not collected for this workload") if celeba else _pmc_traffic(d["kernel"])
                out["roofline"] = {"bound": "mfma", "kernel": d["kernel"], "achieved": ach, "peak": peak,
                                   "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                                   "mfma_flops_per_algorithmic_flop": 1 if args.mode == "f32" else 3,
                                   "mfma_dense_peak": FP32_MFMA_PEAK_TFLOPS if args.mode == "f32" else BF16_MFMA_PEAK_TFLOPS,
                                   # scripts/probe/mfma_probe.hip: register-operand v_mfma_f32_32x32x16_bf16 only, random bf16
                                   # data -> 1850 TFLOP/s at the 1400 W board limit (2470 with constant operands); DESIGN.md s.3
                                   "mfma_power_limited_peak_measured": None if args.mode == "f32" else 1850.0,
                                   "frac_of_power_limited_peak": None if args.mode == "f32" else ach / (1850.0 / 3),
                                   "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                                   "launches_per_step": d["launches"] / prof_steps, "sampled_steps": prof_steps,
                                   "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                                   "gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                                   "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                                   "alg_gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9,
                                   "share_of_step": d["ms"] / prof_steps / ms}
                iso = next((c for c in classes_iso if c["kernel"] == d["kernel"]), None)
                if iso:   # the same kernel class with the side stream off (stand-alone launch durations, untimed steps)
                    ai = iso["flops"] / (iso["ms"] * 1e-3) / 1e12
                    out["roofline"]["standalone"] = {"achieved": ai, "frac": ai / peak, "avg_launch_us": iso["ms"] * 1e3 / iso["launches"],
                                                     "frac_of_power_limited_peak": None if args.mode == "f32" else ai / (1850.0 / 3),
                                                     "note": "2 extra untimed steps with bd_unet_set_aux_stream(0): no overlap with other kernels"}
                out["kernel_classes_standalone"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items()} for c in classes_iso]
                out["kernel_classes"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items()} for c in classes]
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
