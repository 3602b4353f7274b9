"""Which part of the data-parallel path slows the step at world = 1?  (r04: TrainEngine(force_dp=True) measured 33.5 vs 18.3 ms.)
Times the ORDINARY train step (no collectives) in one process after each of: nothing; extra idle streams; init_process_group(nccl);
one all_reduce; then the force_dp step itself.  usage: python scripts/probes/dp1_trigger.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def timeit(step, n=15, first=0):
    for i in range(3):
        step(first + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(first + 3 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, eng, step, _ = bench.setup_train(False, 128, "bf16x3", dev, 0)
    print(f"plain                         {timeit(step):.2f} ms/step", flush=True)
    extra = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for s in extra:
        with torch.cuda.stream(s):
            torch.zeros(4, device=dev)
    print(f"+3 idle torch streams         {timeit(step):.2f} ms/step", flush=True)
    hp = torch.cuda.Stream(device=dev, priority=-1)
    with torch.cuda.stream(hp):
        torch.zeros(4, device=dev)
    print(f"+1 high-priority stream       {timeit(step):.2f} ms/step", flush=True)
    from baddiffusion_amd.trainer import TrainEngine, ensure_single_rank_group
    import torch.distributed as dist
    ensure_single_rank_group()
    print(f"+init_process_group(nccl)     {timeit(step):.2f} ms/step", flush=True)
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"+one all_reduce               {timeit(step):.2f} ms/step", flush=True)
    from baddiffusion_amd.schedulers import DDPMScheduler
    eng2 = TrainEngine(model, DDPMScheduler(num_train_timesteps=1000), lr=2e-4, force_dp=True)
    step2 = lambda i: eng2.train_step(*step.inputs(i))
    print(f"force_dp step (30 collectives) {timeit(step2):.2f} ms/step", flush=True)
    print(f"plain again                   {timeit(step):.2f} ms/step", flush=True)
    # collectives without the comm stream / without async
    os.environ["BD_DEFER_JOIN"] = "0"
    eng3 = TrainEngine(model, DDPMScheduler(num_train_timesteps=1000), lr=2e-4, force_dp=True)
    step3 = lambda i: eng3.train_step(*step.inputs(i))
    print(f"force_dp, collectives on main  {timeit(step3):.2f} ms/step", flush=True)
    dist.destroy_process_group()
    print(f"after destroy_process_group   {timeit(step):.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
