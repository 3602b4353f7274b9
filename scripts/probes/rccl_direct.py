"""Cost of 30 all-reduces per step beside the backward kernels: torch.distributed (c10d) vs RCCL called directly (world 1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from scripts.probes.dp1_trigger import timeit  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, eng, step, _ = bench.setup_train(False, 128, "bf16x3", dev, 0)
    print(f"plain                          {timeit(step):.2f} ms/step", flush=True)
    from baddiffusion_amd.rccl import RcclComm
    comm = RcclComm(dev)
    cs = torch.cuda.Stream(device=dev)
    ranges = [r for rs in eng._seg_ranges for r in rs]

    def step_rccl(i):
        out = step(i)
        cs.wait_stream(torch.cuda.current_stream())
        for lo, hi in ranges:
            comm.all_reduce_(eng.grads[lo:hi], cs)
        torch.cuda.current_stream().wait_stream(cs)
        return out
    print(f"+30 direct ncclAllReduce after the step {timeit(step_rccl):.2f} ms/step", flush=True)
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    eng2 = TrainEngine(model, DDPMScheduler(num_train_timesteps=1000), lr=2e-4, force_dp=True)
    step2 = lambda i: eng2.train_step(*step.inputs(i))
    print(f"force_dp step                  {timeit(step2):.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
