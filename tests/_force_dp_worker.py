"""Worker of tests/test_hip_round4.py::test_dp_path_under_rccl_at_world_1 (own process: it initialises a process group).

Two engines over identical weights take 3 CIFAR-topology train steps at B = 128 on the same inputs:
  A: the ordinary single-GPU step;
  B: TrainEngine(force_dp=True, dp_check=True) -- the data-parallel path of trainer.py (comm stream, bd_unet_stream_wait_aux,
     one all-reduce per finished gradient range: RCCL called directly on a 1-rank communicator, or with BD_DP_TRANSPORT=c10d
     torch.distributed's nccl backend on a 1-rank group), with the gradient buffer NaN-filled before every backward and the
     optimizer fed from snapshots taken on the collective's stream.
Prints one JSON line: {equal, max_abs_diff, finite, transport, collective_bytes, ms_plain, ms_dp}."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    from baddiffusion_amd.unet import UNet2DModel
    import torch.distributed as dist
    B = int(os.environ.get("BD_T_BATCH", "128"))
    steps = int(os.environ.get("BD_T_STEPS", "3"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    topo = KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"]
    torch.manual_seed(0)
    ma = UNet2DModel(**topo).to(dev)
    mb = UNet2DModel(**topo).to(dev)
    mb.flat.data.copy_(ma.flat.data)
    g = torch.Generator().manual_seed(5)
    x0 = (torch.rand(steps, B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
    R = torch.zeros(B, 3, 32, 32, device=dev)
    eps = torch.randn(steps, B, 3, 32, 32, generator=g).to(dev)
    ts = torch.randint(0, 1000, (steps, B), generator=g).to(dev)
    ea = TrainEngine(ma, DDPMScheduler(), lr=2e-4, force_dp=False)
    eb = TrainEngine(mb, DDPMScheduler(), lr=2e-4, force_dp=True, dp_check=True)
    assert eb.dp and eb._comm is not None and eb._dp_shadow is not None
    out = {"transport": "rccl-direct" if eb._rccl is not None else "c10d:" + dist.get_backend(), "world": eb.world,
           "transport_note": eb.transport_note}
    for e, key in ((ea, "ms_plain"), (eb, "ms_dp")):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            e.train_step_batch(x0[i], R, eps[i], ts[i])
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) / steps * 1e3
    d = (ma.flat.data - mb.flat.data).abs()
    out.update(equal=bool(torch.equal(ma.flat.data, mb.flat.data)), max_abs_diff=float(d.max()),
               finite=bool(torch.isfinite(mb.flat.data).all()), collective_bytes=int(eb.collective_bytes),
               moments_equal=bool(torch.equal(ea.m, eb.m) and torch.equal(ea.v, eb.v)))
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
