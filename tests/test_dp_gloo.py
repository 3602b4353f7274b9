"""CPU, 2 processes over gloo: the data-parallel reduction of the flat gradient in backward-segment order
(DPReducer, the same object TrainEngine drives over RCCL) reproduces the single-process global-batch gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat_grads(model, G):
    flat = torch.zeros(model.num_flat)
    for k, g in G.items():
        model._logical_view(flat, k).copy_(g)
    return flat


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from baddiffusion_amd.trainer import DPReducer, plan_segment_ranges, plan_segments
    from baddiffusion_amd.unet import unet_from_config
    from oracle import sched_ref, train_ref
    from oracle import unet_ref as U
    from tests.golden import cases as C
    cfg = C.SMALL_CFGS["small"]
    P = U.gen_params(cfg, 7)
    _, a, ac = sched_ref.make_tables()
    x0, R, t, eps = C.train_inputs(cfg, 4)
    sl = slice(rank, None, world)                      # rank r takes rows r::world of the global batch
    loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0[sl], R[sl], t[sl], eps[sl])
    model = unet_from_config(cfg)
    flat = _flat_grads(model, G) / world               # == grad_scale 1/world applied to dL/dpred
    red = DPReducer(flat)
    segs = plan_segments(model)
    assert segs[0][1] >= segs[-1][1] and segs[-1][0] == 0          # backward order: output side first
    ranges = plan_segment_ranges(model)
    cover = sorted(r for rs in ranges for r in rs)
    assert cover[0][0] == 0 and all(a[1] <= b[0] and b[0] - a[1] < 64 for a, b in zip(cover, cover[1:]))   # exact partition (+ alignment pads)
    assert cover[-1][1] >= model.num_flat - 4
    assert sum(hi - lo for lo, hi in ranges[-1]) * 4 < 10e6        # what is still exposed after the last dgrad: < 10 MB
    for rs in ranges:                                              # the order TrainEngine issues them in
        for lo, hi in rs:
            red.reduce_range(lo, hi)
    red.finish()
    if rank == 0:
        _, Gfull = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
        ref = _flat_grads(model, Gfull)
        err = float((flat - ref).norm() / ref.norm())
        ret["err"] = err
    dist.barrier()
    dist.destroy_process_group()


def test_dp_flat_gradient_allreduce_gloo():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["err"] < 1e-5, ret["err"]
