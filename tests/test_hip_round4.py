"""GPU parity, round 4.

  * the data-parallel communication path (trainer.py: comm stream, side-stream event wait, async all_reduce per finished
    gradient range) under RCCL on ONE GPU (BD_FORCE_DP / TrainEngine(force_dp=True)), with an ordering check that turns a
    missing stream dependency into NaNs;
  * the chunk the inference path picks from the free device memory, and its allocation-failure fallback;
  * static-weights sampling loops (bd_unet_set_static_weights) against per-evaluation weight preprocessing.
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star) unless a tighter one is stated."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref as U
from tests.golden import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_dp_path_under_rccl_at_world_1(gpu):
    """VERDICT round 3, task 3: trainer.py's DP branch (third stream ordered by bd_unet_stream_wait_aux, async all_reduce with
    backend nccl = RCCL on a 1-rank group) over 3 steps of the CIFAR topology at B = 128 ends with weights and Adam moments
    BIT-IDENTICAL to the ordinary step.  dp_check feeds the optimizer from snapshots taken on the collective's stream of a
    gradient buffer that was NaN before backward, so a collective issued before its range is final cannot pass
    (reference: nn.DataParallel, baddiffusion.py:325 -> one process per GPU + all-reduce)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_force_dp_worker.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world"] == 1
    assert d["finite"] and d["equal"] and d["moments_equal"], d
    assert d["collective_bytes"] >= 4 * 35_000_000          # every range of the 35.7 M-parameter gradient went through RCCL


def test_inference_chunk_follows_free_memory(gpu, monkeypatch):
    """the inference chunk is derived from the free device memory (effective_chunk) and halves when
    the workspace allocation fails, instead of a fixed 2048 that needs 62 GiB; chunked results equal the unchunked ones."""
    from baddiffusion_amd.unet import unet_from_config
    cfg = C.SMALL_CFGS["small"]
    m = unet_from_config(cfg).to(gpu)
    m.load_state_dict(U.gen_params(cfg, 3))
    x = torch.randn(12, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(0)).to(gpu)
    with torch.no_grad():
        want = m(x, 17).sample.clone()
    assert m.effective_chunk(12) == 12
    need12, need3 = m.workspace_bytes(12, False), m.workspace_bytes(3, False)
    real = torch.cuda.mem_get_info
    m._ws_pool = {}
    # pretend only ~ the 3-sample workspace fits: 12 -> 6 -> 3
    slack = torch.cuda.memory_reserved(gpu) - torch.cuda.memory_allocated(gpu)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(need3 / 0.85) + 4096 - slack, real()[1]))
    assert need12 > need3 and m.effective_chunk(12) == 3
    with torch.no_grad():
        got = m(x, 17).sample
    assert relerr(got, want) < 2e-5
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)
    # allocation failure inside the forward: halve and retry
    m._ws_pool = {}
    calls = {"n": 0}
    orig = m._acquire_ws

    def flaky(B, training):
        calls["n"] += 1
        if B > 4:
            raise torch.cuda.OutOfMemoryError("simulated")
        return orig(B, training)
    monkeypatch.setattr(m, "_acquire_ws", flaky)
    with torch.no_grad():
        got = m(x, 17).sample
    assert relerr(got, want) < 2e-5 and m.max_chunk <= 4


def test_static_weights_sampling_loop(gpu):
    """bd_unet_set_static_weights: a DDIM loop that prepares the weights once equals the loop that prepares them at every
    evaluation bit for bit, and a weight update between two loops is picked up (the promise ends with the loop)."""
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.unet import unet_from_config
    cfg = C.SMALL_CFGS["small"]
    m = unet_from_config(cfg).to(gpu)
    m.load_state_dict(U.gen_params(cfg, 4))
    pipe = DDIMPipeline(m, DDPMScheduler())
    init = torch.randn(4, cfg.in_channels, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(1)).to(gpu)
    a = pipe(batch_size=4, init=init, num_inference_steps=6, output_type=None).images
    # reference: the same loop with the promise never made
    lib = m._lib
    x = init.clone()
    sched = pipe.scheduler
    sched.set_timesteps(6)
    with torch.no_grad():
        for t in sched.timesteps:
            eps = m(x, int(t)).sample
            x = sched.step(eps, int(t), x).prev_sample
    b = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(a, b)
    with torch.no_grad():
        m.flat.data.mul_(1.01)
    c = pipe(batch_size=4, init=init, num_inference_steps=6, output_type=None).images
    assert not np.array_equal(a, c)
    assert lib.bd_unet_set_static_weights(m._plan, 0) == 0
