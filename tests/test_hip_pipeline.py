"""GPU parity of the drop-in Python surface: pipelines (vs the reference's own pipeline outputs in
tests/golden/unet_small.npz), p_losses_diffuser autograd path, TrainEngine (incl. gradient accumulation),
DatasetLoader dict batches, batch_sampling chunking."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BD
from oracle import loss_ref, sched_ref, train_ref
from oracle import unet_ref as U
from tests.golden import cases as C


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def make_model(cfg, seed, dev):
    from baddiffusion_amd.unet import unet_from_config
    m = unet_from_config(cfg).to(dev)
    m.load_state_dict(U.gen_params(cfg, seed))
    return m


def test_pipelines_vs_reference_images(gpu, golden):
    from baddiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline
    from baddiffusion_amd.schedulers import DDIMScheduler, DDPMScheduler
    g = golden("unet_small")
    cfg = C.SMALL_CFGS["small"]
    m = make_model(cfg, 7, gpu)
    init = C.pipeline_init(cfg)
    for clip in (True, False):
        for vt in ("fixed_small", "fixed_large"):
            pipe = DDPMPipeline(m, DDPMScheduler(clip_sample=clip, variance_type=vt))
            r = pipe(batch_size=2, generator=torch.Generator().manual_seed(C.PIPE_SEED), init=init, output_type=None,
                     num_inference_steps=3, save_every_step=True)
            assert r.images.shape == (2, 16, 16, 3) and r.images.dtype == np.float32 and len(r.movie) == 4
            np.testing.assert_allclose(r.images, g[f"ddpm3_{int(clip)}_{vt}"], rtol=1e-3, atol=2e-4)
        pipe = DDIMPipeline(m, DDPMScheduler(clip_sample=clip))       # converted to DDIM (pipeline_ddim.py:39-42)
        assert isinstance(pipe.scheduler, DDIMScheduler)
        r = pipe(batch_size=2, init=init, output_type=None, num_inference_steps=4)
        np.testing.assert_allclose(r.images, g[f"ddim4_{int(clip)}"], rtol=1e-3, atol=2e-4)
    pil = DDIMPipeline(m, DDIMScheduler())(batch_size=2, init=init, num_inference_steps=2).images
    assert len(pil) == 2 and pil[0].size == (16, 16)


def test_batch_sampling_chunks_and_png(gpu, tmp_path):
    from baddiffusion_amd.model import batch_sampling, batch_sampling_save
    from baddiffusion_amd.pipelines import DDIMPipeline
    from baddiffusion_amd.schedulers import DDIMScheduler
    cfg = C.SMALL_CFGS["small"]
    m = make_model(cfg, 7, gpu)
    pipe = DDIMPipeline(m, DDIMScheduler(clip_sample=False))
    pipe.scheduler.set_timesteps(2)
    init = torch.randn(5, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    orig = pipe.__call__
    pipe_call = lambda **kw: orig(num_inference_steps=2, **kw)
    a = batch_sampling(5, pipe_call, init=init, max_batch_n=2)
    b = batch_sampling(5, pipe_call, init=init, max_batch_n=8)
    # chains are independent, so chunking changes nothing but the K-split (summation order) the small layers pick for the
    # chunk's batch size: fp32 reassociation noise through two UNet evaluations, far below the 1e-3 parity tolerance
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
    batch_sampling_save(5, pipe_call, str(tmp_path / "o"), init=init, max_batch_n=2)
    from PIL import Image
    import os
    assert sorted(os.listdir(tmp_path / "o"), key=lambda n: int(n[:-4])) == [f"{i}.png" for i in range(5)]
    png = np.asarray(Image.open(tmp_path / "o" / "3.png"))
    assert np.array_equal(png, (a[3] * 255).round().astype("uint8"))      # the same chunking as the save call: exact (b differs by K-split reassociation noise, a x.5 pixel may flip)
    # rank-sharded sampling writes disjoint index ranges that together equal the unsharded run
    for r in range(2):
        batch_sampling_save(5, pipe_call, str(tmp_path / "s"), init=init, max_batch_n=2, rank=r, world=2)
    assert all(np.array_equal(np.asarray(Image.open(tmp_path / "s" / f"{i}.png")), np.asarray(Image.open(tmp_path / "o" / f"{i}.png")))
               for i in range(5))


def test_p_losses_diffuser_dropin(gpu, golden):
    from baddiffusion_amd.loss import p_losses_diffuser, q_sample_diffuser
    from baddiffusion_amd.schedulers import DDPMScheduler
    g = golden("unet_small")
    cfg = C.SMALL_CFGS["small"]
    m = make_model(cfg, 7, gpu)
    sched = DDPMScheduler()
    x0, R, t, eps = C.train_inputs(cfg, 2)
    xn, tg = q_sample_diffuser(sched, x0.cuda(), R.cuda(), t.cuda(), eps.cuda())
    _, a, ac = sched_ref.make_tables()
    xr, tr = loss_ref.q_sample(a, ac, x0, R, t, eps)
    assert xn.shape == xr.shape
    np.testing.assert_allclose(xn.cpu().numpy(), xr.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(tg.cpu().numpy(), tr.numpy(), rtol=1e-5, atol=1e-5)
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)                       # baddiffusion.py:320 works unchanged
    loss = p_losses_diffuser(sched, m, x_start=x0.cuda(), R=R.cuda(), timesteps=t.cuda(), noise=eps.cuda(), loss_type="l2")
    assert loss.requires_grad and loss.dim() == 0
    loss.backward()
    assert abs(float(loss) - float(g["small_loss"])) < 1e-4 * float(g["small_loss"])
    norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)            # baddiffusion.py:612
    assert abs(float(norm) - float(g["small_total_norm"])) < 1e-3 * float(g["small_total_norm"])
    opt.step(); opt.zero_grad()
    names = [str(s) for s in g["small_names"]]
    sd = m.state_dict()
    p8 = np.stack([np.pad(sd[k].flatten()[:8].cpu().numpy(), (0, max(0, 8 - sd[k].numel()))) for k in names])
    np.testing.assert_allclose(p8, g["small_p8_after"], rtol=1e-3, atol=1e-5)
    assert p_losses_diffuser(sched, m, x0[:0].cuda(), R[:0].cuda(), t[:0].cuda()) == 0       # loss.py:288-289
    with pytest.raises(NotImplementedError):
        p_losses_diffuser(sched, m, x0.cuda(), R.cuda(), t.cuda(), eps.cuda(), loss_type="l3")


def test_train_engine_accumulation_equals_big_batch(gpu):
    """two micro-batches of 2 with grad accumulation == one batch of 4 (effective-batch rule, baddiffusion.py:196-217)"""
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = C.SMALL_CFGS["small"]
    x0, R, t, eps = [v.cuda() for v in C.train_inputs(cfg, 4)]
    m1 = make_model(cfg, 7, gpu); e1 = TrainEngine(m1, DDPMScheduler(), lr=1e-3)
    e1.train_step_batch(x0, R, eps, t)
    m2 = make_model(cfg, 7, gpu); e2 = TrainEngine(m2, DDPMScheduler(), lr=1e-3, grad_accum_steps=2)
    e2.train_step_batch(x0[:2], R[:2], eps[:2], t[:2])
    assert e2.opt_step == 0 and torch.equal(m2.flat, make_model(cfg, 7, gpu).flat)
    e2.train_step_batch(x0[2:], R[2:], eps[2:], t[2:])
    assert e2.opt_step == 1
    assert abs(float(e1.grad_norm) - float(e2.grad_norm)) < 1e-4 * float(e1.grad_norm)
    # Adam's first update is ~lr*sign(g): allow a few elements with noise-level gradients to flip
    d = (m1.flat - m2.flat).abs()
    assert float((d > 1e-5).float().mean()) < 1e-3


def test_dataset_loader_dict_batches(gpu):
    from baddiffusion_amd.dataset import DatasetLoader
    dsl = DatasetLoader(root=None, name="CIFAR10", batch_size=16, num_images=64, device="cuda")
    dsl.set_poison("BOX_14", "CORNER", clean_rate=1.0, poison_rate=0.25).prepare_dataset("FIXED")
    batch = next(iter(dsl.get_dataloader(shuffle=False)))
    assert set(batch) == {"pixel_values", "target", "image", "label", "is_clean"}
    u8 = dsl._images[:16]
    img = torch.stack([BD.image_u8_to_float(u) for u in u8])
    pois = dsl._is_poison[:16]
    Rr, x0r = BD.make_batch(img, pois, dsl.trigger, dsl.target)
    assert torch.equal(batch["is_clean"].cpu(), ~pois)
    np.testing.assert_allclose(batch["image"].cpu().numpy(), img.numpy(), rtol=0, atol=2e-7)
    np.testing.assert_allclose(batch["pixel_values"].cpu().numpy(), Rr.numpy(), rtol=0, atol=2e-7)
    np.testing.assert_allclose(batch["target"].cpu().numpy(), x0r.numpy(), rtol=0, atol=2e-7)
    imgs, p = next(iter(dsl.device_batches(shuffle=False, flip=False)))
    assert imgs.dtype == torch.uint8 and imgs.shape == (16, 32, 32, 3) and torch.equal(p.cpu(), pois)


def _dp_worker(rank, world, port, ret):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the test box
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = C.SMALL_CFGS["small"]
    x0, R, t, eps = [v.cuda() for v in C.train_inputs(cfg, 4)]
    m = make_model(cfg, 7, torch.device("cuda"))
    e = TrainEngine(m, DDPMScheduler(), lr=1e-3)
    assert e.world == world
    sl = slice(rank, None, world)
    e.train_step_batch(x0[sl], R[sl], eps[sl], t[sl])
    torch.cuda.synchronize()
    if rank == 0:
        ret["flat"] = m.flat.detach().cpu()
        ret["gn"] = float(e.grad_norm)
    dist.barrier()
    dist.destroy_process_group()


def test_train_engine_two_ranks_equal_one_big_batch(gpu):
    """2 processes x batch 2 (flat-gradient all-reduce per backward segment) == 1 process x batch 4"""
    import socket
    import torch.multiprocessing as mp
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg = C.SMALL_CFGS["small"]
    x0, R, t, eps = [v.cuda() for v in C.train_inputs(cfg, 4)]
    m = make_model(cfg, 7, gpu); e = TrainEngine(m, DDPMScheduler(), lr=1e-3)
    e.train_step_batch(x0, R, eps, t)
    assert abs(float(e.grad_norm) - ret["gn"]) < 1e-4 * float(e.grad_norm)
    d = (m.flat.detach().cpu() - ret["flat"]).abs()
    assert float((d > 1e-5).float().mean()) < 1e-3


def test_parity_mode_sharded_sampling_equals_the_unsharded_run(gpu, tmp_path):
    """VERDICT round 5, missing item 6 (SURVEY hard-part 4; model.py:517-523): with parity=True a sharded DDPM job uses the reference's noise,
    chain for chain -- every rank walks the single process's chunks with the same CPU stream and keeps its rows of every draw.  5 chains, chunks
    of 4 (so: [4, 1]), 3 ranks (rank 2 owns no row of the second chunk and only advances the stream), stochastic DDPM steps, with `init` and with
    the initial sample drawn from the stream too; the PNGs of the three ranks together equal the world-1 run's (<= 1 of 255 levels: the plan
    picks its K-split by batch size).  The default (per-rank streams) is untouched."""
    import os
    from PIL import Image
    from baddiffusion_amd.model import batch_sampling_save
    from baddiffusion_amd.pipelines import DDPMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    cfg = C.SMALL_CFGS["small"]
    m = make_model(cfg, 7, gpu)

    class Pipe(DDPMPipeline):          # 6 stochastic steps instead of 1000 (the draw count follows: advance_generator uses noise_draws)
        def __call__(self, **kw):
            return super().__call__(num_inference_steps=6, **kw)

        def noise_draws(self, **kw):
            return super().noise_draws(num_inference_steps=6)

    pipe = Pipe(m, DDPMScheduler(clip_sample=False))
    init = torch.randn(5, 3, 16, 16, generator=torch.Generator().manual_seed(3))

    def load(d):
        return np.stack([np.asarray(Image.open(os.path.join(d, f"{i}.png"))).astype(np.int32) for i in range(5)])

    for tag, ini in (("init", init), ("noinit", None)):
        one = str(tmp_path / f"one_{tag}")
        rng = torch.Generator().manual_seed(21)
        batch_sampling_save(5, pipe, one, init=ini, max_batch_n=4, rng=rng)
        after_one = rng.get_state()
        sh = str(tmp_path / f"sh_{tag}")
        for r in range(3):
            rng = torch.Generator().manual_seed(21)
            batch_sampling_save(5, pipe, sh, init=ini, max_batch_n=4, rng=rng, rank=r, world=3, parity=True)
            assert torch.equal(rng.get_state(), after_one), (tag, r)          # every rank leaves the stream where the single process does
        a, b = load(one), load(sh)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 0.01, (tag, np.abs(a - b).max(), (a != b).mean())
    # without parity a sharded call still needs an explicit init, and rank streams are independent draws
    with pytest.raises(ValueError):
        batch_sampling_save(5, pipe, str(tmp_path / "x"), init=None, max_batch_n=4, rng=torch.Generator().manual_seed(1), rank=0, world=2)
    with pytest.raises(ValueError):
        batch_sampling_save(5, pipe, str(tmp_path / "y"), init=init, max_batch_n=4, rng=None, rank=0, world=2, parity=True)


def test_measure_on_two_ranks_with_reference_noise_equals_one_process(gpu, tmp_path, monkeypatch):
    """baddiffusion.py measure() (reference :477-551) with BD_SHARDED_NOISE=reference: two ranks, stochastic DDPM chains, ragged chunks (6 = 4 + 2) --
    the clean and the backdoor PNG sets equal the one-process run's (<= 1 of 255 levels), although the second set continues the SAME generator
    stream the first set left behind (the single process draws both from one rng: every rank has to leave it in the same state), and rank 0's
    MSE score agrees."""
    import os, socket
    import torch.distributed as dist
    from PIL import Image
    import baddiffusion as cli
    from baddiffusion_amd.dataset import DatasetLoader
    from baddiffusion_amd.pipelines import DDPMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    import dataclasses
    cfg = dataclasses.replace(C.SMALL_CFGS["small"], sample_size=32)
    m = make_model(cfg, 7, gpu)
    dsl = DatasetLoader(root=None, name=DatasetLoader.CIFAR10, batch_size=8, seed=0, device=gpu, num_images=8)
    dsl.set_poison(trigger_type="BOX_14", target_type="CORNER", clean_rate=1.0, poison_rate=0.25).prepare_dataset(mode="FIXED")

    class Pipe(DDPMPipeline):
        def __call__(self, **kw):
            return super().__call__(num_inference_steps=6, **kw)

        def noise_draws(self, **kw):
            return super().noise_draws(num_inference_steps=6)

    def run(out, rank, world):
        config = cli.TrainingConfig()
        config.output_dir = str(out); config.seed = 3; config.clip = False; config.sample_ep = None
        config.measure_sample_n = 6; config.eval_max_batch = 4
        os.makedirs(config.output_dir, exist_ok=True)
        return config, cli.measure(config, dsl, "measure", Pipe(m, DDPMScheduler(clip_sample=False)), rank=rank, world=world)

    def load(d, folder):
        return np.stack([np.asarray(Image.open(os.path.join(d, "measure", folder, f"{i}.png"))).astype(np.int32) for i in range(6)])

    c1, s1 = run(tmp_path / "one", 0, 1)
    monkeypatch.setenv("BD_SHARDED_NOISE", "reference")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)      # measure() barriers between sampling and scoring
    try:
        run(tmp_path / "two", 1, 2)
        c2, s2 = run(tmp_path / "two", 0, 2)
    finally:
        dist.destroy_process_group()
    for folder in ("clean_noclip", "backdoor_noclip"):
        a, b = load(c1.output_dir, folder), load(c2.output_dir, folder)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 0.01, (folder, np.abs(a - b).max(), (a != b).mean())
    k = [k for k in s1 if k.startswith("MSE")][0]
    assert abs(s1[k] - s2[k]) <= 1e-4 * max(abs(s1[k]), 1e-6) + 1e-6
