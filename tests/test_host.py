"""CPU: host-side logic of the drop-in surface (no GPU work): parameter table / state_dict contract, schedulers'
host parts against the reference KATs, Backdoor constructors against golden vectors, CLI rules, checkpoint I/O."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import train_ref
from oracle import unet_ref as U
from tests.golden import cases as C


def test_param_table_matches_reference_state_dict_contract():
    from baddiffusion_amd.unet import unet_from_config
    for cfg, n in ((U.CIFAR10_32, 35746307), (U.CELEBA_HQ_256, 113673219), (C.SMALL_CFGS["small_default"], None)):
        m = unet_from_config(cfg)
        shapes = U.param_shapes(cfg)
        sd = m.state_dict()
        assert set(sd) == set(shapes)
        assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
        if n:
            assert sum(v.numel() for v in sd.values()) == n
        P = U.gen_params(cfg, 1)
        m.load_state_dict(P)
        sd = m.state_dict()
        assert all(torch.equal(sd[k], P[k]) for k in P)
        with pytest.raises(RuntimeError):
            m.load_state_dict({k: v for k, v in list(P.items())[:-1]})
        if cfg is U.CELEBA_HQ_256:
            break


def test_unsupported_configs_fail_loudly():
    from baddiffusion_amd.unet import UNet2DModel
    kw = dict(sample_size=16, block_out_channels=(128, 256), layers_per_block=1)
    with pytest.raises(NotImplementedError):
        UNet2DModel(down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"), **kw)
    with pytest.raises(NotImplementedError):
        UNet2DModel(down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
                    resnet_time_scale_shift="scale_shift", **kw)
    with pytest.raises(ValueError):
        UNet2DModel(down_block_types=("DownBlock2D",), up_block_types=("AttnUpBlock2D", "UpBlock2D"), **kw)
    with pytest.raises(RuntimeError):   # C plan: channels must be multiples of the group count
        UNet2DModel(down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
                    sample_size=16, block_out_channels=(100, 256), layers_per_block=1)


def test_scheduler_host_side(golden):
    from baddiffusion_amd.schedulers import DDIMScheduler, DDPMScheduler
    g = golden("sched")
    s = DDPMScheduler()
    assert np.array_equal(s.betas.numpy(), g["betas"]) and np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    # diffusers/tests/schedulers/test_scheduler_ddpm.py:62-69
    assert abs(float(s._get_variance(0)) - 0.0) < 1e-5 and abs(float(s._get_variance(487)) - 0.00979) < 1e-5
    assert abs(float(s._get_variance(999)) - 0.02) < 1e-5
    s.set_timesteps(50)
    assert np.array_equal(s.timesteps.numpy(), g["ddpm_ts50"]) and int(s.previous_timestep(980)) == 960
    with pytest.raises(ValueError):
        s.set_timesteps(timesteps=[100, 100, 50])                       # test_scheduler_ddpm.py:160-170
    with pytest.raises(ValueError):
        s.set_timesteps(num_inference_steps=10, timesteps=[100, 50])
    d = DDIMScheduler(steps_offset=1); d.set_timesteps(5)
    assert list(d.timesteps.numpy()) == [801, 601, 401, 201, 1]          # test_scheduler_ddim.py:46-54
    d = DDIMScheduler.from_config(DDPMScheduler(clip_sample=False).config); d.set_timesteps(50)
    assert d.config.clip_sample is False and np.array_equal(d.timesteps.numpy(), g["ddim_ts50"])
    assert abs(float(d._get_variance(420, 400)) - 0.14771) < 1e-5        # test_scheduler_ddim.py:94-104
    s.config.clip_sample = True                                          # assignable (model.py:639-641)
    assert s.config["clip_sample"] is True
    with pytest.raises(RuntimeError):
        s.step(torch.zeros(1, 3, 4, 4), 5, torch.zeros(1, 3, 4, 4))      # CPU tensors are rejected


def test_backdoor_product_vs_golden(golden):
    from baddiffusion_amd.dataset import Backdoor, DatasetLoader
    g = golden("backdoor")
    bd = Backdoor(root=None)
    for S in (32, 256):
        for trig in ("BOX_4", "BOX_8", "BOX_11", "BOX_14", "BOX_18", "SM_BOX", "NONE"):
            t = bd.get_trigger(trig, 3, S)
            assert np.array_equal(t.numpy(), g[f"trig_{trig}_{S}"])
            if S == 32 or trig == "BOX_14":
                for tg in ("CORNER", "TRIGGER", "SHIFT"):
                    assert np.array_equal(bd.get_target(tg, t).numpy(), g[f"tgt_{tg}_{trig}_{S}"])
    with pytest.raises(ValueError):
        bd.get_trigger("NOPE", 3, 32)
    with pytest.raises(FileNotFoundError):
        bd.get_target("HAT", bd.get_trigger("BOX_14", 3, 32))            # static/ assets are not redistributed
    dsl = DatasetLoader(root=None, name="CIFAR10", batch_size=128, num_images=1000, device="cpu")
    dsl.set_poison("BOX_14", "CORNER", clean_rate=1.0, poison_rate=0.1).prepare_dataset("FIXED")
    assert int(dsl._is_poison.sum()) == 100 and dsl.num_batch == 8 and dsl.image_size == 32 and dsl.channel == 3
    assert torch.equal(dsl.get_mask(dsl.trigger), torch.from_numpy(g["mask_BOX_14_32"]))
    assert dsl.source == "synthetic"


def test_lr_schedule_and_cli_rules(tmp_path, monkeypatch):
    from baddiffusion_amd.trainer import cosine_schedule_with_warmup
    for s in (0, 1, 499, 500, 7777, 23449, 23450):
        assert cosine_schedule_with_warmup(s, 500, 23450) == train_ref.cosine_lr_lambda(s, 500, 23450)
    import baddiffusion as cli
    monkeypatch.chdir(tmp_path)
    cfg = cli.setup(["--project", "default", "--mode", "train", "--dataset", "CIFAR10", "--batch", "64", "--epoch", "50",
                     "--poison_rate", "0.1", "--trigger", "BOX_14", "--target", "HAT", "--ckpt", "DDPM-CIFAR10-32", "--fclip", "o", "-o"])
    assert cfg.gradient_accumulation_steps == 2 and cfg.learning_rate == 2e-4 and cfg.clip is False
    assert os.path.basename(cfg.output_dir) == "res_DDPM-CIFAR10-32_CIFAR10_ep50_c1.0_p0.1_BOX_14-HAT"
    assert json.load(open(os.path.join(cfg.output_dir, "args.json")))["poison_rate"] == 0.1
    with pytest.raises(ValueError):      # batch must divide 128 (baddiffusion.py:213-217)
        cli.setup(["--mode", "train", "--dataset", "CIFAR10", "--batch", "48", "-o"], write=False)
    with pytest.raises(NotImplementedError):   # per-mode whitelist (baddiffusion.py:163-175)
        cli.setup(["--mode", "sampling", "--ckpt", cfg.output_dir, "--epoch", "3"], write=False)
    with pytest.raises(NotImplementedError):
        cli.setup(["--mode", "train", "--dataset", "CIFAR10", "--batch", "128", "--sample_ep", "3", "-o"], write=False)
    c2 = cli.setup(["--mode", "sampling", "--ckpt", cfg.output_dir, "--fclip", "w", "--eval_max_batch", "2048", "--sched", "DDIM-SCHED"])
    assert c2.clip is True and c2.dataset == "CIFAR10" and c2.sched == "DDIM-SCHED" and c2.eval_max_batch == 2048
    c3 = cli.setup(["--mode", "train", "--dataset", "CELEBA-HQ", "--batch", "4", "-o"], write=False)
    assert c3.gradient_accumulation_steps == 16 and c3.learning_rate == 2e-5


def test_checkpoint_layout_roundtrip(tmp_path):
    from baddiffusion_amd.model import DiffuserModelSched, load_scheduler, load_unet
    with pytest.raises(FileNotFoundError):      # like from_pretrained: no silent random-init fine-tuning
        DiffuserModelSched.get_pretrained("DDPM-CIFAR10-32", clip_sample=False, noise_sched_type="DDPM-SCHED")
    model, sched, get_pipeline = DiffuserModelSched.get_pretrained("DDPM-CIFAR10-32", clip_sample=False, noise_sched_type="DDPM-SCHED",
                                                                   allow_random_init=True)
    assert model.pretrained is False
    d = str(tmp_path / "run")
    get_pipeline(model, sched).save_pretrained(d)
    assert sorted(os.listdir(d)) == ["model_index.json", "scheduler", "unet"]
    assert sorted(os.listdir(os.path.join(d, "unet"))) == ["config.json", "diffusion_pytorch_model.bin"]
    sd = torch.load(os.path.join(d, "unet", "diffusion_pytorch_model.bin"))
    assert tuple(sd["conv_in.weight"].shape) == (128, 3, 3, 3) and sd["conv_in.weight"].is_contiguous()
    m2 = load_unet(os.path.join(d, "unet"))
    assert torch.equal(m2.flat, model.flat)
    m3, s3, _ = DiffuserModelSched.get_trained(d, clip_sample=None)
    assert torch.equal(m3.flat, model.flat) and s3.config.clip_sample is False and m3.pretrained
    # SURVEY f-4 (model.py:598-630): the other `--sched` types return a config carrier + a PNDMPipeline factory
    from baddiffusion_amd.pipelines import PNDMPipeline
    from baddiffusion_amd.schedulers import PNDMScheduler
    m4, s4, gp4 = DiffuserModelSched.get_pretrained("DDPM-CIFAR10-32", noise_sched_type="UNIPC-SCHED", allow_random_init=True)
    assert s4.class_name == "UniPCMultistepScheduler" and s4.config.num_train_timesteps == 1000 and s4.config.clip_sample is False
    p4 = gp4(m4, s4)
    assert isinstance(p4, PNDMPipeline) and isinstance(p4.scheduler, PNDMScheduler) and p4.clip_sample is False
    with pytest.raises(NotImplementedError):
        s4.step(None, 0, None)
    _, s5, _ = DiffuserModelSched.get_pretrained("DDPM-CIFAR10-32", noise_sched_type="PNDM-SCHED", allow_random_init=True)
    assert isinstance(s5, PNDMScheduler)
    with pytest.raises(NotImplementedError):     # named in model.py:556-563, rejected by its __get_model_sched as well
        DiffuserModelSched.get_pretrained("DDPM-CIFAR10-32", noise_sched_type="EDM-VE-SCHED", allow_random_init=True)


def test_dataset_modes_and_rank_sharding():
    """FIXED vs FLEX sizes (/root/reference/dataset.py:162-243) and equal, non-empty per-rank batches incl. the tail."""
    from baddiffusion_amd.dataset import Backdoor, DatasetLoader
    n = 203
    mk = lambda: DatasetLoader(root=None, name="CIFAR10", batch_size=8, seed=3, num_images=n, device="cpu")
    d = mk().set_poison(Backdoor.TRIGGER_BOX_14 if hasattr(Backdoor, "TRIGGER_BOX_14") else "BOX_14", "CORNER", clean_rate=0.5,
                        poison_rate=0.1).prepare_dataset("FIXED")
    assert len(d) == n and int(d._is_poison.sum()) == int(n * 0.1)            # clean_rate ignored in FIXED
    f = mk().set_poison("BOX_14", "CORNER", clean_rate=0.5, poison_rate=0.1).prepare_dataset("FLEX")
    train_n, test_n = int(n * 0.5), int(n * 0.1)
    assert len(f) == train_n + test_n
    rows = f._rows()
    assert int(f._is_poison[rows].sum()) == test_n and int(f._is_poison.sum()) == test_n and len(set(rows.tolist())) == len(rows)
    with pytest.raises(ValueError):
        mk().set_poison("BOX_14", "CORNER", clean_rate=0.95, poison_rate=0.1).prepare_dataset("FLEX")
    with pytest.raises(NotImplementedError):
        mk().set_poison("BOX_14", "CORNER").prepare_dataset("OTHER")
    with pytest.raises(NotImplementedError):      # class filter without labels
        DatasetLoader(root=None, name="CIFAR10", label=[1], batch_size=8, num_images=16, device="cpu").prepare_dataset("FIXED")
    # rank sharding: same count on every rank at every step (203 % (8*3) = 11, 11 % 3 = 2 -> tail wrap-padded), disjoint
    # rows inside a step, and the union over ranks covers the dataset
    world = 3
    per_rank = [list(d.device_batches(shuffle=True, epoch=1, rank=r, world=world, flip=False)) for r in range(world)]
    assert len({len(b) for b in per_rank}) == 1
    seen = 0
    for step in range(len(per_rank[0])):
        sizes = {per_rank[r][step][0].shape[0] for r in range(world)}
        assert len(sizes) == 1 and 0 not in sizes
        seen += sum(per_rank[r][step][0].shape[0] for r in range(world))
    assert n <= seen < n + world
    # flip flags are drawn per global row, so world = 1 and world = 3 flip the same images
    one = torch.cat([b[0] for b in d.device_batches(shuffle=True, epoch=1, rank=0, world=1, flip=True)])
    assert one.shape[0] == n


def test_reference_checkpoint_layout(tmp_path):
    """G9: the directory the REFERENCE's save_pretrained wrote (tests/golden/ckpt, captured by make_ckpt_fixture.py from
    pipeline_utils.py:527-600 / modeling_utils.py:287-301): the product loads it (JSON files verbatim, weights
    re-materialised from the same seed and checked against the stored probes) and saves a directory with the same tree,
    the same JSON keys / values and the same tensor manifest (keys, order, shapes, dtypes, contiguity)."""
    import json
    import shutil
    import numpy as np
    from baddiffusion_amd.model import DiffuserModelSched
    from oracle import unet_ref as U
    from tests.golden import cases as C
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt")
    man = json.load(open(os.path.join(src, "manifest.json")))
    probe = np.load(os.path.join(src, "weights_probe.npz"))
    d = str(tmp_path / "ref")
    shutil.copytree(src, d)
    P = U.gen_params(C.SMALL_CFGS[man["config"]], man["seed"])
    # (key ORDER is a registration-order artefact of nn.Module -- the reference lists up_blocks before mid_block and
    #  attentions before resnets -- and no consumer of a state dict depends on it: compared as a set)
    assert sorted(e["key"] for e in man["state_dict"]) == sorted(P.keys())
    for e in man["state_dict"]:      # the regenerated weights are the ones the reference stored
        v = P[e["key"]]
        assert list(v.shape) == e["shape"] and str(v.dtype) == e["dtype"]
        np.testing.assert_allclose(np.concatenate([v.flatten()[:4].double().numpy(), [float(v.double().sum())]]), probe[e["key"]], rtol=1e-12)
    torch.save({k: v.clone() for k, v in P.items()}, os.path.join(d, "unet", "diffusion_pytorch_model.bin"))
    model, sched, get_pipeline = DiffuserModelSched.get_trained(d, clip_sample=None)
    assert model.pretrained and type(sched).__name__ == "DDPMScheduler"
    assert sched.config.variance_type == "fixed_large" and sched.config.clip_sample is False
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(P.keys()) and all(torch.equal(sd[k].cpu(), P[k]) for k in P)
    out = str(tmp_path / "out")
    get_pipeline(model, sched).save_pretrained(out)
    tree = sorted(os.path.relpath(os.path.join(r, f), out) for r, _, fs in os.walk(out) for f in fs)
    assert tree == man["tree"]
    for rel in tree:
        if rel.endswith(".json"):
            ours, ref = json.load(open(os.path.join(out, rel))), json.load(open(os.path.join(src, rel)))
            assert ours == ref, (rel, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)})
    sd2 = torch.load(os.path.join(out, "unet", "diffusion_pytorch_model.bin"), map_location="cpu")
    key = lambda e: e["key"]
    assert sorted(({"key": k, "shape": list(v.shape), "dtype": str(v.dtype), "contiguous": bool(v.is_contiguous())} for k, v in sd2.items()),
                  key=key) == sorted(man["state_dict"], key=key)
    assert all(torch.equal(sd2[k], P[k]) for k in P)


@pytest.mark.skipif(not os.path.exists("/root/reference/static/fedora-hat.png"),
                    reason="the reference's static/ image assets are not redistributed (build container only)")
def test_image_file_triggers_match_reference_fixture():
    """GLASSES / STOP_SIGN triggers and HAT / CAT targets (dataset.py:428-497, 576-597, 643-655): the product's
    Backdoor, reading the reference's assets in place, against tests/golden/img_triggers.npz -- outputs of the reference's
    own Backdoor class (generator: tests/golden/make_trigger_fixture.py; its header states what the torchvision stand-in
    does and does not pin).  BASELINE configs[1]'s HAT target and configs[3]'s GLASSES -> CAT pair are among them."""
    from baddiffusion_amd.dataset import Backdoor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "img_triggers.npz"))
    bd = Backdoor(root="/root/reference")
    n = 0
    for key in g.files:
        kind, name = key.split("_", 1)
        name, c, s = name.rsplit("_", 2)
        ch, size = int(c[1:]), int(s[1:])
        if kind == "trigger":
            got = bd.get_trigger(name, ch, size)
        else:
            got = bd.get_target(name, bd.get_trigger("BOX_14", ch, size))
        want = torch.from_numpy(g[key])
        assert got.shape == want.shape and got.dtype == want.dtype, key
        assert torch.equal(got, want), (key, float((got - want).abs().max()))
        n += 1
    assert n == 9


def test_resume_rng_state_is_per_rank(tmp_path):
    """rank 0's RNG state must not be installed on every rank.  checkpoint() (rank 0) + save_rank_rng (others)
    write one state per rank; restore_training_state(rank) gives each rank its own stream back, and a rank that has no state of
    its own is re-seeded with seed + rank instead of silently sharing rank 0's."""
    import baddiffusion as cli
    config = cli.TrainingConfig()
    config.seed = 3
    config.ckpt_path = str(tmp_path / "ckpt"); config.data_ckpt_path = str(tmp_path / "data.ckpt")
    torch.manual_seed(100)                                        # "rank 0"
    torch.save({"epoch": 1, "step": 7, "world": 1, "cpu_rng": torch.get_rng_state(), "cuda_rng": None}, config.data_ckpt_path)
    next0 = torch.randn(4)
    torch.manual_seed(101)                                        # "rank 1"
    cli.save_rank_rng(config, 1, cli.capture_rank_rng(1, 7))
    next1 = torch.randn(4)
    assert os.path.exists(config.data_ckpt_path + ".rank1") and not torch.equal(next0, next1)
    torch.manual_seed(999)
    assert cli.restore_training_state(config, None, rank=0) == (1, 7) and torch.equal(torch.randn(4), next0)
    torch.manual_seed(999)
    assert cli.restore_training_state(config, None, rank=1) == (1, 7) and torch.equal(torch.randn(4), next1)
    assert cli.restore_training_state(config, None, rank=2) == (1, 7)     # no file: seed + rank
    want = torch.randn(4, generator=torch.Generator().manual_seed(config.seed + 2))
    assert torch.equal(torch.randn(4), want)
    # a rank file that belongs to ANOTHER checkpoint (rank 0 crashed before writing data.ckpt, ranks failed at
    # different steps, a stale file of an earlier run) must not be installed silently
    torch.manual_seed(555)
    cli.save_rank_rng(config, 1, cli.capture_rank_rng(2, 14))              # one checkpoint ahead of data.ckpt (1, 7)
    assert cli.restore_training_state(config, None, rank=1) == (1, 7)
    want = torch.randn(4, generator=torch.Generator().manual_seed(config.seed + 1))
    assert torch.equal(torch.randn(4), want)
    st = cli.capture_rank_rng(1, 7); st["world"] = 4                       # same step, but written by a 4-rank run
    cli.save_rank_rng(config, 1, st)
    assert cli.restore_training_state(config, None, rank=1) == (1, 7)
    assert torch.equal(torch.randn(4), torch.randn(4, generator=torch.Generator().manual_seed(config.seed + 1)))


def test_fid_inception_manifest():
    """f-3: the FID feature extractor's state-dict manifest (baddiffusion_amd/inception.py) names what pytorch_fid's
    pt_inception-2015-12-05 checkpoint holds -- 94 BasicConv2d (conv.weight + 4 BatchNorm tensors each) + the unused 1008-way
    fc head = 472 tensors, 23 885 392 parameters -- and agrees with the oracle's independent statement key for key.  A state dict
    with a wrong shape, a missing or an unknown key is refused (no GPU needed: the check precedes any device work)."""
    import pytest
    from baddiffusion_amd import inception as P
    from oracle import inception_ref as I
    a, b = P.state_dict_manifest(), I.manifest()
    assert list(a.items()) == list(b.items()) and len(a) == 472
    assert sum(int(np.prod(s)) for s in a.values()) == 23_885_392
    assert a["Conv2d_1a_3x3.conv.weight"] == (32, 3, 3, 3) and a["Mixed_6c.branch7x7dbl_4.conv.weight"] == (160, 160, 7, 1)
    assert a["Mixed_7c.branch3x3dbl_1.conv.weight"] == (448, 2048, 1, 1) and a["fc.weight"] == (1008, 2048)
    with pytest.raises(RuntimeError, match="GPU only"):
        P.FIDInceptionV3(device="cpu")
    assert P.load_fid_weights("/nonexistent/pt_inception.pth") is None


def test_dp_communication_buckets_cover_every_gradient_range():
    """trainer.plan_buckets: backward segments fused into communication buckets.  Every range a segment finalises lies inside a range of
    the bucket that closes at or after that segment (never before: the collective must not precede its producer), bucket ranges are
    disjoint, the fused gaps are alignment pads only, and the last bucket -- the one collective nothing overlaps -- is small."""
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.trainer import merge_ranges, plan_buckets, plan_segment_ranges
    from baddiffusion_amd.unet import UNet2DModel
    assert merge_ranges([(10, 20), (0, 10), (30, 40), (38, 50), (200, 300)]) == [(0, 50), (200, 300)]
    # a gap is bridged only when it is padding: 33 floats could be a whole foreign tensor (conv_out.bias, a 32-channel bias row)
    assert merge_ranges([(0, 96), (129, 200)]) == [(0, 96), (129, 200)] and merge_ranges([(0, 97), (128, 200)]) == [(0, 200)]
    assert merge_ranges([(0, 97), (128, 200)], pads=[(97, 128)]) == [(0, 200)] and merge_ranges([(0, 97), (128, 200)], pads=[(100, 128)]) == [(0, 97), (128, 200)]
    tiny = dict(sample_size=16, block_out_channels=(32, 64), down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1, norm_num_groups=8)     # 32-float tensors next to the range borders
    for name in ("google/ddpm-cifar10-32", "google/ddpm-ema-celebahq-256", "tiny-32-64"):
        m = UNet2DModel(**(tiny if name.startswith("tiny") else KNOWN_TOPOLOGIES[name]))
        sr = plan_segment_ranges(m)
        seg_elems = sum(hi - lo for rs in sr for lo, hi in rs)
        for mb in (0, 16, 32, 1e6):
            b = plan_buckets(sr, int(mb * 2 ** 20), pads=m._pads)
            closes = [s for s, _ in b]
            assert closes == sorted(closes) and closes[-1] == len(sr) - 1
            flat = sorted(r for _, rs in b for r in rs)
            assert all(a[1] <= c[0] for a, c in zip(flat, flat[1:]))                       # disjoint
            assert 0 <= sum(hi - lo for lo, hi in flat) - seg_elems < 64 * len(flat) * 4     # fused gaps = pads
            own = sorted(r for rs in sr for r in rs)
            for blo, bhi in flat:       #  what a bucket range holds beyond its source ranges is a subset of model._pads
                cur = blo
                for lo, hi in [r for r in own if blo <= r[0] and r[1] <= bhi] + [(bhi, bhi)]:
                    if lo > cur:
                        assert any(plo <= cur and lo <= phi for plo, phi in m._pads), (name, mb, cur, lo)
                    cur = max(cur, hi)
            for s, rs in enumerate(sr):
                owner = next(i for i, c in enumerate(closes) if c >= s)
                for lo, hi in rs:
                    assert any(blo <= lo and hi <= bhi for blo, bhi in b[owner][1]), (name, mb, s, lo, hi)
            if mb == 32 and not name.startswith("tiny"):
                assert sum(hi - lo for lo, hi in b[-1][1]) * 4 <= 8 << 20 and len(b) <= 8


def test_rccl_binding_loads_and_exports_the_calls_used():
    """baddiffusion_amd/rccl.py binds the librccl.so that ships with PyTorch-ROCm through ctypes: the library loads without a GPU and
    exports every entry point the gradient exchange uses (no communicator is created here -- that needs a device)."""
    from baddiffusion_amd import rccl
    lib = rccl._load()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclBroadcast", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd",
                 "ncclGetErrorString"):
        assert hasattr(lib, name), name
    import ctypes
    assert ctypes.sizeof(rccl._UniqueId) == 128
    assert b"" != lib.ncclGetErrorString(0)
    # the 128 bytes that travel to the other ranks are the WHOLE id: a c_char array read as a C string stops at the first NUL byte (round 6: the
    # pre-round-6 broadcast did exactly that, and would have failed the first multi-rank ncclCommInitRank)
    u = rccl.unique_id()
    assert isinstance(u, bytes) and len(u) == 128 and 0 < u.find(b"\0") < 127


def test_bench_headline_is_bounded_and_parses():
    """The ONE line bench.py prints is built from the detail object by bench.headline_line; the driver reads a bounded tail of stdout, and round 5's
    20 KB line did not parse there.  Rebuild the line from a recorded full detail object (profiles/r05_bench.json: the 20 KB line of round 5, every
    rider present) and from degenerate ones: < 8 KB, valid JSON, the contract's keys + roofline + cpu_baseline intact, riders reduced to value /
    ms_per_step / roofline.frac / roofline.traffic / cpu_baseline.value."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    out = json.load(open(os.path.join(root, "profiles", "r05_bench.json")))
    assert len(json.dumps(out)) > 16000                           # the recorded object really is the oversized one
    line = b.headline_line(out, "gpurun_out/bench_detail.json")
    assert len(line) < b.HEADLINE_MAX_BYTES == 8192 and "\n" not in line
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in h and (h[k] == out[k] or isinstance(out[k], float)), k
    assert abs(h["value"] - out["value"]) < 1e-5 * out["value"] and abs(h["ms_per_step"] - out["ms_per_step"]) < 1e-5 * out["ms_per_step"]
    r = h["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["standalone"]["frac"] > r["frac"]
    assert h["cpu_baseline"]["value"] > 0 and h["cpu_baseline"]["cores"] >= 1 and h["cpu_baseline"]["kind"] == "port"
    for rider in (h["sampling"]["ddim50"], h["sampling"]["ddpm1000"], h["celeba"]):
        assert rider["value"] > 0 and 0 < rider["roofline"]["frac"] < 1 and rider["cpu_baseline"]["value"] > 0
        assert all(not isinstance(v, str) or len(v) < 64 for v in rider.values())
    assert h["fid_features"]["value"] > 0 and h["detail_file"] == "gpurun_out/bench_detail.json"
    assert not any(k in h for k in ("kernel_classes", "kernel_classes_standalone"))
    # an object whose riders are absurdly large still yields a parsable contract line (riders dropped, never the contract)
    fat = dict(out, sampling={f"k{i}": dict(out["sampling"]["ddim50"]) for i in range(200)})
    line = b.headline_line(fat, None)
    assert len(line) < 8192 and json.loads(line)["roofline"]["kernel"] == r["kernel"] and "cpu_baseline" in json.loads(line)
    # riders that failed carry a bounded error string
    bad = dict(out, celeba={"error": "x" * 5000}, fid_features={"error": "y" * 5000})
    hb = json.loads(b.headline_line(bad, None))
    assert len(hb["celeba"]["error"]) <= 120 and len(hb["fid_features"]["error"]) <= 120


def test_sharded_generator_reproduces_the_single_stream():
    """SURVEY hard-part 4 / VERDICT round 5 missing item 6: the reference draws every sampling noise from ONE CPU generator, chunk after chunk
    (model.py:517-523, scheduling_ddpm.py:400-404).  ShardedGenerator + randn_tensor: each rank draws the full chunk's tensor and keeps its
    rows, so the rows of all ranks together are exactly the single process's draws, draw after draw, and the streams stay in step."""
    from baddiffusion_amd.schedulers import ShardedGenerator, randn_tensor
    full = torch.Generator().manual_seed(11)
    want = [torch.randn(5, 3, 4, 4, generator=full) for _ in range(3)] + [torch.randn(2, 3, 4, 4, generator=full)]
    world = 3
    gens = [torch.Generator().manual_seed(11) for _ in range(world)]
    for k, (bs, w) in enumerate([(5, want[0]), (5, want[1]), (5, want[2]), (2, want[3])]):
        per = (bs + world - 1) // world
        rows = []
        for r in range(world):
            lo, hi = min(bs, r * per), min(bs, (r + 1) * per)
            if hi > lo:
                rows.append(randn_tensor((hi - lo, 3, 4, 4), generator=ShardedGenerator(gens[r], bs, lo, hi)))
            else:
                torch.randn(bs, 3, 4, 4, generator=gens[r])          # what pipeline.advance_generator does for a rank without rows
        assert torch.equal(torch.cat(rows), w), k
    assert all(torch.equal(g.get_state(), full.get_state()) for g in gens)
    with pytest.raises(ValueError):
        randn_tensor((2, 3, 4, 4), generator=ShardedGenerator(gens[0], 5, 0, 3))      # asked for another row count than it owns
    with pytest.raises(ValueError):
        ShardedGenerator(None, 5, 0, 3)
