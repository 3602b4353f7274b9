"""baddiffusion.py -- command line of the MI355X-native BadDiffusion engine.

Same modes, flags, per-mode flag whitelist, defaults, derived settings and output-directory naming as
/root/reference/baddiffusion.py (:20-134 flags/defaults, :144-248 setup, :572-645 train loop, :366-419
sampling, :477-551 measure, :558-570 checkpoint), but:
  * nothing runs at import time (the reference calls setup() + wandb.init on import, :246-250);
  * one process per GPU: launch N ranks with torch.distributed.run and the global batch is sharded
    (`--batch` stays the PER-RANK micro-batch; the 128 / 64 effective-batch rule applies per rank);
  * the dataset lives in HBM as uint8 and the loop body is `TrainEngine.train_step` (fused HIP path);
  * wandb / tensorboard are optional (absent here); losses are logged to <output_dir>/log.jsonl.

    python baddiffusion.py --project default --mode train --dataset CIFAR10 --batch 128 --epoch 50 \
        --poison_rate 0.1 --trigger BOX_14 --target HAT --ckpt DDPM-CIFAR10-32 --fclip o -o --gpu 0
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / CUDA-tensor sharing across ranks fail with hipIpcGetMemHandle: invalid
# argument otherwise); the driver's environment exports it, a bare shell may not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import traceback
from dataclasses import dataclass

import numpy as np
import torch

from baddiffusion_amd.dataset import Backdoor, DatasetLoader

MODE_TRAIN, MODE_RESUME, MODE_SAMPLING, MODE_MEASURE, MODE_TRAIN_MEASURE = "train", "resume", "sampling", "measure", "train+measure"

DEFAULT_LEARNING_RATE_32, DEFAULT_LEARNING_RATE_256 = 2e-4, 8e-5
NOT_MODE_TRAIN_OPTS = ["sample_ep"]
NOT_MODE_TRAIN_MEASURE_OPTS = ["sample_ep"]
MODE_RESUME_OPTS = ["project", "mode", "gpu", "ckpt"]
MODE_SAMPLING_OPTS = ["project", "mode", "eval_max_batch", "gpu", "fclip", "ckpt", "sample_ep", "sched"]
MODE_MEASURE_OPTS = ["project", "mode", "eval_max_batch", "gpu", "fclip", "ckpt", "sample_ep", "sched"]
IGNORE_ARGS = ["overwrite", "is_save_all_model_epochs"]
SCHED_CHOICES = ["DDPM-SCHED", "DDIM-SCHED", "DPM_SOLVER_PP_O1-SCHED", "DPM_SOLVER_O1-SCHED", "DPM_SOLVER_PP_O2-SCHED",
                 "DPM_SOLVER_O2-SCHED", "DPM_SOLVER_PP_O3-SCHED", "DPM_SOLVER_O3-SCHED", "UNIPC-SCHED", "PNDM-SCHED", "DEIS-SCHED",
                 "HEUN-SCHED", "SCORE-SDE-VE-SCHED"]


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--project", "-pj", type=str, help="Project name")
    p.add_argument("--mode", "-m", required=True, type=str, choices=[MODE_TRAIN, MODE_RESUME, MODE_SAMPLING, MODE_MEASURE, MODE_TRAIN_MEASURE])
    p.add_argument("--dataset", "-ds", type=str, choices=[DatasetLoader.MNIST, DatasetLoader.CIFAR10, DatasetLoader.CELEBA, DatasetLoader.CELEBA_HQ])
    p.add_argument("--batch", "-b", type=int, help="(per-rank) batch size")
    p.add_argument("--sched", "-sc", type=str, choices=SCHED_CHOICES, help="Noise scheduler")
    p.add_argument("--eval_max_batch", "-eb", type=int)
    p.add_argument("--epoch", "-e", type=int)
    p.add_argument("--learning_rate", "-lr", type=float)
    p.add_argument("--clean_rate", "-cr", type=float)
    p.add_argument("--poison_rate", "-pr", type=float)
    p.add_argument("--trigger", "-tr", type=str)
    p.add_argument("--target", "-ta", type=str)
    p.add_argument("--dataset_load_mode", "-dlm", type=str, choices=[DatasetLoader.MODE_FIXED, DatasetLoader.MODE_FLEX])
    p.add_argument("--gpu", "-g", type=str)
    p.add_argument("--ckpt", "-c", type=str)
    p.add_argument("--overwrite", "-o", action="store_true")
    p.add_argument("--postfix", "-p", type=str)
    p.add_argument("--fclip", "-fc", type=str, choices=["w", "o"])
    p.add_argument("--save_image_epochs", "-sie", type=int)
    p.add_argument("--save_model_epochs", "-sme", type=int)
    p.add_argument("--is_save_all_model_epochs", "-isame", action="store_true")
    p.add_argument("--sample_ep", "-se", type=int)
    p.add_argument("--result", "-res", type=str)
    return p.parse_args(argv)


@dataclass
class TrainingConfig:
    project: str = "Default"
    batch: int = 512
    epoch: int = 50
    eval_max_batch: int = 256
    learning_rate: float = None
    clean_rate: float = 1.0
    poison_rate: float = 0.007
    trigger: str = Backdoor.TRIGGER_BOX_14
    target: str = Backdoor.TARGET_CORNER
    dataset_load_mode: str = DatasetLoader.MODE_FIXED
    gpu: str = "0"
    ckpt: str = None
    overwrite: bool = False
    postfix: str = ""
    fclip: str = "o"
    save_image_epochs: int = 20
    save_model_epochs: int = 5
    is_save_all_model_epochs: bool = False
    sample_ep: int = None
    result: str = "."
    eval_sample_n: int = 16
    measure_sample_n: int = 2048
    batch_32: int = 128
    batch_256: int = 64
    gradient_accumulation_steps: int = 1
    learning_rate_32_scratch: float = 2e-4
    learning_rate_256_scratch: float = 2e-5
    lr_warmup_steps: int = 500
    mixed_precision: str = "no"      # this engine trains in fp32 (the reference hard-codes fp16 AMP, :116)
    seed: int = 0
    dataset_path: str = "datasets"
    ckpt_dir: str = "ckpt"
    data_ckpt_dir: str = "data.ckpt"
    ep_model_dir: str = "epochs"
    ckpt_path: str = None
    data_ckpt_path: str = None


def naming_fn(config):
    add_on = f"_{config.postfix}" if config.postfix else ""
    return f"res_{config.ckpt}_{config.dataset}_ep{config.epoch}_c{config.clean_rate}_p{config.poison_rate}_{config.trigger}-{config.target}{add_on}"


def _write_json(content, config, file):
    def default(o):
        return str(o)
    with open(os.path.join(config.output_dir, file), "w") as f:
        json.dump(content, f, indent=2, default=default)


def setup(argv=None, write=True):
    """baddiffusion.py:144-248 without side effects other than the run directory / JSON files."""
    args = parse_args(argv)
    config = TrainingConfig()
    if args.mode in (MODE_RESUME, MODE_SAMPLING, MODE_MEASURE):
        if args.ckpt is None:
            raise ValueError(f"--ckpt <run directory> is required in mode {args.mode}")
        with open(os.path.join(args.ckpt, "args.json")) as f:
            for k, v in json.load(f).items():
                if v is not None:
                    setattr(config, k, v)
        config.output_dir = args.ckpt
    allowed = {MODE_RESUME: MODE_RESUME_OPTS, MODE_SAMPLING: MODE_SAMPLING_OPTS, MODE_MEASURE: MODE_MEASURE_OPTS}
    for key, value in vars(args).items():
        if value is None or value is False:
            continue
        if args.mode == MODE_TRAIN and key not in NOT_MODE_TRAIN_OPTS:
            setattr(config, key, value)
        elif args.mode == MODE_TRAIN_MEASURE and key not in NOT_MODE_TRAIN_MEASURE_OPTS:
            setattr(config, key, value)
        elif args.mode in allowed and key in allowed[args.mode]:
            setattr(config, key, value)
        elif key not in IGNORE_ARGS:
            raise NotImplementedError(f"Argument: {key}={value} isn't used in mode: {args.mode}")
    config.mode = args.mode
    if not hasattr(config, "dataset"):
        raise ValueError("--dataset is required")
    if not hasattr(config, "sched"):
        config.sched = None                      # the reference leaves this attribute undefined (SURVEY D-3)
    config.device_ids = [int(i) for i in range(len(str(config.gpu).split(",")))]
    if isinstance(config.sample_ep, int) and config.sample_ep < 0:
        config.sample_ep = None
    config.clip = {"w": True, "o": False}.get(config.fclip)
    if config.dataset in (DatasetLoader.CIFAR10, DatasetLoader.MNIST):
        bs = config.batch_32
        if config.learning_rate is None:
            config.learning_rate = config.learning_rate_32_scratch if config.ckpt is None else DEFAULT_LEARNING_RATE_32
    elif config.dataset in (DatasetLoader.CELEBA, DatasetLoader.CELEBA_HQ, DatasetLoader.LSUN_CHURCH, DatasetLoader.LSUN_BEDROOM):
        bs = config.batch_256
        if config.learning_rate is None:
            config.learning_rate = config.learning_rate_256_scratch if config.ckpt is None else DEFAULT_LEARNING_RATE_256
    else:
        raise NotImplementedError()
    if bs % config.batch != 0:
        raise ValueError(f"batch size {config.batch} should be divisible to {bs} for dataset {config.dataset}")
    if bs < config.batch:
        raise ValueError(f"batch size {config.batch} should be smaller or equal to {bs} for dataset {config.dataset}")
    config.gradient_accumulation_steps = int(bs // config.batch)
    if args.mode in (MODE_TRAIN, MODE_TRAIN_MEASURE):
        config.output_dir = os.path.join(config.result, naming_fn(config))
    rank = int(os.environ.get("RANK", "0"))
    if write and rank == 0:
        if args.mode in (MODE_TRAIN, MODE_TRAIN_MEASURE):
            if not config.overwrite and os.path.isdir(config.output_dir):
                raise ValueError(f"Output directory: {config.output_dir} has already been created, please set overwrite flag --overwrite or -o")
            os.makedirs(config.output_dir, exist_ok=True)
            _write_json(vars(args), config, "args.json")
            _write_json(config.__dict__, config, "config.json")
        elif args.mode == MODE_SAMPLING:
            _write_json(config.__dict__, config, "sampling.json")
        elif args.mode == MODE_MEASURE:
            _write_json(config.__dict__, config, "measure.json")
        if args.mode == MODE_TRAIN_MEASURE:
            _write_json(config.__dict__, config, "measure.json")
    if config.ckpt_path is None:
        config.ckpt_path = os.path.join(config.output_dir, config.ckpt_dir)
        config.data_ckpt_path = os.path.join(config.output_dir, config.data_ckpt_dir)
        if write and rank == 0:
            os.makedirs(config.ckpt_path, exist_ok=True)
    return config


# ---------------------------------------------------------------------------------------------------- runtime
def _dist():
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = local % max(1, torch.cuda.device_count())   # BD_DIST_BACKEND=gloo: several ranks may share one GPU (functional tests)
        torch.cuda.set_device(local)
        backend = os.environ.get("BD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def get_data_loader(config, device):
    n_img = int(os.environ["BD_NUM_IMAGES"]) if os.environ.get("BD_NUM_IMAGES") else None   # synthetic-dataset size (tests)
    dsl = DatasetLoader(root=config.dataset_path if os.path.isdir(config.dataset_path) else None, name=config.dataset,
                        batch_size=config.batch, seed=config.seed, device=device, num_images=n_img)
    dsl.set_poison(trigger_type=config.trigger, target_type=config.target, clean_rate=config.clean_rate,
                   poison_rate=config.poison_rate).prepare_dataset(mode=config.dataset_load_mode)
    print(f"datasetloader len: {len(dsl)} ({dsl.source})")
    return dsl


def get_model_sched(config, device):
    from baddiffusion_amd.model import DiffuserModelSched, _resolve_local
    if config.ckpt is not None:
        if config.mode in (MODE_SAMPLING, MODE_MEASURE, MODE_RESUME):
            path = config.output_dir
            if config.sample_ep is not None:
                path = os.path.join(config.output_dir, config.ep_model_dir, f"ep{config.sample_ep}")   # intent of :311-313
            base = config.__dict__.get("_base_ckpt", None) or path
            model, noise_sched, get_pipeline = DiffuserModelSched.get_trained(ckpt=base, clip_sample=config.clip,
                                                                             noise_sched_type=config.sched)
        else:
            model, noise_sched, get_pipeline = DiffuserModelSched.get_pretrained(ckpt=config.ckpt, clip_sample=config.clip,
                                                                                noise_sched_type=config.sched)
    else:
        model, noise_sched, get_pipeline = DiffuserModelSched.get_model_sched(image_size=config.image_size, channels=config.channel,
                                                                             model_type=DiffuserModelSched.MODEL_DEFAULT,
                                                                             noise_sched_type=config.sched, clip_sample=config.clip)
    return model.to(device), noise_sched, get_pipeline


def make_grid(images, rows, cols):
    from PIL import Image
    w, h = images[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, image in enumerate(images):
        grid.paste(image, box=(i % cols * w, i // cols * h))
    return grid


def sampling(config, file_name, pipeline, dsl):
    """baddiffusion.py:366-419: 16 clean + 16 backdoor samples as 4x4 grids (plus the t0 grids)."""
    from PIL import Image

    def gen(init, folder):
        test_dir = os.path.join(config.output_dir, folder)
        os.makedirs(test_dir, exist_ok=True)
        res = pipeline(batch_size=config.eval_sample_n, generator=torch.manual_seed(config.seed), init=init, output_type=None,
                       save_every_step=True)
        to_pil = lambda arr: [Image.fromarray(im) for im in np.squeeze((arr * 255).round().astype("uint8"))]
        clip_opt = "" if config.clip else "_noclip"
        name = f"{file_name:04d}" if isinstance(file_name, int) else str(file_name)
        make_grid(to_pil(res.images), rows=4, cols=4).save(f"{test_dir}/{name}{clip_opt}.png")
        make_grid(to_pil(res.movie[0]), rows=4, cols=4).save(f"{test_dir}/{name}{clip_opt}_sample_t0.png")

    with torch.no_grad():
        s = pipeline.unet.sample_size
        noise = torch.randn((config.eval_sample_n, pipeline.unet.in_channels, s, s), generator=torch.manual_seed(config.seed))
        gen(noise, "samples")
        gen(noise + dsl.trigger.unsqueeze(0), "backdoor_samples")       # raw trigger incl. its -1 background (:417)


FID_UNAVAILABLE = ("pytorch_fid's InceptionV3 pool3 weights (pt_inception-2015-12-05-6726825d.pth) are a third-party asset that is not in "
                   "the reference repository and cannot be fetched here; point BD_FID_WEIGHTS at that file and measure() computes FID with "
                   "baddiffusion_amd.inception.FIDInceptionV3 (HIP kernels) + metrics.fid_from_features")


def _png_dir_u8(path, channel):
    """every PNG of `path` (numeric file names, in order) as one uint8 [N, H, W, 3] array: what fid_score.py:84-89 feeds the
    Inception network (PIL -> RGB; a grey image is replicated over the three channels)"""
    from PIL import Image
    files = sorted(os.listdir(path), key=lambda n: int(os.path.splitext(n)[0]))
    return np.stack([np.asarray(Image.open(os.path.join(path, f)).convert("RGB"), dtype=np.uint8) for f in files])


def fid_of_dirs(net, real_u8, gen_u8, batch=500):
    """fid_score.py:233-259 on device-resident uint8 image sets: pool3 features batch by batch -> fp64 statistics on the device
    -> Frechet distance."""
    from baddiffusion_amd import metrics

    def feats(arr):
        for s in range(0, arr.shape[0], batch):
            yield net(arr[s:s + batch])
    return metrics.fid_from_features(feats(real_u8), feats(gen_u8))


def update_score_file(config, score_file, fid_sc, mse_sc, ssim_sc, fid_reason=None, extra=None):
    def key(k):
        r = f"{k}_ep{config.sample_ep}" if config.sample_ep is not None else k
        return r + ("_noclip" if not config.clip else "")
    path = os.path.join(config.output_dir, score_file)
    sc = {}
    if os.path.exists(path):
        with open(path) as f:
            sc = json.load(f)
    for k, v in (("FID", fid_sc), ("MSE", mse_sc), ("SSIM", ssim_sc)):
        if v is not None or key(k) not in sc:
            sc[key(k)] = v
    if sc.get(key("FID")) is None:
        sc[key("FID_reason")] = fid_reason or FID_UNAVAILABLE      # a null FID is never silent
    else:
        sc.pop(key("FID_reason"), None)
        sc[key("FID_real_set")] = ("first measure_sample_n rows of np.random.default_rng(seed).permutation(N) (HF datasets' shuffle rule) over this "
                                   "loader's row order, unflipped uint8 images; upstream's row order comes from an unseeded train_test_split and its "
                                   "PNGs carry RandomHorizontalFlip, so the subset differs from upstream's")
    for k, v in (extra or {}).items():
        sc[key(k)] = v
    with open(path, "w") as f:
        json.dump(sc, f, indent=2, sort_keys=True)
    return sc


def measure(config, dsl, folder_name, pipeline, rank=0, world=1):
    """baddiffusion.py:477-551.  Sampling is sharded over ranks (independent chains, no collective); rank 0 scores.
    MSE / SSIM to the target are computed on the device from the written PNGs like the reference; FID needs
    pytorch_fid's Inception weights, absent here -> reported as null (statistics + Frechet distance: baddiffusion_amd/metrics.py)."""
    from baddiffusion_amd.model import batch_sampling_save
    # per-step DDPM noise: one stream per rank (seed + rank).  With a shared seed every shard would consume the SAME
    # noise sequence and chain j of each shard would be correlated with chain j of the others.  world == 1 keeps the
    # reference's stream exactly; a sharded run is a different (equally valid) draw of the same distribution.
    # BD_SHARDED_NOISE=reference (round 6, SURVEY hard-part 4): every rank walks the single process's chunks with the SAME stream (seed, no
    # rank offset) and takes its rows of every draw -- the reference's noise, chain for chain, at the price of drawing all of it on every rank.
    parity = world > 1 and os.environ.get("BD_SHARDED_NOISE", "rank") == "reference"
    rng = torch.Generator().manual_seed(config.seed + (0 if parity else rank))
    if hasattr(pipeline.unet, "chunks_used"):
        pipeline.unet.chunks_used.clear()
    parts = [config.output_dir, folder_name] + ([f"ep{config.sample_ep}"] if config.sample_ep is not None else [])
    suffix = "_noclip" if not config.clip else ""
    clean_path, backdoor_path = os.path.join(*parts, "clean" + suffix), os.path.join(*parts, "backdoor" + suffix)
    s = pipeline.unet.sample_size
    noise = torch.randn((config.measure_sample_n, pipeline.unet.in_channels, s, s), generator=torch.manual_seed(config.seed))
    backdoor_noise = noise + dsl.trigger.unsqueeze(0)
    batch_sampling_save(config.measure_sample_n, pipeline, clean_path, init=noise, max_batch_n=config.eval_max_batch, rng=rng,
                        rank=rank, world=world, parity=parity)
    batch_sampling_save(config.measure_sample_n, pipeline, backdoor_path, init=backdoor_noise, max_batch_n=config.eval_max_batch,
                        rng=rng, rank=rank, world=world, parity=parity)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if rank != 0:
        return None
    from PIL import Image
    files = sorted(os.listdir(backdoor_path), key=lambda n: int(os.path.splitext(n)[0]))
    gen = np.stack([np.asarray(Image.open(os.path.join(backdoor_path, f)).convert("RGB" if dsl.channel == 3 else "L"), dtype=np.float32) / 255.0
                    for f in files])
    gen = torch.from_numpy(gen)
    gen = gen.permute(0, 3, 1, 2) if gen.dim() == 4 else gen[:, None]
    tgt = (dsl.target / 2 + 0.5).clamp(0, 1)[None].expand_as(gen)
    from baddiffusion_amd import metrics
    dev = pipeline.device if hasattr(pipeline, "device") else "cpu"
    gen_d, tgt_d = gen.to(dev), tgt.to(dev)
    mse_sc = metrics.mse(gen_d, tgt_d)
    ssim_sc = metrics.ssim(gen_d, tgt_d, data_range=1.0)     # torchmetrics defaults (parity unpinned: torchmetrics absent)
    # FID (baddiffusion.py:533: fid(path=[dataset_img_dir, clean_path])): pool3 features of pytorch_fid's InceptionV3 for the first
    # measure_sample_n images of the seed-shuffled dataset and for the clean samples, on the device (baddiffusion_amd/inception.py),
    # then fp64 statistics + the Frechet distance (pinned by G8).  The weights file is a third-party asset: without it FID is null
    # and score.json says why.
    fid_sc, fid_reason = None, FID_UNAVAILABLE
    from baddiffusion_amd.inception import load_fid_weights
    net = load_fid_weights(device=dev) if str(dev) != "cpu" else None
    if net is not None:
        n_real = min(config.measure_sample_n, len(dsl))
        # The reference scores against `get_dataset().shuffle(seed=config.seed)[:n]` (:488, 503): HF datasets' shuffle is
        # np.random.default_rng(seed).permutation(len) -- reproduced here -- over ITS row order, which in FIXED mode is the concatenation of an
        # UNSEEDED train_test_split (SURVEY D-5), and it saves the TRANSFORMED images (resize + RandomHorizontalFlip) as PNGs.  So the permutation
        # rule is the reference's, but the subset and the flips are not reproducible from upstream: FID values are comparable in distribution,
        # not digit for digit (score.json: FID_real_set).
        order = torch.from_numpy(np.random.default_rng(config.seed).permutation(len(dsl))[:n_real].astype(np.int64))
        real = dsl.device_images[dsl._rows()[order].to(dsl.device_images.device)]
        if real.shape[-1] == 1:
            real = real.expand(-1, -1, -1, 3).contiguous()
        clean = torch.from_numpy(_png_dir_u8(clean_path, dsl.channel)).to(dev)
        fid_sc = float(fid_of_dirs(net, real.to(dev), clean))
        fid_reason = None
    print(f"[{config.sample_ep}] FID: {fid_sc if fid_sc is not None else 'None (BD_FID_WEIGHTS not set: pytorch_fid Inception weights unavailable)'}, "
          f"MSE: {mse_sc}, SSIM: {ssim_sc}")
    # the inference batch decides which kernels the plan picks (fp32 summation order): EVERY batch size a forward of this measure() ran with is
    # recorded with the scores it produced (`last_chunk` alone is only the tail call, e.g. 8 of 32 = 12 + 12 + 8)
    used = sorted(getattr(pipeline.unet, "chunks_used", ()), reverse=True)
    return update_score_file(config, "score.json", fid_sc, mse_sc, ssim_sc, fid_reason=fid_reason,
                             extra={"inference_chunk": used[0] if len(used) == 1 else (used or None)})


def checkpoint(config, engine, pipeline, cur_epoch, cur_step):
    """baddiffusion.py:558-570: optimizer state + {epoch, step} + the diffusers-layout pipeline."""
    os.makedirs(config.ckpt_path, exist_ok=True)
    torch.save({"m": engine.m.cpu(), "v": engine.v.cpu(), "opt_step": engine.opt_step, "micro": engine.micro},
               os.path.join(config.ckpt_path, "optimizer.bin"))
    # + the random-number state, so that a resumed run draws the noise / timesteps the uninterrupted run would have drawn
    torch.save({"epoch": cur_epoch, "step": cur_step, "world": _world_size(), "cpu_rng": torch.get_rng_state(),
                "cuda_rng": torch.cuda.get_rng_state() if torch.cuda.is_available() else None}, config.data_ckpt_path)
    pipeline.save_pretrained(config.output_dir)
    if config.is_save_all_model_epochs:
        pipeline.save_pretrained(os.path.join(config.output_dir, config.ep_model_dir, f"ep{cur_epoch}"))


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def capture_rank_rng(cur_epoch, cur_step):
    """This rank's random-number state, tagged with the checkpoint it belongs to ({epoch, step, world})."""
    return {"epoch": cur_epoch, "step": cur_step, "world": _world_size(), "cpu_rng": torch.get_rng_state(),
            "cuda_rng": torch.cuda.get_rng_state() if torch.cuda.is_available() else None}


def save_rank_rng(config, rank, state):
    """ranks > 0: their own random-number state next to rank 0's data.ckpt (every rank draws its own noise / timesteps, so a
    resumed multi-rank run needs one state per rank; rank 0's lives in data.ckpt itself, written by checkpoint()).  The file
    carries the {epoch, step, world} of the checkpoint it belongs to: restore_training_state() refuses a state from another
    checkpoint (a crash between the two writes, ranks failing at different steps, a stale file of another world size)."""
    if rank == 0:
        return
    torch.save(state, f"{config.data_ckpt_path}.rank{rank}")


def restore_training_state(config, engine, rank=0):
    """baddiffusion.py:336-342 (accelerator.load_state): Adam moments, optimizer step counters and the RNG state written by
    checkpoint(); the weights themselves come from the diffusers-layout directory (get_trained).  Returns (epoch, step).
    The random-number state is per rank (data.ckpt for rank 0, data.ckpt.rank<r> otherwise): a resumed run continues
    bit-identically for any world size that matches the checkpointed one; a rank without a state file of its own (the run was
    checkpointed with fewer ranks) is re-seeded with seed + rank and says so."""
    opt_file = os.path.join(config.ckpt_path, "optimizer.bin")
    if os.path.exists(opt_file):
        st = torch.load(opt_file, map_location="cpu")
        engine.m.copy_(st["m"]); engine.v.copy_(st["v"]); engine.opt_step = st["opt_step"]; engine.micro = st["micro"]
        engine.sync_state()
    epoch = step = 0
    if os.path.exists(config.data_ckpt_path):
        st = torch.load(config.data_ckpt_path, map_location="cpu")
        epoch, step = st["epoch"], st["step"]
        if rank > 0:
            own = f"{config.data_ckpt_path}.rank{rank}"
            tag = (epoch, step, st.get("world", _world_size()))
            mine = torch.load(own, map_location="cpu") if os.path.exists(own) else None
            if mine is not None and (mine.get("epoch"), mine.get("step"), mine.get("world")) == tag and tag[2] == _world_size():
                st = mine
            else:
                why = "no RNG state of its own in the checkpoint" if mine is None else \
                    f"its RNG state file belongs to another checkpoint ({mine.get('epoch')}, {mine.get('step')}, world {mine.get('world')}) " \
                    f"than data.ckpt {tag}"
                print(f"[rank {rank}] {why}: re-seeding with seed + rank (not bit-identical)")
                torch.manual_seed(config.seed + rank)
                st = {}
        if st.get("cpu_rng") is not None:
            torch.set_rng_state(st["cpu_rng"])
        if st.get("cuda_rng") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(st["cuda_rng"])
    return epoch, step


def train_loop(config, model, noise_sched, get_pipeline, dsl, device, world, rank, start_epoch=0, start_step=0):
    """baddiffusion.py:572-645 with the loop body replaced by the fused engine step."""
    from baddiffusion_amd.trainer import TrainEngine
    num_batch = (len(dsl) + config.batch * world - 1) // (config.batch * world)
    engine = TrainEngine(model, noise_sched, lr=config.learning_rate, lr_warmup_steps=config.lr_warmup_steps,
                         num_training_steps=num_batch * config.epoch // config.gradient_accumulation_steps,
                         grad_accum_steps=config.gradient_accumulation_steps)
    if config.mode == MODE_RESUME:
        restore_training_state(config, engine, rank)
    dsl.to_device(device)
    trigger, target = dsl.trigger.to(device), dsl.target.to(device)
    log = open(os.path.join(config.output_dir, "log.jsonl"), "a") if rank == 0 else None
    cur_step = start_step
    epoch = start_epoch
    saved_at = None
    try:
        for epoch in range(int(start_epoch), int(config.epoch)):
            t0 = time.time()
            for step, (rows, flips, pois) in enumerate(dsl.device_batch_rows(shuffle=True, epoch=epoch, rank=rank, world=world)):
                bs = rows.shape[0]
                noise = torch.randn((bs, dsl.channel, dsl.image_size, dsl.image_size), device=device)       # :596
                timesteps = torch.randint(0, noise_sched.config.num_train_timesteps, (bs,), device=device).long()   # :600
                loss = engine.train_step(dsl.device_images, pois, trigger, target, noise, timesteps, row_index=rows, flip=flips)
                cur_step += 1
                if log is not None and step % 50 == 0:
                    rec = {"loss": float(loss), "lr": engine.current_lr(), "epoch": epoch, "step": cur_step}
                    log.write(json.dumps(rec) + "\n"); log.flush()
            ckpt_now = (epoch + 1) % config.save_model_epochs == 0 or epoch == config.epoch - 1
            rng_now = capture_rank_rng(epoch, cur_step) if (rank > 0 and ckpt_now) else None
            if rank == 0:
                print(f"epoch {epoch}: {time.time() - t0:.1f} s, loss {float(loss):.5f}")
                pipeline = get_pipeline(unet=model, scheduler=noise_sched)
                if (epoch + 1) % config.save_image_epochs == 0 or epoch == config.epoch - 1:
                    sampling(config, epoch, pipeline, dsl)
                if ckpt_now:
                    checkpoint(config, engine, pipeline, epoch, cur_step)
            if ckpt_now and world > 1:
                # rank files are written AFTER rank 0's checkpoint exists (rank 0 samples for minutes first): a crash in between
                # leaves the previous, mutually consistent set; the tags catch whatever is left inconsistent
                import torch.distributed as dist
                dist.barrier()
                save_rank_rng(config, rank, rng_now)
            saved_at = (epoch, cur_step) if ckpt_now else saved_at
    except Exception:
        traceback.print_exc()          # the reference swallows the exception too (:635-637) but we re-raise below
        raise
    finally:
        if saved_at != (epoch, cur_step):      # (the regular end-of-run checkpoint above already holds exactly this state)
            if rank == 0:
                pipeline = get_pipeline(unet=model, scheduler=noise_sched)
                checkpoint(config, engine, pipeline, epoch, cur_step)
            else:       # no barrier on the error path (a failed rank would never arrive): the tags decide on restore
                save_rank_rng(config, rank, capture_rank_rng(epoch, cur_step))
        if log is not None:
            log.close()
        engine.close()
    return get_pipeline(unet=model, scheduler=noise_sched)


def main(argv=None):
    config = setup(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("baddiffusion.py needs an AMD GPU: the hot path runs in libbd_hip.so and has no CPU fallback")
    world, rank, local = _dist()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dsl = get_data_loader(config, device)
    config.image_size, config.channel = dsl.image_size, dsl.channel
    model, noise_sched, get_pipeline = get_model_sched(config, device)
    config.pretrained = bool(getattr(model, "pretrained", True))   # False: topology built with default init (BD_ALLOW_RANDOM_INIT)
    if rank == 0 and config.mode in (MODE_TRAIN, MODE_TRAIN_MEASURE) and os.path.isdir(config.output_dir):
        _write_json(config.__dict__, config, "config.json")
    if config.mode in (MODE_TRAIN, MODE_RESUME, MODE_TRAIN_MEASURE):
        start_epoch = start_step = 0
        if config.mode == MODE_RESUME and os.path.exists(config.data_ckpt_path):
            st = torch.load(config.data_ckpt_path)
            start_epoch, start_step = st["epoch"], st["step"]
        pipeline = train_loop(config, model, noise_sched, get_pipeline, dsl, device, world, rank, start_epoch, start_step)
        if config.mode == MODE_TRAIN_MEASURE:
            measure(config, dsl, "measure", pipeline, rank, world)
    elif config.mode == MODE_SAMPLING:
        pipeline = get_pipeline(unet=model, scheduler=noise_sched)
        if rank == 0:
            sampling(config, "final" if config.sample_ep is None else f"final_ep{config.sample_ep}", pipeline, dsl)
    elif config.mode == MODE_MEASURE:
        measure(config, dsl, "measure", get_pipeline(unet=model, scheduler=noise_sched), rank, world)
    else:
        raise NotImplementedError()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
