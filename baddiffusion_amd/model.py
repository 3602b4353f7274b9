"""DiffuserModelSched + batch sampling helpers -- drop-in for /root/reference/model.py:466-729.

`get_pretrained / get_trained / get_model_sched` return `(model, noise_sched, get_pipeline)` like the
reference; checkpoints are read from a LOCAL diffusers-layout directory (no network here):
    <dir>/model_index.json, <dir>/unet/config.json, <dir>/unet/diffusion_pytorch_model.{bin,safetensors},
    <dir>/scheduler/scheduler_config.json            (utils/constants.py:22-26, scheduling_utils.py:25)
Hub ids ("google/ddpm-cifar10-32") are resolved against $BD_CKPT_ROOT / $HF_HOME-style local folders; when the
weights are not available the documented topology is built with default init and `model.pretrained = False`.
"""
import json
import os
from functools import partial
from typing import Union

import numpy as np
import torch

from .pipelines import DDIMPipeline, DDPMPipeline, PNDMPipeline
from .schedulers import DDIMScheduler, DDPMScheduler, PNDMScheduler, SchedulerConfigCarrier
from .unet import UNet2DModel

WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"
SCHEDULER_CONFIG_NAME = "scheduler_config.json"

# architecture kwargs of the hub checkpoints the reference uses (validated by parameter counts)
KNOWN_TOPOLOGIES = {
    "google/ddpm-cifar10-32": dict(
        sample_size=32, in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 256),
        down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"),
        up_block_types=("UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"), layers_per_block=2,
        downsample_padding=0, flip_sin_to_cos=False, freq_shift=1, norm_eps=1e-6, attention_head_dim=None),
    "google/ddpm-ema-celebahq-256": dict(
        sample_size=256, in_channels=3, out_channels=3, block_out_channels=(128, 128, 256, 256, 512, 512),
        down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
        up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4, layers_per_block=2,
        downsample_padding=0, flip_sin_to_cos=False, freq_shift=1, norm_eps=1e-6, attention_head_dim=None),
}
KNOWN_TOPOLOGIES["google/ddpm-ema-church-256"] = KNOWN_TOPOLOGIES["google/ddpm-ema-celebahq-256"]
KNOWN_TOPOLOGIES["google/ddpm-ema-bedroom-256"] = KNOWN_TOPOLOGIES["google/ddpm-ema-celebahq-256"]
KNOWN_SCHEDULERS = {   # hub scheduler configs (verify when assets are available)
    "google/ddpm-cifar10-32": dict(variance_type="fixed_large", clip_sample=True),
    "google/ddpm-ema-celebahq-256": dict(variance_type="fixed_small", clip_sample=True),
}


# ------------------------------------------------------------------------------------------------ I/O (f-2)
def save_unet(unet, directory):
    os.makedirs(directory, exist_ok=True)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in unet.config_dict().items()}
    with open(os.path.join(directory, CONFIG_NAME), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    sd = {k: v.cpu() for k, v in unet.state_dict().items()}
    torch.save(sd, os.path.join(directory, WEIGHTS_NAME))


def load_unet(directory, device=None):
    with open(os.path.join(directory, CONFIG_NAME)) as f:
        cfg = json.load(f)
    cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    unet = UNet2DModel(**cfg)
    st = os.path.join(directory, SAFETENSORS_WEIGHTS_NAME)
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(directory, WEIGHTS_NAME), map_location="cpu")
    unet.load_state_dict(sd)
    unet.pretrained = True
    return unet.to(device) if device is not None else unet


def save_scheduler(sched, directory):
    os.makedirs(directory, exist_ok=True)
    cfg = dict(sched.config)
    cfg["_class_name"] = type(sched).__name__
    cfg["_diffusers_version"] = "0.16.0.dev0"
    with open(os.path.join(directory, SCHEDULER_CONFIG_NAME), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)


def load_scheduler(directory):
    with open(os.path.join(directory, SCHEDULER_CONFIG_NAME)) as f:
        cfg = json.load(f)
    cls = {"DDPMScheduler": DDPMScheduler, "DDIMScheduler": DDIMScheduler}.get(cfg.get("_class_name", "DDPMScheduler"))
    if cls is None:
        raise NotImplementedError(f"scheduler {cfg.get('_class_name')} is not on the BadDiffusion hot path")
    return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})


def _resolve_local(ckpt_id):
    """A diffusers-layout directory for `ckpt_id`, or None."""
    cands = [ckpt_id]
    for root in (os.environ.get("BD_CKPT_ROOT"), "checkpoints", os.path.expanduser("~/.cache/baddiffusion")):
        if root:
            cands += [os.path.join(root, ckpt_id), os.path.join(root, ckpt_id.replace("/", "--"))]
    for c in cands:
        if c and os.path.isdir(c) and os.path.exists(os.path.join(c, "unet", CONFIG_NAME)):
            return c
    return None


# ------------------------------------------------------------------------------------------------ sampling helpers
def batch_sampling(sample_n: int, pipeline, init: torch.Tensor = None, max_batch_n: int = 256, rng: torch.Generator = None):
    # model.py:469-490
    if init is None:
        if sample_n > max_batch_n:
            replica, residual = sample_n // max_batch_n, sample_n % max_batch_n
            batch_sizes = [max_batch_n] * replica + ([residual] if residual > 0 else [])
        else:
            batch_sizes = [sample_n]
        inits = [None] * len(batch_sizes)
    else:
        inits = torch.split(init, max_batch_n)
        batch_sizes = [len(x) for x in inits]
    out = []
    for i, bs in enumerate(batch_sizes):
        out.append(pipeline(batch_size=bs, generator=rng, init=inits[i], output_type=None).images)
    return np.concatenate(out)


def save_imgs(imgs: np.ndarray, file_dir: Union[str, os.PathLike], file_name: Union[str, os.PathLike] = "", start_cnt: int = 0) -> None:
    # model.py:496-502 : PNG = round(255 x) uint8
    from PIL import Image
    os.makedirs(file_dir, exist_ok=True)
    arr = np.squeeze((imgs * 255).round().astype("uint8"))
    if arr.ndim == 3 and imgs.shape[0] == 1:
        arr = arr[None]
    for i, image in enumerate(arr):
        Image.fromarray(image).save(os.path.join(file_dir, f"{file_name}{start_cnt + i}.png"))


def batch_sampling_save(sample_n: int, pipeline, path: Union[str, os.PathLike], init: torch.Tensor = None, max_batch_n: int = 256,
                        rng: torch.Generator = None, rank: int = 0, world: int = 1, parity: bool = False):
    """model.py:504-529.  With world > 1 the rows of `init` are split contiguously over ranks and every rank writes
    its own PNG index range (sampling chains are independent, no collective).

    parity=True (round 6, SURVEY hard-part 4): REFERENCE-ORDER noise on a sharded job.  The chunks are the single process's (max_batch_n
    over the whole job), every rank walks all of them with the SAME CPU generator `rng`, and inside a chunk rank r samples rows
    [r * ceil(bs / world), ...) through a ShardedGenerator: each draw produces the full chunk's tensor and keeps this rank's rows, so the
    stream advances exactly as in the unsharded run and chain j sees the noise it would see there -- the PNGs equal the world-1 run's up
    to the fp32 summation order the plan picks for another batch size (<= 1 of 255 levels).  init=None works too (the initial sample is a
    draw like any other).  Price: every rank draws all of the job's noise on its host."""
    from .schedulers import ShardedGenerator
    if parity and world > 1:
        if rng is None or rng.device.type != "cpu":
            raise ValueError("parity=True needs the shared CPU generator of the unsharded run as `rng`")
        if init is None:
            replica, residual = sample_n // max_batch_n, sample_n % max_batch_n
            sizes = ([max_batch_n] * replica + ([residual] if residual > 0 else [])) if sample_n > max_batch_n else [sample_n]
            chunks = [None] * len(sizes)
        else:
            chunks = list(torch.split(init, max_batch_n))
            sizes = [len(x) for x in chunks]
        jobs, start = [], 0
        for ch, bs in zip(chunks, sizes):
            per = (bs + world - 1) // world
            lo, hi = min(bs, rank * per), min(bs, (rank + 1) * per)
            jobs.append((hi - lo, ShardedGenerator(rng, bs, lo, hi) if hi > lo else None, None if ch is None else ch[lo:hi], start + lo,
                         bs, ch is not None))
            start += bs
    else:
        jobs = None
    if jobs is not None:
        pass
    elif init is None:
        if sample_n > max_batch_n:
            replica, residual = sample_n // max_batch_n, sample_n % max_batch_n
            batch_sizes = [max_batch_n] * replica + ([residual] if residual > 0 else [])
        else:
            batch_sizes = [sample_n]
        inits = [None] * len(batch_sizes)
        if world > 1:
            raise ValueError("sharded sampling needs an explicit `init` (one generator stream cannot be split) or parity=True")
        offset = 0
    else:
        n = len(init)
        per = (n + world - 1) // world
        offset = rank * per
        init = init[offset: offset + per]
        inits = torch.split(init, max_batch_n)
        batch_sizes = [len(x) for x in inits]
    if jobs is None:
        jobs, cnt0 = [], offset
        for ch, bs in zip(inits, batch_sizes):
            jobs.append((bs, rng, ch, cnt0, bs, ch is not None))
            cnt0 += bs
    # Overlapped save (f-3): the pipeline hands back DEVICE uint8 images (output_type="u8": bd_to_image quantises on the
    # GPU); they go to pinned host memory on a copy stream and are PNG-encoded by a thread pool while the next chunk is
    # being sampled.  A pipeline that returns host arrays (float [0,1]) keeps the reference's serial save_imgs path.
    from concurrent.futures import ThreadPoolExecutor
    pending = []
    copy_stream = torch.cuda.Stream() if torch.cuda.is_available() else None
    # decided up front (not by catching exceptions around the sampling call: an error raised mid-chain must surface, and a retry
    # would re-sample with an rng that has already advanced)
    device_u8 = bool(getattr(pipeline, "supports_u8", False)) and copy_stream is not None

    def _write(host, ev, start):
        from PIL import Image
        ev.synchronize()
        arr = host.numpy()
        os.makedirs(path, exist_ok=True)
        for k in range(arr.shape[0]):
            img = arr[k]
            Image.fromarray(img[..., 0] if img.shape[-1] == 1 else img).save(os.path.join(path, f"{start + k}.png"))

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        for bs, gen, ch, cnt, full_bs, init_given in jobs:
            if bs == 0:       # parity mode, a chunk with fewer rows than ranks: this rank samples nothing of it but keeps the stream in step
                adv = getattr(pipeline, "advance_generator", None)
                if adv is None:
                    raise TypeError("parity=True with a chunk smaller than the world size needs pipeline.advance_generator (the pipelines of "
                                    "baddiffusion_amd.pipelines have it; a bare callable does not)")
                adv(full_bs, rng, init_given)
                continue
            res = pipeline(batch_size=bs, generator=gen, init=ch, output_type="u8" if device_u8 else None)
            imgs = res.images
            if device_u8:
                if not (torch.is_tensor(imgs) and imgs.is_cuda and imgs.dtype == torch.uint8):
                    raise TypeError("pipeline.supports_u8 is set but output_type='u8' did not return device uint8 images")
                host = torch.empty(imgs.shape, dtype=torch.uint8, pin_memory=True)
                copy_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(copy_stream):
                    host.copy_(imgs, non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(copy_stream)
                imgs.record_stream(copy_stream)
                # split the chunk over the pool: PNG encoding is the slow part (zlib), one task per 64 images
                for s0 in range(0, bs, 64):
                    pending.append(pool.submit(_write, host[s0: s0 + 64], ev, cnt + s0))
            else:
                save_imgs(imgs=imgs, file_dir=path, file_name="", start_cnt=cnt)
            del res
        for f in pending:
            f.result()
    return None


class DiffuserModelSched:
    CLIP_SAMPLE_DEFAULT = False
    MODEL_DEFAULT = "DEFAULT"
    DDPM_CIFAR10_DEFAULT = "DDPM-CIFAR10-DEFAULT"
    DDPM_CELEBA_HQ_DEFAULT = "DDPM-CELEBA-HQ-DEFAULT"
    DDPM_CHURCH_DEFAULT = "DDPM-CHURCH-DEFAULT"
    DDPM_BEDROOM_DEFAULT = "DDPM-BEDROOM-DEFAULT"
    LDM_CELEBA_HQ_DEFAULT = "LDM-CELEBA-HQ-DEFAULT"
    DDPM_CIFAR10_32 = "DDPM-CIFAR10-32"
    DDPM_CELEBA_HQ_256 = "DDPM-CELEBA-HQ-256"
    DDPM_CHURCH_256 = "DDPM-CHURCH-256"
    DDPM_BEDROOM_256 = "DDPM-BEDROOM-256"
    LDM_CELEBA_HQ_256 = "LDM-CELEBA-HQ-256"
    DDPM_SCHED = "DDPM-SCHED"
    DDIM_SCHED = "DDIM-SCHED"
    # model.py:598-630: these build their own scheduler object but sample through PNDMPipeline, which converts it to a
    # PNDMScheduler (pipeline_pndm.py:46) -- value = the reference class the returned config stands for
    _PNDM_SCHEDS = {"DPM_SOLVER_PP_O1-SCHED": "DPMSolverMultistepScheduler", "DPM_SOLVER_O1-SCHED": "DPMSolverMultistepScheduler",
                    "DPM_SOLVER_PP_O2-SCHED": "DPMSolverMultistepScheduler", "DPM_SOLVER_O2-SCHED": "DPMSolverMultistepScheduler",
                    "DPM_SOLVER_PP_O3-SCHED": "DPMSolverMultistepScheduler", "DPM_SOLVER_O3-SCHED": "DPMSolverMultistepScheduler",
                    "UNIPC-SCHED": "UniPCMultistepScheduler", "PNDM-SCHED": "PNDMScheduler", "DEIS-SCHED": "DEISMultistepScheduler",
                    "HEUN-SCHED": "HeunDiscreteScheduler", "LMSD-SCHED": "LMSDiscreteScheduler"}
    # named in model.py:556-563 but not handled by its __get_model_sched either (it raises NotImplementedError)
    _OTHER_SCHEDS = ("LDM-SCHED", "SCORE-SDE-VE-SCHED", "EDM-VE-SCHED", "EDM-VE-ODE-SCHED", "EDM-VE-SDE-SCHED")
    _HUB = {DDPM_CIFAR10_32: "google/ddpm-cifar10-32", DDPM_CELEBA_HQ_256: "google/ddpm-ema-celebahq-256",
            DDPM_CHURCH_256: "google/ddpm-ema-church-256", DDPM_BEDROOM_256: "google/ddpm-ema-bedroom-256",
            LDM_CELEBA_HQ_256: "CompVis/ldm-celebahq-256"}

    @staticmethod
    def get_sample_clip(clip_sample: bool, clip_sample_default: bool):
        return clip_sample if clip_sample is not None else clip_sample_default

    @staticmethod
    def _pipeline_generator(pipeline):
        def get_pipeline(unet, scheduler):
            return pipeline(unet, scheduler)
        return get_pipeline

    @staticmethod
    def _get_model_sched(ckpt_id: str, clip_sample: bool, noise_sched_type: str = None, allow_random_init: bool = False):
        # model.py:577-643
        clip_used = DiffuserModelSched.get_sample_clip(clip_sample, DiffuserModelSched.CLIP_SAMPLE_DEFAULT)
        local = _resolve_local(ckpt_id)
        if local is not None:
            model = load_unet(os.path.join(local, "unet"))
            ckpt_sched = load_scheduler(os.path.join(local, "scheduler")) if os.path.isdir(os.path.join(local, "scheduler")) \
                else DDPMScheduler()
        elif ckpt_id in KNOWN_TOPOLOGIES:
            # The reference fails in from_pretrained when the weights are missing; fine-tuning a backdoor from a
            # random network with the fine-tune LR is a different experiment, so it needs an explicit opt-in.
            if not (allow_random_init or os.environ.get("BD_ALLOW_RANDOM_INIT", "0") == "1"):
                raise FileNotFoundError(
                    f"pretrained weights for '{ckpt_id}' are not available locally (put the diffusers-layout checkpoint "
                    f"under $BD_CKPT_ROOT).  Set BD_ALLOW_RANDOM_INIT=1 to build the documented topology with default "
                    f"initialisation instead (recorded as pretrained=False).")
            print(f"[baddiffusion_amd] weights for '{ckpt_id}' are not available locally (set BD_CKPT_ROOT): "
                  f"building the documented topology with default initialisation (BD_ALLOW_RANDOM_INIT)")
            model = UNet2DModel(**KNOWN_TOPOLOGIES[ckpt_id])
            model.pretrained = False
            ckpt_sched = DDPMScheduler(**KNOWN_SCHEDULERS.get(ckpt_id, {}))
        else:
            raise FileNotFoundError(f"checkpoint '{ckpt_id}' not found locally and its topology is unknown")
        kw = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02)
        if noise_sched_type == DiffuserModelSched.DDPM_SCHED:
            noise_sched = DDPMScheduler(clip_sample=clip_used, **kw)
            get_pipeline = DiffuserModelSched._pipeline_generator(DDPMPipeline)
        elif noise_sched_type == DiffuserModelSched.DDIM_SCHED:
            noise_sched = DDIMScheduler(clip_sample=clip_used, **kw)
            get_pipeline = DiffuserModelSched._pipeline_generator(DDIMPipeline)
        elif noise_sched_type is None:
            noise_sched = ckpt_sched
            get_pipeline = DiffuserModelSched._pipeline_generator(DDPMPipeline)
        elif noise_sched_type in DiffuserModelSched._PNDM_SCHEDS:
            # SURVEY f-4: training uses only the betas of this object; sampling = PNDMPipeline with the post-step clip
            if noise_sched_type == "PNDM-SCHED":
                noise_sched = PNDMScheduler(**kw)
            else:
                noise_sched = SchedulerConfigCarrier(DiffuserModelSched._PNDM_SCHEDS[noise_sched_type], **kw)
            get_pipeline = DiffuserModelSched._pipeline_generator(partial(PNDMPipeline, clip_sample=clip_used))
        elif noise_sched_type in DiffuserModelSched._OTHER_SCHEDS:
            raise NotImplementedError(f"{noise_sched_type}: not handled by the reference's model.py:577-643 either")
        else:
            raise NotImplementedError()
        if clip_used is not None:
            noise_sched.config.clip_sample = clip_used
            print(f"noise_sched.config.clip_sample = {noise_sched.config.clip_sample}")
        return model, noise_sched, get_pipeline

    @staticmethod
    def get_model_sched(image_size: int, channels: int, model_type: str = MODEL_DEFAULT, noise_sched_type: str = None,
                        clip_sample: bool = None, **kwargs):
        # model.py:645-698
        if model_type == DiffuserModelSched.MODEL_DEFAULT:
            clip_used = DiffuserModelSched.get_sample_clip(clip_sample, False)
            noise_sched = DDPMScheduler(num_train_timesteps=1000, clip_sample=clip_used)
            model = UNet2DModel(
                sample_size=image_size, in_channels=channels, out_channels=channels, layers_per_block=2,
                block_out_channels=(128, 128, 256, 256, 512, 512),
                down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D"))
            get_pipeline = DiffuserModelSched._pipeline_generator(DDPMPipeline)
            return model, noise_sched, get_pipeline
        table = {DiffuserModelSched.DDPM_CIFAR10_DEFAULT: DiffuserModelSched.DDPM_CIFAR10_32,
                 DiffuserModelSched.DDPM_CELEBA_HQ_DEFAULT: DiffuserModelSched.DDPM_CELEBA_HQ_256,
                 DiffuserModelSched.DDPM_CHURCH_DEFAULT: DiffuserModelSched.DDPM_CHURCH_256,
                 DiffuserModelSched.DDPM_BEDROOM_DEFAULT: DiffuserModelSched.DDPM_BEDROOM_256,
                 DiffuserModelSched.LDM_CELEBA_HQ_DEFAULT: DiffuserModelSched.LDM_CELEBA_HQ_256}
        if model_type not in table:
            raise NotImplementedError()
        # topology only: the weights are re-initialised right below (model.apply(weight_reset), model.py:647-652)
        model, noise_sched, get_pipeline = DiffuserModelSched._get_model_sched(
            ckpt_id=DiffuserModelSched._HUB.get(table[model_type], table[model_type]), clip_sample=clip_sample,
            noise_sched_type=noise_sched_type, allow_random_init=True)
        model.reset_parameters()
        return model, noise_sched, get_pipeline

    @staticmethod
    def get_pretrained(ckpt: str, clip_sample: bool = None, noise_sched_type: str = None, allow_random_init: bool = False):
        ckpt = DiffuserModelSched._HUB.get(ckpt, ckpt)
        return DiffuserModelSched._get_model_sched(ckpt_id=ckpt, clip_sample=clip_sample, noise_sched_type=noise_sched_type,
                                                   allow_random_init=allow_random_init)

    @staticmethod
    def get_trained(ckpt: str, clip_sample: bool = None, noise_sched_type: str = None):
        return DiffuserModelSched._get_model_sched(ckpt_id=ckpt, clip_sample=clip_sample, noise_sched_type=noise_sched_type)
