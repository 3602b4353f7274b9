"""DDPMPipeline / DDIMPipeline -- drop-in for the two (locally modified) pipelines BadDiffusion samples with
(/root/reference/diffusers/src/diffusers/pipelines/ddpm/pipeline_ddpm.py:46-125,
 pipelines/ddim/pipeline_ddim.py:50-142), including the repo's additions `init=`, `save_every_step=`,
`start_from=` and the `.movie` output.

Loop body = one UNet forward (bd_unet_forward) + one fused scheduler step kernel; the image never leaves the
GPU until the final (x/2+0.5).clamp(0,1) -> NHWC conversion (bd_to_image).  With a CPU `generator` the
per-step noise is drawn on the CPU in the reference's order (seed parity, scheduling_ddpm.py:400-404);
with a CUDA generator or `generator=None` it is drawn on the device (throughput mode).
"""
import json
import os

import numpy as np
import torch

from . import ops
from .schedulers import DDIMScheduler, PNDMScheduler, DDPMScheduler, randn_tensor


class ImagePipelineOutput:
    def __init__(self, images, movie=None):
        self.images = images
        self.movie = movie if movie is not None else []


def _static_weights(call):
    """The sampling loop evaluates the UNet num_inference_steps times over the same parameters: promise that to the plan so it
    prepares the weights (split-plane copy, upsample tap planes) once per loop instead of once per evaluation."""
    import functools

    @functools.wraps(call)
    def wrapped(self, *a, **kw):
        sw = getattr(self.unet, "static_weights", None)
        if sw is None:
            return call(self, *a, **kw)
        with sw():
            return call(self, *a, **kw)
    return wrapped


class _PipelineBase:
    supports_u8 = True      # output_type="u8" returns device uint8 images (batch_sampling_save's overlapped PNG path)

    def __init__(self, unet, scheduler):
        self.unet = unet
        self.scheduler = scheduler
        self._progress = {}

    @property
    def device(self):
        return self.unet.device

    def to(self, device):
        self.unet.to(device)
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    def progress_bar(self, it):
        if self._progress.get("disable", True):
            return it
        try:
            from tqdm.auto import tqdm
            return tqdm(it, **{k: v for k, v in self._progress.items() if k != "disable"})
        except ImportError:
            return it

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [Image.fromarray(image.squeeze(), mode="L") for image in images]
        return [Image.fromarray(image) for image in images]

    def _image_shape(self, batch_size):
        s = self.unet.config.sample_size
        return (batch_size, self.unet.config.in_channels, s, s)

    def _to_numpy(self, image_nhwc_storage, shape):
        """image: NHWC storage [B,H,W,C] -> float32 numpy [B,H,W,C] in [0,1] (pipeline_ddpm.py:115-116)."""
        return ops.to_image(image_nhwc_storage, True, shape).cpu().numpy()

    def _to_u8(self, image_nhwc_storage, shape):
        """device uint8 [B,H,W,C] = round(255 * (x/2+0.5).clamp(0,1)): what the PNG writer needs, without leaving the GPU
        and without a host synchronisation (output_type="u8"; model.py:496-502 quantises the same way on the host)."""
        return ops.to_image(image_nhwc_storage, True, shape, want_u8=True)[1]

    def _start(self, batch_size, generator, init):
        shape = self._image_shape(batch_size)
        if init is None:
            image = randn_tensor(shape, generator=generator, device=self.device)
        else:
            image = init.detach().clone().to(self.device)
            shape = tuple(image.shape)
        # keep the running sample in NHWC storage (what the UNet consumes and produces)
        return ops.nchw_to_nhwc(image.float()), shape

    def noise_draws(self, num_inference_steps=None, start_from=0, eta=0.0, **kwargs):
        """How many chunk-shaped tensors ONE call draws from its generator after the initial sample: what a rank that owns no row of a chunk
        must draw and discard to keep a shared CPU stream in the unsharded order (model.batch_sampling_save(parity=True))."""
        return 0

    def advance_generator(self, batch_size, generator, init_given, **kwargs):
        """draw and discard everything __call__(batch_size, generator=generator, init=None / given, **kwargs) would draw"""
        shape = self._image_shape(batch_size)
        for _ in range((0 if init_given else 1) + self.noise_draws(**kwargs)):
            torch.randn(shape, generator=generator)

    # ---- diffusers-layout save (SURVEY f-2) ------------------------------------------------------------------
    def save_pretrained(self, save_directory):
        from .model import save_unet, save_scheduler
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump({"_class_name": type(self).__name__, "_diffusers_version": "0.16.0.dev0",
                       "scheduler": ["diffusers", type(self.scheduler).__name__], "unet": ["diffusers", "UNet2DModel"]}, f, indent=2)
        save_unet(self.unet, os.path.join(save_directory, "unet"))
        save_scheduler(self.scheduler, os.path.join(save_directory, "scheduler"))


class DDPMPipeline(_PipelineBase):
    def noise_draws(self, num_inference_steps=1000, start_from=0, **kwargs):
        self.scheduler.set_timesteps(num_inference_steps)
        return sum(1 for t in self.scheduler.timesteps[start_from:] if int(t) > 0)      # scheduling_ddpm.py:400: `if t > 0`

    @_static_weights
    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, num_inference_steps=1000, start_from=0, output_type="pil", init=None,
                 save_every_step=False, return_dict=True, **kwargs):
        image, shape = self._start(batch_size, generator, init)
        self.scheduler.set_timesteps(num_inference_steps)
        mov = [self._to_numpy(image, shape)] if save_every_step else []
        ts_host = self.scheduler.timesteps[start_from:]
        ts_dev = torch.as_tensor(ts_host, dtype=torch.int64).to(self.device)    # one H2D copy for the whole chain
        for i, t in enumerate(self.progress_bar(ts_host)):
            t = int(t)
            eps = self.unet(image.permute(0, 3, 1, 2), ts_dev[i]).sample.permute(0, 2, 3, 1)        # NHWC storage
            noise = None
            if t > 0:   # same draw order / shape as randn_tensor(model_output.shape) in the reference
                noise = randn_tensor(shape, generator=generator, device=self.device)
                noise = ops.nchw_to_nhwc(noise)
            image = self.scheduler.step(eps.permute(0, 3, 1, 2), t, image.permute(0, 3, 1, 2),
                                        noise=noise.permute(0, 3, 1, 2) if noise is not None else None
                                        ).prev_sample.permute(0, 2, 3, 1)
            if save_every_step:
                mov.append(self._to_numpy(image, shape))
        if output_type == "u8":
            images = self._to_u8(image, shape)
            return ImagePipelineOutput(images=images, movie=mov) if return_dict else (images,)
        images = self._to_numpy(image, shape)
        if output_type == "pil":
            images = self.numpy_to_pil(images)
            if save_every_step:
                mov = list(map(self.numpy_to_pil, mov))
        if not return_dict:
            return (images,)
        return ImagePipelineOutput(images=images, movie=mov)


class DDIMPipeline(_PipelineBase):
    def __init__(self, unet, scheduler):
        # pipeline_ddim.py:39-42: make sure the scheduler can always be converted to DDIM
        scheduler = DDIMScheduler.from_config(scheduler.config)
        super().__init__(unet, scheduler)

    def noise_draws(self, num_inference_steps=50, eta=0.0, **kwargs):
        self.scheduler.set_timesteps(num_inference_steps)
        return len(self.scheduler.timesteps) if eta > 0 else 0                           # scheduling_ddim.py:366: `if eta > 0`

    @_static_weights
    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, use_clipped_model_output=None,
                 output_type="pil", init=None, save_every_step=False, return_dict=True, **kwargs):
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}.")
        image, shape = self._start(batch_size, generator, init)
        self.scheduler.set_timesteps(num_inference_steps)
        mov = [self._to_numpy(image, shape)] if save_every_step else []
        ts_dev = torch.as_tensor(self.scheduler.timesteps, dtype=torch.int64).to(self.device)   # one H2D copy for the whole chain
        for i, t in enumerate(self.progress_bar(self.scheduler.timesteps)):
            t = int(t)
            eps = self.unet(image.permute(0, 3, 1, 2), ts_dev[i]).sample
            image = self.scheduler.step(eps, t, image.permute(0, 3, 1, 2), eta=eta,
                                        use_clipped_model_output=bool(use_clipped_model_output),
                                        generator=generator).prev_sample.permute(0, 2, 3, 1)
            if save_every_step:
                mov.append(self._to_numpy(image, shape))
        if output_type == "u8":
            images = self._to_u8(image, shape)
            return ImagePipelineOutput(images=images, movie=mov) if return_dict else (images,)
        images = self._to_numpy(image, shape)
        if output_type == "pil":
            images = self.numpy_to_pil(images)
            if save_every_step:
                mov = list(map(self.numpy_to_pil, mov))
        if not return_dict:
            return (images,)
        return ImagePipelineOutput(images=images, movie=mov)


class PNDMPipeline(_PipelineBase):
    """pipelines/pndm/pipeline_pndm.py:39-122 (the reference's modified copy: `clip_sample` / `clip_sample_range`, `init`,
    `start_from`, `save_every_step`).  Whatever scheduler it is given is converted to a PNDMScheduler from its config
    (:46) -- which is how every `--sched` other than DDPM / DDIM samples in the reference.  The post-step clamp (:108-109)
    is folded into the scheduler's update kernel."""

    def __init__(self, unet, scheduler, clip_sample=False, clip_sample_range=1.0):
        super().__init__(unet, PNDMScheduler.from_config(scheduler.config))
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range

    def encode(self, image, *args, **kwargs):
        return image

    def decode(self, image, *args, **kwargs):
        return image

    @_static_weights
    @torch.no_grad()
    def __call__(self, batch_size=1, num_inference_steps=50, start_from=0, generator=None, output_type="pil", init=None,
                 save_every_step=False, return_dict=True, **kwargs):
        image, shape = self._start(batch_size, generator, init)
        mov = [self._to_numpy(image, shape)] if save_every_step else []
        self.scheduler.set_timesteps(num_inference_steps)
        ts_host = self.scheduler.timesteps[int(start_from):]
        ts_dev = torch.as_tensor(ts_host, dtype=torch.int64).to(self.device)
        clip = float(self.clip_sample_range) if self.clip_sample else None
        for i, t in enumerate(self.progress_bar(ts_host)):
            eps = self.unet(image.permute(0, 3, 1, 2), ts_dev[i]).sample
            image = self.scheduler.step(eps, int(t), image.permute(0, 3, 1, 2), clip=clip).prev_sample.permute(0, 2, 3, 1)
            if save_every_step:
                mov.append(self._to_numpy(image, shape))
        if output_type == "u8":
            images = self._to_u8(image, shape)
            return ImagePipelineOutput(images=images, movie=mov) if return_dict else (images,)
        images = self._to_numpy(image, shape)
        if output_type == "pil":
            images = self.numpy_to_pil(images)
            if save_every_step:
                mov = list(map(self.numpy_to_pil, mov))
        if not return_dict:
            return (images,)
        return ImagePipelineOutput(images=images, movie=mov)
