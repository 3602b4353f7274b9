"""DDPMScheduler / DDIMScheduler -- drop-in for the two diffusers schedulers BadDiffusion uses
(/root/reference/diffusers/src/diffusers/schedulers/scheduling_ddpm.py:76-481,
 scheduling_ddim.py:79-429), with `step` / `add_noise` running as HIP kernels.

Kept surface: `.betas .alphas .alphas_cumprod .timesteps .config.{num_train_timesteps,
clip_sample (assignable), variance_type, ...} .init_noise_sigma .set_timesteps() .step(...).prev_sample
.add_noise() .previous_timestep() ._get_variance() __len__`.
Tables are built on the host exactly as the reference does (fp32 linspace + fp32 cumprod) and mirrored
once to the device; `step` reads its coefficients from that device table (no per-step host scalar math,
no host<->device sync; scheduling_ddpm.py:350-365 does ~7 host scalar extractions per step).
Only what BadDiffusion exercises is supported: epsilon prediction, linear betas, fixed_small /
fixed_large variance; anything else raises like the reference does for unknown options.
"""
import numpy as np
import torch

from . import ops
from .unet import FrozenConfig


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class ShardedGenerator:
    """Rows [lo, hi) of the noise a SINGLE process would draw for a chunk of `full_batch` chains (SURVEY hard-part 4; the reference draws every
    sampling noise from one CPU generator, chunk after chunk: model.py:517-523, scheduling_ddpm.py:400-404).  Handed to a pipeline as its
    `generator`, every randn_tensor(shape, ...) call draws the FULL chunk's tensor ((full_batch,) + shape[1:]) from `gen` -- so the stream
    advances exactly as in the unsharded run, on every rank -- and returns this rank's rows.  The price of reference-order noise on a sharded
    job: every rank draws all of it (host RNG time), which is why it is opt-in (batch_sampling_save(parity=True), BD_SHARDED_NOISE=reference)."""

    def __init__(self, gen, full_batch, lo, hi):
        if gen is None or gen.device.type != "cpu":
            raise ValueError("ShardedGenerator needs a CPU torch.Generator (the reference's sampling stream is a CPU generator)")
        if not (0 <= lo <= hi <= full_batch):
            raise ValueError(f"rows [{lo}, {hi}) outside a chunk of {full_batch}")
        self.gen, self.full_batch, self.lo, self.hi = gen, int(full_batch), int(lo), int(hi)
        self.device = gen.device


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """utils/torch_utils.py:29-70: with a CPU generator the noise is drawn on the CPU (seed parity with
    the reference) and copied to the device."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, ShardedGenerator):
        if shape[0] != generator.hi - generator.lo:
            raise ValueError(f"ShardedGenerator for rows [{generator.lo}, {generator.hi}) asked for a batch of {shape[0]}")
        full = torch.randn((generator.full_batch,) + tuple(shape[1:]), generator=generator.gen, device="cpu", dtype=dtype)
        return full[generator.lo: generator.hi].contiguous().to(device)
    if generator is not None and generator.device.type == "cpu" and device.type != "cpu":
        return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class _Base:
    order = 1

    def _make_tables(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self._dev = {}

    def device_tables(self, device):
        """(alphas, alphas_cumprod) on `device`, uploaded once."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.alphas.to(device), self.alphas_cumprod.to(device))
        return self._dev[key]

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    @property
    def num_train_timesteps(self):
        """`noise_sched.num_train_timesteps` (baddiffusion.py:600, anp_defense.py:131): diffusers' config attribute shortcut."""
        return self.config.num_train_timesteps

    def add_noise(self, original_samples, noise, timesteps):
        # scheduling_ddpm.py:422-443  (q_sample with R = 0)
        a, ac = self.device_tables(original_samples.device)
        xn, _ = ops.qsample(original_samples, torch.zeros_like(original_samples), noise, timesteps, a, ac)
        return xn.permute(0, 3, 1, 2)

    @staticmethod
    def _flat(x):
        """dense storage view of a (possibly channels_last) tensor: the step kernels are elementwise, so any
        layout works as long as model_output / sample / noise share it."""
        if x.is_contiguous():
            return x, None
        p = x.permute(0, 2, 3, 1)
        if p.is_contiguous():
            return p, "nhwc"
        return x.contiguous(), None

    def _align(self, *ts):
        """bring all tensors to one common dense layout; returns (tensors, restore_fn)."""
        first, tag = self._flat(ts[0])
        outs = [first]
        for t in ts[1:]:
            if t is None:
                outs.append(None)
            elif tag == "nhwc":
                q = t.permute(0, 2, 3, 1)
                outs.append(q if q.is_contiguous() else q.contiguous())
            else:
                outs.append(t.contiguous())
        restore = (lambda y: y.permute(0, 3, 1, 2)) if tag == "nhwc" else (lambda y: y)
        return outs, restore


class DDPMScheduler(_Base):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 clip_defense=False, clip_defense_range=1.0, **unused):
        self.config = FrozenConfig(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
            trained_betas=trained_betas, variance_type=variance_type, clip_sample=clip_sample,
            prediction_type=prediction_type, thresholding=thresholding,
            dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=clip_sample_range,
            sample_max_value=sample_max_value, clip_defense=clip_defense, clip_defense_range=clip_defense_range)
        self._make_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.custom_timesteps = False
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self.variance_type = variance_type

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        # scheduling_ddpm.py:197-248
        if num_inference_steps is not None and timesteps is not None:
            raise ValueError("Can only pass one of `num_inference_steps` or `custom_timesteps`.")
        if timesteps is not None:
            for i in range(1, len(timesteps)):
                if timesteps[i] >= timesteps[i - 1]:
                    raise ValueError("`custom_timesteps` must be in descending order.")
            if timesteps[0] >= self.config.num_train_timesteps:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {self.config.num_train_timesteps}.")
            timesteps = np.array(timesteps, dtype=np.int64)
            self.custom_timesteps = True
        else:
            if num_inference_steps > self.config.num_train_timesteps:
                raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                                 f"`self.config.train_timesteps`: {self.config.num_train_timesteps}")
            self.num_inference_steps = num_inference_steps
            step_ratio = self.config.num_train_timesteps // self.num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            self.custom_timesteps = False
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def previous_timestep(self, timestep):
        # scheduling_ddpm.py:468-481
        if self.custom_timesteps:
            index = (self.timesteps == timestep).nonzero(as_tuple=True)[0][0]
            return torch.tensor(-1) if index == self.timesteps.shape[0] - 1 else self.timesteps[index + 1]
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        # scheduling_ddpm.py:250-288 (host scalar; kept for API parity and the reference's KATs)
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_prev
        variance = torch.clamp((1 - a_prev) / (1 - a_t) * cur_beta, min=1e-20)
        variance_type = variance_type or self.config.variance_type
        if variance_type == "fixed_small":
            return variance
        if variance_type == "fixed_large":
            return cur_beta
        raise NotImplementedError(f"variance_type {variance_type}: only fixed_small / fixed_large are on the BadDiffusion path")

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, noise=None):
        # scheduling_ddpm.py:324-420
        if self.config.prediction_type != "epsilon":
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be `epsilon` on the HIP path")
        if self.config.thresholding:
            raise NotImplementedError("dynamic thresholding is not on the BadDiffusion path")
        if not model_output.is_cuda:
            raise RuntimeError("DDPMScheduler.step runs on the GPU only")
        t = int(timestep)
        prev_t = int(self.previous_timestep(t))
        if t > 0 and noise is None:
            noise = randn_tensor(tuple(model_output.shape), generator=generator, device=model_output.device,
                                 dtype=model_output.dtype)
        (mo, sm, nz), restore = self._align(model_output, sample, noise if t > 0 else None)
        _, ac = self.device_tables(model_output.device)
        prev, x0 = ops.ddpm_step(mo, sm, nz, ac, t, prev_t, self.variance_type, self.config.clip_sample,
                                 self.config.clip_sample_range, self.config.clip_defense, self.config.clip_defense_range,
                                 want_x0=True)
        prev, x0 = restore(prev), restore(x0)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDIMScheduler(_Base):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0, **unused):
        self.config = FrozenConfig(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
            trained_betas=trained_betas, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
            steps_offset=steps_offset, prediction_type=prediction_type, thresholding=thresholding,
            dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=clip_sample_range,
            sample_max_value=sample_max_value)
        self._make_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        """pipeline_ddim.py:39-42 re-creates a DDIM scheduler from any scheduler's config."""
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "trained_betas", "clip_sample",
                "set_alpha_to_one", "steps_offset", "prediction_type", "thresholding", "dynamic_thresholding_ratio",
                "clip_sample_range", "sample_max_value")
        return cls(**{k: config[k] for k in keys if k in config})

    def set_timesteps(self, num_inference_steps, device=None):
        # scheduling_ddim.py:237-259
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {self.config.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // self.num_inference_steps
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(timesteps).to(device)
        self.timesteps += self.config.steps_offset

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        # scheduling_ddim.py:261-381
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.config.prediction_type != "epsilon":
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be `epsilon` on the HIP path")
        if self.config.thresholding:
            raise NotImplementedError("dynamic thresholding is not on the BadDiffusion path")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output is not on the BadDiffusion path")
        if not model_output.is_cuda:
            raise RuntimeError("DDIMScheduler.step runs on the GPU only")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise.")
            if variance_noise is None:
                variance_noise = randn_tensor(tuple(model_output.shape), generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
        (mo, sm, nz), restore = self._align(model_output, sample, variance_noise if eta > 0 else None)
        _, ac = self.device_tables(model_output.device)
        prev, x0 = ops.ddim_step(mo, sm, ac, t, prev_t, eta=float(eta), noise=nz, clip_sample=self.config.clip_sample,
                                 clip_sample_range=self.config.clip_sample_range,
                                 final_alpha_cumprod=float(self.final_alpha_cumprod), want_x0=True)
        prev, x0 = restore(prev), restore(x0)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class PNDMScheduler(_Base):
    """Pseudo numerical method scheduler (scheduling_pndm.py:60-426) on the GPU: 12 Runge-Kutta warm-up evaluations, then a
    4-step Adams-Bashforth tail, every update being `sample_coeff * x + eps_coeff * (combination of stored model outputs)`
    with formula (9) of the PNDM paper for the two scalars -- one fused `bd_lincomb` launch per update (plus one for the
    Runge-Kutta accumulator), coefficients computed on the host in fp32 like the reference's scalar tensor arithmetic.
    This is the scheduler every `--sched` other than DDPM / DDIM ends up as: PNDMPipeline re-creates it from whatever
    scheduler config it is given (pipeline_pndm.py:46).  `step(..., clip=r)` folds the pipeline's post-step clamp in."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon", steps_offset=0, **unused):
        self.config = FrozenConfig(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                   beta_schedule=beta_schedule, trained_betas=trained_betas, skip_prk_steps=skip_prk_steps,
                                   set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type, steps_offset=steps_offset)
        self._make_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.pndm_order = 4
        self.num_inference_steps = None
        self.prk_timesteps = self.plms_timesteps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._reset()

    _KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "trained_betas", "skip_prk_steps",
             "set_alpha_to_one", "prediction_type", "steps_offset")

    @classmethod
    def from_config(cls, config):
        return cls(**{k: config[k] for k in cls._KEYS if k in config})

    def _reset(self):
        self.counter, self.ets, self.cur_sample, self._acc = 0, [], None, None

    def set_timesteps(self, num_inference_steps, device=None):
        # scheduling_pndm.py:150-190
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        if self.config.skip_prk_steps:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        else:
            stages = np.array(base[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, ratio // 2]), self.pndm_order)
            self.prk_timesteps = (stages[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = base[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)).to(device)
        self._reset()

    def _coeffs(self, t, prev_t):
        """(sample_coeff, eps_coeff) of formula (9) as fp32 scalars (scheduling_pndm.py:366-397)"""
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return float((a_p / a_t) ** 0.5), float(-(a_p - a_t) / denom)

    def step(self, model_output, timestep, sample, return_dict=True, clip=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.config.prediction_type != "epsilon":
            raise ValueError(f"prediction_type given as {self.config.prediction_type} must be `epsilon` on the HIP path")
        if not model_output.is_cuda:
            raise RuntimeError("PNDMScheduler.step runs on the GPU only")
        (eps, x), restore = self._align(model_output, sample)
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        if self.counter < len(self.prk_timesteps) and not self.config.skip_prk_steps:
            # Runge-Kutta stage r of the current transition: weights 1/6, 1/3, 1/3, 1/6
            r = self.counter % 4
            prev_t = t - (0 if self.counter % 2 else ratio // 2)
            t0 = int(self.prk_timesteps[self.counter // 4 * 4])
            sc, ec = self._coeffs(t0, prev_t)
            if r == 0:
                self._acc = ops.lincomb([eps], [1.0 / 6.0])
                self.ets.append(eps)
                self.cur_sample = x
            base = self.cur_sample if self.cur_sample is not None else x
            if r in (1, 2):
                self._acc = ops.lincomb([self._acc, eps], [1.0, 1.0 / 3.0])
            if r == 3:
                prev = ops.lincomb([base, self._acc, eps], [sc, ec, ec / 6.0], clip=clip)
                self._acc = None
            else:
                prev = ops.lincomb([base, eps], [sc, ec], clip=clip)
        else:
            # linear multistep (Adams-Bashforth) on the stored model outputs
            if not self.config.skip_prk_steps and len(self.ets) < 3:
                raise ValueError(f"{self.__class__} can only be run AFTER scheduler has been run in 'prk' mode for at least 12 "
                                 "iterations See: https://github.com/huggingface/diffusers/blob/main/src/diffusers/pipelines/"
                                 "pipeline_pndm.py for more information.")
            prev_t = t - ratio
            if self.counter != 1:
                self.ets = self.ets[-3:] + [eps]
            else:
                prev_t, t = t, t + ratio
            k = len(self.ets)
            if k == 1 and self.counter == 0:
                terms, w = [eps], [1.0]
                self.cur_sample = x
            elif k == 1 and self.counter == 1:
                terms, w = [eps, self.ets[-1]], [0.5, 0.5]
                x, self.cur_sample = self.cur_sample, None
            elif k == 2:
                terms, w = [self.ets[-1], self.ets[-2]], [1.5, -0.5]
            elif k == 3:
                terms, w = [self.ets[-1], self.ets[-2], self.ets[-3]], [23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0]
            else:
                terms, w = [self.ets[-1], self.ets[-2], self.ets[-3], self.ets[-4]], [55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0]
            sc, ec = self._coeffs(t, prev_t)
            prev = ops.lincomb([x] + terms, [sc] + [ec * wi for wi in w], clip=clip)
        self.counter += 1
        prev = restore(prev)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev)


class SchedulerConfigCarrier(DDPMScheduler):
    """What DiffuserModelSched returns as `noise_sched` for the `--sched` types the reference constructs but never steps
    (model.py:598-630: DPM-Solver(++), UniPC, DEIS, Heun, LMS): training only uses its betas (`add_noise`,
    `alphas_cumprod`, `config.num_train_timesteps`) and sampling goes through PNDMPipeline, which rebuilds a PNDMScheduler
    from this config.  `class_name` records which reference class the config stands for."""

    def __init__(self, class_name, **kw):
        super().__init__(**kw)
        self.class_name = class_name

    def step(self, *a, **k):
        raise NotImplementedError(f"{self.class_name}.step is never called by the reference either: PNDMPipeline converts the "
                                  "scheduler to PNDMScheduler (pipeline_pndm.py:46)")
