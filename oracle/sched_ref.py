"""Oracle: DDPM / DDIM scheduler math (fp32, CPU).

Follows /root/reference/diffusers/src/diffusers/schedulers/scheduling_ddpm.py:140-171
(tables), :197-248 (set_timesteps), :250-288 (_get_variance), :324-420 (step),
:422-443 (add_noise), :468-481 (previous_timestep) and
scheduling_ddim.py:130-175 (tables), :192-200 (_get_variance), :237-259
(set_timesteps), :261-381 (step).  Only what BadDiffusion exercises is restated:
epsilon prediction, linear betas, fixed_small / fixed_large variance, clip on/off,
the local clip_defense clamp (scheduling_ddpm.py:137-138, 414-415).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch


def make_tables(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
    # scheduling_ddpm.py:142-160 -- fp32 linspace, fp32 cumprod (do NOT use fp64)
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    return betas, alphas, alphas_cumprod


def ddpm_timesteps(num_inference_steps, num_train_timesteps=1000):
    # scheduling_ddpm.py:241-246
    step_ratio = num_train_timesteps // num_inference_steps
    return (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=0):
    # scheduling_ddim.py:254-259
    return ddpm_timesteps(num_inference_steps, num_train_timesteps) + steps_offset


def ddpm_variance(alphas_cumprod, t, prev_t, variance_type="fixed_small"):
    # scheduling_ddpm.py:250-276 (0-dim fp32 tensor arithmetic, as the reference does on CPU)
    one = torch.tensor(1.0)
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else one
    cur_beta = 1 - a_t / a_prev
    var = (1 - a_prev) / (1 - a_t) * cur_beta
    var = torch.clamp(var, min=1e-20)
    if variance_type == "fixed_small":
        return var
    if variance_type == "fixed_large":
        return cur_beta
    raise NotImplementedError(variance_type)


def ddpm_step(alphas_cumprod, model_output, t, sample, noise, num_inference_steps=None,
              num_train_timesteps=1000, variance_type="fixed_small", clip_sample=True,
              clip_sample_range=1.0, clip_defense=False, clip_defense_range=1.0):
    """One reverse step.  ``noise`` is the variance noise the reference would draw with
    randn_tensor (scheduling_ddpm.py:400-404); it is an INPUT here so seeds can be shared.
    Returns (prev_sample, pred_original_sample)."""
    t = int(t)
    n_inf = num_inference_steps if num_inference_steps else num_train_timesteps
    prev_t = t - num_train_timesteps // n_inf                        # :476-479
    one = torch.tensor(1.0)
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else one
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5           # :370
    if clip_sample:
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)         # :384-387
    c0 = (a_prev ** 0.5 * cur_beta) / b_t                            # :391
    ct = cur_alpha ** 0.5 * b_prev / b_t                             # :392
    prev = c0 * x0 + ct * sample                                     # :396
    if t > 0:
        var = ddpm_variance(alphas_cumprod, t, prev_t, variance_type)
        prev = prev + (var ** 0.5) * noise                           # :411-413
    if clip_defense:
        prev = prev.clamp(-clip_defense_range, clip_defense_range)   # :414-415
    return prev, x0


def ddim_step(alphas_cumprod, model_output, t, sample, num_inference_steps, eta=0.0, noise=None,
              num_train_timesteps=1000, clip_sample=True, clip_sample_range=1.0, set_alpha_to_one=True):
    # scheduling_ddim.py:300-381
    t = int(t)
    prev_t = t - num_train_timesteps // num_inference_steps
    final_alpha = torch.tensor(1.0) if set_alpha_to_one else alphas_cumprod[0]
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else final_alpha
    b_t = 1 - a_t
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    eps = model_output
    if clip_sample:
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)
    var = ((1 - a_prev) / b_t) * (1 - a_t / a_prev)                  # :192-200
    std = eta * var ** 0.5
    direction = (1 - a_prev - std ** 2) ** 0.5 * eps
    prev = a_prev ** 0.5 * x0 + direction
    if eta > 0:
        prev = prev + std * noise
    return prev, x0


def add_noise(alphas_cumprod, x0, noise, timesteps):
    # scheduling_ddpm.py:422-443
    a = alphas_cumprod[timesteps] ** 0.5
    s = (1 - alphas_cumprod[timesteps]) ** 0.5
    shape = (-1,) + (1,) * (x0.dim() - 1)
    return a.reshape(shape) * x0 + s.reshape(shape) * noise


def to_image(x):
    # pipeline_ddpm.py:115-116  -> NHWC float32 in [0,1]
    return (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous()


# ---- PNDM (SURVEY f-4): scheduling_pndm.py:150-190 (set_timesteps), :236-289 (Runge-Kutta warm-up), :291-353 (linear
# multistep), :366-397 (formula (9) of the PNDM paper).  Every `--sched` other than DDPM / DDIM reaches this scheduler:
# pipelines/pndm/pipeline_pndm.py:46 rebuilds a PNDMScheduler from the config of whatever scheduler it is handed. --------
def pndm_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=0, skip_prk_steps=False):
    """(prk_timesteps, plms_timesteps): the 12 warm-up evaluations (4 Runge-Kutta stages for each of the last... first 3
    transitions, the two middle stages sharing the half-step time) followed by the multistep tail."""
    ratio = num_train_timesteps // num_inference_steps
    base = (np.arange(0, num_inference_steps) * ratio).round() + steps_offset
    if skip_prk_steps:
        prk = np.array([])
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
    else:
        stages = np.array(base[-4:]).repeat(2) + np.tile(np.array([0, ratio // 2]), 4)
        prk = (stages[:-1].repeat(2)[1:-1])[::-1].copy()
        plms = base[:-3][::-1].copy()
    return prk.astype(np.int64), plms.astype(np.int64)


def pndm_transfer(alphas_cumprod, sample, t, prev_t, eps, final_alpha_cumprod=None):
    """x_{t-d} from x_t and an epsilon estimate, formula (9): both coefficients are fp32 scalars"""
    a_t = alphas_cumprod[t]
    a_p = alphas_cumprod[prev_t] if prev_t >= 0 else (alphas_cumprod[0] if final_alpha_cumprod is None else final_alpha_cumprod)
    b_t, b_p = 1 - a_t, 1 - a_p
    sample_coeff = (a_p / a_t) ** 0.5
    denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
    return sample_coeff * sample - (a_p - a_t) * eps / denom


class PNDMRef:
    """State machine of the pseudo numerical method: feed it (model_output, t, sample) in the order of `timesteps`."""

    def __init__(self, alphas_cumprod, num_inference_steps, num_train_timesteps=1000, skip_prk_steps=False, steps_offset=0):
        self.ac, self.n, self.T, self.skip = alphas_cumprod, num_inference_steps, num_train_timesteps, skip_prk_steps
        self.prk, self.plms = pndm_timesteps(num_inference_steps, num_train_timesteps, steps_offset, skip_prk_steps)
        self.timesteps = np.concatenate([self.prk, self.plms]).astype(np.int64)
        self.counter, self.acc, self.cur_sample, self.ets = 0, 0, None, []

    def step(self, eps, t, sample):
        t = int(t)
        ratio = self.T // self.n
        if self.counter < len(self.prk) and not self.skip:          # Runge-Kutta stage (counter % 4)
            prev_t = t - (0 if self.counter % 2 else ratio // 2)
            t0 = int(self.prk[self.counter // 4 * 4])
            r = self.counter % 4
            if r == 0:
                self.acc = self.acc + eps / 6
                self.ets.append(eps)
                self.cur_sample = sample
            elif r in (1, 2):
                self.acc = self.acc + eps / 3
            else:
                eps = self.acc + eps / 6
                self.acc = 0
            out = pndm_transfer(self.ac, self.cur_sample, t0, prev_t, eps)
        else:                                                         # Adams-Bashforth on the stored estimates
            prev_t = t - ratio
            if self.counter != 1:
                self.ets = self.ets[-3:] + [eps]
            else:
                prev_t, t = t, t + ratio
            k = len(self.ets)
            if k == 1 and self.counter == 0:
                self.cur_sample = sample
            elif k == 1 and self.counter == 1:
                eps = (eps + self.ets[-1]) / 2
                sample, self.cur_sample = self.cur_sample, None
            elif k == 2:
                eps = (3 * self.ets[-1] - self.ets[-2]) / 2
            elif k == 3:
                eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
            else:
                eps = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
            out = pndm_transfer(self.ac, sample, t, prev_t, eps)
        self.counter += 1
        return out
