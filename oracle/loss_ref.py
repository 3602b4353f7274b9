"""Oracle: BadDiffusion forward process + loss (fp32, CPU).

Follows /root/reference/loss.py:257-285 (q_sample_diffuser) and :287-307
(p_losses_diffuser).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from .sched_ref import add_noise


def q_sample(alphas, alphas_cumprod, x_start, R, timesteps, noise):
    # loss.py:264-285 -- same operation order as the reference (pow 0.5 on gathered fp32 values)
    shape = (len(x_start),) + (1,) * (x_start.dim() - 1)
    sqrt_ac = alphas_cumprod[timesteps] ** 0.5
    sqrt_1mac = (1 - alphas_cumprod[timesteps]) ** 0.5
    r_coef = (1 - alphas[timesteps] ** 0.5) * sqrt_1mac / (1 - alphas[timesteps])
    noisy = add_noise(alphas_cumprod, x_start, noise, timesteps)
    return noisy + (1 - sqrt_ac.reshape(shape)) * R, r_coef.reshape(shape) * R + noise


def p_losses(alphas, alphas_cumprod, model_fn, x_start, R, timesteps, noise, loss_type="l2"):
    # loss.py:287-307
    x_noisy, target = q_sample(alphas, alphas_cumprod, x_start, R, timesteps, noise)
    pred = model_fn(x_noisy.contiguous(), timesteps.contiguous())
    if loss_type == "l1":
        return F.l1_loss(target, pred)
    if loss_type == "l2":
        return F.mse_loss(target, pred)
    if loss_type == "huber":
        return F.smooth_l1_loss(target, pred)
    raise NotImplementedError()
