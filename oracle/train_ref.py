"""Oracle: one optimisation step of the BadDiffusion train loop (fp32, CPU).

Follows /root/reference/baddiffusion.py:590-615 (loop body), :320 (Adam(lr), torch defaults
betas=(0.9,0.999), eps=1e-8, no weight decay), :611-612 (clip_grad_norm_(params, 1.0)) and
diffusers/src/diffusers/optimization.py:109-140 (cosine schedule with warmup).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math

import torch

from .loss_ref import p_losses
from .unet_ref import unet_forward


def cosine_lr_lambda(step, num_warmup_steps, num_training_steps, num_cycles=0.5):
    # optimization.py:134-138
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def loss_and_grads(cfg, P, alphas, alphas_cumprod, x0, R, t, noise):
    P = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss = p_losses(alphas, alphas_cumprod, lambda x, tt: unet_forward(cfg, P, x, tt), x0, R, t, noise)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in P.items()}


def clip_and_adam(P, G, state, lr, step, max_norm=1.0, b1=0.9, b2=0.999, eps=1e-8):
    """torch.nn.utils.clip_grad_norm_ (coef = max_norm/(norm+1e-6) clamped to 1) then torch.optim.Adam
    (single-tensor formulation).  ``step`` is 1-based.  Returns (new P, new state, grad_norm)."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in G.values())).float()
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    newP, newS = {}, {}
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, p in P.items():
        g = G[k] * coef
        m, v = state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
        m = m + (g - m) * (1 - b1)                       # lerp_, as torch's _single_tensor_adam
        v = v * b2 + (g * g) * (1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)) + eps
        newP[k] = p - (lr / bc1) * (m / denom)
        newS[k] = (m, v)
    return newP, newS, norm
