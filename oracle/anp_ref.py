"""Oracle: one batch of the Adversarial Neuron Pruning loop (fp32, CPU).

Follows /root/reference/anp_model.py:490-514 (PerturbConv2d = conv, then PerturbBatchNorm2d evaluated with fixed mean 0 / variance 1 / eps 0:
anp_model.py:152-207 -- restated literally in unet_ref.conv2d, which applies `<conv>.bn.weight / .bn.bias` when the parameter dict holds them),
anp_util.py:60-88 (convert_model: every nn.Conv2d, found by attribute type), :130-135 (Adam over the parameters whose name contains 'bn'),
anp_defense.py:136-160 (loop body: loss = -p_losses(clean, R = 0); clip_grad_norm_(1.0); Adam step; clip_weight; backdoor MSE under no_grad),
:47-66 (backdoor_mse_fn), :68-75 (clip_weight: clamp EVERY bn parameter to [-budget, budget]).
Pinned by tests/golden/anp.npz (G13: the reference's own PerturbConv2d on the reference's UNet2DModel).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from .loss_ref import p_losses, q_sample
from .train_ref import clip_and_adam
from .unet_ref import param_shapes, unet_forward


def conv_names(cfg):
    """names of the nn.Conv2d layers (4-D weights) in state-dict order."""
    return [k[: -len(".weight")] for k, shp in param_shapes(cfg).items() if k.endswith(".weight") and len(shp) == 4]


def init_bn(cfg, P):
    """anp_model.py:63-72 reset: bn.weight = 1, bn.bias = 0 for every wrapped convolution."""
    bn = {}
    for n in conv_names(cfg):
        c = P[n + ".weight"].shape[0]
        bn[n + ".bn.weight"] = torch.ones(c)
        bn[n + ".bn.bias"] = torch.zeros(c)
    return bn


def anp_loss_and_grads(cfg, P, bn, alphas, alphas_cumprod, clean, t, noise):
    """(-clean MSE, its gradient w.r.t. every bn parameter, gradients of the conv weights / biases).
    anp_util.freeze runs BEFORE convert_model (anp_util.py:125-126), and PerturbConv2d.__init__ builds a NEW nn.Conv2d whose weight and bias are fresh
    Parameters (anp_model.py:492-505): in the perturbed model the convolution weights require grad again.  The optimiser never sees them
    (anp_util.py:133-135), but `clip_grad_norm_(model.parameters(), 1.0)` (anp_defense.py:152) does: the clip coefficient of the bn step is taken from
    the norm over bn AND conv weight / bias gradients.  Everything else (norms, Linear layers, attention) stays frozen."""
    bn = {k: v.detach().clone().requires_grad_(True) for k, v in bn.items()}
    convp = {n + s: P[n + s].detach().clone().requires_grad_(True) for n in conv_names(cfg) for s in (".weight", ".bias")}
    full = dict(P); full.update(convp); full.update(bn)
    loss = -p_losses(alphas, alphas_cumprod, lambda x, tt: unet_forward(cfg, full, x, tt), clean, torch.zeros_like(clean), t, noise)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in bn.items()}, {k: v.grad for k, v in convp.items()}


def backdoor_mse(cfg, P, bn, alphas, alphas_cumprod, clean, target_images, trigger_images, t, noise):
    """anp_defense.py:47-66: input from the clean images (R = 0), target from the backdoor pair (x_start = target image, R = trigger image)."""
    full = dict(P); full.update(bn)
    with torch.no_grad():
        x_noisy, _ = q_sample(alphas, alphas_cumprod, clean, torch.zeros_like(clean), t, noise)
        _, btarget = q_sample(alphas, alphas_cumprod, target_images, trigger_images, t, noise)
        return F.mse_loss(btarget, unet_forward(cfg, full, x_noisy.contiguous(), t.contiguous()))


def anp_step(cfg, P, bn, state, alphas, alphas_cumprod, clean, trigger_images, target_images, t, noise, lr, step, budget):
    loss, G, Gc = anp_loss_and_grads(cfg, P, bn, alphas, alphas_cumprod, clean, t, noise)
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in list(G.values()) + list(Gc.values()))).float()
    coef = torch.clamp(1.0 / (norm + 1e-6), max=1.0)                      # clip_grad_norm_(all parameters that have a gradient, 1.0)
    bn2, state2, _ = clip_and_adam(bn, {k: g * coef for k, g in G.items()}, state, lr, step, max_norm=float("inf"))
    if budget is not None and budget >= 0:
        bn2 = {k: v.clamp(-budget, budget) for k, v in bn2.items()}
    bm = backdoor_mse(cfg, P, bn2, alphas, alphas_cumprod, clean, target_images, trigger_images, t, noise)
    return loss, G, norm, bn2, state2, bm
