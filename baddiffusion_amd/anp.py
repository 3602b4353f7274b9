"""Adversarial Neuron Pruning on the MI355X path (SURVEY f-4; reference: anp_model.py, anp_util.py, anp_defense.py).

The reference wraps every `nn.Conv2d` of the UNet in `PerturbConv2d` (anp_model.py:490-514): the convolution followed by
`PerturbBatchNorm2d` evaluated with FIXED statistics mean 0 / variance 1 / eps 0 (anp_model.py:152-207), i.e. a learnable
per-output-channel affine map `y_c <- w_c * y_c + b_c` (`bn.weight` init 1, `bn.bias` init 0).  Only those `bn` parameters train
(anp_util.py:130-135), with Adam, to MAXIMISE the clean denoising loss (anp_defense.py:147) under a clamp to +-perturb_budget
(anp_defense.py:68-75).

The affine map commutes with the convolution, so nothing new runs inside the network: the perturbed UNet is the ordinary plan
(`bd_unet_forward / backward`) on EFFECTIVE parameters `W'_c = w_c W_c`, `b'_c = w_c b_c + b_c(bn)` (`bd_anp_apply`), and the gradient
of the perturbation is a row-wise contraction of the ordinary weight gradient (`bd_anp_grad`, include/bd_hip.h).  No aten compute:
loss, network, clip + Adam (`bd_sumsq`, `bd_adam_clip`) and the clamp (`bd_lincomb`) are HIP kernels.

Upstream this defense does not run at the surveyed commit (anp_util.py:123 unpacks a 3-tuple into 2 names); what is
mirrored here is what the code says it computes, pinned by vectors made with the reference's own `PerturbConv2d` class
(tests/golden/anp.npz, G13)."""
import math
from collections import OrderedDict
from contextlib import contextmanager
from dataclasses import dataclass

import torch
from torch import nn

from . import ops
from .loss import p_losses_diffuser, q_sample_diffuser, _LossFn
from .unet import UNet2DModel, UNet2DOutput, _UNetFn


class _AnpApplyFn(torch.autograd.Function):
    """perturbation [2 T] -> effective flat parameters; backward contracts the flat weight gradient row by row."""

    @staticmethod
    def forward(ctx, perturb, model):
        T = model.total_rows
        ctx.model = model
        ctx.save_for_backward(perturb)
        return ops.anp_apply(model.base.flat.detach(), perturb[:T], perturb[T:], model._items, T)

    @staticmethod
    def backward(ctx, geff):
        m = ctx.model
        (perturb,) = ctx.saved_tensors
        T = m.total_rows
        g = torch.empty(2 * T, device=geff.device)
        m.conv_grad_norms = torch.empty(T, device=geff.device)      # read by AnpTrainer: the reference clips over the conv weights' gradients too
        ops.anp_grad(m.base.flat.detach(), geff.contiguous(), m._items, T, g[:T], g[T:], pert_w=perturb.detach()[:T], row_norm=m.conv_grad_norms)
        return g, None


class PerturbedUNet2DModel(nn.Module):
    """`anp_util.convert_model(model)` for this package's UNet2DModel: same call surface (`model(sample, t).sample`, `.config`,
    `.in_channels`, `.sample_size`, `state_dict()` with the extra `<conv>.bn.weight / .bn.bias` keys), the base weights frozen
    (anp_util.freeze), ONE flat trainable tensor `perturb` = [all bn.weight | all bn.bias]."""

    def __init__(self, base: UNet2DModel):
        super().__init__()
        if not isinstance(base, UNet2DModel):
            raise TypeError("PerturbedUNet2DModel wraps baddiffusion_amd.UNet2DModel")
        self.base = base
        base.flat.requires_grad_(False)
        self.conv_names, rows, p = [], [], 0
        for key, (off, shape, _layout) in base._table.items():
            if not key.endswith(".weight") or len(shape) != 4:       # every Conv2d layer of the reference model, nothing else
                continue
            name = key[: -len(".weight")]
            bkey = name + ".bias"
            boff = base._table[bkey][0] if bkey in base._table else -1
            cout = int(shape[0])
            rows.append((off, boff, cout, int(math.prod(shape[1:])), p))
            self.conv_names.append((name, p, cout))
            p += cout
        self.total_rows = p
        self.register_buffer("_items", torch.tensor(rows, dtype=torch.int64), persistent=False)
        self.perturb = nn.Parameter(torch.cat([torch.ones(p), torch.zeros(p)]))
        self.is_perturb = True
        self.to(base.flat.device)

    # ---- reference surface ---------------------------------------------------------------------------------------------------
    config = property(lambda self: self.base.config)
    in_channels = property(lambda self: self.base.in_channels)
    sample_size = property(lambda self: self.base.sample_size)
    device = property(lambda self: self.base.flat.device)
    dtype = property(lambda self: self.base.flat.dtype)

    def enable_perturb(self):       # anp_model.py:516-524
        self.is_perturb = True
        return self

    def disable_perturb(self):
        self.is_perturb = False
        return self

    def named_bn_parameters(self):
        """(name, view) pairs `<conv>.bn.weight` / `<conv>.bn.bias` (views of `perturb`), the parameters anp_util.py:133 selects by 'bn'."""
        T = self.total_rows
        for name, p, cout in self.conv_names:
            yield name + ".bn.weight", self.perturb[p: p + cout]
            yield name + ".bn.bias", self.perturb[T + p: T + p + cout]

    def bn_grads(self):
        T, g, out = self.total_rows, self.perturb.grad, OrderedDict()
        for name, p, cout in self.conv_names:
            out[name + ".bn.weight"] = g[p: p + cout]
            out[name + ".bn.bias"] = g[T + p: T + p + cout]
        return out

    def state_dict(self, *a, **k):
        sd = self.base.state_dict()
        for n, v in self.named_bn_parameters():
            sd[n] = v.detach().clone()
        return sd

    @torch.no_grad()
    def load_bn_state(self, sd):
        for n, v in self.named_bn_parameters():
            if n in sd:
                v.copy_(torch.as_tensor(sd[n], dtype=torch.float32).to(v.device))

    def effective_flat(self):
        return _AnpApplyFn.apply(self.perturb, self) if self.is_perturb else self.base.flat

    @contextmanager
    def static_weights(self):
        """sampling loops (pipelines.py): the effective parameters are built once per loop and the plan prepares them once."""
        self._eff_cache = self.effective_flat().detach()
        try:
            with self.base.static_weights():
                yield self
        finally:
            self._eff_cache = None

    def forward(self, sample, timestep, class_labels=None, return_dict=True):
        if class_labels is not None:
            raise NotImplementedError("class conditioning is not supported")
        base = self.base
        x_nhwc, t = base._prep_inputs(sample, timestep)
        cached = getattr(self, "_eff_cache", None)
        eff = cached if cached is not None and not torch.is_grad_enabled() else self.effective_flat()
        if torch.is_grad_enabled() and eff.requires_grad:
            out = _UNetFn.apply(eff, x_nhwc, t, base)
        else:
            out = base._forward_chunked(x_nhwc, t, base.effective_chunk(x_nhwc.shape[0]), flat=eff.detach())
        sample_out = out.permute(0, 3, 1, 2)
        return UNet2DOutput(sample=sample_out) if return_dict else (sample_out,)


def convert_model(model):
    """anp_util.py:60-88."""
    return PerturbedUNet2DModel(model)


def enable_perturb(model):
    model.enable_perturb()


def disable_perturb(model):
    model.disable_perturb()


def clip_weight(model, budget=None):
    """anp_defense.py:68-75: clamp every bn parameter (weights AND biases) to [-budget, budget]; None / negative = no clamp."""
    if budget is None or budget < 0:
        return
    p = model.perturb.data
    ops.lincomb([p], [1.0], clip=float(budget), out=p)


def backdoor_mse_fn(noise_sched, model, x_start, backdoor_x_start, R, backdoor_R, timesteps, noise=None, loss_type="l2"):
    """anp_defense.py:47-66: the model sees the CLEAN noisy input, its prediction is scored against the BACKDOOR target."""
    if len(x_start) == 0:
        return 0
    if loss_type not in ops.LOSS_TYPES:
        raise NotImplementedError()
    if noise is None:
        noise = torch.randn_like(x_start)
    x_noisy, _ = q_sample_diffuser(noise_sched=noise_sched, x_start=x_start, R=R, timesteps=timesteps, noise=noise)
    _, backdoor_target = q_sample_diffuser(noise_sched=noise_sched, x_start=backdoor_x_start, R=backdoor_R, timesteps=timesteps, noise=noise)
    pred = model(x_noisy.contiguous(), timesteps.contiguous(), return_dict=False)[0]
    return _LossFn.apply(pred.permute(0, 2, 3, 1), backdoor_target.permute(0, 2, 3, 1), loss_type)


@dataclass
class AnpConfig:
    """the fields of anp_config.py the loop reads (defaults as there)."""
    learning_rate: float = 1e-4
    perturb_budget: float = 4.0
    max_grad_norm: float = 1.0
    epoch: int = 10
    batch: int = 128


class AnpTrainer:
    """The body of anp_defense.train_loop (anp_defense.py:136-160) for one batch: loss = -p_losses(clean), backward, clip_grad_norm_(1.0), Adam on
    the bn parameters (anp_util.py:133-135: torch.optim.Adam(perturb_params, lr), defaults beta (0.9, 0.999), eps 1e-8), clamp to the budget,
    then the backdoor MSE under no_grad.  Adam / clip run as bd_sumsq + bd_adam_clip on the flat perturbation."""

    def __init__(self, model: PerturbedUNet2DModel, noise_sched, config: AnpConfig = None):
        self.model, self.noise_sched, self.config = model, noise_sched, config or AnpConfig()
        self.m = torch.zeros_like(model.perturb.data)
        self.v = torch.zeros_like(model.perturb.data)
        self.step_count = 0
        self.last_grad_norm = torch.zeros((), device=model.perturb.device)

    def step(self, clean_images, trigger_images, target_images, timesteps, noise, lr=None):
        cfg, model = self.config, self.model
        zeros = torch.zeros_like(trigger_images)
        model.perturb.grad = None
        loss = -p_losses_diffuser(self.noise_sched, model=model, x_start=clean_images, R=zeros, timesteps=timesteps, noise=noise, loss_type="l2")
        loss.backward()
        g = model.perturb.grad
        # clip_grad_norm_(model.parameters(), 1.0), anp_defense.py:152: over every parameter that HAS a gradient -- the bn parameters and, because
        # PerturbConv2d re-creates them as trainable Parameters (anp_model.py:492-505), the conv weights and biases (not updated: anp_util.py:133-135)
        ss = ops.sumsq(g) + ops.sumsq(model.conv_grad_norms)
        self.step_count += 1
        ops.adam_clip(model.perturb.data, g, self.m, self.v, ss, self.step_count, cfg.learning_rate if lr is None else lr,
                      max_norm=cfg.max_grad_norm, grad_norm_out=self.last_grad_norm)
        clip_weight(model, cfg.perturb_budget)
        with torch.no_grad():
            bmse = backdoor_mse_fn(self.noise_sched, model=model, x_start=clean_images, backdoor_x_start=target_images, R=zeros,
                                   backdoor_R=trigger_images, timesteps=timesteps, noise=noise, loss_type="l2")
        return {"loss": loss.detach(), "clean_mse": -loss.detach(), "backdoor_mse": bmse.detach() if torch.is_tensor(bmse) else bmse,
                "grad_norm": self.last_grad_norm}
