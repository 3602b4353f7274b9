"""q_sample_diffuser / p_losses_diffuser -- drop-in for /root/reference/loss.py:257-307 on the HIP path.

Same signatures and argument meaning.  `p_losses_diffuser` returns a 0-dim tensor that carries grad: calling
`.backward()` on it runs `bd_unet_backward` and fills the model's (flat) parameter gradient, exactly like
`accelerator.backward(loss)` at baddiffusion.py:608.
"""
import torch

from . import ops


def _tables(noise_sched, device):
    if hasattr(noise_sched, "device_tables"):
        return noise_sched.device_tables(device)
    return (noise_sched.alphas.to(device=device, dtype=torch.float32),
            noise_sched.alphas_cumprod.to(device=device, dtype=torch.float32))


def q_sample_diffuser(noise_sched, x_start, R, timesteps, noise=None):
    """loss.py:257-285 -> (x_noisy, target), logical [B,C,H,W] (NHWC storage, i.e. channels_last views)."""
    if noise is None:
        noise = torch.randn_like(x_start)
    if not x_start.is_cuda:
        raise RuntimeError("q_sample_diffuser runs on the GPU only (libbd_hip.so); no CPU fallback on the hot path")
    a, ac = _tables(noise_sched, x_start.device)
    xn, tg = ops.qsample(x_start, R.to(x_start.device), noise.to(x_start.device), timesteps.to(x_start.device), a, ac)
    return xn.permute(0, 3, 1, 2), tg.permute(0, 3, 1, 2)


class _LossFn(torch.autograd.Function):
    """loss + dL/dpred in one fused kernel; backward just scales the stored gradient."""

    @staticmethod
    def forward(ctx, pred_nhwc, target_nhwc, loss_type):
        loss, dpred = ops.loss_fwd_bwd(pred_nhwc, target_nhwc, loss_type)
        ctx.save_for_backward(dpred)
        ctx.shape = pred_nhwc.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (dpred * g).reshape(ctx.shape), None, None


def p_losses_diffuser(noise_sched, model, x_start, R, timesteps, noise=None, loss_type="l2"):
    """loss.py:287-307."""
    if len(x_start) == 0:
        return 0
    if loss_type not in ops.LOSS_TYPES:
        raise NotImplementedError()
    if noise is None:
        noise = torch.randn_like(x_start)
    x_noisy, target = q_sample_diffuser(noise_sched=noise_sched, x_start=x_start, R=R, timesteps=timesteps, noise=noise)
    predicted_noise = model(x_noisy, timesteps.contiguous(), return_dict=False)[0]
    return _LossFn.apply(predicted_noise.permute(0, 2, 3, 1), target.permute(0, 2, 3, 1), loss_type)
