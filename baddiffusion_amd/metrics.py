"""Measure path (SURVEY f-3): MSE / SSIM against the backdoor target and the Frechet statistics, on the device.

Reference call sites: baddiffusion.py:536-547 (`nn.MSELoss`, `StructuralSimilarityIndexMeasure(data_range=1.0)` on
`[N,3,S,S]` images in [0,1]) and fid_score.py:150-230 (activation statistics + Frechet distance over pool3 features).
* `ActivationStats` accumulates sum and sum of outer products in fp64 ON THE GPU, batch by batch (no [N, 2048] host
  array); `frechet_distance` then needs one matrix square root of a d x d product, which stays on scipy / CPU.
* SSIM follows the published torchmetrics defaults the reference relies on (11x11 Gaussian, sigma 1.5, k1 0.01,
  k2 0.03, reflect padding, border crop, mean over C,H,W then over the batch).  torchmetrics is not installed in the
  build container, so SSIM parity is UNPINNED; Frechet / statistics are pinned by G8.
* The Inception pool3 network (pytorch_fid weights) is an asset that does not travel: callers pass any feature
  extractor `f(images) -> [n, d]`.
"""
import numpy as np
import torch
import torch.nn.functional as F   # F.pad only


def mse(a, b):
    """nn.MSELoss(reduction='mean') (baddiffusion.py:545).  GPU tensors: the fused l2-loss kernel (bd_loss_fwd_bwd, fp64
    partial sums); CPU tensors (tests): the same formula in fp64."""
    if a.is_cuda and b.is_cuda and a.shape == b.shape and a.shape[-1] % 1 == 0:
        from . import ops
        a32, b32 = a.float().contiguous(), b.float().contiguous()
        loss, _ = ops.loss_fwd_bwd(a32.reshape(-1, a32.shape[-1]), b32.reshape(-1, b32.shape[-1]), "l2", want_grad=False)
        return float(loss)
    return float(((a.double() - b.double()) ** 2).mean())


def _gauss1d(k, sigma, device, dtype):
    x = torch.arange(k, device=device, dtype=dtype) - (k - 1) / 2
    g = torch.exp(-(x / sigma) ** 2 / 2)
    return g / g.sum()


def ssim(preds, target, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Mean SSIM of two [N,C,H,W] batches.  GPU tensors with the reference's defaults run the HIP kernel bd_ssim; CPU tensors
    (tests, and the statement the kernel is checked against) run the torch restatement below."""
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError(f"expected two [N,C,H,W] tensors of the same shape, got {tuple(preds.shape)} and {tuple(target.shape)}")
    if preds.is_cuda and target.is_cuda and (kernel_size, sigma, k1, k2) == (11, 1.5, 0.01, 0.03) and min(preds.shape[-2:]) > 10:
        from . import ops                     # the HIP kernel (bd_ssim): the reference's defaults, no aten compute
        return float(ops.ssim(preds.float(), target.float(), data_range))
    p, t = preds.float(), target.float()
    C = p.shape[1]
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    g = _gauss1d(kernel_size, sigma, p.device, p.dtype)
    pad = (kernel_size - 1) // 2
    p = F.pad(p, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(t, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat((p, t, p * p, t * t, p * t))
    # the 11x11 window is separable (g g^T): "valid" filtering = two banded matrix products, rows then columns
    def band(n_out):
        m = torch.zeros(n_out, n_out + kernel_size - 1, device=p.device, dtype=p.dtype)
        idx = torch.arange(n_out, device=p.device)
        for j in range(kernel_size):
            m[idx, idx + j] = g[j]
        return m
    H, W = preds.shape[-2:]
    out = torch.einsum("ih,nchw,jw->ncij", band(H), stack, band(W))
    mp, mt, pp, tt, pt = out.split(preds.shape[0])
    sp, st, spt = pp - mp * mp, tt - mt * mt, pt - mp * mt
    full = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (sp + st + c2))
    full = full[..., pad:-pad, pad:-pad]
    return float(full.reshape(full.shape[0], -1).mean(-1).mean())


class ActivationStats:
    """Running mean / covariance of feature rows, accumulated in fp64 on the features' device
    (fid_score.py:207-230: mu = mean(act, 0), sigma = np.cov(act, rowvar=False))."""

    def __init__(self, dim, device=None):
        self.n = 0
        self.s = torch.zeros(dim, dtype=torch.float64, device=device)
        self.ss = torch.zeros(dim, dim, dtype=torch.float64, device=device)

    def update(self, act):
        a = act.reshape(act.shape[0], -1).to(device=self.s.device, dtype=torch.float64)
        self.n += a.shape[0]
        self.s += a.sum(0)
        self.ss += a.t() @ a
        return self

    def finalize(self):
        if self.n < 2:
            raise ValueError("need at least two feature rows")
        mu = self.s / self.n
        cov = (self.ss - self.n * torch.outer(mu, mu)) / (self.n - 1)
        return mu.cpu().numpy(), cov.cpu().numpy()


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2)) (fid_score.py:150-204, incl. its singular-product retry)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(np.asarray(mu1, np.float64)), np.atleast_1d(np.asarray(mu2, np.float64))
    sigma1, sigma2 = np.atleast_2d(np.asarray(sigma1, np.float64)), np.atleast_2d(np.asarray(sigma2, np.float64))
    if mu1.shape != mu2.shape:
        raise ValueError("Training and test mean vectors have different lengths")
    if sigma1.shape != sigma2.shape:
        raise ValueError("Training and test covariances have different dimensions")
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def fid_from_features(features_a, features_b):
    """Frechet distance between two iterables of feature batches ([n, d] tensors)."""
    stats = []
    for feats in (features_a, features_b):
        acc = None
        for f in feats:
            acc = acc or ActivationStats(f.reshape(f.shape[0], -1).shape[1], f.device)
            acc.update(f)
        stats.append(acc.finalize())
    return frechet_distance(stats[0][0], stats[0][1], stats[1][0], stats[1][1])
