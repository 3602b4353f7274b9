"""Measure path (SURVEY f-3): MSE / SSIM against the backdoor target and the Frechet statistics, on the device.

Reference call sites: baddiffusion.py:536-547 (`nn.MSELoss`, `StructuralSimilarityIndexMeasure(data_range=1.0)` on
`[N,3,S,S]` images in [0,1]) and fid_score.py:150-230 (activation statistics + Frechet distance over pool3 features).
* `ActivationStats` accumulates sum and sum of outer products in fp64 ON THE GPU, batch by batch (no [N, 2048] host
  array); `frechet_distance` then needs one matrix square root of a d x d product, which stays on scipy / CPU.
* SSIM follows the published torchmetrics defaults the reference relies on (11x11 Gaussian, sigma 1.5, k1 0.01,
  k2 0.03, reflect padding, border crop, mean over C,H,W then over the batch) and exists only as the HIP kernel bd_ssim
  (csrc/metrics.hip).  torchmetrics is not installed in the build container, so SSIM parity against torchmetrics itself is
  UNPINNED; the kernel is checked against the independent fp64 scipy statement in oracle/metrics_ref.py.
  Frechet / statistics are pinned by G8.
* The Inception pool3 network (pytorch_fid weights) is an asset that does not travel: callers pass any feature
  extractor `f(images) -> [n, d]`.
"""
import numpy as np
import torch


def mse(a, b):
    """nn.MSELoss(reduction='mean') (baddiffusion.py:545) through the fused l2-loss kernel (bd_loss_fwd_bwd): the difference
    is taken in fp32, its squares are accumulated in fp64 partial sums folded in fixed order.  Device tensors only: there is
    no CPU path (tests pin the kernel against the fp64 formula in oracle/metrics_ref.py, |err| <= 1e-6 relative)."""
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("metrics.mse: device tensors required (the measure path runs on the GPU; no CPU fallback)")
    if a.shape != b.shape:
        raise ValueError(f"mse: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    from . import ops
    a32, b32 = a.float().contiguous(), b.float().contiguous()
    loss, _ = ops.loss_fwd_bwd(a32.reshape(-1, a32.shape[-1]), b32.reshape(-1, b32.shape[-1]), "l2", want_grad=False)
    return float(loss)


def ssim(preds, target, data_range=1.0):
    """Mean SSIM of two [N,C,H,W] batches with the torchmetrics defaults the reference relies on (baddiffusion.py:260,546):
    the HIP kernel bd_ssim.  Device tensors only; images must be larger than the 11-tap window."""
    if preds.shape != target.shape or preds.dim() != 4:
        raise ValueError(f"expected two [N,C,H,W] tensors of the same shape, got {tuple(preds.shape)} and {tuple(target.shape)}")
    if not (preds.is_cuda and target.is_cuda):
        raise RuntimeError("metrics.ssim: device tensors required (bd_ssim is a HIP kernel; no CPU fallback)")
    if min(preds.shape[-2:]) <= 10:
        raise ValueError("ssim: images must be larger than the 11x11 window")
    from . import ops
    return float(ops.ssim(preds.float(), target.float(), data_range))


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
    """Squared Frechet distance between N(mu1, sigma1) and N(mu2, sigma2):
    |mu1 - mu2|^2 + Tr(sigma1) + Tr(sigma2) - 2 Tr((sigma1 sigma2)^(1/2)).
    Behaviour kept from fid_score.py:150-204 (pinned by G8): a non-finite square root is retried once with eps added to both
    diagonals, and a complex root is accepted only if its diagonal is real to 1e-3."""
    from scipy import linalg
    m = [np.atleast_1d(np.asarray(v, np.float64)) for v in (mu1, mu2)]
    S = [np.atleast_2d(np.asarray(v, np.float64)) for v in (sigma1, sigma2)]
    if m[0].shape != m[1].shape:
        raise ValueError("Training and test mean vectors have different lengths")
    if S[0].shape != S[1].shape:
        raise ValueError("Training and test covariances have different dimensions")

    def root_of_product(jitter):
        r = linalg.sqrtm((S[0] + jitter) @ (S[1] + jitter), disp=False)
        return r[0] if isinstance(r, tuple) else r
    root = root_of_product(0.0)
    if not np.all(np.isfinite(root)):
        root = root_of_product(eps * np.eye(S[0].shape[0]))
    if np.iscomplexobj(root):
        worst = float(np.abs(np.diagonal(root).imag).max())
        if worst > 1e-3:
            raise ValueError(f"Imaginary component {worst}")
        root = root.real
    gap = m[0] - m[1]
    return float(gap @ gap + np.trace(S[0]) + np.trace(S[1]) - 2.0 * np.trace(root))


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
