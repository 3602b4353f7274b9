"""CPU restatement of the measure path's statistics (test infrastructure only, never imported by the product).

* activation_statistics -- fid_score.py:207-230 (`calculate_activation_statistics`): mean and np.cov(rowvar=False)
* frechet_distance      -- fid_score.py:150-204 (`calculate_frechet_distance`): d^2 = |mu1-mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)),
                           with the eps-on-the-diagonal retry when the product is singular and the real-part rule
Pinned by tests/golden/fid.npz (G8, captured by importing the reference: tests/golden/make_golden.py g8).
"""
import numpy as np
from scipy import linalg


def activation_statistics(act):
    act = np.asarray(act, dtype=np.float64)
    return act.mean(axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise AssertionError("mean vectors / covariances have different shapes")
    d = mu1 - mu2
    root, _ = linalg.sqrtm(sigma1 @ sigma2, disp=False)
    if not np.isfinite(root).all():
        off = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + off) @ (sigma2 + off))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(root.imag))}")
        root = root.real
    return float(d @ d + np.trace(sigma1) + np.trace(sigma2) - 2.0 * np.trace(root))
