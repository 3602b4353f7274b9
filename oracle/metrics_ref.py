"""CPU restatement of the measure path's statistics (test infrastructure only, never imported by the product).

* activation_statistics -- fid_score.py:207-230 (`calculate_activation_statistics`): mean and np.cov(rowvar=False)
* frechet_distance      -- fid_score.py:150-204 (`calculate_frechet_distance`): d^2 = |mu1-mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)),
                           with the eps-on-the-diagonal retry when the product is singular and the real-part rule
Pinned by tests/golden/fid.npz (G8, captured by importing the reference: tests/golden/make_golden.py g8).
* ssim_ref / mse_ref    -- baddiffusion.py:260,545-546: `nn.MSELoss(reduction='mean')` and torchmetrics'
                           `StructuralSimilarityIndexMeasure(data_range=1.0)` on [N,C,H,W] images in [0,1].  torchmetrics is a
                           third-party dependency absent from /root/reference and from this image (requirements.txt pins no
                           version), so this restates the published algorithm from its documented defaults -- Gaussian window
                           11x11, sigma 1.5, k1 0.01, k2 0.03, reflect padding of 5 px, the padded border cropped, mean over the
                           map -- in fp64 with scipy.ndimage (a library and formulation the product does not share).
                           PARITY UNPINNED against torchmetrics itself: no golden vector can be produced here.
"""
import numpy as np
from scipy import linalg


def mse_ref(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(((a - b) ** 2).mean())


def ssim_ref(a, b, data_range=1.0, sigma=1.5, k1=0.01, k2=0.03):
    """mean SSIM of two [N,C,H,W] arrays; kernel = 11 taps (truncate 3.5 at sigma 1.5), mirror boundary == reflect padding"""
    from scipy import ndimage
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)

    def blur(x):
        return ndimage.gaussian_filter(x, sigma=(0, 0, sigma, sigma), truncate=3.5, mode="mirror")
    ma, mb = blur(a), blur(b)
    va, vb, cab = blur(a * a) - ma * ma, blur(b * b) - mb * mb, blur(a * b) - ma * mb
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    m = ((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))
    return float(m[..., 5:-5, 5:-5].reshape(m.shape[0], -1).mean(-1).mean())


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
