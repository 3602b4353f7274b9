"""Stage A of the Winograd F(2x2, 3x3) prototype (task 1): what does the arithmetic cost in accuracy?

Emulates on the CPU, in numpy, exactly what a split-bf16 Winograd kernel would compute for one stride-1 3x3 "same" convolution:
  * weights:  U = G g G^T in fp32, then split  U = hi + lo  (hi = truncated bf16, lo = RNE bf16 of the remainder);
  * input:    V = B^T d B in fp32 (adds only), then the same split;
  * products: hi*hi + hi*lo + lo*hi, fp32 accumulation over the channels (the three v_mfma_f32_32x32x16_bf16 passes);
  * output:   Y = A^T M A in fp32.
and compares with the fp64 convolution, next to the direct split-bf16 convolution (what conv_ps3_kernel computes) on the same data.
Gate of the task: <= 2e-5 relative to fp64.
"""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def split(x):
    """fp32 -> (hi, lo) as fp32 arrays holding bf16 values: hi truncated, lo = RNE bf16 of x - hi"""
    x = x.astype(np.float32)
    hi = (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r = (x - hi).astype(np.float32)
    u = r.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return hi, u.view(np.float32)


def prod3(ah, al, bh, bl):
    """sum over the last axis of a (.., K) and b (.., K): three bf16 passes, fp32 accumulate (numpy float32 matmul order)"""
    f = np.float32
    return (np.einsum("...k,...k->...", al.astype(f), bh.astype(f), dtype=f) + np.einsum("...k,...k->...", ah.astype(f), bl.astype(f), dtype=f)
            + np.einsum("...k,...k->...", ah.astype(f), bh.astype(f), dtype=f)).astype(f)


def run(H, W, C, N, seed, xscale=1.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((H + 2, W + 2, C)) * xscale).astype(np.float32)
    x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0                               # zero padding
    w = (rng.uniform(-1, 1, (N, 3, 3, C)) / np.sqrt(9 * C)).astype(np.float32)
    # fp64 reference
    ref = np.zeros((H, W, N))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("hwc,nc->hwn", x[ky:ky + H, kx:kx + W].astype(np.float64), w[:, ky, kx].astype(np.float64))
    # direct split-bf16 (conv_ps3)
    xh, xl = split(x); wh, wl = split(w)
    direct = np.zeros((H, W, N), np.float32)
    for ky in range(3):
        for kx in range(3):
            a_h, a_l = xh[ky:ky + H, kx:kx + W], xl[ky:ky + H, kx:kx + W]
            direct += (np.einsum("hwc,nc->hwn", a_l, wh[:, ky, kx], dtype=np.float32) + np.einsum("hwc,nc->hwn", a_h, wl[:, ky, kx], dtype=np.float32)
                       + np.einsum("hwc,nc->hwn", a_h, wh[:, ky, kx], dtype=np.float32))
    # Winograd, split-bf16 products
    U = np.einsum("ia,nabc,jb->nijc", G, w.astype(np.float64), G).astype(np.float32)          # fp64 transform rounded once (a kernel would do fp32 adds: same to 1 ulp)
    Uh, Ul = split(U)
    th, tw = H // 2, W // 2
    d = np.stack([np.stack([x[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for tx in range(tw)]) for ty in range(th)])   # [th, tw, 4, 4, C]
    f = np.float32
    t1 = np.einsum("ia,yxabc->yxibc", BT.astype(f), d, dtype=f)             # fp32 adds
    V = np.einsum("yxibc,jb->yxijc", t1, BT.astype(f), dtype=f)
    Vh, Vl = split(V)
    M = (np.einsum("yxijc,nijc->yxijn", Vl, Uh, dtype=f) + np.einsum("yxijc,nijc->yxijn", Vh, Ul, dtype=f) + np.einsum("yxijc,nijc->yxijn", Vh, Uh, dtype=f))
    t2 = np.einsum("pi,yxijn->yxpjn", AT.astype(f), M, dtype=f)
    Y = np.einsum("yxpjn,qj->yxpqn", t2, AT.astype(f), dtype=f)             # [th, tw, 2, 2, N]
    wino = Y.transpose(0, 2, 1, 3, 4).reshape(H, W, N)
    # Winograd with exact fp32 products (what the transforms alone cost)
    Mx = np.einsum("yxijc,nijc->yxijn", V.astype(np.float64), U.astype(np.float64))
    Yx = np.einsum("pi,yxijn,qj->yxpqn", AT, Mx, AT).transpose(0, 2, 1, 3, 4).reshape(H, W, N)
    nrm = np.linalg.norm(ref)
    mx = np.abs(ref).max()
    out = {}
    for tag, y in (("direct_bf16x3", direct), ("winograd_bf16x3", wino), ("winograd_exact_products", Yx)):
        e = y.astype(np.float64) - ref
        out[tag] = (np.linalg.norm(e) / nrm, np.abs(e).max() / mx)
    return out


if __name__ == "__main__":
    import json
    for (H, W, C, N, sc) in ((8, 8, 128, 128, 1.0), (16, 16, 256, 64, 1.0), (8, 8, 512, 64, 1.0), (8, 8, 128, 64, 30.0)):
        r = run(H, W, C, N, 0, sc)
        print(f"H={H} W={W} C={C} N={N} xscale={sc}: " + "  ".join(f"{k}: rel-L2 {v[0]:.2e} max/max {v[1]:.2e}" for k, v in r.items()))
