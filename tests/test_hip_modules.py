"""GPU parity of the UNet's building blocks against the G5 vectors captured from the reference's own nn.Modules
(tests/golden/modules.npz: ResnetBlock2D resnet.py:551-601, AttentionBlock attention.py:121-174, Downsample2D
resnet.py:199-208, Upsample2D resnet.py:126-161 -- outputs, input gradients, time-embedding gradients, norm + first 8
values of every parameter gradient).  Each block is composed from the C-ABI entry points exactly as the plan
(csrc/unet_plan.cpp) wires them, once through the igemm convolutions (exact fp32 and split-bf16) and once through the
LDS-DMA split-plane family (bd_conv3x3_ps / bd_conv3x3_ps_wgrad / bd_split_rows / bd_split_wt); the 16 x 16 attention blocks also through
bd_gemm_sp / bd_attn_sp_fwd / bd_attn_sp_bwd against tests/golden/attn_planes.npz (G11).
Tolerance 1e-3 relative (north_star); observed ~1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden import cases as C

G, EPS = 32, 1e-6


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import baddiffusion_amd.ops as o
    return o


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def w_hwio(w):   # OIHW -> [O, kh, kw, I]
    return w.permute(0, 2, 3, 1).contiguous().cuda()


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().cpu().double()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_grads(grads, g, name, tol=1e-3):
    """grads: key -> tensor in the reference's logical shape"""
    big = max(float(g[f"{name}_gn_{k}"]) for k in grads)
    for k, v in grads.items():
        gn = float(v.double().norm())
        ref = float(g[f"{name}_gn_{k}"])
        assert abs(gn - ref) <= tol * max(ref, 1e-3 * big), (name, k, gn, ref)   # key.bias gradients are mathematically 0: rounding noise
        v8 = v.detach().cpu().flatten()[:8].double().numpy()
        np.testing.assert_allclose(v8, g[f"{name}_g8_{k}"][: len(v8)], rtol=2e-3, atol=2e-3 * max(float(np.abs(g[f"{name}_g8_{k}"]).max()), 1e-3 * big))


class Convs:
    """3x3 stride-1 convolution trio behind one interface: igemm (mode 0 / 1) or the split-plane LDS-DMA kernels"""

    def __init__(self, ops, path):
        self.ops, self.path = ops, path

    def fwd(self, x, w, bias, **kw):
        o = self.ops
        if self.path == "ps":
            B, H, W, Cin = x.shape
            return o.conv3x3_ps(o.split_rows(x), o.split_bf16(w), B, H, W, Cin, w.shape[0], 1, bias=bias, **kw)
        return o.conv3x3_fwd(x, w, bias, mode=self.path, **kw)

    def dgrad(self, dy, w, xshape):
        o = self.ops
        if self.path == "ps":
            B, H, W, Cin = xshape
            return o.conv3x3_ps(o.split_rows(dy), o.split_wT(w), B, H, W, w.shape[0], Cin, -1)
        return o.conv3x3_dgrad(dy, w, xshape, mode=self.path)

    def wgrad(self, x, dy):
        o = self.ops
        if self.path == "ps":
            B, H, W, Cin = x.shape
            return o.conv3x3_ps_wgrad(o.split_rows(x), o.split_rows(dy), B, H, W, Cin, dy.shape[-1], with_db=True)
        return o.conv3x3_wgrad(x, dy, mode=self.path, with_db=True)


@pytest.mark.parametrize("path", [0, 1, "ps"])
@pytest.mark.parametrize("name", list(C.RESNET_CASES))
def test_resnet_block_vs_reference(ops, golden, name, path):
    g = golden("modules")
    cin, cout, hw = C.RESNET_CASES[name]
    P = {k: v.cuda() for k, v in C.module_params(name).items()}
    x, temb, dy = C.resnet_inputs(name)
    B = x.shape[0]
    cv = Convs(ops, path)
    mode = 0 if path == 0 else 1
    xh, dyh, tc = nhwc(x), nhwc(dy), temb.cuda()
    w1, w2 = w_hwio(P["conv1.weight"]), w_hwio(P["conv2.weight"])
    # ---- forward (resnet.py:559-601)
    a1, m1, r1 = ops.gn_fwd(xh.reshape(B, hw * hw, cin), P["norm1.weight"], P["norm1.bias"], G, EPS, True)
    st = ops.silu_fwd(tc)
    tp = ops.gemm(st, P["time_emb_proj.weight"], bias=P["time_emb_proj.bias"], mode=mode)                  # [B, cout]
    h1 = cv.fwd(a1.reshape(B, hw, hw, cin), w1, P["conv1.bias"], rowbias=tp)
    a2, m2, r2 = ops.gn_fwd(h1.reshape(B, hw * hw, cout), P["norm2.weight"], P["norm2.bias"], G, EPS, True)
    if cin != cout:
        ws = P["conv_shortcut.weight"].reshape(cout, cin)
        sc = ops.gemm(xh.reshape(-1, cin), ws, bias=P["conv_shortcut.bias"], mode=mode).reshape(B, hw, hw, cout)
    else:
        sc = xh
    y = cv.fwd(a2.reshape(B, hw, hw, cout), w2, P["conv2.bias"], residual=sc, out_scale=1.0)
    assert rel(nchw(y), g[f"{name}_y"]) < 1e-4
    # ---- backward
    dw2, db2 = cv.wgrad(a2.reshape(B, hw, hw, cout), dyh)
    da2 = cv.dgrad(dyh, w2, (B, hw, hw, cout))
    dh1, dg2, dbeta2, cs = ops.gn_bwd(h1.reshape(B, hw * hw, cout), P["norm2.weight"], P["norm2.bias"], m2, r2,
                                      da2.reshape(B, hw * hw, cout), G, True, with_colsum=True)
    dwt = ops.gemm(cs, st, trans_a=True, trans_b=False, mode=mode)                                         # [cout, 512]
    dbt = cs.sum(0)
    dtemb = ops.silu_bwd(tc, ops.gemm(cs, P["time_emb_proj.weight"], trans_b=False, mode=mode))
    dw1, db1 = cv.wgrad(a1.reshape(B, hw, hw, cin), dh1.reshape(B, hw, hw, cout))
    da1 = cv.dgrad(dh1.reshape(B, hw, hw, cout), w1, (B, hw, hw, cin))
    dx, dg1, dbeta1 = ops.gn_bwd(xh.reshape(B, hw * hw, cin), P["norm1.weight"], P["norm1.bias"], m1, r1, da1.reshape(B, hw * hw, cin), G, True)
    dx = dx.reshape(B, hw, hw, cin)
    grads = {"norm1.weight": dg1, "norm1.bias": dbeta1, "conv1.weight": dw1.permute(0, 3, 1, 2), "conv1.bias": db1,
             "time_emb_proj.weight": dwt, "time_emb_proj.bias": dbt, "norm2.weight": dg2, "norm2.bias": dbeta2,
             "conv2.weight": dw2.permute(0, 3, 1, 2), "conv2.bias": db2}
    if cin != cout:
        dyr = dyh.reshape(-1, cout)
        grads["conv_shortcut.weight"] = ops.gemm(dyr, xh.reshape(-1, cin), trans_a=True, trans_b=False, mode=mode).reshape(cout, cin, 1, 1)
        grads["conv_shortcut.bias"] = dyr.sum(0)
        dx = dx + ops.gemm(dyr, ws, trans_b=False, mode=mode).reshape(B, hw, hw, cin)
    else:
        dx = dx + dyh
    assert rel(nchw(dx), g[f"{name}_dx"]) < 1e-3 and rel(dtemb, g[f"{name}_dtemb"]) < 1e-3
    check_grads({k: v.contiguous() for k, v in grads.items()}, g, name)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name", list(C.ATTN_CASES))
def test_attention_block_vs_reference(ops, golden, name, mode):
    g = golden("modules")
    Cc, hw, hd = C.ATTN_CASES[name]
    heads = 1 if hd is None else Cc // hd
    dh = Cc // heads
    N = hw * hw
    P = {k: v.cuda() for k, v in C.module_params(name).items()}
    x, dy = C.attn_inputs(name)
    B = x.shape[0]
    xh, dyh = nhwc(x).reshape(B, N, Cc), nhwc(dy).reshape(B, N, Cc)
    scale = 1.0 / (Cc / heads) ** 0.5
    n, m0, r0 = ops.gn_fwd(xh, P["group_norm.weight"], P["group_norm.bias"], G, EPS, False)
    nr = n.reshape(B * N, Cc)
    lin = lambda a, nm: ops.gemm(a, P[nm + ".weight"], bias=P[nm + ".bias"], mode=mode)
    split = lambda t: t.reshape(B, N, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, N, dh).contiguous()
    merge = lambda t: t.reshape(B, heads, N, dh).permute(0, 2, 1, 3).reshape(B * N, Cc).contiguous()
    q, k, v = (split(lin(nr, nm)) for nm in ("query", "key", "value"))
    S = ops.gemm(q, k, alpha=scale, mode=mode)                       # [B*heads, N, N]   (attention.py:148-156)
    Pm = ops.softmax_fwd(S)
    O = merge(ops.gemm(Pm, v, trans_b=False, mode=mode))
    y = (lin(O, "proj_attn").reshape(B, N, Cc) + xh)                 # rescale_output_factor = 1
    assert rel(nchw(y.reshape(B, hw, hw, Cc)), g[f"{name}_y"]) < 1e-4
    # ---- backward
    dyr = dyh.reshape(B * N, Cc)
    grads = {"proj_attn.weight": ops.gemm(dyr, O, trans_a=True, trans_b=False, mode=mode), "proj_attn.bias": dyr.sum(0)}
    dO = split(ops.gemm(dyr, P["proj_attn.weight"], trans_b=False, mode=mode))
    dP = ops.gemm(dO, v, mode=mode)
    dV = ops.gemm(Pm, dO, trans_a=True, trans_b=False, mode=mode)
    dS = ops.softmax_bwd(Pm, dP)
    dQ = ops.gemm(dS, k, trans_b=False, alpha=scale, mode=mode)
    dK = ops.gemm(dS, q, trans_a=True, trans_b=False, alpha=scale, mode=mode)
    dn = torch.zeros(B * N, Cc, device="cuda")
    for nm, d in (("query", dQ), ("key", dK), ("value", dV)):
        dm = merge(d)
        grads[nm + ".weight"] = ops.gemm(dm, nr, trans_a=True, trans_b=False, mode=mode)
        grads[nm + ".bias"] = dm.sum(0)
        dn = dn + ops.gemm(dm, P[nm + ".weight"], trans_b=False, mode=mode)
    dx, dgw, dgb = ops.gn_bwd(xh, P["group_norm.weight"], P["group_norm.bias"], m0, r0, dn.reshape(B, N, Cc), G, False)
    grads["group_norm.weight"], grads["group_norm.bias"] = dgw, dgb
    dx = dx + dyh
    assert rel(nchw(dx.reshape(B, hw, hw, Cc)), g[f"{name}_dx"]) < 1e-3
    check_grads(grads, g, name)


@pytest.mark.parametrize("name", list(C.ATTN_SP_CASES))
def test_attention_block_on_split_planes_vs_reference(ops, golden, name):
    """AttentionBlock (attention.py:121-174) forward + backward at 16 x 16 through the split-plane path exactly as the plan wires it
    (unet_plan.cpp node_attention, use_sp): GroupNorm -> planes, bd_gemm_sp (QKV projection, planes out), bd_attn_sp_fwd, bd_gemm_sp (output
    projection + bias + residual); backward: planes of dy, bd_gemm_sp weight / data gradients, bd_attn_sp_bwd, GroupNorm backward --
    against what the reference's own module computed (tests/golden/attn_planes.npz, G11): y, dx, every parameter gradient."""
    g = golden("attn_planes")
    Cc, hw, hd = C.ATTN_SP_CASES[name]
    heads = 1 if hd is None else Cc // hd
    N = hw * hw
    P = {k: v.cuda() for k, v in C.module_params(name).items()}
    x, dy = C.attn_inputs(name)
    B = x.shape[0]
    M = B * N
    xh, dyh = nhwc(x).reshape(B, N, Cc), nhwc(dy).reshape(B, N, Cc)
    scale = 1.0 / (Cc / heads) ** 0.5
    wqkv = torch.cat([P["query.weight"], P["key.weight"], P["value.weight"]]).contiguous()
    bqkv = torch.cat([P["query.bias"], P["key.bias"], P["value.bias"]]).contiguous()
    wq_s, wp_s = ops.split_rows(wqkv), ops.split_rows(P["proj_attn.weight"].contiguous())
    n, m0, r0 = ops.gn_fwd(xh, P["group_norm.weight"], P["group_norm.bias"], G, EPS, False)
    n_s = ops.split_rows(n.reshape(M, Cc))
    _, qkv_s = ops.gemm_sp(n_s, wq_s, M, 3 * Cc, Cc, bias=bqkv, want_f32=False, want_split=True)
    qkv_s = qkv_s.reshape(M, 3 * Cc // 32, 2, 32)
    o_s, pt_s = ops.attn_sp_fwd(qkv_s, B, heads, scale)
    y, _ = ops.gemm_sp(o_s, wp_s, M, Cc, Cc, bias=P["proj_attn.bias"], residual=xh.reshape(1, M, Cc))
    assert rel(nchw(y.reshape(B, hw, hw, Cc)), g[f"{name}_y"]) < 1e-4
    # ---- backward
    dy_s = ops.split_rows(dyh.reshape(M, Cc))
    dwp, _, dbp = ops.gemm_sp(dy_s, o_s, Cc, Cc, M, a_kmajor=True, b_kmajor=True, want_colsum=True)
    _, do_s = ops.gemm_sp(dy_s, wp_s, M, Cc, Cc, b_kmajor=True, want_f32=False, want_split=True)
    dqkv_s, _ = ops.attn_sp_bwd(qkv_s, pt_s, do_s.reshape(M, Cc // 32, 2, 32), B, heads, scale)
    dwqkv, _, dbqkv = ops.gemm_sp(dqkv_s, n_s, 3 * Cc, Cc, M, a_kmajor=True, b_kmajor=True, want_colsum=True)
    dn, _ = ops.gemm_sp(dqkv_s, wq_s, M, Cc, 3 * Cc, b_kmajor=True)
    dx, dgw, dgb = ops.gn_bwd(xh, P["group_norm.weight"], P["group_norm.bias"], m0, r0, dn.reshape(B, N, Cc), G, False)
    dx = dx + dyh
    assert rel(nchw(dx.reshape(B, hw, hw, Cc)), g[f"{name}_dx"]) < 1e-3
    grads = {"proj_attn.weight": dwp[0], "proj_attn.bias": dbp, "group_norm.weight": dgw, "group_norm.bias": dgb}
    for i, nm in enumerate(("query", "key", "value")):
        grads[nm + ".weight"] = dwqkv[0][i * Cc:(i + 1) * Cc]
        grads[nm + ".bias"] = dbqkv[i * Cc:(i + 1) * Cc]
    check_grads(grads, g, name)


@pytest.mark.parametrize("mode", [0, 1])
def test_down_and_up_sample_vs_reference(ops, golden, mode):
    g = golden("modules")
    for name, (Cc, hw, pad) in C.DOWN_CASES.items():
        P = {k: v.cuda() for k, v in C.module_params(name).items()}
        x, dy = C.down_inputs(name)
        xh, dyh, w = nhwc(x), nhwc(dy), w_hwio(P["conv.weight"])
        kw = dict(stride=2, pad=pad, asym=(pad == 0))
        y = ops.conv3x3_fwd(xh, w, P["conv.bias"], mode=mode, **kw)
        assert rel(nchw(y), g[f"{name}_y"]) < 1e-4
        dx = ops.conv3x3_dgrad(dyh, w, tuple(xh.shape), mode=mode, **kw)
        dw, db = ops.conv3x3_wgrad(xh, dyh, mode=mode, with_db=True, **kw)
        assert rel(nchw(dx), g[f"{name}_dx"]) < 1e-3
        check_grads({"conv.weight": dw.permute(0, 3, 1, 2).contiguous(), "conv.bias": db}, g, name)
    for name, (Cc, hw) in C.UP_CASES.items():
        P = {k: v.cuda() for k, v in C.module_params(name).items()}
        x, dy = C.up_inputs(name)
        xh, dyh, w = nhwc(x), nhwc(dy), w_hwio(P["conv.weight"])
        B = xh.shape[0]
        # (a) upsampling folded into the igemm gather; (b) the LDS-DMA form: upsampled input materialised as split planes
        y = ops.conv3x3_fwd(xh, w, P["conv.bias"], ups=1, mode=mode)
        assert rel(nchw(y), g[f"{name}_y"]) < 1e-4
        du = ops.conv3x3_dgrad(dyh, w, tuple(xh.shape), ups=1, mode=mode)
        dw, db = ops.conv3x3_wgrad(xh, dyh, ups=1, mode=mode, with_db=True)
        assert rel(nchw(ops.sum2x2(du)), g[f"{name}_dx"]) < 1e-3
        check_grads({"conv.weight": dw.permute(0, 3, 1, 2).contiguous(), "conv.bias": db}, g, name)
        if mode == 1:
            xus = ops.split_rows_ups2(xh)
            up = xh.repeat_interleave(2, 1).repeat_interleave(2, 2)
            assert torch.equal(xus, ops.split_rows(up))
            y2 = ops.conv3x3_ps(xus, ops.split_bf16(w), B, 2 * hw, 2 * hw, Cc, Cc, 1, bias=P["conv.bias"])
            assert rel(nchw(y2), g[f"{name}_y"]) < 1e-4
            dys = ops.split_rows(dyh)
            du2 = ops.conv3x3_ps(dys, ops.split_wT(w), B, 2 * hw, 2 * hw, Cc, Cc, -1)
            dw2, db2 = ops.conv3x3_ps_wgrad(xus, dys, B, 2 * hw, 2 * hw, Cc, Cc, with_db=True)
            assert rel(nchw(ops.sum2x2(du2)), g[f"{name}_dx"]) < 1e-3
            check_grads({"conv.weight": dw2.permute(0, 3, 1, 2).contiguous(), "conv.bias": db2}, g, name)
