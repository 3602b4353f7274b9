"""bd_gemm_sp (split-plane GEMM) against fp64 matmul at the attention block's twelve shapes; prints max |err| / max |ref|.
usage: python scripts/check_gemm_sp.py [B]"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, C = 256, 256
M = B * N
dev = "cuda"
torch.manual_seed(0)
def rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())
n = torch.randn(M, C, device=dev); wqkv = torch.randn(3 * C, C, device=dev) * 0.05; bqkv = torch.randn(3 * C, device=dev)
wp = torch.randn(C, C, device=dev) * 0.05; bp = torch.randn(C, device=dev)
q = torch.randn(B, N, C, device=dev); k = torch.randn(B, N, C, device=dev); v = torch.randn(B, N, C, device=dev)
p = torch.softmax(torch.randn(B, N, N, device=dev), -1); dp = torch.randn(B, N, N, device=dev)
o = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev); dqkv = torch.randn(M, 3 * C, device=dev)
res = torch.randn(M, C, device=dev)
S = ops.split_rows
ns, wqs, wps, qs, ks, vs, ps_, dps, os_, dys, dqs = S(n), S(wqkv), S(wp), S(q), S(k), S(v), S(p), S(dp), S(o), S(dy), S(dqkv)
D = lambda t: t.double()
bad = 0
def check(name, got, ref, tol=2e-5):
    global bad
    e = rel(got, ref)
    flag = "" if e < tol else "   <-- FAIL"
    bad += e >= tol
    print(f"{name:28s} rel err {e:.2e}{flag}", flush=True)

c, cs = ops.gemm_sp(ns, wqs, M, 3 * C, C, bias=bqkv, want_split=True)
ref = D(n) @ D(wqkv).T + D(bqkv)
check("qkv_fwd NT f32", c[0], ref); check("qkv_fwd NT split", ops.unsplit_rows(cs)[0], ref)
c, _ = ops.gemm_sp(qs, ks, N, N, C, batch=B, alpha=0.0625)
check("QKt NT batched", c, 0.0625 * D(q) @ D(k).transpose(1, 2))
c, cs = ops.gemm_sp(ps_, vs, N, C, N, b_kmajor=True, batch=B, want_split=True)
check("PV NN batched", c, D(p) @ D(v)); check("PV NN batched split", ops.unsplit_rows(cs), D(p) @ D(v))
c, _ = ops.gemm_sp(os_, wps, M, C, C, bias=bp, residual=res, out_scale=0.7)
check("proj_fwd NT +res", c[0], (D(o) @ D(wp).T + D(bp) + D(res)) * 0.7)
c, _, cb = ops.gemm_sp(dys, os_, C, C, M, a_kmajor=True, b_kmajor=True, want_colsum=True)
check("proj_wgrad TN splitK", c[0], D(dy).T @ D(o), 5e-5); check("proj_wgrad colsum", cb, D(dy).sum(0), 5e-5)
c, _ = ops.gemm_sp(dys, wps, M, C, C, b_kmajor=True)
check("proj_dgrad NN", c[0], D(dy) @ D(wp))
c, _ = ops.gemm_sp(dys, vs, N, N, C, batch=B)
check("dP NT batched", c, D(dy).view(B, N, C) @ D(v).transpose(1, 2))
c, _ = ops.gemm_sp(ps_, dys, N, C, N, a_kmajor=True, b_kmajor=True, batch=B)
check("dV TN batched", c, D(p).transpose(1, 2) @ D(dy).view(B, N, C))
c, _ = ops.gemm_sp(dps, ks, N, C, N, b_kmajor=True, batch=B, alpha=0.0625)
check("dQ NN batched", c, 0.0625 * D(dp) @ D(k))
c, _ = ops.gemm_sp(dps, qs, N, C, N, a_kmajor=True, b_kmajor=True, batch=B, alpha=0.0625)
check("dK TN batched", c, 0.0625 * D(dp).transpose(1, 2) @ D(q))
c, _, cb = ops.gemm_sp(dqs, ns, 3 * C, C, M, a_kmajor=True, b_kmajor=True, want_colsum=True)
check("qkv_wgrad TN splitK", c[0], D(dqkv).T @ D(n), 5e-5); check("qkv_wgrad colsum", cb, D(dqkv).sum(0), 5e-5)
c, _ = ops.gemm_sp(dqs, wqs, M, C, 3 * C, b_kmajor=True)
check("qkv_dgrad NN", c[0], D(dqkv) @ D(wqkv))
acc0 = torch.randn(1, M, C, device=dev)
c, _ = ops.gemm_sp(dqs, wqs, M, C, 3 * C, b_kmajor=True, out=acc0.clone(), accumulate=True)
check("qkv_dgrad NN accumulate", c[0], D(dqkv) @ D(wqkv) + D(acc0[0]))
a_km = S(q.transpose(1, 2).contiguous())    # [B, C, N]: K-major A with a K-contiguous B
c, _ = ops.gemm_sp(a_km, ks, N, N, C, a_kmajor=True, batch=B)
check("TN' A K-major, B K-contig", c, D(q) @ D(k).transpose(1, 2))
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
