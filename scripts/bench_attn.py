"""Launch the twelve GEMMs (and two softmax passes) of one attention block at the CIFAR step's shape through ops.gemm, each kind
`reps` times with a silu marker launch between kinds; run under `rocprofv3 --kernel-trace --output-format csv` and feed the
kernel_trace.csv to scripts/attn_trace.py, which prints the average duration per kind.
usage: python scripts/bench_attn.py [B] [reps]"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, C = 256, 256
M = B * N
dev = "cuda"
n = torch.randn(M, C, device=dev); wqkv = torch.randn(3 * C, C, device=dev) * 0.05; bqkv = torch.randn(3 * C, device=dev)
wp = torch.randn(C, C, device=dev) * 0.05; bp = torch.randn(C, device=dev)
q = torch.randn(B, N, C, device=dev); k = torch.randn(B, N, C, device=dev); v = torch.randn(B, N, C, device=dev)
p = torch.softmax(torch.randn(B, N, N, device=dev), -1); dp = torch.randn(B, N, N, device=dev)
o = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev); dqkv = torch.randn(M, 3 * C, device=dev)
do = dy.view(B, N, C)
mark = torch.randn(64, device=dev)
kinds = [
    ("qkv_fwd      M=32768 N=768 K=256 NT", lambda: ops.gemm(n, wqkv, bias=bqkv, mode=1)),
    ("QKt          b128 256x256x256 NT   ", lambda: ops.gemm(q, k, mode=1)),
    ("softmax_fwd                         ", lambda: ops.softmax_fwd(dp)),
    ("PV           b128 256x256x256 NN   ", lambda: ops.gemm(p, v, trans_b=False, mode=1)),
    ("proj_fwd     M=32768 N=256 K=256 NT", lambda: ops.gemm(o, wp, bias=bp, mode=1)),
    ("proj_wgrad   M=256 N=256 K=32768 TN", lambda: ops.gemm(dy, o, trans_a=True, trans_b=False, mode=1)),
    ("proj_dgrad   M=32768 N=256 K=256 NN", lambda: ops.gemm(dy, wp, trans_b=False, mode=1)),
    ("dP=dO Vt     b128 NT               ", lambda: ops.gemm(do, v, mode=1)),
    ("dV=Pt dO     b128 TN               ", lambda: ops.gemm(p, do, trans_a=True, trans_b=False, mode=1)),
    ("softmax_bwd                         ", lambda: ops.softmax_bwd(p, dp)),
    ("dQ=dS K      b128 NN               ", lambda: ops.gemm(dp, k, trans_b=False, mode=1)),
    ("dK=dSt Q     b128 TN               ", lambda: ops.gemm(dp, q, trans_a=True, trans_b=False, mode=1)),
    ("qkv_wgrad    M=768 N=256 K=32768 TN", lambda: ops.gemm(dqkv, n, trans_a=True, trans_b=False, mode=1)),
    ("qkv_dgrad    M=32768 N=256 K=768 NN", lambda: ops.gemm(dqkv, wqkv, trans_b=False, mode=1)),
]
S = ops.split_rows
ns, wqs, wps, qs, ks, vs, ps_, dps, os_, dys, dqs = S(n), S(wqkv), S(wp), S(q), S(k), S(v), S(p), S(dp), S(o), S(dy), S(dqkv)
G = ops.gemm_sp
sc = C ** -0.5
kinds += [
    ("sp qkv_fwd   NT -> split         ", lambda: G(ns, wqs, M, 3 * C, C, bias=bqkv, want_f32=False, want_split=True)),
    ("sp QKt       NT -> f32           ", lambda: G(qs, ks, N, N, C, batch=B, alpha=sc)),
    ("sp PV        NN -> split         ", lambda: G(ps_, vs, N, C, N, b_kmajor=True, batch=B, want_f32=False, want_split=True)),
    ("sp proj_fwd  NT -> f32 +res      ", lambda: G(os_, wps, M, C, C, bias=bp, residual=dy)),
    ("sp proj_wgrad TN splitK +colsum  ", lambda: G(dys, os_, C, C, M, a_kmajor=True, b_kmajor=True, want_colsum=True)),
    ("sp proj_dgrad NN -> split        ", lambda: G(dys, wps, M, C, C, b_kmajor=True, want_f32=False, want_split=True)),
    ("sp dP        NT -> f32           ", lambda: G(dys, vs, N, N, C, batch=B)),
    ("sp dV        TN -> split         ", lambda: G(ps_, dys, N, C, N, a_kmajor=True, b_kmajor=True, batch=B, want_f32=False, want_split=True)),
    ("sp dQ        NN -> split         ", lambda: G(dps, ks, N, C, N, b_kmajor=True, batch=B, alpha=sc, want_f32=False, want_split=True)),
    ("sp dK        TN -> split         ", lambda: G(dps, qs, N, C, N, a_kmajor=True, b_kmajor=True, batch=B, alpha=sc, want_f32=False, want_split=True)),
    ("sp qkv_wgrad TN splitK +colsum   ", lambda: G(dqs, ns, 3 * C, C, M, a_kmajor=True, b_kmajor=True, want_colsum=True)),
    ("sp qkv_dgrad NN -> f32           ", lambda: G(dqs, wqs, M, C, 3 * C, b_kmajor=True)),
]
for name, fn in kinds:
    fn(); torch.cuda.synchronize()
for name, fn in kinds:
    ops.silu_fwd(mark)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
ops.silu_fwd(mark)
torch.cuda.synchronize()
print("KINDS " + "|".join(kn for kn, _ in kinds))
