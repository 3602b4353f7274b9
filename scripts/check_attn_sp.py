"""bd_attn_sp_fwd / bd_attn_sp_bwd (attention core on split planes) against an fp64 reference; then `reps` timed-by-rocprof launches.
usage: python scripts/check_attn_sp.py [B] [reps]"""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, C, heads = 256, 256, 1
dev = "cuda"
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * C, device=dev)
do = torch.randn(B * N, C, device=dev)
scale = C ** -0.5
qs, dos = ops.split_rows(qkv), ops.split_rows(do)
o_s, pt_s = ops.attn_sp_fwd(qs, B, heads, scale)
dqkv_s, dst_s = ops.attn_sp_bwd(qs, pt_s, dos, B, heads, scale)
torch.cuda.synchronize()
D = lambda t: t.double()
q, k, v = (D(qkv[:, i * C:(i + 1) * C]).view(B, N, C) for i in range(3))
S = scale * q @ k.transpose(1, 2)
P = torch.softmax(S, -1)
O = P @ v
dO = D(do).view(B, N, C)
dV = P.transpose(1, 2) @ dO
dP = dO @ v.transpose(1, 2)
dS = P * (dP - (P * dP).sum(-1, keepdim=True))
dQ = scale * dS @ k
dK = scale * dS.transpose(1, 2) @ q
bad = 0
def check(name, got, ref, tol=5e-5):
    global bad
    e = float((got.double() - ref).abs().max() / ref.abs().max())
    bad += not (e < tol)
    print(f"{name:10s} rel err {e:.2e}{'' if e < tol else '   <-- FAIL'}", flush=True)
check("O", ops.unsplit_rows(o_s).view(B, N, C), O)
check("P^T", ops.unsplit_rows(pt_s), P.transpose(1, 2))
dqkv = ops.unsplit_rows(dqkv_s).view(B, N, 3 * C)
check("dS^T", ops.unsplit_rows(dst_s), scale * dS.transpose(1, 2))
check("dQ", dqkv[..., :C], dQ)
check("dK", dqkv[..., C:2 * C], dK)
check("dV", dqkv[..., 2 * C:], dV)
for _ in range(reps):
    o_s, pt_s = ops.attn_sp_fwd(qs, B, heads, scale)
    ops.attn_sp_bwd(qs, pt_s, dos, B, heads, scale)
torch.cuda.synchronize()
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
