import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, H, Cin, Cout in [(128,32,128,128),(64,32,128,128),(128,32,256,128)]:
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 30; bias = torch.randn(Cout, device="cuda")
    t = timeit(lambda: ops.conv3x3_fwd(x, w, bias, mode=1))
    print(f"dbg={os.environ.get('BD_IGEMM_DEBUG','0')} B{B} {Cin}->{Cout}: {t:8.1f} us", flush=True)
