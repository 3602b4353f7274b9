import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baddiffusion_amd import ops
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, H, Cin, Cout in [(128,32,32,128),(128,32,64,128),(128,32,128,128),(128,32,256,128),(128,32,512,128),(128,32,1024,128),
                        (64,32,128,128),(256,32,128,128),(128,32,128,256),(128,32,128,512)]:
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 30; bias = torch.randn(Cout, device="cuda")
    t = timeit(lambda: ops.conv3x3_fwd(x, w, bias, mode=mode))
    fl = 2.0 * B * H * H * Cout * Cin * 9
    tiles = (B*H*H//128) * (Cout//128)
    print(f"B{B} {H}^2 {Cin}->{Cout}: {t:8.1f} us  {fl/t/1e6:6.1f} TF  tiles {tiles} chunks {9*Cin//32}  us/round/chunk {t/max(1,tiles/512)/(9*Cin/32):.3f}", flush=True)
