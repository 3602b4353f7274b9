import os, time, torch, subprocess, sys
sys.path.insert(0, '.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
print(subprocess.run("grep -m1 'model name' /proc/cpuinfo; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null", shell=True, capture_output=True, text=True).stdout, flush=True)
from oracle import sched_ref, train_ref
from oracle import unet_ref as U
cfg = U.CIFAR10_32
nt = int(sys.argv[1])
torch.set_num_threads(nt)
P = U.gen_params(cfg, 0)
_, a, ac = sched_ref.make_tables()
B = 16
g = torch.Generator().manual_seed(0)
x0 = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
R = torch.zeros_like(x0); eps = torch.randn(B, 3, 32, 32, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
t0 = time.time()
with torch.no_grad():
    y = U.unet_forward(cfg, P, x0, t)
print(nt, "fwd only", time.time() - t0, flush=True)
for i in range(3):
    t0 = time.time()
    loss, G = train_ref.loss_and_grads(cfg, P, a, ac, x0, R, t, eps)
    print(nt, "fwd+bwd", time.time() - t0, flush=True)
