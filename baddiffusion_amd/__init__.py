"""baddiffusion_amd -- MI355X-native BadDiffusion hot path (UNet2D train/sample step, DDPM/DDIM scheduler
steps, trigger-blend + q_sample, clip + Adam, DP all-reduce) behind the reference's Python API.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all math of the path
runs in hand-written HIP kernels for gfx950 through the C ABI of include/bd_hip.h (libbd_hip.so).
"""
__version__ = "0.1.0"
