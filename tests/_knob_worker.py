"""Worker of tests/test_hip_round4.py::test_round4_paths_equal_their_fallbacks: one CIFAR-topology forward + backward at B = 8 under
whatever BD_* knobs the environment carries; writes the output and the flat gradient to the .pt file named by argv[1]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.unet import UNet2DModel
    torch.manual_seed(0)
    m = UNet2DModel(**KNOWN_TOPOLOGIES[os.environ.get("BD_T_TOPOLOGY", "google/ddpm-cifar10-32")]).cuda()
    S = m.config.sample_size
    B = int(os.environ.get("BD_T_BATCH", "8"))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    dout = (torch.randn(B, 3, S, S, generator=g) / (B * 3 * S * S)).cuda()
    out = m(x, t, return_dict=False)[0]
    out.backward(dout)
    torch.save({"out": out.detach().cpu(), "grad": m.flat.grad.detach().cpu()}, sys.argv[1])


if __name__ == "__main__":
    main()
