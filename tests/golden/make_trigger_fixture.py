"""Image-file triggers / targets of the reference (dataset.py:428-497, 576-597, 643-655) as golden vectors.

Run once, here (CPU), from the repo root:   python tests/golden/make_trigger_fixture.py
It imports the REFERENCE's Backdoor class and lets it read the reference's own static/ assets (they stay where they are:
nothing of them is copied), with cwd = /root/reference because the asset paths in dataset.py are relative.

torchvision is not installed in this container.  The six transforms dataset.py uses on this path are thin wrappers over
PIL / torch calls, and this script supplies exactly those calls as a stand-in module (documented per class below: what
torchvision 0.15 does for a PIL input).  So the fixture pins everything the REFERENCE computes around them -- channel
conversion order, the resize target of an int / [H, W] size, the padding arithmetic incl. the negative-offset branch,
`trig >= 0.999 -> vmin`, bg2grey -- and is only as good as the stand-in for the resampling itself (PIL BILINEAR, which is
what torchvision hands a PIL image to).  tests/test_host.py::test_image_file_triggers_match_reference_fixture compares
baddiffusion_amd.dataset.Backdoor with these vectors whenever the assets are reachable (skipped on the GPU box).
Outputs only: tests/golden/img_triggers.npz.
"""
import os, sys, types
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from unittest.mock import MagicMock

# ---- stand-in for torchvision.transforms (PIL-input behaviour of torchvision 0.15) ---------------------------------
tv = types.ModuleType("torchvision.transforms")
class Compose:                                   # transforms.Compose: apply in order
    def __init__(self, ts): self.ts = list(ts)
    def __call__(self, x):
        for t in self.ts: x = t(x)
        return x
class Lambda:                                    # transforms.Lambda
    def __init__(self, fn): self.fn = fn
    def __call__(self, x): return self.fn(x)
class Grayscale:                                 # F_pil.to_grayscale(num_output_channels=1): img.convert("L")
    def __init__(self, num_output_channels=1): assert num_output_channels == 1
    def __call__(self, img): return img.convert("L")
class Resize:                                    # F_pil.resize: int -> shorter side = size, long = int(size * long / short);
    def __init__(self, size): self.size = size  # sequence -> (h, w); img.resize((w, h), BILINEAR) (the default interpolation)
    def __call__(self, img):
        w, h = img.size
        if isinstance(self.size, int):
            short, long = (w, h) if w <= h else (h, w)
            ns, nl = self.size, int(self.size * long / short)
            nw, nh = (ns, nl) if w <= h else (nl, ns)
        else:
            nh, nw = self.size
        return img.resize((nw, nh), Image.BILINEAR)
class ToTensor:                                  # F.to_tensor: uint8 HWC -> float CHW / 255
    def __call__(self, img):
        a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
        a = a[:, :, None] if a.dim() == 2 else a
        return a.permute(2, 0, 1).float().div(255)
class Pad:                                       # F_t.pad(constant) on a tensor: [left, top, right, bottom]
    def __init__(self, padding, fill=0): self.p, self.fill = padding, fill
    def __call__(self, x):
        l, t, r, b = self.p
        return torch.nn.functional.pad(x, [l, r, t, b], value=self.fill)
for k, v in dict(Compose=Compose, Lambda=Lambda, Grayscale=Grayscale, Resize=Resize, ToTensor=ToTensor, Pad=Pad,
                 ToPILImage=MagicMock(), CenterCrop=MagicMock()).items():
    setattr(tv, k, v)
import datasets  # noqa: the real HF datasets package, imported before the stand-ins exist
tvroot = types.ModuleType("torchvision"); tvroot.transforms = tv
sys.modules["torchvision"] = tvroot; sys.modules["torchvision.transforms"] = tv
for n in ("torchvision.utils", "torchvision.datasets", "comet_ml", "wandb", "torchmetrics"):
    sys.modules[n] = MagicMock()
sys.path.insert(0, "/root/reference")
os.chdir("/root/reference")                      # dataset.py opens "static/..." relative to the cwd
import dataset as ref_dataset

bd = ref_dataset.Backdoor(root="/tmp")
out = {}
for name, ch, size in [("GLASSES", 3, 64), ("GLASSES", 3, 256), ("STOP_SIGN_14", 3, 32), ("STOP_SIGN_8", 3, 32), ("STOP_SIGN_14", 1, 32)]:
    out[f"trigger_{name}_c{ch}_s{size}"] = bd.get_trigger(type=name, channel=ch, image_size=size).numpy()
for name, ch, size in [("HAT", 3, 32), ("CAT", 3, 64), ("HAT", 1, 32), ("CAT", 3, 256)]:
    trig = bd.get_trigger(type="BOX_14", channel=ch, image_size=size)
    out[f"target_{name}_c{ch}_s{size}"] = bd.get_target(type=name, trigger=trig).numpy()
path = os.path.join(HERE, "img_triggers.npz")
np.savez_compressed(path, **out)
print({k: v.shape for k, v in out.items()}, f"{os.path.getsize(path) / 1024:.1f} KiB")
