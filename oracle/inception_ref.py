"""TEST INFRASTRUCTURE (oracle): CPU restatement of the FID feature extractor, pytorch_fid's InceptionV3 up to pool3.

Reference call sites: /root/reference/fid_score.py:53 (`from pytorch_fid.inception import InceptionV3`), :91-148
(get_activations: ToTensor batches of 50 -> `model(batch)[0]` -> [N, 2048]), :255 (`InceptionV3([block_idx])`, dims 2048 ->
block 3).  pytorch_fid (requirements.txt: pytorch-fid==0.2.1) is a third-party dependency that is NOT in /root/reference and not
installed here; this file restates its published network:
  * torchvision's inception_v3 graph (Conv2d_1a_3x3 .. Mixed_7c, aux_logits off, fc 1008 x 2048 unused by pool3), every
    convolution a BasicConv2d = Conv2d(bias=False) + BatchNorm2d(eps=0.001) in inference mode + ReLU;
  * pytorch_fid's patches: Mixed_5b/5c/5d = FIDInceptionA and Mixed_6b..6e = FIDInceptionC (their 3x3 average pool uses
    count_include_pad=False), Mixed_7b = FIDInceptionE_1 (same average pool), Mixed_7c = FIDInceptionE_2 (a 3x3 MAX pool in the
    pool branch);
  * in front: F.interpolate(size=(299, 299), mode="bilinear", align_corners=False), then x -> 2x - 1; behind: adaptive average
    pool to 1 x 1.
PARITY UNPINNED: neither pytorch_fid nor its weights (pt_inception-2015-12-05-6726825d.pth) can be obtained here, so this
restatement is checked against nothing but itself; the product (baddiffusion_amd/inception.py on libbd_hip.so) is compared with
it on seeded random weights, and both against the same key / shape manifest."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 0.001


def _a(pre, cin, pf):
    return [(pre + ".branch1x1", cin, 64, (1, 1)), (pre + ".branch5x5_1", cin, 48, (1, 1)), (pre + ".branch5x5_2", 48, 64, (5, 5)),
            (pre + ".branch3x3dbl_1", cin, 64, (1, 1)), (pre + ".branch3x3dbl_2", 64, 96, (3, 3)), (pre + ".branch3x3dbl_3", 96, 96, (3, 3)),
            (pre + ".branch_pool", cin, pf, (1, 1))]


def _c(pre, cin, c7):
    return [(pre + ".branch1x1", cin, 192, (1, 1)), (pre + ".branch7x7_1", cin, c7, (1, 1)), (pre + ".branch7x7_2", c7, c7, (1, 7)),
            (pre + ".branch7x7_3", c7, 192, (7, 1)), (pre + ".branch7x7dbl_1", cin, c7, (1, 1)), (pre + ".branch7x7dbl_2", c7, c7, (7, 1)),
            (pre + ".branch7x7dbl_3", c7, c7, (1, 7)), (pre + ".branch7x7dbl_4", c7, c7, (7, 1)), (pre + ".branch7x7dbl_5", c7, 192, (1, 7)),
            (pre + ".branch_pool", cin, 192, (1, 1))]


def _e(pre, cin):
    return [(pre + ".branch1x1", cin, 320, (1, 1)), (pre + ".branch3x3_1", cin, 384, (1, 1)), (pre + ".branch3x3_2a", 384, 384, (1, 3)),
            (pre + ".branch3x3_2b", 384, 384, (3, 1)), (pre + ".branch3x3dbl_1", cin, 448, (1, 1)), (pre + ".branch3x3dbl_2", 448, 384, (3, 3)),
            (pre + ".branch3x3dbl_3a", 384, 384, (1, 3)), (pre + ".branch3x3dbl_3b", 384, 384, (3, 1)), (pre + ".branch_pool", cin, 192, (1, 1))]


def conv_table():
    """(name, Cin, Cout, (kh, kw)) of every BasicConv2d, in state-dict order."""
    t = [("Conv2d_1a_3x3", 3, 32, (3, 3)), ("Conv2d_2a_3x3", 32, 32, (3, 3)), ("Conv2d_2b_3x3", 32, 64, (3, 3)),
         ("Conv2d_3b_1x1", 64, 80, (1, 1)), ("Conv2d_4a_3x3", 80, 192, (3, 3))]
    t += _a("Mixed_5b", 192, 32) + _a("Mixed_5c", 256, 64) + _a("Mixed_5d", 288, 64)
    t += [("Mixed_6a.branch3x3", 288, 384, (3, 3)), ("Mixed_6a.branch3x3dbl_1", 288, 64, (1, 1)), ("Mixed_6a.branch3x3dbl_2", 64, 96, (3, 3)),
          ("Mixed_6a.branch3x3dbl_3", 96, 96, (3, 3))]
    t += _c("Mixed_6b", 768, 128) + _c("Mixed_6c", 768, 160) + _c("Mixed_6d", 768, 160) + _c("Mixed_6e", 768, 192)
    t += [("Mixed_7a.branch3x3_1", 768, 192, (1, 1)), ("Mixed_7a.branch3x3_2", 192, 320, (3, 3)), ("Mixed_7a.branch7x7x3_1", 768, 192, (1, 1)),
          ("Mixed_7a.branch7x7x3_2", 192, 192, (1, 7)), ("Mixed_7a.branch7x7x3_3", 192, 192, (7, 1)), ("Mixed_7a.branch7x7x3_4", 192, 192, (3, 3))]
    t += _e("Mixed_7b", 1280) + _e("Mixed_7c", 2048)
    return t


def manifest():
    """state-dict key -> shape of the FID Inception network (pt_inception-2015-12-05 layout: torchvision inception_v3 names,
    num_classes 1008, no AuxLogits)."""
    m = OrderedDict()
    for name, cin, cout, (kh, kw) in conv_table():
        m[name + ".conv.weight"] = (cout, cin, kh, kw)
        for k in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.bn.{k}"] = (cout,)
    m["fc.weight"] = (1008, 2048)
    m["fc.bias"] = (1008,)
    return m


def gen_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for k, shp in manifest().items():
        if k.endswith("conv.weight"):
            fan_in = shp[1] * shp[2] * shp[3]
            P[k] = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("bn.weight") or k.endswith("running_var"):
            P[k] = torch.rand(shp, generator=g) + 0.5
        elif k.startswith("fc."):
            P[k] = torch.randn(shp, generator=g) * 0.01
        else:
            P[k] = torch.randn(shp, generator=g) * 0.1
    return P


def _bc(P, name, x, stride=1, padding=0):
    x = F.conv2d(x, P[name + ".conv.weight"], None, stride, padding)
    x = F.batch_norm(x, P[name + ".bn.running_mean"], P[name + ".bn.running_var"], P[name + ".bn.weight"], P[name + ".bn.bias"], False, 0.0, BN_EPS)
    return F.relu(x)


def _block_a(P, pre, x):
    b1 = _bc(P, pre + ".branch1x1", x)
    b5 = _bc(P, pre + ".branch5x5_2", _bc(P, pre + ".branch5x5_1", x), padding=2)
    b3 = _bc(P, pre + ".branch3x3dbl_3", _bc(P, pre + ".branch3x3dbl_2", _bc(P, pre + ".branch3x3dbl_1", x), padding=1), padding=1)
    bp = _bc(P, pre + ".branch_pool", F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))
    return torch.cat([b1, b5, b3, bp], 1)


def _block_b(P, pre, x):
    b3 = _bc(P, pre + ".branch3x3", x, stride=2)
    bd = _bc(P, pre + ".branch3x3dbl_3", _bc(P, pre + ".branch3x3dbl_2", _bc(P, pre + ".branch3x3dbl_1", x), padding=1), stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)


def _block_c(P, pre, x):
    b1 = _bc(P, pre + ".branch1x1", x)
    b7 = _bc(P, pre + ".branch7x7_1", x)
    b7 = _bc(P, pre + ".branch7x7_2", b7, padding=(0, 3))
    b7 = _bc(P, pre + ".branch7x7_3", b7, padding=(3, 0))
    bd = _bc(P, pre + ".branch7x7dbl_1", x)
    bd = _bc(P, pre + ".branch7x7dbl_2", bd, padding=(3, 0))
    bd = _bc(P, pre + ".branch7x7dbl_3", bd, padding=(0, 3))
    bd = _bc(P, pre + ".branch7x7dbl_4", bd, padding=(3, 0))
    bd = _bc(P, pre + ".branch7x7dbl_5", bd, padding=(0, 3))
    bp = _bc(P, pre + ".branch_pool", F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))
    return torch.cat([b1, b7, bd, bp], 1)


def _block_d(P, pre, x):
    b3 = _bc(P, pre + ".branch3x3_2", _bc(P, pre + ".branch3x3_1", x), stride=2)
    b7 = _bc(P, pre + ".branch7x7x3_1", x)
    b7 = _bc(P, pre + ".branch7x7x3_2", b7, padding=(0, 3))
    b7 = _bc(P, pre + ".branch7x7x3_3", b7, padding=(3, 0))
    b7 = _bc(P, pre + ".branch7x7x3_4", b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


def _block_e(P, pre, x, max_pool):
    b1 = _bc(P, pre + ".branch1x1", x)
    b3 = _bc(P, pre + ".branch3x3_1", x)
    b3 = torch.cat([_bc(P, pre + ".branch3x3_2a", b3, padding=(0, 1)), _bc(P, pre + ".branch3x3_2b", b3, padding=(1, 0))], 1)
    bd = _bc(P, pre + ".branch3x3dbl_2", _bc(P, pre + ".branch3x3dbl_1", x), padding=1)
    bd = torch.cat([_bc(P, pre + ".branch3x3dbl_3a", bd, padding=(0, 1)), _bc(P, pre + ".branch3x3dbl_3b", bd, padding=(1, 0))], 1)
    pooled = F.max_pool2d(x, 3, 1, 1) if max_pool else F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)
    return torch.cat([b1, b3, bd, _bc(P, pre + ".branch_pool", pooled)], 1)


@torch.no_grad()
def pool3_features(P, images, resize_input=True, normalize_input=True):
    """images [N, 3, H, W] float in [0, 1] (what ToTensor yields, fid_score.py:113) -> [N, 2048] pool3 features."""
    x = images.float()
    if resize_input:
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    if normalize_input:
        x = 2 * x - 1
    x = _bc(P, "Conv2d_1a_3x3", x, stride=2)
    x = _bc(P, "Conv2d_2a_3x3", x)
    x = _bc(P, "Conv2d_2b_3x3", x, padding=1)
    x = F.max_pool2d(x, 3, 2)
    x = _bc(P, "Conv2d_3b_1x1", x)
    x = _bc(P, "Conv2d_4a_3x3", x)
    x = F.max_pool2d(x, 3, 2)
    for pre in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
        x = _block_a(P, pre, x)
    x = _block_b(P, "Mixed_6a", x)
    for pre in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
        x = _block_c(P, pre, x)
    x = _block_d(P, "Mixed_7a", x)
    x = _block_e(P, "Mixed_7b", x, max_pool=False)
    x = _block_e(P, "Mixed_7c", x, max_pool=True)
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
