"""The product never routes through the oracle or a CPU fallback: static checks on the package sources."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "baddiffusion_amd")


def _py_files():
    out = [os.path.join(ROOT, "baddiffusion.py")]
    for d, _, fs in os.walk(PKG):
        out += [os.path.join(d, f) for f in fs if f.endswith(".py")]
    return out


def test_product_never_imports_oracle_or_reference():
    for f in _py_files():
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
        assert "sys.path.insert(0, \"/root/reference" not in src and "import diffusers" not in src, f


def test_no_torch_compute_fallback_for_network_ops():
    """No aten conv / group_norm / attention in the product: those ops exist only as HIP kernels."""
    banned = ("F.conv2d", "nn.Conv2d", "F.group_norm", "nn.GroupNorm", "scaled_dot_product_attention", "F.linear(", "nn.Linear(",
              "torch.bmm", "torch.softmax", "F.silu", "F.avg_pool2d", "F.max_pool2d", "F.interpolate", "F.batch_norm", "nn.BatchNorm2d",
              "adaptive_avg_pool2d")
    for f in _py_files():
        src = open(f).read()
        code = re.sub(r'""".*?"""', "", src, flags=re.S)
        for b in banned:
            assert b not in code, (f, b)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from baddiffusion_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        L.load()


def test_cpu_tensor_is_rejected():
    import pytest
    import torch
    from baddiffusion_amd import ops
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        ops.silu_fwd(torch.zeros(4))
    from baddiffusion_amd.unet import UNet2DModel
    m = UNet2DModel(sample_size=16, block_out_channels=(128, 256), down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                    up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 16, 16), 3)
