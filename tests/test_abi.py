"""CPU: the C-ABI library builds, loads, and exports exactly what include/bd_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "bd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ("bd_poison_qsample", "bd_ddpm_step", "bd_ddim_step", "bd_gn_fwd", "bd_gn_bwd", "bd_igemm", "bd_conv3x3_fwd",
              "bd_conv3x3_dgrad", "bd_conv3x3_wgrad", "bd_loss_fwd_bwd", "bd_adam_clip", "bd_unet_forward", "bd_unet_backward"):
        assert s in syms


def test_library_builds_and_exports_every_declared_symbol():
    from baddiffusion_amd.build import build_lib
    lib_path = build_lib(force=False, verbose=False)
    assert os.path.exists(lib_path)
    from baddiffusion_amd import _lib as L
    lib = L.load()
    exported = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (bd_[a-z0-9_]+)", exported))
    declared = set(header_symbols())
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(L.SIGNATURES), sorted(declared ^ set(L.SIGNATURES))
    assert lib.bd_version() >= 1


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors vs sizeof() computed by the C compiler from the header."""
    from baddiffusion_amd import _lib as L
    names = {"bd_poison_qsample_desc": L.PoisonQsampleDesc, "bd_qsample_desc": L.QsampleDesc, "bd_ddpm_step_desc": L.DdpmStepDesc,
             "bd_ddim_step_desc": L.DdimStepDesc, "bd_gn_fwd_desc": L.GnFwdDesc, "bd_gn_bwd_desc": L.GnBwdDesc, "bd_gn_param_item": L.GnParamItem,
             "bd_operand": L.Operand, "bd_igemm_desc": L.IgemmDesc, "bd_conv3x3_fwd_desc": L.ConvFwdDesc,
             "bd_conv3x3_dgrad_desc": L.ConvDgradDesc, "bd_conv3x3_wgrad_desc": L.ConvWgradDesc, "bd_unet_config": L.UnetConfig,
             "bd_conv3x3_ps_desc": L.ConvPsDesc, "bd_conv3x3_ps_wgrad_desc": L.ConvPsWgradDesc,
             "bd_upsample_conv_desc": L.UpsampleConvDesc, "bd_conv2d_desc": L.Conv2dDesc, "bd_conv3x3_s2_dgrad_desc": L.ConvS2DgradDesc}
    prog = '#include <stdio.h>\n#include "bd_hip.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        n, sz = line.split()
        assert ctypes.sizeof(names[n]) == int(sz), (n, ctypes.sizeof(names[n]), sz)


def test_error_reporting_without_gpu():
    from baddiffusion_amd import _lib as L
    lib = L.load()
    assert lib.bd_poison_qsample(None, None) == -1
    assert b"null descriptor" in lib.bd_last_error()
    cfg = L.UnetConfig()
    cfg.num_blocks = 0
    h = ctypes.c_void_p()
    assert lib.bd_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"num_blocks" in lib.bd_last_error()


def test_round2_entry_points_reject_bad_arguments_without_gpu():
    """argument checks of the entry points added in round 2 run on the host before any launch: they must fail loudly with a
    message (negative status + bd_last_error), never touch the device"""
    from baddiffusion_amd import _lib as L
    lib = L.load()
    assert lib.bd_lincomb(0, None, None, 16, 0, 0.0, None, None) < 0 and b"bd_lincomb" in lib.bd_last_error()
    assert lib.bd_lincomb(7, None, None, 16, 0, 0.0, None, None) < 0
    assert lib.bd_ssim(None, None, 1, 3, 32, 32, 0, 0, 0, 0, 1.0, None, None, 0, None) < 0 and b"bd_ssim" in lib.bd_last_error()
    assert lib.bd_ssim_workspace_bytes(2, 3, 32, 32) == 2 * 3 * 8 and lib.bd_ssim_workspace_bytes(2, 3, 75, 44) == 2 * 3 * 3 * 2 * 8
    assert lib.bd_gn_bwd_params(None, 0, 4, None) < 0 and b"bd_gn_bwd_params" in lib.bd_last_error()
    assert lib.bd_gn_bwd_defers(128, 1024, 128, 32) == 1 and lib.bd_gn_bwd_defers(4, 65536, 128, 32) == 1
    assert lib.bd_gn_bwd_defers(4, 64, 130, 32) == 0                      # C not divisible by G
    assert lib.bd_conv3x3_ps(None, None) < 0 and lib.bd_conv3x3_ps_wgrad(None, None) < 0
    # round 3: phase-decomposed convolutions, deferred join, probe
    assert lib.bd_upsample_conv_fwd(None, None) < 0 and lib.bd_upsample_conv_dgrad(None, None) < 0 and lib.bd_upsample_conv_wgrad(None, None) < 0
    assert lib.bd_conv3x3_s2_dgrad_ps(None, None) < 0 and b"bd_conv3x3_s2_dgrad_ps" in lib.bd_last_error()
    assert lib.bd_upsample_weights(None, 128, 128, None, None, None) < 0 and b"bd_upsample_weights" in lib.bd_last_error()
    d = L.UpsampleConvDesc(B=2, H=8, W=8, Cin=128, Cout=256)
    assert lib.bd_upsample_conv_wgrad_workspace_bytes(ctypes.byref(d)) >= 4 * 256 * 16 * 128       # at least dE itself
    assert lib.bd_unet_set_deferred_join(None, 1) < 0 and lib.bd_unet_stream_wait_aux(None, None) < 0
    assert lib.bd_unet_set_static_weights(None, 1) < 0
    # round 4: FID feature-extractor kernels
    assert lib.bd_conv2d_nhwc(None, None) < 0 and b"bd_conv2d_nhwc" in lib.bd_last_error()
    assert lib.bd_pool2d_nhwc(None, 4, None, 4, 1, 8, 8, 4, 3, 2, 0, 0, 0, None) < 0
    assert lib.bd_resize_bilinear_nhwc(None, 0, None, 1, 8, 8, 3, 299, 299, 2.0, -1.0, None) < 0
    assert lib.bd_global_avgpool_nhwc(None, 4, None, 1, 64, 4, None) < 0
    assert lib.bd_mfma_probe(1, 0, 1, None, None) < 0
