"""The C-ABI library: loads, exports every symbol include/afan_hip.h declares, the ctypes table mirrors the
header, and argument errors are reported without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "afan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(afan_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported(pkg):
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(pkg.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in afan_hip.h but not exported by libafan_hip.so: {missing}"


def test_ctypes_table_matches_header(pkg):
    assert sorted(pkg._lib.SIGNATURES) == _declared()
    txt = open(os.path.join(ROOT, "include", "afan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, (_, args) in pkg._lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, txt, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, ctypes table {len(args)}"


def test_identification(pkg):
    lib = pkg._lib.load()
    assert lib.afan_arch() == b"gfx950"
    assert lib.afan_version() >= 100


def test_argument_errors_without_gpu(pkg):
    lib = pkg._lib.load()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    odd = ctypes.c_void_p(p.value + 1)
    assert lib.afan_pgd_step(p, p, 0, None, None, -1, 0.1, 0.1, 0, None) == -3        # AFAN_ESHAPE
    assert lib.afan_pgd_step(p, p, 0, None, None, 0, 0.1, 0.1, 0, None) == 0          # empty input: no launch
    assert lib.afan_pgd_step(None, p, 0, None, None, 4, 0.1, 0.1, 0, None) == -4      # AFAN_ENULL
    assert lib.afan_pgd_step(p, p, 0, None, None, 4, 0.1, 0.1, 1, None) == -4         # clip needs x_clean
    assert lib.afan_pgd_step(p, p, 7, None, None, 4, 0.1, 0.1, 0, None) == -1         # AFAN_EDTYPE
    assert lib.afan_pgd_step(odd, p, 0, None, None, 4, 0.1, 0.1, 0, None) == -2       # AFAN_EALIGN
    assert lib.afan_mix_feature(p, p, p, 1, 0, 4, 1e-5, 0, None) == -3
    assert lib.afan_mix_feature(p, p, p, 0, 4, 4, 1e-5, 0, None) == 0
    assert lib.afan_bn_train_forward(p, None, p, 3, 0, 1, 1, 1, 1e-5, 0.1, None, None, 0, p, p, None, None, None, None) == -1
    assert lib.afan_bn_train_forward(p, None, p, 0, 9, 1, 1, 1, 1e-5, 0.1, None, None, 0, p, p, None, None, None, None) == -5   # AFAN_ELAYOUT
    assert lib.afan_sgd_step(p, p, p, None, 4, None, 0.9, 0.0, 1.0, 0, None) == -4
    assert lib.afan_norms_workspace_floats(256, 65536) == 2 * 256 * 16
    assert lib.afan_bn_workspace_floats(64) >= 64 * 64 * 4
    assert lib.afan_lerp_points(p, p, p, 4, buf, 9, None) == -3
    # DeepLab layers
    assert lib.afan_upsample_bilinear_fwd(p, p, 0, 0, 1, 0, 4, 4, 8, 8, None) == -3
    assert lib.afan_upsample_bilinear_fwd(p, p, 0, 7, 1, 1, 4, 4, 8, 8, None) == -5
    assert lib.afan_upsample_bilinear_fwd(p, p, 0, 0, 0, 1, 4, 4, 8, 8, None) == 0          # empty batch: no launch
    assert lib.afan_ce2d(p, p, 0, 1, 33, 4, 255, 1.0, p, p, None, None) == -3               # more than 32 classes
    assert lib.afan_ce2d(p, None, 0, 1, 4, 4, 255, 1.0, p, p, None, None) == -4
    assert lib.afan_ce2d_workspace_floats(513 * 513) == 1 + 2 * 1029
    assert lib.afan_maxpool3x3s2_fwd(p, p, 3, 0, 1, 1, 4, 4, None) == -1
    assert lib.afan_pointwise_fwd(p, 1, p, None, p, 4, 12, 4, None) == -3                   # ci % 8
    assert lib.afan_pointwise_fwd(p, 1, p, None, p, 4, 16, 33, None) == -3                  # co > afan_pointwise_max_co()
    assert lib.afan_pointwise_max_co() == 32
    assert lib.afan_dropout(p, p, 0, 4, 1.0, None, None, None, 1, None) == -3               # p must be < 1
    assert lib.afan_dropout(p, p, 0, 4, 0.1, None, None, None, 1, None) == -4               # needs a mask or a generator
    assert lib.afan_conv_stem7_supported(3, 64, 7, 2) == 1 and lib.afan_conv_stem7_supported(3, 64, 3, 1) == 0
    assert lib.afan_conv_stem7_wgrad_workspace_floats(2, 513, 513) == 512 * 147 * 64
    # atrous / ragged-channel convolutions: shapes the tiled kernels take or decline
    assert lib.afan_conv_supported(304, 256, 3, 1) == 1 and lib.afan_conv_supported(256, 48, 1, 1) == 1
    assert lib.afan_conv_supported(20, 256, 3, 1) == 0 and lib.afan_conv_supported(100, 256, 3, 1) == 0
    assert lib.afan_conv_fwd_nhwc_bf16(p, p, p, 1, 8, 8, 256, 256, 3, 2, 2, None, None, None, 1, None) == -3   # dilation needs stride 1
    assert lib.afan_conv_fwd_nhwc_bf16(p, p, p, 1, 8, 8, 256, 256, 1, 1, 2, None, None, None, 1, None) == -3   # ... and a 3x3


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libafan_hip.so")
    with pytest.raises(pkg.AfanLibraryError):
        pkg._lib.load()


def test_product_never_imports_oracle():
    """The shipped package must not reach into oracle/ (the oracle is test infrastructure)."""
    pk = os.path.join(ROOT, "cv_a-fan_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
