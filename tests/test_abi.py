"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/ahv.h
declares.  No compute calls here (no GPU in the CPU test tier)."""
import os
import re
import subprocess

import pytest

from .conftest import REPO


@pytest.fixture(scope="module")
def lib(ahv):
    ahv._lib.build()
    return ahv._lib.load()


def declared_symbols(header="ahv.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ahv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib, ahv):
    syms = declared_symbols()                 # the drop-in boundary
    diag = declared_symbols("ahv_diag.h")     # measurement / developer entry points: a header of their own
    assert len(syms) >= 9 and not set(syms) & set(diag)
    assert all("clocked" not in s and "diag" not in s for s in syms)
    for s in syms + diag:
        assert hasattr(lib, s), "libahv_hip.so lacks %s declared in include/" % s
    assert sorted(ahv._lib.SIGNATURES) == syms  # the ctypes tables mirror the headers one to one
    assert sorted(ahv._lib.DIAG_SIGNATURES) == diag
    out = subprocess.check_output(["nm", "-D", "--defined-only", ahv._lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (ahv_[a-z0-9_]+)", out)))
    assert exported == sorted(syms + diag)  # nothing undeclared leaks out of the ABI


def test_abi_version_and_error_string(lib):
    assert lib.ahv_abi_version() == (2 << 16) | 3  # 2.3: the training pair (train forward + saved-u backward)
    assert isinstance(lib.ahv_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib, ahv):
    # validation happens before any HIP call, so it is testable on a CPU-only box
    rc = lib.ahv_score_hypotheses_f32(None, None, None, 0, 0, None, None, None, 1, 1, None, None, 0, None)
    assert rc == -1 and b"null" in lib.ahv_last_error()
    rc = lib.ahv_rotate_volume_f32(1, 0, 1, -5, 16, 8, 8, 8, 1, None)
    assert rc == -1 and b"bad shape" in lib.ahv_last_error()
    rc = lib.ahv_score_hypotheses_f32(1, 1, 1, 5, 0, 1, 1, 1, 1, 10, None, None, 0, None)
    assert rc == -1 and b"r_batch_stride" in lib.ahv_last_error()
    rc = lib.ahv_score_hypotheses_f32(1, 1, 1, 0, 1 << 32, 1, 1, 1, 1, 10, None, None, 0, None)
    assert rc == -1 and b"32 bits" in lib.ahv_last_error()
    rc = lib.ahv_score_hypotheses_f32(1, 1, 1, 0, 0, 1, 1, 1, 1, 10, None, None, 8, None)
    assert rc == -1 and b"flags" in lib.ahv_last_error()
    with pytest.raises(ahv._lib.AhvError):
        ahv._lib.check(rc, "x")
    # backward entry point: workspace size is a pure function; pointer / stride / workspace checks come first
    # dL/du + one max|du| word per sample (padded to 16 B) + one partial dW1 per workgroup (one per CU; 1024 with no device)
    cu = lib.ahv_device_cu_count()
    assert lib.ahv_score_hypotheses_backward_workspace_bytes(2, 10) == 4 * (2048 * 20 + 4 + (cu if cu > 0 else 1024) * 32 * 384)
    assert lib.ahv_score_hypotheses_backward_workspace_bytes(0, 10) == 0
    bw = lib.ahv_score_hypotheses_backward_f32
    assert bw(1, 1, 1, 0, 1, 1, 1, 1, 10, 1, 16, 1 << 30, 1, 1, None, 1, 1, None) == -1 and b"weight-gradient" in lib.ahv_last_error()
    assert bw(None, 1, 1, 0, 1, 1, 1, 1, 10, 1, 16, 1 << 30, 1, 1, 1, 1, 1, None) == -1 and b"null" in lib.ahv_last_error()
    assert bw(1, 1, 1, 7, 1, 1, 1, 1, 10, 1, 16, 1 << 30, 1, 1, 1, 1, 1, None) == -1 and b"r_batch_stride" in lib.ahv_last_error()
    assert bw(1, 1, 1, 0, 1, 1, 1, 1, 10, 1, 16, 100, 1, 1, 1, 1, 1, None) == -1 and b"workspace of 100" in lib.ahv_last_error()
    assert bw(1, 1, 1, 0, 1, 1, 1, 1, 10, 1, 20, 1 << 30, 1, 1, 1, 1, 1, None) == -1 and b"aligned" in lib.ahv_last_error()
    assert lib.ahv_so3_grid_f32(10, 8, 5, 1, None) == -1 and b"so3_grid" in lib.ahv_last_error()
    assert lib.ahv_so3_grid_f32(10, 0, 0, None, None) == 0
    # the one-launch coarse-to-fine step: argument order (vol_src, vol_tgt, R, r_stride, N, D, N2, W1, W2, b2, B, scores x 2,
    # keys, sync, feat_tgt_out, R_pred, fine score / idx, coarse score / idx, flags, stream)
    c2f = lib.ahv_coarse_to_fine_f32
    ok = [1, 1, 1, 0, 10, 1, 4, 1, 1, 1, 2, None, None, 1, 1, None, 1, 1, 1, 1, 1, 0, None]
    def call(**kw):
        a = list(ok)
        for k, v in kw.items():
            a[int(k[1:])] = v
        return c2f(*a)
    assert call(_13=None) == -1 and b"keys and sync" in lib.ahv_last_error()
    assert call(_5=None) == -1 and b"null input" in lib.ahv_last_error()
    assert call(_6=0) == -1 and b"empty hypothesis set" in lib.ahv_last_error()
    assert call(_3=5) == -1 and b"r_batch_stride" in lib.ahv_last_error()
    assert call(_21=ahv._lib.AHV_SCORE_SPLIT_F16) == -1 and b"flags" in lib.ahv_last_error()
    assert call(_10=0) == 0   # B = 0: nothing to do


def test_ops_refuse_cpu_tensors(ahv):
    import torch
    v = torch.zeros(2, 16, 8, 8, 8)
    R = torch.eye(3)[None].repeat(2, 1, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ahv.ops.rotate_volume(v, R)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ahv.ops.forward_3d2d(v, torch.zeros(32, 384), torch.zeros(32, 32), torch.zeros(32))


def test_product_does_not_import_oracle():
    """The shipped path must never route through the CPU oracle."""
    pkg = os.path.join(REPO, "3dahv_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "libahv_oracle" not in text, f
