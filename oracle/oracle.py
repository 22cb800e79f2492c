"""ctypes/numpy front-end of the CPU oracle (oracle/ahv_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under 3dahv_amd/ imports this module.

Parity status: pinned by tests/golden/*.npz (generated from the reference's own
code by tools/gen_golden.py); see tests/test_oracle.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AHV_ORACLE_LIB: an alternative build of the same source (the sanitizer build, `make -C oracle asan`)
_LIB_PATH = os.path.abspath(os.environ.get("AHV_ORACLE_LIB") or os.path.join(_HERE, "libahv_oracle.so"))
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile libahv_oracle.so with gcc (a few seconds)."""
    src = os.path.join(_HERE, "ahv_oracle.c")
    if os.environ.get("AHV_ORACLE_LIB"):
        if not os.path.exists(_LIB_PATH):
            raise FileNotFoundError("AHV_ORACLE_LIB=%s does not exist (make -C oracle asan)" % _LIB_PATH)
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libahv_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ahv_oracle_num_threads.restype = ctypes.c_int
        _lib.ahv_oracle_score_hypotheses_f32.restype = ctypes.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_f32p)


def num_threads() -> int:
    return int(lib().ahv_oracle_num_threads())


def rotate_volume(volume: np.ndarray, R: np.ndarray) -> np.ndarray:
    """utils.rotate_volume (utils.py:113-131). volume (N|1,C,D,H,W), R (N,3,3) -> (N,C,D,H,W)."""
    volume, R = _f(volume), _f(R)
    N = R.shape[0]
    nb, C, D, H, W = volume.shape
    if nb not in (1, N):
        raise ValueError("volume batch must be 1 (broadcast) or N")
    stride = 0 if (nb == 1 and N != 1) else C * D * H * W
    out = np.empty((N, C, D, H, W), dtype=np.float32)
    lib().ahv_oracle_rotate_volume_f32(_p(volume), ctypes.c_int64(stride), _p(R), ctypes.c_int64(N),
                                       C, D, H, W, _p(out))
    return out


def forward_3d2d(vol: np.ndarray, W1: np.ndarray, W2: np.ndarray, b2: np.ndarray) -> np.ndarray:
    """Feature_Aligner.forward_3d2d (modules/modules.py:112-124). (M,16,8,8,8) -> (M,32,64)."""
    vol, W1, W2, b2 = _f(vol), _f(W1).reshape(32, 384), _f(W2).reshape(32, 32), _f(b2)
    M = vol.shape[0]
    assert vol.shape[1:] == (16, 8, 8, 8)
    out = np.empty((M, 32, 64), dtype=np.float32)
    lib().ahv_oracle_forward_3d2d_f32(_p(vol), _p(W1), _p(W2), _p(b2), ctypes.c_int64(M), _p(out))
    return out


def score_features(f_src: np.ndarray, f_tgt: np.ndarray) -> np.ndarray:
    """(f_src * f_tgt[:, None]).sum(2).mean(-1) (test_co3d.py:143). (B,N,32,64),(B,32,64) -> (B,N)."""
    f_src, f_tgt = _f(f_src), _f(f_tgt)
    B, N = f_src.shape[:2]
    out = np.empty((B, N), dtype=np.float32)
    lib().ahv_oracle_score_features_f32(_p(f_src), _p(f_tgt), B, ctypes.c_int64(N), _p(out))
    return out


def argmax(scores: np.ndarray):
    """torch.max(scores, dim=1) (test_co3d.py:145): (values, first maximal int64 index)."""
    scores = _f(scores)
    B, N = scores.shape
    best = np.empty((B,), dtype=np.float32)
    idx = np.empty((B,), dtype=np.int64)
    lib().ahv_oracle_argmax_f32(_p(scores), B, ctypes.c_int64(N), _p(best), idx.ctypes.data_as(_i64p))
    return best, idx


def score_hypotheses(vol_src, vol_tgt, R, W1, W2, b2):
    """Whole hot loop of test_co3d.py:135-146. R (N,3,3) shared or (B,N,3,3) per sample.

    Returns (scores (B,N), best (B,), best_idx (B,) int64).
    """
    vol_src, vol_tgt, R = _f(vol_src), _f(vol_tgt), _f(R)
    W1, W2, b2 = _f(W1).reshape(32, 384), _f(W2).reshape(32, 32), _f(b2)
    B = vol_src.shape[0]
    if R.ndim == 3:
        N, rstride = R.shape[0], 0
    else:
        assert R.shape[0] == B
        N, rstride = R.shape[1], R.shape[1] * 9
    scores = np.empty((B, N), dtype=np.float32)
    best = np.empty((B,), dtype=np.float32)
    idx = np.empty((B,), dtype=np.int64)
    rc = lib().ahv_oracle_score_hypotheses_f32(_p(vol_src), _p(vol_tgt), _p(R), ctypes.c_int64(rstride),
                                               _p(W1), _p(W2), _p(b2), B, ctypes.c_int64(N), _p(scores),
                                               _p(best), idx.ctypes.data_as(_i64p))
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    return scores, best, idx


def geodesic_deg(R_pred: np.ndarray, R_gt: np.ndarray) -> np.ndarray:
    """Angular error in degrees (test_co3d.py:149-150)."""
    R_pred, R_gt = _f(R_pred).reshape(-1, 9), _f(R_gt).reshape(-1, 9)
    n = R_pred.shape[0]
    out = np.empty((n,), dtype=np.float32)
    lib().ahv_oracle_geodesic_deg_f32(_p(R_pred), _p(R_gt), ctypes.c_int64(n), _p(out))
    return out
