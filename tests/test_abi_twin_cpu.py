"""GPU-free walk of the C ABI's calling convention ("identical signatures with suffix _cpu
operating on host pointers ... used for ABI tests").

The ``*_cpu`` twins live in the ORACLE library (oracle/ahv_oracle.c, test infrastructure) -- the product library has
no CPU path.  What is under test is the product's ctypes signature table, ``3dahv_amd._lib.SIGNATURES``: it is applied
verbatim to the twins and driven with host pointers through the golden vectors, so a wrong argument order, integer
width or stride convention in the binding shows up without a GPU.  Also pinned here: the packed-key convention
(reset / merge flag, n_offset, lowest index among equal scores, NaN above +inf, the empty key)."""
import ctypes

import numpy as np
import pytest
import torch

from .conftest import load_golden

TWINS = ("ahv_score_hypotheses_f32", "ahv_unpack_best", "ahv_rotate_volume_f32", "ahv_forward_3d2d_f32",
         "ahv_score_features_f32", "ahv_argmax_f32", "ahv_select_rotation_f32", "ahv_reset_best", "ahv_verify_pair_f32")


@pytest.fixture(scope="module")
def twin(ahv):
    from oracle import oracle as oracle_mod
    cdll = ctypes.CDLL(oracle_mod.build())
    ns = type("Twin", (), {})()
    for name in TWINS:
        res, args = ahv._lib.SIGNATURES[name]          # the PRODUCT's table, not a copy
        fn = getattr(cdll, name + "_cpu")
        fn.restype, fn.argtypes = res, args
        setattr(ns, name, fn)
    return ns


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a):
    return a.ctypes.data


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-2)))


def test_every_twin_has_the_declared_symbol(twin):
    assert all(hasattr(twin, n) for n in TWINS)


def test_config1_golden_through_the_product_signatures(ahv, twin):
    g = load_golden("score_n128")
    vs, vt, R = f32(g["vol_src"]), f32(g["vol_tgt"]), f32(g["R"])
    W1, W2, b2 = f32(g["W1"]), f32(g["W2"]), f32(g["b2"])
    N = R.shape[0]
    # A1: the stride-0 expand the reference passes
    rot = np.empty((2, 16, 8, 8, 8), np.float32)
    assert twin.ahv_rotate_volume_f32(ptr(vs), 0, ptr(R), 2, 16, 8, 8, 8, ptr(rot), None) == 0
    assert np.max(np.abs(rot - g["rot_first2"])) <= 1e-5 * np.max(np.abs(g["rot_first2"]))
    # A2
    ft = np.empty((1, 32, 64), np.float32)
    assert twin.ahv_forward_3d2d_f32(ptr(vt), ptr(W1), ptr(W2), ptr(b2), 1, ptr(ft), None) == 0
    assert np.max(np.abs(ft - g["f_tgt"])) <= 1e-5
    fs = np.empty((2, 32, 64), np.float32)
    assert twin.ahv_forward_3d2d_f32(ptr(rot), ptr(W1), ptr(W2), ptr(b2), 2, ptr(fs), None) == 0
    assert np.max(np.abs(fs - g["f_src_first2"])) <= 1e-5
    # A3 on materialised features
    s2 = np.empty((1, 2), np.float32)
    assert twin.ahv_score_features_f32(ptr(fs), ptr(ft), 1, 2, ptr(s2), None) == 0
    assert relerr(s2, g["scores"][:, :2]) <= 1e-4
    # fused A1-A4 with the packed key, then unpack / select
    scores = np.empty((1, N), np.float32)
    key = np.full((1,), 0xDEADBEEF, np.int64)
    rc = twin.ahv_score_hypotheses_f32(ptr(vs), ptr(ft), ptr(R), 0, 0, ptr(W1), ptr(W2), ptr(b2), 1, N, ptr(scores),
                                       ptr(key), ahv._lib.AHV_SCORE_RESET_BEST, None)
    assert rc == 0
    assert relerr(scores, g["scores"]) <= 1e-4
    best, idx = np.empty(1, np.float32), np.empty(1, np.int64)
    assert twin.ahv_unpack_best(ptr(key), 1, ptr(best), ptr(idx), None) == 0
    assert int(idx[0]) == int(g["best_idx"][0]) and abs(best[0] - g["best"][0]) <= 1e-6
    R_out = np.empty((1, 3, 3), np.float32)
    # the whole step behind one entry point: same scores, same key, and the target features on request
    scores2, key2, ft2 = np.empty((1, N), np.float32), np.zeros(1, np.int64), np.empty((1, 32, 64), np.float32)
    rc = twin.ahv_verify_pair_f32(ptr(vs), ptr(vt), ptr(R), 0, 0, ptr(W1), ptr(W2), ptr(b2), 1, N, ptr(scores2), ptr(key2),
                                  ptr(ft2), ahv._lib.AHV_SCORE_RESET_BEST, None, None)
    assert rc == 0 and np.array_equal(scores2, scores) and np.array_equal(key2, key) and np.array_equal(ft2, ft)
    kept = key.copy()
    assert twin.ahv_select_rotation_f32(ptr(key), ptr(R), 0, 0, N, 1, ptr(R_out), ptr(best), ptr(idx), 0, None) == 0
    assert np.array_equal(R_out[0], R[int(g["best_idx"][0])]) and np.array_equal(key, kept)
    # AHV_SELECT_RESET_KEY: same outputs, and the key comes back EMPTY for the next step
    R_out2 = np.empty((1, 3, 3), np.float32)
    assert twin.ahv_select_rotation_f32(ptr(key), ptr(R), 0, 0, N, 1, ptr(R_out2), ptr(best), ptr(idx),
                                        ahv._lib.AHV_SELECT_RESET_KEY, None) == 0
    assert np.array_equal(R_out2, R_out) and int(idx[0]) == int(g["best_idx"][0])
    assert int(key[0]) == ahv._lib.AHV_KEY_EMPTY == np.iinfo(np.int64).min
    assert twin.ahv_select_rotation_f32(ptr(key), ptr(R), 0, 0, N, 1, ptr(R_out2), ptr(best), ptr(idx), 2, None) == -1


def test_batched_shared_and_per_sample_strides(ahv, twin):
    g, h = load_golden("batched"), load_golden("score_n128")
    W1, W2, b2 = f32(h["W1"]), f32(h["W2"]), f32(h["b2"])
    vs, vt = f32(g["vol_src"]), f32(g["vol_tgt"])
    B = vs.shape[0]
    ft = np.empty((B, 32, 64), np.float32)
    assert twin.ahv_forward_3d2d_f32(ptr(vt), ptr(W1), ptr(W2), ptr(b2), B, ptr(ft), None) == 0
    for R, want, stride in ((f32(g["R_shared"]), g["scores_shared"], 0), (f32(g["R_per"]), g["scores_per"], 64 * 9)):
        N = R.shape[-3]
        scores, key = np.empty((B, N), np.float32), np.zeros(B, np.int64)
        rc = twin.ahv_score_hypotheses_f32(ptr(vs), ptr(ft), ptr(R), stride, 0, ptr(W1), ptr(W2), ptr(b2), B, N,
                                           ptr(scores), ptr(key), ahv._lib.AHV_SCORE_RESET_BEST, None)
        assert rc == 0 and relerr(scores, want) <= 1e-4
    idx = np.empty(B, np.int64)
    twin.ahv_unpack_best(ptr(key), B, None, ptr(idx), None)
    assert list(idx) == list(np.argmax(scores, axis=1))


def test_key_convention_chunks_offsets_ties_nan_and_empty(ahv, twin):
    RESET = ahv._lib.AHV_SCORE_RESET_BEST
    s = np.array([[0.1, 0.7, 0.7, 0.2, -0.0, 0.0], [np.nan, 1.0, np.nan, np.inf, 0.0, 0.0]], np.float32)
    B, N = s.shape
    key = np.zeros(B, np.int64)
    assert twin.ahv_argmax_f32(ptr(s), B, N, 0, ptr(key), RESET, None) == 0
    best, idx = np.empty(B, np.float32), np.empty(B, np.int64)
    twin.ahv_unpack_best(ptr(key), B, ptr(best), ptr(idx), None)
    tb, ti = torch.max(torch.from_numpy(s), dim=1)           # first maximal index; NaN propagates
    assert list(idx) == list(ti.numpy())
    assert best[0] == np.float32(0.7) and np.isnan(best[1])
    # two chunks with offsets merge into the same key as one pass (flags = 0: merge, reset only on the first)
    key2 = np.full(B, 123, np.int64)
    a, b = np.ascontiguousarray(s[:, :4]), np.ascontiguousarray(s[:, 4:])
    assert twin.ahv_argmax_f32(ptr(a), B, 4, 0, ptr(key2), RESET, None) == 0
    assert twin.ahv_argmax_f32(ptr(b), B, 2, 4, ptr(key2), 0, None) == 0
    assert np.array_equal(key, key2)
    # sharding: the global index is n_offset + local index
    key3 = np.zeros(B, np.int64)
    twin.ahv_argmax_f32(ptr(s), B, N, 1000, ptr(key3), RESET, None)
    twin.ahv_unpack_best(ptr(key3), B, None, ptr(idx), None)
    assert list(idx) == [1001, 1000]
    # -0.0 and +0.0 are equal: the lower index wins
    z = np.array([[-1.0, 0.0, -0.0]], np.float32)
    kz, iz = np.zeros(1, np.int64), np.empty(1, np.int64)
    twin.ahv_argmax_f32(ptr(z), 1, 3, 0, ptr(kz), RESET, None)
    twin.ahv_unpack_best(ptr(kz), 1, None, ptr(iz), None)
    assert int(iz[0]) == 1
    # nothing scored: (-inf, -1); the empty key is INT64_MIN, below every real key (a negative best score must
    # survive a merge into a fresh key: signed order, not "0 = empty")
    k0 = np.full(1, 7, np.int64)
    assert twin.ahv_reset_best(ptr(k0), 1, None) == 0 and int(k0[0]) == ahv._lib.AHV_KEY_EMPTY
    twin.ahv_unpack_best(ptr(k0), 1, ptr(best[:1]), ptr(idx[:1]), None)
    assert best[0] == -np.inf and idx[0] == -1
    neg = np.array([[-0.5, -0.25, -0.75]], np.float32)
    assert twin.ahv_argmax_f32(ptr(neg), 1, 3, 0, ptr(k0), 0, None) == 0   # merge into the empty key
    twin.ahv_unpack_best(ptr(k0), 1, ptr(best[:1]), ptr(idx[:1]), None)
    assert best[0] == np.float32(-0.25) and idx[0] == 1
    # the keys are in SIGNED order: np.max over int64 keys of separate chunks == the one-pass key
    parts = np.zeros((3, B), np.int64)
    for c in range(3):
        chunk = np.ascontiguousarray(s[:, 2 * c:2 * c + 2])
        assert twin.ahv_argmax_f32(ptr(chunk), B, 2, 2 * c, ptr(parts[c]), RESET, None) == 0
    assert np.array_equal(parts.max(axis=0), key)
    # n_offset + N must fit in 32 bits; negative sizes are refused
    assert twin.ahv_argmax_f32(ptr(s), B, N, (1 << 32) - 2, ptr(key), RESET, None) == -1
    assert twin.ahv_rotate_volume_f32(ptr(s), 0, ptr(s), -5, 16, 8, 8, 8, ptr(s), None) == -1
