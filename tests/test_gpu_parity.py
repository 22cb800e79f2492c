"""Parity of the HIP path (through the C ABI) against the oracle and the committed golden vectors.

Bar (BASELINE.json north_star): per-hypothesis scores within 1e-4 relative of the reference PyTorch
path on identical inputs, arg-max index bit-exact.  Scores are mean cosine similarities in [-1, 1]
that can pass through zero, so "relative" is taken against max(|ref|, SCORE_FLOOR); tensors
(rotated volumes, features) are compared with max|diff| / max|ref|.
"""
import hashlib

import numpy as np
import pytest
import torch

from .conftest import load_golden

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-4   # north-star tolerance
SCORE_FLOOR = 1e-2  # |score| floor for the relative measure
TENSOR_RTOL = 1e-5  # rotated volumes / features: fp32 with a different summation order


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(ahv):
    ahv._lib.load()  # raises if libahv_hip.so is missing: no fallback
    return ahv.ops


@pytest.fixture(scope="module")
def G(dev, g128):
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g128[k])).to(dev)
    return {k: t(k) for k in ["vol_src", "vol_tgt", "R", "W1", "W2", "b2"]}


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def score_relerr(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), SCORE_FLOOR)))


def tensor_relerr(got, ref):
    return float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-30))


# ---------------------------------------------------------------- op-level kernels

def test_rotate_volume_golden_and_oracle(ops, oracle, G, g128):
    vol = G["vol_src"][0][None].expand(128, -1, -1, -1, -1)  # the reference's stride-0 expand
    assert vol.stride(0) == 0
    out = ops.rotate_volume(vol, G["R"])
    assert out.shape == (128, 16, 8, 8, 8) and out.is_contiguous()
    got = out.cpu().numpy()
    assert tensor_relerr(got[:2], g128["rot_first2"]) < TENSOR_RTOL
    assert tensor_relerr(got, oracle.rotate_volume(g128["vol_src"], g128["R"])) < TENSOR_RTOL


def test_rotate_volume_edge_cases(ops, G, g128, dev):
    g = load_golden("edge_rotations")
    R = to_dev(g["R"], dev)
    n = R.shape[0]
    got = ops.rotate_volume(G["vol_src"][0][None].expand(n, -1, -1, -1, -1), R).cpu().numpy()
    names = list(g["names"])
    assert np.array_equal(got[names.index("identity")], g128["vol_src"][0])  # bit-exact copy
    for i in range(1, 25):  # cube rotations permute voxels exactly
        assert np.array_equal(np.sort(got[i].ravel()), np.sort(g128["vol_src"][0].ravel()))
    assert tensor_relerr(got[g["rot_keep_idx"]], g["rot_keep"]) < TENSOR_RTOL
    zero_frac = (got == 0).reshape(n, -1).mean(axis=1)
    assert zero_frac[names.index("double")] == pytest.approx(0.875)  # zeros padding
    assert zero_frac[names.index("half")] == 0.0
    assert np.allclose(zero_frac, g["rot_zero_frac"], atol=2e-3)


def test_rotate_volume_materialised_batch_and_n1(ops, oracle, G, g128):
    # same volume materialised N times (non-zero batch stride): generic path
    vol = G["vol_src"][0][None].repeat(5, 1, 1, 1, 1)
    got = ops.rotate_volume(vol, G["R"][:5]).cpu().numpy()
    assert tensor_relerr(got, oracle.rotate_volume(g128["vol_src"], g128["R"][:5])) < TENSOR_RTOL
    got1 = ops.rotate_volume(G["vol_src"], G["R"][:1]).cpu().numpy()  # N == 1
    assert tensor_relerr(got1, got[:1]) < 1e-6


def test_rotate_volume_generic_shape(ops, oracle, dev):
    rng = np.random.RandomState(0)
    vol = rng.standard_normal((7, 3, 4, 5, 6)).astype(np.float32)
    R = rng.standard_normal((7, 3, 3)).astype(np.float32) * 0.7
    got = ops.rotate_volume(to_dev(vol, dev), to_dev(R, dev)).cpu().numpy()
    ref = np.concatenate([oracle.rotate_volume(vol[i:i + 1], R[i:i + 1]) for i in range(7)])
    assert tensor_relerr(got, ref) < TENSOR_RTOL


def test_rotate_volume_errors(ops, G):
    with pytest.raises(NotImplementedError):
        ops.rotate_volume(G["vol_src"], G["R"][:1], padding_mode="border")
    with pytest.raises(RuntimeError):
        ops.rotate_volume(G["vol_src"], G["R"][:3])  # batch mismatch
    with pytest.raises(RuntimeError):
        ops.rotate_volume(G["vol_src"].cpu(), G["R"][:1].cpu())  # no CPU fallback
    out = ops.rotate_volume(G["vol_src"][:0], G["R"][:0])  # empty batch
    assert out.shape == (0, 16, 8, 8, 8)


def test_forward_3d2d_golden_and_oracle(ops, oracle, G, g128, dev):
    f_tgt = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"]).cpu().numpy()
    assert tensor_relerr(f_tgt, g128["f_tgt"]) < TENSOR_RTOL
    rot = oracle.rotate_volume(g128["vol_src"], g128["R"])
    got = ops.forward_3d2d(to_dev(rot, dev), G["W1"], G["W2"], G["b2"]).cpu().numpy()
    assert tensor_relerr(got[:2], g128["f_src_first2"]) < TENSOR_RTOL
    assert tensor_relerr(got, oracle.forward_3d2d(rot, g128["W1"], g128["W2"], g128["b2"])) < TENSOR_RTOL
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    # conv-shaped weights, as stored in the reference state dict
    got2 = ops.forward_3d2d(G["vol_tgt"], G["W1"].reshape(32, 384, 1, 1), G["W2"].reshape(32, 32, 1, 1), G["b2"])
    assert np.array_equal(got2.cpu().numpy(), f_tgt)


def test_forward_3d2d_zero_volume_eps(ops, oracle, G, g128, dev):
    # zero volume and zero bias -> v = 0 -> normalise clamps the norm at 1e-12 -> exact zeros
    z = torch.zeros(3, 16, 8, 8, 8, device=dev)
    out = ops.forward_3d2d(z, G["W1"], G["W2"], torch.zeros_like(G["b2"]))
    assert torch.count_nonzero(out) == 0
    out_b = ops.forward_3d2d(z, G["W1"], G["W2"], G["b2"]).cpu().numpy()
    ref = oracle.forward_3d2d(np.zeros((1, 16, 8, 8, 8), np.float32), g128["W1"], g128["W2"], g128["b2"])
    assert tensor_relerr(out_b[:1], ref) < TENSOR_RTOL


def test_score_features_and_argmax(ops, oracle, dev):
    rng = np.random.RandomState(1)
    fs = rng.standard_normal((2, 300, 32, 64)).astype(np.float32)
    ft = rng.standard_normal((2, 32, 64)).astype(np.float32)
    got = ops.score_features(to_dev(fs, dev), to_dev(ft, dev))
    ref = oracle.score_features(fs, ft)
    assert tensor_relerr(got.cpu().numpy(), ref) < TENSOR_RTOL
    val, idx = ops.argmax(got)
    rv, ri = torch.max(got.cpu(), dim=1)
    assert torch.equal(idx.cpu(), ri) and torch.equal(val.cpu(), rv)


def test_argmax_ties_nan_and_signed_zero(ops, dev):
    s = torch.tensor([[0.1, 0.7, 0.7, 0.2], [float("nan"), 1.0, float("nan"), 0.0], [-0.0, 0.0, -1.0, -2.0],
                      [-3.0, -1.5, -1.5, -9.0]], device=dev)
    val, idx = ops.argmax(s)
    rv, ri = torch.max(s.cpu(), dim=1)
    assert idx.cpu().tolist() == ri.tolist() == [1, 0, 0, 1]
    assert torch.isnan(val[1]) and val[0].item() == rv[0].item() and val[3].item() == -1.5
    big = torch.zeros(1, 100000, device=dev)
    big[0, 77777] = 1e-30
    big[0, 99999] = 1e-30
    assert ops.argmax(big)[1].item() == 77777


# ---------------------------------------------------------------- fused scorer

def fused(ops, G, R, **kw):
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    scores, key = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], **kw)
    val, idx = ops.unpack_best(key)
    return scores, val, idx


def test_fused_n128_config1(ops, G, g128):
    scores, val, idx = fused(ops, G, G["R"])
    assert score_relerr(scores.cpu().numpy(), g128["scores"]) < SCORE_RTOL
    assert idx.item() == int(g128["best_idx"][0])
    assert abs(val.item() - float(g128["best"][0])) <= SCORE_RTOL * abs(float(g128["best"][0]))
    assert val.item() == scores[0, idx.item()].item()  # the key carries the exact fp32 score


def test_head_weights_off_a_16_byte_boundary(ops, G, g128, dev):
    """W1 handed over as a contiguous view that starts 4 bytes into its storage (the staging code reads 16-byte
    rows when it can and single floats when it cannot): same scores, fused and op-level."""
    store = torch.empty(32 * 384 + 1, dtype=torch.float32, device=dev)
    W1u = store[1:].view(32, 384)
    W1u.copy_(G["W1"])
    assert W1u.data_ptr() % 16 == 4 and W1u.is_contiguous()
    ft = ops.forward_3d2d(G["vol_tgt"], W1u, G["W2"], G["b2"])
    assert torch.equal(ft, ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"]))
    s_u, k_u = ops.score_hypotheses(G["vol_src"], ft, G["R"], W1u, G["W2"], G["b2"])
    s_a, k_a = ops.score_hypotheses(G["vol_src"], ft, G["R"], G["W1"], G["W2"], G["b2"])
    assert torch.equal(s_u, s_a) and torch.equal(k_u, k_a)
    assert score_relerr(s_u.cpu().numpy(), g128["scores"]) < SCORE_RTOL


def test_fused_n4096(ops, G, dev):
    g = load_golden("score_n4096")
    scores, val, idx = fused(ops, G, to_dev(g["R"], dev))
    assert score_relerr(scores.cpu().numpy(), g["scores"]) < SCORE_RTOL
    assert idx.item() == int(g["best_idx"][0])


def test_fused_n50k_config2_digest(ops, ahv, oracle, G, g128, dev):
    g = load_golden("score_n50k_digest")
    R = ahv.rotations.haar_rotations_np(int(g["n"]), int(g["seed"]))
    assert hashlib.sha256(R.tobytes()).hexdigest() == str(g["R_sha256"])
    scores, val, idx = fused(ops, G, to_dev(R, dev))
    s = scores[0].cpu().numpy()
    assert idx.item() == int(g["best_idx"][0])  # bit-exact arg-max at the reference's test size
    assert score_relerr(s[::97], g["every97_score"]) < SCORE_RTOL
    assert score_relerr(s[g["top16_idx"]], g["top16_score"]) < SCORE_RTOL
    assert list(np.argsort(-s, kind="stable")[:16]) == list(g["top16_idx"])
    # without the score tensor (arg-max only) the same key comes back
    _, val2, idx2 = fused(ops, G, to_dev(R, dev), want_scores=False)
    assert idx2.item() == idx.item() and val2.item() == val.item()
    # fused == op-level pipeline == oracle on a slice
    sl = slice(20000, 20512)
    ref, _, _ = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], R[sl], g128["W1"], g128["W2"], g128["b2"])
    assert score_relerr(s[sl], ref[0]) < SCORE_RTOL


def test_fused_equals_op_level_pipeline(ops, G, ahv, dev):
    R = to_dev(ahv.rotations.haar_rotations_np(3000, 21), dev)
    scores, val, idx = fused(ops, G, R)
    rot = ops.rotate_volume(G["vol_src"][0][None].expand(3000, -1, -1, -1, -1), R)
    fs = ops.forward_3d2d(rot, G["W1"], G["W2"], G["b2"]).reshape(1, 3000, 32, 64)
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    s2 = ops.score_features(fs, ft)
    assert score_relerr(scores.cpu().numpy(), s2.cpu().numpy()) < 1e-5
    assert ops.argmax(s2)[1].item() == idx.item()


def test_fused_edge_rotations(ops, G, dev):
    g = load_golden("edge_rotations")
    scores, val, idx = fused(ops, G, to_dev(g["R"], dev))
    assert np.max(np.abs(scores.cpu().numpy() - g["scores"])) < 2e-6
    assert idx.item() == int(g["best_idx"][0])


def test_fused_batched_shared_and_per_sample(ops, G, dev):
    g = load_golden("batched")
    vs, vt = to_dev(g["vol_src"], dev), to_dev(g["vol_tgt"], dev)
    ft = ops.forward_3d2d(vt, G["W1"], G["W2"], G["b2"])
    s, key = ops.score_hypotheses(vs, ft, to_dev(g["R_shared"], dev), G["W1"], G["W2"], G["b2"])
    assert score_relerr(s.cpu().numpy(), g["scores_shared"]) < SCORE_RTOL
    assert ops.unpack_best(key)[1].cpu().tolist() == list(g["best_idx_shared"])
    s2, _ = ops.score_hypotheses(vs, ft, to_dev(g["R_per"], dev), G["W1"], G["W2"], G["b2"])
    assert score_relerr(s2.cpu().numpy(), g["scores_per"]) < SCORE_RTOL


def test_fused_ragged_sizes_and_empty(ops, oracle, G, g128, ahv, dev):
    Rall = ahv.rotations.haar_rotations_np(1031, 5)
    ref, _, ref_idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], Rall, g128["W1"], g128["W2"],
                                              g128["b2"])
    for n in (1, 3, 4, 5, 63, 1023, 1024, 1025, 1031):
        scores, val, idx = fused(ops, G, to_dev(Rall[:n], dev))
        assert score_relerr(scores.cpu().numpy(), ref[:, :n]) < SCORE_RTOL
        assert idx.item() == int(np.argmax(ref[0, :n]))
    scores, val, idx = fused(ops, G, to_dev(Rall[:0], dev))  # N = 0: nothing scored
    assert scores.shape == (1, 0) and idx.item() == -1 and val.item() == float("-inf")


def test_fused_tie_break_lowest_index(ops, G, dev, ahv):
    # every hypothesis identical -> identical scores -> torch.max returns index 0
    R = to_dev(ahv.rotations.haar_rotations_np(1, 9), dev).expand(5000, -1, -1).contiguous()
    scores, val, idx = fused(ops, G, R)
    assert torch.unique(scores).numel() == 1
    assert idx.item() == 0
    # best hypothesis duplicated later in the list: the first copy wins
    R2 = to_dev(ahv.rotations.haar_rotations_np(4000, 10), dev)
    _, _, i0 = fused(ops, G, R2)
    R2[3999] = R2[i0.item()]
    _, _, i1 = fused(ops, G, R2)
    assert i1.item() == i0.item()


def test_fused_sharded_offsets_merge(ops, G, dev, ahv):
    """N split into shards with n_offset (what each rank does); max over packed keys = global arg-max, scores and keys
    bit for bit -- with the DEFAULT launches: 10 000 unsharded is 4 full rounds + 1 808 by single waves, a 2 500 shard is
    one round + 452 by teams, and a team's score is a lone wave's (ahv_team.h)."""
    R = to_dev(ahv.rotations.haar_rotations_np(10000, 11), dev)
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    assert ops.score_plan(1, 10000)[2] == 10000 and ops.score_plan(1, 2500)[2] == 2048
    s_full, k_full = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"])
    keys, parts = [], []
    for r in range(4):
        lo, hi = r * 2500, (r + 1) * 2500
        s, k = ops.score_hypotheses(G["vol_src"], ft, R[lo:hi], G["W1"], G["W2"], G["b2"], n_offset=lo)
        keys.append(k)
        parts.append(s)
    assert torch.equal(torch.cat(parts, dim=1), s_full)  # per-hypothesis results do not depend on the shard
    merged = ahv.dist.merge_keys(torch.stack(keys))  # int64 max over ranks: what the all-reduce computes
    assert torch.equal(merged, k_full)
    assert ops.unpack_best(merged)[1].item() == ops.unpack_best(k_full)[1].item()
    # chunked accumulation into one key tensor (flags = 0: merge into the caller's keys)
    key = None
    for r in range(4):
        lo, hi = r * 2500, (r + 1) * 2500
        _, key = ops.score_hypotheses(G["vol_src"], ft, R[lo:hi], G["W1"], G["W2"], G["b2"], n_offset=lo,
                                      want_scores=False, best_key=key)
    assert torch.equal(key, k_full)
    s_1, k_1 = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], no_teams=True)
    assert torch.equal(s_1, s_full) and torch.equal(k_1, k_full)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 33, 64, 257, 511, 512, 2048 + 106, 2048 + 512, 3 * 2048 + 106, 6 * 2048 + 212,
                               12 * 2048 + 424])
def test_team_scores_are_bit_identical(ops, G, dev, ahv, n):
    """Whether a hypothesis is scored by one wave or by a team of four (ahv_team.h: the remainder of a launch) is a
    scheduling decision: a team runs the lone wave's accumulation chains tile by tile and associates the score's sums the
    same way, so scores and keys are equal BIT FOR BIT -- through both entry points, with shared and per-sample rotations,
    with an offset.  (Rounds 1-4: teams exchanged partial sums, 1e-7 apart, and a sharded run could pick another arg-max
    than the unsharded one inside a rounding-level near-tie.)"""
    R = to_dev(ahv.rotations.haar_rotations_np(n, 100 + n), dev)
    W = (G["W1"], G["W2"], G["b2"])
    ft = ops.forward_3d2d(G["vol_tgt"], *W)
    s1, k1 = ops.score_hypotheses(G["vol_src"], ft, R, *W, no_teams=True, n_offset=7)
    s2, k2 = ops.score_hypotheses(G["vol_src"], ft, R, *W, n_offset=7)
    assert torch.equal(s1, s2) and torch.equal(k1, k2)
    v1, kv1 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, *W, no_teams=True)
    v2, kv2 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, *W)
    assert torch.equal(v1, v2) and torch.equal(kv1, kv2)
    if n <= 2048 + 512:   # a batch of three with per-sample rotation sets (A5: r_batch_stride = 9 N)
        g = torch.Generator(device="cpu").manual_seed(n)
        vs = (torch.randn(3, 16, 8, 8, 8, generator=g) * 1.15).to(dev)
        vt = (torch.randn(3, 16, 8, 8, 8, generator=g) * 1.15).to(dev)
        Rb = to_dev(ahv.rotations.haar_rotations_np(3 * n, 200 + n).reshape(3, n, 3, 3), dev)
        b1, kb1 = ops.verify_pair(vs, vt, Rb, *W, no_teams=True)
        b2, kb2 = ops.verify_pair(vs, vt, Rb, *W)
        assert torch.equal(b1, b2) and torch.equal(kb1, kb2)


def test_team_scores_bit_identical_on_edge_rotations(ops, G, dev):
    """The same on the 33 edge rotations of G3 (identity, cube rotations, 45 degrees, non-orthonormal, scaled) and on
    rotations with NaN / inf entries: a launch of 33 + 6 hypotheses is all teams by default."""
    g = load_golden("edge_rotations")
    Rn = np.concatenate([g["R"], np.repeat(np.eye(3, dtype=np.float32)[None], 6, 0)])
    Rn[33, 1, 1] = np.nan
    Rn[34, 0, 2] = np.inf
    Rn[35, 2, 2] = -np.inf
    Rn[36] = 1e30
    Rn[37] = 0.0
    Rn[38, 2, 0] = np.inf
    R = to_dev(Rn, dev)
    W = (G["W1"], G["W2"], G["b2"])
    s1, k1 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, *W, no_teams=True)
    s2, k2 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, *W)
    assert torch.equal(s1, s2) and torch.equal(k1, k2)


def test_configs3_full_size(ops, dev, ahv, oracle):
    """BASELINE.json configs[3] at its full size on one GPU: B = 32 pairs x N = 50 000 shared hypotheses, the
    hypothesis axis cut into 8 contiguous shards with n_offset (what the 8 ranks do).  merge_keys over the shard
    keys == the unsharded key bit for bit; shard scores == the unsharded scores bit for bit (default launches: the
    shards' remainders go to teams, the unsharded launch has none); a 512-hypothesis slice of shard 3 vs the oracle."""
    B, N, G8 = 32, 50000, 8
    g = load_golden("score_n128")
    rng = np.random.default_rng(33)
    vs = (rng.standard_normal((B, 16, 8, 8, 8)) * 1.15).astype(np.float32)
    vt = (rng.standard_normal((B, 16, 8, 8, 8)) * 1.15).astype(np.float32)
    Rn = ahv.rotations.haar_rotations_np(N, 34)
    W = [to_dev(g[k], dev) for k in ("W1", "W2", "b2")]
    vsd, vtd, R = to_dev(vs, dev), to_dev(vt, dev), to_dev(Rn, dev)
    ft = ops.forward_3d2d(vtd, *W)
    s_full, k_full = ops.score_hypotheses(vsd, ft, R, *W)
    assert s_full.shape == (B, N)
    keys = []
    for r in range(G8):
        lo, hi = ahv.dist.shard_range(N, r, G8)
        assert hi - lo == N // G8
        s, k = ops.score_hypotheses(vsd, ft, R[lo:hi], *W, n_offset=lo)
        assert torch.equal(s, s_full[:, lo:hi])            # per-hypothesis results do not depend on the shard
        keys.append(k)
        if r == 3:
            ref, _, _ = oracle.score_hypotheses(vs, vt, Rn[lo:lo + 512], g["W1"], g["W2"], g["b2"])
            assert score_relerr(s[:, :512].cpu().numpy(), ref) < SCORE_RTOL
    merged = ahv.dist.merge_keys(torch.stack(keys))
    assert torch.equal(merged, k_full)                       # bit for bit
    val, idx = ops.unpack_best(merged)
    rv, ri = torch.max(s_full, dim=1)
    assert torch.equal(idx, ri) and torch.equal(val, rv)     # == torch.max over the full score matrix, all 32 pairs
    # the arg-max-only launch (what a rank runs in production) merges to the same keys
    key = None
    for r in range(G8):
        lo, hi = ahv.dist.shard_range(N, r, G8)
        _, key = ops.score_hypotheses(vsd, ft, R[lo:hi], *W, n_offset=lo, want_scores=False, best_key=key)
    assert torch.equal(key, k_full)
    # what bench.py's multi-rank line runs: the one-launch step per shard -- merged keys == the unsharded one-launch key
    keys_t = [ops.verify_pair(vsd, vtd, R[lo:hi], *W, n_offset=lo, want_scores=False)[1]
              for lo, hi in (ahv.dist.shard_range(N, r, G8) for r in range(G8))]
    k_one = ops.verify_pair(vsd, vtd, R, *W, want_scores=False)[1]
    assert torch.equal(ahv.dist.merge_keys(torch.stack(keys_t)), k_one)
    assert torch.equal(ops.unpack_best(k_one)[1], ri)


def test_fused_properties_full_size(ops, G, dev, ahv):
    """Size-independent properties at BASELINE.json's full N (50 000) and the LINEMOD-grid size (200 000)."""
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    for n, gen in ((50000, lambda: ahv.rotations.haar_rotations_np(50000, 31)),
                   (200000, lambda: ahv.rotations.so3_grid_np(200000))):
        R = to_dev(gen(), dev)
        s, k = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"])
        val, idx = ops.unpack_best(k)
        rv, ri = torch.max(s, dim=1)
        assert idx.item() == ri.item() and val.item() == rv.item()        # key == torch.max of the scores
        assert torch.all(s.abs() <= 1.0 + 1e-5)                           # mean cosine similarity
        perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
        s_1, _ = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], no_teams=True)
        s_p, _ = ops.score_hypotheses(G["vol_src"], ft, R[perm].contiguous(), G["W1"], G["W2"], G["b2"])
        assert torch.equal(s_p[0], s[0, perm])                            # order independence, bit for bit
        assert torch.equal(s, s_1)                                        # teams or single waves: the same bits
        s_again, k_again = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"])
        assert torch.equal(s_again, s) and torch.equal(k_again, k)        # deterministic
    # identity hypothesis == un-rotated source features scored against the target
    eye = torch.eye(3, device=dev)[None]
    s_id, _ = ops.score_hypotheses(G["vol_src"], ft, eye, G["W1"], G["W2"], G["b2"])
    f_src = ops.forward_3d2d(G["vol_src"], G["W1"], G["W2"], G["b2"])
    assert abs(s_id.item() - ops.score_features(f_src[None], ft).item()) < 1e-6
    # scoring the target volume against its own features with R = I gives exactly-normalised 1.0
    s_self, _ = ops.score_hypotheses(G["vol_tgt"], ft, eye, G["W1"], G["W2"], G["b2"])
    assert abs(s_self.item() - 1.0) < 1e-5


def test_fused_argument_errors(ops, ahv, G):
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    with pytest.raises(RuntimeError):
        ops.score_hypotheses(G["vol_src"], ft, G["R"].reshape(-1, 9), G["W1"], G["W2"], G["b2"])
    with pytest.raises(RuntimeError):
        ops.score_hypotheses(G["vol_src"].cpu(), ft.cpu(), G["R"].cpu(), G["W1"].cpu(), G["W2"].cpu(), G["b2"].cpu())
    lib = ahv._lib.load()
    rc = lib.ahv_score_hypotheses_f32(None, None, None, 0, 0, None, None, None, 1, 1, None, None, 0, None)
    assert rc == -1 and b"null" in lib.ahv_last_error()
    with pytest.raises(ahv._lib.AhvError):
        ahv._lib.check(rc, "ahv_score_hypotheses_f32")


@pytest.mark.parametrize("split", [False, True])
def test_fused_kernels_agree(ops, ahv, G, g128, split):
    """Both fused kernels (all-fp32 default; opt-in split-f16 via the per-call flag) meet the same parity bar."""
    with ops.split_f16_scorer(split):
        scores, val, idx = fused(ops, G, G["R"])
        assert score_relerr(scores.cpu().numpy(), g128["scores"]) < SCORE_RTOL
        assert idx.item() == int(g128["best_idx"][0])
        g = load_golden("batched")  # B > 1 and per-sample R through both kernels
        vs, vt = to_dev(g["vol_src"], G["R"].device), to_dev(g["vol_tgt"], G["R"].device)
        ft = ops.forward_3d2d(vt, G["W1"], G["W2"], G["b2"])
        s2, _ = ops.score_hypotheses(vs, ft, to_dev(g["R_per"], G["R"].device), G["W1"], G["W2"], G["b2"])
        assert score_relerr(s2.cpu().numpy(), g["scores_per"]) < SCORE_RTOL


def test_device_haar_sampler(ops, dev):
    R = ops.random_rotations(200000, seed=7, device=dev)
    Rd = R.double()
    eye = torch.eye(3, dtype=torch.float64, device=dev)
    assert (Rd @ Rd.transpose(1, 2) - eye).abs().max().item() < 5e-6       # orthonormal
    assert (torch.linalg.det(Rd) - 1).abs().max().item() < 5e-6            # proper rotations
    tr = Rd.diagonal(dim1=1, dim2=2).sum(-1)
    ang = torch.arccos(((tr - 1) / 2).clamp(-1, 1))
    assert abs(tr.mean().item()) < 0.01                                    # Haar: E[trace] = 0
    assert abs(ang.mean().item() - (np.pi / 2 + 2 / np.pi)) < 0.01        # Haar: E[angle] = pi/2 + 2/pi
    for col in range(3):                                                   # columns uniform on the sphere
        assert Rd[:, :, col].mean(0).abs().max().item() < 0.01
    # counter-based: shards reproduce slices of the whole set; other seeds differ
    assert torch.equal(ops.random_rotations(1000, seed=7, offset=50000, device=dev), R[50000:51000])
    assert not torch.equal(ops.random_rotations(1000, seed=8, device=dev), R[:1000])
    assert ops.random_rotations(0, device=dev).shape == (0, 3, 3)


def test_forward_3d2d_throughput_path_matches_small_paths(ops, oracle, G, g128, ahv, dev):
    """M >= 4096 takes the two-waves-per-SIMD kernel; same results as the oracle and as the other paths."""
    R = to_dev(ahv.rotations.haar_rotations_np(5000, 77), dev)
    rot = ops.rotate_volume(G["vol_src"][0][None].expand(5000, -1, -1, -1, -1), R)
    big = ops.forward_3d2d(rot, G["W1"], G["W2"], G["b2"])            # dual kernel
    mid = ops.forward_3d2d(rot[:1000], G["W1"], G["W2"], G["b2"])     # register-resident W1 kernel
    small = ops.forward_3d2d(rot[:8], G["W1"], G["W2"], G["b2"])      # latency kernel
    assert tensor_relerr(big[:1000].cpu().numpy(), mid.cpu().numpy()) < 1e-6
    assert tensor_relerr(big[:8].cpu().numpy(), small.cpu().numpy()) < 1e-6
    ref = oracle.forward_3d2d(rot[4990:].cpu().numpy(), g128["W1"], g128["W2"], g128["b2"])
    assert tensor_relerr(big[4990:].cpu().numpy(), ref) < TENSOR_RTOL


@pytest.mark.parametrize("split", [False, True])
def test_fused_many_samples_few_hypotheses(ops, ahv, oracle, g128, dev, split):
    """B larger than the number of CUs (each workgroup loops over several samples) and N smaller than a workgroup."""
    with ops.split_f16_scorer(split):
        rng = np.random.RandomState(3)
        B, N = 300, 5
        vs = (rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32)
        vt = (rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32)
        R = ahv.rotations.haar_rotations_np(N, 12)
        ft = ops.forward_3d2d(to_dev(vt, dev), to_dev(g128["W1"], dev), to_dev(g128["W2"], dev), to_dev(g128["b2"], dev))
        s, key = ops.score_hypotheses(to_dev(vs, dev), ft, to_dev(R, dev), to_dev(g128["W1"], dev), to_dev(g128["W2"], dev),
                                      to_dev(g128["b2"], dev))
        ref, _, ref_idx = oracle.score_hypotheses(vs, vt, R, g128["W1"], g128["W2"], g128["b2"])
        assert score_relerr(s.cpu().numpy(), ref) < SCORE_RTOL
        assert ops.unpack_best(key)[1].cpu().tolist() == ref_idx.tolist()


def test_planted_rotation_recovered_at_full_size(ops, ahv, G, dev):
    """Size-independent property at BASELINE.json's N = 50 000: if the target volume IS the source rotated by one of
    the hypotheses, that hypothesis scores 1 (its feature equals the target's) and wins the arg-max, wherever it
    sits; and coarse (Haar) -> fine (local perturbations) brings an UNLISTED rotation to within a few degrees."""
    N = 50_000
    R = ops.random_rotations(N, seed=5, device=dev)
    for planted in (0, 31_337, N - 1):
        vt = ops.rotate_volume(G["vol_src"], R[planted][None])
        ft = ops.forward_3d2d(vt, G["W1"], G["W2"], G["b2"])
        scores, key = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"])
        val, idx = ops.unpack_best(key)
        assert idx.item() == planted and abs(val.item() - 1.0) < 1e-5
        assert (scores[0] > 1.0 + 1e-5).sum().item() == 0
    # unlisted ground truth: the winner of 50 000 Haar samples is its nearest-looking neighbour; refine around it
    R_true = torch.from_numpy(ahv.rotations.haar_rotations_np(1, 123)).to(dev)
    vt = ops.rotate_volume(G["vol_src"], R_true)
    ft = ops.forward_3d2d(vt, G["W1"], G["W2"], G["b2"])
    _, key = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], want_scores=False)
    _, _, R1 = ops.select_rotation(key, R)
    e1 = ahv.rotations.geodesic_deg(R1, R_true).item()
    fine = ahv.rotations.refine_rotations(R1[0].cpu(), 4000, max_angle_deg=max(e1 * 1.5, 2.0),
                                          generator=torch.Generator().manual_seed(1)).to(dev)
    _, key2 = ops.score_hypotheses(G["vol_src"], ft, fine, G["W1"], G["W2"], G["b2"], want_scores=False)
    _, _, R2 = ops.select_rotation(key2, fine)
    e2 = ahv.rotations.geodesic_deg(R2, R_true).item()
    print("coarse error %.2f deg -> refined %.2f deg" % (e1, e2))
    assert e1 < 15.0 and e2 < e1 and e2 < 3.0


def test_device_so3_grid_matches_host_grid(ops, ahv, dev):
    """ahv_so3_grid_f32 = rotations.so3_grid_np (fp64 host maths) to fp32 rounding, also for the far end of a
    200 000-point grid (angles ~ 10^6 rad) and for shards generated independently."""
    for n in (1, 1000, 200_000):
        host = ahv.rotations.so3_grid_np(n)
        got = ops.so3_grid(n, dev).cpu().numpy()
        assert got.shape == (n, 3, 3) and np.abs(got - host).max() < 2e-6
    lo, hi = ahv.dist.shard_range(200_000, 3, 8)
    shard = ops.so3_grid(200_000, dev, offset=lo, n=hi - lo)
    assert torch.equal(shard, ops.so3_grid(200_000, dev)[lo:hi])
    Rd = ops.so3_grid(50_000, dev).double()
    eye = torch.eye(3, dtype=torch.float64, device=dev)
    assert (Rd @ Rd.transpose(1, 2) - eye).abs().max().item() < 5e-6 and (torch.linalg.det(Rd) - 1).abs().max().item() < 5e-6
    with pytest.raises(RuntimeError):
        ops.so3_grid(10, dev, offset=8, n=5)


def test_fused_on_a_side_stream_and_in_a_graph(ops, G, g128, dev):
    """Launches go to torch's CURRENT stream (whatever it is) and are capturable: same scores from a side stream and
    from a replayed hipGraph as from the default stream."""
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    ref, ref_key = ops.score_hypotheses(G["vol_src"], ft, G["R"], G["W1"], G["W2"], G["b2"])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s1, k1 = ops.score_hypotheses(G["vol_src"], ft, G["R"], G["W1"], G["W2"], G["b2"])
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(s1, ref) and torch.equal(k1, ref_key)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        s2, k2 = ops.score_hypotheses(G["vol_src"], ft, G["R"], G["W1"], G["W2"], G["b2"])
    s2.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(s2, ref) and torch.equal(k2, ref_key)
