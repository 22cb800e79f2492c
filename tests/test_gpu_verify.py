"""Round-4 parity tests of the fused path through the C ABI:

* ``ahv_verify_pair_f32`` -- the whole of test_co3d.py:137-145 behind one entry point, target features built inside the
  scoring launch -- against the goldens, the oracle and the two-launch path;
* the team tail (four waves per hypothesis for the remainder of a launch, csrc/ahv_team.h) against single waves;
* the reference-generated digests of BASELINE.json configs[2] (200 000-point SO(3) grid, G8) and of a B = 32 batch (G9),
  for the fp32 and the split-f16 kernel;
* signed-order keys: reset through ``ahv_select_rotation_f32``, merge by plain int64 max;
* non-finite inputs: what is guaranteed (any R) and what is documented as unspecified (non-finite voxels).
"""
import hashlib

import numpy as np
import pytest
import torch

from .conftest import load_golden

pytestmark = pytest.mark.gpu

SCORE_RTOL, SCORE_FLOOR = 1e-4, 1e-2   # north-star tolerance (test_gpu_parity.py)
ORDER_ATOL = 1e-6                      # same fp32 arithmetic, sums associated differently (in-launch target features)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(ahv):
    ahv._lib.load()  # raises if libahv_hip.so is missing: no fallback
    return ahv.ops


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def G(dev, g128):
    return {k: to_dev(g128[k], dev) for k in ["vol_src", "vol_tgt", "R", "W1", "W2", "b2"]}


def relerr(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), SCORE_FLOOR)))


def two_launch(ops, G, R, vs=None, vt=None, **kw):
    vs, vt = (G["vol_src"] if vs is None else vs), (G["vol_tgt"] if vt is None else vt)
    ft = ops.forward_3d2d(vt, G["W1"], G["W2"], G["b2"])
    s, k = ops.score_hypotheses(vs, ft, R, G["W1"], G["W2"], G["b2"], **kw)
    return s, k, ft


# ------------------------------------------------------------------------------------------- one-launch verify

def test_verify_pair_config1_golden_and_oracle(ops, oracle, G, g128):
    s, key, ft = ops.verify_pair(G["vol_src"], G["vol_tgt"], G["R"], G["W1"], G["W2"], G["b2"], want_feat_tgt=True)
    val, idx = ops.unpack_best(key)
    assert relerr(s.cpu().numpy(), g128["scores"]) <= SCORE_RTOL
    assert int(idx.item()) == int(g128["best_idx"][0])
    assert val.item() == s[0, idx.item()].item()           # the key carries the exact fp32 score
    assert np.max(np.abs(ft.cpu().numpy() - g128["f_tgt"])) <= 1e-5     # in-launch forward_3d2d(vol_tgt) vs the reference's
    ref, _, ref_idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], g128["R"], g128["W1"], g128["W2"], g128["b2"])
    assert relerr(s.cpu().numpy(), ref) <= SCORE_RTOL and int(idx.item()) == int(ref_idx[0])
    # arg-max only: same key
    _, key2 = ops.verify_pair(G["vol_src"], G["vol_tgt"], G["R"], G["W1"], G["W2"], G["b2"], want_scores=False)
    assert torch.equal(key2, key)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 129, 300, 511, 513, 1024, 2048 + 100, 2048 + 512, 2048 + 513, 4096])
def test_verify_pair_equals_two_launches_at_every_n(ops, G, dev, n):
    """Every grid shape and tail mode of the launch plan (all-team launches, lone waves, full rounds + team tail, full
    rounds + lone tail): the one-launch step against forward_3d2d + score_hypotheses, and both against the golden."""
    g = load_golden("score_n4096")
    R = to_dev(g["R"][:n], dev)
    s1, k1, ft1 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, G["W1"], G["W2"], G["b2"], want_feat_tgt=True)
    s2, k2, ft2 = two_launch(ops, G, R)
    assert (ft1 - ft2).abs().max().item() <= ORDER_ATOL
    assert (s1 - s2).abs().max().item() <= ORDER_ATOL
    assert relerr(s1.cpu().numpy(), g["scores"][:, :n]) <= SCORE_RTOL
    ref_idx = int(np.argmax(g["scores"][0, :n]))
    assert int(ops.unpack_best(k1)[1].item()) == ref_idx == int(ops.unpack_best(k2)[1].item())
    # the key is torch.max of the launch's own scores (first maximal index)
    v, i = torch.max(s1, dim=1)
    bv, bi = ops.unpack_best(k1)
    assert bv.item() == v.item() and bi.item() == i.item()


def test_verify_pair_batched_shared_and_per_sample(ops, dev, G):
    g = load_golden("batched")
    vs, vt = to_dev(g["vol_src"], dev), to_dev(g["vol_tgt"], dev)
    for R, want in ((g["R_shared"], g["scores_shared"]), (g["R_per"], g["scores_per"])):
        s, key = ops.verify_pair(vs, vt, to_dev(R, dev), G["W1"], G["W2"], G["b2"])
        assert relerr(s.cpu().numpy(), want) <= SCORE_RTOL
        assert ops.unpack_best(key)[1].cpu().tolist() == list(np.argmax(want, axis=1))


def test_verify_pair_target_features_only_and_errors(ops, ahv, dev, G):
    empty = torch.empty(0, 3, 3, device=dev)
    s, key, ft = ops.verify_pair(G["vol_src"], G["vol_tgt"], empty, G["W1"], G["W2"], G["b2"], want_feat_tgt=True)
    assert s.shape == (1, 0) and int(key.item()) == ahv._lib.AHV_KEY_EMPTY
    assert (ft - ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])).abs().max().item() <= ORDER_ATOL
    with pytest.raises(RuntimeError):
        ops.verify_pair(G["vol_src"], G["vol_tgt"][:, :8], G["R"], G["W1"], G["W2"], G["b2"])
    lib = ahv._lib.load()   # the split kernel needs the feature buffer
    key = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = lib.ahv_verify_pair_f32(G["vol_src"].data_ptr(), G["vol_tgt"].data_ptr(), G["R"].data_ptr(), 0, 0, G["W1"].data_ptr(),
                                 G["W2"].data_ptr(), G["b2"].data_ptr(), 1, 128, None, key.data_ptr(), None,
                                 ahv._lib.AHV_SCORE_SPLIT_F16, None, None)
    assert rc == -1 and b"feat_tgt_out" in lib.ahv_last_error()


def test_verify_pair_split_kernel_and_graph_replay(ops, G, g128):
    s16, k16 = ops.verify_pair(G["vol_src"], G["vol_tgt"], G["R"], G["W1"], G["W2"], G["b2"], split_f16=True)
    assert relerr(s16.cpu().numpy(), g128["scores"]) <= SCORE_RTOL
    assert int(ops.unpack_best(k16)[1].item()) == int(g128["best_idx"][0])
    ref, ref_key = ops.verify_pair(G["vol_src"], G["vol_tgt"], G["R"], G["W1"], G["W2"], G["b2"])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        s2, k2 = ops.verify_pair(G["vol_src"], G["vol_tgt"], G["R"], G["W1"], G["W2"], G["b2"])
    for _ in range(3):
        s2.zero_()
        k2.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(s2, ref) and torch.equal(k2, ref_key)


# ------------------------------------------------------------------------------------------- team tails

@pytest.mark.parametrize("n", [7, 130, 512, 2048 + 37, 4 * 2048 + 300, 5 * 2048 + 512, 12500])
def test_team_tail_against_single_waves(ops, ahv, G, dev, n):
    """Teams (the default for a remainder of at most two hypotheses per workgroup) against AHV_SCORE_NO_TEAMS (every
    hypothesis by one wave): the SAME scores and keys bit for bit, whatever N, the hypothesis' position in the set or the
    entry point -- a score is a function of (volumes, weights, R_n) alone."""
    R = to_dev(ahv.rotations.haar_rotations_np(12500, seed=41)[:n], dev)
    assert ops.score_plan(1, n)[2] < n, "this case was expected to use the team path"
    s_t, k_t, _ = two_launch(ops, G, R)
    s_1, k_1, _ = two_launch(ops, G, R, no_teams=True)
    assert torch.equal(s_t, s_1) and torch.equal(k_t, k_1)
    big = to_dev(ahv.rotations.haar_rotations_np(12500, seed=41), dev)
    s_big, _, _ = two_launch(ops, G, big)
    assert torch.equal(s_big[:, :n], s_1)
    # a reversed set: every hypothesis changes its wave, workgroup and possibly its mode (team / single wave)
    s_rev_t, _, _ = two_launch(ops, G, R.flip(0).contiguous())
    assert torch.equal(s_rev_t.flip(1), s_1)
    # the one-launch step on the same set: in-launch target features (summed in another order than forward_3d2d's: the
    # scores agree to rounding with the two-launch ones) -- and among themselves bit for bit, teams or not
    s_v, k_v = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, G["W1"], G["W2"], G["b2"])
    assert (s_v - s_1).abs().max().item() <= ORDER_ATOL and torch.equal(ops.unpack_best(k_v)[1], ops.unpack_best(k_1)[1])
    s_v1, k_v1 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, G["W1"], G["W2"], G["b2"], no_teams=True)
    assert torch.equal(s_v1, s_v) and torch.equal(k_v1, k_v)


def test_team_tail_against_the_oracle(ops, oracle, ahv, G, g128, dev):
    """The remainder of a 12 500-hypothesis shard (six full rounds, then 212 hypotheses scored by teams) against the oracle."""
    Rn = ahv.rotations.haar_rotations_np(12500, seed=43)
    s, key, _ = two_launch(ops, G, to_dev(Rn, dev))
    s1, _, _ = two_launch(ops, G, to_dev(Rn, dev), no_teams=True)
    gx, gy, n_main = ops.score_plan(1, 12500)
    assert (gx, gy, n_main) == (256, 1, 12288), "the remainder was expected to go through the team path"
    assert ops.score_plan(1, 12500, no_teams=True)[2] == 12500
    assert torch.equal(s, s1)          # a team's score is the lone wave's, bit for bit (ahv_team.h)
    ref, _, _ = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], Rn[12288:], g128["W1"], g128["W2"], g128["b2"])
    assert relerr(s[:, 12288:].cpu().numpy(), ref) <= SCORE_RTOL
    v, i = torch.max(s, dim=1)
    bv, bi = ops.unpack_best(key)
    assert bv.item() == v.item() and bi.item() == i.item()


def test_spare_cus_change_the_grid_not_the_scores(ops, ahv, G, dev):
    R = to_dev(ahv.rotations.haar_rotations_np(5000, seed=47), dev)
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    ref, ref_key = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], no_teams=True)
    for spare in (1, 2, 17, 255):
        s, k = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], no_teams=True, spare_cus=spare)
        assert torch.equal(s, ref) and torch.equal(k, ref_key)
        s, k = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], spare_cus=spare)
        assert torch.equal(s, ref) and torch.equal(k, ref_key)    # another grid, another team split: the same bits
    with pytest.raises(RuntimeError):
        ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], spare_cus=256)


# ------------------------------------------------------------------------------------------- keys

def test_signed_keys_merge_by_plain_max_and_reset_through_select(ops, ahv, G, dev):
    R = to_dev(ahv.rotations.haar_rotations_np(3000, seed=53), dev)
    ft = ops.forward_3d2d(G["vol_tgt"], G["W1"], G["W2"], G["b2"])
    s, k_full = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], no_teams=True)
    parts = []
    for r in range(3):
        lo, hi = ahv.dist.shard_range(3000, r, 3)
        _, k = ops.score_hypotheses(G["vol_src"], ft, R[lo:hi], G["W1"], G["W2"], G["b2"], n_offset=lo, want_scores=False,
                                    no_teams=True)
        parts.append(k)
    assert torch.equal(torch.stack(parts).max(dim=0).values, k_full)      # what the int64 MAX all-reduce computes
    assert torch.equal(ahv.dist.merge_keys(torch.stack(parts)), k_full)
    # a negative best score survives the merge into a fresh key (signed order: the empty key is INT64_MIN, not 0)
    neg = torch.tensor([[-0.5, -0.25, -0.75]], device=dev)
    bv, bi = ops.unpack_best(ops.argmax(neg, return_key=True))
    assert bv.item() == -0.25 and bi.item() == 1
    # select with reset: same outputs, key handed back empty; a second select sees "nothing scored"
    key = k_full.clone()
    sc, idx, Rp = ops.select_rotation(key, R)
    assert torch.equal(key, k_full)
    sc2, idx2, Rp2 = ops.select_rotation(key, R, reset_key=True)
    assert torch.equal(sc2, sc) and torch.equal(idx2, idx) and torch.equal(Rp2, Rp) and torch.equal(Rp[0], R[idx.item()])
    assert int(key.item()) == ahv._lib.AHV_KEY_EMPTY
    sc3, idx3, Rp3 = ops.select_rotation(key, R)
    assert sc3.item() == float("-inf") and idx3.item() == -1 and Rp3.abs().sum().item() == 0
    # the emptied key is ready for the next step: merging into it (reset_best=False) gives the fresh result
    _, k_next = ops.score_hypotheses(G["vol_src"], ft, R, G["W1"], G["W2"], G["b2"], want_scores=False, best_key=key,
                                     reset_best=False, no_teams=True)
    assert torch.equal(k_next, k_full)
    assert int(ops.reset_best(torch.zeros(4, dtype=torch.int64, device=dev))[2].item()) == ahv._lib.AHV_KEY_EMPTY


# ------------------------------------------------------------------------------------------- reference digests

def _check_digest(scores, key, ops, g, every_name, every):
    s = scores[0].cpu().numpy()
    assert relerr(s[::every], g[every_name]) <= SCORE_RTOL
    assert relerr(s[g["top16_idx"]], g["top16_score"]) <= SCORE_RTOL
    bv, bi = ops.unpack_best(key)
    assert int(bi.item()) == int(g["best_idx"][0])
    assert abs(bv.item() - float(g["best"][0])) <= SCORE_RTOL * max(abs(float(g["best"][0])), SCORE_FLOOR)
    assert float(g["top2_margin"]) > 1e-5, "the fixture's top-2 margin is above fp32 noise: a unique winner"
    # the top-16 set and order of the reference (margins between neighbours above rounding are kept)
    top = np.argsort(-s, kind="stable")[:16]
    gaps = -np.diff(g["top16_score"])
    k = 0
    while k < 15 and gaps[k] > 2e-6:
        k += 1
    assert list(top[:k + 1]) == list(g["top16_idx"][:k + 1])
    v, i = torch.max(scores, dim=1)
    assert bv.item() == v.item() and bi.item() == i.item()


@pytest.mark.parametrize("split", [False, True])
def test_configs2_dense_grid_reference_digest(ops, ahv, G, dev, split):
    """G8: BASELINE.json configs[2] -- the 200 000-point super-Fibonacci SO(3) grid, scored by the REFERENCE's own hot loop
    (tools/gen_golden.py gen_grid_digest): best index, top-16, every 389th score.  The grid is regenerated here from
    rotations.so3_grid_np (bit-reproducible by construction) and checked against the stored SHA-256."""
    g = load_golden("score_n200k_grid_digest")
    Rn = ahv.rotations.so3_grid_np(int(g["n"]))
    assert hashlib.sha256(Rn.tobytes()).hexdigest() == str(g["R_sha256"])
    R = to_dev(Rn, dev)
    s, key, _ = two_launch(ops, G, R, split_f16=split)
    _check_digest(s, key, ops, g, "every389_score", 389)
    assert [int(x) for x in g["tie_set_1e6"]] == [int(g["best_idx"][0])]
    s2, key2 = ops.verify_pair(G["vol_src"], G["vol_tgt"], R, G["W1"], G["W2"], G["b2"], split_f16=split)
    _check_digest(s2, key2, ops, g, "every389_score", 389)
    # the device-generated grid (what a production run uses) selects the same rotation
    Rd = ops.so3_grid(int(g["n"]), dev)
    _, keyd, _ = two_launch(ops, G, Rd, split_f16=split, want_scores=False)
    assert int(ops.unpack_best(keyd)[1].item()) == int(g["best_idx"][0])


@pytest.mark.parametrize("split", [False, True])
def test_batch32_reference_digest(ops, ahv, G, dev, split):
    """G9: B = 32 volume pairs (the reference's forward_2d3d outputs, fp16-representable), one shared R(4096)
    (modules/model.py:184-196): per-sample best index and score, every 61st score."""
    g = load_golden("batched32_digest")
    Rn = ahv.rotations.haar_rotations_np(int(g["n"]), seed=int(g["seed"]))
    assert hashlib.sha256(Rn.tobytes()).hexdigest() == str(g["R_sha256"])
    vs, vt = to_dev(g["vol_src"].astype(np.float32), dev), to_dev(g["vol_tgt"].astype(np.float32), dev)
    R = to_dev(Rn, dev)
    assert float(g["top2_margin"].min()) > 1e-5
    for fn in ("two", "one"):
        if fn == "two":
            s, key, _ = two_launch(ops, G, R, vs=vs, vt=vt, split_f16=split)
        else:
            s, key = ops.verify_pair(vs, vt, R, G["W1"], G["W2"], G["b2"], split_f16=split)
        assert relerr(s[:, ::61].cpu().numpy(), g["every61_score"]) <= SCORE_RTOL
        bv, bi = ops.unpack_best(key)
        assert bi.cpu().tolist() == g["best_idx"].tolist()
        assert relerr(bv.cpu().numpy(), g["best"]) <= SCORE_RTOL
    # sharded 8 ways with the all-reduce's merge (configs[3]'s split): same winners
    keys = []
    for r in range(8):
        lo, hi = ahv.dist.shard_range(int(g["n"]), r, 8)
        keys.append(ops.verify_pair(vs, vt, R[lo:hi], G["W1"], G["W2"], G["b2"], n_offset=lo, want_scores=False,
                                    split_f16=split)[1])
    assert ops.unpack_best(torch.stack(keys).max(dim=0).values)[1].cpu().tolist() == g["best_idx"].tolist()


# ------------------------------------------------------------------------------------------- non-finite inputs

def test_any_rotation_matrix_matches_the_reference(ops, oracle, G, g128, dev):
    """include/ahv.h, "Non-finite inputs": R may hold ANY values.  NaN / inf entries send the sample points out of the
    volume (ATen: floor -> out of bounds -> zeros padding), i.e. a rotated volume of zeros; scores stay finite and equal
    the reference's.  Through both entry points, single waves and teams."""
    Rn = np.array(g128["R"][:40])
    Rn[3, 1, 1] = np.nan
    Rn[7] = np.nan
    Rn[11, 0, 2] = np.inf
    Rn[12, 2, 0] = -np.inf
    Rn[20] = 1e30
    Rn[21] = 0.0
    ref, _, ref_idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], Rn, g128["W1"], g128["W2"], g128["b2"])
    assert np.all(np.isfinite(ref))
    for kw in ({}, {"no_teams": True}):
        s, key, _ = two_launch(ops, G, to_dev(Rn, dev), **kw)
        assert torch.isfinite(s).all() and relerr(s.cpu().numpy(), ref) <= SCORE_RTOL
        assert int(ops.unpack_best(key)[1].item()) == int(ref_idx[0])
    s, key = ops.verify_pair(G["vol_src"], G["vol_tgt"], to_dev(Rn, dev), G["W1"], G["W2"], G["b2"])
    assert relerr(s.cpu().numpy(), ref) <= SCORE_RTOL


@pytest.mark.parametrize("split", [False, True])
def test_non_finite_inputs_match_the_reference(ops, oracle, ahv, G, g128, dev, split):
    """G10 `nonfinite` (the REFERENCE's scores on inputs with ONE non-finite voxel or head weight) and the oracle on more of
    them: the same NaN mask, the same finite scores, torch.max's index (the first NaN) -- for the fp32 and the split-f16
    scorer, through both entry points, single waves and teams, B > 1 with one flagged sample.  A workgroup notices a
    non-finite value while it stages the sample and sends that sample through the exact path (csrc/ahv_exact.h):
    grid_sample's per-corner zeros padding and F.relu's NaN propagation restated literally (utils.py:129,
    modules/modules.py:68).  Rounds 1-4 documented this as outside the contract."""
    from .conftest import nonfinite_cases

    def check(s, key, ref, ref_idx, name):
        got = s.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin), name
        if fin.any():
            assert relerr(got[fin], ref[fin]) <= SCORE_RTOL, name
        if ref_idx is not None:
            assert int(ops.unpack_best(key)[1][0].item()) == ref_idx, name

    cases = list(nonfinite_cases(g128))
    assert len(cases) == 11
    for name, inp, Rn, ref, ref_idx in cases:
        d = {k: to_dev(v, dev) for k, v in inp.items()}
        R = to_dev(Rn, dev)
        W = (d["W1"], d["W2"], d["b2"])
        s1, k1 = ops.verify_pair(d["vol_src"], d["vol_tgt"], R, *W, split_f16=split)
        check(s1, k1, ref, ref_idx, name + " one launch")
        ft = ops.forward_3d2d(d["vol_tgt"], *W)
        s2, k2 = ops.score_hypotheses(d["vol_src"], ft, R, *W, split_f16=split)
        check(s2, k2, ref, ref_idx, name + " two launches")
        s3, k3 = ops.verify_pair(d["vol_src"], d["vol_tgt"], R, *W, split_f16=split, no_teams=True)
        check(s3, k3, ref, ref_idx, name + " single waves")
    # more positions, values and a full-size set against the oracle (itself pinned by G10: tests/test_oracle.py)
    rng = np.random.default_rng(5)
    Rn = ahv.rotations.haar_rotations_np(2500, seed=77)
    for trial in range(6):
        v = np.array(g128["vol_src"])
        for _ in range(1 + trial % 3):
            idx = (0, rng.integers(16), rng.integers(8), rng.integers(8), rng.integers(8))
            v.view(np.uint32)[idx] = [0x7FC00000, 0xFFC00000, 0x7F800000, 0xFF800000][rng.integers(4)]
        ref, _, ref_idx = oracle.score_hypotheses(v, g128["vol_tgt"], Rn, g128["W1"], g128["W2"], g128["b2"])
        s, key = ops.verify_pair(to_dev(v, dev), G["vol_tgt"], to_dev(Rn, dev), G["W1"], G["W2"], G["b2"], split_f16=split)
        check(s, key, ref, int(ref_idx[0]), "random voxels %d" % trial)
        # a batch of three: the flagged sample in the middle, its finite neighbours untouched (bit for bit)
        vs3 = torch.stack([G["vol_src"][0], to_dev(v, dev)[0], G["vol_src"][0]])
        vt3 = G["vol_tgt"].expand(3, -1, -1, -1, -1).contiguous()
        s3, _ = ops.verify_pair(vs3, vt3, to_dev(Rn[:300], dev), G["W1"], G["W2"], G["b2"], split_f16=split)
        s_fin, _ = ops.verify_pair(G["vol_src"], G["vol_tgt"], to_dev(Rn[:300], dev), G["W1"], G["W2"], G["b2"], split_f16=split)
        assert torch.equal(s3[0], s_fin[0]) and torch.equal(s3[2], s_fin[0])
        check(s3[1:2], None, ref[:, :300], None, "batched %d" % trial)
