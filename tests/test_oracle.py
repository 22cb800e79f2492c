"""CPU oracle (oracle/ahv_oracle.c, oracle/torch_ref.py) pinned against the golden vectors that
tools/gen_golden.py produced by running the reference's own code.  No GPU."""
import hashlib

import numpy as np
import pytest
import torch

from .conftest import load_golden

RTOL = 1e-5  # oracle vs reference: same fp32 maths, different summation order only


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def test_rotate_volume_matches_reference(oracle, g128):
    out = oracle.rotate_volume(g128["vol_src"], g128["R"][:2])
    assert out.shape == (2, 16, 8, 8, 8)
    assert relerr(out, g128["rot_first2"]) < RTOL


def test_forward_3d2d_matches_reference(oracle, g128):
    f_tgt = oracle.forward_3d2d(g128["vol_tgt"], g128["W1"], g128["W2"], g128["b2"])
    assert relerr(f_tgt, g128["f_tgt"]) < RTOL
    f_src = oracle.forward_3d2d(g128["rot_first2"], g128["W1"], g128["W2"], g128["b2"])
    assert relerr(f_src, g128["f_src_first2"]) < RTOL
    # unit norm along channels (modules/modules.py:122)
    assert np.allclose(np.linalg.norm(f_src, axis=1), 1.0, atol=1e-5)


def test_score_and_argmax_n128(oracle, g128):
    scores, best, idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], g128["R"], g128["W1"],
                                                g128["W2"], g128["b2"])
    assert relerr(scores, g128["scores"]) < RTOL
    assert idx[0] == g128["best_idx"][0]
    assert abs(best[0] - g128["best"][0]) <= 1e-6


def test_score_n4096(oracle, g128):
    g = load_golden("score_n4096")
    scores, best, idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], g["R"], g128["W1"], g128["W2"],
                                                g128["b2"])
    assert np.max(np.abs(scores - g["scores"]) / np.abs(g["scores"]).clip(1e-3)) < 1e-4
    assert idx[0] == g["best_idx"][0]
    assert float(g["top2_margin"]) > 1e-4  # arg-max is well separated from fp32 noise


def test_score_n50k_digest(oracle, ahv, g128):
    g = load_golden("score_n50k_digest")
    R = ahv.rotations.haar_rotations_np(int(g["n"]), int(g["seed"]))
    assert hashlib.sha256(R.tobytes()).hexdigest() == str(g["R_sha256"])
    scores, best, idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], R, g128["W1"], g128["W2"],
                                                g128["b2"])
    assert idx[0] == g["best_idx"][0]
    assert relerr(scores[0, ::97], g["every97_score"]) < RTOL
    assert relerr(scores[0, g["top16_idx"]], g["top16_score"]) < RTOL
    order = np.argsort(-scores[0], kind="stable")[:16]
    assert list(order) == list(g["top16_idx"])


def test_edge_rotations(oracle, g128):
    g = load_golden("edge_rotations")
    rot = oracle.rotate_volume(g128["vol_src"], g["R"])
    names = list(g["names"])
    # identity reproduces the volume bit for bit; 90-degree rotations permute voxels exactly
    assert np.array_equal(rot[names.index("identity")], g128["vol_src"][0])
    for i in range(1, 25):
        assert np.array_equal(np.sort(rot[i].ravel()), np.sort(g128["vol_src"][0].ravel()))
    assert relerr(rot[g["rot_keep_idx"]], g["rot_keep"]) < RTOL
    zero_frac = (rot == 0).reshape(rot.shape[0], -1).mean(axis=1)
    assert np.allclose(zero_frac, g["rot_zero_frac"], atol=2e-3)
    assert zero_frac[names.index("double")] == pytest.approx(0.875)
    scores, best, idx = oracle.score_hypotheses(g128["vol_src"], g128["vol_tgt"], g["R"], g128["W1"], g128["W2"],
                                                g128["b2"])
    assert np.max(np.abs(scores - g["scores"])) < 2e-6
    assert idx[0] == g["best_idx"][0]


def test_batched_shared_and_per_sample(oracle, g128):
    g = load_golden("batched")
    s, best, idx = oracle.score_hypotheses(g["vol_src"], g["vol_tgt"], g["R_shared"], g128["W1"], g128["W2"],
                                           g128["b2"])
    assert relerr(s, g["scores_shared"]) < RTOL
    assert list(idx) == list(g["best_idx_shared"])
    s2, _, _ = oracle.score_hypotheses(g["vol_src"], g["vol_tgt"], g["R_per"], g128["W1"], g128["W2"], g128["b2"])
    assert relerr(s2, g["scores_per"]) < RTOL


def test_metric(oracle):
    g = load_golden("metric")
    err = oracle.geodesic_deg(g["R_pred"], g["R_gt"])
    assert np.allclose(err, g["err_deg"], atol=2e-2, equal_nan=True)
    assert err[0] == 0.0 and abs(err[1] - 180.0) < 1e-3 and abs(err[2] - 15.0) < 1e-3


def test_argmax_first_index_and_nan(oracle):
    s = np.array([[0.1, 0.7, 0.7, 0.2], [np.nan, 1.0, np.nan, 0.0]], dtype=np.float32)
    best, idx = oracle.argmax(s)
    tb, ti = torch.max(torch.from_numpy(s), dim=1)
    assert list(idx) == list(ti.numpy())
    assert best[0] == tb[0].item() and np.isnan(best[1])


def test_torch_ref_matches_reference(g128):
    """The torch op-sequence restatement timed as cpu_baseline equals the reference outputs."""
    from oracle import torch_ref
    t = lambda k: torch.from_numpy(g128[k])
    scores, best, idx = torch_ref.score_hypotheses(t("vol_src"), t("vol_tgt"), t("R"), t("W1"), t("W2"), t("b2"))
    assert np.array_equal(scores.numpy(), g128["scores"])  # same ATen ops, same order: bit-exact
    assert idx.item() == int(g128["best_idx"][0])
    scores_c, _, idx_c = torch_ref.score_hypotheses(t("vol_src"), t("vol_tgt"), t("R"), t("W1"), t("W2"), t("b2"),
                                                    chunk=50)
    assert relerr(scores_c.numpy(), g128["scores"]) < 1e-6 and idx_c.item() == idx.item()


def test_nonfinite_inputs_match_the_reference(oracle, g128):
    """G10: ONE non-finite voxel / head weight.  F.grid_sample skips an out-of-range corner (a non-finite voxel next to it
    stays out of the sample: 0 * inf would be NaN), an in-range corner is accumulated even with weight 0, and F.relu
    propagates NaN of either sign: the oracle reproduces the reference's NaN mask exactly, its finite scores to RTOL and
    torch.max's index (the first NaN)."""
    from .conftest import nonfinite_cases
    n_cases = 0
    for name, inp, R, ref, ref_idx in nonfinite_cases(g128):
        scores, best, idx = oracle.score_hypotheses(inp["vol_src"], inp["vol_tgt"], R, inp["W1"], inp["W2"], inp["b2"])
        assert np.array_equal(np.isnan(scores), np.isnan(ref)), name
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(scores), fin), name
        if fin.any():
            assert np.max(np.abs(scores[fin] - ref[fin]) / np.abs(ref[fin]).clip(1e-2)) < 1e-4, name
        assert idx[0] == ref_idx, name
        n_cases += 1
    assert n_cases == 11


def infonce_from_torch_ref(g, tag, dtype=torch.float32):
    """modules/model_co3d.py:41-61 over oracle/torch_ref.py's operators on the G11 inputs -> (loss, sim, 5 gradients)."""
    from oracle import torch_ref
    t = lambda k: torch.from_numpy(g[k]).to(dtype)
    vs, vt = t(tag + "_vol_src").requires_grad_(True), t(tag + "_vol_tgt").requires_grad_(True)
    W1, W2, b2 = (t(k).requires_grad_(True) for k in ("W1", "W2", "b2"))
    R, pos = t(tag + "_R"), torch.from_numpy(g[tag + "_positive"])
    ft = torch_ref.forward_3d2d(vt, W1, W2, b2)
    sim = []
    for b in range(vs.shape[0]):
        n = R.shape[1]
        f = torch_ref.forward_3d2d(torch_ref.rotate_volume(vs[b:b + 1].expand(n, -1, -1, -1, -1), R[b]), W1, W2, b2)
        sim.append((f * ft[b:b + 1]).sum(dim=1).mean(dim=-1))
    sim = torch.stack(sim)
    e = torch.exp(sim / 0.1)
    loss = (-torch.log((e * pos).sum(dim=-1) / e.sum(dim=-1).clamp(min=1e-8))).mean()
    return loss, sim, torch.autograd.grad(loss, [vs, vt, W1, W2, b2])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_torch_ref_autograd_matches_the_reference_gradients(tag):
    """G11 `infonce_grad`: loss, similarities and d loss / d (vol_src, vol_tgt, W1, W2, b2) that the reference's own
    rotate_volume / forward_3d2d produced under autograd (tools/gen_golden.py gen_infonce).  The torch restatement
    runs the same ATen operators in the same order, forward and backward; the positive mask is re-derived from R and
    the GT as modules/model_co3d.py:44-46 does."""
    g = load_golden("infonce_grad")
    R, gt = torch.from_numpy(g[tag + "_R"]), torch.from_numpy(g[tag + "_gt"])
    gt_sim = (torch.sum(R.flatten(2) * gt.view(-1, 1, 9), dim=-1).clamp(-1, 3) - 1) / 2
    assert np.array_equal((180 * torch.arccos(gt_sim) / np.pi <= int(g["acc_thr"])).numpy(), g[tag + "_positive"])
    assert g[tag + "_positive"][:, :3].tolist() == [[True, True, False]] * R.shape[0]
    loss, sim, grads = infonce_from_torch_ref(g, tag)
    # same ATen operators in the same order; the reductions' summation order follows the thread count, hence not bit-exact
    assert relerr(sim.detach().numpy(), g[tag + "_sim"]) < 1e-6
    assert abs(loss.item() - float(g[tag + "_loss"])) <= 2e-6
    for got, key in zip(grads, ("d_vol_src", "d_vol_tgt", "d_W1", "d_W2", "d_b2")):
        assert relerr(got.numpy(), g[tag + "_" + key]) < 1e-5, key
    # and the fp64 evaluation of the same graph (what tests/test_gpu_backward.py holds the HIP backward to) agrees
    # with the reference's fp32 gradients to fp32 rounding
    _, _, grads64 = infonce_from_torch_ref(g, tag, torch.float64)
    for got, key in zip(grads64, ("d_vol_src", "d_vol_tgt", "d_W1", "d_W2", "d_b2")):
        assert relerr(got.numpy(), g[tag + "_" + key].astype(np.float64)) < 2e-4, key
