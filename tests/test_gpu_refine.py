"""Coarse-to-fine verify step (BASELINE.json configs[4] shape, scaled down): device-side compose /
select kernels and hipGraph replay against an eager torch recomputation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(ahv, g128):
    dev = torch.device("cuda:0")
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g128[k])).to(dev)
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "batched.npz"))
    vs, vt = torch.from_numpy(g["vol_src"]).to(dev), torch.from_numpy(g["vol_tgt"]).to(dev)
    return dev, vs, vt, T("W1"), T("W2"), T("b2")


def test_compose_and_select_kernels(ahv, setup):
    dev, vs, vt, W1, W2, b2 = setup
    ops = ahv.ops
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(500, 3)).to(dev)
    D = ahv.rotations.refine_rotations(torch.eye(3), 64, 8.0, generator=torch.Generator().manual_seed(1)).to(dev)
    ft = ops.forward_3d2d(vt, W1, W2, b2)
    s, key = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
    val, idx = ops.unpack_best(key)
    score2, idx2, R_pred = ops.select_rotation(key, R)
    assert torch.equal(idx2, idx) and torch.equal(score2, val) and torch.equal(R_pred, R[idx])
    fine = ops.compose_rotations(key, R, D)
    ref = torch.matmul(R[idx][:, None], D[None])
    assert fine.shape == (3, 64, 3, 3) and torch.allclose(fine, ref, atol=1e-6)
    assert torch.allclose(fine[:, 0], R[idx], atol=1e-6)  # D[0] = I
    # sharded view: a rank that does not own the winner writes zeros, the owner writes the row
    lo = int(idx[0].item()) + 1
    _, gidx, Rz = ops.select_rotation(key[:1], R[lo:], n_offset=lo)
    assert gidx.item() == idx[0].item() and torch.count_nonzero(Rz) == 0
    _, _, Ro = ops.select_rotation(key[:1], R[: lo], n_offset=0)
    assert torch.equal(Ro[0], R[idx[0]])


@pytest.mark.parametrize("use_graph", [False, True])
def test_coarse_to_fine_matches_two_stage_reference(ahv, setup, use_graph):
    dev, vs, vt, W1, W2, b2 = setup
    ops = ahv.ops
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(2000, 5)).to(dev)
    c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=300, max_angle_deg=12.0, batch=3, use_graph=use_graph)
    for rep in range(3):  # replays with different inputs
        a, b = (vs, vt) if rep != 1 else (vt, vs)
        score, idx, R_pred, c_score, c_idx = [t.clone() for t in c2f(a, b)]
        ft = ops.forward_3d2d(b, W1, W2, b2)
        s1, _ = ops.score_hypotheses(a, ft, R, W1, W2, b2)
        v1, i1 = torch.max(s1, dim=1)
        assert torch.equal(c_idx, i1) and torch.equal(c_score, v1)
        fine = torch.matmul(R[i1][:, None], c2f.D[None]).contiguous()
        s2, _ = ops.score_hypotheses(a, ft, fine, W1, W2, b2)
        v2, i2 = torch.max(s2, dim=1)
        assert torch.equal(idx, i2)
        assert torch.allclose(score, v2, rtol=1e-5) and torch.all(score >= c_score - 1e-6)
        assert torch.allclose(R_pred, fine[torch.arange(3), i2], atol=1e-6)
