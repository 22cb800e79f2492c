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
        # the step's own sequence: coarse stage = one verify_pair launch whose in-launch target features feed the fine stage
        s1, _, ft = ops.verify_pair(a, b, R, W1, W2, b2, want_feat_tgt=True)
        v1, i1 = torch.max(s1, dim=1)
        assert torch.equal(c_idx, i1) and torch.equal(c_score, v1)
        fine = torch.matmul(R[i1][:, None], c2f.D[None]).contiguous()
        s2, _ = ops.score_hypotheses(a, ft, fine, W1, W2, b2)
        v2, i2 = torch.max(s2, dim=1)
        assert torch.equal(idx, i2)
        assert torch.allclose(score, v2, rtol=1e-5) and torch.all(score >= c_score - 1e-6)
        assert torch.allclose(R_pred, fine[torch.arange(3), i2], atol=1e-6)


# ---- the one-launch step (ahv_coarse_to_fine_f32) against the five-launch one ------------------------------------
@pytest.mark.parametrize("B,n_coarse,n_fine", [(3, 2000, 300), (1, 10_000, 1000), (2, 4096, 2048), (1, 7, 5)])
def test_one_launch_step_equals_the_five_launch_step(ahv, setup, B, n_coarse, n_fine):
    """Same scores bit for bit (every hypothesis by one wave: no_teams), same winners, same R_pred; the keys and the
    meeting point's counters come back empty, so the same state serves step after step."""
    dev, vs3, vt3, W1, W2, b2 = setup
    ops = ahv.ops
    vs, vt = vs3[:B].contiguous(), vt3[:B].contiguous()
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(n_coarse, 17)).to(dev)
    five = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=n_fine, max_angle_deg=10.0, batch=B, use_graph=False,
                                   want_scores=True, no_teams=True)
    assert not five.fused
    state = ops.CoarseToFineState(B, dev)
    out = {}
    for rep in range(3):
        a, b = (vs, vt) if rep != 1 else (vt, vs)
        score, idx, R_pred, c_score, c_idx = [t.clone() for t in five(a, b)]
        r = ops.coarse_to_fine(a, b, R, five.D, W1, W2, b2, state=state, want_scores=True, want_feat_tgt=True, no_teams=True, out=out)
        assert torch.equal(r["coarse_scores"], five.last["coarse_scores"])
        assert torch.equal(r["fine_scores"], five.last["fine_scores"])
        assert torch.equal(r["coarse_idx"], c_idx) and torch.equal(r["coarse_score"], c_score)
        assert torch.equal(r["fine_idx"], idx) and torch.equal(r["fine_score"], score)
        assert torch.equal(r["R_pred"], R_pred)
        assert torch.equal(r["feat_tgt"], ops.verify_pair(a, b, R[:1], W1, W2, b2, want_feat_tgt=True)[2])
        assert torch.all(state.keys == ahv._lib.AHV_KEY_EMPTY) and torch.count_nonzero(state.sync) == 0
    assert not state.gave_up()


def test_one_launch_step_with_team_tails_and_per_sample_sets(ahv, setup):
    """Default plan (team tails where the plan takes them) and a (B,N,3,3) coarse set: winners and R_pred as the
    five-launch step finds them, scores to rounding."""
    dev, vs, vt, W1, W2, b2 = setup
    ops = ahv.ops
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(3 * 8300, 23).reshape(3, 8300, 3, 3)).to(dev)
    five = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=300, batch=3, use_graph=False, want_scores=True)
    score, idx, R_pred, c_score, c_idx = [t.clone() for t in five(vs, vt)]
    r = ops.coarse_to_fine(vs, vt, R, five.D, W1, W2, b2, want_scores=True)
    assert torch.equal(r["coarse_idx"], c_idx) and torch.equal(r["fine_idx"], idx)
    assert torch.equal(r["coarse_scores"], five.last["coarse_scores"])      # teams or single waves: the same bits
    assert torch.allclose(r["fine_scores"], five.last["fine_scores"], atol=5e-7, rtol=0)
    assert torch.allclose(r["fine_score"], score, atol=5e-7, rtol=0) and torch.equal(r["R_pred"], R_pred)
    assert not r["state"].gave_up()


def test_one_launch_step_that_gave_up_is_poisoned_and_named(ahv, setup):
    """a workgroup that abandons the device-wide meeting point (CUs held by another kernel for ~1 s)
    scores stage 1 against an incomplete coarse key.  The launch sets its sticky error word -- and the last workgroup then
    writes POISON instead of results (NaN scores, index -1, NaN rotation), so the failure cannot pass for a result, and
    CoarseToFine.check() names it.  The give-up itself needs a second of starvation; here the error word is set by hand,
    which is the state the decode sees after one."""
    dev, vs3, vt3, W1, W2, b2 = setup
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(3000, 29)).to(dev)
    c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=200, batch=2, use_graph=False, fused=True)
    good = [t.clone() for t in c2f(vs3[:2].contiguous(), vt3[:2].contiguous())]
    c2f.check()                                                    # nothing to report
    assert torch.isfinite(good[0]).all() and (good[1] >= 0).all()
    c2f._fused_state.sync[-1] = 1
    score, idx, R_pred, c_score, c_idx = c2f(vs3[:2].contiguous(), vt3[:2].contiguous())
    assert torch.isnan(score).all() and torch.isnan(c_score).all() and torch.isnan(R_pred).all()
    assert (idx == -1).all() and (c_idx == -1).all()
    with pytest.raises(RuntimeError, match="gave up the meeting point"):
        c2f.check()
    # the word is sticky -- the NEXT step is poisoned as well, until clear_error() resets the state
    again = c2f(vs3[:2].contiguous(), vt3[:2].contiguous())
    assert torch.isnan(again[0]).all()
    with pytest.raises(RuntimeError, match="sticky"):
        c2f.check()
    c2f.clear_error()
    clean = c2f(vs3[:2].contiguous(), vt3[:2].contiguous())
    c2f.check()
    for a, b in zip(clean, good):
        assert torch.equal(a, b)


def test_one_launch_step_through_CoarseToFine_eager_and_from_a_graph(ahv, setup):
    dev, vs, vt, W1, W2, b2 = setup
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(5000, 29)).to(dev)
    eager = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=500, batch=3, use_graph=False, fused=True)
    graph = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=500, batch=3, use_graph=True, fused=True)
    five = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=500, batch=3, use_graph=False)
    assert eager.fused and graph.fused and graph.use_graph and not five.fused
    for rep in range(4):
        a, b = (vs, vt) if rep % 2 == 0 else (vt, vs)
        e = [t.clone() for t in eager(a, b)]
        g = [t.clone() for t in graph(a, b)]
        f = [t.clone() for t in five(a, b)]
        for x, y, z in zip(e, g, f):
            assert torch.equal(x, y)
            assert torch.equal(x, z) if x.dtype == torch.int64 else torch.allclose(x, z, atol=5e-7, rtol=0)
    assert not eager._fused_state.gave_up() and not graph._fused_state.gave_up()


def test_one_launch_step_argument_checks(ahv, setup):
    dev, vs, vt, W1, W2, b2 = setup
    ops = ahv.ops
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(64, 1)).to(dev)
    D = R[:8].contiguous()
    with pytest.raises(RuntimeError, match="made for B"):
        ops.coarse_to_fine(vs, vt, R, D, W1, W2, b2, state=ops.CoarseToFineState(2, dev))
    with pytest.raises(RuntimeError, match="empty hypothesis set"):
        ops.coarse_to_fine(vs, vt, R, D[:0], W1, W2, b2)
    with pytest.raises(RuntimeError, match="one rank"):
        ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=8, batch=3, fused=True, backend=ops)


# ---- BASELINE.json configs[4] at full size: 10 000 coarse + 1 000 refined, hipGraph, against the ORACLE ----------
@pytest.mark.parametrize("B", [1, 3])
def test_configs4_full_size_graph_vs_oracle(ahv, oracle, setup, B):
    """Both stages of CoarseToFine (use_graph=True) against oracle.score_hypotheses on the same two hypothesis
    sets (verify semantics: modules/model.py:183-196): coarse winner exact, scores of both stages <= 1e-4
    relative, fine index exact, R_pred = the winning refinement."""
    dev, vs3, vt3, W1, W2, b2 = setup
    vs, vt = vs3[:B].contiguous(), vt3[:B].contiguous()
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(10_000, 40)).to(dev)
    c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, max_angle_deg=10.0, batch=B, use_graph=True,
                                  want_scores=True)
    assert c2f.use_graph
    W = [t.cpu().numpy() for t in (W1, W2, b2)]
    rel = lambda got, ref: float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2)))
    for rep in range(2):  # second replay: swapped inputs through the same captured graph
        a, b = (vs, vt) if rep == 0 else (vt, vs)
        score, idx, R_pred, c_score, c_idx = [t.clone().cpu().numpy() for t in c2f(a, b)]
        s1 = c2f.last["coarse_scores"].cpu().numpy()
        s2 = c2f.last["fine_scores"].cpu().numpy()
        R_fine = c2f.last["R_fine"].cpu().numpy()
        an, bn = a.cpu().numpy(), b.cpu().numpy()
        ref1, best1, idx1 = oracle.score_hypotheses(an, bn, R.cpu().numpy(), *W)
        assert np.array_equal(c_idx, idx1)
        assert rel(s1, ref1) < 1e-4 and rel(c_score, best1) < 1e-4
        want_fine = np.matmul(R.cpu().numpy()[idx1][:, None], c2f.D.cpu().numpy()[None])
        assert R_fine.shape == (B, 1000, 3, 3) and np.max(np.abs(R_fine - want_fine)) < 1e-6
        ref2, best2, idx2 = oracle.score_hypotheses(an, bn, R_fine, *W)   # per-sample sets (B,N2,3,3)
        assert rel(s2, ref2) < 1e-4 and rel(score, best2) < 1e-4
        assert np.array_equal(idx, idx2), (idx, idx2, np.sort(ref2, axis=1)[:, -2:])
        assert np.array_equal(R_pred, R_fine[np.arange(B), idx2])
        assert np.all(score >= c_score - 1e-6)


RANK_WORKER = r'''
import importlib, os, sys, numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["AHV_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ["AHV_BACKEND"]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
solo = [dist.new_group([r]) for r in range(world)][rank]   # every rank creates every group; keeps its own
ahv = importlib.import_module("3dahv_amd")
g = np.load(os.path.join(os.environ["AHV_REPO"], "tests", "golden", "batched.npz"))
h = np.load(os.path.join(os.environ["AHV_REPO"], "tests", "golden", "score_n128.npz"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vs, vt, W1, W2, b2 = T(g["vol_src"]), T(g["vol_tgt"]), T(h["W1"]), T(h["W2"]), T(h["b2"])
R = torch.from_numpy(ahv.rotations.haar_rotations_np(10_000, 40)).to(dev)
force = backend == "nccl"
# no_teams: every score by one wave, i.e. independent of how the sets are split over the ranks -- bit for bit
multi = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, batch=3, use_graph=True, force_collectives=force, no_teams=True)
single = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, batch=3, use_graph=False, group=solo, no_teams=True)
# the default (teams of four waves may take a shard's remainder): same winners, scores equal to rounding
multi_t = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, batch=3, use_graph=True, force_collectives=force)
assert single.world == 1 and not single.collectives
assert multi.collectives and multi.use_graph == (backend == "nccl")
for rep in range(3):
    a, b = (vs, vt) if rep != 1 else (vt, vs)
    got = [t.clone() for t in multi(a, b)]
    ref = [t.clone() for t in single(a, b)]
    for x, y in zip(got, ref):
        assert torch.equal(x, y), (rank, rep, x, y)
    for x, y in zip([t.clone() for t in multi_t(a, b)], ref):
        assert torch.equal(x, y) if x.dtype == torch.int64 else torch.allclose(x, y, rtol=0, atol=1e-6), (rank, rep, x, y)
    flat = torch.cat([t.double().flatten() for t in got]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    if backend == "gloo":
        dist.all_gather(both, flat)
        assert all(torch.equal(both[0], o) for o in both)
torch.cuda.synchronize()
if rank == 0:
    print("OK world=%d backend=%s graph=%s shards=%s" % (world, backend, multi.use_graph, (multi.c_lo, multi.c_hi, multi.f_lo, multi.f_hi)))
dist.destroy_process_group()
'''


def _launch_ranks(tmp_path, world, backend):
    import os
    import socket
    import subprocess
    import sys
    from .conftest import REPO
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "c2f_worker.py"
    script.write_text(RANK_WORKER)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AHV_REPO=REPO, AHV_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0], outs
    return outs[0]


def test_configs4_two_ranks_on_one_gpu_equal_single_rank(tmp_path):
    """world = 2 (gloo rendezvous, both ranks on the one GPU of the test box; HIP kernels): the sharded two-stage
    step with its three exchanges returns, on every rank, exactly what one rank scoring everything returns."""
    out = _launch_ranks(tmp_path, 2, "gloo")
    assert "world=2" in out and "graph=False" in out


def test_configs4_rccl_collectives_captured_in_the_graph(tmp_path):
    """A 1-rank RCCL group with the collectives forced: the whole step -- 2 scorer launches, compose, select AND
    the three all-reduces -- is captured into ONE hipGraph and replayed (SURVEY 8(d) cfg 5); results equal the
    eager single-rank step.  (Two RCCL ranks cannot share one GPU; the N-rank data path is the gloo test above.)"""
    out = _launch_ranks(tmp_path, 1, "nccl")
    assert "graph=True" in out
