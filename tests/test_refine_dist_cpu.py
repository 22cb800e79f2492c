"""BASELINE.json configs[4] (coarse-to-fine), multi-rank control flow on CPU: ``CoarseToFine._step`` with
``world > 1`` under world-size-2 gloo.  The five device ops are injected as an oracle-backed CPU backend (test
infrastructure; the product backend is ``3dahv_amd.ops`` = HIP only), so what is under test is exactly the part
that differs from the single-GPU path: the two sharded stages and the two packed-key all-reduces (the only two
collectives of a step; the winner's row is composed locally on every rank).
Merged keys, R_pred and indices must equal the single-rank result bit for bit."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import REPO
from .test_dist_cpu import _free_port

N_COARSE, N_FINE = 96, 40


class OracleBackend:
    """CPU stand-ins with the signatures of 3dahv_amd.ops, built on oracle/ (checker) + the host key codec."""

    def __init__(self, ahv, oracle):
        self.ahv, self.oracle = ahv, oracle

    def score_hypotheses(self, vol_src, vol_tgt, R, W1, W2, b2, n_offset=0, want_scores=True, best_key=None,
                         reset_best=None):
        s, _, _ = self.oracle.score_hypotheses(vol_src.numpy(), vol_tgt.numpy(), R.numpy(), W1.numpy(), W2.numpy(),
                                               b2.numpy())
        best, idx = self.oracle.argmax(s)
        key = torch.from_numpy(self.ahv.dist.pack_keys_host(best, idx + n_offset))
        if best_key is not None:  # merge into the caller's key (handed over empty), as the product does
            assert reset_best is False
            best_key.copy_(torch.maximum(best_key, key))
            key = best_key
        return (torch.from_numpy(s) if want_scores else None), key

    def verify_pair(self, vol_src, vol_tgt, R, W1, W2, b2, want_feat_tgt=False, **kw):
        # the oracle's scorer takes the target VOLUME (it applies forward_3d2d itself): "features" = the volume
        s, key = self.score_hypotheses(vol_src, vol_tgt, R, W1, W2, b2, **kw)
        return (s, key, vol_tgt) if want_feat_tgt else (s, key)

    def unpack_best(self, key):
        b, i = self.ahv.dist.unpack_keys_host(key.numpy())
        return torch.from_numpy(b), torch.from_numpy(i)

    def compose_rotations(self, key, R, D, n_offset=0, out=None):
        _, idx = self.unpack_best(key)
        return torch.matmul(R[idx - n_offset][:, None], D[None]).contiguous()

    def select_rotation(self, key, R, n_offset=0, reset_key=False):
        score, idx = self.unpack_best(key)
        if reset_key:
            key.fill_(self.ahv.dist.KEY_EMPTY)
        B, N = key.numel(), R.shape[-3]
        out = torch.zeros(B, 3, 3)
        for b in range(B):
            j = int(idx[b]) - n_offset
            if 0 <= j < N:
                out[b] = R[b, j] if R.dim() == 4 else R[j]
        return score, idx, out


def _inputs(ahv):
    g = np.load(os.path.join(REPO, "tests", "golden", "batched.npz"))
    h = np.load(os.path.join(REPO, "tests", "golden", "score_n128.npz"))
    T = torch.from_numpy
    R = T(ahv.rotations.haar_rotations_np(N_COARSE, seed=21))
    return T(g["vol_src"]), T(g["vol_tgt"]), T(h["W1"]), T(h["W2"]), T(h["b2"]), R


def _run(ahv, oracle):
    vs, vt, W1, W2, b2, R = _inputs(ahv)
    c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=N_FINE, max_angle_deg=12.0, batch=3, use_graph=True,
                                  backend=OracleBackend(ahv, oracle), want_scores=True)
    assert not c2f.use_graph  # CPU tensors / gloo: eager
    out = [t.clone().numpy() for t in c2f(vs, vt)]
    return c2f, out


def _worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ahv = importlib.import_module("3dahv_amd")
        from oracle import oracle
        calls = []
        real_all_reduce = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(tuple(t.shape)), real_all_reduce(t, *a, **k))[1]
        try:
            c2f, out = _run(ahv, oracle)
        finally:
            dist.all_reduce = real_all_reduce
        assert c2f.world == world and c2f.collectives
        # exactly TWO exchanges per step, both the 8*B-byte key all-reduce; R_pred needs none
        assert calls == [(3,), (3,)], calls
        q.put((rank, (c2f.c_lo, c2f.c_hi, c2f.f_lo, c2f.f_hi), out,
               c2f.last["coarse_scores"].numpy(), c2f.last["fine_scores"].numpy()))
    finally:
        dist.destroy_process_group()


def test_coarse_to_fine_world2_equals_single_rank(ahv, oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, ref = _run(ahv, oracle)  # no process group here: world = 1
    assert single.world == 1 and not single.collectives
    names = ["fine score", "fine index", "R_pred", "coarse score", "coarse index"]
    for rank, (c_lo, c_hi, f_lo, f_hi), out, s1, s2 in got:
        for name, a, b in zip(names, out, ref):
            assert np.array_equal(a, b), (rank, name)
        # each rank scored exactly its contiguous slices of both stages
        assert (c_lo, c_hi) == ahv.dist.shard_range(N_COARSE, rank, world)
        assert (f_lo, f_hi) == ahv.dist.shard_range(N_FINE, rank, world)
        assert np.array_equal(s1, single.last["coarse_scores"].numpy()[:, c_lo:c_hi])
        assert np.array_equal(s2, single.last["fine_scores"].numpy()[:, f_lo:f_hi])
    # R_pred is a rotation
    Rp = ref[2]
    assert np.allclose(Rp @ Rp.transpose(0, 2, 1), np.eye(3), atol=1e-5)
    assert np.all(ref[0] >= ref[3] - 1e-6)  # D[0] = I: refinement never scores below the coarse winner
