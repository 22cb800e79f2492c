"""N>1 path on CPU: world_size-2 gloo, hypothesis axis sharded, packed-key all-reduce(max).
The scorer is injected (CPU oracle + host key codec) so that only the partition / merge logic --
the part that differs from the single-GPU path -- is under test here."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import REPO, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
        g = np.load(os.path.join(REPO, "tests", "golden", "batched.npz"))
        h = np.load(os.path.join(REPO, "tests", "golden", "score_n128.npz"))

        def cpu_scorer(vol_src, feat_tgt, R, W1, W2, b2, n_offset=0, want_scores=True):
            s, _, _ = oracle.score_hypotheses(vol_src.numpy(), g["vol_tgt"], R.numpy(), W1.numpy(), W2.numpy(),
                                              b2.numpy())
            best, idx = oracle.argmax(s)
            key = torch.from_numpy(ahv.dist.pack_keys_host(best, idx + n_offset))
            return (torch.from_numpy(s) if want_scores else None), key

        T = torch.from_numpy
        for name, full in (("shared", g["R_shared"]), ("ties", np.repeat(g["R_shared"][:1], 64, axis=0))):
            scores, best, idx = ahv.dist.score_hypotheses_sharded(
                T(g["vol_src"]), None, T(full), T(h["W1"]), T(h["W2"]), T(h["b2"]), want_scores=True,
                score_fn=cpu_scorer)
            lo, hi = ahv.dist.shard_range(full.shape[0], rank, world)
            q.put((rank, name, best.numpy(), idx.numpy(), scores.numpy(), lo, hi))
        # r_is_local: every rank only holds its slice
        lo, hi = ahv.dist.shard_range(64, rank, world)
        _, best, idx = ahv.dist.score_hypotheses_sharded(
            T(g["vol_src"]), None, T(g["R_shared"][lo:hi]), T(h["W1"]), T(h["W2"]), T(h["b2"]),
            r_is_local=True, n_total=64, score_fn=cpu_scorer)
        q.put((rank, "local", best.numpy(), idx.numpy(), None, lo, hi))
    finally:
        dist.destroy_process_group()


def test_sharded_argmax_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(3 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("batched")
    by = {(r, name): (b, i, s, lo, hi) for r, name, b, i, s, lo, hi in results}
    for r in range(world):
        b, i, s, lo, hi = by[(r, "shared")]
        assert list(i) == list(g["best_idx_shared"])          # global arg-max identical on every rank
        assert np.allclose(b, g["best_shared"], rtol=1e-5)
        assert np.allclose(s, g["scores_shared"][:, lo:hi], rtol=1e-4, atol=1e-6)  # rank scored its own slice
        b, i, s, lo, hi = by[(r, "ties")]
        assert list(i) == [0, 0, 0]                            # lowest GLOBAL index wins ties across ranks
        b, i, s, lo, hi = by[(r, "local")]
        assert list(i) == list(g["best_idx_shared"])
    assert np.array_equal(by[(0, "shared")][0], by[(1, "shared")][0])


def _grad_worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ahv = importlib.import_module("3dahv_amd")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
        for p in net[3].parameters():      # a module that takes no part (like the reference's bn_down): grad None
            p.requires_grad_(True)
        x = torch.randn(4, 7, generator=torch.Generator().manual_seed(10 + rank))
        net[:3](x).square().sum().backward()
        n_buckets = ahv.dist.all_reduce_gradients(net.parameters(), bucket_bytes=64)   # tiny buckets: several
        q.put((rank, n_buckets, [None if p.grad is None else p.grad.clone().numpy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_gradient_averaging_world2():
    """all_reduce_gradients = mean over ranks of each rank's gradients, bucketed; unused parameters stay None."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: average of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    ref = None
    for rank in range(2):
        net.zero_grad()
        x = torch.randn(4, 7, generator=torch.Generator().manual_seed(10 + rank))
        net[:3](x).square().sum().backward()
        g = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        ref = g if ref is None else [None if a is None else a + b for a, b in zip(ref, g)]
    for rank, n_buckets, grads in got:
        assert n_buckets >= 2
        for a, b in zip(grads, ref):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.allclose(a, (b / 2).numpy(), atol=1e-6)
