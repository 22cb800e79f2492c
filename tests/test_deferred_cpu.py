"""Bookkeeping of `3dahv_amd/deferred.py` on the CPU: the recognised chain (test_co3d.py:137-146) turns into ONE call of the
backend's fused scorer, and everything else done to a deferred tensor gives what the eager op sequence gives.  The arithmetic
behind the three-function backend is the stock-torch restatement of the reference (oracle/torch_ref.py) here; on the GPU box
tests/test_gpu_estimator.py runs the same chain on the HIP kernels against the reference-run digest."""
import importlib

import pytest
import torch

from oracle import torch_ref as tr

D = importlib.import_module("3dahv_amd.deferred")

N = 9


def fused(v, t, R, W1, W2, b2):
    n = R.shape[-3]
    rot = torch.stack([tr.rotate_volume(x[None].expand(n, -1, -1, -1, -1), R if R.dim() == 3 else R[i]) for i, x in enumerate(v)])
    f = tr.forward_3d2d(rot.reshape(-1, 16, 8, 8, 8), W1, W2, b2).reshape(v.shape[0], n, 32, 64)
    return (f * t[:, None]).sum(2).mean(-1)


@pytest.fixture()
def be():
    calls = {"rotate": 0, "head": 0, "fused": 0}

    def rot(v, R):
        calls["rotate"] += 1
        return tr.rotate_volume(v, R)

    def head(x, W1, W2, b2):
        calls["head"] += 1
        return tr.forward_3d2d(x, W1, W2, b2)

    def sc(*a):
        calls["fused"] += 1
        return fused(*a)

    b = D.Backend(rotate_volume=rot, forward_3d2d=head, score_hypotheses=sc, device_type="cpu")
    prev = D.set_backend(b)
    yield calls
    D.set_backend(prev)


@pytest.fixture()
def data():
    g = torch.Generator().manual_seed(4)
    vols = torch.randn(2, 16, 8, 8, 8, generator=g)
    tgt = torch.randn(2, 16, 8, 8, 8, generator=g)
    R = torch.linalg.qr(torch.randn(N, 3, 3, generator=g))[0]
    W1, W2, b2 = torch.randn(32, 384, generator=g) * 0.05, torch.randn(32, 32, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
    return vols, tgt, R, (W1, W2, b2)


def chain(rotate_volume, forward_3d2d, img_feat_src, img_feat_tgt, proposals):
    """test_co3d.py:135-146"""
    B, C, D_, H, W = img_feat_src.shape
    rot = [rotate_volume(f[None].expand(proposals.shape[0], -1, -1, -1, -1), proposals) for f in img_feat_src]
    rot = torch.stack(rot).reshape(-1, C, D_, H, W)
    f_src = forward_3d2d(rot).reshape(B, proposals.shape[0], -1, H * W)
    f_tgt = forward_3d2d(img_feat_tgt)
    pred_sim = (f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)
    best, idx = torch.max(pred_sim, dim=1)
    return pred_sim, best, idx, proposals[idx]


def patched(head):
    def rotate_volume(v, R):
        d = D.defer_rotate_volume(v, R)
        return d if d is not None else tr.rotate_volume(v, R)

    def forward_3d2d(x):
        if isinstance(x, D.DeferredHypotheses) and x.deferred_kind == "rotated":
            return x.with_head(*head)
        return tr.forward_3d2d(x, *head)
    return rotate_volume, forward_3d2d


@pytest.mark.parametrize("B", [1, 2])
def test_the_reference_chain_is_one_fused_call(be, data, B):
    vols, tgt, R, head = data
    before = dict(D.counters)
    rv, f32 = patched(head)
    sim, best, idx, Rp = chain(rv, f32, vols[:B], tgt[:B], R)
    assert type(sim) is torch.Tensor and sim.shape == (B, N)
    assert be == {"rotate": 0, "head": 0, "fused": 1}     # no op-level kernel ran for the hypotheses
    assert D.counters["fused_score_launches"] == before["fused_score_launches"] + 1
    assert D.counters["materialised"] == before["materialised"]
    ref, rbest, ridx, rR = chain(tr.rotate_volume, lambda x: tr.forward_3d2d(x, *head), vols[:B], tgt[:B], R)
    assert torch.allclose(sim, ref, atol=1e-6) and torch.equal(idx, ridx) and torch.equal(Rp, rR)


def test_metadata_never_materialises(be, data):
    vols, _, R, head = data
    d = D.defer_rotate_volume(vols[0][None].expand(N, -1, -1, -1, -1), R)
    assert isinstance(d, torch.Tensor) and d.shape == (N, 16, 8, 8, 8) and d.dtype == torch.float32 and d.device.type == "cpu"
    assert d.dim() == 5 and d.size(0) == N and len(d) == N and d.numel() == N * 8192 and not d.requires_grad and d.ndim == 5
    assert "rotated" in repr(d)
    f = d.with_head(*head)
    assert f.shape == (N, 32, 64) and f.deferred_kind == "features"
    assert f.view(1, N, 32, 64).deferred_kind == "features" and f.reshape((1, N, -1, 64)).shape == (1, N, 32, 64)
    assert be == {"rotate": 0, "head": 0, "fused": 0} and d.deferred_kind == "rotated"


def test_every_other_use_gives_the_eager_result(be, data):
    vols, tgt, R, head = data
    eager_rot = tr.rotate_volume(vols[0][None].expand(N, -1, -1, -1, -1), R)
    eager_f = tr.forward_3d2d(eager_rot, *head)
    f_tgt = tr.forward_3d2d(tgt[:1], *head)
    mk = lambda: D.defer_rotate_volume(vols[0][None].expand(N, -1, -1, -1, -1), R)
    # rotated volumes used as data
    assert torch.equal(mk() + 1.0, eager_rot + 1.0)
    assert torch.equal(mk()[3], eager_rot[3]) and torch.equal(mk().clone(), eager_rot)
    assert torch.equal(torch.cat([mk(), mk()]), torch.cat([eager_rot, eager_rot]))
    assert torch.equal(mk().reshape(N, 16, 512), eager_rot.reshape(N, 16, 512))          # not a regrouping of the leading dims
    assert torch.equal(mk().permute(0, 2, 1, 3, 4), eager_rot.permute(0, 2, 1, 3, 4))
    assert torch.equal(torch.stack([mk()], dim=1), torch.stack([eager_rot], dim=1))
    assert torch.equal(tr.forward_3d2d(mk(), *head), eager_f)                             # an unpatched forward_3d2d
    # features used differently from the score lines
    fd = lambda: mk().with_head(*head).reshape(1, N, 32, 64)
    assert torch.equal(fd() * 2.0, eager_f[None] * 2.0)
    assert torch.equal(fd() * f_tgt[0], eager_f[None] * f_tgt[0])                         # (32,64): not the (B,1,32,64) broadcast
    assert torch.equal((fd() * f_tgt[:, None]).sum(dim=3), (eager_f[None] * f_tgt[:, None]).sum(dim=3))
    assert torch.equal((fd() * f_tgt[:, None]).sum(dim=2, keepdim=True), (eager_f[None] * f_tgt[:, None]).sum(dim=2, keepdim=True))
    assert torch.equal((fd() * f_tgt[:, None]).sum(dim=2).mean(dim=-1, keepdim=True),
                       (eager_f[None] * f_tgt[:, None]).sum(dim=2).mean(dim=-1, keepdim=True))
    assert torch.equal((fd() * f_tgt[:, None]).sum(dim=2).max(dim=-1).values, (eager_f[None] * f_tgt[:, None]).sum(dim=2).max(dim=-1).values)
    assert be["fused"] == 0
    # real * deferred continues the chain
    s = (f_tgt[:, None] * fd()).sum(2).mean(-1)
    assert be["fused"] == 1 and torch.allclose(s, (eager_f[None] * f_tgt[:, None]).sum(2).mean(-1), atol=1e-6)


def test_in_place_writes_stay_visible(be, data):
    vols, _, R, _ = data
    d = D.defer_rotate_volume(vols[0][None].expand(N, -1, -1, -1, -1), R)
    d[0] = 0.0
    d.mul_(2.0)
    eager = tr.rotate_volume(vols[0][None].expand(N, -1, -1, -1, -1), R)
    eager[0] = 0.0
    assert d.deferred_kind is None and torch.equal(d.materialise(), eager * 2.0) and torch.equal(d + 0.0, eager * 2.0)
    assert be["rotate"] == 1


def test_what_is_not_deferred(be, data):
    vols, _, R, _ = data
    exp = vols[0][None].expand(N, -1, -1, -1, -1)
    assert D.defer_rotate_volume(vols[:1].repeat(N, 1, 1, 1, 1), R) is None            # a materialised batch of volumes
    assert D.defer_rotate_volume(exp, R[:3]) is None                                   # (the kernel raises the batch-size error)
    assert D.defer_rotate_volume(exp.double(), R.double()) is None
    assert D.defer_rotate_volume(torch.randn(N, 4, 8, 8, 8), R) is None                # another shape
    v = vols[0].clone().requires_grad_(True)
    assert D.defer_rotate_volume(v[None].expand(N, -1, -1, -1, -1), R) is None         # autograd is recording
    with torch.no_grad():
        assert D.defer_rotate_volume(v[None].expand(N, -1, -1, -1, -1), R) is not None
    assert D.defer_rotate_volume(vols[:1], R[:1]) is not None                          # N = 1


def infonce(rotate_volume, forward_3d2d, img_feat_1, img_feat_2, sampled_R, pos):
    """The tensor expressions of modules/model_co3d.py:49-59 (per-sample lists; `pos`: the positives' mask)."""
    bs, n = sampled_R.shape[:2]
    warp = [rotate_volume(img_feat_1[i:i + 1].expand(n, -1, -1, -1, -1), sampled_R[i]) for i in range(bs)]
    warp = [forward_3d2d(w) for w in warp]
    f2 = forward_3d2d(img_feat_2)
    sim = [(warp[i] * f2[i:i + 1]).sum(dim=1).mean(dim=-1) for i in range(bs)]
    positive = torch.stack([torch.exp(sim[i][pos[i]] / 0.1).sum(dim=0) for i in range(bs)])
    both = torch.exp(torch.stack(sim) / 0.1).sum(dim=-1)
    return -torch.log(positive / both.clamp(min=1e-8)).mean()


def test_the_training_chain_is_one_differentiable_fused_call_per_batch(be, data):
    vols, tgt, R, head = data
    g = torch.Generator().manual_seed(8)
    Rs = torch.stack([R, torch.linalg.qr(torch.randn(N, 3, 3, generator=g))[0]])
    pos = torch.zeros(2, N, dtype=torch.bool)
    pos[:, :2] = True

    def run(deferred):
        leaves = [t.clone().requires_grad_(True) for t in (vols, tgt) + head]
        v, t, W1, W2, b2 = leaves

        def rotate_volume(x, Rm):
            d = D.defer_rotate_volume(x, Rm, allow_grad=True) if deferred else None
            return d if d is not None else tr.rotate_volume(x, Rm)

        def forward_3d2d(x):
            if isinstance(x, D.DeferredHypotheses) and x.deferred_kind == "rotated":
                return x.with_head(W1, W2, b2)
            return tr.forward_3d2d(x, W1, W2, b2)
        loss = infonce(rotate_volume, forward_3d2d, v, t, Rs, pos)
        loss.backward()
        return loss.detach(), [x.grad for x in leaves]
    before = dict(D.counters)
    loss_d, grads_d = run(True)
    assert be["fused"] == 1 and be["rotate"] == 0 and D.counters["materialised"] == before["materialised"]   # ONE launch for both samples
    loss_e, grads_e = run(False)
    assert torch.allclose(loss_d, loss_e, atol=1e-6)
    for a, b in zip(grads_d, grads_e):
        assert a is not None and torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    # an inference call of forward_3d2d (with_head(detach=True)) cuts every edge, whatever the operands' flags say
    v = vols[:1].clone().requires_grad_(True)
    d = D.defer_rotate_volume(v[0][None].expand(N, -1, -1, -1, -1), R, allow_grad=True)
    W1 = head[0].clone().requires_grad_(True)
    f2 = tr.forward_3d2d(tgt[:1], W1, *head[1:])
    s = (d.with_head(W1, *head[1:], detach=True).reshape(1, N, 32, 64) * f2[:, None]).sum(dim=2).mean(dim=-1)
    assert not s.requires_grad
    # ... and a volume that needs a gradient is only deferred on request
    assert D.defer_rotate_volume(v[0][None].expand(N, -1, -1, -1, -1), R) is None


def test_pending_scores_are_grouped_by_batch_and_survive_odd_orders(be, data):
    vols, tgt, R, head = data
    f2 = tr.forward_3d2d(tgt, *head)
    other_head = tuple(h.clone() for h in head)

    def sample(i, hd, n=N):
        d = D.defer_rotate_volume(vols[i][None].expand(n, -1, -1, -1, -1), R[:n]).with_head(*hd)
        return (d * f2[i:i + 1]).sum(dim=1).mean(dim=-1)
    eager = lambda i, hd, n=N: fused(vols[i:i + 1], f2[i:i + 1], R[:n], *hd)[0]
    a, b, c, d_ = sample(0, head), sample(1, head), sample(0, other_head), sample(1, head, n=N - 2)
    assert all(x.deferred_kind == "scores" and x.shape[0] in (N, N - 2) for x in (a, b, c, d_)) and be["fused"] == 0
    assert torch.allclose(b + 0.0, eager(1, head), atol=1e-6)        # first use: a and b (same weights, same N) in one launch
    assert be["fused"] == 1 and a.deferred_kind is None and c.deferred_kind == "scores" and d_.deferred_kind == "scores"
    assert torch.allclose(a[2:5], eager(0, head)[2:5], atol=1e-6) and be["fused"] == 1
    assert torch.allclose(torch.stack([c]), eager(0, other_head)[None], atol=1e-6) and be["fused"] == 2
    del d_                                                           # never used: nothing is launched for it
    e = sample(1, head)
    assert torch.allclose(e.exp(), eager(1, head).exp(), atol=1e-6) and be["fused"] == 3


def test_random_programs_agree_with_the_eager_sequence(be, data):
    """300 random operator sequences on deferred tensors against the same sequences on the eager tensors: whatever mixture
    of chain links and other operators a script applies, shapes agree after every step and values at the end."""
    import random
    vols, tgt, R, head = data
    f_tgt = tr.forward_3d2d(tgt, *head)
    rng = random.Random(7)

    def start(bv):
        d = [D.defer_rotate_volume(vols[i][None].expand(N, -1, -1, -1, -1), R) for i in range(bv)]
        e = [tr.rotate_volume(vols[i][None].expand(N, -1, -1, -1, -1), R) for i in range(bv)]
        return (torch.stack(d), torch.stack(e)) if (bv > 1 or rng.random() < 0.5) else (d[0], e[0])
    head_fn = lambda x: x.with_head(*head) if isinstance(x, D.DeferredHypotheses) and x.deferred_kind == "rotated" and x.dim() == 5 \
        else tr.forward_3d2d(x.reshape(-1, 16, 8, 8, 8) if x.shape[-4:] == (16, 8, 8, 8) else x, *head)
    menu = [
        ("flat5", lambda x, bv: x.reshape(-1, 16, 8, 8, 8)),
        ("lead5", lambda x, bv: x.reshape(bv, N, 16, 8, 8, 8)),
        ("view5", lambda x, bv: x.view(bv * N, 16, 8, 8, 8)),
        ("head", lambda x, bv: head_fn(x)),
        ("feat4", lambda x, bv: x.reshape(bv, N, -1, 64)),
        ("feat3", lambda x, bv: x.reshape(bv * N, 32, 64)),
        ("mul_tgt", lambda x, bv: x * f_tgt[:bv, None]),
        ("rmul_tgt", lambda x, bv: f_tgt[:bv, None] * x),
        ("mul_one", lambda x, bv: x * f_tgt[:1]),
        ("sum2", lambda x, bv: x.sum(dim=2)),
        ("sum1", lambda x, bv: x.sum(dim=1)),
        ("sum_neg2", lambda x, bv: x.sum(-2)),
        ("mean_last", lambda x, bv: x.mean(dim=-1)),
        ("mean_keep", lambda x, bv: x.mean(dim=-1, keepdim=True)),
        ("add", lambda x, bv: x + 0.5),
        ("neg", lambda x, bv: -x),
        ("index", lambda x, bv: x[..., 1:]),
        ("transpose", lambda x, bv: x.transpose(-1, -2)),
        ("contig", lambda x, bv: x.contiguous()),
        ("double", lambda x, bv: x.double().float()),
    ]
    ran = 0
    for _ in range(300):
        bv = rng.choice([1, 1, 2])
        d, e = start(bv)
        trace = []
        chain = ["flat5", "head", "feat4", rng.choice(["mul_tgt", "rmul_tgt"]), rng.choice(["sum2", "sum_neg2"]), "mean_last"]
        prefix = chain[:rng.randint(0, 6)] if rng.random() < 0.5 else []        # half of the programs start along the chain
        steps = [next(m for m in menu if m[0] == n) for n in prefix] + [rng.choice(menu) for _ in range(rng.randint(1, 4))]
        for name, op in steps:
            try:
                e2 = op(e, bv)
            except Exception:
                continue                      # not applicable to the eager tensor at this point: skip the step for both
            d = op(d, bv)
            e = e2
            trace.append(name)
            assert tuple(d.shape) == tuple(e.shape), (trace, tuple(d.shape), tuple(e.shape))
        got = d + 0.0 if isinstance(d, D.DeferredHypotheses) else d
        assert torch.allclose(got, e, rtol=1e-5, atol=1e-6, equal_nan=True), (trace, (got - e).abs().max().item())   # (means of empty slices: NaN in both)
        ran += 1
    assert ran == 300 and be["fused"] > 0 and be["rotate"] > 0        # both the fused launch and the fallbacks were exercised
