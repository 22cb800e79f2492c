"""Backward of the fused scorer (ahv_score_hypotheses_backward_f32) against torch
autograd through the reference's op sequence (oracle/torch_ref.py, stock torch operators) evaluated in fp64.
Tolerance: gradients are sums over up to B*N hypotheses of fp32 terms -> 2e-4 of the largest entry.

ReLU has a kink: a pre-activation within fp32 rounding of zero (about one element in 5e6; a B = 2 x N = 1100 case
has 4.5e6 of them) makes an fp32 and an fp64 evaluation take different sub-gradients, which moves d vol / d W1 by
1e-3..1e-2 -- arithmetic, not implementation.  The comparison is therefore made against
the fp64 gradient for EVERY assignment of the ambiguous sub-gradients: the pre-activations the fp64 reference finds
within KINK_TAU of zero are listed, the reference gradient is evaluated with each of them flipped (the gradient
is affine in those indicator bits), the bits are fitted to the kernel's gradients by least squares and rounded, and
the kernel must match that one reference within the unchanged tolerance.  Cases with no ambiguous element (or that
already agree) reduce to the plain comparison."""
import numpy as np
import pytest
import torch

from .conftest import load_golden

pytestmark = pytest.mark.gpu
GRAD_RTOL = 2e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(ahv):
    ahv._lib.load()
    return ahv.ops


def ref_scores(vs, ft, R, W1, W2, b2):
    """Differentiable reference: per sample rotate -> forward_3d2d -> mean cosine with ft (B,32,64)."""
    from oracle import torch_ref
    B = vs.shape[0]
    out = []
    for b in range(B):
        Rb = R[b] if R.dim() == 4 else R
        n = Rb.shape[0]
        rot = torch_ref.rotate_volume(vs[b][None].expand(n, -1, -1, -1, -1), Rb)
        f = torch_ref.forward_3d2d(rot, W1, W2, b2)
        out.append((f * ft[b][None]).sum(dim=1).mean(dim=-1))
    return torch.stack(out)


KINK_TAU = 2e-6   # |pre-activation| below this (fp64 reference; typical magnitude ~1) = ambiguous sub-gradient
KINK_MAX = 64     # reference backward passes spent on one case


def ref_scores_masked(vs, ft, R, W1, W2, b2, flips=None):
    """ref_scores with the ReLU written as u * mask, mask = (u > 0) XOR flips; returns (scores, list of u)."""
    from oracle import torch_ref
    import torch.nn.functional as F
    out, us = [], []
    for b in range(vs.shape[0]):
        Rb = R[b] if R.dim() == 4 else R
        n = Rb.shape[0]
        vol = torch_ref.rotate_volume(vs[b][None].expand(n, -1, -1, -1, -1), Rb)
        m, c, d, h, w = vol.shape
        slabs = torch.cat([vol.permute(0, 1, 4, 2, 3).reshape(m, c * w, d, h), vol.permute(0, 1, 3, 2, 4).reshape(m, c * h, d, w),
                           vol.reshape(m, c * d, h, w)], dim=1)                       # modules/modules.py:115-118
        u = F.conv2d(slabs, W1.reshape(32, 384, 1, 1))
        mask = (u.detach() > 0)
        if flips is not None:
            mask = mask ^ flips[b]
        v = F.conv2d(u * mask, W2.reshape(32, 32, 1, 1), b2)
        f = F.normalize(v, p=2, dim=1).flatten(2)
        out.append((f * ft[b][None]).sum(dim=1).mean(dim=-1))
        us.append(u.detach())
    return torch.stack(out), us


def kink_aware_errors(got, vs, ft, R, W1, W2, b2, gs):
    """max relative error of every gradient against the best assignment of the ambiguous ReLU sub-gradients."""
    def grads(flips):
        leaves = [x.double().requires_grad_(True) for x in (vs, ft, W1, W2, b2)]
        s, us = ref_scores_masked(leaves[0], leaves[1], R.double(), leaves[2], leaves[3], leaves[4], flips)
        return torch.autograd.grad(s, leaves, grad_outputs=gs.double()), us
    base, us = grads(None)
    amb = [(b, idx) for b, u in enumerate(us) for idx in torch.nonzero(u.abs() < KINK_TAU).tolist()]
    errs0 = [relerr(a, r.reshape(a.shape)) for a, r in zip(got, base)]
    if not amb or max(errs0) < GRAD_RTOL:
        return errs0, len(amb), 0
    assert len(amb) <= KINK_MAX, "too many near-kink pre-activations: %d" % len(amb)
    deltas = []
    for b, idx in amb:  # gradient change when this one sub-gradient flips
        flips = [torch.zeros_like(u, dtype=torch.bool) for u in us]
        flips[b][tuple(idx)] = True
        g, _ = grads(flips)
        deltas.append([x - y for x, y in zip(g, base)])
    # the gradient is affine in the flip bits: least squares for the bits, rounded to {0, 1}
    flat = lambda ts: torch.cat([t.reshape(-1).double() for t in ts])
    A = torch.stack([flat(d) for d in deltas], dim=1)
    rhs = flat([a.double() - r.reshape(a.shape) for a, r in zip(got, base)])
    bits = (torch.linalg.lstsq(A, rhs[:, None]).solution[:, 0] > 0.5)
    ref = [x.clone() for x in base]
    for i in torch.nonzero(bits).flatten().tolist():
        ref = [x + d for x, d in zip(ref, deltas[i])]
    errs = [relerr(a, r.reshape(a.shape)) for a, r in zip(got, ref)]
    return errs, len(amb), int(bits.sum())


def make_case(ahv, dev, B, N, per_sample, seed):
    g = load_golden("score_n128")
    rng = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vs = t((rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32))
    ft = torch.nn.functional.normalize(t(rng.standard_normal((B, 32, 64)).astype(np.float32)), dim=1)
    R = ahv.rotations.haar_rotations_np(N * (B if per_sample else 1), seed + 1)
    R = t(R.reshape(B, N, 3, 3) if per_sample else R)
    gs = t(rng.standard_normal((B, N)).astype(np.float32))
    return vs, ft, R, t(g["W1"]), t(g["W2"]), t(g["b2"]), gs


def relerr(got, ref):
    return ((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,N,per_sample", [(1, 1, False), (1, 37, False), (2, 9, True), (3, 130, True), (2, 1100, False),
                                            (300, 2, True)])   # more samples than CUs: a workgroup walks several
def test_backward_matches_fp64_autograd(ops, ahv, dev, B, N, per_sample):
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, B, N, per_sample, 5 + B + N)
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    errs, n_amb, combo = kink_aware_errors(got, vs, ft, R, W1, W2, b2, gs)
    names = ["vol_src", "feat_tgt", "W1", "W2", "b2"]
    errs = dict(zip(names, errs))
    print(B, N, per_sample, {k: "%.1e" % v for k, v in errs.items()}, "near-kink pre-activations:", n_amb, "flipped:", combo)
    assert all(v < GRAD_RTOL for v in errs.values()), (errs, n_amb)


@pytest.mark.parametrize("B,N,per_sample", [(1, 1, False), (2, 9, True), (3, 130, True), (2, 1100, False), (5, 2048, True),
                                            (300, 3, True), (260, 17, False)])   # more samples than CUs
def test_saved_preactivations_backward_equals_the_recomputing_one(ops, ahv, dev, B, N, per_sample):
    """The training pair (ABI 2.3): ahv_score_hypotheses_train_f32 leaves every hypothesis' pre-activations in the workspace,
    ahv_score_hypotheses_backward_saved_f32 reads them instead of recomputing rotate_volume + the first projection.  Same
    scores as the inference launch bit for bit (the same accumulation chain), and the five gradients of the recomputing
    backward -- which the other tests hold to the fp64 reference -- to rounding (the head's arithmetic is the same code on
    the same values; the sums across hypotheses use float atomics in both)."""
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, B, N, per_sample, 31 + B + N)
    scores, ws = ops.score_hypotheses_train(vs, ft, R, W1, W2, b2)
    ref_scores, _ = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
    assert torch.equal(scores, ref_scores)
    want = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs, workspace=ws)
    for a, b, name in zip(got, want, ("vol_src", "feat_tgt", "W1", "W2", "b2")):
        assert relerr(a, b.double()) < 5e-6, (name, relerr(a, b.double()))   # (measured <= 2.2e-6: the order of the float-atomic sums)


def test_autograd_function_uses_the_saved_forward_and_survives_a_second_backward(ops, ahv, dev):
    """ops.score_hypotheses_autograd / ops.score_hypotheses under autograd run the training forward; the workspace serves ONE
    backward, a second one (retain_graph) falls back on the recomputing kernels -- same gradients either way; the key that
    score_hypotheses returns beside differentiable scores is still torch.max's."""
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 2, 300, True, 91)
    leaves = [t.clone().requires_grad_(True) for t in (vs, ft, W1, W2, b2)]
    s = ops.score_hypotheses_autograd(leaves[0], leaves[1], R, leaves[2], leaves[3], leaves[4])
    g1 = torch.autograd.grad(s, leaves, grad_outputs=gs, retain_graph=True)
    g2 = torch.autograd.grad(s, leaves, grad_outputs=gs)
    want = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    for a, b, c in zip(g1, g2, want):
        assert relerr(a, c.reshape(a.shape).double()) < 2e-6 and relerr(b, c.reshape(b.shape).double()) < 2e-6
    s2, key = ops.score_hypotheses(leaves[0], leaves[1], R, leaves[2], leaves[3], leaves[4])
    assert s2.requires_grad and torch.equal(s2.detach(), s.detach())
    val, idx = ops.unpack_best(key)
    tv, ti = torch.max(s2.detach(), dim=1)
    assert torch.equal(idx, ti) and torch.equal(val, tv)


def test_saved_backward_poisons_a_non_finite_sample_only(ops, ahv, dev):
    """A sample with a NaN voxel goes through the forward's exact path, which keeps no accumulators: its saved pre-activations
    are NaN and so are its gradients; the finite sample beside it is untouched."""
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 2, 200, True, 92)
    clean = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    bad = vs.clone()
    bad[0, 3, 2, 5, 1] = float("nan")
    scores, ws = ops.score_hypotheses_train(bad, ft, R, W1, W2, b2)
    ref_scores, _ = ops.score_hypotheses(bad, ft, R, W1, W2, b2)
    assert torch.equal(torch.isnan(scores), torch.isnan(ref_scores)) and torch.equal(scores[1], ref_scores[1])
    got = ops.score_hypotheses_backward(bad, ft, R, W1, W2, b2, gs, workspace=ws)
    assert torch.isnan(got[0][0]).float().mean().item() > 0.9 and torch.isfinite(got[0][1]).all()
    assert relerr(got[0][1], clean[0][1].double()) < 2e-6


def test_backward_at_integer_sample_coordinates(ops, ahv, dev):
    """identity and the 24 cube rotations put EVERY sample coordinate exactly on an integer -- where the forward's
    point-mirror gather (quarters 3 and 2 reuse the set-up of 0 and 1, csrc/ahv_dual.h hat_mirror) picks the neighbouring
    base row with weights (0, 1) instead of (1, 0), while the dW1 / dV kernels evaluate all four quarters directly.  Same
    samples, so the five gradients must still be those of the fp64 reference; the 45-degree, scaled, non-orthonormal and
    zero matrices of the G3 fixture ride along (R is never assumed to be a rotation)."""
    g3 = load_golden("edge_rotations")
    vs, ft, _, W1, W2, b2, _ = make_case(ahv, dev, 2, 1, False, 77)
    R = torch.from_numpy(np.ascontiguousarray(g3["R"])).to(dev)
    gs = torch.from_numpy(np.random.RandomState(78).standard_normal((2, R.shape[0])).astype(np.float32)).to(dev)
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    errs, n_amb, combo = kink_aware_errors(got, vs, ft, R, W1, W2, b2, gs)
    assert all(v < GRAD_RTOL for v in errs), (errs, n_amb, combo)
    # and the forward at the same rotations, fused against the reference-generated scores of the fixture's own volume
    g1 = load_golden("score_n128")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s, _ = ops.verify_pair(t(g1["vol_src"]), t(g1["vol_tgt"]), R, W1, W2, b2, no_teams=True)
    ref = g3["scores"]
    assert float(np.max(np.abs(s.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1e-2))) < 1e-4


@pytest.mark.parametrize("seed", [1101, 1102, 1103, 1106])
def test_backward_large_case_any_seed(ops, ahv, dev, seed):
    """B = 2 x N = 1100 on seeds whose plain fp64 comparison trips over a ReLU kink: the kink-aware reference must
    explain the kernel's gradients to the same tolerance (no seed picking)."""
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 2, 1100, False, seed)
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    errs, n_amb, combo = kink_aware_errors(got, vs, ft, R, W1, W2, b2, gs)
    print(seed, ["%.1e" % e for e in errs], "near-kink:", n_amb, "flipped:", combo)
    assert max(errs) < GRAD_RTOL, (errs, n_amb)


def test_autograd_function_trains_like_torch(ops, ahv, dev):
    """ops.score_hypotheses_autograd inside an infoNCE-style loss: same loss and gradients as the all-torch graph."""
    vs, ft, R, W1, W2, b2, _ = make_case(ahv, dev, 2, 64, True, 3)
    def loss_of(score_fn, leaves):
        s = score_fn(*leaves)
        logits = s / 0.1
        return -(torch.log_softmax(logits, dim=1)[:, :4].logsumexp(dim=1)).mean()
    a = [x.clone().requires_grad_(True) for x in (vs, ft, W1, W2, b2)]
    la = loss_of(lambda v, f, w1, w2, b: ops.score_hypotheses_autograd(v, f, R, w1, w2, b), a)
    la.backward()
    b_ = [x.double().clone().requires_grad_(True) for x in (vs, ft, W1, W2, b2)]
    lb = loss_of(lambda v, f, w1, w2, b: ref_scores(v, f, R.double(), w1, w2, b), b_)
    lb.backward()
    assert abs(la.item() - lb.item()) < 1e-5
    for x, y in zip(a, b_):
        assert relerr(x.grad, y.grad) < GRAD_RTOL


def test_backward_edge_cases(ops, ahv, dev):
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 2, 16, False, 9)
    # zero upstream gradient -> exact zeros; empty hypothesis set -> zeros of the right shapes
    z = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, torch.zeros_like(gs))
    assert all(float(t.abs().max()) == 0.0 for t in z)
    e = ops.score_hypotheses_backward(vs, ft, R[:0], W1, W2, b2, gs[:, :0])
    assert [tuple(t.shape) for t in e] == [(2, 16, 8, 8, 8), (2, 32, 64), (32, 384), (32, 32), (32,)]
    assert all(float(t.abs().max()) == 0.0 for t in e)
    with pytest.raises(RuntimeError):
        ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs[:, :3])
    # two calls agree to rounding (float atomics: not bitwise)
    a = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    b = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    assert all(relerr(x, y.double()) < 1e-5 for x, y in zip(a, b))


def tiny_cfg(n):
    return {"RUN_NAME": "t", "DATA": {"NUM_ROTA": n, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256, "ACC_THR": 30,
                                       "VIEW_THR": 90},
            "TRAIN": {"MASK": True, "MASK_RATIO": 0.25, "LR": 1e-4}}


def test_infonce_gradients_match_all_torch_graph(ahv, dev):
    """infoNCE_loss under autograd: loss and parameter / volume gradients equal those of the same loss built from
    stock torch operators (the reference's graph, modules/model_co3d.py:41-61)."""
    import math
    torch.manual_seed(3)
    m = ahv.estimator.EstimatorCo3d(tiny_cfg(48)).to(dev).train()
    g = torch.Generator().manual_seed(4)
    vs = (torch.randn(2, 16, 8, 8, 8, generator=g) * 1.1).to(dev).requires_grad_(True)
    vt = (torch.randn(2, 16, 8, 8, 8, generator=g) * 1.1).to(dev).requires_grad_(True)
    gt = ahv.rotations.random_rotations(2, generator=g).to(dev)
    R = torch.cat([gt[:, None], ahv.rotations.random_rotations(2 * 47, generator=g).to(dev).reshape(2, 47, 3, 3)], dim=1)
    loss = m.infoNCE_loss(vs, vt, R, gt)
    params = list(m.feature_aligner.feature_embedding_2d.parameters())
    got = torch.autograd.grad(loss, [vs, vt] + params)

    W1, W2, b2 = (p.detach().double().requires_grad_(True) for p in m.feature_aligner.head_weights())
    vs64, vt64 = vs.detach().double().requires_grad_(True), vt.detach().double().requires_grad_(True)
    from oracle import torch_ref
    ft = torch_ref.forward_3d2d(vt64, W1, W2, b2)
    sim = ref_scores(vs64, ft, R.double(), W1, W2, b2)
    gs = (torch.sum(R.flatten(2) * gt.reshape(-1, 1, 9), dim=-1).clamp(-1, 3) - 1) / 2
    pos = (180 * torch.arccos(gs) / math.pi <= 30)
    e = torch.exp(sim / 0.1)
    ref_loss = (-torch.log((e * pos).sum(dim=-1) / e.sum(dim=-1).clamp(min=1e-8))).mean()
    ref = torch.autograd.grad(ref_loss, [vs64, vt64, W1, W2, b2])
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    for a, b in zip(got, ref):
        assert relerr(a, b.reshape(a.shape)) < GRAD_RTOL


@pytest.mark.parametrize("tag", ["a", "b"])
def test_infonce_matches_the_reference_run_gradients(ahv, dev, tag):
    """G11 `infonce_grad` (tools/gen_golden.py gen_infonce): the loss, similarities and five gradients the
    REFERENCE's own code produced under autograd (utils.rotate_volume utils.py:113-131 + Feature_Aligner.forward_3d2d
    modules/modules.py:112-124 + the loss lines of modules/model_co3d.py:41-61), per-sample hypothesis sets with the
    GT at index 0.  The mirror's infoNCE_loss (one fused HIP launch forward, the three-kernel HIP backward) must
    give the same loss (1e-5), similarities (1e-4 relative, the forward's contract) and gradients (GRAD_RTOL of
    each tensor's largest entry; reference gradients are fp32 here, so a near-kink pre-activation falls back on
    the kink-aware fp64 comparison)."""
    g = load_golden("infonce_grad")
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    B, N = g[tag + "_R"].shape[:2]
    m = ahv.estimator.EstimatorCo3d(tiny_cfg(N)).to(dev).train()
    c1, c2 = m.feature_aligner.feature_embedding_2d[0], m.feature_aligner.feature_embedding_2d[2]
    with torch.no_grad():
        c1.weight.copy_(t("W1").reshape(c1.weight.shape))
        c2.weight.copy_(t("W2").reshape(c2.weight.shape))
        c2.bias.copy_(t("b2"))
    vs, vt = t(tag + "_vol_src").requires_grad_(True), t(tag + "_vol_tgt").requires_grad_(True)
    R, gt = t(tag + "_R"), t(tag + "_gt")
    per_sample = m.infoNCE_loss(vs, vt, R, gt, reduce_mean=False)
    loss = m.infoNCE_loss(vs, vt, R, gt)
    assert abs(loss.item() - float(g[tag + "_loss"])) < 1e-5
    assert np.max(np.abs(per_sample.detach().cpu().numpy() - g[tag + "_loss_per_sample"])) < 1e-5
    sim = ahv.ops.score_hypotheses(vs.detach(), m.feature_aligner.forward_3d2d(vt.detach()).detach(), R,
                                   *[w.detach() for w in m.feature_aligner.head_weights()])[0]
    ref_sim = g[tag + "_sim"]
    assert np.max(np.abs(sim.cpu().numpy() - ref_sim) / np.abs(ref_sim).clip(1e-2)) < 1e-4
    got = torch.autograd.grad(loss, [vs, vt, c1.weight, c2.weight, c2.bias])
    errs = {}
    for a, key in zip(got, ("d_vol_src", "d_vol_tgt", "d_W1", "d_W2", "d_b2")):
        ref = torch.from_numpy(g[tag + "_" + key]).double().to(dev)
        errs[key] = relerr(a.reshape(ref.shape), ref)
    print(tag, {k: "%.1e" % v for k, v in errs.items()})
    assert all(v < GRAD_RTOL for v in errs.values()), errs


def test_training_steps_reduce_the_loss(ahv, dev):
    """training_step (modules/model_co3d.py:71-91) end to end: backbone -> encoder (torch autograd) -> sampled
    rotations with the ground truth as hypothesis 0 -> InfoNCE through the HIP backward -> AdamW."""
    torch.manual_seed(0)
    cfg = tiny_cfg(64)
    cfg["TRAIN"]["LR"] = 3e-4
    m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev).train()
    (opt,), (sched,) = m.configure_optimizers()
    g = torch.Generator().manual_seed(1)
    batch = {"image": torch.randn(2, 2, 3, 256, 256, generator=g).to(dev),
             "relative_rotation": ahv.rotations.random_rotations(2, generator=g).to(dev)[:, None]}
    losses = []
    for step in range(6):
        opt.zero_grad()
        loss = m.training_step(batch, step)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and len(m.logged["train_loss"]) == 6
    # every aligner parameter takes part, except bn_down, which the reference constructs but never applies
    assert all(p.grad is not None for k, p in m.feature_aligner.named_parameters() if "bn_down" not in k)
    assert min(losses[3:]) < losses[0], losses   # same pair every step: the loss must come down


def test_backward_many_samples_and_gradient_scales(ops, ahv, dev):
    """More samples than workgroups in y (each workgroup flushes / re-zeroes its volume-gradient image between
    samples), and upstream gradients far from 1 (the dV image is fixed point with a per-sample scale derived on the
    device from max|du|)."""
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 260, 3, True, 21)
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    leaves = [x.double().requires_grad_(True) for x in (vs, ft, W1, W2, b2)]
    ref = torch.autograd.grad(ref_scores(leaves[0], leaves[1], R.double(), *leaves[2:]), leaves, grad_outputs=gs.double())
    for a, b in zip(got, ref):
        assert relerr(a, b.reshape(a.shape)) < GRAD_RTOL
    # per-sample check of the volume gradient (a global max would hide a sample that was dropped)
    per = ((got[0].double() - ref[0]).flatten(1).abs().max(dim=1).values / ref[0].flatten(1).abs().max(dim=1).values)
    assert per.max().item() < GRAD_RTOL

    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, 2, 50, False, 22)
    base = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    for scale in (1e-9, 1e7):
        mixed = gs * torch.tensor([[scale], [1.0]], device=dev)       # two samples, very different magnitudes
        out = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, mixed)
        assert relerr(out[0][0], base[0][0].double() * scale) < 1e-5
        assert relerr(out[0][1], base[0][1].double()) < 1e-5
    nan = gs.clone()
    nan[0, 3] = float("nan")                                          # poisoned sample -> NaN, the other one intact
    out = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, nan)
    assert torch.isnan(out[0][0]).any() and torch.isfinite(out[0][1]).all()
    assert relerr(out[0][1], base[0][1].double()) < 1e-5


def test_graphed_train_step_matches_eager(ahv, dev):
    """harness.GraphedTrainStep (whole iteration in one hipGraph) against the eager iteration: same rotations
    (same seed and draw counter), masking off -> same losses and same parameters after three steps."""
    import copy
    cfg = tiny_cfg(48)
    cfg["TRAIN"]["MASK"] = False
    torch.manual_seed(0)
    m1 = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev).train()
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(5)
    batches = [{"image": torch.randn(2, 2, 3, 256, 256, generator=g).to(dev),
                "relative_rotation": ahv.rotations.random_rotations(2, generator=g).to(dev)[:, None]} for _ in range(3)]
    state0 = copy.deepcopy(m2.state_dict())
    step = ahv.harness.GraphedTrainStep(m2, batch_size=2, device=dev, warmup=2)
    # the warm-up iterations trained m2 on zeros: rewind weights, optimizer state and the rotation draw counter
    m2.load_state_dict(state0)
    for st in step.optimizer.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    step._draws = 0
    graphed = [float(step(b).item()) for b in batches]

    opt = torch.optim.AdamW([{"params": m1.feature_aligner.parameters(), "lr": 1e-4},
                             {"params": m1.feature_extractor.parameters(), "lr": 1e-4}], eps=1e-5)
    eager = []
    for i, b in enumerate(batches):
        gt = b["relative_rotation"].squeeze(1)
        R = torch.cat([gt[:, None], ahv.ops.random_rotations(2 * 47, seed=torch.initial_seed() + i + 1, device=dev).reshape(2, 47, 3, 3)], dim=1)
        opt.zero_grad()
        vs, vt = m1.feature_aligner.forward_2d3d(m1.feature_extraction(b["image"][:, 0]), m1.feature_extraction(b["image"][:, 1]),
                                                 random_mask=False, mask_ratio=0.0)
        loss = m1.infoNCE_loss(vs, vt, R, gt, reduce_mean=True)
        loss.backward()
        opt.step()
        eager.append(float(loss.item()))
    assert np.allclose(graphed, eager, rtol=2e-4, atol=1e-5), (graphed, eager)
    # Adam's first steps are +-lr * sign-like, so rounding-level gradient differences (other GEMM backend, float
    # atomics) move single weights by a fraction of lr: compare the UPDATE vectors, not the weights elementwise
    up1 = torch.cat([(a - state0[k]).flatten() for k, a in m1.state_dict().items() if a.is_floating_point()])
    up2 = torch.cat([(b - state0[k]).flatten() for k, b in m2.state_dict().items() if b.is_floating_point()])
    cos = torch.nn.functional.cosine_similarity(up1, up2, dim=0).item()
    assert cos > 0.999 and up1.abs().max().item() > 1e-5, cos
    assert (up1 - up2).abs().max().item() < 6e-4


def test_backward_at_the_co3d_training_size(ops, ahv, dev):
    """The reference's CO3D training size (train_estimator_co3d.py:12-15: NUM_ROTA = 9000, BS = 32; per-sample
    rotation sets, modules/model_co3d.py:41-61): 288 000 hypotheses forward + backward in one call.
      * workspace = the size the ABI reports (2.36 GB: 8 KiB of dL/du per hypothesis);
      * a 2 x 1100 sub-problem EMBEDDED in the batch (samples 5 and 17, hypotheses 3000..4099; dL/dscore = 0
        everywhere else) meets the kink-aware fp64 reference, and samples without upstream gradient get exact zeros;
      * with dL/dscore on all 288 000 hypotheses the gradients are finite and ADDITIVE over a split of the hypothesis
        axis (the 64-bit fixed-point image of dV and its per-sample headroom guard at 9000 hypotheses per sample)."""
    B, N = 32, 9000
    lib = ahv._lib.load()
    cu = lib.ahv_device_cu_count()
    # dL/du (8 KiB per hypothesis) + max|du| per sample (padded to 4 words) + one dW1 partial per workgroup
    assert lib.ahv_score_hypotheses_backward_workspace_bytes(B, N) == 4 * (2048 * B * N + ((B + 3) & ~3) + cu * 32 * 384)
    vs, ft, R, W1, W2, b2, gs = make_case(ahv, dev, B, N, True, 77)
    sub_b, lo, hi = [5, 17], 3000, 4100
    gsub = torch.zeros_like(gs)
    for b in sub_b:
        gsub[b, lo:hi] = gs[b, lo:hi]
    got = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gsub)
    others = [b for b in range(B) if b not in sub_b]
    assert got[0][others].abs().max().item() == 0.0 and got[1][others].abs().max().item() == 0.0
    sub = (got[0][sub_b], got[1][sub_b], got[2], got[3], got[4])
    errs, n_amb, combo = kink_aware_errors(sub, vs[sub_b], ft[sub_b], R[sub_b][:, lo:hi].contiguous(), W1, W2, b2,
                                           gs[sub_b][:, lo:hi].contiguous())
    print("embedded 2 x 1100:", ["%.1e" % e for e in errs], "near-kink:", n_amb, "flipped:", combo)
    assert max(errs) < GRAD_RTOL, (errs, n_amb)
    # every hypothesis carries gradient: finite, and the sum of the two half-batches of hypotheses
    full = ops.score_hypotheses_backward(vs, ft, R, W1, W2, b2, gs)
    assert all(torch.isfinite(g).all().item() for g in full)
    half = N // 2
    ga = ops.score_hypotheses_backward(vs, ft, R[:, :half].contiguous(), W1, W2, b2, gs[:, :half].contiguous())
    gb = ops.score_hypotheses_backward(vs, ft, R[:, half:].contiguous(), W1, W2, b2, gs[:, half:].contiguous())
    for name, f, a, b_ in zip(["vol_src", "feat_tgt", "W1", "W2", "b2"], full, ga, gb):
        err = relerr(f, (a.double() + b_.double()))
        assert err < 2e-5, (name, err)     # fp32 accumulation order differs between one pass and two
