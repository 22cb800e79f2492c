"""Boundary hardening: the op-level drop-ins never drop a gradient silently, the packed
encoder weights follow in-place parameter updates, launches are bound to the tensors' device, fit() keeps one
optimizer / scheduler across epochs and resumes, GraphedTrainStep leaves the model as it was given."""
import copy
import types

import numpy as np
import pytest
import torch

from .conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def ref_infonce(rotate_volume, forward_3d2d, img_feat_1, img_feat_2, sampled_R, gt_delta_R, acc_thr):
    """The shape of the reference's loss (modules/model_co3d.py:41-61): positives within acc_thr degrees of the
    ground truth, per-sample rotate_volume of a stride-0 expand, forward_3d2d, mean cosine, temperature 0.1."""
    B, N = sampled_R.shape[:2]
    cos = ((sampled_R.reshape(B, N, 9) * gt_delta_R.reshape(B, 1, 9)).sum(-1).clamp(-1, 3) - 1) / 2
    pos = (torch.arccos(cos) * 180.0 / np.pi) <= acc_thr
    rot = [rotate_volume(img_feat_1[b:b + 1].expand(N, -1, -1, -1, -1), sampled_R[b]) for b in range(B)]
    f1 = forward_3d2d(torch.stack(rot).reshape(-1, 16, 8, 8, 8)).reshape(B, N, 32, 64)
    f2 = forward_3d2d(img_feat_2)
    sim = torch.exp((f1 * f2[:, None]).sum(dim=2).mean(dim=-1) / 0.1)
    return (-torch.log((sim * pos).sum(1) / sim.sum(1))).mean()


def test_patched_reference_shaped_infonce_backpropagates_on_hip(ahv, dev):
    """patch.install() on stand-in `utils` / `modules.modules`, then a reference-shaped infoNCE_loss under autograd:
    non-zero gradients for both volumes and the head weights, equal to fp64 autograd through the stock-operator
    restatement (oracle/torch_ref.py).  Before round 2 the patched callables detached their inputs: every
    gradient was silently zero."""
    from oracle import torch_ref
    g = load_golden("score_n128")
    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None

    class Feature_Aligner(torch.nn.Module):  # noqa: N801
        def __init__(self):
            super().__init__()
            self.feature_embedding_2d = torch.nn.Sequential(torch.nn.Conv2d(384, 32, 1, bias=False), torch.nn.ReLU(),
                                                            torch.nn.Conv2d(32, 32, 1))

        def forward_3d2d(self, x):
            raise AssertionError("not patched")
    mm.Feature_Aligner = Feature_Aligner
    fa = Feature_Aligner().to(dev)
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    with torch.no_grad():
        fa.feature_embedding_2d[0].weight.copy_(T("W1").reshape(32, 384, 1, 1))
        fa.feature_embedding_2d[2].weight.copy_(T("W2").reshape(32, 32, 1, 1))
        fa.feature_embedding_2d[2].bias.copy_(T("b2"))
    rng = np.random.RandomState(3)
    B, N = 2, 24
    v1 = torch.tensor((rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32), device=dev, requires_grad=True)
    v2 = torch.tensor((rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32), device=dev, requires_grad=True)
    gt = torch.from_numpy(ahv.rotations.haar_rotations_np(B, 5)).to(dev)
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(B * N, 6)).reshape(B, N, 3, 3).to(dev).clone()
    R[:, 0] = gt
    ahv.patch.install(um, mm)
    try:
        loss = ref_infonce(um.rotate_volume, fa.forward_3d2d, v1, v2, R, gt, 30.0)
        loss.backward()
    finally:
        ahv.patch.uninstall()
    W1, W2, b2 = fa.feature_embedding_2d[0].weight, fa.feature_embedding_2d[2].weight, fa.feature_embedding_2d[2].bias
    got = [v1.grad, v2.grad, W1.grad.reshape(32, 384), W2.grad.reshape(32, 32), b2.grad]
    # fp64 reference through stock torch operators
    d = lambda t: t.detach().double().cpu().requires_grad_(True)
    r1, r2, rW1, rW2, rb2 = d(v1), d(v2), d(W1.reshape(32, 384)), d(W2.reshape(32, 32)), d(b2)
    ref_loss = ref_infonce(torch_ref.rotate_volume, lambda x: torch_ref.forward_3d2d(x, rW1, rW2, rb2), r1, r2,
                           R.double().cpu(), gt.double().cpu(), 30.0)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * max(1.0, abs(ref_loss.item()))
    for name, a, b in zip(["vol_src", "vol_tgt", "W1", "W2", "b2"], got, [r1.grad, r2.grad, rW1.grad, rW2.grad, rb2.grad]):
        assert a is not None and a.abs().max().item() > 0, name
        assert rel(a.cpu(), b) < 2e-4, (name, rel(a.cpu(), b))


def ref_infonce_verbatim(rotate_volume, forward_3d2d, img_feat_1, img_feat_2, sampled_R, gt_delta_R, acc_thr, num_rota):
    """modules/model_co3d.py:41-61 expression for expression (per-sample lists, `sim[idx][posi_indices[idx]]`)."""
    bs = gt_delta_R.shape[0]
    with torch.no_grad():
        gt_sim = (torch.sum(sampled_R.flatten(2) * gt_delta_R.view(-1, 1, 9), dim=-1).clamp(-1, 3) - 1) / 2
        gt_dis = torch.arccos(gt_sim) / np.pi
        posi_indices = [torch.nonzero(180 * gt_dis[i] <= acc_thr).squeeze(-1) for i in range(bs)]
    img_feat_warp = [rotate_volume(img_feat_1[idx:idx + 1].expand(num_rota, -1, -1, -1, -1), sampled_R[idx]) for idx in range(bs)]
    img_feat_warp = [forward_3d2d(img_feat) for img_feat in img_feat_warp]
    img_feat_2 = forward_3d2d(img_feat_2)
    sim = [(img_feat_warp[idx] * img_feat_2[idx:idx + 1]).sum(dim=1).mean(dim=-1) for idx in range(bs)]
    positive_sim = torch.stack([torch.exp(sim[idx][posi_indices[idx]] / 0.1).sum(dim=0) for idx in range(bs)])
    positive_negative_sim = (torch.exp(torch.stack(sim) / 0.1)).sum(dim=-1)
    return -torch.log(positive_sim / positive_negative_sim.clamp(min=1e-8)).mean()


@pytest.mark.parametrize("defer", [True, False])
def test_unchanged_infonce_lines_in_training_mode(ahv, dev, defer):
    """The reference's infoNCE_loss lines under patch.install() with the module in TRAINING mode: with the deferral (the
    default) all samples' `rotate_volume` .. `.mean(dim=-1)` chains are ONE differentiable fused launch (the training pair: HIP
    forward that keeps its pre-activations + HIP backward; the per-sample score tensors stay pending until the first is
    used) and nothing is materialised; without it every line is its own differentiable op-level kernel.  Either way: the loss and all five gradients of fp64 autograd through the stock-operator
    restatement."""
    from oracle import torch_ref
    g = load_golden("score_n128")
    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None

    class Feature_Aligner(torch.nn.Module):  # noqa: N801
        def __init__(self):
            super().__init__()
            self.feature_embedding_2d = torch.nn.Sequential(torch.nn.Conv2d(384, 32, 1, bias=False), torch.nn.ReLU(),
                                                            torch.nn.Conv2d(32, 32, 1))

        def forward_3d2d(self, x):
            raise AssertionError("not patched")
    mm.Feature_Aligner = Feature_Aligner
    fa = Feature_Aligner().to(dev).train()
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    with torch.no_grad():
        fa.feature_embedding_2d[0].weight.copy_(T("W1").reshape(32, 384, 1, 1))
        fa.feature_embedding_2d[2].weight.copy_(T("W2").reshape(32, 32, 1, 1))
        fa.feature_embedding_2d[2].bias.copy_(T("b2"))
    rng = np.random.RandomState(11)
    B, N = 3, 40
    v1 = torch.tensor((rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32), device=dev, requires_grad=True)
    v2 = torch.tensor((rng.standard_normal((B, 16, 8, 8, 8)) * 1.1).astype(np.float32), device=dev, requires_grad=True)
    gt = torch.from_numpy(ahv.rotations.haar_rotations_np(B, 5)).to(dev)
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(B * N, 6)).reshape(B, N, 3, 3).to(dev).clone()
    R[:, 0] = gt
    ahv.patch.install(um, mm, defer=defer)
    before, dbefore = dict(ahv.patch.calls), dict(ahv.deferred.counters)
    torch.autograd.set_detect_anomaly(True)        # as the reference's training runs (modules/model_co3d.py:22)
    try:
        loss = ref_infonce_verbatim(um.rotate_volume, fa.forward_3d2d, v1, v2, R, gt, 30.0, N)
        loss.backward()
    finally:
        torch.autograd.set_detect_anomaly(False)
        ahv.patch.uninstall()
    ran = {k: ahv.patch.calls[k] - before[k] for k in before}
    dran = {k: ahv.deferred.counters[k] - dbefore[k] for k in dbefore}
    if defer:
        assert dran == {"deferred_rotations": B, "deferred_forward_3d2d": B, "fused_score_launches": 1, "materialised": 0}, dran   # ONE launch pair for the batch
        assert ran["forward_3d2d_autograd"] == 1 and ran["rotate_volume_kernel"] == 0 and ran["forward_3d2d_inference"] == 0, ran
    else:
        assert dran == {k: 0 for k in dran} and ran["forward_3d2d_autograd"] == B + 1 and ran["rotate_volume_kernel"] == B, (ran, dran)
    W1, W2, b2 = fa.feature_embedding_2d[0].weight, fa.feature_embedding_2d[2].weight, fa.feature_embedding_2d[2].bias
    got = [v1.grad, v2.grad, W1.grad.reshape(32, 384), W2.grad.reshape(32, 32), b2.grad]
    d = lambda t: t.detach().double().cpu().requires_grad_(True)
    r1, r2, rW1, rW2, rb2 = d(v1), d(v2), d(W1.reshape(32, 384)), d(W2.reshape(32, 32)), d(b2)
    ref_loss = ref_infonce_verbatim(torch_ref.rotate_volume, lambda x: torch_ref.forward_3d2d(x, rW1, rW2, rb2), r1, r2,
                                    R.double().cpu(), gt.double().cpu(), 30.0, N)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * max(1.0, abs(ref_loss.item()))
    for name, a, b in zip(["vol_src", "vol_tgt", "W1", "W2", "b2"], got, [r1.grad, r2.grad, rW1.grad, rW2.grad, rb2.grad]):
        assert a is not None and a.abs().max().item() > 0, name
        assert rel(a.cpu(), b) < 2e-4, (name, rel(a.cpu(), b))


@pytest.mark.parametrize("shape,shared", [((16, 8, 8, 8), True), ((16, 8, 8, 8), False), ((3, 4, 5, 6), True),
                                          ((3, 4, 5, 6), False)])
def test_rotate_volume_adjoint_matches_grid_sample_autograd(ahv, dev, shape, shared):
    from oracle import torch_ref
    N = 19
    rng = np.random.RandomState(11)
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(N, 3)).to(dev)
    R[3] = R[3] * 1.7   # not a rotation: corners leave the volume (zeros padding), weights no longer sum to 1
    base = torch.tensor(rng.standard_normal(((1 if shared else N),) + shape).astype(np.float32), device=dev,
                        requires_grad=True)
    vol = base.expand(N, -1, -1, -1, -1) if shared else base
    gout = torch.tensor(rng.standard_normal((N,) + shape).astype(np.float32), device=dev)
    out = ahv.ops.rotate_volume(vol, R)
    assert out.requires_grad
    out.backward(gout)
    b64 = base.detach().double().cpu().requires_grad_(True)
    v64 = b64.expand(N, -1, -1, -1, -1) if shared else b64
    torch_ref.rotate_volume(v64, R.double().cpu()).backward(gout.double().cpu())
    assert base.grad.shape == base.shape and rel(base.grad.cpu(), b64.grad) < 1e-5
    with pytest.raises(NotImplementedError):
        ahv.ops.rotate_volume(vol, R.clone().requires_grad_(True))
    with torch.no_grad():  # inference form unchanged: no graph
        assert not ahv.ops.rotate_volume(vol, R).requires_grad


def test_plain_ops_never_drop_a_gradient(ahv, dev):
    g = load_golden("score_n128")
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    vs, vt, R = T("vol_src").requires_grad_(True), T("vol_tgt"), T("R")[:16]
    W1, W2, b2 = T("W1").requires_grad_(True), T("W2"), T("b2")
    ft = ahv.ops.forward_3d2d(vt, W1, W2, b2)          # W1 requires grad -> autograd edge
    assert ft.requires_grad
    scores, key = ahv.ops.score_hypotheses(vs, ft, R, W1, W2, b2)
    assert scores.requires_grad and not key.requires_grad
    scores.sum().backward()
    ref = ahv.ops.score_hypotheses_autograd(vs.detach().requires_grad_(True), ft.detach(), R, W1.detach(), W2, b2)
    assert vs.grad.abs().max().item() > 0 and W1.grad.abs().max().item() > 0
    assert torch.equal(scores.detach(), ref.detach())
    with pytest.raises(RuntimeError, match="autograd"):
        ahv.ops.score_features(torch.zeros(1, 2, 32, 64, device=dev, requires_grad=True), torch.zeros(1, 32, 64, device=dev))
    with pytest.raises(RuntimeError, match="GPU only"):
        ahv.ops.unpack_best(torch.zeros(1, dtype=torch.int64))
    # nothing differentiable requested: the plain launch (teams or single waves: same bits)
    _, key2 = ahv.ops.score_hypotheses(vs, ft, R, W1, W2, b2, want_scores=False)
    assert torch.equal(key2, key)


def test_one_launch_steps_refuse_or_route_a_gradient(ahv, dev):
    """verify_pair / coarse_to_fine have no autograd edge and must SAY so (the refusal sat inside a
    no_grad decorator, where it could never fire); Feature_Aligner.score_hypotheses -- which routes to verify_pair for
    inference -- takes the differentiable pair of ops when something requires grad."""
    g = load_golden("score_n128")
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    vs, vt, R = T("vol_src"), T("vol_tgt"), T("R")[:16]
    W1, W2, b2 = T("W1"), T("W2"), T("b2")
    for bad in ("vs", "W1"):
        a = vs.clone().requires_grad_(bad == "vs")
        w = W1.clone().requires_grad_(bad == "W1")
        with pytest.raises(RuntimeError, match="autograd"):
            ahv.ops.verify_pair(a, vt, R, w, W2, b2)
        with pytest.raises(RuntimeError, match="autograd"):
            ahv.ops.coarse_to_fine(a, vt, R, R[:4], w, W2, b2)
        with torch.no_grad():                       # explicitly not recording: the plain launch
            s, _ = ahv.ops.verify_pair(a, vt, R, w, W2, b2)
        assert not s.requires_grad
    # the module-level entry point: trainable head, volumes under grad -> scores with the HIP backward behind them
    torch.manual_seed(0)
    m = ahv.aligner.Feature_Aligner(768, 256, 32, 4, 4).to(dev)
    with torch.no_grad():
        m.feature_embedding_2d[0].weight.copy_(W1.reshape(32, 384, 1, 1))
        m.feature_embedding_2d[2].weight.copy_(W2.reshape(32, 32, 1, 1))
        m.feature_embedding_2d[2].bias.copy_(b2)
    a = vs.clone().requires_grad_(True)
    scores, key = m.score_hypotheses(a, vt, R)
    assert scores.requires_grad and not key.requires_grad
    scores.sum().backward()
    assert a.grad.abs().max().item() > 0 and m.feature_embedding_2d[0].weight.grad.abs().max().item() > 0
    with torch.no_grad():
        s_inf, k_inf = m.score_hypotheses(vs, vt, R)
    assert (s_inf - scores.detach()).abs().max().item() < 1e-6 and torch.equal(
        ahv.ops.unpack_best(k_inf)[1], ahv.ops.unpack_best(key)[1])


def test_packed_encoder_weights_follow_inplace_updates(ahv, dev):
    """pack -> optimizer.step() -> the no_grad HIP encoder must see the NEW weights everywhere
    (q|k|v concatenation, tap-major conv copies, padded 3-D conv copies are copies, not aliases)."""
    torch.manual_seed(3)
    m = ahv.aligner.Feature_Aligner(768, 256, 32, 4, 4).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(1, 768, 8, 8, generator=g).to(dev), torch.randn(1, 768, 8, 8, generator=g).to(dev)

    def both():
        with torch.no_grad():
            m.use_hip_encoder = m.att.use_hip = True
            hip = m.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
            m.use_hip_encoder = m.att.use_hip = False
            ref = m.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
            m.use_hip_encoder = m.att.use_hip = True
        return hip, ref
    hip0, ref0 = both()
    assert rel(hip0[0], ref0[0]) < 2e-5
    run = m.graphed_forward_2d3d(batch=1)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    gg = torch.Generator(device="cpu").manual_seed(2)
    for p in m.parameters():
        p.grad = (torch.randn(p.shape, generator=gg) * p.detach().abs().mean().cpu()).to(dev)
    opt.step()                                          # in-place update of every parameter
    hip1, ref1 = both()
    assert rel(ref1[0], ref0[0]) > 1e-3                 # the step really changed the function
    assert rel(hip1[0], ref1[0]) < 2e-5 and rel(hip1[1], ref1[1]) < 2e-5
    out = run(a, b)                                     # captured graph: packed copies refilled at their addresses
    assert rel(out[0], ref1[0]) < 2e-5 and rel(out[1], ref1[1]) < 2e-5
    with torch.no_grad():                               # token stage alone (its own table)
        x = torch.randn(1, 256, 8, 8, device=dev)
        y_hip, _ = m.att(x, x)
        m.att.use_hip = False
        y_ref, _ = m.att(x, x)
        m.att.use_hip = True
    assert rel(y_hip, y_ref) < 2e-5
    # storage replaced (not in place): new pointers, the graph is captured again
    with torch.no_grad():
        m.att.proj_in.bias.data = m.att.proj_in.bias.data.clone() + 0.5
    _, ref2 = both()
    out2 = run(a, b)
    assert rel(out2[0], ref2[0]) < 2e-5 and rel(ref2[0], ref1[0]) > 1e-4


def tiny_cfg(num_rota=32):
    return {"RUN_NAME": "t", "DATA": {"NUM_ROTA": num_rota, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256, "ACC_THR": 30,
                                      "VIEW_THR": 90},
            "TRAIN": {"MASK": False, "MASK_RATIO": 0.0, "LR": 1e-4, "STEP_SIZE": 1, "GAMMA": 0.5}}


def test_fit_keeps_one_optimizer_and_scheduler_across_epochs_and_resumes(ahv, dev, tmp_path):
    """the reference trains many epochs with ONE AdamW and ONE StepLR (modules/model_co3d.py:95-99,
    130-145); fit(epochs=k) must do the same, and fit(ckpt_path=) must continue moments, step counts and the LR
    schedule."""
    cfg = tiny_cfg()
    torch.manual_seed(0)
    m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev)
    loader = ahv.harness.SyntheticTrainingPairs(batch_size=1, steps=2, seed=3)

    def opt_sched(model):  # the reference's AdamW; a fast StepLR so that three epochs show the decay
        (o,), _ = model.configure_optimizers()
        return o, torch.optim.lr_scheduler.StepLR(o, step_size=1, gamma=0.5)
    opt0, sch0 = opt_sched(m)
    lr0 = opt0.param_groups[0]["lr"]
    ck = str(tmp_path / "last.ckpt")
    res = ahv.harness.fit(cfg, m, loader, device=dev, epochs=3, save_path=ck, optimizer=opt0, scheduler=sch0)
    assert len(res) == 6 and res.epoch == 3 and res.global_step == 6 and res.optimizer is opt0
    steps = {int(st["step"]) for st in res.optimizer.state.values() if "step" in st}
    assert steps == {6}                                              # one optimizer saw all six steps
    assert res.scheduler.last_epoch == 3
    assert abs(res.optimizer.param_groups[0]["lr"] - lr0 * 0.5 ** 3) < 1e-12      # the decay took effect
    # default path: configure_optimizers() once per fit call, StepLR(200, 0.1) as in modules/model_co3d.py:93-99
    torch.manual_seed(0)
    m1 = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev)
    r1 = ahv.harness.fit(cfg, m1, loader, device=dev, epochs=2, max_steps=3)
    assert len(r1) == 3 and r1.scheduler.step_size == 200 and r1.scheduler.last_epoch == 2
    # resume into a fresh model + fresh optimizer: everything comes from the checkpoint
    torch.manual_seed(1)
    m2 = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=9)).to(dev)
    opt2, sch2 = opt_sched(m2)
    r0 = ahv.harness.fit(cfg, m2, loader, device=dev, epochs=0, ckpt_path=ck, optimizer=opt2, scheduler=sch2)
    assert len(r0) == 0 and r0.epoch == 3 and r0.global_step == 6 and sch2.last_epoch == 3
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    for sa, sb in zip(res.optimizer.state_dict()["state"].values(), opt2.state_dict()["state"].values()):
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert int(sa["step"]) == int(sb["step"]) == 6
    assert abs(opt2.param_groups[0]["lr"] - lr0 * 0.5 ** 3) < 1e-12
    res2 = ahv.harness.fit(cfg, m2, loader, device=dev, epochs=1, optimizer=opt2, scheduler=sch2)
    assert res2.epoch == 1 and sch2.last_epoch == 4    # this call's own epoch count; the scheduler carries the run's
    assert {int(st["step"]) for st in opt2.state.values() if "step" in st} == {8}
    assert abs(opt2.param_groups[0]["lr"] - lr0 * 0.5 ** 4) < 1e-12
    # a missing checkpoint file is skipped, as the reference's os.path.exists guard does (modules/model_co3d.py:130-137)
    r3 = ahv.harness.fit(cfg, m2, loader, device=dev, epochs=0, ckpt_path=str(tmp_path / "absent.ckpt"),
                         optimizer=opt2, scheduler=sch2)
    assert r3.epoch == 0


def test_graphed_train_step_leaves_the_model_as_given(ahv, dev):
    """the capture warm-up must not train the model on placeholder data."""
    cfg = tiny_cfg(24)
    torch.manual_seed(0)
    m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev).train()
    state0 = copy.deepcopy(m.state_dict())
    step = ahv.harness.GraphedTrainStep(m, batch_size=1, device=dev, warmup=2)
    for k, v in m.state_dict().items():
        assert torch.equal(v, state0[k]), k
    for st in step.optimizer.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                assert float(v.abs().max()) == 0.0
    assert step._draws == 0
