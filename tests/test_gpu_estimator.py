"""Estimator / harness call surface on the GPU: the verify step inside them is the fused HIP launch;
checked against the reference's op sequence issued with stock torch operators (oracle/torch_ref.py)
on the same device and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(n=3000, bg=True):
    return {"RUN_NAME": "Synthetic_3DAHV", "DATA": {"NUM_ROTA": n, "BG": bg, "SIZE_THR": 25, "OBJ_SIZE": 256}}


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model_obj(ahv, dev):
    torch.manual_seed(0)
    m = ahv.estimator.EstimatorObjaverse(cfg(), feature_extractor=ahv.estimator.PatchifyBackbone()).to(dev).eval()
    return m


def batch(dev, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    import importlib
    rot = importlib.import_module("3dahv_amd.rotations")
    return {"src_img": torch.rand(B, 3, 256, 256, generator=g).to(dev), "ref_img": torch.rand(B, 3, 256, 256, generator=g).to(dev),
            "src_mask": torch.ones(B, 1, 256, 256, device=dev), "ref_mask": torch.ones(B, 1, 256, 256, device=dev),
            "src_R": rot.random_rotations(B, generator=g).to(dev), "ref_R": rot.random_rotations(B, generator=g).to(dev)}


def test_test_step_matches_torch_op_sequence(ahv, model_obj, dev):
    from oracle import torch_ref
    b = batch(dev, 3)
    P = torch.from_numpy(ahv.rotations.haar_rotations_np(3000, 41)).to(dev)
    geo = model_obj.test_step(b, 0, proposals=P)
    assert geo.shape == (3,) and len(model_obj.pred_Rs) == 1 and model_obj.pred_Rs[0].shape == (27,)
    with torch.no_grad():
        vs, vt = model_obj(b["src_img"], b["src_mask"], b["ref_img"], b["ref_mask"])
        W1, W2, b2 = model_obj.feature_aligner.head_weights()
        s_ref, best_ref, idx_ref = torch_ref.score_hypotheses(vs, vt, P, W1, W2, b2, chunk=1000)
        s_hip, best, idx, R_pred = model_obj.verify(vs, vt, P, want_scores=True)
    rel = ((s_hip - s_ref).abs() / s_ref.abs().clamp_min(1e-2)).max().item()
    assert rel < 1e-4, rel
    assert torch.equal(idx, idx_ref)
    gt = torch.bmm(b["ref_R"], torch.inverse(b["src_R"]))
    assert torch.allclose(geo, ahv.rotations.geodesic_deg(P[idx_ref], gt), atol=1e-3)
    assert "test_error" in model_obj.logged


def test_validation_step_gt_sim(ahv, model_obj, dev):
    from oracle import torch_ref
    b = batch(dev, 2, seed=3)
    P = torch.from_numpy(ahv.rotations.haar_rotations_np(500, 42)).to(dev)
    out = model_obj.validation_step(b, 0, proposals=P)
    with torch.no_grad():
        vs, vt = model_obj(b["src_img"], b["src_mask"], b["ref_img"], b["ref_mask"])
        W1, W2, b2 = model_obj.feature_aligner.head_weights()
        gt = torch.bmm(b["ref_R"], torch.inverse(b["src_R"]))
        ref = torch.stack([torch_ref.score_hypotheses(vs[i:i + 1], vt[i:i + 1], gt[i:i + 1], W1, W2, b2)[0][0, 0]
                           for i in range(2)])
    assert torch.allclose(out["gt_sim"], ref, rtol=1e-4, atol=1e-6)
    a15, a30 = model_obj.on_validation_epoch_end()
    assert 0 <= a15 <= a30 <= 100 and model_obj.step_outputs == []


def test_batch32_shared_proposals_config4_shape(ahv, dev):
    """BASELINE.json configs[3] shape: B=32 pairs, shared proposals (one rank's shard here)."""
    from oracle import torch_ref
    torch.manual_seed(2)
    m = ahv.estimator.EstimatorCo3d(cfg()).to(dev).eval()
    g = torch.Generator().manual_seed(7)
    l4 = torch.randn(2, 32, 768, 8, 8, generator=g).to(dev)
    P = torch.from_numpy(ahv.rotations.haar_rotations_np(2000, 43)).to(dev)
    with torch.no_grad():
        vs, vt = m.forward_features(l4[0], l4[1])
        s, best, idx, R_pred = m.verify(vs, vt, P, want_scores=True)
        W1, W2, b2 = m.feature_aligner.head_weights()
        s_ref, _, idx_ref = torch_ref.score_hypotheses(vs, vt, P, W1, W2, b2, chunk=500)
    assert s.shape == (32, 2000) and R_pred.shape == (32, 3, 3)
    assert ((s - s_ref).abs() / s_ref.abs().clamp_min(1e-2)).max().item() < 1e-4
    assert torch.equal(idx, idx_ref)


def test_harness_gpu_equals_oracle_verify(ahv, oracle, dev):
    torch.manual_seed(3)
    c = cfg(256)
    m = ahv.estimator.EstimatorCo3d(c).to(dev).eval()
    W1, W2, b2 = (t.detach().cpu().numpy() for t in m.feature_aligner.head_weights())

    def oracle_verify(vs, vt, P):
        s, best, idx = oracle.score_hypotheses(vs.cpu().numpy(), vt.cpu().numpy(), P.cpu().numpy(), W1, W2, b2)
        return torch.from_numpy(best).to(dev), torch.from_numpy(idx).to(dev)

    seqs = ahv.harness.SyntheticSequences(2, 3, seed=4)
    np.random.seed(0)
    e_hip, d_hip = ahv.harness.evaluate_category(c, m, seqs, device=dev, return_details=True,
                                                 proposals=torch.from_numpy(ahv.rotations.haar_rotations_np(256, 5)))
    np.random.seed(0)
    e_ref, d_ref = ahv.harness.evaluate_category(c, m, seqs, device=dev, return_details=True, verify_fn=oracle_verify,
                                                 proposals=torch.from_numpy(ahv.rotations.haar_rotations_np(256, 5)))
    assert [d["idx"] for d in d_hip] == [d["idx"] for d in d_ref]
    assert np.allclose(e_hip, e_ref, atol=1e-4)
    # the default on the GPU batches the ordered pairs of a sequence; one by one (as the reference) gives the same
    np.random.seed(0)
    e_one, d_one = ahv.harness.evaluate_category(c, m, seqs, device=dev, return_details=True, batch_pairs=False,
                                                 proposals=torch.from_numpy(ahv.rotations.haar_rotations_np(256, 5)))
    assert [(d["pair"], d["idx"]) for d in d_one] == [(d["pair"], d["idx"]) for d in d_hip]
    assert np.allclose(e_one, e_hip, atol=1e-5) and np.allclose([d["best"] for d in d_one], [d["best"] for d in d_hip], atol=1e-6)
    # several sequences per batch: same pairs in the same order, same answers
    np.random.seed(0)
    e_grp, d_grp = ahv.harness.evaluate_category(c, m, seqs, device=dev, return_details=True, batch_sequences=2,
                                                 proposals=torch.from_numpy(ahv.rotations.haar_rotations_np(256, 5)))
    assert [(d["model_id"], d["pair"], d["idx"]) for d in d_grp] == [(d["model_id"], d["pair"], d["idx"]) for d in d_hip]
    assert np.allclose(e_grp, e_hip, atol=1e-5)


def test_patched_reference_callables_run_on_hip(ahv, dev, g128):
    """patch.install() on stand-in modules: the reference's own call sequence then runs on the HIP ops."""
    import types
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
    ahv.patch.install(um, mm)
    try:
        fa = Feature_Aligner().to(dev)
        T = lambda k: torch.from_numpy(np.ascontiguousarray(g128[k])).to(dev)
        with torch.no_grad():
            fa.feature_embedding_2d[0].weight.copy_(T("W1").reshape(32, 384, 1, 1))
            fa.feature_embedding_2d[2].weight.copy_(T("W2").reshape(32, 32, 1, 1))
            fa.feature_embedding_2d[2].bias.copy_(T("b2"))
        proposals, img_feat_src, img_feat_tgt = T("R"), T("vol_src"), T("vol_tgt")
        # verbatim shape of test_co3d.py:135-146
        B, C, D, H, W = img_feat_src.shape
        warped = [um.rotate_volume(f[None].expand(proposals.shape[0], -1, -1, -1, -1), proposals) for f in img_feat_src]
        warped = torch.stack(warped).reshape(-1, C, D, H, W)
        f_src = fa.forward_3d2d(warped).reshape(B, proposals.shape[0], -1, H * W)
        f_tgt = fa.forward_3d2d(img_feat_tgt)
        pred_sim = (f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)
        pred_sim, pred_index = torch.max(pred_sim, dim=1)
    finally:
        ahv.patch.uninstall()
    assert pred_index.item() == int(g128["best_idx"][0])
    assert abs(pred_sim.item() - float(g128["best"][0])) < 1e-5


@pytest.mark.parametrize("defer", [True, False])
def test_option_a_under_the_reference_scripts_own_conditions(ahv, dev, defer):
    """INTEGRATION.md option A exactly as /root/reference/test_co3d.py would exercise it: grad mode ON (the script never
    enters no_grad), anomaly detection ON (modules/model_co3d.py:22), ``model.eval()`` (test_co3d.py:219), a backbone
    whose parameters require grad (so ``layer_4`` does too), N = 50 000.  The verbatim sequence test_co3d.py:133-146
    runs on a reference-shaped stand-in: an Estimator-like module whose Feature_Aligner's OWN forward_2d3d is the
    stock-torch implementation and whose forward_3d2d raises unless patched.  Seed 0 builds the same weights and the
    same layer_4 pair as tools/gen_golden.py's reference aligner (identical registration order; checked bit for bit on
    the CPU), so the result is held to the reference-run digest G2 (50 000 Haar rotations, seed 3).
    Asserts: (i) G2's arg-max, best score and top-16 scores; (ii) ahv_forward_2d3d_f32 was the encoder that ran and
    every head call took the inference path; (iii) peak memory stays below the reference's own dataflow.
    ``defer`` (the default of patch.install, round 6): the script's score lines run as ONE fused launch -- no rotated
    volume and no hypothesis feature is ever materialised -- else every line is its own op-level kernel."""
    import hashlib
    import types
    from .conftest import load_golden
    dg = load_golden("score_n50k_digest")

    class RefAligner(ahv.aligner.Feature_Aligner):     # torch-operator forward_2d3d, like the reference's class
        def forward_2d3d(self, a, b, random_mask=True, mask_ratio=0.25):
            self.use_hip_encoder = self.att.use_hip = False
            return super().forward_2d3d(a, b, random_mask, mask_ratio)

        def forward_3d2d(self, x):
            raise AssertionError("not patched")

    class Backbone(torch.nn.Module):                    # stands in for MiDaS: image id -> layer_4, parameters require grad
        def __init__(self, layer4):
            super().__init__()
            self.register_buffer("layer4", layer4)
            self.gain = torch.nn.Parameter(torch.ones(()))

        def forward(self, img):
            return self.layer4[img.flatten()[0].long()][None] * self.gain

    class Estimator(torch.nn.Module):                   # modules/model_co3d.py:26-39,63-69
        def __init__(self, layer4):
            super().__init__()
            self.feature_extractor = Backbone(layer4)
            self.feature_aligner = RefAligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4)

        def feature_extraction(self, img):
            return self.feature_extractor(img)

        def forward(self, img_src, img_tgt):
            img_feat_src = self.feature_extraction(img_src)
            img_feat_tgt = self.feature_extraction(img_tgt)
            return self.feature_aligner.forward_2d3d(img_feat_src, img_feat_tgt, random_mask=False, mask_ratio=0)

    torch.manual_seed(0)                                # tools/gen_golden.py seeded_pair: aligner first, then layer_4
    fa = RefAligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4)
    layer4 = torch.randn(2, 3, 768, 8, 8)
    model = Estimator(torch.stack([layer4[0, 0], layer4[1, 0]]))
    model.feature_aligner.load_state_dict(fa.state_dict())
    model = model.to(dev)
    R_np = ahv.rotations.haar_rotations_np(50000, seed=int(dg["seed"]))
    assert hashlib.sha256(R_np.tobytes()).hexdigest() == str(dg["R_sha256"])
    proposals = torch.from_numpy(R_np).to(dev)
    image1, image2 = torch.zeros(3, 4, 4, device=dev), torch.ones(3, 4, 4, device=dev)

    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None
    mm.Feature_Aligner = RefAligner
    assert torch.is_grad_enabled()
    torch.autograd.set_detect_anomaly(True)
    ahv.patch.install(um, mm, defer=defer)
    before = dict(ahv.patch.calls)
    dbefore = dict(ahv.deferred.counters)
    try:
        model.eval()
        rotate_volume = um.rotate_volume
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        # ---- test_co3d.py:133-146, verbatim
        img_feat_src, img_feat_tgt = model(image1[None], image2[None])

        B, C, D, H, W = img_feat_src.shape

        img_feat_src_2_tgt = [rotate_volume(img_feat[None].expand(proposals.shape[0], -1, -1, -1, -1), proposals) for img_feat in img_feat_src]
        img_feat_src_2_tgt = torch.stack(img_feat_src_2_tgt).reshape(-1, C, D, H, W)

        img_feat_src_2_tgt = model.feature_aligner.forward_3d2d(img_feat_src_2_tgt).reshape(B, proposals.shape[0], -1, H*W)
        img_feat_tgt_vol = img_feat_tgt
        img_feat_tgt = model.feature_aligner.forward_3d2d(img_feat_tgt)

        pred_sim = (img_feat_src_2_tgt * img_feat_tgt[:, None]).sum(dim=2).mean(dim=-1)
        all_sim = pred_sim

        pred_sim, pred_index = torch.max(pred_sim, dim=1)
        pred_src_2_tgt_R = proposals[pred_index]
        # ----
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated(dev) - base
        # option B under the same conditions: the method install() adds to the reference's class (one fused launch + one select);
        # the bare op refuses weights that require grad while autograd records -- nothing is detached silently
        with pytest.raises(RuntimeError, match="no autograd edge"):
            fe = model.feature_aligner.feature_embedding_2d
            ahv.ops.verify_pair(img_feat_src, img_feat_tgt_vol, proposals, fe[0].weight, fe[2].weight, fe[2].bias, want_scores=False)
        _, key = model.feature_aligner.verify_hypotheses(img_feat_src, img_feat_tgt_vol, proposals)
        b_sim, b_index, b_R = ahv.ops.select_rotation(key, proposals)
    finally:
        ahv.patch.uninstall()
        torch.autograd.set_detect_anomaly(False)
    ran = {k: ahv.patch.calls[k] - before[k] for k in before}
    dran = {k: ahv.deferred.counters[k] - dbefore[k] for k in dbefore}
    # (ii) which code ran
    if defer:
        assert ran == {"forward_2d3d_hip": 1, "forward_2d3d_reference": 0, "forward_3d2d_inference": 1, "forward_3d2d_autograd": 0,
                       "forward_3d2d_deferred": 1, "rotate_volume_deferred": 1, "rotate_volume_kernel": 0}, ran
        assert dran == {"deferred_rotations": 1, "deferred_forward_3d2d": 1, "fused_score_launches": 1, "materialised": 0}, dran
    else:
        assert ran == {"forward_2d3d_hip": 1, "forward_2d3d_reference": 0, "forward_3d2d_inference": 2, "forward_3d2d_autograd": 0,
                       "forward_3d2d_deferred": 0, "rotate_volume_deferred": 0, "rotate_volume_kernel": 1}, ran
        assert dran == {k: 0 for k in dran}, dran
    assert type(all_sim) is torch.Tensor and not img_feat_src.requires_grad and not all_sim.requires_grad
    # (i) the reference-run digest
    assert pred_index.item() == int(dg["best_idx"][0])
    assert abs(pred_sim.item() - float(dg["best"][0])) < 1e-4 * abs(float(dg["best"][0]))
    top = all_sim[0, torch.from_numpy(dg["top16_idx"]).to(dev)].cpu().numpy()
    assert np.max(np.abs(top - dg["top16_score"]) / np.abs(dg["top16_score"])) < 1e-4
    every = all_sim[0, ::97].cpu().numpy()
    assert np.max(np.abs(every - dg["every97_score"]) / np.abs(dg["every97_score"]).clip(1e-2)) < 1e-4
    assert torch.equal(pred_src_2_tgt_R[0], proposals[int(dg["best_idx"][0])])
    assert b_index.item() == pred_index.item() and torch.equal(b_R, pred_src_2_tgt_R) and abs(b_sim.item() - pred_sim.item()) < 1e-6
    # (iii) memory: the script's own tensors are the rotated volumes twice (list + torch.stack's copy, 1.64 GB each),
    # the features (0.41 GB) and their product (0.41 GB) = 4.1 GB.  The reference's dataflow needs those AND the
    # sampling grid (0.3 GB), three permuted slab copies + their concatenation (4.9 + 4.9 GB) and the conv / relu /
    # normalize temporaries: > 14 GB (~560 KB per hypothesis).
    n = proposals.shape[0]
    own = n * (2 * 32768 + 2 * 8192)
    if defer:   # nothing per hypothesis but its score (and the encoder's workspace for one pair)
        assert peak <= 0.02 * own, (peak, own)
    else:
        assert peak <= 1.1 * own, (peak, own)
    print("option A (defer=%s) peak memory %.3f GB (the script's own tensors, materialised: %.2f GB)" % (defer, peak / 1e9, own / 1e9))


def test_patched_linemod_and_validation_lines_with_a_batch(ahv, dev, g128):
    """The other two call sites of the chain under patch.install(): test_linemod.py:45-63 (B pairs per step, plus the
    ground-truth line `rotate_volume(img_feat_src, gt_R)` .. `(f_gt * f_tgt).sum(dim=1).mean(dim=-1)`) and
    modules/model.py:183-196 (`.reshape(B, self.num_rota, -1, H*W)`), here with B = 2: deferred and op-level runs agree with
    each other and with the stock-operator restatement; the deferred run is ONE fused launch for the B x N hypotheses."""
    import types
    from oracle import torch_ref
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
    fa = Feature_Aligner().to(dev).eval()
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g128[k])).to(dev)
    with torch.no_grad():
        fa.feature_embedding_2d[0].weight.copy_(T("W1").reshape(32, 384, 1, 1))
        fa.feature_embedding_2d[2].weight.copy_(T("W2").reshape(32, 32, 1, 1))
        fa.feature_embedding_2d[2].bias.copy_(T("b2"))
    codebook = T("R")
    img_feat_src = torch.cat([T("vol_src"), T("vol_tgt").flip(2)])          # B = 2
    img_feat_tgt0 = torch.cat([T("vol_tgt"), T("vol_src").flip(3)])
    gt_src_2_tgt_R = codebook[[5, 9]]

    def lines(rotate_volume, forward_3d2d):     # test_linemod.py:45-63
        img_feat_tgt = img_feat_tgt0
        B, C, D, H, W = img_feat_src.shape
        img_feat_src_2_tgt = [rotate_volume(img_feat[None].expand(codebook.shape[0], -1, -1, -1, -1), codebook) for img_feat in img_feat_src]
        img_feat_src_2_tgt = torch.stack(img_feat_src_2_tgt).reshape(-1, C, D, H, W)
        img_feat_src_2_tgt = forward_3d2d(img_feat_src_2_tgt).reshape(B, codebook.shape[0], -1, H*W)
        img_feat_src_2_tgt_gt = rotate_volume(img_feat_src, gt_src_2_tgt_R)
        img_feat_src_2_tgt_gt = forward_3d2d(img_feat_src_2_tgt_gt)
        img_feat_tgt = forward_3d2d(img_feat_tgt)
        pred_sim = (img_feat_src_2_tgt * img_feat_tgt[:, None]).sum(dim=2).mean(dim=-1)
        gt_sim = (img_feat_src_2_tgt_gt * img_feat_tgt).sum(dim=1).mean(dim=-1)
        pred_index = torch.max(pred_sim, dim=1)[1]
        return pred_sim, gt_sim, pred_index, codebook[pred_index]
    out = {}
    for defer in (True, False):
        ahv.patch.install(um, mm, defer=defer)
        before = dict(ahv.deferred.counters)
        try:
            out[defer] = lines(um.rotate_volume, fa.forward_3d2d)
        finally:
            ahv.patch.uninstall()
        ran = {k: ahv.deferred.counters[k] - before[k] for k in before}
        if defer:   # the B x N chain: one launch; the ground-truth line is a batch of materialised volumes: not deferred at all
            assert ran == {"deferred_rotations": 2, "deferred_forward_3d2d": 1, "fused_score_launches": 1, "materialised": 0}, ran
        else:
            assert ran == {k: 0 for k in ran}, ran
    W = [T("W1"), T("W2"), T("b2")]
    ref = lines(torch_ref.rotate_volume, lambda x: torch_ref.forward_3d2d(x, *W))
    for got in out.values():
        assert torch.allclose(got[0], ref[0], rtol=1e-4, atol=1e-6) and torch.allclose(got[1], ref[1], rtol=1e-4, atol=1e-6)
        assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-6)


def test_unchanged_script_through_the_runner(ahv, dev, g128, tmp_path):
    """`python ahv_run.py script.py`: a stand-in checkout whose `utils.rotate_volume` and `Feature_Aligner.forward_3d2d` RAISE
    unless rebound, and a script with the reference's own lines (test_co3d.py:135-146) and not one line of ours -- run in its
    own process on the GPU, it prints fixture G1's arg-max and best score."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "modules").mkdir()
    (tmp_path / "modules" / "__init__.py").write_text("")
    (tmp_path / "utils.py").write_text("def rotate_volume(volume, rotation_matrix, padding_mode='zeros'):\n"
                                       "    raise AssertionError('utils.rotate_volume was not rebound')\n")
    (tmp_path / "modules" / "modules.py").write_text(
        "import torch\n"
        "class Feature_Aligner(torch.nn.Module):\n"
        "    def __init__(self):\n"
        "        super().__init__()\n"
        "        self.feature_embedding_2d = torch.nn.Sequential(torch.nn.Conv2d(384, 32, 1, bias=False), torch.nn.ReLU(), torch.nn.Conv2d(32, 32, 1))\n"
        "    def forward_3d2d(self, img_feat):\n"
        "        raise AssertionError('Feature_Aligner.forward_3d2d was not rebound')\n")
    (tmp_path / "test_script.py").write_text(
        "import sys\n"
        "import numpy as np\n"
        "import torch\n"
        "from utils import rotate_volume\n"
        "from modules.modules import Feature_Aligner\n"
        "g = np.load(sys.argv[1])\n"
        "T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda()\n"
        "class M(torch.nn.Module):\n"
        "    def __init__(self):\n"
        "        super().__init__()\n"
        "        self.feature_aligner = Feature_Aligner()\n"
        "model = M().cuda()\n"
        "with torch.no_grad():\n"
        "    fe = model.feature_aligner.feature_embedding_2d\n"
        "    fe[0].weight.copy_(T('W1').reshape(32, 384, 1, 1)); fe[2].weight.copy_(T('W2').reshape(32, 32, 1, 1)); fe[2].bias.copy_(T('b2'))\n"
        "model.eval()\n"
        "proposals, img_feat_src, img_feat_tgt = T('R'), T('vol_src'), T('vol_tgt')\n"
        "B, C, D, H, W = img_feat_src.shape\n"
        "img_feat_src_2_tgt = [rotate_volume(img_feat[None].expand(proposals.shape[0], -1, -1, -1, -1), proposals) for img_feat in img_feat_src]\n"
        "img_feat_src_2_tgt = torch.stack(img_feat_src_2_tgt).reshape(-1, C, D, H, W)\n"
        "img_feat_src_2_tgt = model.feature_aligner.forward_3d2d(img_feat_src_2_tgt).reshape(B, proposals.shape[0], -1, H*W)\n"
        "img_feat_tgt = model.feature_aligner.forward_3d2d(img_feat_tgt)\n"
        "pred_sim = (img_feat_src_2_tgt * img_feat_tgt[:, None]).sum(dim=2).mean(dim=-1)\n"
        "pred_sim, pred_index = torch.max(pred_sim, dim=1)\n"
        "pred_src_2_tgt_R = proposals[pred_index]\n"
        "print('RESULT', pred_index.item(), '%.8f' % pred_sim.item())\n")
    env = {k: v for k, v in os.environ.items() if k != "AHV_PATCH_DEFER"}
    out = subprocess.run([sys.executable, os.path.join(repo, "ahv_run.py"), "test_script.py", os.path.join(repo, "tests", "golden", "score_n128.npz")],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == int(g128["best_idx"][0]) and abs(float(line[2]) - float(g128["best"][0])) < 1e-5
    # without the runner the same script dies on the reference's own stand-in
    bare = subprocess.run([sys.executable, "test_script.py", os.path.join(repo, "tests", "golden", "score_n128.npz")],
                          cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert bare.returncode != 0 and "was not rebound" in bare.stderr


def test_deferred_hypotheses_fall_back_to_the_op_level_kernels(ahv, dev, g128):
    """Whatever else a script does with the deferred tensors of patch.install(): the values are those of the op-level kernels
    (deferred.py: the deferral can cost time, never a result).  The CPU suite checks the bookkeeping on a torch backend
    (tests/test_deferred_cpu.py); here the HIP backend."""
    D = ahv.deferred
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g128[k])).to(dev)
    vol, tgt, R, head = T("vol_src"), T("vol_tgt"), T("R"), (T("W1"), T("W2"), T("b2"))
    n = R.shape[0]
    exp = vol[0][None].expand(n, -1, -1, -1, -1)
    eager_rot = ahv.ops.rotate_volume(exp, R)
    eager_f = ahv.ops.forward_3d2d(eager_rot, *head)
    f_tgt = ahv.ops.forward_3d2d(tgt[:1], *head)
    before = dict(D.counters)
    d = D.defer_rotate_volume(exp, R)
    assert d is not None and d.is_cuda and d.shape == eager_rot.shape and d.deferred_kind == "rotated"
    assert torch.equal(d.cpu(), eager_rot.cpu()) and d.deferred_kind is None               # materialised on first touch, once
    assert torch.equal(D.defer_rotate_volume(exp, R)[5:7], eager_rot[5:7])
    assert torch.equal(torch.nn.functional.avg_pool3d(D.defer_rotate_volume(exp, R), 2), torch.nn.functional.avg_pool3d(eager_rot, 2))
    f = D.defer_rotate_volume(exp, R).with_head(*head)
    assert torch.equal(f.flatten(1), eager_f.flatten(1))
    prod = D.defer_rotate_volume(exp, R).with_head(*head).reshape(1, n, 32, 64) * f_tgt[:, None]
    assert prod.deferred_kind == "product" and torch.equal(prod.amax(dim=(2, 3)), (eager_f[None] * f_tgt[:, None]).amax(dim=(2, 3)))
    assert D.counters["fused_score_launches"] == before["fused_score_launches"]
    # ... and the chain itself against the same eager tensors (fused launch: its own summation order)
    s = (D.defer_rotate_volume(exp, R).with_head(*head).reshape(1, n, -1, 64) * f_tgt[:, None]).sum(dim=2).mean(dim=-1)
    assert D.counters["fused_score_launches"] == before["fused_score_launches"] + 1
    ref = (eager_f[None] * f_tgt[:, None]).sum(dim=2).mean(dim=-1)
    assert torch.allclose(s, ref, rtol=1e-5, atol=1e-6) and torch.equal(s.argmax(1), ref.argmax(1))
    # volumes that need a gradient are never deferred
    vg = vol[0].clone().requires_grad_(True)
    assert D.defer_rotate_volume(vg[None].expand(n, -1, -1, -1, -1), R) is None


def test_graphed_encoder_matches_eager(ahv, dev):
    torch.manual_seed(5)
    fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).to(dev).eval()
    run = fa.graphed_forward_2d3d(batch=1)
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        a, b = torch.randn(1, 768, 8, 8, generator=g).to(dev), torch.randn(1, 768, 8, 8, generator=g).to(dev)
        with torch.no_grad():
            ref = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
        got = run(a, b)
        for x, y in zip(got, ref):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-5)


def test_test_step_default_proposals_are_device_haar(ahv, model_obj, dev):
    b = batch(dev, 2, seed=9)
    n0 = len(model_obj.step_outputs)
    geo = model_obj.test_step(b, 0)  # no proposals given: fresh Haar set generated on the GPU
    assert geo.shape == (2,) and len(model_obj.step_outputs) == n0 + 1
    P1, P2 = model_obj.fresh_proposals(dev), model_obj.fresh_proposals(dev)
    assert P1.shape == (model_obj.num_rota, 3, 3) and P1.is_cuda and not torch.equal(P1, P2)


def test_infonce_forward_matches_torch_op_sequence(ahv, model_obj, dev):
    """Forward value of infoNCE_loss (modules/model.py:43-63) with per-sample hypothesis sets."""
    from oracle import torch_ref
    import math
    c = cfg(64)
    c["DATA"]["ACC_THR"] = 30
    torch.manual_seed(1)
    m = ahv.estimator.EstimatorObjaverse(c).to(dev).eval()
    g = torch.Generator().manual_seed(2)
    l4 = torch.randn(2, 2, 768, 8, 8, generator=g).to(dev)
    with torch.no_grad():
        vs, vt = m.forward_features(l4[0], l4[1])
    gt = ahv.rotations.random_rotations(2, generator=g).to(dev)
    R = ahv.rotations.random_rotations(2 * 63, generator=g).to(dev).reshape(2, 63, 3, 3)
    R = torch.cat([gt[:, None], R], dim=1)  # GT at index 0, as in training_step (modules/model.py:102-103)
    loss = m.infoNCE_loss(vs, vt, R, gt)
    assert loss.shape == (2,)
    W1, W2, b2 = m.feature_aligner.head_weights()
    with torch.no_grad():
        ref = []
        for i in range(2):
            s, _, _ = torch_ref.score_hypotheses(vs[i:i + 1], vt[i:i + 1], R[i], W1, W2, b2)
            gs = (torch.sum(R[i].flatten(1) * gt[i].reshape(1, 9), dim=-1).clamp(-1, 3) - 1) / 2
            pos = 180 * torch.arccos(gs) / math.pi <= 30
            e = torch.exp(s[0] / 0.1)
            ref.append(-torch.log(e[pos].sum() / e.sum().clamp(min=1e-8)))
    assert torch.allclose(loss, torch.stack(ref), rtol=1e-4, atol=1e-5)
    mc = ahv.estimator.EstimatorCo3d(c).to(dev).eval()
    mc.load_state_dict(m.state_dict())
    assert torch.allclose(mc.infoNCE_loss(vs, vt, R, gt), loss.mean(), rtol=1e-5)


def test_eval_co3d_tool_end_to_end_on_gpu(ahv, tmp_path, capsys):
    """tools/eval_co3d.py as a user would run it (counterpart of ``python test_co3d.py``, test_co3d.py:201-253), on
    the synthetic CO3D dataset with a randomly initialised Lightning-style checkpoint and the synthetic backbone:
    yaml -> checkpoint -> annotations -> HIP encoder + fused verify per pair -> co3d_result.txt."""
    import importlib.util
    import os
    import yaml
    from .conftest import REPO
    from .test_co3d_cpu import write_dataset
    spec = importlib.util.spec_from_file_location("eval_co3d", os.path.join(REPO, "tools", "eval_co3d.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    cfg, _ = write_dataset(str(tmp_path), n_frames=4)
    cfg["DATA"]["OBJ_SIZE"] = 256
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    torch.manual_seed(5)
    m = ahv.estimator.Estimator(dict(cfg, DATA=dict(cfg["DATA"], NUM_ROTA=16)),
                                feature_extractor=ahv.estimator.PatchifyBackbone())
    ahv.checkpoint.save_lightning_style(str(tmp_path / "c.ckpt"), m)
    argv = ["--config", str(tmp_path / "config.yaml"), "--ckpt", str(tmp_path / "c.ckpt"), "--categories", "ball",
            "--repeats", "2", "--num-rota", "3000", "--out-dir", str(tmp_path / "out")]
    assert tool.main(argv) == 0                       # default backbone (MiDaS) cannot be built offline
    assert "not measurable (no backbone" in capsys.readouterr().out
    assert tool.main(argv + ["--backbone", "patchify"]) == 0
    out = capsys.readouterr().out
    assert "Acc@15 [synthetic backbone: plumbing only]:" in out
    lines = (tmp_path / "out" / "co3d_result.txt").read_text().splitlines()
    assert [l[:10].strip() for l in lines] == ["ball", "mean"]
    assert 0.0 <= float(lines[0][10:16]) <= 180.0
