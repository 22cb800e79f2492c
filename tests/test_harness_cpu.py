"""Harness / Estimator plumbing on CPU (BASELINE.json configs[0]: N=128, no GPU).  The verify step is
injected as the CPU oracle here; the product default (HIP) is covered by the -m gpu tests."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from .conftest import load_golden


def small_cfg(n=128):
    return {"RUN_NAME": "Synthetic_3DAHV", "DATA": {"NUM_ROTA": n, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256}}


def oracle_verify(oracle, model):
    W1, W2, b2 = (t.detach().numpy() for t in model.feature_aligner.head_weights())

    def fn(vol_src, vol_tgt, proposals):
        s, best, idx = oracle.score_hypotheses(vol_src.numpy(), vol_tgt.numpy(), proposals.numpy(), W1, W2, b2)
        return torch.from_numpy(best), torch.from_numpy(idx)
    return fn


def test_config1_golden_end_to_end(ahv, oracle, g128):
    """Fixture G1 through the harness: one pair, N=128 -> same best index / score as the reference."""
    model = ahv.estimator.EstimatorCo3d(small_cfg(), feature_extractor=None)
    with torch.no_grad():
        fe = model.feature_aligner.feature_embedding_2d
        fe[0].weight.copy_(torch.from_numpy(g128["W1"]).reshape(32, 384, 1, 1))
        fe[2].weight.copy_(torch.from_numpy(g128["W2"]).reshape(32, 32, 1, 1))
        fe[2].bias.copy_(torch.from_numpy(g128["b2"]))
    vol_src, vol_tgt = torch.from_numpy(g128["vol_src"]), torch.from_numpy(g128["vol_tgt"])
    model.forward_features = lambda a, b: (vol_src, vol_tgt)  # volumes of the fixture instead of an encoder run
    seq = [{"n": 2, "model_id": "g1", "R": torch.eye(3)[None].repeat(2, 1, 1), "layer4": torch.zeros(2, 768, 8, 8)}]
    np.random.seed(0)
    errs, det = ahv.harness.evaluate_category(small_cfg(), model, seq, proposals=torch.from_numpy(g128["R"]),
                                              device=torch.device("cpu"), verify_fn=oracle_verify(oracle, model),
                                              return_details=True)
    assert len(errs) == 2 and [d["pair"] for d in det] == [(0, 1), (1, 0)]
    for d in det:
        assert d["idx"] == int(g128["best_idx"][0])
        assert abs(d["best"] - float(g128["best"][0])) < 1e-5
        assert np.allclose(d["R_pred"], g128["R"][d["idx"]])
    # GT is the identity here, so the error is the rotation angle of the predicted hypothesis
    R = g128["R"][int(g128["best_idx"][0])]
    ang = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))
    assert abs(errs[0] - ang) < 1e-2


def test_synthetic_run_writes_reference_format(ahv, oracle, tmp_path):
    torch.manual_seed(1)
    cfg = small_cfg(64)
    model = ahv.estimator.EstimatorCo3d(cfg).eval()
    cats = {"apple": ahv.harness.SyntheticSequences(2, 3, seed=1), "ball": ahv.harness.SyntheticSequences(1, 3, seed=2)}
    lines = ahv.harness.run_co3d(cfg, model, cats, repeats=2, out_dir=str(tmp_path), device=torch.device("cpu"),
                                 verify_fn=oracle_verify(oracle, model))
    text = open(tmp_path / "co3d_result.txt").read().splitlines()
    assert [l.rstrip() for l in text] == [l.rstrip() for l in lines] and len(lines) == 3
    assert lines[0].startswith("     apple") and lines[2].startswith("      mean")
    for l in lines:
        assert len(l) == 10 + 18
        err, a15, a30 = float(l[10:16]), float(l[16:22]), float(l[22:28])
        assert 0 <= err <= 180 and 0 <= a15 <= a30 <= 100
    assert ahv.harness.format_result_line("banana", 12.3456, 50.0, 75.5) == "    banana 12.35 50.00 75.50"
    # determinism: the harness reseeds (test_co3d.py:24-25)
    lines2 = ahv.harness.run_co3d(cfg, model, cats, repeats=2, out_dir=str(tmp_path), device=torch.device("cpu"),
                                  verify_fn=oracle_verify(oracle, model))
    assert lines2 == lines


def test_permutations_match_reference_order(ahv):
    assert ahv.harness.get_permutations(2).tolist() == [[0, 1], [1, 0]]
    assert ahv.harness.get_permutations(3).tolist() == [[0, 1], [0, 2], [1, 0], [1, 2], [2, 0], [2, 1]]


def test_checkpoint_roundtrip_and_surface(ahv, tmp_path):
    cfg = small_cfg()
    m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone())
    path = str(tmp_path / "checkpoint_co3d.ckpt")
    ahv.checkpoint.save_lightning_style(path, m)
    sd = ahv.checkpoint.read_state_dict(path)
    assert any(k.startswith("feature_aligner.att.transformer_blocks.3.") for k in sd)
    assert any(k.startswith("feature_extractor.") for k in sd)
    m2 = ahv.estimator.EstimatorCo3d.load_from_checkpoint(path, cfg=cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=5))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(TypeError):
        ahv.estimator.EstimatorCo3d.load_from_checkpoint(path)  # cfg is mandatory, as in the reference call
    assert m2.num_rota == 128 and m2.step_outputs == [] and m2.gt_dis == [] and m2.pred_Rs == []
    assert isinstance(m2.eval(), ahv.estimator.EstimatorCo3d)
    with pytest.raises(KeyError):
        m2.training_step({}, 0)  # implemented (HIP backward on the GPU); an empty batch has no "image"
    with torch.no_grad():
        v1, v2 = m2(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256))
    assert v1.shape == v2.shape == (1, 16, 8, 8, 8)
    bare = ahv.estimator.EstimatorCo3d(cfg)
    with pytest.raises(RuntimeError, match="feature_extractor"):
        bare(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256))


def test_objaverse_forward_masks_and_skip(ahv):
    cfg = small_cfg()
    cfg["DATA"]["BG"] = False
    m = ahv.estimator.EstimatorObjaverse(cfg, feature_extractor=ahv.estimator.PatchifyBackbone()).eval()
    img = torch.rand(2, 3, 256, 256)
    with torch.no_grad():
        a, _ = m(img, torch.zeros(2, 1, 256, 256), img, torch.ones(2, 1, 256, 256))
        b, _ = m(torch.zeros_like(img), torch.ones(2, 1, 256, 256), img, torch.ones(2, 1, 256, 256))
    assert torch.allclose(a, b)  # BG False: the source image is multiplied by its (zero) mask
    batch = {"src_mask": torch.zeros(2, 1, 256, 256), "ref_mask": torch.ones(2, 1, 256, 256), "src_img": img,
             "ref_img": img, "src_R": torch.eye(3)[None].repeat(2, 1, 1), "ref_R": torch.eye(3)[None].repeat(2, 1, 1)}
    assert m.test_step(batch, 0) == 0 and m.step_outputs == []  # mask area < SIZE_THR -> "Skip bad case"


def test_patch_install_rebinds_reference_callables(ahv):
    utils_mod = types.ModuleType("utils")
    ref_fn = lambda volume, rotation_matrix, padding_mode="zeros": "reference"
    utils_mod.rotate_volume = ref_fn
    mm = types.ModuleType("modules.modules")

    class Feature_Aligner:  # noqa: N801 (the reference's class name)
        def forward_3d2d(self, x):
            return "reference"
    mm.Feature_Aligner = Feature_Aligner
    script = types.ModuleType("fake_script")
    script.rotate_volume = ref_fn  # what `from utils import rotate_volume` leaves behind
    sys.modules["fake_script"] = script
    try:
        ahv.patch.install(utils_mod, mm)
        assert utils_mod.rotate_volume is ahv.patch._hip_rotate_volume     # (ops.rotate_volume behind the deferral check)
        assert script.rotate_volume is ahv.patch._hip_rotate_volume
        assert Feature_Aligner.forward_3d2d is not None and Feature_Aligner().forward_3d2d.__func__ is ahv.patch._hip_forward_3d2d
        ahv.patch.uninstall()
        assert utils_mod.rotate_volume is ref_fn and script.rotate_volume is ref_fn
        assert Feature_Aligner().forward_3d2d(None) == "reference"
    finally:
        sys.modules.pop("fake_script", None)


def test_configure_optimizers_follow_each_variant(ahv):
    """AdamW eps 1e-5 in both; co3d: one rate, StepLR(200) (model_co3d.py:93-99); objaverse: backbone at a tenth,
    StepLR(20) (model.py:212-218)."""
    cfg = small_cfg()
    cfg["TRAIN"] = {"LR": 1e-4, "MASK": True, "MASK_RATIO": 0.25}
    for cls, rates, step in ((ahv.estimator.EstimatorCo3d, (1e-4, 1e-4), 200), (ahv.estimator.EstimatorObjaverse, (1e-4, 1e-5), 20)):
        m = cls(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1))
        (opt,), (sched,) = m.configure_optimizers()
        assert isinstance(opt, torch.optim.AdamW) and all(g["eps"] == 1e-5 for g in opt.param_groups)
        assert tuple(pytest.approx(g["lr"]) for g in opt.param_groups) == rates
        assert sched.step_size == step and sched.gamma == 0.1
