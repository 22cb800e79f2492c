"""Host-side logic (no GPU): hypothesis samplers, key codec, shard partition, error metric."""
import hashlib

import numpy as np
import pytest
import torch

from .conftest import load_golden


def test_haar_sampler_reproducible_and_orthonormal(ahv):
    R = ahv.rotations.haar_rotations_np(2000, 3)
    assert R.dtype == np.float32 and R.shape == (2000, 3, 3)
    assert np.array_equal(R, ahv.rotations.haar_rotations_np(50000, 3)[:2000])
    eye = np.einsum("nij,nkj->nik", R.astype(np.float64), R.astype(np.float64))
    assert np.allclose(eye, np.eye(3), atol=1e-6)
    assert np.allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-6)
    g = load_golden("score_n50k_digest")
    assert hashlib.sha256(ahv.rotations.haar_rotations_np(50000, 3).tobytes()).hexdigest() == str(g["R_sha256"])


def test_haar_statistics(ahv):
    # Haar measure: rotation angle density (1-cos t)/pi -> E[trace] = 0, E[angle] = pi/2 + 2/pi
    R = ahv.rotations.random_rotations(20000, generator=torch.Generator().manual_seed(0)).double()
    tr = R.diagonal(dim1=1, dim2=2).sum(-1)
    assert abs(tr.mean().item()) < 0.03
    ang = torch.arccos(((tr - 1) / 2).clamp(-1, 1))
    assert abs(ang.mean().item() - (np.pi / 2 + 2 / np.pi)) < 0.02
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand_as(R), atol=1e-6)


def test_so3_grid_covers_uniformly(ahv):
    G = ahv.rotations.so3_grid_np(4096).astype(np.float64)
    assert np.allclose(np.einsum("nij,nkj->nik", G, G), np.eye(3), atol=1e-5)
    Q = ahv.rotations.haar_rotations_np(200, 1).astype(np.float64)
    tr = np.einsum("qij,nij->qn", Q, G)  # trace(Q^T G)
    nearest = np.degrees(np.arccos(np.clip((tr.max(axis=1) - 1) / 2, -1, 1)))
    assert nearest.max() < 25.0 and nearest.mean() < 12.0  # 4096 points: ~10 deg spacing


def test_refine_rotations_stay_local(ahv):
    Rs = torch.from_numpy(ahv.rotations.haar_rotations_np(1, 2)[0])
    out = ahv.rotations.refine_rotations(Rs, 500, 10.0, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(out[0], Rs, atol=1e-6)
    err = ahv.rotations.geodesic_deg(out, Rs[None].expand_as(out))
    assert err.max().item() <= 10.0 + 1e-2 and err.mean().item() > 4.0


def test_geodesic_metric_golden(ahv):
    g = load_golden("metric")
    err = ahv.rotations.geodesic_deg(torch.from_numpy(g["R_pred"]), torch.from_numpy(g["R_gt"])).numpy()
    assert np.array_equal(err, g["err_deg"], equal_nan=True)  # same torch expression: bit-exact


def test_key_codec_matches_torch_max(ahv):
    d = ahv.dist
    rng = np.random.RandomState(0)
    s = rng.standard_normal(1000).astype(np.float32)
    s[[5, 17]] = s.max() + 1  # tie: lowest index wins
    keys = d.pack_keys_host(s, np.arange(1000))
    merged = d.merge_keys(torch.from_numpy(keys)[:, None])
    score, idx = d.unpack_keys_host(merged.numpy())
    assert idx[0] == 5 and score[0] == s[5]
    # ordering across sign, zero and NaN
    vals = np.array([-np.inf, -2.0, -0.0, 0.0, 1e-30, 3.0, np.inf, np.nan], dtype=np.float32)
    k = d.pack_keys_host(vals, np.zeros(8))   # int64, signed order (include/ahv.h "Packed keys")
    assert k.dtype == np.int64 and k[2] == k[3] and all(k[i] < k[i + 1] for i in (0, 1, 3, 4, 5, 6))
    assert np.all(k > d.KEY_EMPTY) and d.KEY_EMPTY == np.iinfo(np.int64).min
    sc, ix = d.unpack_keys_host(d.pack_keys_host(vals, np.arange(8)))
    assert np.array_equal(sc[[0, 1, 4, 5, 6]], vals[[0, 1, 4, 5, 6]]) and np.isnan(sc[7]) and list(ix) == list(range(8))
    sc, ix = d.unpack_keys_host(np.full(2, d.KEY_EMPTY, dtype=np.int64))
    assert list(ix) == [-1, -1] and np.all(np.isneginf(sc))


def test_shard_range_partitions(ahv):
    for n in (0, 1, 7, 50000, 200000):
        for w in (1, 2, 3, 8):
            parts = [ahv.dist.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        ahv.dist.shard_range(10, 2, 2)


def test_lds_conflict_simulation_backs_the_table_in_ahv_device_h():
    """tools/lds_conflict_sim.py: the simulated bank-conflict factors quoted in 3dahv_amd/csrc/ahv_device.h (and measured on
    the GPU: SQ_LDS_BANK_CONFLICT 1 477 -> 684 cycles per hypothesis) -- x-run lane map at (8, 74) rows 2.43, the box map
    at (9, 76) rows 1.64 -- and the lane -> voxel map itself (a bijection of the quarter, as lane_vox computes it)."""
    import importlib.util
    import os
    import numpy as np
    from .conftest import REPO
    spec = importlib.util.spec_from_file_location("lds_conflict_sim", os.path.join(REPO, "tools", "lds_conflict_sim.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    R = sim.haar(300, np.random.default_rng(0))
    assert abs(sim.factor(sim.lane_map_xrun(), 8, 74, R) - 2.43) < 0.06
    assert abs(sim.factor(sim.lane_map_box(), 9, 76, R) - 1.64) < 0.06
    box = sim.lane_map_box()
    for Q in range(4):
        vox = {tuple(v) for p in range(2) for v in box[Q, p].astype(int).tolist()}
        assert vox == {(x, y, z) for x in range(8) for y in range(8) for z in (2 * Q, 2 * Q + 1)}
    # lane_vox of ahv_dual.h, restated: first-group mask 0x0FF0F00F, rank inside the group by popcount
    for lane in range(64):
        l, first = lane & 31, (0x0FF0F00F >> (lane & 31)) & 1
        mask = (0x0FF0F00F if first else ~0x0FF0F00F & 0xFFFFFFFF) & ((1 << l) - 1)
        k = bin(mask).count("1")
        want = ((0 if first else 4) + ((k >> 1) & 3), 2 * (lane >> 5) + (k >> 3), k & 1)
        assert tuple(box[0, 0, lane].astype(int)) == want, lane


def test_estimator_tries_the_reference_backbone_factory(ahv, tmp_path, monkeypatch):
    """modules/model_co3d.py:32: ``Estimator(cfg)`` builds ``DPT_SwinV2_T_256(pretrained=True)`` from MiDaS/hubconf.py.  The
    mirror attempts exactly that (MiDaS dir from cfg / $AHV_MIDAS_DIR / ./MiDaS): with a hubconf present the one-argument
    constructor yields a model whose forward(img, img) runs; without one (or with a failing factory) it says why."""
    import sys
    est = ahv.estimator
    cfg = {"DATA": {"NUM_ROTA": 8, "ACC_THR": 15.0}, "TRAIN": {"LR": 1e-4, "MASK": False, "MASK_RATIO": 0.0}}
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("AHV_MIDAS_DIR", raising=False)
    sys.modules.pop("hubconf", None)
    m = est.Estimator(cfg)
    assert m.feature_extractor is None and "no MiDaS/hubconf.py" in est.midas_unavailable_reason
    with pytest.raises(RuntimeError, match="hubconf"):
        m.forward(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256))
    # a stand-in MiDaS checkout: same factory name, same forward_transformer contract (layer_4 = (B,768,8,8))
    midas = tmp_path / "MiDaS"
    midas.mkdir()
    (midas / "hubconf.py").write_text(
        "import torch\n"
        "class _DPT(torch.nn.Module):\n"
        "    def __init__(self, pretrained):\n"
        "        super().__init__()\n"
        "        self.loaded = pretrained\n"
        "        self.pretrained = torch.nn.Conv2d(3, 768, 32, stride=32)\n"
        "    def forward_transformer(self, pretrained, x):\n"
        "        y = pretrained(x)\n"
        "        return y, y, y, y\n"
        "def DPT_SwinV2_T_256(pretrained=True, **kw):\n"
        "    return _DPT(pretrained)\n")
    try:
        m = est.Estimator(cfg)                      # found through ./MiDaS, the reference's working directory
        assert type(m.feature_extractor).__name__ == "_DPT" and m.feature_extractor.loaded is True
        assert est.midas_unavailable_reason == ""
        torch.manual_seed(0)
        with torch.no_grad():
            v_src, v_tgt = m.forward(torch.randn(1, 3, 256, 256), torch.randn(1, 3, 256, 256))
        assert v_src.shape == (1, 16, 8, 8, 8) and v_tgt.shape == (1, 16, 8, 8, 8) and torch.isfinite(v_src).all()
        assert any(k.startswith("feature_extractor.pretrained.") for k in m.state_dict())   # checkpoint keys as the reference's
        monkeypatch.setenv("AHV_MIDAS_PRETRAINED", "0")
        assert est.Estimator(cfg).feature_extractor.loaded is False
        # a factory that fails (timm missing, no network): None + the reason, no exception out of the constructor
        (midas / "hubconf.py").write_text("def DPT_SwinV2_T_256(pretrained=True, **kw):\n    raise ImportError('No module named timm')\n")
        sys.modules.pop("hubconf", None)
        m = est.Estimator({**cfg, "MODEL": {"MIDAS_DIR": str(midas)}})
        assert m.feature_extractor is None and "No module named timm" in est.midas_unavailable_reason
    finally:
        sys.modules.pop("hubconf", None)
        if str(midas) in sys.path:
            sys.path.remove(str(midas))


def test_bench_lane_rule(monkeypatch):
    """bench.py default_lanes: one lane at 1 and 2 ranks (25 000 hypotheses per rank: two chip-sized grids would queue for the
    same CUs), two lanes from 4 ranks on (the fixed ~18 us of a short launch are worth overlapping); the same on every rank."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("AHV_BENCH_TWO_LANES_PG", "AHV_BENCH_TWO_LANES_MAX_N"):
        monkeypatch.delenv(k, raising=False)
    assert [bench.default_lanes(w, 50000) for w in (1, 2, 3, 4, 6, 8)] == [1, 1, 1, 2, 2, 2]
    monkeypatch.setenv("AHV_BENCH_TWO_LANES_MAX_N", "25000")
    assert bench.default_lanes(2, 50000) == 2 and bench.default_lanes(1, 50000) == 1
    monkeypatch.setenv("AHV_BENCH_TWO_LANES_PG", "1")
    assert bench.default_lanes(1, 50000) == 2
