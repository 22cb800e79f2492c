"""tools/eval_co3d.py -- the one-command counterpart of ``python test_co3d.py`` (test_co3d.py:201-253) -- on CPU:
absent inputs end in "Acc@15: not measurable (...)" with exit code 0, and with the synthetic CO3D dataset of
tests/test_co3d_cpu.py the whole chain (yaml -> cfg overrides -> annotations -> key frames -> verify -> geodesic error
-> 5-repeat average -> co3d_result.txt) runs with a stand-in estimator and an oracle-backed verify step (test
infrastructure injected through ``main(model=, verify_fn=)``; the product's verify step is HIP only)."""
import importlib.util
import os

import numpy as np
import torch
import yaml

from .conftest import REPO
from .test_co3d_cpu import write_dataset


def _tool():
    spec = importlib.util.spec_from_file_location("eval_co3d", os.path.join(REPO, "tools", "eval_co3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_not_measurable_paths(tmp_path, capsys):
    tool = _tool()
    assert tool.main(["--config", str(tmp_path / "nope.yaml")]) == 0
    assert "Acc@15: not measurable (no config file" in capsys.readouterr().out
    cfg = {"RUN_NAME": "x", "DATA": {"NUM_ROTA": 10}, "CO3D": {"CO3D_DIR": str(tmp_path / "img"),
                                                               "CO3D_ANNOTATION_DIR": str(tmp_path / "ann")}}
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    assert tool.main(["--config", str(tmp_path / "config.yaml"), "--ckpt", str(tmp_path / "missing.ckpt")]) == 0
    out = capsys.readouterr().out
    assert "not measurable" in out and "no data" in out and "no checkpoint" in out


def test_synthetic_dataset_end_to_end(ahv, oracle, tmp_path, capsys):
    tool = _tool()
    cfg, _ = write_dataset(str(tmp_path), n_frames=4)
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    g = np.load(os.path.join(REPO, "tests", "golden", "score_n128.npz"))

    class Model:  # stands in for Estimator.forward: two frames -> two volumes (deterministic in the images)
        def __call__(self, a, b):
            gen = torch.Generator().manual_seed(int(a.abs().sum().item() * 1000) % 2**31)
            v = torch.randn(2, a.shape[0], 16, 8, 8, 8, generator=gen)
            return v[0], v[1]

    def verify(vs, vt, P):  # oracle (the checker) in place of the HIP launch
        _, best, idx = oracle.score_hypotheses(vs.numpy(), vt.numpy(), P.numpy(), g["W1"], g["W2"], g["b2"])
        return torch.from_numpy(best), torch.from_numpy(idx)

    out_dir = tmp_path / "out"
    rc = tool.main(["--config", str(tmp_path / "config.yaml"), "--categories", "ball,book", "--repeats", "2",
                    "--num-rota", "64", "--device", "cpu", "--out-dir", str(out_dir)], model=Model(), verify_fn=verify)
    assert rc == 0
    out = capsys.readouterr().out
    assert "categories without annotations skipped: book" in out and "Acc@15:" in out
    lines = (out_dir / "co3d_result.txt").read_text().splitlines()
    assert len(lines) == 2 and lines[0].startswith(f"{'ball':>10s}") and lines[0].endswith(" ")
    assert lines[1].startswith(f"{'mean':>10s}") and lines[1][10:] == lines[0][10:]   # one category: mean == it
    err, a15, a30 = float(lines[0][10:16]), float(lines[0][16:22]), float(lines[0][22:28])
    assert 0.0 <= err <= 180.0 and 0.0 <= a15 <= a30 <= 100.0
