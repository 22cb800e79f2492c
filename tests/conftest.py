import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def ahv():
    """The package, with libahv_hip.so brought up to date FROM SOURCE first wherever a GPU is present (incremental
    make: a no-op when the binary that travelled with the snapshot is newer than every source; a stale or missing
    binary is rebuilt, and a build failure fails the session instead of testing an old library)."""
    pkg = importlib.import_module("3dahv_amd")
    import torch
    if torch.cuda.is_available():
        pkg._lib.build()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def g128():
    return load_golden("score_n128")


def nonfinite_cases(g128):
    """G10 `nonfinite` (tools/gen_golden.py gen_nonfinite: the reference's hot loop on inputs with ONE non-finite element):
    yields (name, inputs dict with vol_src / vol_tgt / W1 / W2 / b2, R, reference scores (1, N), best idx)."""
    g = load_golden("nonfinite")
    for k, name in enumerate(g["names"]):
        inp = {key: np.array(g128[key]) for key in ("vol_src", "vol_tgt", "W1", "W2", "b2")}
        if "abs_src" in g.files and bool(g["abs_src"][k]):   # strictly positive volumes (gen_nonfinite)
            for key in ("vol_src", "vol_tgt"):
                inp[key] = (np.abs(inp[key]) + np.float32(0.1)).astype(np.float32)
        which = {"src": "vol_src", "tgt": "vol_tgt"}.get(str(g["tensor"][k]), str(g["tensor"][k]))
        idx = tuple(int(i) for i in g["index"][k] if i >= 0)
        inp[which].view(np.uint32)[idx] = g["bits"][k]
        yield str(name), inp, g["R"], g["scores"][k], int(g["best_idx"][k][0])
