"""The GPU tests must not depend on a binary that happened to travel with the snapshot: on the GPU box itself
(1) the session's library is brought up to date by ``_lib.build()`` (incremental make; see conftest.py), and
(2) here the whole library is rebuilt FROM SOURCE into a scratch directory (every object, ``make -B``), loaded in a
fresh process and checked against the N = 128 golden case and the exported-symbol list."""
import os
import subprocess
import sys

import pytest

from .conftest import REPO

pytestmark = pytest.mark.gpu

CHECK = r'''
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ["AHV_REPO"])
import importlib
ahv = importlib.import_module("3dahv_amd")
ahv._lib.LIB_PATH = os.environ["AHV_LIB"]          # the freshly built library, not the in-tree one
lib = ahv._lib.load()
assert lib._name == os.environ["AHV_LIB"], lib._name
g = np.load(os.path.join(os.environ["AHV_REPO"], "tests", "golden", "score_n128.npz"))
T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda()
ft = ahv.ops.forward_3d2d(T("vol_tgt"), T("W1"), T("W2"), T("b2"))
s, key = ahv.ops.score_hypotheses(T("vol_src"), ft, T("R"), T("W1"), T("W2"), T("b2"))
best, idx = ahv.ops.unpack_best(key)
rel = float(np.max(np.abs(s.cpu().numpy() - g["scores"]) / np.maximum(np.abs(g["scores"]), 1e-2)))
assert rel < 1e-4 and int(idx.item()) == int(g["best_idx"][0]), (rel, idx)
print("REBUILT_OK", rel)
'''


def test_library_rebuilds_from_source_on_the_gpu_box(tmp_path):
    build = tmp_path / "build"
    build.mkdir()
    csrc = os.path.join(REPO, "3dahv_amd", "csrc")
    proc = subprocess.run(["make", "-C", csrc, "-B", "-j8", "BUILD=%s" % build], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stdout[-4000:]
    so = build / "libahv_hip.so"
    assert so.exists() and (build / "ahv_score.o").exists()
    script = tmp_path / "check.py"
    script.write_text(CHECK)
    env = dict(os.environ, AHV_REPO=REPO, AHV_LIB=str(so))
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert out.returncode == 0 and "REBUILT_OK" in out.stdout, out.stdout[-4000:]
    # same exported C ABI as the header declares (the in-tree check of tests/test_abi.py, on the fresh binary)
    nm = subprocess.run(["nm", "-D", "--defined-only", str(so)], stdout=subprocess.PIPE, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if l.split()[-1].startswith("ahv_")}
    import importlib
    ahv = importlib.import_module("3dahv_amd")
    declared = set(ahv._lib.SIGNATURES) | set(ahv._lib.DIAG_SIGNATURES)   # include/ahv.h + include/ahv_diag.h
    assert exported == declared, exported ^ declared
