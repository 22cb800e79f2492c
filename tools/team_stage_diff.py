#!/usr/bin/env python3
"""Developer tool: at which stage do a team's and a lone wave's results first differ?  Needs diagnostic builds of the library
(make -C 3dahv_amd/csrc BUILD=tools/_dbg/s<k> CXXFLAGS="... -DAHV_DIAG_STAGE=<k>", k = 1..4: the 'score' of a hypothesis is
then an XOR checksum of the bits of stage k -- u, v, per-position sums, per-position cosines).  One process per build."""
import importlib, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--stage":
    import numpy as np, torch
    sys.path.insert(0, REPO)
    ahv = importlib.import_module("3dahv_amd")
    ahv._lib.LIB_PATH = os.path.join(REPO, "tools", "_dbg", "s" + sys.argv[2], "libahv_hip.so")
    ops = ahv.ops
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(REPO, "tests/golden/score_n128.npz"))
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev)
    vs, vt, W1, W2, b2 = (T(k) for k in ["vol_src", "vol_tgt", "W1", "W2", "b2"])
    ft = ops.forward_3d2d(vt, W1, W2, b2)
    for n in (64, 512):
        R = torch.from_numpy(ahv.rotations.haar_rotations_np(n, 100 + n)).to(dev)
        s1, _ = ops.score_hypotheses(vs, ft, R, W1, W2, b2, no_teams=True)
        s2, _ = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
        d = s1.view(torch.int32) != s2.view(torch.int32)
        print("stage %s, n=%d: %d of %d checksums differ (first at %s)" % (sys.argv[2], n, int(d.sum()), n, d.nonzero()[:6, 1].tolist()))
else:
    for k in "1234":
        subprocess.run([sys.executable, os.path.abspath(__file__), "--stage", k])
