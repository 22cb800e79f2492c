#!/usr/bin/env python3
"""Quick GPU sanity + timing (developer tool; run through gpurun)."""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ahv = importlib.import_module("3dahv_amd")
from oracle import oracle
ops = ahv.ops
g = np.load(os.path.join(REPO, "tests/golden/score_n128.npz"))
dev = torch.device("cuda")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vs, vt, R, W1, W2, b2 = T(g["vol_src"]), T(g["vol_tgt"]), T(g["R"]), T(g["W1"]), T(g["W2"]), T(g["b2"])
print("CUs", ahv._lib.load().ahv_device_cu_count())
rot = ops.rotate_volume(vs[0][None].expand(128, -1, -1, -1, -1), R)
print("rotate err", (rot[:2].cpu().numpy() - g["rot_first2"]).__abs__().max())
ft = ops.forward_3d2d(vt, W1, W2, b2)
print("f_tgt err", np.abs(ft.cpu().numpy() - g["f_tgt"]).max())
fs = ops.forward_3d2d(rot, W1, W2, b2)
print("f_src err", np.abs(fs[:2].cpu().numpy() - g["f_src_first2"]).max())
sc = ops.score_features(fs[None], ft)
print("oplevel score err", np.abs(sc.cpu().numpy() - g["scores"]).max())
bs, bi = ops.argmax(sc)
print("oplevel argmax", bs.item(), bi.item(), g["best"], g["best_idx"])
scores, key = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
e = np.abs(scores.cpu().numpy() - g["scores"])
print("fused score abs err", e.max(), "rel", (e / np.abs(g["scores"])).max())
print("fused best", [x.tolist() for x in ops.unpack_best(key)])
# timing at N=50000
Rn = T(ahv.rotations.haar_rotations_np(50000, 3))
lib = ahv._lib.load()
for variant in (0, 1, 0, 1):
  lib.ahv_set_option(b"score_variant", variant)
  sv, kv = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
  print("variant", variant, "n128 abs err", np.abs(sv.cpu().numpy() - g["scores"]).max(), [x.tolist() for x in ops.unpack_best(kv)])
  for want in (False,):
    for _ in range(3):
        ops.score_hypotheses(vs, ft, Rn, W1, W2, b2, want_scores=want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 20
    for _ in range(it):
        s50, k50 = ops.score_hypotheses(vs, ft, Rn, W1, W2, b2, want_scores=want)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print("variant", variant, "fused N=50000 want_scores=%s: %.3f ms -> %.3e hyp/s (%.1f TFLOP/s)" % (want, dt * 1e3, 50000 / dt, 50000 / dt * 1839104 / 1e12))
g50 = np.load(os.path.join(REPO, "tests/golden/score_n50k_digest.npz"))
s50, k50 = ops.score_hypotheses(vs, ft, Rn, W1, W2, b2)
print("50k best", [x.tolist() for x in ops.unpack_best(k50)], g50["best"], g50["best_idx"])
print("50k every97 err", np.abs(s50[0, ::97].cpu().numpy() - g50["every97_score"]).max())
for name, fn in [("rotate", lambda: ops.rotate_volume(vs[0][None].expand(50000, -1, -1, -1, -1), Rn))]:
    out = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%s N=50000: %.3f ms, %.1f GB/s" % (name, dt * 1e3, 50000 * 32804 / dt / 1e9))
rot50 = out
fn = lambda: ops.forward_3d2d(rot50, W1, W2, b2)
f50 = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f50 = fn()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("forward_3d2d M=50000: %.3f ms, %.1f GB/s, %.3e items/s" % (dt * 1e3, 50000 * 40960 / dt / 1e9, 50000 / dt))
fn = lambda: ops.score_features(f50[None], ft)
o = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): o = fn()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("score_features N=50000: %.3f ms, %.1f GB/s" % (dt * 1e3, 50000 * 8196 / dt / 1e9))
print("oplevel vs fused 50k max abs diff", (o - s50).abs().max().item())
