#!/usr/bin/env python3
"""Developer tool: time ops.rotate_volume at N = 200 000 with the library given as argv[1] (default: the in-tree build).
`make -C 3dahv_amd/csrc BUILD=tools/_dbg/lin CXXFLAGS="... -DAHV_DIAG_LINEAR_GATHER"` gives the conflict-free bound of the
gather (wrong results): how much of the kernel's time is LDS bank conflicts."""
import importlib, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ahv = importlib.import_module("3dahv_amd")
if len(sys.argv) > 1:
    ahv._lib.LIB_PATH = os.path.abspath(sys.argv[1])
ops = ahv.ops
dev = torch.device("cuda:0")
N = 200_000
R = ops.so3_grid(N, dev)
v = (torch.randn(1, 16, 8, 8, 8, generator=torch.Generator().manual_seed(0)) * 1.15).to(dev)
src = v[0][None].expand(N, -1, -1, -1, -1)
out = ops.rotate_volume(src, R)
for _ in range(3):
    ops.rotate_volume(src, R)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.rotate_volume(src, R); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print("%s: rotate_volume N=%d  min %.4f  median %.4f ms  -> %.2f TB/s (32 804 B per hypothesis)" % (
    os.path.basename(os.path.dirname(ahv._lib.LIB_PATH)), N, ts[0], ts[5], N * 32804 / ts[5] / 1e9))
