#!/usr/bin/env python3
"""forward_2d3d at B = 32 (the evaluation harness' default batch), a few eager calls: what rocprofv3 --kernel-trace --stats
is pointed at to see which encoder kernels the 2.3 ms are (tools/gpu_run.sh encoder32)."""
import importlib, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ahv = importlib.import_module("3dahv_amd")
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).to(dev).eval()
a, b = torch.randn(B, 768, 8, 8, device=dev), torch.randn(B, 768, 8, 8, device=dev)
with torch.no_grad():
    for _ in range(3):
        fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
    e1.record()
    torch.cuda.synchronize()
print("B = %d: %.1f us per forward" % (B, e0.elapsed_time(e1) * 100))
