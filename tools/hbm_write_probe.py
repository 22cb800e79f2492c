#!/usr/bin/env python3
"""Developer probe: what a PURE WRITE stream of rotate_volume's size (200 000 x 32 KiB = 6.55 GB) achieves on this box with stock
fills (torch.fill_, zero_ = hipMemset) and, for reference, a device-to-device copy of the same buffer (read + write) and a pure
read (sum): the yardsticks for the HBM-bound op-level kernels."""
import torch, time
dev = torch.device("cuda:0")
n = 200_000 * 8192
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[0], ts[len(ts) // 2]
gb = n * 4 / 1e9
for name, fn, traffic in (("fill_(1.0)", lambda: a.fill_(1.0), gb), ("zero_()", lambda: a.zero_(), gb),
                          ("copy_ (read + write)", lambda: b.copy_(a), 2 * gb), ("sum (read)", lambda: a.sum(), gb)):
    mn, med = t(fn)
    print("%-22s min %.4f median %.4f ms -> %.2f TB/s (median), %.2f (best)" % (name, mn, med, traffic / med, traffic / mn))
