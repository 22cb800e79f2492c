#!/usr/bin/env python3
"""Kernel-trace timeline of bench.py's timed steps (rocprofv3 --kernel-trace CSV): what runs between two launches of
the fused verify kernel, on which queue, and how long the device is in no kernel at all.

    python3 tools/summarize_timeline.py <kernel_trace.csv> [<label>]

Prints a window of consecutive dispatches in steady state and, over the middle launches of the verify kernel, the
launch-to-launch period, the kernel duration and their difference (= time per step that is not the scorer)."""
import csv
import sys

import numpy as np

path = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else path
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
main = [i for i, r in enumerate(rows) if "score_hypotheses_dual_kernel<false, true>" in r["Kernel_Name"]]
# bench.py: pre-warm, warm-up, then the timed steps, then the strong-scaling legs: the longest run of equal-period launches
# sits in the timed region; take the launches between 30 % and 60 % of the B = 1 launches as the steady-state sample
b1 = [i for i in main if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 2_000_000]
sel = b1[int(0.3 * len(b1)):int(0.6 * len(b1))]
starts = np.array([int(rows[i]["Start_Timestamp"]) for i in sel])
durs = np.array([int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in sel]) / 1e3
period = np.diff(starts) / 1e3
print("== %s: %d dispatches, %d launches of the verify kernel, steady-state sample of %d" % (label, len(rows), len(main), len(sel)))
print("   launch-to-launch period: median %.1f us (mean %.1f); kernel duration: median %.1f us; not in the scorer: %.1f us per step"
      % (np.median(period), period.mean(), np.median(durs), np.median(period) - np.median(durs)))
names = {}
for i in range(sel[0], sel[-1]):
    n = rows[i]["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
    names.setdefault(n, []).append(d)
for n, d in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print("   %-72s x%-5d mean %8.2f us   per step %.2f" % (n, len(d), np.mean(d), len(d) / len(sel)))
i0 = sel[len(sel) // 2]
t0, prev = int(rows[i0]["Start_Timestamp"]), None
print("   window:")
for r in rows[i0:i0 + 10]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("     +%8.1f us  dur %7.1f us  idle before %5.1f us  queue %-3s %s" % (
        (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Queue_Id"),
        r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
    prev = e
