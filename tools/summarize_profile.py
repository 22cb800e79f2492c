#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the small files kept under profiles/.
Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> [kernel-substring [summary-suffix]]"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "score_hypotheses_dual_kernel<false, true>"  # the one-launch verify step
suffix = sys.argv[4] if len(sys.argv) > 4 else ""
out = {}
for d in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"]:
    files = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0]
    for k, v in agg.items():
        out[k] = {"dispatches": len(v), "mean_per_launch": sum(v) / len(v)}
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join("profiles", tag + "_kernel_stats.csv"))
avg_ns = None
for r in csv.DictReader(open(stats)):
    if kern in r["Name"]:
        avg_ns = float(r["AverageNs"]); calls = int(r["Calls"]); kname = r["Name"].split("(")[0]
# Per-launch durations from the raw trace: the chip ramps its clock for ~30 ms after an idle period (the first
# launches run ~18 % slower), so the average over ALL launches mixes ramp and steady state.  Report the plateau:
# launches from the first run of three consecutive durations within 2 % of the level the run ends at (the median
# of its last third) onwards.
def plateau(durs):
    tail = sorted(durs[-max(len(durs) // 3, 1):])
    level = tail[len(tail) // 2]  # where the run ends up
    for i in range(len(durs) - 2):
        if all(abs(d - level) <= 0.02 * level for d in durs[i:i + 3]):
            return i
    return 0


trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
steady = None
if trace:
    durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace[0]))
            if kern in r["Kernel_Name"] and kname in r["Kernel_Name"]]
    if durs:
        k = plateau(durs)
        rest = sorted(durs[k:])
        steady = {"launches_dropped_before_plateau": k, "calls": len(rest), "median_ns": rest[len(rest) // 2],
                  "mean_ns": sum(rest) / len(rest), "min_ns": rest[0], "max_ns": rest[-1],
                  "first_launches_ns": [round(d) for d in durs[:8]]}
summary = {"kernel": kname, "command": "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --skip-strong-scaling (tools/profile_bench.sh)",
           "kernel_trace": {"calls": calls, "average_ns": avg_ns, "steady_state": steady}, "counters": out}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    f, w = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
    summary["hbm_bytes_per_launch"] = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                                       "value": (2 * f + w) * 1024, "fetch_kb": f, "write_kb": w}
json.dump(summary, open(os.path.join("profiles", tag + suffix + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
