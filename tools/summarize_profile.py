#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the small files kept under profiles/.
Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> [kernel-substring [summary-suffix]]"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "score_hypotheses"
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
summary = {"kernel": kname, "command": "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline (tools/profile_bench.sh)",
           "kernel_trace": {"calls": calls, "average_ns": avg_ns}, "counters": out}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    f, w = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
    summary["hbm_bytes_per_launch"] = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                                       "value": (2 * f + w) * 1024, "fetch_kb": f, "write_kb": w}
json.dump(summary, open(os.path.join("profiles", tag + suffix + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
