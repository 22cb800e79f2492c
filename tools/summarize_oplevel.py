#!/usr/bin/env python3
"""Condense a tools/profile_oplevel.sh output directory into profiles/<tag>_oplevel_N200k.json: per kernel the
rocprofv3 average duration and the FETCH_SIZE / WRITE_SIZE-derived HBM bytes and GB/s against the 8 TB/s peak and the
~6.3 TB/s a streaming copy achieves (MI355X_MICROARCH.md).  Usage: python tools/summarize_oplevel.py <dir> <tag>"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
PEAK, ACHIEVABLE = 8000.0, 6290.0  # GB/s
short = lambda n: n.replace("void ", "").split("(")[0].split("<")[0]
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join("profiles", tag + "_oplevel_N200k_kernel_stats.csv"))
kern = {}
for r in csv.DictReader(open(stats)):
    if "ahv::" in r["Name"]:
        k = kern.setdefault(short(r["Name"]), {"calls": 0, "total_ns": 0.0})
        k["calls"] += int(r["Calls"])
        k["total_ns"] += float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"])
for d, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])) if files else []:
        if r["Counter_Name"] == cname and "ahv::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k in kern:
            kern[k][cname + "_KB_mean"] = sum(v) / len(v)
for k, v in kern.items():
    v["avg_ns"] = v.pop("total_ns") / v["calls"]
    if "FETCH_SIZE_KB_mean" in v and "WRITE_SIZE_KB_mean" in v:
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KB_mean"] + v["WRITE_SIZE_KB_mean"]) * 1024
        v["hbm_GBps"] = v["hbm_bytes_per_launch"] / v["avg_ns"]
        v["frac_of_8TBps_peak"] = v["hbm_GBps"] / PEAK
        v["frac_of_6.29TBps_achievable"] = v["hbm_GBps"] / ACHIEVABLE
out = {"command": "python3 tools/bench_configs.py 3  (N=200000 SO(3) grid; op-level pipeline + fused kernel), "
                  "tools/profile_oplevel.sh: --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own",
       "note": "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B); "
               "kernels launched with several N in this command (warm-ups, the per-pair target feature) are averaged together",
       "peak_GBps": PEAK, "achievable_GBps": ACHIEVABLE, "kernels": kern}
json.dump(out, open(os.path.join("profiles", tag + "_oplevel_N200k.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
