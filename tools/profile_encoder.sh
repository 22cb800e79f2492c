#!/bin/bash
# Per-kernel durations and the timeline of ONE forward_2d3d at B = 1: rocprofv3 --kernel-trace over tools/bench_configs.py enc.
# Usage (through gpurun): bash tools/profile_encoder.sh <tag>
set -o pipefail
TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_encoder_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_configs.py enc > $OUT/enc.log 2>&1 || exit 1
grep "encoder forward_2d3d" $OUT/enc.log
python3 - <<PY
import csv, glob, collections
t = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nchw_to_tokens" in r["Kernel_Name"]]
# one B = 1 forward in the middle of the eager timing loop (grid of nchw_to_tokens tells B)
segs = [(a, b) for a, b in zip(idx, idx[1:]) if 30 <= b - a <= 90]
a, b = segs[len(segs) // 2]
seg = rows[a:b]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
print("launches %d  sum of durations %.1f us  span %.1f us  mean gap %.2f us (profiler attached)" % (len(seg), sum(dur), (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3, sum(gaps) / len(gaps)))
agg = collections.defaultdict(list)
for r, d in zip(seg, dur):
    agg[r["Kernel_Name"].split("(")[0]].append(d)
with open("$OUT/encoder_kernels.csv", "w") as o:
    o.write("kernel,launches_per_forward,avg_us,total_us\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        o.write("%s,%d,%.2f,%.1f\n" % (k, len(v), sum(v) / len(v), sum(v)))
        print("%-48s x%2d avg %6.2f us total %6.1f us" % (k[-48:], len(v), sum(v) / len(v), sum(v)))
PY
