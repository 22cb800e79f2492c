#!/bin/bash
# Per-kernel durations of the scorer's training step (forward + 3 backward kernels) at the reference's
# training size: rocprofv3 --kernel-trace --stats over `tools/bench_configs.py train`.
# Usage (through gpurun): bash tools/profile_train.sh <tag>
set -o pipefail
TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_train_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_configs.py train > $OUT/train.log 2>&1 || exit 1
grep "training scorer step" $OUT/train.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
import collections
# full-size launches only: the same kernels also run at N = 1 for forward_3d2d's backward (a few microseconds)
t = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    if "score_" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in sorted(d.items()):
    big = [x for x in v if x > 100.0]
    if not big:
        continue
    rows.append((k, len(big), sum(big) / max(len(big), 1), min(big), max(big)))
    print("%-48s full-size launches %3d  avg %8.1f us  min %8.1f  max %8.1f" % rows[-1])
with open("$OUT/training_kernels.csv", "w") as o:
    o.write("kernel,full_size_launches,avg_us,min_us,max_us\n")
    for r in rows:
        o.write("%s,%d,%.1f,%.1f,%.1f\n" % r)
PY
