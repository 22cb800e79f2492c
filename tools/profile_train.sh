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
for r in csv.DictReader(open(f)):
    if "score_" in r["Name"]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
