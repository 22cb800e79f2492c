#!/bin/bash
# Round 4: where a 6 250-hypothesis launch (one rank's share at 8 GPUs) spends its time: in-kernel stamps.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z
mkdir -p $O
(for n in 6144 6250 2048 1000; do for v in 3 5; do echo "-- N $n variant $v"; timeout -k 10 120 tools/kbench_stamps $n 300 $v; done; done) > $O/stamps.txt 2>&1; echo rc=$?
grep -E "^--|variant [35]: 0|workgroups|total|idle" $O/stamps.txt
