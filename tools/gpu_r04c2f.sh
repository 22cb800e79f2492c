#!/bin/bash
# Round 4: the one-launch coarse-to-fine step (ahv_coarse_to_fine_f32): parity tests, configs[4] timing, kernel trace.
set -o pipefail
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r04c2f}
mkdir -p $O
echo "== tests" && timeout -k 10 600 python -m pytest tests/test_gpu_refine.py -x -q -m gpu -rf > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc" | tee -a $O/pytest.log; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
echo "== configs[4]" && timeout -k 10 600 python3 tools/bench_configs.py 5 > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?; cut -c1-330 $O/secondary.jsonl
echo "== kernel trace" && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_configs.py 5 > $O/trace.log 2>&1; echo rc=$?
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
echo done
