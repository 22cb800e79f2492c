#!/bin/bash
# Round-3 GPU pass B: full tests, bench, rocprof of bench (trace + PMC), backward kernels.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== kbench_bwd" && timeout -k 10 200 tools/kbench_bwd 12 3000 > $O/kbench_bwd.txt 2>&1; tail -12 $O/kbench_bwd.txt
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh r03b > $O/profile.log 2>&1; echo rc=$?
echo done
