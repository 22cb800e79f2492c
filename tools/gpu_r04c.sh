#!/bin/bash
# Round 4, third GPU pass: the whole GPU suite, the bench line (driver command, long run, process-group variants),
# the evaluation loop, rocprofv3 passes of the bench command.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c
mkdir -p $O
echo "== pytest" && timeout -k 10 1100 python -m pytest tests -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -12 $O/pytest.log
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== forced pg (default: stream-ordered, 8 steps per collective)" && AHV_BENCH_FORCE_PG=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_pg_sync_k8.json 2> $O/bench_forced_pg_sync_k8.err; echo rc=$?
echo "== forced pg async k8" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_COLLECTIVE=async timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_pg_async_k8.json 2> $O/bench_forced_pg_async_k8.err; echo rc=$?
echo "== gloo 2 ranks" && timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo rc=$?
echo "== pairs" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs.jsonl 2> $O/pairs.err; echo rc=$?
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh r04c > $O/profile.log 2>&1; echo rc=$?
echo done
