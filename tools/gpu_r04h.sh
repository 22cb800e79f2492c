#!/bin/bash
# Round 4 end-of-round style pass: whole GPU suite, bench (driver command + long run + process-group variants), profile
# passes of the bench command, secondary configs, evaluation loop.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r04h}
mkdir -p $O
echo "== pytest" && timeout -k 10 1100 python -m pytest tests -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== forced pg" && AHV_BENCH_FORCE_PG=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_pg.json 2> $O/bench_forced_pg.err; echo rc=$?
echo "== gloo 2 ranks" && timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo rc=$?
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh ${1:-r04h} > $O/profile.log 2>&1; echo rc=$?
echo "== secondary" && timeout -k 10 900 python3 tools/bench_configs.py 3 4 5 shard > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?
echo "== pairs" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs.jsonl 2> $O/pairs.err; echo rc=$?
echo "== train" && timeout -k 10 600 python3 tools/bench_configs.py train train9000 > $O/train.jsonl 2> $O/train.err; echo rc=$?
echo done
