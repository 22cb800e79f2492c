#!/bin/bash
# Round-3 GPU pass A: tests, bench (driver command + long run), rocprof of bench, scorer segments, encoder host time.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a
mkdir -p $O
echo "== pytest" && timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== bench forced pg" && AHV_BENCH_FORCE_PG=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_forced_pg.json 2> $O/bench_forced_pg.err; echo rc=$?
echo "== segments"
for b in kbench kbench_lin kbench_stamps kbench_stamps_lin; do echo "--- $b" >> $O/segments.txt; timeout -k 10 120 tools/$b 50000 40 3 >> $O/segments.txt 2>&1; done
echo "== enc host" && timeout -k 10 300 python3 tools/bench_configs.py enchost > $O/enchost.jsonl 2> $O/enchost.err; echo rc=$?
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh r03a > $O/profile.log 2>&1; echo rc=$?
echo "== profile oplevel" && timeout -k 10 900 bash tools/profile_oplevel.sh r03 > $O/profile_oplevel.log 2>&1; echo rc=$?
echo done
