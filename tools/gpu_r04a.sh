#!/bin/bash
# Round 4, first GPU pass (through gpurun: bash tools/gpu_r04a.sh): new parity tests first, kernel A/B numbers, the
# bench line, the forced-process-group variants and the collective-contention experiment.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a
mkdir -p $O
echo "== new tests" && timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_boundary.py tests/test_gpu_split.py tests/test_gpu_graph_replay.py -q -m gpu -rf > $O/pytest_new.log 2>&1; echo "rc=$?" | tee -a $O/pytest_new.log; tail -15 $O/pytest_new.log
echo "== kbench" && (for n in 50000 6250 12500 25000 1000 10000; do timeout -k 10 120 tools/kbench $n 50 3; timeout -k 10 120 tools/kbench $n 50 5; timeout -k 10 120 tools/kbench $n 50 3 0 1; done; timeout -k 10 120 tools/kbench 50000 50 3 2; timeout -k 10 120 tools/kbench 50000 50 5 2; timeout -k 10 120 tools/kbench 50000 50 4) > $O/kbench.txt 2>&1; echo rc=$?; grep -c variant $O/kbench.txt
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
for lag in 0 1 2; do for spare in 0 2; do
  echo "== forced pg lag $lag spare $spare" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_FINALIZE_LAG=$lag AHV_BENCH_SPARE_CUS=$spare timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_pg_lag${lag}_spare${spare}.json 2> $O/bench_forced_pg_lag${lag}_spare${spare}.err; echo rc=$?
done; done
echo "== contention" && timeout -k 10 600 python3 tools/collective_contention.py > $O/collective_contention.jsonl 2> $O/collective_contention.err; echo rc=$?
echo "== secondary" && timeout -k 10 600 python3 tools/bench_configs.py 5 shard > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?
echo "== rest of the tests" && timeout -k 10 1100 python -m pytest tests -q -m gpu -rf --deselect tests/test_gpu_verify.py --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_refine.py --deselect tests/test_gpu_boundary.py --deselect tests/test_gpu_split.py --deselect tests/test_gpu_graph_replay.py > $O/pytest_rest.log 2>&1; echo "rc=$?" | tee -a $O/pytest_rest.log; tail -15 $O/pytest_rest.log
echo done
