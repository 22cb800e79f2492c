#!/bin/bash
# Round 4, second GPU pass: fixed tests, kernel A/B after the team pipeline, the process-group variants of bench.py
# (steps per collective, async / stream-ordered), the evaluation loop, kernel-trace timeline of the forced-PG run.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b
mkdir -p $O
echo "== tests" && timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_refine.py tests/test_gpu_bench_contract.py tests/test_gpu_estimator.py tests/test_gpu_encoder.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -12 $O/pytest.log
echo "== kbench" && (for n in 50000 6250 25000 1000; do timeout -k 10 120 tools/kbench $n 200 3; timeout -k 10 120 tools/kbench $n 200 5; timeout -k 10 120 tools/kbench $n 200 3 0 1; done) > $O/kbench.txt 2>&1; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
for mode in async sync; do for k in 1 8; do
  echo "== forced pg $mode k $k" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_COLLECTIVE=$mode AHV_BENCH_STEPS_PER_COLLECTIVE=$k timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_pg_${mode}_k${k}.json 2> $O/bench_forced_pg_${mode}_k${k}.err; echo rc=$?
done; done
echo "== gloo 2 ranks" && timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo rc=$?
echo "== secondary" && timeout -k 10 600 python3 tools/bench_configs.py 5 shard pairs > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "== timeline forced pg" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_STEPS_PER_COLLECTIVE=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_pg_k1 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/trace_pg_k1.log 2>&1; echo rc=$?
echo "== timeline single" && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_single -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/trace_single.log 2>&1; echo rc=$?
find $O -name "*kernel_trace.csv" | head
echo done
