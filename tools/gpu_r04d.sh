#!/bin/bash
# Round 4, fourth GPU pass: scratch-free kernels, team-tail sweep, low_half A/B, evaluation loop, profile passes.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d
mkdir -p $O
echo "== tests" && timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_estimator.py tests/test_gpu_refine.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== kbench teams" && (for n in 6144 6250 6400 2154 12288 12500; do timeout -k 10 120 tools/kbench $n 300 3; timeout -k 10 120 tools/kbench $n 300 3 0 1; done; timeout -k 10 120 tools/kbench 50000 200 3; timeout -k 10 120 tools/kbench 50000 200 5;  timeout -k 10 120 tools/kbench 50000 200 4) > $O/kbench.txt 2>&1; echo rc=$?
echo "== low_half A/B" && (for r in 1 2; do timeout -k 10 120 tools/kbench 50000 200 3; timeout -k 10 120 tools/kbench_lowhalf 50000 200 3; done; timeout -k 10 300 tools/kbench_bwd; timeout -k 10 300 tools/kbench_bwd_lowhalf; timeout -k 10 300 tools/kbench_bwd; timeout -k 10 300 tools/kbench_bwd_lowhalf) > $O/lowhalf.txt 2>&1; echo rc=$?
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== secondary" && timeout -k 10 900 python3 tools/bench_configs.py pairs 5 shard > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh r04d > $O/profile.log 2>&1; echo rc=$?
echo done
