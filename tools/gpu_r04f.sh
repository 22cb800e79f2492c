#!/bin/bash
# Round 4, sixth GPU pass: XCD-balanced split (tests, kbench A/B, bench), evaluation loop after the uploader fixes.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f
mkdir -p $O
echo "== tests" && timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_refine.py tests/test_gpu_bench_contract.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
echo "== kbench balance A/B" && (for r in 1 2; do timeout -k 10 120 tools/kbench 50000 300 5 0 0 0; timeout -k 10 120 tools/kbench 50000 300 5 0 0 1; timeout -k 10 120 tools/kbench 50000 300 3 0 0 0; timeout -k 10 120 tools/kbench 50000 300 3 0 0 1; done; timeout -k 10 120 tools/kbench 25000 300 5 0 0 0; timeout -k 10 120 tools/kbench 25000 300 5 0 0 1; timeout -k 10 120 tools/kbench 200000 50 5 0 0 0; timeout -k 10 120 tools/kbench 200000 50 5 0 0 1; timeout -k 10 120 tools/kbench 50000 300 4) > $O/kbench_balance.txt 2>&1; echo rc=$?; grep -E "xcd shares|variant [345]: [0-9.]+ ms" $O/kbench_balance.txt | awk '{print}' | tail -40
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== bench 200/20 no balance" && AHV_BENCH_XCD_BALANCE=0 timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20_nobalance.json 2> $O/bench_200_20_nobalance.err; echo rc=$?
echo "== pairs" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs.jsonl 2> $O/pairs.err; echo rc=$?; cat $O/pairs.jsonl
echo done
