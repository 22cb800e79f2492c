#!/bin/bash
# Round 4, fifth GPU pass: whole GPU suite after the launch-plan / low_half / uploader changes, the evaluation loop.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e
mkdir -p $O
echo "== pytest" && timeout -k 10 1100 python -m pytest tests -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
echo "== pairs" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs.jsonl 2> $O/pairs.err; echo rc=$?; cat $O/pairs.jsonl
echo "== pairs again" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs2.jsonl 2> $O/pairs2.err; echo rc=$?
echo "== train" && timeout -k 10 600 python3 tools/bench_configs.py train > $O/train.jsonl 2> $O/train.err; echo rc=$?
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo done
