#!/bin/bash
# whole GPU suite, the three scorer instances timed stand-alone, the driver's bench command
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r04fin}
mkdir -p $O
echo "== pytest" && timeout -k 10 1100 python -m pytest tests -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo "== kbench" && (timeout -k 10 120 tools/kbench 50000 200; timeout -k 10 120 tools/kbench 6250 300 5) > $O/kbench.txt 2>&1; echo rc=$?; grep -E "variant [345]: 0|max" $O/kbench.txt
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?; cut -c1-400 $O/bench_20_5.json
echo done
