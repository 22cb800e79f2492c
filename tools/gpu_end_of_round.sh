#!/bin/bash
# End-of-round GPU pass (through gpurun: bash tools/gpu_end_of_round.sh <tag> [a|b]): full tests, bench (driver command + long run +
# the multi-rank rehearsals), rocprof of bench (trace + PMC) -- half "a" --, secondary configs, evaluation loops, encoder / training
# profiles -- half "b"; without the second argument both (more than one 20-minute gpurun call holds).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-eor}
mkdir -p $O
HALF=${2:-ab}
if [[ $HALF == *a* ]]; then
echo "== pytest" && timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo "== bench 20/5" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
echo "== bench 200/20" && timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo rc=$?
echo "== bench forced pg, one lane" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_LANES=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_forced_pg.json 2> $O/bench_forced_pg.err; echo rc=$?
echo "== bench forced pg, two lanes (what a run of 4 ranks and more takes)" && AHV_BENCH_FORCE_PG=1 AHV_BENCH_TWO_LANES_PG=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_forced_pg_two_lanes.json 2> $O/bench_forced_pg_two_lanes.err; echo rc=$?
echo "== profile bench" && timeout -k 10 900 bash tools/profile_bench.sh ${1:-eor} > $O/profile.log 2>&1; echo rc=$?
echo "== bench 2-rank gloo (rehearsal on one GPU; two lanes forced: at 2 ranks the default is one)" && AHV_BENCH_TWO_LANES_MAX_N=25000 timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo rc=$?
echo "== bench 4-rank gloo (rehearsal on one GPU: the default lane choice of 4 ranks -- two lanes, two groups per rank)" && timeout -k 10 700 python3 bench.py --gpus 4 --backend gloo --steps 40 --warmup 8 --no-cpu-baseline > $O/bench_4rank_gloo.json 2> $O/bench_4rank_gloo.err; echo rc=$?
fi
if [[ $HALF == *b* ]]; then
echo "== secondary" && timeout -k 10 600 python3 tools/bench_configs.py 3 4 5 shard > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?
echo "== pairs" && timeout -k 10 600 python3 tools/bench_configs.py pairs > $O/pairs.jsonl 2> $O/pairs.err; echo rc=$?
echo "== option A / B under the reference's conditions" && timeout -k 10 300 python3 tools/bench_configs.py optiona > $O/optiona.jsonl 2> $O/optiona.err; echo rc=$?
echo "== enc" && timeout -k 10 300 python3 tools/bench_configs.py enc enchost > $O/enc.jsonl 2> $O/enc.err; echo rc=$?
echo "== train" && timeout -k 10 600 python3 tools/bench_configs.py train train9000 trainstep > $O/train.jsonl 2> $O/train.err; echo rc=$?
echo "== unchanged infoNCE lines (training mode)" && timeout -k 10 600 python3 tools/bench_configs.py trainlines > $O/trainlines.jsonl 2> $O/trainlines.err; echo rc=$?
echo "== kbench_enc" && timeout -k 10 300 tools/kbench_enc.bin 1 --each > $O/enc_marginal.txt 2>&1; echo rc=$?
echo "== kbench_bwd" && timeout -k 10 200 tools/kbench_bwd 32 9000 5 > $O/kbench_bwd.txt 2>&1; echo rc=$?
echo "== profile backward (PMC)" && timeout -k 10 600 bash tools/profile_kbench_bwd.sh ${1:-eor} 32 9000 > $O/profile_bwd.log 2>&1; echo rc=$?
echo "== profile train" && timeout -k 10 600 bash tools/profile_train.sh ${1:-eor} > $O/profile_train.log 2>&1; echo rc=$?
echo "== profile encoder" && timeout -k 10 600 bash tools/profile_encoder.sh ${1:-eor} > $O/profile_encoder.log 2>&1; echo rc=$?
echo "== encoder B = 32 kernel stats" && bash tools/gpu_run.sh encoder32 ${1:-eor}_enc32 > $O/enc32.log 2>&1; echo rc=$?
echo "== profile op-level" && timeout -k 10 600 bash tools/profile_oplevel.sh ${1:-eor} > $O/profile_oplevel.log 2>&1; echo rc=$?
fi
# raw traces are large: keep the stats and counter summaries only
find gpurun_out -name "*_kernel_trace.csv" -size +3M -delete 2>/dev/null
find gpurun_out -name "*counter_collection.csv" -size +3M -delete 2>/dev/null
echo done
