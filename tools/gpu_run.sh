#!/bin/bash
# GPU-side passes of a round, one sub-command each (through gpurun: `gpurun -- bash tools/gpu_run.sh <cmd> [tag]`).
# Everything writes under gpurun_out/<tag>/; what is kept goes to profiles/ by hand.
#   teams     bit-identity of team and lone-wave scores + kernel times of small launches, both team orders
#   quick     the whole GPU test suite + kernel times at three sizes
#   ab        two kbench builds alternating on one box
#   stamps    in-kernel stamps of small launches (where a 6 250-hypothesis launch spends its time)
#   tests     the whole GPU test suite
#   bench     bench.py: the driver's command, a long run, the forced-RCCL one-rank run, the 2-rank gloo rehearsal
#   oplevel   rocprofv3 of the op-level (materialising, HBM-bound) pipeline at N = 200 000
#   encoder   encoder marginal costs + times
#   eor       the end-of-round pass (tools/gpu_end_of_round.sh)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
cmd=${1:-tests}
tag=${2:-$cmd}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp

kb() {  # kb <binary> <N> <iters> <variant> [spare] [no_teams]
    echo "-- $1 N $2 variant $4 no_teams ${6:-0}"
    timeout -k 10 120 tools/$1 $2 $3 $4 ${5:-0} ${6:-0}
}

case $cmd in
teams)
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify.py tests/test_gpu_refine.py -x -q -m gpu > $O/pytest.log 2>&1
    echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
    (for n in 6144 6250 2154 12500 25000 50000 1000 300; do for v in 3 5; do
        kb kbench $n 300 $v 0 0; kb kbench $n 300 $v 0 1; kb kbench_teams_last $n 300 $v 0 0
    done; done) > $O/team_head.txt 2>&1
    grep -E "^--|variant [35]: 0|max" $O/team_head.txt | tail -60
    ;;
quick)
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
    (for n in 6250 12500 50000; do for v in 3 5 4; do kb kbench $n 300 $v 0 0; done; done) > $O/kbench.txt 2>&1
    grep -E "^--|variant [345]: 0" $O/kbench.txt
    ;;
ab)  # ab <tag> <binA> <binB>: two kbench builds alternating on one box (boxes differ by ~1 %: never compare across calls)
    A=${3:-kbench}; B=${4:-kbench_noexact}
    (for rep in 1 2; do for n in 50000 6250; do for v in 5 3 4; do kb $A $n 300 $v; kb $B $n 300 $v; done; done; done) > $O/ab.txt 2>&1
    grep -E "^--|variant [345]: 0" $O/ab.txt | awk '/^--/{h=$0;c=0;next} {c++; if(c==3) print h" | "$3" ms"}'
    ;;
stamps)
    (for n in 6144 6250 12500 1000; do for v in 3 5; do kb kbench_stamps $n 300 $v; done; done) > $O/stamps.txt 2>&1
    grep -E "^--|variant [35]: 0|workgroups|total|idle" $O/stamps.txt
    ;;
tests)
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
    ;;
bench)
    timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "20/5 rc=$?"
    timeout -k 10 600 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200_20.json 2> $O/bench_200_20.err; echo "200/20 rc=$?"
    AHV_BENCH_FORCE_PG=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_forced_pg.json 2> $O/bench_forced_pg.err; echo "forced pg rc=$?"
    timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank gloo rc=$?"
    tail -c 1500 $O/bench_20_5.json
    ;;
oplevel)
    timeout -k 10 600 bash tools/profile_oplevel.sh $tag > $O/profile_oplevel.log 2>&1; echo rc=$?; tail -5 $O/profile_oplevel.log
    timeout -k 10 300 python3 tools/bench_configs.py 3 4 5 shard > $O/secondary.jsonl 2> $O/secondary.err; echo rc=$?; cat $O/secondary.jsonl
    ;;
benchtest)
    timeout -k 10 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_bench.log
    ;;
encoder)
    timeout -k 10 300 tools/kbench_enc.bin 1 --each > $O/enc_marginal.txt 2>&1; echo rc=$?
    timeout -k 10 300 python3 tools/bench_configs.py enc enchost > $O/enc.jsonl 2> $O/enc.err; echo rc=$?; cat $O/enc.jsonl
    ;;
encoder32)
    cd /tmp; cd "$GRAFT_REPO_ROOT"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/enc_b32.py 32 > $O/enc_b32.log 2>&1; echo rc=$?; tail -2 $O/enc_b32.log
    python3 - <<PY
import csv, glob
st = glob.glob("$O/trace/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(st)) if "ahv::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("%-70s calls %4s avg %8.1f us  %5.1f %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
    ;;
eor)
    bash tools/gpu_end_of_round.sh $tag
    ;;
*)
    echo "unknown sub-command $cmd"; exit 2
    ;;
esac
echo done
