#!/bin/bash
# rocprofv3 of the op-level (materialising, HBM-bound) pipeline at N=200k (BASELINE.json configs[2]).
set -o pipefail
TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_oplevel_$TAG
mkdir -p $OUT
CMD="python3 tools/bench_configs.py 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
ls $OUT/*/*/ | head
