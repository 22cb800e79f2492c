#!/bin/bash
# Round 4: the split-f16 image with the row XOR (conflict-free slab reads): parity, speed, LDS conflict counters.
set -o pipefail
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k
mkdir -p $O
echo "== tests" && timeout -k 10 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_verify.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== kbench split" && (for r in 1 2 3; do timeout -k 10 120 tools/kbench 50000 200 4; done; timeout -k 10 120 tools/kbench 50000 200 3) > $O/kbench.txt 2>&1; echo rc=$?; cat $O/kbench.txt
echo "== pmc split" && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_split -- tools/kbench 50000 20 4 > $O/pmc_split.log 2>&1; echo rc=$?
echo done
