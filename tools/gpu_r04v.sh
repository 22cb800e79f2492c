#!/bin/bash
# Round 4: the split-f16 scorer as shipped (two-plane image, GEMM2 on the XDL pipe, ring depth 2, five resident W1 groups):
# parity suites, timing of the three scorer instances, stamps, LDS counters.
set -o pipefail
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r04v}
mkdir -p $O
echo "== tests" && timeout -k 10 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_verify.py tests/test_gpu_parity.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== kbench" && (for r in 1 2; do for b in kbench; do echo "-- $b"; timeout -k 10 120 tools/$b 50000 200 4; done; done; echo "-- kbench all variants"; timeout -k 10 120 tools/kbench 50000 100) > $O/kbench.txt 2>&1; echo rc=$?; grep -E "^--|variant [345]: 0|max" $O/kbench.txt
echo "== stamps" && (echo "-- kbench_stamps"; timeout -k 10 120 tools/kbench_stamps 50000 50 4) > $O/stamps.txt 2>&1; echo rc=$?; grep -E "^--|ticks|clock" $O/stamps.txt
echo "== pmc" && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_kbench -- tools/kbench 50000 20 4 > $O/pmc_kbench.log 2>&1; echo rc=$?
echo done
