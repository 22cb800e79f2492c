#!/bin/bash
# Round 4: split-f16 image swizzle A/B on one box (old = chunk swizzle only; new = conflict-free for the 8-lane store
# groups and the 16-lane read groups), stamps and counters.
set -o pipefail
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r04m}
mkdir -p $O
echo "== tests" && timeout -k 10 600 python -m pytest tests/test_gpu_split.py -q -m gpu -rf > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
echo "== A/B" && (for r in 1 2 3; do for b in kbench kbench_oldswz kbench_swzD kbench_swzA; do echo "-- $b"; timeout -k 10 120 tools/$b 50000 200 4; done; done; for b in kbench_lin kbench_lin_oldswz; do echo "-- $b"; timeout -k 10 120 tools/$b 50000 200 4; done) > $O/ab.txt 2>&1; echo rc=$?; grep -E "^--|variant 4: 0" $O/ab.txt
echo "== stamps" && (for b in kbench_stamps kbench_stamps_oldswz; do echo "-- $b"; timeout -k 10 120 tools/$b 50000 50 4; done) > $O/stamps.txt 2>&1; echo rc=$?
for b in kbench kbench_oldswz; do
echo "== pmc $b" && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_$b -- tools/$b 50000 20 4 > $O/pmc_$b.log 2>&1; echo rc=$?
done
echo done
