#!/bin/bash
# Regression check for the packed-fp32 op_sel hazard (low_half, 3dahv_amd/csrc/ahv_dual.h): builds tools/first_launch.cpp at
# 16 code positions inside the 64-byte fetch lines and with gaps forced between the MFMAs, runs each build twice in fresh
# processes (on the GPU box) and prints one line per build.  Usage: tools/first_launch_sweep.sh build | run
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I3dahv_amd/csrc -Iinclude -Itools"
mkdir -p tools/_fl
if [ "$1" = build ]; then
    for g in "0 1 2 3" "4 5 6 7" "8 9 10 11" "12 13 14 15"; do
        for k in $g; do timeout 900 hipcc $F -DAHV_DIAG_CODE_SHIFT=$k tools/first_launch.cpp -o tools/_fl/shift$k & done
        wait
    done
    timeout 900 hipcc $F '-DAHV_DIAG_MFMA_GAP="s_nop 7"' tools/first_launch.cpp -o tools/_fl/gap_nop7 &
    timeout 900 hipcc $F '-DAHV_DIAG_MFMA_GAP="s_sleep 1"' tools/first_launch.cpp -o tools/_fl/gap_sleep1 &
    timeout 900 hipcc $F '-DAHV_DIAG_MFMA_GAP="s_nop 3\n s_nop 3"' tools/first_launch.cpp -o tools/_fl/gap_nop3x2 &
    wait
    # the kernel WITHOUT the protection (expected to fail; reported, not counted): the hazard itself
    mkdir -p tools/_fl_unprotected
    timeout 900 hipcc $F -DAHV_DIAG_NO_LOW_HALF '-DAHV_DIAG_MFMA_GAP="s_nop 7"' tools/first_launch.cpp -o tools/_fl_unprotected/gap_nop7 &
    timeout 900 hipcc $F -DAHV_DIAG_NO_LOW_HALF '-DAHV_DIAG_MFMA_GAP="s_nop 4"' tools/first_launch.cpp -o tools/_fl_unprotected/gap_nop4 &
    timeout 900 hipcc $F -DAHV_DIAG_NO_LOW_HALF '-DAHV_DIAG_MFMA_GAP="v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop\n v_nop"' tools/first_launch.cpp -o tools/_fl_unprotected/gap_vnop8 &
    wait
    for k in 0 1 2 3; do timeout 900 hipcc $F -DAHV_DIAG_NO_LOW_HALF -DAHV_DIAG_CODE_SHIFT=$k tools/first_launch.cpp -o tools/_fl_unprotected/shift$k & done
    wait
    for k in 4 5 6 7; do timeout 900 hipcc $F -DAHV_DIAG_NO_LOW_HALF -DAHV_DIAG_CODE_SHIFT=$k tools/first_launch.cpp -o tools/_fl_unprotected/shift$k & done
    wait
else
    rc=0
    for b in $(ls tools/_fl | sort -V); do
        for i in 1 2; do
            out=$(timeout -k 10 60 tools/_fl/$b 8192 ABAAB) || rc=1
            echo "$b run $i: $(echo "$out" | tail -1)  first launch: $(echo "$out" | head -1 | sed 's/.*max/max/')"
        done
    done
    echo "--- unprotected builds (-DAHV_DIAG_NO_LOW_HALF): the hazard itself, not counted ---"
    for b in $(ls tools/_fl_unprotected | sort -V); do
        out=$(timeout -k 10 60 tools/_fl_unprotected/$b 8192 ABAAB) || true
        echo "unprotected $b: $(echo "$out" | tail -1)  first launch: $(echo "$out" | head -1 | sed 's/.*max/max/')  third launch: $(echo "$out" | sed -n 3p | sed 's/.*max/max/')"
    done
    exit $rc
fi
