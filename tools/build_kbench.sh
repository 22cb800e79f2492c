#!/bin/bash
# Builds the developer micro-benchmarks (plain and with in-kernel stamps) next to their source.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I3dahv_amd/csrc -Iinclude -Itools"
timeout 900 hipcc $F tools/kbench.cpp -o tools/kbench
timeout 900 hipcc $F -DAHV_STAMPS tools/kbench.cpp -o tools/kbench_stamps
# conflict-free bound of the gather (wrong results; timing and stamps only)
timeout 900 hipcc $F -DAHV_DIAG_LINEAR_GATHER tools/kbench.cpp -o tools/kbench_lin
timeout 900 hipcc $F -DAHV_STAMPS -DAHV_DIAG_LINEAR_GATHER tools/kbench.cpp -o tools/kbench_stamps_lin
# what low_half() (the packed-fp32 op_sel protection) costs the fp32 scorers, which ship with it since round 5
timeout 900 hipcc $F -DAHV_DIAG_NO_FP32_LOW_HALF tools/kbench.cpp -o tools/kbench_nolowhalf
# what the PRESENCE of the exact (non-finite) path costs finite inputs; team rounds behind the main rounds as in round 4
timeout 900 hipcc $F -DAHV_DIAG_NO_EXACT tools/kbench.cpp -o tools/kbench_noexact
timeout 900 hipcc $F -DAHV_DIAG_TEAMS_LAST tools/kbench.cpp -o tools/kbench_teams_last
timeout 900 hipcc $F -DAHV_DIAG_NO_FP32_LOW_HALF tools/kbench_bwd.cpp -o tools/kbench_bwd_nolowhalf   # the backward kernels WITHOUT it (they ship with it)
timeout 900 hipcc $F tools/kbench_bwd.cpp -o tools/kbench_bwd
timeout 900 hipcc $F -DAHV_BWD_DU_AMAX tools/kbench_bwd.cpp -o tools/kbench_bwd_atomics   # + the LDS-atomic dV kernel of rounds 2-5 for A/B
# forward_2d3d from a replayed hipGraph with per-launch marginal costs (tools/gpu_run.sh encoder)
timeout 900 hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAHV_ENC_PROBE -I3dahv_amd/csrc -Iinclude tools/kbench_enc.cpp -o tools/kbench_enc.bin
