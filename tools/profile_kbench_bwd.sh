#!/bin/bash
# PMC counters of the three backward kernels in tools/kbench_bwd (B = 12, N = 3000).  Usage: bash tools/profile_kbench_bwd.sh <tag>
set -o pipefail
TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bwd_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- tools/kbench_bwd 12 3000 3 > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- tools/kbench_bwd 12 3000 3 > $OUT/pmc2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "score_backward" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k)
        for n, v in c.items():
            print("   %-24s %16.0f   per hypothesis %12.1f" % (n, sum(v) / len(v), sum(v) / len(v) / 36000))
PY
