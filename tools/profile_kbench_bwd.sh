#!/bin/bash
# PMC counters of the three backward kernels in tools/kbench_bwd.  Usage: bash tools/profile_kbench_bwd.sh <tag> [B] [N]
# (defaults: the reference's CO3D training size, B = 32 per-sample sets of N = 9000 rotations; round 2-5 profiles used 12 3000).
# Separate --pmc passes with --kernel-trace only (gpurun refuses --pmc next to the other trace domains); writes
# gpurun_out/prof_bwd_<tag>/training_pmc_summary.json (copy to profiles/ by hand).
set -o pipefail
TAG=${1:-r06}
B=${2:-32}
N=${3:-9000}
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bwd_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- tools/kbench_bwd $B $N 5 > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- tools/kbench_bwd $B $N 3 > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -- tools/kbench_bwd $B $N 3 > $OUT/pmc2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc3 -- tools/kbench_bwd $B $N 3 > $OUT/pmc3.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- tools/kbench_bwd $B $N 3 > $OUT/pmc_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- tools/kbench_bwd $B $N 3 > $OUT/pmc_write.log 2>&1 || true
python3 - <<PY
import csv, glob, collections, json
NH = $B * $N
FLOPS = {"score_backward_head_kernel": 2101248, "score_backward_w1_kernel": 1703936, "score_backward_volume_kernel": 1703936,
         "score_backward_volume_rmw_kernel": 1703936,
         "score_backward_head_saved_kernel": 393216}   # GEMM2 again from the saved u + dr + dW2 (3 x 2*64*32*32)
SHIPPED = ("score_backward_head_kernel", "score_backward_w1_kernel", "score_backward_w1_reduce_kernel", "score_backward_volume_rmw_kernel")
out = {"B": $B, "N": $N, "hypotheses": NH, "command": "tools/kbench_bwd $B $N", "kernels": {}}
# durations: full-size launches of the kernel trace (the stats pass)
t = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    k = r["Kernel_Name"].split("(")[0].replace("ahv::", "").replace("void ", "")
    if k.startswith("score_backward"):
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    v = sorted(v)[: max(1, len(v) - 2)]  # drop the two slowest (first launches: clock ramp)
    out["kernels"][k] = {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v)}
    if k in FLOPS:
        tf = NH * FLOPS[k] / (sum(v) / len(v)) / 1e6
        out["kernels"][k].update(algorithmic_flops_per_hypothesis=FLOPS[k], tflops=tf, frac_fp32_mfma_peak=tf / 157.3)
for d in ("pmc1", "pmc2", "pmc3", "pmc_fetch", "pmc_write"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("ahv::", "").replace("void ", "")
        if k.startswith("score_backward"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        e = out["kernels"].setdefault(k, {})
        for n, v in c.items():
            e.setdefault("counters_per_launch", {})[n] = sum(v) / len(v)
            e.setdefault("counters_per_hypothesis", {})[n] = sum(v) / len(v) / NH
# the backward the library launches: head + dW1 (+ reduce) + the read-modify-write dV kernel (kbench_bwd also times the LDS-atomic
# dV kernel of rounds 2-5 for A/B: listed under "kernels", not part of the sum)
tot = sum(out["kernels"][k].get("avg_us", 0.0) for k in SHIPPED if k in out["kernels"])
out["backward_us"] = tot
out["backward_kernels"] = [k for k in SHIPPED if k in out["kernels"]]
out["backward_tflops"] = NH * 5509120 / tot / 1e6
out["backward_frac_fp32_mfma_peak"] = out["backward_tflops"] / 157.3
# the training pair's backward (what autograd runs): the head kernel that starts from the forward's saved pre-activations
PAIR = ("score_backward_head_saved_kernel",) + SHIPPED[1:]
if all(k in out["kernels"] for k in PAIR):
    tp = sum(out["kernels"][k].get("avg_us", 0.0) for k in PAIR)
    out["training_pair_backward_us"] = tp
    out["training_pair_backward_kernels"] = list(PAIR)
    out["training_pair_backward_executed_flops_per_hypothesis"] = 393216 + 2 * 1703936
    out["training_pair_backward_frac_fp32_mfma_peak"] = NH * (393216 + 2 * 1703936) / tp / 1e6 / 157.3
for k, e in out["kernels"].items():
    c = e.get("counters_per_launch", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:   # MI355X_MICROARCH.md, gfx950 correction: (2 FETCH + WRITE) KiB
        e["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    cph = e.get("counters_per_hypothesis", {})
    if "SQ_INSTS_MFMA" in cph:
        e["note_mfma"] = "MFMA wave-instructions per hypothesis %.1f" % cph["SQ_INSTS_MFMA"]
out["lds_atomics"] = "score_backward_volume_kernel issues 1 024 ds_add_u64 wave-instructions per hypothesis (512 positions x 8 corners x 16 channels / 64 lanes); there is no PMC counter for LDS atomics alone -- they are inside SQ_INSTS_LDS"
json.dump(out, open("$OUT/training_pmc_summary.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in e.items() if not isinstance(y, dict)})
print("backward", round(tot, 1), "us", round(out["backward_tflops"], 1), "TFLOP/s", round(out["backward_frac_fp32_mfma_peak"], 3))
PY
