#!/usr/bin/env python3
"""bench.py -- rotation hypotheses scored per second (BASELINE.json metric) on MI355X.

One "step" = the reference's per-pair hot loop (test_co3d.py:137-146) for ONE synthetic image
pair with N_hyp = 50 000 hypotheses on each GPU (BASELINE.json configs[1]):
    forward_3d2d(vol_tgt)                       1 small launch   (test_co3d.py:141)
    fused rotate + forward_3d2d + score + max   1 launch         (test_co3d.py:137-145)
    [N>1: all-reduce(max) of the packed key over RCCL]
    unpack key, gather R_pred = proposals[idx]                    (test_co3d.py:145-146)
Inputs are resident in HBM before the timed region.  N > 1 shards the hypothesis axis: every
rank scores its own 50 000 (weak scaling), the only exchange is the 8-byte key all-reduce.

Prints ONE JSON line on rank 0 (see the task contract), with `roofline` for the fused kernel
(fp32 MFMA bound) and `cpu_baseline` (the reference's torch-CPU op sequence on the host cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_HYP = 50_000
FLOPS_PER_HYP = 1_839_104      # trilinear 131072 + GEMM1 1572864 + GEMM2 131072 + dot 4096
HBM_BYTES_PER_HYP = 40         # 36 B of R in + 4 B of score out (fused kernel)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak


def synth_inputs(ahv, dev, rank):
    """SURVEY.md 8(d) cfg 2: vol ~ N(0, 1.15^2); torch-default conv init for the head; Haar R."""
    g = torch.Generator().manual_seed(0)
    vol_src = torch.randn(1, 16, 8, 8, 8, generator=g) * 1.15
    vol_tgt = torch.randn(1, 16, 8, 8, 8, generator=g) * 1.15
    W1 = (torch.rand(32, 384, generator=g) * 2 - 1) / np.sqrt(384.0)
    W2 = (torch.rand(32, 32, generator=g) * 2 - 1) / np.sqrt(32.0)
    b2 = (torch.rand(32, generator=g) * 2 - 1) / np.sqrt(32.0)
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(N_HYP, seed=1000 + rank))
    return [t.to(dev).contiguous() for t in (vol_src, vol_tgt, W1, W2, b2, R)]


def cpu_baseline(vol_src, vol_tgt, W1, W2, b2, R, budget_s=20.0):
    """The reference's op sequence with stock torch CPU ops (oracle/torch_ref.py) on the host cores:
    the same workload in chunks of 1 000 hypotheses (fastest variant found in the survey, BASELINE.md
    section 2), stopped after ~budget_s seconds of CPU work.  Threads = the box's CPU share for one
    GPU (16) or fewer; override with AHV_CPU_THREADS."""
    from oracle import torch_ref
    cores = int(os.environ.get("AHV_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    vs, vt, Rc, w1, w2, bb = [t.cpu() for t in (vol_src, vol_tgt, R, W1, W2, b2)]
    torch_ref.score_hypotheses(vs, vt, Rc[:2000], w1, w2, bb, chunk=1000)  # warm-up
    parts, done = [], 0
    t0 = time.perf_counter()
    while done < N_HYP and time.perf_counter() - t0 < budget_s:
        s, _, _ = torch_ref.score_hypotheses(vs, vt, Rc[done:done + 1000], w1, w2, bb)
        parts.append(s)
        done += s.shape[1]
    dt = time.perf_counter() - t0
    scores = torch.cat(parts, dim=1)
    return {"value": done / dt, "unit": "hypotheses/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "first %d of the %d hypotheses of the same workload (B=1), chunks of 1000, torch %s CPU "
                      "ops, %.1f s" % (done, N_HYP, torch.__version__, dt)}, scores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--score-variant", type=int, default=3, help="fused-kernel variant to time (3 = all-fp32 dual, "
                    "the default and the headline; 4 = split-f16 GEMM1, opt-in, reported beside it)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank logic on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    dev = torch.device("cuda", local_rank % max(ndev, 1))  # ranks share a GPU only in a gloo rehearsal
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ahv = importlib.import_module("3dahv_amd")
    ops, adist = ahv.ops, ahv.dist
    lib = ahv._lib.load()  # fails loudly without the HIP library
    lib.ahv_set_option(b"score_variant", args.score_variant)

    vol_src, vol_tgt, W1, W2, b2, R = synth_inputs(ahv, dev, rank)
    n_offset = rank * N_HYP
    ring = 4  # key buffers in flight: step i's tiny all-reduce overlaps step i+1's kernel
    keys = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ring)]
    pending = {}
    out = {}

    def finalize(i):
        """all-reduce result -> (best score, global idx) -> R_pred = proposals[idx] (test_co3d.py:145-146)."""
        key = keys[i % ring]
        work = pending.pop(i, None)
        if work is not None:
            work.wait()  # stream-level dependency only, the host does not block
            key.bitwise_xor_(adist._SIGN)
        # unpack + gather in one launch; with sharding the owner rank holds the winning row, the others get zeros
        out["best"], out["idx"], out["R_pred"] = ops.select_rotation(key, R, n_offset=n_offset)

    def step(i, ev=None):
        feat_tgt = ops.forward_3d2d(vol_tgt, W1, W2, b2)
        key = keys[i % ring]
        if ev is not None:
            ev[0].record()
        ops.score_hypotheses(vol_src, feat_tgt, R, W1, W2, b2, n_offset=n_offset, want_scores=False, best_key=key,
                             reset_best=True)
        if ev is not None:
            ev[1].record()
        if world > 1:
            key.bitwise_xor_(adist._SIGN)  # unsigned order -> signed order for ReduceOp.MAX
            pending[i] = dist.all_reduce(key, op=dist.ReduceOp.MAX, async_op=True)
        if i > 0:
            finalize(i - 1)  # finish the previous step while this step's kernel runs

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    if args.warmup:
        finalize(args.warmup - 1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    finalize(args.steps - 1)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))  # fused kernel, HIP events on its stream

    # correctness of what was timed: the key equals torch.max over the materialised scores (all ranks)
    feat_tgt = ops.forward_3d2d(vol_tgt, W1, W2, b2)
    scores, key = ops.score_hypotheses(vol_src, feat_tgt, R, W1, W2, b2, n_offset=n_offset)
    lv, li = torch.max(scores, dim=1)
    cand = torch.stack([lv.double(), (li + n_offset).double()], dim=1)
    if world > 1:
        allc = [torch.zeros_like(cand) for _ in range(world)]
        dist.all_gather(allc, cand)
        cand = torch.cat(allc)
    gbest = cand[torch.argmax(cand[:, 0])]
    assert int(out["idx"].item()) == int(gbest[1].item()), (out["idx"], gbest)
    assert float(out["best"].item()) == float(gbest[0].item())

    if rank == 0:
        total_hyp = N_HYP * world * args.steps
        value = total_hyp / dt
        achieved = FLOPS_PER_HYP * N_HYP / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 --pmc (see profiles/README.md)
            with open(tpath) as f:
                traffic = json.load(f).get("fused_hbm_bytes_per_launch")
        res = {
            "metric": "rotation hypotheses scored/sec (B=1)", "value": value, "unit": "hypotheses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "CO3D pair (BASELINE.json configs[1]): B=1, N_hyp=50000 Haar rotations per GPU, "
                                   "source volume 16x8x8x8 (P=512 voxel sites x 16 ch), head 384->32->32, 64 positions",
                       "n_hyp_per_gpu": N_HYP, "n_hyp_total": N_HYP * world,
                       "parallelism": "hypothesis axis sharded x%d, 8-byte key all-reduce(max)" % world,
                       "step": "forward_3d2d(tgt) + fused score/argmax + select (unpack + gather R_pred)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": "score_hypotheses_dual_kernel", "kernel_ms": kern_ms,
                         "algorithmic_flops_per_launch": FLOPS_PER_HYP * N_HYP,
                         "algorithmic_hbm_bytes_per_launch": HBM_BYTES_PER_HYP * N_HYP},
        }
        if args.score_variant != 3:
            res["roofline"]["kernel"] = "score_variant %d" % args.score_variant
        if args.score_variant == 4:  # opt-in kernel: priced against the f16 matrix peak (16 x the fp32 one)
            res["dtype"] = "f16 hi/lo split products, f32 accumulate"
            res["roofline"].update(peak=16 * PEAK_F32_MFMA_TFLOPS, frac=achieved / (16 * PEAK_F32_MFMA_TFLOPS),
                                   note="GEMM1 runs 3 f16 MFMA products per algorithmic MAC; the kernel is bound by "
                                        "LDS bandwidth (trilinear gather), not by the matrix pipe")
        if world == 1 and args.score_variant == 3:
            # The opt-in split-f16 kernel on the same inputs, reported beside the fp32 headline (never as `value`).
            lib.ahv_set_option(b"score_variant", 4)
            s4, k4 = ops.score_hypotheses(vol_src, feat_tgt, R, W1, W2, b2)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for a, b in ev:
                a.record()
                ops.score_hypotheses(vol_src, feat_tgt, R, W1, W2, b2, want_scores=False, best_key=keys[0], reset_best=True)
                b.record()
            torch.cuda.synchronize()
            lib.ahv_set_option(b"score_variant", 3)
            ms4 = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            res["split_f16_kernel"] = {
                "note": "score_variant 4 (opt-in): GEMM1 as 3 f16 MFMA products of hi/lo split operands, f32 accumulate",
                "kernel_ms": ms4, "hypotheses_per_s_kernel_only": N_HYP / (ms4 * 1e-3),
                "max_abs_score_diff_vs_f32_kernel": float((s4 - scores).abs().max().item()),
                "same_argmax": bool(torch.equal(ops.unpack_best(k4)[1], ops.unpack_best(key)[1]))}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_scores = cpu_baseline(vol_src, vol_tgt, W1, W2, b2, R)
            res["cpu_baseline"] = cb
            # same inputs, same answers on the sampled part: scores <= 1e-4 relative, arg-max exact
            n = cpu_scores.shape[1]
            gpu_part = scores[:, :n].cpu()
            rel = ((gpu_part - cpu_scores).abs() / cpu_scores.abs().clamp_min(1e-2)).max().item()
            assert rel < 1e-4, rel
            assert int(torch.argmax(cpu_scores, dim=1).item()) == int(torch.argmax(gpu_part, dim=1).item())
            res["cpu_baseline"]["gpu_vs_cpu_max_rel_err"] = rel
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
