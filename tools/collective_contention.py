#!/usr/bin/env python3
"""What a small kernel of ANOTHER stream costs the verify step when it needs a compute unit while the scorer's persistent
grid runs -- the situation of the RCCL all-reduce of step i's key beside step i+1's scorer.

A one-rank RCCL group launches no kernel for an all-reduce, and two RCCL ranks cannot share the one GPU of the test box,
so the collective's kernel is emulated: a spin kernel of ~20 us on a side stream (torch.cuda._sleep: one wave, no LDS),
made to depend on the scorer of its step and consumed `lag` steps later by the select launch -- the event structure
torch's ProcessGroupNCCL builds around an async collective.  The scorer's workgroups hold all 512 vector registers of
every SIMD and 159.5 of 160 KiB of LDS, so ANY other wave needs a CU the grid does not occupy.

Sweeps: spare CUs left by the scorer (AHV_SCORE_SPARE_CUS) x how many steps later the result is consumed x with/without
the side kernel.  One JSON object per line (kept as profiles/r04_collective_contention.jsonl).
"""
import importlib, json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ahv = importlib.import_module("3dahv_amd")
ops = ahv.ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
STEPS = 300

g = torch.Generator().manual_seed(0)
W1 = ((torch.rand(32, 384, generator=g) * 2 - 1) / np.sqrt(384.0)).to(dev)
W2 = ((torch.rand(32, 32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
b2 = ((torch.rand(32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
vs, vt = [(torch.randn(1, 16, 8, 8, 8, generator=g) * 1.15).to(dev) for _ in range(2)]
R = torch.from_numpy(ahv.rotations.haar_rotations_np(N, 9)).to(dev)
side = torch.cuda.Stream()


def calibrate_sleep(target_us=20.0):
    """cycles argument of torch.cuda._sleep for ~target_us, measured."""
    cyc = 100_000
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.cuda._sleep(cyc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0  # per call, us
        cyc = max(1, int(cyc * target_us / max(us, 1e-3)))
    return cyc, us


def run(spare, lag, with_side, cyc):
    ring = 4
    keys = [torch.full((1,), -(1 << 63), dtype=torch.int64, device=dev) for _ in range(ring)]
    evs = [torch.cuda.Event() for _ in range(ring)]
    back = [torch.cuda.Event() for _ in range(ring)]
    main = torch.cuda.current_stream()

    def finalize(i):
        if with_side:
            main.wait_event(back[i % ring])
        ops.select_rotation(keys[i % ring], R, reset_key=True)

    def loop(steps):
        for i in range(steps):
            ops.verify_pair(vs, vt, R, W1, W2, b2, want_scores=False, best_key=keys[i % ring], reset_best=False,
                            spare_cus=spare)
            if with_side:
                evs[i % ring].record(main)
                side.wait_event(evs[i % ring])
                with torch.cuda.stream(side):
                    torch.cuda._sleep(cyc)
                    back[i % ring].record(side)
            if i - lag >= 0:
                finalize(i - lag)
        for i in range(max(steps - lag, 0), steps):
            finalize(i)

    loop(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(STEPS)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


with torch.no_grad():
    for _ in range(100):  # clock ramp
        ops.verify_pair(vs, vt, R, W1, W2, b2, want_scores=False)
    torch.cuda.synchronize()
    cyc, us = calibrate_sleep()
    print(json.dumps({"side_kernel": "torch.cuda._sleep", "cycles": cyc, "measured_us": us, "N": N, "steps": STEPS}))
    for rep in range(2):
        for spare in (0, 1, 2, 4):
            row = {"spare_cus": spare, "rep": rep}
            row["ms_no_side_kernel"] = run(spare, 0, False, cyc)
            for lag in (0, 1, 2):
                row["ms_side_kernel_lag%d" % lag] = run(spare, lag, True, cyc)
            print(json.dumps(row))
