#!/usr/bin/env python3
"""Why does the captured configs[4] step replay slower than it runs eagerly?  

Two modes:

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/graph_timeline.py run
        runs four legs of the coarse-to-fine step (10 000 + 1 000 hypotheses, B = 1, inputs in place), each 300 steps
        behind a marker kernel of its own (a compose launch with a recognisable N): five launches eager / five launches
        from a hipGraph / one launch eager / one launch from a hipGraph; also a leg that captures EIGHT steps into one
        graph (40 kernel nodes, one hipGraphLaunch).  Prints wall-clock us per step of every leg (host timer around
        synchronize: what a caller sees).

    python3 tools/graph_timeline.py summarize <kernel_trace.csv>
        per leg: step period (first kernel of a step to first kernel of the next), the kernels' own time per step, the
        idle time between kernels INSIDE a step and the idle time BETWEEN the last kernel of a step and the first of the
        next -- the place a graph launch differs from five plain launches.
"""
import csv
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LEGS = [("eager_5", False, False, 1), ("graph_5", False, True, 1), ("eager_1", True, False, 1), ("graph_1", True, True, 1),
        ("graph_5x8", False, True, 8), ("graph_1x8", True, True, 8)]
# AHV_GTL_PG=1: a one-rank RCCL group, the five-launch step with its two key all-reduces issued (force_collectives): what a
# rank of a multi-GPU run enqueues per step, eagerly and from the graph (the collectives captured with the kernels)
PG = os.environ.get("AHV_GTL_PG", "0") == "1"
if PG:
    LEGS = [("eager_5_rccl", False, False, 1), ("graph_5_rccl", False, True, 1), ("graph_5x8_rccl", False, True, 8)]
STEPS = 320


def run():
    import numpy as np
    import torch
    ahv = importlib.import_module("3dahv_amd")
    ops = ahv.ops
    dev = torch.device("cuda:0")
    if PG:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = torch.Generator().manual_seed(0)
    W1 = ((torch.rand(32, 384, generator=g) * 2 - 1) / np.sqrt(384.0)).to(dev)
    W2 = ((torch.rand(32, 32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
    b2 = ((torch.rand(32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
    vol = (torch.randn(2, 1, 16, 8, 8, 8, generator=g) * 1.15).to(dev)
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(10_000, 11)).to(dev)
    marker_out = torch.empty(4096, dtype=torch.float32, device=dev)
    # warm the chip: the legs must not sit in the clock ramp
    key = torch.full((1,), -(1 << 63), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            ops.verify_pair(vol[0], vol[1], R, W1, W2, b2, want_scores=False, best_key=key, reset_best=False)
        torch.cuda.synchronize()
    for k, (name, fused, use_graph, per_graph) in enumerate(LEGS):
        c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, max_angle_deg=10.0, batch=1, use_graph=use_graph and per_graph == 1,
                                      fused=fused, force_collectives=PG)
        c2f.buffers[0].copy_(vol[0])
        c2f.buffers[1].copy_(vol[1])
        if per_graph > 1:   # several steps in ONE captured graph
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    c2f()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(per_graph):
                    out = c2f()
            step = graph.replay
        else:
            step = c2f
        for _ in range(16 // per_graph + 1):
            step()
        torch.cuda.synchronize()
        # the marker: an so3_grid launch of 1000 + k points (no other launch of this run has that name)
        ops.so3_grid(1000 + k, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS // per_graph):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"leg": name, "launches_per_step": 1 if fused else 5, "graph": use_graph, "steps_per_graph": per_graph,
                          "us_per_step_wall": dt / STEPS * 1e6}), flush=True)
        ops.so3_grid(2000 + k, dev)
        torch.cuda.synchronize()
    if PG:
        dist.destroy_process_group()


def summarize(path):
    import numpy as np
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "so3_grid" in r["Kernel_Name"]]
    assert len(marks) == 2 * len(LEGS), (len(marks), "marker launches")
    for k, (name, fused, use_graph, per_graph) in enumerate(LEGS):
        seg = rows[marks[2 * k] + 1:marks[2 * k + 1]]
        per_step = len(seg) // STEPS      # 1 / 5 kernels per step, plus whatever the collectives launch
        assert len(seg) == STEPS * per_step and per_step >= (1 if fused else 5), (name, len(seg))
        s = np.array([int(r["Start_Timestamp"]) for r in seg], dtype=np.int64).reshape(STEPS, per_step)
        e = np.array([int(r["End_Timestamp"]) for r in seg], dtype=np.int64).reshape(STEPS, per_step)
        sel = slice(STEPS // 4, STEPS - 8)
        period = np.diff(s[:, 0])[sel] / 1e3
        busy = (e - s).sum(axis=1)[sel] / 1e3
        inside = (s[:, 1:] - e[:, :-1]).sum(axis=1)[sel] / 1e3 if per_step > 1 else np.zeros(1)
        between = (s[1:, 0] - e[:-1, -1])[sel] / 1e3
        if per_graph > 1:   # the boundary between two graph launches falls behind every per_graph-th step
            idx = np.arange(STEPS - 1)[sel]
            edge = between[(idx % per_graph) == per_graph - 1]
            mid = between[(idx % per_graph) != per_graph - 1]
            extra = "  [between steps inside a graph %.2f us, across two graph launches %.2f us]" % (np.median(mid), np.median(edge))
        else:
            extra = ""
        names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ahv::", "")[:40] for r in seg[:per_step]]
        print("%-10s period %7.2f us | kernels %7.2f | idle inside a step %5.2f | idle between steps %5.2f (mean %5.2f)%s" % (
            name, np.median(period), np.median(busy), np.median(inside), np.median(between), between.mean(), extra))
        if not (use_graph and per_graph == 1) and per_graph == 1:
            durs = np.median((e - s)[sel], axis=0) / 1e3
            print("           kernels: " + ", ".join("%s %.1f" % (n, d) for n, d in zip(names, durs)))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        run()
