"""bench.py keeps the driver's contract: one JSON line with the agreed fields (short run, no CPU baseline)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["n_hyp_total"] == 50000 and d["config"]["n_hyp_per_gpu"] == 50000
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    # value = hypotheses of all steps / wall time; consistent with ms_per_step
    assert abs(d["value"] - 50000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert r["algorithmic_hbm_bytes_per_launch"] == 36 * 50000          # want_scores=False launch: R only
    # HBM traffic: the committed PMC figure, reported only while it belongs to THIS kernel (source hash) -- else null + why
    import hashlib
    h = hashlib.sha256()
    for name in ("ahv_score.hip", "ahv_device.h", "ahv_dual.h", "ahv_team.h", "ahv_exact.h", "ahv_split.h"):
        h.update(open(os.path.join(REPO, "3dahv_amd", "csrc", name), "rb").read())
    tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
    ts = r["traffic_source"]
    if tj.get("kernel_src_sha") == h.hexdigest():
        assert r["traffic"] == tj["fused_hbm_bytes_per_launch"] >= 36 * 50000
        assert ts["kernel_src_sha"] == h.hexdigest() and ts["source"] == tj["source"] and "source_commit" in ts
        assert "not measured in this run" in ts["note"]
    else:
        assert r["traffic"] is None and "stale" in ts and "re-run tools/profile_bench.sh" in ts["stale"]
    assert 1.0 < r["shader_clock_ghz"] <= 2.5                             # measured live under this kernel
    assert r["frac_at_delivered_clock"] >= r["frac"] - 1e-9
    assert "ms" not in d["config"] and d["config"]["backend"] == "single process"
    # the time-based pre-warm ran (>= 60 ms of the same step) and is disclosed; the per-step list is complete
    assert d["prewarm_ms"] >= 60.0 and d["prewarm_steps"] >= 8
    # 5 steps = one (partial) group of launches = one event bracket INSIDE the timed region, ahead of the group's select:
    # the kernel time can never exceed the step time it is part of
    assert len(r["kernel_ms_per_bracket"]) == 1 and "(1 brackets, 5 launches)" in r["kernel_ms_is"]
    assert "INSIDE the timed region" in r["kernel_ms_is"] and r["achieved_is"] == "algorithmic flops per launch / kernel_ms"
    assert r["kernel_ms"] <= d["ms_per_step"]
    assert 0.5 * r["kernel_ms"] < r["hypothesis_loop_ms_all_launches_median"] <= r["kernel_ms"]
    assert r["kernel_ms_min"] <= r["kernel_ms_median"] <= max(r["kernel_ms_per_bracket"]) + 1e-4   # (the list is rounded to 4 places)
    assert abs(r["kernel_ms_mean"] - r["kernel_ms"]) < 1e-12
    # one rank, one lane, and which GPU it was
    assert d["config"]["lanes"] == 1 and len(d["config"]["ranks"]) == 1
    who = d["config"]["ranks"][0]
    assert who["rank"] == 0 and who["device_index"] == 0 and who["gcn_arch"].startswith("gfx950") and who["pci_bus_id"]
    # the reference's literal cadence (select after every pair) is a top-level figure, never better than the headline's
    assert d["ms_per_step_select_every_step"] == d["secondary"]["n50k_b1_collective_per_step"]["ms_per_step"]
    assert d["value_select_every_step"] == d["secondary"]["n50k_b1_collective_per_step"]["hypotheses_per_s"]
    assert 0.97 * d["ms_per_step"] < d["ms_per_step_select_every_step"] < d["ms_per_step"] + 0.03
    assert "TIMED launches" in r["shader_clock_source"]
    assert out.stdout.strip() == lines[0]                                  # stdout = the one JSON line, nothing else
    assert r["kernel"].startswith("score_hypotheses_dual_kernel<false, true>")   # the one-launch verify step
    # the secondary records of the same run
    sec = d["secondary"]
    a, b, w, two = (sec[k] for k in ("n50k_b1_collective_per_step", "configs3_b32_n50k", "weak_n50k_per_rank_b1", "n50k_b1_two_lanes"))
    assert a["n_hyp_total"] == 50000 and a["n_hyp_per_rank"] == 50000 and a["B"] == 1 and a["steps_per_select"] == 1
    assert d["config"]["steps_per_select"] == 8
    assert b["n_hyp_total"] == 50000 and b["B"] == 32 and b["steps"] == 5
    assert w["n_hyp_total"] == 50000 and two["lanes"] == 2 and a["lanes"] == 1
    assert abs(a["hypotheses_per_s"] - 50000 / (a["ms_per_step"] * 1e-3)) / a["hypotheses_per_s"] < 1e-6
    assert abs(b["hypotheses_per_s"] - 32 * 50000 / (b["ms_per_step"] * 1e-3)) / b["hypotheses_per_s"] < 1e-6
    assert 0.5 * d["value"] < a["hypotheses_per_s"] < 1.2 * d["value"]          # the same workload at one rank
    assert b["hypotheses_per_s"] > 0.9 * a["hypotheses_per_s"]                   # batching never costs throughput
    assert two["hypotheses_per_s"] > 0.97 * a["hypotheses_per_s"]                # overlapping steps never costs either
    # what an n-GPU strong-scaling run can reach, from the shard timings of this GPU
    p = sec["predicted_strong_scaling"]
    for n, per in ((2, 25000), (4, 12500), (8, 6250)):
        row = p["n_gpus_%d" % n]
        assert row["n_hyp_per_rank"] == per and 0.5 < row["efficiency"] <= 1.05 and 0.5 < row["efficiency_two_lanes"] <= 1.1
        assert abs(row["efficiency"] - p["ms_per_step_n50k"] / (n * row["ms_per_step"])) < 1e-9
        # the lane count a --gpus n run takes by default, and the efficiency the driver would compute for it
        assert row["default_lanes"] == (1 if n == 2 else 2)
        assert row["efficiency_default"] == row["efficiency_vs_one_lane_n1" + ("" if n == 2 else "_two_lanes")]
    assert p["n_gpus_2"]["efficiency"] > p["n_gpus_8"]["efficiency"]


def test_bench_rccl_branch_on_one_gpu():
    """AHV_BENCH_FORCE_PG=1: a single rank creates the RCCL process group (device_id given) and runs every
    collective of the world > 1 path -- the async key all-reduce ring, the barriers, the all-reduce of the time,
    the all-gather check -- so the first multi-GPU run is not the first run of that code.  Same arg-max and score
    as the single-process run."""
    base = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}

    def run(extra):
        out = subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=REPO, env=dict(env, **extra))
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads(out.stdout.strip())

    a, b = run({}), run({"AHV_BENCH_FORCE_PG": "1", "AHV_BENCH_LANES": "1"})
    assert a["config"]["backend"] == "single process" and b["config"]["backend"] == "rccl"
    assert b["config"]["lanes"] == 1 and "rccl" in b["config"]["ranks"][0]["collectives"]
    assert b["n_gpus"] == 1 and a["result"] == b["result"]
    assert b["config"]["steps_per_collective"] == 8 and a["config"]["steps_per_collective"] is None
    assert b["scaling"] == "strong" and b["secondary"]["n50k_b1_collective_per_step"]["steps_per_collective"] == 1
    assert b["secondary"]["configs3_b32_n50k"]["steps_per_collective"] == 8
    # the per-step cadence costs what a collective's fixed part costs (17 ... 36 us on this stack), never a multiple
    assert b["secondary"]["n50k_b1_collective_per_step"]["ms_per_step"] < b["ms_per_step"] + 0.08
    # The collective is cheap: the int64 keys go into the all-reduce as the kernel packed them (no re-encoding launches),
    # eight steps' keys per collective, consumed one group later -- so the step with the process group costs what the step
    # without it costs (round 3: +4.6 %).  Two processes seconds apart differ by the clock they are granted (<= 1 %):
    # one repeat before judging.
    ta, tb = a["ms_per_step"], b["ms_per_step"]
    gap = a["ms_per_step"] - a["roofline"]["kernel_ms"]
    if tb > 1.02 * ta or gap > 0.010:
        a2 = run({})
        ta, tb = min(ta, a2["ms_per_step"]), min(tb, run({"AHV_BENCH_FORCE_PG": "1", "AHV_BENCH_LANES": "1"})["ms_per_step"])
        gap = min(gap, a2["ms_per_step"] - a2["roofline"]["kernel_ms"])
    assert tb <= 1.03 * ta, (ta, tb)   # (two processes: the clock they are granted differs by up to ~1 %)
    # one step = ONE fused launch + ONE select launch: what is not the scorer stays small (21 us in round 3)
    assert 0 <= gap < 0.012, (a["ms_per_step"], a["roofline"]["kernel_ms"], gap)
    # What a multi-rank run takes by default: TWO lanes, each with its own stream, key buffer and RCCL communicator
    # (AHV_BENCH_TWO_LANES_PG=1 selects it at one rank).  Same winner, no slower than one stream, priced from wall time with
    # the kernel time of a one-lane calibration leg -- so an 8-GPU run is not the first run of two communicators either.
    c = run({"AHV_BENCH_FORCE_PG": "1", "AHV_BENCH_TWO_LANES_PG": "1"})
    assert c["config"]["backend"] == "rccl" and c["config"]["lanes"] == 2 and c["result"] == a["result"]
    assert "own communicator" in c["config"]["lanes_note"] and c["config"]["steps_per_collective"] == 8
    r = c["roofline"]
    assert "calibration leg" in r["kernel_ms_is"] and "wall time" in r["achieved_is"]
    assert abs(r["achieved"] - 1839104 * 50000 / (c["ms_per_step"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    assert 0.5 < r["frac"] < 1.0 and r["kernel_ms"] <= 1.02 * c["ms_per_step"]
    assert c["ms_per_step"] <= 1.03 * tb, (c["ms_per_step"], tb)
    one = c["secondary"]["n50k_b1_one_stream"]
    assert one["lanes"] == 1 and one["steps_per_collective"] == 8 and "n50k_b1_two_lanes" not in c["secondary"]


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` (no external launcher) spawns its two workers before touching the GPU; on this
    1-GPU box the ranks share the device over gloo (a rehearsal of the sharded path, not a measurement)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # (two lanes are the default from 4 ranks on -- shards of at most 16 384 hypotheses; this 2-rank rehearsal raises the bound so
    # that the path with two communicators per rank is the one that runs)
    env["AHV_BENCH_TWO_LANES_MAX_N"] = "25000"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    # strong scaling: ONE set of 50 000 hypotheses, 25 000 per rank; value = the whole job's hypotheses per second
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["n_hyp_total"] == 50000 and d["config"]["n_hyp_per_gpu"] == 25000
    assert d["config"]["backend"] == "gloo" and d["scaling"] == "strong"
    assert abs(d["value"] - 50000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["algorithmic_flops_per_launch"] == 25000 * 1839104 and d["roofline"]["traffic"] is None
    sec = d["secondary"]
    assert sec["n50k_b1_collective_per_step"]["steps_per_collective"] == 1 and d["config"]["steps_per_collective"] == 8
    assert sec["weak_n50k_per_rank_b1"]["n_hyp_total"] == 100000 and sec["weak_n50k_per_rank_b1"]["n_hyp_per_rank"] == 50000
    assert sec["configs3_b32_n50k"]["n_hyp_per_rank"] == 25000 and "predicted_strong_scaling" not in sec
    # two lanes (two communicators per rank), the one-stream loop timed beside it
    assert d["config"]["lanes"] == 2 and sec["n50k_b1_one_stream"]["lanes"] == 1 and "n50k_b1_two_lanes" not in sec
    assert "calibration leg" in d["roofline"]["kernel_ms_is"]
    # which GPU each rank drove (here: both ranks on the one device of the box, over gloo)
    who = d["config"]["ranks"]
    assert [w["rank"] for w in who] == [0, 1] and all(w["device_index"] == 0 and w["collectives"] == "gloo" for w in who)
    assert who[0]["pid"] != who[1]["pid"] and who[0]["pci_bus_id"] == who[1]["pci_bus_id"]
    assert d["ms_per_step_select_every_step"] == sec["n50k_b1_collective_per_step"]["ms_per_step"]
    # a failing rank takes the launch down with a non-zero code instead of hanging in a collective
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "nccl",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=600, cwd=REPO, env=env)
    assert bad.returncode != 0 and "only 1 GPU(s) visible" in bad.stderr
