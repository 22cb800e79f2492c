#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] on ONE GPU (the headline number is bench.py).
Prints one JSON object per line; run through gpurun and keep the output under profiles/."""
import importlib, json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ahv = importlib.import_module("3dahv_amd")
ops = ahv.ops
dev = torch.device("cuda:0")
FLOPS = 1_839_104
# Backward of the fused scorer, algorithmic FLOP per hypothesis and kernel:
#   head  recompute the forward 1 839 104 + dW2 = dv relu(u)^T 2*64*32*32 + dr = W2^T dv 2*64*32*32          = 2 101 248
#   dW1   gather the rotated volume 131 072 + dW1 += du X^T 2*64*384*32                                      = 1 703 936
#   dV    dX = W1^T du 2*64*384*32 + adjoint of the gather (scatter) 131 072                                  = 1 703 936
FLOPS_BWD_KERNELS = {"head": 2_101_248, "dW1": 1_703_936, "dV": 1_703_936}
FLOPS_BWD = sum(FLOPS_BWD_KERNELS.values())   # 5 509 120
PEAK = 157.3


def bwd_roofline(n_hyp, ms):
    """The contract's `roofline` object for the three-kernel backward as a whole (fp32 MFMA bound; the only per-hypothesis
    HBM traffic is dL/du: 8 KB written once, read twice = 24 KB against 5.5 MFLOP)."""
    ach = n_hyp * FLOPS_BWD / ms / 1e9
    return {"bound": "mfma", "achieved": ach, "peak": PEAK, "unit": "TFLOP/s", "frac": ach / PEAK, "traffic": None,
            "kernels": "score_backward_head_kernel + score_backward_w1_kernel (+ reduce) + score_backward_volume_rmw_kernel",
            "algorithmic_flops_per_hypothesis": FLOPS_BWD, "algorithmic_flops_per_kernel": FLOPS_BWD_KERNELS,
            "algorithmic_hbm_bytes_per_hypothesis": 3 * 8192 + 36 + 4,
            "note": "includes the head kernel's recompute of the forward (1 839 104 FLOP): nothing of the forward is kept in HBM"}


def timeit(fn, iters, warm=3, warm_ms=80.0):
    """ms per call between two HIP events.  The warm-up is time-based as well as counted: the chip ramps its clock for
    ~30 ms after an idle period, and round 5's "B = 32 x 6 250 at 0.79" was three 3-ms warm-up calls, i.e. a measurement
    INSIDE the ramp."""
    t0 = time.perf_counter()
    n = 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


g = torch.Generator().manual_seed(0)
W1 = ((torch.rand(32, 384, generator=g) * 2 - 1) / np.sqrt(384.0)).to(dev)
W2 = ((torch.rand(32, 32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
b2 = ((torch.rand(32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
vol = (torch.randn(2, 32, 16, 8, 8, 8, generator=g) * 1.15).to(dev)
only = sys.argv[1:]

if not only or "3" in only:
    # configs[2]: LINEMOD pair, dense SO(3) grid N=200k; fused and op-level (materialising, HBM-bound) pipelines
    N = 200_000
    R = ops.so3_grid(N, dev)  # deterministic super-Fibonacci grid, generated on the device
    vs, vt = vol[0, :1], vol[1, :1]
    ft = ops.forward_3d2d(vt, W1, W2, b2)
    ms = timeit(lambda: ops.score_hypotheses(vs, ft, R, W1, W2, b2, want_scores=False), 10)
    print(json.dumps({"config": "3 fused", "N": N, "ms": ms, "hyp_per_s": N / ms * 1e3, "tflops": N * FLOPS / ms / 1e9,
                      "frac_fp32_mfma_peak": N * FLOPS / ms / 1e9 / 157.3}))
    src = vs[0][None].expand(N, -1, -1, -1, -1)
    rot = ops.rotate_volume(src, R)
    ms_rot = timeit(lambda: ops.rotate_volume(src, R), 5)
    f = ops.forward_3d2d(rot, W1, W2, b2)
    ms_f = timeit(lambda: ops.forward_3d2d(rot, W1, W2, b2), 5)
    fs = f.reshape(1, N, 32, 64)
    ms_s = timeit(lambda: ops.score_features(fs, ft), 5)
    sc = ops.score_features(fs, ft)
    ms_a = timeit(lambda: ops.argmax(sc), 5)
    s_fused, key = ops.score_hypotheses(vs, ft, R, W1, W2, b2)
    assert ops.argmax(sc)[1].item() == ops.unpack_best(key)[1].item()
    tot = ms_rot + ms_f + ms_s + ms_a
    print(json.dumps({"config": "3 op-level", "N": N, "ms_total": tot, "hyp_per_s": N / tot * 1e3,
                      "rotate_volume": {"ms": ms_rot, "GBps": N * 32804 / ms_rot / 1e6},
                      "forward_3d2d": {"ms": ms_f, "GBps": N * 40960 / ms_f / 1e6, "tflops": N * 1703936 / ms_f / 1e9},
                      "score_features": {"ms": ms_s, "GBps": N * 8196 / ms_s / 1e6}, "argmax": {"ms": ms_a},
                      "algorithmic_bytes_per_hyp": 81960, "GBps_pipeline": N * 81960 / tot / 1e6,
                      "max_abs_diff_vs_fused": (sc - s_fused).abs().max().item()}))
    del rot, f, fs, src

if not only or "4" in only:
    # configs[3]: B=32 pairs, shared proposals; one rank's 1/8 shard and the whole set on one GPU
    vs, vt = vol[0], vol[1]
    ft = ops.forward_3d2d(vt, W1, W2, b2)
    for N in (6250, 50_000):
        R = torch.from_numpy(ahv.rotations.haar_rotations_np(N, 9)).to(dev)
        ms = timeit(lambda: ops.score_hypotheses(vs, ft, R, W1, W2, b2, want_scores=False), 20 if N < 20000 else 5)
        row = {"config": "4 B=32", "N_per_gpu": N, "ms": ms, "hyp_per_s": 32 * N / ms * 1e3,
               "frac_fp32_mfma_peak": 32 * N * FLOPS / ms / 1e9 / 157.3}
        # the whole step of one rank: verify_pair (target features built in the launch) + select
        key = torch.full((32,), -(1 << 63), dtype=torch.int64, device=dev)
        def step():
            ops.verify_pair(vs, vt, R, W1, W2, b2, want_scores=False, best_key=key, reset_best=False)
            ops.select_rotation(key, R, reset_key=True)
        row["ms_verify_step"] = timeit(step, 20 if N < 20000 else 5)
        print(json.dumps(row))

if not only or "5" in only:
    # configs[4]: coarse 10k + 1k refined hypotheses, the whole verify step replayed from one hipGraph
    vs, vt = vol[0, :1], vol[1, :1]
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(10_000, 11)).to(dev)
    # one rank: the step is ONE launch (ahv_coarse_to_fine_f32); fused=False = the five launches a sharded step is made of
    for fused, use_graph in ((True, False), (True, True), (False, False), (False, True)):
        c2f = ahv.refine.CoarseToFine(W1, W2, b2, R, n_fine=1000, max_angle_deg=10.0, batch=1, use_graph=use_graph, fused=fused)
        ms = timeit(lambda: c2f(vs, vt), 200, warm=5)
        out = c2f(vs, vt)
        # the producer writes straight into the step's static inputs: no staging copies in front of the replay
        c2f.buffers[0].copy_(vs)
        c2f.buffers[1].copy_(vt)
        ms_in_place = timeit(lambda: c2f(), 200, warm=5)
        assert not (fused and c2f._fused_state.gave_up())
        # eight steps per replay (run_many): a hipGraphLaunch idles the device ~9 us between two replays, so the gap is paid once
        ms_many = timeit(lambda: c2f.run_many(steps=8), 25, warm=3) / 8
        print(json.dumps({"config": "5 coarse10k+fine1k", "graph": use_graph, "us_per_step": ms * 1e3,
                          "us_per_step_inputs_in_place": ms_in_place * 1e3, "us_per_step_8_steps_per_call": ms_many * 1e3,
                          "launches_per_step": 1 if fused else 5,
                          "hyp_per_s": 11_000 / ms_in_place * 1e3, "fine_score": out[0].item(), "coarse_score": out[3].item()}))

if not only or "shard" in only:
    # what one rank of a strong-scaling run does per step (B = 1): the one-launch verify + select on its shard of 50 000
    vs, vt = vol[0, :1], vol[1, :1]
    key = torch.full((1,), -(1 << 63), dtype=torch.int64, device=dev)
    for N in (1000, 6250, 10_000, 12_500, 25_000, 50_000):
        R = torch.from_numpy(ahv.rotations.haar_rotations_np(N, 9)).to(dev)
        row = {"config": "shard B=1", "N": N}
        def make_step(kw, k=None, R_=R):
            k = key if k is None else k
            def step():
                ops.verify_pair(vs, vt, R_, W1, W2, b2, want_scores=False, best_key=k, reset_best=False, **kw)
                ops.select_rotation(k, R_, reset_key=True)
            return step
        for name, kw in (("teams", {}), ("single_waves", {"no_teams": True})):
            row["us_per_step_" + name] = timeit(make_step(kw), 200, warm=20) * 1e3
        row["hyp_per_s"] = N / row["us_per_step_teams"] * 1e6
        # the multi-rank cadence: 8 verify launches per select (bench.py under a process group), one stream and two.
        # Two lanes: the groups alternate between two streams, each with its own keys, so that one step's drain -- the wait
        # for the slowest workgroup, the launch gap, the next prologue -- overlaps the next step's hypotheses.
        keys8 = [torch.full((8, 1), -(1 << 63), dtype=torch.int64, device=dev) for _ in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def run_groups(n_groups, lanes):
            main = torch.cuda.current_stream()
            for g in range(n_groups):
                k8 = keys8[g % lanes]
                if lanes > 1:
                    torch.cuda.set_stream(streams[g % lanes])
                for j in range(8):
                    ops.verify_pair(vs, vt, R, W1, W2, b2, want_scores=False, best_key=k8[j], reset_best=False)
                ops.select_rotation(k8.view(-1), R, reset_key=True)
            if lanes > 1:
                torch.cuda.set_stream(main)

        for lanes in (1, 2):
            for s_ in streams:
                s_.wait_stream(torch.cuda.current_stream())
            run_groups(8, lanes)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_groups(50, lanes)
            torch.cuda.synchronize()
            row["us_per_step_8_per_select" + ("" if lanes == 1 else "_two_lanes")] = (time.perf_counter() - t0) / 400 * 1e6
        print(json.dumps(row))

if not only or "enc" in only:
    # once-per-pair encoder (forward_2d3d), stock PyTorch-ROCm operators: eager vs one hipGraph replay
    torch.manual_seed(0)
    fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).to(dev).eval()
    a, b = torch.randn(1, 768, 8, 8, device=dev), torch.randn(1, 768, 8, 8, device=dev)
    with torch.no_grad():
        ms_eager = timeit(lambda: fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0), 30)
    run = fa.graphed_forward_2d3d(1)
    ms_graph = timeit(lambda: run(a, b), 100)
    print(json.dumps({"config": "encoder forward_2d3d B=1", "eager_us": ms_eager * 1e3, "hipgraph_us": ms_graph * 1e3,
                      "weights_MB": sum(p.numel() for p in fa.parameters()) * 4 / 1e6}))
    for B in (1, 32):  # HIP kernels vs the stock PyTorch-ROCm operator path of the same module
        a, b = torch.randn(B, 768, 8, 8, device=dev), torch.randn(B, 768, 8, 8, device=dev)
        with torch.no_grad():
            fa.use_hip_encoder = fa.att.use_hip = True
            ms_hip = timeit(lambda: fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0), 20)
            ref_hip = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
            fa.use_hip_encoder = fa.att.use_hip = False
            ms_torch = timeit(lambda: fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0), 10)
            ref_t = fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0)
            fa.use_hip_encoder = fa.att.use_hip = True
        rel = max(((x - y).abs().max() / y.abs().max()).item() for x, y in zip(ref_hip, ref_t))
        print(json.dumps({"config": "encoder forward_2d3d", "B": B, "hip_us": ms_hip * 1e3, "torch_ops_us": ms_torch * 1e3,
                          "max_rel_diff": rel}))

if "enchost" in only:
    # HOST time of forward_2d3d at B = 1 (the packed-weight version check cost ~350 us of Python
    # per call): perf_counter around calls that only enqueue (no sync inside the loop), eager and graphed; GPU time of
    # the same loops from events; the version check alone.
    torch.manual_seed(0)
    fa = ahv.aligner.Feature_Aligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4).to(dev).eval()
    a, b = torch.randn(1, 768, 8, 8, device=dev), torch.randn(1, 768, 8, 8, device=dev)

    def host_us(fn, iters=300):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / iters * 1e6
        torch.cuda.synchronize()
        return host, e0.elapsed_time(e1) / iters * 1e3

    with torch.no_grad():
        h_e, g_e = host_us(lambda: fa.forward_2d3d(a, b, random_mask=False, mask_ratio=0.0))
    run = fa.graphed_forward_2d3d(1)
    h_g, g_g = host_us(lambda: run(a, b))
    t0 = time.perf_counter()
    for _ in range(2000):
        ahv.aligner._packed_aligner(fa, dev)
    chk = (time.perf_counter() - t0) / 2000 * 1e6
    print(json.dumps({"config": "encoder forward_2d3d B=1 host time", "eager": {"host_us_per_call": h_e, "gpu_us_per_call": g_e},
                      "hipgraph": {"host_us_per_call": h_g, "gpu_us_per_call": g_g},
                      "packed_weight_version_check_us": chk, "parameters_checked": len(list(fa.parameters()))}))

if not only or "train" in only:
    # training-size scorer step (reference config.yaml: TRAIN.BS 12, DATA.NUM_ROTA 3000, per-sample rotations):
    # forward + backward of the (B,N) similarities, HIP fused kernels vs autograd over stock PyTorch-ROCm operators
    import types
    import torch.nn.functional as F

    def _rotate_volume(volume, R):  # the reference's operator sequence (utils.py:123-129), stock operators
        theta = torch.cat([R, R.new_zeros(R.shape[0], 3, 1)], dim=-1)
        grid = F.affine_grid(theta, list(volume.shape), align_corners=False)
        return F.grid_sample(volume, grid, mode="bilinear", padding_mode="zeros", align_corners=False)

    def _forward_3d2d(v, W1, W2, b2):  # modules/modules.py:112-124
        m, c, d, h, w = v.shape
        slabs = torch.cat([v.permute(0, 1, 4, 2, 3).reshape(m, c * w, d, h), v.permute(0, 1, 3, 2, 4).reshape(m, c * h, d, w),
                           v.reshape(m, c * d, h, w)], dim=1)
        u = F.relu(F.conv2d(slabs, W1.reshape(32, 384, 1, 1)))
        return F.normalize(F.conv2d(u, W2.reshape(32, 32, 1, 1), b2), p=2, dim=1).flatten(2)

    torch_ref = types.SimpleNamespace(rotate_volume=_rotate_volume, forward_3d2d=_forward_3d2d)
    B, N = 12, 3000
    vs = vol[0, :B].clone().requires_grad_(True)
    vt = vol[1, :B].clone()
    R = torch.from_numpy(ahv.rotations.haar_rotations_np(B * N, 13)).to(dev).reshape(B, N, 3, 3)
    P = [W1.clone().requires_grad_(True), W2.clone().requires_grad_(True), b2.clone().requires_grad_(True)]
    gs = torch.randn(B, N, device=dev)

    def hip_step():
        ft = ops.forward_3d2d_autograd(vt, *P)
        s = ops.score_hypotheses_autograd(vs, ft, R, *P)
        return torch.autograd.grad(s, [vs] + P, grad_outputs=gs)

    def torch_step():
        ft = torch_ref.forward_3d2d(vt, *P)
        out = []
        for b in range(B):
            rot = torch_ref.rotate_volume(vs[b][None].expand(N, -1, -1, -1, -1), R[b])
            out.append((torch_ref.forward_3d2d(rot, *P) * ft[b][None]).sum(dim=1).mean(dim=-1))
        return torch.autograd.grad(torch.stack(out), [vs] + P, grad_outputs=gs)

    ms_hip = timeit(hip_step, 10)
    ms_fwd = timeit(lambda: ops.score_hypotheses(vs, ops.forward_3d2d(vt, W1, W2, b2), R, W1, W2, b2), 10)
    ms_bwd = timeit(lambda: ops.score_hypotheses_backward(vs, ops.forward_3d2d(vt, W1, W2, b2), R, W1, W2, b2, gs), 10)
    ms_torch = timeit(torch_step, 3, warm=1)
    ga, gb = hip_step(), torch_step()
    rel = max(((x - y).abs().max() / y.abs().max()).item() for x, y in zip(ga, gb))
    print(json.dumps({"config": "training scorer step", "B": B, "N": N, "hip_fwd_bwd_ms": ms_hip, "hip_fwd_only_ms": ms_fwd,
                      "hip_bwd_only_ms": ms_bwd, "torch_autograd_ms": ms_torch, "speedup": ms_torch / ms_hip,
                      "max_rel_grad_diff_vs_torch_fp32": rel, "hyp_per_s_fwd_bwd": B * N / ms_hip * 1e3,
                      "roofline": bwd_roofline(B * N, ms_bwd)}))

if "train9000" in only:
    # the reference's CO3D training size (train_estimator_co3d.py:12-15: NUM_ROTA = 9000, BS = 32): 288 000 hypothesis
    # evaluations forward + backward per step, per-sample rotation sets; HIP kernels only (the all-torch graph needs
    # ~37 GB of activations at this size)
    B, N = 32, 9000
    g9 = torch.Generator().manual_seed(9)
    vs9 = (torch.randn(B, 16, 8, 8, 8, generator=g9) * 1.15).to(dev)
    vt9 = (torch.randn(B, 16, 8, 8, 8, generator=g9) * 1.15).to(dev)
    R9 = ops.random_rotations(B * N, seed=9, device=dev).reshape(B, N, 3, 3)
    gs9 = torch.randn(B, N, generator=g9).to(dev)
    ft9 = ops.forward_3d2d(vt9, W1, W2, b2)
    lib = ahv._lib.load()
    ms_fwd = timeit(lambda: ops.score_hypotheses(vs9, ft9, R9, W1, W2, b2), 5)
    ms_bwd = timeit(lambda: ops.score_hypotheses_backward(vs9, ft9, R9, W1, W2, b2, gs9), 5)
    # the training pair (ABI 2.3): a forward that keeps the pre-activations + a backward that does not recompute them
    ms_fwd_train = timeit(lambda: ops.score_hypotheses_train(vs9, ft9, R9, W1, W2, b2), 5)
    def train_pair():
        _, ws9 = ops.score_hypotheses_train(vs9, ft9, R9, W1, W2, b2)
        ops.score_hypotheses_backward(vs9, ft9, R9, W1, W2, b2, gs9, workspace=ws9)
    ms_pair = timeit(train_pair, 5)
    print(json.dumps({"config": "training scorer step, CO3D training size", "B": B, "N": N, "hip_fwd_only_ms": ms_fwd,
                      "hip_bwd_only_ms": ms_bwd, "hyp_per_s_fwd_bwd": B * N / (ms_fwd + ms_bwd) * 1e3,
                      "fwd_frac_fp32_mfma_peak": B * N * FLOPS / ms_fwd / 1e9 / 157.3,
                      "bwd_tflops": B * N * FLOPS_BWD / ms_bwd / 1e9, "roofline": bwd_roofline(B * N, ms_bwd),
                      "training_pair": {"fwd_keeping_preactivations_ms": ms_fwd_train, "fwd_plus_bwd_ms": ms_pair,
                                        "bwd_saved_preactivations_ms": ms_pair - ms_fwd_train,
                                        "against_recomputing_pair_ms": ms_fwd + ms_bwd,
                                        "algorithmic_flops_per_hypothesis_bwd": FLOPS_BWD - FLOPS,
                                        "bwd_frac_fp32_mfma_peak": B * N * (FLOPS_BWD - FLOPS) / (ms_pair - ms_fwd_train) / 1e9 / PEAK,
                                        "note": "the backward without the forward recompute does 3 670 016 FLOP per hypothesis "
                                                "(dW2 + dr + gather + dW1 + dX + scatter) and moves 16 KB more per hypothesis"},
                      "backward_workspace_GB": lib.ahv_score_hypotheses_backward_workspace_bytes(B, N) / 1e9}))

if not only or "trainstep" in only:
    # whole training_step (modules/model_co3d.py:71-91) at the reference's batch: where the time goes
    cfg = {"RUN_NAME": "t", "DATA": {"NUM_ROTA": 3000, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256, "ACC_THR": 30, "VIEW_THR": 90},
           "TRAIN": {"MASK": True, "MASK_RATIO": 0.25, "LR": 1e-4}}
    torch.manual_seed(0)
    m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev).train()
    (opt,), _ = m.configure_optimizers()
    gg = torch.Generator().manual_seed(3)
    batch = {"image": torch.randn(12, 2, 3, 256, 256, generator=gg).to(dev),
             "relative_rotation": ahv.rotations.random_rotations(12, generator=gg).to(dev)[:, None]}

    def full_step():
        opt.zero_grad(set_to_none=True)
        loss = m.training_step(batch, 0)
        loss.backward()
        opt.step()

    def encoder_only():
        opt.zero_grad(set_to_none=True)
        vs, vt = m.feature_aligner.forward_2d3d(m.feature_extraction(batch["image"][:, 0]),
                                                m.feature_extraction(batch["image"][:, 1]), random_mask=True, mask_ratio=0.25)
        (vs.square().mean() + vt.square().mean()).backward()

    for blas in ("cublaslt", "cublas"):  # hipBLASLt (torch's default here) / rocBLAS (what harness.fit selects)
        torch.backends.cuda.preferred_blas_library(blas)
        ms_full = timeit(full_step, 5, warm=2)
        ms_enc = timeit(encoder_only, 5, warm=2)
        print(json.dumps({"config": "training_step B=12 N=3000 (synthetic backbone)",
                          "blas": {"cublaslt": "hipBLASLt", "cublas": "rocBLAS"}[blas], "full_step_ms": ms_full,
                          "encoder_fwd_bwd_torch_autograd_ms": ms_enc, "scorer_and_loss_and_adamw_ms": ms_full - ms_enc}))
    # the same iteration captured once in a hipGraph and replayed (harness.GraphedTrainStep)
    torch.backends.cuda.preferred_blas_library("cublaslt")
    gstep = ahv.harness.GraphedTrainStep(m, batch_size=12, device=dev)
    ms_graph = timeit(lambda: gstep(batch), 10, warm=2)
    print(json.dumps({"config": "training_step B=12 N=3000 (synthetic backbone)", "mode": "one hipGraph per iteration (rocBLAS)",
                      "full_step_ms": ms_graph, "loss": float(gstep.loss.item())}))

if not only or "pairs" in only:
    # end-to-end per-pair throughput of the evaluation loop (test_co3d.py:93-154 counterpart): encoder + verify at
    # N = 50 000 + metric, synthetic layer_4 features; ordered pairs one by one (reference order) vs batched
    cfgp = {"RUN_NAME": "t", "DATA": {"NUM_ROTA": 50000, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256}}
    torch.manual_seed(0)
    mp_ = ahv.estimator.EstimatorCo3d(cfgp).to(dev).eval()
    P = ops.random_rotations(50000, seed=1, device=dev)
    graphed = mp_.feature_aligner.graphed_forward_2d3d(2)
    for name, kw in (("one by one (reference order)", dict(batch_pairs=False, batch_sequences=1)),
                     ("ordered pairs batched", dict(batch_pairs=True, batch_sequences=1)),
                     ("batched + encoder replayed from a hipGraph", dict(batch_pairs=True, batch_sequences=1, encoder_fn=graphed)),
                     ("4 sequences (8 pairs) per batch", dict(batch_pairs=True, batch_sequences=4)),
                     ("16 sequences (32 pairs) per batch = the default", dict())):
        seqs = list(ahv.harness.SyntheticSequences(40, 2, seed=1))     # materialised: data generation is not timed
        np.random.seed(0)
        ahv.harness.evaluate_category(cfgp, mp_, seqs[:32], device=dev, proposals=P, **kw)   # warm-up at the timed batch shape
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = ahv.harness.evaluate_category(cfgp, mp_, seqs, device=dev, proposals=P, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "evaluation loop, N=50000 per pair", "mode": name, "pairs": len(e),
                          "ms_per_pair": dt / len(e) * 1e3, "pairs_per_s": len(e) / dt, "mean_err_deg": float(np.mean(e))}))
    with ops.split_f16_scorer():  # opt-in split-f16 scorer (per-call flag)
        seqs = list(ahv.harness.SyntheticSequences(40, 2, seed=1))
        np.random.seed(0)
        ahv.harness.evaluate_category(cfgp, mp_, seqs[:32], device=dev, proposals=P, batch_sequences=16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = ahv.harness.evaluate_category(cfgp, mp_, seqs, device=dev, proposals=P, batch_sequences=16)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "evaluation loop, N=50000 per pair", "mode": "16 sequences per batch, AHV_SCORE_SPLIT_F16 (split-f16, opt-in)",
                          "pairs": len(e), "ms_per_pair": dt / len(e) * 1e3, "pairs_per_s": len(e) / dt, "mean_err_deg": float(np.mean(e))}))

if "trainlines" in only:
    # The reference's infoNCE_loss lines (modules/model_co3d.py:41-61) under patch.install() with the module in training mode,
    # forward + backward: every sample's rotate_volume .. mean(dim=-1) deferred into ONE differentiable fused launch (default)
    # against every line as its own differentiable op-level kernel (install(defer=False)), and the mirror's batched training
    # pair (one launch pair for all samples) for scale.
    import types
    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None

    class Feature_Aligner(torch.nn.Module):  # noqa: N801
        def __init__(self):
            super().__init__()
            self.feature_embedding_2d = torch.nn.Sequential(torch.nn.Conv2d(384, 32, 1, bias=False), torch.nn.ReLU(),
                                                            torch.nn.Conv2d(32, 32, 1))

        def forward_3d2d(self, x):
            raise AssertionError("not patched")
    mm.Feature_Aligner = Feature_Aligner
    fa = Feature_Aligner().to(dev).train()
    with torch.no_grad():
        fa.feature_embedding_2d[0].weight.copy_(W1.reshape(32, 384, 1, 1))
        fa.feature_embedding_2d[2].weight.copy_(W2.reshape(32, 32, 1, 1))
        fa.feature_embedding_2d[2].bias.copy_(b2)

    def infonce(rotate_volume, img_feat_1, img_feat_2, sampled_R, gt_delta_R, num_rota, acc_thr=30.0):
        bs = gt_delta_R.shape[0]
        with torch.no_grad():
            gt_sim = (torch.sum(sampled_R.flatten(2) * gt_delta_R.view(-1, 1, 9), dim=-1).clamp(-1, 3) - 1) / 2
            gt_dis = torch.arccos(gt_sim) / np.pi
            posi_indices = [torch.nonzero(180 * gt_dis[i] <= acc_thr).squeeze(-1) for i in range(bs)]
        img_feat_warp = [rotate_volume(img_feat_1[idx:idx+1].expand(num_rota, -1, -1, -1, -1), sampled_R[idx]) for idx in range(bs)]
        img_feat_warp = [fa.forward_3d2d(img_feat) for img_feat in img_feat_warp]
        img_feat_2 = fa.forward_3d2d(img_feat_2)
        sim = [(img_feat_warp[idx] * img_feat_2[idx:idx+1]).sum(dim=1).mean(dim=-1) for idx in range(bs)]
        positive_sim = torch.stack([torch.exp(sim[idx][posi_indices[idx]] / 0.1).sum(dim=0) for idx in range(bs)])
        positive_negative_sim = (torch.exp(torch.stack(sim) / 0.1)).sum(dim=-1)
        return -torch.log(positive_sim / positive_negative_sim.clamp(min=1e-8)).mean()

    for B, N in ((12, 3000), (32, 9000)):
        gT = torch.Generator().manual_seed(21)
        v1 = (torch.randn(B, 16, 8, 8, 8, generator=gT) * 1.15).to(dev).requires_grad_(True)
        v2 = (torch.randn(B, 16, 8, 8, 8, generator=gT) * 1.15).to(dev).requires_grad_(True)
        gt = ops.random_rotations(B, seed=5, device=dev)
        Rs = ops.random_rotations(B * N, seed=6, device=dev).reshape(B, N, 3, 3).clone()
        Rs[:, 0] = gt
        row = {"config": "unchanged infoNCE_loss lines, module in training mode, forward + backward", "B": B, "N": N}
        losses = {}
        for mode in ("deferred", "op_level"):
            ahv.patch.install(um, mm, defer=(mode == "deferred"))
            try:
                def step():
                    for p_ in list(fa.parameters()) + [v1, v2]:
                        p_.grad = None
                    loss = infonce(um.rotate_volume, v1, v2, Rs, gt, N)
                    loss.backward()
                    return loss
                step()
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats(dev)
                row[mode + "_ms"] = timeit(step, 3, warm=1)
                row[mode + "_peak_memory_GB"] = torch.cuda.max_memory_allocated(dev) / 1e9
                losses[mode] = step().item()
            finally:
                ahv.patch.uninstall()
        row["same_loss"] = abs(losses["deferred"] - losses["op_level"]) < 1e-5 * max(1.0, abs(losses["op_level"]))
        # the mirror's batched form: one training pair for all samples
        P = [W1.clone().requires_grad_(True), W2.clone().requires_grad_(True), b2.clone().requires_grad_(True)]
        gs = torch.randn(B, N, device=dev)

        def batched():
            ft = ops.forward_3d2d_autograd(v2, *P)
            s = ops.score_hypotheses_autograd(v1, ft, Rs, *P)
            return torch.autograd.grad(s, [v1, v2] + P, grad_outputs=gs)
        row["batched_training_pair_ms"] = timeit(batched, 3, warm=1)
        print(json.dumps(row), flush=True)

if "optiona" in only:
    # INTEGRATION.md option A under the reference script's own conditions (grad mode on, anomaly detection on, model.eval()):
    # the verbatim per-pair sequence of test_co3d.py:133-152 on a reference-shaped stand-in whose two callables are patched,
    # beside option B (one fused launch + one select) on the same pairs.  N = 50 000 per pair, one pair at a time.
    import types

    class RefAligner(ahv.aligner.Feature_Aligner):     # torch-operator forward_2d3d, like the reference's class
        def forward_2d3d(self, a, b, random_mask=True, mask_ratio=0.25):
            self.use_hip_encoder = self.att.use_hip = False
            return super().forward_2d3d(a, b, random_mask, mask_ratio)

        def forward_3d2d(self, x):
            raise AssertionError("not patched")

    class Estimator(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.feature_aligner = RefAligner(in_channel=768, mid_channel=256, out_channel=32, n_heads=4, depth=4)
            self.gain = torch.nn.Parameter(torch.ones(()))

        def forward(self, f_src, f_tgt):   # (stands behind the backbone: layer_4 features in, parameters that require grad)
            return self.feature_aligner.forward_2d3d(f_src * self.gain, f_tgt * self.gain, random_mask=False, mask_ratio=0)

    torch.manual_seed(0)
    model = Estimator().to(dev)
    model.eval()
    um, mm = types.ModuleType("utils"), types.ModuleType("modules.modules")
    um.rotate_volume = lambda *a, **k: None
    mm.Feature_Aligner = RefAligner
    proposals = ops.random_rotations(50000, seed=1, device=dev)
    gt = ops.random_rotations(64, seed=2, device=dev)
    feats = torch.randn(64, 2, 768, 8, 8, generator=torch.Generator().manual_seed(3)).to(dev)
    torch.autograd.set_detect_anomaly(True)
    assert torch.is_grad_enabled()

    def reference_pairs(n_pairs, patched):
        rotate_volume = um.rotate_volume
        errs = []
        for i in range(n_pairs):
            img_feat_src, img_feat_tgt = model(feats[i % 64, 0][None], feats[i % 64, 1][None])
            gt_src_2_tgt_R = gt[i % 64][None]
            if patched == "B":   # INTEGRATION.md option B: the two-line change
                _, key = model.feature_aligner.verify_hypotheses(img_feat_src, img_feat_tgt, proposals)   # added by patch.install()
                pred_sim, pred_index, pred_src_2_tgt_R = ops.select_rotation(key, proposals)
            else:                # test_co3d.py:135-146, verbatim
                B, C, D, H, W = img_feat_src.shape
                img_feat_src_2_tgt = [rotate_volume(img_feat[None].expand(proposals.shape[0], -1, -1, -1, -1), proposals) for img_feat in img_feat_src]
                img_feat_src_2_tgt = torch.stack(img_feat_src_2_tgt).reshape(-1, C, D, H, W)
                img_feat_src_2_tgt = model.feature_aligner.forward_3d2d(img_feat_src_2_tgt).reshape(B, proposals.shape[0], -1, H*W)
                img_feat_tgt = model.feature_aligner.forward_3d2d(img_feat_tgt)
                pred_sim = (img_feat_src_2_tgt * img_feat_tgt[:, None]).sum(dim=2).mean(dim=-1)
                pred_sim, pred_index = torch.max(pred_sim, dim=1)
                pred_src_2_tgt_R = proposals[pred_index]
            sim = (torch.sum(pred_src_2_tgt_R.view(-1, 9) * gt_src_2_tgt_R.view(-1, 9), dim=-1).clamp(-1, 3) - 1) / 2
            err = torch.arccos(sim) * 180. / np.pi
            errs.append(err.mean().item())   # the reference's host sync per pair (test_co3d.py:152)
        return errs

    rows = {}
    try:
        # "A": the unchanged lines with the default install (round 6: the score lines deferred into one fused launch);
        # "A_oplevel": install(defer=False) -- every line its own kernel, the tensors materialised as the reference does;
        # "B": the two-line change
        for mode in ("A", "A_oplevel", "B"):
            ahv.patch.install(um, mm, defer=(mode != "A_oplevel"))
            reference_pairs(8, mode)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            t0 = time.perf_counter()
            errs = reference_pairs(64, mode)
            torch.cuda.synchronize()
            rows[mode] = ((time.perf_counter() - t0) / 64 * 1e3, errs, torch.cuda.max_memory_allocated(dev) / 1e9)
        calls = dict(ahv.patch.calls)
        deferred_counters = dict(ahv.deferred.counters)
    finally:
        ahv.patch.uninstall()
        torch.autograd.set_detect_anomaly(False)
    for m in ("A", "A_oplevel"):
        assert rows[m][1] == rows["B"][1] or max(abs(a - b) for a, b in zip(rows[m][1], rows["B"][1])) < 1e-3
    print(json.dumps({"config": "evaluation loop under the reference script's conditions (grad mode on, anomaly mode on, model.eval()), "
                                "one pair at a time, N=50000",
                      "option_A_unchanged_lines_ms_per_pair": rows["A"][0],
                      "option_A_op_level_kernels_ms_per_pair": rows["A_oplevel"][0],
                      "option_B_fused_launch_ms_per_pair": rows["B"][0],
                      "option_A_peak_memory_GB": rows["A"][2], "option_A_op_level_peak_memory_GB": rows["A_oplevel"][2],
                      "option_B_peak_memory_GB": rows["B"][2], "same_predictions": True,
                      "patch_calls": calls, "deferred_counters": deferred_counters}))
