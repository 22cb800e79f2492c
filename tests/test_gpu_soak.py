"""Randomised agreement of the fused launch with the op-level pipeline (two independent kernel families) over
many shapes: batch sizes, hypothesis counts around the workgroup / grid granularities, shared and per-sample R,
orthonormal and arbitrary 3x3 matrices."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_vs_oplevel_random_shapes(ahv):
    ops = ahv.ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(2024)
    g = torch.Generator().manual_seed(5)
    W1 = ((torch.rand(32, 384, generator=g) * 2 - 1) / np.sqrt(384.0)).to(dev)
    W2 = ((torch.rand(32, 32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
    b2 = ((torch.rand(32, generator=g) * 2 - 1) / np.sqrt(32.0)).to(dev)
    sizes = [1, 2, 7, 8, 9, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4097, 6000]
    worst = 0.0
    for trial in range(24):
        B = int(rng.choice([1, 1, 2, 3, 5, 17]))
        N = int(rng.choice(sizes))
        per_sample = bool(rng.rand() < 0.4)
        vs = (torch.randn(B, 16, 8, 8, 8, generator=g) * rng.uniform(0.2, 3.0)).to(dev)
        vt = (torch.randn(B, 16, 8, 8, 8, generator=g) * 1.1).to(dev)
        nR = B * N if per_sample else N
        R = ops.random_rotations(nR, seed=trial, device=dev)
        if trial % 5 == 4:  # arbitrary (non-orthonormal) matrices: the reference never assumes rotations
            R = R + 0.3 * torch.randn(nR, 3, 3, generator=g).to(dev)
        Rf = R.reshape(B, N, 3, 3) if per_sample else R
        ft = ops.forward_3d2d(vt, W1, W2, b2)
        s, key = ops.score_hypotheses(vs, ft, Rf, W1, W2, b2)
        ref = torch.empty_like(s)
        for b in range(B):
            Rb = Rf[b] if per_sample else Rf
            rot = ops.rotate_volume(vs[b][None].expand(N, -1, -1, -1, -1), Rb)
            ref[b] = ops.score_features(ops.forward_3d2d(rot, W1, W2, b2)[None], ft[b:b + 1])[0]
        err = ((s - ref).abs() / ref.abs().clamp_min(1e-2)).max().item()
        worst = max(worst, err)
        assert err < 1e-5, (trial, B, N, per_sample, err)
        val, idx = ops.unpack_best(key)
        rv, ri = torch.max(s, dim=1)
        assert torch.equal(idx, ri) and torch.equal(val, rv), (trial, B, N)
    assert worst < 1e-5
