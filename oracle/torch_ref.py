"""The reference's op sequence issued with stock torch CPU operators.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py).  This is the "reference CPU
PyTorch path" that bench.py times as `cpu_baseline` (kind "port"): the same
ATen operators the reference calls, in the same order, under no_grad:

    affine_grid -> grid_sample           utils.py:123-129
    3 axis collapses + channel concat    modules/modules.py:115-118
    conv1x1 -> relu -> conv1x1 + bias    modules/modules.py:66-70,120
    normalize(dim=1).flatten(2)          modules/modules.py:122
    mul / sum / mean / max               test_co3d.py:143-146

The reference's Python files are never shipped or executed on the GPU box; that
this restatement equals them is pinned in the authoring container by the
golden vectors (tests/test_oracle.py::test_torch_ref_*).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rotate_volume(volume: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    theta = torch.cat([R, R.new_zeros(R.shape[0], 3, 1)], dim=-1)
    grid = F.affine_grid(theta, list(volume.shape), align_corners=False)
    return F.grid_sample(volume, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def forward_3d2d(vol: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    m, c, d, h, w = vol.shape
    along_z = vol.reshape(m, c * d, h, w)
    along_y = vol.permute(0, 1, 3, 2, 4).reshape(m, c * h, d, w)
    along_x = vol.permute(0, 1, 4, 2, 3).reshape(m, c * w, d, h)
    slabs = torch.cat([along_x, along_y, along_z], dim=1)
    u = F.relu(F.conv2d(slabs, W1.reshape(32, 384, 1, 1)))
    v = F.conv2d(u, W2.reshape(32, 32, 1, 1), b2)
    return F.normalize(v, p=2, dim=1).flatten(2)


@torch.no_grad()
def score_hypotheses(vol_src, vol_tgt, R, W1, W2, b2, chunk: int | None = None):
    """vol_* (B,16,8,8,8); R (N,3,3) shared.  Returns (scores (B,N), best (B,), idx (B,)).

    chunk=None is the reference's behaviour (all N hypotheses materialised at
    once, test_co3d.py:137-140); chunk=k bounds the temporaries.
    """
    B = vol_src.shape[0]
    N = R.shape[0]
    f_tgt = forward_3d2d(vol_tgt, W1, W2, b2)
    step = N if chunk is None else chunk
    parts = []
    for n0 in range(0, N, step):
        Rc = R[n0:n0 + step]
        n = Rc.shape[0]
        rot = torch.stack([rotate_volume(v[None].expand(n, -1, -1, -1, -1), Rc) for v in vol_src])
        f_src = forward_3d2d(rot.reshape(-1, *vol_src.shape[1:]), W1, W2, b2).reshape(B, n, 32, 64)
        parts.append((f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1))
    scores = torch.cat(parts, dim=1)
    best, idx = torch.max(scores, dim=1)
    return scores, best, idx
