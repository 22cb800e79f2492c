"""Rotation-hypothesis generators and the angular error metric (host side).

The reference draws hypotheses with ``pytorch3d.transforms.random_rotations``
(test_co3d.py:106, test_linemod.py:43, modules/model.py:184) -- third-party code
that is not in the reference tree.  Only the *distribution* (Haar-uniform on
SO(3)) matters: hypotheses are inputs of the hot path, not results.  These are
our own samplers: normalised Gaussian quaternions, a deterministic
super-Fibonacci SO(3) grid (BASELINE.json config 3; no grid exists in the
reference) and a local refinement sampler (config 5).
"""
from __future__ import annotations

import math

import numpy as np
import torch


def quaternion_to_matrix_np(q: np.ndarray) -> np.ndarray:
    """(n,4) real-first quaternions (any norm > 0) -> (n,3,3) rotation matrices, float64 maths."""
    q = np.asarray(q, dtype=np.float64)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    s = 2.0 / (q * q).sum(axis=1)
    m = np.stack(
        [
            1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
            s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
            s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j),
        ],
        axis=1,
    )
    return m.reshape(-1, 3, 3)


def haar_rotations_np(n: int, seed: int) -> np.ndarray:
    """n Haar-uniform rotations, float32 (n,3,3); bit-reproducible from `seed`.

    Uses numpy's frozen legacy RandomState stream so that tests on any box can
    regenerate the exact hypothesis set whose SHA-256 is stored in the golden
    fixtures.
    """
    rs = np.random.RandomState(seed)
    q = rs.standard_normal(size=(n, 4))
    return quaternion_to_matrix_np(q).astype(np.float32)


def random_rotations(n: int, dtype=torch.float32, device=None, generator: torch.Generator | None = None):
    """Drop-in for ``pytorch3d.transforms.random_rotations(n)``: Haar-uniform (n,3,3)."""
    q = torch.randn((n, 4), dtype=torch.float64, device=device, generator=generator)
    r, i, j, k = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    m = torch.stack(
        [
            1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
            s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
            s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j),
        ],
        dim=-1,
    )
    return m.reshape(n, 3, 3).to(dtype)


def _sincos_turns(f: np.ndarray):
    """sin(2 pi f), cos(2 pi f) for f in [0, 1) from +, -, * alone (nearest octant + Taylor series on |x| <= pi/8):
    correct to ~1e-16 and, unlike libm / SIMD sin and cos, bit-reproducible on every IEEE-754 machine -- the SHA-256 of
    the grid is stored in a golden fixture and re-derived on another box."""
    f = np.asarray(f, dtype=np.float64)
    k = np.floor(f * 8.0 + 0.5)
    x = (f - k * 0.125) * (2.0 * math.pi)
    x2 = x * x
    sn = np.ones_like(x)
    cs = np.ones_like(x)
    for m in range(9, 0, -1):  # Horner: sin x = x (1 - x2/(2.3) (1 - x2/(4.5) ...)), cos x = 1 - x2/(1.2) (1 - x2/(3.4) ...)
        sn = 1.0 - sn * x2 * (1.0 / ((2 * m) * (2 * m + 1)))
        cs = 1.0 - cs * x2 * (1.0 / ((2 * m - 1) * (2 * m)))
    sn = sn * x
    h = math.sqrt(0.5)
    tab_s = np.array([0.0, h, 1.0, h, 0.0, -h, -1.0, -h])
    tab_c = np.array([1.0, h, 0.0, -h, -1.0, -h, 0.0, h])
    ki = k.astype(np.int64) & 7
    s0, c0 = tab_s[ki], tab_c[ki]
    return s0 * cs + c0 * sn, c0 * cs - s0 * sn


def so3_grid_np(n: int) -> np.ndarray:
    """Deterministic, nearly uniform SO(3) grid of n rotations (super-Fibonacci spiral).

    Build-defined (the reference has no grid: test_linemod.py:43 samples randomly).
    Alexa, "Super-Fibonacci Spirals: Fast, Low-Discrepancy Sampling of SO(3)", CVPR 2022.
    Point i is the quaternion (sqrt(t) sin a, sqrt(t) cos a, sqrt(1-t) sin b, sqrt(1-t) cos b) with t = (i + 1/2)/n,
    a = 2 pi (i + 1/2)/sqrt(2), b = 2 pi (i + 1/2)/psi; the angles are reduced as turn fractions first (they reach
    10^6 rad) and sin / cos come from ``_sincos_turns``, so the float32 result is the same bytes on every machine.
    """
    psi = 1.533751168755204288118041
    i = np.arange(n, dtype=np.float64)
    s = i + 0.5
    t = s / n
    fa, fb = s * math.sqrt(0.5), s * (1.0 / psi)
    fa, fb = fa - np.floor(fa), fb - np.floor(fb)
    r, R = np.sqrt(t), np.sqrt(1.0 - t)
    sa, ca = _sincos_turns(fa)
    sb, cb = _sincos_turns(fb)
    q = np.stack([r * sa, r * ca, R * sb, R * cb], axis=1)
    return quaternion_to_matrix_np(q).astype(np.float32)


def axis_angle_to_matrix(omega: torch.Tensor) -> torch.Tensor:
    """Rodrigues: (n,3) rotation vectors -> (n,3,3)."""
    theta = omega.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = omega / theta
    K = omega.new_zeros(omega.shape[0], 3, 3)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    st, ct = torch.sin(theta)[..., None], torch.cos(theta)[..., None]
    eye = torch.eye(3, dtype=omega.dtype, device=omega.device)[None]
    return eye + st * K + (1 - ct) * (K @ K)


def refine_rotations(R_star: torch.Tensor, n: int, max_angle_deg: float, generator=None) -> torch.Tensor:
    """Local refinement set around R_star (3,3): R_star @ exp([w]x), |w| <= max_angle (config 5).

    Index 0 is R_star itself so that refinement can never lower the best score.
    """
    dev = R_star.device
    v = torch.randn((n, 3), dtype=torch.float64, device=dev, generator=generator)
    v = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    # uniform in the ball of rotation vectors: radius ~ U^(1/3)
    rad = torch.rand((n, 1), dtype=torch.float64, device=dev, generator=generator) ** (1.0 / 3.0)
    w = v * rad * math.radians(max_angle_deg)
    w[0] = 0
    return (R_star.to(torch.float64)[None] @ axis_angle_to_matrix(w)).to(R_star.dtype)


def geodesic_deg(R_pred: torch.Tensor, R_gt: torch.Tensor) -> torch.Tensor:
    """Angular error in degrees, exactly the reference's expression (test_co3d.py:149-150,
    modules/model.py:199-200, test_linemod.py:66-67): arccos(((sum(Rp*Rg)).clamp(-1,3)-1)/2)*180/pi."""
    sim = (torch.sum(R_pred.reshape(-1, 9) * R_gt.reshape(-1, 9), dim=-1).clamp(-1, 3) - 1) / 2
    return torch.arccos(sim) * 180.0 / math.pi
