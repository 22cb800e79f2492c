#!/usr/bin/env python3
"""Bank-conflict simulation behind the lane map and the source-image geometry of the fused scorer's gather
(3dahv_amd/csrc/ahv_device.h: kSrcRowsY / kSrcPlaneRows; ahv_dual.h: lane_vox).

Model (MI355X_MICROARCH.md, LDS): a wave64 ds_read_b128 is served in four groups of 16 lanes
({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32), one LDS cycle per group when its 16 lanes touch 16 different
16-byte slots of a 256-byte line; lanes that read the SAME address broadcast; every extra distinct address on a busy slot
costs one more cycle.  A gather instruction reads, for every lane, one 16-byte chunk of the row of a voxel's clamped base
corner (jz, jy, jx) + a constant corner offset, so only the base rows matter.  Rows are 80 bytes: the slot of row r is
5 r mod 16, a unit multiple of r mod 16.

Prints LDS cycles per conflict-free cycle, averaged over Haar rotations, for the x-run lane map of rounds 1-2 and the
4 x 2 x 2-box lane map of round 3 over the row strides (Sy, Sz) that fit the LDS budget.  Measured on the GPU
(SQ_LDS_BANK_CONFLICT per hypothesis / 1024 gather cycles + 1): 2.44 for (x-run, 8, 74), 1.64-1.67 for (box, 9, 76).
"""
import argparse
import numpy as np

GROUPS = np.array([[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                   [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]])
GROUPS = np.concatenate([GROUPS, GROUPS + 32])  # (4, 16) lanes


def haar(n, rng):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, i, j, k = q.T
    return np.stack([1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r), 2 * (i * j + k * r),
                     1 - 2 * (i * i + k * k), 2 * (j * k - i * r), 2 * (i * k - j * r), 2 * (j * k + i * r),
                     1 - 2 * (i * i + j * j)], 1).reshape(n, 3, 3)


def base_voxels(R, vox):
    """Clamped base corner (jx, jy, jz) of output voxels `vox` (..., 3) = (x, y, z) under rotations R (n, 3, 3):
    F.affine_grid + F.grid_sample(align_corners=False) coordinates, base = clamp(floor(i), 0, 6) (ahv_dual.h)."""
    p = (2 * vox + 1) / 8.0 - 1.0
    g = np.einsum("nab,...b->n...a", R, p)
    i = ((g + 1) * 8 - 1) / 2
    return np.clip(np.floor(i), 0, 6).astype(np.int64)


def lane_map_xrun():
    """Rounds 1-2: lane -> (x = lane & 7, z = (lane >> 3) & 1, y = 2 * ((lane >> 5) & 1) + ((lane >> 4) & 1) + 4 * pass)."""
    m = np.zeros((4, 2, 64, 3))
    l = np.arange(64)
    for Q in range(4):
        for p in range(2):
            m[Q, p] = np.stack([l & 7, 2 * ((l >> 5) & 1) + ((l >> 4) & 1) + 4 * p, 2 * Q + ((l >> 3) & 1)], 1)
    return m


def lane_map_box():
    """Round 3 (lane_vox): each b128 lane group owns a 4 (x) x 2 (y) x 2 (z) block."""
    m = np.zeros((4, 2, 64, 3))
    for Q in range(4):
        for p in range(2):
            for g in range(4):
                for k, lane in enumerate(GROUPS[g]):
                    m[Q, p, lane] = (4 * (g & 1) + ((k >> 1) & 3), 2 * (g >> 1) + (k >> 3) + 4 * p, 2 * Q + (k & 1))
    return m


_TRI = np.tril(np.ones((16, 16), bool), -1)


def cycles(rows):
    """rows (..., 16): distinct row indices of the 16 lanes of a group -> LDS cycles of that group."""
    slot = rows % 16
    eq = rows[..., :, None] == rows[..., None, :]
    first = ~np.any(eq & _TRI, axis=-1)                      # first lane of every distinct address
    same = slot[..., :, None] == slot[..., None, :]
    return np.sum(same & first[..., None, :], axis=-1).max(-1)


def factor(lane_map, Sy, Sz, R):
    j = base_voxels(R, lane_map.reshape(-1, 3)).reshape(len(R), 4, 2, 64, 3)
    rows = j[..., 0] + Sy * j[..., 1] + Sz * j[..., 2]
    return float(cycles(rows[..., GROUPS]).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rotations", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    R = haar(args.rotations, np.random.default_rng(args.seed))
    maps = {"x-run (rounds 1-2)": lane_map_xrun(), "4x2x2 box (round 3)": lane_map_box()}
    print("LDS cycles per conflict-free cycle of the gather's ds_read_b128, %d Haar rotations" % args.rotations)
    print("  dense 8 x 64 rows, x-run map: %.2f" % factor(maps["x-run (rounds 1-2)"], 8, 64, R))
    for name, m in maps.items():
        res = sorted((factor(m, Sy, Sz, R), Sy, Sz) for Sy in (8, 9) for Sz in range(7 * Sy + 8, 77))
        print("  %-20s best (Sy, Sz): %s" % (name, ", ".join("(%d, %d) %.2f" % (a, b, f) for f, a, b in res[:3])))
        for Sy, Sz in ((8, 74), (9, 76)):
            print("  %-20s (%d, %d): %.2f" % (name, Sy, Sz, factor(m, Sy, Sz, R)))


if __name__ == "__main__":
    main()
