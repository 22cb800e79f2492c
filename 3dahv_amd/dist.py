"""Sharding the hypothesis axis across GPUs (one process per GPU, RCCL over xGMI).

Hypotheses are independent given the per-pair constants (source volume, target
feature, head weights: 94 KB, replicated on every rank), so N is partitioned
contiguously: rank r scores [lo_r, hi_r) with ``n_offset = lo_r``.  The data
path has ONE exchange: an all-reduce(max) of B packed 64-bit keys
``(ordered_i32(score) << 32) | (0xFFFFFFFF - global_idx)``, i.e. 8*B bytes --
latency-bound, bandwidth-irrelevant.  Signed max on the key = largest score,
lowest global index among equal scores (torch.max semantics, test_co3d.py:145),
independent of how N was split.

The kernels pack the key in SIGNED int64 order (include/ahv.h, "Packed keys"), which is
what ``ReduceOp.MAX`` on an int64 tensor computes on ``nccl`` (= RCCL) and ``gloo`` alike:
the all-reduce takes the key as it is, with no re-encoding launch on either side.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

KEY_EMPTY = -(1 << 63)  # AHV_KEY_EMPTY: below every real key ("nothing scored")


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of range(n): sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (n * rank) // world, (n * (rank + 1)) // world


def merge_keys(keys: torch.Tensor) -> torch.Tensor:
    """Max over dim 0 of packed int64 keys (what the all-reduce computes): (G,B) -> (B,)."""
    return keys.max(dim=0).values


def all_reduce_best(key: torch.Tensor, group=None) -> torch.Tensor:
    """In-place global merge of per-rank packed keys (B,) int64; returns ``key``."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
    return key


# ---- host-side key codec (numpy): used by host logic and CPU tests ------------------

def pack_keys_host(scores: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Same packing as the device code (csrc/ahv_device.h pack_key); returns int64 keys."""
    s = np.asarray(scores, dtype=np.float32) + np.float32(0.0)
    u = s.view(np.uint32).copy()
    u[np.isnan(s)] = np.uint32(0x7FC00000)
    neg = (u & np.uint32(0x80000000)) != 0
    u = np.where(neg, u ^ np.uint32(0x7FFFFFFF), u).astype(np.uint64)
    k = (u << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(idx, dtype=np.uint64))
    return k.view(np.int64)


def unpack_keys_host(keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    ks = np.asarray(keys, dtype=np.int64)
    k = ks.view(np.uint64)
    u = (k >> np.uint64(32)).astype(np.uint32)
    neg = (u & np.uint32(0x80000000)) != 0
    bits = np.where(neg, u ^ np.uint32(0x7FFFFFFF), u).astype(np.uint32)
    score = bits.view(np.float32).copy()
    idx = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    empty = ks == KEY_EMPTY
    score[empty] = -np.inf
    idx[empty] = -1
    return score, idx


# ---- sharded verify step -------------------------------------------------------------

def score_hypotheses_sharded(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1, W2, b2,
                             rank: Optional[int] = None, world: Optional[int] = None, group=None,
                             want_scores: bool = False, r_is_local: bool = False, n_total: Optional[int] = None,
                             score_fn: Optional[Callable] = None):
    """One verify step with N sharded over the process group.

    ``R`` is the full (N,3,3) hypothesis set present on every rank (the reference samples one
    codebook per category and reuses it, test_co3d.py:106) unless ``r_is_local``; in that case it
    is this rank's slice and ``n_total`` gives N.  Returns ``(local_scores or None, best_score (B,),
    best_idx (B,) global int64)`` -- identical on every rank.
    ``score_fn`` defaults to the HIP fused scorer; tests inject a CPU scorer to exercise the
    partition/merge logic without a GPU.
    """
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if score_fn is None:
        from . import ops
        score_fn = ops.score_hypotheses
        unpack = ops.unpack_best
    else:
        unpack = None
    if r_is_local:
        if n_total is None:
            raise ValueError("n_total is required with r_is_local")
        lo, hi = shard_range(n_total, rank, world)
        R_loc = R
        if R.shape[-3] != hi - lo:
            raise ValueError("local R has %d hypotheses, shard expects %d" % (R.shape[-3], hi - lo))
    else:
        n = R.shape[-3]
        lo, hi = shard_range(n, rank, world)
        R_loc = R[..., lo:hi, :, :]
    scores, key = score_fn(vol_src, feat_tgt, R_loc.contiguous(), W1, W2, b2, n_offset=lo, want_scores=want_scores)
    key = all_reduce_best(key, group=group)
    if unpack is not None:
        best, idx = unpack(key)
    else:
        b, i = unpack_keys_host(key.cpu().numpy())
        best, idx = torch.from_numpy(b), torch.from_numpy(i)
    return scores, best, idx


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20) -> int:
    """Data-parallel gradient averaging, the exchange Lightning's DDP strategy performs for the reference's
    ``trainer.fit`` (modules/model_co3d.py:101-145 with ``strategy='ddp'``, train_estimator_co3d.py:23-24).  Gradients are packed
    into flat buckets of up to ``bucket_bytes`` (the aligner has ~230 tensors / 192 MB: a few large RCCL
    all-reduces instead of hundreds of small ones), summed, divided by the world size and unpacked in place.
    Parameters without a gradient on this rank (``bn_down``) are skipped on every rank alike.  Returns the
    number of buckets.  With no process group it is a no-op."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and (size + nbytes > bucket_bytes or g.dtype != cur[0].dtype or g.device != cur[0].device):
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nbytes
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch._utils._flatten_dense_tensors(b)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        for g, f in zip(b, torch._utils._unflatten_dense_tensors(flat, b)):
            g.copy_(f)
    return len(buckets)
