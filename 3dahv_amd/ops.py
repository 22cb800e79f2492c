"""Operator mirrors of the reference's hot-path call surface, backed by libahv_hip.so.

Same names, argument meaning and error behaviour as the reference callables:

* ``rotate_volume(volume, rotation_matrix, padding_mode='zeros')``  -- utils.py:113-131
* ``forward_3d2d(img_feat, W1, W2, b2)``  -- Feature_Aligner.forward_3d2d, modules/modules.py:112-124
* ``score_features`` / ``argmax``  -- the inline lines test_co3d.py:143 / :145
* ``score_hypotheses``  -- all of the above fused into one launch (test_co3d.py:137-145)
* ``score_hypotheses_autograd`` / ``forward_3d2d_autograd`` / ``score_hypotheses_backward``  -- the same with
  autograd edges for training (infoNCE_loss, modules/model_co3d.py:41-61): HIP forward + HIP backward

Tensors must live on the GPU (``torch.device('cuda')`` is HIP on ROCm); launches go
to torch's current stream.  The plain ops carry no autograd graph; the ``*_autograd`` ones do.
"""
from __future__ import annotations

import torch

from . import _lib

_VOL = (16, 8, 8, 8)


class score_variant:
    """Select the fused-scorer kernel: ``ops.score_variant(4)`` as a statement switches for the process,
    ``with ops.score_variant(4): ...`` for a block.  3 = all-fp32 (default); 4 = GEMM1 as split-f16 MFMA
    products with fp32 accumulation (1.85x faster, scores as close to the fp64 truth as the fp32 kernel's --
    DESIGN.md section 4.1); 0-2 = earlier fp32 kernels kept for A/B runs."""

    def __init__(self, variant: int):
        self.prev = _lib.load().ahv_set_option(b"score_variant", int(variant))
        if self.prev < 0:
            _lib.check(self.prev, "ahv_set_option")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _lib.load().ahv_set_option(b"score_variant", self.prev)
        return False


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "3dahv_amd ops run on the GPU only (no CPU fallback); got a tensor on %s" % t.device)
        if t.device != dev:
            raise RuntimeError("Expected all tensors to be on the same device, found %s and %s" % (dev, t.device))
        if t.dtype != torch.float32:
            raise RuntimeError("expected float32 tensors (the reference path is fp32), got %s" % t.dtype)
    return dev


def _head(W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor):
    if W1.numel() != 32 * 384 or W2.numel() != 32 * 32 or b2.numel() != 32:
        raise RuntimeError("head weights must be (32,384[,1,1]), (32,32[,1,1]), (32,)")
    return W1.detach().reshape(32, 384).contiguous(), W2.detach().reshape(32, 32).contiguous(), b2.detach().contiguous()


@torch.no_grad()
def rotate_volume(volume: torch.Tensor, rotation_matrix: torch.Tensor, padding_mode: str = "zeros") -> torch.Tensor:
    """Rotate ``volume (N,C,D,H,W)`` by ``rotation_matrix (N,3,3)``; returns a new contiguous tensor.

    The batch dimension of ``volume`` may be a stride-0 expand of one volume (what the
    reference passes): it is read once, never materialised.
    """
    if padding_mode != "zeros":
        raise NotImplementedError("only padding_mode='zeros' (the only mode the reference uses) is implemented")
    if volume.dim() != 5:
        raise RuntimeError("volume must be 5-D (N,C,D,H,W), got %s" % (tuple(volume.shape),))
    if rotation_matrix.dim() != 3 or tuple(rotation_matrix.shape[1:]) != (3, 3):
        raise RuntimeError("rotation_matrix must be (N,3,3), got %s" % (tuple(rotation_matrix.shape),))
    N, C, D, H, W = volume.shape
    if rotation_matrix.shape[0] != N:
        raise RuntimeError("Expected volume and rotation_matrix to have the same batch size, got %d and %d"
                           % (N, rotation_matrix.shape[0]))
    _need_gpu(volume, rotation_matrix)
    volume = volume.detach()
    if N > 1 and volume.stride(0) == 0 and volume[0].is_contiguous():
        src, stride = volume[0], 0
    else:
        src = volume.contiguous()
        stride = C * D * H * W
    R = rotation_matrix.detach().contiguous()
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=volume.device)
    lib = _lib.load()
    _lib.check(lib.ahv_rotate_volume_f32(src.data_ptr(), stride, R.data_ptr(), N, C, D, H, W, out.data_ptr(),
                                         _stream()), "ahv_rotate_volume_f32")
    return out


@torch.no_grad()
def forward_3d2d(img_feat: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """``(M,16,8,8,8) -> (M,32,64)``: slabs -> conv1x1 -> ReLU -> conv1x1+bias -> L2-normalise."""
    if img_feat.dim() != 5 or tuple(img_feat.shape[1:]) != _VOL:
        raise RuntimeError("img_feat must be (M,16,8,8,8), got %s" % (tuple(img_feat.shape),))
    _need_gpu(img_feat, W1, W2, b2)
    W1, W2, b2 = _head(W1, W2, b2)
    x = img_feat.detach().contiguous()
    M = x.shape[0]
    out = torch.empty((M, 32, 64), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ahv_forward_3d2d_f32(x.data_ptr(), W1.data_ptr(), W2.data_ptr(), b2.data_ptr(), M,
                                        out.data_ptr(), _stream()), "ahv_forward_3d2d_f32")
    return out


@torch.no_grad()
def score_features(f_src: torch.Tensor, f_tgt: torch.Tensor) -> torch.Tensor:
    """``(f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)``: (B,N,32,64),(B,32,64) -> (B,N)."""
    if f_src.dim() != 4 or tuple(f_src.shape[2:]) != (32, 64) or tuple(f_tgt.shape) != (f_src.shape[0], 32, 64):
        raise RuntimeError("expected f_src (B,N,32,64) and f_tgt (B,32,64)")
    _need_gpu(f_src, f_tgt)
    B, N = f_src.shape[:2]
    a, t = f_src.detach().contiguous(), f_tgt.detach().contiguous()
    out = torch.empty((B, N), dtype=torch.float32, device=a.device)
    lib = _lib.load()
    _lib.check(lib.ahv_score_features_f32(a.data_ptr(), t.data_ptr(), B, N, out.data_ptr(), _stream()),
               "ahv_score_features_f32")
    return out


@torch.no_grad()
def unpack_best(best_key: torch.Tensor):
    """Packed keys (B,) int64 -> (best_score (B,) f32, best_idx (B,) int64)."""
    B = best_key.numel()
    score = torch.empty((B,), dtype=torch.float32, device=best_key.device)
    idx = torch.empty((B,), dtype=torch.int64, device=best_key.device)
    lib = _lib.load()
    _lib.check(lib.ahv_unpack_best(best_key.data_ptr(), B, score.data_ptr(), idx.data_ptr(), _stream()),
               "ahv_unpack_best")
    return score, idx


@torch.no_grad()
def argmax(scores: torch.Tensor, n_offset: int = 0, return_key: bool = False):
    """``torch.max(scores, dim=1)`` -> (values, first maximal int64 index); (B,N) -> (B,),(B,)."""
    if scores.dim() != 2:
        raise RuntimeError("scores must be (B,N)")
    _need_gpu(scores)
    B, N = scores.shape
    if N == 0:
        raise RuntimeError("max(): Expected reduction dim 1 to have non-zero size.")
    s = scores.detach().contiguous()
    key = torch.empty((B,), dtype=torch.int64, device=s.device)
    lib = _lib.load()
    _lib.check(lib.ahv_argmax_f32(s.data_ptr(), B, N, n_offset, key.data_ptr(), _lib.AHV_SCORE_RESET_BEST,
                                  _stream()), "ahv_argmax_f32")
    if return_key:
        return key
    return unpack_best(key)


@torch.no_grad()
def score_hypotheses(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor,
                     W2: torch.Tensor, b2: torch.Tensor, n_offset: int = 0, want_scores: bool = True,
                     best_key: torch.Tensor | None = None, reset_best: bool | None = None):
    """Fused hot loop (one launch): returns ``(scores (B,N) or None, best_key (B,) int64)``.

    vol_src (B,16,8,8,8); feat_tgt (B,32,64) = forward_3d2d(vol_tgt); R (N,3,3) shared by the
    batch (modules/model.py:184) or (B,N,3,3) per sample (modules/model.py:51).  ``best_key``
    given: merge into it (chunked / multi-call N) unless ``reset_best``; else a fresh key tensor
    is reset and returned.
    Decode with ``unpack_best``; ``n_offset`` is the global index of R[0] when N is sharded.
    """
    if vol_src.dim() != 5 or tuple(vol_src.shape[1:]) != _VOL:
        raise RuntimeError("vol_src must be (B,16,8,8,8), got %s" % (tuple(vol_src.shape),))
    B = vol_src.shape[0]
    if tuple(feat_tgt.shape) != (B, 32, 64):
        raise RuntimeError("feat_tgt must be (B,32,64), got %s" % (tuple(feat_tgt.shape),))
    if R.dim() == 3 and tuple(R.shape[1:]) == (3, 3):
        N, rstride = R.shape[0], 0
    elif R.dim() == 4 and R.shape[0] == B and tuple(R.shape[2:]) == (3, 3):
        N, rstride = R.shape[1], R.shape[1] * 9
    else:
        raise RuntimeError("R must be (N,3,3) or (B,N,3,3), got %s" % (tuple(R.shape),))
    dev = _need_gpu(vol_src, feat_tgt, R, W1, W2, b2)
    W1, W2, b2 = _head(W1, W2, b2)
    vs, ft, Rc = vol_src.detach().contiguous(), feat_tgt.detach().contiguous(), R.detach().contiguous()
    scores = torch.empty((B, N), dtype=torch.float32, device=dev) if want_scores else None
    if best_key is None:
        best_key = torch.empty((B,), dtype=torch.int64, device=dev)
        reset_best = True
    elif best_key.dtype != torch.int64 or best_key.numel() != B or not best_key.is_cuda:
        raise RuntimeError("best_key must be a GPU int64 tensor of B elements")
    flags = _lib.AHV_SCORE_RESET_BEST if reset_best else 0
    lib = _lib.load()
    _lib.check(lib.ahv_score_hypotheses_f32(vs.data_ptr(), ft.data_ptr(), Rc.data_ptr(), rstride, n_offset,
                                            W1.data_ptr(), W2.data_ptr(), b2.data_ptr(), B, N,
                                            scores.data_ptr() if scores is not None else None,
                                            best_key.data_ptr(), flags, _stream()), "ahv_score_hypotheses_f32")
    return scores, best_key


@torch.no_grad()
def score_hypotheses_backward(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor,
                              W2: torch.Tensor, b2: torch.Tensor, grad_scores: torch.Tensor):
    """Gradients of ``score_hypotheses`` w.r.t. ``(vol_src, feat_tgt, W1, W2, b2)`` given ``dL/dscores (B,N)``
    (three launches; what ``infoNCE_loss`` back-propagates, modules/model_co3d.py:41-61).  R gets no gradient."""
    if vol_src.dim() != 5 or tuple(vol_src.shape[1:]) != _VOL:
        raise RuntimeError("vol_src must be (B,16,8,8,8), got %s" % (tuple(vol_src.shape),))
    B = vol_src.shape[0]
    if tuple(feat_tgt.shape) != (B, 32, 64):
        raise RuntimeError("feat_tgt must be (B,32,64), got %s" % (tuple(feat_tgt.shape),))
    N, rstride = _rot_layout(R, B)
    if tuple(grad_scores.shape) != (B, N):
        raise RuntimeError("grad_scores must be (B,N) = %s, got %s" % ((B, N), tuple(grad_scores.shape)))
    dev = _need_gpu(vol_src, feat_tgt, R, W1, W2, b2, grad_scores)
    W1c, W2c, b2c = _head(W1, W2, b2)
    vs, ft, Rc, gs = (t.detach().contiguous() for t in (vol_src, feat_tgt, R, grad_scores))
    lib = _lib.load()
    nbytes = lib.ahv_score_hypotheses_backward_workspace_bytes(B, N)
    ws = torch.empty((max(nbytes, 16) // 4,), dtype=torch.float32, device=dev)
    g_vol = torch.empty((B,) + _VOL, dtype=torch.float32, device=dev)
    g_ft = torch.empty((B, 32, 64), dtype=torch.float32, device=dev)
    g_W1 = torch.empty((32, 384), dtype=torch.float32, device=dev)
    g_W2 = torch.empty((32, 32), dtype=torch.float32, device=dev)
    g_b2 = torch.empty((32,), dtype=torch.float32, device=dev)
    _lib.check(lib.ahv_score_hypotheses_backward_f32(vs.data_ptr(), ft.data_ptr(), Rc.data_ptr(), rstride,
                                                     W1c.data_ptr(), W2c.data_ptr(), b2c.data_ptr(), B, N,
                                                     gs.data_ptr(), ws.data_ptr(), ws.numel() * 4, g_vol.data_ptr(),
                                                     g_ft.data_ptr(), g_W1.data_ptr(), g_W2.data_ptr(),
                                                     g_b2.data_ptr(), _stream()),
               "ahv_score_hypotheses_backward_f32")
    return g_vol, g_ft, g_W1, g_W2, g_b2


class _ScoreFn(torch.autograd.Function):
    """Differentiable fused scorer: forward = one fused launch, backward = ``score_hypotheses_backward``."""

    @staticmethod
    def forward(ctx, vol_src, feat_tgt, R, W1, W2, b2):
        scores, _ = score_hypotheses(vol_src, feat_tgt, R, W1, W2, b2)
        ctx.save_for_backward(vol_src, feat_tgt, R, W1, W2, b2)
        return scores

    @staticmethod
    def backward(ctx, grad_scores):
        vol_src, feat_tgt, R, W1, W2, b2 = ctx.saved_tensors
        g_vol, g_ft, g_W1, g_W2, g_b2 = score_hypotheses_backward(vol_src, feat_tgt, R, W1, W2, b2, grad_scores)
        return g_vol, g_ft, None, g_W1.reshape(W1.shape), g_W2.reshape(W2.shape), g_b2.reshape(b2.shape)


def score_hypotheses_autograd(vol_src, feat_tgt, R, W1, W2, b2) -> torch.Tensor:
    """``scores (B,N)`` with autograd support for vol_src, feat_tgt and the head weights (training path)."""
    return _ScoreFn.apply(vol_src, feat_tgt, R, W1, W2, b2)


class _Forward3d2dFn(torch.autograd.Function):
    """Differentiable ``forward_3d2d``.  The backward reuses the scorer's: with R = identity the scorer's feature
    IS forward_3d2d(vol) (an identity rotation reproduces the voxels exactly), and for L = sum(dF * f) the
    gradients equal those of 64 * score computed against "target" dF -- so one call with N = 1, R = I,
    feat_tgt = dF and grad_scores = 64 returns d vol, d W1, d W2, d b2."""

    @staticmethod
    def forward(ctx, vol, W1, W2, b2):
        ctx.save_for_backward(vol, W1, W2, b2)
        return forward_3d2d(vol, W1, W2, b2)

    @staticmethod
    def backward(ctx, dF):
        vol, W1, W2, b2 = ctx.saved_tensors
        M = vol.shape[0]
        eye = torch.eye(3, dtype=torch.float32, device=vol.device).reshape(1, 3, 3)
        gs = torch.full((M, 1), 64.0, dtype=torch.float32, device=vol.device)
        g_vol, _, g_W1, g_W2, g_b2 = score_hypotheses_backward(vol, dF.contiguous(), eye, W1, W2, b2, gs)
        return g_vol, g_W1.reshape(W1.shape), g_W2.reshape(W2.shape), g_b2.reshape(b2.shape)


def forward_3d2d_autograd(img_feat, W1, W2, b2) -> torch.Tensor:
    """``forward_3d2d`` with autograd support for the volume and the head weights (training path)."""
    return _Forward3d2dFn.apply(img_feat, W1, W2, b2)


def _rot_layout(R: torch.Tensor, B: int):
    if R.dim() == 3 and tuple(R.shape[1:]) == (3, 3):
        return R.shape[0], 0
    if R.dim() == 4 and R.shape[0] == B and tuple(R.shape[2:]) == (3, 3):
        return R.shape[1], R.shape[1] * 9
    raise RuntimeError("R must be (N,3,3) or (B,N,3,3), got %s" % (tuple(R.shape),))


@torch.no_grad()
def select_rotation(best_key: torch.Tensor, R: torch.Tensor, n_offset: int = 0):
    """(best_score (B,), best_idx (B,) global int64, R_pred (B,3,3)) in ONE launch:
    ``pred_sim, pred_index = torch.max(...)``; ``proposals[pred_index]`` (test_co3d.py:145-146)."""
    B = best_key.numel()
    _need_gpu(R)
    N, rstride = _rot_layout(R, B)
    Rc = R.detach().contiguous()
    dev = Rc.device
    score = torch.empty((B,), dtype=torch.float32, device=dev)
    idx = torch.empty((B,), dtype=torch.int64, device=dev)
    R_out = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _lib.check(lib.ahv_select_rotation_f32(best_key.data_ptr(), Rc.data_ptr(), rstride, n_offset, N, B,
                                           R_out.data_ptr(), score.data_ptr(), idx.data_ptr(), _stream()),
               "ahv_select_rotation_f32")
    return score, idx, R_out


@torch.no_grad()
def compose_rotations(best_key: torch.Tensor, R: torch.Tensor, D: torch.Tensor, n_offset: int = 0,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    """Refinement hypotheses ``out[b, n] = R[idx_b] @ D[n]`` with idx_b decoded from the packed key on the
    device (coarse-to-fine, BASELINE.json configs[4]).  R (N,3,3) or (B,N,3,3); D (N2,3,3) -> (B,N2,3,3)."""
    B = best_key.numel()
    _need_gpu(R, D)
    N, rstride = _rot_layout(R, B)
    if D.dim() != 3 or tuple(D.shape[1:]) != (3, 3):
        raise RuntimeError("D must be (N2,3,3)")
    N2 = D.shape[0]
    Rc, Dc = R.detach().contiguous(), D.detach().contiguous()
    if out is None:
        out = torch.empty((B, N2, 3, 3), dtype=torch.float32, device=Rc.device)
    lib = _lib.load()
    _lib.check(lib.ahv_compose_rotations_f32(best_key.data_ptr(), Rc.data_ptr(), rstride, n_offset, N, Dc.data_ptr(),
                                             N2, B, out.data_ptr(), _stream()), "ahv_compose_rotations_f32")
    return out


@torch.no_grad()
def so3_grid(n_total: int, device, offset: int = 0, n: int | None = None) -> torch.Tensor:
    """Rows ``[offset, offset + n)`` of the deterministic ``n_total``-point super-Fibonacci SO(3) grid, generated on
    the device (``rotations.so3_grid_np`` is the host form of the same grid)."""
    n = n_total - offset if n is None else n
    out = torch.empty((n, 3, 3), dtype=torch.float32, device=device)
    if out.device.type != "cuda":
        raise RuntimeError("3dahv_amd ops run on the GPU only (no CPU fallback); got device %s" % out.device)
    lib = _lib.load()
    _lib.check(lib.ahv_so3_grid_f32(n_total, offset, n, out.data_ptr(), _stream()), "ahv_so3_grid_f32")
    return out


@torch.no_grad()
def random_rotations(n: int, seed: int = 0, offset: int = 0, device=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Drop-in for ``pytorch3d.transforms.random_rotations(n)`` generated on the GPU: Haar-uniform (n,3,3) fp32.
    Rotation i depends only on ``(seed, offset + i)``: ``random_rotations(n, s)[a:b]`` equals
    ``random_rotations(b - a, s, offset=a)``, so every rank can generate its own shard."""
    if out is None:
        out = torch.empty((n, 3, 3), dtype=torch.float32, device=device if device is not None else "cuda")
    if not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n * 9:
        raise RuntimeError("out must be a contiguous float32 GPU tensor of n*9 elements")
    lib = _lib.load()
    _lib.check(lib.ahv_random_rotations_f32(seed & (2**64 - 1), offset, n, out.data_ptr(), _stream()),
               "ahv_random_rotations_f32")
    return out
