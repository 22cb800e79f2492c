"""Operator mirrors of the reference's hot-path call surface, backed by libahv_hip.so.

Same names, argument meaning and error behaviour as the reference callables:

* ``rotate_volume(volume, rotation_matrix, padding_mode='zeros')``  -- utils.py:113-131
* ``forward_3d2d(img_feat, W1, W2, b2)``  -- Feature_Aligner.forward_3d2d, modules/modules.py:112-124
* ``score_features`` / ``argmax``  -- the inline lines test_co3d.py:143 / :145
* ``score_hypotheses``  -- all of the above fused into one launch (test_co3d.py:137-145)
* ``verify_pair``  -- the same with ``forward_3d2d(vol_tgt)`` (test_co3d.py:141) inside that launch
* ``score_hypotheses_autograd`` / ``forward_3d2d_autograd`` / ``score_hypotheses_backward``  -- the same with
  autograd edges for training (infoNCE_loss, modules/model_co3d.py:41-61): HIP forward + HIP backward

Tensors must live on the GPU (``torch.device('cuda')`` is HIP on ROCm); launches go
to torch's current stream.  The plain ops carry no autograd graph; the ``*_autograd`` ones do.
"""
from __future__ import annotations

import contextvars

import torch

from . import _lib

_VOL = (16, 8, 8, 8)


_SPLIT_F16 = contextvars.ContextVar("ahv_split_f16", default=False)


class split_f16_scorer:
    """``with ops.split_f16_scorer(): ...`` -- inside the block (this thread / context only) ``score_hypotheses``
    passes ``AHV_SCORE_SPLIT_F16``: both GEMMs as split-f16 MFMA products with fp32 accumulation (2.1x faster, scores as
    close to the fp64 truth as the fp32 kernel's -- DESIGN.md section 4.1).  Opt-in; the default is the all-fp32
    kernel.  The choice travels with each call as a flag bit: the library keeps no process-wide selector, so
    concurrent threads / streams cannot disturb each other.  ``score_hypotheses(..., split_f16=True)`` selects
    it for one call."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)
        self._token = None

    def __enter__(self):
        self._token = _SPLIT_F16.set(self.enabled)
        return self

    def __exit__(self, *exc):
        _SPLIT_F16.reset(self._token)
        return False


def _stream(dev=None) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _call(dev: torch.device, name: str, *args) -> None:
    """Run entry point ``name`` with ``dev`` as the current HIP device (the C side sizes its grids from
    hipGetDevice) on torch's current stream OF THAT DEVICE, so that tensors on cuda:1 are never processed on
    cuda:0's stream while cuda:0 happens to be current.  The stream is the last argument of every entry point."""
    lib = _lib.load()
    if dev.type != "cuda":
        raise RuntimeError("3dahv_amd ops run on the GPU only (no CPU fallback); got a tensor on %s" % dev)
    with torch.cuda.device(dev):
        _lib.check(getattr(lib, name)(*args, torch.cuda.current_stream(dev).cuda_stream), name)


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


def _refuse_grad(name: str, *tensors: torch.Tensor) -> None:
    """Ops without an autograd edge must not swallow a gradient silently."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError("%s has no autograd edge: call it under torch.no_grad() / on detached tensors, or use "
                           "score_hypotheses_autograd / rotate_volume / forward_3d2d, which are differentiable" % name)


def _head(W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor):
    if W1.numel() != 32 * 384 or W2.numel() != 32 * 32 or b2.numel() != 32:
        raise RuntimeError("head weights must be (32,384[,1,1]), (32,32[,1,1]), (32,)")
    return W1.detach().reshape(32, 384).contiguous(), W2.detach().reshape(32, 32).contiguous(), b2.detach().contiguous()


def rotate_volume(volume: torch.Tensor, rotation_matrix: torch.Tensor, padding_mode: str = "zeros") -> torch.Tensor:
    """Rotate ``volume (N,C,D,H,W)`` by ``rotation_matrix (N,3,3)``; returns a new contiguous tensor.

    The batch dimension of ``volume`` may be a stride-0 expand of one volume (what the
    reference passes): it is read once, never materialised.  Differentiable w.r.t. ``volume`` like the
    reference's (utils.py:113-131; infoNCE_loss back-propagates through it, modules/model_co3d.py:49-54): when
    autograd is recording and ``volume`` requires grad the call goes through ``_RotateVolumeFn`` (HIP forward, HIP
    adjoint).  A rotation matrix that requires grad is refused loudly (the reference samples its rotations).
    """
    if torch.is_grad_enabled():
        if rotation_matrix.requires_grad:
            raise NotImplementedError("rotate_volume: no gradient w.r.t. rotation_matrix is implemented (the "
                                      "reference samples its rotations); detach it")
        if volume.requires_grad:
            return _RotateVolumeFn.apply(volume, rotation_matrix)
    return _rotate_volume_nograd(volume, rotation_matrix, padding_mode)


@torch.no_grad()
def _rotate_volume_nograd(volume: torch.Tensor, rotation_matrix: torch.Tensor, padding_mode: str = "zeros") -> torch.Tensor:
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
    _call(out.device, "ahv_rotate_volume_f32", src.data_ptr(), stride, R.data_ptr(), N, C, D, H, W, out.data_ptr())
    return out


def forward_3d2d(img_feat: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """``(M,16,8,8,8) -> (M,32,64)``: slabs -> conv1x1 -> ReLU -> conv1x1+bias -> L2-normalise.
    When autograd is recording and an input requires grad, the call carries an autograd edge (HIP backward,
    ``_Forward3d2dFn``) -- it never silently drops a gradient."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (img_feat, W1, W2, b2)):
        return _Forward3d2dFn.apply(img_feat, W1, W2, b2)
    return _forward_3d2d_nograd(img_feat, W1, W2, b2)


@torch.no_grad()
def _forward_3d2d_nograd(img_feat: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    if img_feat.dim() != 5 or tuple(img_feat.shape[1:]) != _VOL:
        raise RuntimeError("img_feat must be (M,16,8,8,8), got %s" % (tuple(img_feat.shape),))
    _need_gpu(img_feat, W1, W2, b2)
    W1, W2, b2 = _head(W1, W2, b2)
    x = img_feat.detach().contiguous()
    M = x.shape[0]
    out = torch.empty((M, 32, 64), dtype=torch.float32, device=x.device)
    _call(out.device, "ahv_forward_3d2d_f32", x.data_ptr(), W1.data_ptr(), W2.data_ptr(), b2.data_ptr(), M,
          out.data_ptr())
    return out


def score_features(f_src: torch.Tensor, f_tgt: torch.Tensor) -> torch.Tensor:
    """``(f_src * f_tgt[:, None]).sum(dim=2).mean(dim=-1)``: (B,N,32,64),(B,32,64) -> (B,N).  Inference only."""
    _refuse_grad("score_features", f_src, f_tgt)
    if f_src.dim() != 4 or tuple(f_src.shape[2:]) != (32, 64) or tuple(f_tgt.shape) != (f_src.shape[0], 32, 64):
        raise RuntimeError("expected f_src (B,N,32,64) and f_tgt (B,32,64)")
    _need_gpu(f_src, f_tgt)
    B, N = f_src.shape[:2]
    a, t = f_src.detach().contiguous(), f_tgt.detach().contiguous()
    out = torch.empty((B, N), dtype=torch.float32, device=a.device)
    _call(out.device, "ahv_score_features_f32", a.data_ptr(), t.data_ptr(), B, N, out.data_ptr())
    return out


@torch.no_grad()
def unpack_best(best_key: torch.Tensor):
    """Packed keys (B,) int64 -> (best_score (B,) f32, best_idx (B,) int64)."""
    B = best_key.numel()
    score = torch.empty((B,), dtype=torch.float32, device=best_key.device)
    idx = torch.empty((B,), dtype=torch.int64, device=best_key.device)
    _call(best_key.device, "ahv_unpack_best", best_key.data_ptr(), B, score.data_ptr(), idx.data_ptr())
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
    _call(key.device, "ahv_argmax_f32", s.data_ptr(), B, N, n_offset, key.data_ptr(), _lib.AHV_SCORE_RESET_BEST)
    if return_key:
        return key
    return unpack_best(key)


def score_hypotheses(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor,
                     W2: torch.Tensor, b2: torch.Tensor, n_offset: int = 0, want_scores: bool = True,
                     best_key: torch.Tensor | None = None, reset_best: bool | None = None,
                     split_f16: bool | None = None, clock_stamps: torch.Tensor | None = None,
                     no_teams: bool = False, spare_cus: int = 0):
    """Fused hot loop (one launch): returns ``(scores (B,N) or None, best_key (B,) int64)``.

    vol_src (B,16,8,8,8); feat_tgt (B,32,64) = forward_3d2d(vol_tgt); R (N,3,3) shared by the
    batch (modules/model.py:184) or (B,N,3,3) per sample (modules/model.py:51).  ``best_key``
    given: merge into it (chunked / multi-call N) unless ``reset_best``; else a fresh key tensor
    is reset and returned.
    Decode with ``unpack_best``; ``n_offset`` is the global index of R[0] when N is sharded.
    ``split_f16``: opt-in kernel for this call (None: the enclosing ``split_f16_scorer`` block, else fp32).
    ``clock_stamps`` (int64, 4 * CU count, zeroed): diagnostic launch that also records the shader clock.
    ``no_teams``: AHV_SCORE_NO_TEAMS -- every hypothesis by one wave.  A scheduling knob only: by default a launch's
    remainder goes to teams of four waves, whose score is the lone wave's bit for bit (a score is a function of the
    volumes, the weights and R_n alone -- not of N, of ``n_offset`` or of how a set is sharded); ``spare_cus``: compute
    units left without a workgroup of the persistent grid (a measurement knob, include/ahv_diag.h).
    With autograd recording and an input that requires grad, the returned scores carry the autograd edge of
    ``score_hypotheses_autograd`` (HIP backward) -- like the reference's op sequence, nothing is silently detached.
    """
    if (want_scores and clock_stamps is None and torch.is_grad_enabled()
            and any(t.requires_grad for t in (vol_src, feat_tgt, W1, W2, b2))):
        return _ScoreFn.apply(vol_src, feat_tgt, R, W1, W2, b2, n_offset, best_key, reset_best, split_f16)
    with torch.no_grad():
        return _score_hypotheses_nograd(vol_src, feat_tgt, R, W1, W2, b2, n_offset, want_scores, best_key, reset_best,
                                        split_f16, clock_stamps, no_teams=no_teams, spare_cus=spare_cus)


def score_plan(B: int, N: int, no_teams: bool = False, split_f16: bool = False, spare_cus: int = 0):
    """How a launch of B samples x N hypotheses is laid out on the current device (``ahv_diag_score_plan``, pure host
    arithmetic): ``(gx, gy, n_main)`` -- the persistent grid and how many hypotheses of each sample go to single waves; the
    rest, ``[n_main, N)``, go to teams of four.  Scores do not depend on it (a team's score is a lone wave's bit for bit)."""
    import ctypes
    flags = ((_lib.AHV_SCORE_NO_TEAMS if no_teams else 0) | (_lib.AHV_SCORE_SPLIT_F16 if split_f16 else 0) |
             (int(spare_cus) << _lib.AHV_SCORE_SPARE_CUS_SHIFT))
    gx, gy, n_main = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    _lib.check(_lib.load().ahv_diag_score_plan(B, N, flags, ctypes.byref(gx), ctypes.byref(gy), ctypes.byref(n_main)),
               "ahv_diag_score_plan")
    return gx.value, gy.value, n_main.value


def verify_pair(vol_src: torch.Tensor, vol_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor,
                b2: torch.Tensor, n_offset: int = 0, want_scores: bool = True, best_key: torch.Tensor | None = None,
                reset_best: bool | None = None, split_f16: bool | None = None, want_feat_tgt: bool = False,
                clock_stamps: torch.Tensor | None = None, no_teams: bool = False, spare_cus: int = 0):
    """The whole per-pair verify step of test_co3d.py:137-145 behind ONE entry point (``ahv_verify_pair_f32``):
    ``forward_3d2d(vol_tgt)`` is built inside the scoring launch instead of in a launch of its own.  Arguments as
    ``score_hypotheses`` with the target VOLUME ``vol_tgt (B,16,8,8,8)`` in place of ``feat_tgt``.  Returns
    ``(scores (B,N) or None, best_key (B,) int64)`` and, with ``want_feat_tgt``, the target features ``(B,32,64)`` as
    third element (always materialised for the split-f16 kernel, which runs forward_3d2d as a launch of its own).
    Inference only (no autograd edge): with autograd recording and an input that requires grad it REFUSES (the check runs
    before the no_grad block -- as a decorator the block hid the recording state from the check, round 4) -- use
    ``forward_3d2d`` + ``score_hypotheses``, which carry the HIP backward."""
    _refuse_grad("verify_pair", vol_src, vol_tgt, W1, W2, b2)
    if vol_tgt.dim() != 5 or tuple(vol_tgt.shape) != tuple(vol_src.shape):
        raise RuntimeError("vol_tgt must have vol_src's shape (B,16,8,8,8), got %s" % (tuple(vol_tgt.shape),))
    split = bool(_SPLIT_F16.get() if split_f16 is None else split_f16)
    with torch.no_grad():
        feat = None
        if want_feat_tgt or split:
            feat = torch.empty((vol_src.shape[0], 32, 64), dtype=torch.float32, device=vol_src.device)
        scores, key = _score_hypotheses_nograd(vol_src, vol_tgt, R, W1, W2, b2, n_offset, want_scores, best_key, reset_best,
                                               split, clock_stamps, no_teams=no_teams, spare_cus=spare_cus,
                                               tgt_is_volume=True, feat_tgt_out=feat)
    return (scores, key, feat) if want_feat_tgt else (scores, key)


def _score_hypotheses_nograd(vol_src, feat_tgt, R, W1, W2, b2, n_offset, want_scores, best_key, reset_best, split_f16,
                             clock_stamps, no_teams=False, spare_cus=0, tgt_is_volume=False, feat_tgt_out=None):
    if vol_src.dim() != 5 or tuple(vol_src.shape[1:]) != _VOL:
        raise RuntimeError("vol_src must be (B,16,8,8,8), got %s" % (tuple(vol_src.shape),))
    B = vol_src.shape[0]
    if not tgt_is_volume and tuple(feat_tgt.shape) != (B, 32, 64):
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
    if _SPLIT_F16.get() if split_f16 is None else split_f16:
        flags |= _lib.AHV_SCORE_SPLIT_F16
    if no_teams:
        flags |= _lib.AHV_SCORE_NO_TEAMS
    if not 0 <= int(spare_cus) <= 255:
        raise RuntimeError("spare_cus must be in 0..255")
    flags |= int(spare_cus) << _lib.AHV_SCORE_SPARE_CUS_SHIFT
    lib = _lib.load()
    head = (vs.data_ptr(), ft.data_ptr(), Rc.data_ptr(), rstride, n_offset, W1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
            B, N, scores.data_ptr() if scores is not None else None, best_key.data_ptr())
    if clock_stamps is not None and (clock_stamps.dtype != torch.int64 or clock_stamps.device != dev or
                                     clock_stamps.numel() < 4 * lib.ahv_device_cu_count()):
        raise RuntimeError("clock_stamps must be an int64 tensor of 4 * CU-count elements on %s" % dev)
    with torch.cuda.device(dev):
        if tgt_is_volume:
            _lib.check(lib.ahv_verify_pair_f32(*head, feat_tgt_out.data_ptr() if feat_tgt_out is not None else None, flags,
                                               clock_stamps.data_ptr() if clock_stamps is not None else None,
                                               _stream(dev)), "ahv_verify_pair_f32")
        elif clock_stamps is None:
            _lib.check(lib.ahv_score_hypotheses_f32(*head, flags, _stream(dev)), "ahv_score_hypotheses_f32")
        else:
            _lib.check(lib.ahv_score_hypotheses_clocked_f32(*head, flags, clock_stamps.data_ptr(), _stream(dev)),
                       "ahv_score_hypotheses_clocked_f32")
    return scores, best_key


@torch.no_grad()
def score_hypotheses_train(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor,
                           b2: torch.Tensor):
    """The training forward: ``scores (B,N)`` as ``score_hypotheses`` computes them, plus the workspace in which the launch
    left every hypothesis' pre-activations (8 KB each) for ``score_hypotheses_backward(..., workspace=ws)`` -- the backward
    then skips the recompute of rotate_volume + the first projection (``ahv_score_hypotheses_train_f32``).  Returns
    ``(scores, workspace)``; the workspace serves ONE backward."""
    if vol_src.dim() != 5 or tuple(vol_src.shape[1:]) != _VOL:
        raise RuntimeError("vol_src must be (B,16,8,8,8), got %s" % (tuple(vol_src.shape),))
    B = vol_src.shape[0]
    if tuple(feat_tgt.shape) != (B, 32, 64):
        raise RuntimeError("feat_tgt must be (B,32,64), got %s" % (tuple(feat_tgt.shape),))
    N, rstride = _rot_layout(R, B)
    dev = _need_gpu(vol_src, feat_tgt, R, W1, W2, b2)
    W1c, W2c, b2c = _head(W1, W2, b2)
    vs, ft, Rc = (t.detach().contiguous() for t in (vol_src, feat_tgt, R))
    lib = _lib.load()
    nbytes = lib.ahv_score_hypotheses_backward_workspace_bytes(B, N)
    ws = torch.empty((max(nbytes, 16) // 4,), dtype=torch.float32, device=dev)
    scores = torch.empty((B, N), dtype=torch.float32, device=dev)
    _call(dev, "ahv_score_hypotheses_train_f32", vs.data_ptr(), ft.data_ptr(), Rc.data_ptr(), rstride, W1c.data_ptr(),
          W2c.data_ptr(), b2c.data_ptr(), B, N, scores.data_ptr(), ws.data_ptr(), ws.numel() * 4)
    return scores, ws


@torch.no_grad()
def score_hypotheses_backward(vol_src: torch.Tensor, feat_tgt: torch.Tensor, R: torch.Tensor, W1: torch.Tensor,
                              W2: torch.Tensor, b2: torch.Tensor, grad_scores: torch.Tensor, workspace: torch.Tensor | None = None):
    """Gradients of ``score_hypotheses`` w.r.t. ``(vol_src, feat_tgt, W1, W2, b2)`` given ``dL/dscores (B,N)``
    (three launches; what ``infoNCE_loss`` back-propagates, modules/model_co3d.py:41-61).  R gets no gradient.
    ``workspace``: what ``score_hypotheses_train`` returned for the SAME inputs -- the first kernel then reads the saved
    pre-activations instead of recomputing the forward (and consumes them: one backward per workspace)."""
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
    saved = workspace is not None
    if saved:
        if workspace.device != dev or workspace.dtype != torch.float32 or workspace.numel() * 4 < nbytes:
            raise RuntimeError("workspace is not the one score_hypotheses_train returned for these shapes")
        ws = workspace
    else:
        ws = torch.empty((max(nbytes, 16) // 4,), dtype=torch.float32, device=dev)
    g_vol = torch.empty((B,) + _VOL, dtype=torch.float32, device=dev)
    g_ft = torch.empty((B, 32, 64), dtype=torch.float32, device=dev)
    g_W1 = torch.empty((32, 384), dtype=torch.float32, device=dev)
    g_W2 = torch.empty((32, 32), dtype=torch.float32, device=dev)
    g_b2 = torch.empty((32,), dtype=torch.float32, device=dev)
    _call(dev, "ahv_score_hypotheses_backward_saved_f32" if saved else "ahv_score_hypotheses_backward_f32", vs.data_ptr(),
          ft.data_ptr(), Rc.data_ptr(), rstride,
          W1c.data_ptr(), W2c.data_ptr(), b2c.data_ptr(), B, N, gs.data_ptr(), ws.data_ptr(), ws.numel() * 4,
          g_vol.data_ptr(), g_ft.data_ptr(), g_W1.data_ptr(), g_W2.data_ptr(), g_b2.data_ptr())
    return g_vol, g_ft, g_W1, g_W2, g_b2


class _RotateVolumeFn(torch.autograd.Function):
    """Differentiable ``rotate_volume`` (w.r.t. the volume): HIP gather forward, HIP scatter adjoint.  A stride-0
    batch (``v[None].expand(N, ...)``) is read once in the forward; its gradient is accumulated into ONE volume on
    the device.  Autograd's expand-backward then sums the N rows of what this function returns, so row 0 carries
    that volume and the other rows are zero (exact; no division by N)."""

    @staticmethod
    def forward(ctx, volume, rotation_matrix):
        out = _rotate_volume_nograd(volume, rotation_matrix)
        N = volume.shape[0]
        ctx.shared = bool(N > 1 and volume.stride(0) == 0 and volume[0].is_contiguous())
        ctx.vshape = tuple(volume.shape)
        ctx.save_for_backward(rotation_matrix.detach().contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (R,) = ctx.saved_tensors
        N, C, D, H, W = ctx.vshape
        g = grad_out.contiguous()
        dev = g.device
        if ctx.shared:
            gv = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
            _call(dev, "ahv_rotate_volume_backward_f32", g.data_ptr(), 0, R.data_ptr(), N, C, D, H, W, gv.data_ptr())
            # the expand's backward sums over the batch: put the whole sum in row 0
            full = torch.zeros((N, C, D, H, W), dtype=torch.float32, device=dev)
            full[0] = gv[0]
            return full, None
        gv = torch.empty((N, C, D, H, W), dtype=torch.float32, device=dev)
        _call(dev, "ahv_rotate_volume_backward_f32", g.data_ptr(), C * D * H * W, R.data_ptr(), N, C, D, H, W,
              gv.data_ptr())
        return gv, None


class _ScoreFn(torch.autograd.Function):
    """Differentiable fused scorer: forward = one fused launch, backward = ``score_hypotheses_backward``."""

    @staticmethod
    def forward(ctx, vol_src, feat_tgt, R, W1, W2, b2, n_offset=0, best_key=None, reset_best=None, split_f16=None, need_key=True):
        ctx.ws = None
        plain = n_offset == 0 and best_key is None and not split_f16 and (split_f16 is not None or not _SPLIT_F16.get())
        if plain:
            # the training forward: the same scores, and the pre-activations kept for the backward (no recompute there);
            # the arg-max key of the inference launch is not produced -- callers of the differentiable form use the scores
            scores, ctx.ws = score_hypotheses_train(vol_src, feat_tgt, R, W1, W2, b2)
            if need_key:   # (score_hypotheses' contract: the packed arg-max key beside the scores)
                key = argmax(scores, return_key=True)
            else:
                key = torch.full((vol_src.shape[0],), _lib.AHV_KEY_EMPTY, dtype=torch.int64, device=scores.device)
        else:
            scores, key = _score_hypotheses_nograd(vol_src, feat_tgt, R, W1, W2, b2, n_offset, True, best_key, reset_best,
                                                   split_f16, None)
        ctx.save_for_backward(vol_src, feat_tgt, R, W1, W2, b2)
        ctx.mark_non_differentiable(key)
        return scores, key

    @staticmethod
    def backward(ctx, grad_scores, _grad_key):
        vol_src, feat_tgt, R, W1, W2, b2 = ctx.saved_tensors
        ws, ctx.ws = ctx.ws, None   # one backward per workspace: a second one (retain_graph) recomputes the forward
        g_vol, g_ft, g_W1, g_W2, g_b2 = score_hypotheses_backward(vol_src, feat_tgt, R, W1, W2, b2,
                                                                  grad_scores.contiguous(), workspace=ws)
        return (g_vol, g_ft, None, g_W1.reshape(W1.shape), g_W2.reshape(W2.shape), g_b2.reshape(b2.shape),
                None, None, None, None, None)


def score_hypotheses_autograd(vol_src, feat_tgt, R, W1, W2, b2) -> torch.Tensor:
    """``scores (B,N)`` with autograd support for vol_src, feat_tgt and the head weights (training path)."""
    return _ScoreFn.apply(vol_src, feat_tgt, R, W1, W2, b2, 0, None, None, None, False)[0]


class _Forward3d2dFn(torch.autograd.Function):
    """Differentiable ``forward_3d2d``.  The backward reuses the scorer's: with R = identity the scorer's feature
    IS forward_3d2d(vol) (an identity rotation reproduces the voxels exactly), and for L = sum(dF * f) the
    gradients equal those of 64 * score computed against "target" dF -- so one call with N = 1, R = I,
    feat_tgt = dF and grad_scores = 64 returns d vol, d W1, d W2, d b2."""

    @staticmethod
    def forward(ctx, vol, W1, W2, b2):
        ctx.save_for_backward(vol, W1, W2, b2)
        return _forward_3d2d_nograd(vol, W1, W2, b2)

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
def reset_best(best_key: torch.Tensor) -> torch.Tensor:
    """``best_key[:] = AHV_KEY_EMPTY`` (one tiny launch): a key tensor ready to be merged into."""
    if best_key.dtype != torch.int64 or not best_key.is_cuda:
        raise RuntimeError("best_key must be a GPU int64 tensor")
    _call(best_key.device, "ahv_reset_best", best_key.data_ptr(), best_key.numel())
    return best_key


@torch.no_grad()
def select_rotation(best_key: torch.Tensor, R: torch.Tensor, n_offset: int = 0, reset_key: bool = False):
    """(best_score (B,), best_idx (B,) global int64, R_pred (B,3,3)) in ONE launch:
    ``pred_sim, pred_index = torch.max(...)``; ``proposals[pred_index]`` (test_co3d.py:145-146).
    ``reset_key``: hand ``best_key`` back EMPTY (AHV_SELECT_RESET_KEY), ready for the next step's scorer."""
    B = best_key.numel()
    _need_gpu(R)
    N, rstride = _rot_layout(R, B)
    Rc = R.detach().contiguous()
    dev = Rc.device
    score = torch.empty((B,), dtype=torch.float32, device=dev)
    idx = torch.empty((B,), dtype=torch.int64, device=dev)
    R_out = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    _call(dev, "ahv_select_rotation_f32", best_key.data_ptr(), Rc.data_ptr(), rstride, n_offset, N, B,
          R_out.data_ptr(), score.data_ptr(), idx.data_ptr(), _lib.AHV_SELECT_RESET_KEY if reset_key else 0)
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
    _call(out.device, "ahv_compose_rotations_f32", best_key.data_ptr(), Rc.data_ptr(), rstride, n_offset, N,
          Dc.data_ptr(), N2, B, out.data_ptr())
    return out


class CoarseToFineState:
    """Scratch of ``coarse_to_fine`` for B samples on one device: the two packed keys (EMPTY between steps), the meeting
    point's counters (zero between steps) and the launch's error word.  One per caller and stream; reusing it keeps the
    step free of clearing launches."""

    def __init__(self, B: int, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("3dahv_amd ops run on the GPU only (no CPU fallback); got device %s" % dev)
        self.B = B
        self.keys = torch.full((2, B), _lib.AHV_KEY_EMPTY, dtype=torch.int64, device=dev)
        self.sync = torch.zeros((2 * B + 1,), dtype=torch.int32, device=dev)

    def gave_up(self) -> bool:
        """True if a workgroup of some step abandoned the meeting point (host sync).  The flag is STICKY: the kernel never
        clears it, and every later step on this state poisons its outputs (NaN / -1) too until ``clear_error()``."""
        return bool(self.sync[-1].item() != 0)

    def clear_error(self) -> None:
        """Make the state usable again after a give-up: keys back to empty, the meeting point's counters to zero (a step
        that gave up may have left them mid-count) and the error word cleared -- on the current stream, after every launch
        that used the state."""
        self.keys.fill_(_lib.AHV_KEY_EMPTY)
        self.sync.zero_()


def coarse_to_fine(vol_src: torch.Tensor, vol_tgt: torch.Tensor, R: torch.Tensor, D: torch.Tensor, W1: torch.Tensor,
                   W2: torch.Tensor, b2: torch.Tensor, state: CoarseToFineState | None = None, want_scores: bool = False,
                   want_feat_tgt: bool = False, no_teams: bool = False, spare_cus: int = 0, out: dict | None = None) -> dict:
    """BASELINE.json configs[4] on one rank as ONE launch (``ahv_coarse_to_fine_f32``): the verify step on the coarse set
    ``R (N,3,3) or (B,N,3,3)``, then -- behind a device-wide meeting point -- on the refinements ``R* @ D[n]`` of its winner,
    ``D (N2,3,3)``; the last workgroup decodes both keys.  Returns a dict with ``fine_score, fine_idx`` (index into D),
    ``R_pred (B,3,3)``, ``coarse_score, coarse_idx`` and, on request, ``coarse_scores (B,N)``, ``fine_scores (B,N2)``,
    ``feat_tgt (B,32,64)``.  ``out``: a dict from an earlier call whose tensors are written again (static buffers for
    graph capture).  With the hypothesis sets sharded over ranks use ``refine.CoarseToFine`` (five launches, two
    all-reduces).  Inference only: refuses inputs that require grad while autograd is recording."""
    _refuse_grad("coarse_to_fine", vol_src, vol_tgt, W1, W2, b2)
    with torch.no_grad():
        return _coarse_to_fine_nograd(vol_src, vol_tgt, R, D, W1, W2, b2, state, want_scores, want_feat_tgt, no_teams,
                                      spare_cus, out)


def _coarse_to_fine_nograd(vol_src, vol_tgt, R, D, W1, W2, b2, state, want_scores, want_feat_tgt, no_teams, spare_cus, out):
    if vol_src.dim() != 5 or tuple(vol_src.shape[1:]) != _VOL or tuple(vol_tgt.shape) != tuple(vol_src.shape):
        raise RuntimeError("vol_src and vol_tgt must be (B,16,8,8,8), got %s and %s" % (tuple(vol_src.shape), tuple(vol_tgt.shape)))
    B = vol_src.shape[0]
    dev = _need_gpu(vol_src, vol_tgt, R, D, W1, W2, b2)
    N, rstride = _rot_layout(R, B)
    if D.dim() != 3 or tuple(D.shape[1:]) != (3, 3):
        raise RuntimeError("D must be (N2,3,3)")
    N2 = D.shape[0]
    W1, W2, b2 = _head(W1, W2, b2)
    if state is None:
        state = CoarseToFineState(B, dev)
    elif state.B != B or state.keys.device != dev:
        raise RuntimeError("the CoarseToFineState was made for B = %d on %s" % (state.B, state.keys.device))
    vs, vt, Rc, Dc = (t.detach().contiguous() for t in (vol_src, vol_tgt, R, D))
    o = out if out is not None else {}
    def buf(name, shape, dtype=torch.float32, wanted=True):
        if not wanted:
            return None
        if name not in o:
            o[name] = torch.empty(shape, dtype=dtype, device=dev)
        return o[name]
    sc = buf("coarse_scores", (B, N), wanted=want_scores)
    sf = buf("fine_scores", (B, N2), wanted=want_scores)
    ft = buf("feat_tgt", (B, 32, 64), wanted=want_feat_tgt)
    rp, fs, cs = buf("R_pred", (B, 3, 3)), buf("fine_score", (B,)), buf("coarse_score", (B,))
    fi, ci = buf("fine_idx", (B,), torch.int64), buf("coarse_idx", (B,), torch.int64)
    p = lambda t: 0 if t is None else t.data_ptr()
    if not 0 <= spare_cus <= 255:
        raise RuntimeError("spare_cus must be in [0, 255]")
    flags = (_lib.AHV_SCORE_NO_TEAMS if no_teams else 0) | (spare_cus << _lib.AHV_SCORE_SPARE_CUS_SHIFT)
    _call(dev, "ahv_coarse_to_fine_f32", vs.data_ptr(), vt.data_ptr(), Rc.data_ptr(), rstride, N, Dc.data_ptr(), N2,
          W1.data_ptr(), W2.data_ptr(), b2.data_ptr(), B, p(sc), p(sf), state.keys.data_ptr(), state.sync.data_ptr(), p(ft),
          rp.data_ptr(), fs.data_ptr(), fi.data_ptr(), cs.data_ptr(), ci.data_ptr(), flags)
    o["state"] = state
    return o


@torch.no_grad()
def so3_grid(n_total: int, device, offset: int = 0, n: int | None = None) -> torch.Tensor:
    """Rows ``[offset, offset + n)`` of the deterministic ``n_total``-point super-Fibonacci SO(3) grid, generated on
    the device (``rotations.so3_grid_np`` is the host form of the same grid)."""
    n = n_total - offset if n is None else n
    out = torch.empty((n, 3, 3), dtype=torch.float32, device=device)
    if out.device.type != "cuda":
        raise RuntimeError("3dahv_amd ops run on the GPU only (no CPU fallback); got device %s" % out.device)
    _call(out.device, "ahv_so3_grid_f32", n_total, offset, n, out.data_ptr())
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
    _call(out.device, "ahv_random_rotations_f32", seed & (2**64 - 1), offset, n, out.data_ptr())
    return out
