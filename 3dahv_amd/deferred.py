"""Deferred hypothesis tensors: the UNCHANGED score lines of the reference's evaluation loop as ONE fused launch.

test_co3d.py:137-146 (the same lines are in test_linemod.py:49-63 and modules/model.py:186-196) spell the hot loop as separate torch calls:

    rot  = [rotate_volume(v[None].expand(N, -1, -1, -1, -1), proposals) for v in img_feat_src]      # (N,16,8,8,8): 1.6 GB at N = 50 000
    rot  = torch.stack(rot).reshape(-1, C, D, H, W)
    f    = model.feature_aligner.forward_3d2d(rot).reshape(B, N, -1, H * W)                         # (B,N,32,64): 0.4 GB
    sim  = (f * img_feat_tgt[:, None]).sum(dim=2).mean(dim=-1)                                      # (B,N)
    pred_sim, pred_index = torch.max(sim, dim=1)

With ``patch.install()`` each of these can run as its own HIP kernel (INTEGRATION.md option A as of round 5: the tensors
above are materialised exactly as the reference materialises them, 2.4 ms and 3.5 GB per pair).  This module removes the
materialisation without touching the script: the patched ``rotate_volume`` of an inference call returns a
``DeferredHypotheses`` -- a ``torch.Tensor`` subclass that has the shape, dtype and device of the rotated volumes but no
storage, only (volume, rotations) -- and the subclass's ``__torch_function__`` recognises exactly the chain above:

    rotated --stack / reshape--> rotated --forward_3d2d (patched)--> features --reshape--> features
            --mul by (B,1,32,64)--> product --sum(dim=2)--> channel sum --mean(dim=-1)--> ONE ``ops.score_hypotheses`` launch

and returns the ordinary ``(B, N)`` score tensor the script's ``torch.max`` consumes.  ANY other use of a deferred tensor (another
operator, another argument, printing a value, ``.cpu()``, indexing, an in-place write ...) materialises it first with the
op-level kernels -- the same tensors option A produced before -- and carries on with plain torch: the deferral can cost time,
never a result.

Training (``infoNCE_loss``, modules/model_co3d.py:49-54) spells the same chain per sample, with operands that need gradients:

    warp = [rotate_volume(img_feat_1[i:i+1].expand(N, ...), sampled_R[i]) for i in range(bs)]
    warp = [forward_3d2d(w) for w in warp]                       # patched; module in training mode: parameters stay attached
    sim  = [(warp[i] * img_feat_2[i:i+1]).sum(dim=1).mean(dim=-1) for i in range(bs)]

Here ``mean`` returns one more deferred tensor per sample (kind "scores", shape (N,)); the first use of any of them evaluates
ALL pending ones of the batch -- same head parameters, same N -- in ONE launch of the backend's scorer with per-sample
rotation sets, under the caller's grad mode: with the HIP backend that is ``ops.score_hypotheses_autograd``, the training pair
of DESIGN 4.4, and ``loss.backward()`` runs its HIP backward once for the batch.  Fallbacks of a chain that needs gradients run
the differentiable op-level kernels.  An inference call of ``forward_3d2d`` (``with_head(detach=True)``) cuts every edge.

The arithmetic lives behind a three-function backend (``rotate_volume``, ``forward_3d2d``, ``score_hypotheses``): the HIP
``ops`` in production (there is no CPU fallback: ``ops`` raises without the library), a stock-torch backend in the CPU tests
of the bookkeeping (``tests/test_deferred_cpu.py``).
"""
from __future__ import annotations

import contextlib
import math
import weakref
from dataclasses import dataclass, replace
from typing import Callable, Optional, Tuple

import torch
from torch.utils._pytree import tree_map

_VOL = (16, 8, 8, 8)
_FEAT = (32, 64)

counters = {"deferred_rotations": 0, "deferred_forward_3d2d": 0, "fused_score_launches": 0, "materialised": 0}
_pending: list = []   # weak references to per-sample score tensors that have not been evaluated yet (see _evaluate_pending)


@dataclass(frozen=True)
class Backend:
    rotate_volume: Callable       # (volume (N,16,8,8,8) -- may be a stride-0 expand --, R (N,3,3)) -> (N,16,8,8,8)
    forward_3d2d: Callable        # (vol (M,16,8,8,8), W1, W2, b2) -> (M,32,64)
    score_hypotheses: Callable    # (vol_src (B,16,8,8,8), feat_tgt (B,32,64), R (N,3,3) or (B,N,3,3), W1, W2, b2) -> scores (B,N)
    device_type: str = "cuda"


def _hip_backend() -> Backend:
    from . import ops

    def score(v, t, R, W1, W2, b2):
        if torch.is_grad_enabled() and any(x.requires_grad for x in (v, t, W1, W2, b2)):
            return ops.score_hypotheses_autograd(v, t, R, W1, W2, b2)     # the training pair: HIP forward + backward
        return ops.score_hypotheses(v, t, R, W1, W2, b2)[0]
    return Backend(rotate_volume=ops.rotate_volume, forward_3d2d=ops.forward_3d2d, score_hypotheses=score)


_backend: Optional[Backend] = None


def backend() -> Backend:
    global _backend
    if _backend is None:
        _backend = _hip_backend()
    return _backend


def set_backend(b: Optional[Backend]) -> Optional[Backend]:
    """Install another backend (tests); ``None`` restores the HIP one.  Returns the previous backend."""
    global _backend
    prev, _backend = _backend, b
    return prev


@dataclass(frozen=True)
class _State:
    kind: str                       # "rotated" | "features" | "product" | "channel_sum" | "scores"
    vols: torch.Tensor              # (Bv,16,8,8,8) source volumes, real
    R: torch.Tensor                 # (N,3,3), real, shared by the Bv volumes
    head: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None    # forward_3d2d's weights (features onwards)
    tgt: Optional[torch.Tensor] = None                                       # (Bv,32,64) target features (product onwards)
    backend: Optional[Backend] = None
    R_key: tuple = ()               # identity of the caller's rotation tensor (storage, layout, version): same key = same set
    grad: bool = False              # an operand needs a gradient: the launches run under the caller's grad mode (autograd edges)
    inference: bool = False         # forward_3d2d was an inference call: later operands are detached, nothing records a graph


def _trailing(kind: str) -> Tuple[int, ...]:
    return {"rotated": _VOL, "features": _FEAT, "product": _FEAT, "channel_sum": (_FEAT[1],), "scores": ()}[kind]


# attribute getters and methods that only look at a tensor's metadata: answered by the wrapper itself
_META_GETTERS = {"shape", "dtype", "device", "ndim", "is_cuda", "requires_grad", "layout", "grad", "grad_fn", "is_leaf",
                 "names", "is_sparse", "is_quantized", "is_meta", "is_cpu", "is_nested", "itemsize", "nbytes", "_version",
                 "output_nr", "is_mkldnn", "is_xpu", "is_mps", "is_xla", "is_sparse_csr", "retains_grad"}
_META_METHODS = {"dim", "size", "numel", "nelement", "ndimension", "__len__", "is_floating_point", "is_complex",
                 "element_size", "get_device", "is_contiguous", "is_signed", "is_inference", "is_shared", "is_pinned",
                 "has_names", "is_conj", "is_neg", "_is_view", "is_same_size"}


class DeferredHypotheses(torch.Tensor):
    """See the module docstring.  Construct through ``defer_rotate_volume`` / ``with_head``."""

    @staticmethod
    def __new__(cls, shape, state: _State):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=state.vols.device,
                                                requires_grad=False)
        t._ahv = state
        t._real = None
        return t

    # ---- bookkeeping ------------------------------------------------------------------------------------
    @property
    def deferred_kind(self) -> Optional[str]:
        """The stage of the recognised chain this tensor stands for; None once it has been materialised."""
        return None if self._real is not None else self._ahv.kind

    def _meta_shape(self) -> Tuple[int, ...]:
        with torch._C.DisableTorchFunctionSubclass():
            return tuple(self.shape)

    def _counts(self) -> Tuple[int, int]:
        return self._ahv.vols.shape[0], self._ahv.R.shape[0]

    def _valid_lead(self, lead: Tuple[int, ...]) -> bool:
        Bv, N = self._counts()
        return lead == (Bv * N,) or lead == (Bv, N) or (Bv == 1 and lead == (N,))

    def with_head(self, W1, W2, b2, detach: bool = False) -> "DeferredHypotheses":
        """``forward_3d2d`` of deferred rotated volumes: (M,16,8,8,8) -> (M,32,64), still deferred.  ``detach``: an inference
        call (patch._inference_call) -- nothing downstream records a graph, whatever the operands' flags say."""
        assert self.deferred_kind == "rotated"
        shape = self._meta_shape()
        if len(shape) != 5:
            raise RuntimeError("forward_3d2d expects (M,16,8,8,8), got %s" % (shape,))
        counters["deferred_forward_3d2d"] += 1
        st = self._ahv
        if detach:
            st = replace(st, vols=st.vols.detach(), grad=False, inference=True)
            W1, W2, b2 = W1.detach(), W2.detach(), b2.detach()
        else:
            st = replace(st, grad=st.grad or (torch.is_grad_enabled() and any(w.requires_grad for w in (W1, W2, b2))))
        return DeferredHypotheses(shape[:1] + _FEAT, replace(st, kind="features", head=(W1, W2, b2)))

    def _mode(self):
        return contextlib.nullcontext() if self._ahv.grad else torch.no_grad()

    def materialise(self) -> torch.Tensor:
        """The real tensor this object stands for, by the op-level kernels (cached: later uses see the same storage)."""
        if self._real is not None:
            return self._real
        st, be = self._ahv, self._ahv.backend or backend()
        if st.kind == "scores":
            _evaluate_pending(self)
            return self._real
        Bv, N = self._counts()
        counters["materialised"] += 1
        with self._mode():
            rot = torch.stack([be.rotate_volume(v[None].expand(N, -1, -1, -1, -1), st.R) for v in st.vols])   # (Bv,N,16,8,8,8)
            if st.kind == "rotated":
                out = rot
            else:
                out = be.forward_3d2d(rot.reshape(-1, *_VOL), *st.head).reshape(Bv, N, *_FEAT)
                del rot
                if st.kind in ("product", "channel_sum"):
                    out = out * st.tgt[:, None]
                if st.kind == "channel_sum":
                    out = out.sum(dim=2)
            self._real = out.reshape(self._meta_shape())
        return self._real

    def _scores(self) -> torch.Tensor:
        st, be = self._ahv, self._ahv.backend or backend()
        counters["fused_score_launches"] += 1
        with self._mode():
            s = be.score_hypotheses(st.vols.contiguous(), st.tgt.contiguous(), st.R, *st.head)     # (Bv,N)
        return s.reshape(self._meta_shape()[:-1])

    # ---- the recognised chain ---------------------------------------------------------------------------
    def _lazy_reshape(self, shape) -> Optional["DeferredHypotheses"]:
        if self._ahv.kind not in ("rotated", "features"):
            return None
        shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else list(shape)
        if not all(isinstance(s, int) for s in shape):
            return None
        numel = math.prod(self._meta_shape())
        if shape.count(-1) == 1:
            known = -math.prod(shape)
            if known <= 0 or numel % known:
                return None
            shape[shape.index(-1)] = numel // known
        if any(s < 0 for s in shape) or math.prod(shape) != numel:
            return None
        tr = _trailing(self._ahv.kind)
        lead, tail = tuple(shape[:len(shape) - len(tr)]), tuple(shape[len(shape) - len(tr):])
        if tail != tr or not self._valid_lead(lead):
            return None
        return DeferredHypotheses(shape, self._ahv)

    def _lazy_mul(self, other) -> Optional["DeferredHypotheses"]:
        """features * target features: (Bv,N,32,64) * (Bv,1,32,64) (test_co3d.py:143) or, for one volume, (N,32,64) * (1,32,64)
        (the per-sample form of infoNCE_loss, modules/model_co3d.py:54)."""
        Bv, N = self._counts()
        shape = self._meta_shape()
        if (self._ahv.kind != "features" or not isinstance(other, torch.Tensor) or isinstance(other, DeferredHypotheses)
                or other.dtype != torch.float32 or other.device != self._ahv.vols.device):
            return None
        if shape == (Bv, N) + _FEAT and tuple(other.shape) == (Bv, 1) + _FEAT:
            tgt = other[:, 0]
        elif Bv == 1 and shape == (N,) + _FEAT and tuple(other.shape) == (1,) + _FEAT:
            tgt = other
        else:
            return None
        grad = (not self._ahv.inference) and (self._ahv.grad or (torch.is_grad_enabled() and other.requires_grad))
        return DeferredHypotheses(shape, replace(self._ahv, kind="product", tgt=tgt if grad else tgt.detach(), grad=grad))

    @staticmethod
    def _reduction_dim(args, kwargs, ndim):
        """The single ``dim`` of a ``sum`` / ``mean`` call with no other option set, normalised; else None."""
        rest = list(args)
        dim = kwargs.get("dim", kwargs.get("axis", rest.pop(0) if rest else None))
        if rest or set(kwargs) - {"dim", "axis", "keepdim"} or kwargs.get("keepdim", False):
            return None
        if isinstance(dim, (tuple, list)) and len(dim) == 1:
            dim = dim[0]
        if not isinstance(dim, int):
            return None
        return dim % ndim

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        owner = getattr(getattr(func, "__self__", None), "__name__", None)   # attribute getters: <getset 'shape'>.__get__
        if (name == "__get__" and owner in _META_GETTERS) or name in _META_METHODS:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        out = cls._try_lazy(name, args, kwargs)
        if out is not None:
            return out
        real = lambda x: x.materialise() if isinstance(x, DeferredHypotheses) else x
        args, kwargs = tree_map(real, (tuple(args), dict(kwargs)))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # the safety net below __torch_function__ (an operator reached from C++ without passing it): the same fallback
        real = lambda x: x.materialise() if isinstance(x, DeferredHypotheses) else x
        args, kwargs = tree_map(real, (tuple(args), dict(kwargs or {})))
        return func(*args, **kwargs)

    @classmethod
    def _try_lazy(cls, name, args, kwargs):
        if name == "stack":
            seq = args[0] if args else kwargs.get("tensors")
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if (isinstance(seq, (list, tuple)) and seq and dim == 0 and len(args) <= 2 and not (set(kwargs) - {"tensors", "dim"})
                    and all(isinstance(t, DeferredHypotheses) and t.deferred_kind == "rotated" and t._counts()[0] == 1
                            and len(t._meta_shape()) == 5 for t in seq)):
                f = seq[0]._ahv
                if all(t._ahv.R_key == f.R_key and t._ahv.backend is f.backend for t in seq):
                    vols = f.vols if len(seq) == 1 else torch.cat([t._ahv.vols for t in seq])
                    return DeferredHypotheses((len(seq), f.R.shape[0]) + _VOL, replace(f, vols=vols))
            return None
        self = args[0] if args else None
        if not isinstance(self, DeferredHypotheses) or self._real is not None:
            other = args[1] if len(args) > 1 else None
            if name in ("mul", "__mul__", "__rmul__", "multiply") and isinstance(other, DeferredHypotheses) and other._real is None \
                    and len(args) == 2 and not kwargs:
                return other._lazy_mul(self)          # real * deferred
            return None
        if name in ("reshape", "view") and not kwargs:
            return self._lazy_reshape(args[1:])
        if name in ("mul", "__mul__", "__rmul__", "multiply") and len(args) == 2 and not kwargs:
            return self._lazy_mul(args[1])
        if name == "sum" and self._ahv.kind == "product":
            shape = self._meta_shape()
            if cls._reduction_dim(args[1:], kwargs, len(shape)) == len(shape) - 2:       # the channel axis
                return DeferredHypotheses(shape[:-2] + shape[-1:], replace(self._ahv, kind="channel_sum"))
            return None
        if name == "mean" and self._ahv.kind == "channel_sum":
            shape = self._meta_shape()
            if cls._reduction_dim(args[1:], kwargs, len(shape)) == len(shape) - 1:       # the position axis
                if len(shape) == 2:
                    # the per-sample form (infoNCE_loss builds `sim` as a list over the batch, modules/model_co3d.py:54): the
                    # scores stay deferred until the first of them is used, and ALL samples' scores are then one launch
                    d = DeferredHypotheses(shape[:1], replace(self._ahv, kind="scores"))
                    if len(_pending) >= 256:   # (references to tensors that were dropped unused)
                        _pending[:] = [r for r in _pending if r() is not None]
                    _pending.append(weakref.ref(d))
                    return d
                return self._scores()
            return None
        return None

    def __repr__(self):   # (never materialise for a debugger's sake)
        if self._real is not None:
            return "DeferredHypotheses(materialised, shape=%s)" % (self._meta_shape(),)
        Bv, N = self._counts()
        return "DeferredHypotheses(%s, shape=%s, volumes=%d, rotations=%d)" % (self._ahv.kind, self._meta_shape(), Bv, N)


def _evaluate_pending(first: "DeferredHypotheses") -> None:
    """Evaluate ``first`` and every other pending per-sample score tensor of the same batch -- same head weights (the module's
    parameters: identity), same number of rotations, same backend and grad mode -- in ONE fused launch with per-sample rotation
    sets (B,N,3,3); each tensor then stands for its row.  A tensor evaluated alone is the B = 1 case of the same call."""
    st = first._ahv
    same = lambda o: (o._real is None and o._ahv.kind == "scores" and o._ahv.backend is st.backend and o._ahv.grad == st.grad
                      and all(a is b for a, b in zip(o._ahv.head, st.head)) and o._ahv.R.shape == st.R.shape
                      and o._ahv.vols.device == st.vols.device)
    live = [r() for r in _pending]
    group = [o for o in live if o is not None and o is not first and same(o)]
    group.insert(0, first)
    _pending[:] = [weakref.ref(o) for o in live if o is not None and not any(o is g for g in group)]
    be = st.backend or backend()
    counters["fused_score_launches"] += 1
    with first._mode():
        if len(group) == 1:
            s = be.score_hypotheses(st.vols.contiguous(), st.tgt.contiguous(), st.R, *st.head)
        else:
            s = be.score_hypotheses(torch.cat([o._ahv.vols for o in group]), torch.cat([o._ahv.tgt for o in group]),
                                    torch.stack([o._ahv.R for o in group]), *st.head)
        for i, o in enumerate(group):
            o._real = s[i]


def defer_rotate_volume(volume: torch.Tensor, R: torch.Tensor, be: Optional[Backend] = None,
                        allow_grad: bool = False) -> Optional[DeferredHypotheses]:
    """A deferred ``rotate_volume(volume, R)`` if the call has the shape of the reference's hot loops -- ONE (16,8,8,8) volume
    expanded over N rotations, fp32, on the backend's device -- else None (the caller then runs the kernel).  A volume that
    needs a gradient is deferred only with ``allow_grad`` (the chain then ends in the differentiable fused scorer, and every
    fallback runs the differentiable op-level kernels under the caller's grad mode); rotations that need one never are.
    The deferred tensor holds a VIEW of the volume: like the lazily evaluated expression it stands for, it sees writes to the
    volume that happen before the scores are taken (the reference's loops have none)."""
    be = be or backend()
    if (not isinstance(volume, torch.Tensor) or isinstance(volume, DeferredHypotheses) or volume.dim() != 5
            or tuple(volume.shape[1:]) != _VOL or volume.dtype != torch.float32 or volume.device.type != be.device_type
            or R.dim() != 3 or tuple(R.shape) != (volume.shape[0], 3, 3) or R.dtype != torch.float32 or R.device != volume.device
            or (volume.shape[0] > 1 and volume.stride(0) != 0)):
        return None
    grad = torch.is_grad_enabled() and volume.requires_grad
    if (torch.is_grad_enabled() and R.requires_grad) or (grad and not allow_grad):
        return None
    counters["deferred_rotations"] += 1
    st = _State(kind="rotated", vols=(volume if grad else volume.detach())[:1], R=R.detach().contiguous(), backend=be,
                R_key=(R.data_ptr(), tuple(R.shape), tuple(R.stride()), R._version), grad=grad)
    return DeferredHypotheses(tuple(volume.shape), st)
