"""Run the UNCHANGED reference scripts on the HIP kernels by rebinding the callables on the hot
path (option A):

    utils.rotate_volume                          -> 3dahv_amd.ops.rotate_volume        (utils.py:113-131)
    modules.modules.Feature_Aligner.forward_3d2d -> HIP head kernel                    (modules/modules.py:112-124)
    modules.modules.Feature_Aligner.forward_2d3d -> HIP encoder (inference calls: no_grad OR module.eval())  (modules/modules.py:86-110)

Usage from the reference's checkout, before the script's own imports bind the names:

    import ahv_amd; ahv_amd.patch.install()      # then: from utils import rotate_volume  (now the HIP one)

Calls of the hot loops' shape (one (16,8,8,8) volume expanded over N rotations) are DEFERRED (round 6, ``deferred.py``):
``rotate_volume`` returns a tensor subclass without storage, the patched ``forward_3d2d`` passes it on, and the script's own
``(f * f_tgt[:, None]).sum(dim=2).mean(dim=-1)`` becomes ONE fused ``score_hypotheses`` launch -- the unchanged lines
test_co3d.py:137-146 at the speed and memory of the two-line change of option B; in training mode the per-sample form of
``infoNCE_loss`` (modules/model_co3d.py:49-54) ends in the differentiable fused scorer instead (HIP forward that keeps its
pre-activations + HIP backward).  Any other use of such a tensor materialises it with the op-level kernels, exactly what
``install(defer=False)`` (or ``AHV_PATCH_DEFER=0``) runs for every call.
"""
from __future__ import annotations

import importlib
import os
import sys

from . import deferred, ops

_saved = {}
_defer = True


def _inference_call(module) -> bool:
    """What the patched callables treat as an inference call.  The reference's evaluation scripts never enter
    ``torch.no_grad()``: test_co3d.py:219 only sets ``model.eval()`` (and modules/model_co3d.py:22 switches anomaly
    detection on), so grad mode is ON and every parameter requires grad while they run.  An eval-mode module is
    therefore inference here: the HIP kernels run under ``no_grad`` and the results come back detached, which keeps
    the downstream calls (``rotate_volume`` of a detached volume, the score lines) on their plain paths too.  A module
    in training mode keeps autograd: ``forward_3d2d`` / ``rotate_volume`` through the HIP backward,
    ``forward_2d3d`` through the reference's own implementation."""
    import torch
    return (not torch.is_grad_enabled()) or (not module.training)


calls = {"forward_2d3d_hip": 0, "forward_2d3d_reference": 0, "forward_3d2d_inference": 0, "forward_3d2d_autograd": 0,
         "forward_3d2d_deferred": 0, "rotate_volume_deferred": 0, "rotate_volume_kernel": 0}


def _hip_rotate_volume(volume, rotation_matrix, padding_mode="zeros"):
    """``utils.rotate_volume`` (utils.py:113-131).  The evaluation loop's call -- a stride-0 expand of one detached volume --
    comes back deferred (``deferred.DeferredHypotheses``: same shape, dtype, device; materialised by the kernel below the
    moment anything but the recognised score chain touches it); every other call runs ``ops.rotate_volume`` at once."""
    if _defer and padding_mode == "zeros":
        d = deferred.defer_rotate_volume(volume, rotation_matrix, allow_grad=True)
        if d is not None:
            calls["rotate_volume_deferred"] += 1
            return d
    calls["rotate_volume_kernel"] += 1
    return ops.rotate_volume(volume, rotation_matrix, padding_mode)


def _hip_forward_3d2d(self, img_feat):
    import torch
    c1, c2 = self.feature_embedding_2d[0], self.feature_embedding_2d[2]
    if isinstance(img_feat, deferred.DeferredHypotheses) and img_feat.deferred_kind == "rotated":
        # inference: everything downstream is detached; training: the chain ends in the differentiable fused scorer
        calls["forward_3d2d_deferred"] += 1
        return img_feat.with_head(c1.weight, c2.weight, c2.bias, detach=_inference_call(self))
    if _inference_call(self):
        calls["forward_3d2d_inference"] += 1
        with torch.no_grad():
            return ops.forward_3d2d(img_feat, c1.weight, c2.weight, c2.bias)
    calls["forward_3d2d_autograd"] += 1
    return ops.forward_3d2d(img_feat, c1.weight, c2.weight, c2.bias)


def _hip_forward_2d3d(self, img_feat_src, img_feat_tgt, random_mask=True, mask_ratio=0.25):
    """The reference module's own weights, packed once for the C ABI.  Inference calls (``_inference_call``; the
    scripts pass ``random_mask=False``, modules/model_co3d.py:67) run ``ahv_forward_2d3d_f32`` and return detached
    volumes; training-style calls (module in training mode with autograd on, or ``random_mask=True``) and other
    shapes go to the reference implementation -- ``patch.calls`` counts which one ran."""
    from .aligner import hip_forward_2d3d
    ok = (img_feat_src.is_cuda and _inference_call(self) and random_mask is not True
          and tuple(img_feat_src.shape[1:]) == (768, 8, 8) and getattr(self, "mid_channel", 0) == 256)
    if not ok:
        calls["forward_2d3d_reference"] += 1
        return _saved["forward_2d3d"][1](self, img_feat_src, img_feat_tgt, random_mask, mask_ratio)
    calls["forward_2d3d_hip"] += 1
    return hip_forward_2d3d(self, img_feat_src, img_feat_tgt)   # runs under no_grad; fresh tensors, no graph


def verify_hypotheses(self, img_feat_src, img_feat_tgt, proposals, want_scores=False, **kw):
    """INTEGRATION.md option B as a method of the (reference or mirror) ``Feature_Aligner`` -- ``install()`` adds it to the
    reference's class: lines 137-145 of test_co3d.py as ONE launch (``ops.verify_pair`` with this module's head weights).
    Returns ``(scores | None, packed keys)``; ``ops.select_rotation(keys, proposals)`` decodes them (lines 145-146).
    An inference call by the same rule as the patched callables (``_inference_call``: no_grad, or the module in eval mode --
    the reference scripts never enter no_grad, and ``ops.verify_pair`` on weights that require grad is refused because the
    fused launch has no autograd edge).  A module in training mode with autograd recording is refused loudly here too."""
    import torch
    c1, c2 = self.feature_embedding_2d[0], self.feature_embedding_2d[2]
    if not _inference_call(self):
        raise RuntimeError("verify_hypotheses is the inference step (one fused launch, no autograd edge); the module is in "
                           "training mode with autograd recording -- use rotate_volume / forward_3d2d (differentiable) or "
                           "ops.score_hypotheses_autograd for the loss")
    with torch.no_grad():
        return ops.verify_pair(img_feat_src.detach(), img_feat_tgt.detach(), proposals, c1.weight, c2.weight, c2.bias,
                               want_scores=want_scores, **kw)[:2]


def install(utils_module=None, modules_module=None, defer=None):
    """Patch the reference's modules (already imported, importable from sys.path, or passed in).  ``defer``: run the
    evaluation loop's score lines as one fused launch (module docstring); default: on unless ``AHV_PATCH_DEFER=0``."""
    global _defer
    _defer = (os.environ.get("AHV_PATCH_DEFER", "1") != "0") if defer is None else bool(defer)
    if utils_module is None:
        utils_module = sys.modules.get("utils") or importlib.import_module("utils")
    if modules_module is None:
        modules_module = sys.modules.get("modules.modules") or importlib.import_module("modules.modules")
    if "rotate_volume" not in _saved:
        _saved["rotate_volume"] = (utils_module, utils_module.rotate_volume)
        _saved["forward_3d2d"] = (modules_module.Feature_Aligner, modules_module.Feature_Aligner.forward_3d2d)
        if hasattr(modules_module.Feature_Aligner, "forward_2d3d"):
            _saved["forward_2d3d"] = (modules_module.Feature_Aligner, modules_module.Feature_Aligner.forward_2d3d)
    utils_module.rotate_volume = _hip_rotate_volume
    modules_module.Feature_Aligner.forward_3d2d = _hip_forward_3d2d
    if "forward_2d3d" in _saved:
        modules_module.Feature_Aligner.forward_2d3d = _hip_forward_2d3d
    if not hasattr(modules_module.Feature_Aligner, "verify_hypotheses"):
        modules_module.Feature_Aligner.verify_hypotheses = verify_hypotheses
        _saved["verify_hypotheses"] = (modules_module.Feature_Aligner, None)
    # scripts that did `from utils import *` / `from utils import rotate_volume` earlier hold their own binding
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "rotate_volume", None) is _saved["rotate_volume"][1]:
            setattr(mod, "rotate_volume", _hip_rotate_volume)
    return utils_module, modules_module


def uninstall():
    if not _saved:
        return
    um, f = _saved.pop("rotate_volume")
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "rotate_volume", None) is _hip_rotate_volume:
            setattr(mod, "rotate_volume", f)
    um.rotate_volume = f
    cls, g = _saved.pop("forward_3d2d")
    cls.forward_3d2d = g
    if "forward_2d3d" in _saved:
        cls2, h = _saved.pop("forward_2d3d")
        cls2.forward_2d3d = h
    if "verify_hypotheses" in _saved:
        cls3, _ = _saved.pop("verify_hypotheses")
        if cls3.__dict__.get("verify_hypotheses") is verify_hypotheses:
            del cls3.verify_hypotheses
