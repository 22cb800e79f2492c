"""Run the UNCHANGED reference scripts on the HIP kernels by rebinding the two callables on the hot
path (option A):

    utils.rotate_volume                          -> 3dahv_amd.ops.rotate_volume        (utils.py:113-131)
    modules.modules.Feature_Aligner.forward_3d2d -> HIP head kernel                    (modules/modules.py:112-124)

Usage from the reference's checkout, before the script's own imports bind the names:

    import ahv_amd; ahv_amd.patch.install()      # then: from utils import rotate_volume  (now the HIP one)

The op-level kernels materialise exactly the tensors the reference materialises; the fused
single-launch path needs the three-line change shown in INTEGRATION.md, option B.
"""
from __future__ import annotations

import importlib
import sys

from . import ops

_saved = {}


def _hip_forward_3d2d(self, img_feat):
    c1, c2 = self.feature_embedding_2d[0], self.feature_embedding_2d[2]
    return ops.forward_3d2d(img_feat, c1.weight, c2.weight, c2.bias)


def install(utils_module=None, modules_module=None):
    """Patch the reference's modules (already imported, importable from sys.path, or passed in)."""
    if utils_module is None:
        utils_module = sys.modules.get("utils") or importlib.import_module("utils")
    if modules_module is None:
        modules_module = sys.modules.get("modules.modules") or importlib.import_module("modules.modules")
    if "rotate_volume" not in _saved:
        _saved["rotate_volume"] = (utils_module, utils_module.rotate_volume)
        _saved["forward_3d2d"] = (modules_module.Feature_Aligner, modules_module.Feature_Aligner.forward_3d2d)
    utils_module.rotate_volume = ops.rotate_volume
    modules_module.Feature_Aligner.forward_3d2d = _hip_forward_3d2d
    # scripts that did `from utils import *` / `from utils import rotate_volume` earlier hold their own binding
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "rotate_volume", None) is _saved["rotate_volume"][1]:
            setattr(mod, "rotate_volume", ops.rotate_volume)
    return utils_module, modules_module


def uninstall():
    if not _saved:
        return
    um, f = _saved.pop("rotate_volume")
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "rotate_volume", None) is ops.rotate_volume and mod is not ops:
            setattr(mod, "rotate_volume", f)
    um.rotate_volume = f
    cls, g = _saved.pop("forward_3d2d")
    cls.forward_3d2d = g
