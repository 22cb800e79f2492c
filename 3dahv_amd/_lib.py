"""ctypes binding of libahv_hip.so (the C ABI in include/ahv.h).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.  ``build()`` compiles it with hipcc for gfx950 (works
without a GPU: hipcc cross-compiles).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libahv_hip.so")

AHV_SCORE_RESET_BEST = 1
AHV_SCORE_SPLIT_F16 = 2
AHV_SCORE_NO_TEAMS = 4
AHV_SCORE_SPARE_CUS_SHIFT = 8
AHV_SELECT_RESET_KEY = 1
AHV_KEY_EMPTY = -(1 << 63)

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_u32 = ctypes.c_uint

# name -> (restype, argtypes); mirrors include/ahv.h declaration by declaration
SIGNATURES = {
    "ahv_abi_version": (_int, []),
    "ahv_last_error": (ctypes.c_char_p, []),
    "ahv_device_cu_count": (_int, []),
    "ahv_score_hypotheses_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _int, _i64, _vp, _vp, _u32, _vp]),
    "ahv_score_hypotheses_clocked_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _int, _i64, _vp, _vp, _u32,
                                                _vp, _vp]),
    "ahv_verify_pair_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp, _u32, _vp, _vp]),
    "ahv_unpack_best": (_int, [_vp, _int, _vp, _vp, _vp]),
    "ahv_reset_best": (_int, [_vp, _int, _vp]),
    "ahv_rotate_volume_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _int, _vp, _vp]),
    "ahv_rotate_volume_backward_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _int, _vp, _vp]),
    "ahv_forward_3d2d_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "ahv_score_features_f32": (_int, [_vp, _vp, _int, _i64, _vp, _vp]),
    "ahv_argmax_f32": (_int, [_vp, _int, _i64, _i64, _vp, _u32, _vp]),
    "ahv_random_rotations_f32": (_int, [ctypes.c_uint64, ctypes.c_uint64, _i64, _vp, _vp]),
    "ahv_so3_grid_f32": (_int, [_i64, _i64, _i64, _vp, _vp]),
    "ahv_select_rotation_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _u32, _vp]),
    "ahv_compose_rotations_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _int, _vp, _vp]),
    "ahv_coarse_to_fine_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
}

class BlockWeights(ctypes.Structure):
    """Mirror of ``ahv_block_weights`` (include/ahv.h)."""
    _fields_ = [(n, _vp) for n in ("w_qkv", "w_out", "b_out", "ln1_g", "ln1_b", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
                                   "ln2_g", "ln2_b")]


class AlignerWeights(ctypes.Structure):
    """Mirror of ``ahv_aligner_weights`` (include/ahv.h)."""
    _fields_ = ([(n, _vp) for n in ("w_emb", "w_conv1", "w_conv2", "posemb", "gn_g", "gn_b")] +
                [("w_in", _vp * 2), ("b_in", _vp * 2), ("w_out", _vp * 2), ("b_out", _vp * 2),
                 ("w3d_1", _vp), ("w3d_2", _vp), ("blocks", ctypes.POINTER(BlockWeights)), ("depth", _int)])


SIGNATURES["ahv_forward_2d3d_workspace_bytes"] = (ctypes.c_size_t, [_int])
SIGNATURES["ahv_forward_2d3d_f32"] = (_int, [ctypes.POINTER(AlignerWeights), _vp, _vp, _int, _vp, ctypes.c_size_t, _vp,
                                             _vp, _vp])
SIGNATURES["ahv_score_hypotheses_backward_workspace_bytes"] = (ctypes.c_size_t, [_int, _i64])
SIGNATURES["ahv_score_hypotheses_backward_f32"] = (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, _i64, _vp, _vp,
                                                          ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["ahv_score_hypotheses_backward_saved_f32"] = SIGNATURES["ahv_score_hypotheses_backward_f32"]
SIGNATURES["ahv_score_hypotheses_train_f32"] = (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, _i64, _vp, _vp, ctypes.c_size_t, _vp])
SIGNATURES["ahv_transformer_workspace_bytes"] = (ctypes.c_size_t, [_int])
SIGNATURES["ahv_transformer_blocks_f32"] = (_int, [ctypes.POINTER(BlockWeights), _int, _vp, _vp, _int, _vp,
                                                   ctypes.c_size_t, _vp])

# measurement / developer entry points (include/ahv_diag.h): not part of the drop-in boundary
DIAG_SIGNATURES = {
    "ahv_score_hypotheses_clocked_f32": SIGNATURES.pop("ahv_score_hypotheses_clocked_f32"),
    "ahv_diag_score_plan": (_int, [_int, _i64, _u32, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_i64)]),
}

_lib = None


class AhvError(RuntimeError):
    """A libahv_hip call returned a negative status."""


def build(force: bool = False) -> str:
    """Compile libahv_hip.so in-tree with hipcc --offload-arch=gfx950."""
    args = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    proc = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libahv_hip.so failed:\n" + proc.stdout)
    return LIB_PATH


def load():
    """Load the library and type its entry points; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: import it FIRST so that libahv_hip.so binds to the same
    # libamdhip64 (one runtime per process: shared device context, streams and pointers).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C 3dahv_amd/csrc`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status < 0:
        msg = load().ahv_last_error().decode("utf-8", "replace")
        raise AhvError(f"{what} failed with status {status}: {msg}")
