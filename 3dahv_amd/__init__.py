"""3dahv_amd -- MI355X-native rotation-hypothesis verification for 3DAHV.

The directory name starts with a digit, so import it with
``importlib.import_module("3dahv_amd")`` (or ``import ahv_amd``, the one-line
alias module at the repo root).

Only the hot path of the reference lives here: the HIP kernels + C-ABI library
(``csrc/``), the ctypes binding (``_lib``), the operator mirrors of the
reference call surface (``ops``, ``aligner``, ``estimator``), hypothesis
sharding across GPUs (``dist``) and the evaluation harness counterpart
(``harness``).  Sub-modules are imported lazily so that CPU-only tools (golden
generation, host logic tests) never touch the GPU library.
"""
import importlib as _importlib

__version__ = "0.1.0"
_SUBMODULES = ("rotations", "_lib", "ops", "aligner", "estimator", "dist", "harness", "checkpoint", "patch", "deferred", "refine", "co3d")


def __getattr__(name):
    if name in _SUBMODULES:
        return _importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
