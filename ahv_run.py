"""``python ahv_run.py <reference script.py> [args]``: the same as ``python -m 3dahv_amd ...`` from any directory
(3dahv_amd/__main__.py: run an UNCHANGED reference script on the HIP kernels)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":
    sys.exit(importlib.import_module("3dahv_amd.__main__").main())
