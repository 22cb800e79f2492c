"""Run an UNCHANGED reference script on the HIP kernels (INTEGRATION.md option A with zero edited lines):

    cd /path/to/3DAHV && python -m 3dahv_amd test_co3d.py [the script's own arguments]
    (or: python /path/to/this/repo/ahv_run.py test_co3d.py ...)

What it does, in this order: puts the script's directory at the head of ``sys.path`` (as ``python script.py`` would), imports
the reference's ``utils`` and ``modules.modules`` from there, rebinds the hot path's callables (``patch.install()``:
``utils.rotate_volume``, ``Feature_Aligner.forward_3d2d`` / ``forward_2d3d``; the score lines then run as one fused launch,
``deferred.py``), and hands control to the script with ``__name__ == "__main__"`` and ``sys.argv`` as the script expects
them.  ``--ahv-no-defer`` in front of the script name keeps every line its own kernel (``install(defer=False)``).
"""
from __future__ import annotations

import os
import runpy
import sys


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    defer = None
    while argv and argv[0].startswith("--ahv-"):
        flag = argv.pop(0)
        if flag == "--ahv-no-defer":
            defer = False
        else:
            print("unknown option %s (known: --ahv-no-defer)" % flag, file=sys.stderr)
            return 2
    if not argv:
        print(__doc__, file=sys.stderr)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print("no such script: %s" % argv[0], file=sys.stderr)
        return 2
    sys.argv = [argv[0]] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))
    from . import patch
    patch.install(defer=defer)          # imports the reference's utils / modules.modules from the script's directory
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        patch.uninstall()
    return 0


if __name__ == "__main__":
    sys.exit(main())
