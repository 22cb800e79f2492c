"""Sanitizer tier for the CPU-side C (oracle/ahv_oracle.c: the restatement and the ``*_cpu`` ABI twins, manual malloc
and index arithmetic): ``make -C oracle asan`` builds it with -fsanitize=address,undefined, and the oracle tests, the
ABI-twin tests and the edge-case goldens run against that build in a child process with libasan preloaded.  CPU only --
the GPU build is never sanitised (not available on this pool)."""
import os
import subprocess
import sys

import pytest

from .conftest import REPO


def test_oracle_and_abi_twins_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan on this machine")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, AHV_ORACLE_LIB=os.path.join(REPO, "oracle", "libahv_oracle_asan.so"),
               # leaks: CPython itself never frees everything; everything else aborts the run at the first report
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=67", OMP_NUM_THREADS="4")
    probe = subprocess.run([sys.executable, "-c", "from oracle import oracle; print(oracle.lib()._name)"], cwd=REPO,
                           env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libahv_oracle_asan.so"), probe.stderr[-2000:]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          "tests/test_oracle.py", "tests/test_abi_twin_cpu.py"], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-4000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in out.stdout
