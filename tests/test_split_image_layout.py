"""The split-f16 scorer's quarter image (3dahv_amd/csrc/ahv_split.h: split_addr) against the LDS bank model of
tools/split_image_sim.py: the address expression is read out of the header, so a change of the layout that brings bank
conflicts back (or breaks the one-to-one map) fails here, on the CPU."""
import importlib.util
import os
import re

from .conftest import REPO


def _sim():
    spec = importlib.util.spec_from_file_location("split_image_sim", os.path.join(REPO, "tools", "split_image_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _split_addr():
    src = open(os.path.join(REPO, "3dahv_amd", "csrc", "ahv_split.h")).read()
    body = re.search(r"int split_addr\(int a0, int b, int e, int chunk\)[^{]*\{(.*?)\n\}", src, flags=re.S).group(1)
    x = re.search(r"const int x = ([^;]+);", body).group(1)
    ret = re.search(r"return ([^;]+);", body).group(1)
    lo = int(re.search(r"kSplitLoPlane = (\d+)", src).group(1))
    step = int(re.search(r"kSplitPass = (\d+)", src).group(1))

    def addr(a0, b, e, chunk):
        env = {"a0": a0, "b": b, "e": e, "chunk": chunk}
        env["x"] = eval(x, {}, env)
        return eval(ret, {}, env)
    return addr, lo, step


def test_image_is_a_bijection_with_constant_pass_and_lo_offsets():
    addr, lo, step = _split_addr()
    seen = set()
    for a0 in range(2):
        for b in range(8):
            for e in range(8):
                for c in range(2):
                    a = addr(a0, b, e, c)
                    assert a % 16 == 0 and 0 <= a < lo
                    seen.add(a)
                    if b < 4:
                        assert addr(a0, b + 4, e, c) == a + step   # the gather's second pass rides in the offset field
    assert len(seen) == 2 * 8 * 8 * 2 and 2 * lo == 128 * 64         # hi plane full; lo plane = + lo; 8 KB per wave


def test_shipped_image_has_no_bank_conflicts_in_the_model():
    sim = _sim()
    addr, _, _ = _split_addr()
    assert sim.conflicts_of_address_function(addr) == (0, 0)
    # the model itself: round 3's layout and the unswizzled planes as known answers (512 = what SQ_LDS_BANK_CONFLICT showed)
    assert sim.store_conflicts(sim.E1 | sim.E2, sim.E2 | sim.B0, 0) + sim.read_conflicts(sim.E1 | sim.E2, sim.E2 | sim.B0, 0, 0) == 512
    plain = lambda a0, b, e, c: (a0 + 2 * e + 16 * b) * 32 + 16 * c
    assert sim.conflicts_of_address_function(plain) == (sim.plane_store_conflicts(0, 0, 0), sim.plane_read_conflicts(0, 0, 0, 0))
