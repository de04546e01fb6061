"""Regression guard for the hidden variable of the timed kernel (EXPERIMENTS.md round 4): where hipcc puts its spills.
Twice in round 4 a change elsewhere moved scratch reloads into the traversal -- once in front of every push and pop, once (a 64-bit lane mask of the
claim index, 2 % of the kernel) into every refill round.  This test compiles the kernel to assembly (cross-compile: no GPU needed) and checks, by LLVM
loop depth, that the loops a ray spends its time in touch no scratch: tile loop = 1, sampling / refill / shading loops = 2, node and leaf loops = 3."""
import os
import re
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_view_kernel_traversal_loops_touch_no_scratch():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_spills.py"), "--kernel", "bake_view_kernelILi3"], check=True, capture_output=True, text=True, timeout=600).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"depth (\d+) instructions (\d+) (\{.*\})", line)
        if m:
            rows[int(m.group(1))] = (int(m.group(2)), eval(m.group(3)))          # (the tool prints a plain dict of counts)
    assert {1, 2, 3} <= set(rows), out
    for depth in (2, 3):
        n, c = rows[depth]
        assert n > 500, f"depth {depth}: {n} instructions -- the loop structure the guard keys on has changed:\n{out}"
        assert c.get("scratch_load", 0) == 0 and c.get("scratch_store", 0) == 0, f"scratch traffic inside the depth-{depth} loops:\n{out}"
    # lane moves (scalar registers spilled to vector lanes) inside the node / leaf loops: only the rare paths (stack overflow) may carry them
    assert rows[3][1].get("v_readlane", 0) <= 48, out
