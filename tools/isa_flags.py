"""The compiler flags of the shipped library, read from iris_amd/csrc/Makefile (used by tools/isa_spills.py, tools/isa_mix.py and tests/test_isa_guard.py:
the assembly they inspect must be the assembly of the build that ships)."""
import os, re, shlex, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def makefile_flags():
    """HIPCC, ARCH and CXXFLAGS exactly as iris_amd/csrc/Makefile expands them (make -pn prints the variable database): the assembly this tool reads is
    compiled with the flags of the SHIPPED library, not with a copy of them that can drift."""
    db = subprocess.run(["make", "-pn", "-C", os.path.join(REPO, "iris_amd", "csrc"), "EXTRA="], capture_output=True, text=True).stdout
    var = {}
    for name in ("HIPCC", "ARCH", "CXXFLAGS"):
        m = re.search(r"^" + name + r"\s*[:?]?=\s*(.*)$", db, re.M)
        var[name] = m.group(1).strip()
    def expand(v):
        for _ in range(8):
            v = re.sub(r"\$\((\w+)\)", lambda m: var.get(m.group(1), ""), v)
        return v
    flags = [f for f in shlex.split(expand(var["CXXFLAGS"])) if f not in ("-shared",) and not f.startswith("-I")]
    return expand(var["HIPCC"]), expand(var["ARCH"]), flags


