"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol include/iris_hip.h declares,
the ctypes prototypes cover exactly that set, and the product never falls back to a CPU path."""
import os
import re
import subprocess

import pytest
import torch

from conftest import REPO


def _header_symbols(name="iris_hip.h"):
    src = open(os.path.join(REPO, "include", name)).read()
    return sorted(set(re.findall(r"IRIS_API[^;(]*?\b(iris_\w+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = _header_symbols()
    for must in ("iris_scene_create", "iris_intersect", "iris_sample_diffuse", "iris_sample_specular", "iris_eval_emitter",
                 "iris_bake_diffuse", "iris_bake_specular", "iris_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from iris_amd import _lib as L
    assert os.path.exists(L.LIB_PATH), "libiris_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
    exported = set(re.findall(r" T (iris_\w+)", out))
    public, debug = set(_header_symbols()), set(_header_symbols("iris_hip_debug.h"))
    assert all(n.startswith("iris_debug_") for n in debug) and not any(n.startswith("iris_debug_") for n in public)
    declared = public | debug                                  # the drop-in boundary + the diagnostics entry points
    assert declared <= exported, declared - exported
    assert exported == declared, exported ^ declared          # nothing undeclared leaks out either
    assert set(L.PROTOTYPES) == declared
    lib = L.lib()                                              # dlopen + prototype binding
    assert lib.iris_version().startswith(b"iris_hip")


def test_public_boundary_has_no_experiment_knobs():
    """include/iris_hip.h is the drop-in boundary: no kernel-variant / instrumentation arguments, and the library reads no
    environment variables (tuning goes through iris_debug_set of the diagnostics header)."""
    src = open(os.path.join(REPO, "include", "iris_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for word in ("variant", "stats", "IRIS_BAKE_"):
        assert word not in code, word
    for f in os.listdir(os.path.join(REPO, "iris_amd", "csrc")):
        assert "getenv" not in open(os.path.join(REPO, "iris_amd", "csrc", f)).read(), f


def test_no_cpu_fallback():
    """CPU tensors are rejected loudly; nothing in the product imports the oracle."""
    from iris_amd import _lib as L
    from iris_amd.model.brdf import BaseBRDF
    with pytest.raises(L.IrisError):
        BaseBRDF().sample_diffuse(torch.rand(4, 2), torch.rand(4, 3))
    for root, _, files in os.walk(os.path.join(REPO, "iris_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert "iris_oracle" not in txt or "oracle/iris_oracle.c" in txt, f


def test_build_id_comes_from_the_loaded_binary_and_covers_every_source():
    """refine_shading --resume keys on L.build_id(): the hash the Makefile embedded in the library, over ALL the files the library is compiled from
    (the path-tracing stages, the material network, the shading cache and the denoiser included -- their arithmetic decides the refined maps), not a
    hash of whatever lies in csrc/ at run time."""
    import hashlib
    from iris_amd import _lib as L
    csrc = os.path.join(REPO, "iris_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    listed = re.search(r"^SOURCES\s*=\s*(.+)$", mk, re.M).group(1).split()
    on_disk = {f for f in os.listdir(csrc) if f.endswith((".h", ".hip", ".cpp"))}
    assert {os.path.basename(f) for f in listed if not f.startswith("..")} == on_disk, "a source file is missing from the Makefile's SOURCES (and from the hash)"
    hip = open(os.path.join(csrc, "iris_hip.hip")).read() + open(os.path.join(csrc, "iris_bake.h")).read() + open(os.path.join(csrc, "iris_pt.h")).read()
    for inc in set(re.findall(r'#include "(\w+\.h)"', hip)):
        assert inc in {os.path.basename(f) for f in listed}, inc
    h = hashlib.sha256()
    for f in listed:
        h.update(open(os.path.join(csrc, f), "rb").read())
    embedded = L.lib().iris_debug_source_hash().decode()
    assert embedded == h.hexdigest()[:16], "libiris_hip.so is older than its sources: rebuild (make -C iris_amd/csrc)"
    assert L.build_id().endswith("|" + embedded) and L.lib().iris_debug_build_flags().decode() in L.build_id()
    src = open(os.path.join(REPO, "iris_amd", "_lib.py")).read()
    body = src[src.index("def build_id"):src.index("class StageTimer")]
    assert "open(" not in body and not re.search(r"(?<!debug_)source_hash\(\)", body)          # nothing read from disk
