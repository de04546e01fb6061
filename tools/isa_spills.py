#!/usr/bin/env python3
"""Where a kernel's scratch traffic and lane moves sit: counts scratch_load / scratch_store / v_readlane / v_writelane by LLVM loop depth
(tile loop = 1, trace_stream's round loop = 2, node / leaf phase loops = 3).   python tools/isa_spills.py [--kernel bake_view_kernelILi3] [--extra "-D..."]"""
import argparse, collections, os, re, shlex, subprocess, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import importlib.util
_spec = importlib.util.spec_from_file_location("isa_flags", os.path.join(REPO, "tools", "isa_flags.py")); _fl = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_fl)
makefile_flags = _fl.makefile_flags
ap = argparse.ArgumentParser(); ap.add_argument("--kernel", default="bake_view_kernelILi3"); ap.add_argument("--extra", default=""); ap.add_argument("--dump", default="")
a = ap.parse_args()
with tempfile.TemporaryDirectory() as tmp:
    out = a.dump or os.path.join(tmp, "k.s")
    hipcc, arch, flags = makefile_flags()
    subprocess.check_call([hipcc, "--offload-arch=" + arch] + flags + ["-I" + os.path.join(REPO, "include"),
                           "-S", "--cuda-device-only", "-o", out, os.path.join(REPO, "iris_amd", "csrc", "iris_hip.hip")] + a.extra.split(), stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4iris\d+" + re.escape(a.kernel) + r".*:\s*(;.*)?$", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
depth = 0; cnt = collections.defaultdict(collections.Counter); total = collections.Counter()
for l in lines[start:end]:
    if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb\.\d+:", l):
        d = re.search(r"Depth=(\d+)", l); depth = int(d.group(1)) if d else 0; continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]; total[depth] += 1
    for key in ("scratch_load", "scratch_store", "v_readlane", "v_writelane", "v_accvgpr"):
        if op.startswith(key): cnt[depth][key] += 1
for d in sorted(total): print("depth", d, "instructions", total[d], dict(cnt[d]))
