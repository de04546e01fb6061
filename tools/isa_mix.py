#!/usr/bin/env python3
"""Static VALU instruction mix of the traversal loops of a bake kernel, by gfx950 issue class (tools/microbench `valu`):
  fast    (2 cycles/wave64: two such instructions issue per quad-cycle): v_fma_f32 v_fmac_f32 v_mul_f32 v_add/sub_f32 v_add/sub_u32
          v_mov_b32 v_and_b32 v_or_b32 v_xor_b32 v_lshrrev_b32
  trans   (8 cycles): v_rcp/rsq/sqrt/exp/log/sin/cos
  complex (4 cycles): everything else (v_fma_mix_f32 v_perm_b32 v_cndmask_b32 v_cmp_* v_min/max* v_cvt_* v_bfe v_lshlrev ...)
Compiles iris_hip.hip to assembly (hipcc -S, device only) and counts the instructions of the basic blocks LLVM annotates with loop
depth >= 3 inside the chosen kernel (tile loop = depth 1, trace_stream's round loop = 2, node / leaf phase loops = 3): the code a ray
spends its time in.  Prints one JSON object; bench.py's VALU roof uses `complex_frac`.

    python tools/isa_mix.py [--kernel bake_view_kernelILi3] [--extra "-DIRIS_..."]
"""
import argparse, json, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_and_b32",
        "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_add_co_u32", "v_sub_co_u32", "v_mac_f32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="bake_view_kernelILi3")
    ap.add_argument("--extra", default="")
    ap.add_argument("--min-depth", type=int, default=3)
    args = ap.parse_args()
    src = os.path.join(REPO, "iris_amd", "csrc", "iris_hip.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        import importlib.util
        spec = importlib.util.spec_from_file_location("isa_flags", os.path.join(REPO, "tools", "isa_flags.py")); fl = importlib.util.module_from_spec(spec); spec.loader.exec_module(fl)
        hipcc, arch, flags = fl.makefile_flags()                # the flags of the SHIPPED library (iris_amd/csrc/Makefile), not a copy of them
        cmd = [hipcc, "--offload-arch=" + arch] + flags + [
               "-I" + os.path.join(REPO, "include"), "-S", "--cuda-device-only", "-o", out, src] + args.extra.split()
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4iris\d+" + re.escape(args.kernel) + r".*:\s*(;.*)?$", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    depth, counts, other = 0, {"fast": 0, "complex": 0, "trans": 0}, {"salu": 0, "vmem": 0, "lds": 0, "lane_ops": 0}
    per_op = {}
    for l in lines[start:end]:
        m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", l)
        if m:
            d = re.search(r"Depth=(\d+)", l)
            depth = int(d.group(1)) if d else 0
            continue
        if re.match(r"^; %bb\.\d+:", l):
            d = re.search(r"Depth=(\d+)", l)
            depth = int(d.group(1)) if d else 0
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or depth < args.min_depth:
            continue
        op = t.split()[0]
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base.startswith("v_"):
            if base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
                other["lane_ops"] += 1          # (SGPR spill traffic and wave-uniform broadcasts: VALU issue slots too, reported separately)
                continue
            if base.startswith("v_mfma"):
                continue
            cls = "fast" if base in FAST else "trans" if base in TRANS else "complex"
            counts[cls] += 1
            per_op[base] = per_op.get(base, 0) + 1
        elif base.startswith("s_"):
            other["salu"] += 1
        elif base.startswith(("global_", "buffer_", "scratch_", "flat_")):
            other["vmem"] += 1
        elif base.startswith("ds_"):
            other["lds"] += 1
    n = sum(counts.values())
    res = {"kernel": args.kernel, "min_loop_depth": args.min_depth, "valu_static": n, **counts,
           "complex_frac": round((counts["complex"] + 2 * counts["trans"]) / max(n, 1), 4), "fast_frac": round(counts["fast"] / max(n, 1), 4), **other,
           "top_ops": dict(sorted(per_op.items(), key=lambda kv: -kv[1])[:14]),
           "note": "complex_frac counts a transcendental as two complex issue slots (8 cycles)"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
