#!/bin/bash
# Static vector-instruction count of one BVH4 / BVH8 node visit (tools/bvh_eval/node_step8_isa.hip), compiled with the shipped library's flags; no GPU needed.
cd "$(dirname "$0")/../.."
read -r HIPCC ARCH FLAGS <<< "$(python3 -c "
import sys; sys.path.insert(0, 'tools')
import isa_flags
h, a, f = isa_flags.makefile_flags()
print(h, a, ' '.join(x for x in f if x != '-fPIC'))")"
$HIPCC --offload-arch=$ARCH $FLAGS -Iinclude -S --cuda-device-only -o /tmp/node_step8.s tools/bvh_eval/node_step8_isa.hip 2>/dev/null || exit 1
python3 - <<'PY'
import re, json
from collections import Counter
txt = open("/tmp/node_step8.s").read()
out = {}
for w in (4, 8):
    body = re.search(r"^_Z12visit_kernelILi%d.*?s_endpgm" % w, txt, re.S | re.M).group(0).splitlines()
    depth, ops = 0, []
    for l in body:                       # the loop body (LLVM loop depth 1) IS one visit
        if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb\.\d+:", l):
            d = re.search(r"Depth=(\d+)", l); depth = int(d.group(1)) if d else 0; continue
        t = l.strip()
        if t and not t.startswith((";", ".")) and depth >= 1:
            ops.append(t.split()[0])
    c = Counter(ops)
    vg = re.search(r"\.name:\s+_Z12visit_kernelILi%d.*?\.vgpr_count:\s+(\d+)" % w, txt, re.S)
    out["bvh%d" % w] = {"vector_alu": sum(v for k, v in c.items() if k.startswith("v_")), "scalar": sum(v for k, v in c.items() if k.startswith("s_")),
                        "vector_loads": sum(v for k, v in c.items() if k.startswith("global_load")), "lds": sum(v for k, v in c.items() if k.startswith("ds_")),
                        "v_perm_b32": c["v_perm_b32"], "v_fma_mix_f32": c["v_fma_mix_f32"], "scratch": sum(v for k, v in c.items() if k.startswith("scratch")),
                        "vgprs_of_the_bare_loop": int(vg.group(1)) if vg else None}
print(json.dumps(out))
PY
