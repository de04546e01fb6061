#!/usr/bin/env python3
"""Assemble profiles/pmc_r4.json -- what bench.py's roofline reads -- from
  * gpurun_out/pmc_<tag>/pmc.json   per-launch counter means + kernel duration + source hash (tools/pmc_profile.sh on the GPU box)
  * gpurun_out/pmc_<tag>/trace.log  bench.py's own JSON line of the traced run (rays per launch)
  * profiles/r2_microbench.jsonl    tools/microbench results: TA cycles per wave-load as a function of the distinct lines it touches
  * tools/isa_mix.py                static issue-class mix of the traversal loops of the same sources
    python tools/make_pmc_json.py gpurun_out/pmc_<tag> [out.json]"""
import json, os, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "pmc_r4.json")
pj = json.load(open(os.path.join(src, "pmc.json")))
line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
bj = json.loads(line)
pj["rays_per_launch"] = bj["config"]["rays_per_step"]
pj["workload"] = bj["config"]["workload"]
if bj.get("emulated"):       # rank 0's stripes of an N-rank run, baked on one GPU (bench.py --emulate-world N): what bench.py's roofline reads at N > 1, labelled as such
    pj["emulated_world"] = bj["emulated"]["world"]; pj["emulated_rank"] = bj["emulated"]["rank"]
    pj["workload"] += " -- EMULATED: the stripes of rank %d of %d, on one GPU, no collective" % (bj["emulated"]["rank"], bj["emulated"]["world"])
rows = [json.loads(l) for l in open(os.path.join(REPO, "profiles", "r2_microbench.jsonl"))]


def g(level, lanes_per_record, active=64):
    r = [x for x in rows if x["bench"] == "gather" and x["level"] == level and x["loads_per_record"] == 4 and x["lanes_per_record"] == lanes_per_record
         and x["active_lanes"] == active and not x["dependent"] and x["waves_per_simd"] == 7]
    return r[0]["cycles_per_wave_inst_per_cu"]


def slope(level):      # cycles per extra distinct 64-B line: 64 lines (one lane per record) against 32 lines (two lanes per record)
    return (g(level, 1) - g(level, 2)) / 32.0


l1 = slope("L1")
cal = {"ta_cycles_per_line_l1": round(l1, 3), "ta_cycles_per_line_l2": round(slope("L2"), 3),
       "ta_cycles_per_line_mall": round(0.5 * (slope("MALL16") + slope("MALL80")), 3),
       "ta_cycles_base_per_load": round(g("L1", 1) - 64 * l1, 2), "ta_cycles_min_per_load": round(g("L1", 16), 2),
       "source": "profiles/r2_microbench.jsonl (tools/microbench gather, 7 waves/SIMD, 4 x 16-B loads per 64-B record)"}
mix = json.loads(subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "isa_mix.py")]).decode().strip().splitlines()[-1])
cal["complex_frac"] = mix["complex_frac"]
cal["isa_mix"] = mix
valu = {r["op"]: r["chip_Gwinst_per_s"] for r in rows if r["bench"] == "valu" and r["waves_per_simd"] == 7}
cal["valu_rates_Gwinst_per_s_at_7_waves"] = {k: valu[k] for k in ("v_fma_f32", "v_mul_f32", "v_add_u32", "v_mov_b32", "v_fma_mix_f32", "v_perm_b32", "v_max_f32", "v_cmp_lt_f32",
                                                                  "v_cndmask_b32_e64(sgpr mask)", "v_cndmask_b32", "v_rcp_f32") if k in valu}
pj["calibration"] = cal
json.dump(pj, open(out, "w"), indent=1)
print("wrote", out, "source_hash", pj["source_hash"], "rays_per_launch", pj["rays_per_launch"], "kernel avg ms", pj["duration"]["avg_ns"] / 1e6 if pj.get("duration") else None)
