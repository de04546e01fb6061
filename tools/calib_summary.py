#!/usr/bin/env python3
"""Reduce tools/calibrate_counters.sh's output to <dir>/calib.json (committed as profiles/r3_counter_calibration.json): per known
workload, the ground truth printed by `mb calib <what>` next to the counter values of the measured dispatch, and the ratios the
roofline depends on:
  valu : SQ_INSTS_VALU / true wave-instructions;  SQ_ACTIVE_INST_VALU2 / SQ_ACTIVE_INST_VALU (share of issue quad-cycles with two
         instructions: ~1/2 expected for a pure dual-issue class, 0 for a pure 4-cycle class);
         occupied quad-cycles (ACTIVE_INST_VALU - ACTIVE_INST_VALU2) against the kernel's SIMD quad-cycles (1024 x GRBM_GUI_ACTIVE/8 / 4)
  mem  : FETCH_SIZE x 1024 / bytes requested and / bytes of the distinct 64-B (128-B) lines touched"""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d) or not os.path.exists(os.path.join(d, "truth.json")):
        continue
    what = os.path.basename(d)
    truth = [json.loads(l) for l in open(os.path.join(d, "truth.json")) if l.startswith("{")]
    cal = [t for t in truth if t.get("bench") == "calib"]
    if not cal:
        continue
    cal = cal[0]
    # the measured dispatch = the LAST launch of the benchmark's own kernel (hipMemset's fill kernel and the short warm-up launch come first)
    kname = {"valu": "valu_kernel", "stream": "stream_kernel"}.get(cal["what"], "gather_kernel")
    c = {}
    for f in glob.glob(os.path.join(d, "g*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                c[r["Counter_Name"]] = float(r["Counter_Value"])
    e = {"truth": cal, "counters": c, "others": [t for t in truth if t.get("bench") != "calib" and t.get("bench") != "device"]}
    if cal["what"] == "valu":
        n = cal["loop_wave_insts"]
        if c.get("SQ_INSTS_VALU"):
            e["insts_valu_over_truth"] = c["SQ_INSTS_VALU"] / n
        if c.get("SQ_ACTIVE_INST_VALU"):
            a, a2 = c["SQ_ACTIVE_INST_VALU"], c.get("SQ_ACTIVE_INST_VALU2", 0.0)
            e["active_inst_valu_per_inst"] = a / c["SQ_INSTS_VALU"]
            e["valu2_over_active"] = a2 / a
            if c.get("GRBM_GUI_ACTIVE"):
                quad = 1024.0 * (c["GRBM_GUI_ACTIVE"] / 8.0) / 4.0
                e["occupied_quad_cycles_over_simd_quad_cycles"] = (a - a2) / quad
                e["active_inst_valu_over_simd_quad_cycles"] = a / quad
    else:
        if c.get("FETCH_SIZE") is not None:
            b = c["FETCH_SIZE"] * 1024.0
            e["fetch_bytes"] = b
            e["fetch_over_requested"] = b / cal["bytes_requested"]
            if "bytes_of_lines_touched_64B" in cal:
                e["fetch_over_64B_lines"] = b / cal["bytes_of_lines_touched_64B"]
                e["fetch_over_128B_lines"] = b / cal["bytes_of_lines_touched_128B"]
        if c.get("TCC_EA0_RDREQ_sum") is not None:
            e["rdreq_32B_share"] = c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / max(c["TCC_EA0_RDREQ_sum"], 1.0)
            e["rdreq_128B_share(TCC_BUBBLE)"] = c.get("TCC_BUBBLE_sum", 0.0) / max(c["TCC_EA0_RDREQ_sum"], 1.0)
    out[what] = e
json.dump(out, open(os.path.join(root, "calib.json"), "w"), indent=1)
for k, e in out.items():
    print("==", k, json.dumps({x: e[x] for x in e if x not in ("truth", "counters", "others")}))
    print("   truth", json.dumps(e["truth"]))
    print("   counters", json.dumps(e["counters"]))
