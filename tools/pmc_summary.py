"""Summarise the rocprofv3 output of tools/pmc_profile.sh: mean counter value per kernel name (per dispatch) as text on stdout, and
<dir>/pmc.json = per-launch means of the dominant kernel + its mean duration from the kernel-trace pass + the source hash of the
kernels (iris_amd._lib.source_hash) -- the file bench.py's roofline reads (committed as profiles/pmc_r4.json)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "bake" not in k and "trace" not in k and "intersect" not in k:
                continue
            k = k.split("(")[0][-60:]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            dur[row["Name"].split("(")[0][-60:]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
for k, d in sorted(acc.items()):
    print("==", k, dur.get(k, ""))
    for c, v in sorted(d.items()):
        print("  %-40s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
view = [k for k in acc if "bake_view_kernel" in k]
if view:
    k = view[0]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from iris_amd._lib import source_hash
        sh = source_hash()
    except Exception as e:       # noqa
        sh = None
    out = {"kernel": k, "bench_args": bench_args, "source_hash": sh, "duration": dur.get(k), "counters": {c: sum(v) / len(v) for c, v in sorted(acc[k].items())},
           "dispatches_per_counter": {c: len(v) for c, v in sorted(acc[k].items())}}
    json.dump(out, open(os.path.join(root, "pmc.json"), "w"), indent=1)
