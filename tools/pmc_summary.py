"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name (per dispatch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "bake" not in k and "trace" not in k and "intersect" not in k:
                continue
            k = k.split("(")[0][-60:]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("==", k)
    for c, v in sorted(d.items()):
        print("  %-40s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
