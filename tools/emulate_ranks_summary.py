#!/usr/bin/env python3
"""Summarise tools/emulate_ranks.sh: per world size N the per-rank, per-view kernel-inclusive step times of `bench.py --emulate-world N --emulate-rank r`,
the imbalance max_r / mean_r per view, and t(N = 1) / max_r t_r -- what strong scaling would be BEFORE the gather (<= 323 MB over xGMI, overlapped with the next
view) if the other GPUs behaved as this one.  Emulated on one GPU: functional evidence, not a scaling curve."""
import glob, json, os, re, sys
d = sys.argv[1]
runs = {}
for f in sorted(glob.glob(os.path.join(d, "w*_r*.json"))):
    m = re.search(r"w(\d+)_r(\d+)\.json", f)
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    runs.setdefault(int(m.group(1)), {})[int(m.group(2))] = j
out = {"what": "bench.py --emulate-world N --emulate-rank r for every rank of N = 1, 2, 4, 8 on ONE MI355X: rank r bakes the interleaved 8-row stripes it would own of the same 1920x1080 views "
               "(SPP 128, 7 lobes, 1.0 M-triangle room), no collective.  EMULATED: no second GPU, no xGMI; not a scaling curve.", "worlds": {}}
t1 = None
for N in sorted(runs):
    ranks = runs[N]
    views = [ranks[r]["config"]["ms_by_view"] for r in sorted(ranks)]
    nv = min(len(v) for v in views)
    per_view_max = [max(v[k] for v in views) for k in range(nv)]
    per_view_mean = [sum(v[k] for v in views) / len(views) for k in range(nv)]
    row = {"ranks_measured": len(ranks), "ms_by_rank_and_view": {str(r): ranks[r]["config"]["ms_by_view"] for r in sorted(ranks)},
           "max_over_ranks_ms_by_view": [round(x, 2) for x in per_view_max], "mean_over_ranks_ms_by_view": [round(x, 2) for x in per_view_mean],
           "imbalance_max_over_mean_by_view": [round(a / b, 4) for a, b in zip(per_view_max, per_view_mean)],
           "sum_of_max_ms": round(sum(per_view_max), 2)}
    if N == 1:
        t1 = per_view_max
    if t1:
        row["t1_over_max_rank_by_view"] = [round(a / b, 3) for a, b in zip(t1, per_view_max)]
        row["t1_over_max_rank_all_views"] = round(sum(t1[:nv]) / sum(per_view_max), 3)
        row["work_inflation_sum_over_ranks_vs_t1"] = round(sum(sum(v[:nv]) for v in views) / sum(t1[:nv]), 4)
    out["worlds"][str(N)] = row
json.dump(out, open(os.path.join(d, "emulated_ranks.json"), "w"), indent=1)
for N, row in out["worlds"].items():
    print("N =", N, "max-rank ms by view", row["max_over_ranks_ms_by_view"], "imbalance", row["imbalance_max_over_mean_by_view"], "t1/max", row.get("t1_over_max_rank_all_views"))
