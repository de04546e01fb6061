#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name calls / mean duration, and how much of the wall time between the first and
the last kernel had 1, 2, ... kernels in flight (overlap of the streamed bake's pipeline).  usage: tools/trace_overlap.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev, per = [], collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-48:]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per[n].append(b - a); ev.append((a, 1)); ev.append((b, -1))
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("%-50s calls %4d  mean %9.3f ms  total %9.1f ms" % (n, len(v), sum(v) / len(v) / 1e6, sum(v) / 1e6))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
tot = sum(hist.values())
print("wall %.1f ms; kernels in flight: " % (tot / 1e6) + ", ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
