#!/bin/bash
# A/B of library builds on the bench workload (run on the GPU box): tools/ab_bench.sh <outdir> <steps> lib1.so lib2.so ...
# AB_ARGS: extra bench.py arguments (e.g. "--debug-set bvh_max_leaf=2"); AB_TAG: suffix of the result names
# Each library runs the same `bench.py --steps N` (timed region only: no roofline pass, no CPU baseline); one JSON line per library.
OUT=$1; STEPS=$2; shift 2
mkdir -p $OUT
for LIB in "$@"; do
  name=$(basename $LIB .so)$AB_TAG
  IRIS_HIP_LIB=$PWD/$LIB timeout 300 python3 bench.py --steps $STEPS --warmup 2 --no-roofline --no-extras --cpu-seconds 0 $AB_ARGS > $OUT/$name.json 2> $OUT/$name.err
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f Mrays/s  %7.2f ms/step  views %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["ms_by_view"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
