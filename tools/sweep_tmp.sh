mkdir -p gpurun_out
for cfg in "" "bvh_tri_cost_x100=50" "bvh_tri_cost_x100=100" "bvh_tri_cost_x100=140" "bvh_max_leaf=3" "bvh_max_leaf=5" "bvh_max_leaf=6" "bvh_max_leaf=3 bvh_tri_cost_x100=100" "bvh_max_leaf=5 bvh_tri_cost_x100=100" "bvh_presplit_x10=40" "bvh_presplit_x10=160"; do
  a=""; for kv in $cfg; do a="$a --debug-set $kv"; done
  echo "== $cfg"
  python3 bench.py --steps 8 --warmup 1 --cpu-seconds 0 --no-extras --no-roofline $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
