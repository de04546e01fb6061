#!/bin/bash
run(){ tag=$1; shift; python3 bench.py --steps 8 --warmup 2 --no-roofline --no-extras --cpu-seconds 0 --parity-pixels 0 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])"; }
run base
run pm8 --debug-set phase_min=8
run pm10 --debug-set phase_min=10
run pm14 --debug-set phase_min=14
run pm16 --debug-set phase_min=16
run tile3584 --debug-set tile_target_rays=3584
run tile4608 --debug-set tile_target_rays=4608
run tile5120 --debug-set tile_target_rays=5120
run tc50 --debug-set bvh_tri_cost_x100=50
run tc100 --debug-set bvh_tri_cost_x100=100
run leaf2 --debug-set bvh_max_leaf=2
run leaf4 --debug-set bvh_max_leaf=4
run base
