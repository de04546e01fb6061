#!/bin/bash
# N = 8, all ranks, 2 views, stripe heights 4 / 8 / 16
for S in 4 8 16; do
  for r in 0 1 2 3 4 5 6 7; do
    IRIS_STRIPE_ROWS=$S timeout 300 python3 bench.py --steps 2 --warmup 1 --no-roofline --no-extras --cpu-seconds 0 --parity-pixels 0 --emulate-world 8 --emulate-rank $r 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=$S r=$r', d['config']['ms_by_view'])"
  done
done
