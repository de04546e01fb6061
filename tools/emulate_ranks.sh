#!/bin/bash
# Per-rank cost of a strong-scaling run of ONE 1080p view sharded over N GPUs, measured on ONE MI355X (run on the GPU box):
#   tools/emulate_ranks.sh <outdir> [steps=4]
# For N in 1 2 4 8 and every rank r of N: `bench.py --emulate-world N --emulate-rank r` bakes only the stripes rank r would own (no collective), on the same
# `steps` views of the circle.  EMULATED / FUNCTIONAL EVIDENCE, NOT A SCALING CURVE: no xGMI transfer, no second GPU's clock or HBM is involved;
# what it measures is how evenly the interleaved 8-row stripes split the work and what tile granularity costs a rank that holds an N-th of a view.
# tools/emulate_ranks_summary.py <outdir> turns the lines into profiles/r5_emulated_ranks.json.
OUT=$1; STEPS=${2:-4}
mkdir -p $OUT
for N in 1 2 4 8; do
  for ((r=0; r<N; r++)); do
    timeout 300 python3 bench.py --steps $STEPS --warmup 1 --no-roofline --no-extras --cpu-seconds 0 --parity-pixels 0 --emulate-world $N --emulate-rank $r > $OUT/w${N}_r${r}.json 2> $OUT/w${N}_r${r}.err || echo "w$N r$r FAILED"
  done
done
python3 tools/emulate_ranks_summary.py $OUT
