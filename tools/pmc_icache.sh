#!/bin/bash
# rocprofv3 counter passes on the instruction / scalar-data caches of the bake kernel (run on the GPU box via gpurun).  Same shape as tools/pmc_profile.sh:
# counters in their own runs, never mixed with tracing; per-dispatch means in gpurun_out/pmc_<tag>/summary.txt.
# usage: [IRIS_HIP_LIB=...] tools/pmc_icache.sh <tag> [bench args...]
set -u
TAG=${1:-icache}; shift || true
ARGS=${@:---steps 4 --warmup 1 --cpu-seconds 0 --no-roofline --no-extras --parity-pixels 0}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for GROUP in \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB GRBM_GUI_ACTIVE" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT "$ARGS" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
