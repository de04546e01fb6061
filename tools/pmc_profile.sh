#!/bin/bash
# rocprofv3 passes for the bake kernels (run on the GPU box via gpurun).  One kernel-trace pass (durations), then one pass per counter
# group -- counters are collected in their own runs, never mixed with tracing.  Per-dispatch means land in gpurun_out/pmc_<tag>/summary.txt
# and, with the source hash of the kernels they were taken on, in gpurun_out/pmc_<tag>/pmc.json (bench.py reads profiles/pmc_r4.json).
# usage: [IRIS_HIP_LIB=...] tools/pmc_profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ARGS=${@:---steps 4 --warmup 1 --cpu-seconds 0 --no-roofline --no-extras}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
i=0
for GROUP in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU2" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCP_TCC_READ_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT "$ARGS" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
