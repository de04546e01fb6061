#!/bin/bash
# Ground-truth calibration of the counters bench.py's roofline rests on (run on the GPU box via gpurun).
# Every workload is a separate process of tools/microbench/mb with a KNOWN instruction / byte count (`mb calib <what>` prints it); every
# counter group is its own rocprofv3 --pmc run (never mixed with tracing; the program itself directly after `--`).
# Output: gpurun_out/calib/<what>/<group>/..._counter_collection.csv + truth.json; tools/calib_summary.py reduces them to calib.json.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib
mkdir -p $OUT
MB=tools/microbench/mb
for W in fma mix blend max rcp; do
  mkdir -p $OUT/$W
  $MB calib $W > $OUT/$W/truth.json 2> $OUT/$W/truth.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$W/g1 -- $MB calib $W > $OUT/$W/g1.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $OUT/$W/g2 -- $MB calib $W > $OUT/$W/g2.log 2>&1
done
for W in gather gather1 stream; do
  mkdir -p $OUT/$W
  $MB calib $W > $OUT/$W/truth.json 2> $OUT/$W/truth.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$W/g1 -- $MB calib $W > $OUT/$W/g1.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/$W/g2 -- $MB calib $W > $OUT/$W/g2.log 2>&1
  rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/$W/g3 -- $MB calib $W > $OUT/$W/g3.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/$W/g4 -- $MB calib $W > $OUT/$W/g4.log 2>&1
done
python3 tools/calib_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
