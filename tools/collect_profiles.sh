#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes of the bench workload -> gpurun_out/prof/.
# Usage: tools/collect_profiles.sh [steps] [warmup]      then tools/summarise_profiles.py gpurun_out/prof profiles/rNN
set -e
STEPS=${1:-300}; WARM=${2:-100}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-also --no-host-rate --no-steady"
echo "[prof] kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[prof] pmc $c"; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH > $OUT/pmc_$c.log 2>&1
done
echo "[prof] pmc SQ set 1"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
echo "[prof] pmc SQ set 2"; rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
echo "[prof] pmc SQ set 3"; rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_sq3 -- $BENCH > $OUT/pmc_sq3.log 2>&1 || echo "[prof] set 3 not available"
echo "[prof] pmc SQ set 4"; rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_sq4 -- $BENCH > $OUT/pmc_sq4.log 2>&1 || echo "[prof] set 4 not available"
echo "[prof] pmc SQ set 5 (fp32 instruction mix)"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/pmc_sq5 -- $BENCH > $OUT/pmc_sq5.log 2>&1 || echo "[prof] set 5 not available"
echo "[prof] pmc SQ set 6 (active lanes)"; rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_sq6 -- $BENCH > $OUT/pmc_sq6.log 2>&1 || echo "[prof] set 6 not available"
echo "[prof] pmc GRBM"; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1 || echo "[prof] grbm not available"
cd $ROOT
find $OUT -name "*.csv" | head -40
