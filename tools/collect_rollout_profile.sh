#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes of ONE fused K-step launch (jb_step_many_device, BASELINE configs[2])
# -> gpurun_out/prof_rollout/ ; then  python3 tools/summarise_rollout_profile.py gpurun_out/prof_rollout profiles/r04_rollout
# (counters in their own runs with --kernel-trace only, as the pool requires; the program itself after --)
set -e
K=${1:-1000}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_rollout
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="python3 $ROOT/tools/rollout_profile_target.py $K"
echo "[prof] kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $T > $OUT/stats.log 2>&1
echo "[prof] pmc 1"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- $T > $OUT/pmc_sq1.log 2>&1
echo "[prof] pmc 2"; rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $T > $OUT/pmc_sq2.log 2>&1
echo "[prof] pmc 3"; rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq3 -- $T > $OUT/pmc_sq3.log 2>&1 || echo "[prof] set 3 not available"
echo "[prof] pmc 4"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_sq4 -- $T > $OUT/pmc_sq4.log 2>&1 || echo "[prof] set 4 not available"
echo "[prof] pmc grbm"; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $T > $OUT/pmc_grbm.log 2>&1 || echo "[prof] grbm not available"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[prof] pmc $c"; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- $T > $OUT/pmc_$c.log 2>&1 || echo "[prof] $c not available"
done
cd $ROOT
grep -h "fused K=" $OUT/*.log | head -3
