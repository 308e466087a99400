#!/bin/bash
# Runs on the GPU box: rocprofv3 evidence for the LEAN kernel variant at two waves per SIMD (65 536 envs, move_from_origin) next to the
# ordinary kernel on the same batch -> gpurun_out/prof_lean/.   tools/collect_lean_profile.sh ; then read with tools/summarise_lean.py
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lean
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in lean ordinary; do
  FLAG=""; [ $v = ordinary ] && FLAG="--no-lean"
  BENCH="python3 $ROOT/bench.py --envs-per-gpu 65536 --steps 60 --warmup 20 --no-cpu-baseline --no-also --no-host-rate --no-steady $FLAG"
  echo "[prof] $v stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_stats -- $BENCH > $OUT/${v}_stats.log 2>&1
  echo "[prof] $v pmc 1"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${v}_pmc1 -- $BENCH > $OUT/${v}_pmc1.log 2>&1
  echo "[prof] $v pmc 2"; rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${v}_pmc2 -- $BENCH > $OUT/${v}_pmc2.log 2>&1
  echo "[prof] $v pmc 3"; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --output-format csv -d $OUT/${v}_pmc3 -- $BENCH > $OUT/${v}_pmc3.log 2>&1 || echo "[prof] pmc3 not available"
done
cd $ROOT
python3 tools/summarise_lean.py $OUT > $ROOT/gpurun_out/r03_lean_summary.json
cat $ROOT/gpurun_out/r03_lean_summary.json
