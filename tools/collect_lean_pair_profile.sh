#!/bin/bash
# Runs on the GPU box: rocprofv3 evidence for the LEAN + PAIR kernel (one model per env, per-substep table overlay: eight waves per CU) next to the
# one-wave PAIR kernel on the same randomised batch -> gpurun_out/prof_lean_pair/ ; summary -> gpurun_out/r04_lean_pair_summary.json
#   tools/collect_lean_pair_profile.sh [envs]          (default 16384: where jitterbug_amd.variants selects it)
set -e
N=${1:-16384}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lean_pair
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in lean ordinary; do
  FLAG="--lean"; [ $v = ordinary ] && FLAG="--no-lean"
  BENCH="python3 $ROOT/bench.py --augmented --task move_to_pose --envs-per-gpu $N --steps 60 --warmup 40 --no-cpu-baseline --no-also --no-host-rate --no-steady $FLAG"
  echo "[prof] $v stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_stats -- $BENCH > $OUT/${v}_stats.log 2>&1
  echo "[prof] $v pmc 1"; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${v}_pmc1 -- $BENCH > $OUT/${v}_pmc1.log 2>&1
  echo "[prof] $v pmc 2"; rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${v}_pmc2 -- $BENCH > $OUT/${v}_pmc2.log 2>&1
  echo "[prof] $v pmc 3"; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${v}_pmc3 -- $BENCH > $OUT/${v}_pmc3.log 2>&1 || echo "[prof] pmc3 not available"
done
cd $ROOT
python3 tools/summarise_lean.py $OUT > $ROOT/gpurun_out/r04_lean_pair_summary.json
python3 - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/r04_lean_pair_summary.json"))
for v in ("lean", "ordinary"):
    r = d[v]
    print(v, r.get("kernel", "")[:60], "avg %.3f ms" % (r["avg_ns"] / 1e6), "resident waves per SIMD %.2f" % r.get("mean_resident_waves_per_simd", float("nan")), "VALU issue slots %.1f %%" % (100 * r.get("valu_issue_slot_frac_of_chip", float("nan"))), "LDS", r.get("dispatch", {}).get("LDS_Block_Size"))
print("speedup", d.get("speedup_lean_over_ordinary"))
PY
