#!/bin/bash
# tools/ab_pmc.sh lib1.so lib2.so ... : total wave cycles and VALU instruction count per launch for each build (rocprofv3 PMC).
# The launch time is the SLOWEST wave's time and moves by several % with the seed; these sums resolve sub-% code changes.
ROOT=$(pwd)
for lib in "$@"; do
  OUT=$ROOT/gpurun_out/prof/abpmc_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && JITTERBUG_HIP_LIB=$ROOT/$lib rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-also --no-host-rate --no-steady > $OUT.log 2>&1 )
  python3 - "$OUT" "$lib" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc, n = {}, {}
for r in csv.DictReader(open(f)):
    if "jb_step_kernel" not in r["Kernel_Name"]: continue
    k = r["Counter_Name"]; acc[k] = acc.get(k, 0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
v = {k: acc[k] / n[k] for k in acc}
print("%-26s wave cycles %.2f M (mean wave %.4f ms)  VALU %.2f M  LDS %.2f M  waitcnt %.1f %%" % (sys.argv[2], v["SQ_WAVE_CYCLES"] / 1e6, v["SQ_WAVE_CYCLES"] * 4 / 1024 / 2.38e9 * 1e3, v["SQ_INSTS_VALU"] / 1e6, v["SQ_INSTS_LDS"] / 1e6, 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
done
