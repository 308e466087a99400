#!/bin/bash
# A/B of step-kernel builds on one box: tools/ab_bench.sh lib1.so lib2.so ...   (prints env-steps/s and launch ms per build)
for lib in "$@"; do
  JITTERBUG_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 50 --no-cpu-baseline --no-also --no-host-rate --seed ${SEED:-0} > gpurun_out/ab_$(basename $lib .so).json || exit 1
  python - "$lib" <<PY
import json, sys, os
f = "gpurun_out/ab_%s.json" % os.path.basename(sys.argv[1])[:-3]
d = json.loads(open(f).read().strip().split("\n")[-1]); print("%-28s %10.0f env-steps/s  launch %.4f ms  cap_hits %s  finite %s%s" % (sys.argv[1], d["value"], d["roofline"]["launch_ms"], d["solver_cap_hits"], d["finite"], "" if d["finite"] else "   <-- BROKEN BUILD, timing meaningless"))
PY
done
