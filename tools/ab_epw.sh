#!/bin/bash
# A/B of builds at a given batch size and envs-per-wave: tools/ab_epw.sh N EPW lib1.so lib2.so ...
N=$1; EPW=$2; shift 2
for lib in "$@"; do
  JITTERBUG_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps ${STEPS:-300} --warmup 50 --envs-per-gpu $N --envs-per-wave $EPW --no-cpu-baseline --no-also --no-host-rate > gpurun_out/abe.json 2> gpurun_out/abe.err || { tail -3 gpurun_out/abe.err; continue; }
  python - "$lib" $N $EPW <<PY
import json, sys
d = json.loads(open("gpurun_out/abe.json").read().strip().split("\n")[-1]); print("%-28s N=%-6s epw=%s %10.0f env-steps/s  launch %.4f ms  finite %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"], d["roofline"]["launch_ms"], d["finite"]))
PY
done
