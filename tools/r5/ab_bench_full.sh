# tools/r5/ab_bench_full.sh lib1.so lib2.so ... : the driver's bench line (steps 5-25 window + steady + full episode + fused blocks) for each build, on ONE box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for lib in "$@"; do
  name=$(basename $lib .so)
  JITTERBUG_HIP_LIB=$lib timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-host-rate > gpurun_out/r5/abfull_$name.json 2> gpurun_out/r5/abfull_$name.err || { tail -5 gpurun_out/r5/abfull_$name.err; continue; }
  python3 - gpurun_out/r5/abfull_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
rf = d["rollout_fused"]
print("%-12s window %.3f M | steady %.3f | full episode %.3f | fused k1000 %.3f (%s; mean wave %.4f ms) | k100 %.3f | policy %.3f | const1 %.3f | unconverged %s" % (sys.argv[2], d["value"]/1e6, d["steady"]["value"]/1e6, d["value_full_episode"]/1e6, rf["k1000"]["value"]/1e6, " ".join("%.3f" % x for x in rf["k1000"]["ms_per_step_by_episode"]), rf["k1000"]["wave_clock"]["mean_wave_ms_per_step"], rf["k100"]["value"]/1e6, rf["k1000_policy"]["value"]/1e6, rf["k1000_const1"]["value"]/1e6, d["solver_cap_hits"]))
PY
done
