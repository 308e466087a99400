cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout -k 10 400 python -m pytest tests/test_bench_launch.py -q -m gpu -k "degraded or rccl_on_one or one_rank" > gpurun_out/r5/gputests5.log 2>&1; tail -5 gpurun_out/r5/gputests5.log
timeout -k 10 400 bash tools/ab_pmc.sh ab_build/r3.so ab_build/base.so ab_build/new.so > gpurun_out/r5/abpmc3.txt 2>&1
cat gpurun_out/r5/abpmc3.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_driver.json 2> gpurun_out/r5/bench_driver.err; tail -c 600 gpurun_out/r5/bench_driver.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_driver.json").read().strip().split("\n")[-1])
rf = d["rollout_fused"]
print("driver window %.3f M | steady %.3f M | full episode %.3f M | fused k1000 %.3f M %s | k100 %.3f | policy %.3f | const1 %.3f | cap %s" % (d["value"]/1e6, d["steady"]["value"]/1e6, d["value_full_episode"]/1e6, rf["k1000"]["value"]/1e6, ["%.3f" % x for x in rf["k1000"]["ms_per_step_by_episode"]], rf["k100"]["value"]/1e6, rf["k1000_policy"]["value"]/1e6, rf["k1000_const1"]["value"]/1e6, d["solver_cap_hits"]))
PY
