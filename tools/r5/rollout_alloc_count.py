"""Counts hipMalloc / hipFree calls per rollout call in a rocprofv3 --hip-trace of tools/r5/rollout10.py (marks: wall-clock ns after each call)."""
import csv, glob, sys
d, marks_f = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Function"] in ("hipMalloc", "hipFree", "hipMallocAsync", "hipFreeAsync")]
ts = sorted((int(r["Start_Timestamp"]), r["Function"]) for r in rows)
print("allocation calls in the whole run: %d hipMalloc, %d hipFree" % (sum(1 for _, fn in ts if "Malloc" in fn), sum(1 for _, fn in ts if "Free" in fn)))
# order of calls: the staging allocations all sit before the first call's end; index them by position among all HIP calls of the run
allrows = list(csv.DictReader(open(f)))
launch_idx = [i for i, r in enumerate(allrows) if r["Function"] in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtLaunchKernel")]
alloc_idx = [i for i, r in enumerate(allrows) if r["Function"] in ("hipMalloc", "hipMallocAsync")]
# the rollout launches are the LAST 20 step launches (10 x jb_step_many + 10 x jb_rollout_policy, plus observe kernels): count the allocations after the 4th launch from the first rollout on
n_launch = len(launch_idx)
first_rollout_end = launch_idx[-(10 * 3) + 2] if n_launch >= 30 else launch_idx[0]
late = [i for i in alloc_idx if i > first_rollout_end]
print("HIP API calls %d, kernel launches %d; hipMalloc calls AFTER the first rollout()/rollout_policy() pair: %d" % (len(allrows), n_launch, len(late)))
