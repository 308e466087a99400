"""Condenses tools/collect_rollout_profile.sh's rocprofv3 output into <prefix>_kernel_stats.csv and <prefix>_pmc.json (one fused K-step launch)."""
import csv, glob, hashlib, json, os, sys
src, prefix = sys.argv[1], sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
KERNEL = "jb_step_kernel"
def find(d, pat):
    r = glob.glob(os.path.join(src, d, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None          # (gpurun merges into a directory that may still hold an earlier run's files: take the newest)
out = {"what": "ONE launch of jb_step_kernel<4> advancing %d envs by K = %d control steps (jb_step_many_device; BASELINE configs[2], uniform action tape, seed 0, steps 0-%d of the episode, auto-reset included)" % (N, K, K), "K": K, "n_envs": N}
st = find("stats", "*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(prefix + "_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_ALL); w.writeheader(); w.writerows(rows[:8])
    for r in rows:
        if KERNEL in r["Name"]:
            out.update(kernel=r["Name"][:70], kernel_calls=int(r["Calls"]), kernel_avg_ns=float(r["AverageNs"]), kernel_max_ns=float(r["MaxNs"]))
def counters(d):
    f = find(d, "*counter_collection.csv")
    acc = {}
    if not f: return acc
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc
sq = {}
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_sq4", "pmc_grbm"):
    sq.update(counters(d))
out["sq"] = sq
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = counters("pmc_" + c)
    if c in v: out[c + "_KB"] = v[c]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jitterbug_amd", "libjitterbug_hip.so")
out["lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
if out.get("kernel_avg_ns"):
    t = out["kernel_max_ns"] * 1e-9          # the K-step launch (the reset-time launches of other kernels are not jb_step_kernel)
    out["launch_ms"] = t * 1e3
    out["ms_per_step"] = t * 1e3 / K
    out["env_steps_per_s_kernel_only"] = N * K / t
    if "GRBM_GUI_ACTIVE" in sq:
        out["clock_hz"] = sq["GRBM_GUI_ACTIVE"] / 8.0 / t
    if "SQ_WAVE_CYCLES" in sq and "SQ_WAVES" in sq and out.get("clock_hz"):
        out["mean_wave_life_ms"] = sq["SQ_WAVE_CYCLES"] * 4.0 / sq["SQ_WAVES"] / out["clock_hz"] * 1e3
        out["mean_wave_life_over_launch"] = out["mean_wave_life_ms"] / out["launch_ms"]
    if "SQ_INSTS_VALU" in sq:
        out["valu_insts_per_wave_substep"] = sq["SQ_INSTS_VALU"] / sq.get("SQ_WAVES", 1024.0) / (K * 50.0)
        out["valu_issue_slot_frac_of_chip"] = sq["SQ_INSTS_VALU"] * 2.0 / (1024.0 * t * out.get("clock_hz", 2.4e9))
    if "FETCH_SIZE_KB" in out and "WRITE_SIZE_KB" in out:
        out["hbm_bytes_per_env_step"] = (out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024.0 / (N * K)
        out["algorithmic_bytes_per_env_step"] = "tape 4 + reward 4 per step; state 248 + obs 60 + done 1 once per launch"
json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
