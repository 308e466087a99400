"""Condenses the rocprofv3 output of tools/collect_profiles.sh into <prefix>_kernel_stats.csv and <prefix>_pmc_raw.json."""
import csv, glob, json, os, sys
src, prefix = sys.argv[1], sys.argv[2]
KERNEL = "jb_step_kernel"
def find(d, pat):
    r = glob.glob(os.path.join(src, d, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None          # (gpurun merges into a directory that may still hold an earlier run's files: take the newest)
out = {}
st = find("stats", "*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(prefix + "_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_ALL); w.writeheader(); w.writerows(rows[:8])
    for r in rows:
        if KERNEL in r["Name"]:
            out.update(kernel=r["Name"].split("(")[1].split(")")[-1] if False else r["Name"][:60], kernel_calls=int(r["Calls"]), kernel_avg_ns=float(r["AverageNs"]), kernel_min_ns=float(r["MinNs"]), kernel_max_ns=float(r["MaxNs"]))
def counters(d):
    f = find(d, "*counter_collection.csv")
    acc, n, disp = {}, {}, {}
    if not f: return acc, disp
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
        if not disp: disp = {x: r.get(x) for x in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size") if x in r}
    return {k: acc[k] / n[k] for k in acc}, disp
sq = {}
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_sq4", "pmc_sq5", "pmc_grbm", "pmc_sq6"):      # sq6 last: its SQ_ACTIVE_INST_VALU pairs with its SQ_THREAD_CYCLES_VALU
    c, disp = counters(d); sq.update(c)
    if disp: out["dispatch"] = disp
out["sq"] = sq
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, _ = counters("pmc_" + c)
    if c in v: out[c + "_KB"] = v[c]
# which build this is: the summary is only quoted by bench.py for the same library
import hashlib, subprocess
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jitterbug_amd", "libjitterbug_hip.so")
out["lib_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
if "SQ_WAVE_CYCLES" in sq and "SQ_WAVES" in sq and "GRBM_GUI_ACTIVE" in sq and out.get("kernel_avg_ns"):
    out["clock_hz"] = sq["GRBM_GUI_ACTIVE"] / 8.0 / (out["kernel_avg_ns"] * 1e-9)          # effective clock (MI355X_MICROARCH.md: sum over 8 XCDs)
    out["mean_wave_life_ms"] = sq["SQ_WAVE_CYCLES"] * 4.0 / sq["SQ_WAVES"] / out["clock_hz"] * 1e3   # SQ_WAVE_CYCLES counts quad-cycles
# the dispatch fields rocprofv3 prints are allocation-granule figures of the ARCH VGPR file only; the code object's own numbers
# (compiler remarks, tools/kernel_resources.sh) are recorded next to them
try:
    rep = subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_resources.sh")], capture_output=True, text=True, timeout=600).stdout
    out["code_object"] = {l.split()[0][:40]: " ".join(l.split()[1:]) for l in rep.splitlines() if "jb_step_kernel" in l}
except Exception:
    pass
json.dump(out, open(prefix + "_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
