"""Condenses tools/collect_lean_profile.sh's rocprofv3 output: LEAN kernel (two waves per SIMD) vs the ordinary one, 65 536 envs."""
import csv, glob, json, os, sys
src = sys.argv[1]
out = {}
for v in ("lean", "ordinary"):
    r = {}
    f = glob.glob(os.path.join(src, v + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        for row in csv.DictReader(open(f[0])):
            if "jb_step_kernel" in row["Name"]:
                r.update(kernel=row["Name"][:70], calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]), min_ns=float(row["MinNs"]), max_ns=float(row["MaxNs"]))
    sq = {}
    for d in ("_pmc1", "_pmc2", "_pmc3"):
        g = glob.glob(os.path.join(src, v + d, "**", "*counter_collection.csv"), recursive=True)
        if not g:
            continue
        acc, n = {}, {}
        for row in csv.DictReader(open(g[0])):
            if "jb_step_kernel" not in row["Kernel_Name"]:
                continue
            k = row["Counter_Name"]; acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
            r.setdefault("dispatch", {x: row.get(x) for x in ("LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Workgroup_Size", "Grid_Size") if x in row})
        sq.update({k: acc[k] / n[k] for k in acc})
    r["sq"] = sq
    if sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_INSTS_VALU"):
        r["valu_cycles_per_inst_per_wave"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
    if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_BUSY_CYCLES"):
        # SQ_WAVE_CYCLES (summed over waves) / SQ_BUSY_CYCLES (summed over SEs) ~ resident waves; per SIMD: / 1024 SIMDs x (32 SE-instances)
        r["wave_cycles_over_busy_cycles"] = sq["SQ_WAVE_CYCLES"] / sq["SQ_BUSY_CYCLES"]
    if sq.get("GRBM_GUI_ACTIVE") and r.get("avg_ns") and sq.get("SQ_WAVE_CYCLES"):
        clk = sq["GRBM_GUI_ACTIVE"] / 8.0 / (r["avg_ns"] * 1e-9)
        r["clock_hz"] = clk
        # mean number of resident waves per SIMD over the launch = total wave-cycles / (1024 SIMDs x launch cycles)
        r["mean_resident_waves_per_simd"] = sq["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * r["avg_ns"] * 1e-9 * clk)
        r["valu_issue_slot_frac_of_chip"] = sq.get("SQ_INSTS_VALU", 0) * 2.0 / (1024.0 * r["avg_ns"] * 1e-9 * clk)
    out[v] = r
if out["lean"].get("avg_ns") and out["ordinary"].get("avg_ns"):
    out["speedup_lean_over_ordinary"] = out["ordinary"]["avg_ns"] / out["lean"]["avg_ns"]
print(json.dumps(out, indent=1))
