"""Summarise rocprofv3 CSV output of tools/prof_passes.sh: per-kernel time and the PMC counters of rt_extend_kernel."""
import csv, glob, os, sys, json
d = sys.argv[1]
out = {}
ks = os.path.join(d, "trace", "t_kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats")
    for r in csv.DictReader(open(ks)):
        print("  %-60s calls %s total %.3f ms avg %.3f ms  %s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
        if "rt_extend_kernel" in r["Name"]:
            out["render_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
print("== counters (rt_extend_kernel)")
for f in sorted(glob.glob(os.path.join(d, "pmc*", "p_counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "rt_extend_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        out["vgpr"] = r.get("VGPR_Count"); out["sgpr"] = r.get("SGPR_Count"); out["lds"] = r.get("LDS_Block_Size"); out["scratch"] = r.get("Scratch_Size")
    for k, v in acc.items():
        print("  %-40s %.6g" % (k, v))
        out[k] = v
print(json.dumps(out))
