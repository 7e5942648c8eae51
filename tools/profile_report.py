"""Turns the rocprofv3 CSVs of a profiling run (gpurun_out/<tag>/...) into the summaries committed under profiles/:
  profiles/<tag>_kernel_stats.csv     kernel-trace --stats summary of `python bench.py`
  profiles/<tag>_pmc.json             PMC counters of rt_extend_kernel / rt_shade_kernel (one pass per counter group)
  profiles/hbm_traffic.json           HBM bytes per rt_extend_kernel launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; on gfx950
                                      FETCH_SIZE reports half of the bytes of wide loads), read by bench.py
usage: tools/profile_report.py gpurun_out/r01 r01 cu_cover2
"""
import csv, glob, json, os, shutil, sys
src, tag, scene = sys.argv[1], sys.argv[2], sys.argv[3]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)
ks = os.path.join(src, "bench_trace", "t_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copyfile(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(ks)):
        print("%-50s calls %6s total %10.3f ms avg %9.3f us %6s%%" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    shutil.copyfile(bj, os.path.join(out, f"{tag}_bench_under_rocprof.json"))
pmc = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = "extend" if "rt_extend_kernel" in r["Kernel_Name"] else ("shade" if "rt_shade_kernel" in r["Kernel_Name"] else None)
        if not k:
            continue
        d = pmc.setdefault(k, {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d.setdefault("_dispatches", {}).setdefault(r["Counter_Name"], 0)
        d["_dispatches"][r["Counter_Name"]] += 1
        d["VGPR_Count"], d["SGPR_Count"], d["LDS_Block_Size"], d["Scratch_Size"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")
for k, d in pmc.items():
    disp = d.pop("_dispatches")
    d["dispatches"] = max(disp.values()) if disp else 0
json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
if "extend" in pmc and "FETCH_SIZE" in pmc["extend"]:
    e = pmc["extend"]
    per_launch = (2.0 * e["FETCH_SIZE"] + e.get("WRITE_SIZE", 0.0)) * 1024.0 / e["dispatches"]
    path = os.path.join(out, "hbm_traffic.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[scene] = per_launch
    cur[scene + "_profile"] = tag
    cur[scene + "_note"] = "bytes per rt_extend_kernel launch: (2*FETCH_SIZE + WRITE_SIZE) KB over %d launches, %s" % (e["dispatches"], tag)
    # issue-side view of the same kernel: share of SIMD cycles with a VALU instruction in flight, and lanes active in it
    if all(k in e for k in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")):
        cur[scene + "_valu_busy"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
        cur[scene + "_lanes_active"] = e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"] / 64.0
    json.dump(cur, open(path, "w"), indent=1)
    print("HBM bytes per extend launch: %.1f MB" % (per_launch / 1e6))
