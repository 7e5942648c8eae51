"""Per-kernel sums of the counters tools/prof_f64.sh collected: python tools/f64_pmc_summary.py gpurun_out/<tag> > profiles/<tag>_pmc.json"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
kernels = {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        kind = "extend" if "rt64_extend" in name else "shade" if "rt64_shade" in name else None
        if kind:
            k = kernels.setdefault(kind, {})
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in kernels.values():
    if k.get("SQ_INSTS_VALU") and "SQ_THREAD_CYCLES_VALU" in k:
        k["lanes_per_valu_inst"] = k["SQ_THREAD_CYCLES_VALU"] / k["SQ_INSTS_VALU"]
        k["lanes_active"] = k["lanes_per_valu_inst"] / 64.0 * k["SQ_INSTS_VALU"] / k["SQ_ACTIVE_INST_VALU"] if k.get("SQ_ACTIVE_INST_VALU") else None
    if k.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in k:
        k["valu_busy_while_running"] = k["SQ_ACTIVE_INST_VALU"] / (k["GRBM_GUI_ACTIVE"] * 32.0)
    if k.get("TCC_REQ_sum") and "TCC_HIT_sum" in k:
        k["l2_hit"] = k["TCC_HIT_sum"] / k["TCC_REQ_sum"]
    if k.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in k:
        k["waves_waiting"] = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
print(json.dumps({"scene": "cu_cover2 800x800, 16 columns, Float64; counters are serialised per kernel by the profiler (no overlap between the streams)",
                  "kernels": kernels}, indent=1))
