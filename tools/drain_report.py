#!/usr/bin/env python3
"""The last frame of a kernel trace (tools/drain_trace.sh): its launches of rt_extend_kernel / rt_shade_kernel in order, in
groups of eight bounces -- where a short frame's time goes once the unstarted paths have run out.
python tools/drain_report.py gpurun_out/<tag>/trace/t_kernel_trace.csv"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
inits = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("rt_init_kernel")]
frame = rows[inits[-1]:]
t0 = int(frame[0]["Start_Timestamp"])
ext = [r for r in frame if "rt_extend_kernel" in r["Kernel_Name"]]
shd = [r for r in frame if "rt_shade_kernel" in r["Kernel_Name"]]
end = max(int(r["End_Timestamp"]) for r in frame)
print("frame: %.2f ms, %d bounces, kernels busy %.2f ms" % ((end - t0) / 1e6, len(ext), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in frame) / 1e6))
for k in range(0, len(ext), 8):
    e = ext[k:k + 8]
    s = shd[k:k + 8]
    de = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in e) / 1e3 / len(e)
    ds = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) / 1e3 / max(1, len(s))
    grid = int(s[0].get("Grid_Size_X", s[0].get("Grid_Size", 0))) if s else 0
    print("bounces %3d-%3d at %6.2f ms: extend %7.1f us, shade %7.1f us a launch, shade grid %8d slots" % (
        k, k + len(e) - 1, (int(e[0]["Start_Timestamp"]) - t0) / 1e6, de, ds, grid))
