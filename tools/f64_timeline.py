"""The kernels of the last Float64 frame in a rocprofv3 kernel trace, as a timeline: python tools/f64_timeline.py <trace csv>
Prints, per 2 ms of the frame, the time the extend / shade kernels of all streams were running and their launch counts."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        kind = "extend" if "rt64_extend" in name else "shade" if "rt64_shade" in name else "compact" if "rt64_compact" in name else \
               "sum" if "rt64_sum" in name else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
sums = [i for i, r in enumerate(rows) if r[2] == "sum"]
start = sums[-2] + 1 if len(sums) > 1 else 0          # the frame after the warm-up one
frame = rows[start:sums[-1] + 1]
t0, t1 = frame[0][0], frame[-1][1]
print("frame %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(frame)))
step = 2_000_000
b = t0
while b < t1:
    sel = [r for r in frame if b <= r[0] < b + step]
    busy = {k: sum(r[1] - r[0] for r in sel if r[2] == k) / 1e3 for k in ("extend", "shade", "compact")}
    n = {k: sum(1 for r in sel if r[2] == k) for k in ("extend", "shade", "compact")}
    longest = max((r[1] - r[0] for r in sel if r[2] == "extend"), default=0) / 1e3
    print("%6.1f ms: extend %3d launches %7.0f us (longest %5.0f), shade %3d %6.0f us, compact %d" %
          ((b - t0) / 1e6, n["extend"], busy["extend"], longest, n["shade"], busy["shade"], n["compact"]))
    b += step
