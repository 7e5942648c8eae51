#!/usr/bin/env python3
"""Cost of one frame as a function of the pattern columns in it: t(columns) = fixed + per_column * columns.

The fixed part is what a shard of a frame pays however few columns it holds (the first bounces' cold caches, the bounces
after the last unstarted path was handed out, the host round trips); it sets the multi-GPU efficiency of a frame that
is sharded by pattern column.  Renders cu_cover2 (depth 50) with 125, 250, 500 and 1000 columns, best of
--repeat frames each, and prints a least-squares fit as one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # before the library: its own HIP runtime has to be the first one in the process

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cu_cover2")
    ap.add_argument("--columns", type=int, nargs="+", default=[125, 250, 500, 1000])
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    world, kw = rt.get_scene(args.scene, spp=1)
    camera, bg, W, H = kw["camera"], kw["background"], kw["width"], kw["height"]
    dev = rt.DeviceScene(world, device=0)
    accum = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
    pattern_all = rt.Halton_sequence32(max(args.columns))
    rows = []
    for n in args.columns:
        pattern = np.ascontiguousarray(pattern_all[:, :n])
        best = float("inf")
        for _ in range(args.repeat + 1):  # the first frame of a size allocates
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.accumulate(camera, pattern, bg, W, H, accum.data_ptr(), max_depth=50, seed=0)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        rows.append((n, best * 1e3))
        print(f"{n:5d} columns: {best * 1e3:8.2f} ms  {W * H * n / best / 1e6:8.1f} Mpaths/s", flush=True)
    x = np.array([r[0] for r in rows], dtype=np.float64)
    y = np.array([r[1] for r in rows], dtype=np.float64)
    per_col, fixed = np.polyfit(x, y, 1) if len(rows) > 1 else (y[0] / x[0], 0.0)
    print(json.dumps({"scene": args.scene, "ms": {int(a): round(b, 3) for a, b in rows},
                      "per_column_ms": round(float(per_col), 4), "fixed_ms": round(float(fixed), 3)}))
    dev.close()


if __name__ == "__main__":
    main()
