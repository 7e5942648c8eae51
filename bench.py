#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the path-tracing loop on cu_cover2 (800x800, 1000 spp), the configuration
the reference's headline metric is quoted on, with the spp dimension sharded over N GPUs (one process per GPU) and the per-GPU
accumulators summed by one RCCL all-reduce.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame: every rank traces its share of the pattern columns into an un-normalised FP32
accumulator in HBM, the accumulators are all-reduced, and the image is finalised (x 1/N, sqrt) on the device.
The scene, camera and pattern are resident on the GPU before the timed region starts.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic cost per counted event of the reference algorithm (flops, bytes), from the formulas cited:
#   node visit: src/bounding_box.jl:70-84 (6 sub, 6 mul, 3 min, 3 max, 3+3 for the 4-way max/min), 24 B box + 8 B links
#   sphere: src/sphere.jl:46-63; rect: src/aarect.jl:50-57; box = 6 rects; transform: src/transform.jl:76-98,105-113
#   triangle: src/triangle.jl:74-84; medium: two boundary tests + log; scatter per material: src/hittable.jl:36-75
COST = {
    "node": (24, 32), "sphere": (31, 16), "moving_sphere": (45, 36), "rect": (8, 20), "box": (48, 24),
    "transform": (54, 96), "medium": (72, 40), "triangle": (52, 36),
    "hit_completion": (25, 0), "raygen": (40, 8),
    "lambertian": (29, 0), "metal": (42, 0), "dielectric": (45, 0), "isotropic": (17, 0), "emissive": (0, 0),
}
PEAK_FP32_TFLOPS = 157.3  # MI355X FP32 vector peak (spec), the compute roof of this kernel
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 vector peak (spec), for the Float64 mode


def algorithmic_work(st):
    """(flops, bytes) per path from the event counts of the reference algorithm (counted by the CPU checker)."""
    pt = list(st.prim_tests)
    sc = list(st.scatters)
    ev = {
        "node": st.node_visits, "sphere": pt[1], "moving_sphere": pt[2], "rect": pt[3] + pt[4] + pt[5], "box": pt[6],
        "transform": pt[7], "medium": pt[8], "triangle": pt[10], "hit_completion": sum(sc), "raygen": st.paths,
        "lambertian": sc[1], "metal": sc[2], "dielectric": sc[3], "emissive": sc[4], "isotropic": sc[5],
    }
    flops = sum(ev[k] * COST[k][0] for k in ev)
    nbytes = sum(ev[k] * COST[k][1] for k in ev)
    return flops / st.paths, nbytes / st.paths, {k: v / st.paths for k, v in ev.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cu_cover2")
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--cpu-spp", type=int, default=24, help="pattern columns of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip the other benchmark configurations of the reference (one frame each)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import raytracing_amd as rt

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    world, kw = rt.get_scene(args.scene, spp=args.spp)
    W, H, cam, bg, pattern = kw["width"], kw["height"], kw["camera"], kw["background"], kw["pattern"]
    n = pattern.shape[1]
    scene = rt.DeviceScene(world, device=local_rank)
    accum = torch.zeros(3 * W * H, dtype=torch.float32, device=dev)
    out = torch.zeros(3 * W * H, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    from raytracing_amd.distributed import sharded_frame

    def step(k=None):
        if k is not None:
            ev0[k].record()
        sharded_frame(scene, cam, pattern, bg, W, H, accum, out, rank, world_size,
                      all_reduce=dist.all_reduce if world_size > 1 else None, stream=stream,
                      after_accumulate=(lambda: ev1[k].record()) if k is not None else None)

    def fence():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(1, args.steps)

    if rank == 0:
        paths_per_frame = W * H * n
        value = paths_per_frame * args.steps / elapsed / 1e6
        line = {
            "metric": "Mpaths/s (whole node), %s %d spp" % (args.scene, n), "value": value, "unit": "Mpaths/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3, "s_per_frame": elapsed / max(1, args.steps),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp (Halton), depth 50, seeded scene" % (args.scene, W, H, n),
                       "parallelism": "spp sharded over %d GPU(s), RCCL all-reduce of the FP32 accumulator" % world_size,
                       "paths_per_frame": paths_per_frame},
        }
        roofline, cpu = None, None
        if not args.no_cpu:
            # bounded CPU sample: the checker (a C++ port of the reference's CPU path, threads over columns) renders
            # the same scene at full resolution with a few pattern columns; it also counts the events of the
            # reference algorithm, which define the algorithmic flops/bytes per path used for the roofline.
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_ffi as O
            # threads: the CPU share of this job (OMP_NUM_THREADS where the harness sets it -- 16 per GPU on the boxes
            # this runs on, although os.cpu_count() shows the whole machine), else every core the host has
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
            ref = O.OracleScene(world)
            # (with more than one GPU the CPU baseline is not reported; a few columns then give the event counts)
            cpu_cols = args.cpu_spp if world_size == 1 else min(4, args.cpu_spp)
            cpu_pat = rt.Halton_sequence32(cpu_cols)
            runs = []
            for _ in range(3 if world_size == 1 else 1):
                _, st = ref.render(cam, cpu_pat, bg, W, H, threads=cores, stats=True)
                runs.append(st.paths / st.seconds / 1e6)
            if world_size == 1:
                cpu = {"value": max(runs), "unit": "Mpaths/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
                       "runs": [round(r, 3) for r in runs],
                       "sample": "%s %dx%d, %d of %d pattern columns, best of %d runs (%.1f s each)" % (
                           args.scene, W, H, cpu_cols, n, len(runs), st.seconds)}
            flops_pp, bytes_pp, events = algorithmic_work(st)
            # Dominant kernel = rt_extend_kernel (BVH traversal + intersection).  Its launches of one extra frame are
            # bracketed with HIP events on the launch stream (RT_FLAG_TIMING); algorithmic work per launch = the
            # traversal/intersection share of the per-path work x the paths of a frame / launches.
            from raytracing_amd import _ffi
            scene.accumulate(cam, pattern, bg, W, H, accum.data_ptr(), sample_first=rank, sample_stride=world_size,
                             flags=_ffi.RT_FLAG_TIMING, stream=stream)  # this rank's shard of the columns
            tst = scene.counters()
            trav = sum(events[k] * COST[k][0] for k in ("node", "sphere", "moving_sphere", "rect", "box", "transform", "medium", "triangle"))
            trav_b = sum(events[k] * COST[k][1] for k in ("node", "sphere", "moving_sphere", "rect", "box", "transform", "medium", "triangle"))
            launches = max(1, tst.extend_launches)
            avg_s = tst.extend_seconds / launches
            paths_per_launch = W * H * len(range(rank, n, world_size)) / launches
            tflops = trav * paths_per_launch / avg_s / 1e12
            roofline = {"bound": "valu", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / PEAK_FP32_TFLOPS, "traffic": None,
                        "kernel": "rt_extend_kernel", "launches_per_frame": int(launches),
                        "avg_launch_ms": avg_s * 1e3, "extend_ms_per_frame": tst.extend_seconds * 1e3,
                        "shade_ms_per_frame": tst.shade_seconds * 1e3,
                        "flops_per_path": trav, "bytes_per_path": trav_b, "flops_per_path_incl_shading": flops_pp,
                        "node_fetch_GBps": trav_b * paths_per_launch / avg_s / 1e9,
                        "events_per_path": {k: round(v, 3) for k, v in events.items()}}
            # HBM bytes per launch and the VALU readings come from PMC passes (rocprofv3 --pmc, run apart from the timed
            # run as the counters need); the committed summary of the latest passes is read here and named as the source
            pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(pmc):
                with open(pmc) as f:
                    prof = json.load(f)
                roofline["traffic"] = prof.get(args.scene)
                roofline["traffic_source"] = "profiles/hbm_traffic.json (%s): not measured by this run" % prof.get(args.scene + "_profile", "PMC passes")
                if args.scene + "_valu_busy" in prof:
                    roofline["valu_busy_profiled"] = prof[args.scene + "_valu_busy"]
                    roofline["lanes_active_profiled"] = prof[args.scene + "_lanes_active"]
        line["roofline"] = roofline
        line["cpu_baseline"] = cpu
        if world_size == 1 and not args.no_other and args.scene == "cu_cover2":
            # the reference's other benchmark configurations, one GPU each, second frame of a resident scene (buffers warm)
            other = {}
            for key, spp in (("cover1", 500), ("teapot", 1000), ("cornell_box", 1000), ("cornell_smoke_box", 10000)):
                w2, k2 = rt.get_scene(key, spp=spp)
                sc2 = rt.DeviceScene(w2, device=local_rank)
                a2 = (k2["camera"], k2["pattern"], k2["background"], k2["width"], k2["height"])
                acc2 = torch.zeros(3 * k2["width"] * k2["height"], dtype=torch.float32, device=dev)
                best = None
                for _ in range(2):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    sharded_frame(sc2, *a2, acc2, acc2, 0, 1, stream=stream)
                    torch.cuda.synchronize()
                    best = time.perf_counter() - t1
                paths = k2["width"] * k2["height"] * spp
                other["%s %d spp" % (key, spp)] = {"Mpaths_per_s": paths / best / 1e6, "s_per_frame": best, "paths": paths,
                                                  "size": "%dx%d" % (k2["width"], k2["height"])}
                sc2.close()
            # the Float64 mode (render(Float64, ...)): the same scene with double tables through rt_*_f64, a tenth of the
            # columns (it walks the reference's own trees in double: about a sixth of the Float32 path's rate)
            f64_cols = max(1, n // 10)
            w3, k3 = rt.get_scene(args.scene, spp=f64_cols, F=np.float64)
            sc3 = rt.DeviceScene(w3, device=local_rank, F=np.float64)
            acc3 = torch.zeros(3 * W * H, dtype=torch.float64, device=dev)
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sc3.accumulate(k3["camera"], k3["pattern"], k3["background"], W, H, acc3.data_ptr(), stream=stream)
                sc3.finalize(acc3.data_ptr(), f64_cols, W, H, acc3.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                best = time.perf_counter() - t1
            f64 = {"Mpaths_per_s": W * H * f64_cols / best / 1e6, "s_per_frame": best, "paths": W * H * f64_cols,
                   "size": "%dx%d" % (W, H), "dtype": "f64"}
            if roofline is not None:
                # the same algorithmic work per path (the events of the reference algorithm, counted on the Float32 scene:
                # the Float64 scene draws its random parts from the same stream two words at a time) against the FP64
                # vector peak; whole frame (extend + shade), wall clock
                tf = roofline["flops_per_path_incl_shading"] * W * H * f64_cols / best / 1e12
                f64["roofline"] = {"bound": "valu", "achieved": tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP64_TFLOPS,
                                   "kernel": "rt64_extend_kernel + rt64_shade_kernel (whole frame)"}
            other["%s %d spp float64" % (args.scene, f64_cols)] = f64
            sc3.close()
            line["other_configs"] = other
        print(json.dumps(line), flush=True)
    scene.close()
    if world_size > 1:
        dist.barrier()  # rank 0 measured one more frame for the roofline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
