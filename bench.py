#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the path-tracing loop on cu_cover2 (800x800, 1000 spp), the configuration
the reference's headline metric is quoted on, with the spp dimension sharded over N GPUs (one process per GPU) and the per-GPU
accumulators summed by one RCCL all-reduce.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame: every rank traces its share of the pattern columns into an un-normalised FP32
accumulator in HBM, the accumulators are all-reduced, the image is finalised (x 1/N, sqrt) on the device and rank 0
copies it to host memory (pinned), all inside the timed region.  The scene, camera and pattern are resident on the GPU
before the timed region starts.  Rank 0 prints one JSON line.

  python bench.py --mode group --gpus N     one process, N GPUs: the same frames through rt_group_render (the call the
                                            reference's one-process render(...; gpu=true) makes), members' shards summed
                                            by RCCL inside the library.  With more than one rank the default mode also
                                            runs this once the ranks' own frames are timed (in a child process of rank
                                            0, the other ranks idle) and reports it as "group_mode" in the same line.
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


TRAVERSAL = ("node", "sphere", "moving_sphere", "rect", "box", "transform", "medium", "triangle")


def extend_roofline(rt, scene, events, cam, pattern, bg, W, H, accum, stream, rank=0, world_size=1):
    """Roofline object of rt_extend_kernel (BVH traversal + intersection, the dominant kernel) for one scene: its
    launches of one frame are bracketed with HIP events on the launch stream (RT_FLAG_TIMING); algorithmic work per
    launch = the traversal / intersection share of the per-path work of the reference algorithm (`events`, counted by the
    CPU checker) x the paths of the frame / launches."""
    from raytracing_amd import _ffi
    n = pattern.shape[1]
    scene.accumulate(cam, pattern, bg, W, H, accum.data_ptr(), sample_first=rank, sample_stride=world_size,
                     flags=_ffi.RT_FLAG_TIMING, stream=stream)  # this rank's shard of the columns
    tst = scene.counters()
    trav = sum(events[k] * COST[k][0] for k in TRAVERSAL)
    trav_b = sum(events[k] * COST[k][1] for k in TRAVERSAL)
    launches = max(1, tst.extend_launches)
    avg_s = tst.extend_seconds / launches
    paths_per_launch = W * H * len(range(rank, n, world_size)) / launches
    tflops = trav * paths_per_launch / avg_s / 1e12
    return {"bound": "valu", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / PEAK_FP32_TFLOPS, "traffic": None,
            "kernel": "rt_extend_kernel", "launches_per_frame": int(launches),
            "avg_launch_ms": avg_s * 1e3, "extend_ms_per_frame": tst.extend_seconds * 1e3,
            "shade_ms_per_frame": tst.shade_seconds * 1e3,
            "flops_per_path": trav, "bytes_per_path": trav_b,
            "node_fetch_GBps": trav_b * paths_per_launch / avg_s / 1e9}


def group_mode(args):
    """--mode group: one process drives --gpus devices through rt_group_render (scene replicated per device, pattern
    columns dealt to the members, accumulators summed on the first device by RCCL, image copied to host memory: all of
    it inside the timed call).  Prints one JSON line."""
    import raytracing_amd as rt
    from raytracing_amd import _ffi
    import ctypes as C
    count = C.c_int(0)
    _ffi.check(_ffi.lib().rt_device_count(C.byref(count)))
    if count.value == 0:
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    g = min(args.gpus, count.value) if args.gpus > 0 else count.value
    world, kw = rt.get_scene(args.scene, spp=args.spp)
    W, H, cam, bg, pattern = kw["width"], kw["height"], kw["camera"], kw["background"], kw["pattern"]
    n = pattern.shape[1]
    group = rt.DeviceGroup(world, devices=list(range(g)))
    try:
        for _ in range(args.warmup):
            group.render(cam, pattern, bg, W, H)
        t0 = time.perf_counter()
        device_s = 0.0
        for _ in range(args.steps):
            _, st = group.render(cam, pattern, bg, W, H)
            device_s += st.seconds
        elapsed = time.perf_counter() - t0
    finally:
        group.close()
    paths = W * H * n
    line = {"metric": "Mpaths/s (whole node), %s %d spp" % (args.scene, n), "value": paths * args.steps / elapsed / 1e6,
            "unit": "Mpaths/s", "n_gpus": g, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3, "first_device_ms_per_step": device_s / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp (Halton), depth 50, seeded scene" % (args.scene, W, H, n),
                       "parallelism": "one process, rt_group_render over %d GPU(s): spp dealt to the members, ncclReduce of the FP32 accumulators" % g,
                       "paths_per_frame": paths},
            "roofline": None, "cpu_baseline": None}
    print(json.dumps(line), flush=True)


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
    ap.add_argument("--mode", choices=("ranks", "group"), default="ranks",
                    help="ranks: one process per GPU (torch.distributed); group: this one process drives --gpus devices through rt_group_render")
    ap.add_argument("--with-group", action="store_true", help="ranks mode: also time the group mode afterwards (default with more than one rank)")
    args = ap.parse_args()
    if args.mode == "group":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--mode group is one process for all GPUs: start it without torchrun")
        return group_mode(args)

    import numpy as np
    import torch
    import raytracing_amd as rt

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RT_BENCH_BACKEND=gloo is a rehearsal of the N > 1 control flow on a host with fewer GPUs than ranks: the ranks then
    # share the devices there are and the all-reduce goes through host memory (its numbers mean nothing)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    if world_size > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    world, kw = rt.get_scene(args.scene, spp=args.spp)
    W, H, cam, bg, pattern = kw["width"], kw["height"], kw["camera"], kw["background"], kw["pattern"]
    n = pattern.shape[1]
    scene = rt.DeviceScene(world, device=local_rank)
    accum = torch.zeros(3 * W * H, dtype=torch.float32, device=dev)
    out = torch.zeros(3 * W * H, dtype=torch.float32, device=dev)
    host_img = torch.empty(3 * W * H, dtype=torch.float32).pin_memory() if rank == 0 else None  # where the frame ends
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
        if host_img is not None:
            host_img.copy_(out, non_blocking=True)   # the finished image in host memory (what Array(d_img) is, main.jl:38)

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
            roofline = extend_roofline(rt, scene, events, cam, pattern, bg, W, H, accum, stream, rank, world_size)
            roofline["flops_per_path_incl_shading"] = flops_pp
            roofline["events_per_path"] = {k: round(v, 3) for k, v in events.items()}
            # HBM bytes per launch and the VALU readings come from PMC passes (rocprofv3 --pmc, run apart from the timed
            # run as the counters need); the committed summary of the latest passes is read here and named as the source
            pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(pmc) and world_size == 1:  # (the passes were made with one GPU tracing every column)
                with open(pmc) as f:
                    prof = json.load(f)
                roofline["traffic"] = prof.get(args.scene)
                roofline["traffic_source"] = "profiles/hbm_traffic.json (%s): not measured by this run" % prof.get(args.scene + "_profile", "PMC passes")
                if args.scene + "_valu_busy" in prof:
                    roofline["valu_busy_profiled"] = prof[args.scene + "_valu_busy"]
                    roofline["lanes_active_profiled"] = prof[args.scene + "_lanes_active"]
        line["roofline"] = roofline
        line["cpu_baseline"] = cpu
        if world_size == 1 and not args.no_cpu and args.scene == "cu_cover2":
            # the reference's own CPU-runnable configuration in full: cornell_box 400x400, 100 pattern columns, every
            # host core of this job (threads over image columns, src/Ray_Tracing.jl:78), and the GPU beside it
            w1, k1 = rt.get_scene("cornell_box", spp=100)
            a1 = (k1["camera"], k1["pattern"], k1["background"], k1["width"], k1["height"])
            ref1 = O.OracleScene(w1)
            runs = []
            for _ in range(2):
                _, st1 = ref1.render(*a1, threads=cores, stats=True)
                runs.append(st1.paths / st1.seconds / 1e6)
            sc1 = rt.DeviceScene(w1, device=local_rank)
            gpu_runs = []
            for _ in range(3):
                t1 = time.perf_counter()
                sc1.render(*a1)
                gpu_runs.append(time.perf_counter() - t1)
            sc1.close()
            paths1 = k1["width"] * k1["height"] * 100
            line["cpu_baseline_configs"] = {"cornell_box 100 spp": {
                "value": max(runs), "unit": "Mpaths/s", "cores": cores, "kind": "port", "runs": [round(r, 3) for r in runs],
                "s_per_frame": paths1 / max(runs) / 1e6,
                "sample": "the whole configuration: cornell_box %dx%d, 100 pattern columns = %d paths, best of %d frames" % (
                    k1["width"], k1["height"], paths1, len(runs)),
                "gpu_Mpaths_per_s": paths1 / min(gpu_runs) / 1e6, "gpu_s_per_frame": min(gpu_runs),
                "gpu_note": "rt_render of the same frame on one MI355X, host to host (scene resident), best of 3"}}
        if world_size == 1 and not args.no_other and args.scene == "cu_cover2":
            # the reference's other benchmark configurations, one GPU each, on a resident scene: one warm-up frame, then
            # three timed frames (accumulate, finalize, image to pinned host memory), and the roofline object of the
            # extend kernel from a fourth (events of the reference algorithm from 4 pattern columns of the CPU checker)
            other = {}
            for key, spp in (("cover1", 500), ("teapot", 1000), ("cornell_box", 1000), ("cornell_smoke_box", 10000)):
                w2, k2 = rt.get_scene(key, spp=spp)
                sc2 = rt.DeviceScene(w2, device=local_rank)
                a2 = (k2["camera"], k2["pattern"], k2["background"], k2["width"], k2["height"])
                acc2 = torch.zeros(3 * k2["width"] * k2["height"], dtype=torch.float32, device=dev)
                host2 = torch.empty(acc2.numel(), dtype=torch.float32).pin_memory()
                frames = []
                for f in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    sharded_frame(sc2, *a2, acc2, acc2, 0, 1, stream=stream)
                    host2.copy_(acc2, non_blocking=True)
                    torch.cuda.synchronize()
                    if f > 0:
                        frames.append(time.perf_counter() - t1)
                paths = k2["width"] * k2["height"] * spp
                mean = sum(frames) / len(frames)
                entry = {"Mpaths_per_s": paths / mean / 1e6, "s_per_frame": mean, "frames_s": [round(x, 5) for x in frames],
                         "paths": paths, "size": "%dx%d" % (k2["width"], k2["height"])}
                if not args.no_cpu:
                    _, st2 = O.OracleScene(w2).render(k2["camera"], rt.Halton_sequence32(4), k2["background"], k2["width"], k2["height"],
                                                      threads=cores, stats=True)
                    entry["cpu_Mpaths_per_s"] = st2.paths / st2.seconds / 1e6
                    entry["cpu_sample"] = "4 of %d pattern columns at full size, %d threads, one run (%.1f s)" % (spp, cores, st2.seconds)
                    entry["roofline"] = extend_roofline(rt, sc2, algorithmic_work(st2)[2], *a2, acc2, stream)
                other["%s %d spp" % (key, spp)] = entry
                sc2.close()
            # the Float64 mode (render(Float64, ...)): the same scene with double tables through rt_*_f64, a tenth of the
            # columns (it walks the reference's own trees in double: about a sixth of the Float32 path's rate)
            f64_cols = max(1, n // 10)
            w3, k3 = rt.get_scene(args.scene, spp=f64_cols, F=np.float64)
            sc3 = rt.DeviceScene(w3, device=local_rank, F=np.float64)
            acc3 = torch.zeros(3 * W * H, dtype=torch.float64, device=dev)
            frames = []
            for f in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sc3.accumulate(k3["camera"], k3["pattern"], k3["background"], W, H, acc3.data_ptr(), stream=stream)
                sc3.finalize(acc3.data_ptr(), f64_cols, W, H, acc3.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                if f > 0:
                    frames.append(time.perf_counter() - t1)
            best = sum(frames) / len(frames)
            f64 = {"Mpaths_per_s": W * H * f64_cols / best / 1e6, "s_per_frame": best, "frames_s": [round(x, 5) for x in frames],
                   "paths": W * H * f64_cols, "size": "%dx%d" % (W, H), "dtype": "f64"}
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
    scene.close()
    store = dist.distributed_c10d._get_default_store() if world_size > 1 else None
    if rank == 0:
        if world_size > 1 or args.with_group:
            # the other multi-GPU path, the one the reference's one-process render() would take: rt_group_render over
            # the same GPUs, in a child process of its own (this process keeps its ranks' state; the other ranks wait on
            # the host side, their GPUs idle)
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", "group", "--gpus", str(world_size), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--scene", args.scene, "--spp", str(args.spp)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                    "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60 + 20 * (args.steps + args.warmup))
                rows = [x for x in r.stdout.splitlines() if x.startswith("{")]
                if r.returncode == 0 and rows:
                    g = json.loads(rows[-1])
                    line["group_mode"] = {k: g[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step", "first_device_ms_per_step")}
                    line["group_mode"]["parallelism"] = g["config"]["parallelism"]
                else:
                    line["group_mode"] = {"error": (r.stderr or r.stdout)[-400:], "returncode": r.returncode}
            except subprocess.TimeoutExpired:
                line["group_mode"] = {"error": "the child process did not finish in time"}
        print(json.dumps(line), flush=True)
        if store is not None:
            store.set("bench_done", "1")
    elif store is not None:
        from datetime import timedelta
        store.wait(["bench_done"], timedelta(hours=1))  # on the host: a collective here would keep this rank's GPU busy
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
