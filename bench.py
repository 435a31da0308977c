#!/usr/bin/env python3
"""bench.py -- Msamples/s of HomemadeRayTracer's render loop on N MI355X.

Default workload: the Cornell box with three balls, 1024x1024 @1024 spp, depth 50 (BASELINE.json configs[1], the
configuration the metric is quoted on).  `--workload` selects the other BASELINE configurations:

    cornell    Cornell box 3-ball 1024x1024 @1024 spp              (configs[1]; list walk, VALU-bound)
    spheres4   ground + three big spheres 200x100 @100 spp          (configs[0]; list walk, VALU-bound)
    spheres    ~485 random spheres 1200x800 @500 spp                (configs[2]; own SAH tree walked near-first)
    spheres1m  ~1 M spheres 1920x1080 @256 spp                      (configs[3]; same walk)

One step = one complete TraceRay of the workload: every rank renders its slice of the samples-per-pixel range into a
device-resident float radiance-sum buffer (mirt_render_device), the buffers are summed onto rank 0 with one RCCL reduce,
rank 0 resolves to RGBA8 (mean, sqrt gamma, clamp, x255).  Scene upload and image files are outside the timed region;
nothing crosses PCIe inside it.  Total work is fixed, so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
        --gpus N > 1 without a launcher: this process starts N ranks itself (one per GPU, RCCL), before touching a GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...          (also fine)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8(d), algorithmic work (FLOP = one fp32 add/sub/mul/compare/min/max/div/sqrt; RNG integer work and
# address arithmetic excluded).  Cornell: brute force over its 9 primitives, 256 FLOP per ray.  Sphere scenes:
# boxes_tested x 27 + spheres_tested x 32 + 20 (closest-hit finalise) + 32 (mean scatter) per ray, the two counters
# taken per workload from the CPU oracle walking the same tree in the same (near-first) order on a bounded sample.
FLOP_PER_RAY_CORNELL = 256.0
FLOP_BOX, FLOP_SPHERE, FLOP_FINALISE, FLOP_SCATTER = 27.0, 32.0, 20.0, 32.0
FLOP_PER_SAMPLE_FIXED = 38.0    # ray generation + accumulate
BYTES_BOX, BYTES_SPHERE = 32.0, 16.0            # SURVEY 8(d), configs[3]: bytes per ray = boxes x 32 + spheres x 16 (path state lives in registers here: 0)
# MI355X_MICROARCH.md: 256 CU x 4 SIMD, a wave64 VALU op issues in 2 cycles at 2.4 GHz -> 78.6 T lane-ops/s for non-FMA
# fp32 (157.3 TFLOP/s is the same rate counted as FMAs); contraction is off on this path, so one flop = one lane-op.
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
RAD_BYTES = 12.0                # one 12-byte slot (three floats) per finished path, written once (k_trace), read once (k_accumulate)
PATH_STATE_BYTES = 44.0         # wavefront engine only

WORKLOADS = {
    # name: (label, width, height, spp, cpu sample spp, binding roofline)
    "cornell": ("Cornell box 3-ball", 1024, 1024, 1024, 128, "valu"),
    "spheres4": ("4-sphere scene (ground + lambertian/metal/dielectric)", 200, 100, 100, 1000, "valu"),
    "spheres": ("random-spheres scene (485 spheres)", 1200, 800, 500, 48, "valu"),
    "spheres1m": ("1M-sphere procedural scene", 1920, 1080, 256, 4, "hbm"),
}


# The synthetic inputs: what generates each scene, recorded in the JSON line's config.  spheres1m: the reference's
# generator (World.cpp:41-54) on a 1000 x 1000 grid of cells; the camera keeps its target and lens, its position is raised from
# the reference's (11,2,3) to (26,6,6) so that a large part of the field is in view (the survey proposed (11,20,3)*k and asked for
# the value used to be recorded: this is it).
SCENE_PARAMS = {
    "cornell": {"camera": {"look_from": [0.0, 0.64, 4.63], "look_at": [0.0, -0.26, -1.0], "fov_deg": 30.0, "aperture": 0.0}},
    "spheres4": {"camera": {"look_from": [11.0, 2.0, 3.0], "look_at": [0.0, 0.6, 0.0], "fov_deg": 20.0, "aperture": 0.1}, "grid": 0},
    "spheres": {"camera": {"look_from": [11.0, 2.0, 3.0], "look_at": [0.0, 0.6, 0.0], "fov_deg": 20.0, "aperture": 0.1}, "grid": 11, "scene_seed": 1},
    "spheres1m": {"camera": {"look_from": [26.0, 6.0, 6.0], "look_at": [0.0, 0.6, 0.0], "fov_deg": 20.0, "aperture": 0.1}, "grid": 500, "scene_seed": 1},
}


def make_scene(name, aspect):
    from raytracinginaweekend_amd import scene
    if name == "cornell":
        return scene.cornell_box(aspect)
    par = SCENE_PARAMS[name]
    return scene.spheres_world(aspect, small_spheres=par["grid"] > 0, grid=par["grid"] or 11, look_from=tuple(par["camera"]["look_from"]))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cornell")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0: the workload's default)")
    ap.add_argument("--engine", type=int, default=1, help="1: persistent k_trace (default), 0: wavefront k_bounce per depth")
    ap.add_argument("--ppm", default="", help="write the last image as ASCII PPM (outside the timed region)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend; gloo (through host memory) only to rehearse the N>1 path where RCCL cannot run")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="--gpus N > 1 started without a launcher: seconds the self-started ranks get before they are stopped (exit status 124)")
    ap.add_argument("--log-dir", default="", help="where the self-started ranks' stderr goes (bench_rank<r>.err; default: the current directory)")
    return ap.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # Started plainly as `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched a GPU
        # (torch is not even imported yet); the N ranks are fresh child processes, rank 0 prints the JSON line.
        from raytracinginaweekend_amd import sharding
        raise SystemExit(sharding.launch_local_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                                     timeout=args.launch_timeout if args.launch_timeout > 0 else None, log_dir=args.log_dir or None))

    import torch                                   # BEFORE libmirt.so: the loader then binds libmirt to the libamdhip64 torch mapped
    from raytracinginaweekend_amd import capi, sharding
    hip_runtime = capi.hip_runtime_info()          # raises if two HIP runtimes are mapped (torch's pointers and stream go to libmirt)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world                             # the launcher's world size wins
    dev_index = 0 if args.same_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or "RANK" in os.environ:          # one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # librccl prints a version banner on STDOUT when its first communicator comes up; stdout is for the one JSON line, so file
        # descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            warm = torch.zeros(1, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(warm)                    # creates the communicator here, not inside a timed step (--warmup 0)
            dist.reduce(warm, dst=0)
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
        sharding.passed_first_barrier()              # tells a self-started launcher that the process group came up

    label, W0, H0, SPP0, cpu_spp0, bound = WORKLOADS[args.workload]
    W, H, SPP = args.width or W0, args.height or H0, args.spp or SPP0
    sc, cam = make_scene(args.workload, W / H)
    list_walk = sc.n_prims <= 32
    if not list_walk:
        capi.build_bvh(sc)                           # the library's own SAH tree, attached so that the CPU oracle walks the same one
    ctx = capi.Context(dev_index)
    ctx.set_option("engine", args.engine)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    s_begin, s_count = sharding.spp_range(rank, world, SPP)
    flags = capi.FLAG_ACCEL_AUTO
    params = capi.RenderParams(W, H, s_begin, s_count, args.max_depth, flags, args.seed)

    d_sum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    d_rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def reduce_to_root():
        if args.backend == "nccl":
            dist.reduce(d_sum, dst=0, op=dist.ReduceOp.SUM)
        else:                                        # rehearsal path: through host memory
            h = d_sum.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                d_sum.copy_(h)

    def step(stats=None):
        d_sum.zero_()
        ctx.render_device(params, d_sum.data_ptr(), stream, stats)
        if dist is not None:
            if stats is not None:
                ev[0].record()
            reduce_to_root()
            if stats is not None:
                ev[1].record()
        if rank == 0:
            ctx.resolve_device(d_sum.data_ptr(), W, H, SPP, flags, None, d_rgba.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # a few extra, untimed, instrumented steps: per-launch HIP events on the render stream + ray counts; the launch
    # times are averaged over them (a single launch can be off by several per cent)
    ctx.set_option("time_launches", 1)
    n_inst = max(1, min(args.steps, 3))
    st = None
    ext_ms, ext_launches, dev_ms, reduce_ms = 0.0, 0, 0.0, 0.0
    for _ in range(n_inst):
        st = capi.Stats()
        step(st)
        torch.cuda.synchronize(dev)
        ext_ms += st.extend_ms; ext_launches += int(st.extend_launches); dev_ms += st.device_ms
        if dist is not None:
            reduce_ms += float(ev[0].elapsed_time(ev[1]))       # the one exchange of the path, rank 0's view
    reduce_ms /= n_inst
    ctx.set_option("time_launches", 0)
    per_rank = torch.tensor([float(st.rays), float(st.samples), dev_ms / n_inst], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    gathered = [per_rank.clone() for _ in range(world)] if dist is not None else [per_rank]
    if dist is not None:
        dist.all_gather(gathered, per_rank)
    rays_total = sum(float(g[0]) for g in gathered)
    samples_total = sum(float(g[1]) for g in gathered)
    device_ms_per_rank = [round(float(g[2]), 3) for g in gathered]

    if rank == 0:
        samples = float(W) * H * SPP
        ms_per_step = dt / args.steps * 1e3
        msamples = samples / (dt / args.steps) / 1e6
        rays_per_sample = rays_total / samples_total
        out = {
            "metric": f"Msamples/sec (whole node), {label} {W}x{H}" if args.workload != "cornell" or (W, H) != (1024, 1024) else "Msamples/sec (whole node), Cornell box 3-ball 1024^2",
            "value": round(msamples, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label} {W}x{H} @{SPP} spp, depth {args.max_depth}, spp-sharded over {world} GPU(s)",
                       "name": args.workload, "seed": args.seed, "objects": int(sc.n_objects), "scene": SCENE_PARAMS[args.workload],
                       "hip_runtime": hip_runtime, "launch_attempts": int(os.environ.get("MIRT_LAUNCH_ATTEMPT", "1")),
                       "launcher": "self-started ranks (sharding.launch_local_ranks)" if "MIRT_LAUNCH_ATTEMPT" in os.environ else ("torch.distributed.run / external" if "RANK" in os.environ else "single process"),
                       "accel": f"list ({sc.n_prims} primitives through the scalar cache)" if list_walk else "own binned-SAH tree, near-first wave-synchronous walk in resumable slices",
                       "spp_per_gpu": [sharding.spp_range(r, world, SPP)[1] for r in range(world)],
                       "engine": "persistent (k_trace)" if args.engine == 1 else "wavefront (k_bounce per depth)",
                       "collective": (("RCCL reduce (torch.distributed nccl)" if args.backend == "nccl" else "gloo reduce through host memory (rehearsal)") if world > 1 else "none")},
            "mrays_per_s": round(msamples * rays_per_sample, 2), "rays_per_sample": round(rays_per_sample, 4),
        }
        work = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], work, parity = cpu_baseline(sc, cam, W, H, args, ctx, list_walk)
            out["image_parity"] = parity
        out["roofline"], out["roofline_other"] = rooflines(args, st, sc, list_walk, bound, ext_ms, ext_launches, n_inst, work)
        out["whole_render"] = {"device_ms_rank0": round(dev_ms / n_inst, 3), "device_ms_per_rank": device_ms_per_rank,
                               "dominant_kernel_ms": round(ext_ms / n_inst, 3), "batches": int(st.batches), "launches": int(st.bounce_launches),
                               "rccl_reduce_ms": round(reduce_ms, 3), "reduce_bytes": int(W) * H * 3 * 4 if world > 1 else 0}
        if args.ppm:
            capi.write_ppm(args.ppm, d_rgba.cpu().numpy())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def rooflines(args, st, sc, list_walk, bound, ext_ms, ext_launches, n_inst, work):
    """The dominant kernel (k_trace*: one launch per batch, every bounce of every path; rank 0's launches, HIP events on their
    stream) against the VALU and HBM rooflines; `bound` names the binding one."""
    launches = max(1, int(st.extend_launches))                 # per render
    k_s = ext_ms / n_inst * 1e-3                               # kernel time per render, mean of the instrumented renders
    rays, samples = float(st.rays), float(st.samples)
    note = None
    if args.engine != 1:
        kname = "k_bounce<depth>=1>"
        rays -= samples
        hist = list(st.depth_hist)
        written = float(sum(hist[2:]))
        k_flop = rays * FLOP_PER_RAY_CORNELL
        k_bytes = rays * PATH_STATE_BYTES + written * PATH_STATE_BYTES + (rays - written) * RAD_BYTES
        flop_per_ray = FLOP_PER_RAY_CORNELL
    elif args.workload == "cornell":
        kname = "k_trace<true,false>"
        flop_per_ray = FLOP_PER_RAY_CORNELL
        k_flop = rays * flop_per_ray + samples * FLOP_PER_SAMPLE_FIXED
        k_bytes = samples * RAD_BYTES                          # the only per-path HBM traffic: one radiance slot
    else:
        kname = "k_trace<true,false>" if list_walk else ("k_trace_bvh<4,false,false>" if sc.n_objects > 65536 else "k_trace_bvh<3,false,false>")   # the near-first walk in resumable slices (the default from 2 Mi paths per batch); 4: top of the tree from LDS (trees of 65536 nodes and more)
        if work is None:                                       # no oracle counters in this run (--no-cpu-baseline or N > 1): the committed ones
            work = committed_work(args.workload)
            note = "boxes/spheres per ray from profiles/r02_work_counters.json (no CPU sample in this run)"
        boxes, prims = (work or {}).get("boxes_per_ray", 0.0), (work or {}).get("prims_per_ray", float(sc.n_prims) if list_walk else 0.0)
        flop_per_ray = boxes * FLOP_BOX + prims * FLOP_SPHERE + FLOP_FINALISE + FLOP_SCATTER
        k_flop = rays * flop_per_ray + samples * FLOP_PER_SAMPLE_FIXED
        k_bytes = rays * (boxes * BYTES_BOX + prims * BYTES_SPHERE) + samples * RAD_BYTES
    valu_achieved = k_flop / k_s / 1e12 if k_s > 0 else 0.0
    hbm_achieved = k_bytes / k_s / 1e9 if k_s > 0 else 0.0
    traffic = None
    try:    # HBM bytes per launch from the committed PMC passes of this same workload (profiles/)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pm = json.load(f)
        pm = pm.get(args.workload) if "kernel" not in pm else (pm if args.workload == "cornell" else None)   # per-workload entries, or the round-1 flat form
        if pm and pm.get("kernel") == kname and pm.get("paths_per_launch"):
            traffic = round(pm["hbm_bytes_per_launch"] * (samples / launches) / pm["paths_per_launch"], 1)
    except (OSError, ValueError):
        pass
    common = {"kernel": kname, "launches": launches, "avg_launch_ms": round(ext_ms / max(1, ext_launches), 4), "timed_launches": ext_launches}
    if note:
        common["note"] = note
    if work:
        common["boxes_per_ray"] = round(work.get("boxes_per_ray", 0.0), 3); common["prims_per_ray"] = round(work.get("prims_per_ray", 0.0), 3)
    valu = {"bound": "valu", "achieved": round(valu_achieved, 4), "peak": round(VALU_PEAK_TOPS, 1), "unit": "TFLOP/s",
            "frac": round(valu_achieved / VALU_PEAK_TOPS, 5), "traffic": traffic, **common,
            "flop_per_ray": round(flop_per_ray, 1), "flop_per_launch": round(k_flop / launches, 1)}
    hbm = {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "traffic": traffic, **common,
           "bytes_per_launch": round(k_bytes / launches, 1)}
    if not list_walk and args.engine == 1 and work:
        # what the walk really moves, through the CU's L1: one 64-byte node (both child boxes) per two boxes tested and one
        # 32-byte sphere record per sphere tested, per lane; the 256 L1s deliver 64 B per clock each (profiles/r01_bvh_l1.txt)
        l1_bytes = rays * (work["boxes_per_ray"] * 32.0 + work["prims_per_ray"] * 32.0)
        l1_peak = 256 * 64 * 2.4e9 / 1e9
        hbm["l1_view"] = {"bound": "l1", "achieved": round(l1_bytes / k_s / 1e9, 1), "peak": round(l1_peak, 1), "unit": "GB/s",
                          "frac": round(l1_bytes / k_s / 1e9 / l1_peak, 4), "bytes_per_launch": round(l1_bytes / launches, 1)}
        if hbm["frac"] > 1.0:
            hbm["note"] = ("frac > 1: the node and sphere records a ray needs are served by L1/L2/MALL, not by HBM (the whole scene is " +
                           f"{(sc.n_objects * 2 * 64 + sc.n_prims * 64) / 1e6:.0f} MB); HBM is not what binds this walk, see l1_view and the VALU view")
    # Small scenes live in LDS / the scalar cache and the kernel is VALU-issue bound (profiles/): the VALU roofline binds and
    # the HBM view of the same launches is beside it.  The 1 M-sphere scene (80 MB of nodes and spheres) is priced in bytes
    # per ray against HBM as SURVEY 8(d) asks; what actually limits that walk is the CUs' L1 (profiles/r01_bvh_l1.txt).
    return (hbm, valu) if bound == "hbm" else (valu, hbm)


def committed_work(name):
    try:
        with open(os.path.join(ROOT, "profiles", "r02_work_counters.json")) as f:
            return json.load(f).get(name)
    except (OSError, ValueError):
        return None


def cpu_baseline(sc, cam, W, H, args, ctx, list_walk):
    """The CPU oracle (a port: oracle/rt_oracle.c, OpenMP over rows like the reference) timed on a bounded sample of the
    same workload: the full-resolution image at a few spp, Philox draws, same tree, same visiting order.  The same sample
    range is rendered on the GPU and compared bit for bit.  Returns (cpu_baseline, work counters per ray, image_parity)."""
    from tests import orc
    from tests import helpers as Hh
    from raytracinginaweekend_amd import capi
    spp = args.cpu_spp or WORKLOADS[args.workload][4]
    oflag = orc.FLAG_ACCEL_LIST if list_walk else orc.FLAG_ACCEL_ORDERED
    gflag = capi.FLAG_ACCEL_LIST if list_walk else capi.FLAG_ACCEL_ORDERED
    threads = min(orc.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0") or orc.max_threads()))
    t0 = time.perf_counter()
    osum, ost = orc.trace(sc, cam, W, H, spp, args.max_depth, args.seed, oflag, threads=threads)
    dt = time.perf_counter() - t0
    g = ctx.render(W, H, spp, args.max_depth, args.seed, gflag, want=("sum",))
    same = bool(np.array_equal(osum.view(np.uint32), g["sum"].view(np.uint32)))
    work = {"boxes_per_ray": ost.nodes_visited / max(1, ost.rays), "prims_per_ray": ost.prims_tested / max(1, ost.rays),
            "rays_per_sample": ost.rays / float(W * H * spp)}
    base = {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{W}x{H} @{spp} spp of the same scene/seed (Philox draws), {dt:.1f} s",
            "gpu_image_bit_identical_on_sample": same, "gpu_rays_equal_on_sample": bool(g["stats"].rays == ost.rays)}
    if args.workload == "cornell":
        # the reference's own RNG cost (a fresh std::random_device per draw), on a much smaller sample
        t1 = time.perf_counter()
        orc.trace(sc, cam, 128, 128, 1, args.max_depth, args.seed, oflag, rng=orc.RNG_ENTROPY, threads=threads)
        base["with_os_entropy_per_draw_msamples"] = round(128 * 128 / (time.perf_counter() - t1) / 1e6, 4)

    # image parity at equal spp with DIFFERENT random streams (north_star: "within a stated PSNR tolerance, since the RNG
    # streams differ"): PSNR(GPU seed C, oracle seed A/B) >= PSNR(oracle A, oracle B) - 1 dB, on gamma-corrected pixels in [0, 1].
    parity = None
    gold = os.path.join(ROOT, "tests", "golden")
    try:
        side = json.load(open(os.path.join(gold, "f6_converged.json"))).get(args.workload)
    except (OSError, ValueError):
        side = None
    if side:                                                    # committed converged oracle images (two seeds) of this scene
        ref = np.load(os.path.join(gold, "f6_converged.npz"))
        sc2, cam2 = make_scene(args.workload, side["width"] / side["height"])
        ctx.upload_scene(sc2); ctx.set_camera(cam2)
        r = ctx.render(side["width"], side["height"], side["spp"], 50, 303, capi.FLAG_ACCEL_AUTO, want=("pixels",))
        img = np.clip(r["pixels"], 0, 1)
        ps = [Hh.psnr(img, ref[f"{args.workload}_seed{s}"].astype(np.float32)) for s in side["seeds"]]
        parity = {"psnr_db": round(min(ps), 2), "threshold_db": round(side["mutual_psnr_db"] - 1.0, 2), "spp": side["spp"],
                  "image": f"{side['width']}x{side['height']}", "gpu_seed": 303, "oracle_seeds": side["seeds"],
                  "threshold": "PSNR(oracle seed A, oracle seed B) - 1 dB (tests/golden/f6_converged.json)", "pass": bool(min(ps) >= side["mutual_psnr_db"] - 1.0)}
        ctx.upload_scene(sc); ctx.set_camera(cam)
    else:                                                       # no fixture for this scene: calibrate on the bounded sample itself
        osum_b, _ = orc.trace(sc, cam, W, H, spp, args.max_depth, args.seed + 1, oflag, threads=threads)
        gc = ctx.render(W, H, spp, args.max_depth, args.seed + 2, gflag, want=("pixels",))
        pa, _ = orc.resolve(osum, spp); pb, _ = orc.resolve(osum_b, spp)
        pa, pb, pc = np.clip(pa, 0, 1), np.clip(pb, 0, 1), np.clip(gc["pixels"], 0, 1)
        mutual = Hh.psnr(pa, pb)
        ps = [Hh.psnr(pc, pa), Hh.psnr(pc, pb)]
        parity = {"psnr_db": round(min(ps), 2), "threshold_db": round(mutual - 1.0, 2), "spp": spp, "image": f"{W}x{H}",
                  "gpu_seed": args.seed + 2, "oracle_seeds": [args.seed, args.seed + 1],
                  "threshold": "PSNR(oracle seed A, oracle seed B) - 1 dB, measured in this run", "pass": bool(min(ps) >= mutual - 1.0)}
    return base, work, parity


if __name__ == "__main__":
    main()
