#!/usr/bin/env python3
"""bench.py -- Msamples/s of the Cornell box 3-ball render (BASELINE.json configs[1]) on N MI355X.

One step = one complete TraceRay of the workload: every rank renders its slice of the
samples-per-pixel range into a device-resident float radiance-sum buffer (mirt_render_device), the
buffers are summed onto rank 0 with one RCCL reduce, rank 0 resolves to RGBA8 (mean, sqrt gamma,
clamp, x255).  Scene upload and the PPM write are outside the timed region; nothing crosses PCIe
inside it.  Total work is fixed (1024x1024 @ 1024 spp), so scaling is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1024 --height 1024 --spp 1024]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of the Cornell box, brute force over its 9 primitives
FLOP_PER_RAY = 256.0            # 3 sphere tests + 6 rect tests + closest-hit finalise + mean scatter
FLOP_PER_SAMPLE_FIXED = 38.0    # ray generation + accumulate
# MI355X_MICROARCH.md: 256 CU x 4 SIMD, a wave64 VALU op issues in 2 cycles at 2.4 GHz ->
# 78.6 T lane-ops/s for non-FMA fp32 (157.3 TFLOP/s is the same rate counted as FMAs); contraction
# is off on this path, so one flop = one lane-op.
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
PATH_STATE_BYTES = 44.0         # wavefront engine only: origin, direction, throughput (9 floats) + pixel + sample per queue slot
RAD_BYTES = 16.0                # one float4 per finished path, written once (k_trace / k_bounce), read once (k_accumulate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=128, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--engine", type=int, default=1, help="1: persistent k_trace (default), 0: wavefront k_bounce per depth")
    ap.add_argument("--ppm", default="", help="write the last image as ASCII PPM (outside the timed region)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from raytracinginaweekend_amd import capi, scene, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)                        # creates the RCCL communicator here, not inside a timed step (--warmup 0)
        dist.reduce(warm, dst=0)
        torch.cuda.synchronize(dev)

    W, H, SPP = args.width, args.height, args.spp
    sc, cam = scene.cornell_box(W / H)
    ctx = capi.Context(local_rank)
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

    def step(stats=None):
        d_sum.zero_()
        ctx.render_device(params, d_sum.data_ptr(), stream, stats)
        if dist is not None:
            if stats is not None:
                ev[0].record()
            dist.reduce(d_sum, dst=0, op=dist.ReduceOp.SUM)
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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
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
    rays_local = torch.tensor([float(st.rays), float(st.samples)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(rays_local, op=dist.ReduceOp.SUM)
    rays_total, samples_total = float(rays_local[0].item()), float(rays_local[1].item())

    if rank == 0:
        samples = float(W) * H * SPP
        ms_per_step = dt / args.steps * 1e3
        msamples = samples / (dt / args.steps) / 1e6
        rays_per_sample = rays_total / samples_total
        # dominant kernel: k_trace (engine 1: one launch per batch, every bounce of every path) or k_bounce at
        # depth >= 1 (engine 0); rank 0's launches, HIP events on their stream
        hist = list(st.depth_hist)
        launches = max(1, int(st.extend_launches))                 # per render
        k_s = ext_ms / n_inst * 1e-3                               # k_trace time per render, mean of the instrumented renders
        if args.engine == 1:
            kname = "k_trace<true,false>"
            k_rays = float(st.rays)
            k_flop = k_rays * FLOP_PER_RAY + float(st.samples) * FLOP_PER_SAMPLE_FIXED
            k_bytes = float(st.samples) * RAD_BYTES                       # the only per-path HBM traffic: one radiance slot
        else:
            kname = "k_bounce<depth>=1>"
            k_rays = float(st.rays - st.samples)
            k_flop = k_rays * FLOP_PER_RAY
            written = float(sum(hist[2:]))
            k_bytes = k_rays * PATH_STATE_BYTES + written * PATH_STATE_BYTES + (k_rays - written) * RAD_BYTES
        valu_achieved = k_flop / k_s / 1e12 if k_s > 0 else 0.0
        hbm_achieved = k_bytes / k_s / 1e9 if k_s > 0 else 0.0
        traffic = None
        try:    # HBM bytes per launch from the committed PMC passes of this same workload (profiles/)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f)
            if pm.get("kernel") == kname and pm.get("paths_per_launch"):
                traffic = round(pm["hbm_bytes_per_launch"] * (float(st.samples) / launches) / pm["paths_per_launch"], 1)
        except (OSError, ValueError):
            pass
        common = {"kernel": kname, "launches": launches, "avg_launch_ms": round(ext_ms / max(1, ext_launches), 4), "timed_launches": ext_launches}
        valu = {"bound": "valu", "achieved": round(valu_achieved, 4), "peak": round(VALU_PEAK_TOPS, 1), "unit": "TFLOP/s",
                "frac": round(valu_achieved / VALU_PEAK_TOPS, 5), "traffic": traffic, **common,
                "flop_per_ray": FLOP_PER_RAY, "flop_per_launch": round(k_flop / launches, 1)}
        hbm = {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "traffic": traffic, **common,
               "bytes_per_launch": round(k_bytes / launches, 1)}
        # The kernel is VALU-issue bound (profiles/: VALU instructions x 2 cycles ~ 70 % of all SIMD cycles, HBM traffic
        # a few GB per launch), so the VALU roofline is the binding one; the HBM figure of the same launches is beside it.
        roofline, other = valu, hbm
        out = {
            "metric": "Msamples/sec (whole node), Cornell box 3-ball 1024^2",
            "value": round(msamples, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Cornell box 3-ball {W}x{H} @{SPP} spp, depth {args.max_depth}, spp-sharded over {world} GPU(s)",
                       "seed": args.seed, "accel": "list (9 primitives in LDS)", "spp_per_gpu": s_count,
                       "engine": "persistent (k_trace)" if args.engine == 1 else "wavefront (k_bounce per depth)"},
            "mrays_per_s": round(msamples * rays_per_sample, 2), "rays_per_sample": round(rays_per_sample, 4),
            "roofline": roofline, "roofline_other": other,
            "whole_render": {"device_ms_rank0": round(dev_ms / n_inst, 3), "dominant_kernel_ms": round(ext_ms / n_inst, 3),
                             "batches": int(st.batches), "launches": int(st.bounce_launches),
                             "rccl_reduce_ms": round(reduce_ms, 3), "reduce_bytes": int(W) * H * 3 * 4 if world > 1 else 0},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sc, cam, W, H, args, ctx)
        if args.ppm:
            capi.write_ppm(args.ppm, d_rgba.cpu().numpy())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def cpu_baseline(sc, cam, W, H, args, ctx):
    """The CPU oracle (a port: oracle/rt_oracle.c, OpenMP over rows like the reference) timed on a
    bounded sample of the same workload: the full-resolution image at a few spp, Philox draws.  The
    same sample range is rendered on the GPU and compared bit for bit."""
    from tests import orc
    from raytracinginaweekend_amd import capi
    spp = args.cpu_spp
    threads = min(orc.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "0") or orc.max_threads()))
    t0 = time.perf_counter()
    osum, ost = orc.trace(sc, cam, W, H, spp, args.max_depth, args.seed, orc.FLAG_ACCEL_LIST, threads=threads)
    dt = time.perf_counter() - t0
    g = ctx.render(W, H, spp, args.max_depth, args.seed, capi.FLAG_ACCEL_LIST, want=("sum",))
    same = bool(np.array_equal(np.where(osum == 0, 0, osum).view(np.uint32), np.where(g["sum"] == 0, 0, g["sum"]).view(np.uint32)))
    # the reference's own RNG cost (a fresh std::random_device per draw), on a much smaller sample
    t1 = time.perf_counter()
    _, est = orc.trace(sc, cam, 128, 128, 1, args.max_depth, args.seed, orc.FLAG_ACCEL_LIST, rng=orc.RNG_ENTROPY, threads=threads)
    dte = time.perf_counter() - t1
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{W}x{H} @{spp} spp of the same scene/seed (Philox draws), {dt:.1f} s",
            "gpu_image_bit_identical_on_sample": same,
            "with_os_entropy_per_draw_msamples": round(128 * 128 / dte / 1e6, 4)}


if __name__ == "__main__":
    main()
