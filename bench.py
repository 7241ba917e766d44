#!/usr/bin/env python3
"""Benchmark of the render hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one frame: one call of the library's render entry point over the whole image at the
workload's samples per pixel, with the scene and the accumulation buffer already in HBM. The
default workload is BASELINE.json configs[1] (floor + sphere.obj, 1920x1080, 64 spp, depth cap
200). With N > 1 every rank renders the same frame with its own block of each pixel's sample
stream (weak scaling: per-GPU work fixed) into an fp32 sum buffer, and one RCCL all-reduce per
step combines the buffers. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp, BASELINE.json config it is)
    "sphere_floor_1080p_64spp": ("sphere_floor", 1920, 1080, 64, "configs[1]"),
    "ogre_4k_256spp": ("ogre", 3840, 2160, 256, "configs[2]"),
    "spot_4k_1024spp": ("spot", 3840, 2160, 1024, "configs[3]"),
    "spot_default_10spp": ("spot", 1600, 1200, 10, "configs[0]"),
    "spot_20k_default_10spp": ("spot_20k", 1600, 1200, 10, "configs[0] with the 20000x20000 texture src/main.rs:1273 opens"),
    "spot_4k_256spp": ("spot", 3840, 2160, 256, "the sec/frame figure of BASELINE.json's metric"),
    "spot_grid14_8k_4spp": ("spot_grid:14x14", 7680, 4320, 4, "configs[4] (1 spp-block of it)"),
}

FORM_NAMES = {
    "wave": "wavefront: one launch per bounce over a compacted ray queue",
    "mega": "single persistent kernel with path regeneration",
    "noregen": "single kernel, one thread per path",
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec
L2_PEAK_GBS = 34500.0  # aggregate over the eight XCDs' L2s, measured
FP64_PEAK_TFLOPS = 78.6
FP32_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)  # 0.6 s of frames: the first frames after a pause run a few per cent slower
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sphere_floor_1080p_64spp", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Spot 4K 256 spp frames timed after the main loop")
    ap.add_argument("--cpu-sample-spp", type=int, default=0, help="spp of the CPU baseline sample (0 = auto)")
    ap.add_argument("--split", default="samples", choices=["samples", "tiles", "samples_strong"],
                    help="N > 1: 'samples' = every rank renders the frame with its own samples (weak scaling, the default); "
                         "'tiles' = the frame's 64x64 tiles are dealt round-robin over the ranks (strong scaling); "
                         "'samples_strong' = the frame's samples are divided over the ranks (strong scaling: 1024 spp "
                         "as 512/256/128 per GPU on 2/4/8 GPUs)")
    ap.add_argument("--form", default="wave", choices=["wave", "mega", "noregen"], help="scheduling form (A/B); default wavefront")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import _ffi, parallel, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rt.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the render path has no CPU fallback")
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks then share devices and the reduction goes through the host); the driver uses nccl.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene_name, W, H, spp, cfg = WORKLOADS[args.workload]
    scene = scenes.build(scene_name)
    scene.upload(device)
    info = scene.info()
    stream = torch.cuda.current_stream().cuda_stream
    base_flags = {"wave": 0, "mega": _ffi.RT_FLAG_MEGAKERNEL, "noregen": _ffi.RT_FLAG_NO_REGEN}[args.form]

    strong = world > 1 and args.split in ("tiles", "samples_strong")

    def frame_params(step, flags):
        first = step * spp * (1 if strong else world)  # every frame takes fresh samples of each pixel's stream
        p = rt.render_params(width=W, height=H, spp=spp, first_sample=first, seed=0, flags=flags | base_flags)
        if world == 1:
            return p
        if args.split == "tiles":
            return parallel.shard_tiles(p, rank, world, 64)
        if args.split == "samples_strong":
            return parallel.shard_samples(p, rank, world)
        return parallel.weak_samples(p, rank, world)

    canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda") if world == 1 else None
    # N > 1: two sum buffers, so that the all-reduce of one frame runs (on the collective's own
    # stream) while the next frame renders; a buffer is reused only after its reduction is done
    sums = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)] if world > 1 else None
    in_flight = [None, None]

    debug = bool(os.environ.get("RT_BENCH_DEBUG"))

    def note(msg):
        if debug:
            print(f"[rank {rank} {time.perf_counter():.3f}] {msg}", file=sys.stderr, flush=True)

    # N > 1: the reduction of a frame is queued on a stream of its own, behind a device-side wait for
    # the frame (rt_scene_stream_wait): the frame's overlapped tail is then free to finish on the
    # library's stream while the next frame's first stage already runs on the render stream.
    reduce_stream = torch.cuda.Stream() if world > 1 else None

    def run_step(step, flags):
        note(f"step {step} flags {flags}")
        p = frame_params(step, flags)
        if world == 1:
            scene.render_device(canvas.data_ptr(), p, stream)
        else:
            b = step & 1
            if in_flight[b] is not None:
                in_flight[b].wait()  # the render stream waits for the reduction that last used this buffer
            sums[b].zero_()
            scene.render_sum_f32_device(sums[b].data_ptr(), p, stream)
            note("rendered")
            if os.environ.get("RT_BENCH_SYNC_REDUCE"):  # A/B: reduce each frame before the next starts
                scene.wait()
                parallel.reduce_frame_sum(sums[b], dist)
            else:
                with torch.cuda.stream(reduce_stream):
                    scene.stream_wait(reduce_stream.cuda_stream)
                    in_flight[b] = parallel.reduce_frame_sum(sums[b], dist, async_op=True)
                if backend != "nccl":  # the rehearsal backend reduces through the host and hands back no handle
                    reduce_stream.synchronize()
            note("reduce queued")

    def drain():
        for b in range(2):
            if in_flight[b] is not None:
                in_flight[b].wait()
                in_flight[b] = None
                note(f"drained {b}")
        if reduce_stream is not None:
            reduce_stream.synchronize()

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Work counters (node visits, triangle tests per ray, per kernel) from the counting build of
    # the kernels, on the first timed frame's parameters, outside the timed region.
    run_step(args.warmup, _ffi.RT_FLAG_COUNT_WORK)
    drain()
    torch.cuda.synchronize()
    cst = scene.last_stats()
    per_ray = {
        "nodes": cst.node_visits / max(1, cst.rays),
        "tris": cst.tri_tests / max(1, cst.rays),
        "spheres": cst.sphere_tests / max(1, cst.rays),
        "shaded": cst.shaded_hits / max(1, cst.rays),
        "rays_per_path": cst.rays / max(1, cst.paths),
    }
    counted = {}  # kernel kind -> [rays, node visits, triangle tests]
    for l in scene.last_launches():
        c = counted.setdefault(l.kind, [0, 0, 0])
        c[0] += l.rays; c[1] += l.node_visits; c[2] += l.tri_tests
    if canvas is not None:
        canvas.zero_()

    # Timed region. The frames form a pipelined run (RT_FLAG_PIPELINE): each frame's thin tail of
    # long paths and its sum into the canvas (or sum buffer) overlap the next frame's first stage,
    # as in the reference's display loop where frame follows frame; rt_scene_wait at the end is
    # inside the timed region. With N > 1 every frame's reduction follows its tail on the reduction
    # stream, also behind the next frame.
    pipelined = (args.form == "wave" and not os.environ.get("RT_BENCH_NO_PIPELINE")
                 and not (world > 1 and os.environ.get("RT_BENCH_SYNC_REDUCE")))
    step_flags = _ffi.RT_FLAG_TIMING | (_ffi.RT_FLAG_PIPELINE if pipelined else 0)
    rays_total = 0
    timed = {}  # kernel kind -> [ms, launches, rays]

    def harvest():
        for l in scene.last_launches():
            if l.kind == 1 and l.items == 0 and not l.folded:
                continue  # a bounce launch queued beyond what the frame needed: nothing ran
            c = timed.setdefault(l.kind, [0.0, 0, 0])
            c[0] += l.ms; c[1] += max(1, l.folded); c[2] += l.rays

    # Warm-up frames run exactly as the timed ones (same flags: whatever a timed frame needs — work
    # buffers, the events of its timed launches, the launch plan of its shape — exists afterwards),
    # as a run of their own.
    for w_ in range(args.warmup):
        run_step(w_, step_flags)
    if pipelined:
        scene.wait()
    if canvas is not None:
        canvas.zero_()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        run_step(args.warmup + k, step_flags)
        if not pipelined:
            rays_total += scene.last_stats().rays  # waits for the frame, as the reference's display loop does
            harvest()
    if pipelined:
        scene.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    if pipelined:  # the run's totals
        rays_total = scene.last_stats().rays
        harvest()

    t = torch.tensor([elapsed, float(rays_total)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_all = float(t[1])
    else:
        rays_all = float(rays_total)

    # Which GPUs took part: every rank's device UUID, gathered, and the collective library's version.
    prop0 = torch.cuda.get_device_properties(device)
    my_id = f"{prop0.name} {getattr(prop0, 'uuid', '')} (rank {rank}, device {device})"
    ids = [my_id]
    if world > 1:
        raw = torch.zeros(160, dtype=torch.uint8, device="cuda")
        enc = torch.tensor(list(my_id.encode()[:160]), dtype=torch.uint8, device="cuda")
        raw[: enc.numel()] = enc
        gathered = [torch.zeros_like(raw) for _ in range(world)]
        dist.all_gather(gathered, raw)
        ids = [bytes(g.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for g in gathered]
    try:
        coll = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        coll = None
    ranks_seen = {"devices": ids, "distinct_devices": len({i.split(" (rank")[0] for i in ids}),
                  "collective": None if world == 1 else f"{backend} (RCCL {coll})" if backend == "nccl" else backend}

    # The second half of BASELINE.json's metric, "sec/frame Spot 4K 256spp": frames of that workload
    # timed after the main loop, outside `value` (one GPU, default workload only).
    secondary = None
    if world == 1 and args.workload == "sphere_floor_1080p_64spp" and not args.no_secondary and args.form == "wave":
        s_name, sW, sH, s_spp, _ = WORKLOADS["spot_4k_256spp"]
        spot = scenes.build(s_name)
        spot.upload(device)
        s_canvas = torch.zeros((sH, sW, 3), dtype=torch.float64, device="cuda")
        # the per-ray means of this workload's kernels: one frame of the counting build (it also allocates the work buffers)
        spot.render_device(s_canvas.data_ptr(), rt.render_params(width=sW, height=sH, spp=s_spp, first_sample=0, seed=0,
                                                                 flags=_ffi.RT_FLAG_COUNT_WORK), stream)
        s_cst = spot.last_stats()
        s_per_ray = {"nodes": s_cst.node_visits / max(1, s_cst.rays), "tris": s_cst.tri_tests / max(1, s_cst.rays),
                     "spheres": s_cst.sphere_tests / max(1, s_cst.rays), "shaded": s_cst.shaded_hits / max(1, s_cst.rays),
                     "rays_per_path": s_cst.rays / max(1, s_cst.paths)}
        s_counted = {}
        for l in spot.last_launches():
            c = s_counted.setdefault(l.kind, [0, 0, 0])
            c[0] += l.rays; c[1] += l.node_visits; c[2] += l.tri_tests
        frames = 5
        s_flags = _ffi.RT_FLAG_PIPELINE | _ffi.RT_FLAG_TIMING
        spot.render_device(s_canvas.data_ptr(), rt.render_params(width=sW, height=sH, spp=s_spp, first_sample=s_spp, seed=0, flags=s_flags), stream)
        spot.wait()  # one warm-up frame as the timed ones are run; the timed frames are a run of their own
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(frames):
            spot.render_device(s_canvas.data_ptr(), rt.render_params(width=sW, height=sH, spp=s_spp, first_sample=(2 + k) * s_spp, seed=0,
                                                                     flags=s_flags), stream)
        spot.wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / frames
        s_rays_total = spot.last_stats().rays
        s_timed = {}
        for l in spot.last_launches():
            if l.kind == 1 and l.items == 0 and not l.folded:
                continue
            c = s_timed.setdefault(l.kind, [0.0, 0, 0])
            c[0] += l.ms; c[1] += max(1, l.folded); c[2] += l.rays
        s_roof, s_kernels = roofline_block("spot_4k_256spp", s_timed, s_counted, s_cst, s_per_ray, frames, s_rays_total,
                                           spot.export_grid_bvh() is not None, torch.cuda.get_device_properties(device))
        s_rays = s_rays_total / frames
        secondary = {"metric": "sec/frame Spot 4K 256spp", "value": round(dt, 5), "unit": "s", "higher_is_better": False,
                     "workload": f"spot_4k_256spp: scene {s_name}, {sW}x{sH}, {s_spp} spp, depth cap 200",
                     "frames_timed": frames, "mrays_per_s": round(s_rays / dt / 1e6, 1), "rays_per_frame": int(s_rays),
                     "per_ray": {k: round(v, 3) for k, v in s_per_ray.items()}, "kernels": s_kernels, "roofline": s_roof}
        del spot, s_canvas

    if rank == 0:
        value = rays_all / elapsed / 1e6
        roofline, kernels = roofline_block(args.workload, timed, counted, cst, per_ray, args.steps, rays_total, scene.export_grid_bvh() is not None,
                                           torch.cuda.get_device_properties(device), first_on_grid=first_stage_on_grid(scene))
        library_digest = (_ffi.lib.rt_source_digest() or b"unknown").decode()  # digest of the sources the library was built from
        out = {
            "metric": "Mrays/s (primary+secondary)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "reference scene files (assets/), counter-based sampling seed 0",
            "config": {
                "workload": f"{args.workload} = BASELINE.json {cfg}: scene {scene_name}, {W}x{H}, {spp} spp per GPU per frame, depth cap 200",
                "triangles": int(info.n_triangles),
                "spheres": int(info.n_spheres),
                "bvh_nodes": int(info.bvh_nodes),
                "frames": "pipelined run (tail and canvas sum of a frame overlap the next frame)" if pipelined else "one after the other",
                "sharding": "none" if world == 1 else (
                    f"tiles: 64x64 tiles dealt round-robin over {world} ranks at {spp} spp, disjoint pixels combined by one RCCL all-reduce(sum, fp32, {W * H * 3 * 4 / 1e6:.1f} MB) per frame"
                    if args.split == "tiles" else
                    f"samples: {spp} spp divided over {world} ranks, one RCCL all-reduce(sum, fp32, {W * H * 3 * 4 / 1e6:.1f} MB) per frame"
                    if args.split == "samples_strong" else
                    f"samples: {world} x {spp} spp, one RCCL all-reduce(sum, fp32, {W * H * 3 * 4 / 1e6:.1f} MB) per frame"),
                "form": FORM_NAMES[args.form],
            },
            "sec_per_frame": round(elapsed / args.steps, 5),
            "mrays_per_gpu": round(value / world, 2),
            "rays_per_frame_per_gpu": int(rays_total / max(1, args.steps)),
            "per_ray": {k: round(v, 3) for k, v in per_ray.items()},
            "kernels": kernels,
            "roofline": roofline,
            "library": library_digest,
            "ranks_seen": ranks_seen,
        }
        if secondary is not None:
            out["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported with the one-GPU line only
            out["cpu_baseline"] = cpu_baseline(scene_name, W, H, spp, args.cpu_sample_spp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

KERNEL_NAMES = {0: "k_wave<false,0> (camera + trace + first bounce)", 1: "k_wave<false,1> (shade + trace, hit queue)",
                2: "k_wave<false,2> (tail loop)", 3: "k_paths (single-kernel form)", 4: "k_resolve"}
L2_LINE_BYTES = 128.0  # one L2 request moves at most one 128-byte line
L1_ACCESS_BYTES = 64.0  # one vector-L1 access moves at most 64 bytes


def first_stage_on_grid(scene):
    """Whether the first stage walks the 32-byte copy of the nodes too: the library's rule (trees whose
    binary32 nodes exceed 32 MiB; RT_QUANT_FIRST=0/1 forces the choice) for scenes that have the copy."""
    if scene.export_grid_bvh() is None:
        return False
    forced = os.environ.get("RT_QUANT_FIRST", "")
    if forced:
        return forced[0] != "0"
    return int(scene.info().bvh_nodes) * 64 > (32 << 20)


def roofline_block(workload, timed, counted, cst, per_ray, steps, rays_total, grid_nodes, prop, first_on_grid=False):
    """Roofline of the dominant kernel (largest total time among the timed launches) of a run.

    Algorithmic work per ray, from that kernel's own per-ray means (counting build of the same
    kernel): bytes = node bytes per interior node visited (64; 32 in the launches over the queue when
    the scene qualifies for the half-size grid nodes) + 192 per triangle tested (72 vertices + 72
    vertex normals + 48 texture coordinates: the reference's FlatTriangle and its per-hit lookups)
    + 48 ray in + 40 hit out; fetch-only bytes = what the kernel's loads move: node bytes per visit
    + 80 per triangle test (TriGeom) + 128 per shaded hit (TriShade) + 88. Flops = 24 per node visit
    (binary32 slab tests of two boxes) + 52 per triangle test + 30 per sphere test (binary64).
    Candidate bounds: VALU flops at the vector peaks; measured HBM bytes at the HBM peak;
    algorithmic bytes at the aggregate L2 bandwidth (l2.algorithmic) next to what the L2 and the
    vector L1s were actually asked for (l2.measured, l1.measured: request counters of the
    committed PMC passes). frac is the largest of the candidate fractions."""
    dom = max((k for k in timed if k != 4), key=lambda k: timed[k][0])
    d_ms, d_launches, d_rays = timed[dom]
    c_rays, c_nodes, c_tris = counted.get(dom, [cst.rays, cst.node_visits, cst.tri_tests])
    if c_rays == 0:  # the single-kernel form logs no per-launch ray count when not counting
        c_rays, c_nodes, c_tris = cst.rays, cst.node_visits, cst.tri_tests
    if d_rays == 0:
        d_rays = rays_total
    # the launches over the queue walk the 32-byte copy of the nodes in scenes that qualify for it; the first
    # stage does so only for trees far larger than the caches (first_on_grid); the tail never
    node_bytes = 32.0 if grid_nodes and (dom == 1 or (dom == 0 and first_on_grid)) else 64.0
    bytes_per_ray = node_bytes * c_nodes / c_rays + 192.0 * c_tris / c_rays + 88.0
    fetch_bytes_per_ray = node_bytes * c_nodes / c_rays + 80.0 * c_tris / c_rays + 128.0 * per_ray["shaded"] + 88.0
    f32_flops_per_ray = 24.0 * c_nodes / c_rays
    f64_flops_per_ray = 52.0 * c_tris / c_rays + 30.0 * per_ray["spheres"]
    flops_per_ray = f32_flops_per_ray + f64_flops_per_ray
    launch_ms = d_ms / d_launches
    launch_s = launch_ms * 1e-3
    rays_per_launch = d_rays / d_launches
    kernels = [
        {"kernel": KERNEL_NAMES[k], "launches_per_frame": round(v[1] / steps, 2), "ms_per_frame": round(v[0] / steps, 4),
         "mrays_per_frame": round(v[2] / steps / 1e6, 3)}
        for k, v in sorted(timed.items())
    ]
    resolve = timed.get(4, [0.0, 1, 0])
    # Counter-derived figures of the dominant kernel are replayed from the committed rocprofv3 PMC
    # passes of this same command (profiles/hbm_traffic.json, stamped with the digest of the library
    # they were taken on); durations are measured live.
    traffic = None
    valu_issue = None
    pmc = {}
    counters_note = "none for this workload"
    library_digest = (_ffi_lib().rt_source_digest() or b"unknown").decode()
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            pmc = tj.get(workload, {}).get(KERNEL_NAMES[dom].split(" ")[0], {})
            traffic = pmc.get("bytes_per_launch")
            if traffic is not None:
                taken_on = tj.get("_library_digests", {}).get(workload, "unstamped")
                counters_note = (f"replayed from profiles/hbm_traffic.json, taken on library {taken_on} (git {tj.get('_library_revision', '?')}): "
                                 + ("the library at hand" if taken_on == library_digest else f"STALE, the library at hand is {library_digest}"))
            if pmc.get("valu_wave_insts_per_launch"):
                simds = 4 * prop.multi_processor_count
                clock_ghz = (getattr(prop, "clock_rate", 0) or 0) / 1e6  # kHz -> GHz
                if not (1.0 < clock_ghz <= 2.4):
                    clock_ghz = 2.4  # the MI355X's peak engine clock
                peak = simds * clock_ghz / 4.0
                ach = pmc["valu_wave_insts_per_launch"] / launch_s / 1e9
                valu_issue = {"achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                              "frac": round(ach / peak, 4), "lane_use": pmc.get("valu_lane_use"),
                              "insts_per_ray": round(pmc["valu_wave_insts_per_launch"] * 64.0 / rays_per_launch, 1),
                              "note": "SQ_INSTS_VALU per launch (rocprofv3 PMC pass of this command, profiles/) over the live launch "
                                      "duration, against 4 SIMDs x CUs x clock / 4: one wave-instruction per SIMD every 4 cycles, the rate "
                                      "measured for binary64, 64-bit integer and 32-bit multiply instructions (4.2-4.8 cycles; 32-bit add and "
                                      "logic instructions take 2.6: profiles/r03/valu_rates.txt); lane_use = share of the 64 lanes active per "
                                      "VALU instruction; insts_per_ray = lane-slots issued per ray"}
        except Exception:
            traffic = None
    # (i) VALU: time the algorithmic flops would take at the vector peaks (binary32 slab tests at
    # the FP32 peak, triangle and sphere tests at the FP64 peak) over the launch time
    valu_s = rays_per_launch * (f32_flops_per_ray / (FP32_PEAK_TFLOPS * 1e12) + f64_flops_per_ray / (FP64_PEAK_TFLOPS * 1e12))
    ach_tflops = flops_per_ray * rays_per_launch / launch_s / 1e12
    alg_gbs = bytes_per_ray * rays_per_launch / launch_s / 1e9
    l2 = {"achieved": round(alg_gbs, 2), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / L2_PEAK_GBS, 5),
          "algorithmic": {"bytes_per_ray": round(bytes_per_ray, 1), "gb_per_s": round(alg_gbs, 2), "frac": round(alg_gbs / L2_PEAK_GBS, 5)},
          "fetch_only": {"bytes_per_ray": round(fetch_bytes_per_ray, 1), "gb_per_s": round(fetch_bytes_per_ray * rays_per_launch / launch_s / 1e9, 2),
                         "frac": round(fetch_bytes_per_ray * rays_per_launch / launch_s / 1e9 / L2_PEAK_GBS, 5)},
          "note": "algorithmic bytes of every node visit and triangle test over the launch time, against the aggregate L2 bandwidth; "
                  "fetch_only counts what the kernel's loads move (80-byte triangle geometry per test, 128-byte shading record per hit); "
                  "measured = requests the L2s actually received (TCC_HIT + TCC_MISS) x 128-byte line, an upper bound on the bytes they "
                  "served: the tree and the triangles are mostly served by the CUs' vector L1s (l1_measured = TCP_TOTAL_CACHE_ACCESSES x 64 "
                  "bytes, also an upper bound)"}
    if pmc.get("tcc_requests_per_launch"):
        gbs = pmc["tcc_requests_per_launch"] * L2_LINE_BYTES / launch_s / 1e9
        l2["measured"] = {"requests_per_launch": int(pmc["tcc_requests_per_launch"]), "hit_rate": pmc.get("tcc_hit_rate"),
                          "gb_per_s": round(gbs, 2), "frac": round(gbs / L2_PEAK_GBS, 5)}
    if pmc.get("tcp_accesses_per_launch"):
        gbs = pmc["tcp_accesses_per_launch"] * L1_ACCESS_BYTES / launch_s / 1e9
        l2["l1_measured"] = {"accesses_per_launch": int(pmc["tcp_accesses_per_launch"]), "gb_per_s": round(gbs, 2),
                             "bytes_per_ray": round(pmc["tcp_accesses_per_launch"] * L1_ACCESS_BYTES / rays_per_launch, 1)}
    cand = {
        "valu": {"achieved": round(ach_tflops, 4), "peak": round(ach_tflops * launch_s / valu_s, 2), "unit": "TFLOP/s",
                 "frac": round(valu_s / launch_s, 5),
                 "note": f"{f32_flops_per_ray:.1f} binary32 + {f64_flops_per_ray:.1f} binary64 algorithmic flops per ray; peak = this mix at "
                         f"{FP32_PEAK_TFLOPS} / {FP64_PEAK_TFLOPS} TFLOP/s vector"},
        "l2": l2,
    }
    if traffic is not None:  # (ii) HBM: measured bytes per launch
        cand["hbm"] = {"achieved": round(traffic / launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 5),
                       "note": "measured HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) over the live launch time"}
    bound = max(cand, key=lambda k: cand[k]["frac"])
    roofline = {
        "kernel": KERNEL_NAMES[dom],
        "bound": bound,
        "achieved": cand[bound]["achieved"],
        "peak": cand[bound]["peak"],
        "unit": cand[bound]["unit"],
        "frac": cand[bound]["frac"],
        "traffic": traffic,
        "counters": counters_note,
        "launch_ms": round(launch_ms, 4),
        "bytes_per_ray": round(bytes_per_ray, 1),
        "flops_per_ray": round(flops_per_ray, 1),
        "rays_per_launch": int(rays_per_launch),
        "launches_per_frame": round(d_launches / steps, 2),
        "resolve_ms": round(resolve[0] / max(1, resolve[1]), 4),
        "candidates": cand,
        "valu_issue": valu_issue,
        "note": "frac is the largest of the candidate fractions (VALU flops at the vector peaks, algorithmic bytes at the aggregate L2 "
                "bandwidth, measured HBM bytes at the HBM peak): the bound the kernel is closest to, not the share of the machine in use — the "
                "scene's tree and triangles stay in the CUs' L1s, HBM carries only the sample and queue records, and what the kernel "
                "actually runs out of is vector-instruction issue (valu_issue)",
    }
    return roofline, kernels


def _ffi_lib():
    from raytracer_amd import _ffi

    return _ffi.lib


def cpu_baseline(scene_name, W, H, spp, sample_spp):
    """The CPU restatement of the reference's render -> raytrace -> Scene::hit path (oracle/, with
    the reference's exhaustive-candidate hit semantics), threads over pixels standing in for
    rayon, timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import oracle_ffi as orc
    import raytracer_amd as rt

    threads = min(16, os.cpu_count() or 1)
    o = orc.OracleScene.build(scene_name)
    # calibrate on a 1/16-size frame, then size the sample for about 10-20 s
    p = rt.render_params(width=W // 4, height=H // 4, spp=1, seed=0)
    c = np.zeros((H // 4, W // 4, 3))
    t0 = time.perf_counter()
    rays, _ = o.render(c, p, threads=threads, mode=orc.HIT_REFLIKE)
    rate = rays / (time.perf_counter() - t0)
    if sample_spp <= 0:
        rays_per_spp = rays * 16.0
        sample_spp = int(max(1, min(spp, 15.0 * rate / rays_per_spp)))
    p = rt.render_params(width=W, height=H, spp=sample_spp, seed=0)
    c = np.zeros((H, W, 3))
    t0 = time.perf_counter()
    rays, cands = o.render(c, p, threads=threads, mode=orc.HIT_REFLIKE)
    dt = time.perf_counter() - t0
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{W}x{H}, {sample_spp} of {spp} spp, same scene and seed ({rays} rays in {dt:.1f} s; {cands / max(1, rays):.1f} triangle candidates per ray)",
    }


if __name__ == "__main__":
    main()
