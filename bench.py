#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json: "Mrays/sec + extend-kernel HBM GB/s,
sponza 1080p 256spp").

A step = one Renderer::RayTrace() frame (1 sample per pixel) of the sponza-class scene at 1920x1080:
generate, 7 x (extend, shade), connect.  `value` = W*H*steps*N / t / 1e6 — the reference's own
"Mrays/s" definition (primary samples per second, src/renderer.cpp:60-62) summed over all ranks.
Scene, BVH and seeds are resident in HBM before the timed region.  The samples are partitioned twice: across the N ranks
(magr_ray_tracer_amd/dist.py: one process per GPU, torch.distributed / RCCL, one all_reduce(SUM) of the accumulator inside the
timed region) and, inside a GPU, across the `--lanes` sample streams of the LIBRARY's group handle (rt_group_*, include/rt355.h:
what stands behind one Renderer::Tick), whose frames are queued interleaved so that the tails of one lane's launches are filled
by the others' kernels; a rank's `steps` frames are dealt to its lanes round-robin.

Launching: `python bench.py --gpus N` starts its own N ranks (fresh child processes, spawned before this process touches the
GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
`--total-steps T` renders a FIXED T-spp image split over the ranks ("scaling": "strong"; north star: 256 / 1024 spp over 8 GPUs).

Timed region: exactly `--steps` frames between barrier + synchronize brackets, MAX over ranks.  A region shorter than 0.25 s says
little (20 steps are 41 ms), so the bracketed K-step region is repeated until 0.25 s of it have been timed; `value` and
`ms_per_step` are over all repeats (`repeats` in the line).

Roofline: the traversal kernels fetch ~21 MB of node and triangle records over and over, from the
vector L1s, the L2s and the Infinity Cache - so the line prices the extend kernel at EVERY memory level it uses: bytes per launch
from committed rocprofv3 PMC passes (profiles/r03_traffic.json, made by tools/make_traffic.py from profiles/r03_pmc_*.csv) over the
kernel's own launch time (ONE context with the GPU to itself, kernel-attached HIP events), against that level's peak
(/opt/skills/guides/MI355X_MICROARCH.md).  `bound` is the level with the largest fraction; every fraction is <= 1 by construction.
SURVEY 8(d)'s algorithmic bytes stay in the line as a throughput figure (`roofline.algorithmic`), not as a fraction of HBM.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks of the MI355X memory levels, /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; 6.3 measured); L2 34.5 TB/s aggregate;
# vector L1 64 B per clock and CU = 256 CUs x 64 B x 2.4 GHz (the table's max clock: the chip clocks lower under load, so the fraction
# is a lower estimate)
PEAK_GBS = {"hbm": 8000.0, "l2": 34500.0, "vl1d": 256 * 64 * 2.4}
MIN_TIMED_S = 0.25      # repeat the K-step region until this much has been timed

# The BASELINE.json configurations that fit one GPU (config 1 is the CPU plumbing case).  `python bench.py` = config 3, the one the
# metric is quoted on; the others are parity-test cases that can be timed / profiled with --config (tools/r3_pmc.sh).
CONFIGS = {
    2: dict(label="bunny-class closed mesh, Kajiya (SHADING_SIMPLE)", scene=lambda a: __import__("magr_ray_tracer_amd.scenes", fromlist=["x"]).bunny_class(187),
            W=1280, H=720, spp=64, accel="bvh2", shading=0, lanes=8),
    3: dict(label="sponza-class procedural atrium", scene=lambda a: __import__("magr_ray_tracer_amd.scenes", fromlist=["x"]).sponza_class(a.detail),
            W=1920, H=1080, spp=256, accel="bvh2", shading=1),
    4: dict(label="sponza-class procedural atrium through the QBVH", scene=lambda a: __import__("magr_ray_tracer_amd.scenes", fromlist=["x"]).sponza_class(a.detail),
            W=1920, H=1080, spp=1024, accel="bvh4", shading=1),
    5: dict(label="robo-orb + terrarium_bot, two BLAS under a TLAS, SBVH alpha 0", scene=lambda a: __import__("magr_ray_tracer_amd.scenes", fromlist=["x"]).config5_scene(0.0),
            W=3840, H=2160, spp=4096, accel="bvh2", shading=1, lanes=8),
}


def extend_bytes(c, accel, prefix="extend"):
    """Algorithmic bytes of SURVEY.md §8(d): R*48 + V_tlas*96 + L_inst*68 + V_int*96 (BVH4: V4*160) + T_prim*52."""
    node = 160 if accel == 1 else 96
    return (c[prefix + "_rays"] * 48 + c[prefix + "_tlas_visits"] * 96 + c[prefix + "_inst_visits"] * 68 +
            c[prefix + "_node_visits"] * node + c[prefix + "_prim_tests"] * 52)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="frames (1 spp each) per GPU in the timed region; 256 = BASELINE config 3")
    ap.add_argument("--total-steps", type=int, default=0, help="render a fixed image of this many spp split over the ranks (strong scaling) instead of --steps per GPU")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (3 = the one the metric is quoted on)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--detail", type=float, default=1.0, help="sponza-class tessellation (1.0 = ~262k triangles)")
    ap.add_argument("--model", default=None, help="an OBJ file (e.g. a real sponza.obj) instead of the procedural atrium: scenes.model_scene")
    ap.add_argument("--view", default=None, help="camera for --model: ox,oy,oz,fx,fy,fz[,fov] (default: the reference's CameraManager defaults)")
    ap.add_argument("--accel", choices=["bvh2", "bvh4"], default=None)
    ap.add_argument("--shard", choices=["samples", "bands", "ibands"], default="samples")
    ap.add_argument("--band-rows", type=int, default=0, help="ibands: rows per band (0 = a quarter of a rank's contiguous share)")
    ap.add_argument("--persist-blocks", type=int, default=0, help="workgroups per CU of the persistent traversal grids of lanes that share the GPU (0 = the library's choice, 2)")
    ap.add_argument("--lanes", type=int, default=0, help="sample streams per GPU (and per row band) whose frames overlap; 0 = 4 (configs 2 and 5, whose late launches are nearly empty: 8; ibands: 1), 1 = the reference's single in-order queue")
    ap.add_argument("--extend-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stage launches with HIP events")
    ap.add_argument("--profile-lanes", action="store_true", help="bracket the extend launches of the first lane in the timed region too (diagnostic: lane0_launch_ms_while_sharing)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (default); gloo only for rehearsals")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses GPU 0 (requires --backend gloo)")
    ap.add_argument("--no-single", action="store_true", help="skip the untimed single-context pass (profiler runs: only the timed workload's launches)")
    ap.add_argument("--no-repeat", action="store_true", help="time the K-step region once, however short it is")
    ap.add_argument("--dump-accum", default=None, help="rank 0 writes the reduced accumulator (npy) here after the timed region")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (this process has not touched the GPU and does
    not exec), pass their output through, wait for all of them; a failing rank ends the others and makes the exit code non-zero."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:        # the others would wait for it in a collective forever
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc if rc >= 0 else 1


def load_traffic(match):
    """PMC-derived bytes per launch / per step for this configuration, if a committed measurement matches it exactly."""
    path = os.path.join(ROOT, "profiles", "r03_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    for e in tj.get("entries", []):
        if all(e.get("config", {}).get(k) == v for k, v in match.items()):
            return e
    return None


# wave64 vector instructions per ns the chip issues when it does nothing else - MEASURED (tools/pmc_calib.hip calib_valu_mix: the slab test's
# own op mix, 8 waves per SIMD; profiles/r03_pmc_calibration*.csv): 0.43 per clock and SIMD at the 2.04 GHz it holds under that load.
# On paper 1,024 SIMDs x 2.4 GHz / 2 clocks = 1,229.
PEAK_VALU_GINST = 928.0


def issue_entry(insts, seconds):
    """vector-issue roofline: wave-level VALU instructions (PMC: SQ_INSTS_VALU, unit pinned by tools/pmc_calib.hip calib_valu) over a time."""
    if not insts or seconds <= 0:
        return None
    g = insts / seconds / 1e9
    return {"valu_wave_insts": int(insts), "ginst_per_s": round(g, 1), "peak": PEAK_VALU_GINST, "peak_source": "measured: tools/pmc_calib.hip calib_valu_mix",
            "frac": round(g / PEAK_VALU_GINST, 4)}


def level_table(bytes_by_level, seconds):
    """{level: {bytes, gbs, peak, frac}} and the bound (the level with the largest fraction) for `bytes` moved in `seconds`."""
    lv = {}
    for k in ("hbm", "l2", "vl1d"):
        b = bytes_by_level.get(k)
        if b is None or seconds <= 0:
            continue
        gbs = b / seconds / 1e9
        lv[k] = {"bytes": int(b), "gbs": round(gbs, 1), "peak": PEAK_GBS[k], "frac": round(gbs / PEAK_GBS[k], 4)}
    bound = max(lv, key=lambda k: lv[k]["frac"]) if lv else None
    return lv, bound


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.same_device and args.gpus > 1 and args.backend != "gloo":
        raise SystemExit("--same-device puts every rank on GPU 0, which RCCL refuses: use --backend gloo")

    import numpy as np
    # Load order decides which HIP runtime the process runs on: torch first = the ROCm 7.0 runtime bundled in the torch wheel, librt355.so
    # first = the system's ROCm 7.2 (the library's RUNPATH).  A single context is 12 % faster on the former (732 against 645 M samples/s,
    # EXPERIMENTS.md (44)); RT355_IMPORT_ORDER=lib-first selects the latter for A/B runs.  Either way librt355.so is loaded before HIP
    # INITIALISES (importing torch does not initialise it), which is when its request for sixteen hardware queues has to be in place.
    from magr_ray_tracer_amd import _lib
    if os.environ.get("RT355_IMPORT_ORDER") != "lib-first":
        import torch
    _lib.device_lib()
    import torch
    if os.environ.get("RT355_IMPORT_ORDER"):
        print("libamdhip64 in use:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}), file=sys.stderr)
    from magr_ray_tracer_amd import dist as rdist, scenes
    from magr_ray_tracer_amd.renderer import Device, Group

    rank, world, local = rdist.init_process_group(args.backend if args.gpus > 1 else None, set_device=not args.same_device)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    conf = CONFIGS[args.config]
    W, H = args.width or conf["W"], args.height or conf["H"]
    accel_name = args.accel or conf["accel"]
    accel = 1 if accel_name == "bvh4" else 0
    shading = conf["shading"]

    # ---- scene + BVH on the host (not timed), upload, seeds -------------------------------------------------
    t0 = time.time()
    if args.model:
        s, view = scenes.model_scene(args.model)
        if args.view:
            v = [float(x) for x in args.view.split(",")]
            view.update(origin=tuple(v[0:3]), forward=tuple(v[3:6]))
            if len(v) > 6:
                view["fov"] = v[6]
    else:
        s, view = conf["scene"](args)
    sa = s.arrays(bvh4=bool(accel))
    build_s = time.time() - t0
    lanes = args.lanes if args.lanes > 0 else (1 if args.shard == "ibands" else conf.get("lanes", 4))
    cam = scenes.camera_for(view, W, H)
    reduced = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")

    # one group (= `lanes` sample streams behind one handle) per row band this rank owns.  The timed region runs WITHOUT HIP-event brackets
    # when lanes share the GPU (they cost 1.2 % of the frame rate and time nothing usable: a lane's launch takes longer while other lanes'
    # kernels run beside it); --profile-lanes brackets the extend launches of the first lane for that diagnostic.  The roofline's kernel
    # time comes from the single-context pass below.
    made = []
    for p in rdist.plans(args.shard, W, H, rank, world, 0, lanes, args.band_rows or None):
        g = Group(W, H, lanes=lanes, y0=p["y0"], y1=p["y1"], accel=accel, shading=shading, device=local,
                  profile=1 if args.profile_lanes and not args.no_profile and not made else 0, extend_variant=args.extend_variant, persist_blocks_per_cu=args.persist_blocks)
        if made:
            g.share_scene(made[0])          # one device copy of the scene for all contexts of this rank
        else:
            g.upload(sa)
        g.seed(rank * lanes if args.shard == "samples" else 0)
        made.append(g)
    groups = rdist.Groups(made)
    dev = groups.devs[0]
    cam["focalLength"] = dev.focus(W // 2, H // 2, cam)
    strong = args.total_steps > 0 or args.shard != "samples"
    # frames this rank renders in one pass of the timed region: weak scaling (default) - `steps` per GPU; a fixed `total_steps`-spp
    # image - its share of the samples (sample plan) or all of them for its rows (band plans)
    if args.total_steps > 0:
        my_steps = rdist.rank_frames(args.total_steps, rank, world) if args.shard == "samples" else args.total_steps
        job_frames = args.total_steps
    else:
        my_steps = args.steps
        job_frames = args.steps * (world if args.shard == "samples" else 1)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def reduce_all():
        groups.sum_into(reduced)              # the rank's accumulator = its lanes added up in lane order, on the device ...
        groups.synchronize()
        rdist.reduce_accumulator(reduced)     # ... then the one exchange step across ranks
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local}" if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    # ---- warmup ------------------------------------------------------------------------------------------
    if args.warmup > 0:
        groups.render(cam, max(args.warmup, 1) * (lanes if args.shard == "samples" else 1))
        groups.synchronize()
        reduce_all()
    groups.reset()
    groups.synchronize()
    for d in groups.devs:
        d.reset_counters()
        d.reset_stage_times()
    barrier()

    # ---- timed region: this rank's frames (dealt to its lanes) + the accumulator reduction, repeated -----------------------------
    dts, render_s, reduce_s = [], 0.0, 0.0
    while True:
        barrier()
        t0 = time.perf_counter()
        if my_steps > 0:
            groups.render(cam, my_steps)
        groups.synchronize()
        t1 = time.perf_counter()
        reduce_all()
        t2 = time.perf_counter()
        barrier()
        dts.append(max_over_ranks(time.perf_counter() - t0))
        render_s += t1 - t0
        reduce_s += t2 - t1
        if args.no_repeat or sum(dts) >= MIN_TIMED_S or len(dts) >= 256:
            break
    dt, repeats = sum(dts), len(dts)
    frames_job = job_frames * repeats            # 1-spp frames of the whole job in the timed region
    frames_mine = my_steps * repeats
    checksum = float(reduced[..., :3].sum().item())
    if args.dump_accum and rank == 0:
        np.save(args.dump_accum, reduced.cpu().numpy())
    per_rank = [[render_s, reduce_s]]
    if world > 1:
        mine = torch.tensor([render_s, reduce_s], dtype=torch.float64, device=f"cuda:{local}" if args.backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [[float(x[0]), float(x[1])] for x in allr]

    ctr = {}
    for d in groups.devs:          # work totals over this rank's contexts
        for k, v in d.counters().items():
            ctr[k] = ctr.get(k, 0) + v
    st_lane0 = dev.stage_times()   # lane 0's extend launches while the other lanes share the GPU (diagnostic only, never a roofline time)
    concurrency = made[0].concurrency()

    # ---- untimed: ONE context with the GPU to itself (the reference's own shape: one Renderer, one in-order queue) ---------------
    single, stage_tab, con, dctr, st2 = {}, {}, {}, None, None
    value_single = None
    if args.shard == "samples" and not args.no_single:
        barrier()
        solo = Device(W, H, accel=accel, shading=shading, device=local, profile=0 if args.no_profile else 1, extend_variant=args.extend_variant)
        solo.share_scene(dev)
        solo.bind_accum(torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}"))
        seeds = np.zeros(W * H, np.uint32)
        _seed_stream(seeds, rdist.plan("samples", W, H, rank, world)["seed_first"])
        solo.set_seeds(seeds)
        solo.render(cam, 2)
        solo.synchronize()
        n1 = max(16, min(args.steps, 64))
        t0 = time.perf_counter()
        solo.render(cam, n1)
        solo.synchronize()
        value_single = W * H * n1 / max_over_ranks(time.perf_counter() - t0) / 1e6 * world
        if not args.no_profile and rank == 0:   # 16 more frames of that context with every stage bracketed (kernels undisturbed)
            solo.set_profile(2)
            solo.reset_stage_times()
            c0 = solo.counters()
            solo.render(cam, 16)
            solo.synchronize()
            st2, c1 = solo.stage_times(), solo.counters()
            stage_tab = {k[:-3]: round(st2[k] / 16, 4) for k in st2 if k.endswith("_ms") and k != "compact_ms"}
            dctr = {k: c1[k] - c0[k] for k in c1}
            # lane utilisation of the event loop's two paths: counted by the instantiation that also records per-ray `steps` (4 more frames)
            solo.enable_steps(True)
            solo.render(cam, 4)
            solo.synchronize()
            c2 = solo.counters()
            solo.enable_steps(False)
            for k in c2:
                if "issues" in k or "loop" in k:
                    dctr[k] = c2[k] - c1[k]
        kernel_name = solo.extend_kernel_name()
        solo.close()
        barrier()
    else:
        kernel_name = dev.extend_kernel_name()

    samples = W * H * frames_job
    value = samples / dt / 1e6

    steps_timed = (args.total_steps if args.total_steps > 0 else args.steps) * repeats
    if rank == 0:
        ms_per_step = dt / max(frames_mine, 1) * 1e3             # per frame THIS GPU rendered (the job-level roofline's time base)
        roof = {"bound": None, "kernel": "extend = " + kernel_name, "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None}
        base_match = {"config": args.config, "accel": accel_name, "detail": args.detail, "width": W, "height": H, "model": bool(args.model)}
        if dctr is not None and st2["extend_launches"] > 0:
            e_launches = st2["extend_launches"]
            e_s = st2["extend_ms"] * 1e-3 / e_launches          # the kernel's own average launch time: one context, nothing else on the GPU
            alg = extend_bytes(dctr, accel) / e_launches
            tr1 = load_traffic(dict(base_match, lanes=1))
            lv, bound = ({}, None)
            if tr1:
                lv, bound = level_table(tr1["extend_bytes_per_launch"], e_s)
            roof.update({
                "timed": "ONE context with the GPU to itself: the %d extend launches of 16 frames, kernel-attached HIP events (hipExtLaunchKernelGGL); "
                         "7 x avg_launch_ms <= the single-context frame time by construction; rocprofv3 --kernel-trace --stats of the same context: "
                         "profiles/r03_kernel_stats_lanes1.csv ((6 x k_trace_persist<false, false, false> + k_trace_persist<false, true, false>) / 7)" % e_launches,
                "avg_launch_ms": round(e_s * 1e3, 5), "launches": e_launches,
                "levels": lv, "bound": bound,
                "achieved": lv[bound]["gbs"] if bound else None, "peak": lv[bound]["peak"] if bound else None,
                "frac": lv[bound]["frac"] if bound else None,
                "traffic": lv["hbm"]["bytes"] if "hbm" in lv else None,
                "traffic_source": (tr1 or {}).get("source", "no committed PMC measurement matches this configuration (profiles/r03_traffic.json)"),
                # what bounds the kernel when no memory level is near its peak: a dependent chain of record fetches per ray (PMC, same passes)
                "limiter": dict({"what": "latency of a dependent fetch chain x lane divergence, not bandwidth at any level: waves of the extend launches are "
                                         "parked in s_waitcnt for `wait_any` of their lifetime, `valu_lane_utilisation` of the lanes of an issued vector "
                                         "instruction hold a ray in that state, the vector L1s are busy (any request in flight) for `vl1d_busy` of the launch"},
                                **({"wait_any": tr1["extend_wave_time"]["wait_any"], "valu_lane_utilisation": tr1["extend_wave_time"]["valu_lane_utilisation"],
                                    "vl1d_busy": tr1.get("extend_vl1d_busy")} if tr1 and tr1.get("extend_wave_time") else {})),
                # the same launches against the vector ALUs' issue rate (not a memory level: reported beside `levels`, `bound` stays a memory level)
                "valu_issue": issue_entry((tr1 or {}).get("extend_valu_insts_per_launch"), e_s),
                "algorithmic": {"bytes_per_launch": int(alg), "gbs": round(alg / e_s / 1e9, 1),
                                "note": "SURVEY 8(d): R*48 + V_int*96 (+ TLAS / instance terms; BVH4: V4*160) + T_prim*52 from the device work counters, over the same "
                                        "launch time: a throughput, NOT a fraction of HBM peak - the records are re-fetched from the vector L1s and L2s "
                                        "(levels.*), so it may exceed the HBM peak"},
                # lane utilisation of the event loop's two code paths (device counters): the share of a wave's 64 lanes that had an event of
                # the kind when that path was issued
                "lane_utilisation": {"node_path": round(dctr["extend_loop_node_events"] / max(64 * dctr["extend_node_issues"], 1), 3),
                                     "triangle_path": round(dctr["extend_loop_leaf_events"] / max(64 * dctr["extend_leaf_issues"], 1), 3),
                                     "node_issues_share": round(dctr["extend_node_issues"] / max(dctr["extend_node_issues"] + dctr["extend_leaf_issues"], 1), 3)},
                "per_ray": {"node_visits": round(dctr["extend_node_visits"] / max(dctr["extend_rays"], 1), 2),
                            "prim_tests": round(dctr["extend_prim_tests"] / max(dctr["extend_rays"], 1), 2),
                            "tlas_visits": round(dctr["extend_tlas_visits"] / max(dctr["extend_rays"], 1), 3)},
            })
            if st2["connect_launches"] > 0:
                c_s = st2["connect_ms"] * 1e-3 / st2["connect_launches"]
                c_alg = extend_bytes(dctr, accel, "connect") / st2["connect_launches"]
                clv, cbound = level_table(tr1["connect_bytes_per_launch"], c_s) if tr1 and tr1.get("connect_bytes_per_launch") else ({}, None)
                con = {"avg_launch_ms": round(c_s * 1e3, 5), "levels": clv, "bound": cbound, "frac": clv[cbound]["frac"] if cbound else None,
                       "algorithmic_gbs": round(c_alg / c_s / 1e9, 1), "rays_per_launch": dctr["connect_rays"] // st2["connect_launches"],
                       "note": "connect is an any-hit traversal with its own visit order; its node / triangle counters are its own work"}
        # job level: every byte the PMC passes saw per frame (all kernels, all lanes) over the frame time of the timed region
        trj = load_traffic(dict(base_match, lanes=lanes)) if args.shard == "samples" else None
        job = {"ms_per_step": round(ms_per_step, 4), "lanes": lanes}
        if trj:
            jlv, jbound = level_table(trj["frame_bytes"], ms_per_step * 1e-3)
            job.update({"levels": jlv, "bound": jbound, "frac": jlv[jbound]["frac"] if jbound else None, "traffic_source": trj.get("source", ""),
                        "note": "bytes of ALL kernels of a frame (PMC passes of this configuration with %d lane(s)) over ms_per_step" % lanes})
            job["valu_issue"] = issue_entry(trj.get("frame_valu_insts"), ms_per_step * 1e-3)
        job["algorithmic_gbs"] = round((extend_bytes(ctr, accel) + extend_bytes(ctr, accel, "connect")) / max(frames_mine, 1) / (ms_per_step * 1e-3) / 1e9, 1)
        roof["job"] = job
        if st_lane0["extend_launches"] > 0 and lanes > 1:
            roof["lane0_launch_ms_while_sharing"] = round(st_lane0["extend_ms"] / st_lane0["extend_launches"], 5)   # diagnostic: NOT a kernel time (other lanes' kernels run beside it)
        traced = ctr["extend_rays"] + ctr["connect_rays"]
        out = {
            "metric": "Mrays/sec + extend-kernel HBM GB/s, sponza 1080p 256spp",
            "value": round(value, 3), "unit": "Mrays/s (primary samples/s, reference definition renderer.cpp:60-62)",
            "n_gpus": world, "steps": args.total_steps if args.total_steps > 0 else args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(steps_timed, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"model {os.path.basename(args.model)}" if args.model else f"BASELINE config {args.config}: " + conf["label"]) +
                                   f" ({len(sa.prims)} prims, {accel_name.upper()}, {len(sa.bvh2)} BVH2 nodes, {len(sa.blas)} BLAS) {W}x{H}, "
                                   f"{'NEE' if shading else 'Kajiya'}+cosine+RR+firefly, 7 bounces; step = 1 spp frame, "
                                   + (f"a fixed {args.total_steps}-spp image split over the ranks" if args.total_steps > 0 else f"{args.steps} spp timed per GPU (config {args.config}: {conf['spp']} spp)")
                                   + f", rendered as {lanes} interleaved sample stream(s) per GPU behind one rt_group handle",
                       "baseline_config": args.config, "shard": args.shard, "lanes": lanes, "contexts": len(groups), "streams_concurrent": concurrency,
                       "triangles": int(len(sa.prims)), "extend_variant": args.extend_variant},
            "repeats": repeats, "timed_s": round(dt, 4),
            "value_single_context": round(value_single, 3) if value_single else None,
            "rank_s": {"render": [round(x[0], 4) for x in per_rank], "all_reduce": [round(x[1], 5) for x in per_rank]},
            "traced_mrays_per_s": round(traced * (world if args.shard == "samples" else 1) / dt / 1e6, 2),
            "rays_per_step": {"extend": ctr["extend_rays"] // max(frames_mine, 1), "connect": ctr["connect_rays"] // max(frames_mine, 1)},
            "roofline": roof, "connect_roofline": con, "stage_ms_per_step": stage_tab,
            "host_build_s": round(build_s, 2), "accum_rgb_sum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sa, cam, W, H, accel, shading)
        print(json.dumps(out), flush=True)
    groups.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def _seed_stream(out, first):
    """seeds[i] = (first+i+1)-th xorshift32 output from 0x12345678 (reference renderer.cpp:195-196), computed by the
    host library's C loop (rth_seed_stream)."""
    import ctypes as C
    from magr_ray_tracer_amd import _lib
    if _lib.host_lib().rth_seed_stream(out.ctypes.data_as(C.c_void_p), int(first), int(out.size)) != 0:
        raise RuntimeError("rth_seed_stream failed")


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box exposes all
    logical CPUs but grants a share; oversubscribing it would make the baseline look slower than the hardware is)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sa, cam, W, H, accel, shading=1):
    """The oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores on a bounded
    sample of the same workload: full-resolution frames, row-band parallel over all cores."""
    from oracle.oracle_py import Oracle
    cores = usable_cores()
    o = Oracle(sa, W, H, accel=accel, shading=shading)
    t0 = time.perf_counter()
    o.render(cam, 1, threads=cores)
    t1 = time.perf_counter() - t0
    frames = int(max(1, min(16, round(15.0 / max(t1, 1e-3)))))
    t0 = time.perf_counter()
    _, _, e, c = o.render(cam, frames, threads=cores)
    t = time.perf_counter() - t0
    # CPU-B of BASELINE.md §3: the upstream template's trace loop shape (scan-line parallel, one primary ray per pixel, nearest
    # hit through the BVH, normal visualisation), reported with the reference's own W*H*fps formula
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        o.trace_normals(cam, threads=cores)
    tb = (time.perf_counter() - t0) / reps
    return {"value": round(W * H * frames / t / 1e6, 4), "unit": "Mrays/s (primary samples/s)", "cores": cores, "kind": "port",
            "sample": f"{frames} full {W}x{H} frames (1 spp each) of the same scene, {cores} row bands in parallel, {t:.1f} s",
            "traced_mrays_per_s": round((e["rays"] + c["rays"]) / t / 1e6, 3),
            "template_trace_mrays_per_s": round(W * H / tb / 1e6, 3)}


if __name__ == "__main__":
    main()
