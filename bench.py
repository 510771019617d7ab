#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json: "Mrays/sec + extend-kernel HBM GB/s,
sponza 1080p 256spp").

A step = one Renderer::RayTrace() frame (1 sample per pixel) of the sponza-class scene at 1920x1080:
generate, 7 x (extend, shade), connect.  `value` = W*H*steps*N / t / 1e6 — the reference's own
"Mrays/s" definition (primary samples per second, src/renderer.cpp:60-62) summed over all ranks.
Scene, BVH and seeds are resident in HBM before the timed region.  The samples are partitioned twice (magr_ray_tracer_amd/dist.py):
across the N ranks (one process per GPU, torch.distributed / RCCL, one all_reduce(SUM) of the accumulator inside the timed
region) and, inside a GPU, across `--lanes` independent contexts whose frames are interleaved so that the tails of one context's
launches are filled by the other's kernels; a rank's `steps` frames are shared out over its lanes.

Launching: `python bench.py --gpus N` starts its own N ranks (fresh child processes, spawned before this process touches the
GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Timed region: exactly `--steps` frames between barrier + synchronize brackets, MAX over ranks.  A region shorter than 0.25 s says
little (20 steps are 57 ms), so the bracketed K-step region is repeated until 0.25 s of it have been timed; `value` and
`ms_per_step` are over all repeats (`repeats` in the line).
"""
import argparse
import json
import os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included): lanes whose streams share a
# hardware queue are serialised (four lanes on four queues: 860 M samples/s; on eight: 1,018).  Has to be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MIN_TIMED_S = 0.25      # repeat the K-step region until this much has been timed
# tools/gather_probe.hip on the MI355X (profiles/r02_gather_probe.log): dependent random 64-byte record fetches + two slab tests,
# no divergence.  By table size: <= 4 MB 200 G records/s (L1/L2 resident), 8 MB 168, 32 MB 118 (Infinity Cache).  The traversal's
# 21 MB of records are fetched with 86 % vector-L1 and 82 % L2 hits (profiles/r01_pmc_final_summary.csv), i.e. its EFFECTIVE
# working set is the small-table regime: the honest ceiling for the access pattern is the 200 G/s row.
GATHER_CEILING_RECORDS_PER_S = 200e9


def extend_bytes(c, accel, prefix="extend"):
    """Algorithmic bytes of SURVEY.md §8(d): R*48 + V_tlas*96 + L_inst*68 + V_int*96 (BVH4: V4*160) + T_prim*52."""
    node = 160 if accel == 1 else 96
    return (c[prefix + "_rays"] * 48 + c[prefix + "_tlas_visits"] * 96 + c[prefix + "_inst_visits"] * 68 +
            c[prefix + "_node_visits"] * node + c[prefix + "_prim_tests"] * 52)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="frames (1 spp each) in the timed region; 256 = BASELINE config 3")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0, help="sponza-class tessellation (1.0 = ~262k triangles)")
    ap.add_argument("--model", default=None, help="an OBJ file (e.g. a real sponza.obj) instead of the procedural atrium: scenes.model_scene")
    ap.add_argument("--view", default=None, help="camera for --model: ox,oy,oz,fx,fy,fz[,fov] (default: the reference's CameraManager defaults)")
    ap.add_argument("--accel", choices=["bvh2", "bvh4"], default="bvh2")
    ap.add_argument("--shard", choices=["samples", "bands", "ibands"], default="samples")
    ap.add_argument("--band-rows", type=int, default=0, help="ibands: rows per band (0 = a quarter of a rank's contiguous share)")
    ap.add_argument("--persist-blocks", type=int, default=2, help="workgroups per CU of the persistent traversal grids of contexts that share the GPU")
    ap.add_argument("--lanes", type=int, default=4, help="independent sample streams per GPU whose frames overlap (samples plan only; 1 = one context)")
    ap.add_argument("--extend-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stage launches with HIP events")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (default); gloo only for rehearsals")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses GPU 0 (requires --backend gloo)")
    ap.add_argument("--no-share-scene", action="store_true", help="every context uploads its own copy of the scene (A/B of rt_share_scene)")
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


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.same_device and args.gpus > 1 and args.backend != "gloo":
        raise SystemExit("--same-device puts every rank on GPU 0, which RCCL refuses: use --backend gloo")

    import numpy as np
    import torch
    from magr_ray_tracer_amd import dist as rdist, scenes
    from magr_ray_tracer_amd.renderer import Device

    rank, world, local = rdist.init_process_group(args.backend if args.gpus > 1 else None, set_device=not args.same_device)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    W, H = args.width, args.height
    accel = 1 if args.accel == "bvh4" else 0

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
        s, view = scenes.sponza_class(args.detail)
    sa = s.arrays(bvh4=bool(accel))
    build_s = time.time() - t0
    lanes = max(1, args.lanes) if args.shard == "samples" else 1
    # timed region: HIP events around the extend launches only (the roofline's kernel); the per-stage table comes from a short
    # fully-bracketed pass afterwards, outside the timed region
    cam = scenes.camera_for(view, W, H)
    accums = [torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}") for _ in range(lanes)]
    reduced = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")

    # one lane = one context; the interleaved-band plan gives a rank several row bands = several contexts
    ctx_plans = [(m, p) for m in range(lanes) for p in rdist.plans(args.shard, W, H, rank, world, m, lanes, args.band_rows or None)]
    share = len(ctx_plans) > 1      # several contexts share the GPU: smaller footprints, so that they leave each other room

    made = []

    def make_device(m, p):
        # HIP events bracket the extend launches of the FIRST context only: hundreds of launches of the roofline's kernel are timed either
        # way, and the other lanes run without the event packets (with four lanes they cost 1.3 % of the frame rate)
        d = Device(W, H, y0=p["y0"], y1=p["y1"], accel=accel, device=local, profile=0 if args.no_profile or made else 1,
                   extend_variant=args.extend_variant, shade_blocks_per_cu=1 if share else 0, persist_blocks_per_cu=args.persist_blocks if share else 0)
        if made and not args.no_share_scene:
            d.share_scene(made[0])          # one device copy of the scene for all contexts of this rank
        else:
            d.upload(sa)
        made.append(d)
        d.bind_accum(accums[m])
        seeds = np.zeros(p["seed_count"], np.uint32)
        _seed_stream(seeds, p["seed_first"])
        d.set_seeds(seeds)
        return d

    group = rdist.Lanes([make_device(m, p) for m, p in ctx_plans])
    dev = group.devs[0]
    each = args.shard != "samples"     # band plans: the contexts are parts of ONE frame, every one renders every step
    cam["focalLength"] = dev.focus(W // 2, H // 2, cam)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def reduce_all():
        reduced.copy_(accums[0])
        for a in accums[1:]:      # the rank's accumulator = sum of its lanes in lane order ...
            reduced.add_(a)
        rdist.reduce_accumulator(reduced)   # ... then the one exchange step across ranks
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local}" if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    # ---- warmup ------------------------------------------------------------------------------------------
    if args.warmup > 0:
        group.render(cam, args.warmup * (1 if each else len(group)), each=each)
        group.synchronize()
        reduce_all()
    for d in group.devs:
        d.reset()
        d.synchronize()
        d.reset_counters()
        d.reset_stage_times()
    for a in accums:
        a.zero_()
    barrier()

    # ---- timed region: exactly `steps` frames (shared out over the lanes) + the accumulator reduction, repeated ---------------
    dts, render_s, reduce_s = [], 0.0, 0.0
    while True:
        barrier()
        t0 = time.perf_counter()
        group.render(cam, args.steps, each=each)
        group.synchronize()
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
    frames_timed = args.steps * repeats
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
    for d in group.devs:          # work totals over the lanes
        for k, v in d.counters().items():
            ctr[k] = ctr.get(k, 0) + v
    st, ctr0 = group.devs[0].stage_times(), group.devs[0].counters()   # the context whose extend launches carry HIP events, and ITS work

    # ---- untimed: ONE context with the GPU to itself (the reference's own shape: one Renderer) -----------------------------------
    single, stage_tab, con = {}, {}, {}
    value_single = None
    if args.shard == "samples" and not args.no_single:
        barrier()
        solo = dev
        if lanes > 1:     # the lanes' contexts are configured for sharing the GPU; the single-Renderer figure gets a context of its own
            solo = Device(W, H, accel=accel, device=local, profile=0 if args.no_profile else 1, extend_variant=args.extend_variant)
            solo.upload(sa)
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
        dev = solo
        if not args.no_profile and rank == 0:   # 16 more frames of that context with every stage bracketed (kernels undisturbed)
            dev.set_profile(2)
            dev.reset_stage_times()
            c0 = dev.counters()
            dev.render(cam, 16)
            dev.synchronize()
            st2, c1 = dev.stage_times(), dev.counters()
            stage_tab = {k[:-3]: round(st2[k] / 16, 4) for k in st2 if k.endswith("_ms") and k != "compact_ms"}
            dctr = {k: c1[k] - c0[k] for k in c1}
            e_ms = st2["extend_ms"] / max(st2["extend_launches"], 1)
            e_gbs = extend_bytes(dctr, accel) / max(st2["extend_launches"], 1) / (e_ms * 1e-3) / 1e9 if e_ms > 0 else 0.0
            e_rec = (dctr["extend_node_visits"] + dctr["extend_prim_tests"]) / (st2["extend_ms"] * 1e-3) if st2["extend_ms"] > 0 else 0.0
            single = {"achieved": round(e_gbs, 2), "frac": round(e_gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(e_ms, 5),
                      "gather_records_per_s": round(e_rec, 0), "gather_frac": round(e_rec / GATHER_CEILING_RECORDS_PER_S, 4),
                      "note": "the same kernel with the GPU to itself (one context, 16 untimed frames after the timed region)"}
            c_ms = st2["connect_ms"] / max(st2["connect_launches"], 1)
            c_rec = (dctr["connect_node_visits"] + dctr["connect_prim_tests"]) / (st2["connect_ms"] * 1e-3) if st2["connect_ms"] > 0 else 0.0
            c_gbs = extend_bytes(dctr, accel, "connect") / max(st2["connect_launches"], 1) / (c_ms * 1e-3) / 1e9 if c_ms > 0 else 0.0
            con = {"achieved": round(c_gbs, 2), "frac": round(c_gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(c_ms, 5),
                   "gather_records_per_s": round(c_rec, 0), "gather_frac": round(c_rec / GATHER_CEILING_RECORDS_PER_S, 4),
                   "rays_per_launch": dctr["connect_rays"] // max(st2["connect_launches"], 1),
                   "note": "connect is an any-hit traversal with its own visit order; its counters are its own work, not the reference's"}
        barrier()

    nshare = world if args.shard == "samples" else 1
    samples = W * H * frames_timed * nshare
    value = samples / dt / 1e6

    if rank == 0:
        ext_launches = max(st["extend_launches"], 1)
        ext_ms = st["extend_ms"] / ext_launches
        ext_bytes = extend_bytes(ctr0, accel) / ext_launches
        ext_gbs = ext_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
        ext_rec = (ctr0["extend_node_visits"] + ctr0["extend_prim_tests"]) / (st["extend_ms"] * 1e-3) if st["extend_ms"] > 0 else 0.0
        job_rec = (ctr["extend_node_visits"] + ctr["extend_prim_tests"] + ctr["connect_node_visits"] + ctr["connect_prim_tests"]) * nshare / dt / world   # per GPU
        job_gbs = (extend_bytes(ctr, accel) + extend_bytes(ctr, accel, "connect")) * nshare / dt / world / 1e9   # per GPU, extend + connect of all lanes
        traffic, traffic_note = None, "no PMC measurement committed for this configuration"
        tpath = os.path.join(ROOT, "profiles", "extend_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                same = not args.model and tj.get("config", {}) == {"accel": args.accel, "detail": args.detail, "width": W, "height": H}
                if same:
                    traffic, traffic_note = tj.get("hbm_bytes_per_launch"), tj.get("source", "")
            except Exception:
                pass
        traced = ctr["extend_rays"] + ctr["connect_rays"]
        roof = {"bound": "hbm", "kernel": "extend = " + dev.extend_kernel_name(), "achieved": round(ext_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ext_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(ext_bytes), "avg_launch_ms": round(ext_ms, 5), "launches": st["extend_launches"],
                "dram_frac": round(traffic / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and ext_ms > 0 else None,
                "traffic_note": traffic_note,
                "limiter": "memory latency + divergence of a dependent random-gather chain (vector L1 / L2 hits), not HBM bandwidth: `frac` is "
                           "SURVEY 8(d)'s algorithmic bytes over the HBM peak and may exceed 1 because ~21 MB of node and triangle records "
                           "are served from L1/L2/Infinity Cache; `dram_frac` is the measured DRAM-side traffic over the same peak; "
                           "`gather` prices the kernel against what the chip sustains for its access pattern",
                "gather": {"records_per_s": round(ext_rec, 0), "ceiling_records_per_s": GATHER_CEILING_RECORDS_PER_S,
                           "frac": round(ext_rec / GATHER_CEILING_RECORDS_PER_S, 4),
                           # a whole GPU's share of the job: every node and triangle record its lanes fetched in the timed region (extend + connect) over the wall time
                           "job_records_per_s": round(job_rec, 0), "job_frac": round(job_rec / GATHER_CEILING_RECORDS_PER_S, 4),
                           "note": "records = interior-node pair fetches + triangle-record fetches of all extend launches / their summed HIP-event "
                                   "time; ceiling = tools/gather_probe.hip at the table size that matches the measured L1/L2 hit rates (<= 4 MB rows, "
                                   "profiles/r02_gather_probe.log).  With lanes > 1 two contexts' launches overlap, so the per-launch rate is below "
                                   "`single_stream` while the frame rate is higher"},
                "single_stream": single,
                "timed_context": "HIP events on the extend launches of lane 0 (of %d)" % len(group.devs),
                # `achieved` / `frac` are per LAUNCH: a launch takes longer while other contexts' kernels share the GPU.  The
                # job-level figure: algorithmic bytes of every extend and connect launch of all lanes over the wall time of the timed region
                "job_algorithmic_gbs": round(job_gbs, 1), "job_algorithmic_frac": round(job_gbs / HBM_PEAK_GBS, 4),
                "per_ray": {"node_visits": round(ctr["extend_node_visits"] / max(ctr["extend_rays"], 1), 2),
                            "prim_tests": round(ctr["extend_prim_tests"] / max(ctr["extend_rays"], 1), 2)}}
        out = {
            "metric": "Mrays/sec + extend-kernel HBM GB/s, sponza 1080p 256spp",
            "value": round(value, 3), "unit": "Mrays/s (primary samples/s, reference definition renderer.cpp:60-62)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / frames_timed * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if args.shard == "samples" else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"model {os.path.basename(args.model)}" if args.model else "sponza-class procedural atrium") +
                                   f" ({len(sa.prims)} prims, SAH {args.accel.upper()}, "
                                   f"{len(sa.bvh2)} nodes) {W}x{H}, NEE+cosine+RR+firefly, 7 bounces; step = 1 spp frame, "
                                   f"{args.steps} spp timed per GPU (BASELINE config 3 = 256 spp) as {lanes} interleaved sample stream(s)",
                       "shard": args.shard, "lanes": lanes, "contexts": len(ctx_plans), "triangles": int(len(sa.prims)), "extend_variant": args.extend_variant},
            "repeats": repeats, "timed_s": round(dt, 4),
            "value_single_context": round(value_single, 3) if value_single else None,
            "rank_s": {"render": [round(x[0], 4) for x in per_rank], "all_reduce": [round(x[1], 5) for x in per_rank]},
            "traced_mrays_per_s": round(traced * nshare / dt / 1e6, 2),
            "rays_per_step": {"extend": ctr["extend_rays"] // frames_timed, "connect": ctr["connect_rays"] // frames_timed},
            "roofline": roof, "connect_roofline": con, "stage_ms_per_step": stage_tab,
            "host_build_s": round(build_s, 2), "accum_rgb_sum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sa, cam, W, H, accel)
        print(json.dumps(out), flush=True)
    group.close()
    if dev not in group.devs:
        dev.close()
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


def cpu_baseline(sa, cam, W, H, accel):
    """The oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores on a bounded
    sample of the same workload: full-resolution frames, row-band parallel over all cores."""
    from oracle.oracle_py import Oracle
    cores = usable_cores()
    o = Oracle(sa, W, H, accel=accel)
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
