"""N>1 path on CPU: two processes over gloo exercise the sharding plan and the single accumulator reduction of
magr_ray_tracer_amd/dist.py.  The per-rank "renderer" here is the oracle (tests may use it); on the GPU box the
same plan + reduce run with the HIP path (bench.py --gpus N)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from magr_ray_tracer_amd import dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_partitions_rows_and_seeds():
    W, H = 64, 37
    for world in (1, 2, 3, 8):
        rows = []
        firsts = []
        for r in range(world):
            p = rdist.plan("bands", W, H, r, world)
            rows += list(range(p["y0"], p["y1"]))
            assert p["seed_first"] == p["y0"] * W and p["seed_count"] == (p["y1"] - p["y0"]) * W
            q = rdist.plan("samples", W, H, r, world)
            assert (q["y0"], q["y1"]) == (0, H) and q["seed_count"] == W * H
            firsts.append(q["seed_first"])
        assert rows == list(range(H))                       # every row exactly once, in order
        assert firsts == [r * W * H for r in range(world)]  # disjoint seed slices


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from magr_ray_tracer_amd import dist as rdist, scenes
    from oracle.oracle_py import Oracle, seed_stream
    shard = sys.argv[1]
    rank, world, local = rdist.init_process_group("gloo")
    W, H, frames = 48, 28, 2
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, W, H)
    o = Oracle(sa, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    for p in rdist.plans(shard, W, H, rank, world, band_rows=5 if shard == "ibands" else None):   # one context per owned band
        o.render(cam, frames, accum=acc, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
    t = torch.from_numpy(acc)
    rdist.reduce_accumulator(t)
    if rank == 0:
        np.save(sys.argv[2], t.numpy())
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("shard", ["bands", "samples", "ibands"])
def test_two_ranks_gloo_reduce(tmp_path, shard):
    from magr_ray_tracer_amd import scenes
    from oracle.oracle_py import Oracle, seed_stream
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    out = tmp_path / "acc.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(script), shard, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=ROOT)
    got = np.load(out)
    W, H, frames = 48, 28, 2
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, W, H)
    o = Oracle(sa, W, H)
    exp = np.zeros((H, W, 4), np.float32)
    for r in range(2):
        part = np.zeros((H, W, 4), np.float32)
        for p in rdist.plans(shard, W, H, r, 2, band_rows=5 if shard == "ibands" else None):
            o.render(cam, frames, accum=part, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
        exp = exp + part          # two addends: the sum is order-independent, so gloo's reduction must match bit for bit
    assert np.array_equal(got, exp)
    if shard == "ibands":
        # bands of 5 rows dealt out round-robin: the two ranks' rows interleave and together cover the frame exactly once; a band
        # renders with the frame's own seed slice, so the result equals the oracle rendering the same bands one after another
        rows = sorted(y for r in range(2) for y0, y1 in rdist.interleaved_bands(H, r, 2, 5) for y in range(y0, y1))
        assert rows == list(range(H)) and rdist.interleaved_bands(H, 0, 2, 5)[:2] == [(0, 5), (10, 15)]
    if shard == "bands":
        # zero-padded reduce == gather of the bands
        p0, p1 = rdist.plan(shard, W, H, 0, 2), rdist.plan(shard, W, H, 1, 2)
        assert got[p0["y0"]:p0["y1"]].any() and got[p1["y0"]:p1["y1"]].any()


WORKER8 = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    torch.set_num_threads(1)
    from magr_ray_tracer_amd import dist as rdist, scenes
    from oracle.oracle_py import Oracle, seed_stream
    shard, lanes, total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world, local = rdist.init_process_group("gloo")
    W, H = 40, 24
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, W, H)
    o = Oracle(sa, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    # a FIXED `total`-spp image: the sample plan splits the samples over ranks x lanes, the band plans split the rows over the ranks and
    # the samples of a band over its lanes (bench.py --total-steps)
    mine = rdist.rank_frames(total, rank, world) if shard == "samples" else total
    for m, frames in enumerate(rdist.lane_frames(mine, lanes)):
        for p in rdist.plans(shard, W, H, rank, world, m, lanes, band_rows=2 if shard == "ibands" else None):
            if frames:
                o.render(cam, frames, accum=acc, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
    t = torch.from_numpy(acc)
    rdist.reduce_accumulator(t)
    if rank == 0:
        np.save(sys.argv[4], t.numpy())
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("shard", ["samples", "bands", "ibands"])
def test_eight_ranks_gloo_fixed_image_all_plans(tmp_path, shard):
    """world_size 8 (the node BASELINE configs 4 and 5 are quoted on), a fixed 16-spp image (bench.py --total-steps), two lanes per rank.
    Band plans: a pixel has ONE non-zero addend in the reduction (the other seven ranks add exact zeros), so the reduced accumulator is
    bit-identical to the bands rendered one after another, whatever order the collective sums in.  Sample plan: eight non-zero addends
    per pixel; a ring or tree all-reduce adds them in another order than rank order and float addition is not associative, so the
    statement is a bound, not equality: |reduced - exact| <= 7 * 2^-24 * sum |addend| per component (7 roundings of at most half an ulp
    of a partial sum each) - about 4e-7 relative, far inside the 1e-4 of the north star."""
    from magr_ray_tracer_amd import scenes
    from oracle.oracle_py import Oracle, seed_stream
    world, lanes, total = 8, 2, 16
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % dict(root=ROOT))
    out = tmp_path / "acc8.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29578", str(script), shard, str(lanes), str(total), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    W, H = 40, 24
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, W, H)
    o = Oracle(sa, W, H)
    parts = []
    for r in range(world):
        part = np.zeros((H, W, 4), np.float32)
        mine = rdist.rank_frames(total, r, world) if shard == "samples" else total
        for m, frames in enumerate(rdist.lane_frames(mine, lanes)):
            for p in rdist.plans(shard, W, H, r, world, m, lanes, band_rows=2 if shard == "ibands" else None):
                if frames:
                    o.render(cam, frames, accum=part, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
        parts.append(part)
    if shard == "samples":
        exact = np.sum([p.astype(np.float64) for p in parts], axis=0)
        bound = 7 * 2.0 ** -24 * np.sum([np.abs(p).astype(np.float64) for p in parts], axis=0)
        assert (np.abs(got.astype(np.float64) - exact) <= bound + 1e-300).all()
        assert got[..., :3].sum() > 0 and all(p.any() for p in parts)       # every rank contributed samples
    else:
        seq = np.zeros((H, W, 4), np.float32)
        for p in parts:
            seq = seq + p
        assert np.array_equal(got, seq)                                      # bit for bit, independent of the reduction order
        rows = [y for r in range(world) for p in rdist.plans(shard, W, H, r, world, band_rows=2 if shard == "ibands" else None) for y in range(p["y0"], p["y1"])]
        assert sorted(rows) == list(range(H))                                # the ranks' bands tile the frame exactly once


def test_lanes_are_virtual_ranks_of_the_sample_plan():
    """Lanes partition the samples once more inside a rank: (rank, lane) takes the seed slice of virtual rank rank*lanes + lane, the slices
    of all (rank, lane) pairs tile the stream without gap or overlap, and `steps` frames are shared out exactly."""
    W, H, world, lanes = 7, 5, 3, 2
    P = W * H
    seen = []
    for r in range(world):
        for m in range(lanes):
            p = rdist.plan("samples", W, H, r, world, m, lanes)
            assert (p["y0"], p["y1"], p["seed_count"]) == (0, H, P)
            seen.append(p["seed_first"])
    assert seen == [v * P for v in range(world * lanes)]
    assert rdist.plan("samples", W, H, 2, world) == rdist.plan("samples", W, H, 2, world, 0, 1)
    # band plans: the ranks split the rows, the lanes of a band are sample streams 0 .. lanes-1 of THOSE rows (lane 0 = the frame's own slice)
    for r in range(world):
        y0, y1 = rdist.band_rows(H, r, world)
        for m in range(lanes):
            p = rdist.plan("bands", W, H, r, world, m, lanes)
            assert (p["y0"], p["y1"], p["seed_first"], p["seed_count"], p["stream"]) == (y0, y1, m * P + y0 * W, (y1 - y0) * W, m)
    assert [rdist.rank_frames(1024, r, 8) for r in range(8)] == [128] * 8 and sum(rdist.rank_frames(10, r, 4) for r in range(4)) == 10
    for frames in (0, 1, 5, 256):
        for n in (1, 2, 3):
            parts = rdist.lane_frames(frames, n)
            assert sum(parts) == frames and max(parts) - min(parts) <= 1 and parts == sorted(parts, reverse=True)


def test_bench_launcher_reports_a_failing_rank():
    """`python bench.py --gpus 2` starts its own ranks (no torch.distributed.run around it).  Here there is no GPU, so every rank fails
    in rt_create ("no HIP device visible: this library has no CPU path") - the launcher must end the other rank, not hang in the
    rendezvous, and exit non-zero without printing a result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        import ctypes
        from magr_ray_tracer_amd import _lib
        if _lib.device_lib().rt_device_count() > 0:
            pytest.skip("a GPU is present: the ranks would succeed (covered by test_bench_starts_its_own_ranks_two_on_one_gpu)")
    except Exception:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "0",
                        "--width", "64", "--height", "36", "--detail", "0.1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no HIP device" in r.stderr or "RtError" in r.stderr or "hip" in r.stderr.lower()
