"""Multi-GPU harness: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU
for tests).  The path shards with no data-path collective; the only exchange is ONE reduction of the float
accumulator per render (BASELINE.json north star: "samples partition across the 8 GPUs ... final RCCL reduce").

  shard="samples": every rank renders the full frame with its own disjoint seed slice
                   (rank r: seeds[i] = (r*P + i + 1)-th xorshift32 output), then all_reduce(SUM).
  shard="bands":   rank r renders rows [r*H/N, (r+1)*H/N) with seeds[a..b) of the frame's stream; the
                   all_reduce(SUM) of the zero-padded full-frame accumulators is exact (adds zeros) and
                   equals a gather.
  shard="ibands":  the same with the frame cut into bands of `band_rows` rows dealt out round-robin (band k -> rank k mod N;
                   SURVEY.md §8(e): "bands of 16-32 rows round-robin" against sky-vs-geometry imbalance; BASELINE config 4's
                   "screen-tile shard").  A rank runs one context per band it owns, interleaved like lanes.

Lanes: the sample partition applied once more INSIDE a GPU.  Every launch of a frame ends in a tail of a few long rays during which
most of the chip idles, and consecutive frames of one accumulation cannot overlap (each continues the RNG state of the one before).
Independent sample streams can: a rank runs `lanes` contexts (own HIP stream, queues, accumulator and seed slice - virtual rank
rank*lanes + lane) and interleaves their frames, so one context's tails are filled by the other's kernels (+14 % on the bench scene with
2 lanes).  The rank's accumulator is the sum of its lanes' accumulators in lane order, then the ranks are reduced as before.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def band_rows(height, rank, world):
    return (height * rank) // world, (height * (rank + 1)) // world


def seed_offset(shard, width, height, rank, world, lane=0, lanes=1):
    """First index into the reference's host seed stream for this rank (and lane), and the number of seeds.  A context renders sample
    stream `stream` (the stream-th slice of W*H seeds) restricted to its rows: samples plan - every (rank, lane) has a stream of its
    own, rank * lanes + lane, and renders all rows; band plans - the ranks split the ROWS, and the lanes of a band are sample streams
    0 .. lanes-1 of those rows (lane 0 alone is the frame's own seed slice, as the reference would render the band)."""
    if shard == "samples":
        return (rank * lanes + lane) * width * height, width * height
    y0, y1 = band_rows(height, rank, world)
    return lane * width * height + y0 * width, (y1 - y0) * width


def plan(shard, width, height, rank, world, lane=0, lanes=1):
    if shard == "samples":
        y0, y1 = 0, height
    elif shard == "bands":
        y0, y1 = band_rows(height, rank, world)
    else:
        raise ValueError(f"unknown shard mode {shard!r}")
    first, n = seed_offset(shard, width, height, rank, world, lane, lanes)
    return dict(y0=y0, y1=y1, seed_first=first, seed_count=n, stream=(rank * lanes + lane) if shard == "samples" else lane)


def interleaved_bands(height, rank, world, band_rows=None):
    """Row bands [y0, y1) of rank `rank` under the round-robin plan: band k (rows [k*b, (k+1)*b)) belongs to rank k mod world.
    Default band height: a quarter of a contiguous band, at least 8 rows (so that a rank owns ~4 bands spread over the frame)."""
    b = band_rows or max(8, -(-height // (world * 4)))
    return [(y, min(y + b, height)) for k, y in enumerate(range(0, height, b)) if k % world == rank]


def plans(shard, width, height, rank, world, lane=0, lanes=1, band_rows=None):
    """All contexts of this rank (and lane): one plan for "samples" / "bands", one per owned band for "ibands"."""
    if shard != "ibands":
        return [plan(shard, width, height, rank, world, lane, lanes)]
    return [dict(y0=y0, y1=y1, seed_first=lane * width * height + y0 * width, seed_count=(y1 - y0) * width, stream=lane)
            for y0, y1 in interleaved_bands(height, rank, world, band_rows)]


def rank_frames(total, rank, world):
    """Strong scaling of a fixed render (`--total-steps`): how many of `total` 1-spp frames rank `rank` renders under the sample
    plan (the first total % world ranks take one more)."""
    return total // world + (1 if rank < total % world else 0)


def lane_frames(frames, lanes):
    """How many of `frames` 1-spp frames each lane renders (the first frames % lanes lanes take one more)."""
    return [frames // lanes + (1 if m < frames % lanes else 0) for m in range(lanes)]


class Lanes:
    """`lanes` Device contexts of one rank rendering disjoint sample streams of the same frame concurrently (see the module text)."""

    def __init__(self, lanes, make_device=None, seeds_for=None):
        """Either a list of uploaded, seeded Devices, or a count with make_device(lane) -> an uploaded Device and
        seeds_for(lane) -> its uint32 seed slice."""
        if make_device is None:
            self.devs = list(lanes)
            return
        self.devs = [make_device(m) for m in range(lanes)]
        for m, d in enumerate(self.devs):
            d.set_seeds(seeds_for(m))

    def __len__(self):
        return len(self.devs)

    def render(self, cam, frames, each=False):
        """`frames` frames in total, interleaved over the lanes so that their kernels overlap on the GPU (each=True: `frames`
        frames on EVERY context - the band plans, where the contexts are parts of one frame)."""
        todo = [frames] * len(self.devs) if each else lane_frames(frames, len(self.devs))
        for f in range(max(todo)):
            for m, d in enumerate(self.devs):
                if f < todo[m]:
                    d.render(cam, 1)

    def synchronize(self):
        for d in self.devs:
            d.synchronize()

    def each(self, fn):
        return [fn(d) for d in self.devs]

    def read_accum(self):
        """Sum of the lanes' accumulators in lane order (host side; bench.py does the same on the device with torch)."""
        acc = self.devs[0].read_accum()
        for d in self.devs[1:]:
            acc = acc + d.read_accum()
        return acc

    def close(self):
        for d in self.devs:
            d.close()


class Groups:
    """The library-level lanes of one rank: one renderer.Group (rt_group_*: `lanes` sample streams behind one handle, one device copy
    of the scene) per row band the rank owns - ONE group for the "samples" and "bands" plans, one per owned band for "ibands"."""

    def __init__(self, groups):
        self.groups = list(groups)

    @property
    def devs(self):
        return [d for g in self.groups for d in g.devs]

    def __len__(self):
        return sum(len(g) for g in self.groups)

    def render(self, cam, frames):
        """`frames` frames of EVERY group (a group deals its frames to its lanes), queued in turns of one frame per lane so that all
        contexts' kernels overlap on the GPU."""
        left = [frames] * len(self.groups)
        while any(left):
            for k, g in enumerate(self.groups):
                n = min(left[k], len(g))
                if n:
                    g.render(cam, n)
                    left[k] -= n

    def synchronize(self):
        for g in self.groups:
            g.synchronize()

    def sum_into(self, tensor):
        """Every group adds up its lanes (lane order) into its rows of `tensor`; the bands of a rank are disjoint."""
        for g in self.groups:
            g.sum_into(tensor)

    def reset(self):
        for g in self.groups:
            g.reset()

    def close(self):
        for g in self.groups:
            g.close()


def init_process_group(backend=None, set_device=True):
    import torch
    import torch.distributed as dist
    rank, world, local = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and set_device:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def reduce_accumulator(tensor):
    """The single exchange step of a render: SUM of the per-rank float accumulators (in place)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if tensor.is_cuda and dist.get_backend() == "gloo":   # rehearsal path: gloo reduces on the host
            host = tensor.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
