"""Multi-GPU harness: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU
for tests).  The path shards with no data-path collective; the only exchange is ONE reduction of the float
accumulator per render (BASELINE.json north star: "samples partition across the 8 GPUs ... final RCCL reduce").

  shard="samples": every rank renders the full frame with its own disjoint seed slice
                   (rank r: seeds[i] = (r*P + i + 1)-th xorshift32 output), then all_reduce(SUM).
  shard="bands":   rank r renders rows [r*H/N, (r+1)*H/N) with seeds[a..b) of the frame's stream; the
                   all_reduce(SUM) of the zero-padded full-frame accumulators is exact (adds zeros) and
                   equals a gather.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def band_rows(height, rank, world):
    return (height * rank) // world, (height * (rank + 1)) // world


def seed_offset(shard, width, height, rank, world):
    """First index into the reference's host seed stream for this rank, and the number of seeds."""
    if shard == "samples":
        return rank * width * height, width * height
    y0, y1 = band_rows(height, rank, world)
    return y0 * width, (y1 - y0) * width


def plan(shard, width, height, rank, world):
    if shard == "samples":
        y0, y1 = 0, height
    elif shard == "bands":
        y0, y1 = band_rows(height, rank, world)
    else:
        raise ValueError(f"unknown shard mode {shard!r}")
    first, n = seed_offset(shard, width, height, rank, world)
    return dict(y0=y0, y1=y1, seed_first=first, seed_count=n)


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
