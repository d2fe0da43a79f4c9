"""Multi-GPU plumbing: one process per GPU, triples / images sharded over ranks, one
RCCL all-gather of the finished images over xGMI.  The generator
and encoders hold frozen parameters and images are independent, so there is no
collective inside the forward; each rank owns a full replica (121 MB generator).

`torch.distributed` backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.  Returns
    (rank, world_size, local_rank).  Single-process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("HF_FORCE_DIST", "0") == "1"  # single-rank RCCL smoke test of the N>1 code path
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): the first (n % world) ranks get one
    extra item.  Returns (start, stop)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def to_uint8_image(img):
    """Final image quantisation of the pipeline: ((I+1)/2).clip(0,1) (Blending.py:80)
    followed by the 8-bit conversion save_image applies."""
    return ((img + 1.0) * 127.5).clamp_(0.0, 255.0).round_().to(torch.uint8)


def all_gather_images(local, n_total=None, group=None, async_op=False):
    """Gather per-rank result tensors [n_local, ...] into [sum n_local, ...] on every rank.
    Ranks may hold different counts (block partition): shorter shards are padded to the
    longest one for the collective and trimmed afterwards.  Returns the gathered tensor
    (or (tensor, work, finalize) when async_op=True)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and os.environ.get("HF_FORCE_DIST", "0") != "1"):
        return (local, None, lambda t: t) if async_op else local
    world = dist.get_world_size(group)
    if n_total is None:
        counts = [local.shape[0]] * world
    else:
        counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    n_max = max(counts)
    send = local
    if local.shape[0] < n_max:
        pad = local.new_zeros((n_max - local.shape[0],) + tuple(local.shape[1:]))
        send = torch.cat([local, pad], 0)
    send = send.contiguous()
    out = send.new_empty((world * n_max,) + tuple(send.shape[1:]))
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)

    def finalize(t):
        if all(c == n_max for c in counts):
            return t
        return torch.cat([t[r * n_max: r * n_max + counts[r]] for r in range(world)], 0)

    if async_op:
        return out, work, finalize
    return finalize(out)
