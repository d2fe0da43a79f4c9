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


def pin_rank_to_cores(local_rank, local_world):
    """Give rank `local_rank` of `local_world` on this node its own contiguous slice of the host's hardware threads
    (os.sched_setaffinity) and size torch's intra-op pool to it: eight launch-bound ranks otherwise migrate over - and
    oversubscribe - the same cores.  Returns the slice (or None where affinity is not available / not divisible)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        if per < 1:
            return None
        mine = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, 32)))
        return mine
    except OSError:
        return None


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


def swap_many(swap_fn, n_total, load_triple, device=None, chunk=8, group=None, batch=1, swap_batch_fn=None, stats=None):
    """BASELINE.json configs[3]: hair swaps of triples 0..n_total-1, block-partitioned over the ranks
    (one process per GPU, a full replica each - triples share no state, models/Net.py:44-46), results
    returned to every rank as uint8 images in triple order [n_total, 3, H, W].

      load_triple(i) -> (face, shape, color)   three CPU tensors (uint8 [3,H,W]); pinned memory lets the
                                               host-to-device copy of triple i+1 overlap swap i
      swap_fn(face, shape, color) -> float image [3,H,W] in [0,1]  (HairFast.swap)
      batch > 1 with swap_batch_fn(list of triples) -> list of images (HairFast.swap_batch): `batch` consecutive
                                               local triples per call - one batched pass over the hot path

    The only collective is the RCCL all-gather of finished images over xGMI (3.1 MB per 1024^2 image),
    issued asynchronously once per `chunk` local triples so that it overlaps the following swaps (one
    final round trip instead of a 100 MB-per-rank tail); a rank whose shard is shorter pads.  Works
    without a process group (world 1) and under gloo (CPU tests).  Returns (images_u8, n_local).

    stats: a dict that receives this call's balance figures (costs one device synchronisation before the gather tail):
      compute_s            this rank's wall time from the first H2D to the completion of its last swap
      gather_tail_s        ... from there to the completion of the last all-gather round (the exposed part of the collective)
      per_rank_compute_s   every rank's compute_s (one small all-gather); imbalance = max / min of them"""
    import time as _time

    t_begin = _time.perf_counter()
    init = dist.is_initialized()
    world = dist.get_world_size(group) if init else 1
    rank = dist.get_rank(group) if init else 0
    if n_total < world:
        raise ValueError(f"{n_total} triples cannot be sharded over {world} ranks")
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    lo, hi = shard_range(n_total, rank, world)
    n_local, n_max = hi - lo, max(counts)
    use_cuda = device is not None and torch.device(device).type == "cuda"
    copy_stream = torch.cuda.Stream(device) if use_cuda else None
    collective = init and (world > 1 or os.environ.get("HF_FORCE_DIST", "0") == "1")

    def fetch(i):  # host -> device copy of triple i on a side stream
        imgs = load_triple(i)
        if not use_cuda:
            return imgs, None
        with torch.cuda.stream(copy_stream):
            on_dev = tuple(t.to(device, non_blocking=True) for t in imgs)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return on_dev, ev

    if batch > 1 and swap_batch_fn is None:
        raise ValueError("batch > 1 needs swap_batch_fn")
    batch = max(1, batch)
    chunk = max(chunk, batch)  # a batched pass is never split across gather rounds: one all-gather per `chunk` >= batch triples

    def fetch_group(j0, j1):  # local triples j0..j1-1
        return [fetch(lo + j) for j in range(j0, j1)]

    def group_end(j0, limit):
        return min(j0 + batch, limit)

    rounds = []  # (k, gathered-or-local tensor, work|None, chunk size)
    first_end = group_end(0, min(chunk, n_local))
    nxt = fetch_group(0, first_end)
    done = []
    for k in range((n_max + chunk - 1) // chunk):
        cs = min(chunk, n_max - k * chunk)
        limit = min(k * chunk + cs, n_local)
        j = k * chunk
        while j < limit:
            j1 = group_end(j, limit)
            cur_group = nxt
            # prefetch the next group (of this chunk or the first of the next one) while this one is swapped
            if j1 < n_local:
                nlimit = limit if j1 < limit else min(j1 + chunk, n_local)
                nxt = fetch_group(j1, group_end(j1, nlimit))
            else:
                nxt = None
            triples = []
            for imgs, ev in cur_group:
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)
                    for t in imgs:  # allocated on the copy stream, consumed on this one: keep the block from being
                        t.record_stream(cur)  # handed to the NEXT prefetch while these kernels still read it
                triples.append(imgs)
            if batch > 1:
                for img in swap_batch_fn(triples):
                    done.append(to_uint8_image(img * 2.0 - 1.0))
            else:
                done.append(to_uint8_image(swap_fn(*triples[0]) * 2.0 - 1.0))
            j = j1
        mine = done[k * chunk:k * chunk + cs]
        send = torch.stack(mine + [torch.zeros_like(done[0])] * (cs - len(mine)))
        if collective:
            out = send.new_empty((world * cs,) + tuple(send.shape[1:]))
            work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)
            rounds.append((k, out, work, cs))
        else:
            rounds.append((k, send, None, cs))
    if stats is not None:
        if use_cuda:
            torch.cuda.current_stream().synchronize()
        stats["compute_s"] = _time.perf_counter() - t_begin
    starts = [shard_range(n_total, r, world)[0] for r in range(world)]
    result = done[0].new_empty((n_total,) + tuple(done[0].shape))
    for k, out, work, cs in rounds:
        if work is not None:
            work.wait()
        for r in range(world if collective else 1):
            valid = max(0, min(cs, counts[r if collective else rank] - k * chunk))
            if valid:
                g0 = starts[r if collective else rank] + k * chunk
                result[g0:g0 + valid] = out[r * cs:r * cs + valid]
    if stats is not None:
        if use_cuda:
            torch.cuda.current_stream().synchronize()
        stats["gather_tail_s"] = _time.perf_counter() - t_begin - stats["compute_s"]
        mine_t = torch.tensor([stats["compute_s"]], dtype=torch.float64, device=device if use_cuda else "cpu")
        if collective:
            every = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(every, mine_t, group=group)
            stats["per_rank_compute_s"] = [float(t.item()) for t in every]
        else:
            stats["per_rank_compute_s"] = [stats["compute_s"]]
        stats["imbalance"] = max(stats["per_rank_compute_s"]) / max(min(stats["per_rank_compute_s"]), 1e-9)
    return result, n_local
