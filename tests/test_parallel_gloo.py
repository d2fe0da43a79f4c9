"""world_size-2 gloo tests (CPU) of the sharding + all-gather path used for N>1 GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hairfastgan_amd import parallel

    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_range(n_total, rank, world)
    # "images": item i is filled with the value i
    local = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 3, 4, 4)
    u8 = parallel.to_uint8_image(local / 127.5 - 1.0)
    got = parallel.all_gather_images(u8, n_total=n_total)
    out, work, fin = parallel.all_gather_images(u8, n_total=n_total, async_op=True)
    work.wait()
    got2 = fin(out)
    q.put((rank, got[:, 0, 0, 0].tolist(), got2[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 5])
def test_shard_and_all_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, got2 in res:
        assert got == list(range(n_total)), (rank, got)
        assert got2 == got


def test_shard_range_covers_everything():
    from hairfastgan_amd.parallel import shard_range

    for n in (0, 1, 7, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
