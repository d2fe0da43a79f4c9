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


def _swap_worker(rank, world, port, n_total, chunk, q, batch=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hairfastgan_amd import parallel

    parallel.init_from_env("gloo")
    calls = []

    def load_triple(i):  # three "images" whose content identifies the triple
        return tuple(torch.full((3, 4, 4), (3 * i + k) % 256, dtype=torch.uint8) for k in range(3))

    def swap_fn(face, shape, color):  # a stand-in for HairFast.swap: float image in [0,1] derived from all three inputs
        calls.append(int(face[0, 0, 0]) // 3)
        return ((face.float() + shape.float() + color.float()) / 3.0 / 255.0)

    groups = []

    def swap_batch_fn(triples):  # HairFast.swap_batch stand-in
        groups.append(len(triples))
        return [swap_fn(*t) for t in triples]

    stats = {}
    got, n_local = parallel.swap_many(swap_fn, n_total, load_triple, chunk=chunk, batch=batch,
                                      swap_batch_fn=swap_batch_fn if batch > 1 else None, stats=stats)
    # balance figures: every rank's compute time on every rank, the exposed gather tail, max / min
    assert len(stats["per_rank_compute_s"]) == world and stats["per_rank_compute_s"][rank] == stats["compute_s"]
    assert stats["imbalance"] >= 1.0 and stats["gather_tail_s"] >= 0.0
    assert all(1 <= g <= min(batch, chunk) for g in groups) and (batch == 1 or sum(groups) == n_local)
    q.put((rank, n_local, calls, got[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,chunk,batch", [(7, 2, 1), (8, 3, 1), (2, 8, 1), (9, 4, 3), (7, 8, 4)])
def test_swap_many_world2(n_total, chunk, batch):
    """BASELINE configs[3] driver (parallel.swap_many) at world size 2 over gloo: contiguous shards, ragged
    chunked all-gathers, every rank ends with all images in triple order; batch > 1: groups of consecutive local
    triples go through the batched swap."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_swap_worker, args=(r, 2, port, n_total, chunk, q, batch)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [float(round(((3 * i) % 256 + (3 * i + 1) % 256 + (3 * i + 2) % 256) / 3.0)) for i in range(n_total)]
    seen = []
    for rank, n_local, calls, got in res:
        assert [float(v) for v in got] == want, (rank, got, want)
        assert len(calls) == n_local
        seen += calls
    assert sorted(seen) == list(range(n_total))  # every triple swapped exactly once, on one rank


def test_swap_many_single_process():
    from hairfastgan_amd import parallel

    got, n = parallel.swap_many(lambda a, b, c: a.float() / 255.0, 5, lambda i: tuple(torch.full((3, 2, 2), 10 * i, dtype=torch.uint8) for _ in range(3)), chunk=2)
    assert n == 5 and got[:, 0, 0, 0].tolist() == [0, 10, 20, 30, 40]
    # rank pinning: a rank's slice of the host's hardware threads (no-op for a single rank)
    assert parallel.pin_rank_to_cores(0, 1) is None
    before, threads = os.sched_getaffinity(0), torch.get_num_threads()
    try:
        if len(before) >= 2:
            mine = parallel.pin_rank_to_cores(1, 2)
            half = len(before) // 2
            assert mine == sorted(before)[half:2 * half] and os.sched_getaffinity(0) == set(mine)
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)  # the oracle's bit-for-bit goldens depend on ATen's thread partition
    sizes = []
    got, n = parallel.swap_many(None, 7, lambda i: tuple(torch.full((3, 2, 2), 10 * i, dtype=torch.uint8) for _ in range(3)), chunk=5, batch=3,
                                swap_batch_fn=lambda ts: (sizes.append(len(ts)), [t[0].float() / 255.0 for t in ts])[1])
    assert n == 7 and got[:, 0, 0, 0].tolist() == [0, 10, 20, 30, 40, 50, 60] and sizes == [3, 2, 2]


def test_bench_gpus2_self_launches_two_ranks():
    """`python bench.py --gpus 2` without a torchrun environment re-executes itself under torch.distributed.run
    with two ranks (here: gloo, the launcher / process-group workload that needs no GPU) and rank 0 prints ONE
    JSON line carrying the rank count the process group observed."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "launch-check", "--steps", "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["ranks_observed"] == 2 and doc["gather_ok"] and doc["backend"] == "gloo"
    # a WORLD_SIZE that contradicts --gpus is refused, not silently reported as a different job
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "launch-check"],
                        env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in r2.stderr
    if not torch.cuda.is_available():  # and the GPU workloads refuse to run with fewer GPUs than ranks
        r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                            text=True, timeout=300)
        assert r3.returncode != 0 and "GPU(s) visible" in r3.stderr


def test_bench_gpus8_dry_run_under_gloo():
    """The rank count the driver's scaling run uses, before any 8-GPU node exists for it: `python bench.py --gpus 8
    --workload launch-check` self-launches EIGHT gloo ranks on this container's cores - rendezvous on 127.0.0.1, the max-over-ranks
    clock, every rank's core slice (len(cpus) // 8, disjoint), and a ragged 29-triple block partition gathered in triple
    order through parallel.swap_many (shards of 4 and 3, padded gather rounds of 2)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "launch-check", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 8 and doc["ranks_observed"] == 8 and doc["gather_ok"] and doc["ragged_swap_many_ok"]
    assert doc["triples"] == 29 and doc["shards"] == [4, 4, 4, 4, 4, 3, 3, 3] and doc["core_slices_disjoint"]
    n_cpu = len(os.sched_getaffinity(0))
    if n_cpu >= 8:
        assert all(c is not None and c[2] == n_cpu // 8 for c in doc["core_slices_first_last_count"])
