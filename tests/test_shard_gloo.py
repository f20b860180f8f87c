"""CPU: the N>1 path of bench.py -- stream assignment and whole-job throughput -- on a world_size-2
gloo group (the data path itself has no collective: streams are independent)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from cmix_amd import shard


def test_assign_streams_silesia_over_8():
    sizes = [10192446, 51220480, 9970564, 33553445, 6152192, 10085684, 6627202, 21606400, 7251944, 41458703,
             5345280, 8474240]
    a = shard.assign_streams(sizes, 8)
    assert sorted(i for g in a for i in g) == list(range(12))
    loads = [sum(sizes[i] for i in g) for g in a]
    assert max(loads) == 51220480  # the largest file alone bounds the makespan
    assert a[0] == [1]
    assert shard.assign_streams([5, 5], 1) == [[0, 1]]


def test_shard_seeds_distinct():
    seeds = {shard.shard_seed(r, s, 4) for r in range(8) for s in range(4)}
    assert len(seeds) == 32 and min(seeds) == 1000


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0: 1000 bytes in 2 s, rank 1: 3000 bytes in 4 s -> 4000 bytes / 4 s
    res = shard.aggregate_throughput(1000 * (1 + 2 * rank), 2.0 * (1 + rank))
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_aggregate_throughput_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        total, secs, rate = got[r]
        assert total == 4000.0 and secs == 4.0 and rate == 1000.0


def test_single_process_passthrough():
    assert shard.aggregate_throughput(10, 2.0) == (10, 2.0, 5.0)
