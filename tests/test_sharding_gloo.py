"""The N > 1 path of bench.py (one process per GPU, independent circuits, barrier + max-over-ranks timing)
exercised with world_size 2 on the gloo backend - no GPU needed."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from plonky2_goldibear_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.circuits_for_rank(8, world, rank)
    sharding.barrier()
    secs = 1.0 + rank  # rank 1 is the slow one
    agg = sharding.aggregate_throughput(len(mine), secs)
    mx = sharding.max_over_ranks(secs)
    out.put((rank, mine, agg, mx))
    sharding.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]      # disjoint, complete, i mod world
    for _, _, agg, mx in res:
        assert mx == 2.0 and agg == 8 / 2.0                           # all units / slowest rank


def test_single_process_is_identity():
    assert sharding.circuits_for_rank(5, 1, 0) == [0, 1, 2, 3, 4]
    assert sharding.max_over_ranks(1.5) == 1.5 and sharding.aggregate_throughput(3, 1.5) == 2.0
