"""The N > 1 path of bench.py (one process per GPU, independent circuits, barrier + max-over-ranks timing)
exercised with world_size 2 on the gloo backend - no GPU needed."""
import json
import os
import socket
import subprocess
import sys

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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench_parent(extra_env, *args, gpus=2):
    env = dict(os.environ, GB_BENCH_STUB="1", GB_BENCH_BACKEND="gloo", GB_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", *args],
                          env=env, capture_output=True, text=True, timeout=240)


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the parent starts two rank processes, relays rank 0's
    line and exits 0 - the launch path of the driver's multi-GPU run, with the GPU work stubbed out."""
    r = _run_bench_parent({})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["stub"] is True
    assert j["ms_per_step"] >= 4.0                      # the slower rank (2 x 2 ms per step) sets the time: max over ranks
    assert abs(j["value"] - 2 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    assert "rank-local 0" in r.stderr and "rank-local 1" in r.stderr   # each rank printed its CPU binding (or that it has none)
    ranks = j["affinity"]["ranks"]                                      # and the line carries both
    assert len(ranks) == 2 and all(a["cpus"] >= 1 for a in ranks)
    if all(a["bound"] for a in ranks):
        assert ranks[0]["last"] < ranks[1]["first"] or ranks[1]["last"] < ranks[0]["first"]
    assert j["control_plane"] == {"backend": "gloo", "requested": "gloo"}


def test_eight_ranks_get_disjoint_cpu_sets():
    """the launch path at the node's size: `python bench.py --gpus 8` (stub), eight ranks, eight disjoint CPU sets carved out of
    this process's affinity mask (16 CPUs on a GPU box: two each; fewer than eight here: not bound, and the line says so)"""
    avail = len(os.sched_getaffinity(0))
    r = _run_bench_parent({}, gpus=8)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r)
    ranks = j["affinity"]["ranks"]
    assert j["n_gpus"] == 8 and len(ranks) == 8
    if avail >= 8:
        assert all(a["bound"] for a in ranks)
        spans = sorted((a["first"], a["last"]) for a in ranks)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans        # pairwise disjoint
        assert sum(a["cpus"] for a in ranks) <= avail
    else:
        assert not any(a["bound"] for a in ranks)


def test_control_plane_falls_back_to_gloo_when_rccl_cannot_come_up():
    """GB_BENCH_BACKEND unset = nccl requested; on a box without GPUs RCCL cannot come up: the run goes on over gloo and says so"""
    r = _run_bench_parent({"GB_BENCH_BACKEND": "nccl"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r)
    cp = j["control_plane"]
    assert cp["backend"] == "gloo" and cp["requested"] == "nccl" and cp["fallback_reason"] and cp["failed_at"] == "precondition"
    assert "RCCL control plane unavailable" in r.stderr


def test_control_plane_survives_one_rank_failing_rccl():
    """RCCL fails on ONE rank while its peer is already inside the collective (a gloo group stands in for the RCCL one:
    GB_BENCH_FAKE_RCCL; rank 1 raises instead of entering the probe all-reduce).  The peer's bring-up runs under a bounded wait, the
    outcome is agreed over gloo, and the run goes on over gloo - one line, exit 0, the reason names the rank."""
    r = _run_bench_parent({"GB_BENCH_BACKEND": "nccl", "GB_BENCH_FAKE_RCCL": "1", "GB_BENCH_RCCL_FAIL_RANK": "1", "GB_BENCH_RCCL_TIMEOUT": "6"})
    assert r.returncode == 0, r.stderr[-2000:]
    cp = _line(r)["control_plane"]
    assert cp["backend"] == "gloo" and cp["requested"] == "nccl" and cp["failed_at"] == "bring-up"
    assert "rank 1: RuntimeError: RCCL bring-up failed on this rank" in cp["fallback_reason"]
    assert "RCCL control plane unavailable" in r.stderr


def test_control_plane_second_group_carries_the_barrier_when_it_comes_up():
    """the branch in which the second group DID come up on every rank (here a gloo group in RCCL's place): the barrier and the
    max-over-ranks of the timed region run over it"""
    r = _run_bench_parent({"GB_BENCH_BACKEND": "nccl", "GB_BENCH_FAKE_RCCL": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r)
    assert j["control_plane"]["requested"] == "nccl" and "standing in for RCCL" in j["control_plane"]["backend"]
    assert j["ms_per_step"] >= 4.0 and "RCCL control plane unavailable" not in r.stderr


def test_bench_parent_fails_when_a_rank_fails():
    r = _run_bench_parent({"GB_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_mismatched_world_size():
    env = dict(os.environ, GB_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
