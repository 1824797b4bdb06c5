"""The N > 1 path on CPU: channel partitioning, the ROUTE hand-off (torch.distributed P2P,
gloo here / RCCL on GPUs) and the bench timing contract, at world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phaneron_amd import multigpu as mg


def test_channel_partition_round_robin_and_blocks():
    assert [mg.channel_rank(c, 8) for c in range(8)] == list(range(8))
    assert mg.channels_of_rank(1, 8, 2) == [1, 3, 5, 7]
    # config 5: 16 channels, 2 per GPU: channel k and k+8 never share a GPU
    for k in range(16):
        assert mg.channel_rank(k, 8, 2) != mg.channel_rank((k + 8) % 16, 8, 2)
    assert mg.channels_of_rank(3, 16, 8, 2) == [6, 7]


def test_route_plan_is_consistent_between_ranks():
    routes = [mg.Route(src=(k + 8) % 16, dst=k) for k in range(16)]
    sends, recvs = {}, {}
    for r in range(8):
        p = mg.plan_routes(routes, r, 8, 2)
        assert not p.local
        for rt, peer in p.sends:
            sends.setdefault((r, peer), []).append(rt)
        for rt, peer in p.recvs:
            recvs.setdefault((peer, r), []).append(rt)
    assert sends == recvs  # same routes, same order, on both ends of every rank pair
    assert sum(len(v) for v in sends.values()) == 16
    # everything on one GPU: the reference's zero-copy case
    p = mg.plan_routes(routes, 0, 1, 0)
    assert len(p.local) == 16 and not p.sends and not p.recvs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_ch, numel = 4, 4096
        routes = [mg.Route(src=(k + 2) % 4, dst=k) for k in range(4)] + [mg.Route(src=0, dst=2)]
        mine = mg.channels_of_rank(rank, n_ch, world)
        ex = mg.RouteExchange(routes, rank, world, numel, torch.float32, "cpu")
        results = []
        for frame in range(3):
            frames = {c: torch.full((numel,), float(100 * frame + c)) for c in mine}
            got = ex.exchange(frames)
            for rt in routes:
                if mg.channel_rank(rt.dst, world) == rank:
                    # several routes can end at one channel in this test: the later one wins the dict
                    results.append((frame, rt.dst, float(got[rt.dst][0]), float(got[rt.dst][-1])))
        # timing contract: max over ranks, barrier on both sides
        import time
        el = mg.timed_steps(lambda i: time.sleep(0.01 * (rank + 1)), steps=3, warmup=1, sync=lambda: None, dist=dist,
                            device="cpu")
        q.put((rank, results, el, ex.traffic_bytes()))
    finally:
        dist.destroy_process_group()


def test_route_exchange_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_rank = {r: (res, el, tb) for r, res, el, tb in out}
    # channel c lives on rank c % 2; route (src=(k+2)%4 -> k) stays on one rank (local alias),
    # route (0 -> 2) is also local; so add a crossing check below with explicit values
    for rank, (res, el, tb) in by_rank.items():
        for frame, dst, first, last in res:
            assert first == last
            assert first in (100 * frame + (dst + 2) % 4, 100 * frame + 0)
    # both ranks report the same (max) elapsed time: rank 1 sleeps 2x longer
    assert abs(by_rank[0][1] - by_rank[1][1]) < 1e-9
    assert by_rank[0][1] >= 3 * 0.02 * 0.9


def _worker_cross(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 4 channels in blocks of 2 per rank: channel k shows channel (k + 2) % 4 -> every route crosses ranks
        routes = [mg.Route(src=(k + 2) % 4, dst=k) for k in range(4)]
        mine = mg.channels_of_rank(rank, 4, world, 2)
        ex = mg.RouteExchange(routes, rank, world, 1024, torch.float32, "cpu", channels_per_rank=2)
        frames = {c: torch.arange(1024, dtype=torch.float32) + 1000 * c for c in mine}
        got = ex.exchange(frames)
        q.put((rank, mine, {d: float(t[0]) for d, t in got.items()}, len(ex.plan.sends), len(ex.plan.recvs)))
    finally:
        dist.destroy_process_group()


def test_route_exchange_every_route_crosses_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cross, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mine, got, ns, nr in out:
        assert mine == ([0, 1] if rank == 0 else [2, 3])
        assert ns == 2 and nr == 2
        for dst, first in got.items():
            assert dst in mine and first == 1000.0 * ((dst + 2) % 4)


def test_bench_gpus_flag_plans_one_rank_per_gpu():
    """`python bench.py --gpus N` must run N ranks on its own: bare, it plans a torch.distributed.run launch of N
    processes on 127.0.0.1; under a launcher it is one rank of the world it was given; and it refuses (non-zero
    exit) when fewer than N GPUs are visible - here, none."""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}

    def plan(args, extra_env=None):
        out = subprocess.run([sys.executable, bench, "--plan"] + args, env=dict(env, **(extra_env or {})), check=True,
                             capture_output=True, text=True).stdout
        return json.loads(out.strip().splitlines()[-1])

    p = plan(["--gpus", "2", "--steps", "7"])
    assert p["mode"] == "spawn" and p["world"] == 2 and p["ranks"] == [0, 1]
    cmd = p["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "7"] and "--plan" not in cmd
    assert plan([]) == {"mode": "single", "world": 1}
    r = plan(["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "3"})
    assert r == {"mode": "rank", "world": 8, "rank": 3, "gpus_flag_matches": True}
    assert plan(["--gpus", "2"], {"WORLD_SIZE": "8", "RANK": "0"})["gpus_flag_matches"] is False
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "only 0 GPU(s) are visible" in r.stderr


def test_bench_plan_carries_every_flag_into_the_spawned_command():
    """`python bench.py --gpus N --plan` (no GPU needed): the self-spawned torch.distributed.run command is one process per
    GPU on one node, rendezvous on 127.0.0.1, and hands --steps / --warmup (and every other flag) through to the ranks"""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "37", "--warmup", "9", "--no-route", "--plan"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["mode"] == "spawn" and plan["world"] == 8 and plan["ranks"] == list(range(8))
    cmd = plan["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "37", "--warmup", "9", "--no-route"]  # --plan itself is not passed on
    # started by a launcher (WORLD_SIZE set) the same script is a rank and must agree with --gpus
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--plan"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="8", RANK="3"), timeout=120)
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan == {"mode": "rank", "world": 8, "rank": 3, "gpus_flag_matches": False}
