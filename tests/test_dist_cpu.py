"""world_size-2 gloo tests (CPU) of the multi-rank path: weight broadcast and frame sharding produce a
disjoint, complete, deterministic cover of the walk - the only two things ranks ever coordinate on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from stable_diffusion_videos_amd import config, parallel, weights
    r, ws, _ = parallel.init_from_env(backend="gloo")
    assert (r, ws) == (rank, world) and parallel.world() == (rank, world)
    shapes = weights.unet_shapes(config.tiny_unet())
    sd = weights.synthetic_state_dict(shapes, seed=0) if rank == 0 else None
    got = parallel.broadcast_state_dict(sd, shapes, "cpu")
    ref = weights.synthetic_state_dict(shapes, seed=0)
    same = all(torch.equal(got[k], ref[k]) for k in shapes)
    share = parallel.partition_frames([7, 4], world, rank)
    parallel.barrier()
    q.put((rank, same, share))
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(same for _, same, _ in res)
    frames = [(c, k) for _, _, share in res for c, a, b in share for k in range(a, b)]
    assert frames == [(0, k) for k in range(7)] + [(1, k) for k in range(4)]
    assert res[0][2] == [(0, 0, 6)] and res[1][2] == [(0, 6, 7), (1, 0, 4)]


def _plan_worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    os.environ.pop("OMP_NUM_THREADS", None)
    os.environ.pop("SDV_WRITER_THREADS", None)
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from stable_diffusion_videos_amd import parallel
    from stable_diffusion_videos_amd.utils import FrameWriter
    r, ws, _ = parallel.init_from_env(backend="gloo")
    plan = bench.config3_plan(ws, r, 32)
    # every rank learns every plan through the process group itself (what a straggler report would use)
    plans = [None] * ws
    dist.all_gather_object(plans, plan)
    w = FrameWriter()
    info = dict(rank=r, plans=plans, threads=torch.get_num_threads(), writer=w.workers, share=parallel.host_threads_per_rank())
    w.close()
    parallel.barrier()
    q.put(info)
    dist.destroy_process_group()


def test_config3_partition_and_host_threads_world8():
    """BASELINE config 3 (4 prompts, 240 frames, 8 ranks) as ``bench.py --config 3`` shards it: 30 frames per rank, calls of
    at most B frames, the union is every frame exactly once, no rank is special - and the per-rank host hygiene: each rank's PNG
    writer pool and torch CPU pool are sized from ITS share of the host cores, not from ``os.cpu_count()``
    (flax_stable_diffusion_pipeline.py:568-597 is the reference's only multi-device path)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    plans = res[0]["plans"]
    assert all(d["plans"] == plans for d in res)                      # the all-gather gave every rank the same picture
    frames = [(c, k) for plan in plans for c, a, b in plan for k in range(a, b)]
    assert frames == [(c, k) for c in range(3) for k in range(80)]    # complete, disjoint, in walk order
    assert [sum(b - a for _, a, b in plan) for plan in plans] == [30] * world
    assert all(0 < b - a <= 32 for plan in plans for _, a, b in plan)
    cores = os.cpu_count() or 4
    for d in res:
        assert d["share"] == max(1, cores // world)
        assert d["writer"] == max(2, d["share"] - 1)
        assert d["threads"] <= d["share"]
