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
