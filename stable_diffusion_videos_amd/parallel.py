"""Frame-sharded data parallelism for the walk.

A frame is a pure function of (endpoint embeddings, endpoint noise, t, weights), so the walk is
partitioned into contiguous blocks of frames, one block per rank, one process per GPU.  The only
collectives are a one-time RCCL broadcast of the weights over xGMI and a barrier before the video is
muxed; there are NO per-step collectives.  This mirrors the reference's only multi-device strategy
(replicated params + sharded batch under ``jax.pmap``: flax_stable_diffusion_pipeline.py:568-597,
:898-927) without its padding waste, because ranks need not run in lock-step.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size) of the initialised process group, or (0, 1)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def local_world_size() -> int:
    """Ranks that share THIS host (torchrun's LOCAL_WORLD_SIZE; one node is all this path supports, so WORLD_SIZE otherwise)."""
    return max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or "1"))


def host_threads_per_rank() -> int:
    """This rank's share of the host cores: every rank of an 8-GPU node runs its own PNG writer pool and its own torch CPU
    thread pool - sized from ``os.cpu_count()`` each, 8 ranks would put 8 x (all cores) threads on one host."""
    return max(1, (os.cpu_count() or 4) // local_world_size())


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    backend "nccl" is RCCL on ROCm.  Returns (rank, world_size, local_rank)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SDV_FORCE_DEVICE") is not None:      # functional tests of the N-rank path on a 1-GPU box
        local = int(os.environ["SDV_FORCE_DEVICE"])
    # SDV_DIST_INIT=1 builds a process group even for ONE rank: the RCCL broadcast path then runs on a 1-GPU box
    if (ws > 1 or os.environ.get("SDV_DIST_INIT") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("SDV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if ws > 1 and "OMP_NUM_THREADS" not in os.environ:
            # one CPU thread pool per rank, each its share of the host - not N pools of cpu_count() threads
            torch.set_num_threads(min(torch.get_num_threads(), host_threads_per_rank()))
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=ws, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws, local


def partition_frames(frame_counts: Sequence[int], world_size: int, rank: int,
                     skips: Sequence[int] = None) -> List[Tuple[int, int, int]]:
    """Split the walk's frames into ``world_size`` contiguous blocks and return rank's share as a list of
    ``(clip_index, first_frame, stop_frame)``.

    ``frame_counts[i]`` is clip i's ``num_interpolation_steps``; ``skips[i]`` the number of leading frames
    already on disk (resume).  Blocks are contiguous in (clip, frame) order so a rank keeps its clip
    endpoints hot and works in full batches; sizes differ by at most one frame."""
    skips = list(skips) if skips is not None else [0] * len(frame_counts)
    return partition_frame_list([list(range(skips[i], n)) for i, n in enumerate(frame_counts)], world_size, rank)


def partition_frame_list(todo_per_clip: Sequence[Sequence[int]], world_size: int, rank: int) -> List[Tuple[int, int, int]]:
    """As ``partition_frames`` for an ARBITRARY set of frames per clip (``todo_per_clip[i]`` = sorted frame indices of
    clip i that still have to be generated).  A resumed multi-rank (or hard-killed) run leaves holes, not a prefix: rank 0
    may have died at frame 20 of its block and rank 1 at frame 70 of its own, so "continue after the last frame on disk"
    (the reference's rule, which assumes one sequential writer) would never produce frames 21-49.  The share is returned
    as runs of consecutive frames ``(clip, first, stop)``."""
    todo = [(i, k) for i, frames in enumerate(todo_per_clip) for k in frames]
    total = len(todo)
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    mine = todo[start:stop]
    out: List[Tuple[int, int, int]] = []
    for clip, k in mine:
        if out and out[-1][0] == clip and out[-1][2] == k:
            out[-1] = (clip, out[-1][1], k + 1)
        else:
            out.append((clip, k, k + 1))
    return out


def broadcast_object(obj, src: int = 0):
    """Rank ``src``'s picklable object on every rank (run name, resume decisions)."""
    rank, ws = world()
    if ws == 1:
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_state_dict(sd: Dict[str, torch.Tensor], shapes: Dict[str, tuple], device, src: int = 0
                         ) -> Dict[str, torch.Tensor]:
    """One-time weight distribution: rank ``src`` holds ``sd``; everyone returns an identical dict.

    All parameters travel as ONE packed fp32 buffer (3.4 GB for the SD-1.4 UNet - a single large broadcast, which is what a
    point-to-point xGMI fabric wants, instead of ~700 small ones).  fp32 MASTERS, not bf16: the engines fold LayerNorm gains
    into weights, sum the up-convs' coincident taps and quantise to e4m3 from what they are given, and each of those must
    round ONCE from the checkpoint's values - with a bf16 transport a real fp32 / fp16 checkpoint gave an N-rank run that
    differed from the 1-rank run.  Every entry starts on a 16-byte boundary.  The matrices are returned as views
    of the packed buffer (the engines convert / re-lay them out into their own bf16 storage, after which the caller drops the
    dict and the buffer is freed); the small vectors are cloned so that nothing the engines keep pins the buffer."""
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return sd
    # (a 1-rank process group still takes the broadcast path: that is how the RCCL code is exercised on a 1-GPU box)
    use_gpu = dist.get_backend() == "nccl"
    dev = torch.device(device) if use_gpu else torch.device("cpu")
    offs, o = {}, 0
    for k, shp in shapes.items():
        n = 1
        for d in shp:
            n *= int(d)
        offs[k] = (o, n)
        o += (n + 3) // 4 * 4                  # 4 floats = 16 bytes
    buf = torch.empty(o, dtype=torch.float32, device=dev)
    if rank == src:
        for k, (a, n) in offs.items():
            buf[a:a + n] = sd[k].reshape(-1).to(dev, torch.float32)
    dist.broadcast(buf, src=src)
    out: Dict[str, torch.Tensor] = {}
    for k, (a, n) in offs.items():
        v = buf[a:a + n].view(*shapes[k])
        out[k] = v.clone() if len(shapes[k]) == 1 or n <= 4096 else v
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
