#!/usr/bin/env python
"""bench.py - interpolated frames/sec of the StableDiffusionWalkPipeline hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-size B]      (N > 1: bench.py spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU over RCCL.  Under torch.distributed.run the ranks come from the environment; a bare
``python bench.py --gpus N`` (no WORLD_SIZE) fans out by itself - it spawns N copies of this script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set (HIP_VISIBLE_DEVICES untouched), rank 0 prints the one JSON line, and the
launcher exits non-zero if the node has fewer than N GPUs or any rank fails - the one-call fan-out of the reference's
multi-device path (flax_stable_diffusion_pipeline.py:568-597, :898-927).  ``n_gpus`` / ``rccl_ranks`` in the line are the size
of the process group that actually formed, never the flag.

Metric (BASELINE.json): interpolated frames/sec, 512x512, 50 DDIM steps.  Workload at N=1 = BASELINE config[1]:
SD-v1-4 architecture, bf16, 2 prompts, walk of K*B interpolated frames (default 2 x 128 = 256), CFG 7.5, eta 0.
A "step" is one pass of the hot path over one batch of B frames: lerp(text embeddings) + slerp(noise) for the
batch -> 50 x (UNet on 2B samples + fused CFG/DDIM update), each a hipGraph replay -> VAE decode -> uint8 frames
copied to the host.  Endpoint embeddings / endpoint noise are resident in HBM before the timed region; PNG
encoding is outside `value` (reported separately as `png_frames_per_sec_per_core`).  Weights are seeded synthetic
(no checkpoints offline) - the arithmetic is shape-identical.  With N > 1 every rank runs the same number of
steps on its own frames (weak scaling, no data-path collective; the weights arrive by ONE RCCL broadcast).

Extra objects in the JSON line:
  roofline     whole hot path against the dense bf16 MFMA peak: achieved = 82.84 TFLOP/frame (SURVEY.md 8d,
               algorithmic, padding not counted) x frames/s/GPU; plus `kernels`: per-kernel algorithmic
               TFLOP/s and share of GPU time from a HIP-event-bracketed eager pass (one UNet forward + one VAE
               decode of the same batch) run outside the timed region.
  attention    the UNet's 64x64-level self-attention: algorithmic / issued TFLOP/s (live) + PMC MFMA-busy share (committed profile)
  frames_per_sec_incl_png / walk_60_frames   the literal BASELINE config 2 (60 frames) through walk(), PNG files included:
               `frames_per_sec_incl_png` is the WARM walk (step graph already captured - what PNG encode + write cost), the first
               walk at that batch size is `walk_60_frames` with `cold_start_s` = what the first call paid on top
  other_configs.batch_sweep   the reference's own operating points: frames per call 1 (walk's default batch_size, :571), 4, 16
               (tests/test_pipeline.py:66; examples use 12) and 60 - first call (cold) and steady state, 50 steps each
  cpu_baseline the CPU oracle (PyTorch eager fp32 restatement of the reference path) timed on this host's cores
               on a bounded sample - 1 CFG UNet forward (2 samples) + 1 VAE decode at full size - and
               extrapolated to 50 steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

FLOP_PER_FRAME = {"sd14": 82.84e12, "sd21": 220.66e12}   # 2*(UNet MAC*2*50 + VAE MAC)
MFMA_PEAK_TFLOPS = 2500.0                                 # MI355X dense bf16 (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # frames per pipeline call = the reference's own `batch_size` argument.  128 measured 10.94 -> 11.28 frames/s over 64 (the
    # 8x8 / 16x16 UNet levels fill the 256 CUs better); 288 GB of HBM holds far more
    ap.add_argument("--batch-size", type=int, default=int(os.environ.get("SDV_BENCH_BATCH", "128")))
    ap.add_argument("--arch", default="sd14", choices=["sd14", "sd21", "tiny"])
    ap.add_argument("--size", type=int, default=0, help="image size (default: 512 for sd14, 768 for sd21)")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE.json configs[] index + 1: 2 = weak-scaled K x B frames per rank (default, the driver's "
                         "contract); 3 = the literal 8-GPU config: 4 prompts, 240 frames in total, STRONG-scaled over the ranks")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8 = BASELINE config 5: e4m3 operands in the UNet's ResBlock convs (everything else bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walk-pass", action="store_true", help="skip the 60-frame walk() pass (frames/s including PNG files)")
    ap.add_argument("--no-kernel-pass", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the self-check of the last batch (frames recomputed at batch 4)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short passes of BASELINE configs 4 (SD-2.1 768x768) and 5 (fp8) that ride in `other_configs`")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="spawn the ranks, form the process group (gloo without GPUs), all-reduce once, print the line - no model")
    return ap.parse_args()


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: spawn the N ranks (one process per GPU) and wait.
    Returns the exit code for the launcher process."""
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    forced = os.environ.get("SDV_FORCE_DEVICE")      # functional tests of the N-rank path on a 1-GPU box: every rank on that GPU
    if have < n and forced is None and not (args.launcher_selftest and have == 0):
        print(f"bench.py --gpus {n}: this node exposes {have} GPU(s) - refusing to report an {n}-GPU number from fewer "
              f"devices (set SDV_FORCE_DEVICE=<i> only to TEST the {n}-rank path on one GPU)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SDV_BENCH_RANKS_SPAWNED=str(n))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if forced is not None and have > 0:
            # every rank on ONE GPU: RCCL refuses that ("Duplicate GPU detected", measured on the 1-GPU pool), so the functional
            # test of the N-rank path runs its (host-side) collectives over gloo - the line then says dist_backend "gloo",
            # rccl_ranks 0, and is a plumbing check, not a scaling number
            env.setdefault("SDV_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:            # one rank died: the others would wait at a barrier for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc != 0:
        print(f"bench.py --gpus {n}: a rank exited with code {rc}", file=sys.stderr)
    return rc if rc >= 0 else 1


def launcher_selftest(args):
    """The rank side of ``--launcher-selftest``: process group + one all-reduce, no model (runs on CPU with gloo)."""
    import torch.distributed as dist
    from stable_diffusion_videos_amd import parallel
    rank, world, local = parallel.init_from_env()
    if os.environ.get("SDV_BENCH_SELFTEST_FAIL_RANK") == str(rank):     # (tests: one rank dies before the collective)
        raise SystemExit(3)
    t = torch.tensor([float(rank + 1)], device=torch.device("cuda", local) if world > 1 and dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        backend = dist.get_backend() if world > 1 else None
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rccl_ranks": world if backend == "nccl" else 0,
                          "dist_backend": backend, "sum_of_ranks_plus_1": float(t.item()), "flag_gpus": args.gpus}), flush=True)
    parallel.barrier()
    if world > 1:
        dist.destroy_process_group()


CONFIG3 = dict(prompts=["a cat", "a dog", "a horse", "a cow"], seeds=[42, 1337, 2022, 4321], counts=[80, 80, 80])


def config3_plan(world: int, rank: int, B: int, counts=None):
    """BASELINE.json configs[2] (4 prompts, 3 x 80 = 240 interpolated frames, frame-sharded): this rank's batches as
    ``(clip, first_frame, stop_frame)`` with ``stop - first <= B`` - its contiguous block of the flattened (clip, frame) list,
    exactly as ``walk()`` shards it (parallel.partition_frames), cut into calls of at most B frames.  Pure host logic, no rank
    is special: tests/test_dist_cpu.py runs it for world = 8 over gloo."""
    from stable_diffusion_videos_amd import parallel
    counts = list(counts or CONFIG3["counts"])
    plan = []
    for ci, a, b in parallel.partition_frames(counts, world, rank):
        for s in range(a, b, B):
            plan.append((ci, s, min(s + B, b)))
    return plan


def parity_check(pipe, embeds, noise, frames_u8, size, inference_steps):
    """Self-check of the benchmarked configuration, OUTSIDE the clock: four frames of the last timed batch are generated again in
    a 4-frame call and compared with what the timed batch returned -
    (a) with the tile selection the library makes for 8 samples.  When the timed batch WAS a 4-frame batch this is the same
        computation and must be BIT-IDENTICAL; otherwise it is a second, independent realisation of the bf16 roundings (other
        tiles -> other partial-sum groupings of the LayerNorm row statistics -> a flipped rounding early on,
        tests/test_bench_config_gpu.py) and must sit where two bf16 realisations sit, PSNR >= 37 dB;
    (b) with every igemm forced onto the 256 x 320 tile (hip.FORCE_TILE = 6) - only for chip-filling timed batches (>= 64 frames),
        which run that tile themselves: then every output element sees the same sequence of MFMA k-steps whatever the batch, the
        GroupNorm statistics are split by image size only, attention is per (sample, head), and the frames must be BIT-IDENTICAL.
        (For a small timed batch the comparison would pit the cost model's small tiles against a forced big one - two realisations.)
    (c) for a timed batch below 64 frames (no forced-tile check applies): the WHOLE batch once more, eagerly, at its own size - the
        same launches the captured step graph replays, so graph replay vs eager launches must be BIT-IDENTICAL.
    Both recomputations run eagerly (no step graph is captured for them), so (b)'s launches really are the forced tile's."""
    from stable_diffusion_videos_amd import hip
    B = embeds.shape[0]
    idx = sorted({0, max(B // 2 - 1, 0), B // 2, B - 1})
    sel = torch.tensor(idx, device=embeds.device)
    kw = dict(latents=noise[sel].contiguous(), text_embeddings=embeds[sel].contiguous(), height=size, width=size,
              num_inference_steps=inference_steps, guidance_scale=7.5, eta=0.0, output_type="numpy_u8")
    graphs, prev_tile = pipe.use_graphs, hip.FORCE_TILE
    same_batch, forced = len(idx) == B, B >= 64
    out = {"frames": idx, "vs": f"the same frames recomputed in a {len(idx)}-frame call", "min_psnr_db": 37.0}
    b = None
    try:
        pipe.use_graphs = False          # (a new batch size runs eagerly: (b) must not find a step captured with (a)'s tiles)
        a = pipe(**kw)["images"]
        if forced:
            hip.FORCE_TILE = 6
            b = pipe(**kw)["images"]
        elif not same_batch:
            kw.update(latents=noise, text_embeddings=embeds)
            out["max_abs_u8_same_batch_eager"] = int(np.abs(pipe(**kw)["images"].astype(np.int32) - frames_u8.astype(np.int32)).max())
    finally:
        pipe.use_graphs, hip.FORCE_TILE = graphs, prev_tile
    ref = frames_u8[idx].astype(np.int32)
    da = np.abs(a.astype(np.int32) - ref)
    mse = float((da.astype(np.float64) ** 2).mean())
    out.update(max_abs_u8=int(da.max()), mean_abs_u8=round(float(da.mean()), 4),
               psnr_db=round(10.0 * float(np.log10(255.0 ** 2 / max(mse, 1e-12))), 2),
               frames_differ=bool(np.abs(ref[0] - ref[-1]).mean() > 0.5))
    ok = out["frames_differ"] and (out["max_abs_u8"] == 0 if same_batch else out["psnr_db"] >= out["min_psnr_db"])
    if b is not None:
        out["max_abs_u8_same_tiles"] = int(np.abs(b.astype(np.int32) - ref).max())
        ok = ok and out["max_abs_u8_same_tiles"] == 0
    ok = ok and out.get("max_abs_u8_same_batch_eager", 0) == 0
    out["ok"] = bool(ok)
    return out


def short_pass(arch, size, B, dtype, inference_steps, steps=2):
    """A short measured pass of another BASELINE config on this GPU (own pipeline, own warm-up batch, `steps` timed batches)."""
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline
    name = {"sd14": "CompVis/stable-diffusion-v1-4", "sd21": "stabilityai/stable-diffusion-2-1"}[arch]
    pipe = StableDiffusionWalkPipeline.from_pretrained(name, arch=arch, fp8=dtype == "fp8").to(torch.device("cuda", torch.cuda.current_device()))
    h = size // 8
    T = np.linspace(0.0, 1.0, (steps + 1) * B)
    gen = pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, h, h), T, B)

    def one():
        _, e, n = next(gen)
        return pipe(latents=n, text_embeddings=e, height=size, width=size, num_inference_steps=inference_steps, guidance_scale=7.5,
                    eta=0.0, output_type="numpy_u8")["images"]

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fps = steps * B / dt
    res = {"value": round(fps, 4), "unit": "frames/s", "steps": steps, "warmup": 1, "batch_size": B, "ms_per_step": round(1e3 * dt / steps, 2),
           "workload": f"{arch} walk, 2 prompts, {size}x{size}, {inference_steps} DDIM steps, CFG 7.5, {B} frames per call",
           "dtype": "bf16" if dtype == "bf16" else "fp8 (e4m3 ResBlock convs, MX form) + bf16", "frames_shape": list(out.shape)}
    if FLOP_PER_FRAME.get(arch) and inference_steps == 50:
        ach = fps * FLOP_PER_FRAME[arch] / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "flop_per_frame": FLOP_PER_FRAME[arch]}
    del pipe, gen
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


class EventProfiler:
    """LAUNCH_HOOK that brackets every observed launch with HIP events on the launch stream."""

    def __init__(self):
        self.records = []

    def __call__(self, kind, info, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((kind, info, s, e))

    def by_shape(self):
        """Per distinct launch shape: launches, total ms, algorithmic TFLOP/s - the optimisation work-list."""
        torch.cuda.synchronize()
        agg = {}
        for kind, info, s, e in self.records:
            key = (kind,) + tuple((k, info[k]) for k in ("M", "N", "K", "batch", "mode", "epi", "B", "H", "Lq", "Lk", "dh")
                                  if k in info)
            a = agg.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["flops"] += info.get("flops", 0.0)
            a["bytes"] += info.get("bytes", 0.0)
        rows = []
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            rows.append({"kind": key[0], **dict(key[1:]), "launches": a["launches"], "ms": round(a["ms"], 4),
                         "tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 1) if a["flops"] else None,
                         "gbps": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["bytes"] else None})
        return rows

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kind, info, s, e in self.records:
            a = agg.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["flops"] += info.get("flops", 0.0)
            a["bytes"] += info.get("bytes", 0.0)
        return agg


def kernel_pass(pipe, embeds, noise, size, steps):
    """One eager UNet forward + one VAE decode of the bench batch with every hot launch timed by HIP events."""
    from stable_diffusion_videos_amd import hip
    B = embeds.shape[0]
    h = size // 8
    uncond = pipe._uncond_embeddings(None, B)
    ctx = torch.cat([uncond, embeds])
    pipe._schedule(steps, 0.0)
    pipe.unet.prepare_context(ctx)
    pipe.unet.reserve(2 * B, h, h)
    x2 = torch.zeros((2 * B * h * h, 4), dtype=torch.bfloat16, device=pipe.device)
    lat = hip.nchw_to_nhwc(noise)
    hip.latents_to_unet_input(lat, x2, True, lat.numel())
    step = torch.zeros(1, dtype=torch.int32, device=pipe.device)
    pipe.unet.forward(x2, 2 * B, h, h, step)          # warm
    torch.cuda.synchronize()
    prof_u, prof_v = EventProfiler(), EventProfiler()
    hip.LAUNCH_HOOK = prof_u
    t0 = time.perf_counter()
    pipe.unet.forward(x2, 2 * B, h, h, step)
    torch.cuda.synchronize()
    t_unet = time.perf_counter() - t0
    hip.LAUNCH_HOOK = prof_v
    pipe.vae.decode(lat * 0.18215)
    torch.cuda.synchronize()
    hip.LAUNCH_HOOK = None
    su, sv = prof_u.summary(), prof_v.summary()
    if os.environ.get("SDV_SHAPE_REPORT"):
        with open(os.environ["SDV_SHAPE_REPORT"], "w") as f:
            json.dump({"unet": prof_u.by_shape(), "vae": prof_v.by_shape()}, f, indent=1)
    out = {}
    tot_u = sum(a["ms"] for a in su.values())
    tot_v = sum(a["ms"] for a in sv.values())
    for name, s, tot in (("unet", su, tot_u), ("vae", sv, tot_v)):
        for kind, a in sorted(s.items(), key=lambda kv: -kv[1]["ms"]):
            out[f"{name}.{kind}"] = {
                "launches": a["launches"], "ms": round(a["ms"], 3), "share": round(a["ms"] / max(tot, 1e-9), 3),
                "tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 1) if a["flops"] and a["ms"] > 0 else None,
                "gbps": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["bytes"] and not a["flops"] and a["ms"] > 0 else None,
                "avg_us": round(1e3 * a["ms"] / a["launches"], 2)}
    out["unet_forward_event_ms"] = round(tot_u, 3)
    out["unet_forward_wall_ms_eager"] = round(t_unet * 1e3, 3)
    out["vae_decode_event_ms"] = round(tot_v, 3)
    out["_unet_shapes"] = prof_u.by_shape()
    return out


def csrc_fingerprint() -> str:
    """sha256 over the kernel sources + the C header + the build recipe, in name order: what a committed profile was collected FROM.  (The GPU box has
    no .git; the content hash needs none.)"""
    import hashlib
    h = hashlib.sha256()
    files = (sorted((ROOT / "stable_diffusion_videos_amd" / "csrc").glob("*.h*")) + [ROOT / "include" / "sdv_hip.h"] +
             [ROOT / "stable_diffusion_videos_amd" / "build.py"])          # (the compiler flags are part of what was measured)
    for f in files:
        h.update(f.name.encode() + b"\0" + f.read_bytes())
    return h.hexdigest()[:16]


def profile_fingerprint(path: Path) -> str:
    """The `# csrc=<fingerprint>` comment tools/pmc_summary.py / tools/rocpd_stats.py put at the top of a committed profile."""
    try:
        first = open(path).readline()
    except OSError:
        return ""
    return first.split("csrc=", 1)[1].split()[0] if first.startswith("#") and "csrc=" in first else ""


def pmc_profile(batch):
    """Counters of the committed rocprofv3 --pmc passes over one UNet forward at this bench's batch
    (profiles/round6_pmc_unet_b<batch>.csv, made by tools/pmc_summary.py from `rocprofv3 --pmc ... tools/unet_once.py <batch>`;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Returns {kernel: {counter: mean per launch}} - or {} when
    there is no profile for this batch size OR the profile was collected from other kernel sources than this tree's
    (`# csrc=` fingerprint in its first line != csrc_fingerprint()): a replayed number must describe the code that ships."""
    path = ROOT / "profiles" / f"round6_pmc_unet_b{batch}.csv"
    pmc_profile.stale = None
    if not path.exists():
        return {}
    fp = profile_fingerprint(path)
    if fp != csrc_fingerprint():
        pmc_profile.stale = (f"profiles/{path.name} was collected from kernel sources {fp or '(unrecorded)'}, this tree is "
                             f"{csrc_fingerprint()}: not replayed")
        return {}
    pmc_profile.source = f"profiles/{path.name}"
    import csv
    acc = {}
    for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")):
        a = acc.setdefault((r["kernel"], r["counter"]), [0.0, 0])
        a[0] += float(r["mean"]) * int(r["dispatches"])
        a[1] += int(r["dispatches"])
    out = {}
    for (k, c), (tot, n) in acc.items():
        out.setdefault(k, {})[c] = tot / max(n, 1)
        out[k]["_dispatches"] = max(out[k].get("_dispatches", 0), n)
    return out


def dominant_kernel_traffic(pmc):
    """HBM-side bytes per launch of the dominant kernel (the 256x320 implicit-GEMM conv), averaged over its launches in
    one UNet forward: FETCH_SIZE [KiB] x 2 (gfx950 correction) + WRITE_SIZE [KiB]."""
    # (since round 4 the tile has two bf16 conv variants - plain, and with the GroupNorm-statistics epilogue: averaged together,
    #  weighted by their launches)
    ks = [(k, c) for k, c in pmc.items() if k.startswith("igemm_kernel<4, 2, 2, 5, 64, true, 2, ") and k.rstrip(">")[-1] in "04"
          and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
    if not ks:
        return None
    n = sum(c["_dispatches"] for _, c in ks)
    fetch = sum(c["FETCH_SIZE"] * c["_dispatches"] for _, c in ks) / n
    write = sum(c["WRITE_SIZE"] * c["_dispatches"] for _, c in ks) / n
    return {"kernel": " + ".join(k for k, _ in ks), "launches_in_profile": n, "fetch_bytes": round(fetch * 2 * 1024),
            "write_bytes": round(write * 1024),
            "source": f"{getattr(pmc_profile, 'source', 'profiles/')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                      f"COMMITTED profile of the same forward from the same kernel sources - fingerprint {csrc_fingerprint()} - not "
                      "collected in this run)"}


def rocprof_conv_average(batch):
    """Average launch duration of the dominant kernel (both bf16 variants of the 256 x 320 conv) in the committed
    `rocprofv3 --kernel-trace --stats` summary of this bench command - only when that summary was collected from this tree's
    kernel sources - and the algorithmic TFLOP/s it implies, next to the live HIP-event figure."""
    path = ROOT / "profiles" / f"round6_bench_b{batch}_kernel_stats.csv"
    if not path.exists() or profile_fingerprint(path) != csrc_fingerprint():
        return None
    import csv
    calls = ns = 0
    for r in csv.DictReader(ln for ln in open(path) if not ln.startswith("#")):
        name = r.get("Name", "")
        if "igemm_kernel<4, 2, 2, 5, 64, true, 2, " in name and name.split("igemm_kernel<4, 2, 2, 5, 64, true, 2, ")[1][0] in "04":
            calls += int(r["Calls"])
            ns += float(r["TotalDurationNs"])
    if not calls:
        return None
    avg_us = ns / calls / 1e3
    # conv bucket of one forward: 200.0 GMAC per sample (every conv3x3 incl. down / up) x 2 x 2B samples over 51 launches
    tf = 200.0e9 * 2 * 2 * batch / 51 / (avg_us * 1e-6) / 1e12
    return {"avg_launch_us": round(avg_us, 2), "launches": calls, "algorithmic_tflops": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "source": f"profiles/{path.name} (rocprofv3 --kernel-trace --stats of `bench.py --steps 1` on these kernel sources, eager launches)"}


def batch_sweep(pipe, size, inference_steps, sizes=(1, 4, 16, 60)):
    """The reference's own operating points on this GPU (SURVEY.md 8d asks for the sweep; walk()'s default batch_size is 1 -
    stable_diffusion_pipeline.py:571 -, its test uses 16 - tests/test_pipeline.py:66 -, its example 12): per batch size the FIRST
    call (buffers + first eager step + graph capture inside the clock) and the steady state (best of two further calls)."""
    h = size // 8
    out = {}
    for B in sizes:
        T = np.linspace(0.0, 1.0, 3 * B)
        gen = pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, h, h), T, B)
        secs = []
        for _ in range(3):
            _, e, n = next(gen)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe(latents=n, text_embeddings=e, height=size, width=size, num_inference_steps=inference_steps, guidance_scale=7.5, eta=0.0,
                 output_type="numpy_u8")
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
        warm = min(secs[1:])
        row = {"frames_per_call": B, "frames_per_sec": round(B / warm, 4), "first_call_frames_per_sec": round(B / secs[0], 4),
               "seconds_per_call": round(warm, 4), "first_call_seconds": round(secs[0], 4)}
        if FLOP_PER_FRAME.get("sd14") and inference_steps == 50 and size == 512:
            row["frac_of_mfma_peak"] = round(B / warm * FLOP_PER_FRAME["sd14"] / 1e12 / MFMA_PEAK_TFLOPS, 4)
        out[str(B)] = row
    return out


def attention_object(shapes, pmc):
    """UNet 64x64-level self-attention (dh 40, 4096 tokens): algorithmic and issued TFLOP/s + PMC MFMA-busy share."""
    row = next((r for r in shapes if r["kind"] == "attention" and r.get("dh") == 40 and r.get("Lk") == r.get("Lq") == 4096
                and r.get("B", 0) > 64), None) or next((r for r in shapes if r["kind"] == "attention" and r.get("dh") == 40
                                                        and r.get("Lk") == 4096), None)
    if row is None:
        return None
    alg = row["tflops"]
    out = {"shape": {k: row[k] for k in ("B", "H", "Lq", "Lk", "dh")}, "algorithmic_tflops": alg,
           # the kernel pads dh 40 -> 48 for Q.K^T and -> 64 rows for P.V: (48 + 64) / (40 + 40) of the algorithmic MFMA work
           "issued_tflops": round(alg * 1.4, 1), "frac_of_mfma_peak_algorithmic": round(alg / MFMA_PEAK_TFLOPS, 3),
           "frac_of_mfma_peak_issued": round(alg * 1.4 / MFMA_PEAK_TFLOPS, 3),
           "north_star_metric": "mfma_busy_useful (PMC matrix-pipe busy share / 1.4: the padding MFMAs of dh 40 -> 48 / 64 are NOT "
                                "counted as utilisation); BASELINE.json asks for >= 0.50"}
    for k, c in pmc.items():
        if k.startswith("attention_kernel<40, 2") and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs
            out["mfma_busy_pmc"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
            # useful share of the matrix pipe: the busy counter also counts the dh 40 -> 48 / 64 padding MFMAs (x1.4)
            out["mfma_busy_useful"] = round(out["mfma_busy_pmc"] / 1.4, 3)
            out["mfma_busy_source"] = (f"{getattr(pmc_profile, 'source', 'profiles/')} (committed profile of THESE kernel sources - "
                                       f"fingerprint {csrc_fingerprint()} - not collected in this run)")
    return out


def cpu_baseline(pipe_cfgs, size, inference_steps):
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "tests"))
    from helpers import make_oracle_unet, make_oracle_vae
    from stable_diffusion_videos_amd import weights
    ucfg, vcfg = pipe_cfgs
    # PyTorch eager CPU kernels stop scaling (and regress badly) far below this host's 256 hardware threads:
    # 256 threads measured 158 s for the UNet sample vs seconds on 8-32, so the port uses at most 32 threads
    # and reports that count as `cores`.
    cores = min(os.cpu_count() or 1, int(os.environ.get("SDV_CPU_BASELINE_THREADS", "32")))
    torch.set_num_threads(cores)
    u = make_oracle_unet(ucfg, weights.synthetic_state_dict(weights.unet_shapes(ucfg), seed=0))
    v = make_oracle_vae(vcfg, weights.synthetic_state_dict(weights.vae_decoder_shapes(vcfg), seed=1))
    h = size // 8
    x = torch.randn(2, 4, h, h)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim)
    with torch.no_grad():
        u(x[:, :, :8, :8], torch.tensor(981), ctx)      # page in the weights
        t0 = time.perf_counter()
        u(x, torch.tensor(981), ctx)
        t_unet = time.perf_counter() - t0
        t0 = time.perf_counter()
        v.decode(x[:1])
        t_vae = time.perf_counter() - t0
    per_frame = inference_steps * t_unet + t_vae
    return {"value": round(1.0 / per_frame, 6), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (PyTorch eager fp32 restatement): 1 CFG UNet forward (2 samples, {t_unet:.2f} s) + 1 VAE "
                      f"decode ({t_vae:.2f} s) at {size}x{size}, extrapolated to {inference_steps} steps "
                      f"({per_frame:.1f} s/frame)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))          # this process is only the launcher
    if args.launcher_selftest:
        return launcher_selftest(args)
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE')})")
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    size = args.size or {"sd14": 512, "sd21": 768, "tiny": 128}[args.arch]
    B = args.batch_size

    pipe = StableDiffusionWalkPipeline.from_pretrained({"sd14": "CompVis/stable-diffusion-v1-4",
                                                        "sd21": "stabilityai/stable-diffusion-2-1", "tiny": "tiny"}[args.arch],
                                                       arch=args.arch, fp8=args.dtype == "fp8")
    cfgs_for_cpu = (pipe.unet.config, pipe.vae.config)
    pipe.to(dev)                                           # weight relayout (+ RCCL broadcast when world > 1)

    h = size // 8
    if args.config == 3:
        # BASELINE.json configs[2]: 4 prompts, [80, 80, 80] = 240 interpolated frames in TOTAL, frame-sharded over the ranks
        # (30 per rank on 8 GPUs): strong scaling.  Every rank takes its contiguous block of the flattened (clip, frame) list
        # exactly as walk() does and runs it in batches of at most B frames (one untimed warm-up batch first).
        prompts, seeds, counts = CONFIG3["prompts"], CONFIG3["seeds"], CONFIG3["counts"]
        work = []
        for ci, a, b in config3_plan(world, rank, B):
            T = np.linspace(0.0, 1.0, counts[ci])[a:b]
            _, embeds, noise = next(pipe.generate_inputs(prompts[ci], prompts[ci + 1], seeds[ci], seeds[ci + 1], (1, 4, h, h), T, B))
            work.append((embeds, noise))
        my_T = np.linspace(0.0, 1.0, B)

        def run(embeds, noise):
            return pipe(latents=noise, text_embeddings=embeds, height=size, width=size, num_inference_steps=args.inference_steps,
                        guidance_scale=7.5, eta=0.0, output_type="numpy_u8")["images"]

        last = run(*max(work, key=lambda w: w[0].shape[0]))       # warm-up: captures the graph of the largest batch
        torch.cuda.synchronize()
        parallel.barrier()
        t0 = time.perf_counter()
        for embeds, noise in work:
            last = run(embeds, noise)
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
        steps_done = len(work)
        frames = sum(counts)
        workload = (f"{args.arch} walk, 4 prompts, {frames} interpolated frames in total ({frames // world} per rank, batches of "
                    f"<= {B}), {size}x{size}, {args.inference_steps} DDIM steps, CFG 7.5")
        scaling = "strong"
    else:
        # the walk: 2 prompts, seeds 42/1337; rank r owns frames [r*K*B, (r+1)*K*B) of world*K*B
        total_frames = world * (args.steps + args.warmup) * B
        T_all = np.linspace(0.0, 1.0, total_frames)
        my_T = T_all[rank * (args.steps + args.warmup) * B:(rank + 1) * (args.steps + args.warmup) * B]
        gen = pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, h, h), my_T, B)

        last_in = [None]

        def one_step():
            _, embeds, noise = next(gen)
            last_in[0] = (embeds, noise)
            out = pipe(latents=noise, text_embeddings=embeds, height=size, width=size,
                       num_inference_steps=args.inference_steps, guidance_scale=7.5, eta=0.0, output_type="numpy_u8")
            return out["images"]                               # uint8 NHWC frames on the host

        for _ in range(args.warmup):
            last = one_step()
        torch.cuda.synchronize()
        parallel.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = one_step()
        torch.cuda.synchronize()
        parallel.barrier()
        elapsed = time.perf_counter() - t0
        steps_done = args.steps
        frames = world * args.steps * B
        workload = (f"{args.arch} walk, 2 prompts, {frames} interpolated frames ({args.steps} steps x {B} frames x {world} GPU; "
                    f"BASELINE config 2 names 60 frames - this is the same walk with more frames, see `walk_60_frames` for the "
                    f"literal one), {size}x{size}, {args.inference_steps} DDIM steps, CFG 7.5")
        scaling = "weak"
    rank_seconds = [round(elapsed, 4)]
    if world > 1:
        # every rank's own time over the same barrier-to-barrier region: the line's `value` uses the MAX, the list shows a
        # straggler (rank 0 does nothing the others do not inside the region; it builds the synthetic state dict before it)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        rank_seconds = [round(float(x.item()), 4) for x in allt]
        elapsed = max(float(x.item()) for x in allt)

    fps = frames / elapsed
    flop_per_frame = FLOP_PER_FRAME.get(args.arch)
    result = {
        "metric": "interpolated frames/sec (512x512, 50 DDIM steps)" if args.arch == "sd14" and size == 512 and
        args.inference_steps == 50 else f"interpolated frames/sec ({size}x{size}, {args.inference_steps} DDIM steps)",
        "value": round(fps, 4), "unit": "frames/s", "n_gpus": world,
        "rccl_ranks": world if (world > 1 and torch.distributed.get_backend() == "nccl") else (1 if world == 1 else 0),
        "dist_backend": torch.distributed.get_backend() if world > 1 else None,
        "steps": steps_done, "warmup": args.warmup, "rank_seconds": rank_seconds,
        "ms_per_step": round(1e3 * elapsed / max(steps_done, 1), 2), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "fp8 (e4m3 ResBlock convs) + bf16", "data": "synthetic (seeded random-init SD weights, hash-tokenised prompts)",
        "config": {"workload": workload, "batch_size": B, "frames": frames, "parallelism": f"frame-sharded dp{world}",
                   "hipgraph": pipe.use_graphs, "baseline_config": args.config},
    }
    if rank == 0:
        if flop_per_frame and args.inference_steps == 50:
            ach = fps / world * flop_per_frame / 1e12
            result["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                  "flop_per_frame": flop_per_frame}
        else:
            result["roofline"] = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": None, "traffic": None}
        if not args.no_kernel_pass:
            _, embeds, noise = next(pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, h, h), my_T[:B], B))
            kp = kernel_pass(pipe, embeds, noise, size, args.inference_steps)
            shapes = kp.pop("_unet_shapes")
            result["roofline"]["kernels"] = kp
            pmc = pmc_profile(B) if (args.arch == "sd14" and size == 512) else {}
            # dominant kernel = the implicit-GEMM conv (largest share of GPU time): algorithmic TFLOP/s live (HIP events),
            # HBM-side traffic per launch from the committed PMC passes of the same forward
            dom = kp.get("unet.conv3x3")
            if dom:
                result["roofline"]["dominant_kernel"] = {"name": "igemm_kernel (conv3x3, 256x320 tile)", "bound": "mfma",
                                                         "achieved": dom["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                         "frac": round(dom["tflops"] / MFMA_PEAK_TFLOPS, 4),
                                                         "avg_launch_us": dom["avg_us"], "share_of_unet_time": dom["share"]}
                tr = dominant_kernel_traffic(pmc)
                if tr:
                    result["roofline"]["traffic"] = tr["fetch_bytes"] + tr["write_bytes"]
                    result["roofline"]["traffic_source"] = tr["source"]
                    result["roofline"]["dominant_kernel"]["traffic"] = tr
                elif getattr(pmc_profile, "stale", None):
                    result["roofline"]["traffic_source"] = pmc_profile.stale
                rp = rocprof_conv_average(B)
                if rp:
                    result["roofline"]["dominant_kernel"]["rocprof"] = rp
            att = attention_object(shapes, pmc)
            if att:
                result["attention"] = att
        if world == 1 and args.config == 2 and not args.no_walk_pass:
            # the LITERAL BASELINE config 2 through the public API: walk() of 2 prompts, 60 interpolated frames, files and all
            # (lerp / slerp, 50-step loop, VAE, D2H, PNG encode + write by the asynchronous writer pool); make_video=False
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="sdv_bench_walk_")
            try:
                t1 = time.perf_counter()
                pipe.walk(["a cat", "a dog"], seeds=[42, 1337], num_interpolation_steps=60, output_dir=tmp, name="w",
                          batch_size=60, height=size, width=size, num_inference_steps=args.inference_steps, make_video=False)
                dt = time.perf_counter() - t1
                n_png = len(list(Path(tmp).rglob("frame*.png")))
                result["walk_60_frames"] = {"frames": n_png, "seconds": round(dt, 3), "batch_size": 60, "frames_per_sec": round(n_png / dt, 4),
                                            "includes": "COLD (first call at this batch size): buffers, first eager step + graph capture, "
                                                        "text encoder, interpolation, denoise, VAE, D2H, PNG encode + write"}
                # the same walk a second time: the 60-frame step graph is cached, the number a long-running service sees
                t1 = time.perf_counter()
                pipe.walk(["a cat", "a dog"], seeds=[42, 1337], num_interpolation_steps=60, output_dir=tmp, name="w2",
                          batch_size=60, height=size, width=size, num_inference_steps=args.inference_steps, make_video=False)
                dt2 = time.perf_counter() - t1
                n2 = len(list((Path(tmp) / "w2").rglob("frame*.png")))
                result["walk_60_frames_warm"] = {"frames": n2, "seconds": round(dt2, 3), "frames_per_sec": round(n2 / dt2, 4),
                                                 "batch_size": 60, "includes": "as walk_60_frames, step graph already captured"}
                # the PNG-inclusive rate of the WARM walk (step graph cached) and of the COLD one (a one-shot CLI call, which pays
                # the first call's buffers + graph capture), under keys that say which
                result["frames_per_sec_incl_png"] = round(n_png / dt, 4)              # (the key of rounds 1-4: the COLD walk)
                result["frames_per_sec_incl_png_cold"] = round(n_png / dt, 4)
                result["frames_per_sec_incl_png_warm"] = round(n2 / dt2, 4)
                result["walk_60_frames"]["cold_start_s"] = round(dt - dt2, 3)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        if args.config == 2 and not args.no_parity_check:
            result["parity_check"] = parity_check(pipe, last_in[0][0], last_in[0][1], last, size, args.inference_steps)
        # host-side PNG encode rate (outside `value`; the reference pays it serially at :553)
        from PIL import Image
        import io
        img = Image.fromarray(last[0])
        t1 = time.perf_counter()
        for _ in range(3):
            img.save(io.BytesIO(), format="PNG")
        result["png_frames_per_sec_per_core"] = round(3 / (time.perf_counter() - t1), 2)
        if world == 1 and not args.no_cpu_baseline and args.arch != "tiny":
            try:
                result["cpu_baseline"] = cpu_baseline(cfgs_for_cpu, size, args.inference_steps)
            except Exception as exc:  # the GPU number must still be reported
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {exc!r}"}
        if (world == 1 and args.config == 2 and args.arch == "sd14" and args.dtype == "bf16" and size == 512 and
                args.inference_steps == 50 and not args.no_other_configs):
            # BASELINE configs 4 and 5 on this GPU, short passes with their own warm-up (the headline `value` above stays
            # config 2 in bf16).  The bf16 pipeline's graphs and buffers go first.
            import gc
            oc = {}
            try:
                pipe._drop_graphs()          # so that every size's first call really is a cold one (60 was captured by the walk pass)
                oc["batch_sweep"] = batch_sweep(pipe, size, args.inference_steps)
            except Exception as exc:  # the headline line must still be printed
                oc["batch_sweep"] = {"error": repr(exc)}
            pipe._drop_graphs()
            del pipe, gen
            gc.collect()
            torch.cuda.empty_cache()
            for key, kw in (("sd21_768", dict(arch="sd21", size=768, B=32, dtype="bf16")),
                            ("fp8_mx", dict(arch="sd14", size=512, B=B, dtype="fp8"))):
                try:
                    oc[key] = short_pass(inference_steps=50, **kw)
                except Exception as exc:  # the headline line must still be printed
                    oc[key] = {"value": None, "error": repr(exc)}
            result["other_configs"] = oc
        print(json.dumps(result), flush=True)
        if result.get("parity_check") is not None and not result["parity_check"]["ok"]:
            parallel.barrier()
            raise SystemExit("bench.py: the parity self-check of the last batch failed: " + json.dumps(result["parity_check"]))
    parallel.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
