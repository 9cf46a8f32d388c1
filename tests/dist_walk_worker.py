"""Worker for the multi-rank walk tests: one rank of a torch.distributed job running the tiny pipeline's walk() into a
shared directory.  Launched by tests/test_model_gpu.py with RANK / WORLD_SIZE / MASTER_PORT / SDV_DIST_BACKEND set
(torchrun-style), every rank on cuda:0 (SDV_FORCE_DEVICE=0) - the pool has 1-GPU boxes only.

    python tests/dist_walk_worker.py OUT_DIR NAME [resume]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, parallel  # noqa: E402


def main():
    out_dir, name = sys.argv[1], sys.argv[2]
    resume = len(sys.argv) > 3 and sys.argv[3] == "resume"
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    pipe = StableDiffusionWalkPipeline.from_pretrained("tiny").to(torch.device("cuda", local))
    kw = dict(output_dir=out_dir, name=name if name != "-" else None, fps=3, num_inference_steps=3, height=64, width=64,
              make_video=False, batch_size=2)
    if resume:
        pipe.walk(resume=True, **kw)
    else:
        pipe.walk(["a cat", "a dog", "a horse"], seeds=[42, 1337, 7], num_interpolation_steps=[5, 4], **kw)
    parallel.barrier()
    if torch.distributed.is_initialized():
        print(f"rank {rank}/{world} backend {torch.distributed.get_backend()} done", flush=True)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
