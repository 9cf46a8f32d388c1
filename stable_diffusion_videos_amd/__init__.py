"""MI355X-native drop-in for the hot path of nateraw/stable-diffusion-videos:
``StableDiffusionWalkPipeline.walk(prompts, seeds, num_interpolation_steps, ...)``.

Public names mirror /root/reference/stable_diffusion_videos/__init__.py:99-119 for the parts of the
package that are on (or directly around) the walk path.
"""
from . import ops  # noqa: F401  (registers torch.ops.sdv.*)
from .image_generation import generate_images
from .pipeline import StableDiffusionPipelineOutput, StableDiffusionWalkPipeline
from .scheduler import (DDIMScheduler, DPMSolverMultistepScheduler, EulerAncestralDiscreteScheduler, EulerDiscreteScheduler,
                        LMSDiscreteScheduler, PNDMScheduler)
from .utils import get_timesteps_arr, make_video_pyav, pad_along_axis, slerp

__version__ = "0.1.0"
__all__ = ["StableDiffusionWalkPipeline", "StableDiffusionPipelineOutput", "DDIMScheduler", "PNDMScheduler",
           "LMSDiscreteScheduler", "EulerDiscreteScheduler", "EulerAncestralDiscreteScheduler", "DPMSolverMultistepScheduler",
           "slerp", "get_timesteps_arr", "make_video_pyav", "pad_along_axis", "generate_images"]
