#!/usr/bin/env python
"""Audio-driven walk - the shape of the reference's examples/make_music_video.py on the MI355X path: the interpolation
schedule T follows the percussive energy of the audio slice between consecutive offsets (audio.get_timesteps_arr).

    python examples/make_music_video.py tests/samples/choice.wav

Differences from the reference script: DDIM is the scheduler this path implements (the reference example swaps in
LMSDiscreteScheduler), bf16 instead of fp16, and no xformers switch - the flash-attention kernel is always on.
"""
import random
import sys

from stable_diffusion_videos_amd import StableDiffusionWalkPipeline

audio_filepath = sys.argv[1] if len(sys.argv) > 1 else "tests/samples/choice.wav"
pipe = StableDiffusionWalkPipeline.from_pretrained("runwayml/stable-diffusion-v1-5", safety_checker=None).to("cuda")

audio_offsets = [0, 2, 4]                        # seconds into the song, one per prompt
fps = 25
num_interpolation_steps = [(b - a) * fps for a, b in zip(audio_offsets, audio_offsets[1:])]
prompts = ["a cat with a funny hat", "snoop dogg at the dmv", "steak flavored ice cream"]
seeds = [random.randint(0, int(9e9)) for _ in prompts]

print(pipe.walk(prompts=prompts, seeds=seeds, num_interpolation_steps=num_interpolation_steps, fps=fps,
                audio_filepath=audio_filepath, audio_start_sec=audio_offsets[0], batch_size=50, num_inference_steps=50,
                guidance_scale=15, margin=1.0, smooth=0.2))
