"""CPU oracle for the StableDiffusionWalkPipeline hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU (PyTorch eager fp32 / numpy) restatement of the reference path

    /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:412-438  (denoise loop + decode)
    /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:457-479  (generate_inputs)
    /root/reference/stable_diffusion_videos/utils.py:42-66                        (slerp)

plus the third-party arithmetic those lines dispatch into (diffusers
``UNet2DConditionModel`` / ``AutoencoderKL.decode`` / ``DDIMScheduler``), which is NOT
vendored under /root/reference (``pyproject.toml:14`` lists ``diffusers`` unpinned) and is not
installed in the build container.

Pinning status
--------------
* ``slerp`` / lerp / batching (``oracle.interp``): PINNED.  Checked bit-for-bit against the
  reference's own ``slerp`` function, lifted by AST from ``utils.py:42-66`` in the build
  container; the vectors are committed under ``tests/golden/`` together with the generating
  script ``tests/golden/make_golden.py``.
* DDIM scheduler, UNet, VAE decoder (``oracle.scheduler`` / ``oracle.models``):
  **parity unpinned**.  The reference's tests assert only that an mp4 exists
  (``tests/test_pipeline.py:50,68,81``) and diffusers cannot be imported here, so these are
  restatements of the published diffusers algorithms, anchored on the reference call sites
  (``stable_diffusion_pipeline.py:394,401,415,418,426,433``) and cross-checked by parameter
  counts (859.52 M UNet, 49.49 M VAE decoder + post_quant_conv) and diffusers state-dict keys.

Usage rule: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker.  The product package
(``stable_diffusion_videos_amd``) never imports it.
"""
