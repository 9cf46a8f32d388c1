"""Real-ESRGAN x4 upsampling (reference: stable_diffusion_videos/upsampling.py:13-99) is SURVEY.md section 8(f)
rank 3 - outside the hot path built this round.  ``walk(upsample=True)`` therefore fails loudly instead of
silently returning 512x512 frames."""


class RealESRGANModel:
    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        raise NotImplementedError(
            "upsample=True (Real-ESRGAN, reference upsampling.py) is not part of the MI355X hot path yet "
            "(SURVEY.md section 8f rank 3); run walk(..., upsample=False)")
