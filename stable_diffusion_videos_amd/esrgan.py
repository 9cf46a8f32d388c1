"""Real-ESRGAN x4 generator (RRDBNet) on the HIP kernels - SURVEY.md section 8(f) rank 3.

Replaces ``RealESRGANer.enhance`` as reached from the reference's
``RealESRGANModel.forward`` (/root/reference/stable_diffusion_videos/upsampling.py:30-54) for the case the walk
uses (:513-516, :552): 3-channel 8-bit frames, outscale = 4, tile = 0, pre_pad = 0.

Data layout in HBM (per chunk of n frames, M = n*H*W low-resolution pixels):
  * three dense-block buffers [M, 192] bf16, NHWC rows = [x (64) | x1 | x2 | x3 | x4 (32 each)]: every growth conv of
    a ResidualDenseBlock reads a column prefix of its buffer (``torch.cat`` never materialises) and writes its 32
    outputs next to it through ``ldc``; conv5 writes the next block's x (0.2 * conv + x, residual epilogue) into the
    next buffer.  Prefix widths 96 / 160 are read as 128 / 192 against zero-padded weights (K granularity 64).
  * the trunk feature, the two up-sampled tensors ([16 M, 64] at 4x) and the uint8 output frames.
Every conv is ``sdv_gemm_bf16`` in implicit-GEMM conv mode with the LeakyReLU(0.2) epilogue (epi 3); conv_first is
im2col(4 channels) + K = 64 GEMM; conv_last is the same igemm with its image epilogue (``out_mode`` 3: clamp / round / uint8).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import hip
from .config import RRDBNetConfig
from .weights import StateDict, conv_w, conv_w_c4, conv_w_kpad, vec

BF16 = torch.bfloat16
F32 = torch.float32


class RRDBNetEngine:
    def __init__(self, cfg: RRDBNetConfig, sd: StateDict, device, max_chunk_pixels: int = 4 * 512 * 512):
        hip.load()
        if cfg.scale != 4 or cfg.num_in_ch != 3 or cfg.num_feat % 64 != 0 or cfg.num_grow_ch % 8 != 0:
            raise hip.SdvHipError("RRDBNetEngine: only the scale-4 RGB generator with num_feat % 64 == 0 is supported")
        self.cfg, self.device = cfg, torch.device(device)
        self.max_chunk_pixels = max_chunk_pixels
        dev = self.device
        nf, g = cfg.num_feat, cfg.num_grow_ch
        self.width = nf + 4 * g                                   # dense-block row width (192)
        self.ld = (self.width + 63) // 64 * 64
        w_first = torch.zeros((nf, 4, 3, 3), dtype=torch.float32)
        w_first[:, :3] = sd["conv_first.weight"].float().cpu()
        self.first_w, self.first_b = conv_w_c4(w_first, dev), vec(sd["conv_first.bias"], dev)
        self.blocks = []
        for i in range(cfg.num_block):
            rdbs = []
            for r in (1, 2, 3):
                convs = []
                for k in range(1, 6):
                    p = f"body.{i}.rdb{r}.conv{k}"
                    last = k == 5
                    # conv5: out = 0.2 * (conv + b) + x  ->  alpha = 0.2 on the accumulator, bias pre-scaled
                    convs.append((conv_w_kpad(sd[p + ".weight"].float().cpu(), dev),
                                  vec(sd[p + ".bias"].float() * (0.2 if last else 1.0), dev)))
                rdbs.append(convs)
            self.blocks.append(rdbs)
        self.tail = {n: (conv_w(sd[n + ".weight"], dev), vec(sd[n + ".bias"], dev))
                     for n in ("conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last")}
        self._ws: Dict[int, Tuple[torch.Tensor, ...]] = {}

    # -------------------------------------------------------------------------------------------
    def _workspace(self, M: int):
        ws = self._ws.get(M)
        if ws is None:
            self._ws.clear()                                       # one resident chunk size at a time
            # zero-initialised ONCE: the padded K columns of the 96- / 160-wide convs must only ever see finite data
            ws = tuple(torch.zeros((M, self.ld), dtype=BF16, device=self.device) for _ in range(3))
            self._ws[M] = ws
        return ws

    def _rdb(self, src: torch.Tensor, dst_x: torch.Tensor, convs, n: int, H: int, W: int):
        """One ResidualDenseBlock: reads/extends ``src`` ([M, ld]) in place, writes 0.2*conv5 + x to ``dst_x`` ([M, nf] view)."""
        nf, g = self.cfg.num_feat, self.cfg.num_grow_ch
        for k in range(4):
            cin = nf + k * g
            w, b = convs[k]
            hip.conv3x3(src[:, : w.shape[1] // 9], w, b, nimg=n, H=H, W=W, out=src[:, cin:cin + g], epi=3)
        w, b = convs[4]
        hip.conv3x3(src[:, : w.shape[1] // 9], w, b, nimg=n, H=H, W=W, out=dst_x, residual=src[:, :nf], alpha=0.2)

    def _chunk(self, img_u8: torch.Tensor, want_float: bool):
        n, H, W, _ = img_u8.shape
        M = n * H * W
        nf = self.cfg.num_feat
        A, B, C = self._workspace(M)
        x4 = hip.rgb_u8_to_bf16_c4(img_u8)                                            # img / 255, {r, g, b, 0}
        feat = hip.conv3x3_c4(x4, self.first_w, self.first_b, nimg=n, H=H, W=W)       # [M, nf]
        hip.axpby(feat, feat, A[:, :nf], 1.0, 0.0)
        for rdbs in self.blocks:
            self._rdb(A, B[:, :nf], rdbs[0], n, H, W)
            self._rdb(B, C[:, :nf], rdbs[1], n, H, W)
            self._rdb(C, B[:, :nf], rdbs[2], n, H, W)                                 # B's x is dead by now
            hip.axpby(B[:, :nf], A[:, :nf], A[:, :nf], 0.2, 1.0)                      # RRDB: out * 0.2 + x
        w, b = self.tail["conv_body"]
        body = hip.conv3x3(A[:, :nf], w, b, nimg=n, H=H, W=W, residual=feat)          # feat + conv_body(body(feat))
        w, b = self.tail["conv_up1"]
        up1 = hip.conv3x3(body, w, b, nimg=n, H=H, W=W, mode=3, epi=3)
        w, b = self.tail["conv_up2"]
        up2 = hip.conv3x3(up1, w, b, nimg=n, H=2 * H, W=2 * W, mode=3, epi=3)
        del up1
        w, b = self.tail["conv_hr"]
        hr = hip.conv3x3(up2, w, b, nimg=n, H=4 * H, W=4 * W, epi=3)
        del up2
        w, b = self.tail["conv_last"]
        u8 = torch.empty((n, 4 * H, 4 * W, 3), dtype=torch.uint8, device=self.device)
        f32 = torch.empty((n, 4 * H, 4 * W, 3), dtype=F32, device=self.device) if want_float else None
        hip.conv3x3(hr, w, b, nimg=n, H=4 * H, W=4 * W, out_mode=3, out_f32=f32.view(-1, 3) if f32 is not None else None,
                    out_u8=u8.view(-1, 3) if u8 is not None else None)
        return u8, f32

    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, want_float: bool = False):
        """img_u8: uint8 RGB NHWC [n, H, W, 3] in GPU memory -> (uint8 RGB NHWC [n, 4H, 4W, 3], optional fp32 in [0,1])."""
        if img_u8.dtype != torch.uint8 or img_u8.ndim != 4 or img_u8.shape[-1] != 3:
            raise hip.SdvHipError("RRDBNetEngine.forward: expected uint8 [n, H, W, 3]")
        if not img_u8.is_cuda:
            raise hip.SdvHipError("RRDBNetEngine.forward: frames must live in GPU memory (no CPU fallback)")
        img_u8 = img_u8.contiguous()
        n, H, W, _ = img_u8.shape
        per = max(1, self.max_chunk_pixels // (H * W))
        outs, outs_f = [], []
        for i in range(0, n, per):
            u8, f32 = self._chunk(img_u8[i:i + per], want_float)
            outs.append(u8)
            outs_f.append(f32)
        u8 = outs[0] if len(outs) == 1 else torch.cat(outs)
        f32 = None if not want_float else (outs_f[0] if len(outs_f) == 1 else torch.cat(outs_f))
        return u8, f32

    __call__ = forward
