"""A dependency-free H.264 (ITU-T Rec. H.264 | ISO/IEC 14496-10) intra encoder for the mp4 output of ``walk()``.

The reference writes its frames with ``torchvision.io.write_video(..., options={"crf": "10", "pix_fmt": "yuv420p"})``
(/root/reference/stable_diffusion_videos/utils.py:69-128): libx264, yuv420p, near-lossless.  ffmpeg / pyav / torchvision do not
exist in this image, so this module produces the same KIND of stream - H.264, 4:2:0, 8 bit, every picture an IDR picture - with
the one macroblock type that needs no entropy-coding tables: ``I_PCM`` (clause 7.3.5, mb_type 25 of an I slice), whose 384 sample
bytes are carried verbatim.  That makes the video track LOSSLESS in YUV (libx264 at crf 10 is merely close to it) at 1.5 bytes per
pixel (393 KB per 512 x 512 frame - UNCOMPRESSED, several times what an entropy-coded crf 10 stream takes; video.py falls back to
Motion-JPEG above H264_PCM_MAX_BYTES), and every conforming decoder - hardware ones included - plays it; what it gives up is
compression, not compatibility.

Layout of what is written (all syntax elements in the order of clauses 7.3.2.1.1, 7.3.2.2, 7.3.3, 7.3.5 and Annex E):
  * SPS: profile_idc 66 (Baseline) with constraint_set0/1 flags, level from Table A-1 by picture size / macroblock rate / bit rate,
    pic_order_cnt_type 2 (output order = decoding order), one reference frame, frame cropping for sizes that are not multiples of
    16, VUI with BT.601 limited-range colour description, the frame rate (timing_info) and a bitstream restriction of zero
    re-ordered frames.
  * PPS: CAVLC, one slice group, deblocking_filter_control_present_flag so that the slices can switch the loop filter off.
  * one IDR slice per picture (slice_type 7, frame_num 0, alternating idr_pic_id): per macroblock ``ue(25)`` + alignment zeros +
    256 luma + 64 Cb + 64 Cr samples; emulation-prevention bytes (7.4.1) over the whole NAL unit.
RGB -> Y'CbCr is the BT.601 studio-range matrix ffmpeg's swscale applies for ``pix_fmt=yuv420p``; chroma is the 2 x 2 box average.
tests/test_video.py re-parses every syntax element with an independently written reader and recovers the pictures.
"""
from __future__ import annotations

import re
import struct
from typing import List, Tuple

import numpy as np

# Table A-1 (level limits): level_idc, MaxMBPS (macroblocks / s), MaxFS (macroblocks), MaxBR (1000 bit/s units, Baseline)
_LEVELS = ((30, 40500, 1620, 10000), (31, 108000, 3600, 14000), (32, 216000, 5120, 20000), (40, 245760, 8192, 20000),
           (41, 245760, 8192, 50000), (42, 522240, 8704, 50000), (50, 589824, 22080, 135000), (51, 983040, 36864, 240000),
           (52, 2073600, 36864, 240000))


class BitWriter:
    """MSB-first bit string with the Exp-Golomb codes of clause 9.1."""

    def __init__(self):
        self.bits: List[int] = []

    def u(self, n: int, v: int):
        assert 0 <= v < (1 << n)
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def ue(self, v: int):
        assert v >= 0
        n = (v + 1).bit_length()
        self.bits.extend([0] * (n - 1))
        self.u(n, v + 1)

    def se(self, v: int):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def align_zero(self):
        self.bits.extend([0] * (-len(self.bits) % 8))

    def trailing(self):                                   # rbsp_trailing_bits(): stop bit, then zeros up to the byte boundary
        self.bits.append(1)
        self.align_zero()

    def tobytes(self) -> bytes:
        assert len(self.bits) % 8 == 0
        return np.packbits(np.asarray(self.bits, dtype=np.uint8)).tobytes()


_EPB = re.compile(rb"\x00\x00(?=[\x00-\x03])")


def nal_unit(ref_idc: int, unit_type: int, rbsp: bytes) -> bytes:
    """nal_unit() of 7.3.1: header byte + the RBSP with an emulation-prevention 0x03 behind every 00 00 that a byte <= 3 follows."""
    assert not rbsp.endswith(b"\x00")
    return bytes([(ref_idc << 5) | unit_type]) + _EPB.sub(b"\x00\x00\x03", rbsp)


def pick_level(mbs: int, fps: float) -> int:
    need_br = mbs * 386 * 8 * fps / 1000.0
    for level, max_mbps, max_fs, max_br in _LEVELS:
        if mbs <= max_fs and mbs * fps <= max_mbps and need_br <= max_br * 1.2:      # cpbBrVclFactor 1200 bit/s per unit
            return level
    return _LEVELS[-1][0]


def sps_pps(width: int, height: int, fps: float) -> Tuple[bytes, bytes]:
    """The two parameter-set NAL units (as they go into the ``avcC`` box)."""
    mbw, mbh = (width + 15) // 16, (height + 15) // 16
    w = BitWriter()
    w.u(8, 66)                      # profile_idc: Baseline
    w.u(8, 0b11000000)              # constraint_set0_flag, constraint_set1_flag (also a Main-profile stream), rest 0
    w.u(8, pick_level(mbw * mbh, fps))
    w.ue(0)                         # seq_parameter_set_id
    w.ue(0)                         # log2_max_frame_num_minus4 -> frame_num is u(4)
    w.ue(2)                         # pic_order_cnt_type 2
    w.ue(1)                         # max_num_ref_frames
    w.u(1, 0)                       # gaps_in_frame_num_value_allowed_flag
    w.ue(mbw - 1)                   # pic_width_in_mbs_minus1
    w.ue(mbh - 1)                   # pic_height_in_map_units_minus1
    w.u(1, 1)                       # frame_mbs_only_flag
    w.u(1, 1)                       # direct_8x8_inference_flag
    crop_r, crop_b = mbw * 16 - width, mbh * 16 - height
    if crop_r or crop_b:
        assert crop_r % 2 == 0 and crop_b % 2 == 0, "4:2:0 needs even picture sizes"
        w.u(1, 1)                   # frame_cropping_flag; offsets in units of 2 luma samples (CropUnitX = CropUnitY = 2)
        w.ue(0), w.ue(crop_r // 2), w.ue(0), w.ue(crop_b // 2)
    else:
        w.u(1, 0)
    w.u(1, 1)                       # vui_parameters_present_flag (Annex E.1.1)
    w.u(1, 0)                       # aspect_ratio_info_present_flag
    w.u(1, 0)                       # overscan_info_present_flag
    w.u(1, 1)                       # video_signal_type_present_flag
    w.u(3, 5)                       # video_format: unspecified
    w.u(1, 0)                       # video_full_range_flag: studio range
    w.u(1, 1)                       # colour_description_present_flag
    w.u(8, 2), w.u(8, 2), w.u(8, 6)  # primaries / transfer unspecified, matrix_coefficients 6 = BT.601 (SMPTE 170M)
    w.u(1, 0)                       # chroma_loc_info_present_flag
    w.u(1, 1)                       # timing_info_present_flag: frame rate = time_scale / (2 * num_units_in_tick)
    tick, scale = _frame_rate_fraction(fps)
    w.u(32, tick), w.u(32, 2 * scale)
    w.u(1, 1)                       # fixed_frame_rate_flag
    w.u(1, 0)                       # nal_hrd_parameters_present_flag
    w.u(1, 0)                       # vcl_hrd_parameters_present_flag
    w.u(1, 0)                       # pic_struct_present_flag
    w.u(1, 1)                       # bitstream_restriction_flag
    w.u(1, 1)                       # motion_vectors_over_pic_boundaries_flag
    w.ue(0), w.ue(0)                # max_bytes_per_pic_denom, max_bits_per_mb_denom: no limit (I_PCM exceeds the defaults' spirit)
    w.ue(16), w.ue(16)              # log2_max_mv_length_horizontal / vertical (the defaults)
    w.ue(0)                         # max_num_reorder_frames
    w.ue(1)                         # max_dec_frame_buffering
    w.trailing()
    sps = nal_unit(3, 7, w.tobytes())
    p = BitWriter()
    p.ue(0), p.ue(0)                # pic_parameter_set_id, seq_parameter_set_id
    p.u(1, 0)                       # entropy_coding_mode_flag: CAVLC
    p.u(1, 0)                       # bottom_field_pic_order_in_frame_present_flag
    p.ue(0)                         # num_slice_groups_minus1
    p.ue(0), p.ue(0)                # num_ref_idx_l0 / l1_default_active_minus1
    p.u(1, 0), p.u(2, 0)            # weighted_pred_flag, weighted_bipred_idc
    p.se(0), p.se(0), p.se(0)       # pic_init_qp_minus26, pic_init_qs_minus26, chroma_qp_index_offset
    p.u(1, 1)                       # deblocking_filter_control_present_flag
    p.u(1, 0), p.u(1, 0)            # constrained_intra_pred_flag, redundant_pic_cnt_present_flag
    p.trailing()
    return sps, nal_unit(3, 8, p.tobytes())


def _frame_rate_fraction(fps: float) -> Tuple[int, int]:
    """(num_units_in_tick, frames-per-second numerator) with fps = numerator / num_units_in_tick."""
    for den in (1, 1001, 1000):
        num = fps * den
        if abs(num - round(num)) < 1e-6:
            return den, int(round(num))
    return 1000, int(round(fps * 1000))


def rgb_to_yuv420(frame: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """uint8 RGB [H, W, 3] -> studio-range BT.601 Y' [H, W], Cb, Cr [H/2, W/2] (2 x 2 box average), uint8."""
    f = frame.astype(np.float32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    h, w = y.shape
    box = lambda c: c.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    q = lambda c: np.clip(np.rint(c), 0, 255).astype(np.uint8)
    return q(y), q(box(cb)), q(box(cr))


def idr_picture(frame: np.ndarray, index: int) -> bytes:
    """One coded picture = one IDR slice NAL unit of I_PCM macroblocks.  ``frame``: uint8 RGB [H, W, 3], H and W even."""
    assert frame.shape[0] % 2 == 0 and frame.shape[1] % 2 == 0, "4:2:0 needs even picture sizes"
    return idr_picture_yuv(*rgb_to_yuv420(frame), index)


def idr_picture_yuv(y: np.ndarray, cb: np.ndarray, cr: np.ndarray, index: int) -> bytes:
    """The same from planes: Y' [H, W], Cb / Cr [H/2, W/2], uint8, any sample values (0 included: 7.4.1 emulation prevention)."""
    h, w = y.shape
    assert cb.shape == cr.shape == (h // 2, w // 2)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    y = np.pad(y, ((0, mbh * 16 - h), (0, mbw * 16 - w)), mode="edge")            # the cropped-away border repeats the edge
    cb = np.pad(cb, ((0, mbh * 8 - h // 2), (0, mbw * 8 - w // 2)), mode="edge")
    cr = np.pad(cr, ((0, mbh * 8 - h // 2), (0, mbw * 8 - w // 2)), mode="edge")
    n = mbw * mbh
    blocks = lambda p, s: p.reshape(mbh, s, mbw, s).transpose(0, 2, 1, 3).reshape(n, s * s)   # raster scan inside each macroblock
    s = BitWriter()
    s.ue(0)                         # first_mb_in_slice
    s.ue(7)                         # slice_type: I, and every slice of the picture is
    s.ue(0)                         # pic_parameter_set_id
    s.u(4, 0)                       # frame_num (0 in an IDR picture)
    s.ue(index & 1)                 # idr_pic_id: differs between consecutive IDR pictures
    s.u(1, 0), s.u(1, 0)            # dec_ref_pic_marking(): no_output_of_prior_pics_flag, long_term_reference_flag
    s.se(0)                         # slice_qp_delta
    s.ue(1)                         # disable_deblocking_filter_idc: loop filter off
    s.ue(25)                        # macroblock 0: mb_type I_PCM ...
    s.align_zero()                  # ... pcm_alignment_zero_bit
    head = s.tobytes()
    body = np.empty((n, 2 + 384), dtype=np.uint8)
    body[:, 0], body[:, 1] = 0x0D, 0x00          # ue(25) = 000011010 + 7 alignment zeros, from a byte boundary
    body[:, 2:258] = blocks(y, 16)
    body[:, 258:322] = blocks(cb, 8)
    body[:, 322:386] = blocks(cr, 8)
    rbsp = head + body.reshape(-1)[2:].tobytes() + b"\x80"                          # rbsp_slice_trailing_bits
    return nal_unit(3, 5, rbsp)


def avcc_box_payload(sps: bytes, pps: bytes) -> bytes:
    """AVCDecoderConfigurationRecord (ISO/IEC 14496-15, 5.3.3.1) with 4-byte NAL unit lengths."""
    return (bytes([1, sps[1], sps[2], sps[3], 0xFC | 3, 0xE0 | 1]) + struct.pack(">H", len(sps)) + sps +
            bytes([1]) + struct.pack(">H", len(pps)) + pps)
