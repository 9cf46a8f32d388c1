"""mp4 muxing (reference utils.py:69-128).  Without ffmpeg the built-in writer emits an ISO-BMFF
file with a Motion-JPEG track; the test re-parses the container and decodes the samples back."""
import io
import struct

import numpy as np
import pytest
import torch
from PIL import Image

from stable_diffusion_videos_amd import make_video_pyav


def parse_boxes(buf, start=0, end=None):
    end = len(buf) if end is None else end
    out, pos = [], start
    while pos < end:
        size, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        head = 8
        if size == 1:                                                     # 64-bit largesize follows the type
            size, head = struct.unpack(">Q", buf[pos + 8:pos + 16])[0], 16
        out.append((kind.decode("latin1"), pos + head, pos + size))
        pos += size
    return out


def find(buf, path):
    lo, hi = 0, len(buf)
    for name in path:
        skip = 8 if name == "stsd_entry" else 0
        boxes = parse_boxes(buf, lo, hi)
        kind, lo, hi = next(b for b in boxes if b[0] == name)
        lo += skip
    return lo, hi


def test_mjpeg_mp4_round_trip(tmp_path, monkeypatch):
    monkeypatch.setenv("SDV_VIDEO_CODEC", "mjpeg")
    rng = np.random.default_rng(0)
    base = np.kron(rng.integers(0, 255, (8, 8, 3)), np.ones((8, 8, 1))).astype(np.uint8)      # 64x64 blocky image
    frames = [np.roll(base, 4 * k, axis=1) for k in range(5)]
    d = tmp_path / "clip"
    d.mkdir()
    for k, fr in enumerate(frames):
        Image.fromarray(fr).save(d / f"frame{k:06d}.png")
    out = make_video_pyav(d, fps=5, output_filepath=tmp_path / "clip" / "clip.mp4", glob_pattern="*.png")
    buf = open(out, "rb").read()
    top = parse_boxes(buf)
    assert [b[0] for b in top] == ["ftyp", "mdat", "moov"]
    stbl = ("moov", "trak", "mdia", "minf", "stbl")
    lo, hi = find(buf, stbl + ("stsz",))
    _, _, n = struct.unpack(">III", buf[lo:lo + 12])
    sizes = struct.unpack(f">{n}I", buf[lo + 12:lo + 12 + 4 * n])
    assert n == 5
    lo, hi = find(buf, stbl + ("stco",))
    offset = struct.unpack(">III", buf[lo:lo + 12])[2]
    lo, hi = find(buf, stbl + ("stts",))
    _, entries, count, delta = struct.unpack(">IIII", buf[lo:lo + 16])
    assert (entries, count, delta) == (1, 5, 18000)                     # 90 kHz timescale / 5 fps
    lo, hi = find(buf, ("moov", "trak", "tkhd"))
    w, h = struct.unpack(">II", buf[hi - 8:hi])
    assert (w >> 16, h >> 16) == (64, 64)
    pos = offset
    for k, sz in enumerate(sizes):                                       # every sample is a decodable JPEG of the frame
        im = np.asarray(Image.open(io.BytesIO(buf[pos:pos + sz])).convert("RGB")).astype(int)
        assert im.shape == (64, 64, 3) and np.abs(im - frames[k].astype(int)).mean() < 15
        pos += sz
    # tensor input form (T, C, H, W), as the reference accepts
    t = torch.from_numpy(np.stack(frames)).permute(0, 3, 1, 2)
    out2 = make_video_pyav(t, fps=5, output_filepath=tmp_path / "t.mp4")
    assert abs(len(open(out2, "rb").read()) - len(buf)) < 64


def traks(buf):
    lo, hi = find(buf, ("moov",))
    return [(a, b) for kind, a, b in parse_boxes(buf, lo, hi) if kind == "trak"]


def find_in(buf, lo, hi, path):
    for name in path:
        kind, lo, hi = next(b for b in parse_boxes(buf, lo, hi) if b[0] == name)
    return lo, hi


@pytest.mark.parametrize("codec", ["mjpeg", "h264"])
def test_mp4_carries_the_audio_window(tmp_path, monkeypatch, codec):
    """make_video_pyav(audio_filepath=..., audio_offset, audio_duration) (utils.py:69-128, called at
    stable_diffusion_pipeline.py:786-807): the dependency-free writer adds a 16-bit PCM track holding exactly the
    [offset, offset + duration) window of the file at ``sr`` - re-parsed here and compared sample by sample."""
    from pathlib import Path
    from stable_diffusion_videos_amd.audio import load_audio
    monkeypatch.setenv("SDV_VIDEO_CODEC", codec)
    wav = Path(__file__).parent / "samples" / "choice.wav"
    frames = torch.randint(0, 255, (6, 3, 32, 48), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    sr, off, dur, fps = 44100, 1.5, 2.0, 3
    out = make_video_pyav(frames, audio_filepath=str(wav), fps=fps, audio_offset=off, audio_duration=dur, sr=sr,
                          output_filepath=tmp_path / "a.mp4")
    buf = open(out, "rb").read()
    tk = traks(buf)
    assert len(tk) == 2
    # movie header: 90 kHz timescale, duration = the longer track (6 frames at 3 fps = 2 s = the audio window)
    lo, hi = find(buf, ("moov", "mvhd"))
    _, _, _, timescale, duration = struct.unpack(">IIIII", buf[lo:lo + 20])
    assert timescale == 90000 and duration == 180000
    assert struct.unpack(">I", buf[hi - 4:hi])[0] == 3                    # next_track_ID
    # handler types
    kinds = []
    for a, b in tk:
        lo, hi = find_in(buf, a, b, ("mdia", "hdlr"))
        kinds.append(buf[lo + 8:lo + 12])
    assert kinds == [b"vide", b"soun"]
    a, b = tk[1]
    lo, hi = find_in(buf, a, b, ("mdia", "mdhd"))
    _, _, _, a_timescale, a_duration = struct.unpack(">IIIII", buf[lo:lo + 20])
    assert a_timescale == sr and a_duration == int(sr * dur)
    stbl = ("mdia", "minf", "stbl")
    lo, hi = find_in(buf, a, b, stbl + ("stsd",))
    entry = parse_boxes(buf, lo + 8, hi)[0]
    assert entry[0] == "ipcm"
    channels, bits = struct.unpack(">HH", buf[entry[1] + 16:entry[1] + 20])
    assert (channels, bits) == (1, 16)
    assert struct.unpack(">I", buf[entry[1] + 24:entry[1] + 28])[0] == sr << 16
    sub = {k: (x, y) for k, x, y in parse_boxes(buf, entry[1] + 28, entry[2])}
    assert set(sub) == {"pcmC", "chnl"} and buf[sub["pcmC"][0] + 4:sub["pcmC"][0] + 6] == bytes([1, 16])
    lo, hi = find_in(buf, a, b, stbl + ("stsz",))
    _, sample_size, count = struct.unpack(">III", buf[lo:lo + 12])
    assert (sample_size, count) == (2, int(sr * dur))
    lo, hi = find_in(buf, a, b, stbl + ("stco",))
    offset = struct.unpack(">III", buf[lo:lo + 12])[2]
    mdat = next(x for x in parse_boxes(buf) if x[0] == "mdat")
    assert mdat[1] < offset and offset + 2 * count == mdat[2]            # the PCM samples close the mdat box
    pcm = np.frombuffer(buf[offset:offset + 2 * count], dtype="<i2").astype(np.float64) / 32767.0
    ref, _ = load_audio(wav, sr=sr, mono=True, offset=off, duration=dur)
    assert ref.shape[0] == count and np.abs(pcm - np.clip(ref, -1, 1)).max() <= 1.0 / 32767.0
    assert np.abs(pcm).max() > 0.05                                      # (the window is not silence)
    # the video samples still start where the video track says
    va, vb = tk[0]
    lo, hi = find_in(buf, va, vb, stbl + ("stco",))
    v_off = struct.unpack(">III", buf[lo:lo + 12])[2]
    if codec == "mjpeg":
        assert buf[v_off:v_off + 2] == b"\xff\xd8"
    else:
        assert buf[v_off + 4] == 0x65                                     # length prefix, then an IDR slice NAL unit


def test_h264_aac_path_when_pyav_is_present(tmp_path):
    """The reference's encoding (utils.py:69-128: ``torchvision.io.write_video(..., options={"crf": "10", "pix_fmt": "yuv420p"},
    audio_codec="aac")``) is taken whenever torchvision + pyav are importable; neither is in this image, so this test skips
    here and runs wherever they exist: the file must then carry an H.264 video track (sample entry ``avc1``) instead of the
    Motion-JPEG fallback's ``mp4v``."""
    pytest.importorskip("av")
    pytest.importorskip("torchvision")
    from PIL import Image
    d = tmp_path / "frames"
    d.mkdir()
    rng = np.random.RandomState(0)
    for k in range(6):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / f"frame{k:06d}.png")
    out = make_video_pyav(d, fps=6, output_filepath=tmp_path / "h264.mp4", glob_pattern="*.png")
    data = open(out, "rb").read()
    assert b"avc1" in data and b"mp4v" not in data


# ---- an H.264 reader written from the syntax tables of the Recommendation (independent of h264.py) ---------------------------
class Bits:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def aligned(self):
        return self.p % 8 == 0

    def trailing_ok(self):
        """rbsp_trailing_bits: a one, then zeros to the end of the data."""
        if self.u(1) != 1:
            return False
        while self.p % 8:
            if self.u(1):
                return False
        return self.p == 8 * len(self.d)


def unescape(nal: bytes) -> bytes:
    """7.4.1: drop every emulation_prevention_three_byte; assert the forbidden patterns never occur in the NAL unit."""
    out, zeros, i = bytearray(), 0, 0
    while i < len(nal):
        b = nal[i]
        if zeros >= 2:
            assert b >= 3, "00 00 followed by 00 / 01 / 02 inside a NAL unit"
            if b == 3:
                assert i + 1 == len(nal) or nal[i + 1] <= 3, "an 03 after 00 00 must be an emulation prevention byte"
                zeros, i = 0, i + 1
                continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
        i += 1
    return bytes(out)


def read_sps(nal):
    assert nal[0] == 0x67                                                  # forbidden_zero_bit 0, nal_ref_idc 3, type 7
    r = Bits(unescape(nal[1:]))
    s = dict(profile=r.u(8), constraints=r.u(8), level=r.u(8), sps_id=r.ue())
    assert s["profile"] == 66                                              # (no chroma_format_idc branch below 100)
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    assert s["poc_type"] == 2                                              # types 0 / 1 carry more syntax
    s["num_ref_frames"], s["gaps"] = r.ue(), r.u(1)
    s["mbw"], s["mbh"] = r.ue() + 1, r.ue() + 1
    s["frame_mbs_only"] = r.u(1)
    assert s["frame_mbs_only"] == 1
    s["direct_8x8"] = r.u(1)
    s["crop"] = (r.ue(), r.ue(), r.ue(), r.ue()) if r.u(1) else (0, 0, 0, 0)
    if r.u(1):                                                             # vui_parameters()
        if r.u(1):
            if r.u(8) == 255:
                r.u(32)
        if r.u(1):
            r.u(1)
        if r.u(1):
            s["video_format"], s["full_range"] = r.u(3), r.u(1)
            if r.u(1):
                s["colour"] = (r.u(8), r.u(8), r.u(8))
        if r.u(1):
            r.ue(), r.ue()
        if r.u(1):
            s["num_units_in_tick"], s["time_scale"], s["fixed_frame_rate"] = r.u(32), r.u(32), r.u(1)
        nal_hrd, vcl_hrd = r.u(1), None
        assert not nal_hrd
        vcl_hrd = r.u(1)
        assert not vcl_hrd
        s["pic_struct_present"] = r.u(1)
        if r.u(1):
            s["restriction"] = (r.u(1), r.ue(), r.ue(), r.ue(), r.ue(), r.ue(), r.ue())
    assert r.trailing_ok()
    return s


def read_pps(nal):
    assert nal[0] == 0x68
    r = Bits(unescape(nal[1:]))
    p = dict(pps_id=r.ue(), sps_id=r.ue(), cabac=r.u(1), bottom_field_poc=r.u(1), slice_groups=r.ue() + 1)
    assert p["slice_groups"] == 1
    p["ref_l0"], p["ref_l1"] = r.ue(), r.ue()
    p["weighted"], p["bipred"] = r.u(1), r.u(2)
    p["qp"], p["qs"], p["chroma_qp_offset"] = 26 + r.se(), 26 + r.se(), r.se()
    p["deblock_control"], p["constrained_intra"], p["redundant_pic_cnt"] = r.u(1), r.u(1), r.u(1)
    assert r.trailing_ok()
    return p


def read_idr_picture(nal, sps, pps):
    """slice_layer_without_partitioning_rbsp() of an IDR picture whose macroblocks are all I_PCM -> (Y, Cb, Cr, header)."""
    assert nal[0] == 0x65                                                  # nal_ref_idc 3, nal_unit_type 5
    r = Bits(unescape(nal[1:]))
    hdr = dict(first_mb=r.ue(), slice_type=r.ue(), pps_id=r.ue(), frame_num=r.u(sps["log2_max_frame_num"]), idr_pic_id=r.ue())
    assert hdr["slice_type"] in (2, 7) and hdr["first_mb"] == 0 and hdr["frame_num"] == 0
    assert not pps["redundant_pic_cnt"]
    hdr["no_output_of_prior_pics"], hdr["long_term_reference"] = r.u(1), r.u(1)      # dec_ref_pic_marking() of an IDR picture
    assert not pps["cabac"]
    hdr["qp"] = pps["qp"] + r.se()
    if pps["deblock_control"]:
        hdr["disable_deblocking"] = r.ue()
        if hdr["disable_deblocking"] != 1:
            r.se(), r.se()
    mbw, mbh = sps["mbw"], sps["mbh"]
    Y = np.zeros((mbh * 16, mbw * 16), np.uint8)
    Cb, Cr = np.zeros((mbh * 8, mbw * 8), np.uint8), np.zeros((mbh * 8, mbw * 8), np.uint8)
    data = r.d
    for mb in range(mbw * mbh):                                            # macroblock_layer(), raster order (one slice group)
        assert r.ue() == 25, "I slice mb_type 25 = I_PCM"
        while not r.aligned():
            assert r.u(1) == 0                                             # pcm_alignment_zero_bit
        o = r.p // 8
        px = np.frombuffer(data[o:o + 384], np.uint8)
        r.p += 384 * 8
        my, mx = divmod(mb, mbw)
        Y[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] = px[:256].reshape(16, 16)
        Cb[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] = px[256:320].reshape(8, 8)
        Cr[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] = px[320:384].reshape(8, 8)
    assert r.trailing_ok()                                                 # no more_rbsp_data(): the slice ends here
    cl, cr_, ct, cb_ = sps["crop"]
    h, w = mbh * 16 - 2 * cb_, mbw * 16 - 2 * cr_
    assert cl == ct == 0
    return Y[:h, :w], Cb[:h // 2, :w // 2], Cr[:h // 2, :w // 2], hdr


def yuv_to_rgb(Y, Cb, Cr):
    y = (Y.astype(np.float64) - 16.0) * (255.0 / 219.0)
    cb = np.kron(Cb.astype(np.float64) - 128.0, np.ones((2, 2))) * (255.0 / 224.0)
    cr = np.kron(Cr.astype(np.float64) - 128.0, np.ones((2, 2))) * (255.0 / 224.0)
    return np.stack([y + 1.402 * cr, y - 0.344136 * cb - 0.714136 * cr, y + 1.772 * cb], axis=-1)


@pytest.mark.parametrize("size", [(64, 64), (36, 50), (512, 512)])
def test_h264_mp4_is_decodable_from_the_syntax_tables(tmp_path, size):
    """The default writer: ISO-BMFF with an ``avc1`` track of I_PCM IDR pictures (h264.py).  No H.264 decoder exists in this
    image, so the stream is read back by a reader written from the Recommendation's syntax tables (7.3.1 NAL unit + emulation
    prevention, 7.3.2.1.1 SPS + Annex E VUI, 7.3.2.2 PPS, 7.3.3 slice header, 7.3.5 macroblock layer): every element has the
    value an all-intra Constrained-Baseline stream needs, every picture comes back sample-exact in Y'CbCr and within the
    4:2:0 / studio-range rounding in RGB; sizes that are not multiples of 16 go through frame cropping."""
    from stable_diffusion_videos_amd import h264
    H, W = size
    rng = np.random.default_rng(1)
    base = np.kron(rng.integers(0, 256, ((H + 7) // 8, (W + 7) // 8, 3)), np.ones((8, 8, 1)))[:H, :W].astype(np.uint8)   # chroma-friendly
    frames = [np.roll(base, 8 * k, axis=1) for k in range(3)]
    frames[1] = frames[1].copy()
    frames[1][:8, :8] = 0                                                 # pure black and pure white blocks
    frames[1][8:16, :8] = 255
    t = torch.from_numpy(np.stack(frames)).permute(0, 3, 1, 2)
    out = make_video_pyav(t, fps=30, output_filepath=tmp_path / "v.mp4")
    buf = open(out, "rb").read()
    assert [b[0] for b in parse_boxes(buf)] == ["ftyp", "mdat", "moov"]
    assert b"avc1" in buf[8:32]                                            # a compatible brand
    stbl = ("moov", "trak", "mdia", "minf", "stbl")
    lo, hi = find(buf, stbl + ("stsd",))
    entry = parse_boxes(buf, lo + 8, hi)[0]
    assert entry[0] == "avc1"
    assert struct.unpack(">HH", buf[entry[1] + 24:entry[1] + 28]) == (W, H)
    avcc = parse_boxes(buf, entry[1] + 78, entry[2])[0]
    assert avcc[0] == "avcC"
    rec = buf[avcc[1]:avcc[2]]
    assert rec[0] == 1 and rec[4] == 0xFF and rec[5] == 0xE1              # version 1, 4-byte NAL lengths, one SPS
    n_sps = struct.unpack(">H", rec[6:8])[0]
    sps_nal = rec[8:8 + n_sps]
    assert rec[8 + n_sps] == 1
    n_pps = struct.unpack(">H", rec[9 + n_sps:11 + n_sps])[0]
    pps_nal = rec[11 + n_sps:11 + n_sps + n_pps]
    assert 11 + n_sps + n_pps == len(rec)
    assert rec[1:4] == sps_nal[1:4]                                        # profile / compatibility / level copied from the SPS
    sps, pps = read_sps(sps_nal), read_pps(pps_nal)
    assert sps["constraints"] & 0xC0 == 0xC0 and sps["num_ref_frames"] == 1 and sps["gaps"] == 0
    assert (sps["mbw"], sps["mbh"]) == ((W + 15) // 16, (H + 15) // 16)
    assert sps["crop"] == (0, (sps["mbw"] * 16 - W) // 2, 0, (sps["mbh"] * 16 - H) // 2)
    assert sps["colour"] == (2, 2, 6) and sps["full_range"] == 0
    assert sps["time_scale"] / (2 * sps["num_units_in_tick"]) == 30 and sps["fixed_frame_rate"] == 1
    assert sps["restriction"][5:] == (0, 1)                               # no re-ordering, one frame of decoder buffering
    mbs = sps["mbw"] * sps["mbh"]
    limits = {30: (40500, 1620), 31: (108000, 3600), 32: (216000, 5120), 40: (245760, 8192), 41: (245760, 8192), 42: (522240, 8704),
              50: (589824, 22080), 51: (983040, 36864), 52: (2073600, 36864)}[sps["level"]]
    assert mbs * 30 <= limits[0] and mbs <= limits[1]                     # Table A-1
    assert pps == dict(pps_id=0, sps_id=0, cabac=0, bottom_field_poc=0, slice_groups=1, ref_l0=0, ref_l1=0, weighted=0, bipred=0,
                       qp=26, qs=26, chroma_qp_offset=0, deblock_control=1, constrained_intra=0, redundant_pic_cnt=0)
    lo, hi = find(buf, stbl + ("stsz",))
    _, _, n = struct.unpack(">III", buf[lo:lo + 12])
    sizes = struct.unpack(f">{n}I", buf[lo + 12:lo + 12 + 4 * n])
    assert n == 3 and "stss" not in [b[0] for b in parse_boxes(buf, *find(buf, stbl))]
    lo, hi = find(buf, stbl + ("stco",))
    pos = struct.unpack(">III", buf[lo:lo + 12])[2]
    mdat = next(x for x in parse_boxes(buf) if x[0] == "mdat")
    assert pos == mdat[1] and pos + sum(sizes) == mdat[2]
    for k, sz in enumerate(sizes):
        nal_len = struct.unpack(">I", buf[pos:pos + 4])[0]
        assert nal_len + 4 == sz                                           # one NAL unit per sample
        nal = buf[pos + 4:pos + sz]
        assert b"\x00\x00\x00" not in nal and b"\x00\x00\x01" not in nal and b"\x00\x00\x02" not in nal
        Y, Cb, Cr, hdr = read_idr_picture(nal, sps, pps)
        assert hdr["idr_pic_id"] == k % 2 and hdr["disable_deblocking"] == 1 and hdr["qp"] == 26
        ey, ecb, ecr = h264.rgb_to_yuv420(frames[k])
        assert np.array_equal(Y, ey) and np.array_equal(Cb, ecb) and np.array_equal(Cr, ecr)        # lossless in Y'CbCr
        rgb = yuv_to_rgb(Y, Cb, Cr)
        assert np.abs(np.clip(rgb, 0, 255) - frames[k]).max() <= 2.5      # 8x8 colour blocks: only the studio-range rounding
        pos += sz


def test_h264_emulation_prevention_on_hostile_samples():
    """7.4.1: sample bytes are free to be anything, also long zero runs and the start-code-like 00 00 01 / 00 00 03 patterns -
    the NAL unit must never contain 00 00 0x (x <= 2) and must read back exactly."""
    from stable_diffusion_videos_amd import h264
    rng = np.random.default_rng(2)
    H, W = 32, 48
    Y = rng.integers(0, 4, (H, W)).astype(np.uint8)                       # values 0..3 only: a maximum of escape sites
    Y[:4] = 0
    Cb = np.zeros((H // 2, W // 2), np.uint8)
    Cr = np.tile(np.array([0, 0, 1, 0, 0, 3, 0, 0, 2, 0, 0, 0], np.uint8), (H // 2, W // 24))
    nal = h264.idr_picture_yuv(Y, Cb, Cr, 1)
    assert b"\x00\x00\x00" not in nal and b"\x00\x00\x01" not in nal and b"\x00\x00\x02" not in nal
    assert nal.count(b"\x00\x00\x03") > 100
    sps_nal, pps_nal = h264.sps_pps(W, H, 24000 / 1001)
    sps, pps = read_sps(sps_nal), read_pps(pps_nal)
    assert (sps["num_units_in_tick"], sps["time_scale"]) == (1001, 48000)
    y2, cb2, cr2, hdr = read_idr_picture(nal, sps, pps)
    assert np.array_equal(y2, Y) and np.array_equal(cb2, Cb) and np.array_equal(cr2, Cr) and hdr["idr_pic_id"] == 1


def test_exp_golomb_codes_known_answers():
    """Clause 9.1, Table 9-2: the first code words, bit for bit."""
    from stable_diffusion_videos_amd.h264 import BitWriter
    want = {0: "1", 1: "010", 2: "011", 3: "00100", 4: "00101", 5: "00110", 6: "00111", 7: "0001000", 8: "0001001", 25: "000011010"}
    for v, code in want.items():
        w = BitWriter()
        w.ue(v)
        assert "".join(map(str, w.bits)) == code
    for v, k in {0: 0, 1: 1, -1: 2, 2: 3, -2: 4, 3: 5}.items():           # Table 9-3: se(v) -> codeNum
        a, b = BitWriter(), BitWriter()
        a.se(v), b.ue(k)
        assert a.bits == b.bits


def test_codec_selection_and_odd_sizes(tmp_path, monkeypatch):
    """``SDV_VIDEO_CODEC`` picks the dependency-free writer's codec; yuv420p needs even sizes, so odd frames fall back to
    Motion-JPEG (the reference's libx264 call would fail on them: "height not divisible by 2"); unknown names are an error."""
    odd = torch.randint(0, 255, (2, 3, 33, 48), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    monkeypatch.delenv("SDV_VIDEO_CODEC", raising=False)
    buf = open(make_video_pyav(odd, fps=8, output_filepath=tmp_path / "odd.mp4"), "rb").read()
    lo, hi = find(buf, ("moov", "trak", "mdia", "minf", "stbl", "stsd"))
    assert parse_boxes(buf, lo + 8, hi)[0][0] == "mp4v"
    from stable_diffusion_videos_amd import video
    assert video.LAST_CODEC["video"] == "mjpeg" and "odd frame size" in video.LAST_CODEC["why"]      # the choice is recorded
    even = odd[:, :, :32]
    buf = open(make_video_pyav(even, fps=8, output_filepath=tmp_path / "even.mp4"), "rb").read()
    lo, hi = find(buf, ("moov", "trak", "mdia", "minf", "stbl", "stsd"))
    assert parse_boxes(buf, lo + 8, hi)[0][0] == "avc1"
    assert video.LAST_CODEC["video"] == "h264 (I_PCM)" and video.LAST_CODEC["path"].endswith("even.mp4") and video.LAST_CODEC["audio"] is None
    # ... including the size fallback: above the I_PCM limit the default writer switches to Motion-JPEG and says why
    monkeypatch.setattr(video, "H264_PCM_MAX_BYTES", 1000)
    make_video_pyav(even, fps=8, output_filepath=tmp_path / "big.mp4")
    assert video.LAST_CODEC["video"] == "mjpeg" and "I_PCM stream" in video.LAST_CODEC["why"]
    monkeypatch.undo()
    monkeypatch.delenv("SDV_VIDEO_CODEC", raising=False)
    monkeypatch.setenv("SDV_VIDEO_CODEC", "vp9")
    with pytest.raises(ValueError, match="SDV_VIDEO_CODEC"):
        make_video_pyav(even, fps=8, output_filepath=tmp_path / "x.mp4")
