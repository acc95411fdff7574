"""The host producer (dryv_amd/host/h264_islice.hpp via libdryv_h264.so): mp4 demux + parameter sets + slice header +
I-slice CABAC parse, and the CABAC I-slice encoder. CPU only.

What pins the parser: (1) a real stream produced by someone else's encoder -- tests/golden/realshort.mp4 (imageio's
96 KB test clip: High profile, 320x240, CABAC, 8x8 transform enabled; a data fixture, not part of the reference):
arithmetic decoding must hit end_of_slice_flag exactly at the last macroblock with the engine's read position right
behind the rbsp_stop_one_bit of the NAL unit -- a single wrong context index or bin anywhere desynchronises the engine
long before that; (2) encode -> parse round trips of synthetic batches of every macroblock kind.
The reference (Rust) cannot be built here, so its own parse of the same file is not available: parity with it stays
unpinned."""
import os

import numpy as np
import pytest

import oracle
from dryv_amd import abi, h264, synth

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "realshort.mp4")


def test_parse_real_mp4_first_idr():
    fp, mbs, co, info = h264.parse_first_islice(open(FIXTURE, "rb").read())
    assert (fp.pic_width_in_mbs, fp.pic_height_in_mbs) == (20, 15)
    assert fp.transform_8x8_mode_flag == 1 and fp.chroma_array_type == 1
    assert info["tail_ok"] == 1 and info["bits_unread"] < 8, info       # terminate bin at MB 299, NAL fully consumed
    assert info["n_i4x4"] + info["n_i8x8"] + info["n_i16x16"] == 300
    assert mbs["qp"].max() <= 51 and mbs["mb_kind"].max() <= 2
    assert mbs["i16_pred_mode"].max() <= 3 and mbs["intra_chroma_pred_mode"].max() <= 3
    assert np.abs(co.astype(np.int32)).max() < 2048                      # a conformant 8-bit stream
    st, yuv = oracle.reconstruct(fp, 1, mbs, co)                          # every derived mode legal, no unsupported record
    assert st == 0
    # a natural picture, not noise: neighbouring luma rows correlate strongly
    Y = yuv[:320 * 240].reshape(240, 320).astype(np.float64)
    assert np.corrcoef(Y[:-1].ravel(), Y[1:].ravel())[0, 1] > 0.9


def test_truncated_and_corrupt_streams_are_rejected():
    data = bytearray(open(FIXTURE, "rb").read())
    with pytest.raises(h264.H264Error):
        h264.parse_first_islice(bytes(data[:4000]))
    with pytest.raises(h264.H264Error):
        h264.parse_first_islice(b"\x00\x00\x00\x01\x65\x88\x84\x00")


CASES = [
    ("i16_only", 7, 5, dict(i4x4=0.0, i8x8=0.0), {}),
    ("i4x4_only", 7, 5, dict(i4x4=1.0, i8x8=0.0), {}),
    ("i8x8_only", 7, 5, dict(i4x4=0.0, i8x8=1.0), dict(transform_8x8=True)),
    ("c2_mix", 12, 9, dict(i4x4=0.7, i8x8=0.0), {}),
    ("c3_mix", 12, 9, dict(i4x4=0.35, i8x8=0.40), dict(transform_8x8=True)),
    ("dense_all_qp", 8, 6, dict(i4x4=0.4, i8x8=0.3, coded=1.0, p0=0.9, decay4=0.97, decay8=0.99, qp=(0, 51)),
     dict(transform_8x8=True)),
    ("sparse", 9, 7, dict(i4x4=0.4, i8x8=0.3, coded=0.1), dict(transform_8x8=True)),
    ("zero_residual", 5, 4, dict(i4x4=0.4, i8x8=0.3, coded=0.0), dict(transform_8x8=True)),
    ("chroma_offsets", 6, 5, dict(i4x4=0.5, i8x8=0.2), dict(transform_8x8=True, cqo_cb=-5, cqo_cr=7)),
    ("one_mb", 1, 1, dict(i4x4=0.5, i8x8=0.3), dict(transform_8x8=True)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_encode_parse_round_trip(case):
    """A synthetic batch -> CABAC Annex-B stream -> parse: the records and coefficients come back, and the oracle
    reconstructs the same picture from both. (A macroblock without any coded coefficient carries no mb_qp_delta, so
    its qp reads back as its predecessor's: irrelevant to reconstruction, excluded from the record comparison.)"""
    name, W, H, skw, fkw = case
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 300 + CASES.index(case), 0, 1)
    stream = h264.encode_idr(fp, mbs, co, slice_qp=int(mbs["qp"][0]))
    fp2, mbs2, co2, info = h264.parse_first_islice(stream)
    assert bytes(fp2) == bytes(fp)
    assert info["tail_ok"] == 1 and info["bits_unread"] < 8
    assert np.array_equal(co2, co)
    for f in ("mb_kind", "intra_chroma_pred_mode"):
        assert np.array_equal(mbs2[f], mbs[f]), f
    k2 = mbs["mb_kind"] == 2
    assert np.array_equal(mbs2["i16_pred_mode"][k2], mbs["i16_pred_mode"][k2])
    n_modes = np.where(mbs["mb_kind"] == 1, 4, 16)
    for a in np.flatnonzero(~k2):
        n = n_modes[a]
        fl1, fl2 = int(mbs["prev_flags"][a]) & ((1 << n) - 1), int(mbs2["prev_flags"][a])
        assert fl1 == fl2
        for b in range(n):
            if not (fl1 >> b) & 1:
                r1 = (mbs["rem_modes"][a][b >> 1] >> (4 * (b & 1))) & 7
                r2 = (mbs2["rem_modes"][a][b >> 1] >> (4 * (b & 1))) & 7
                assert r1 == r2
    coded = (np.abs(co).sum(axis=1) > 0) | k2
    assert np.array_equal(mbs2["qp"][coded], mbs["qp"][coded])
    st1, y1 = oracle.reconstruct(fp, 1, mbs, co)
    st2, y2 = oracle.reconstruct(fp2, 1, mbs2, co2)
    assert st1 == 0 and st2 == 0 and np.array_equal(y1, y2)


def test_real_stream_survives_reencoding():
    """parse(realshort) -> encode -> parse gives the same batch: the encoder writes what the parser of a third-party
    stream reads."""
    fp, mbs, co, info = h264.parse_first_islice(open(FIXTURE, "rb").read())
    fp2, mbs2, co2, info2 = h264.parse_first_islice(h264.encode_idr(fp, mbs, co, slice_qp=info["slice_qp"]))
    assert np.array_equal(co2, co) and np.array_equal(mbs2["mb_kind"], mbs["mb_kind"])
    assert info2["bins"] == info["bins"]     # bin for bin the same arithmetic-coded sequence
    st1, y1 = oracle.reconstruct(fp, 1, mbs, co)
    st2, y2 = oracle.reconstruct(fp2, 1, mbs2, co2)
    assert np.array_equal(y1, y2)


def test_frame_cropping_rectangle_round_trips():
    """The SPS's frame cropping rectangle (sps.rs:252-267: parsed by the reference, never applied) comes out of the parser
    in luma samples, as the output stage takes it; the encoder writes it. realshort.mp4 (320x240 = 20x15 macroblocks)
    has none."""
    assert h264.parse_first_islice(open(FIXTURE, "rb").read())[3]["crop"] == (0, 0, 0, 0)
    fp = abi.make_frame_params(8, 5)                          # 128 x 80 coded
    mbs, co = synth.generate(fp, synth.config(i4x4=0.6, i8x8=0.0), 4242, 0, 1)
    stream = h264.encode_idr(fp, mbs, co, slice_qp=int(mbs["qp"][0]), crop=(2, 6, 0, 8))   # a 120 x 72 picture
    fp2, mbs2, co2, info = h264.parse_first_islice(stream)
    assert info["crop"] == (2, 6, 0, 8) and info["tail_ok"] == 1
    assert np.array_equal(co2, co)
    with pytest.raises(h264.H264Error):
        h264.encode_idr(fp, mbs, co, crop=(1, 0, 0, 0))      # odd: not expressible in crop units


def test_all_intra_stream_round_trip():
    """A batch of pictures -> one all-intra Annex-B stream (SPS, PPS, an IDR slice per picture) -> parse_all_islices: the
    batch comes back picture for picture, ready for one submit. (The reference stops after sample 0, quirk Q9: batches are
    this build's own unit of work.)"""
    fp = abi.make_frame_params(9, 6, transform_8x8=True)
    frames = 5
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 5151, 0, frames)
    stream = h264.encode_stream(fp, frames, mbs, co, slice_qp=int(mbs["qp"][0]))
    fp2, n2, mbs2, co2, info = h264.parse_all_islices(stream)
    assert n2 == frames and bytes(fp2) == bytes(fp) and info["tails_ok"] == 1 and info["skipped"] == 0
    assert np.array_equal(co2, co) and np.array_equal(mbs2["mb_kind"], mbs["mb_kind"])
    st1, y1 = oracle.reconstruct(fp, frames, mbs, co)
    st2, y2 = oracle.reconstruct(fp2, n2, mbs2, co2)
    assert st1 == 0 and st2 == 0 and np.array_equal(y1, y2)
    # a limit on the number of pictures, and the single-picture entry point on the same stream
    assert h264.parse_all_islices(stream, max_pictures=2)[1] == 2
    fp1, mbs1, co1, _ = h264.parse_first_islice(stream)
    assert np.array_equal(co1, co[:54])


def test_inter_slice_with_a_missing_parameter_set_does_not_cost_the_intra_pictures():
    """An inter slice that names a picture parameter set which was never sent (arrives late, was lost, is not supported) is
    skipped like every inter slice: the stream's intra pictures still parse. An intra picture that names one is refused --
    when it is asked for."""
    fp = abi.make_frame_params(5, 4)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.6), 616, 0, 2)
    stream = bytes(h264.encode_stream(fp, 2, mbs, co, slice_qp=int(mbs["qp"][0])))
    # a P slice, first_mb_in_slice 0, slice_type 0, pic_parameter_set_id 7 (ue: 1, 1, 0001000), then the stop bit
    stray = b"\x00\x00\x00\x01\x41\xc4\x40"
    k = stream.rfind(b"\x00\x00\x00\x01")            # in front of the second IDR slice
    fp2, n2, mbs2, co2, info = h264.parse_all_islices(stream[:k] + stray + stream[k:])
    assert n2 == 2 and info["skipped"] == 1 and np.array_equal(co2, co)
    # the same stray unit as an IDR I slice (slice_type 7: ue 0001000; pps 7): asked for, refused
    bad = b"\x00\x00\x00\x01\x65" + bytes([0b10001000, 0b00010001, 0b00000000])
    with pytest.raises(h264.H264Error) as e:
        h264.parse_all_islices(stream[:k] + bad)
    assert "parameter set" in str(e.value)


def test_real_mp4_intra_pictures_among_inter_ones():
    """realshort.mp4 is an ordinary IPB stream: sample positions come from stco / stsc / stsz, the intra pictures are
    parsed (each must end at its terminating bin), the inter pictures are counted and skipped."""
    data = open(FIXTURE, "rb").read()
    fp, n_pic, mbs, co, info = h264.parse_all_islices(data)
    assert n_pic >= 1 and info["tails_ok"] == 1 and info["skipped"] >= 1
    fp1, mbs1, co1, _ = h264.parse_first_islice(data)
    assert bytes(fp) == bytes(fp1) and np.array_equal(co[:300], co1) and np.array_equal(mbs[:300], mbs1)
    st, yuv = oracle.reconstruct(fp, n_pic, mbs, co)
    assert st == 0


def test_parse_into_batch_buffers_in_parallel():
    """stream_params sizes the buffers from the parameter sets alone; parse_all_islices_into then parses the pictures on
    several threads straight into the caller's batch arrays: same result as the copying, single-threaded entry point."""
    fp = abi.make_frame_params(10, 6, transform_8x8=True)
    frames = 7
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 99, 0, frames)
    stream = h264.encode_stream(fp, frames, mbs, co, slice_qp=int(mbs["qp"][0]))
    fps, n_slices = h264.stream_params(stream)
    assert bytes(fps) == bytes(fp) and n_slices == frames
    per = 60
    m_out = np.zeros(n_slices * per, dtype=abi.MB_DESC_DTYPE)
    c_out = np.full((n_slices * per, 384), 77, dtype=np.int16)      # (must be cleared by the parser)
    fp2, n2, info = h264.parse_all_islices_into(stream, m_out, c_out, threads=4)
    ref = h264.parse_all_islices(stream)
    assert n2 == frames and info["tails_ok"] == 1 and np.array_equal(c_out, ref[3]) and np.array_equal(m_out, ref[2])
    with pytest.raises(h264.H264Error):                              # too small a buffer is refused, not overrun
        h264.parse_all_islices_into(stream, m_out[:3 * per], c_out[:3 * per], threads=2)


# ---- hardening (round 3): sizes are bounded where they are parsed, nothing crosses the C boundary by exception ---------
class _Bits:
    """A minimal RBSP writer for hand-made parameter sets."""
    def __init__(self):
        self.b = []

    def u(self, v, n):
        self.b += [(v >> i) & 1 for i in range(n - 1, -1, -1)]

    def ue(self, v):
        x = int(v) + 1
        n = x.bit_length() - 1
        self.u(0, n)
        self.u(x, n + 1)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def nal(self, header):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, zeros = bytearray(b"\x00\x00\x00\x01" + bytes([header])), 0
        for c in raw:   # emulation prevention
            if zeros >= 2 and c <= 3:
                out.append(3)
                zeros = 0
            out.append(c)
            zeros = zeros + 1 if c == 0 else 0
        return bytes(out)


def _sps(width_mbs_minus1, height_mbs_minus1, matrix=None):
    """High-profile SPS; matrix: None or a list of 8 entries, each None (list not present) or a list of values."""
    w = _Bits()
    w.u(100, 8); w.u(0, 8); w.u(40, 8); w.ue(0)
    w.ue(1); w.ue(0); w.ue(0); w.u(0, 1)
    w.u(1 if matrix else 0, 1)
    if matrix:
        for lst in matrix:
            w.u(0 if lst is None else 1, 1)
            last = 8
            for v in (lst or []):
                v = int(v)
                d = v - last
                d = d - 256 if d > 127 else d + 256 if d < -128 else d
                w.se(d)
                last = v
    w.ue(0); w.ue(2); w.ue(1); w.u(0, 1)
    w.ue(width_mbs_minus1); w.ue(height_mbs_minus1)
    w.u(1, 1); w.u(1, 1); w.u(0, 1); w.u(0, 1)
    return w.nal(0x67)


def _replace_sps(stream, sps):
    """The Annex-B stream with its first NAL unit (the encoder writes the SPS first) replaced."""
    nxt = stream.index(b"\x00\x00\x00\x01", 4)
    return sps + stream[nxt:]


def test_oversized_picture_in_sps_is_refused_not_truncated():
    """pic_width_in_mbs_minus1 = 65536 used to be truncated to uint16 in the parameters the caller sizes
    its buffers from (1x1) while the parse loop kept the full width and wrote past them."""
    fp = abi.make_frame_params(1, 1)
    mbs, co = synth.generate(fp, synth.config(), 5, 0, 1)
    good = h264.encode_stream(fp, 1, mbs, co, slice_qp=int(mbs["qp"][0]))
    assert h264.stream_params(good)[0].pic_width_in_mbs == 1
    for wm1, hm1 in ((65536, 0), (0, 65536), (1024, 0), (70000, 70000), (2 ** 31, 3)):
        bad = _replace_sps(good, _sps(wm1, hm1))
        with pytest.raises(h264.H264Error, match="larger than"):
            h264.stream_params(bad)
        m1, c1 = np.zeros(1, dtype=abi.MB_DESC_DTYPE), np.zeros(384, dtype=np.int16)
        with pytest.raises(h264.H264Error):
            h264.parse_all_islices_into(bad, m1, c1)
        with pytest.raises(h264.H264Error):
            h264.parse_first_islice(bad)
    # the largest picture the library takes is still parsed as such (the slice then fails: it is a 1-macroblock slice)
    big = _replace_sps(good, _sps(1023, 9))
    assert (h264.stream_params(big)[0].pic_width_in_mbs, h264.stream_params(big)[0].pic_height_in_mbs) == (1024, 10)


def test_scaling_matrix_round_trip_and_reference_fallback():
    """Non-flat lists travel in the SPS and come back entry for entry; a list that is not present becomes the Default_*
    table -- the reference's rule (atom/avcc/sps.rs:206-249), not the standard's fall-back rule A."""
    fp = abi.make_frame_params(5, 4, transform_8x8=True)
    rng = np.random.default_rng(11)
    l4 = rng.integers(1, 256, size=(6, 16), dtype=np.uint8)
    l8 = np.full((6, 64), 16, dtype=np.uint8)
    l8[:2] = rng.integers(1, 256, size=(2, 64), dtype=np.uint8)
    np.ctypeslib.as_array(fp.scaling_list4x4)[:] = l4
    np.ctypeslib.as_array(fp.scaling_list8x8)[:] = l8
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 13, 0, 2)
    stream = h264.encode_stream(fp, 2, mbs, co, slice_qp=int(mbs["qp"][0]))
    fp2, n, m2, c2, info = h264.parse_all_islices(stream)
    assert n == 2 and bytes(fp2) == bytes(fp) and np.array_equal(m2, mbs) and np.array_equal(c2, co)
    # lists 1 and 7 left out, list 3 switched to its default by a leading zero
    matrix = [list(l4[0]), None, list(l4[2]), [0], list(l4[4]), list(l4[5]), list(l8[0]), None]
    # ([0] codes delta_scale = -8 for entry 0: nextScale == 0 -> useDefaultScalingMatrixFlag)
    s2 = _replace_sps(stream, _sps(4, 3, matrix))
    fp3 = h264.stream_params(s2)[0]
    got4, got8 = np.ctypeslib.as_array(fp3.scaling_list4x4), np.ctypeslib.as_array(fp3.scaling_list8x8)
    d4i = [6, 13, 13, 20, 20, 20, 28, 28, 28, 28, 32, 32, 32, 37, 37, 42]
    d4p = [10, 14, 14, 20, 20, 20, 24, 24, 24, 24, 27, 27, 27, 30, 30, 34]
    assert list(got4[0]) == list(l4[0]) and list(got4[1]) == d4i and list(got4[3]) == d4p and list(got4[5]) == list(l4[5])
    assert list(got8[0]) == list(l8[0]) and got8[1][0] == 9 and got8[1][63] == 35 and (got8[2:] == 16).all()


def test_parser_survives_mutated_streams_under_sanitizers(tmp_path):
    """A malformed-input fuzz of the host parser's source built with -fsanitize=address,undefined (CPU build only)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = tmp_path / "h264_fuzz"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-o", str(exe), os.path.join(here, "fuzz", "h264_fuzz.cpp")], check=True)
    fp = abi.make_frame_params(6, 4, transform_8x8=True)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 17, 0, 3)
    annexb = tmp_path / "s.h264"
    annexb.write_bytes(h264.encode_stream(fp, 3, mbs, co, slice_qp=int(mbs["qp"][0])))
    for seedfile, iters in ((FIXTURE, 250), (str(annexb), 400)):
        r = subprocess.run([str(exe), seedfile, str(iters), "12345"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        assert "rejected" in r.stdout
