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
