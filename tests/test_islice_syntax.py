"""The host parser (dryv_amd/host/h264_islice.hpp) against an independent restatement of the I-slice syntax
(oracle/islice_syntax.py: written from clauses 7.3.5 / 9.3.2, decode-only, no shared code with the parser's templated
parse / encode walk). The restatement re-derives every syntax element from the parser's logged bin string and must produce
the same records and coefficient lists: this breaks the symmetry of the encode -> parse round trips (a mapping error common
to both directions -- block order, coefficient position in a list, sign, mb_qp_delta accumulation, prev / rem packing --
passes those, not this). Plus image-domain checks on the real stream: a wrong mapping that still parses gives garbage
pictures. CPU only. Parity with the reference's own parse (cabac/mod.rs:433-675) stays unpinned: no Rust toolchain here."""
import os
import sys

import numpy as np
import pytest

import oracle
from dryv_amd import abi, h264, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import islice_syntax  # noqa: E402

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "realshort.mp4")


def _same(data, picture, mbs, co):
    log, W, H, t8, qp = h264.bin_log(data, picture)
    rec, rem, coeffs = islice_syntax.parse_picture(log, W, H, t8, qp)
    assert mbs.size == W * H
    for f in ("mb_kind", "i16_pred_mode", "intra_chroma_pred_mode", "qp1y", "prev_flags"):
        want = mbs["qp" if f == "qp1y" else f].astype(np.int64)
        if f == "i16_pred_mode":
            want = np.where(mbs["mb_kind"] == 2, want, 0)
        assert np.array_equal(rec[f], want), (f, np.flatnonzero(rec[f] != want)[:5])
    assert np.array_equal(rem, np.asarray(mbs["rem_modes"]).reshape(-1, 8))
    assert np.array_equal(coeffs, co.reshape(-1, 384)), np.argwhere(coeffs != co.reshape(-1, 384))[:5]
    return log.size


def test_real_stream_pictures_reparse_from_their_bins():
    data = open(FIXTURE, "rb").read()
    fp, n_pic, mbs, co, info = h264.parse_all_islices(data)
    assert n_pic == 2
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    for p in range(n_pic):
        nb = _same(data, p, mbs[p * per:(p + 1) * per], co[p * per:(p + 1) * per])
        assert nb > 10000   # (a real picture: tens of thousands of bins)


CASES = [
    ("c3_mix", 9, 7, dict(i4x4=0.35, i8x8=0.40), dict(transform_8x8=True)),
    ("dense_all_qp", 6, 5, dict(i4x4=0.4, i8x8=0.3, coded=1.0, p0=0.9, decay4=0.97, decay8=0.99, qp=(0, 51), max_level=2047),
     dict(transform_8x8=True)),
    ("no_8x8_sparse", 8, 6, dict(i4x4=0.7, i8x8=0.0, coded=0.3), {}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_encoded_pictures_reparse_from_their_bins(case):
    name, W, H, skw, fkw = case
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 41, 0, 2)
    stream = h264.encode_stream(fp, 2, mbs, co, slice_qp=int(mbs["qp"][0]))
    fp2, n_pic, m2, c2, info = h264.parse_all_islices(stream)
    assert n_pic == 2
    for p in range(2):
        _same(stream, p, m2[p * W * H:(p + 1) * W * H], c2[p * W * H:(p + 1) * W * H])


def test_real_pictures_look_like_pictures():
    """Image-domain checks on realshort.mp4's two intra pictures (reconstructed by the oracle from the parsed batch). They
    show one scene 36 frames apart under a moving camera, so their pixels correlate (0.58 measured; floor 0.4: unrelated or
    scrambled pictures give ~0) without being equal (PSNR 13 dB: no floor on that). And the pictures are smooth the way
    decoded pictures are: the mean luma step between horizontal neighbours is 3.9 / 4.6 (bound 5.5), and across the 4x4
    block edges inside a macroblock it is 1.3 ... 1.65 x the step inside the blocks (QP 31, no deblocking; bound 1.9).
    Negative controls, measured on the same picture: every 4x4 list reversed -> mean step 8.2; the blocks of a macroblock
    in reverse order -> edge ratio 2.0, mean step 5.8. (A flipped residual sign leaves these statistics alone: that is the
    bin-level restatement's job above.)"""
    data = open(FIXTURE, "rb").read()
    fp, n_pic, mbs, co, info = h264.parse_all_islices(data)
    st, yuv = oracle.reconstruct(fp, n_pic, mbs, co)
    assert st == 0
    Wp, Hp = 16 * fp.pic_width_in_mbs, 16 * fp.pic_height_in_mbs
    fb = Wp * Hp * 3 // 2
    Y = [yuv[k * fb:k * fb + Wp * Hp].reshape(Hp, Wp).astype(np.float64) for k in range(2)]
    assert np.corrcoef(Y[0].ravel(), Y[1].ravel())[0, 1] > 0.4

    def steps(y):
        dx = np.abs(np.diff(y, axis=1))
        cols = np.arange(Wp - 1)
        return dx.mean(), dx[:, (cols % 4 == 3) & (cols % 16 != 15)].mean() / dx[:, cols % 4 != 3].mean()
    for y in Y:
        mean_step, ratio = steps(y)
        assert mean_step < 5.5 and ratio < 1.9, (mean_step, ratio)
    # negative control: the first picture with every 4x4 luma list reversed
    bad = np.ascontiguousarray(co.reshape(-1, 384)[:fp.pic_width_in_mbs * fp.pic_height_in_mbs]).copy()
    bad[:, :256] = bad[:, :256].reshape(-1, 16, 16)[:, :, ::-1].reshape(-1, 256)
    st, yb = oracle.reconstruct(fp, 1, mbs[:bad.shape[0]], bad)
    mean_step, ratio = steps(yb[:Wp * Hp].reshape(Hp, Wp).astype(np.float64))
    assert mean_step > 5.5
