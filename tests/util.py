"""Helpers shared by the tests: building macroblock records by hand, plane views."""
import numpy as np

from dryv_amd import abi


def make_mb(kind=2, qp=26, i16=2, chroma=0, prev=0, rem=None, nz=0xFFFF):
    rec = np.zeros(1, dtype=abi.MB_DESC_DTYPE)[0]
    rec["mb_kind"] = kind
    rec["qp"] = qp
    rec["i16_pred_mode"] = i16
    rec["intra_chroma_pred_mode"] = chroma
    rec["prev_flags"] = prev
    rem = [0] * 16 if rem is None else rem
    packed = np.zeros(8, dtype=np.uint8)
    for i, r in enumerate(rem):
        packed[i >> 1] |= (r & 7) << (4 * (i & 1))
    rec["rem_modes"] = packed
    rec["nz_mask"] = nz
    return rec


def make_coeffs(sparse=None):
    c = np.zeros(384, dtype=np.int16)
    for k, v in (sparse or {}).items():
        c[int(k)] = v
    return c


def split_planes(yuv, W, H, frame=0):
    """(Y, Cb, Cr) views of frame `frame` of a write_to_yuv_file-layout buffer."""
    fb = 384 * W * H
    f = yuv[frame * fb:(frame + 1) * fb]
    nl, nc = 256 * W * H, 64 * W * H
    return (f[:nl].reshape(16 * H, 16 * W), f[nl:nl + nc].reshape(8 * H, 8 * W),
            f[nl + nc:].reshape(8 * H, 8 * W))


def first_mismatch(a, b, W, H):
    """Human-readable location of the first differing byte of two frame buffers."""
    idx = np.flatnonzero(a != b)
    if idx.size == 0:
        return "identical"
    i = int(idx[0])
    fb = 384 * W * H
    f, o = divmod(i, fb)
    nl, nc = 256 * W * H, 64 * W * H
    if o < nl:
        pl, y, x, mbw = "Y", o // (16 * W), o % (16 * W), 16
    elif o < nl + nc:
        o -= nl
        pl, y, x, mbw = "Cb", o // (8 * W), o % (8 * W), 8
    else:
        o -= nl + nc
        pl, y, x, mbw = "Cr", o // (8 * W), o % (8 * W), 8
    return "%d bytes differ; first: frame %d plane %s x=%d y=%d (mb %d,%d) got %d want %d" % (
        idx.size, f, pl, x, y, x // mbw, y // mbw, a[i], b[i])
