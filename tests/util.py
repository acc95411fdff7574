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


V4 = np.array([[10, 16, 13], [11, 18, 14], [13, 20, 16], [14, 23, 18], [16, 25, 20], [18, 29, 23]])


def packed16_bound_batches(fp, synth, frames=1, qps=(0, 11, 23, 24, 29, 30, 35, 36, 40, 47, 51)):
    """Adversarial input for the band kernel's packed 16-bit residual path (flat scaling lists): every luma block's
    sum of |c| sits exactly AT the bound under which a step takes that path (32700 // max LS'), or one above it, spent on
    one entry, on all sixteen, on the first row / column, with the sign patterns that maximise the butterflies' growth.
    Chroma and the Intra16x16 DC terms stay at whatever the generator drew. Yields (qp, bump, mbs, coeffs)."""
    rng = np.random.default_rng(16)
    for qp in qps:
        qd, qm = qp // 6, qp % 6
        ls_max = (16 * int(V4[qm].max())) << max(qd - 4, 0)
        lim = 32700 // ls_max
        for bump in (0, 1):
            mbs, co = synth.generate(fp, synth.config(i4x4=0.8, i8x8=0.0, coded=1.0, qp=(qp, qp)), 1600 + qp, 0, frames)
            co = co.astype(np.int64)
            for a in range(co.shape[0]):
                if mbs["mb_kind"][a] != 0:
                    continue
                for b in range(16):
                    blk = np.zeros(16, dtype=np.int64)
                    tot = lim + bump
                    pat = int(rng.integers(0, 5))
                    if pat == 0:                                  # everything on one entry
                        blk[int(rng.integers(0, 16))] = tot
                    elif pat == 1:                                # spread over all sixteen
                        blk[:] = tot // 16
                        blk[0] += tot - blk.sum()
                    elif pat == 2:                                # the first four list entries
                        blk[:4] = tot // 4
                        blk[0] += tot - blk.sum()
                    elif pat == 3:                                # two entries
                        k = rng.choice(16, size=2, replace=False)
                        blk[k[0]] = tot // 2
                        blk[k[1]] = tot - tot // 2
                    else:                                         # random split
                        w = rng.random(16)
                        blk = np.floor(w / w.sum() * tot).astype(np.int64)
                        blk[int(rng.integers(0, 16))] += tot - blk.sum()
                    sgn = int(rng.integers(0, 3))
                    sign = np.ones(16, dtype=np.int64) if sgn == 0 else (-np.ones(16, dtype=np.int64) if sgn == 1
                                                                         else rng.choice([-1, 1], size=16))
                    co[a, 16 * b:16 * b + 16] = np.clip(blk * sign, -32768, 32767)
            yield qp, bump, mbs, co.astype(np.int16)


V8 = np.array([[20, 18, 32, 19, 25, 24], [22, 19, 35, 21, 28, 26], [26, 23, 42, 24, 33, 31], [28, 25, 45, 26, 35, 33],
               [32, 28, 51, 30, 40, 38], [36, 32, 58, 34, 46, 43]])


def packed16_8x8_limit(qp, ls_max):
    """The bound of the packed 16-bit 8x8 residual (band_kernel.h, T_THR8P), restated: the largest sum of |c| over an 8x8
    block for which the step may take it."""
    qd = qp // 6
    by_product = 32700 // ls_max
    by_d = 14400 // (ls_max << (qd - 6)) if qd >= 6 else ((14400 - 96) << (6 - qd)) // ls_max
    return min(by_product, by_d, 65535)


def packed16_8x8_bound_batches(fp, synth, frames=1, qps=(0, 7, 17, 24, 29, 30, 35, 36, 41, 42, 47, 51)):
    """Adversarial input for the packed 16-bit form of the 8x8 residual (flat scaling lists): every 8x8 block of every
    Intra8x8 macroblock has its sum of |c| exactly AT the bound, or one above it (the step then takes the 32-bit passes):
    on one entry (the positions with the largest butterfly weights and LevelScale among them), on two, on a row of the
    list, spread over all 64, random; one sign, alternating signs, random signs. Yields (qp, bump, mbs, coeffs)."""
    rng = np.random.default_rng(88)
    for qp in qps:
        lim = packed16_8x8_limit(qp, 16 * int(V8[qp % 6].max()))
        for bump in (0, 1):
            mbs, co = synth.generate(fp, synth.config(i4x4=0.3, i8x8=0.6, coded=1.0, qp=(qp, qp)), 2600 + qp, 0, frames)
            co = co.astype(np.int64)
            for a in range(co.shape[0]):
                if mbs["mb_kind"][a] != 1:
                    continue
                for b in range(4):
                    blk = np.zeros(64, dtype=np.int64)
                    tot = lim + bump
                    pat = int(rng.integers(0, 5))
                    if pat == 0:                                  # everything on one entry (early list entries: low frequencies)
                        blk[int(rng.integers(0, 64)) if rng.random() < 0.5 else int(rng.integers(0, 6))] = tot
                    elif pat == 1:                                # spread over all 64
                        blk[:] = tot // 64
                        blk[0] += tot - blk.sum()
                    elif pat == 2:                                # the first eight list entries
                        blk[:8] = tot // 8
                        blk[1] += tot - blk.sum()
                    elif pat == 3:                                # two entries
                        k = rng.choice(64, size=2, replace=False)
                        blk[k[0]] = tot // 2
                        blk[k[1]] = tot - tot // 2
                    else:                                         # random split
                        w = rng.random(64)
                        blk = np.floor(w / w.sum() * tot).astype(np.int64)
                        blk[int(rng.integers(0, 64))] += tot - blk.sum()
                    sgn = int(rng.integers(0, 4))
                    sign = (np.ones(64, dtype=np.int64) if sgn == 0 else -np.ones(64, dtype=np.int64) if sgn == 1
                            else np.where(np.arange(64) % 2, -1, 1) if sgn == 2 else rng.choice([-1, 1], size=64))
                    co[a, 64 * b:64 * b + 64] = np.clip(blk * sign, -32768, 32767)
            yield qp, bump, mbs, co.astype(np.int16)
