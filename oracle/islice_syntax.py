"""islice_syntax.py -- TEST INFRASTRUCTURE: an independent, decode-only restatement of the I-slice macroblock layer
syntax (ITU-T H.264 clauses 7.3.5, 7.3.5.1, 7.3.5.3, 7.3.5.3.3, 7.4.5, 9.3.2 binarizations), written from the clauses and
not from dryv_amd/host/h264_islice.hpp -- whose one templated walk serves both its parser and its encoder, so that an
encode -> parse round trip cannot see a mapping error the two directions share (block order, coefficient position in a
list, sign, mb_qp_delta accumulation, prev / rem packing).

Input: the arithmetic decoder's bin string as that parser logs it (dryv_amd.h264.bin_log: value | kind << 1 per bin). This
module never sees a context index: it re-derives every syntax element from the bins by the binarizations of 9.3.2 alone,
walks the syntax structure itself, and lays the coefficient levels out in the lists of the reconstruction ABI
(include/dryv_recon.h). Only tests import it. Parity with the reference's own parse (src/video/cabac/mod.rs:89-210,
433-675) stays unpinned: the reference cannot be built here; what this pins is the host parser against the standard's
syntax tables, given a bin string that the real-stream CABAC synchronisation test pins separately.
"""
import numpy as np

CTX, BYPASS, TERMINATE = 0, 1, 2


class Bins:
    """The logged bins, read in order; every read states what kind of bin the syntax expects there."""

    def __init__(self, log):
        self.log = np.asarray(log, dtype=np.uint8)
        self.pos = 0

    def get(self, kind):
        if self.pos >= self.log.size:
            raise ValueError("bin string exhausted")
        b = int(self.log[self.pos])
        self.pos += 1
        if (b >> 1) != kind:
            raise ValueError("bin %d is of kind %d, the syntax expects kind %d" % (self.pos - 1, b >> 1, kind))
        return b & 1

    def done(self):
        return self.pos == self.log.size


# ---- binarizations (9.3.2) --------------------------------------------------------------------------------------------
def tu(bins, cmax):                       # 9.3.2.2 truncated unary
    v = 0
    while v < cmax and bins.get(CTX):
        v += 1
    return v


def fl_lsb_first(bins, nbits):            # 9.3.2.5 fixed length: binIdx 0 is the least significant bit
    return sum(bins.get(CTX) << k for k in range(nbits))


def mb_qp_delta(bins):                    # 9.3.2.7 + table 9-3: unary of the mapped value k; delta = (-1)^(k+1) ceil(k / 2)
    k = 0
    while bins.get(CTX):
        k += 1
    return (k + 1) // 2 if k & 1 else -(k // 2)


def coeff_abs_level_minus1(bins):         # 9.3.2.3 UEG0, uCoff = 14, signedValFlag = 0
    v = tu(bins, 14)
    if v < 14:
        return v
    k = 0
    while bins.get(BYPASS):
        v += 1 << k
        k += 1
    while k:
        k -= 1
        v += bins.get(BYPASS) << k
    return v


# ---- residual_block_cabac (7.3.5.3.3) ---------------------------------------------------------------------------------
def residual_block(bins, start, end, max_num, has_cbf):
    """Returns coeffLevel[0 .. max_num-1] (zeros outside start..end)."""
    lev = [0] * max_num
    if has_cbf and not bins.get(CTX):                  # coded_block_flag
        return lev
    num = end + 1
    sig = [False] * max_num
    i = start
    while i < num - 1:
        sig[i] = bool(bins.get(CTX))                   # significant_coeff_flag[i]
        if sig[i] and bins.get(CTX):                   # last_significant_coeff_flag[i]
            num = i + 1
        i += 1
    sig[num - 1] = True
    for i in range(num - 1, start - 1, -1):            # levels are coded from the last significant position down
        if sig[i]:
            mag = coeff_abs_level_minus1(bins) + 1
            lev[i] = -mag if bins.get(BYPASS) else mag  # coeff_sign_flag
    return lev


# blkIdx -> position inside the macroblock in units of 4x4 blocks (6.4.3, figure 6-10): the z order
def blk_xy(blk):
    return ((blk >> 1) & 2) | (blk & 1), ((blk >> 2) & 2) | ((blk >> 1) & 1)


def parse_picture(log, W, H, transform_8x8_mode, slice_qp):
    """The macroblock layer of one I slice covering a picture of W x H macroblocks. Returns (mbs, coeffs) in the
    reconstruction ABI's layout: records of (mb_kind, i16_pred_mode, intra_chroma_pred_mode, qp1y, prev_flags,
    rem_modes[8]) as a dict of arrays, and int16 [n][384] lists."""
    bins = Bins(log)
    n = W * H
    out = {k: np.zeros(n, dtype=np.int64) for k in ("mb_kind", "i16_pred_mode", "intra_chroma_pred_mode", "qp1y", "prev_flags")}
    rem_modes = np.zeros((n, 8), dtype=np.uint8)
    coeffs = np.zeros((n, 384), dtype=np.int64)
    qp = slice_qp                                       # QPY,PRED of the first macroblock: SliceQPY (7.4.5)
    for a in range(n):
        # ---- mb_type (table 9-36, I slices): 0 = I_NxN; 1 then a terminate bin: I_PCM; else the Intra16x16 types
        cbp_luma = cbp_chroma = 0
        i16 = bool(bins.get(CTX))
        t8 = False
        if i16:
            if bins.get(TERMINATE):
                raise ValueError("I_PCM macroblock: outside the backend's domain")
            luma15 = bins.get(CTX)                      # b2: AC residual of luma coded (CodedBlockPatternLuma 15)
            cbp_chroma = bins.get(CTX)                  # b3 (+ b4): CodedBlockPatternChroma 0 / 1 / 2
            if cbp_chroma:
                cbp_chroma = 1 + bins.get(CTX)
            pred = (bins.get(CTX) << 1) | bins.get(CTX)  # b5 b6: Intra16x16PredMode
            cbp_luma = 15 if luma15 else 0
            out["mb_kind"][a], out["i16_pred_mode"][a] = 2, pred
        else:
            if transform_8x8_mode:
                t8 = bool(bins.get(CTX))                # transform_size_8x8_flag
            out["mb_kind"][a] = 1 if t8 else 0
            # mb_pred(): prev_intra{4x4,8x8}_pred_mode_flag, rem_intra{4x4,8x8}_pred_mode in luma4x4BlkIdx / luma8x8BlkIdx order
            flags, rem = 0, [0] * 16
            for k in range(4 if t8 else 16):
                if bins.get(CTX):
                    flags |= 1 << k
                else:
                    rem[k] = fl_lsb_first(bins, 3)
            out["prev_flags"][a] = flags
            for k in range(8):                          # the ABI packs two 4-bit fields per byte, low nibble first
                rem_modes[a, k] = rem[2 * k] | (rem[2 * k + 1] << 4)
        out["intra_chroma_pred_mode"][a] = tu(bins, 3)
        if not i16:
            # coded_block_pattern (9.3.2.6): prefix FL of 4 bins, bin b8 = luma 8x8 block b8; suffix TU cMax 2 (chroma)
            for b8 in range(4):
                cbp_luma |= bins.get(CTX) << b8
            cbp_chroma = tu(bins, 2)
        if cbp_luma or cbp_chroma or i16:
            # QPY = (QPY,PRED + mb_qp_delta + 52) % 52 for 8-bit video (7.4.5, equation 7-37); it becomes the next QPY,PRED
            qp = (qp + mb_qp_delta(bins) + 52) % 52
        out["qp1y"][a] = qp
        # ---- residual(): residual_luma, then chroma DC of Cb, Cr, then chroma AC of Cb, Cr (7.3.5.3)
        c = coeffs[a]
        if i16:
            c[0:16] = residual_block(bins, 0, 15, 16, True)                 # Intra16x16DCLevel
        for b8 in range(4):
            if t8:
                if cbp_luma >> b8 & 1:                  # level8x8: no coded_block_flag in a CABAC stream that is not 4:4:4
                    c[64 * b8:64 * b8 + 64] = residual_block(bins, 0, 63, 64, False)
                continue
            for k in range(4):
                blk = 4 * b8 + k
                if not (cbp_luma >> b8 & 1):
                    continue
                if i16:                                 # Intra16x16ACLevel: 15 entries, list position k = AC k
                    c[16 + 15 * blk:16 + 15 * blk + 15] = residual_block(bins, 0, 14, 15, True)
                else:
                    c[16 * blk:16 * blk + 16] = residual_block(bins, 0, 15, 16, True)
        if cbp_chroma & 3:
            for pl in range(2):
                c[256 + 64 * pl:256 + 64 * pl + 4] = residual_block(bins, 0, 3, 4, True)       # ChromaDCLevel
        if cbp_chroma & 2:
            for pl in range(2):
                for blk in range(4):
                    o = 256 + 64 * pl + 4 + 15 * blk
                    c[o:o + 15] = residual_block(bins, 0, 14, 15, True)                        # ChromaACLevel
        # ---- end_of_slice_flag
        if bins.get(TERMINATE) != (1 if a == n - 1 else 0):
            raise ValueError("end_of_slice_flag at macroblock %d of %d" % (a, n))
    if not bins.done():
        raise ValueError("%d bins behind the last macroblock" % (bins.log.size - bins.pos))
    if np.abs(coeffs).max(initial=0) > 32768 or coeffs.max(initial=0) > 32767:
        raise ValueError("coefficient outside int16")
    return out, rem_modes, coeffs.astype(np.int16)
