/* dryv_deblock.c — CPU restatement of the H.264 in-loop deblocking filter for intra pictures.
 *
 * TEST INFRASTRUCTURE ONLY (like dryv_oracle.c): only tests/ may link or call it; the product path is the HIP kernel
 * in dryv_amd/csrc/deblock.hip.
 *
 * PARITY: there is nothing in the reference to be equal to. dryv parses the deblocking syntax elements
 * (slice/header.rs:609-640, `DeblockingFilterControl`) and never filters (README.md:15 is an unchecked to-do). This file
 * follows ITU-T H.264 clause 8.7 (edge order 8.7, boundary strength 8.7.2.1, thresholds 8.7.2.2 with tables 8-16 / 8-17,
 * sample filters 8.7.2.3 / 8.7.2.4) for the domain the rest of this library has: frame macroblocks, 4:2:0, 8 bit, one slice
 * per picture, all macroblocks intra (so bS is 4 on macroblock edges and 3 inside). oracle/deblock_model.py is a second,
 * differently structured restatement of the same clauses; tests/test_deblock.py holds the two against each other and
 * against hand-derived known answers. Neither is pinned to an independent decoder: none is available in this image.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dryv_recon.h"

/* Table 8-16 */
static const uint8_t ALPHA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22,
                                  25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const uint8_t BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7,
                                 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
/* Table 8-17: tC0 for bS = 1, 2, 3 */
static const uint8_t TC0[52][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1},
    {0, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3},
    {2, 2, 3}, {2, 2, 4}, {2, 3, 4}, {2, 3, 4}, {3, 3, 5}, {3, 4, 6}, {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10},
    {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16}, {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

static int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static int iabs(int v) { return v < 0 ? -v : v; }

/* 8.5.8 (the same mapping the reconstruction uses): QPc of a macroblock for one chroma plane */
static int qpc_of(int qpy, int offset) {
  static const uint8_t QPCS[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
  const int qpi = clip3(0, 51, qpy + offset);
  return qpi < 30 ? qpi : QPCS[qpi - 30];
}

/* 8.7.2.2 - 8.7.2.4 for one line of samples across an edge: s[-4 * step .. 3 * step] = p3 p2 p1 p0 | q0 q1 q2 q3 */
static void filter_line(uint8_t *s, ptrdiff_t step, int bS, int qp_p, int qp_q, int chroma, int offA, int offB) {
  const int qpav = (qp_p + qp_q + 1) >> 1;
  const int indexA = clip3(0, 51, qpav + offA), indexB = clip3(0, 51, qpav + offB);
  const int alpha = ALPHA[indexA], beta = BETA[indexB];
  const int p0 = s[-1 * step], p1 = s[-2 * step], p2 = s[-3 * step], q0 = s[0], q1 = s[1 * step], q2 = s[2 * step];
  if (bS == 0 || !(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return; /* filterSamplesFlag */
  if (bS < 4) { /* 8.7.2.3 */
    const int tc0 = TC0[indexA][bS - 1];
    const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
    const int tc = chroma ? tc0 + 1 : tc0 + (ap < beta ? 1 : 0) + (aq < beta ? 1 : 0);
    const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    s[-1 * step] = (uint8_t)clip3(0, 255, p0 + delta);
    s[0] = (uint8_t)clip3(0, 255, q0 - delta);
    if (!chroma) {
      if (ap < beta) s[-2 * step] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
      if (aq < beta) s[1 * step] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    }
  } else { /* 8.7.2.4 */
    if (chroma) {
      s[-1 * step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
      s[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
      return;
    }
    const int p3 = s[-4 * step], q3 = s[3 * step];
    const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
    const int small = iabs(p0 - q0) < ((alpha >> 2) + 2);
    if (ap < beta && small) {
      s[-1 * step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
      s[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
      s[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
    } else {
      s[-1 * step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
    }
    if (aq < beta && small) {
      s[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
      s[1 * step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
      s[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
    } else {
      s[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
  }
}

/* Filters n_frames pictures in place (write_to_yuv_file plane order). mbs: the batch's macroblock records (qp and kind
 * are read). Returns DRYV_OK, DRYV_E_INVALID or DRYV_E_UNSUPPORTED. */
int dryv_oracle_deblock(const dryv_frame_params *fp, const dryv_deblock_params *dp, uint32_t n_frames, const dryv_mb_desc *mbs,
                        uint8_t *yuv) {
  if (!fp || !dp || !mbs || !yuv) return DRYV_E_INVALID;
  if (fp->chroma_array_type != 1 || fp->bit_depth_y != 8 || fp->bit_depth_c != 8) return DRYV_E_UNSUPPORTED;
  if (dp->disable_deblocking_filter_idc > 2 || dp->slice_alpha_c0_offset_div2 < -6 || dp->slice_alpha_c0_offset_div2 > 6 ||
      dp->slice_beta_offset_div2 < -6 || dp->slice_beta_offset_div2 > 6)
    return DRYV_E_INVALID;
  if (dp->disable_deblocking_filter_idc == 1) return DRYV_OK;
  const int W = fp->pic_width_in_mbs, H = fp->pic_height_in_mbs;
  const int offA = dp->slice_alpha_c0_offset_div2 * 2, offB = dp->slice_beta_offset_div2 * 2; /* filterOffsetA / B */
  const ptrdiff_t pitchY = 16 * W, pitchC = 8 * W;
  for (uint32_t f = 0; f < n_frames; f++) {
    uint8_t *Y = yuv + (size_t)f * W * H * 384, *C[2] = {Y + (size_t)W * H * 256, Y + (size_t)W * H * 320};
    const dryv_mb_desc *m = mbs + (size_t)f * W * H;
    for (int my = 0; my < H; my++)
      for (int mx = 0; mx < W; mx++) { /* macroblocks in raster order (8.7) */
        const dryv_mb_desc *cur = &m[my * W + mx];
        const int t8 = cur->mb_kind == 1; /* transform_size_8x8_flag: luma edges 4 and 12 are not transform edges */
        const int qp = cur->qp;
        /* luma: vertical edges left to right, then horizontal edges top to bottom */
        for (int e = 0; e < 4; e++) {
          if (e == 0 && mx == 0) continue; /* filterLeftMbEdgeFlag */
          if (t8 && (e & 1)) continue;
          const int bS = e == 0 ? 4 : 3, qp_p = e == 0 ? m[my * W + mx - 1].qp : qp;
          for (int k = 0; k < 16; k++) filter_line(Y + (16 * my + k) * pitchY + 16 * mx + 4 * e, 1, bS, qp_p, qp, 0, offA, offB);
        }
        for (int e = 0; e < 4; e++) {
          if (e == 0 && my == 0) continue; /* filterTopMbEdgeFlag */
          if (t8 && (e & 1)) continue;
          const int bS = e == 0 ? 4 : 3, qp_p = e == 0 ? m[(my - 1) * W + mx].qp : qp;
          for (int k = 0; k < 16; k++) filter_line(Y + (16 * my + 4 * e) * pitchY + 16 * mx + k, pitchY, bS, qp_p, qp, 0, offA, offB);
        }
        /* chroma (4:2:0): edges 0 and 4 of each 8x8 plane block; bS of the corresponding luma edge; QPc per plane */
        for (int pl = 0; pl < 2; pl++) {
          const int off = pl ? fp->second_chroma_qp_index_offset : fp->chroma_qp_index_offset;
          const int qc = qpc_of(qp, off);
          for (int e = 0; e < 2; e++) {
            if (e == 0 && mx == 0) continue;
            const int bS = e == 0 ? 4 : 3, qc_p = e == 0 ? qpc_of(m[my * W + mx - 1].qp, off) : qc;
            for (int k = 0; k < 8; k++) filter_line(C[pl] + (8 * my + k) * pitchC + 8 * mx + 4 * e, 1, bS, qc_p, qc, 1, offA, offB);
          }
          for (int e = 0; e < 2; e++) {
            if (e == 0 && my == 0) continue;
            const int bS = e == 0 ? 4 : 3, qc_p = e == 0 ? qpc_of(m[(my - 1) * W + mx].qp, off) : qc;
            for (int k = 0; k < 8; k++) filter_line(C[pl] + (8 * my + 4 * e) * pitchC + 8 * mx + k, pitchC, bS, qc_p, qc, 1, offA, offB);
          }
        }
      }
  }
  return DRYV_OK;
}
