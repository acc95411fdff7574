/*
 * dryv_oracle.c — CPU restatement of dryv's macroblock-reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the parity checker for the HIP backend; it is never
 * linked into, imported by or called from the product library (dryv_amd/). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference (Stuff7/dryv, /root/reference) ships no tests, golden vectors or
 * fixtures for this path, and it cannot be built here (no rustc/cargo). This file is
 * therefore a line-by-line restatement of the Rust sources, checked against hand-derived
 * known-answer vectors (tests/golden/kat_vectors.json + make_kat_vectors.py, each expectation derived by
 * hand from the cited reference lines, not produced by this file) and against a second restatement with a
 * different structure and source (oracle/spec_model.py: written from the H.264 clauses, the reference's
 * quirks patched in; tests/test_spec_model.py). Neither pins it to the reference's actual output.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Arithmetic is int64_t throughout because the reference computes in 64-bit `isize`; `>>` on
 * negative values is an arithmetic shift in Rust and (implementation-defined but universally so)
 * in gcc/clang; `x << n` on possibly-negative x is written as x * (1 << n).
 *
 * Structure deliberately mirrors the reference, including what makes it slow: scaling() is
 * recomputed per macroblock, every reference sample goes through the neighbour derivation with
 * div/mod, planes are column-major [x][y]. The reference's quirks Q1-Q5 (SURVEY.md §8a') are
 * reproduced and marked "QUIRK".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dryv_recon.h"

typedef int64_t isize;

/* ------------------------------------------------------------------------------------------- */
/* src/math.rs                                                                                 */
/* ------------------------------------------------------------------------------------------- */

/* math.rs:109-117 */
static isize clamp(isize value, isize min, isize max) {
  if (value < min) return min;
  if (value > max) return max;
  return value;
}

/* math.rs:119-125 */
static isize inverse_raster_scan(isize a, isize b, isize c, isize d, isize e) {
  if (e == 0) return (a % (d / b)) * b;
  return (a / (d / b)) * c;
}

/* ------------------------------------------------------------------------------------------- */
/* Records (the fields of slice/macroblock.rs:21-129 that the path reads or writes)            */
/* ------------------------------------------------------------------------------------------- */

enum { MODE_INTRA4X4 = 0, MODE_INTRA8X8 = 1, MODE_INTRA16X16 = 2, MODE_NA = 3 };

/* What survives of a decoded macroblock for its neighbours: type + derived prediction modes. */
typedef struct {
  int unavailable; /* MbType::Unavailable (slice/consts.rs:3 MB_UNAVAILABLE_INTRA)                   */
  int mode;        /* PartPredMode of mb_type.mode() (macroblock.rs:593-599)                   */
  isize intra4x4_pred_mode[16];
  isize intra8x8_pred_mode[4];
} MbRec;

/* The current macroblock's full record (macroblock.rs:21-129); freshly "empty()" per MB
 * (macroblock.rs:156-202: all prediction-sample arrays start at 0 — QUIRK Q4 depends on it). */
typedef struct {
  int i16_pred_mode;
  int intra_chroma_pred_mode;
  isize qpy, qp1y, qp1c, qpc;
  uint8_t prev_intra4x4_pred_mode_flag[16], rem_intra4x4_pred_mode[16];
  uint8_t prev_intra8x8_pred_mode_flag[4], rem_intra8x8_pred_mode[4];
  isize luma_pred_samples[16][4][4];   /* [blk][x][y] */
  isize luma16x16_pred_samples[16][16]; /* [x][y]      */
  isize luma8x8_pred_samples[4][8][8]; /* [blk][x][y] */
  isize chroma_pred_samples[8][16];    /* [x][y]      */
  isize block_luma_dc[16];
  isize block_luma_ac[16][15];
  isize block_luma_4x4[16][16];
  isize block_luma_8x8[4][64];
  isize block_chroma_dc[2][8];
  isize block_chroma_ac[2][8][15];
} Macroblock;

typedef struct {
  const dryv_frame_params *fp;
  isize curr_mb_addr;
  isize first_mb_in_slice; /* always 0: one slice per picture */
  isize pic_width_in_mbs, pic_height_in_mbs, pic_size_in_mbs;
  isize pic_width_in_samples_l, pic_height_in_samples_l;
  isize pic_width_in_samples_c, pic_height_in_samples_c;
  isize mb_width_c, mb_height_c, sub_width_c, sub_height_c; /* header.rs:168-183: 8,8,2,2 for 4:2:0 */
  isize chroma_array_type, bit_depth_y, bit_depth_c, qp_bd_offset_c;
  isize scaling_list4x4[6][16];
  isize scaling_list8x8[6][64];
  MbRec *macroblocks; /* slice.macroblocks, one per mbaddr                                     */
  MbRec unavailable;  /* Macroblock::unavailable(0)                                            */
  Macroblock mb;      /* heavy part of slice.mb()                                              */
} Slice;

/* frame/mod.rs:16-26 — planes are indexed [x][y] (column-major), as in the reference. */
typedef struct {
  uint8_t *luma_data, *chroma_cb_data, *chroma_cr_data;
  isize level_scale4x4[6][4][4];
  isize level_scale8x8[6][8][8];
  isize width_l, height_l, width_c, height_c;
} Frame;

#define LUMA(f, x, y) ((f)->luma_data[(size_t)(x) * (size_t)(f)->height_l + (size_t)(y)])
#define CB(f, x, y) ((f)->chroma_cb_data[(size_t)(x) * (size_t)(f)->height_c + (size_t)(y)])
#define CR(f, x, y) ((f)->chroma_cr_data[(size_t)(x) * (size_t)(f)->height_c + (size_t)(y)])

static MbRec *slice_mb(Slice *s) { return &s->macroblocks[s->curr_mb_addr]; } /* slice/mod.rs:176 */

/* ------------------------------------------------------------------------------------------- */
/* Neighbour geometry: slice/macroblock.rs:447-477, slice/mod.rs:576-622                       */
/* ------------------------------------------------------------------------------------------- */

enum { POS_NONE = 0, POS_THIS, POS_A, POS_B, POS_C, POS_D };

/* macroblock.rs:448-462 */
static int mbpos_from_coords(isize x, isize y, isize max_w, isize max_h) {
  if (x < 0 && y < 0) return POS_D;
  if (x < 0 && (y >= 0 && y < max_h)) return POS_A;
  if ((x >= 0 && x < max_w) && y < 0) return POS_B;
  if (x > max_w - 1 && y < 0) return POS_C;
  if ((x >= 0 && x < max_w) && (y >= 0 && y < max_h)) return POS_THIS;
  return POS_NONE;
}

/* macroblock.rs:464-466 */
static void mbpos_coords(isize x, isize y, isize max_w, isize max_h, isize *xw, isize *yw) {
  *xw = (x + max_w) % max_w;
  *yw = (y + max_h) % max_h;
}

/* macroblock.rs:468-471 (+ :136-141: -1 when the macroblock is unavailable) */
static isize mb_blk_idx4x4(const MbRec *mb, isize x, isize y, isize max_w, isize max_h) {
  if (mb->unavailable) return -1;
  isize xw, yw;
  mbpos_coords(x, y, max_w, max_h, &xw, &yw);
  return 8 * (yw / 8) + 4 * (xw / 8) + 2 * ((yw % 8) / 4) + ((xw % 8) / 4);
}

/* macroblock.rs:473-476 (+ :143-148) */
static isize mb_blk_idx8x8(const MbRec *mb, isize x, isize y, isize max_w, isize max_h) {
  if (mb->unavailable) return -1;
  isize xw, yw;
  mbpos_coords(x, y, max_w, max_h, &xw, &yw);
  return 2 * (yw / 8) + (xw / 8);
}

/* slice/mod.rs:615-622 — no slice groups in scope, so the group test is always equal. */
static int mb_available(const Slice *s, isize mbaddr) {
  if (mbaddr < s->first_mb_in_slice || mbaddr > s->curr_mb_addr) return 0;
  return 1;
}

/* slice/mod.rs:576-613, non-MBAFF. Returns the neighbour record or the static unavailable one. */
static const MbRec *mb_nb_p(const Slice *s, int position) {
  isize mbaddr = s->curr_mb_addr;
  isize w = s->pic_width_in_mbs;
  switch (position) {
    case POS_THIS:
      return &s->macroblocks[s->curr_mb_addr];
    case POS_A:
      if ((mbaddr % w) == 0) return &s->unavailable;
      mbaddr -= 1;
      break;
    case POS_B:
      mbaddr -= w;
      break;
    case POS_C:
      if (((mbaddr + 1) % w) == 0) return &s->unavailable;
      mbaddr -= w - 1;
      break;
    case POS_D:
      if ((mbaddr % w) == 0) return &s->unavailable;
      mbaddr -= w + 1;
      break;
    default:
      return &s->unavailable;
  }
  if (!mb_available(s, mbaddr)) return &s->unavailable;
  return &s->macroblocks[mbaddr];
}

/* `pos.map(|pos| slice.mb_nb_p(pos, 0)).unwrap_or(unavailable)` — the idiom at pred4x4.rs:33-36 */
static const MbRec *mb_at(const Slice *s, isize x, isize y, isize max_w, isize max_h) {
  int pos = mbpos_from_coords(x, y, max_w, max_h);
  if (pos == POS_NONE) return &s->unavailable;
  return mb_nb_p(s, pos);
}

/* Macroblock::index (macroblock.rs:225-232): pointer offset into slice.macroblocks */
static isize mb_index(const Slice *s, const MbRec *mb) { return (isize)(mb - s->macroblocks); }

/* ------------------------------------------------------------------------------------------- */
/* frame/mod.rs:185-284 — inverse zig-zag scans (c[row][col])                                  */
/* ------------------------------------------------------------------------------------------- */

static void inverse_scanner4x4(const isize value[16], isize c[4][4]) {
  c[0][0] = value[0];
  c[0][1] = value[1];
  c[1][0] = value[2];
  c[2][0] = value[3];
  c[1][1] = value[4];
  c[0][2] = value[5];
  c[0][3] = value[6];
  c[1][2] = value[7];
  c[2][1] = value[8];
  c[3][0] = value[9];
  c[3][1] = value[10];
  c[2][2] = value[11];
  c[1][3] = value[12];
  c[2][3] = value[13];
  c[3][2] = value[14];
  c[3][3] = value[15];
}

/* frame/mod.rs:212-284: (row, col) of list position k */
static const uint8_t ZZ8[64][2] = {
    {0, 0}, {0, 1}, {1, 0}, {2, 0}, {1, 1}, {0, 2}, {0, 3}, {1, 2}, {2, 1}, {3, 0}, {4, 0},
    {3, 1}, {2, 2}, {1, 3}, {0, 4}, {0, 5}, {1, 4}, {2, 3}, {3, 2}, {4, 1}, {5, 0}, {6, 0},
    {5, 1}, {4, 2}, {3, 3}, {2, 4}, {1, 5}, {0, 6}, {0, 7}, {1, 6}, {2, 5}, {3, 4}, {4, 3},
    {5, 2}, {6, 1}, {7, 0}, {7, 1}, {6, 2}, {5, 3}, {4, 4}, {3, 5}, {2, 6}, {1, 7}, {2, 7},
    {3, 6}, {4, 5}, {5, 4}, {6, 3}, {7, 2}, {7, 3}, {6, 4}, {5, 5}, {4, 6}, {3, 7}, {4, 7},
    {5, 6}, {6, 5}, {7, 4}, {7, 5}, {6, 6}, {5, 7}, {6, 7}, {7, 6}, {7, 7}};

static void inverse_scanner_8x8(const isize value[64], isize c[8][8]) {
  for (int k = 0; k < 64; k++) c[ZZ8[k][0]][ZZ8[k][1]] = value[k];
}

/* ------------------------------------------------------------------------------------------- */
/* frame/transform.rs                                                                          */
/* ------------------------------------------------------------------------------------------- */

/* transform.rs:8-78 — 8.5.9. mb_is_inter_flag is always false here (intra only); color_plane_id
 * is None. QUIRK Q3: the path only ever calls this with is_luma = true, so i_y_cb_cr = 0 and the
 * chroma blocks later read the luma tables. */
static void frame_scaling(Frame *f, const Slice *s, int is_luma, int is_chroma_cb) {
  int i_y_cb_cr = is_luma ? 0 : (is_chroma_cb ? 1 : 2);
  int idx = i_y_cb_cr + 0;
  isize weight_scale4x4[4][4];
  inverse_scanner4x4(s->scaling_list4x4[idx], weight_scale4x4);

  static const isize V4X4[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16},
                                   {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
  for (int m = 0; m < 6; m++)
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        if (i % 2 == 0 && j % 2 == 0)
          f->level_scale4x4[m][i][j] = weight_scale4x4[i][j] * V4X4[m][0];
        else if (i % 2 == 1 && j % 2 == 1)
          f->level_scale4x4[m][i][j] = weight_scale4x4[i][j] * V4X4[m][1];
        else
          f->level_scale4x4[m][i][j] = weight_scale4x4[i][j] * V4X4[m][2];
      }

  idx = 2 * i_y_cb_cr + 0;
  isize weight_scale8x8[8][8];
  inverse_scanner_8x8(s->scaling_list8x8[idx], weight_scale8x8);

  static const isize V8X8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26},
                                   {26, 23, 42, 24, 33, 31}, {28, 25, 45, 26, 35, 33},
                                   {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
  for (int m = 0; m < 6; m++)
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) {
        isize v;
        if (i % 4 == 0 && j % 4 == 0)
          v = V8X8[m][0];
        else if (i % 2 == 1 && j % 2 == 1)
          v = V8X8[m][1];
        else if (i % 4 == 2 && j % 4 == 2)
          v = V8X8[m][2];
        else if ((i % 4 == 0 && j % 2 == 1) || (i % 2 == 1 && j % 4 == 0))
          v = V8X8[m][3];
        else if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0))
          v = V8X8[m][4];
        else
          v = V8X8[m][5];
        f->level_scale8x8[m][i][j] = weight_scale8x8[i][j] * v;
      }
}

/* transform.rs:194-216 — 8.5.8 / table 8-15 */
static isize get_qpc(const Slice *s, isize qpy, int is_chroma_cb) {
  isize qp_offset =
      is_chroma_cb ? s->fp->chroma_qp_index_offset : s->fp->second_chroma_qp_index_offset;
  isize qpi = clamp(qpy + qp_offset, -s->qp_bd_offset_c, 51);
  if (qpi < 30) return qpi;
  static const isize QPCS[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36,
                                 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
  return QPCS[qpi - 30];
}

/* transform.rs:218-226 (slice_type is I: the switching branch never runs) */
static void chroma_quantization_parameters(Slice *s, int is_chroma_cb) {
  s->mb.qpc = get_qpc(s, s->mb.qpy, is_chroma_cb);
  s->mb.qp1c = s->mb.qpc + s->qp_bd_offset_c;
}

/* transform.rs:116-191 — 8.5.12. transform_bypass_mode_flag and s_mb_flag are false in scope. */
static void scaling_and_transform4x4(const Frame *f, Slice *s, isize c[4][4], int is_luma,
                                     int is_chroma_cb, isize r[4][4]) {
  chroma_quantization_parameters(s, is_chroma_cb);
  isize q_p = is_luma ? s->mb.qp1y : s->mb.qp1c;
  int is_intra_16x16 = slice_mb(s)->mode == MODE_INTRA16X16;

  isize d[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      if ((is_intra_16x16 || !is_luma) && j == 0 && i == 0) {
        d[0][0] = c[0][0];
      } else if (q_p >= 24) {
        d[i][j] = (c[i][j] * f->level_scale4x4[q_p % 6][i][j]) * ((isize)1 << (q_p / 6 - 4));
      } else {
        d[i][j] = (c[i][j] * f->level_scale4x4[q_p % 6][i][j] + ((isize)1 << (3 - q_p / 6))) >>
                  (4 - q_p / 6);
      }
    }

  isize ff[4][4], h[4][4];
  for (int i = 0; i < 4; i++) {
    isize ei0 = d[i][0] + d[i][2];
    isize ei1 = d[i][0] - d[i][2];
    isize ei2 = (d[i][1] >> 1) - d[i][3];
    isize ei3 = d[i][1] + (d[i][3] >> 1);
    ff[i][0] = ei0 + ei3;
    ff[i][1] = ei1 + ei2;
    ff[i][2] = ei1 - ei2;
    ff[i][3] = ei0 - ei3;
  }
  for (int j = 0; j <= 3; j++) {
    isize g0j = ff[0][j] + ff[2][j];
    isize g1j = ff[0][j] - ff[2][j];
    isize g2j = (ff[1][j] >> 1) - ff[3][j];
    isize g3j = ff[1][j] + (ff[3][j] >> 1);
    h[0][j] = g0j + g3j;
    h[1][j] = g1j + g2j;
    h[2][j] = g1j - g2j;
    h[3][j] = g0j - g3j;
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r[i][j] = (h[i][j] + 32) >> 6;
}

/* ------------------------------------------------------------------------------------------- */
/* frame/mod.rs:93-165 — 8.5.14 picture construction                                           */
/* ------------------------------------------------------------------------------------------- */

enum { B16x16, B8x8, B4x4 };

static void picture_construction(Frame *f, const Slice *s, const isize *u, int blk_type,
                                 isize blk_idx, int is_luma, int is_chroma_cb) {
  isize x_p = inverse_raster_scan(s->curr_mb_addr, 16, 16, s->pic_width_in_samples_l, 0);
  isize y_p = inverse_raster_scan(s->curr_mb_addr, 16, 16, s->pic_width_in_samples_l, 1);
  isize x_o = 0, y_o = 0;
  if (is_luma) {
    isize n_e;
    if (blk_type == B16x16) {
      x_o = 0;
      y_o = 0;
      n_e = 16;
    } else if (blk_type == B4x4) {
      x_o = inverse_raster_scan(blk_idx / 4, 8, 8, 16, 0) + inverse_raster_scan(blk_idx % 4, 4, 4, 8, 0);
      y_o = inverse_raster_scan(blk_idx / 4, 8, 8, 16, 1) + inverse_raster_scan(blk_idx % 4, 4, 4, 8, 1);
      n_e = 4;
    } else {
      x_o = inverse_raster_scan(blk_idx, 8, 8, 16, 0);
      y_o = inverse_raster_scan(blk_idx, 8, 8, 16, 1);
      n_e = 8;
    }
    for (isize i = 0; i < n_e; i++)
      for (isize j = 0; j < n_e; j++) {
        isize x = x_p + x_o + j;
        isize y = y_p + y_o + i;
        LUMA(f, x, y) = (uint8_t)u[i * n_e + j];
      }
  } else {
    isize mb_width_c = s->mb_width_c, mb_height_c = s->mb_height_c;
    if (s->chroma_array_type == 1 || s->chroma_array_type == 2) {
      for (isize i = 0; i < mb_width_c; i++)
        for (isize j = 0; j < mb_height_c; j++) {
          isize x = x_p / s->sub_width_c + x_o + j;
          isize y = y_p / s->sub_height_c + y_o + i;
          uint8_t v = (uint8_t)u[i * mb_width_c + j];
          if (is_chroma_cb)
            CB(f, x, y) = v;
          else
            CR(f, x, y) = v;
        }
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* frame/pred4x4.rs                                                                            */
/* ------------------------------------------------------------------------------------------- */

/* pred4x4.rs:363-427 — 8.3.1.1 */
static void intra4x4_pred_mode(Slice *s, isize luma4x4_block_idx) {
  const isize INTRA4X4_DC = 2;
  isize x = inverse_raster_scan(luma4x4_block_idx / 4, 8, 8, 16, 0) +
            inverse_raster_scan(luma4x4_block_idx % 4, 4, 4, 8, 0);
  isize y = inverse_raster_scan(luma4x4_block_idx / 4, 8, 8, 16, 1) +
            inverse_raster_scan(luma4x4_block_idx % 4, 4, 4, 8, 1);
  const isize max_w = 16, max_h = 16;

  const MbRec *mb_a = mb_at(s, x - 1, y, max_w, max_h);
  isize idx_a = mb_blk_idx4x4(mb_a, x - 1, y, max_w, max_h);
  const MbRec *mb_b = mb_at(s, x, y - 1, max_w, max_h);
  isize idx_b = mb_blk_idx4x4(mb_b, x, y - 1, max_w, max_h);

  /* constrained_intra_pred only matters for inter neighbours, which do not exist here */
  int dc_pred_mode_predicted_flag = mb_a->unavailable || mb_b->unavailable;

  isize mode_a, mode_b;
  if (dc_pred_mode_predicted_flag || (mb_a->mode != MODE_INTRA4X4 && mb_a->mode != MODE_INTRA8X8))
    mode_a = INTRA4X4_DC;
  else if (mb_a->mode == MODE_INTRA4X4)
    mode_a = mb_a->intra4x4_pred_mode[idx_a];
  else
    mode_a = mb_a->intra8x8_pred_mode[idx_a >> 2];

  if (dc_pred_mode_predicted_flag || (mb_b->mode != MODE_INTRA4X4 && mb_b->mode != MODE_INTRA8X8))
    mode_b = INTRA4X4_DC;
  else if (mb_b->mode == MODE_INTRA4X4)
    mode_b = mb_b->intra4x4_pred_mode[idx_b];
  else
    mode_b = mb_b->intra8x8_pred_mode[idx_b >> 2];

  isize pred = mode_a < mode_b ? mode_a : mode_b;
  MbRec *cur = slice_mb(s);
  if (s->mb.prev_intra4x4_pred_mode_flag[luma4x4_block_idx] != 0)
    cur->intra4x4_pred_mode[luma4x4_block_idx] = pred;
  else if ((isize)s->mb.rem_intra4x4_pred_mode[luma4x4_block_idx] < pred)
    cur->intra4x4_pred_mode[luma4x4_block_idx] = s->mb.rem_intra4x4_pred_mode[luma4x4_block_idx];
  else
    cur->intra4x4_pred_mode[luma4x4_block_idx] =
        (isize)s->mb.rem_intra4x4_pred_mode[luma4x4_block_idx] + 1;
}

/* SampleP for the 9-wide grid (pred4x4.rs:430-434, trans_chroma.rs:459-463) */
#define P9(a, x, y) ((a)[((y) + 1) * 9 + ((x) + 1)])
/* SampleP for the 17-wide grid (pred8x8.rs:767-771, pred16x16.rs:485-489) */
#define P17(a, x, y) ((a)[((y) + 1) * 17 + ((x) + 1)])

/* pred4x4.rs:10-360 — 8.3.1.2 */
static void intra4x4_prediction(Frame *f, Slice *s, isize blk) {
  static const isize RX[13] = {-1, -1, -1, -1, -1, 0, 1, 2, 3, 4, 5, 6, 7};
  static const isize RY[13] = {-1, 0, 1, 2, 3, -1, -1, -1, -1, -1, -1, -1, -1};

  isize x_o = inverse_raster_scan(blk / 4, 8, 8, 16, 0) + inverse_raster_scan(blk % 4, 4, 4, 8, 0);
  isize y_o = inverse_raster_scan(blk / 4, 8, 8, 16, 1) + inverse_raster_scan(blk % 4, 4, 4, 8, 1);

  isize samples[45];
  for (int i = 0; i < 45; i++) samples[i] = -1;

  for (int i = 0; i < 13; i++) {
    isize x = RX[i], y = RY[i];
    isize x_n = x_o + x, y_n = y_o + y;
    const isize max_w = 16, max_h = 16;
    const MbRec *mb_n = mb_at(s, x_n, y_n, max_w, max_h);
    isize x_w, y_w;
    mbpos_coords(x_n, y_n, max_w, max_h, &x_w, &y_w);
    if (mb_n->unavailable || ((x > 3) && (blk == 3 || blk == 11))) {
      P9(samples, x, y) = -1;
    } else {
      isize mbaddr_n = mb_index(s, mb_n);
      isize x_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 0);
      isize y_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 1);
      P9(samples, x, y) = LUMA(f, x_m + x_w, y_m + y_w);
    }
  }

  if (P9(samples, 4, -1) < 0 && P9(samples, 5, -1) < 0 && P9(samples, 6, -1) < 0 &&
      P9(samples, 7, -1) < 0 && P9(samples, 3, -1) >= 0) {
    P9(samples, 4, -1) = P9(samples, 3, -1);
    P9(samples, 5, -1) = P9(samples, 3, -1);
    P9(samples, 6, -1) = P9(samples, 3, -1);
    P9(samples, 7, -1) = P9(samples, 3, -1);
  }

  intra4x4_pred_mode(s, blk);

  isize mode = slice_mb(s)->intra4x4_pred_mode[blk];
  isize(*pred)[4] = s->mb.luma_pred_samples[blk]; /* [x][y] */
#define P(x, y) P9(samples, (x), (y))
  int top4 = P(0, -1) >= 0 && P(1, -1) >= 0 && P(2, -1) >= 0 && P(3, -1) >= 0;
  int left4 = P(-1, 0) >= 0 && P(-1, 1) >= 0 && P(-1, 2) >= 0 && P(-1, 3) >= 0;
  int tr4 = P(4, -1) >= 0 && P(5, -1) >= 0 && P(6, -1) >= 0 && P(7, -1) >= 0;

  if (mode == 0) { /* vertical :92-103 */
    if (top4)
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) pred[x][y] = P(x, -1);
  } else if (mode == 1) { /* horizontal :104-115 */
    if (left4)
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) pred[x][y] = P(-1, y);
  } else if (mode == 2) { /* DC :116-167 */
    isize val;
    if (top4 && left4)
      val = (P(0, -1) + P(1, -1) + P(2, -1) + P(3, -1) + P(-1, 0) + P(-1, 1) + P(-1, 2) + P(-1, 3) + 4) >> 3;
    else if (!top4 && left4)
      val = (P(-1, 0) + P(-1, 1) + P(-1, 2) + P(-1, 3) + 2) >> 2;
    else if (top4 && !left4)
      val = (P(0, -1) + P(1, -1) + P(2, -1) + P(3, -1) + 2) >> 2;
    else
      val = (isize)1 << (s->bit_depth_y - 1);
    for (int x = 0; x < 4; x++)
      for (int y = 0; y < 4; y++) pred[x][y] = val;
  } else if (mode == 3) { /* diagonal down-left :168-193 */
    if (top4 && tr4)
      for (isize y = 0; y < 4; y++)
        for (isize x = 0; x < 4; x++) {
          if (x == 3 && y == 3)
            pred[x][y] = (P(6, -1) + 3 * P(7, -1) + 2) >> 2;
          else
            pred[x][y] = (P(x + y, -1) + 2 * P(x + y + 1, -1) + P(x + y + 2, -1) + 2) >> 2;
        }
  } else if (mode == 4) { /* diagonal down-right :194-231 */
    if (top4 && P(-1, -1) >= 0 && left4)
      for (isize y = 0; y <= 3; y++)
        for (isize x = 0; x <= 3; x++) {
          if (x > y)
            pred[x][y] = (P(x - y - 2, -1) + 2 * P(x - y - 1, -1) + P(x - y, -1) + 2) >> 2;
          else if (x < y)
            pred[x][y] = (P(-1, y - x - 2) + 2 * P(-1, y - x - 1) + P(-1, y - x) + 2) >> 2;
          else
            pred[x][y] = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2;
        }
  } else if (mode == 5) { /* vertical-right :232-267 */
    if (top4 && P(-1, -1) >= 0 && left4)
      for (isize y = 0; y <= 3; y++)
        for (isize x = 0; x <= 3; x++) {
          isize z_vr = 2 * x - y;
          if (z_vr == 0 || z_vr == 2 || z_vr == 4 || z_vr == 6)
            pred[x][y] = (P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 1) >> 1;
          else if (z_vr == 1 || z_vr == 3 || z_vr == 5)
            pred[x][y] =
                (P(x - (y >> 1) - 2, -1) + 2 * P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 2) >> 2;
          else if (z_vr == -1)
            pred[x][y] = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
          else
            pred[x][y] = (P(-1, y - 1) + 2 * P(-1, y - 2) + P(-1, y - 3) + 2) >> 2;
        }
  } else if (mode == 6) { /* horizontal-down :268-303 */
    if (top4 && P(-1, -1) >= 0 && left4)
      for (isize y = 0; y <= 3; y++)
        for (isize x = 0; x <= 3; x++) {
          isize z_hd = 2 * y - x;
          if (z_hd == 0 || z_hd == 2 || z_hd == 4 || z_hd == 6)
            pred[x][y] = (P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 1) >> 1;
          else if (z_hd == 1 || z_hd == 3 || z_hd == 5)
            pred[x][y] =
                (P(-1, y - (x >> 1) - 2) + 2 * P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 2) >> 2;
          else if (z_hd == -1)
            pred[x][y] = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
          else
            pred[x][y] = (P(x - 1, -1) + 2 * P(x - 2, -1) + P(x - 3, -1) + 2) >> 2;
        }
  } else if (mode == 7) { /* vertical-left :304-329 */
    if (top4 && tr4)
      for (isize y = 0; y <= 3; y++)
        for (isize x = 0; x <= 3; x++) {
          if (y == 0 || y == 2)
            pred[x][y] = (P(x + (y >> 1), -1) + P(x + (y >> 1) + 1, -1) + 1) >> 1;
          else
            pred[x][y] =
                (P(x + (y >> 1), -1) + 2 * P(x + (y >> 1) + 1, -1) + P(x + (y >> 1) + 2, -1) + 2) >> 2;
        }
  } else if (mode == 8 && left4) { /* horizontal-up :330-359 */
    for (isize y = 0; y <= 3; y++)
      for (isize x = 0; x <= 3; x++) {
        isize z_hu = x + 2 * y;
        if (z_hu == 0 || z_hu == 2 || z_hu == 4)
          pred[x][y] = (P(-1, y + (x >> 1)) + P(-1, y + (x >> 1) + 1) + 1) >> 1;
        else if (z_hu == 1 || z_hu == 3)
          pred[x][y] =
              (P(-1, y + (x >> 1)) + 2 * P(-1, y + (x >> 1) + 1) + P(-1, y + (x >> 1) + 2) + 2) >> 2;
        else if (z_hu == 5)
          pred[x][y] = (P(-1, 2) + 3 * P(-1, 3) + 2) >> 2;
        else
          pred[x][y] = P(-1, 3);
      }
  }
  /* QUIRK Q4: when a mode's reference samples are missing, pred keeps its zero initial value. */
#undef P
}

/* transform.rs:81-113 — 8.5.1 */
static void transform_for_4x4_luma_residual_blocks(Frame *f, Slice *s) {
  frame_scaling(f, s, 1, 0);
  for (isize blk = 0; blk < 16; blk++) {
    isize c[4][4], r[4][4];
    inverse_scanner4x4(s->mb.block_luma_4x4[blk], c);
    scaling_and_transform4x4(f, s, c, 1, 0, r);
    intra4x4_prediction(f, s, blk);
    isize u[16];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
        u[i * 4 + j] =
            clamp(s->mb.luma_pred_samples[blk][j][i] + r[i][j], 0, ((isize)1 << s->bit_depth_y) - 1);
    picture_construction(f, s, u, B4x4, blk, 1, 0);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* frame/pred8x8.rs                                                                            */
/* ------------------------------------------------------------------------------------------- */

/* pred8x8.rs:51-150 — 8.5.13 */
static void scaling_and_transform8x8(const Frame *f, Slice *s, isize c[8][8], isize r[8][8]) {
  chroma_quantization_parameters(s, 0);
  isize q_p = s->mb.qp1y;
  isize d[8][8];
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      if (q_p >= 36)
        d[i][j] = (c[i][j] * f->level_scale8x8[q_p % 6][i][j]) * ((isize)1 << (q_p / 6 - 6));
      else
        d[i][j] = (c[i][j] * f->level_scale8x8[q_p % 6][i][j] + ((isize)1 << (5 - q_p / 6))) >>
                  (6 - q_p / 6);
    }

  isize g[8][8], m[8][8];
  for (int i = 0; i < 8; i++) {
    isize ei0 = d[i][0] + d[i][4];
    isize ei1 = -d[i][3] + d[i][5] - d[i][7] - (d[i][7] >> 1);
    isize ei2 = d[i][0] - d[i][4];
    isize ei3 = d[i][1] + d[i][7] - d[i][3] - (d[i][3] >> 1);
    isize ei4 = (d[i][2] >> 1) - d[i][6];
    isize ei5 = -d[i][1] + d[i][7] + d[i][5] + (d[i][5] >> 1);
    isize ei6 = d[i][2] + (d[i][6] >> 1);
    isize ei7 = d[i][3] + d[i][5] + d[i][1] + (d[i][1] >> 1);

    isize fi0 = ei0 + ei6;
    isize fi1 = ei1 + (ei7 >> 2);
    isize fi2 = ei2 + ei4;
    isize fi3 = ei3 + (ei5 >> 2);
    isize fi4 = ei2 - ei4;
    isize fi5 = (ei3 >> 2) - ei5;
    isize fi6 = ei0 - ei6;
    isize fi7 = ei7 - (ei1 >> 2);

    g[i][0] = fi0 + fi7;
    g[i][1] = fi2 + fi5;
    g[i][2] = fi4 + fi3;
    g[i][3] = fi6 + fi1;
    g[i][4] = fi6 - fi1;
    g[i][5] = fi4 - fi3;
    g[i][6] = fi2 - fi5;
    g[i][7] = fi0 - fi7;
  }
  for (int j = 0; j < 8; j++) {
    isize h0j = g[0][j] + g[4][j];
    isize h1j = -g[3][j] + g[5][j] - g[7][j] - (g[7][j] >> 1);
    isize h2j = g[0][j] - g[4][j];
    isize h3j = g[1][j] + g[7][j] - g[3][j] - (g[3][j] >> 1);
    isize h4j = (g[2][j] >> 1) - g[6][j];
    isize h5j = -g[1][j] + g[7][j] + g[5][j] + (g[5][j] >> 1);
    isize h6j = g[2][j] + (g[6][j] >> 1);
    isize h7j = g[3][j] + g[5][j] + g[1][j] + (g[1][j] >> 1);

    isize k0j = h0j + h6j;
    isize k1j = h1j + (h7j >> 2);
    isize k2j = h2j + h4j;
    isize k3j = h3j + (h5j >> 2);
    isize k4j = h2j - h4j;
    isize k5j = (h3j >> 2) - h5j;
    isize k6j = h0j - h6j;
    isize k7j = h7j - (h1j >> 2);

    m[0][j] = k0j + k7j;
    m[1][j] = k2j + k5j;
    m[2][j] = k4j + k3j;
    m[3][j] = k6j + k1j;
    m[4][j] = k6j - k1j;
    m[5][j] = k4j - k3j;
    m[6][j] = k2j - k5j;
    m[7][j] = k0j - k7j;
  }
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) r[i][j] = (m[i][j] + 32) >> 6;
}

/* pred8x8.rs:698-764 — 8.3.2.1 */
static void intra8x8_pred_mode(Slice *s, isize blk8) {
  const isize max_w = 16, max_h = 16;
  isize x = (blk8 % 2) * 8;
  isize y = (blk8 / 2) * 8;

  const MbRec *mb_a = mb_at(s, x - 1, y, max_w, max_h);
  isize idx_a = mb_blk_idx8x8(mb_a, x - 1, y, max_w, max_h);
  const MbRec *mb_b = mb_at(s, x, y - 1, max_w, max_h);
  isize idx_b = mb_blk_idx8x8(mb_b, x, y - 1, max_w, max_h);

  int dc_pred_mode_predicted_flag = mb_a->unavailable || mb_b->unavailable;
  isize mode_a, mode_b;
  if (dc_pred_mode_predicted_flag || (mb_a->mode != MODE_INTRA4X4 && mb_a->mode != MODE_INTRA8X8))
    mode_a = 2;
  else if (mb_a->mode == MODE_INTRA8X8)
    mode_a = mb_a->intra8x8_pred_mode[idx_a];
  else
    mode_a = mb_a->intra4x4_pred_mode[idx_a * 4 + 1];

  if (dc_pred_mode_predicted_flag || (mb_b->mode != MODE_INTRA4X4 && mb_b->mode != MODE_INTRA8X8))
    mode_b = 2;
  else if (mb_b->mode == MODE_INTRA8X8)
    mode_b = mb_b->intra8x8_pred_mode[idx_b];
  else
    mode_b = mb_b->intra4x4_pred_mode[idx_b * 4 + 2];

  isize pred = mode_a < mode_b ? mode_a : mode_b;
  MbRec *cur = slice_mb(s);
  if (s->mb.prev_intra8x8_pred_mode_flag[blk8] != 0)
    cur->intra8x8_pred_mode[blk8] = pred;
  else if ((isize)s->mb.rem_intra8x8_pred_mode[blk8] < pred)
    cur->intra8x8_pred_mode[blk8] = s->mb.rem_intra8x8_pred_mode[blk8];
  else
    cur->intra8x8_pred_mode[blk8] = (isize)s->mb.rem_intra8x8_pred_mode[blk8] + 1;
}

/* pred8x8.rs:152-696 — 8.3.2.2. Returns -1 where the reference would panic (:694). */
static int intra8x8_prediction(Frame *f, Slice *s, isize blk8) {
  static const isize RX[25] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 1,  2, 3,
                               4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15};
  static const isize RY[25] = {-1, 0,  1,  2,  3,  4,  5,  6,  7,  -1, -1, -1, -1,
                               -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
  isize p[9 * 17], p1[9 * 17];
  for (int i = 0; i < 9 * 17; i++) p[i] = p1[i] = -1;

  isize x_o = inverse_raster_scan(blk8, 8, 8, 16, 0);
  isize y_o = inverse_raster_scan(blk8, 8, 8, 16, 1);

  for (int i = 0; i < 25; i++) {
    const isize max_w = 16, max_h = 16;
    isize x = RX[i], y = RY[i];
    isize x_n = x_o + x, y_n = y_o + y;
    const MbRec *mb_n = mb_at(s, x_n, y_n, max_w, max_h);
    isize x_w, y_w;
    mbpos_coords(x_n, y_n, max_w, max_h, &x_w, &y_w);
    if (mb_n->unavailable) {
      P17(p, x, y) = -1;
    } else {
      isize mbaddr_n = mb_index(s, mb_n);
      isize x_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 0);
      isize y_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 1);
      P17(p, x, y) = LUMA(f, x_m + x_w, y_m + y_w);
    }
  }
#define P(x, y) P17(p, (x), (y))
#define P1(x, y) P17(p1, (x), (y))
  /* :202-220 top-right substitution */
  {
    int all_neg = 1;
    for (int x = 8; x <= 15; x++) all_neg = all_neg && (P(x, -1) < 0);
    if (all_neg && P(7, -1) >= 0)
      for (int x = 8; x <= 15; x++) P(x, -1) = P(7, -1);
  }
  /* :222-250 filter the top row */
  {
    int all16 = 1;
    for (int x = 0; x <= 15; x++) all16 = all16 && (P(x, -1) >= 0);
    if (all16) {
      if (P(-1, -1) >= 0)
        P1(0, -1) = (P(-1, -1) + 2 * P(0, -1) + P(1, -1) + 2) >> 2;
      else
        P1(0, -1) = (3 * P(0, -1) + P(1, -1) + 2) >> 2;
      /* QUIRK Q1: the loop starts at x = 0 (spec: 1) and overwrites p'[0,-1] using p[-1,-1],
       * which is -1 when the corner is unavailable (pred8x8.rs:245-247). */
      for (isize x = 0; x < 15; x++) P1(x, -1) = (P(x - 1, -1) + 2 * P(x, -1) + P(x + 1, -1) + 2) >> 2;
      P1(15, -1) = (P(14, -1) + 3 * P(15, -1) + 2) >> 2;
    }
  }
  /* :252-264 filter the corner */
  if (P(-1, -1) >= 0) {
    if (P(0, -1) < 0 || P(-1, 0) < 0) {
      if (P(0, -1) >= 0)
        P1(-1, -1) = (3 * P(-1, -1) + P(0, -1) + 2) >> 2;
      else if (P(0, -1) < 0 && P(-1, 0) >= 0)
        P1(-1, -1) = (3 * P(-1, -1) + P(-1, 0) + 2) >> 2;
      else
        P1(-1, -1) = P(-1, -1);
    } else {
      P1(-1, -1) = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2;
    }
  }
  /* :266-286 filter the left column */
  {
    int all8 = 1;
    for (int y = 0; y <= 7; y++) all8 = all8 && (P(-1, y) >= 0);
    if (all8) {
      if (P(-1, -1) >= 0)
        P1(-1, 0) = (P(-1, -1) + 2 * P(-1, 0) + P(-1, 1) + 2) >> 2;
      else
        P1(-1, 0) = (3 * P(-1, 0) + P(-1, 1) + 2) >> 2;
      for (isize y = 1; y < 7; y++) P1(-1, y) = (P(-1, y - 1) + 2 * P(-1, y) + P(-1, y + 1) + 2) >> 2;
      P1(-1, 7) = (P(-1, 6) + 3 * P(-1, 7) + 2) >> 2;
    }
  }
  memcpy(p, p1, sizeof(p)); /* :288 */

  intra8x8_pred_mode(s, blk8);
  isize mode = slice_mb(s)->intra8x8_pred_mode[blk8];
  isize(*pred)[8] = s->mb.luma8x8_pred_samples[blk8]; /* [x][y] */

  int top8 = 1, left8 = 1, tr8 = 1;
  for (int x = 0; x <= 7; x++) top8 = top8 && (P(x, -1) >= 0);
  for (int x = 8; x <= 15; x++) tr8 = tr8 && (P(x, -1) >= 0);
  for (int y = 0; y <= 7; y++) left8 = left8 && (P(-1, y) >= 0);
  int corner = P(-1, -1) >= 0;

  if (mode == 0) {
    if (top8)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) pred[x][y] = P(x, -1);
  } else if (mode == 1) {
    if (left8)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) pred[x][y] = P(-1, y);
  } else if (mode == 2) {
    isize val;
    if (top8 && left8) {
      val = 8;
      for (int k = 0; k < 8; k++) val += P(k, -1) + P(-1, k);
      val >>= 4;
    } else if (!top8 && left8) {
      val = 4;
      for (int k = 0; k < 8; k++) val += P(-1, k);
      val >>= 3;
    } else if (top8 && !left8) {
      val = 4;
      for (int k = 0; k < 8; k++) val += P(k, -1);
      val >>= 3;
    } else {
      val = (isize)1 << (s->bit_depth_y - 1);
    }
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++) pred[x][y] = val;
  } else if (mode == 3) {
    if (top8 && tr8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          if (x == 7 && y == 7)
            pred[x][y] = (P(14, -1) + 3 * P(15, -1) + 2) >> 2;
          else
            pred[x][y] = (P(x + y, -1) + 2 * P(x + y + 1, -1) + P(x + y + 2, -1) + 2) >> 2;
        }
  } else if (mode == 4) {
    if (top8 && corner && left8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          if (x > y)
            pred[x][y] = (P(x - y - 2, -1) + 2 * P(x - y - 1, -1) + P(x - y, -1) + 2) >> 2;
          else if (x < y)
            pred[x][y] = (P(-1, y - x - 2) + 2 * P(-1, y - x - 1) + P(-1, y - x) + 2) >> 2;
          else
            pred[x][y] = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2;
        }
  } else if (mode == 5) {
    if (top8 && corner && left8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          isize z_vr = 2 * x - y;
          if (z_vr >= 0 && z_vr <= 14 && (z_vr % 2) == 0)
            pred[x][y] = (P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 1) >> 1;
          else if (z_vr >= 1 && z_vr <= 13 && (z_vr % 2) == 1)
            pred[x][y] =
                (P(x - (y >> 1) - 2, -1) + 2 * P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 2) >> 2;
          else if (z_vr == -1)
            pred[x][y] = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
          else
            pred[x][y] =
                (P(-1, y - 2 * x - 1) + 2 * P(-1, y - 2 * x - 2) + P(-1, y - 2 * x - 3) + 2) >> 2;
        }
  } else if (mode == 6) {
    if (top8 && corner && left8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          isize z_hd = 2 * y - x;
          if (z_hd >= 0 && z_hd <= 14 && (z_hd % 2) == 0)
            pred[x][y] = (P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 1) >> 1;
          else if (z_hd >= 1 && z_hd <= 13 && (z_hd % 2) == 1)
            pred[x][y] =
                (P(-1, y - (x >> 1) - 2) + 2 * P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 2) >> 2;
          else if (z_hd == -1)
            pred[x][y] = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
          else
            pred[x][y] =
                (P(x - 2 * y - 1, -1) + 2 * P(x - 2 * y - 2, -1) + P(x - 2 * y - 3, -1) + 2) >> 2;
        }
  } else if (mode == 7) {
    if (top8 && tr8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          if (y == 0 || y == 2 || y == 4 || y == 6)
            pred[x][y] = (P(x + (y >> 1), -1) + P(x + (y >> 1) + 1, -1) + 1) >> 1;
          else
            pred[x][y] =
                (P(x + (y >> 1), -1) + 2 * P(x + (y >> 1) + 1, -1) + P(x + (y >> 1) + 2, -1) + 2) >> 2;
        }
  } else if (mode == 8) {
    if (left8)
      for (isize y = 0; y < 8; y++)
        for (isize x = 0; x < 8; x++) {
          isize z_hu = x + 2 * y;
          if (z_hu <= 12 && (z_hu % 2) == 0)
            pred[x][y] = (P(-1, y + (x >> 1)) + P(-1, y + (x >> 1) + 1) + 1) >> 1;
          else if (z_hu <= 11 && (z_hu % 2) == 1)
            pred[x][y] =
                (P(-1, y + (x >> 1)) + 2 * P(-1, y + (x >> 1) + 1) + P(-1, y + (x >> 1) + 2) + 2) >> 2;
          else if (z_hu == 13)
            pred[x][y] = (P(-1, 6) + 3 * P(-1, 7) + 2) >> 2;
          else
            pred[x][y] = P(-1, 7);
        }
  } else {
    return -1; /* panic!("Could not do 8x8 prediction") :694 — unreachable with 3-bit rem modes */
  }
#undef P
#undef P1
  return 0;
}

/* pred8x8.rs:17-48 — 8.5.3 */
static int transform_for_8x8_luma_residual_blocks(Frame *f, Slice *s) {
  frame_scaling(f, s, 1, 0);
  for (isize blk8 = 0; blk8 < 4; blk8++) {
    isize c[8][8], r[8][8];
    inverse_scanner_8x8(s->mb.block_luma_8x8[blk8], c);
    scaling_and_transform8x8(f, s, c, r);
    if (intra8x8_prediction(f, s, blk8) != 0) return -1;
    isize u[64];
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++)
        u[i * 8 + j] = clamp(s->mb.luma8x8_pred_samples[blk8][j][i] + r[i][j], 0,
                             ((isize)1 << s->bit_depth_y) - 1);
    picture_construction(f, s, u, B8x8, blk8, 1, 0);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* frame/pred16x16.rs                                                                          */
/* ------------------------------------------------------------------------------------------- */

/* pred16x16.rs:428-482 — 8.5.10 */
static void transform_intra16x16_dc(const Frame *f, Slice *s, isize c[4][4], isize dc_y[4][4]) {
  isize q_p = s->mb.qp1y;
  static const isize A[4][4] = {{1, 1, 1, 1}, {1, 1, -1, -1}, {1, -1, -1, 1}, {1, -1, 1, -1}};
  isize g[4][4] = {{0}}, ff[4][4] = {{0}};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 4; k++) g[i][j] += A[i][k] * c[k][j];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 4; k++) ff[i][j] += g[i][k] * A[k][j];
  if (q_p >= 36) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
        dc_y[i][j] = (ff[i][j] * f->level_scale4x4[q_p % 6][0][0]) * ((isize)1 << (q_p / 6 - 6));
  } else {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++)
        dc_y[i][j] = (ff[i][j] * f->level_scale4x4[q_p % 6][0][0] + ((isize)1 << (5 - q_p / 6))) >>
                     (6 - q_p / 6);
  }
}

/* pred16x16.rs:79-425 — 8.3.3 */
static void intra16x16_prediction(Frame *f, Slice *s) {
  isize p[17 * 17];
  for (int i = 0; i < 17 * 17; i++) p[i] = -1;
  for (int i = 0; i < 33; i++) {
    /* REFERENCE_COORDINATE_X/Y :80-87 */
    isize x = i < 17 ? -1 : i - 17;
    isize y = i < 17 ? i - 1 : -1;
    const isize max_w = 16, max_h = 16;
    const MbRec *mb_n = mb_at(s, x, y, max_w, max_h);
    isize x_w, y_w;
    mbpos_coords(x, y, max_w, max_h, &x_w, &y_w);
    if (mb_n->unavailable) {
      P17(p, x, y) = -1;
    } else {
      isize mbaddr_n = mb_index(s, mb_n);
      isize x_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 0);
      isize y_m = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 1);
      P17(p, x, y) = LUMA(f, x_m + x_w, y_m + y_w);
    }
  }
#define P(x, y) P17(p, (x), (y))
  int top16 = 1, left16 = 1;
  for (int k = 0; k < 16; k++) {
    top16 = top16 && (P(k, -1) >= 0);
    left16 = left16 && (P(-1, k) >= 0);
  }
  isize(*pred)[16] = s->mb.luma16x16_pred_samples; /* [x][y] */
  int mode = s->mb.i16_pred_mode;
  if (mode == 0) {
    if (top16)
      for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) pred[x][y] = P(x, -1);
  } else if (mode == 1) {
    if (left16)
      for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) pred[x][y] = P(-1, y);
  } else if (mode == 2) {
    isize val;
    if (top16 && left16) {
      val = 16;
      for (int k = 0; k < 16; k++) val += P(k, -1) + P(-1, k);
      val >>= 5;
    } else if (!top16 && left16) {
      val = 8;
      for (int k = 0; k < 16; k++) val += P(-1, k);
      val >>= 4;
    } else if (top16 && !left16) {
      val = 8;
      for (int k = 0; k < 16; k++) val += P(k, -1);
      val >>= 4;
    } else {
      val = (isize)1 << (s->bit_depth_y - 1);
    }
    for (int x = 0; x < 16; x++)
      for (int y = 0; y < 16; y++) pred[x][y] = val;
  } else if (mode == 3 && top16 && left16) {
    /* QUIRK Q5: p[-1,-1] is read (x = 7 below) without an availability test (:366-404). */
    isize h = 0, v = 0;
    for (isize x = 0; x <= 7; x++) h += (x + 1) * (P(8 + x, -1) - P(6 - x, -1));
    for (isize y = 0; y <= 7; y++) v += (y + 1) * (P(-1, 8 + y) - P(-1, 6 - y));
    isize a = 16 * (P(-1, 15) + P(15, -1));
    isize b = (5 * h + 32) >> 6;
    isize c = (5 * v + 32) >> 6;
    for (isize y = 0; y < 16; y++)
      for (isize x = 0; x < 16; x++)
        pred[x][y] =
            clamp((a + b * (x - 7) + c * (y - 7) + 16) >> 5, 0, ((isize)1 << s->bit_depth_y) - 1);
  }
#undef P
}

/* pred16x16.rs:13-76 — 8.5.2 */
static void transform_for_16x16_luma_residual_blocks(Frame *f, Slice *s) {
  frame_scaling(f, s, 1, 0);
  isize c[4][4], dc_y[4][4];
  inverse_scanner4x4(s->mb.block_luma_dc, c);
  transform_intra16x16_dc(f, s, c, dc_y);

  isize r_mb[16][16]; /* [x][y] */
  memset(r_mb, 0, sizeof(r_mb));
  const isize dc_y_to_luma[16] = {dc_y[0][0], dc_y[0][1], dc_y[1][0], dc_y[1][1],
                                  dc_y[0][2], dc_y[0][3], dc_y[1][2], dc_y[1][3],
                                  dc_y[2][0], dc_y[2][1], dc_y[3][0], dc_y[3][1],
                                  dc_y[2][2], dc_y[2][3], dc_y[3][2], dc_y[3][3]};
  for (isize blk = 0; blk < 16; blk++) {
    isize luma_list[16];
    luma_list[0] = dc_y_to_luma[blk];
    for (int k = 0; k < 15; k++) luma_list[1 + k] = s->mb.block_luma_ac[blk][k];
    isize cc[4][4], r[4][4];
    inverse_scanner4x4(luma_list, cc);
    scaling_and_transform4x4(f, s, cc, 1, 0, r);
    isize x_o = inverse_raster_scan(blk / 4, 8, 8, 16, 0) + inverse_raster_scan(blk % 4, 4, 4, 8, 0);
    isize y_o = inverse_raster_scan(blk / 4, 8, 8, 16, 1) + inverse_raster_scan(blk % 4, 4, 4, 8, 1);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r_mb[x_o + j][y_o + i] = r[i][j];
  }

  intra16x16_prediction(f, s);

  isize u[256];
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++)
      u[i * 16 + j] = clamp(s->mb.luma16x16_pred_samples[j][i] + r_mb[j][i], 0,
                            ((isize)1 << s->bit_depth_y) - 1);
  picture_construction(f, s, u, B16x16, 0, 1, 0);
}

/* ------------------------------------------------------------------------------------------- */
/* frame/trans_chroma.rs                                                                       */
/* ------------------------------------------------------------------------------------------- */

/* trans_chroma.rs:369-456 — 8.5.11, ChromaArrayType 1 branch (:389-415) */
static void transform_chroma_dc(const Frame *f, Slice *s, isize c[2][2], int is_chroma_cb,
                                isize dc_c[4][2]) {
  memset(dc_c, 0, sizeof(isize) * 8);
  chroma_quantization_parameters(s, is_chroma_cb);
  isize q_p = s->mb.qp1c;
  static const isize a[2][2] = {{1, 1}, {1, -1}};
  isize g[2][2] = {{0}}, ff[2][2] = {{0}};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) g[i][j] += a[i][k] * c[k][j];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) ff[i][j] += g[i][k] * a[k][j];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      dc_c[i][j] = ((ff[i][j] * f->level_scale4x4[q_p % 6][0][0]) * ((isize)1 << (q_p / 6))) >> 5;
}

/* trans_chroma.rs:96-366 — 8.3.4 */
static void intra_chroma_prediction(Frame *f, Slice *s, int is_chroma_cb) {
  isize mb_width_c = s->mb_width_c, mb_height_c = s->mb_height_c;
  isize max_samples_val = mb_width_c + mb_height_c + 1;
  isize rx[17], ry[17];
  for (isize i = -1; i < mb_height_c; i++) {
    rx[i + 1] = -1;
    ry[i + 1] = i;
  }
  for (isize i = 0; i < mb_width_c; i++) {
    rx[mb_height_c + 1 + i] = i;
    ry[mb_height_c + 1 + i] = -1;
  }
  isize samples[81];
  for (int i = 0; i < 81; i++) samples[i] = -1;

  for (isize i = 0; i < max_samples_val; i++) {
    isize x = rx[i], y = ry[i];
    const MbRec *mb_n = mb_at(s, x, y, mb_width_c, mb_height_c);
    isize x_w, y_w;
    mbpos_coords(x, y, mb_width_c, mb_height_c, &x_w, &y_w);
    if (mb_n->unavailable) {
      P9(samples, x, y) = -1;
    } else {
      isize mbaddr_n = mb_index(s, mb_n);
      isize x_l = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 0);
      isize y_l = inverse_raster_scan(mbaddr_n, 16, 16, s->pic_width_in_samples_l, 1);
      isize x_m = (x_l >> 4) * mb_width_c;
      isize y_m = ((y_l >> 4) * mb_height_c) + (y_l % 2);
      isize p_x = x_m + x_w, p_y = y_m + y_w;
      P9(samples, x, y) = is_chroma_cb ? CB(f, p_x, p_y) : CR(f, p_x, p_y);
    }
  }
#define P(x, y) P9(samples, (x), (y))
  isize(*pred)[16] = s->mb.chroma_pred_samples; /* [x][y] */
  int mode = s->mb.intra_chroma_pred_mode;

  if (mode == 0) {
    for (isize blk = 0; blk < ((isize)1 << (s->chroma_array_type + 1)); blk++) {
      isize x_o = inverse_raster_scan(blk, 4, 4, 8, 0);
      isize y_o = inverse_raster_scan(blk, 4, 4, 8, 1);
      isize t0 = P(x_o, -1), t1 = P(1 + x_o, -1), t2 = P(2 + x_o, -1), t3 = P(3 + x_o, -1);
      isize l0 = P(-1, y_o), l1 = P(-1, 1 + y_o), l2 = P(-1, 2 + y_o), l3 = P(-1, 3 + y_o);
      isize val = 0;
      if ((x_o == 0 && y_o == 0) || (x_o > 0 && y_o > 0)) {
        if (t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 >= 0 && l0 >= 0 && l1 >= 0 && l2 >= 0 && l3 >= 0) {
          val = (t0 + t1 + t2 + t3 + l0 + l1 + l2 + l3 + 4) >> 3;
        } else if (!(t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 >= 0) &&
                   (l0 >= 0 && l1 >= 0 && l2 >= 0 && l3 >= 0)) {
          val = (l0 + l1 + l2 + l3 + 2) >> 2;
        } else if ((t0 > 0 && t1 > 0 && t2 > 0 && t3 > 0) &&
                   !(l0 > 0 && l1 > 0 && l2 > 0 && l3 > 0)) {
          /* QUIRK Q2: `> 0` where the spec means "available" (:209-216) */
          val = (t0 + t1 + t2 + t3 + 2) >> 2;
        } else {
          val = (isize)1 << (s->bit_depth_c - 1);
        }
      } else if (x_o > 0 && y_o == 0) {
        if (t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 >= 0)
          val = (t0 + t1 + t2 + t3 + 2) >> 2;
        else if (l0 >= 0 && l1 >= 0 && l2 >= 0 && l3 > 0) /* QUIRK Q2 :239-242 */
          val = (l0 + l1 + l2 + l3 + 2) >> 2;
        else
          val = (isize)1 << (s->bit_depth_c - 1);
      } else if (x_o == 0 && y_o > 0) {
        if (l0 >= 0 && l1 >= 0 && l2 >= 0 && l3 > 0) /* QUIRK Q2 :254-257 */
          val = (l0 + l1 + l2 + l3 + 2) >> 2;
        else if (t0 >= 0 && t1 >= 0 && t2 >= 0 && t3 > 0) /* QUIRK Q2 :265-268 */
          val = (t0 + t1 + t2 + t3 + 2) >> 2;
        else
          val = (isize)1 << (s->bit_depth_c - 1);
      }
      for (isize y = 0; y < 4; y++)
        for (isize x = 0; x < 4; x++) pred[x + x_o][y + y_o] = val;
    }
  } else if (mode == 1) { /* horizontal :287-303 */
    int flag = 1;
    for (isize y = 0; y < mb_height_c; y++)
      if (P(-1, y) < 0) {
        flag = 0;
        break;
      }
    if (flag)
      for (isize y = 0; y < mb_height_c; y++)
        for (isize x = 0; x < mb_width_c; x++) pred[x][y] = P(-1, y);
  } else if (mode == 2) { /* vertical :304-318 */
    int flag = 1;
    for (isize x = 0; x < mb_width_c; x++)
      if (P(x, -1) < 0) {
        flag = 0;
        break;
      }
    if (flag)
      for (isize y = 0; y < mb_height_c; y++)
        for (isize x = 0; x < mb_width_c; x++) pred[x][y] = P(x, -1);
  } else if (mode == 3) { /* plane :319-363 */
    int flag = 1;
    for (isize x = 0; x < mb_width_c; x++)
      if (P(x, -1) < 0) {
        flag = 0;
        break;
      }
    for (isize y = -1; y < mb_height_c; y++)
      if (P(-1, y) < 0) {
        flag = 0;
        break;
      }
    if (flag) {
      isize x_cf = s->chroma_array_type == 3 ? 4 : 0;
      isize y_cf = s->chroma_array_type != 1 ? 4 : 0;
      isize h = 0, v = 0;
      for (isize x1 = 0; x1 <= 3 + x_cf; x1++)
        h += (x1 + 1) * (P(4 + x_cf + x1, -1) - P(2 + x_cf - x1, -1));
      for (isize y1 = 0; y1 <= 3 + y_cf; y1++)
        v += (y1 + 1) * (P(-1, 4 + y_cf + y1) - P(-1, 2 + y_cf - y1));
      isize a = 16 * (P(-1, mb_height_c - 1) + P(mb_width_c - 1, -1));
      isize b = ((34 - 29 * (isize)(s->chroma_array_type == 3)) * h + 32) >> 6;
      isize c = ((34 - 29 * (isize)(s->chroma_array_type != 1)) * v + 32) >> 6;
      for (isize y = 0; y < mb_height_c; y++)
        for (isize x = 0; x < mb_width_c; x++)
          pred[x][y] = clamp((a + b * (x - 3 - x_cf) + c * (y - 3 - y_cf) + 16) >> 5, 0,
                             ((isize)1 << s->bit_depth_c) - 1);
    }
  }
#undef P
}

/* trans_chroma.rs:14-94 — 8.5.4, ChromaArrayType 1 */
static void transform_chroma_samples(Frame *f, Slice *s, int is_chroma_cb) {
  isize mb_width_c = s->mb_width_c, mb_height_c = s->mb_height_c;
  isize num_chroma4x4_blks = (mb_width_c / 4) * (mb_height_c / 4);
  int i_cb_cr = is_chroma_cb ? 0 : 1;

  isize dc_c[4][2];
  isize c[2][2];
  c[0][0] = s->mb.block_chroma_dc[i_cb_cr][0];
  c[0][1] = s->mb.block_chroma_dc[i_cb_cr][1];
  c[1][0] = s->mb.block_chroma_dc[i_cb_cr][2];
  c[1][1] = s->mb.block_chroma_dc[i_cb_cr][3];
  transform_chroma_dc(f, s, c, is_chroma_cb, dc_c);

  const isize dc_cto_chroma[8] = {dc_c[0][0], dc_c[0][1], dc_c[1][0], dc_c[1][1],
                                  dc_c[2][0], dc_c[2][1], dc_c[3][0], dc_c[3][1]};
  isize r_mb[8][16]; /* [x][y] */
  memset(r_mb, 0, sizeof(r_mb));
  for (isize blk = 0; blk < num_chroma4x4_blks; blk++) {
    isize chroma_list[16];
    chroma_list[0] = dc_cto_chroma[blk];
    for (int k = 0; k < 15; k++) chroma_list[1 + k] = s->mb.block_chroma_ac[i_cb_cr][blk][k];
    isize cc[4][4], r[4][4];
    inverse_scanner4x4(chroma_list, cc);
    scaling_and_transform4x4(f, s, cc, 0, is_chroma_cb, r);
    isize x_o = inverse_raster_scan(blk, 4, 4, 8, 0);
    isize y_o = inverse_raster_scan(blk, 4, 4, 8, 1);
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r_mb[x_o + j][y_o + i] = r[i][j];
  }

  intra_chroma_prediction(f, s, is_chroma_cb);

  isize u[64];
  for (isize i = 0; i < mb_width_c; i++)
    for (isize j = 0; j < mb_height_c; j++)
      u[i * mb_width_c + j] =
          clamp(s->mb.chroma_pred_samples[j][i] + r_mb[j][i], 0, ((isize)1 << s->bit_depth_c) - 1);
  picture_construction(f, s, u, B4x4, 0, 0, is_chroma_cb);
}

/* ------------------------------------------------------------------------------------------- */
/* frame/mod.rs:72-90 — Frame::decode                                                          */
/* ------------------------------------------------------------------------------------------- */

static int frame_decode(Frame *f, Slice *s) {
  int mode = slice_mb(s)->mode;
  if (mode == MODE_INTRA4X4) {
    transform_for_4x4_luma_residual_blocks(f, s);
  } else if (mode == MODE_INTRA8X8) {
    if (transform_for_8x8_luma_residual_blocks(f, s) != 0) return DRYV_E_UNSUPPORTED;
  } else if (mode == MODE_INTRA16X16) {
    transform_for_16x16_luma_residual_blocks(f, s);
  } else {
    return DRYV_E_UNSUPPORTED; /* todo!() :86,88 */
  }
  transform_chroma_samples(f, s, 1);
  transform_chroma_samples(f, s, 0);
  return DRYV_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* Harness: what the host shim does around the path (unpack records, drive the MB loop, and
 * frame/mod.rs:48-70 write_to_yuv_file's byte order)                                          */
/* ------------------------------------------------------------------------------------------- */

static int params_supported(const dryv_frame_params *fp) {
  if (!fp) return DRYV_E_INVALID;
  if (fp->pic_width_in_mbs == 0 || fp->pic_height_in_mbs == 0) return DRYV_E_INVALID;
  if (fp->chroma_array_type != 1 || fp->bit_depth_y != 8 || fp->bit_depth_c != 8)
    return DRYV_E_UNSUPPORTED;
  return DRYV_OK;
}

/* Fills slice.mb() from one FFI record + 384 coefficients (the inverse of the host batcher). */
static int load_macroblock(Slice *s, const dryv_mb_desc *d, const int16_t *co) {
  if (d->mb_kind > 2 || d->qp > 51 || d->i16_pred_mode > 3 || d->intra_chroma_pred_mode > 3)
    return DRYV_E_UNSUPPORTED;
  Macroblock *mb = &s->mb;
  memset(mb, 0, sizeof(*mb)); /* Macroblock::empty() */
  MbRec *rec = slice_mb(s);
  rec->unavailable = 0;
  rec->mode = d->mb_kind;
  memset(rec->intra4x4_pred_mode, 0, sizeof(rec->intra4x4_pred_mode));
  memset(rec->intra8x8_pred_mode, 0, sizeof(rec->intra8x8_pred_mode));
  mb->i16_pred_mode = d->i16_pred_mode;
  mb->intra_chroma_pred_mode = d->intra_chroma_pred_mode;
  mb->qpy = d->qp;
  mb->qp1y = d->qp; /* qp_bd_offset_y = 0 for 8-bit (cabac/mod.rs:191) */
  for (int i = 0; i < 16; i++) {
    int rem = (d->rem_modes[i >> 1] >> (4 * (i & 1))) & 7;
    int prev = (d->prev_flags >> i) & 1;
    mb->prev_intra4x4_pred_mode_flag[i] = (uint8_t)prev;
    mb->rem_intra4x4_pred_mode[i] = (uint8_t)rem;
    if (i < 4) {
      mb->prev_intra8x8_pred_mode_flag[i] = (uint8_t)prev;
      mb->rem_intra8x8_pred_mode[i] = (uint8_t)rem;
    }
  }
  const int16_t *p = co;
  if (d->mb_kind == 0) {
    for (int b = 0; b < 16; b++)
      for (int k = 0; k < 16; k++) mb->block_luma_4x4[b][k] = *p++;
  } else if (d->mb_kind == 1) {
    for (int b = 0; b < 4; b++)
      for (int k = 0; k < 64; k++) mb->block_luma_8x8[b][k] = *p++;
  } else {
    for (int k = 0; k < 16; k++) mb->block_luma_dc[k] = *p++;
    for (int b = 0; b < 16; b++)
      for (int k = 0; k < 15; k++) mb->block_luma_ac[b][k] = *p++;
  }
  for (int pl = 0; pl < 2; pl++) {
    for (int k = 0; k < 4; k++) mb->block_chroma_dc[pl][k] = *p++;
    for (int b = 0; b < 4; b++)
      for (int k = 0; k < 15; k++) mb->block_chroma_ac[pl][b][k] = *p++;
  }
  return DRYV_OK;
}

/* Reconstructs n_frames pictures. yuv_out: n_frames * 384*W*H bytes. modes_out (optional):
 * per macroblock 20 int8 = derived intra4x4_pred_mode[16] + intra8x8_pred_mode[4]. */
int dryv_oracle_reconstruct(const dryv_frame_params *fp, uint32_t n_frames, const dryv_mb_desc *mbs,
                            const int16_t *coeffs, uint8_t *yuv_out, int8_t *modes_out) {
  int st = params_supported(fp);
  if (st != DRYV_OK) return st;
  if (!mbs || !coeffs || !yuv_out) return DRYV_E_INVALID;

  Slice *s = (Slice *)calloc(1, sizeof(Slice));
  if (!s) return DRYV_E_NOMEM;
  s->fp = fp;
  s->pic_width_in_mbs = fp->pic_width_in_mbs;
  s->pic_height_in_mbs = fp->pic_height_in_mbs;
  s->pic_size_in_mbs = s->pic_width_in_mbs * s->pic_height_in_mbs;
  s->mb_width_c = 8;
  s->mb_height_c = 8;
  s->sub_width_c = 2;
  s->sub_height_c = 2;
  s->pic_width_in_samples_l = s->pic_width_in_mbs * 16;
  s->pic_height_in_samples_l = s->pic_height_in_mbs * 16;
  s->pic_width_in_samples_c = s->pic_width_in_mbs * s->mb_width_c;
  s->pic_height_in_samples_c = s->pic_height_in_mbs * s->mb_height_c;
  s->chroma_array_type = 1;
  s->bit_depth_y = 8;
  s->bit_depth_c = 8;
  s->qp_bd_offset_c = 0;
  for (int l = 0; l < 6; l++) {
    for (int k = 0; k < 16; k++) s->scaling_list4x4[l][k] = fp->scaling_list4x4[l][k];
    for (int k = 0; k < 64; k++) s->scaling_list8x8[l][k] = fp->scaling_list8x8[l][k];
  }
  s->unavailable.unavailable = 1;
  s->unavailable.mode = MODE_NA;
  s->macroblocks = (MbRec *)calloc((size_t)s->pic_size_in_mbs, sizeof(MbRec));

  Frame fr;
  memset(&fr, 0, sizeof(fr));
  fr.width_l = s->pic_width_in_samples_l;
  fr.height_l = s->pic_height_in_samples_l;
  fr.width_c = s->pic_width_in_samples_c;
  fr.height_c = s->pic_height_in_samples_c;
  size_t nl = (size_t)fr.width_l * (size_t)fr.height_l, nc = (size_t)fr.width_c * (size_t)fr.height_c;
  fr.luma_data = (uint8_t *)malloc(nl);
  fr.chroma_cb_data = (uint8_t *)malloc(nc);
  fr.chroma_cr_data = (uint8_t *)malloc(nc);
  if (!s->macroblocks || !fr.luma_data || !fr.chroma_cb_data || !fr.chroma_cr_data) {
    st = DRYV_E_NOMEM;
    goto done;
  }

  int worst = DRYV_OK;
  for (uint32_t fi = 0; fi < n_frames; fi++) {
    /* Frame::new + Slice::new: fresh planes and records per slice NAL (video/decoder.rs:123-124) */
    memset(fr.luma_data, 0, nl);
    memset(fr.chroma_cb_data, 0, nc);
    memset(fr.chroma_cr_data, 0, nc);
    memset(s->macroblocks, 0, (size_t)s->pic_size_in_mbs * sizeof(MbRec));
    size_t base = (size_t)fi * (size_t)s->pic_size_in_mbs;
    for (isize addr = 0; addr < s->pic_size_in_mbs; addr++) {
      s->curr_mb_addr = addr;
      int lst = load_macroblock(s, &mbs[base + addr], coeffs + (base + addr) * DRYV_COEFFS_PER_MB);
      if (lst == DRYV_OK) lst = frame_decode(&fr, s);
      if (lst != DRYV_OK) {
        worst = lst;
        /* the product leaves such a macroblock zero-filled and keeps going; so does the checker.
         * For its neighbours it then behaves like an Intra16x16 macroblock of zeros. */
        slice_mb(s)->mode = MODE_INTRA16X16;
      }
      if (modes_out) {
        int8_t *mo = modes_out + (base + addr) * 20;
        for (int k = 0; k < 16; k++) mo[k] = (int8_t)slice_mb(s)->intra4x4_pred_mode[k];
        for (int k = 0; k < 4; k++) mo[16 + k] = (int8_t)slice_mb(s)->intra8x8_pred_mode[k];
      }
    }
    /* write_to_yuv_file byte order (frame/mod.rs:48-70) */
    uint8_t *o = yuv_out + (size_t)fi * (nl + 2 * nc);
    for (isize y = 0; y < fr.height_l; y++)
      for (isize x = 0; x < fr.width_l; x++) *o++ = LUMA(&fr, x, y);
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) *o++ = CB(&fr, x, y);
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) *o++ = CR(&fr, x, y);
  }
  st = worst;

done:
  free(fr.luma_data);
  free(fr.chroma_cb_data);
  free(fr.chroma_cr_data);
  free(s->macroblocks);
  free(s);
  return st;
}

/* Decodes ONE macroblock at `mbaddr` of a picture whose earlier macroblocks are given: their
 * reconstructed samples are taken from yuv_inout (one frame, write_to_yuv_file layout), their
 * types from nb_kind[addr] (0/1/2) and their derived modes from nb_modes[addr*20 ..] (may be NULL:
 * zeros). The reconstructed macroblock is written back into yuv_inout. Used by the known-answer
 * tests to put exact neighbour samples in place (quirk vectors K7/K8). */
int dryv_oracle_decode_mb(const dryv_frame_params *fp, uint32_t mbaddr, const dryv_mb_desc *d,
                          const int16_t *co, uint8_t *yuv_inout, const uint8_t *nb_kind,
                          const int8_t *nb_modes, int8_t *modes_out) {
  int st = params_supported(fp);
  if (st != DRYV_OK) return st;
  if (!d || !co || !yuv_inout) return DRYV_E_INVALID;
  isize W = fp->pic_width_in_mbs, H = fp->pic_height_in_mbs;
  if ((isize)mbaddr >= W * H) return DRYV_E_INVALID;
  Slice *s = (Slice *)calloc(1, sizeof(Slice));
  if (!s) return DRYV_E_NOMEM;
  s->fp = fp;
  s->pic_width_in_mbs = W;
  s->pic_height_in_mbs = H;
  s->pic_size_in_mbs = W * H;
  s->mb_width_c = 8;
  s->mb_height_c = 8;
  s->sub_width_c = 2;
  s->sub_height_c = 2;
  s->pic_width_in_samples_l = W * 16;
  s->pic_height_in_samples_l = H * 16;
  s->pic_width_in_samples_c = W * 8;
  s->pic_height_in_samples_c = H * 8;
  s->chroma_array_type = 1;
  s->bit_depth_y = 8;
  s->bit_depth_c = 8;
  for (int l = 0; l < 6; l++) {
    for (int k = 0; k < 16; k++) s->scaling_list4x4[l][k] = fp->scaling_list4x4[l][k];
    for (int k = 0; k < 64; k++) s->scaling_list8x8[l][k] = fp->scaling_list8x8[l][k];
  }
  s->unavailable.unavailable = 1;
  s->unavailable.mode = MODE_NA;
  s->macroblocks = (MbRec *)calloc((size_t)s->pic_size_in_mbs, sizeof(MbRec));
  Frame fr;
  memset(&fr, 0, sizeof(fr));
  fr.width_l = W * 16;
  fr.height_l = H * 16;
  fr.width_c = W * 8;
  fr.height_c = H * 8;
  size_t nl = (size_t)fr.width_l * (size_t)fr.height_l, nc = (size_t)fr.width_c * (size_t)fr.height_c;
  fr.luma_data = (uint8_t *)malloc(nl);
  fr.chroma_cb_data = (uint8_t *)malloc(nc);
  fr.chroma_cr_data = (uint8_t *)malloc(nc);
  if (!s->macroblocks || !fr.luma_data || !fr.chroma_cb_data || !fr.chroma_cr_data) {
    st = DRYV_E_NOMEM;
    goto done;
  }
  {
    const uint8_t *in = yuv_inout;
    for (isize y = 0; y < fr.height_l; y++)
      for (isize x = 0; x < fr.width_l; x++) LUMA(&fr, x, y) = *in++;
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) CB(&fr, x, y) = *in++;
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) CR(&fr, x, y) = *in++;
  }
  for (isize a = 0; a < (isize)mbaddr; a++) {
    s->macroblocks[a].mode = nb_kind ? nb_kind[a] : MODE_INTRA16X16;
    if (nb_modes) {
      for (int k = 0; k < 16; k++) s->macroblocks[a].intra4x4_pred_mode[k] = nb_modes[a * 20 + k];
      for (int k = 0; k < 4; k++) s->macroblocks[a].intra8x8_pred_mode[k] = nb_modes[a * 20 + 16 + k];
    }
  }
  s->curr_mb_addr = mbaddr;
  st = load_macroblock(s, d, co);
  if (st == DRYV_OK) st = frame_decode(&fr, s);
  if (modes_out) {
    for (int k = 0; k < 16; k++) modes_out[k] = (int8_t)slice_mb(s)->intra4x4_pred_mode[k];
    for (int k = 0; k < 4; k++) modes_out[16 + k] = (int8_t)slice_mb(s)->intra8x8_pred_mode[k];
  }
  {
    uint8_t *o = yuv_inout;
    for (isize y = 0; y < fr.height_l; y++)
      for (isize x = 0; x < fr.width_l; x++) *o++ = LUMA(&fr, x, y);
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) *o++ = CB(&fr, x, y);
    for (isize y = 0; y < fr.height_c; y++)
      for (isize x = 0; x < fr.width_c; x++) *o++ = CR(&fr, x, y);
  }
done:
  free(fr.luma_data);
  free(fr.chroma_cb_data);
  free(fr.chroma_cr_data);
  free(s->macroblocks);
  free(s);
  return st;
}

/* Function-level entry points for the known-answer tests. c / r are row-major [i][j]. */
void dryv_oracle_residual4x4(const dryv_frame_params *fp, int qp, int is_luma, int is_chroma_cb,
                             int is_intra16x16, const int64_t c_in[16], int64_t r_out[16]) {
  Slice s;
  memset(&s, 0, sizeof(s));
  MbRec rec;
  memset(&rec, 0, sizeof(rec));
  Frame f;
  memset(&f, 0, sizeof(f));
  s.fp = fp;
  s.macroblocks = &rec;
  s.curr_mb_addr = 0;
  s.qp_bd_offset_c = 0;
  for (int l = 0; l < 6; l++) {
    for (int k = 0; k < 16; k++) s.scaling_list4x4[l][k] = fp->scaling_list4x4[l][k];
    for (int k = 0; k < 64; k++) s.scaling_list8x8[l][k] = fp->scaling_list8x8[l][k];
  }
  rec.mode = is_intra16x16 ? MODE_INTRA16X16 : MODE_INTRA4X4;
  s.mb.qpy = qp;
  s.mb.qp1y = qp;
  frame_scaling(&f, &s, 1, 0);
  isize c[4][4], r[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) c[i][j] = c_in[i * 4 + j];
  scaling_and_transform4x4(&f, &s, c, is_luma, is_chroma_cb, r);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r_out[i * 4 + j] = r[i][j];
}

void dryv_oracle_residual8x8(const dryv_frame_params *fp, int qp, const int64_t c_in[64],
                             int64_t r_out[64]) {
  Slice s;
  memset(&s, 0, sizeof(s));
  MbRec rec;
  memset(&rec, 0, sizeof(rec));
  Frame f;
  memset(&f, 0, sizeof(f));
  s.fp = fp;
  s.macroblocks = &rec;
  for (int l = 0; l < 6; l++) {
    for (int k = 0; k < 16; k++) s.scaling_list4x4[l][k] = fp->scaling_list4x4[l][k];
    for (int k = 0; k < 64; k++) s.scaling_list8x8[l][k] = fp->scaling_list8x8[l][k];
  }
  rec.mode = MODE_INTRA8X8;
  s.mb.qpy = qp;
  s.mb.qp1y = qp;
  frame_scaling(&f, &s, 1, 0);
  isize c[8][8], r[8][8];
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) c[i][j] = c_in[i * 8 + j];
  scaling_and_transform8x8(&f, &s, c, r);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) r_out[i * 8 + j] = r[i][j];
}

int64_t dryv_oracle_get_qpc(const dryv_frame_params *fp, int qpy, int is_chroma_cb) {
  Slice s;
  memset(&s, 0, sizeof(s));
  s.fp = fp;
  return get_qpc(&s, qpy, is_chroma_cb);
}

int64_t dryv_oracle_clamp(int64_t v, int64_t lo, int64_t hi) { return clamp(v, lo, hi); }
int64_t dryv_oracle_inverse_raster_scan(int64_t a, int64_t b, int64_t c, int64_t d, int64_t e) {
  return inverse_raster_scan(a, b, c, d, e);
}
