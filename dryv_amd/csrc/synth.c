/*
 * synth.c — deterministic synthetic all-intra macroblock batches at the FFI boundary.
 *
 * Produces what dryv's CABAC layer would hand to Frame::decode: one dryv_mb_desc
 * and 384 zig-zag-ordered int16 coefficients per macroblock, with prediction modes that are legal
 * for each block's neighbour availability and encoded as prev_intra*_pred_mode_flag /
 * rem_intra*_pred_mode by inverting the derivation of 8.3.1.1 / 8.3.2.1
 * (reference: src/video/frame/pred4x4.rs:363-427, pred8x8.rs:698-764).
 *
 * Host-only C (no GPU code); built as libdryv_synth.so. Used by tests and bench.py for inputs.
 * Not part of the oracle and does not call it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dryv_recon.h"

typedef struct dryv_synth_config {
  uint32_t permille_i4x4;       /* share of Intra4x4 macroblocks (of 1000)                      */
  uint32_t permille_i8x8;       /* share of Intra8x8; the rest is Intra16x16                    */
  uint32_t qp_min, qp_max;      /* qp uniform in [qp_min, qp_max]                               */
  uint32_t permille_coded;      /* P(block coded)                                               */
  uint32_t p0_q16;              /* P(nonzero at scan 0), Q16                                    */
  uint32_t decay4_q16;          /* per-position decay for 4x4 / DC / AC blocks, Q16             */
  uint32_t decay8_q16;          /* per-position decay for 8x8 blocks, Q16                       */
  uint32_t max_level;           /* |level| = 1 + Geom(0.5), clipped to this (<= 32767)          */
  uint32_t legal_modes_only;    /* 1: only modes whose reference samples exist; 0: any mode     */
  uint32_t permille_prev_flag;  /* when the chosen mode equals the predicted one, always prev=1;
                                   with legal_modes_only=0 this is P(prev_flag=1)               */
} dryv_synth_config;

static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

typedef struct {
  uint64_t s;
  uint64_t bits;
  int nbits;
} Rng;

static uint32_t rng_u32(Rng *r) { return (uint32_t)(splitmix64(&r->s) >> 32); }
static uint32_t rng_below(Rng *r, uint32_t n) { return (uint32_t)(((uint64_t)rng_u32(r) * n) >> 32); }
static int rng_bit(Rng *r) {
  if (r->nbits == 0) {
    r->bits = splitmix64(&r->s);
    r->nbits = 64;
  }
  int b = (int)(r->bits & 1);
  r->bits >>= 1;
  r->nbits--;
  return b;
}
static int rng_q16(Rng *r, uint32_t p_q16) { return (rng_u32(r) >> 16) < p_q16; }

/* fill a zig-zag list of n positions */
static void gen_list(Rng *r, const dryv_synth_config *c, int16_t *out, int n, uint32_t decay_q16) {
  memset(out, 0, sizeof(int16_t) * (size_t)n);
  if (rng_below(r, 1000) >= c->permille_coded) return;
  uint32_t p = c->p0_q16;
  for (int k = 0; k < n; k++) {
    if (p == 0) break;
    if (rng_q16(r, p)) {
      uint32_t lvl = 1;
      while (lvl < c->max_level && rng_bit(r)) lvl++;
      out[k] = (int16_t)(rng_bit(r) ? -(int32_t)lvl : (int32_t)lvl);
    }
    p = (uint32_t)(((uint64_t)p * decay_q16) >> 16);
  }
}

/* derived modes kept per macroblock for the neighbour rule */
typedef struct {
  uint8_t kind;
  uint8_t m4[16];
  uint8_t m8[4];
} ModeRec;

static const uint8_t BLK_X[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3};
static const uint8_t BLK_Y[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};
static int blk_of(int bx, int by) { return 8 * (by / 2) + 4 * (bx / 2) + 2 * (by % 2) + (bx % 2); }

/* mode of the 4x4 position (bx,by) of a neighbour macroblock as seen by an Intra4x4 block */
static int nb_mode_for4(const ModeRec *m, int bx, int by) {
  if (m->kind == 0) return m->m4[blk_of(bx, by)];
  if (m->kind == 1) return m->m8[blk_of(bx, by) >> 2];
  return 2;
}
/* ... as seen by an Intra8x8 block: n = 1 for A, 2 for B (pred8x8.rs:738-750) */
static int nb_mode_for8(const ModeRec *m, int b8, int n) {
  if (m->kind == 1) return m->m8[b8];
  if (m->kind == 0) return m->m4[b8 * 4 + n];
  return 2;
}

/* legal luma modes given which reference samples exist */
static int pick_luma_mode(Rng *r, int legal_only, int top, int left, int corner) {
  if (!legal_only) return (int)rng_below(r, 9);
  int cand[9], n = 0;
  cand[n++] = 2;
  if (top) {
    cand[n++] = 0;
    cand[n++] = 3;
    cand[n++] = 7;
  }
  if (left) {
    cand[n++] = 1;
    cand[n++] = 8;
  }
  if (top && left && corner) {
    cand[n++] = 4;
    cand[n++] = 5;
    cand[n++] = 6;
  }
  return cand[rng_below(r, (uint32_t)n)];
}

static void encode_mode(Rng *r, const dryv_synth_config *c, dryv_mb_desc *d, int idx, int pred,
                        int *mode_io) {
  int prev, rem;
  if (c->legal_modes_only) {
    int m = *mode_io;
    if (m == pred) {
      prev = 1;
      rem = 0;
    } else {
      prev = 0;
      rem = m < pred ? m : m - 1;
    }
  } else {
    /* raw syntax: any flag / rem combination; report the mode it derives to */
    prev = rng_below(r, 1000) < c->permille_prev_flag;
    rem = (int)rng_below(r, 8);
    *mode_io = prev ? pred : (rem < pred ? rem : rem + 1);
  }
  if (prev) d->prev_flags |= (uint16_t)(1u << idx);
  d->rem_modes[idx >> 1] |= (uint8_t)(rem << (4 * (idx & 1)));
}

int dryv_synth_generate(const dryv_frame_params *fp, const dryv_synth_config *cfg, uint64_t config_id,
                        uint32_t first_frame, uint32_t n_frames, dryv_mb_desc *mbs, int16_t *coeffs) {
  if (!fp || !cfg || !mbs || !coeffs) return DRYV_E_INVALID;
  int W = fp->pic_width_in_mbs, H = fp->pic_height_in_mbs;
  if (W <= 0 || H <= 0 || cfg->qp_max > 51 || cfg->qp_min > cfg->qp_max || cfg->max_level == 0 ||
      cfg->max_level > 32767)
    return DRYV_E_INVALID;
  ModeRec *rec = (ModeRec *)malloc(sizeof(ModeRec) * (size_t)W * (size_t)H);
  if (!rec) return DRYV_E_NOMEM;

  for (uint32_t fi = 0; fi < n_frames; fi++) {
    Rng r;
    r.s = 0x6472797600000000ull ^ (config_id << 24) ^ (uint64_t)(first_frame + fi);
    r.bits = 0;
    r.nbits = 0;
    size_t base = (size_t)fi * (size_t)W * (size_t)H;
    for (int my = 0; my < H; my++)
      for (int mx = 0; mx < W; mx++) {
        size_t addr = base + (size_t)my * W + mx;
        dryv_mb_desc *d = &mbs[addr];
        int16_t *co = coeffs + addr * DRYV_COEFFS_PER_MB;
        ModeRec *me = &rec[my * W + mx];
        const ModeRec *A = mx > 0 ? &rec[my * W + mx - 1] : NULL;
        const ModeRec *B = my > 0 ? &rec[(my - 1) * W + mx] : NULL;
        memset(d, 0, sizeof(*d));
        memset(me, 0, sizeof(*me));
        d->nz_mask = 0xFFFF;
        d->qp = (uint8_t)(cfg->qp_min + rng_below(&r, cfg->qp_max - cfg->qp_min + 1));

        uint32_t t = rng_below(&r, 1000);
        int kind = t < cfg->permille_i4x4 ? 0 : (t < cfg->permille_i4x4 + cfg->permille_i8x8 ? 1 : 2);
        d->mb_kind = (uint8_t)kind;
        me->kind = (uint8_t)kind;
        int mb_top = my > 0, mb_left = mx > 0;

        if (kind == 0) {
          for (int b = 0; b < 16; b++) {
            int bx = BLK_X[b], by = BLK_Y[b];
            int top = by > 0 || mb_top, left = bx > 0 || mb_left;
            int corner = (bx > 0 || mb_left) && (by > 0 || mb_top);
            /* predicted mode (pred4x4.rs:386-414) */
            int a_av = bx > 0 || mb_left, b_av = by > 0 || mb_top;
            int pred;
            if (!a_av || !b_av) {
              pred = 2;
            } else {
              int ma = bx > 0 ? me->m4[blk_of(bx - 1, by)] : nb_mode_for4(A, 3, by);
              int mb = by > 0 ? me->m4[blk_of(bx, by - 1)] : nb_mode_for4(B, bx, 3);
              pred = ma < mb ? ma : mb;
            }
            int m = pick_luma_mode(&r, (int)cfg->legal_modes_only, top, left, corner);
            encode_mode(&r, cfg, d, b, pred, &m);
            me->m4[b] = (uint8_t)m;
          }
          for (int b = 0; b < 16; b++) gen_list(&r, cfg, co + b * 16, 16, cfg->decay4_q16);
        } else if (kind == 1) {
          for (int b8 = 0; b8 < 4; b8++) {
            int bx = b8 & 1, by = b8 >> 1;
            int top = by > 0 || mb_top, left = bx > 0 || mb_left;
            int corner = (bx > 0 || mb_left) && (by > 0 || mb_top);
            int a_av = bx > 0 || mb_left, b_av = by > 0 || mb_top;
            int pred;
            if (!a_av || !b_av) {
              pred = 2;
            } else {
              int ma = bx > 0 ? me->m8[b8 - 1] : nb_mode_for8(A, b8 + 1, 1);
              int mb = by > 0 ? me->m8[b8 - 2] : nb_mode_for8(B, b8 + 2, 2);
              pred = ma < mb ? ma : mb;
            }
            int m = pick_luma_mode(&r, (int)cfg->legal_modes_only, top, left, corner);
            encode_mode(&r, cfg, d, b8, pred, &m);
            me->m8[b8] = (uint8_t)m;
          }
          for (int b8 = 0; b8 < 4; b8++) gen_list(&r, cfg, co + b8 * 64, 64, cfg->decay8_q16);
        } else {
          int m;
          if (!cfg->legal_modes_only) {
            m = (int)rng_below(&r, 4);
          } else {
            int cand[4], n = 0;
            cand[n++] = 2;
            if (mb_top) cand[n++] = 0;
            if (mb_left) cand[n++] = 1;
            if (mb_top && mb_left) cand[n++] = 3;
            m = cand[rng_below(&r, (uint32_t)n)];
          }
          d->i16_pred_mode = (uint8_t)m;
          gen_list(&r, cfg, co, 16, cfg->decay4_q16);
          for (int b = 0; b < 16; b++) gen_list(&r, cfg, co + 16 + b * 15, 15, cfg->decay4_q16);
        }
        {
          int m;
          if (!cfg->legal_modes_only) {
            m = (int)rng_below(&r, 4);
          } else {
            int cand[4], n = 0;
            cand[n++] = 0;
            if (mb_left) cand[n++] = 1;
            if (mb_top) cand[n++] = 2;
            if (mb_top && mb_left) cand[n++] = 3;
            m = cand[rng_below(&r, (uint32_t)n)];
          }
          d->intra_chroma_pred_mode = (uint8_t)m;
        }
        for (int pl = 0; pl < 2; pl++) {
          int16_t *cc = co + 256 + pl * 64;
          gen_list(&r, cfg, cc, 4, cfg->decay4_q16);
          for (int b = 0; b < 4; b++) gen_list(&r, cfg, cc + 4 + b * 15, 15, cfg->decay4_q16);
        }
      }
  }
  free(rec);
  return DRYV_OK;
}
