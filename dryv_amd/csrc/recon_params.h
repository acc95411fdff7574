// recon_params.h — host-side derivation of the per-submit constant tables (LevelScale, prediction gather tables,
// exactness thresholds) from a dryv_frame_params block. Plain C++ (no HIP): shared by the C-ABI library
// (recon_api.hip) and the CPU emulation harness of the band kernel (tests/emu).
#pragma once
#include <algorithm>
#include <stdint.h>
#include <string.h>

#include "../../include/dryv_recon.h"
#include "kparams.h"

namespace dryv {
namespace params {

// frame/mod.rs:212-284 — (row, col) of 8x8 zig-zag list position k
static const uint8_t ZZ8[64][2] = {
    {0, 0}, {0, 1}, {1, 0}, {2, 0}, {1, 1}, {0, 2}, {0, 3}, {1, 2}, {2, 1}, {3, 0}, {4, 0},
    {3, 1}, {2, 2}, {1, 3}, {0, 4}, {0, 5}, {1, 4}, {2, 3}, {3, 2}, {4, 1}, {5, 0}, {6, 0},
    {5, 1}, {4, 2}, {3, 3}, {2, 4}, {1, 5}, {0, 6}, {0, 7}, {1, 6}, {2, 5}, {3, 4}, {4, 3},
    {5, 2}, {6, 1}, {7, 0}, {7, 1}, {6, 2}, {5, 3}, {4, 4}, {3, 5}, {2, 6}, {1, 7}, {2, 7},
    {3, 6}, {4, 5}, {5, 4}, {6, 3}, {7, 2}, {7, 3}, {6, 4}, {5, 5}, {4, 6}, {3, 7}, {4, 7},
    {5, 6}, {6, 5}, {7, 4}, {7, 5}, {6, 6}, {5, 7}, {6, 7}, {7, 6}, {7, 7}};
// frame/mod.rs:185-209 — (row, col) of 4x4 zig-zag list position k
static const uint8_t ZZ4[16][2] = {{0, 0}, {0, 1}, {1, 0}, {2, 0}, {1, 1}, {0, 2}, {0, 3}, {1, 2},
                            {2, 1}, {3, 0}, {3, 1}, {2, 2}, {1, 3}, {2, 3}, {3, 2}, {3, 3}};

enum { SEL_E = 0, SEL_F = 1, SEL_G = 2 };
inline uint8_t ent(int sel, int idx) { return (uint8_t)(idx | (sel << 5)); }

// Prediction as a gather. The block's reference samples are laid out on one line
//   E = [ left column bottom..top | corner | top row (+ top-right) ]
// and F[i] = (E[i-1] + 2E[i] + E[i+1] + 2) >> 2, G[i] = (E[i] + E[i+1] + 1) >> 1 (ends replicated).
// Every directional mode of 8.3.1.2 / 8.3.2.2 then reads exactly one of E/F/G per pixel
// (pred4x4.rs:92-359, pred8x8.rs:294-692). n = 4: E[0..3]=L3..L0, E[4]=corner, E[5..12]=T0..T7;
// n = 8: E[0..7]=L7..L0, E[8]=corner, E[9..24]=T0..T15.
inline void build_pred_table(int n, uint8_t* t) {
  const int C = n;          // index of the corner sample
  const int T0 = n + 1;     // index of p[0,-1]
  const int L0 = n - 1;     // index of p[-1,0]
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) {
      const int p = y * n + x;
      t[0 * n * n + p] = ent(SEL_E, T0 + x);
      t[1 * n * n + p] = ent(SEL_E, L0 - y);
      t[2 * n * n + p] = 0;
      t[3 * n * n + p] = ent(SEL_F, T0 + 1 + x + y);
      t[4 * n * n + p] = ent(SEL_F, C + x - y);
      {  // vertical-right
        const int z = 2 * x - y, k = x - (y >> 1);
        uint8_t e;
        if (z >= 0) e = (z & 1) ? ent(SEL_F, C + k) : ent(SEL_G, C + k);
        else if (z == -1) e = ent(SEL_F, C);
        else e = ent(SEL_F, T0 - y + 2 * x);
        t[5 * n * n + p] = e;
      }
      {  // horizontal-down
        const int z = 2 * y - x, k = y - (x >> 1);
        uint8_t e;
        if (z >= 0) e = (z & 1) ? ent(SEL_F, C - k) : ent(SEL_G, L0 - k);
        else if (z == -1) e = ent(SEL_F, C);
        else e = ent(SEL_F, L0 + x - 2 * y);
        t[6 * n * n + p] = e;
      }
      t[7 * n * n + p] = (y & 1) ? ent(SEL_F, T0 + 1 + x + (y >> 1)) : ent(SEL_G, T0 + x + (y >> 1));
      {  // horizontal-up
        const int z = x + 2 * y, k = y + (x >> 1);
        const int zmax = 2 * n - 3;  // 5 for 4x4, 13 for 8x8
        uint8_t e;
        if (z < zmax) e = (z & 1) ? ent(SEL_F, n - 2 - k) : ent(SEL_G, n - 2 - k);
        else if (z == zmax) e = ent(SEL_F, 0);
        else e = ent(SEL_E, 0);
        t[8 * n * n + p] = e;
      }
    }
}

inline int build_params(const dryv_frame_params* fp, uint32_t n_frames, KParams* P) {
  if (!fp) return DRYV_E_INVALID;
  if (fp->pic_width_in_mbs == 0 || fp->pic_height_in_mbs == 0 || fp->pic_width_in_mbs > 1024) return DRYV_E_INVALID;
  // the kernels address a frame's planes, records and coefficients with 32-bit byte offsets (768 B of coefficients per
  // macroblock is the largest) and count the batch's macroblocks in 31 bits
  const unsigned long long per = (unsigned long long)fp->pic_width_in_mbs * fp->pic_height_in_mbs;
  if (per * 768ull > 0xFFFFFFFFull) return DRYV_E_INVALID;
  if (n_frames == 0 || per * (unsigned long long)n_frames > 0x7FFFFFFFull) return DRYV_E_INVALID;
  if (fp->chroma_array_type != 1 || fp->bit_depth_y != 8 || fp->bit_depth_c != 8) return DRYV_E_UNSUPPORTED;
  memset(P, 0, sizeof(*P));
  P->W = fp->pic_width_in_mbs;
  P->H = fp->pic_height_in_mbs;
  P->n_frames = (int)n_frames;
  P->cqo_cb = fp->chroma_qp_index_offset;
  P->cqo_cr = fp->second_chroma_qp_index_offset;
  // 8.5.9 (transform.rs:8-78). Only scaling list 0 of each size is ever read on this path
  // (intra, luma call; chroma re-uses the tables: quirk Q3).
  static const int V4[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
  static const int V8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                               {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
  int w4[4][4], w8[8][8];
  for (int k = 0; k < 16; k++) w4[ZZ4[k][0]][ZZ4[k][1]] = fp->scaling_list4x4[0][k];
  for (int k = 0; k < 64; k++) w8[ZZ8[k][0]][ZZ8[k][1]] = fp->scaling_list8x8[0][k];
  for (int m = 0; m < 6; m++) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        const int cls = (i % 2 == 0 && j % 2 == 0) ? 0 : ((i % 2 == 1 && j % 2 == 1) ? 1 : 2);
        P->ls4[m * 16 + i * 4 + j] = (uint16_t)(w4[i][j] * V4[m][cls]);
      }
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) {
        int cls;
        if (i % 4 == 0 && j % 4 == 0) cls = 0;
        else if (i % 2 == 1 && j % 2 == 1) cls = 1;
        else if (i % 4 == 2 && j % 4 == 2) cls = 2;
        else if ((i % 4 == 0 && j % 2 == 1) || (i % 2 == 1 && j % 4 == 0)) cls = 3;
        else if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) cls = 4;
        else cls = 5;
        P->ls8[m * 64 + i * 8 + j] = (uint16_t)(w8[i][j] * V8[m][cls]);
      }
  }
  build_pred_table(4, P->t4);
  build_pred_table(8, P->t8);
  for (int k = 0; k < 64; k++) P->zz8i[ZZ8[k][0] * 8 + ZZ8[k][1]] = (uint8_t)k;
  P->transform8x8 = fp->transform_8x8_mode_flag ? 1 : 0;
  // band kernel: LevelScale4x4 in list order; int32 exactness thresholds.
  // 4x4 (transform.rs:147-187): |d| <= 2^26 keeps both butterfly passes (gain <= 3.5 each) and the +32 below 2^31;
  // the product c*LS itself needs |c| * maxLS < 2^31 (always true: 2^15 * 7395).
  // 8x8 (pred8x8.rs:73-147): a butterfly pass has gain < 12.25 -> |d| <= 2^23.
  for (int m = 0; m < 6; m++)
    for (int k = 0; k < 16; k++) P->ls4z[m * 16 + k] = P->ls4[m * 16 + ZZ4[k][0] * 4 + ZZ4[k][1]];
  for (int qp = 0; qp < 52; qp++) {
    const int qd = qp / 6, qm = qp % 6;
    long long m4 = 1, m8 = 1;
    for (int k = 0; k < 16; k++) if (P->ls4[qm * 16 + k] > m4) m4 = P->ls4[qm * 16 + k];
    for (int k = 0; k < 64; k++) if (P->ls8[qm * 64 + k] > m8) m8 = P->ls8[qm * 64 + k];
    const long long s4 = m4 << (qd > 4 ? qd - 4 : 0), s8 = m8 << (qd > 6 ? qd - 6 : 0);
    const long long t4 = (1ll << 26) / s4, t8 = (1ll << 23) / s8;
    P->thr4[qp] = (uint16_t)(t4 >= 32768 ? 0xFFFF : t4);
    P->thr8[qp] = (uint16_t)(t8 >= 32768 ? 0xFFFF : t8);
  }
  return DRYV_OK;
}

}  // namespace params
}  // namespace dryv
