// deblock_params.h — host side of the deblocking kernel's constant block (plain C++: also used by tests/emu).
#pragma once
#include <string.h>

#include "deblock_kernel_params.h"

namespace dryv {
namespace deblock {

// DRYV_OK and *P filled; DRYV_E_UNSUPPORTED outside the library's picture formats; DRYV_E_INVALID for syntax elements out of
// range. *skip = 1 when disable_deblocking_filter_idc says that nothing is filtered.
inline int build_dparams(const dryv_frame_params* fp, const dryv_deblock_params* dp, uint32_t n_frames, DParams* P, int* skip) {
  if (!fp || !dp || !P || !skip) return DRYV_E_INVALID;
  if (fp->pic_width_in_mbs == 0 || fp->pic_height_in_mbs == 0 || fp->pic_width_in_mbs > 1024 || n_frames == 0) return DRYV_E_INVALID;
  if ((unsigned long long)fp->pic_width_in_mbs * fp->pic_height_in_mbs * 384ull * n_frames > 0xFFFFFFFFFFFFull) return DRYV_E_INVALID;
  if ((unsigned long long)fp->pic_width_in_mbs * fp->pic_height_in_mbs * 384ull > 0xFFFFFFFFull) return DRYV_E_INVALID;
  if (fp->chroma_array_type != 1 || fp->bit_depth_y != 8 || fp->bit_depth_c != 8) return DRYV_E_UNSUPPORTED;
  if (dp->disable_deblocking_filter_idc > 2 || dp->slice_alpha_c0_offset_div2 < -6 || dp->slice_alpha_c0_offset_div2 > 6 ||
      dp->slice_beta_offset_div2 < -6 || dp->slice_beta_offset_div2 > 6)
    return DRYV_E_INVALID;
  *skip = dp->disable_deblocking_filter_idc == 1;
  memset(P, 0, sizeof(*P));
  P->W = fp->pic_width_in_mbs;
  P->H = fp->pic_height_in_mbs;
  P->n_frames = (int)n_frames;
  P->offA = 2 * dp->slice_alpha_c0_offset_div2;
  P->offB = 2 * dp->slice_beta_offset_div2;
  P->cqo_cb = fp->chroma_qp_index_offset;
  P->cqo_cr = fp->second_chroma_qp_index_offset;
  // ITU-T H.264 tables 8-16 (alpha', beta') and 8-17 (tC0', bS = 3)
  static const uint8_t ALPHA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22,
                                    25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
  static const uint8_t BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7,
                                   8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
  static const uint8_t TC0_BS3[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                      1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25};
  memcpy(P->alpha, ALPHA, 52);
  memcpy(P->beta, BETA, 52);
  memcpy(P->tc0, TC0_BS3, 52);
  return DRYV_OK;
}

// Workspace: [luma task counter | chroma task counter (128 bytes apart) | pad to 256][luma progress words][chroma progress
// words | pad to 256][luma side buffer][chroma side buffer]
inline size_t prog_words(const DParams& P) { return (size_t)P.n_frames * ((size_t)(P.H + 3) / 4); }
inline size_t reset_bytes(const DParams& P) {  // leading part a launch needs zeroed: task counters, progress words
  return 256 + ((2 * prog_words(P) * 4 + 255) & ~(size_t)255);
}
inline size_t workspace_bytes(const DParams& P) { return reset_bytes(P) + prog_words(P) * P.W * (64 + 32); }
template <class ARGS>
inline void place_workspace(const DParams& P, unsigned char* ws, ARGS* A) {
  A->taskCounter[0] = (unsigned*)ws;
  A->taskCounter[1] = (unsigned*)(ws + 128);
  A->prog[0] = (unsigned*)(ws + 256);
  A->prog[1] = A->prog[0] + prog_words(P);
  A->side[0] = ws + reset_bytes(P);
  A->side[1] = A->side[0] + prog_words(P) * P.W * 64;
}

}  // namespace deblock
}  // namespace dryv
