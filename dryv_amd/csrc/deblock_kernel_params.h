// deblock_kernel_params.h — the constant block a deblocking launch passes to the kernel (plain C++).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"

namespace dryv {
namespace deblock {

struct DParams {
  int W, H, n_frames;
  int offA, offB;        // filterOffsetA / B = slice_alpha_c0_offset_div2 * 2, slice_beta_offset_div2 * 2
  int cqo_cb, cqo_cr;    // chroma_qp_index_offset, second_chroma_qp_index_offset
  uint8_t alpha[52], beta[52], tc0[52];  // tables 8-16, 8-17 (bS = 3 column)
};

}  // namespace deblock
}  // namespace dryv
