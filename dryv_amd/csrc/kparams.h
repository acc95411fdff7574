// kparams.h — the constant block a launch passes to the kernels (plain C++, no HIP: also used by tests/emu).
#pragma once
#include <stdint.h>

namespace dryv {

// Everything the kernel needs that is constant for a submit; passed by value in the kernarg
// segment and copied to LDS once per workgroup.
struct KParams {
  int W, H;         // picture size in macroblocks
  int n_frames;
  int cqo_cb;       // pps.chroma_qp_index_offset
  int cqo_cr;       // second_chroma_qp_index_offset
  uint16_t ls4[96];   // LevelScale4x4[m][i*4+j], scaling list 0 (transform.rs:22-45, quirk Q3)
  uint16_t ls8[384];  // LevelScale8x8[m][i*8+j], scaling list 0 (transform.rs:47-77)
  uint8_t t4[144];    // Intra4x4 gather table [mode][y*4+x]:  idx | sel << 5 (sel 0 E, 1 F, 2 G)
  uint8_t t8[576];    // Intra8x8 gather table [mode][y*8+x]
  uint8_t zz8i[64];   // raster position i*8+j -> index in the 8x8 zig-zag list (frame/mod.rs:212-284)
  // band kernel
  uint16_t ls4z[96];  // LevelScale4x4[m] in zig-zag LIST order: ls4z[m*16+k] = ls4[m*16 + row(k)*4 + col(k)]
  // Largest |coefficient| for which every intermediate of the int32 transform provably equals the reference's 64-bit
  // result, per qp (depends on the scaling list); 0xFFFF = any int16 coefficient is fine. Blocks beyond it take the
  // 64-bit path. thr4: 4x4 blocks (8.5.12), thr8: 8x8 blocks (8.5.13).
  uint16_t thr4[52];
  uint16_t thr8[52];
  int transform8x8;   // frame parameter transform_8x8_mode_flag: whether mb_kind 1 may occur
};

}  // namespace dryv
