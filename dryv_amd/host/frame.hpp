// frame.hpp — host-side C++ mirror of the reference's reconstruction interface, over the C ABI.
//
// The reference exposes reconstruction as three inherent methods on `Frame`
// (/root/reference/src/video/frame/mod.rs:16-90):
//     Frame::new(&slice)              -> dryv::Frame(frame_params, ctx)
//     frame.decode(&mut slice)        -> frame.decode(mb)                 once per macroblock, in mbaddr order
//     frame.write_to_yuv_file(path)   -> frame.write_to_yuv_file(path)    same byte order (frame/mod.rs:48-70)
// `decode` is called from inside the CABAC macroblock loop (cabac/mod.rs:208); nothing the parser does
// later depends on reconstructed samples, so here it only narrows the macroblock's fields into the
// 16-byte record + 384 int16 coefficients of include/dryv_recon.h and appends them to the frame's batch.
// The first call that needs pixels submits the whole frame to the GPU. All arithmetic happens in
// libdryv_recon.so; this header moves bytes and mirrors names and error behaviour (status codes instead
// of todo!()/panic!()).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dryv_recon.h"

namespace dryv {

// The fields of the reference's `Macroblock` (slice/macroblock.rs:21-129) that Frame::decode reads, with the
// reference's names; coefficient lists are the reference's `isize` arrays (zig-zag order).
struct Macroblock {
  enum Mode { Intra4x4 = 0, Intra8x8 = 1, Intra16x16 = 2, Pcm = 25, Inter = 26 };
  int mode = Intra16x16;             // mb_type.mode(): frame/mod.rs:73-84
  int intra16x16_pred_mode = 2;      // mb_type.intra16x16_pred_mode(): macroblock.rs:584-591
  int intra_chroma_pred_mode = 0;    // :88
  long qpy = 26;                     // :40 (== qp1y for 8-bit video, cabac/mod.rs:186-191)
  uint8_t prev_intra4x4_pred_mode_flag[16] = {0}, rem_intra4x4_pred_mode[16] = {0};  // :76-79
  uint8_t prev_intra8x8_pred_mode_flag[4] = {0}, rem_intra8x8_pred_mode[4] = {0};    // :82-85
  long block_luma_4x4[16][16] = {{0}};   // [0] plane of :108
  long block_luma_8x8[4][64] = {{0}};    // :112
  long block_luma_dc[16] = {0};          // :100
  long block_luma_ac[16][15] = {{0}};    // :104
  long block_chroma_dc[2][4] = {{0}};    // :115 (4:2:0 uses the first four)
  long block_chroma_ac[2][4][15] = {{{0}}};  // :119
};

class Frame {
 public:
  // Frame::new(&slice): planes are allocated on the device by the library; here only the batch.
  Frame(const dryv_frame_params& fp, dryv_recon_ctx* ctx) : fp_(fp), ctx_(ctx) {
    n_ = (size_t)fp.pic_width_in_mbs * fp.pic_height_in_mbs;
    mbs_.resize(n_);
    coeffs_.assign(n_ * DRYV_COEFFS_PER_MB, 0);
    width_l = fp.pic_width_in_mbs * 16;
    height_l = fp.pic_height_in_mbs * 16;
    width_c = fp.pic_width_in_mbs * 8;
    height_c = fp.pic_height_in_mbs * 8;
  }

  // Frame::decode(&mut slice) for the macroblock at the next mbaddr. I_PCM / inter are todo!() in the
  // reference (frame/mod.rs:86,88): here DRYV_E_UNSUPPORTED, nothing is queued.
  int decode(const Macroblock& mb) {
    if (next_ >= n_) return DRYV_E_INVALID;
    if (mb.mode != Macroblock::Intra4x4 && mb.mode != Macroblock::Intra8x8 && mb.mode != Macroblock::Intra16x16)
      return DRYV_E_UNSUPPORTED;
    if (mb.qpy < 0 || mb.qpy > 51) return DRYV_E_UNSUPPORTED;
    dryv_mb_desc d;
    std::memset(&d, 0, sizeof(d));
    d.mb_kind = (uint8_t)mb.mode;
    d.i16_pred_mode = (uint8_t)mb.intra16x16_pred_mode;
    d.intra_chroma_pred_mode = (uint8_t)mb.intra_chroma_pred_mode;
    d.qp = (uint8_t)mb.qpy;
    d.nz_mask = 0xFFFF;
    const bool i8 = mb.mode == Macroblock::Intra8x8;
    for (int i = 0; i < (i8 ? 4 : 16); i++) {
      const int prev = i8 ? mb.prev_intra8x8_pred_mode_flag[i] : mb.prev_intra4x4_pred_mode_flag[i];
      const int rem = i8 ? mb.rem_intra8x8_pred_mode[i] : mb.rem_intra4x4_pred_mode[i];
      if (prev) d.prev_flags |= (uint16_t)(1u << i);
      d.rem_modes[i >> 1] |= (uint8_t)((rem & 7) << (4 * (i & 1)));
    }
    int16_t* c = &coeffs_[next_ * DRYV_COEFFS_PER_MB];
    int bad = 0;
    auto put = [&](long v) {  // isize -> i16 with range check (the FFI carries int16)
      if (v < -32768 || v > 32767) bad = 1;
      *c++ = (int16_t)v;
    };
    if (mb.mode == Macroblock::Intra4x4) {
      for (int b = 0; b < 16; b++) for (int k = 0; k < 16; k++) put(mb.block_luma_4x4[b][k]);
    } else if (i8) {
      for (int b = 0; b < 4; b++) for (int k = 0; k < 64; k++) put(mb.block_luma_8x8[b][k]);
    } else {
      for (int k = 0; k < 16; k++) put(mb.block_luma_dc[k]);
      for (int b = 0; b < 16; b++) for (int k = 0; k < 15; k++) put(mb.block_luma_ac[b][k]);
    }
    for (int pl = 0; pl < 2; pl++) {
      for (int k = 0; k < 4; k++) put(mb.block_chroma_dc[pl][k]);
      for (int b = 0; b < 4; b++) for (int k = 0; k < 15; k++) put(mb.block_chroma_ac[pl][b][k]);
    }
    if (bad) return DRYV_E_UNSUPPORTED;
    mbs_[next_++] = d;
    yuv_.clear();
    return DRYV_OK;
  }

  // Submits the frame (once) and returns the planes: Y, then Cb, then Cr, row-major, uncropped.
  int planes(const uint8_t** yuv, size_t* bytes) {
    if (yuv_.empty()) {
      if (next_ != n_) return DRYV_E_STATE;
      yuv_.resize(dryv_recon_frame_bytes(&fp_));
      int st = dryv_recon_submit(ctx_, &fp_, 1, mbs_.data(), coeffs_.data());
      if (st != DRYV_OK) { yuv_.clear(); return st; }
      st = dryv_recon_wait(ctx_, yuv_.data(), yuv_.size());
      if (st != DRYV_OK) { yuv_.clear(); return st; }
    }
    *yuv = yuv_.data();
    *bytes = yuv_.size();
    return DRYV_OK;
  }

  // Frame::write_to_yuv_file (frame/mod.rs:48-70)
  int write_to_yuv_file(const char* file_path) {
    const uint8_t* p;
    size_t n;
    int st = planes(&p, &n);
    if (st != DRYV_OK) return st;
    FILE* f = std::fopen(file_path, "wb");
    if (!f) return DRYV_E_INVALID;
    const size_t w = std::fwrite(p, 1, n, f);
    std::fclose(f);
    return w == n ? DRYV_OK : DRYV_E_INVALID;
  }

  int width_l = 0, height_l = 0, width_c = 0, height_c = 0;  // frame/mod.rs:22-25

 private:
  dryv_frame_params fp_;
  dryv_recon_ctx* ctx_;
  size_t n_ = 0, next_ = 0;
  std::vector<dryv_mb_desc> mbs_;
  std::vector<int16_t> coeffs_;
  std::vector<uint8_t> yuv_;
};

}  // namespace dryv
