// h264_islice.hpp — host producer for the reconstruction backend: ISO-BMFF / Annex-B demux, SPS / PPS / slice-header
// parse and the CABAC macroblock layer of an I slice, producing exactly the batch the C ABI takes (dryv_frame_params,
// dryv_mb_desc[], int16 coefficient lists); plus the inverse (a CABAC I-slice ENCODER writing an Annex-B stream), so
// that synthetic all-intra batches exist as real bitstreams.
//
// This is SURVEY.md section 8(f) row 1: the caller's side of the path, which in dryv stays on the CPU. It restates
//   src/video/cabac/mod.rs:89-210 (macroblock_layer), :433-675 (residual), :1207-1278 (decision / bypass / terminate)
//   src/video/slice/header.rs:145-315, src/video/atom/avcc/sps.rs:42-121, pps.rs:30-58,
//   src/video/sample/nal.rs:232-253 (length-prefixed NAL units), src/byte/bit.rs:96-108,144-149 (ue/se, emulation
//   prevention)
// as a plain-C++ H.264 (ITU-T H.264 clauses 7.3, 9.1, 9.3) implementation restricted to the backend's domain:
// I slices, CABAC, frame macroblocks, 4:2:0, 8 bit, one slice per picture, flat scaling lists, no I_PCM.
// Anything else is reported as unsupported. Header-only, no dependencies; used by libdryv_h264.so and frame_harness.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dryv_recon.h"

namespace dryv {
namespace h264 {

#include "cabac_tables.inc"

struct Error {
  std::string what;
};
[[noreturn]] inline void fail(const std::string& w) { throw Error{w}; }

// ---- bits ------------------------------------------------------------------------------------------------------------
struct BitReader {  // over an RBSP (emulation prevention already removed)
  const uint8_t* p = nullptr;
  size_t n = 0, pos = 0;  // pos in bits
  unsigned bit() {
    if (pos >= 8 * n) fail("bitstream exhausted");
    const unsigned b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
    pos++;
    return b;
  }
  unsigned bits(int k) {
    unsigned v = 0;
    for (int i = 0; i < k; i++) v = (v << 1) | bit();
    return v;
  }
  unsigned ue() {  // 9.1 (byte/bit.rs:96-108)
    int z = 0;
    while (bit() == 0) {
      if (++z > 31) fail("ue(v) too long");
    }
    return z == 0 ? 0u : ((1u << z) - 1u + bits(z));
  }
  int se() {
    const unsigned k = ue();
    return (k & 1u) ? (int)((k + 1) >> 1) : -(int)(k >> 1);
  }
  bool aligned() const { return (pos & 7) == 0; }
  size_t bits_left() const { return 8 * n - pos; }
};

struct BitWriter {
  std::vector<uint8_t> out;
  int nbits = 0;
  void bit(unsigned b) {
    if ((nbits & 7) == 0) out.push_back(0);
    if (b) out.back() |= (uint8_t)(0x80u >> (nbits & 7));
    nbits++;
  }
  void bits(unsigned v, int k) {
    for (int i = k - 1; i >= 0; i--) bit((v >> i) & 1u);
  }
  void ue(unsigned v) {
    const unsigned x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    bits(0, len);
    bits(x, len + 1);
  }
  void se(int v) { ue(v > 0 ? (unsigned)(2 * v - 1) : (unsigned)(-2 * v)); }
  void trailing() {  // rbsp_trailing_bits
    bit(1);
    while (nbits & 7) bit(0);
  }
};

// NAL payload -> RBSP: drops the emulation_prevention_three_byte of every 00 00 03 (byte/bit.rs:144-149)
inline std::vector<uint8_t> unescape(const uint8_t* p, size_t n) {
  std::vector<uint8_t> r;
  r.reserve(n);
  int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && p[i] == 3) {
      zeros = 0;
      continue;
    }
    r.push_back(p[i]);
    zeros = p[i] == 0 ? zeros + 1 : 0;
  }
  return r;
}
inline void append_nal_annexb(std::vector<uint8_t>& out, uint8_t header, const std::vector<uint8_t>& rbsp) {
  out.insert(out.end(), {0, 0, 0, 1, header});
  int zeros = 0;
  for (uint8_t b : rbsp) {
    if (zeros >= 2 && b <= 3) {
      out.push_back(3);
      zeros = 0;
    }
    out.push_back(b);
    zeros = b == 0 ? zeros + 1 : 0;
  }
}

// ---- parameter sets ----------------------------------------------------------------------------------------------------
// Scaling matrices as the reference decodes them (atom/avcc/sps.rs:150-250): entries in the order they are coded (zig-zag),
// which is the order dryv_frame_params takes. Two things the reference does differently from 7.4.2.1.1 / 7.4.2.2 are kept,
// because the output must be the reference's: a list whose scaling_list_present_flag is 0 becomes the Default_* table
// (the standard's fall-back rules A / B would copy the previous list for lists 1, 2, 4, 5, 7..), and when both parameter
// sets carry a matrix the SPS's wins (slice/header.rs:317-332; the standard lets the PPS's override it).
struct ScalingLists {
  bool present = false;
  int n8 = 0;               // 8x8 lists carried (2 for 4:2:0)
  uint8_t l4[6][16];
  uint8_t l8[6][64];
};
static const uint8_t SL_DEFAULT_4X4[2][16] = {   // Table 7-3: Default_4x4_Intra, Default_4x4_Inter
    {6, 13, 13, 20, 20, 20, 28, 28, 28, 28, 32, 32, 32, 37, 37, 42},
    {10, 14, 14, 20, 20, 20, 24, 24, 24, 24, 27, 27, 27, 30, 30, 34}};
static const uint8_t SL_DEFAULT_8X8[2][64] = {   // Table 7-4: Default_8x8_Intra, Default_8x8_Inter
    {6,  10, 10, 13, 11, 13, 16, 16, 16, 16, 18, 18, 18, 18, 18, 23, 23, 23, 23, 23, 23, 25, 25, 25, 25, 25, 25, 25, 27, 27, 27, 27,
     27, 27, 27, 27, 29, 29, 29, 29, 29, 29, 29, 31, 31, 31, 31, 31, 31, 33, 33, 33, 33, 33, 36, 36, 36, 36, 38, 38, 38, 40, 40, 42},
    {9,  13, 13, 15, 13, 15, 17, 17, 17, 17, 19, 19, 19, 19, 19, 21, 21, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 24, 24, 24, 24,
     24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 27, 27, 27, 27, 27, 27, 28, 28, 28, 28, 28, 30, 30, 30, 30, 32, 32, 32, 33, 33, 35}};
// 7.3.2.1.1.1 scaling_list(): returns useDefaultScalingMatrixFlag (sps.rs:179-198)
inline bool read_scaling_list(BitReader& r, uint8_t* out, int size) {
  bool useDefault = false;
  int last = 8, next = 8;
  for (int j = 0; j < size; j++) {
    if (next != 0) {
      const int delta = r.se();
      if (delta < -128 || delta > 127) fail("scaling list: delta_scale out of range");
      next = (last + delta + 256) % 256;
      useDefault = j == 0 && next == 0;
    }
    out[j] = (uint8_t)(next == 0 ? last : next);
    last = out[j];
  }
  return useDefault;
}
inline ScalingLists read_scaling_matrix(BitReader& r, int n_lists) {   // sps.rs:206-249 (ScalingLists::new)
  ScalingLists L;
  L.present = true;
  L.n8 = n_lists - 6;
  memset(L.l4, 16, sizeof L.l4);
  memset(L.l8, 16, sizeof L.l8);
  for (int i = 0; i < n_lists; i++) {
    const bool present = r.bit() != 0;
    if (i < 6) {
      if (!present || read_scaling_list(r, L.l4[i], 16)) memcpy(L.l4[i], SL_DEFAULT_4X4[i < 3 ? 0 : 1], 16);
    } else {
      if (!present || read_scaling_list(r, L.l8[i - 6], 64)) memcpy(L.l8[i - 6], SL_DEFAULT_8X8[(i & 1) ? 1 : 0], 64);
    }
  }
  return L;
}
inline void write_scaling_list(BitWriter& w, const uint8_t* l, int size) {
  int last = 8;
  for (int j = 0; j < size; j++) {
    if (l[j] == 0) fail("encoder: a scaling list entry of 0 cannot be coded");
    int d = (int)l[j] - last;
    if (d > 127) d -= 256;
    if (d < -128) d += 256;
    w.se(d);
    last = l[j];
  }
}

struct Sps {
  int id = 0, profile_idc = 0, level_idc = 0, chroma_format_idc = 1, bit_depth_luma = 8, bit_depth_chroma = 8;
  int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4;
  int width_mbs = 0, height_map_units = 0;
  bool frame_mbs_only = true, delta_pic_order_always_zero = false;
  int crop[4] = {0, 0, 0, 0};
  ScalingLists scaling;
};
struct Pps {
  int id = 0, sps_id = 0;
  bool cabac = false, bottom_field_pic_order = false, deblocking_control = false, constrained_intra = false;
  bool redundant_pic_cnt = false, transform8x8 = false;
  int num_slice_groups = 1, pic_init_qp = 26, chroma_qp_offset = 0, second_chroma_qp_offset = 0;
  ScalingLists scaling;
};
// What the reconstruction library accepts (dryv_recon_check_params): the parser refuses anything larger, so that no later
// size computation can overflow and the caller's buffers, sized from the parameters it is given, are the ones written.
constexpr int MAX_WIDTH_MBS = 1024, MAX_HEIGHT_MBS = 65535;

inline Sps parse_sps(const std::vector<uint8_t>& rbsp) {  // 7.3.2.1.1 (atom/avcc/sps.rs:42-121)
  BitReader r{rbsp.data(), rbsp.size(), 0};
  Sps s;
  s.profile_idc = (int)r.bits(8);
  r.bits(8);
  s.level_idc = (int)r.bits(8);
  const unsigned sid = r.ue();
  if (sid > 31) fail("SPS: seq_parameter_set_id out of range");
  s.id = (int)sid;
  const int p = s.profile_idc;
  if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128 || p == 138 ||
      p == 139 || p == 134 || p == 135) {
    const unsigned cfi = r.ue();
    if (cfi > 3) fail("SPS: chroma_format_idc out of range");
    s.chroma_format_idc = (int)cfi;
    if (s.chroma_format_idc == 3) r.bit();
    const unsigned bl = r.ue(), bc = r.ue();
    if (bl > 6 || bc > 6) fail("SPS: bit depth out of range");
    s.bit_depth_luma = 8 + (int)bl;
    s.bit_depth_chroma = 8 + (int)bc;
    r.bit();  // qpprime_y_zero_transform_bypass_flag
    if (r.bit()) s.scaling = read_scaling_matrix(r, s.chroma_format_idc != 3 ? 8 : 12);  // seq_scaling_matrix_present_flag
  }
  const unsigned lfn = r.ue();
  if (lfn > 12) fail("SPS: log2_max_frame_num out of range");
  s.log2_max_frame_num = 4 + (int)lfn;
  const unsigned pt = r.ue();
  if (pt > 2) fail("SPS: pic_order_cnt_type out of range");
  s.poc_type = (int)pt;
  if (s.poc_type == 0) {
    const unsigned lp = r.ue();
    if (lp > 12) fail("SPS: log2_max_pic_order_cnt_lsb out of range");
    s.log2_max_poc_lsb = 4 + (int)lp;
  } else if (s.poc_type == 1) {
    s.delta_pic_order_always_zero = r.bit() != 0;
    r.se();
    r.se();
    const unsigned k = r.ue();
    if (k > 255) fail("SPS: num_ref_frames_in_pic_order_cnt_cycle out of range");
    for (unsigned i = 0; i < k; i++) r.se();
  }
  r.ue();   // max_num_ref_frames
  r.bit();  // gaps_in_frame_num_value_allowed_flag
  // picture size: bounded here, by what the reconstruction library takes, before anything is sized from it
  const unsigned wm1 = r.ue(), hm1 = r.ue();
  if (wm1 >= (unsigned)MAX_WIDTH_MBS || hm1 >= (unsigned)MAX_HEIGHT_MBS) fail("SPS: picture larger than 1024 x 65535 macroblocks");
  s.width_mbs = 1 + (int)wm1;
  s.height_map_units = 1 + (int)hm1;
  s.frame_mbs_only = r.bit() != 0;
  if (!s.frame_mbs_only) r.bit();
  r.bit();  // direct_8x8_inference_flag
  if (r.bit()) {
    unsigned c[4];
    for (int k = 0; k < 4; k++) c[k] = r.ue();
    // in units of two luma samples (4:2:0 frame pictures): what is left must be a picture
    if (c[0] > 8u * MAX_WIDTH_MBS || c[1] > 8u * MAX_WIDTH_MBS || c[0] + c[1] >= 8u * (unsigned)s.width_mbs ||
        c[2] > 8u * MAX_HEIGHT_MBS || c[3] > 8u * MAX_HEIGHT_MBS || c[2] + c[3] >= 8u * (unsigned)s.height_map_units)
      fail("SPS: frame cropping rectangle outside the picture");
    for (int k = 0; k < 4; k++) s.crop[k] = (int)c[k];
  }
  return s;  // (VUI not needed)
}

// chroma_format_idc: of the sequence parameter set the PPS refers to (it decides how many 8x8 lists a matrix carries)
inline Pps parse_pps(const std::vector<uint8_t>& rbsp, int chroma_format_idc = 1) {  // 7.3.2.2 (atom/avcc/pps.rs:30-58)
  BitReader r{rbsp.data(), rbsp.size(), 0};
  Pps p;
  const unsigned pid = r.ue(), sid = r.ue();
  if (pid > 255 || sid > 31) fail("PPS: parameter set id out of range");
  p.id = (int)pid;
  p.sps_id = (int)sid;
  p.cabac = r.bit() != 0;
  p.bottom_field_pic_order = r.bit() != 0;
  p.num_slice_groups = 1 + (int)r.ue();
  if (p.num_slice_groups > 1) fail("unsupported: slice groups");
  r.ue();
  r.ue();
  r.bit();    // weighted_pred_flag
  r.bits(2);  // weighted_bipred_idc
  p.pic_init_qp = 26 + r.se();
  if (p.pic_init_qp < 0 || p.pic_init_qp > 51) fail("PPS: pic_init_qp out of range");
  r.se();
  p.chroma_qp_offset = r.se();
  if (p.chroma_qp_offset < -12 || p.chroma_qp_offset > 12) fail("PPS: chroma_qp_index_offset out of range");
  p.second_chroma_qp_offset = p.chroma_qp_offset;  // transform.rs:198-203: falls back to the first offset
  p.deblocking_control = r.bit() != 0;
  p.constrained_intra = r.bit() != 0;
  p.redundant_pic_cnt = r.bit() != 0;
  // more_rbsp_data(): anything left besides the trailing bits
  size_t last = rbsp.size();
  while (last > 0 && rbsp[last - 1] == 0) last--;
  if (last > 0) {
    int tz = 0;
    while (((rbsp[last - 1] >> tz) & 1) == 0) tz++;
    const size_t end_bit = 8 * last - tz - 1;  // position of the rbsp_stop_one_bit
    if (r.pos < end_bit) {
      p.transform8x8 = r.bit() != 0;
      if (r.bit())  // pic_scaling_matrix_present_flag (pps.rs:77-82)
        p.scaling = read_scaling_matrix(r, 6 + (chroma_format_idc != 3 ? 2 : 6) * (p.transform8x8 ? 1 : 0));
      p.second_chroma_qp_offset = r.se();
      if (p.second_chroma_qp_offset < -12 || p.second_chroma_qp_offset > 12) fail("PPS: second_chroma_qp_index_offset out of range");
    }
  }
  return p;
}

struct SliceHeader {
  int first_mb = 0, slice_type = 0, frame_num = 0, idr_pic_id = 0, slice_qp = 26;
  int disable_deblocking_filter_idc = 0, alpha_c0_offset_div2 = 0, beta_offset_div2 = 0;  // slice/header.rs:609-640
  size_t data_bit_pos = 0;  // where slice_data() starts (byte aligned for CABAC)
};
inline SliceHeader parse_slice_header(BitReader& r, const Sps& s, const Pps& p, int nal_unit_type, int nal_ref_idc) {
  SliceHeader h;  // 7.3.3 (slice/header.rs:145-315), I slices only
  h.first_mb = (int)r.ue();
  h.slice_type = (int)r.ue();
  if (h.slice_type % 5 != 2) fail("unsupported: not an I slice");
  if (r.ue() != (unsigned)p.id) fail("slice header: not the picture parameter set it was paired with");
  h.frame_num = (int)r.bits(s.log2_max_frame_num);
  if (!s.frame_mbs_only && r.bit()) fail("unsupported: field picture");
  if (nal_unit_type == 5) h.idr_pic_id = (int)r.ue();
  if (s.poc_type == 0) {
    r.bits(s.log2_max_poc_lsb);
    if (p.bottom_field_pic_order) r.se();
  } else if (s.poc_type == 1 && !s.delta_pic_order_always_zero) {
    r.se();
    if (p.bottom_field_pic_order) r.se();
  }
  if (p.redundant_pic_cnt) r.ue();
  if (nal_ref_idc != 0) {  // dec_ref_pic_marking()
    if (nal_unit_type == 5) {
      r.bit();
      r.bit();
    } else if (r.bit()) {
      for (;;) {
        const unsigned op = r.ue();
        if (op == 0) break;
        if (op == 1 || op == 3) r.ue();
        if (op == 2) r.ue();
        if (op == 3 || op == 4) r.ue();
      }
    }
  }
  h.slice_qp = p.pic_init_qp + r.se();
  if (h.slice_qp < 0 || h.slice_qp > 51) fail("slice header: slice_qp out of range");
  if (p.deblocking_control) {
    h.disable_deblocking_filter_idc = (int)r.ue();
    if (h.disable_deblocking_filter_idc != 1) {
      h.alpha_c0_offset_div2 = r.se();
      h.beta_offset_div2 = r.se();
    }
    if (h.disable_deblocking_filter_idc > 2 || h.alpha_c0_offset_div2 < -6 || h.alpha_c0_offset_div2 > 6 || h.beta_offset_div2 < -6 ||
        h.beta_offset_div2 > 6)
      fail("slice header: deblocking syntax elements out of range");
  }
  if (h.first_mb != 0) fail("unsupported: more than one slice per picture");
  return h;
}

// ---- CABAC engines (9.3.1.2, 9.3.3.2; cabac/mod.rs:1207-1278) ---------------------------------------------------------------
struct CabacContexts {
  uint8_t state[1024], mps[1024];
  void init(int slice_qp) {  // 9.3.1.1
    const int q = slice_qp < 0 ? 0 : (slice_qp > 51 ? 51 : slice_qp);
    for (int i = 0; i < 1024; i++) {
      int pre = ((CABAC_INIT_I[i][0] * q) >> 4) + CABAC_INIT_I[i][1];
      pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
      if (pre <= 63) {
        state[i] = (uint8_t)(63 - pre);
        mps[i] = 0;
      } else {
        state[i] = (uint8_t)(pre - 64);
        mps[i] = 1;
      }
    }
  }
};

struct CabacDecoder {
  BitReader* r = nullptr;
  CabacContexts c;
  unsigned range = 510, offset = 0;
  long bins = 0;
  // test hook: every decoded bin in order, value | kind << 1 (kind 0 context-coded, 1 bypass, 2 terminate): what an
  // independent restatement of the syntax (oracle/islice_syntax.py) parses to check this parser's record / list mapping
  std::vector<uint8_t>* log = nullptr;
  unsigned note(unsigned bin, unsigned kind) {
    if (log) log->push_back((uint8_t)(bin | (kind << 1)));
    return bin;
  }
  void start(BitReader* br, int slice_qp) {
    r = br;
    c.init(slice_qp);
    range = 510;
    offset = r->bits(9);
  }
  unsigned decision(int ctx) {
    bins++;
    const unsigned lps = CABAC_RANGE_LPS[c.state[ctx]][(range >> 6) & 3];
    unsigned bin;
    range -= lps;
    if (offset >= range) {
      bin = 1u - c.mps[ctx];
      offset -= range;
      range = lps;
      if (c.state[ctx] == 0) c.mps[ctx] = (uint8_t)(1 - c.mps[ctx]);
      c.state[ctx] = CABAC_NEXT_LPS[c.state[ctx]];
    } else {
      bin = c.mps[ctx];
      c.state[ctx] = CABAC_NEXT_MPS[c.state[ctx]];
    }
    while (range < 256) {
      range <<= 1;
      offset = (offset << 1) | r->bit();
    }
    return note(bin, 0);
  }
  unsigned bypass() {
    bins++;
    offset = (offset << 1) | r->bit();
    if (offset >= range) {
      offset -= range;
      return note(1, 1);
    }
    return note(0, 1);
  }
  unsigned terminate() {
    bins++;
    range -= 2;
    if (offset >= range) return note(1, 2);
    while (range < 256) {
      range <<= 1;
      offset = (offset << 1) | r->bit();
    }
    return note(0, 2);
  }
};

struct CabacEncoder {  // 9.3.4.2-9.3.4.5
  BitWriter* w = nullptr;
  CabacContexts c;
  unsigned low = 0, range = 510;
  int outstanding = 0;
  bool first = true;
  void start(BitWriter* bw, int slice_qp) {
    w = bw;
    c.init(slice_qp);
    low = 0;
    range = 510;
    outstanding = 0;
    first = true;
  }
  void put(unsigned b) {
    if (first) first = false;
    else w->bit(b);
    for (; outstanding > 0; outstanding--) w->bit(1 - b);
  }
  void renorm() {
    while (range < 256) {
      if (low < 256) put(0);
      else if (low >= 512) {
        low -= 512;
        put(1);
      } else {
        low -= 256;
        outstanding++;
      }
      range <<= 1;
      low <<= 1;
    }
  }
  void decision(int ctx, unsigned bin) {
    const unsigned lps = CABAC_RANGE_LPS[c.state[ctx]][(range >> 6) & 3];
    range -= lps;
    if (bin != c.mps[ctx]) {
      low += range;
      range = lps;
      if (c.state[ctx] == 0) c.mps[ctx] = (uint8_t)(1 - c.mps[ctx]);
      c.state[ctx] = CABAC_NEXT_LPS[c.state[ctx]];
    } else {
      c.state[ctx] = CABAC_NEXT_MPS[c.state[ctx]];
    }
    renorm();
  }
  void bypass(unsigned bin) {
    low <<= 1;
    if (bin) low += range;
    if (low >= 1024) {
      put(1);
      low -= 1024;
    } else if (low < 512) put(0);
    else {
      low -= 512;
      outstanding++;
    }
  }
  void terminate(unsigned bin) {
    range -= 2;
    if (bin) {
      low += range;
      range = 2;
      renorm();
      put((low >> 9) & 1);
      w->bits(((low >> 7) & 3) | 1, 2);  // (the last bit written is the rbsp_stop_one_bit)
    } else {
      renorm();
    }
  }
};

// ---- macroblock layer ---------------------------------------------------------------------------------------------------
// What a decoded macroblock leaves behind for its neighbours' context selection (9.3.3.1.1.x)
struct MbCtx {
  uint8_t avail = 0, i_nxn = 0, i16 = 0, t8 = 0, chroma_mode = 0, cbp_luma = 0, cbp_chroma = 0;
  uint16_t cbf_luma = 0;     // bit blkIdx (z-order): coded_block_flag of the 4x4 luma block (or of its 8x8 block)
  uint8_t cbf_dc = 0;        // bit 0 Intra16x16 DC, bit 1 Cb DC, bit 2 Cr DC
  uint8_t cbf_ac[2] = {0, 0};  // bit blkIdx: chroma AC blocks of Cb / Cr
};

// neighbouring 4x4 luma block of blkIdx (z-order) to the left (A) / above (B): {blkIdx there, 1 if in the neighbour MB}
inline void luma_nb(int blk, bool left, int& nblk, bool& outside) {
  const int bx = ((blk >> 1) & 2) | (blk & 1), by = ((blk >> 2) & 2) | ((blk >> 1) & 1);
  int nx = bx, ny = by;
  outside = false;
  if (left) {
    if (bx == 0) {
      nx = 3;
      outside = true;
    } else nx = bx - 1;
  } else {
    if (by == 0) {
      ny = 3;
      outside = true;
    } else ny = by - 1;
  }
  nblk = 8 * (ny >> 1) + 4 * (nx >> 1) + 2 * (ny & 1) + (nx & 1);
}

static const int CAT_CBF[5] = {0, 4, 8, 12, 16};
static const int CAT_SIG[5] = {0, 15, 29, 44, 47};
static const int CAT_ABS[5] = {0, 10, 20, 30, 39};

struct ParsedFrame {
  dryv_frame_params fp;
  std::vector<dryv_mb_desc> mbs;
  std::vector<int16_t> coeffs;  // 384 per macroblock
  // diagnostics
  long bins = 0;
  size_t slice_bytes = 0, bits_unread = 0;
  bool tail_ok = false;  // the engine stopped right behind the rbsp_stop_one_bit and only zero bits follow
  int kinds[3] = {0, 0, 0};
  int slice_qp = 0;
  // the SPS's frame cropping rectangle in luma samples (left, right, top, bottom): sps.rs:252-267. The reference parses
  // and ignores it; here it is handed to the caller, who may pass it on to the output stage (dryv_output_desc)
  int crop[4] = {0, 0, 0, 0};
  // the slice header's deblocking syntax elements (parsed and unused by the reference: header.rs:609-640, README.md:15);
  // what dryv_recon_deblock_device takes
  dryv_deblock_params deblock = {0, 0, 0, 0};
};

// Shared walk over the macroblock layer; CODER is CabacDecoder (fills mbs/coeffs) or CabacEncoder (reads them).
template <bool ENCODE, class CODER>
struct MbLayer {
  CODER& cd;
  int W, H;
  bool transform8x8_mode;
  std::vector<MbCtx> ctx;
  int qp_prev = 26;
  bool prev_delta_nonzero = false;

  MbLayer(CODER& c, int w, int h, bool t8, int slice_qp) : cd(c), W(w), H(h), transform8x8_mode(t8), ctx((size_t)w * h), qp_prev(slice_qp) {}

  unsigned dec(int ctxIdx, unsigned bin) {
    if constexpr (ENCODE) {
      cd.decision(ctxIdx, bin);
      return bin;
    } else {
      (void)bin;
      return cd.decision(ctxIdx);
    }
  }
  unsigned byp(unsigned bin) {
    if constexpr (ENCODE) {
      cd.bypass(bin);
      return bin;
    } else {
      (void)bin;
      return cd.bypass();
    }
  }
  unsigned term(unsigned bin) {
    if constexpr (ENCODE) {
      cd.terminate(bin);
      return bin;
    } else {
      (void)bin;
      return cd.terminate();
    }
  }

  // residual_block_cabac (7.3.5.3.3, cabac/mod.rs:433-675): coefficient list `c` of maxNum entries, positions
  // start..end coded. Returns coded_block_flag.
  bool residual_block(int cat, int16_t* c, int start, int end, int maxNum, int cbfInc) {
    bool cbf = true;
    if (ENCODE) {
      cbf = false;
      for (int k = start; k <= end; k++) cbf = cbf || c[k] != 0;
    }
    if (maxNum != 64) cbf = dec(85 + CAT_CBF[cat] + cbfInc, cbf) != 0;
    if (!cbf) {
      if (!ENCODE)
        for (int k = start; k <= end; k++) c[k] = 0;
      return false;
    }
    const int sigBase = cat == 5 ? 402 : 105 + CAT_SIG[cat], lastBase = cat == 5 ? 417 : 166 + CAT_SIG[cat];
    const int absBase = cat == 5 ? 426 : 227 + CAT_ABS[cat];
    int numCoeff = end + 1;
    bool sig[64];
    int lastNz = start;
    if (ENCODE)
      for (int k = start; k <= end; k++)
        if (c[k] != 0) lastNz = k;
    for (int i = start; i < numCoeff - 1; i++) {
      const int inc = cat == 5 ? CABAC_SIG8X8_INC[i] : (cat == 3 ? (i < 2 ? i : 2) : i);
      sig[i] = dec(sigBase + inc, ENCODE ? (c[i] != 0) : 0) != 0;
      if (sig[i]) {
        const int linc = cat == 5 ? CABAC_LAST8X8_INC[i] : (cat == 3 ? (i < 2 ? i : 2) : i);
        if (dec(lastBase + linc, ENCODE ? (i == lastNz) : 0)) {
          numCoeff = i + 1;
          break;
        }
      }
    }
    sig[numCoeff - 1] = true;
    int eq1 = 0, gt1 = 0;
    for (int i = numCoeff - 1; i >= start; i--) {
      if (!sig[i]) {
        if (!ENCODE) c[i] = 0;
        continue;
      }
      const int want = ENCODE ? ((c[i] < 0 ? -(int)c[i] : (int)c[i]) - 1) : 0;
      const int inc0 = gt1 != 0 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4);
      const int capN = 4 - (cat == 3 ? 1 : 0);
      const int incN = 5 + (gt1 < capN ? gt1 : capN);
      int v = 0;  // coeff_abs_level_minus1: UEG0, prefix TU cMax 14, suffix Exp-Golomb order 0 (bypass)
      if (dec(absBase + inc0, want > 0)) {
        v = 1;
        while (v < 14 && dec(absBase + incN, want > v)) v++;
        if (v == 14) {
          int k = 0;
          int rest = ENCODE ? want - 14 : 0;
          if (ENCODE) {
            while (rest >= (1 << k)) {
              byp(1);
              rest -= 1 << k;
              k++;
            }
            byp(0);
            while (k--) byp((rest >> k) & 1);
          } else {
            while (byp(0)) {
              v += 1 << k;
              if (++k > 20) fail("coeff_abs_level_minus1 suffix too long");
            }
            while (k--) v += (int)byp(0) << k;
          }
        }
      }
      const unsigned neg = byp(ENCODE ? (c[i] < 0) : 0);
      if (!ENCODE) {
        const int mag = v + 1;
        if (mag > 32768 || (mag == 32768 && !neg)) fail("coefficient outside int16");
        c[i] = (int16_t)(neg ? -mag : mag);
      }
      const int mag1 = ENCODE ? want + 1 : v + 1;
      if (mag1 == 1) eq1++;
      else gt1++;
    }
    if (!ENCODE)
      for (int i = numCoeff; i <= end; i++) c[i] = 0;
    return true;
  }

  void macroblock(int addr, dryv_mb_desc& d, int16_t* co, bool last_mb) {
    const int mx = addr % W, my = addr / W;
    const MbCtx* A = mx > 0 ? &ctx[addr - 1] : nullptr;
    const MbCtx* B = my > 0 ? &ctx[addr - W] : nullptr;
    MbCtx& M = ctx[addr];
    M = MbCtx();
    M.avail = 1;

    // ---- what the encoder knows in advance
    int kind = ENCODE ? d.mb_kind : 0, i16mode = ENCODE ? d.i16_pred_mode : 0;
    int cbpL = 0, cbpC = 0;
    if (ENCODE) {
      if (kind > 2) fail("encoder: mb_kind out of range");
      bool acC = false, dcC = false;
      for (int pl = 0; pl < 2; pl++) {
        const int16_t* c = co + 256 + 64 * pl;
        for (int k = 0; k < 4; k++) dcC = dcC || c[k] != 0;
        for (int k = 4; k < 64; k++) acC = acC || c[k] != 0;
      }
      cbpC = acC ? 2 : (dcC ? 1 : 0);
      if (kind == 2) {
        bool ac = false;
        for (int k = 16; k < 256; k++) ac = ac || co[k] != 0;
        cbpL = ac ? 15 : 0;
      } else {
        for (int b8 = 0; b8 < 4; b8++) {
          bool nz = false;
          for (int k = 64 * b8; k < 64 * b8 + 64; k++) nz = nz || co[k] != 0;
          if (nz) cbpL |= 1 << b8;
        }
      }
    }

    // ---- mb_type (9.3.3.1.1.3, binarization table 9-36)
    const int incT = ((A && !A->i_nxn) ? 1 : 0) + ((B && !B->i_nxn) ? 1 : 0);
    const unsigned notNxN = dec(3 + incT, ENCODE ? kind == 2 : 0);
    if (!notNxN) {
      M.i_nxn = 1;
      if (transform8x8_mode) {  // transform_size_8x8_flag (9.3.3.1.1.10)
        const int inc = ((A && A->t8) ? 1 : 0) + ((B && B->t8) ? 1 : 0);
        M.t8 = (uint8_t)dec(399 + inc, ENCODE ? kind == 1 : 0);
      } else if (ENCODE && kind == 1) fail("encoder: Intra8x8 macroblock without transform_8x8_mode_flag");
      kind = M.t8 ? 1 : 0;
      // prev_intra*_pred_mode_flag / rem_intra*_pred_mode
      const int n = M.t8 ? 4 : 16;
      unsigned flags = 0;
      uint8_t rem[16] = {0};
      for (int k = 0; k < n; k++) {
        const unsigned pf = dec(68, ENCODE ? (d.prev_flags >> k) & 1u : 0);
        flags |= pf << k;
        if (!pf) {
          const unsigned want = ENCODE ? (d.rem_modes[k >> 1] >> (4 * (k & 1))) & 7u : 0;
          unsigned v = dec(69, want & 1);
          v |= dec(69, (want >> 1) & 1) << 1;
          v |= dec(69, (want >> 2) & 1) << 2;
          rem[k] = (uint8_t)v;
        }
      }
      if (!ENCODE) {
        d.prev_flags = (uint16_t)flags;
        for (int k = 0; k < 8; k++) d.rem_modes[k] = (uint8_t)(rem[2 * k] | (rem[2 * k + 1] << 4));
      }
    } else {
      if (term(0)) fail("unsupported: I_PCM macroblock");
      M.i16 = 1;
      kind = 2;
      const unsigned l = dec(3 + 3, ENCODE ? cbpL != 0 : 0);
      unsigned ch = dec(3 + 4, ENCODE ? cbpC != 0 : 0);
      if (ch) ch = 1 + dec(3 + 5, ENCODE ? cbpC == 2 : 0);
      unsigned pm = dec(3 + 6, ENCODE ? (i16mode >> 1) & 1 : 0) << 1;
      pm |= dec(3 + 7, ENCODE ? i16mode & 1 : 0);
      if (!ENCODE) {
        cbpL = l ? 15 : 0;
        cbpC = (int)ch;
        i16mode = (int)pm;
      }
    }
    // ---- intra_chroma_pred_mode (9.3.3.1.1.8)
    {
      const int inc = ((A && A->chroma_mode != 0) ? 1 : 0) + ((B && B->chroma_mode != 0) ? 1 : 0);
      const int want = ENCODE ? d.intra_chroma_pred_mode : 0;
      int v = 0;
      if (dec(64 + inc, want > 0)) {
        v = 1;
        if (dec(64 + 3, want > 1)) {
          v = 2;
          if (dec(64 + 3, want > 2)) v = 3;
        }
      }
      M.chroma_mode = (uint8_t)v;
      if (!ENCODE) d.intra_chroma_pred_mode = (uint8_t)v;
    }
    // ---- coded_block_pattern (9.3.3.1.1.4) for I_NxN
    if (M.i_nxn) {
      int got = 0;
      for (int b8 = 0; b8 < 4; b8++) {
        // neighbouring 8x8 blocks: inside this macroblock or in A / B
        auto bit_of = [&](bool left) -> int {  // condTermFlagN: 1 when the neighbouring 8x8 block is available and NOT coded
          const int bx = b8 & 1, by = b8 >> 1;
          if (left) {
            if (bx == 1) return ((got >> (b8 - 1)) & 1) ? 0 : 1;
            return A ? (((A->cbp_luma >> (b8 + 1)) & 1) ? 0 : 1) : 0;
          }
          if (by == 1) return ((got >> (b8 - 2)) & 1) ? 0 : 1;
          return B ? (((B->cbp_luma >> (b8 + 2)) & 1) ? 0 : 1) : 0;
        };
        const int inc = bit_of(true) + 2 * bit_of(false);
        got |= (int)dec(73 + inc, ENCODE ? (cbpL >> b8) & 1 : 0) << b8;
      }
      const int incA = (A && A->cbp_chroma != 0) ? 1 : 0, incB = (B && B->cbp_chroma != 0) ? 1 : 0;
      int ch = (int)dec(77 + incA + 2 * incB, ENCODE ? cbpC != 0 : 0);
      if (ch) {
        const int a2 = (A && A->cbp_chroma == 2) ? 1 : 0, b2 = (B && B->cbp_chroma == 2) ? 1 : 0;
        ch = 1 + (int)dec(77 + 4 + a2 + 2 * b2, ENCODE ? cbpC == 2 : 0);
      }
      if (!ENCODE) {
        cbpL = got;
        cbpC = ch;
      }
    }
    M.cbp_luma = (uint8_t)cbpL;
    M.cbp_chroma = (uint8_t)cbpC;
    // ---- mb_qp_delta (9.3.3.1.1.5), then QPY (7.4.5, cabac/mod.rs:186-193)
    int qp = qp_prev;
    if (cbpL != 0 || cbpC != 0 || M.i16) {
      int delta = 0;
      if (ENCODE) {
        delta = (int)d.qp - qp_prev;
        if (delta > 25) delta -= 52;
        if (delta < -26) delta += 52;
      }
      const int want = ENCODE ? (delta > 0 ? 2 * delta - 1 : -2 * delta) : 0;
      int v = 0;
      if (dec(60 + (prev_delta_nonzero ? 1 : 0), want > 0)) {
        v = 1;
        if (dec(60 + 2, want > 1)) {
          v = 2;
          while (dec(60 + 3, want > v)) {
            if (++v > 104) fail("mb_qp_delta out of range");
          }
        }
      }
      if (!ENCODE) delta = (v & 1) ? (v + 1) / 2 : -(v / 2);
      prev_delta_nonzero = delta != 0;
      qp = (qp_prev + delta + 52) % 52;
    } else {
      prev_delta_nonzero = false;
    }
    qp_prev = qp;
    if (!ENCODE) {
      d.mb_kind = (uint8_t)kind;
      d.i16_pred_mode = (uint8_t)i16mode;
      d.qp = (uint8_t)qp;
      d.nz_mask = 0xFFFF;
      if (kind == 2) {
        d.prev_flags = 0;
        memset(d.rem_modes, 0, 8);
      }
    }

    // ---- residual(): luma (7.3.5.3.1). coded_block_flag neighbours (9.3.3.1.1.9): an unavailable macroblock counts
    // as coded (the current one is intra), an available block that was not transmitted as not coded.
    auto luma_cbf_nb = [&](int blk, bool left) -> bool {
      int nb;
      bool outside;
      luma_nb(blk, left, nb, outside);
      if (!outside) return (M.cbf_luma >> nb) & 1;
      const MbCtx* N = left ? A : B;
      if (!N) return true;
      return (N->cbf_luma >> nb) & 1;
    };
    if (M.i16) {
      const bool a = A ? (A->cbf_dc & 1) : true, b = B ? (B->cbf_dc & 1) : true;
      if (residual_block(0, co, 0, 15, 16, (a ? 1 : 0) + (b ? 2 : 0))) M.cbf_dc |= 1;
    }
    for (int b8 = 0; b8 < 4; b8++) {
      if (M.t8) {
        int16_t* c = co + 64 * b8;
        if ((cbpL >> b8) & 1) {
          residual_block(5, c, 0, 63, 64, 0);
          M.cbf_luma |= (uint16_t)(0xF << (4 * b8));
        } else if (!ENCODE) memset(c, 0, 128);
        continue;
      }
      for (int k = 0; k < 4; k++) {
        const int blk = 4 * b8 + k;
        int16_t* c = M.i16 ? co + 16 + 15 * blk : co + 16 * blk;
        if ((cbpL >> b8) & 1) {
          const int inc = (luma_cbf_nb(blk, true) ? 1 : 0) + (luma_cbf_nb(blk, false) ? 2 : 0);
          // (Intra16x16 AC: a 15-entry list, levelListIdx 0..14 = positions 1..15 of the block: 7.3.5.3.1)
          const bool cbf = M.i16 ? residual_block(1, c, 0, 14, 15, inc) : residual_block(2, c, 0, 15, 16, inc);
          if (cbf) M.cbf_luma |= (uint16_t)(1u << blk);
        } else if (!ENCODE) {
          memset(c, 0, M.i16 ? 30 : 32);
        }
      }
    }
    // ---- chroma (7.3.5.3, ChromaArrayType 1)
    for (int pl = 0; pl < 2; pl++) {
      int16_t* c = co + 256 + 64 * pl;
      if (cbpC != 0) {
        const bool a = A ? (A->cbf_dc >> (1 + pl)) & 1 : true, b = B ? (B->cbf_dc >> (1 + pl)) & 1 : true;
        if (residual_block(3, c, 0, 3, 4, (a ? 1 : 0) + (b ? 2 : 0))) M.cbf_dc |= (uint8_t)(2 << pl);
      } else if (!ENCODE) memset(c, 0, 8);
    }
    for (int pl = 0; pl < 2; pl++) {
      for (int blk = 0; blk < 4; blk++) {
        int16_t* c = co + 256 + 64 * pl + 4 + 15 * blk;
        if (cbpC == 2) {
          const int bx = blk & 1, by = blk >> 1;
          const bool a = bx ? (M.cbf_ac[pl] >> (blk - 1)) & 1 : (A ? (A->cbf_ac[pl] >> (blk + 1)) & 1 : true);
          const bool b = by ? (M.cbf_ac[pl] >> (blk - 2)) & 1 : (B ? (B->cbf_ac[pl] >> (blk + 2)) & 1 : true);
          if (residual_block(4, c, 0, 14, 15, (a ? 1 : 0) + (b ? 2 : 0))) M.cbf_ac[pl] |= (uint8_t)(1 << blk);
        } else if (!ENCODE) memset(c, 0, 30);
      }
    }
    // ---- end_of_slice_flag
    const unsigned eos = term(ENCODE ? last_mb : 0);
    if (!ENCODE && (eos != 0) != last_mb) fail(eos ? "end_of_slice_flag before the last macroblock" : "no end_of_slice_flag at the last macroblock");
  }
};

// The frame parameters of a picture coded under (s, p). Scaling lists: the SPS's matrix if it has one, else the PPS's,
// else flat 16 (slice/header.rs:317-332 -- the reference's order, see ScalingLists).
inline void set_params(dryv_frame_params& fp, const Sps& s, const Pps& p) {
  const int W = s.width_mbs, H = s.height_map_units;
  if (W < 1 || W > MAX_WIDTH_MBS || H < 1 || H > MAX_HEIGHT_MBS) fail("picture size out of range");
  memset(&fp, 0, sizeof fp);
  fp.pic_width_in_mbs = (uint16_t)W;
  fp.pic_height_in_mbs = (uint16_t)H;
  fp.chroma_array_type = 1;
  fp.bit_depth_y = fp.bit_depth_c = 8;
  fp.chroma_qp_index_offset = (int8_t)p.chroma_qp_offset;
  fp.second_chroma_qp_index_offset = (int8_t)p.second_chroma_qp_offset;
  fp.constrained_intra_pred_flag = p.constrained_intra;
  fp.transform_8x8_mode_flag = p.transform8x8;
  memset(fp.scaling_list4x4, 16, sizeof fp.scaling_list4x4);  // flat (header.rs:330)
  memset(fp.scaling_list8x8, 16, sizeof fp.scaling_list8x8);
  const ScalingLists* L = s.scaling.present ? &s.scaling : p.scaling.present ? &p.scaling : nullptr;
  if (L) {
    memcpy(fp.scaling_list4x4, L->l4, sizeof fp.scaling_list4x4);
    for (int k = 0; k < L->n8 && k < 6; k++) memcpy(fp.scaling_list8x8[k], L->l8[k], 64);
  }
}

// Parses one coded slice NAL unit (header byte included) given the active parameter sets.
// mbs_out / co_out: where the picture's records and coefficients go (W * H records, W * H * 384 coefficients, e.g. a
// slice of a page-locked batch buffer); NULL = into the returned ParsedFrame's own vectors.
inline ParsedFrame parse_islice_nal(const uint8_t* nal, size_t n, const Sps& s, const Pps& p, dryv_mb_desc* mbs_out = nullptr,
                                    int16_t* co_out = nullptr, std::vector<uint8_t>* bin_log = nullptr) {
  if (n < 2) fail("empty NAL unit");
  const int ref_idc = (nal[0] >> 5) & 3, type = nal[0] & 31;
  if (type != 5 && type != 1) fail("not a slice NAL unit");
  if (!p.cabac) fail("unsupported: CAVLC");
  if (s.chroma_format_idc != 1 || s.bit_depth_luma != 8 || s.bit_depth_chroma != 8 || !s.frame_mbs_only)
    fail("unsupported: not 4:2:0 8-bit frame macroblocks");
  const std::vector<uint8_t> rbsp = unescape(nal + 1, n - 1);
  BitReader r{rbsp.data(), rbsp.size(), 0};
  const SliceHeader h = parse_slice_header(r, s, p, type, ref_idc);
  while (!r.aligned())
    if (r.bit() != 1) fail("cabac_alignment_one_bit is 0");  // cabac/mod.rs:71-73
  ParsedFrame F;
  const int W = s.width_mbs, H = s.height_map_units;
  set_params(F.fp, s, p);
  if (!mbs_out || !co_out) {
    F.mbs.resize((size_t)W * H);
    F.coeffs.assign((size_t)W * H * 384, 0);
    mbs_out = F.mbs.data();
    co_out = F.coeffs.data();
  } else {
    memset(co_out, 0, (size_t)W * H * 768);
  }
  F.slice_qp = h.slice_qp;
  for (int k = 0; k < 4; k++) F.crop[k] = 2 * s.crop[k];  // CropUnitX = CropUnitY = 2 for 4:2:0 frame pictures (7.4.2.1.1)
  F.deblock.disable_deblocking_filter_idc = (uint8_t)h.disable_deblocking_filter_idc;
  F.deblock.slice_alpha_c0_offset_div2 = (int8_t)h.alpha_c0_offset_div2;
  F.deblock.slice_beta_offset_div2 = (int8_t)h.beta_offset_div2;
  CabacDecoder cd;
  cd.log = bin_log;
  cd.start(&r, h.slice_qp);
  MbLayer<false, CabacDecoder> L(cd, W, H, p.transform8x8, h.slice_qp);
  for (int a = 0; a < W * H; a++) {
    L.macroblock(a, mbs_out[a], co_out + (size_t)a * 384, a == W * H - 1);
    F.kinds[mbs_out[a].mb_kind]++;
  }
  F.bins = cd.bins;
  F.slice_bytes = n;
  // The arithmetic decoder has read 9 bits at initialisation and one per renormalisation shift; at the terminating bin
  // the standard's 9.3.1.2 does a last byte alignment read of rbsp_stop_one_bit: what is left must be < 2 bytes
  F.bits_unread = r.bits_left();
  // (the engine reads 9 bits ahead and does not renormalise on the terminating bin; the encoder's flush writes 7 + 1 + 2
  // bits, the last of them the rbsp_stop_one_bit: 9.3.1.2, 9.3.4.5 -- so the read position is right behind that bit)
  bool ok = r.pos > 0 && ((rbsp[(r.pos - 1) >> 3] >> (7 - ((r.pos - 1) & 7))) & 1) == 1;
  while (ok && r.bits_left() > 0) ok = r.bit() == 0;
  F.tail_ok = ok;
  return F;
}

// ---- containers ------------------------------------------------------------------------------------------------------------
// Parameter sets are kept by id, every version of them (a stream may send a set again with other content: 7.4.1.2.1);
// a coded slice is paired, when it is seen, with the sets that are active for it at that point of the stream.
struct Stream {
  std::vector<Sps> spsAll;   // append-only
  std::vector<Pps> ppsAll;
  int spsById[32], ppsById[256];  // index of the latest version, -1 = none
  struct Nal { const uint8_t* p; size_t n; int sps, pps; };
  std::vector<Nal> slices;  // coded slice NAL units (with header byte) inside the caller's buffer, in decoding order
  Stream() {
    for (int& v : spsById) v = -1;
    for (int& v : ppsById) v = -1;
  }
  // (a coded slice whose parameter sets were not sent is recorded with -1 and refused only when it is asked for: a stream whose
  // inter slices use a picture parameter set that arrives late, or not at all, still yields its intra pictures)
  const Sps& sps_of(const Nal& nal) const {
    if (nal.pps < 0) fail("coded slice refers to a picture parameter set that was not sent");
    if (nal.sps < 0) fail("picture parameter set refers to a sequence parameter set that was not sent");
    return spsAll[(size_t)nal.sps];
  }
  const Pps& pps_of(const Nal& nal) const {
    if (nal.pps < 0) fail("coded slice refers to a picture parameter set that was not sent");
    return ppsAll[(size_t)nal.pps];
  }
};

inline void take_nal(Stream& S, const uint8_t* p, size_t n) {
  if (n < 1) return;
  const int type = p[0] & 31;
  if (type == 7) {
    S.spsAll.push_back(parse_sps(unescape(p + 1, n - 1)));
    S.spsById[S.spsAll.back().id] = (int)S.spsAll.size() - 1;
  } else if (type == 8) {
    // (the PPS's matrix size depends on the chroma format of the SPS it names; an unknown one counts as 4:2:0 and the
    // slice is refused later for want of its SPS)
    const std::vector<uint8_t> rbsp = unescape(p + 1, n - 1);
    BitReader peek{rbsp.data(), rbsp.size(), 0};
    peek.ue();
    const unsigned sid = peek.ue();
    const int si = sid < 32 ? S.spsById[sid] : -1;
    S.ppsAll.push_back(parse_pps(rbsp, si >= 0 ? S.spsAll[(size_t)si].chroma_format_idc : 1));
    S.ppsById[S.ppsAll.back().id] = (int)S.ppsAll.size() - 1;
  } else if (type == 5 || type == 1) {
    // first_mb_in_slice, slice_type, pic_parameter_set_id (7.3.3): the first three ue(v) of the header
    const std::vector<uint8_t> rbsp = unescape(p + 1, std::min<size_t>(n - 1, 24));
    BitReader r{rbsp.data(), rbsp.size(), 0};
    r.ue();
    r.ue();
    const unsigned pid = r.ue();
    const int pi = pid < 256 ? S.ppsById[pid] : -1;
    const int si = pi < 0 ? -1 : S.spsById[S.ppsAll[(size_t)pi].sps_id];
    S.slices.push_back(Stream::Nal{p, n, si, pi});   // (pi / si < 0: refused when the slice is asked for, Stream::sps_of)
  }
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
inline uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }

// Finds child atom `name` inside [p, p+n); returns its payload.
inline bool find_atom(const uint8_t* p, size_t n, const char* name, const uint8_t*& out, size_t& outn, size_t skip = 0) {
  size_t off = skip;
  while (off + 8 <= n) {
    uint64_t sz = be32(p + off);
    size_t hdr = 8;
    if (sz == 1 && off + 16 <= n) {
      sz = be64(p + off + 8);
      hdr = 16;
    } else if (sz == 0) sz = n - off;
    if (sz < hdr || sz > n - off) return false;   // (off <= n - 8 here: no sum that could wrap)
    if (memcmp(p + off + 4, name, 4) == 0) {
      out = p + off + hdr;
      outn = (size_t)sz - hdr;
      return true;
    }
    off += (size_t)sz;
  }
  return false;
}

// ISO-BMFF: the first video track's avcC (SPS / PPS, NAL length size) and its first `max_samples` samples (the reference
// decodes only sample 0: video/decoder.rs:88), split into length-prefixed NAL units (sample/nal.rs:232-253). Sample
// positions: chunk offsets (stco / co64) + sample-to-chunk runs (stsc) + sample sizes (stsz).
inline Stream demux_mp4(const uint8_t* f, size_t n, size_t max_samples) {
  const uint8_t *moov, *trak, *mdia, *minf, *stbl, *stsd, *q;
  size_t nmoov, ntrak, nmdia, nminf, nstbl, nstsd, nq;
  if (!find_atom(f, n, "moov", moov, nmoov)) fail("mp4: no moov atom");
  // walk the traks until one has an avc1 sample entry
  size_t off = 0;
  for (;;) {
    if (!find_atom(moov + off, nmoov - off, "trak", trak, ntrak)) fail("mp4: no AVC video track");
    off = (size_t)(trak - moov) + ntrak;
    if (find_atom(trak, ntrak, "mdia", mdia, nmdia) && find_atom(mdia, nmdia, "minf", minf, nminf) &&
        find_atom(minf, nminf, "stbl", stbl, nstbl) && find_atom(stbl, nstbl, "stsd", stsd, nstsd) && nstsd > 16 &&
        memcmp(stsd + 12, "avc1", 4) == 0)
      break;
  }
  // stsd: version/flags(4) count(4) | entry: size(4) 'avc1' + 78 bytes of VisualSampleEntry, then child atoms
  const uint8_t* entry = stsd + 8;
  const size_t nentry = be32(entry);
  if (nentry > nstsd - 8 || nentry < 86) fail("mp4: bad avc1 sample entry");
  if (!find_atom(entry + 86, nentry - 86, "avcC", q, nq) || nq < 7) fail("mp4: no avcC");
  Stream S;
  const int lenSize = (q[4] & 3) + 1;
  size_t o = 5;
  const int nsps = q[o++] & 31;
  auto take_sets = [&](int count) {
    for (int k = 0; k < count; k++) {
      if (nq - o < 2) fail("mp4: avcC truncated");
      const size_t l = ((size_t)q[o] << 8) | q[o + 1];
      o += 2;
      if (l > nq - o) fail("mp4: avcC truncated");
      take_nal(S, q + o, l);
      o += l;
    }
  };
  take_sets(nsps);
  if (o >= nq) fail("mp4: avcC truncated");
  const int npps = q[o++];
  take_sets(npps);
  const uint8_t *stsz, *stco, *stsc;
  size_t nstsz, nstco, nstsc;
  if (!find_atom(stbl, nstbl, "stsz", stsz, nstsz) || nstsz < 12) fail("mp4: no stsz");
  const uint32_t fixedSize = be32(stsz + 4), nSamples = be32(stsz + 8);
  if (fixedSize == 0 && nstsz < 12 + 4 * (size_t)nSamples) fail("mp4: stsz truncated");
  bool co64 = false;
  if (find_atom(stbl, nstbl, "stco", stco, nstco) && nstco >= 8) co64 = false;
  else if (find_atom(stbl, nstbl, "co64", stco, nstco) && nstco >= 8) co64 = true;
  else fail("mp4: no chunk offsets");
  const uint32_t nChunks = be32(stco + 4);
  if (nstco < 8 + (size_t)nChunks * (co64 ? 8 : 4)) fail("mp4: chunk offset table truncated");
  // stsc: runs of chunks with the same number of samples; absent or empty = one sample per chunk
  uint32_t nRuns = 0;
  if (find_atom(stbl, nstbl, "stsc", stsc, nstsc) && nstsc >= 8) {
    nRuns = be32(stsc + 4);
    if (nstsc < 8 + 12 * (size_t)nRuns) fail("mp4: stsc truncated");
  }
  size_t sample = 0;
  uint32_t run = 0;
  for (uint32_t c = 0; c < nChunks && sample < nSamples && sample < max_samples; c++) {
    while (run + 1 < nRuns && be32(stsc + 8 + 12 * (run + 1)) <= c + 1) run++;
    const uint32_t perChunk = nRuns ? be32(stsc + 8 + 12 * run + 4) : 1;
    uint64_t off = co64 ? be64(stco + 8 + 8 * (size_t)c) : be32(stco + 8 + 4 * (size_t)c);
    for (uint32_t k = 0; k < perChunk && sample < nSamples && sample < max_samples; k++, sample++) {
      const uint32_t size = fixedSize ? fixedSize : be32(stsz + 12 + 4 * sample);
      if (size > n || off > n - size) fail("mp4: sample outside the file");   // (no sum that could wrap: off is 64 bits from co64)
      const uint8_t* sp = f + off;
      size_t so = 0;
      while (size - so >= (size_t)lenSize) {
        size_t l = 0;
        for (int q2 = 0; q2 < lenSize; q2++) l = (l << 8) | sp[so + q2];
        so += lenSize;
        if (l > size - so) fail("mp4: NAL unit outside the sample");
        take_nal(S, sp + so, l);
        so += l;
      }
      off += size;
    }
  }
  return S;
}

inline Stream demux_annexb(const uint8_t* f, size_t n) {
  Stream S;
  size_t start = (size_t)-1;
  auto flush = [&](size_t end) {
    if (start == (size_t)-1) return;
    size_t e = end;
    while (e > start && f[e - 1] == 0) e--;  // trailing_zero_8bits
    take_nal(S, f + start, e - start);
  };
  // start codes 00 00 01: look for the 01 bytes (memchr) and check the two bytes in front
  size_t i = 2;
  while (i < n) {
    const uint8_t* q = (const uint8_t*)memchr(f + i, 1, n - i);
    if (!q) break;
    i = (size_t)(q - f);
    if (f[i - 1] == 0 && f[i - 2] == 0) {
      flush(i - 2);
      start = i + 1;
    }
    i++;
  }
  flush(n);
  return S;
}

inline ParsedFrame parse_first_islice(const uint8_t* f, size_t n) {
  const bool mp4 = n >= 12 && memcmp(f + 4, "ftyp", 4) == 0;
  Stream S = mp4 ? demux_mp4(f, n, 1) : demux_annexb(f, n);
  if (S.slices.empty()) fail(S.spsAll.empty() || S.ppsAll.empty() ? "no SPS / PPS" : "no coded slice");
  return parse_islice_nal(S.slices[0].p, S.slices[0].n, S.sps_of(S.slices[0]), S.pps_of(S.slices[0]));
}

// Whether a coded slice NAL unit starts a picture that consists of one I slice (what this parser decodes).
inline bool is_whole_picture_islice(const uint8_t* nal, size_t n) {
  if (n < 2) return false;
  const std::vector<uint8_t> rbsp = unescape(nal + 1, std::min<size_t>(n - 1, 16));
  BitReader r{rbsp.data(), rbsp.size(), 0};
  try {
    const unsigned first_mb = r.ue(), type = r.ue();
    return first_mb == 0 && type % 5 == 2;
  } catch (const Error&) {
    return false;
  }
}

// Every picture of the stream that is a single I slice, up to max_pictures (the reference stops after sample 0:
// video/decoder.rs:88, quirk Q9 -- a batch of pictures is this build's own unit of work), all under the stream's first SPS /
// PPS. Inter pictures in between are skipped; their count is returned in *skipped.
// n_threads: intra pictures are independent of each other (CABAC contexts and the QP predictor restart with every slice:
// cabac/mod.rs:72-87, slice/mod.rs:153), so their macroblock layers are parsed in parallel; 0 = one thread per hardware
// thread (at most one per picture).
// mbs_out / co_out (optional): one batch buffer for all pictures, `capacity` pictures large; the returned ParsedFrames
// then carry only parameters and diagnostics.
inline std::vector<ParsedFrame> parse_all_islices(const uint8_t* f, size_t n, size_t max_pictures, size_t* skipped = nullptr,
                                                  unsigned n_threads = 1, dryv_mb_desc* mbs_out = nullptr,
                                                  int16_t* co_out = nullptr, size_t capacity = 0) {
  const bool mp4 = n >= 12 && memcmp(f + 4, "ftyp", 4) == 0;
  Stream S = mp4 ? demux_mp4(f, n, (size_t)-1) : demux_annexb(f, n);
  if (S.spsAll.empty() || S.ppsAll.empty()) fail("no SPS / PPS");
  std::vector<Stream::Nal> todo;
  size_t skip = 0;
  for (const Stream::Nal& nal : S.slices) {
    if (todo.size() >= max_pictures) break;
    if (!is_whole_picture_islice(nal.p, nal.n)) {
      skip++;
      continue;
    }
    todo.push_back(nal);
  }
  if (skipped) *skipped = skip;
  if (todo.empty()) fail("no intra picture");
  // one batch buffer = pictures of one size: every picture must have the first one's (bounded by parse_sps, so none of
  // the products below can overflow), and the caller's capacity is in pictures of that size
  const size_t perPic = (size_t)S.sps_of(todo[0]).width_mbs * (size_t)S.sps_of(todo[0]).height_map_units;
  for (const Stream::Nal& nal : todo)
    if ((size_t)S.sps_of(nal).width_mbs * (size_t)S.sps_of(nal).height_map_units != perPic ||
        S.sps_of(nal).width_mbs != S.sps_of(todo[0]).width_mbs)
      fail("pictures of different sizes in one stream");
  if (mbs_out && co_out && todo.size() > capacity) fail("batch buffer too small");
  auto parse_one = [&](size_t k) {
    return parse_islice_nal(todo[k].p, todo[k].n, S.sps_of(todo[k]), S.pps_of(todo[k]), mbs_out && co_out ? mbs_out + k * perPic : nullptr,
                            mbs_out && co_out ? co_out + k * perPic * 384 : nullptr);
  };
  std::vector<ParsedFrame> out(todo.size());
  if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
  n_threads = (unsigned)std::min<size_t>(n_threads, todo.size());
  if (n_threads <= 1) {
    for (size_t k = 0; k < todo.size(); k++) out[k] = parse_one(k);
    return out;
  }
  std::atomic<size_t> next{0};
  std::vector<std::string> errs(n_threads);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_threads; t++)
    pool.emplace_back([&, t]() {
      try {
        for (size_t k = next++; k < todo.size(); k = next++) out[k] = parse_one(k);
      } catch (const Error& e) {
        errs[t] = e.what.empty() ? "parse error" : e.what;
        next = todo.size();
      } catch (const std::exception& e) {  // (bad_alloc and friends must not leave a std::thread)
        errs[t] = std::string("parse: ") + e.what();
        next = todo.size();
      } catch (...) {
        errs[t] = "parse: unknown exception";
        next = todo.size();
      }
    });
  for (std::thread& th : pool) th.join();
  for (const std::string& e : errs)
    if (!e.empty()) fail(e.c_str());
  return out;
}

// Test hook: the CABAC bins of the stream's `picture`-th intra picture (see CabacDecoder::log) and its parse.
inline ParsedFrame parse_islice_with_bins(const uint8_t* f, size_t n, size_t picture, std::vector<uint8_t>& bins) {
  const bool mp4 = n >= 12 && memcmp(f + 4, "ftyp", 4) == 0;
  Stream S = mp4 ? demux_mp4(f, n, (size_t)-1) : demux_annexb(f, n);
  size_t k = 0;
  for (const Stream::Nal& nal : S.slices) {
    if (!is_whole_picture_islice(nal.p, nal.n)) continue;
    if (k++ == picture) return parse_islice_nal(nal.p, nal.n, S.sps_of(nal), S.pps_of(nal), nullptr, nullptr, &bins);
  }
  fail("no such intra picture");
}

// ---- encoder: one IDR picture as an Annex-B byte stream (SPS, PPS, one I slice) ------------------------------------------
// crop: optional frame cropping rectangle in luma samples (left, right, top, bottom; even), written to the SPS
// n_pictures > 1: an all-intra stream, one IDR picture after the other (mbs / coeffs hold them back to back)
inline std::vector<uint8_t> encode_idr_annexb(const dryv_frame_params& fp, const dryv_mb_desc* mbs, const int16_t* coeffs,
                                              int slice_qp = 26, const int* crop = nullptr, int n_pictures = 1) {
  const int W = fp.pic_width_in_mbs, H = fp.pic_height_in_mbs;
  if (W < 1 || W > MAX_WIDTH_MBS || H < 1 || n_pictures < 1) fail("encoder: picture size out of range");
  // Non-flat lists travel as a sequence scaling matrix with every list sent explicitly (a list left out would come back
  // as the Default_* table: ScalingLists above). A 4:2:0 stream carries two 8x8 lists (Intra Y, Inter Y).
  bool flat = true;
  for (int l = 0; l < 6; l++) {
    for (int k = 0; k < 16; k++) flat = flat && fp.scaling_list4x4[l][k] == 16;
    for (int k = 0; k < 64; k++) {
      flat = flat && fp.scaling_list8x8[l][k] == 16;
      if (l >= 2 && fp.scaling_list8x8[l][k] != 16) fail("encoder: a 4:2:0 stream carries only the two luma 8x8 scaling lists");
    }
  }
  std::vector<uint8_t> out;
  {
    BitWriter w;  // SPS, High profile (7.3.2.1.1)
    w.bits(100, 8);
    w.bits(0, 8);
    w.bits(51, 8);
    w.ue(0);
    w.ue(1);  // chroma_format_idc
    w.ue(0);
    w.ue(0);
    w.bit(0);
    w.bit(flat ? 0 : 1);  // seq_scaling_matrix_present_flag
    if (!flat)
      for (int l = 0; l < 8; l++) {
        w.bit(1);  // seq_scaling_list_present_flag
        if (l < 6) write_scaling_list(w, fp.scaling_list4x4[l], 16);
        else write_scaling_list(w, fp.scaling_list8x8[l - 6], 64);
      }
    w.ue(0);   // log2_max_frame_num_minus4
    w.ue(2);   // pic_order_cnt_type
    w.ue(1);   // max_num_ref_frames
    w.bit(0);
    w.ue((unsigned)W - 1);
    w.ue((unsigned)H - 1);
    w.bit(1);  // frame_mbs_only_flag
    w.bit(1);  // direct_8x8_inference_flag
    const bool cropped = crop && (crop[0] | crop[1] | crop[2] | crop[3]) != 0;
    w.bit(cropped ? 1 : 0);  // frame_cropping_flag
    if (cropped)
      for (int k = 0; k < 4; k++) {
        if (crop[k] < 0 || (crop[k] & 1)) fail("encoder: crop offsets must be even and non-negative");
        w.ue((unsigned)crop[k] / 2);
      }
    w.bit(0);  // vui_parameters_present_flag
    w.trailing();
    append_nal_annexb(out, 0x67, w.out);
  }
  {
    BitWriter w;  // PPS (7.3.2.2)
    w.ue(0);
    w.ue(0);
    w.bit(1);  // entropy_coding_mode_flag
    w.bit(0);
    w.ue(0);
    w.ue(0);
    w.ue(0);
    w.bit(0);
    w.bits(0, 2);
    w.se(0);  // pic_init_qp_minus26
    w.se(0);
    w.se(fp.chroma_qp_index_offset);
    w.bit(0);  // deblocking_filter_control_present_flag
    w.bit(fp.constrained_intra_pred_flag ? 1 : 0);
    w.bit(0);
    w.bit(fp.transform_8x8_mode_flag ? 1 : 0);
    w.bit(0);  // pic_scaling_matrix_present_flag
    w.se(fp.second_chroma_qp_index_offset);
    w.trailing();
    append_nal_annexb(out, 0x68, w.out);
  }
  std::vector<std::vector<uint8_t>> slices((size_t)n_pictures);
  auto encode_picture = [&](int pic) {
    const dryv_mb_desc* pm = mbs + (size_t)pic * W * H;
    const int16_t* pc = coeffs + (size_t)pic * W * H * 384;
    BitWriter w;  // slice header (7.3.3), IDR, I slice
    w.ue(0);
    w.ue(7);
    w.ue(0);
    w.bits(0, 4);  // frame_num
    w.ue((unsigned)(pic & 1));  // idr_pic_id: consecutive IDR pictures must differ (7.4.3)
    w.bit(0);
    w.bit(0);  // dec_ref_pic_marking: no_output_of_prior_pics_flag, long_term_reference_flag
    w.se(slice_qp - 26);
    while (w.nbits & 7) w.bit(1);  // cabac_alignment_one_bit
    CabacEncoder ce;
    ce.start(&w, slice_qp);
    MbLayer<true, CabacEncoder> L(ce, W, H, fp.transform_8x8_mode_flag != 0, slice_qp);
    std::vector<int16_t> co(384);
    for (int a = 0; a < W * H; a++) {
      dryv_mb_desc d = pm[a];
      memcpy(co.data(), pc + (size_t)a * 384, 768);
      L.macroblock(a, d, co.data(), a == W * H - 1);
    }
    while (w.nbits & 7) w.bit(0);
    append_nal_annexb(slices[(size_t)pic], 0x65, w.out);
  };
  // pictures are independent: encode them on all hardware threads, then append in order
  const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), (size_t)n_pictures);
  if (nt <= 1) {
    for (int pic = 0; pic < n_pictures; pic++) encode_picture(pic);
  } else {
    std::atomic<int> next{0};
    std::vector<std::string> errs(nt);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++)
      pool.emplace_back([&, t]() {
        try {
          for (int pic = next++; pic < n_pictures; pic = next++) encode_picture(pic);
        } catch (const Error& e) {
          errs[t] = e.what.empty() ? "encode error" : e.what;
          next = n_pictures;
        } catch (const std::exception& e) {
          errs[t] = std::string("encode: ") + e.what();
          next = n_pictures;
        } catch (...) {
          errs[t] = "encode: unknown exception";
          next = n_pictures;
        }
      });
    for (std::thread& th : pool) th.join();
    for (const std::string& e : errs)
      if (!e.empty()) fail(e.c_str());
  }
  for (const std::vector<uint8_t>& sl : slices) out.insert(out.end(), sl.begin(), sl.end());
  return out;
}

}  // namespace h264
}  // namespace dryv
