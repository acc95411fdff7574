// frame_harness.cpp — exercises dryv::Frame (host/frame.hpp) the way dryv's decoder would: one
// Frame per picture, Frame::decode per macroblock in mbaddr order, write_to_yuv_file at the end.
//
//   frame_harness <in.batch> <out.yuv>
// in.batch: dryv_frame_params (496 B) | u32 n_frames | dryv_mb_desc[n] | int16 coeffs[n][384]
// (written by tests/test_host_harness.py from the synthetic generator). Frames are written back to back.
//
//   frame_harness decode-all <in.mp4 | in.h264> <out.yuv>      every intra picture of the stream as one batch
//   frame_harness decode <in.mp4 | in.h264> <out.yuv>
// BASELINE.json configs[0], end to end: what `dryv <path>` does for its one decoded picture (video/decoder.rs:88-143:
// sample 0 of the video track -> slice NAL -> CABAC macroblock loop -> Frame::decode per macroblock ->
// write_to_yuv_file("temp/yuv_frame")), with the entropy decoding on the host (h264_islice.hpp) and the reconstruction
// behind the C ABI. Prints one line of parse statistics.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "frame.hpp"
#include "h264_islice.hpp"

static dryv::Macroblock unpack(const dryv_mb_desc& d, const int16_t* c) {
  dryv::Macroblock mb;
  mb.mode = d.mb_kind;
  mb.intra16x16_pred_mode = d.i16_pred_mode;
  mb.intra_chroma_pred_mode = d.intra_chroma_pred_mode;
  mb.qpy = d.qp;
  for (int i = 0; i < 16; i++) {
    const int prev = (d.prev_flags >> i) & 1, rem = (d.rem_modes[i >> 1] >> (4 * (i & 1))) & 7;
    mb.prev_intra4x4_pred_mode_flag[i] = (uint8_t)prev;
    mb.rem_intra4x4_pred_mode[i] = (uint8_t)rem;
    if (i < 4) {
      mb.prev_intra8x8_pred_mode_flag[i] = (uint8_t)prev;
      mb.rem_intra8x8_pred_mode[i] = (uint8_t)rem;
    }
  }
  if (d.mb_kind == 0) {
    for (int b = 0; b < 16; b++) for (int k = 0; k < 16; k++) mb.block_luma_4x4[b][k] = *c++;
  } else if (d.mb_kind == 1) {
    for (int b = 0; b < 4; b++) for (int k = 0; k < 64; k++) mb.block_luma_8x8[b][k] = *c++;
  } else {
    for (int k = 0; k < 16; k++) mb.block_luma_dc[k] = *c++;
    for (int b = 0; b < 16; b++) for (int k = 0; k < 15; k++) mb.block_luma_ac[b][k] = *c++;
  }
  for (int pl = 0; pl < 2; pl++) {
    for (int k = 0; k < 4; k++) mb.block_chroma_dc[pl][k] = *c++;
    for (int b = 0; b < 4; b++) for (int k = 0; k < 15; k++) mb.block_chroma_ac[pl][b][k] = *c++;
  }
  return mb;
}

static int decode_file(const char* in, const char* outp) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::perror("open"); return 2; }
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
  std::fclose(f);
  dryv::h264::ParsedFrame P;
  try {
    P = dryv::h264::parse_first_islice(data.data(), data.size());
  } catch (const dryv::h264::Error& e) {
    std::fprintf(stderr, "parse: %s\n", e.what.c_str());
    return 6;
  }
  const size_t per = (size_t)P.fp.pic_width_in_mbs * P.fp.pic_height_in_mbs;
  std::printf("parsed %ux%u macroblocks: %d Intra4x4 %d Intra8x8 %d Intra16x16, slice qp %d, %ld bins, %zu bytes, tail %s\n",
              P.fp.pic_width_in_mbs, P.fp.pic_height_in_mbs, P.kinds[0], P.kinds[1], P.kinds[2], P.slice_qp, P.bins,
              P.slice_bytes, P.tail_ok ? "ok" : "BAD");
  if (!P.tail_ok) return 6;
  dryv_recon_ctx* ctx = nullptr;
  int st = dryv_recon_create(&ctx, 0);
  if (st != DRYV_OK) { std::fprintf(stderr, "dryv_recon_create: %s\n", dryv_recon_strerror(st)); return 3; }
  dryv::Frame frame(P.fp, ctx);                                       // Frame::new(&slice)      video/decoder.rs:124
  for (size_t a = 0; a < per; a++) {                                  // cabac/mod.rs:208
    st = frame.decode(unpack(P.mbs[a], &P.coeffs[a * DRYV_COEFFS_PER_MB]));
    if (st != DRYV_OK) { std::fprintf(stderr, "decode: %s\n", dryv_recon_strerror(st)); return 4; }
  }
  st = frame.write_to_yuv_file(outp);                                 // video/decoder.rs:142
  if (st != DRYV_OK) { std::fprintf(stderr, "reconstruct: %s\n", dryv_recon_strerror(st)); return 5; }
  dryv_recon_destroy(ctx);
  return 0;
}

// frame_harness decode-all <in.mp4 | in.h264> <out.yuv>: every intra picture of the stream (h264_islice.hpp:
// parse_all_islices) as ONE batch through dryv_recon_submit / dryv_recon_wait -- what INTEGRATION.md recommends for a
// dryv that decodes more than sample 0. Pictures are written back to back in write_to_yuv_file order.
static int decode_all(const char* in, const char* outp) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::perror("open"); return 2; }
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
  std::fclose(f);
  std::vector<dryv::h264::ParsedFrame> P;
  size_t skipped = 0;
  try {
    P = dryv::h264::parse_all_islices(data.data(), data.size(), (size_t)-1, &skipped);
  } catch (const dryv::h264::Error& e) {
    std::fprintf(stderr, "parse: %s\n", e.what.c_str());
    return 6;
  }
  const dryv_frame_params fp = P[0].fp;
  const size_t per = (size_t)fp.pic_width_in_mbs * fp.pic_height_in_mbs;
  std::vector<dryv_mb_desc> mbs;
  std::vector<int16_t> co;
  bool tails = true;
  for (const auto& F : P) {
    if (std::memcmp(&F.fp, &fp, sizeof fp) != 0) { std::fprintf(stderr, "parse: parameters change inside the stream\n"); return 6; }
    mbs.insert(mbs.end(), F.mbs.begin(), F.mbs.end());
    co.insert(co.end(), F.coeffs.begin(), F.coeffs.end());
    tails = tails && F.tail_ok;
  }
  std::printf("parsed %zu intra pictures of %ux%u macroblocks (%zu other coded slices skipped), tails %s\n", P.size(),
              fp.pic_width_in_mbs, fp.pic_height_in_mbs, skipped, tails ? "ok" : "BAD");
  if (!tails) return 6;
  dryv_recon_ctx* ctx = nullptr;
  int st = dryv_recon_create(&ctx, 0);
  if (st != DRYV_OK) { std::fprintf(stderr, "dryv_recon_create: %s\n", dryv_recon_strerror(st)); return 3; }
  std::vector<uint8_t> yuv(P.size() * per * 384);
  st = dryv_recon_submit(ctx, &fp, (uint32_t)P.size(), mbs.data(), co.data());
  if (st == DRYV_OK) st = dryv_recon_wait(ctx, yuv.data(), yuv.size());
  if (st != DRYV_OK) { std::fprintf(stderr, "reconstruct: %s\n", dryv_recon_strerror(st)); return 5; }
  FILE* out = std::fopen(outp, "wb");
  if (!out) return 2;
  std::fwrite(yuv.data(), 1, yuv.size(), out);
  std::fclose(out);
  dryv_recon_destroy(ctx);
  return 0;
}

// frame_harness decode-deblocked <in.mp4 | in.h264> <out.yuv>: the first intra picture, reconstructed AND deblocked with the
// slice header's own deblocking syntax elements (h264_islice.hpp: ParsedFrame::deblock): what a conformant decoder outputs
// for that picture -- beyond dryv, which parses those elements and does not filter (README.md:15).
static int decode_deblocked(const char* in, const char* outp) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::perror("open"); return 2; }
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
  std::fclose(f);
  dryv::h264::ParsedFrame P;
  try {
    P = dryv::h264::parse_first_islice(data.data(), data.size());
  } catch (const dryv::h264::Error& e) {
    std::fprintf(stderr, "parse: %s\n", e.what.c_str());
    return 6;
  }
  if (!P.tail_ok) return 6;
  std::printf("parsed %ux%u macroblocks; deblocking: disable_idc %d, alpha offset %d, beta offset %d\n", P.fp.pic_width_in_mbs,
              P.fp.pic_height_in_mbs, P.deblock.disable_deblocking_filter_idc, 2 * P.deblock.slice_alpha_c0_offset_div2,
              2 * P.deblock.slice_beta_offset_div2);
  dryv_recon_ctx* ctx = nullptr;
  int st = dryv_recon_create(&ctx, 0);
  if (st != DRYV_OK) { std::fprintf(stderr, "dryv_recon_create: %s\n", dryv_recon_strerror(st)); return 3; }
  std::vector<uint8_t> yuv(dryv_recon_frame_bytes(&P.fp));
  st = dryv_recon_submit(ctx, &P.fp, 1, P.mbs.data(), P.coeffs.data());
  if (st == DRYV_OK) st = dryv_recon_wait_filtered(ctx, &P.deblock, nullptr, yuv.data(), yuv.size());
  if (st != DRYV_OK) { std::fprintf(stderr, "reconstruct / deblock: %s\n", dryv_recon_strerror(st)); return 5; }
  FILE* out = std::fopen(outp, "wb");
  if (!out) return 2;
  std::fwrite(yuv.data(), 1, yuv.size(), out);
  std::fclose(out);
  dryv_recon_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::strcmp(argv[1], "decode") == 0) return decode_file(argv[2], argv[3]);
  if (argc == 4 && std::strcmp(argv[1], "decode-deblocked") == 0) return decode_deblocked(argv[2], argv[3]);
  if (argc == 4 && std::strcmp(argv[1], "decode-all") == 0) return decode_all(argv[2], argv[3]);
  if (argc != 3) { std::fprintf(stderr, "usage: %s in.batch out.yuv\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  dryv_frame_params fp;
  uint32_t n_frames = 0;
  if (std::fread(&fp, sizeof(fp), 1, f) != 1 || std::fread(&n_frames, 4, 1, f) != 1) return 2;
  const size_t per = (size_t)fp.pic_width_in_mbs * fp.pic_height_in_mbs, n = per * n_frames;
  std::vector<dryv_mb_desc> mbs(n);
  std::vector<int16_t> co(n * DRYV_COEFFS_PER_MB);
  if (std::fread(mbs.data(), sizeof(dryv_mb_desc), n, f) != n) return 2;
  if (std::fread(co.data(), 2, co.size(), f) != co.size()) return 2;
  std::fclose(f);

  dryv_recon_ctx* ctx = nullptr;
  int st = dryv_recon_create(&ctx, 0);
  if (st != DRYV_OK) { std::fprintf(stderr, "dryv_recon_create: %s\n", dryv_recon_strerror(st)); return 3; }
  FILE* out = std::fopen(argv[2], "wb");
  if (!out) return 2;
  for (uint32_t fi = 0; fi < n_frames; fi++) {
    dryv::Frame frame(fp, ctx);                                     // Frame::new(&slice)      video/decoder.rs:124
    for (size_t a = 0; a < per; a++) {                              // slice.data(): the CABAC MB loop
      st = frame.decode(unpack(mbs[fi * per + a], &co[(fi * per + a) * DRYV_COEFFS_PER_MB]));  // cabac/mod.rs:208
      if (st != DRYV_OK) { std::fprintf(stderr, "decode: %s\n", dryv_recon_strerror(st)); return 4; }
    }
    const uint8_t* p;
    size_t bytes;
    st = frame.planes(&p, &bytes);                                  // write_to_yuv_file       video/decoder.rs:142
    if (st != DRYV_OK) { std::fprintf(stderr, "reconstruct: %s\n", dryv_recon_strerror(st)); return 5; }
    std::fwrite(p, 1, bytes, out);
  }
  std::fclose(out);
  dryv_recon_destroy(ctx);
  return 0;
}
