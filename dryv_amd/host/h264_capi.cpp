// h264_capi.cpp — C ABI over h264_islice.hpp (libdryv_h264.so, host only): the host producer of SURVEY.md 8(f)-1.
#include "h264_islice.hpp"

#include <stdlib.h>

#include <memory>
#include <new>

using namespace dryv::h264;

// Nothing may leave an extern "C" function by exception: the parser's own Error, and whatever the standard library
// throws underneath it (bad_alloc from a vector sized by a stream's parameters, length_error ...), all end as a message.
#define DRYV_CATCH_ALL(ret)                                    \
  catch (const Error& e) {                                     \
    g_err = e.what;                                            \
    return ret;                                                \
  }                                                            \
  catch (const std::exception& e) {                            \
    g_err = std::string("host parser: ") + e.what();           \
    return ret;                                                \
  }                                                            \
  catch (...) {                                                \
    g_err = "host parser: unknown exception";                  \
    return ret;                                                \
  }

extern "C" {

struct dryv_h264_frame {
  ParsedFrame F;
  std::string err;
};

/* Parses the first coded picture (an I slice) of an .mp4 / Annex-B buffer. Returns NULL on failure (message via
 * dryv_h264_last_error). */
static thread_local std::string g_err;
dryv_h264_frame* dryv_h264_parse(const uint8_t* data, size_t n) {
  try {
    std::unique_ptr<dryv_h264_frame> h(new dryv_h264_frame());
    h->F = parse_first_islice(data, n);
    return h.release();
  }
  DRYV_CATCH_ALL(nullptr)
}
const char* dryv_h264_last_error(void) { return g_err.c_str(); }
void dryv_h264_free(dryv_h264_frame* h) { delete h; }
const dryv_frame_params* dryv_h264_params(const dryv_h264_frame* h) { return &h->F.fp; }
const dryv_mb_desc* dryv_h264_mbs(const dryv_h264_frame* h) { return h->F.mbs.data(); }
const int16_t* dryv_h264_coeffs(const dryv_h264_frame* h) { return h->F.coeffs.data(); }
/* info[0..7] = bins decoded, slice NAL bytes, bits left unread, tail ok, Intra4x4 / Intra8x8 / Intra16x16 counts, slice qp */
void dryv_h264_info(const dryv_h264_frame* h, long long* info) {
  info[0] = h->F.bins;
  info[1] = (long long)h->F.slice_bytes;
  info[2] = (long long)h->F.bits_unread;
  info[3] = h->F.tail_ok ? 1 : 0;
  info[4] = h->F.kinds[0];
  info[5] = h->F.kinds[1];
  info[6] = h->F.kinds[2];
  info[7] = h->F.slice_qp;
}

/* The SPS's frame cropping rectangle in luma samples: left, right, top, bottom (0 when frame_cropping_flag is 0). */
void dryv_h264_crop(const dryv_h264_frame* h, int* crop4) {
  for (int k = 0; k < 4; k++) crop4[k] = h->F.crop[k];
}

/* The slice header's deblocking syntax elements (disable_deblocking_filter_idc, slice_alpha_c0_offset_div2, slice_beta_offset_div2). */
void dryv_h264_deblock_params(const dryv_h264_frame* h, dryv_deblock_params* out) { *out = h->F.deblock; }

/* Encodes one picture (flat scaling lists) as an Annex-B stream: SPS, PPS, one IDR I slice. Returns the byte count, or
 * 0 on failure / when `cap` is too small (call with cap = 0 to size the buffer: returns the needed size negated). */
long long dryv_h264_encode_idr(const dryv_frame_params* fp, const dryv_mb_desc* mbs, const int16_t* coeffs, int slice_qp,
                               uint8_t* out, size_t cap) {
  try {
    const std::vector<uint8_t> v = encode_idr_annexb(*fp, mbs, coeffs, slice_qp);
    if (v.size() > cap) return -(long long)v.size();
    memcpy(out, v.data(), v.size());
    return (long long)v.size();
  }
  DRYV_CATCH_ALL(0)
}

/* The same with a frame cropping rectangle (luma samples: left, right, top, bottom; even) in the SPS. */
long long dryv_h264_encode_idr_cropped(const dryv_frame_params* fp, const dryv_mb_desc* mbs, const int16_t* coeffs, int slice_qp,
                                       const int* crop4, uint8_t* out, size_t cap) {
  try {
    const std::vector<uint8_t> v = encode_idr_annexb(*fp, mbs, coeffs, slice_qp, crop4);
    if (v.size() > cap) return -(long long)v.size();
    memcpy(out, v.data(), v.size());
    return (long long)v.size();
  }
  DRYV_CATCH_ALL(0)
}

/* ---- batches of pictures ------------------------------------------------------------------------------------------------ */
struct dryv_h264_batch {
  std::vector<ParsedFrame> F;
  std::vector<dryv_mb_desc> mbs;
  std::vector<int16_t> coeffs;
  size_t skipped = 0;
};
/* Parses every picture of the stream that is one I slice (at most max_pictures; 0 = all). NULL on failure. */
dryv_h264_batch* dryv_h264_parse_all_mt(const uint8_t* data, size_t n, size_t max_pictures, unsigned n_threads);
dryv_h264_batch* dryv_h264_parse_all(const uint8_t* data, size_t n, size_t max_pictures) {
  return dryv_h264_parse_all_mt(data, n, max_pictures, 1);
}
/* n_threads: pictures are parsed in parallel (0 = one thread per hardware thread) */
dryv_h264_batch* dryv_h264_parse_all_mt(const uint8_t* data, size_t n, size_t max_pictures, unsigned n_threads) {
  try {
    std::unique_ptr<dryv_h264_batch> b(new dryv_h264_batch);
    b->F = parse_all_islices(data, n, max_pictures ? max_pictures : (size_t)-1, &b->skipped, n_threads);
    for (const ParsedFrame& f : b->F) {
      if (memcmp(&f.fp, &b->F[0].fp, sizeof(f.fp)) != 0) {
        g_err = "pictures of different parameters in one stream";
        return nullptr;
      }
      b->mbs.insert(b->mbs.end(), f.mbs.begin(), f.mbs.end());
      b->coeffs.insert(b->coeffs.end(), f.coeffs.begin(), f.coeffs.end());
    }
    return b.release();
  }
  DRYV_CATCH_ALL(nullptr)
}
/* Parameter sets only: fills fp_out from the stream's first SPS / PPS and returns the number of coded slice NAL units
 * (an upper bound on the pictures parse_all will deliver); 0 on failure. Cheap: no slice data is decoded. */
long long dryv_h264_stream_params(const uint8_t* data, size_t n, dryv_frame_params* fp_out) {
  try {
    const bool mp4 = n >= 12 && memcmp(data + 4, "ftyp", 4) == 0;
    const Stream S = mp4 ? demux_mp4(data, n, (size_t)-1) : demux_annexb(data, n);
    if (S.spsAll.empty() || S.ppsAll.empty()) fail("no SPS / PPS");
    if (S.slices.empty()) fail("no coded slice");
    // the parameter sets of the first picture parse_all will deliver (the first coded slice's, if none qualifies)
    const Stream::Nal* first = &S.slices[0];
    for (const Stream::Nal& nal : S.slices)
      if (is_whole_picture_islice(nal.p, nal.n)) {
        first = &nal;
        break;
      }
    if (fp_out) set_params(*fp_out, S.sps_of(*first), S.pps_of(*first));
    return (long long)S.slices.size();
  }
  DRYV_CATCH_ALL(0)
}
/* The same, with the records and coefficients written straight into the caller's batch buffers (e.g. page-locked memory from
 * dryv_recon_alloc_host: no intermediate copy), `capacity` pictures large. Returns the number of pictures, 0 on failure;
 * fp_out / info4 (pictures, skipped slices, all tails ok, reserved) are filled. */
long long dryv_h264_parse_all_into(const uint8_t* data, size_t n, size_t max_pictures, unsigned n_threads, dryv_mb_desc* mbs_out,
                                   int16_t* coeffs_out, size_t capacity, dryv_frame_params* fp_out, long long* info4) {
  try {
    size_t skipped = 0;
    const std::vector<ParsedFrame> F = parse_all_islices(data, n, max_pictures ? max_pictures : (size_t)-1, &skipped, n_threads,
                                                         mbs_out, coeffs_out, capacity);
    bool tails = true;
    for (const ParsedFrame& f : F) {
      if (memcmp(&f.fp, &F[0].fp, sizeof(f.fp)) != 0) {
        g_err = "pictures of different parameters in one stream";
        return 0;
      }
      tails = tails && f.tail_ok;
    }
    if (fp_out) *fp_out = F[0].fp;
    if (info4) {
      info4[0] = (long long)F.size();
      info4[1] = (long long)skipped;
      info4[2] = tails ? 1 : 0;
      info4[3] = 0;
    }
    return (long long)F.size();
  }
  DRYV_CATCH_ALL(0)
}
/* Test hook: the CABAC bins (value | kind << 1; kind 0 context-coded, 1 bypass, 2 terminate) of the stream's `picture`-th
 * intra picture, in decoding order, for an independent restatement of the slice-data syntax to parse. hdr4 = picture width
 * and height in macroblocks, transform_8x8_mode_flag, SliceQPY. Returns the number of bins (negated if `cap` is smaller; call
 * with cap = 0 to size the buffer), 0 on failure. */
long long dryv_h264_bin_log(const uint8_t* data, size_t n, size_t picture, uint8_t* out, size_t cap, int* hdr4) {
  try {
    std::vector<uint8_t> bins;
    const ParsedFrame F = parse_islice_with_bins(data, n, picture, bins);
    if (hdr4) {
      hdr4[0] = F.fp.pic_width_in_mbs;
      hdr4[1] = F.fp.pic_height_in_mbs;
      hdr4[2] = F.fp.transform_8x8_mode_flag;
      hdr4[3] = F.slice_qp;
    }
    if (bins.size() > cap) return -(long long)bins.size();
    memcpy(out, bins.data(), bins.size());
    return (long long)bins.size();
  }
  DRYV_CATCH_ALL(0)
}
void dryv_h264_batch_free(dryv_h264_batch* b) { delete b; }
size_t dryv_h264_batch_pictures(const dryv_h264_batch* b) { return b->F.size(); }
size_t dryv_h264_batch_skipped(const dryv_h264_batch* b) { return b->skipped; }   /* coded slices that were not whole intra pictures */
const dryv_frame_params* dryv_h264_batch_params(const dryv_h264_batch* b) { return &b->F[0].fp; }
const dryv_mb_desc* dryv_h264_batch_mbs(const dryv_h264_batch* b) { return b->mbs.data(); }
const int16_t* dryv_h264_batch_coeffs(const dryv_h264_batch* b) { return b->coeffs.data(); }
/* 1 if every picture's CABAC data ended at its terminating bin with only the stop bit and zeros behind it */
int dryv_h264_batch_tails_ok(const dryv_h264_batch* b) {
  for (const ParsedFrame& f : b->F)
    if (!f.tail_ok) return 0;
  return 1;
}
void dryv_h264_batch_crop(const dryv_h264_batch* b, int* crop4) {
  for (int k = 0; k < 4; k++) crop4[k] = b->F[0].crop[k];
}
/* n_pictures pictures (records / coefficients back to back) as one all-intra Annex-B stream: SPS, PPS, IDR slices. */
long long dryv_h264_encode_stream(const dryv_frame_params* fp, int n_pictures, const dryv_mb_desc* mbs, const int16_t* coeffs,
                                  int slice_qp, const int* crop4, uint8_t* out, size_t cap) {
  try {
    const std::vector<uint8_t> v = encode_idr_annexb(*fp, mbs, coeffs, slice_qp, crop4, n_pictures);
    if (v.size() > cap) return -(long long)v.size();
    memcpy(out, v.data(), v.size());
    return (long long)v.size();
  }
  DRYV_CATCH_ALL(0)
}

}  // extern "C"
