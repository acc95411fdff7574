// h264_fuzz.cpp — test infrastructure (CPU only): mutates a seed stream and feeds every entry point of libdryv_h264's
// source (compiled in, with -fsanitize=address,undefined) the result. The parser must either deliver a batch or report
// an error; the sanitizers turn any out-of-bounds access, overflow or escaping exception into a non-zero exit.
//   h264_fuzz <seed file> <iterations> <prng seed>
#include "../../dryv_amd/host/h264_capi.cpp"

#include <stdio.h>
#include <stdlib.h>

static uint64_t rng_state;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(rng_state >> 33);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> seed;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) seed.insert(seed.end(), buf, buf + n);
  fclose(f);
  const int iters = atoi(argv[2]);
  rng_state = strtoull(argv[3], nullptr, 10);
  long ok = 0, rejected = 0;
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> d = seed;
    const int kind = (int)(rnd() % 4);
    // where mutations go: anywhere, or concentrated on the first kilobytes (container tables / parameter sets)
    const size_t span = (rnd() & 1) ? d.size() : std::min<size_t>(d.size(), 2048);
    const int flips = 1 + (int)(rnd() % 8);
    for (int k = 0; k < flips; k++) {
      const size_t at = rnd() % span;
      if (kind == 0) d[at] ^= (uint8_t)(1u << (rnd() % 8));
      else if (kind == 1) d[at] = (uint8_t)rnd();
      else if (kind == 2) d[at] = (rnd() & 1) ? 0xFF : 0x00;
      else if (at + 4 <= d.size()) { d[at] = 0xFF; d[at + 1] = 0xFF; d[at + 2] = 0xFF; d[at + 3] = (uint8_t)rnd(); }
    }
    if ((rnd() % 8) == 0) d.resize(rnd() % (d.size() + 1));   // truncation
    dryv_frame_params fp;
    const long long slices = dryv_h264_stream_params(d.data(), d.size(), &fp);
    dryv_h264_frame* h = dryv_h264_parse(d.data(), d.size());
    if (h) dryv_h264_free(h);
    bool good = false;
    if (slices > 0) {
      // the caller's side of parse_all_into: buffers sized from the parameters it was given, bounded for the test
      const size_t per = (size_t)fp.pic_width_in_mbs * fp.pic_height_in_mbs;
      const size_t cap = per && per <= 4096 ? std::min<size_t>((size_t)slices, 4) : 0;
      if (cap) {
        std::vector<dryv_mb_desc> mbs(per * cap);
        std::vector<int16_t> co(per * cap * 384);
        long long info[4];
        dryv_frame_params fp2;
        good = dryv_h264_parse_all_into(d.data(), d.size(), cap, 2, mbs.data(), co.data(), cap, &fp2, info) > 0;
      }
    }
    dryv_h264_batch* b = dryv_h264_parse_all_mt(d.data(), d.size(), 3, 2);
    if (b) dryv_h264_batch_free(b);
    (good ? ok : rejected)++;
  }
  printf("fuzz: %ld parsed, %ld rejected\n", ok, rejected);
  return 0;
}
