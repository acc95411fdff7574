// Microbenchmark (tuning aid, not part of the product): the band kernel's global-memory access pattern without any of
// its arithmetic, dependencies or LDS traffic. One wave per band task walks 126 steps; per step it loads the four
// macroblocks' records and coefficients the way FRONT / CHROMA do (16-byte loads at 32-byte stride, 4 rows 92 KB apart)
// and every other step stores two macroblocks' pixel rows per row of the band the way BACK / CHROMA do (16-byte stores,
// 32 contiguous bytes per pixel row at pitch 1920). Prints the time for the 300-frame 1080p batch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NS>
__global__ void __launch_bounds__(64) k(const uint4* __restrict__ coefs, const uint4* __restrict__ descs, uint8_t* __restrict__ yuv,
                                        unsigned* counter, int W, int H, int nF, int doStore, unsigned* sink) {
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  const int nBands = (H + 3) / 4;
  const unsigned total = (unsigned)nF * nBands;
  uint4 acc = {0, 0, 0, 0};
  for (;;) {
    unsigned task = 0;
    if (lane == 0) task = atomicAdd(counter, 1u);
    task = __shfl(task, 0);
    if (task >= total) break;
    const int b = task / nF, f = task % nF, r0 = 4 * b, nR = min(4, H - r0), nSteps = W + 2 * (nR - 1);
    const size_t frameMb = (size_t)f * W * H;
    uint8_t* plane = yuv + frameMb * 384;
    for (int s = 0; s < nSteps; s++) {
      const int x = min(max(s - 2 * g, 0), W - 1), r = min(r0 + g, H - 1);
      const size_t mb = frameMb + (size_t)r * W + x;
      const uint4* c = coefs + mb * 48;  // 768 B = 48 x 16 B
      const uint4 a0 = c[2 * i], a1 = c[2 * i + 1], a2 = c[32 + (i & 15)], d = descs[mb];
      acc.x ^= a0.x ^ a1.y ^ a2.z ^ d.w;
      if (doStore) {
        // luma: per row of the band, when its macroblock x completes a group of NS: 16 pixel rows x NS x 16 B
        for (int it = 0; it < NS; it++) {
          const int q = lane + 64 * it, fg = q / (16 * NS), fy = (q / NS) & 15, seg = q % NS;
          const int fx = s - 2 * fg;
          if (fg < nR && fx >= 0 && fx < W && (fx % NS) == NS - 1)
            *(uint4*)(plane + (size_t)(16 * (r0 + fg) + fy) * (W * 16) + 16 * ((fx - (NS - 1)) + seg)) = acc;
        }
        // chroma: 2 planes x 8 pixel rows x NS / 2 x 16 B
        for (int it = 0; it < (NS + 1) / 2; it++) {
          const int q = lane + 64 * it, LR = 8 * NS, fg = q / LR, w = q % LR, pl = w / (LR / 2), fy = (w / (NS / 2)) & 7, seg = w % (NS / 2);
          const int fx = s - 2 * fg;
          if (fg < nR && fx >= 0 && fx < W && (fx % NS) == NS - 1)
            *(uint4*)(plane + (size_t)W * H * 256 + (size_t)pl * W * H * 64 + (size_t)(8 * (r0 + fg) + fy) * (W * 8) + 8 * (fx - (NS - 1)) + 16 * seg) = acc;
        }
      }
    }
  }
  if (acc.x == 0x12345678u) sink[0] = acc.x;
}
int main() {
  const int W = 120, H = 68, nF = 300;
  const size_t nMb = (size_t)W * H * nF;
  uint4 *coefs, *descs; uint8_t* yuv; unsigned *counter, *sink;
  hipMalloc(&coefs, nMb * 768 + 4096); hipMalloc(&descs, nMb * 16); hipMalloc(&yuv, nMb * 384 + 4096); hipMalloc(&counter, 4); hipMalloc(&sink, 4);
  hipMemset(coefs, 1, nMb * 768); hipMemset(descs, 2, nMb * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ns : {0, 2, 4, 8})
    for (int teams : {2048, 4096}) {
      const int doStore = ns != 0;
      float best = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        hipMemset(counter, 0, 4);
        hipEventRecord(e0);
        if (ns == 4) k<4><<<teams, 64>>>(coefs, descs, yuv, counter, W, H, nF, doStore, sink);
        else if (ns == 8) k<8><<<teams, 64>>>(coefs, descs, yuv, counter, W, H, nF, doStore, sink);
        else k<2><<<teams, 64>>>(coefs, descs, yuv, counter, W, H, nF, doStore, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double bytes = nMb * (784.0 + (doStore ? 384.0 : 0.0));
      printf("%s (segments of %d macroblocks), %5d waves: %.3f ms  = %.2f TB/s algorithmic\n", doStore ? "loads + stores" : "loads only    ", ns, teams, best, bytes / best / 1e9);
    }
  return 0;
}
