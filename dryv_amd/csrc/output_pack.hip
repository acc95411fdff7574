// output_pack.hip — the output stage behind reconstruction: frame cropping and NV12 packing on the
// device, so that only the bytes the caller wants cross PCIe.
//
// The reference parses frame_crop_*_offset (sps.rs:252-267) but never applies it (README.md:13 unchecked):
// write_to_yuv_file dumps the full coded planes (frame/mod.rs:48-70), and that stays this library's default output.
// This stage is what a caller that wants display-size pictures, or NV12 for a display / encoder API, adds: a pure
// byte-moving kernel (HBM-bound; 16 output bytes per lane, coalesced rows), no arithmetic on pixel values.
#include <hip/hip_runtime.h>

#include "output_pack.h"

namespace dryv {

namespace {

struct __attribute__((packed, aligned(1))) u128_a1 { unsigned x, y, z, w; };
struct __attribute__((packed, aligned(1))) u64_a1 { unsigned x, y; };

// One lane = 16 output bytes of one output row (the row's tail lane copies what is left, byte by byte).
// Rows of a frame: [0, oh) luma, then oh/2 chroma rows of Cb (I420) followed by oh/2 of Cr, or oh/2 interleaved (NV12).
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, PackGeo G) {
  const unsigned groupsY = (G.ow + 15u) >> 4;                 // 16-byte groups per luma row
  const unsigned cwOut = G.nv12 ? G.ow : G.ow >> 1;           // bytes per output chroma row
  const unsigned groupsC = (cwOut + 15u) >> 4;
  const unsigned rowsC = G.nv12 ? (G.oh >> 1) : G.oh;         // chroma output rows (I420: Cb rows then Cr rows)
  const unsigned long long perFrame = (unsigned long long)groupsY * G.oh + (unsigned long long)groupsC * rowsC;
  const unsigned long long total = perFrame * G.n_frames;
  for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned f = (unsigned)(t / perFrame);
    unsigned long long r = t - (unsigned long long)f * perFrame;
    const uint8_t* sf = src + (size_t)f * G.src_frame_bytes;
    uint8_t* df = dst + (size_t)f * G.dst_frame_bytes;
    if (r < (unsigned long long)groupsY * G.oh) {
      const unsigned y = (unsigned)(r / groupsY), gx = (unsigned)(r - (unsigned long long)y * groupsY) * 16u;
      const uint8_t* s = sf + (size_t)(y + G.crop_top) * G.sw + G.crop_left + gx;
      uint8_t* d = df + (size_t)y * G.ow + gx;
      if (gx + 16u <= G.ow) *(u128_a1*)d = *(const u128_a1*)s;
      else for (unsigned k = 0; gx + k < G.ow; k++) d[k] = s[k];
      continue;
    }
    r -= (unsigned long long)groupsY * G.oh;
    const unsigned y = (unsigned)(r / groupsC), gx = (unsigned)(r - (unsigned long long)y * groupsC) * 16u;
    const unsigned scw = G.sw >> 1, sch = G.sh >> 1, och = G.oh >> 1;
    const uint8_t* cb = sf + (size_t)G.sw * G.sh;
    const uint8_t* cr = cb + (size_t)scw * sch;
    uint8_t* dC = df + (size_t)G.ow * G.oh;
    if (!G.nv12) {
      const unsigned pl = y >= och ? 1u : 0u, yy = y - pl * och;
      const uint8_t* s = (pl ? cr : cb) + (size_t)(yy + (G.crop_top >> 1)) * scw + (G.crop_left >> 1) + gx;
      uint8_t* d = dC + (size_t)y * cwOut + gx;
      if (gx + 16u <= cwOut) *(u128_a1*)d = *(const u128_a1*)s;
      else for (unsigned k = 0; gx + k < cwOut; k++) d[k] = s[k];
    } else {
      const size_t so = (size_t)(y + (G.crop_top >> 1)) * scw + (G.crop_left >> 1) + (gx >> 1);
      uint8_t* d = dC + (size_t)y * cwOut + gx;
      if (gx + 16u <= cwOut) {
        const u64_a1 b = *(const u64_a1*)(cb + so), c = *(const u64_a1*)(cr + so);
        u128_a1 o;  // Cb0 Cr0 Cb1 Cr1 ...
        o.x = __builtin_amdgcn_perm(c.x, b.x, 0x05010400u);
        o.y = __builtin_amdgcn_perm(c.x, b.x, 0x07030602u);
        o.z = __builtin_amdgcn_perm(c.y, b.y, 0x05010400u);
        o.w = __builtin_amdgcn_perm(c.y, b.y, 0x07030602u);
        *(u128_a1*)d = o;
      } else {
        for (unsigned k = 0; gx + k < cwOut; k++) d[k] = (k & 1u) ? cr[so + (k >> 1)] : cb[so + (k >> 1)];
      }
    }
  }
}

}  // namespace

hipError_t pack_launch(const PackGeo& G, const void* d_src, void* d_dst, int num_cus, hipStream_t stream) {
  const unsigned long long lanes = ((unsigned long long)((G.ow + 15u) >> 4) * G.oh * 2ull) * G.n_frames;  // upper bound
  unsigned long long blocks = (lanes + 255) / 256;
  const unsigned long long cap = (unsigned long long)num_cus * 32;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)d_src, (uint8_t*)d_dst, G);
  return hipGetLastError();
}

}  // namespace dryv
