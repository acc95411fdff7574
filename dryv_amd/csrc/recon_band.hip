// recon_band.hip — device entry point and launch of the band kernel (band_kernel.h) for gfx950.
#include <hip/hip_runtime.h>

#include "band_kernel.h"
#include "recon_kernel.h"

namespace dryv {

// 320 threads = 5 band waves; 4 workgroups per CU = 5 waves per SIMD (<= 96 VGPRs), 20 bands per CU: the 5100 bands of
// the 300-frame 1080p batch are all resident at once.
template <bool HAS_I8, bool WIDE>
__global__ void __launch_bounds__(64 * band::WAVES_PER_WG, 5) band_kernel(const KParams P, band::Args A) {
  extern __shared__ __attribute__((aligned(64))) unsigned char lds[];
  const int ldsBase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  A.waveBase = (int)blockIdx.x * band::WAVES_PER_WG;
  band::build_tables(P, ldsBase, (int)threadIdx.x, (int)blockDim.x, HAS_I8);
  __syncthreads();  // the only workgroup-level synchronisation: the waves are independent from here on
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  band::band_wave<HAS_I8, WIDE>(P, A, ldsBase, ldsBase + (HAS_I8 ? band::T_END_I8 : band::T_END) + wave * band::S_BYTES);
}

size_t band_lds_bytes(bool hasI8) { return (size_t)(hasI8 ? band::T_END_I8 : band::T_END) + (size_t)band::WAVES_PER_WG * band::S_BYTES; }
int band_waves_per_block() { return band::WAVES_PER_WG; }
int band_blocks_per_cu() { return 4; }

// Workspace: [task counter | pad to 256][progress words][bottom-row modes][diagnostics]
size_t band_profile_offset(const KParams& P) {
  return ((256 + (((size_t)P.n_frames * P.H * 4) + 255) / 256 * 256 + (size_t)P.n_frames * P.W * P.H * 4) + 255) & ~(size_t)255;
}
static size_t band_prog_bytes(const KParams& P) { return (((size_t)P.n_frames * P.H * 4) + 255) & ~(size_t)255; }

// Workspace layout (recon_workspace_bytes): [task counter | pad to 256][progress words][bottom-row modes].
// wide: the build whose residual passes fall back to 64-bit arithmetic (see band_kernel.h, residual_pass).
hipError_t band_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                       void* d_workspace, int grid, bool wide, hipStream_t stream) {
  unsigned char* wsb = (unsigned char*)d_workspace;
  band::Args A;
  A.mbs = (const dryv_mb_desc*)d_mbs;
  A.coeffs = (const int16_t*)d_coeffs;
  A.yuv = (uint8_t*)d_yuv;
  A.status = d_status;
  A.taskCounter = (unsigned*)wsb;
  A.bandProg = (unsigned*)(wsb + 256);
  A.rowModes = (unsigned*)(wsb + 256 + band_prog_bytes(P));
  A.profile = nullptr;
  A.waveBase = 0;
#if defined(DRYV_BAND_PROFILE) || defined(DRYV_BAND_TRACE)
  A.profile = (unsigned long long*)(wsb + band_profile_offset(P));
#endif
  const bool i8 = P.transform8x8 != 0;
  const size_t ldsBytes = band_lds_bytes(i8);
  const dim3 g(grid), b(64 * band::WAVES_PER_WG);
  if (i8 && wide) hipLaunchKernelGGL((band_kernel<true, true>), g, b, ldsBytes, stream, P, A);
  else if (i8) hipLaunchKernelGGL((band_kernel<true, false>), g, b, ldsBytes, stream, P, A);
  else if (wide) hipLaunchKernelGGL((band_kernel<false, true>), g, b, ldsBytes, stream, P, A);
  else hipLaunchKernelGGL((band_kernel<false, false>), g, b, ldsBytes, stream, P, A);
  return hipGetLastError();
}

}  // namespace dryv
