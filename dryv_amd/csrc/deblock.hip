// deblock.hip — device entry point and launch of the deblocking kernel (deblock_kernel.h) for gfx950.
#include <hip/hip_runtime.h>

#include "deblock_kernel.h"
#include "deblock_params.h"
#include "deblock_launch.h"

namespace dryv {

#ifndef DRYV_DEBLOCK_WAVES
#define DRYV_DEBLOCK_WAVES 3   // independent band waves per workgroup (tools/db_variants.sh)
#endif
#ifndef DRYV_DEBLOCK_CHROMA_EVERY
#define DRYV_DEBLOCK_CHROMA_EVERY 3
#endif
#ifndef DRYV_DEBLOCK_WGS_PER_CU
#define DRYV_DEBLOCK_WGS_PER_CU 2   // 2 x (2 luma + 1 chroma waves) per CU measured best on 300 x 1080p: 2.61 ms (4 + 2 in one
                                    // workgroup 2.65, 2 + 2 in two 2.8, 9 or 12 waves per CU 2.9 - 3.0, 3 or 4 per CU 3.3 - 3.7)
#endif

__global__ void __launch_bounds__(64 * DRYV_DEBLOCK_WAVES) deblock_kernel(const deblock::DParams P, deblock::Args A) {
  extern __shared__ __attribute__((aligned(64))) unsigned char lds[];
  const int ldsBase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  deblock::build_tables(P, ldsBase, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();  // the only workgroup-level synchronisation: the waves are independent from here on
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // one wave in DRYV_DEBLOCK_CHROMA_EVERY filters Cb + Cr, the others luma (two independent band queues; a luma band takes
  // about twice as long as a chroma band)
  if (wave % DRYV_DEBLOCK_CHROMA_EVERY == DRYV_DEBLOCK_CHROMA_EVERY - 1) deblock::deblock_wave<false>(P, A, ldsBase, ldsBase + deblock::T_END + wave * deblock::S_BYTES);
  else deblock::deblock_wave<true>(P, A, ldsBase, ldsBase + deblock::T_END + wave * deblock::S_BYTES);
}

int deblock_waves_per_block() { return DRYV_DEBLOCK_WAVES; }
int deblock_blocks_per_cu() { return DRYV_DEBLOCK_WGS_PER_CU; }

hipError_t deblock_launch(const deblock::DParams& P, const void* d_mbs, void* d_yuv, unsigned* d_status, void* d_workspace,
                          int grid, hipStream_t stream) {
  unsigned char* wsb = (unsigned char*)d_workspace;
  deblock::Args A;
  A.mbs = (const dryv_mb_desc*)d_mbs;
  A.yuv = (uint8_t*)d_yuv;
  A.status = d_status;
  deblock::place_workspace(P, wsb, &A);
  const size_t ldsBytes = (size_t)deblock::T_END + (size_t)DRYV_DEBLOCK_WAVES * deblock::S_BYTES;
  hipLaunchKernelGGL(deblock_kernel, dim3(grid), dim3(64 * DRYV_DEBLOCK_WAVES), ldsBytes, stream, P, A);
  return hipGetLastError();
}

}  // namespace dryv
