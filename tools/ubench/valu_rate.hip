// Microbenchmark: issue cost of wave64 integer VALU ops on one SIMD (gfx950), for the roofline reasoning
// in DESIGN.md. Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(int* out, int iters) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (KIND == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }            // v_add_u32
      if (KIND == 1) { a0 = (a0 >> 1) + a1; a1 = (a1 >> 1) + a2; a2 = (a2 >> 1) + a3; a3 = (a3 >> 1) + a4; a4 = (a4 >> 1) + a5; a5 = (a5 >> 1) + a6; a6 = (a6 >> 1) + a7; a7 = (a7 >> 1) + a0; }
      if (KIND == 2) { a0 = __builtin_amdgcn_update_dpp(0, a1, 0x111, 0xf, 0xf, true) + a0; a1 = __builtin_amdgcn_update_dpp(0, a2, 0x111, 0xf, 0xf, true) + a1; a2 = __builtin_amdgcn_update_dpp(0, a3, 0x111, 0xf, 0xf, true) + a2; a3 = __builtin_amdgcn_update_dpp(0, a4, 0x111, 0xf, 0xf, true) + a3; a4 = __builtin_amdgcn_update_dpp(0, a5, 0x111, 0xf, 0xf, true) + a4; a5 = __builtin_amdgcn_update_dpp(0, a6, 0x111, 0xf, 0xf, true) + a5; a6 = __builtin_amdgcn_update_dpp(0, a7, 0x111, 0xf, 0xf, true) + a6; a7 = __builtin_amdgcn_update_dpp(0, a0, 0x111, 0xf, 0xf, true) + a7; }
      if (KIND == 3) { a0 = min(max(a0 + a1, 0), 255); a1 = min(max(a1 + a2, 0), 255); a2 = min(max(a2 + a3, 0), 255); a3 = min(max(a3 + a4, 0), 255); a4 = min(max(a4 + a5, 0), 255); a5 = min(max(a5 + a6, 0), 255); a6 = min(max(a6 + a7, 0), 255); a7 = min(max(a7 + a0, 0), 255); }
      if (KIND == 4) { a0 = __mul24(a0, a1) + a2; a1 = __mul24(a1, a2) + a3; a2 = __mul24(a2, a3) + a4; a3 = __mul24(a3, a4) + a5; a4 = __mul24(a4, a5) + a6; a5 = __mul24(a5, a6) + a7; a6 = __mul24(a6, a7) + a0; a7 = __mul24(a7, a0) + a1; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
void run(const char* name, int instrPerInner, int* d) {
  const int iters = 2000, blocks = 256 * 8, threads = 256;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, threads>>>(d, 10);
  hipEventRecord(e0); k<KIND><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 16 * instrPerInner;           // per wave
  const double wavesPerSimd = 8.0;
  printf("%-28s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms,
         ms * 1e6 / (instr * wavesPerSimd), ms * 1e6 / (instr * wavesPerSimd) * 2.4);
}
int main() {
  int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", 8, d); run<1>("v_lshr + v_add (2 ops)", 16, d); run<2>("v_add_u32 dpp", 8, d);
  run<3>("add + med3 (2 ops)", 16, d); run<4>("v_mad_i32_i24", 8, d);
  return 0;
}
