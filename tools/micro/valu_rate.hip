// Microbenchmark (tuning aid, not part of the product): issue cost of the integer vector instruction classes the band
// kernel is made of, each forced by inline asm, with 1..8 waves per SIMD. Prints ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OP8(S) asm volatile(S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc")
template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13 + threadIdx.x, f = 17;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {  // 4 x (8 x 4) = 128 instructions per iteration
      if (KIND == 0) OP8("v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_add_u32 %2, %3, %4\n v_add_u32 %3, %0, %5");
      if (KIND == 1) OP8("v_perm_b32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_perm_b32 %2, %3, %4, %5\n v_perm_b32 %3, %0, %5, %4");
      if (KIND == 2) OP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf");
      if (KIND == 3) OP8("v_add3_u32 %0, %1, %4, %5\n v_add3_u32 %1, %2, %5, %4\n v_add3_u32 %2, %3, %4, %5\n v_add3_u32 %3, %0, %5, %4");
      if (KIND == 4) OP8("v_lshl_add_u32 %0, %1, 1, %4\n v_lshl_add_u32 %1, %2, 2, %5\n v_lshl_add_u32 %2, %3, 1, %4\n v_lshl_add_u32 %3, %0, 3, %5");
      if (KIND == 5) OP8("v_mul_i32_i24 %0, %1, %4\n v_mul_i32_i24 %1, %2, %5\n v_mul_i32_i24 %2, %3, %4\n v_mul_i32_i24 %3, %0, %5");
      if (KIND == 6) OP8("v_mul_lo_u32 %0, %1, %4\n v_mul_lo_u32 %1, %2, %5\n v_mul_lo_u32 %2, %3, %4\n v_mul_lo_u32 %3, %0, %5");
      if (KIND == 7) OP8("v_add_u32_sdwa %0, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %2, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %0, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD");
      if (KIND == 8) OP8("v_pk_add_i16 %0, %1, %4\n v_pk_add_i16 %1, %2, %5 clamp\n v_pk_add_i16 %2, %3, %4\n v_pk_add_i16 %3, %0, %5 clamp");
      if (KIND == 9) OP8("v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %5, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %5, vcc");
      if (KIND == 10) OP8("v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %0, %1, %4, vcc\n v_cmp_lt_u32 vcc, %3, %5\n v_cndmask_b32 %2, %3, %5, vcc");
      if (KIND == 11) OP8("v_sad_u8 %0, %1, %4, %5\n v_bfe_u32 %1, %2, 8, 8\n v_med3_i32 %2, %3, %4, %5\n v_ashrrev_i32 %3, 3, %0");
      if (KIND == 12) OP8("v_and_b32 %0, %1, %4\n v_lshlrev_b32 %1, 3, %2\n v_sub_u32 %2, %3, %4\n v_max_i32 %3, %0, %5");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[13] = {"v_add_u32 (VOP2)", "v_perm_b32 (VOP3)", "v_mov_b32_dpp", "v_add3_u32 (VOP3)", "v_lshl_add_u32 (VOP3)", "v_mul_i32_i24 (VOP2)",
                           "v_mul_lo_u32 (VOP3)", "v_add_u32_sdwa", "v_pk_add_i16 (VOP3P)", "v_cndmask_b32 (VOP2, vcc)", "v_cmp + v_cndmask (VOPC, VOP2)",
                           "sad_u8 / bfe / med3 / ashr mix", "and / lshl / sub / max (VOP2) mix"};
  for (int kind = 0; kind < 13; kind++) {
    if (kind == 9) continue;  // (selects without a compare in front read an undefined vcc: not a meaningful number)
    for (int wps = 1; wps <= 8; wps *= 8) {   // blocks of 256 threads = 1 wave per SIMD each
      const int grid = cus * wps;
      auto launch = [&]() {
        switch (kind) {
          case 0: k<0><<<grid, 256>>>(d, iters); break; case 1: k<1><<<grid, 256>>>(d, iters); break; case 2: k<2><<<grid, 256>>>(d, iters); break;
          case 3: k<3><<<grid, 256>>>(d, iters); break; case 4: k<4><<<grid, 256>>>(d, iters); break; case 5: k<5><<<grid, 256>>>(d, iters); break;
          case 6: k<6><<<grid, 256>>>(d, iters); break; case 7: k<7><<<grid, 256>>>(d, iters); break; case 8: k<8><<<grid, 256>>>(d, iters); break;
          case 9: k<9><<<grid, 256>>>(d, iters); break; case 10: k<10><<<grid, 256>>>(d, iters); break; case 11: k<11><<<grid, 256>>>(d, iters); break;
          default: k<12><<<grid, 256>>>(d, iters);
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * 128 * wps;
      printf("%-34s waves/SIMD %d: %.2f ns per wave-instruction per SIMD\n", names[kind], wps, ms * 1e6 / inst_per_simd);
    }
  }
  return 0;
}
