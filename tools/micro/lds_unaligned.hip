// Microbenchmark (tuning aid, not part of the product): what a byte-unaligned ds_read_b32 costs on gfx950 (the Intra4x4 chain
// reads three-tap windows of a block's edge array at any byte offset), against aligned dword reads and byte reads of the same
// pattern; and the issue cost of the opcodes the window form of the chain is built from.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(S) S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n"

// every lane reads `n` times from its own address base + k * stride; mis = byte misalignment added to every address.
// pattern 0: lane l reads 16 * (l >> 3) + (l & 7) [8 lanes inside one 16-byte array, like the chain], 1: 4 * l (one dword per lane)
template <int KIND>   // 0: ds_read_b32, 1: ds_read_u8 x 3, 2: ds_read_b64 (aligned 8) , 3: ds_read2_b32
__global__ void __launch_bounds__(256) k_lds(unsigned* out, int iters, int mis, int pattern) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int k = threadIdx.x; k < 16384; k += blockDim.x) lds[k] = (unsigned char)(k * 7 + 3);
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  int a = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + 4096 * w +
          (pattern == 0 ? 16 * (l >> 3) + (l & 7) : pattern == 1 ? 4 * l : 272 * (l >> 4) + 16 * ((l >> 3) & 1) + (l & 7)) + mis;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    unsigned v0, v1, v2, v3;
    if (KIND == 0) {
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:128\n ds_read_b32 %2, %4 offset:256\n ds_read_b32 %3, %4 offset:384\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
    } else if (KIND == 1) {
      asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:128\n ds_read_u8 %2, %4 offset:256\n ds_read_u8 %3, %4 offset:384\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
    } else {
      unsigned long long q0, q1;
      asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n ds_read2_b32 %1, %2 offset0:32 offset1:33\n s_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1) : "v"(a & ~3) : "memory");
      v0 = (unsigned)q0; v1 = (unsigned)(q0 >> 32); v2 = (unsigned)q1; v3 = (unsigned)(q1 >> 32);
    }
    acc += v0 ^ v1 ^ v2 ^ v3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// correctness: out[l] = the dword at byte address l (any alignment) of a known pattern
__global__ void k_check(unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) lds[k] = (unsigned char)(k * 7 + 3);
  __syncthreads();
  const int a = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + threadIdx.x;
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x] = v;
}

#define KOP(NAME, BODY)                                                                                       \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters) {                                     \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13 + threadIdx.x, f = 17;           \
    for (int i = 0; i < iters; i++) {                                                                         \
      asm volatile(REP8(REP8(BODY)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");          \
    }                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                               \
  }
KOP(o_dot4, "v_dot4_u32_u8 %0, %1, %4, %5\n v_dot4_u32_u8 %1, %2, %5, %4\n v_dot4_u32_u8 %2, %3, %4, %5\n v_dot4_u32_u8 %3, %0, %5, %4")
KOP(o_dot4_inl, "v_dot4_u32_u8 %0, %1, %4, 4\n v_dot4_u32_u8 %1, %2, %5, 4\n v_dot4_u32_u8 %2, %3, %4, 4\n v_dot4_u32_u8 %3, %0, %5, 4")
KOP(o_sadhi, "v_sad_hi_u8 %0, %1, %4, %5\n v_sad_hi_u8 %1, %2, %5, %4\n v_sad_hi_u8 %2, %3, %4, %5\n v_sad_hi_u8 %3, %0, %5, %4")
KOP(o_pklshr, "v_pk_lshrrev_b16 %0, 3, %1\n v_pk_lshrrev_b16 %1, 3, %2\n v_pk_lshrrev_b16 %2, 3, %3\n v_pk_lshrrev_b16 %3, 3, %0")
KOP(o_lshladd, "v_lshl_add_u32 %0, %1, 16, %4\n v_lshl_add_u32 %1, %2, 16, %5\n v_lshl_add_u32 %2, %3, 16, %4\n v_lshl_add_u32 %3, %0, 16, %5")
KOP(o_chain, "v_add_u32 %0, %1, %4\n v_dot4_u32_u8 %1, %2, %5, 4\n v_lshl_add_u32 %2, %3, 16, %4\n v_pk_lshrrev_b16 %3, 3, %0")
KOP(o_add, "v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_add_u32 %2, %3, %4\n v_add_u32 %3, %0, %5")

int main() {
  unsigned* d;
  hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // correctness
  {
    std::vector<unsigned> h(256);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 2048, 0, d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 256; l++) {
      unsigned want = 0;
      for (int b = 0; b < 4; b++) want |= (unsigned)(unsigned char)((l + b) * 7 + 3) << (8 * b);
      if (h[l] != want) bad++;
    }
    printf("unaligned ds_read_b32 correctness: %d of 256 addresses wrong\n", bad);
  }
  const int iters = 2000, blocks = 2048;
  auto timeit = [&](auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
  };
  for (int pattern = 0; pattern < 3; pattern++)
    for (int mis = 0; mis < 4; mis++) {
      const float t0 = timeit([&]() { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 16384, 0, d, iters, mis, pattern); });
      const float t1 = timeit([&]() { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 16384, 0, d, iters, mis, pattern); });
      const float t2 = timeit([&]() { hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(256), 16384, 0, d, iters, mis, pattern); });
      // wave-instructions: blocks * 4 waves * iters * 4 reads, on 256 CUs; cycles per read per CU at 2.4 GHz
      const double n = (double)blocks * 4 * iters * 4 / 256.0;
      printf("pattern %d misalign %d: ds_read_b32 %.3f ms (%.1f cyc/instr/CU)  ds_read_u8 %.3f ms (%.1f)  ds_read2_b32(aligned, 2 per 4) %.3f ms (%.1f per instr)\n",
             pattern, mis, t0, t0 * 2.4e6 / n, t1, t1 * 2.4e6 / n, t2, t2 * 2.4e6 / (n / 2));
    }
#define RUN(NAME)                                                                                              \
  {                                                                                                            \
    const float t = timeit([&]() { hipLaunchKernelGGL(NAME, dim3(blocks), dim3(256), 0, 0, d, 100); });       \
    const double n = (double)blocks * 4 * 100 * 256 / 1024.0;                                                  \
    printf("%-12s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", #NAME, t, t * 2.4e6 / n);            \
  }
  RUN(o_add) RUN(o_dot4) RUN(o_dot4_inl) RUN(o_sadhi) RUN(o_pklshr) RUN(o_lshladd) RUN(o_chain)
  return 0;
}
