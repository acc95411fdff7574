// Microbenchmark (tuning aid, not part of the product): issue cost of the integer vector instructions the band kernel
// is made of, with 1..8 waves per SIMD. Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (KIND == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }                     // v_add_u32
      if (KIND == 1) { a = __builtin_amdgcn_perm(a, b, 0x07020500u + c); c = __builtin_amdgcn_perm(c, d, a); e = __builtin_amdgcn_perm(e, f, c); g = __builtin_amdgcn_perm(g, h, e);
                       b = __builtin_amdgcn_perm(b, a, g); d = __builtin_amdgcn_perm(d, c, b); f = __builtin_amdgcn_perm(f, e, d); h = __builtin_amdgcn_perm(h, g, f); }
      if (KIND == 2) { a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true) + b; c = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true) + d;
                       e = (unsigned)__builtin_amdgcn_update_dpp(0, (int)e, 0xB1, 0xF, 0xF, true) + f; g = (unsigned)__builtin_amdgcn_update_dpp(0, (int)g, 0x4E, 0xF, 0xF, true) + h;
                       b += a; d += c; f += e; h += g; }
      if (KIND == 3) { a = a * b + c; c = c * d + e; e = e * f + g; g = g * h + a; b = b * a + d; d = d * c + f; f = f * e + h; h = h * g + b; }   // v_mul_lo / mad
      if (KIND == 4) { a = __builtin_amdgcn_sad_u8(a, b, c); c = __builtin_amdgcn_sad_u8(c, d, e); e = __builtin_amdgcn_sad_u8(e, f, g); g = __builtin_amdgcn_sad_u8(g, h, a);
                       b = (b << 3) + a; d = (d << 3) + c; f = (f << 3) + e; h = (h << 3) + g; }             // v_sad_u8, v_lshl_add
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[5] = {"v_add_u32", "v_perm_b32", "dpp+add", "mul/mad u32", "sad_u8+lshl_add"};
  for (int kind = 0; kind < 5; kind++)
    for (int wps = 1; wps <= 8; wps *= 2) {   // blocks of 256 threads = 1 wave per SIMD each
      const int grid = cus * wps;
      auto launch = [&]() {
        switch (kind) { case 0: k<0><<<grid, 256>>>(d, iters); break; case 1: k<1><<<grid, 256>>>(d, iters); break;
                        case 2: k<2><<<grid, 256>>>(d, iters); break; case 3: k<3><<<grid, 256>>>(d, iters); break; default: k<4><<<grid, 256>>>(d, iters); }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * 16 * 8 * wps;   // wave-instructions per SIMD (source-level count)
      printf("%-16s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", names[kind], wps, ms, ms * 1e6 / inst_per_simd);
    }
  return 0;
}
