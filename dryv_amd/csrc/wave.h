// wave.h — the wave-level primitives the band kernel is written against.
//
// One implementation: gfx950 builtins / inline asm. (The tests' CPU lane emulator implements the same interface in
// tests/emu/wave_emu.h; its builds include that header first, which pre-empts the implementation below.)
#pragma once
#include <stdint.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

#define DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))
#define DPP_ROW_HALF_MIRROR 0x141   // lane i of every eight <- lane 7 - i
#define DPP_WAVE_SHR1 0x138         // lane i <- lane i - 1 across the whole wave; lane 0 has no source

#ifndef DRYV_WAVE_IMPL   // (a test build that brings its own implementation defines it first: tests/emu/wave_emu.h)
#define DRYV_WAVE_IMPL
#include <hip/hip_runtime.h>
#define WV __device__ __forceinline__
// a region that only the lanes with `cond` enter, around code that keeps its own per-lane predicates (so that the lane
// emulator, whose every lane must walk through the region's wave_sync calls, can take all lanes through it)
#define WV_LANES_IF(cond) if (cond)

namespace wv {

WV int lane_id() { return (int)(threadIdx.x & 63u); }
WV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <int CTRL>
WV int dpp(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false); }
// for controls under which every lane has a source (quad_perm, row_ror): no `old` value, so that the compiler can fold the
// move into the instruction that uses it
template <int CTRL>
WV int dppx(int src) { return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xF, 0xF, true); }
// the same for controls under which some lanes have no source (row_shr / row_shl): those lanes get 0 (bound_ctrl)
template <int CTRL>
WV int dppz(int src) { return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xF, 0xF, true); }
// lanes 32..63 of a <-> lanes 0..31 of b (v_permlane32_swap_b32)
WV void swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
WV int bperm(int v, int srcLane) { return __builtin_amdgcn_ds_bpermute(srcLane << 2, v); }
WV int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
WV int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
WV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
WV bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// ---- LDS by byte address ----------------------------------------------------------------------------
#define WV_LDS(T, a) ((__attribute__((address_space(3))) T*)(uintptr_t)(unsigned)(a))
WV unsigned lds_u8(int a) { return *WV_LDS(const uint8_t, a); }
WV unsigned lds_u16(int a) { return *WV_LDS(const uint16_t, a); }
WV int lds_i16(int a) { return *WV_LDS(const int16_t, a); }
WV unsigned lds_u32(int a) { return *WV_LDS(const unsigned, a); }
WV u32x2 lds_u64(int a) { return *WV_LDS(const u32x2, a); }
WV u32x4 lds_u128(int a) { return *WV_LDS(const u32x4, a); }
WV u32x3 lds_u96(int a) { return *WV_LDS(const u32x3, a); }   // 16-byte aligned (ds_read_b96)
WV void lds_st8(int a, unsigned v) { *WV_LDS(uint8_t, a) = (uint8_t)v; }
WV void lds_st16(int a, unsigned v) { *WV_LDS(uint16_t, a) = (uint16_t)v; }
WV void lds_st32(int a, unsigned v) { *WV_LDS(unsigned, a) = v; }
WV void lds_st64(int a, u32x2 v) { *WV_LDS(u32x2, a) = v; }
WV void lds_or32(int a, unsigned v) { __hip_atomic_fetch_or(WV_LDS(unsigned, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV void lds_st128(int a, u32x4 v) { *WV_LDS(u32x4, a) = v; }

// ---- VALU helpers -------------------------------------------------------------------------------------
WV unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
WV unsigned sad4(unsigned w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }
// sum of the four bytes of w, plus acc (v_sad_u8 against 0); the same sum in bits 31:16, plus acc (v_sad_hi_u8)
WV unsigned sum4(unsigned w, unsigned acc) { return __builtin_amdgcn_sad_u8(w, 0u, acc); }
WV unsigned sum4_hi(unsigned w, unsigned acc) { return __builtin_amdgcn_sad_hi_u8(w, 0u, acc); }
// a.b0 * b.b0 + ... + a.b3 * b.b3 + acc on unsigned bytes (v_dot4_u32_u8)
WV unsigned dot4(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }
WV int med3(int a, int lo, int hi) { return min(max(a, lo), hi); }
// the same as one instruction for bounds the compiler cannot order (lo <= hi is the caller's business)
WV int clamp3(int a, int lo, int hi) {
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(lo), "v"(hi));
  return d;
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
// packed signed 16-bit add with saturation (v_pk_add_i16 clamp)
WV unsigned pk_add_sat(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
WV unsigned pk_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
WV unsigned pk_ashr5(unsigned a) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) >> 5)); }
WV unsigned pk_ashr1(unsigned a) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) >> 1)); }
WV unsigned pk_ashr6(unsigned a) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) >> 6)); }
WV unsigned pk_ashr2(unsigned a) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) >> 2)); }
// both unsigned halves >> 2 (v_pk_lshrrev_b16)
WV unsigned pk_lshr2(unsigned a) {
  typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, (u16x2_)(__builtin_bit_cast(u16x2_, a) >> 2));
}
// per-half shift left by the low four bits of the matching half of sh (v_pk_lshlrev_b16)
WV unsigned pk_shl(unsigned a, unsigned sh) {
  return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) << (__builtin_bit_cast(s16x2, sh) & (s16x2)15)));
}
// per-half arithmetic shift right by the low four bits of the matching half of sh (v_pk_ashrrev_i16)
WV unsigned pk_ashr(unsigned a, unsigned sh) {
  return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) >> (__builtin_bit_cast(s16x2, sh) & (s16x2)15)));
}
WV unsigned pk_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
// per half: low 16 bits of a * b + c (v_pk_mad_u16; the same bits for signed and unsigned operands)
WV unsigned pk_mad(unsigned a, unsigned b, unsigned c) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c)));
}
// |a.lo - b.lo| + |a.hi - b.hi| + acc on unsigned halves (v_sad_u16)
WV unsigned sad_u16(unsigned a, unsigned b, unsigned acc) { return __builtin_amdgcn_sad_u16(a, b, acc); }
WV unsigned pk_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
WV unsigned pk_min(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// two signed 16-bit halves -> two unsigned bytes with saturation, in bits 15:0 (v_sat_pk_u8_i16)
WV unsigned sat_pk_u8(unsigned pair) {
  unsigned d;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(pair));
  return d;
}
// two int32 -> packed int16 pair with saturation (v_cvt_pk_i16_i32)
WV unsigned cvt_pk_i16(int lo, int hi) {
  unsigned d;
  asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
  return d;
}
WV unsigned alignbit(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// ---- global memory ------------------------------------------------------------------------------------
WV unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one aligned 8-byte relaxed agent-scope access (global_load / store_dwordx2 sc1): a data-tagged granule is one of these
WV unsigned long long ld_sc1_64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV void st_sc1_64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV unsigned atomic_add_task(unsigned* p, unsigned v) { return atomicAdd(p, v); }
WV void atomic_or(unsigned* p, unsigned v) { atomicOr(p, v); }
WV void atomic_max(unsigned* p, unsigned v) { atomicMax(p, v); }
struct __attribute__((packed, aligned(2))) U128a2 { u32x4 v; };
WV u32x4 ld_u128_a2(const void* p) { return ((const U128a2*)p)->v; }  // 2-byte-aligned 16-byte load (global_load_dwordx4)
struct __attribute__((packed, aligned(4))) U128a4 { u32x4 v; };
struct __attribute__((packed, aligned(4))) U64a4 { u32x2 v; };
WV void st_g128(void* p, u32x4 v) { ((U128a4*)p)->v = v; }  // dword-aligned 16-byte store (global_store_dwordx4)
WV void st_g64(void* p, u32x2 v) { ((U64a4*)p)->v = v; }
// 16-byte write-through store (global_store_dwordx4 ... sc1): for bytes another workgroup loads with sc1 loads
WV void st_g128_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
// all but the n youngest vector-memory operations of this wave have completed (n wave-uniform, 0..8)
WV void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}
#ifndef DRYV_BAND_SLEEP
#define DRYV_BAND_SLEEP 4  // s_sleep argument (x64 clocks) between two polls of a progress word
#endif
WV void sleep_short() { __builtin_amdgcn_s_sleep(DRYV_BAND_SLEEP); }
WV void sleep_long() { __builtin_amdgcn_s_sleep(32); }
#ifndef DRYV_TEAM_SLEEP
#define DRYV_TEAM_SLEEP 4
#endif
WV void sleep_team() { __builtin_amdgcn_s_sleep(DRYV_TEAM_SLEEP); }  // between two polls of the partner wave's LDS flag
WV void compiler_fence() { asm volatile("" ::: "memory"); }
// agent-scope acquire: invalidates this CU's L1 (buffer_inv sc1) and waits for it
WV void acquire_agent() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// wave priority for the SIMD's issue arbiter (s_setprio 0..3)
template <int P>
WV void setprio() { __builtin_amdgcn_s_setprio(P); }
// the value, behind a barrier the optimiser cannot see through: what is derived from it is recomputed, not kept live
WV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// a use of v the optimiser cannot remove (the loaded value of a cache-warming load: the load stays, its wait lands here)
WV void consume(unsigned v) { asm volatile("" : : "v"(v)); }

}  // namespace wv

#endif  // DRYV_WAVE_IMPL
