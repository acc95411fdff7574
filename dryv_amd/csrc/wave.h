// wave.h — the wave-level primitives the band kernel is written against.
//
// Two implementations of one interface:
//   * device (default): gfx950 builtins / inline asm. This is the product.
//   * -DDRYV_EMU (tests/emu only): every lane of a wave is a ucontext fiber on the host; cross-lane operations
//     exchange values through a shared slot behind a fiber barrier that also checks that all 64 lanes execute the
//     same operation (a cross-lane op under divergent control flow is a bug on the GPU too). Test infrastructure
//     for index / schedule logic; nothing in the shipped library is built with it.
#pragma once
#include <stdint.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))
#define DPP_ROW_HALF_MIRROR 0x141   // lane i of every eight <- lane 7 - i
#define DPP_WAVE_SHR1 0x138         // lane i <- lane i - 1 across the whole wave; lane 0 has no source

#ifndef DRYV_EMU
// =====================================================================================================
// device
// =====================================================================================================
#include <hip/hip_runtime.h>
#define WV __device__ __forceinline__

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
WV void lds_st8(int a, unsigned v) { *WV_LDS(uint8_t, a) = (uint8_t)v; }
WV void lds_st16(int a, unsigned v) { *WV_LDS(uint16_t, a) = (uint16_t)v; }
WV void lds_st32(int a, unsigned v) { *WV_LDS(unsigned, a) = v; }
WV void lds_st64(int a, u32x2 v) { *WV_LDS(u32x2, a) = v; }
WV void lds_or32(int a, unsigned v) { __hip_atomic_fetch_or(WV_LDS(unsigned, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WV void lds_st128(int a, u32x4 v) { *WV_LDS(u32x4, a) = v; }

// ---- VALU helpers -------------------------------------------------------------------------------------
WV unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
WV unsigned sad4(unsigned w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }
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
WV unsigned atomic_add_task(unsigned* p, unsigned v) { return atomicAdd(p, v); }
WV void atomic_or(unsigned* p, unsigned v) { atomicOr(p, v); }
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
// wave priority for the SIMD's issue arbiter (s_setprio 0..3)
template <int P>
WV void setprio() { __builtin_amdgcn_s_setprio(P); }
// the value, behind a barrier the optimiser cannot see through: what is derived from it is recomputed, not kept live
WV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

}  // namespace wv

#else
// =====================================================================================================
// host emulation (tests/emu): one fiber per lane
// =====================================================================================================
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define WV static inline

namespace wv {

struct EmuState {
  int cur_lane;
  int xbuf[64];
  unsigned long long xbuf64[64];
  const char* tag[64];
  uint8_t* lds;      // the workgroup's LDS (shared by the waves of a team)
  int lds_bytes;
};
extern EmuState* g_emu_cur;  // the wave that is running (several waves are interleaved: see band_emu.cpp)
#define g_emu (*g_emu_cur)
void emu_barrier(const char* tag);  // yields to the scheduler; returns once all 64 lanes arrived with the same tag

WV int lane_id() { return g_emu.cur_lane; }
WV void wave_sync() { emu_barrier("wave_sync"); }

WV int emu_dpp_src(int ctrl, int lane) {
  const int row = lane & ~15, i = lane & 15;
  if (ctrl < 0x100) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; return i + n <= 15 ? row + i + n : -1; }
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; return i - n >= 0 ? row + i - n : -1; }
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl - 0x120; return row + ((i - n) & 15); }
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));
  if (ctrl == 0x138) return lane - 1;
  fprintf(stderr, "emu: unsupported dpp ctrl %x\n", ctrl);
  abort();
}
template <int CTRL>
WV int dpp(int old, int src) {
  const int l = lane_id();
  g_emu.xbuf[l] = src;
  emu_barrier("dpp");
  const int s = emu_dpp_src(CTRL, l);
  const int r = s < 0 ? old : g_emu.xbuf[s];
  emu_barrier("dpp2");
  return r;
}
template <int CTRL>
WV int dppx(int src) { return dpp<CTRL>(src, src); }
template <int CTRL>
WV int dppz(int src) { return dpp<CTRL>(0, src); }
WV void swap32(unsigned& a, unsigned& b) {
  const int l = lane_id();
  g_emu.xbuf[l] = (int)(l < 32 ? b : a);  // what this lane gives away
  emu_barrier("swap32");
  if (l < 32) b = (unsigned)g_emu.xbuf[l + 32];
  else a = (unsigned)g_emu.xbuf[l - 32];
  emu_barrier("swap32b");
}
WV int bperm(int v, int srcLane) {
  const int l = lane_id();
  g_emu.xbuf[l] = v;
  emu_barrier("bperm");
  const int r = g_emu.xbuf[srcLane & 63];
  emu_barrier("bperm2");
  return r;
}
WV int rdlane(int v, int src) {
  const int l = lane_id();
  g_emu.xbuf[l] = v;
  emu_barrier("rdlane");
  const int r = g_emu.xbuf[src & 63];
  emu_barrier("rdlane2");
  return r;
}
WV int rfl(int v) { return rdlane(v, 0); }
WV unsigned long long ballot(bool p) {
  const int l = lane_id();
  g_emu.xbuf[l] = p ? 1 : 0;
  emu_barrier("ballot");
  unsigned long long m = 0;
  for (int k = 0; k < 64; k++) m |= (unsigned long long)(g_emu.xbuf[k] & 1) << k;
  emu_barrier("ballot2");
  return m;
}
WV bool any(bool p) { return ballot(p) != 0ull; }

WV void emu_lds_check(int a, int n) {
  if (a < 0 || a + n > g_emu.lds_bytes || (a % (n > 8 ? 8 : n)) != 0) {
    fprintf(stderr, "emu: bad LDS access addr %d size %d (lane %d)\n", a, n, lane_id());
    abort();
  }
}
#define EMU_LD(T, a) (emu_lds_check((a), sizeof(T)), *(const T*)(g_emu.lds + (a)))
WV unsigned lds_u8(int a) { return EMU_LD(uint8_t, a); }
WV unsigned lds_u16(int a) { return EMU_LD(uint16_t, a); }
WV int lds_i16(int a) { return EMU_LD(int16_t, a); }
WV unsigned lds_u32(int a) { return EMU_LD(unsigned, a); }
WV u32x2 lds_u64(int a) { emu_lds_check(a, 8); u32x2 v; memcpy(&v, g_emu.lds + a, 8); return v; }
WV u32x4 lds_u128(int a) { emu_lds_check(a, 16); if (a & 15) { fprintf(stderr, "emu: unaligned b128 %d\n", a); abort(); } u32x4 v; memcpy(&v, g_emu.lds + a, 16); return v; }
WV void lds_st8(int a, unsigned v) { emu_lds_check(a, 1); g_emu.lds[a] = (uint8_t)v; }
WV void lds_st16(int a, unsigned v) { emu_lds_check(a, 2); *(uint16_t*)(g_emu.lds + a) = (uint16_t)v; }
WV void lds_st32(int a, unsigned v) { emu_lds_check(a, 4); *(unsigned*)(g_emu.lds + a) = v; }
WV void lds_st64(int a, u32x2 v) { emu_lds_check(a, 8); memcpy(g_emu.lds + a, &v, 8); }
WV void lds_or32(int a, unsigned v) { emu_lds_check(a, 4); *(unsigned*)(g_emu.lds + a) |= v; }
WV void lds_st128(int a, u32x4 v) { emu_lds_check(a, 16); if (a & 15) { fprintf(stderr, "emu: unaligned b128 st %d\n", a); abort(); } memcpy(g_emu.lds + a, &v, 16); }

WV unsigned perm(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned long long src = ((unsigned long long)hi << 32) | lo;
  unsigned r = 0;
  for (int k = 0; k < 4; k++) {
    const unsigned s = (sel >> (8 * k)) & 0xff;
    unsigned b;
    if (s <= 7) b = (unsigned)(src >> (8 * s)) & 0xff;
    else if (s == 0x0c) b = 0;
    else if (s >= 0x0d) b = 0xff;
    else { fprintf(stderr, "emu: perm selector %x not modelled\n", s); abort(); }
    r |= b << (8 * k);
  }
  return r;
}
WV unsigned sad4(unsigned w) { return (w & 0xff) + ((w >> 8) & 0xff) + ((w >> 16) & 0xff) + (w >> 24); }
WV int med3(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }
WV int clamp3(int a, int lo, int hi) { return med3(a, lo, hi); }
WV int emu_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
WV unsigned pk_max(unsigned a, unsigned b) {
  const int16_t al = (int16_t)a, ah = (int16_t)(a >> 16), bl = (int16_t)b, bh = (int16_t)(b >> 16);
  return (unsigned)(uint16_t)(al > bl ? al : bl) | ((unsigned)(uint16_t)(ah > bh ? ah : bh) << 16);
}
WV unsigned pk_min(unsigned a, unsigned b) {
  const int16_t al = (int16_t)a, ah = (int16_t)(a >> 16), bl = (int16_t)b, bh = (int16_t)(b >> 16);
  return (unsigned)(uint16_t)(al < bl ? al : bl) | ((unsigned)(uint16_t)(ah < bh ? ah : bh) << 16);
}
WV unsigned pk_add_sat(unsigned a, unsigned b) {
  const int lo = emu_sat16((int)(int16_t)a + (int)(int16_t)b), hi = emu_sat16((int)(int16_t)(a >> 16) + (int)(int16_t)(b >> 16));
  return ((unsigned)lo & 0xffff) | ((unsigned)hi << 16);
}
WV unsigned pk_add(unsigned a, unsigned b) { return ((a + b) & 0xffff) | (((a >> 16) + (b >> 16)) << 16); }
WV unsigned pk_ashr5(unsigned a) {
  const int lo = (int)(int16_t)a >> 5, hi = (int)(int16_t)(a >> 16) >> 5;
  return ((unsigned)lo & 0xffff) | ((unsigned)hi << 16);
}
WV unsigned pk_ashr(unsigned a, unsigned sh) {
  const int lo = (int)(int16_t)a >> (sh & 15), hi = (int)(int16_t)(a >> 16) >> ((sh >> 16) & 15);
  return ((unsigned)lo & 0xffff) | ((unsigned)hi << 16);
}
WV unsigned pk_ashr1(unsigned a) { return pk_ashr(a, 0x00010001u); }
WV unsigned pk_ashr6(unsigned a) { return pk_ashr(a, 0x00060006u); }
WV unsigned pk_sub(unsigned a, unsigned b) { return ((a - b) & 0xffff) | (((a >> 16) - (b >> 16)) << 16); }
WV unsigned pk_mad(unsigned a, unsigned b, unsigned c) {
  return (((a & 0xffff) * (b & 0xffff) + (c & 0xffff)) & 0xffff) | ((((a >> 16) * (b >> 16) + (c >> 16)) & 0xffff) << 16);
}
WV unsigned sad_u16(unsigned a, unsigned b, unsigned acc) {
  const int dl = (int)(a & 0xffff) - (int)(b & 0xffff), dh = (int)(a >> 16) - (int)(b >> 16);
  return acc + (unsigned)(dl < 0 ? -dl : dl) + (unsigned)(dh < 0 ? -dh : dh);
}
WV unsigned sat_pk_u8(unsigned pair) {
  const int lo = (int16_t)pair, hi = (int16_t)(pair >> 16);
  return (unsigned)med3(lo, 0, 255) | ((unsigned)med3(hi, 0, 255) << 8);
}
WV unsigned cvt_pk_i16(int lo, int hi) { return ((unsigned)emu_sat16(lo) & 0xffff) | ((unsigned)emu_sat16(hi) << 16); }
WV unsigned alignbit(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
}

WV unsigned ld_sc1(const unsigned* p) { return *(const volatile unsigned*)p; }
WV void st_sc1(unsigned* p, unsigned v) { *(volatile unsigned*)p = v; }
WV unsigned atomic_add_task(unsigned* p, unsigned v) { const unsigned o = *p; *p += v; return o; }
WV void atomic_or(unsigned* p, unsigned v) { *p |= v; }
WV u32x4 ld_u128_a2(const void* p) { u32x4 v; memcpy(&v, p, 16); return v; }
WV void st_g128(void* p, u32x4 v) { memcpy(p, &v, 16); }
WV void st_g64(void* p, u32x2 v) { memcpy(p, &v, 8); }
WV void st_g128_sc1(void* p, u32x4 v) { memcpy(p, &v, 16); }
WV void wait_vm(int) {}
// a poll that failed: the wave yields to the other emulated waves (all 64 lanes get here together)
WV void sleep_short() { emu_barrier("@sleep"); }
WV void sleep_long() { emu_barrier("@sleep"); }
WV void sleep_team() { emu_barrier("@sleep"); }
WV void compiler_fence() {}
template <int P>
WV void setprio() {}
WV int opaque(int v) { return v; }

}  // namespace wv
#ifndef __HIPCC__
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
#endif
#endif
