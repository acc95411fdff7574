// wave_emu.h -- TEST INFRASTRUCTURE (tests/emu): the interface of dryv_amd/csrc/wave.h on the host. Every lane of a wave is a
// ucontext fiber; cross-lane operations exchange values through a shared slot behind a fiber barrier that also checks that
// all 64 lanes execute the same operation (a cross-lane op under divergent control flow is a bug on the GPU too). For index /
// schedule logic of the band and deblocking kernels. Include it BEFORE the kernel headers: it pre-empts the device
// implementation of wave.h (DRYV_WAVE_IMPL) and takes only the shared typedefs / DPP control macros from it.
#pragma once
#define DRYV_WAVE_IMPL
#include "../../dryv_amd/csrc/wave.h"
// =====================================================================================================
// host emulation (tests/emu): one fiber per lane
// =====================================================================================================
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define WV static inline
#define WV_LANES_IF(cond) if (true)

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
WV u32x3 lds_u96(int a) { emu_lds_check(a, 12); if (a & 15) { fprintf(stderr, "emu: unaligned b96 %d\n", a); abort(); } u32x3 v; memcpy(&v, g_emu.lds + a, 12); return v; }
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
WV unsigned sum4(unsigned w, unsigned acc) { return sad4(w) + acc; }
WV unsigned sum4_hi(unsigned w, unsigned acc) { return (sad4(w) << 16) + acc; }
WV unsigned dot4(unsigned a, unsigned b, unsigned acc) {
  for (int k = 0; k < 4; k++) acc += ((a >> (8 * k)) & 0xff) * ((b >> (8 * k)) & 0xff);
  return acc;
}
WV unsigned pk_lshr2(unsigned a) { return ((a & 0xffff) >> 2) | ((a >> 18) << 16); }
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
WV unsigned pk_ashr2(unsigned a) { return pk_ashr(a, 0x00020002u); }
WV unsigned pk_shl(unsigned a, unsigned sh) {
  return (((a & 0xffff) << (sh & 15)) & 0xffff) | ((((a >> 16) << ((sh >> 16) & 15)) & 0xffff) << 16);
}
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
WV unsigned long long ld_sc1_64(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
WV void st_sc1_64(unsigned long long* p, unsigned long long v) { *(volatile unsigned long long*)p = v; }
WV unsigned atomic_add_task(unsigned* p, unsigned v) { const unsigned o = *p; *p += v; return o; }
WV void atomic_or(unsigned* p, unsigned v) { *p |= v; }
WV void atomic_max(unsigned* p, unsigned v) { if (v > *p) *p = v; }
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
WV void acquire_agent() {}
template <int P>
WV void setprio() {}
WV int opaque(int v) { return v; }
WV void consume(unsigned) {}

}  // namespace wv
#ifndef __HIPCC__
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
#endif
