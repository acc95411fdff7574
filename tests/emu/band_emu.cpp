// band_emu.cpp — CPU emulation harness for dryv_amd/csrc/band_kernel.h (TEST INFRASTRUCTURE, not the product).
//
// Compiles the band kernel's source with -DDRYV_EMU: every lane of a wavefront is a ucontext fiber, cross-lane
// operations (DPP, ds_bpermute, readlane, ballot, wave barriers) meet at a fiber barrier that also checks that all
// 64 lanes execute the same operation. One wave processes every band task in queue order, so inter-band
// dependencies are always already satisfied. Used by tests/test_band_emu.py to check the kernel's index and
// schedule logic against the oracle without a GPU; timing, memory ordering and inline asm are out of its reach.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <vector>

#include "../../dryv_amd/csrc/band_kernel.h"
#include "../../dryv_amd/csrc/recon_params.h"

namespace wv {
EmuState g_emu;
static ucontext_t g_sched, g_fiber[64];
static int g_state[64];  // 0 runnable, 1 at barrier, 2 done
static void (*g_body)();

void emu_barrier(const char* tag) {
  const int l = g_emu.cur_lane;
  g_emu.tag[l] = tag;
  g_state[l] = 1;
  swapcontext(&g_fiber[l], &g_sched);
}
void emu_spin(const char* what) {
  fprintf(stderr, "emu: lane %d would spin (%s): a dependency is not satisfied in queue order\n", g_emu.cur_lane, what);
  abort();
}
static void trampoline() {
  g_body();
  g_state[g_emu.cur_lane] = 2;
  swapcontext(&g_fiber[g_emu.cur_lane], &g_sched);
}
static void run_wave(void (*body)()) {
  static std::vector<char> stacks;
  const size_t SS = 512 * 1024;
  stacks.resize(64 * SS);
  g_body = body;
  for (int l = 0; l < 64; l++) {
    getcontext(&g_fiber[l]);
    g_fiber[l].uc_stack.ss_sp = stacks.data() + l * SS;
    g_fiber[l].uc_stack.ss_size = SS;
    g_fiber[l].uc_link = &g_sched;
    makecontext(&g_fiber[l], trampoline, 0);
    g_state[l] = 0;
  }
  for (;;) {
    int done = 0, waiting = 0;
    for (int l = 0; l < 64; l++) {
      if (g_state[l] == 2) { done++; continue; }
      g_emu.cur_lane = l;
      g_state[l] = 0;
      swapcontext(&g_sched, &g_fiber[l]);
      if (g_state[l] == 2) done++;
      else waiting++;
    }
    if (done == 64) return;
    if (done != 0) {
      fprintf(stderr, "emu: %d lanes finished while %d wait at a cross-lane operation (divergent control flow)\n", done, waiting);
      abort();
    }
    for (int l = 1; l < 64; l++)
      if (strcmp(g_emu.tag[l], g_emu.tag[0]) != 0) {
        fprintf(stderr, "emu: lane %d is at '%s' while lane 0 is at '%s' (divergent cross-lane operation)\n", l, g_emu.tag[l], g_emu.tag[0]);
        abort();
      }
  }
}
}  // namespace wv

static dryv::KParams g_P;
static dryv::band::Args g_A;
static void body() {
  if (g_P.transform8x8) dryv::band::band_wave<true>(g_P, g_A, 0, dryv::band::T_END_I8);
  else dryv::band::band_wave<false>(g_P, g_A, 0, dryv::band::T_END);
}

extern "C" int dryv_emu_reconstruct(const dryv_frame_params* fp, uint32_t n_frames, const dryv_mb_desc* mbs,
                                    const int16_t* coeffs, uint8_t* yuv, unsigned* status_out) {
  int st = dryv::params::build_params(fp, n_frames, &g_P);
  if (st != DRYV_OK) return st;
  const int nBands = (g_P.H + 3) / 4;
  std::vector<unsigned> prog((size_t)n_frames * nBands, 0u), modes((size_t)n_frames * g_P.W * g_P.H, 0xEEEEEEEEu);
  unsigned counter = 0, status = 0;
  g_A = dryv::band::Args{mbs, coeffs, yuv, &status, prog.data(), modes.data(), &counter};
  memset(wv::g_emu.lds, 0xA5, sizeof(wv::g_emu.lds));
  dryv::band::build_tables(g_P, 0, 0, 1, g_P.transform8x8 != 0);
  wv::run_wave(body);
  if (status_out) *status_out = status;
  return DRYV_OK;
}
