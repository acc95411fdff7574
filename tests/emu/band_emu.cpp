// band_emu.cpp — CPU emulation harness for dryv_amd/csrc/band_kernel.h (TEST INFRASTRUCTURE, not the product).
//
// Compiles the band kernel's source with -DDRYV_EMU: every lane of a wavefront is a ucontext fiber, cross-lane
// operations (DPP, ds_bpermute, readlane, ballot, wave barriers) meet at a fiber barrier that also checks that all
// 64 lanes execute the same operation. Several waves run interleaved: a wave runs until it polls a progress word in
// vain (its s_sleep) or finishes, then the next wave gets the processor -- so the inter-band hand-off protocol
// (claim order, progress words, deferred write-through) is exercised with real concurrency and a deadlock shows up
// as "no wave can make progress". Used by tests/test_band_emu.py to check the kernel's logic against the oracle
// without a GPU; timing, memory ordering and inline asm are out of its reach.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "wave_emu.h"
#include "../../dryv_amd/csrc/band_kernel.h"
#include "../../dryv_amd/csrc/recon_params.h"

namespace wv {
EmuState* g_emu_cur = nullptr;

struct Wave {
  EmuState st;
  int index = 0;
  ucontext_t fiber[64];
  int state[64];  // 0 runnable, 1 at barrier, 2 done
  std::vector<char> stacks;
  bool finished = false;
  unsigned long long slices = 0;
};
static Wave* g_wave = nullptr;
static ucontext_t g_sched;
static void (*g_body)();

void emu_barrier(const char* tag) {
  const int l = g_emu.cur_lane;
  g_emu.tag[l] = tag;
  g_wave->state[l] = 1;
  swapcontext(&g_wave->fiber[l], &g_sched);
}
static void trampoline() {
  g_body();
  Wave* w = g_wave;
  w->state[w->st.cur_lane] = 2;
  swapcontext(&w->fiber[w->st.cur_lane], &g_sched);
}
static void init_wave(Wave* w) {
  const size_t SS = 384 * 1024;
  w->stacks.resize(64 * SS);
  for (int l = 0; l < 64; l++) {
    getcontext(&w->fiber[l]);
    w->fiber[l].uc_stack.ss_sp = w->stacks.data() + l * SS;
    w->fiber[l].uc_stack.ss_size = SS;
    w->fiber[l].uc_link = &g_sched;
    makecontext(&w->fiber[l], trampoline, 0);
    w->state[l] = 0;
  }
}
// runs wave w until it finishes (returns 2), or until all its lanes sit in a failed poll (returns 1)
static int run_slice(Wave* w) {
  g_wave = w;
  g_emu_cur = &w->st;
  for (;;) {
    int done = 0, waiting = 0;
    for (int l = 0; l < 64; l++) {
      if (w->state[l] == 2) { done++; continue; }
      w->st.cur_lane = l;
      w->state[l] = 0;
      swapcontext(&g_sched, &w->fiber[l]);
      if (w->state[l] == 2) done++;
      else waiting++;
    }
    if (done == 64) { w->finished = true; return 2; }
    if (done != 0) {
      fprintf(stderr, "emu: %d lanes finished while %d wait at a cross-lane operation (divergent control flow)\n", done, waiting);
      abort();
    }
    for (int l = 1; l < 64; l++)
      if (strcmp(w->st.tag[l], w->st.tag[0]) != 0) {
        fprintf(stderr, "emu: lane %d is at '%s' while lane 0 is at '%s' (divergent cross-lane operation)\n", l, w->st.tag[l], w->st.tag[0]);
        abort();
      }
    if (w->st.tag[0][0] == '@') return 1;  // a poll failed: let the other waves run
  }
}
}  // namespace wv

static dryv::KParams g_P;
static dryv::band::Args g_A;
static bool g_wide;
static int wave_index() { return wv::g_wave->index; }
static void body() {
  using namespace dryv::band;
  const bool i8 = g_P.transform8x8 != 0;
  const int role = wave_index() % waves_per_team(i8);  // FRONT / BACK / CHROMA (/ BACK8) of team t (each team has its own LDS here)
  const int ts = i8 ? T_END_I8 : T_END;
  if (role == 3) {
    if (g_wide) band_back8<true>(g_P, g_A, 0, ts);
    else band_back8<false>(g_P, g_A, 0, ts);
  } else if (role == 1) {
    if (i8) band_back<true>(g_P, g_A, 0, ts);
    else band_back<false>(g_P, g_A, 0, ts);
  } else if (role == 2) {
    if (g_wide) {
      if (i8) band_chroma<true, true>(g_P, g_A, 0, ts);
      else band_chroma<false, true>(g_P, g_A, 0, ts);
    } else {
      if (i8) band_chroma<true, false>(g_P, g_A, 0, ts);
      else band_chroma<false, false>(g_P, g_A, 0, ts);
    }
  } else if (g_wide) {
    if (i8) band_front<true, true>(g_P, g_A, 0, ts);
    else band_front<false, true>(g_P, g_A, 0, ts);
  } else {
    if (i8) band_front<true, false>(g_P, g_A, 0, ts);
    else band_front<false, false>(g_P, g_A, 0, ts);
  }
}

// n_teams teams (a FRONT, a BACK and a CHROMA wave each, sharing one LDS array) claim bands concurrently; the 3 * n_teams waves are
// scheduled round-robin from wave `first`, `order` = +1 / -1, each until it polls in vain or finishes.
extern "C" int dryv_emu_reconstruct(const dryv_frame_params* fp, uint32_t n_frames, const dryv_mb_desc* mbs,
                                    const int16_t* coeffs, uint8_t* yuv, unsigned* status_out, int n_teams, int first,
                                    int order) {
  int st = dryv::params::build_params(fp, n_frames, &g_P);
  if (st != DRYV_OK) return st;
  if (n_teams < 1) n_teams = 1;
  const int WPT = dryv::band::waves_per_team(g_P.transform8x8 != 0);
  const int n_waves = WPT * n_teams;
  const int nBands = (g_P.H + 3) / 4;
  std::vector<unsigned> prog((size_t)n_frames * nBands, 0u), modes((size_t)n_frames * g_P.W * g_P.H * dryv::band::MREC_WORDS, 0xEEEEEEEEu);
  // the hand-off records: zeroed once, like the host API's; each pass is a launch with a generation of its own
  std::vector<unsigned> hand((size_t)n_frames * (nBands > 1 ? nBands - 1 : 0) * g_P.W * dryv::band::HAND_WORDS + 1, 0u);
  // like the host API: the fast build first; if it flags a block beyond int32 (status bit 1), the batch again with the wide build
  unsigned status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned counter = 0;
  for (int pass = 0; pass < 2; pass++) {
    memset(status, 0, sizeof status);
    g_wide = pass == 1;
    // (like the host API, nothing is reset between the two launches: the counter keeps counting, the words keep their tags)
    g_A = dryv::band::Args{mbs, coeffs, yuv, status, hand.data(), (unsigned)(pass + 1), prog.data(), modes.data(), &counter, counter, nullptr, 0, 0u};
    wv::g_body = body;
    std::vector<std::unique_ptr<wv::Wave>> waves;
    const int teamBytes = dryv::band::team_bytes(g_P.transform8x8 != 0, true);   // (one size for both passes: the wide build's)
    const int ldsBytes = (g_P.transform8x8 ? dryv::band::T_END_I8 : dryv::band::T_END) + teamBytes;
    std::vector<std::vector<uint8_t>> teamLds(n_teams, std::vector<uint8_t>(ldsBytes, 0xA5));
    for (int w = 0; w < n_waves; w++) {
      waves.emplace_back(new wv::Wave());
      waves.back()->index = w;
      waves.back()->st.lds = teamLds[w / WPT].data();
      waves.back()->st.lds_bytes = ldsBytes;
      wv::init_wave(waves.back().get());
      wv::g_emu_cur = &waves.back()->st;
      if (w % WPT == 0) {
        dryv::band::build_tables(g_P, 0, 0, 1, g_P.transform8x8 != 0);
        memset(teamLds[w / WPT].data() + ldsBytes - teamBytes + dryv::band::S_FLAGS, 0, 64);  // the team's flags
        if (g_P.transform8x8) memset(teamLds[w / WPT].data() + ldsBytes - teamBytes + dryv::band::S_F8, 0, 64);
        memset(teamLds[w / WPT].data() + ldsBytes - teamBytes + dryv::band::S_MSEQ, 0, 96 * dryv::band::NBUF);  // (as the kernel's prologue does)
      }
    }
    int live = n_waves;
    unsigned long long idle_rounds = 0;
    for (int k = 0; live > 0; k++) {
      wv::Wave* w = waves[(((first + order * k) % n_waves) + n_waves) % n_waves].get();
      if (w->finished) continue;
      const unsigned before = counter;
      auto progress = [&]() {  // anything a waiting wave could be waiting for: progress words, the teams' flag words
        unsigned long long t = 0;
        for (unsigned v : prog) t += v;
        for (unsigned v : hand) t += v;   // (the tags of the hand-off records)
        for (auto& L : teamLds)
          for (int q = 0; q < 64; q += 4) {
            unsigned v;
            memcpy(&v, L.data() + ldsBytes - teamBytes + dryv::band::S_FLAGS + q, 4);
            t += v;
            if (g_P.transform8x8) { memcpy(&v, L.data() + ldsBytes - teamBytes + dryv::band::S_F8 + q, 4); t += v; }
          }
        return t;
      };
      const unsigned long long sum0 = progress();
      const int r = wv::run_slice(w);
      if (r == 2) live--;
      const unsigned long long sum1 = progress();
      if (r == 2 || counter != before || sum1 != sum0) idle_rounds = 0;
      else if (++idle_rounds > (unsigned long long)n_waves * 64) {
        fprintf(stderr, "emu: deadlock: %d waves are polling and nothing makes progress\n", live);
        abort();
      }
    }
    if (!(status[0] & 2u)) break;
    if (getenv("DRYV_EMU_VERBOSE")) fprintf(stderr, "emu: batch flagged for the wide build\n");
  }
  if (status_out) *status_out = status[0] & ~2u;
  return DRYV_OK;
}

// ---- the deblocking kernel (dryv_amd/csrc/deblock_kernel.h) under the same emulator ---------------------------------------
#include "../../dryv_amd/csrc/deblock_kernel.h"
#include "../../dryv_amd/csrc/deblock_params.h"

static dryv::deblock::DParams g_DP;
static dryv::deblock::Args g_DA;
static void deblock_body() {  // even waves: luma, odd waves: chroma
  if (wave_index() & 1) dryv::deblock::deblock_wave<false>(g_DP, g_DA, 0, dryv::deblock::T_END);
  else dryv::deblock::deblock_wave<true>(g_DP, g_DA, 0, dryv::deblock::T_END);
}

// n_waves independent band waves (each with its own LDS), scheduled round-robin from `first`, `order` = +1 / -1.
extern "C" int dryv_emu_deblock(const dryv_frame_params* fp, const dryv_deblock_params* dp, uint32_t n_frames, const dryv_mb_desc* mbs,
                                uint8_t* yuv, unsigned* status_out, int n_waves, int first, int order) {
  int skip = 0;
  int st = dryv::deblock::build_dparams(fp, dp, n_frames, &g_DP, &skip);
  if (st != DRYV_OK) return st;
  if (status_out) *status_out = 0;
  if (skip) return DRYV_OK;
  if (n_waves < 1) n_waves = 1;
  n_waves *= 2;  // a luma and a chroma wave each
  std::vector<uint8_t> ws(dryv::deblock::workspace_bytes(g_DP), 0xC3);
  memset(ws.data(), 0, dryv::deblock::reset_bytes(g_DP));
  unsigned status[4] = {0, 0, 0, 0};
  g_DA.mbs = mbs;
  g_DA.yuv = yuv;
  g_DA.status = status;
  dryv::deblock::place_workspace(g_DP, ws.data(), &g_DA);
  wv::g_body = deblock_body;
  const int ldsBytes = dryv::deblock::T_END + dryv::deblock::S_BYTES;
  std::vector<std::unique_ptr<wv::Wave>> waves;
  std::vector<std::vector<uint8_t>> lds(n_waves, std::vector<uint8_t>(ldsBytes, 0xA5));
  for (int w = 0; w < n_waves; w++) {
    waves.emplace_back(new wv::Wave());
    waves.back()->index = w;
    waves.back()->st.lds = lds[w].data();
    waves.back()->st.lds_bytes = ldsBytes;
    wv::init_wave(waves.back().get());
    wv::g_emu_cur = &waves.back()->st;
    dryv::deblock::build_tables(g_DP, 0, 0, 1);
  }
  const size_t nProg = 2 * (size_t)n_frames * ((g_DP.H + 3) / 4);
  int live = n_waves;
  unsigned long long idle_rounds = 0;
  for (int k = 0; live > 0; k++) {
    wv::Wave* w = waves[(((first + order * k) % n_waves) + n_waves) % n_waves].get();
    if (w->finished) continue;
    auto progress = [&]() {
      unsigned long long t = *g_DA.taskCounter[0] + *g_DA.taskCounter[1];
      for (size_t q = 0; q < nProg; q++) t += g_DA.prog[0][q];
      return t;
    };
    const unsigned long long before = progress();
    const int r = wv::run_slice(w);
    if (r == 2) live--;
    if (r == 2 || progress() != before) idle_rounds = 0;
    else if (++idle_rounds > (unsigned long long)n_waves * 64) {
      fprintf(stderr, "emu: deblock deadlock: %d waves are polling and nothing makes progress\n", live);
      abort();
    }
  }
  if (status_out) *status_out = status[0];
  return DRYV_OK;
}
