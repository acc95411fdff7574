// recon_band.hip — device entry point and launch of the band kernel (band_kernel.h) for gfx950.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "band_kernel.h"
#include "band_launch.h"

namespace dryv {

// A workgroup = TEAMS_PER_WG teams of a FRONT, a BACK and a CHROMA wave (and a BACK8 wave in the builds for streams with the
// 8x8 transform). The fast build for streams without the 8x8 transform uses all 80 VGPRs that six waves per SIMD allow
// (lane-constant work hoisted up to that limit); the one with it is compiled for 5 waves per SIMD (96 VGPRs: five
// four-wave teams per CU), the wide builds (64-bit residual arithmetic, re-run of a flagged batch only) for 4.
// Grid shape: tools/band_variants.sh (measurements in DESIGN.md).
#ifndef DRYV_BAND_WPS
#define DRYV_BAND_WPS 6   // waves per SIMD the fast build is compiled for (<= 80 VGPRs)
#endif
#ifndef DRYV_BAND_WPS_I8
#define DRYV_BAND_WPS_I8 5   // ... the build for streams with the 8x8 transform (<= 96 VGPRs)
#endif
#ifndef DRYV_BAND_WGS_PER_CU
#define DRYV_BAND_WGS_PER_CU 2
#endif
#ifndef DRYV_BAND_WGS_PER_CU_I8
#define DRYV_BAND_WGS_PER_CU_I8 5
#endif
template <bool HAS_I8, bool WIDE>
__global__ void __launch_bounds__(64 * ((HAS_I8 || WIDE) ? 4 : band::WAVES_PER_WG), WIDE ? 4 : HAS_I8 ? DRYV_BAND_WPS_I8 : DRYV_BAND_WPS) band_kernel(const KParams P, band::Args A) {
  extern __shared__ __attribute__((aligned(64))) unsigned char lds[];
  const int ldsBase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  constexpr int tEnd = HAS_I8 ? band::T_END_I8 : band::T_END;
  A.waveBase = (int)blockIdx.x * band::WAVES_PER_WG;
  band::build_tables(P, ldsBase, (int)threadIdx.x, (int)blockDim.x, HAS_I8);
  constexpr int WPT = band::waves_per_team(HAS_I8);
  if (threadIdx.x < 64 * (blockDim.x / (64 * WPT))) {
    // the teams' flag words (16), and the table-row bytes of the block chain (24 words per queued record): some are never
    // written (the second block half has no block in four of the ten rounds) and must still be offsets of table rows
    const int q = (int)(threadIdx.x & 63);
    const int tsq = ldsBase + tEnd + (int)(threadIdx.x >> 6) * band::team_bytes(HAS_I8, WIDE);
    if (q < 16) wv::lds_st32(tsq + band::S_FLAGS + 4 * q, 0u);
    if (HAS_I8 && q < 16) wv::lds_st32(tsq + band::S_F8 + 4 * q, 0u);
    for (int k = q; k < 24 * band::NBUF; k += 64) wv::lds_st32(tsq + band::S_MSEQ + 4 * k, 0u);
  }
  __syncthreads();  // the only workgroup-level synchronisation: the teams are independent from here on
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // consecutive waves of a workgroup go to different SIMDs: a team's three (four) waves never share one
  const int team = wave / WPT, role = wave - team * WPT;
  const int ts = ldsBase + tEnd + team * band::team_bytes(HAS_I8, WIDE);
  // Wave priority by role (s_setprio: the SIMD's arbiter prefers the higher one when several waves can issue). BACK is the
  // wave on a band's critical path: one above the other two. Measured on the 300-picture batch, in one process against the
  // shipped 2 / 1 / 1 (BACK / CHROMA / FRONT): 2 / 2 / 0 (rounds 2 and 3 until the hand-off records) +1.0 ... +1.8 %, 3 / 2 / 2
  // equal, 3 / 1 / 1 and 3 / 2 / 1 +1 %, 2 / 2 / 2 +4 %, FRONT above CHROMA (2 / 1 / 2, 3 / 1 / 2) +6 ... +7 %, 1 / 0 / 0 +1.3 %;
  // round 2, before the mode pre-pass and the hand-off changes: no priorities 1.444 ms / BACK 2 alone 1.388 / 2 / 2 / 0 1.336-1.370.
  if (role == 1 && DRYV_BAND_PRIO_BACK) __builtin_amdgcn_s_setprio(DRYV_BAND_PRIO_BACK);
  if (HAS_I8 && role == 3 && DRYV_BAND_PRIO_BACK8) __builtin_amdgcn_s_setprio(DRYV_BAND_PRIO_BACK8);
  if (role == 2 && DRYV_BAND_PRIO_CHROMA) __builtin_amdgcn_s_setprio(DRYV_BAND_PRIO_CHROMA);
  if (role == 0 && (HAS_I8 ? DRYV_BAND_PRIO_FRONT_I8 : DRYV_BAND_PRIO_FRONT)) __builtin_amdgcn_s_setprio(HAS_I8 ? DRYV_BAND_PRIO_FRONT_I8 : DRYV_BAND_PRIO_FRONT);
#ifdef DRYV_BAND_ONLY_ROLE   // analysis only (tools/resource_usage.py): the registers one role needs when compiled alone
  if (role != DRYV_BAND_ONLY_ROLE) return;
#endif
  if (role == 1) band::band_back<HAS_I8>(P, A, ldsBase, ts);
  else if (role == 0) band::band_front<HAS_I8, WIDE>(P, A, ldsBase, ts);
  else if (HAS_I8 && role == 3) band::band_back8<WIDE>(P, A, ldsBase, ts);
  else band::band_chroma<HAS_I8, WIDE>(P, A, ldsBase, ts);
}

#ifndef DRYV_BAND_LDS_PAD
#define DRYV_BAND_LDS_PAD 0   // (tuning: extra LDS per workgroup, to cap the workgroups a CU accepts)
#endif
size_t band_lds_bytes(bool hasI8, bool wide, int teams) { return (size_t)(hasI8 ? band::T_END_I8 : band::T_END) + (size_t)teams * band::team_bytes(hasI8, wide) + DRYV_BAND_LDS_PAD; }
// Workgroup geometry per build (tools/ab_inproc.py with -DDRYV_BAND_TEAMS / -DDRYV_BAND_WGS_PER_CU variants):
//   * fast build, no 8x8 transform (80 VGPRs): 4 teams x 2 workgroups per CU; 1 x 8 measures the same, 2 x 4, 3 x 2 and
//     1 x 7 are 3 ... 13 % slower.
//   * fast build with the 8x8 transform (96 VGPRs, 5 waves per SIMD; more LDS per team): ONE four-wave team per workgroup,
//     five per CU = all 20 wave slots. Three-team workgroups left only one of them resident on a CU: 4.21 ms for the
//     100 x 4K batch against 3.59 (2 teams) and 3.05 (1 team, 5 or 6 workgroups per CU); with the fourth wave 2.68;
//     four workgroups per CU 3.00, six (80 VGPRs, 14 spilled) 2.72.
//   * wide builds (re-run of a flagged batch only; up to 128 VGPRs): one team per workgroup, four per CU.
int band_teams_per_block(bool hasI8, bool wide) { return (hasI8 || wide) ? 1 : band::TEAMS_PER_WG; }
int band_blocks_per_cu(bool hasI8, bool wide) { return wide ? 4 : hasI8 ? DRYV_BAND_WGS_PER_CU_I8 : DRYV_BAND_WGS_PER_CU; }

// Workspace: [task counter | pad to 256][modes progress words | pad to 256][mode records, 32 bytes per macroblock]
// [hand-off records, 64 bytes per macroblock of every band's last row][diagnostics]
static size_t band_prog_words(const KParams& P) { return (size_t)P.n_frames * ((P.H + 3) / 4); }
static size_t band_prog_bytes(const KParams& P) { return (band_prog_words(P) * 4 + 255) & ~(size_t)255; }
size_t band_reset_bytes(const KParams& P) { return 256 + band_prog_bytes(P); }
size_t band_handoff_offset(const KParams& P) {
  return (256 + band_prog_bytes(P) + (size_t)P.n_frames * P.W * P.H * (4 * band::MREC_WORDS) + 255) & ~(size_t)255;
}
size_t band_handoff_bytes(const KParams& P) {
  return ((size_t)P.n_frames * ((P.H + 3) / 4 - 1) * P.W * (4 * band::HAND_WORDS) + 255) & ~(size_t)255;
}
size_t band_profile_offset(const KParams& P) { return band_handoff_offset(P) + band_handoff_bytes(P); }
size_t band_workspace_bytes(const KParams& P) { return band_profile_offset(P); }

unsigned band_claims_per_launch(const KParams& P, int grid, bool wide) {
  return (unsigned)P.n_frames * (unsigned)((P.H + 3) / 4) + (unsigned)grid * (unsigned)band_teams_per_block(P.transform8x8 != 0, wide);
}

// wide: the build whose residual passes fall back to 64-bit arithmetic (see band_kernel.h, residual_pass).
hipError_t band_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                       void* d_workspace, int grid, bool wide, unsigned batch_seq, unsigned gen, unsigned task_base, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
  unsigned char* wsb = (unsigned char*)d_workspace;
  band::Args A;
  A.mbs = (const dryv_mb_desc*)d_mbs;
  A.coeffs = (const int16_t*)d_coeffs;
  A.yuv = (uint8_t*)d_yuv;
  A.status = d_status;
  A.taskCounter = (unsigned*)wsb;
  A.progM = (unsigned*)(wsb + 256);
  A.handoff = (unsigned*)(wsb + band_handoff_offset(P));
  A.gen = gen;
  A.taskBase = task_base;
  A.rowModes = (unsigned*)(wsb + 256 + band_prog_bytes(P));
  A.profile = nullptr;
  A.waveBase = 0;
  A.batchSeq = batch_seq;
#if defined(DRYV_BAND_PROFILE) || defined(DRYV_BAND_TRACE) || defined(DRYV_BAND_TIMELINE)
  A.profile = (unsigned long long*)(wsb + band_profile_offset(P));
#endif
  const bool i8 = P.transform8x8 != 0;
  const int teams = band_teams_per_block(i8, wide);
  const size_t ldsBytes = band_lds_bytes(i8, wide, teams);
  const dim3 g(grid), b(64 * band::waves_per_team(i8) * teams);
  if (ldsBytes > 65536) {  // (build variants with wide staging: beyond the default dynamic LDS limit)
    const void* fn = i8 ? (wide ? (const void*)band_kernel<true, true> : (const void*)band_kernel<true, false>)
                        : (wide ? (const void*)band_kernel<false, true> : (const void*)band_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e != hipSuccess) return e;
  }
  // (the events, where given, take the dispatch's own start and end: no marker packets of their own in the stream, which cost a
  // queue of launches several microseconds per launch between its kernels)
  if (i8 && wide) hipExtLaunchKernelGGL((band_kernel<true, true>), g, b, (unsigned)ldsBytes, stream, ev_start, ev_stop, 0, P, A);
  else if (i8) hipExtLaunchKernelGGL((band_kernel<true, false>), g, b, (unsigned)ldsBytes, stream, ev_start, ev_stop, 0, P, A);
  else if (wide) hipExtLaunchKernelGGL((band_kernel<false, true>), g, b, (unsigned)ldsBytes, stream, ev_start, ev_stop, 0, P, A);
  else hipExtLaunchKernelGGL((band_kernel<false, false>), g, b, (unsigned)ldsBytes, stream, ev_start, ev_stop, 0, P, A);
  return hipGetLastError();
}

}  // namespace dryv
