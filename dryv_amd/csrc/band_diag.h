// band_diag.h -- the band kernel's analysis scaffolding, in one place. The SHIPPED library is built with none of the
// switches below: every macro here is then empty (or the plain statement) and band_kernel.h reads as the kernel.
//
//   -DDRYV_BAND_EXP_SKIP=<mask>  drops phase k of the step (bit k): upper bounds of a phase's cost (tools/band_ablate.sh);
//                                such a build reconstructs wrong pictures
//   -DDRYV_BAND_EXP_DUP=k        runs phase k twice (the phases are idempotent): its dynamic instruction count as a
//                                difference of SQ_INSTS_VALU (tools/var_build.sh, var_run.sh)
//   -DDRYV_BAND_PROFILE          per-wave cycle sums per phase (tools/band_phases.py): PH(k) marks
//   -DDRYV_BAND_MARK             the same marks as comments in the assembly (tools/band_static.py)
//   -DDRYV_BAND_TRACE            breadcrumbs a host thread reads while the kernel runs (tools/band_trace.py)
//   -DDRYV_BAND_TIMELINE [+ _TLMODES / _TLENDS]  100 MHz stamps per band task (tools/band_timeline.py, modes_timeline.py,
//                                ends_timeline.py)
// The diagnostic builds write to a buffer of their own behind the workspace (Args::profile).
#pragma once

// Analysis only (tools/band_ablate.sh): -DDRYV_BAND_EXP_SKIP=<mask> drops phase k of the step (bit k) so that the difference
// in SQ_INSTS_VALU against the full kernel is that phase's dynamic instruction count. Such a build reconstructs wrong
// pictures; the shipped library is built with mask 0 and contains none of it.
#ifndef DRYV_BAND_EXP_SKIP
#define DRYV_BAND_EXP_SKIP 0
#endif
#define EXP_SKIP(k) ((((DRYV_BAND_EXP_SKIP) >> (k)) & 1) != 0)
// The opposite measurement, which does not let the compiler simplify anything around the phase: -DDRYV_BAND_EXP_DUP=k runs
// phase k of the step TWICE (the phases are idempotent), so that the difference in SQ_INSTS_VALU is its dynamic count.
#ifndef DRYV_BAND_EXP_DUP
#define DRYV_BAND_EXP_DUP (-1)
#endif
#define EXP_REP(k) for (int rep_ = 0, nrep_ = (DRYV_BAND_EXP_DUP) == (k) ? wv::opaque(2) : 1; rep_ < nrep_; rep_++)
#define EXP_DUP_IS(k) ((DRYV_BAND_EXP_DUP) == (k))

// Diagnostic builds only. -DDRYV_BAND_PROFILE (tools/band_phases.py): per-wave cycle sums per phase of the step.
// -DDRYV_BAND_TRACE (tools/band_trace.py): breadcrumbs only. Both write to a buffer of their own; the shipped library
// contains none of this.
#if defined(DRYV_BAND_PROFILE) && !defined(DRYV_BAND_TRACE)
#define DRYV_BAND_TRACE
#endif
#define BAND_NPH 16
#if defined(DRYV_BAND_PROFILE) && !defined(DRYV_EMU)
#define PH(k)                                                     \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                           \
    phAcc[k] += now_ - phT;                                       \
    phT = now_;                                                   \
  } while (0)
#elif defined(DRYV_BAND_MARK) && !defined(DRYV_EMU)
// analysis only (tools/band_static.py): phase boundaries as comments in the assembly, to count instructions between them
#define PH_STR2(x) #x
#define PH_STR(x) PH_STR2(x)
#define PH(k) asm volatile("; DRYV_MARK " PH_STR(__LINE__) " " #k ::: "memory")
#else
#define PH(k) do { } while (0)
#endif
// -DDRYV_BAND_TIMELINE (tools/band_timeline.py): per band task, 100 MHz timestamps of the claim (FRONT) and of BACK's
// first and last step, behind the trace records
// (with -DDRYV_BAND_TLMODES as well, tools/modes_timeline.py: the four stamps are the mode pre-pass's instead -- its start, the
// end of its first wait for the band above, its end, and FRONT's arrival at the wait for it)
#if defined(DRYV_BAND_TIMELINE) && !defined(DRYV_EMU)
#define TLINE_(task, k, val)                                                                                  \
  do {                                                                                                        \
    if (wv::lane_id() == 0 && A.profile)                                                                      \
      (A.profile + (size_t)65536 * (BAND_NPH + 4))[(size_t)(task) * 4 + (k)] = (unsigned long long)(val);   \
  } while (0)
#define TNOW() __builtin_amdgcn_s_memrealtime()
#if defined(DRYV_BAND_TLENDS)   // (tools/ends_timeline.py: when FRONT, BACK and CHROMA finish a task, and when CHROMA begins it)
#define TLINE(task, k, val) do { } while (0)
#define TLM(task, k, val) do { } while (0)
#define TLE(task, k, val) TLINE_(task, k, val)
#elif defined(DRYV_BAND_TLMODES)
#define TLINE(task, k, val) do { } while (0)
#define TLM(task, k, val) TLINE_(task, k, val)
#else
#define TLINE(task, k, val) TLINE_(task, k, val)
#define TLM(task, k, val) do { } while (0)
#endif
#else
#define TLINE(task, k, val) do { } while (0)
#define TLM(task, k, val) do { } while (0)
#define TNOW() 0
#endif
#ifndef TLE
#define TLE(task, k, val) do { } while (0)
#endif
// breadcrumbs: word k of this wave's 8-word trace record (behind the phase sums), written through so that a host
// thread can read them while the kernel is still running
#if defined(DRYV_BAND_TRACE) && !defined(DRYV_EMU)
#define TRACE(k, val)                                                                                                 \
  do {                                                                                                                \
    if (lane0 == 0 && A.profile)                                                                                      \
      wv::st_sc1((unsigned*)(A.profile + (size_t)65536 * BAND_NPH) + (size_t)(A.waveBase + (int)(threadIdx.x >> 6)) * 8 + (k), \
                 (unsigned)(val));                                                                                    \
  } while (0)
#else
#define TRACE(k, val) do { } while (0)
#endif

// per-wave accumulators of the phase marks: declared at the head of a wave's role, written out at its end
#if defined(DRYV_BAND_PROFILE) && !defined(DRYV_EMU)
#define BAND_DIAG_BEGIN()                                         \
  unsigned long long phAcc[BAND_NPH];                             \
  for (int k = 0; k < BAND_NPH; k++) phAcc[k] = 0;                \
  unsigned long long phT = __builtin_amdgcn_s_memtime();          \
  __builtin_amdgcn_s_waitcnt(0xC07F)
#define BAND_DIAG_END()                                                                                                     \
  do {                                                                                                                      \
    if (lane0 == 0 && A.profile)                                                                                            \
      for (int k = 0; k < BAND_NPH; k++) A.profile[(size_t)(A.waveBase + (int)(threadIdx.x >> 6)) * BAND_NPH + k] = phAcc[k]; \
  } while (0)
#else
#define BAND_DIAG_BEGIN() do { } while (0)
#define BAND_DIAG_END() do { } while (0)
#endif
