// band_kernel.h — macroblock reconstruction, one TEAM of three wavefronts (four in builds with the 8x8 transform) per band of
// four macroblock rows (gfx950).
// The one reconstruction kernel of the library: every stream (with or without the 8x8 transform; fast and WIDE builds).
//
// Decomposition
//   * The unit of work is a band: 4 consecutive macroblock rows of one frame. Lanes 16g..16g+15 of a wave belong to
//     row g of the band. The rows advance in lockstep as a 2:1 diagonal: at step s row g works on macroblock
//     x = s - 2g, so neighbours A, B, C, D of every macroblock (slice/mod.rs:576-613) were finished by the same wave
//     one or two steps earlier -- no synchronisation between the rows of a band at all.
//   * A band is worked on by a team of three waves of one workgroup that run on different SIMDs:
//       FRONT   the step's record for BACK: luma residuals (into a queue of records in LDS), the step's slice of the
//               band's mode records, and the fetch of the band above's bottom luma lines;
//       BACK    luma prediction (Intra16x16, the Intra4x4 block wavefront), luma staging, stores: the wave on the frame's
//               critical path, and it never touches global memory except to store;
//       BACK8   (builds with the 8x8 transform only) the step's Intra8x8 macroblocks -- four serial 8x8 blocks each --
//               next to BACK's Intra4x4 chain instead of behind it: macroblocks of one step never depend on each
//               other. Two more LDS words: BACK8 -> BACK "the step's Intra8x8 macroblocks are in the tiles", BACK ->
//               BACK8 "the step's write-out (left columns, line rings, tile borders) is through";
//       CHROMA  all of chroma (residuals, prediction, staging, stores, hand-off): nothing in it depends on a luma
//               pixel or on the other two waves.
//     A band's prediction modes are derived before its first step by a pre-pass over the whole band, one lane per
//     macroblock (band_modes): by FRONT in the builds without the 8x8 transform, by CHROMA (between two tasks) with it
//     -- whichever is not the wave that finishes a task last (DRYV_BAND_MODES_IN_FRONT). The waves' priorities follow
//     who is behind: CHROMA takes BACK's while it trails BACK by more than a few steps (DRYV_BAND_CHROMA_TRAIL), FRONT
//     (8x8 builds) while its lead over BACK is short (DRYV_BAND_FRONT_LEAD).
//     FRONT -> BACK coupling is the per-step record in LDS (residuals, table rows, macroblock kinds) and two LDS words
//     per buffer (ready / free). A lone wave issues about one instruction every 4-5 cycles whatever its kind, so a
//     step's latency is the instruction count of the slowest wave; the split cuts it, and with it the frame's critical
//     path (a frame is a 2:1 wavefront of 120 + 2 x 67 macroblock steps that no amount of parallel frames shortens).
//   * Bands come off one queue in band-major order (band 0 of every frame, band 1 of every frame, ...): a band's
//     predecessor (same frame, band above) has a smaller number, was claimed earlier and is running, done, or some
//     team's next task, so there is no deadlock at any residency. FRONT claims a team's next task DRYV_BAND_CLAIM_AHEAD
//     steps before the current one ends, so that CHROMA has it when it gets there.
//   * Between bands the hand-off goes through memory, in records of its own (Args::handoff), as 8-byte granules that
//     carry their own tag (cdna_hip_programming.md Guideline 16, R2: the data is the flag; one aligned 8-byte sc1 store,
//     one 8-byte sc1 load): {4 pixels of the last row's bottom line, the launch's generation}, four for luma (BACK stores,
//     the FRONT of the band below fetches), two each for Cb and Cr (CHROMA). A granule is there when its tag is: no progress word, no wait
//     for the stores on the producer's side, one load per step (issued a step ahead) on the consumer's; a stale tag means
//     poll. The records are zeroed once per workspace layout, and a generation is never reused. Only the mode records
//     (CHROMA's pre-pass) keep a flag per band: sc1 stores, drained, then the flag.
//
// Inside a step (4 macroblocks)
//   * residual: ONE LANE PER 4x4 BLOCK. The lane loads its block's 16 coefficients (32 contiguous bytes of the
//     reference's list order) straight into registers, so the inverse zig-zag is register renaming, both butterfly
//     passes are in-lane, and no transpose exists. Luma: 64 lanes = 4 MB x 16 blocks; chroma: 32 lanes = 4 MB x
//     2 planes x 4 blocks, so CHROMA computes two steps' residuals per pass and then hands the halves over
//     (v_permlane32_swap): every step's chroma prediction runs on all 64 lanes, half a block per lane.
//     Intra16x16 DC: 4x4 Hadamard over the 16 lanes of the row group by DPP. What a pass derives from qp comes from one
//     LDS table row (T_QP). Packed 16-bit arithmetic where the block provably fits, int32 with a per-qp coefficient
//     bound (KParams::thr4) otherwise. A block beyond that flags the batch, and the host re-runs it (from the first
//     flagged batch of a queue) with the WIDE build of this kernel, whose passes switch to int64 (reference: isize)
//     for such waves: the result is the reference's for every int16 input.
//   * Intra16x16 and chroma prediction use the same lane-per-block layout: V, H and DC are one v_perm_b32
//     byte-select per pixel pair, plane is packed 16-bit arithmetic.
//   * Intra4x4 / Intra8x8 prediction modes: not here -- band_modes, once per band. Intra4x4 pixels by a 10-step 2:1 block
//     wavefront with 8 lanes per block (2 pixels each; 16 lanes and one pixel each in the four rounds that have one
//     block per macroblock): every pixel is (E[p] + 2E[q] + E[r] + 2) >> 2 of three samples whose tile offsets come
//     from a per-(mode, pixel) table.
//   * pixels are staged in LDS and stored as whole 64-byte row segments (NSY / NSC macroblocks at a time).
//   * Whatever depends on the lane id alone is computed once per kernel (BACK, FRONT's constants) or per task (CHROMA)
//     where the registers allow: the fast build sits at the 80-register limit of six waves per SIMD.
//
// Written against wave.h: the same source runs on the GPU and, lane by lane, in the CPU emulator of tests/emu.
#pragma once
#include "../../include/dryv_recon.h"
#include "kparams.h"
#include "wave.h"
#include "band_diag.h"

namespace dryv {
namespace band {

// ---- per-workgroup constant tables in LDS (byte offsets) ----------------------------------------------------
constexpr int T_LS4Z = 0;     // u16 [6][16]    LevelScale4x4 in list order
constexpr int T_QPC = 192;    // u8  [2][52]    QP'c for Cb / Cr as a function of QPY (transform.rs:194-216)
constexpr int T_THR4 = 304;   // u16 [52]
constexpr int T_THR8 = 408;   // u16 [52]
constexpr int T_LS4Q = 512;   // u16 [52][16]   per qp: LevelScale4x4 << max(qp/6 - 4, 0) in list order, 0xFFFF where that needs 17 bits
constexpr int T_LSMAX = 2176; // u16 [52]       per qp: the largest entry of its T_LS4Q row
constexpr int T_QP = 2304;    // u32 [52][8]    per qp, what a 4x4 residual pass derives from it: [0] rounding term and [1] right shift
                              //                of the packed path (both halves), [2] T_LSMAX | T_THR4 << 16, [3] shl | rnd << 8 |
                              //                shr << 16 | (32 * (qp % 6)) << 24; DC terms: [4] LevelScale(0,0), [5] qp / 6,
                              //                [6] rounding term and [7] shl | shr << 8 of the Intra16x16 DC scaling
constexpr int T_T4W = 3968;   // u32 [8][17][4] the block chain's prediction table, [pixel pair][row]: {byte offset of an aligned 8-byte
                              //                window of the block's edge array (S_EDGE), byte selector of the pair's left pixel, of its
                              //                right pixel (v_perm_b32 on the window: four bytes whose sum + 2 >> 2 is the pixel; 0x0d0d0d0d:
                              //                the block's DC), 0}. Rows: T4R_*
constexpr int T4W_PAIR = 272;  // a pixel pair's 16 entries + 16: the pairs' entries of one row on different banks
constexpr int T_END = 6144;
constexpr int T_LS8 = T_END;          // u16 [6][64]    LevelScale8x8, raster order (HAS_I8 only, like the next two)
// Intra8x8 prediction (BACK8), per (table row, lane): lane (py, half) of a macroblock's sixteen predicts pixels 4 * half .. + 3 of row
// py. Rows 0..8 = the modes, T8R_ZERO = quirk Q4's zero prediction (band_modes puts it in the mode record). A pixel is (the sum of
// four bytes of the block's filtered edge + 2) >> 2, as in the Intra4x4 chain: (a, b, b, c), (a, a, b, b) or (a, a, a, a); the
// bytes of a pixel PAIR lie in one aligned 8-byte window of the edge (tests/test_abi.py::test_prediction_tables_fit_aligned_windows).
constexpr int T_T8S = T_LS8 + 768;    // u32 [10][4 + pad]  the four pixels' byte selectors (v_perm_b32) on their pair's window; a row every T8S_ROW
constexpr int T8S_ROW = 272;          //                    (16 lanes x 16 + 16: the row groups of a wave, on different rows, on different banks)
constexpr int T_T8O = T_T8S + 10 * T8S_ROW;   // u32 [10][16]  byte offset of the first pair's window in the edge | the second pair's << 16
constexpr int T8R_ZERO = 9;
constexpr int T_ZZ8 = T_T8O + 640;    // u8  [64]       8x8 list index -> 2 * raster position
// the packed 16-bit form of the 8x8 residual (residual8x8_pk16): a lane works on rows 2p (low halves) and 2p + 1 (high halves)
constexpr int T_LS8P = T_ZZ8 + 64;    // u32 [6][4][8]  LevelScale8x8 of (row 2p, column j) | (row 2p + 1, column j) << 16
constexpr int T_ZZ8P = T_LS8P + 768;  // u8  [64]       8x8 list index -> byte offset of (row, column) in the pair-interleaved block:
                                      //                32 * (row >> 1) + 4 * column + 2 * (row & 1)
constexpr int T_THR8P = T_ZZ8P + 64;  // u16 [52]       per qp: the largest sum of |coefficients| of an 8x8 block for which the packed form is
                                      //                exact (0: none)
constexpr int T_END_I8 = (T_THR8P + 128 + 63) & ~63;
static_assert(T_END % 64 == 0 && T_END_I8 % 64 == 0, "table layout");

// The block chain's table rows (T_T4W; a mode record holds 16 x the row): the nine Intra4x4 modes of 8.3.1.2 by number
// (DC = 2: both neighbours there), and
constexpr int T4R_DC = 2;
constexpr int T4R_ZERO = 9;      // zero prediction: quirk Q4 (reference samples missing), an unsupported record
constexpr int T4R_NOTR3 = 10;    // modes 3 / 7 without a top-right block: T4..T7 := T3
constexpr int T4R_NOTR7 = 11;
constexpr int T4R_DC_TOP = 12;   // DC with the top neighbour alone
constexpr int T4R_DC_LEFT = 13;  // DC with the left neighbour alone; with neither: the left column then reads 128 (band_back)
constexpr int T4R_H16 = 14;      // Intra16x16 horizontal: the macroblock's left column, four samples per block row, in the T0..T3 bytes
constexpr int T4R_PRED16 = 15;   // Intra16x16 DC and plane: the block's sixteen predicted pixels ARE its edge array, [y][x] (Intra16x16
                                 // vertical = row 0 on the macroblock's top line)
// A block's edge array (S_EDGE, 16 bytes): everything its prediction reads, gathered where the neighbours' pixels are written
constexpr int E_DUMP = 0;        // (bytes 0, 1: where lanes that have nothing to scatter write)
constexpr int E_L3 = 3;          // L3 L2 L1 L0 (left column, bottom to top), corner, T0..T3, T4..T7 (top-right)
constexpr int E_CORNER = 7, E_T0 = 8, E_T4 = 12;
constexpr int E_SLOT = 16;
constexpr int E_ROW = 20 * E_SLOT;   // a row group's blocks 4 * by + bx, by = 0..4 (row 4: where the last block row's scatter lands);
                                     // 320 = 64 banks + 16: the four row groups' blocks of a round sit on different banks

// ---- per-team scratch in LDS (byte offsets from the team's base) ---------------------------------------------
// Output staging: a row's pixels are flushed to global memory NSY (luma) / NSC (chroma) macroblocks at a time, as
// 16 x NSY / 8 x NSC contiguous bytes per pixel row: the wider the segment, the fewer partial lines memory sees.
#ifndef DRYV_BAND_NSY
#define DRYV_BAND_NSY 4
#endif
#ifndef DRYV_BAND_NSC
#define DRYV_BAND_NSC 8
#endif
constexpr int NSY = DRYV_BAND_NSY, NSC = DRYV_BAND_NSC;
static_assert((NSY == 2 || NSY == 4 || NSY == 8) && (NSC == 2 || NSC == 4 || NSC == 8), "staging widths");
// luma: NSY / 2 tiles of two macroblocks per row (the Intra4x4 table holds tile offsets in 8 bits: the stride stays 40)
constexpr int TILE_STRIDE = 40;
constexpr int TILE_BYTES = 704;  // 17 rows x 40 + 8 (row y = -1 of slot 1 reaches 8 bytes into row y = 0), 64-aligned
constexpr int NP = NSY / 2;
constexpr int CW = 8 * NSC;      // chroma staging: bytes per pixel row
// FRONT -> BACK, double-buffered by the parity of the team's global step count
constexpr int S_RES = 0;         // i16 [2][4][16 blk][16]  luma residual, [4 * by + bx][y][x]; a row group every RES_ROW bytes (512 + 32:
constexpr int RES_ROW = 544;     //     8 banks behind the previous one: the rows of a band address their records alike in every
constexpr int RES_BUF = 4 * RES_ROW;          //     instruction), a buffer every RES_BUF
// the FRONT -> BACK record queue is NBUF steps deep (buffer = the team's global step count mod NBUF)
#ifndef DRYV_BAND_NBUF
#define DRYV_BAND_NBUF 2
#endif
constexpr int NBUF = DRYV_BAND_NBUF;
static_assert(NBUF >= 2 && NBUF <= 4, "record queue depth");
constexpr int S_MSEQ = S_RES + NBUF * RES_BUF;   // u8  [NBUF][4][2][12]  8 x the Intra4x4 table row per chain step and block half (always a multiple of 8)
constexpr int S_INFO = S_MSEQ + 96 * NBUF;       // u32 [NBUF][8]      kinds of the 4 macroblocks, Intra16x16 modes, task, step, parity, chain rounds with a DC block
constexpr int S_FLAGS = S_INFO + 32 * NBUF;      // u32 ready[4], free[4] (global step count + 1 of the record in / consumed from the
                                 //     buffer), taskRing[4], taskHead, taskTailC (FRONT -> CHROMA: the claimed tasks),
                                 //     modesDone (the wave that derives the modes -> the other: tasks whose mode pre-pass is through)
constexpr int F_READY = 0, F_FREE = 16, F_TASKS = 32, F_HEAD = 48, F_TAILC = 52, F_MODES = 56;
// BACK (+ FRONT writes row 0 of the luma ring: lines fetched from the band above)
constexpr int S_TILE = (S_FLAGS + 64 + 63) & ~63;  // u8  [4][NP][TILE_BYTES]  luma: tile (x >> 1) % NP, row j = y + 1, column 8 + 16 * (x & 1) + xr
constexpr int S_RINGY = S_TILE + 4 * NP * TILE_BYTES;  // bottom luma lines of the row above: row 0 [2][8][16], rows 1..3 [4][16]
constexpr int S_LEFTY = S_RINGY + 448;            // u8 [4][16]     column 15 of the macroblock to the left
// CHROMA
constexpr int S_STC = S_LEFTY + 64;               // u8 [4][2][8][CW] chroma staging, NSC macroblocks wide
constexpr int S_RINGC = S_STC + 64 * CW;             // [4][4][16]  bottom chroma lines of the row above: Cb[8] Cr[8]
constexpr int RINGC_ROW = 64, RINGC_ENT = 16;
constexpr int S_LEFTC = S_RINGC + 256;            // u8 [4][2][8]
// FRONT
constexpr int S_CARRYM = S_LEFTC + 64;            // u32 [4]     mode pre-pass: right-column modes of the macroblock left of the batch, per row
// BACK
constexpr int S_EDGE = S_CARRYM + 64;             // u8 [4][20][16]  the edge arrays (E_*) of the step's macroblocks, a row group every E_ROW
constexpr int S_BYTES = (S_EDGE + 4 * E_ROW + 4 * E_SLOT + 63) & ~63;   // (+ 4 slots: the dump bytes reach that far beyond the last row group's)
// builds that serve the 8x8 transform (HAS_I8) append, per team:
// (FRONT -> BACK8: an Intra8x8 macroblock's coefficients travel in the record itself, in its row group's part of S_RES, where
// BACK8 later puts the residuals: in raster order (the packed form: pair-interleaved, T_ZZ8P); a block every C8_BLK bytes, a
// macroblock every C8_MB = RES_ROW: the lanes that address their blocks alike land 2 banks apart (a 128-byte stride put the
// eight of a 32-lane group on ONE bank); 8-byte aligned: the passes read it with ds_read_b64)
constexpr int C8_BLK = 136, C8_MB = 4 * C8_BLK;
static_assert(C8_MB == RES_ROW, "an Intra8x8 macroblock's coefficients take its residuals' place in the record");
constexpr int S_E8 = S_BYTES;           // u8 [4][32]  BACK8: the filtered edge E1 of the current 8x8 block, a row group every 32 bytes: L7..L0, TL, T0..T15
constexpr int S_F8 = S_E8 + 128;       // u32 [16]  flags between BACK and BACK8: b8Done (BACK8 -> BACK: Intra8x8 macroblocks of step n - 1 are in
                                       //           the tiles), woDone (BACK -> BACK8: the write-out of step n - 1 is through)
constexpr int F8_DONE = 0, F8_WO = 4;
constexpr int G8_BLK = 144, G8_MB = 4 * G8_BLK;
#ifndef DRYV_BAND_MREC_I8_BYTES
#define DRYV_BAND_MREC_I8_BYTES 1024
#endif
constexpr int S_MREC_I8 = S_F8 + 64;   // [32][32]  (the mode pre-pass's staging area, as S_MREC below, in the builds with the 8x8 transform: half an
                                       //           iteration's records at a time -- five of these workgroups have to fit a CU's LDS)
constexpr int S_G8 = S_MREC_I8 + DRYV_BAND_MREC_I8_BYTES; // BACK8: row-pass output. 32-bit passes: T [4][2 blk8][8][8] (T: 4 bytes; 8 in the WIDE build, whose teams
                                       //           are that much larger); packed form: i16 [4][4 blk8][8][8] with G8_BLK / G8_MB strides, the four
                                       //           column pairs of row r rotated by r >> 1 (rows written and columns read without conflicts)
// builds without it append instead:
constexpr int S_MREC = S_BYTES;        // [64][32]  the wave that derives the modes: the mode records of a pre-pass iteration, on their way to memory
                                       //           as whole lines
constexpr int mrec_off(bool hasI8) { return hasI8 ? S_MREC_I8 : S_MREC; }
constexpr int mrec_bytes(bool hasI8) { return hasI8 ? DRYV_BAND_MREC_I8_BYTES : 2048; }
constexpr int team_bytes(bool hasI8, bool wide) { return hasI8 ? S_G8 + (wide ? 4096 : 4 * G8_MB) : S_BYTES + 2048; }
static_assert(S_EDGE % 16 == 0 && S_TILE % 64 == 0 && S_STC % 16 == 0, "scratch layout");
// luma ring entry (16 bytes) of macroblock e of the row above row g. Row 0's ring is written by FRONT, which runs up to
// two steps ahead of BACK -- also across a task boundary, hence one ring per task parity.
// the luma tile of macroblock x of row g
WV int tile_of(int ts, int g, int x) { return ts + S_TILE + TILE_BYTES * (NP * g + ((x >> 1) & (NP - 1))); }
WV int ringy(int ts, int g, int e, int par) {
  return g == 0 ? ts + S_RINGY + 128 * par + 16 * (e & 7) : ts + S_RINGY + 256 + 64 * (g - 1) + 16 * (e & 3);
}

// Which wave derives a band's prediction modes before its first step. Without the 8x8 transform: FRONT itself (-1.7 % on the
// 300-picture batch: it stood waiting for CHROMA, the wave that finishes a task last, 55 us per task; the pre-pass is 16 us).
// With it: CHROMA (FRONT carries the 8x8 residuals there and is the later one: +1.6 % on the 4K batch the other way round).
#ifndef DRYV_BAND_MODES_IN_FRONT
#define DRYV_BAND_MODES_IN_FRONT(hasI8) (!(hasI8))
#endif
// FRONT's priority by its lead over BACK (records published and not yet consumed): BACK's priority below this lead, its own
// otherwise; -1: no feedback. With the 8x8 transform, where FRONT is the later wave: 2 (4K batch -1.6 %; 1: +1.6 %); without:
// none (300-picture batch +1.7 % / +3.7 % for 1 / 2)
#ifndef DRYV_BAND_FRONT_LEAD
#define DRYV_BAND_FRONT_LEAD(hasI8) ((hasI8) ? 2 : -1)
#endif
#ifndef DRYV_BAND_MODES_ACQUIRE
#define DRYV_BAND_MODES_ACQUIRE 1
#endif
#ifndef DRYV_BAND_CHROMA_TRAIL
#define DRYV_BAND_CHROMA_TRAIL 4
#endif
#ifndef DRYV_BAND_I8_PK16
#define DRYV_BAND_I8_PK16 1   // (0: the 8x8 residuals always in 32 bits -- A/B builds)
#endif
#ifndef DRYV_BAND_TEAMS
#define DRYV_BAND_TEAMS 4   // teams per workgroup (tools/band_variants.sh)
#endif
constexpr int TEAMS_PER_WG = DRYV_BAND_TEAMS;
constexpr int WAVES_PER_TEAM = 3;  // FRONT, BACK, CHROMA
// builds that serve the 8x8 transform give a team a fourth wave, BACK8: the Intra8x8 macroblocks of a step (four serial
// blocks each) next to BACK's Intra4x4 chain instead of behind it
constexpr int waves_per_team(bool hasI8) { return hasI8 ? 4 : WAVES_PER_TEAM; }
constexpr int WAVES_PER_WG = WAVES_PER_TEAM * TEAMS_PER_WG;
// wave priorities by role (recon_band.hip); FRONT raises its own for the mode pre-pass of a task, when the other two
// waves of the team have nothing to do until it is through
#ifndef DRYV_BAND_PRIO_BACK
#define DRYV_BAND_PRIO_BACK 2
#endif
#ifndef DRYV_BAND_PRIO_CHROMA
#define DRYV_BAND_PRIO_CHROMA 1
#endif
#ifndef DRYV_BAND_PRIO_FRONT
#define DRYV_BAND_PRIO_FRONT 1
#endif
// (builds with the 8x8 transform, 4K batch, against the shipped 2 / 2 / 1 / 1 (BACK / BACK8 / CHROMA / FRONT) = 2.563 ms: FRONT 0
// +3.0 %, CHROMA 2 +3.7 %, BACK8 1 +0.5 %, 3 / 2 / 1 / 1 and 3 / 3 / 2 / 2 equal; with CHROMA at 2: FRONT 2 +2.8 %, 3 +18 %)
#ifndef DRYV_BAND_PRIO_FRONT_I8
#define DRYV_BAND_PRIO_FRONT_I8 DRYV_BAND_PRIO_FRONT
#endif
#ifndef DRYV_BAND_PRIO_BACK8
#define DRYV_BAND_PRIO_BACK8 DRYV_BAND_PRIO_BACK
#endif
#ifndef DRYV_BAND_PRIO_MODES
#define DRYV_BAND_PRIO_MODES 3
#endif
constexpr unsigned SPIN_LIMIT = 1u << 21;  // polls of a hand-off record / progress word (about a second) before a band gives up
constexpr unsigned TASK_END = 0xFFFFFFFFu;
constexpr int PROG_SHIFT = 11;                      // (a row has at most 1024 macroblocks)
constexpr unsigned PROG_GEN_MASK = 0x1FFFFFu;       // the host never hands out a generation whose low 21 bits are 0

struct Args {
  const dryv_mb_desc* mbs;
  const int16_t* coeffs;
  uint8_t* yuv;
  unsigned* status;     // [0] bit 0 unsupported record, bit 1 a block beyond the fast build's arithmetic, bit 2 a band gave up waiting;
                        // [1..3] where it gave up; [4] ~(sequence number of the first queued batch that raised bit 1)
  unsigned* handoff;    // [frame][band that has a band below][mb][16]: the bottom lines of the band's last row, in eight 8-byte
                        // granules {4 pixels, tag}: four of luma, two of Cb, two of Cr. tag = gen: a granule is there when its tag is
  unsigned gen;         // the launch's generation (never 0, never repeated over the life of the workspace)
  unsigned* progM;      // [frame][band]: (gen & PROG_GEN_MASK) << PROG_SHIFT | the macroblocks of the band's last row whose mode
                        // records (below) are visible; a word of another launch counts as 0: nothing is zeroed per launch
  unsigned* rowModes;   // [mb][8]: the mode record of every macroblock (MREC_*), written by the band's mode pre-pass
  unsigned* taskCounter;
  unsigned taskBase;    // ... which keeps counting from launch to launch: this launch's tasks are its values from taskBase on
  unsigned long long* profile;  // DRYV_BAND_PROFILE builds only
  int waveBase;                 // (global index of the workgroup's first wave, for the same)
  unsigned batchSeq;            // position of this launch in the host's queue of batches (status[4])
};


// zig-zag (frame/mod.rs:185-209): list index of matrix element (row, col)
#define ZZ4IDX(r, c) ((int)((0xFEA9DB83C7426510ull >> (4 * ((r) * 4 + (c)))) & 15ull))
// Intra4x4 block wavefront: step T runs the blocks with bx + 2*by == T (at most two)
constexpr int stepByLo(int t) { return t < 2 ? 0 : (t - 2) >> 1; }
constexpr int stepByHi(int t) { return (t >> 1) < 3 ? (t >> 1) : 3; }
constexpr int zidx(int bx, int by) { return 8 * (by >> 1) + 4 * (bx >> 1) + 2 * (by & 1) + (bx & 1); }

// Builds the per-workgroup tables. Called by every thread of the workgroup (tid / nthreads), followed by a barrier.
WV void build_tables(const KParams& P, int ldsBase, int tid, int nthreads, bool hasI8) {
  for (int k = tid; k < 96; k += nthreads) wv::lds_st16(ldsBase + T_LS4Z + 2 * k, P.ls4z[k]);
  for (int k = tid; k < 104; k += nthreads) {
    // 8.5.8: qPI = Clip3(0, 51, QPY + offset); QPc = qPI < 30 ? qPI : table 8-15
    const int qpi = min(max((k % 52) + (k < 52 ? P.cqo_cb : P.cqo_cr), 0), 51);
    const int d = qpi - 30;
    const int delta = d < 0 ? 0 : d < 16 ? (int)((0x7765544332221111ull >> (4 * d)) & 15ull) : (int)((0xCBA998u >> (4 * (d - 16))) & 15u);
    wv::lds_st8(ldsBase + T_QPC + k, (unsigned)(qpi - delta));
  }
  for (int k = tid; k < 52; k += nthreads) {
    wv::lds_st16(ldsBase + T_THR4 + 2 * k, P.thr4[k]);
    wv::lds_st16(ldsBase + T_THR8 + 2 * k, P.thr8[k]);
  }
  // The packed path's row of a qp below 24, where the dequantisation shifts right ((c * LS + 2^(3 - qp/6)) >> (4 - qp/6)): when
  // every LevelScale entry of qp % 6 is a multiple of 2^(4 - qp/6) -- the flat lists: 16 * normAdjust -- the row holds
  // LS >> (4 - qp/6) and the pass neither rounds nor shifts: c * (LS >> s) is exactly (c * LS + 2^(s - 1)) >> s
  auto folded_shr = [&](int qd, int qm) -> int {
    if (qd >= 4) return 0;
    unsigned all = 0;
    for (int k = 0; k < 16; k++) all |= (unsigned)P.ls4z[16 * qm + k];
    return (all & ((1u << (4 - qd)) - 1u)) == 0u ? 4 - qd : 0;
  };
  for (int k = tid; k < 832; k += nthreads) {
    const int qp = k >> 4, qd = qp / 6, qm = qp - 6 * qd;
    const unsigned v = ((unsigned)P.ls4z[16 * qm + (k & 15)] << (qd > 4 ? qd - 4 : 0)) >> folded_shr(qd, qm);
    wv::lds_st16(ldsBase + T_LS4Q + 2 * k, v > 0xFFFFu ? 0xFFFFu : v);
  }
  for (int qp = tid; qp < 52; qp += nthreads) {
    const int qd = qp / 6, qm = qp - 6 * qd;
    const int fold = folded_shr(qd, qm);
    unsigned m = 0;
    for (int k = 0; k < 16; k++) m = max(m, ((unsigned)P.ls4z[16 * qm + k] << (qd > 4 ? qd - 4 : 0)) >> fold);
    m = m > 0xFFFFu ? 0xFFFFu : m;
    wv::lds_st16(ldsBase + T_LSMAX + 2 * qp, m);
    // d = ((c * LS) << shl + rnd) >> shr with shl = max(qp/6 - 4, 0), shr = max(4 - qp/6, 0), rnd = 2^(3 - qp/6) below qp 24
    const unsigned shl = qd > 4 ? qd - 4 : 0, shr = qd < 4 ? 4 - qd : 0, rnd = qd < 4 ? 1u << (3 - qd) : 0u;
    wv::lds_st32(ldsBase + T_QP + 32 * qp, fold ? 0u : rnd * 0x10001u);
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 4, fold ? 0u : shr * 0x10001u);
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 8, m | ((unsigned)P.thr4[qp] << 16));
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 12, shl | (rnd << 8) | (shr << 16) | ((unsigned)(32 * qm) << 24));
    // Intra16x16 DC (pred16x16.rs:465-479): qp >= 36: (f * LS) << (qp/6 - 6), else (f * LS + 2^(5 - qp/6)) >> (6 - qp/6)
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 16, (unsigned)P.ls4z[16 * qm]);
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 20, (unsigned)qd);
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 24, qd < 6 ? 1u << (5 - qd) : 0u);
    wv::lds_st32(ldsBase + T_QP + 32 * qp + 28, (unsigned)(qd > 6 ? qd - 6 : 0) | ((unsigned)(qd < 6 ? 6 - qd : 0) << 8));
  }
  for (int k = tid; k < 128; k += nthreads) {
    // entry of (pixel pair p, table row m). P.t4 names a sample by its index j on the line [L3 L2 L1 L0 | corner | T0..T7]
    // (byte E_L3 + j of the block's edge array) and says whether the pixel is E[j], the 3-tap or the 2-tap value there: as
    // four bytes whose sum + 2 >> 2 is the pixel -- (a, b, b, c), (a, a, b, b), (a, a, a, a). The bytes of both pixels of a
    // pair lie inside one aligned 8-byte window of the edge array for every row (tests/test_abi.py::test_prediction_tables_fit_aligned_windows).
    const int p = k >> 4, m = k & 15;
    const int y = p >> 1;
    int by[2][4];
    bool zero = false;
    for (int e = 0; e < 2; e++) {
      const int x = 2 * (p & 1) + e;
      if (m == T4R_ZERO || m == T4R_DC) { zero = true; by[e][0] = by[e][1] = by[e][2] = by[e][3] = 0; }
      else if (m == T4R_DC_TOP) { for (int q = 0; q < 4; q++) by[e][q] = E_T0 + q; }
      else if (m == T4R_DC_LEFT) { for (int q = 0; q < 4; q++) by[e][q] = E_L3 + q; }
      else if (m == T4R_H16) { by[e][0] = by[e][1] = by[e][2] = by[e][3] = E_T0 + y; }
      else if (m == T4R_PRED16) { by[e][0] = by[e][1] = by[e][2] = by[e][3] = 4 * y + x; }
      else {
        const int mode = m < 9 ? m : (m == T4R_NOTR3 ? 3 : 7), jmax = m < 9 ? 12 : 8;
        const int en = P.t4[mode * 16 + y * 4 + x], j = en & 31, sel = en >> 5;
        const int ja = sel == 1 ? j - 1 : j, jb = sel == 2 ? j + 1 : j, jc = sel == 1 ? j + 1 : j;
        by[e][0] = E_L3 + min(max(ja, 0), jmax);
        by[e][1] = by[e][2] = E_L3 + min(max(jb, 0), jmax);
        by[e][3] = E_L3 + min(max(jc, 0), jmax);
      }
    }
    int lo = 15;
    for (int e = 0; e < 2; e++) for (int q = 0; q < 4; q++) lo = min(lo, by[e][q]);
    const int off = zero ? 0 : lo & ~3;
    unsigned sel[2];
    for (int e = 0; e < 2; e++) {
      sel[e] = 0;
      for (int q = 0; q < 4; q++) sel[e] |= (zero ? (m == T4R_DC ? 0x0du : 0x0cu) : (unsigned)((by[e][q] - off) & 7)) << (8 * q);
    }
    wv::lds_st128(ldsBase + T_T4W + T4W_PAIR * p + 16 * m, u32x4{(unsigned)off, sel[0], sel[1], 0u});
  }
  if (hasI8) {
    for (int k = tid; k < 384; k += nthreads) wv::lds_st16(ldsBase + T_LS8 + 2 * k, P.ls8[k]);
    for (int k = tid; k < 160; k += nthreads) {
      // entry of (table row m, lane i). P.t8 names a sample by its index j on the filtered edge [L7..L0 | corner | T0..T15]
      // (byte j of the block's edge array, S_E8) and says whether the pixel is E1[j], the 3-tap or the 2-tap value there
      const int m = k >> 4, i = k & 15, y = i >> 1, x0 = 4 * (i & 1);
      const bool zero = m == T8R_ZERO || m == 2;   // (DC: computed in the step)
      unsigned sel[4], off[2] = {0u, 0u};
      for (int pr = 0; pr < 2; pr++) {
        int by[2][4], lo = 31;
        for (int e = 0; e < 2; e++) {
          const int en = zero ? 0 : P.t8[m * 64 + y * 8 + x0 + 2 * pr + e], j = en & 31, sl = en >> 5;
          const int ja = sl == 1 ? j - 1 : j, jb = sl == 2 ? j + 1 : j, jc = sl == 1 ? j + 1 : j;
          by[e][0] = min(max(ja, 0), 24);
          by[e][1] = by[e][2] = min(max(jb, 0), 24);
          by[e][3] = min(max(jc, 0), 24);
          for (int q = 0; q < 4; q++) lo = min(lo, by[e][q]);
        }
        off[pr] = zero ? 0u : (unsigned)(lo & ~3);
        for (int e = 0; e < 2; e++) {
          unsigned v = 0;
          for (int q = 0; q < 4; q++) v |= (zero ? 0x0cu : (unsigned)((by[e][q] - (int)off[pr]) & 7)) << (8 * q);
          sel[2 * pr + e] = v;
        }
      }
      wv::lds_st128(ldsBase + T_T8S + T8S_ROW * m + 16 * i, u32x4{sel[0], sel[1], sel[2], sel[3]});
      wv::lds_st32(ldsBase + T_T8O + 64 * m + 4 * i, off[0] | (off[1] << 16));
    }
    for (int k = tid; k < 64; k += nthreads) wv::lds_st8(ldsBase + T_ZZ8 + P.zz8i[k], (unsigned)(2 * k));
    for (int k = tid; k < 192; k += nthreads) {
      const int m = k >> 5, p = (k >> 3) & 3, j = k & 7;
      wv::lds_st32(ldsBase + T_LS8P + 4 * k, (unsigned)P.ls8[64 * m + 16 * p + j] | ((unsigned)P.ls8[64 * m + 16 * p + 8 + j] << 16));
    }
    for (int k = tid; k < 64; k += nthreads) {
      const int r = k >> 3, j = k & 7;
      wv::lds_st8(ldsBase + T_ZZ8P + P.zz8i[k], (unsigned)(32 * (r >> 1) + 4 * j + 2 * (r & 1)));
    }
    for (int qp = tid; qp < 52; qp += nthreads) {
      // Exactness of the packed form. D = sum of |d| over the block's dequantised entries. Every value either 8-point pass
      // forms is a sum of its inputs with weights of magnitude <= 1.5 (idct8: d1 -> e7, d7 -> e1) plus < 4 from the
      // truncating shifts, so a row pass stays below 1.5 * (the row's share of D) + 4 and a column pass below
      // 1.5 * sum over the rows of that + 4 + 32 <= 2.25 D + 84: D <= 14400 keeps everything inside int16. With S = the
      // block's sum of |c| (>= any single |c|) and L = the largest LevelScale entry of qp % 6:
      //   qp >= 36: |d| = |c| LS << (qp/6 - 6)                        -> D <= S L << shl
      //   qp <  36: |d| <= (|c| LS + rnd) >> shr <= |c| LS / 2^shr + 1 -> D <= S L >> shr + 64 + 32 (rnd >> shr <= 1/2 per entry)
      // and the product itself, which v_pk_mad_u16 forms modulo 2^16 before the shift: S L + 32 <= 32767.
      const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
      int L = 1;
      for (int k = 0; k < 64; k++) L = max(L, (int)P.ls8[64 * qm + k]);
      const int byProduct = 32700 / L;
      const int byD = qd >= 6 ? 14400 / (L << (qd - 6)) : (int)min(((long long)(14400 - 96) << (6 - qd)) / L, 65535ll);
      wv::lds_st16(ldsBase + T_THR8P + 2 * qp, (unsigned)min(min(byProduct, byD), 65535));
    }
  }
}

// ---- 4x4 residual of one block, all in this lane (transform.rs:116-191, 8.5.12) --------------------------------
// c0/c1: the block's 16 list entries as packed int16 pairs; ls: LevelScale of qp%6 in list order, same packing.
// d = ((c * LS) << shl + rnd) >> shr  ==  transform.rs:147-152 with shl = max(qp/6-4,0), shr = max(4-qp/6,0),
// rnd = 2^(3-qp/6) below qp 24. useDc: element (0,0) arrives already scaled (Intra16x16 / chroma, :145-146).
// Out: residual [y][x] as 8 saturated int16 pairs (clip255(pred + r) cannot tell r from its clamp to int16).
template <typename T>
WV void idct4x4(const u32x4 c0, const u32x4 c1, const u32x4 l0, const u32x4 l1, int shl, int rnd, int shr, bool useDc,
                T dcVal, unsigned out[8]) {
  const unsigned cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const unsigned lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
  T d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int c = (k & 1) ? ((int)cw[k >> 1] >> 16) : (int)(int16_t)cw[k >> 1];
    const int l = (k & 1) ? (int)(lw[k >> 1] >> 16) : (int)(lw[k >> 1] & 0xffffu);
    d[k] = ((((T)(c * l)) * ((T)1 << shl)) + (T)rnd) >> shr;   // |c * l| < 2^15 * 7395 fits int32
  }
  if (useDc) d[0] = dcVal;
  d[0] += 32;  // the rounding term of :183-187: element (0,0) reaches every output with weight 1 and through no shift
  T f[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {  // row butterflies (transform.rs:159-169)
    const T m0 = d[ZZ4IDX(r, 0)], m1 = d[ZZ4IDX(r, 1)], m2 = d[ZZ4IDX(r, 2)], m3 = d[ZZ4IDX(r, 3)];
    const T e0 = m0 + m2, e1 = m0 - m2, e2 = (m1 >> 1) - m3, e3 = m1 + (m3 >> 1);
    f[r][0] = e0 + e3;
    f[r][1] = e1 + e2;
    f[r][2] = e1 - e2;
    f[r][3] = e0 - e3;
  }
  T h[4][4];
#pragma unroll
  for (int c = 0; c < 4; c++) {  // column butterflies (:171-181), rounding (:183-187)
    const T g0 = f[0][c] + f[2][c], g1 = f[0][c] - f[2][c], g2 = (f[1][c] >> 1) - f[3][c], g3 = f[1][c] + (f[3][c] >> 1);
    h[0][c] = (g0 + g3) >> 6;
    h[1][c] = (g1 + g2) >> 6;
    h[2][c] = (g1 - g2) >> 6;
    h[3][c] = (g0 - g3) >> 6;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (sizeof(T) == 4) {
      out[2 * r] = wv::cvt_pk_i16((int)h[r][0], (int)h[r][1]);
      out[2 * r + 1] = wv::cvt_pk_i16((int)h[r][2], (int)h[r][3]);
    } else {
      const T lo = (T)-32768, hi = (T)32767;
      const int a0 = (int)min(max(h[r][0], lo), hi), a1 = (int)min(max(h[r][1], lo), hi);
      const int a2 = (int)min(max(h[r][2], lo), hi), a3 = (int)min(max(h[r][3], lo), hi);
      out[2 * r] = ((unsigned)a0 & 0xffffu) | ((unsigned)a1 << 16);
      out[2 * r + 1] = ((unsigned)a2 & 0xffffu) | ((unsigned)a3 << 16);
    }
  }
}

// The same transform in 64-bit arithmetic (the reference's isize), for blocks whose intermediates may not fit int32.
// Taken by a whole wave when any of its lanes needs it; never for a conformant stream. Written for few live
// registers rather than speed: the column butterflies are done one column at a time and the row pass is
// recomputed for each, so only a handful of wide values are ever live besides the packed inputs.
WV void idct4x4_wide(const u32x4 c0, const u32x4 c1, int lsAddr, int shl, int rnd, int shr, bool useDc, long long dcVal,
                     unsigned out[8]) {
  typedef long long T;
  unsigned cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  auto dq = [&](int k) -> T {
    const int c = (k & 1) ? ((int)cw[k >> 1] >> 16) : (int)(int16_t)cw[k >> 1];
    const int l = (int)wv::lds_u16(lsAddr + 2 * k);  // (re-read per use: registers matter here, time does not)
    const T d = (T)(((unsigned long long)(T)(c * l) << shl) + (unsigned long long)rnd) >> shr;
    return (k == 0 && useDc) ? dcVal : d;
  };
  int hold[4] = {0, 0, 0, 0};
#pragma clang loop unroll(disable)
  for (int col = 0; col < 4; col++) {
    T fa[4];  // column `col` of the row-pass output
#pragma unroll
    for (int k = 0; k < 8; k++) cw[k] = (unsigned)wv::opaque((int)cw[k]);  // (keeps the dequantisation inside the loop)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const T m0 = dq(ZZ4IDX(r, 0)), m1 = dq(ZZ4IDX(r, 1)), m2 = dq(ZZ4IDX(r, 2)), m3 = dq(ZZ4IDX(r, 3));
      const T e0 = m0 + m2, e1 = m0 - m2, e2 = (m1 >> 1) - m3, e3 = m1 + (m3 >> 1);
      fa[r] = col == 0 ? e0 + e3 : col == 1 ? e1 + e2 : col == 2 ? e1 - e2 : e0 - e3;
    }
    const T g0 = fa[0] + fa[2], g1 = fa[0] - fa[2], g2 = (fa[1] >> 1) - fa[3], g3 = fa[1] + (fa[3] >> 1);
    const T lo = (T)-32768, hi = (T)32767;
    const int h0 = (int)min(max((g0 + g3 + 32) >> 6, lo), hi), h1 = (int)min(max((g1 + g2 + 32) >> 6, lo), hi);
    const int h2 = (int)min(max((g1 - g2 + 32) >> 6, lo), hi), h3 = (int)min(max((g0 - g3 + 32) >> 6, lo), hi);
    if ((col & 1) == 0) {
      hold[0] = h0; hold[1] = h1; hold[2] = h2; hold[3] = h3;
    } else {
      const unsigned o0 = ((unsigned)hold[0] & 0xffffu) | ((unsigned)h0 << 16), o1 = ((unsigned)hold[1] & 0xffffu) | ((unsigned)h1 << 16);
      const unsigned o2 = ((unsigned)hold[2] & 0xffffu) | ((unsigned)h2 << 16), o3 = ((unsigned)hold[3] & 0xffffu) | ((unsigned)h3 << 16);
      if (col == 1) { out[0] = o0; out[2] = o1; out[4] = o2; out[6] = o3; }
      else { out[1] = o0; out[3] = o1; out[5] = o2; out[7] = o3; }
    }
  }
}

// ---- the same transform on packed 16-bit pairs, for blocks that provably fit -----------------------------------
// Every value the two passes form is a sum of the dequantised entries with weights of magnitude <= 1 (|x >> 1| <= |x|),
// so nothing exceeds B = sum |d_k| + 32 <= sum |c_k| * max LS' + |dc| + 32 with LS' = LS << max(qp/6 - 4, 0): when
// B < 2^15 the wrap-around 16-bit arithmetic below is exact, and so are the 16-bit products c * LS' (+ rnd). A step
// whose 64 blocks all pass (block_fits16) takes this path; one block that does not sends the wave through idct4x4<int>.
// sum of |c| over the block's 16 entries (entry 0 excluded when it is not a coefficient of this block)
WV unsigned sum_abs16(const u32x4 c0, const u32x4 c1, bool skip0) {
  // On unsigned halves |c - 0x8000| = 32768 - |c| for either sign of the int16 c (c >= 0: 32768 - c; c < 0: its bits are
  // 65536 + c, so 32768 + c), hence eight v_sad_u16 against 0x8000 give 16 * 32768 - sum |c| without touching the inputs
  const unsigned b = 0x80008000u;
  unsigned a = wv::sad_u16(skip0 ? c0.x & 0xffff0000u : c0.x, b, 0u);
  a = wv::sad_u16(c0.y, b, a);
  a = wv::sad_u16(c0.z, b, a);
  a = wv::sad_u16(c0.w, b, a);
  a = wv::sad_u16(c1.x, b, a);
  a = wv::sad_u16(c1.y, b, a);
  a = wv::sad_u16(c1.z, b, a);
  return 16u * 32768u - wv::sad_u16(c1.w, b, a);
}
WV bool block_fits16(const u32x4 c0, const u32x4 c1, unsigned lsmax, bool useDc, int dc, bool dcHuge) {
  const unsigned sa = min(sum_abs16(c0, c1, useDc), 32768u);            // (<= 2^15: the product below fits 32 bits)
  unsigned bound = sa * lsmax;
  if (useDc) {
    if (dcHuge || dc > 32767 || dc < -32767) return false;
    bound += (unsigned)(dc < 0 ? -dc : dc);
  }
  return bound <= 32700u;
}
// entries za (low half) and zb (high half) of the list, as one pair
template <int ZA, int ZB>
WV unsigned zz_pair(const unsigned z[8]) {
  if ((ZA >> 1) == (ZB >> 1) && (ZA & 1) == 0 && (ZB & 1) == 1) return z[ZA >> 1];
  constexpr unsigned sel = (unsigned)(2 * (ZA & 1)) | ((unsigned)(2 * (ZA & 1) + 1) << 8) | ((unsigned)(4 + 2 * (ZB & 1)) << 16) |
                           ((unsigned)(5 + 2 * (ZB & 1)) << 24);
  return wv::perm(z[ZB >> 1], z[ZA >> 1], sel);
}
// one butterfly (transform.rs:159-181) on four pairs
WV void butterfly_pk(const unsigned m[4], unsigned f[4]) {
  const unsigned e0 = wv::pk_add(m[0], m[2]), e1 = wv::pk_sub(m[0], m[2]);
  const unsigned e2 = wv::pk_sub(wv::pk_ashr1(m[1]), m[3]), e3 = wv::pk_add(m[1], wv::pk_ashr1(m[3]));
  f[0] = wv::pk_add(e0, e3);
  f[1] = wv::pk_add(e1, e2);
  f[2] = wv::pk_sub(e1, e2);
  f[3] = wv::pk_sub(e0, e3);
}
// lq0/lq1: the qp's T_LS4Q row. rnd2 / shr2: rnd / shr of idct4x4 in both halves (0 from qp 24 up: the shift is in the table).
// inputsDone(): called as soon as the coefficient registers have been read for the last time (the caller's prefetch of the
// next coefficients goes there: the earlier it is issued, the more of the step it has to come back).
template <typename F>
WV void idct4x4_pk16(const u32x4 c0, const u32x4 c1, const u32x4 lq0, const u32x4 lq1, unsigned rnd2, unsigned shr2, bool useDc, int dcVal,
                     unsigned out[8], F&& inputsDone) {
  const unsigned cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const unsigned lw[8] = {lq0.x, lq0.y, lq0.z, lq0.w, lq1.x, lq1.y, lq1.z, lq1.w};
  unsigned z[8];
#pragma unroll
  for (int k = 0; k < 8; k++) z[k] = wv::pk_mad(cw[k], lw[k], rnd2);
  inputsDone();
  if (wv::any(shr2 != 0)) {
#pragma unroll
    for (int k = 0; k < 8; k++) z[k] = wv::pk_ashr(z[k], shr2);
  }
  if (useDc) z[0] = (z[0] & 0xffff0000u) | ((unsigned)dcVal & 0xffffu);
  z[0] = wv::pk_add(z[0], 32u);  // the rounding term of :183-187, through element (0,0) (see idct4x4)
  // rows (0 | 3) and (1 | 2) as pairs: two of the eight are list neighbours already
  const unsigned ra[4] = {zz_pair<ZZ4IDX(0, 0), ZZ4IDX(3, 0)>(z), zz_pair<ZZ4IDX(0, 1), ZZ4IDX(3, 1)>(z),
                          zz_pair<ZZ4IDX(0, 2), ZZ4IDX(3, 2)>(z), zz_pair<ZZ4IDX(0, 3), ZZ4IDX(3, 3)>(z)};
  const unsigned rb[4] = {zz_pair<ZZ4IDX(1, 0), ZZ4IDX(2, 0)>(z), zz_pair<ZZ4IDX(1, 1), ZZ4IDX(2, 1)>(z),
                          zz_pair<ZZ4IDX(1, 2), ZZ4IDX(2, 2)>(z), zz_pair<ZZ4IDX(1, 3), ZZ4IDX(2, 3)>(z)};
  unsigned fa[4], fb[4];   // fa[c] = f[0][c] | f[3][c] << 16, fb[c] = f[1][c] | f[2][c] << 16
  butterfly_pk(ra, fa);
  butterfly_pk(rb, fb);
#pragma unroll
  for (int q = 0; q < 2; q++) {  // columns 2q | 2q+1 as pairs
    const unsigned col[4] = {wv::perm(fa[2 * q + 1], fa[2 * q], 0x05040100u), wv::perm(fb[2 * q + 1], fb[2 * q], 0x05040100u),
                             wv::perm(fb[2 * q + 1], fb[2 * q], 0x07060302u), wv::perm(fa[2 * q + 1], fa[2 * q], 0x07060302u)};
    unsigned h[4];
    butterfly_pk(col, h);
#pragma unroll
    for (int r = 0; r < 4; r++) out[2 * r + q] = wv::pk_ashr6(h[r]);
  }
}

// largest |c| over the block's entries (entry 0 excluded when it is not a coefficient of this block); 32768 for -32768
WV int max_abs16(const u32x4 c0, const u32x4 c1, bool skip0) {
  const unsigned w0 = skip0 ? c0.x & 0xffff0000u : c0.x;
  // packed maxima and minima of the eight pairs, then max(hi, -lo) per half
  const unsigned mx = wv::pk_max(wv::pk_max(wv::pk_max(w0, c0.y), wv::pk_max(c0.z, c0.w)), wv::pk_max(wv::pk_max(c1.x, c1.y), wv::pk_max(c1.z, c1.w)));
  const unsigned mn = wv::pk_min(wv::pk_min(wv::pk_min(w0, c0.y), wv::pk_min(c0.z, c0.w)), wv::pk_min(wv::pk_min(c1.x, c1.y), wv::pk_min(c1.z, c1.w)));
  const int hi = max((int)(int16_t)(mx & 0xffffu), (int)mx >> 16), lo = min((int)(int16_t)(mn & 0xffffu), (int)mn >> 16);
  return max(hi, -lo);
}

// One pass of lane-per-block residuals. big = this lane's block may overflow int32. WIDE build: the whole wave then
// takes the 64-bit path (wave-uniform branch). Fast build: the batch is flagged (status bit 1) and the host re-runs it
// with the WIDE build before anything is reported -- the fast kernel carries no 64-bit code, which would cost it its
// register budget; no conformant stream ever takes this route.
// unused: this lane's result is not needed (an Intra8x8 macroblock's lanes): it must not keep the wave off the packed path.
// The DC term that replaces entry 0 (useDc): dc when it fits 32 bits with room to spare (|dc| <= 2^26), else dcHuge (the
// fast build then knows no more than that; the WIDE build carries the exact value in dcWide).
template <bool WIDE, typename F>
WV void residual_pass(const u32x4 c0, const u32x4 c1, int ldsBase, int qp, bool useDc, int dc, bool dcHuge, long long dcWide,
                      bool unused, unsigned* status, unsigned batchSeq, unsigned out[8], F&& inputsDone) {
  const u32x4 q = wv::lds_u128(ldsBase + T_QP + 32 * qp);   // everything the pass derives from qp (build_tables)
  if (!wv::any(!unused && !block_fits16(c0, c1, q.z & 0xffffu, useDc, dc, dcHuge))) {
    const u32x4 lq0 = wv::lds_u128(ldsBase + T_LS4Q + 32 * qp), lq1 = wv::lds_u128(ldsBase + T_LS4Q + 32 * qp + 16);
    idct4x4_pk16(c0, c1, lq0, lq1, q.x, q.y, useDc, dc, out, inputsDone);
    return;
  }
  const int thr = unused ? 0xFFFF : (int)(q.z >> 16);
  const int shl = (int)(q.w & 0xffu), rnd = (int)((q.w >> 8) & 0xffu), shr = (int)((q.w >> 16) & 0xffu);
  const int lsAddr = ldsBase + T_LS4Z + (int)(q.w >> 24);   // LevelScale of qp % 6, list order
  bool big = false;
  if (wv::any(thr != 0xFFFF)) big = thr != 0xFFFF && max_abs16(c0, c1, useDc) > thr;
  if (useDc && (dcHuge || dc > (1 << 26) || dc < -(1 << 26))) big = true;
  if (WIDE) {
    if (wv::any(big)) {
      idct4x4_wide(c0, c1, lsAddr, shl, rnd, shr, useDc, dcWide, out);
      inputsDone();
      return;
    }
  } else if (big) {
    wv::atomic_or(status, 2u);
    wv::atomic_max(status + 4, ~batchSeq);   // (the earliest flagged batch of a queue: the host re-runs from there)
  }
  const u32x4 l0 = wv::lds_u128(lsAddr), l1 = wv::lds_u128(lsAddr + 16);
  idct4x4<int>(c0, c1, l0, l1, shl, rnd, shr, useDc, dc, out);
  inputsDone();
}

// lane i ^ 4 and i ^ 8 inside a 16-lane DPP row
// (i ^ 4 = the quads of every eight lanes swapped = the eight mirrored, then every quad mirrored back: two DPP moves, the
// second of which the compiler folds into the instruction that uses it; no select)
WV int xor4(int v, bool) { return wv::dppx<DPP_QUAD(3, 2, 1, 0)>(wv::dppx<DPP_ROW_HALF_MIRROR>(v)); }
WV int xor8(int v) { return wv::dppx<DPP_ROW_ROR(8)>(v); }
WV int xor1(int v) { return wv::dppx<DPP_QUAD(1, 0, 3, 2)>(v); }
WV int xor2(int v) { return wv::dppx<DPP_QUAD(2, 3, 0, 1)>(v); }

// ---- 8x8 residuals of the step's Intra8x8 macroblocks (8.5.13, pred8x8.rs:51-150) ----------------------------------
// 8-point butterfly shared by the row and the column pass (pred8x8.rs:85-141)
template <typename T>
WV void idct8(const T d[8], T o[8]) {
  const T e0 = d[0] + d[4];
  const T e1 = -d[3] + d[5] - d[7] - (d[7] >> 1);
  const T e2 = d[0] - d[4];
  const T e3 = d[1] + d[7] - d[3] - (d[3] >> 1);
  const T e4 = (d[2] >> 1) - d[6];
  const T e5 = -d[1] + d[7] + d[5] + (d[5] >> 1);
  const T e6 = d[2] + (d[6] >> 1);
  const T e7 = d[3] + d[5] + d[1] + (d[1] >> 1);
  const T f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
  const T f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
  o[0] = f0 + f7;
  o[1] = f2 + f5;
  o[2] = f4 + f3;
  o[3] = f6 + f1;
  o[4] = f6 - f1;
  o[5] = f4 - f3;
  o[6] = f2 - f5;
  o[7] = f0 - f7;
}

// Two passes over the macroblock's four 8x8 blocks, two blocks per pass; the 16 lanes of a macroblock are (block, row),
// then (block, column), with the row-pass output transposed through LDS. In: the coefficients already in raster order
// at c8 (the record's coefficient area). Out: out[4 * pass + m] = rows 2m, 2m+1 of this lane's column as a saturated int16 pair.
template <typename T>
WV void residual8x8_passes(bool mine, int g, int i, int qp, int ldsBase, int ts, int c8, unsigned out[8]) {
  const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
  // qp >= 36: (c * LS) << (qp/6 - 6), else (c * LS + 2^(5 - qp/6)) >> (6 - qp/6)   (pred8x8.rs:73-78)
  const int shl = max(qd - 6, 0), shr = max(6 - qd, 0), rnd = qd < 6 ? (1 << (5 - qd)) : 0;
  const int g8 = ts + S_G8 + (int)sizeof(T) * (128 * g + 64 * (i >> 3));
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int b8 = 2 * p + (i >> 3), r = i & 7;
    T dd[8], oo[8];
    if (mine) {
      const u32x2 ca = wv::lds_u64(c8 + C8_MB * g + C8_BLK * b8 + 16 * r), cb = wv::lds_u64(c8 + C8_MB * g + C8_BLK * b8 + 16 * r + 8);
      const u32x4 lr = wv::lds_u128(ldsBase + T_LS8 + 128 * qm + 16 * r);
      const unsigned cw[4] = {ca.x, ca.y, cb.x, cb.y}, lw[4] = {lr.x, lr.y, lr.z, lr.w};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = (j & 1) ? ((int)cw[j >> 1] >> 16) : (int)(int16_t)cw[j >> 1];
        const int l = (j & 1) ? (int)(lw[j >> 1] >> 16) : (int)(lw[j >> 1] & 0xffffu);
        dd[j] = ((((T)(c * l)) * ((T)1 << shl)) + (T)rnd) >> shr;   // |c * l| <= 2^15 * 255 * 58 fits int32
      }
      idct8<T>(dd, oo);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (sizeof(T) == 4) wv::lds_st32(g8 + 4 * (8 * r + j), (unsigned)oo[j]);
        else wv::lds_st64(g8 + 8 * (8 * r + j), u32x2{(unsigned)(unsigned long long)oo[j], (unsigned)((unsigned long long)oo[j] >> 32)});
      }
    }
    wv::wave_sync();
    if (mine) {
      const int j = i & 7;  // this lane now owns column j of its block
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (sizeof(T) == 4) dd[k] = (T)(int)wv::lds_u32(g8 + 4 * (8 * k + j));
        else {
          const u32x2 v = wv::lds_u64(g8 + 8 * (8 * k + j));
          dd[k] = (T)(long long)(((unsigned long long)v.y << 32) | v.x);
        }
      }
      dd[0] += (T)32;   // the rounding term of (x + 32) >> 6: d0 enters every output of the 8-point transform once, unscaled
      idct8<T>(dd, oo);
#pragma unroll
      for (int m = 0; m < 4; m++) {
        if (sizeof(T) == 4) {
          out[4 * p + m] = wv::cvt_pk_i16((int)(oo[2 * m] >> 6), (int)(oo[2 * m + 1] >> 6));   // (saturating)
        } else {
          const T lo = (T)-32768, hi = (T)32767;
          const int a = (int)min(max(oo[2 * m] >> 6, lo), hi), b = (int)min(max(oo[2 * m + 1] >> 6, lo), hi);
          out[4 * p + m] = ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
        }
      }
    }
    wv::wave_sync();  // (S_G8 is reused by the next pass)
  }
}

// 8-point butterfly on pairs (two rows, or two columns, per register)
WV void idct8_pk(const unsigned d[8], unsigned o[8]) {
  using namespace wv;
  const unsigned e0 = pk_add(d[0], d[4]);
  const unsigned e1 = pk_sub(pk_sub(pk_sub(d[5], d[3]), d[7]), pk_ashr1(d[7]));
  const unsigned e2 = pk_sub(d[0], d[4]);
  const unsigned e3 = pk_sub(pk_sub(pk_add(d[1], d[7]), d[3]), pk_ashr1(d[3]));
  const unsigned e4 = pk_sub(pk_ashr1(d[2]), d[6]);
  const unsigned e5 = pk_add(pk_add(pk_sub(d[7], d[1]), d[5]), pk_ashr1(d[5]));
  const unsigned e6 = pk_add(d[2], pk_ashr1(d[6]));
  const unsigned e7 = pk_add(pk_add(pk_add(d[3], d[5]), d[1]), pk_ashr1(d[1]));
  const unsigned f0 = pk_add(e0, e6), f1 = pk_add(e1, pk_ashr2(e7)), f2 = pk_add(e2, e4), f3 = pk_add(e3, pk_ashr2(e5));
  const unsigned f4 = pk_sub(e2, e4), f5 = pk_sub(pk_ashr2(e3), e5), f6 = pk_sub(e0, e6), f7 = pk_sub(e7, pk_ashr2(e1));
  o[0] = pk_add(f0, f7);
  o[1] = pk_add(f2, f5);
  o[2] = pk_add(f4, f3);
  o[3] = pk_add(f6, f1);
  o[4] = pk_sub(f6, f1);
  o[5] = pk_sub(f4, f3);
  o[6] = pk_sub(f2, f5);
  o[7] = pk_sub(f0, f7);
}

// The packed 16-bit form (exact under T_THR8P's bound, which the caller has checked for every Intra8x8 block of the step):
// ONE pass over the macroblock's four blocks. Lane i of the macroblock: block i >> 2, rows 2p and 2p + 1 (p = i & 3) in
// the halves of a register, then columns 2p and 2p + 1. In: the coefficients at c8, pair-interleaved (T_ZZ8P). Out: out[k] =
// row k, columns 2p | 2p + 1 << 16 of block i >> 2.
WV void residual8x8_pk16(bool mine, int lane, int qp, int ldsBase, int ts, int c8, unsigned out[8]) {
  const int g = lane >> 4, i = lane & 15, p = i & 3;
  const int blk = c8 + C8_MB * g + C8_BLK * (i >> 2), g8 = ts + S_G8 + G8_MB * g + G8_BLK * (i >> 2);
  // S_G8: (row r, column pair q) of the block at dword 4 r + ((q + (r >> 1)) & 3)
  const int sw[4] = {4 * (p & 3), 4 * ((p + 1) & 3), 4 * ((p + 2) & 3), 4 * ((p + 3) & 3)};
  const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
  const unsigned shl2 = (unsigned)max(qd - 6, 0) * 0x10001u, shr2 = (unsigned)max(6 - qd, 0) * 0x10001u;
  const unsigned rnd2 = qd < 6 ? (0x10001u << (5 - qd)) : 0u;
  const bool anyShl = wv::any(mine && shl2 != 0u);   // (qp >= 42 only: shl or shr is 0)
  if (mine) {
    unsigned dd[8], oo[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const u32x2 ra = wv::lds_u64(blk + 32 * p + 16 * h), rb = wv::lds_u64(blk + 32 * p + 16 * h + 8);
      const u32x4 l = wv::lds_u128(ldsBase + T_LS8P + 128 * qm + 32 * p + 16 * h);
      const unsigned rw[4] = {ra.x, ra.y, rb.x, rb.y}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
      for (int j = 0; j < 4; j++) dd[4 * h + j] = wv::pk_ashr(wv::pk_mad(rw[j], lw[j], rnd2), shr2);
    }
    if (anyShl) {
#pragma unroll
      for (int j = 0; j < 8; j++) dd[j] = wv::pk_shl(dd[j], shl2);
    }
    idct8_pk(dd, oo);
#pragma unroll
    for (int j = 0; j < 8; j++) {   // rows 2p and 2p + 1 (both rotated by p), column j
      wv::lds_st16(g8 + 32 * p + sw[j >> 1] + 2 * (j & 1), oo[j]);
      wv::lds_st16(g8 + 32 * p + 16 + sw[j >> 1] + 2 * (j & 1), oo[j] >> 16);
    }
  }
  wv::wave_sync();
  if (mine) {
    unsigned dd[8], oo[8];
#pragma unroll
    for (int k = 0; k < 8; k++) dd[k] = wv::lds_u32(g8 + 16 * k + sw[k >> 1]);   // row k, column pair p
    dd[0] = wv::pk_add(dd[0], 0x00200020u);   // the rounding term of (x + 32) >> 6
    idct8_pk(dd, oo);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = wv::pk_ashr6(oo[k]);
  }
  wv::wave_sync();  // (the coefficient area becomes the residuals'; S_G8 is reused by the next step)
}

// The 8x8 residuals are FRONT's and BACK8's: FRONT, which holds the coefficients, checks them and scatters them into the
// record (its row group's part of S_RES) in the order the passes read; BACK8 runs the passes when it takes the record (it used to idle half of
// every step while FRONT, the wave every other one of the team waits for, carried them).
constexpr unsigned I8F_PK = 1u, I8F_WIDE = 2u;   // the step's Intra8x8 residuals: in the packed 16-bit form / in 64-bit arithmetic
// c0/c1: the 16 list entries this lane loaded: entries 16 * (i & 3) .. +15 of 8x8 block i >> 2. mine: this lane's macroblock
// is a valid Intra8x8 one. Overflow handling as in residual_pass (thr8: |d| <= 2^23 keeps both 8-point passes in int32).
// Returns the form the passes take (wave-uniform: I8F_*).
template <bool WIDE>
WV unsigned residual8x8_scatter(const u32x4 c0, const u32x4 c1, bool mine, int lane, int qp, int ldsBase, int c8, unsigned* status,
                                unsigned batchSeq) {
  const int g = lane >> 4, i = lane & 15;
  // the block's sum of |c| (four lanes hold a block's 64 entries) against the qp's bound of the packed form
  bool pk = false;
  if (DRYV_BAND_I8_PK16) {
    unsigned sa = sum_abs16(c0, c1, false);
    sa += (unsigned)xor1((int)sa);
    sa += (unsigned)xor2((int)sa);
    pk = !wv::any(mine && sa > wv::lds_u16(ldsBase + T_THR8P + 2 * qp));
  }
  const int thr = (int)wv::lds_u16(ldsBase + T_THR8 + 2 * qp);
  bool big = false;
  if (!pk && wv::any(mine && thr != 0xFFFF)) big = mine && thr != 0xFFFF && max_abs16(c0, c1, false) > thr;
  // list order -> raster (frame/mod.rs:212-284) resp. the packed form's pair-interleaved block: one 16-bit store per entry
  if (mine) {
    const int dst = c8 + C8_MB * g + C8_BLK * (i >> 2);
    const u32x4 zp = wv::lds_u128(ldsBase + (pk ? T_ZZ8P : T_ZZ8) + 16 * (i & 3));
    const unsigned zw[4] = {zp.x, zp.y, zp.z, zp.w}, cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int k = 0; k < 16; k++)
      wv::lds_st16(dst + (int)((zw[k >> 2] >> (8 * (k & 3))) & 0xffu), (k & 1) ? cw[k >> 1] >> 16 : cw[k >> 1]);
  }
  if (pk) return I8F_PK;
  if (WIDE) {
    if (wv::any(big)) return I8F_WIDE;
  } else if (big) {
    wv::atomic_or(status, 2u);
    wv::atomic_max(status + 4, ~batchSeq);
  }
  return 0u;
}
// BACK8's half: the passes over what FRONT left in the record. `out`: in residual8x8_pk16's layout for I8F_PK,
// else in residual8x8_passes'.
template <bool WIDE>
WV void residual8x8_run(unsigned form, bool mine, int lane, int qp, int ldsBase, int ts, int c8, unsigned out[8]) {
  if (form & I8F_PK) residual8x8_pk16(mine, lane, qp, ldsBase, ts, c8, out);
  else if (WIDE && (form & I8F_WIDE)) residual8x8_passes<long long>(mine, lane >> 4, lane & 15, qp, ldsBase, ts, c8, out);
  else residual8x8_passes<int>(mine, lane >> 4, lane & 15, qp, ldsBase, ts, c8, out);
}
// ... and the step's Intra8x8 residuals from the passes' registers into the record, [blkIdx][y][x] like FRONT's 4x4 ones.
// Element (row k, column j) of 8x8 block b8 goes to 4x4 block (bx, by) = (2 * (b8 & 1) + (j >> 2), 2 * (b8 >> 1) + (k >> 2)),
// position (k & 3, j & 3).
WV void residual8x8_store(unsigned form, int lane, int resBuf, const unsigned r[8]) {
  const int g = lane >> 4, i = lane & 15;
  if (form & I8F_PK) {
    // the packed form: r[k] = row k, columns 2q | 2q + 1 of 8x8 block b8 = i >> 2 (q = i & 3): one dword of 4x4 block
    // (2 * (b8 & 1) + (q >> 1), 2 * (b8 >> 1) + (k >> 2))
    const int b8 = i >> 2, q = i & 3;
    const int dst = resBuf + RES_ROW * g + 32 * (8 * (b8 >> 1) + 2 * (b8 & 1) + (q >> 1)) + 4 * (q & 1);
#pragma unroll
    for (int k = 0; k < 8; k++) wv::lds_st32(dst + 128 * (k >> 2) + 8 * (k & 3), r[k]);
  } else {
    // this lane holds column j = i & 7 of 8x8 blocks i >> 3 (r[0..3]) and 2 + (i >> 3) (r[4..7]), rows 2m, 2m + 1 per word
    const int j = i & 7;
    const int dst = resBuf + RES_ROW * g + 64 * (i >> 3) + 32 * (j >> 2) + 2 * (j & 3);
#pragma unroll
    for (int pk = 0; pk < 16; pk++) {
      const int p = pk >> 3, k = pk & 7;
      const unsigned w = r[4 * p + (k >> 1)];
      wv::lds_st16(dst + 256 * p + 128 * (k >> 2) + 8 * (k & 3), (k & 1) ? w >> 16 : w);
    }
  }
}


// One reconstructed row of 4 pixels: prediction as two u16 pairs, residual as two i16 pairs
WV unsigned recon_row(unsigned p01, unsigned p23, unsigned r01, unsigned r23) {
  const unsigned a = wv::sat_pk_u8(wv::pk_add_sat(p01, r01)), b = wv::sat_pk_u8(wv::pk_add_sat(p23, r23));
  return wv::perm(b, a, 0x05040100u);  // bytes 0, 1 of a, then bytes 0, 1 of b
}


// ---- geometry of a band task (wave-uniform) ----------------------------------------------------------------------
struct BandGeo {
  int b, f, r0, nR, gl, nSteps;
  bool hasAbove, hasBelow;
};
// Queue order: groups of TASK_GROUP consecutive bands; within a group frame by frame, band after band. A band's
// predecessor (same frame, band above) always has a smaller number, whatever the group size. 1 = band-major order (band 0
// of every frame, band 1 of every frame, ...), which ships: larger groups put more bands of fewer pictures in flight
// (deeper chains of bands following each other) and measured slower on the 300-picture batch: 1.65 ms (1), 1.69 (2),
// 1.75 (4), 2.01 (8), 2.70 (17 = picture after picture).
#ifndef DRYV_BAND_TASK_GROUP
#define DRYV_BAND_TASK_GROUP 1
#endif
constexpr int TASK_GROUP = DRYV_BAND_TASK_GROUP;
WV BandGeo band_geo(unsigned task, int nF, int W, int H) {
  BandGeo G;
  if (TASK_GROUP == 1) {
    G.b = (int)(task / (unsigned)nF);
    G.f = (int)(task - (unsigned)G.b * (unsigned)nF);
  } else {
    const unsigned nB = (unsigned)((H + 3) >> 2), perGroup = (unsigned)TASK_GROUP * (unsigned)nF;
    const unsigned g = task / perGroup, r = task - g * perGroup;
    const unsigned nb = min((unsigned)TASK_GROUP, nB - g * (unsigned)TASK_GROUP);  // bands in this group (the last may be short)
    const unsigned f = r / nb;
    G.f = (int)f;
    G.b = (int)(g * (unsigned)TASK_GROUP + (r - f * nb));
  }
  G.r0 = 4 * G.b;
  G.nR = min(4, H - G.r0);
  G.gl = G.nR - 1;  // the band's last row
  G.nSteps = W + 2 * (G.nR - 1);
  G.hasAbove = G.b > 0;
  G.hasBelow = G.r0 + G.nR < H;
  return G;
}

// Waits (polling LDS) until the team's flag word at `addr` equals `want` / has reached it.
WV void team_wait(int addr, unsigned want) {
  while ((unsigned)wv::rfl((int)wv::lds_u32(addr)) != want) wv::sleep_team();
}
WV void team_wait_ge(int addr, unsigned want) {
  while ((int)((unsigned)wv::rfl((int)wv::lds_u32(addr)) - want) < 0) wv::sleep_team();
}

constexpr int HAND_WORDS = 16;   // dwords per macroblock of the hand-off records (Args::handoff)
// The hand-off record of a macroblock of the band above: four lanes (li = 0..3) read one 8-byte granule each, {four pixels,
// tag} (luma: the bottom line's four quarters; chroma: Cb, Cb, Cr, Cr). `v`: what the lanes requested a step ago (or just
// now). Polls -- the lanes whose tag is not there yet again -- until every tag is (bounded: see SPIN_LIMIT).
WV unsigned long long await_granules(const unsigned long long* rec, unsigned long long v, bool fetchLane, unsigned tag, unsigned* status,
                                     unsigned task, int s, int lane) {
  unsigned spins = 0;
  while (wv::any(fetchLane && (unsigned)(v >> 32) != tag)) {
    wv::sleep_short();
    if (fetchLane && (unsigned)(v >> 32) != tag) v = wv::ld_sc1_64(rec);
    if (++spins > SPIN_LIMIT) {
      // every spin is bounded: a band above that never gets there is reported (status bit 2 + where), not waited for
      if (lane == 0) {
        wv::atomic_or(status, 4u);
        status[1] = task;
        status[2] = ((unsigned)s << 16) | 0xffffu;
        status[3] = (unsigned)(v >> 32);
      }
      break;
    }
  }
  return v;
}

// Polls progress words of the band above until they reach `need` (bounded: see SPIN_LIMIT). Lanes 0..31 read pa,
// lanes 32..63 pb (the same word twice where a wave follows only one).
WV unsigned poll_progress(const unsigned* pa, const unsigned* pb, unsigned tag, unsigned known, unsigned need, unsigned W, unsigned* status,
                          unsigned task, int s, int lane) {
  unsigned spins = 0;
  while (known < need) {
    const unsigned w = wv::ld_sc1((const unsigned*)((const uint8_t*)pa + (lane < 32 ? 0u : 4u * (unsigned)(pb - pa))));  // (all lanes: see the claim in band_front)
    const unsigned v = (w >> PROG_SHIFT) == tag ? w & ((1u << PROG_SHIFT) - 1u) : 0u;   // (a word of an earlier launch: nothing yet)
    known = min((unsigned)wv::rfl((int)v), (unsigned)wv::rdlane((int)v, 32));
    if (known < need) {
      wv::sleep_short();
      if (++spins > SPIN_LIMIT) {
        // every spin is bounded: a band above that never gets there is reported (status bit 2 + where), not waited for
        if (lane == 0) {
          wv::atomic_or(status, 4u);
          status[1] = task;
          status[2] = ((unsigned)s << 16) | need;
          status[3] = known;
        }
        known = W;
      }
    }
  }
  return known;
}

// ==================================================================================================================
// Prediction modes of a whole band, derived before its first step (8.3.1.1 / 8.3.2.1; pred4x4.rs:363-427,
// pred8x8.rs:698-764). Nothing in the derivation depends on a pixel, only on the records -- so it does not have to ride
// the 2:1 diagonal, where it cost seven DPP sweeps over the block grid per step (a fifth of FRONT's instructions).
//   * ONE LANE PER MACROBLOCK, 64 consecutive macroblocks of a row per pass. The lane walks its sixteen grid positions
//     in raster order with the modes in registers: neighbour A of a position is the register one to the left, B the
//     one above; only column 0 reaches into the macroblock to the left (the lane to the left: one DPP wave shift of
//     the packed right column) and row 0 into the macroblock above (the same lane, one row earlier: a register).
//   * The dependence on the lane to the left is resolved by relaxation: every lane starts from "the left macroblock's
//     right column is all DC", and the pass is repeated with the right columns the previous pass produced until no lane's
//     right column changes (then every lane has seen its final inputs). A change only travels on when it flips a
//     comparison, so two or three passes are the rule; the bound is one pass per lane.
//   * Out: one 32-byte mode record per macroblock in global memory (workspace), in exactly the form the steps consume:
//       bytes 0..19  the Intra4x4 chain's table rows x 16, [block half][chain round] (= S_MSEQ's layout; rows 9..11 =
//                    zero prediction of quirk Q4 / modes 3, 7 without a top-right block); an Intra8x8 macroblock:
//                    bytes 0..3 = the table rows of its four blocks x 16 (T_T8S: the modes, or T8R_ZERO for quirk Q4)
//       word 5       the record's first word (kind, Intra16x16 / chroma modes, qp), checked: an unsupported record is
//                    reported here (status bit 0) and reads as kind 3 / qp 0 from then on
//       word 6       chain rounds of the macroblock that have a DC-predicted block (bit t)
//       word 7       the raw modes of its bottom grid row (DC for any other macroblock kind): neighbour B of the row below
//     written through (sc1) and published per band by one progress word (progM: the last row's macroblocks whose records are
//     there; batch by batch for wide pictures, else W at the end) for the band below; the band's own
//     steps load them back one step ahead, like the records.
constexpr int MREC_WORDS = 8;
template <bool HAS_I8>
WV void band_modes(const KParams& P, const Args& A, const BandGeo& G, const unsigned task, const int ts, const uint8_t* mbsF,
                   unsigned* recF, const unsigned* upProgM, unsigned* myProgM) {
  const int lane = wv::lane_id();
  const int W = P.W;
  unsigned upKnownM = 0;   // macroblocks of the band above's last row whose records are known to be there
  const bool perBatch = W > 128;
  TLM(task, 0, TNOW());
  if (lane < 4) wv::lds_st32(ts + S_CARRYM + 4 * lane, 0x02020202u);
  wv::wave_sync();
  // One iteration = 64 macroblocks of one row: batch after batch, the band's rows inside a batch (row g's neighbour B
  // is the same lane one iteration earlier). (No prefetch across iterations: requesting the next record an iteration ahead
  // measured 1.9 % slower.)
  const int nR = G.nR, nIter = ((W + 63) >> 6) * nR;
  auto mb_of = [&](int it) -> unsigned {   // this lane's macroblock in iteration `it` (the last one of the row beyond it)
    const int xb = it / nR, gg = it - xb * nR;
    return (unsigned)((G.r0 + gg) * W + min(64 * xb + lane, W - 1));
  };
  auto load_top = [&](int it) -> unsigned {  // row 0's neighbour B: the band above's bottom grid row
    const unsigned mbn = mb_of(it);
    return (G.hasAbove && it % nR == 0) ? wv::ld_sc1(recF + (size_t)MREC_WORDS * (mbn - (unsigned)W) + 7) : 0x02020202u;
  };
  unsigned bottom = 0x02020202u;  // raw modes of the bottom grid row of the macroblock above (same lane, previous iteration)
#pragma clang loop unroll(disable)
  for (int it = 0; it < nIter; it++) {
    const int xb = it / nR, g = it - xb * nR, x0 = 64 * xb;
    // Batch by batch: the band above publishes its last row's records after every batch, and this band needs those of a
    // batch only when it gets there -- the pre-passes of a picture's bands overlap instead of queueing up behind each other
    // (pictures of three batches or more; narrower ones publish once: 120 macroblocks wide, the second publication and its
    // drain cost 1 % and overlap nothing; 240 wide, four batches: -1.7 %)
    if (G.hasAbove && g == 0) {
      const unsigned before = upKnownM;
      upKnownM = poll_progress(upProgM, upProgM, A.gen & PROG_GEN_MASK, upKnownM, (unsigned)(perBatch ? min(x0 + 64, W) : W), (unsigned)W, A.status, task, -1, lane);
      // (the records behind this flag are read with sc1 loads, which MI355X_MICROARCH.md measured in
      // place of an acquire for one workgroup per CU only -- this kernel runs 2 to 5: so the consumer's form is the guide's
      // "always" one, ONE relaxed poll, ONE agent acquire, the wait for it, then the loads; once per band, or per batch of
      // wide pictures: not measurable in the launch (tools/ab_inproc.py: 300 x 1080p -1.1 %, 100 x 4K +0.1 %))
      if (DRYV_BAND_MODES_ACQUIRE && upKnownM != before) wv::acquire_agent();
    }
    if (it == 0) TLM(task, 1, TNOW());
    const int x = x0 + lane;
    const bool valid = x < W;
    const bool xIs0 = x == 0, xLast = x + 1 >= W;
    {
      const int r = G.r0 + g;
      const bool rowTop = r > 0;  // (wave-uniform) macroblock B exists
      const unsigned mb = mb_of(it);
      const u32x4 d = wv::ld_u128_a2(mbsF + 16u * mb);
      const unsigned topM = g == 0 ? load_top(it) : bottom;
      const unsigned carry = wv::lds_u32(ts + S_CARRYM + 4 * g);
      // the record checks of the step (an unsupported record reconstructs as zero and counts as DC for its neighbours)
      int kind = (int)(d.x & 0xffu);
      unsigned word0 = d.x;   // kind | Intra16x16 mode << 8 | chroma mode << 16 | qp << 24: what the steps read (MREC word 5)
      if (kind > 2 || (!HAS_I8 && kind == 1) || (d.x >> 24) > 51u || ((d.x >> 8) & 0xffu) > 3u || ((d.x >> 16) & 0xffu) > 3u) {
        if (valid) wv::atomic_or(A.status, 1u);
        kind = 3;
        word0 = 3u;   // (kind 3, qp 0)
      }
      const bool isI4 = kind == 0, is8 = HAS_I8 && kind == 1;
      // rem fields as nibbles in blkIdx order, bit 3 of a nibble set where prev_intra*_pred_mode_flag is: such a block
      // takes the predicted mode, and its "rem" of 8..15 is not below any mode
      const unsigned prevFlags = d.y & 0xffffu;
      auto spread8 = [](unsigned p) -> unsigned {  // bit k of p -> bit 4 * k + 3
        unsigned v = p & 0xffu;
        v = (v | (v << 12)) & 0x000f000fu;
        v = (v | (v << 6)) & 0x03030303u;
        v = (v | (v << 3)) & 0x11111111u;
        return v << 3;
      };
      const unsigned remLo = (wv::alignbit(d.z, d.y, 16) & 0x77777777u) | spread8(prevFlags);
      const unsigned remHi = (wv::alignbit(d.w, d.z, 16) & 0x77777777u) | spread8(prevFlags >> 8);
      int M[16];
#pragma unroll
      for (int b = 0; b < 16; b++) M[b] = 2;
      unsigned rc = 0x02020202u;
      for (int pass = 0; pass < 66; pass++) {
        // the right column of the macroblock to the left as of the previous pass; lane 0: of the previous batch (final)
        const unsigned lc = (unsigned)wv::dpp<DPP_WAVE_SHR1>((int)carry, (int)rc);
#pragma unroll
        for (int b = 0; b < 16; b++) {
          const int bx = b & 3, by = b >> 2;
          const int An = bx ? M[b - 1] : (int)((lc >> (8 * by)) & 0xffu);
          const int Bn = by ? M[b - 4] : (int)((topM >> (8 * bx)) & 0xffu);
          int pm = min(An, Bn);
          if (bx == 0) pm = xIs0 ? 2 : pm;       // dcPredModePredictedFlag: no macroblock A
          if (by == 0) pm = rowTop ? pm : 2;     // ... no macroblock B
          const int z = zidx(bx, by);
          int rm = (int)(((z < 8 ? remLo : remHi) >> (4 * (z & 7))) & 15u);
          if (HAS_I8) {
            // an 8x8 block: its top-left position derives the mode like a 4x4 block there would, from the block's own
            // flag and rem field; the other three positions copy it
            const int z8 = 2 * (by >> 1) + (bx >> 1);
            if (is8) rm = (int)((remLo >> (4 * z8)) & 15u);
          }
          int m = rm > 7 ? pm : (rm < pm ? rm : rm + 1);
          if (HAS_I8 && ((bx | by) & 1)) m = is8 ? M[(b & ~1) & ~4] : m;
          M[b] = m;
        }
        unsigned rcN = (unsigned)M[3] | ((unsigned)M[7] << 8) | ((unsigned)M[11] << 16) | ((unsigned)M[15] << 24);
        if (!(isI4 || is8)) rcN = 0x02020202u;
        const bool changed = valid && rcN != rc;
        rc = rcN;
        if (!wv::any(changed)) break;
      }
      // the batch's last macroblock is the next batch's macroblock A (every lane has converged: final)
      if (lane == 63) wv::lds_st32(ts + S_CARRYM + 4 * g, rc);
      bottom = (isI4 || is8) ? ((unsigned)M[12] | ((unsigned)M[13] << 8) | ((unsigned)M[14] << 16) | ((unsigned)M[15] << 24)) : 0x02020202u;

      // ---- the record: table rows of the chain (T4R_*), 16 x the row = the byte offset of its entry
      unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, dcMask = 0;
      if (wv::any(isI4)) {
        int T[16];
#pragma unroll
        for (int b = 0; b < 16; b++) T[b] = M[b];
        // modes 3 and 7 without a top-right block: T4..T7 := T3 (table rows 10 and 11). Inside the macroblock positions
        // (1,1) (3,1) (3,2) (1,3) (3,3) never have one; the top row has the neighbours' (B, B, B, C)
        auto no_tr = [](int m) -> int { return (m & 3) == 3 ? T4R_NOTR3 + (m >> 2) : m; };
        T[5] = no_tr(T[5]); T[7] = no_tr(T[7]); T[11] = no_tr(T[11]); T[13] = no_tr(T[13]); T[15] = no_tr(T[15]);
        if (!rowTop) { T[0] = no_tr(T[0]); T[1] = no_tr(T[1]); T[2] = no_tr(T[2]); }
        T[3] = (rowTop && !xLast) ? T[3] : no_tr(T[3]);
        if (!rowTop || x0 == 0) {
          // picture edges only. Quirk Q4: a mode whose reference samples are missing leaves the zero-initialised prediction
          // (row T4R_ZERO); DC with one neighbour or none has rows of its own (none: the left column reads 128)
#pragma unroll
          for (int b = 0; b < 16; b++) {
            const int bx = b & 3, by = b >> 2;
            if (bx != 0 && by != 0) continue;
            const bool topAv = by > 0 || rowTop, leftAv = bx > 0 || !xIs0;
            const int have = (topAv ? 1 : 0) | (leftAv ? 2 : 0) | ((topAv && leftAv) ? 4 : 0);
            const int req = (int)((0x217771021ull >> (4 * M[b])) & 7ull);  // per mode: bit0 top, bit1 left, bit2 corner
            if ((req & ~have) != 0) T[b] = T4R_ZERO;
            if (M[b] == 2 && !(topAv && leftAv)) T[b] = topAv ? T4R_DC_TOP : T4R_DC_LEFT;
          }
        }
        // x 16, in the order [block half][chain round]: round t = bx + 2 by; the first half is the block with the smaller by
        auto pk = [](int a, int b2, int c, int e) -> unsigned {
          return ((unsigned)a << 4) | ((unsigned)b2 << 12) | ((unsigned)c << 20) | ((unsigned)e << 28);
        };
        w0 = pk(T[0], T[1], T[2], T[3]);            // rounds 0..3: (0,0) (1,0) (2,0) (3,0)
        w1 = pk(T[6], T[7], T[10], T[11]);          // rounds 4..7: (2,1) (3,1) (2,2) (3,2)
        w2 = pk(T[14], T[15], 0, 0);                // rounds 8, 9: (2,3) (3,3)
        w3 = pk(0, 0, T[4], T[5]);                  // second half, rounds 2, 3: (0,1) (1,1)
        w4 = pk(T[8], T[9], T[12], T[13]);          // ... rounds 4..7: (0,2) (1,2) (0,3) (1,3)
        // rounds with a block whose value is the block's DC: bytes equal to 16 * T4R_DC (exact per byte: no carry between bytes)
        auto dc4 = [](unsigned w) -> unsigned {
          const unsigned v = w ^ (0x01010101u * (16u * T4R_DC));
          return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u;
        };
        auto gather = [](unsigned zz) -> unsigned {  // bits 7, 15, 23, 31 -> bits 0..3
          unsigned q = zz >> 7;
          q |= q >> 7;
          return (q | (q >> 14)) & 15u;
        };
        dcMask = gather(dc4(w0) | (dc4(w3) & 0x80800000u)) | (gather(dc4(w1) | dc4(w4)) << 4) | (gather(dc4(w2) & 0x00008080u) << 8);
      }
      if (!isI4) {
        // every other kind: one row for all sixteen blocks. Intra16x16 rides the block chain on edge arrays that band_back
        // fills from the macroblock's own neighbours (vertical / horizontal: the top line / left column; DC and plane: the
        // predicted pixels themselves); without the samples a mode needs, and for an unsupported record: zero prediction
        const int m16 = (int)((word0 >> 8) & 3u);
        int r16 = T4R_ZERO;
        if (kind == 2) r16 = m16 == 0 ? (rowTop ? 0 : T4R_ZERO) : m16 == 1 ? (xIs0 ? T4R_ZERO : T4R_H16) : (m16 == 2 || (rowTop && !xIs0)) ? T4R_PRED16 : T4R_ZERO;
        w0 = w1 = w2 = w3 = w4 = 0x10101010u * (unsigned)r16;
        dcMask = 0u;
      }
      if (HAS_I8 && wv::any(is8)) {
        // an Intra8x8 macroblock: the table rows of its four blocks (T_T8S), x 16 like the others
        int R[4] = {M[0], M[2], M[8], M[10]};
        if (!rowTop || x0 == 0) {
          // picture edges only: quirk Q4 as above (a directional mode without its samples: zero prediction)
#pragma unroll
          for (int b8 = 0; b8 < 4; b8++) {
            const bool topAv = (b8 >> 1) > 0 || rowTop, leftAv = (b8 & 1) > 0 || !xIs0;
            const int have = (topAv ? 1 : 0) | (leftAv ? 2 : 0) | ((topAv && leftAv) ? 4 : 0);
            const int req = (int)((0x217771021ull >> (4 * R[b8])) & 7ull);
            if ((req & ~have) != 0) R[b8] = T8R_ZERO;
          }
        }
        if (is8) w0 = ((unsigned)R[0] << 4) | ((unsigned)R[1] << 12) | ((unsigned)R[2] << 20) | ((unsigned)R[3] << 28);
      }
      {
        // The 64 records of the iteration are 2 KB of contiguous memory: through LDS, so that each of the two store
        // instructions writes whole lines -- lane by lane (16 of every 32 bytes per instruction) every write-through store
        // reaches memory as a partial line of its own: 1.4 M write requests more per 300 pictures, and 3 % of the launch
        const int stg = ts + mrec_off(HAS_I8);
        unsigned* rec0 = recF + (size_t)MREC_WORDS * (unsigned)((G.r0 + g) * W + x0);   // the batch's first record
        unsigned* const dst = rec0 + 4 * lane;
        // lane l: 16-byte chunks l and 64 + l of the batch (chunk c = half c & 1 of its record c >> 1)
        if (mrec_bytes(HAS_I8) >= 2048) {
          wv::lds_st128(stg + 32 * lane, u32x4{w0, w1, w2, w3});
          wv::lds_st128(stg + 32 * lane + 16, u32x4{w4, word0, dcMask, bottom});
          wv::wave_sync();
          const u32x4 va = wv::lds_u128(stg + 16 * lane), vb = wv::lds_u128(stg + 1024 + 16 * lane);
          if (x0 + (lane >> 1) < W) wv::st_g128_sc1(dst, va);
          if (x0 + 32 + (lane >> 1) < W) wv::st_g128_sc1(dst + 256, vb);
        } else {
          // (a staging area of 1 KB: the records of lanes 0..31, then those of lanes 32..63; measures the same)
          if (lane < 32) {
            wv::lds_st128(stg + 32 * lane, u32x4{w0, w1, w2, w3});
            wv::lds_st128(stg + 32 * lane + 16, u32x4{w4, word0, dcMask, bottom});
          }
          wv::wave_sync();
          const u32x4 va = wv::lds_u128(stg + 16 * lane);
          wv::wave_sync();
          if (lane >= 32) {
            wv::lds_st128(stg + 32 * (lane - 32), u32x4{w0, w1, w2, w3});
            wv::lds_st128(stg + 32 * (lane - 32) + 16, u32x4{w4, word0, dcMask, bottom});
          }
          wv::wave_sync();
          const u32x4 vb = wv::lds_u128(stg + 16 * lane);
          if (x0 + (lane >> 1) < W) wv::st_g128_sc1(dst, va);
          if (x0 + 32 + (lane >> 1) < W) wv::st_g128_sc1(dst + 256, vb);
        }
      }
    }
    wv::wave_sync();  // (the records' staging area; lane 63's right columns: the next batch's macroblock A)
    // a batch's records of the band's last row are complete once their stores have been written through (the last batch's
    // count is W: FRONT of this band's team waits for the LDS flag, the band below for this word)
    if (G.hasBelow && g == nR - 1 && (perBatch || it == nIter - 1)) {
      wv::wait_vm(0);
      if (lane == 0) wv::st_sc1(myProgM, ((A.gen & PROG_GEN_MASK) << PROG_SHIFT) | (unsigned)min(x0 + 64, W));
    }
  }
  wv::wait_vm(0);
  TLM(task, 2, TNOW());
}

// ==================================================================================================================
// FRONT wave: records, luma residuals, mode derivation, the band above's luma lines for BACK
// ==================================================================================================================
template <bool HAS_I8, bool WIDE>
WV void band_front(const KParams& P, const Args& A, const int ldsBase, const int ts) {
  const int lane0 = wv::lane_id();
  BAND_DIAG_BEGIN();
  const int W = P.W, H = P.H, nF = P.n_frames;
  const int nBands = (H + 3) >> 2;
  const unsigned totalTasks = (unsigned)nF * (unsigned)nBands;
  const int pitchY = W * 16;
  const size_t frameBytes = (size_t)W * H * 384;
  unsigned gstep = 0;  // steps of this team so far, over all its tasks: buffer = parity, flags carry gstep + 1
  // FRONT is the wave at the register limit: its lane roles are recomputed in every step, except these, which save
  // the most arithmetic per register (kept behind an optimisation barrier)
  const int hi4 = lane0 & 15, hzbx = ((hi4 >> 1) & 2) | (hi4 & 1), hzby = ((hi4 >> 2) & 2) | ((hi4 >> 1) & 1);
  const int hPermDc = wv::opaque((lane0 & 48) + zidx((0x1320 >> (4 * hzbx)) & 3, (0x1320 >> (4 * hzby)) & 3));  // Intra16x16 DC: source lane of the last stage
  const int hResOff = wv::opaque(RES_ROW * (lane0 >> 4) + 32 * (4 * hzby + hzbx));                                   // this lane's block in the residual record

  // Claims the task of sequence number q of this team and hands it to CHROMA (a ring of four task numbers).
  // Every lane takes part in the claim (lane 0 adds 1, the others 0) and in the progress-word loads further down:
  // a single-lane conditional in front of a readfirstlane invites the compiler to thread that condition through the
  // loop, after which the readfirstlane executes under a partial exec mask and returns another lane's value.
  auto claim_push = [&](unsigned q) -> unsigned {
    const unsigned tsk = wv::atomic_add_task(A.taskCounter, lane0 == 0 ? 1u : 0u);
    unsigned t = (unsigned)wv::rfl((int)tsk) - A.taskBase;
    if (t >= totalTasks) t = TASK_END;
    if (q >= 4) team_wait_ge(ts + S_FLAGS + F_TAILC, q - 3);
    if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_TASKS + 4 * (int)(q & 3u), t);
    wv::wave_sync();
    if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_HEAD, q + 1);
    return t;
  };
  // A task is claimed CLAIM_AHEAD steps before the previous one ends, so that CHROMA, which follows the tasks through the
  // team's ring, has the next one when it gets there (12 / 24 / 48 / 96 steps measure the same). (Any unfinished task
  // with the smallest number is some team's current one and waits only for smaller ones: no deadlock.)
#ifndef DRYV_BAND_CLAIM_AHEAD
#define DRYV_BAND_CLAIM_AHEAD 24
#endif
  unsigned nextTask = claim_push(0u);
  for (unsigned seq = 0;; seq++) {
    const unsigned task = nextTask;
    bool claimedNext = false;
    if (task == TASK_END) {
      // tell BACK to stop: an end record in the next buffer
      const int buf = (int)(gstep % (unsigned)NBUF);
      if (gstep >= (unsigned)NBUF) team_wait(ts + S_FLAGS + F_FREE + 4 * buf, gstep - NBUF + 1);
      if (lane0 == 0) wv::lds_st32(ts + S_INFO + 32 * buf + 16, TASK_END);
      wv::wave_sync();
      if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_READY + 4 * buf, gstep + 1);
      break;
    }
    TRACE(0, task + 1u);
    TLINE(task, 0, TNOW());
    TLINE(task, 3, A.waveBase + (int)(threadIdx.x >> 6));
    const BandGeo G = band_geo(task, nF, W, H);
    const int r0 = G.r0, nR = G.nR, nSteps = G.nSteps;
    const bool hasAbove = G.hasAbove;
    // per-frame bases (wave-uniform); everything below addresses them with 32-bit offsets: a frame's planes, records
    // and coefficients are each < 4 GB (the host API checks)
    const uint8_t* const planeY = A.yuv + (size_t)G.f * frameBytes;
    const size_t mbFrame = (size_t)G.f * (size_t)(W * H);
    const uint8_t* const coefF = (const uint8_t*)(A.coeffs + mbFrame * 384);
    const unsigned* const recF = A.rowModes + mbFrame * MREC_WORDS;   // the frame's mode records
    // the band above's hand-off records (bands that have a band below: nBands - 1 per frame)
    const unsigned* const handUp = A.handoff + ((size_t)G.f * (nBands - 1) + (G.b - 1)) * (size_t)W * HAND_WORDS;
    const int claimStep = max(nSteps - DRYV_BAND_CLAIM_AHEAD, 0);

    // ---- the band's prediction modes, all of them, before its first step (band_modes), by this wave: CHROMA, which derived
    // them until the end of round 3, is the wave that finishes a task last (tools/modes_timeline.py: FRONT stood here 55 us
    // per task, 13 % of it, most of the time before CHROMA had even begun); the pre-pass itself is 16 us
    TLM(task, 3, TNOW());
    if (DRYV_BAND_MODES_IN_FRONT(HAS_I8)) {
      unsigned* const myProgM = A.progM + (size_t)G.f * nBands + G.b;
      wv::setprio<DRYV_BAND_PRIO_MODES>();
      EXP_REP(10) band_modes<HAS_I8>(P, A, G, task, ts, (const uint8_t*)(A.mbs + mbFrame), A.rowModes + mbFrame * MREC_WORDS, myProgM - 1, myProgM);
      wv::setprio<(HAS_I8 ? DRYV_BAND_PRIO_FRONT_I8 : DRYV_BAND_PRIO_FRONT)>();
      if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_MODES, seq + 1);   // (CHROMA reads the records too)
    } else {
      team_wait_ge(ts + S_FLAGS + F_MODES, seq + 1);
    }
    PH(6);  // mode pre-pass (or the wait for it)

    // ---- software pipeline: a step's record is fetched one step ahead (its first word, which decides the
    // coefficient layout, two steps ahead); its coefficients are fetched right after the previous step's residual
    // pass, into the registers that pass has just freed. Lane roles are recomputed from an opaque lane id wherever
    // they are needed: kept live across the step they would cost more registers than the few VALU they take.
    auto mb_index = [&](int step, int g) -> unsigned {  // within the frame
      const int x = wv::clamp3(step - 2 * g, 0, W - 1);
      return (unsigned)(min(r0 + g, H - 1) * W + x);
    };
    // (the record's first word comes checked from the mode record: band_modes)
    auto load_kind = [&](int step) -> unsigned {
      const int l = lane0;
      return wv::ld_sc1(recF + (MREC_WORDS * mb_index(step, l >> 4) + 5u));
    };
    // word i (0..7) of the mode record of row g's macroblock, on lanes 16 g + i (i < 8)
    auto load_rec = [&](int step) -> unsigned {
      const int l = lane0;
      return wv::ld_sc1(recF + (MREC_WORDS * mb_index(step, l >> 4) + (unsigned)(l & 7)));
    };
    u32x4 cA0, cA1;
    int dcA;
    auto load_coefs_luma = [&](int step, unsigned d0) {
      const int l = lane0;
      const int i = l & 15;
      const int zbx = ((i >> 1) & 2) | (i & 1), zby = ((i >> 2) & 2) | ((i >> 1) & 1);
      const int kind = (int)(d0 & 3u);
      const unsigned mo = mb_index(step, l >> 4) * 768u;
      // Intra16x16: [DC 16][blk x AC 15]; the lane takes the 16 entries that END with its block's 15 AC, so list
      // position k (1..15) is AC k-1 (pred16x16.rs:33-46) and entry 0 is replaced by the DC term
      const int off = kind == 2 ? 30 * (i + 1) : 32 * i;
      cA0 = wv::ld_u128_a2(coefF + (mo + (unsigned)off));
      cA1 = wv::ld_u128_a2(coefF + (mo + (unsigned)off + 16u));
      dcA = *(const int16_t*)(coefF + (mo + 2u * (unsigned)ZZ4IDX(zby, zbx)));
    };
    unsigned kN1 = load_kind(0);
    unsigned kN2 = load_kind(1);
    unsigned mN1 = load_rec(0);
    load_coefs_luma(0, kN1);

    PH(0);  // claim, prologue loads
    unsigned long long lineN = 0;   // this lane's granule of the band above's hand-off record of macroblock s+1, requested during the previous step

    for (int s = 0; s < nSteps; s++, gstep++) {
      TRACE(1, s + 1);
      // (the same feedback for this wave: BACK's priority while BACK has nothing of this wave's left to work on)
      if (DRYV_BAND_FRONT_LEAD(HAS_I8) >= 0) {
        const unsigned bs = max(wv::lds_u32(ts + S_FLAGS + F_FREE), wv::lds_u32(ts + S_FLAGS + F_FREE + 4));
        if ((int)(gstep - (unsigned)wv::rfl((int)bs)) < DRYV_BAND_FRONT_LEAD(HAS_I8)) wv::setprio<DRYV_BAND_PRIO_BACK>();
        else wv::setprio<(HAS_I8 ? DRYV_BAND_PRIO_FRONT_I8 : DRYV_BAND_PRIO_FRONT)>();
      }
      if (s == claimStep) {
        nextTask = claim_push(seq + 1);
        claimedNext = true;
      }
      const int buf = (int)(gstep % (unsigned)NBUF);
      const unsigned kCur = kN1;  // first record word of this step's macroblock (checked: band_modes)
      const unsigned mCur = mN1;  // this lane's word of the step's mode record
      kN1 = kN2;                  // ... of step s+1
      // lane roles (see the pipeline comment above)
      const int lane = wv::opaque(lane0);
      const int g = lane >> 4, i = lane & 15;
      const int zbx = ((i >> 1) & 2) | (i & 1), zby = ((i >> 2) & 2) | ((i >> 1) & 1);  // lane-per-block: blkIdx i (z-order)
      const bool rowOk = g < nR;
      const int x = s - 2 * g;
      const bool valid = rowOk && (unsigned)x < (unsigned)W;
      const bool needUp = hasAbove && s < W;
      unsigned lineV = 0;

      // ---- record decode (lane-per-block luma organisation: row g) ------------------------------------------
      const int kind = (int)(kCur & 3u), i16mode = (int)((kCur >> 8) & 3u), qp = (int)(kCur >> 24);
      // the mode record of step s+1, the first record word of step s+2: requested now, a whole step (two) before they are
      // needed
      kN2 = load_kind(s + 2);
      mN1 = load_rec(s + 1);

      PH(1);  // record decode
      // ================= residuals ================================================================================
      unsigned rA[8];
      unsigned form8 = 0;   // (wave-uniform) the form BACK8's passes over the step's Intra8x8 coefficients take (I8F_*)
      EXP_REP(0)
      if (EXP_SKIP(0)) {
#pragma unroll
        for (int k = 0; k < 8; k++) rA[k] = cA0.x;
      } else {
        // Intra16x16 luma DC: 8.5.10 (pred16x16.rs:428-482). Lane (bx,by) holds c[by][bx]; f = A c A = P (H c H) P^T
        // with H the natural-order Hadamard (butterflies over the lane bits) and A row k = H row s(k), s = [0,2,3,1]
        int dcY = 0;
        bool dcHuge = false;
        long long dcWide = 0;
        if (wv::any(kind == 2)) {
          int v = dcA;
          int o = xor1(v);
          v = (zbx & 1) ? o - v : v + o;
          o = xor4(v, (i & 4) != 0);
          v = (zbx & 2) ? o - v : v + o;
          o = xor2(v);
          v = (zby & 1) ? o - v : v + o;
          o = xor8(v);
          v = (zby & 2) ? o - v : v + o;
          const int fv = wv::bperm(v, hPermDc);  // lane (s(bx), s(by)) of the row group, s = [0, 2, 3, 1]
          // qp >= 36: (f*LS) << (qp/6-6), else (f*LS + 2^(5-qp/6)) >> (6-qp/6)   (pred16x16.rs:465-479)
          const u32x4 qq = wv::lds_u128(ldsBase + T_QP + 32 * qp + 16);
          const int shlY = (int)(qq.w & 0xffu), shrY = (int)(qq.w >> 8);
          if (WIDE) {
            const long long prod = (long long)fv * (int)qq.x;
            dcWide = (long long)(((unsigned long long)prod << shlY) + (unsigned long long)qq.z) >> shrY;
            dcHuge = dcWide > (1ll << 26) || dcWide < -(1ll << 26);
            dcY = (int)dcWide;
          } else {
            // |f| < 2^16, LevelScale(0,0) < 2^13 and shl <= 2 keep the product inside 32 bits; a larger f (no conformant
            // stream has one) flags the batch for the WIDE build
            dcHuge = (unsigned)(fv + 65535) > 131070u;
            dcY = (int)(((unsigned)(fv * (int)qq.x) << shlY) + qq.z) >> shrY;
          }
        }
        // (an Intra8x8 lane's result of this pass is not used: it must not raise the 4x4 overflow flag either)
        // (the next step's coefficients are requested from inside the pass, as soon as this step's have been read. The Intra8x8
        // pass below reads them again: such a stream requests them behind it)
        const u32x4 cC0 = cA0, cC1 = cA1;
        residual_pass<WIDE>(cC0, cC1, ldsBase, qp, kind == 2, dcY, kind == 2 && dcHuge, dcWide, HAS_I8 && kind == 1, A.status, A.batchSeq, rA,
                            [&]() { if (!HAS_I8 && !EXP_DUP_IS(0)) load_coefs_luma(s + 1, kN1); });
        // Intra8x8 macroblocks: checked here, where the coefficients are, and scattered into the record (the macroblock's part of S_RES) for BACK8,
        // which runs the 8x8 passes (the buffer goes with the record: it is free once BACK has freed the record two back)
        if (HAS_I8 && wv::any(valid && kind == 1)) {
          if (gstep >= (unsigned)NBUF) team_wait(ts + S_FLAGS + F_FREE + 4 * buf, gstep - NBUF + 1);
          form8 = residual8x8_scatter<WIDE>(cC0, cC1, valid && kind == 1, lane, qp, ldsBase, ts + S_RES + RES_BUF * buf, A.status, A.batchSeq);
        }
        if (wv::any(kind == 3)) {  // (an unsupported record: reconstructs as zero)
#pragma unroll
          for (int k = 0; k < 8; k++) rA[k] = kind == 3 ? 0u : rA[k];
        }
      }
      PH(2);  // luma residuals
      // ---- hand-off traffic, placed here so that nothing in front of the residuals waits for it.
      // lanes 0..3: the four luma granules of macroblock s+1 of the band above's last row; lanes 16..19: of macroblock 0 at
      // step 0. A granule carries its own tag: no progress word, no ordering between the band above's stores.
      // Macroblock s+2 is requested now and looked at in the next step: the request has a whole step to come back.
      const int li = lane & 15;
      const int mbx = lane < 16 ? s + 1 : 0;
      const bool fetchLane = needUp && li < 4 && (lane < 16 ? s + 1 < W : (lane < 32 && s == 0));
      if (needUp && !EXP_SKIP(2)) {
        const unsigned long long* const rec = (const unsigned long long*)handUp + ((HAND_WORDS / 2) * mbx + li);
        unsigned long long g64 = 0;
        if (fetchLane) g64 = (s > 0 && lane < 16) ? lineN : wv::ld_sc1_64(rec);
        g64 = await_granules(rec, g64, fetchLane, A.gen, A.status, task, s, lane);
        lineV = (unsigned)g64;
        if (lane < 4 && s + 2 < W) lineN = wv::ld_sc1_64(rec + HAND_WORDS / 2);
      }
      if (HAS_I8 || EXP_SKIP(0) || EXP_DUP_IS(0)) load_coefs_luma(s + 1, kN1);  // (otherwise: requested inside the residual pass)
      PH(3);  // hand-off traffic, coefficient prefetch

      // ---- the step's record for BACK: residuals [blkIdx][y][x], table rows, kinds. The buffer is free once BACK has
      // finished the step two back.
      if (gstep >= (unsigned)NBUF) team_wait(ts + S_FLAGS + F_FREE + 4 * buf, gstep - NBUF + 1);
      if (!(HAS_I8 && kind == 1)) {   // (an Intra8x8 macroblock's residuals: BACK8's)
        const int dst = ts + S_RES + RES_BUF * buf + hResOff;
        wv::lds_st128(dst, u32x4{rA[0], rA[1], rA[2], rA[3]});
        wv::lds_st128(dst + 16, u32x4{rA[4], rA[5], rA[6], rA[7]});
      }
      // kind | Intra16x16 mode << 8 | qp << 16 | form of the step's 8x8 passes << 24
      if (i == 0) wv::lds_st32(ts + S_INFO + 32 * buf + 4 * g, (unsigned)kind | ((unsigned)i16mode << 8) | ((unsigned)qp << 16) | (form8 << 24));
      if (lane == 0) {
        wv::lds_st32(ts + S_INFO + 32 * buf + 16, task);
        wv::lds_st32(ts + S_INFO + 32 * buf + 20, (unsigned)s);
        wv::lds_st32(ts + S_INFO + 32 * buf + 24, seq & 1u);
        wv::lds_st32(ts + S_INFO + 32 * buf + 28, 0u);  // rounds of the block chain that have a DC block: below
      }
      // what was fetched from the band above goes into row 0's luma ring (BACK's)
      if (fetchLane) wv::lds_st32(ringy(ts, 0, mbx, (int)(seq & 1u)) + 4 * li, lineV);
      wv::wave_sync();
      // the step's modes from the band's pre-pass (band_modes): words 0..5 of the macroblock's mode record are the table
      // rows of BACK's block chain (S_MSEQ), word 6 the chain rounds that have a DC block -- BACK skips the DC arithmetic
      // (and its samples) in the rounds where no block of the step is predicted DC
      if (valid && !EXP_SKIP(1)) {
        if (i < 6) wv::lds_st32(ts + S_MSEQ + 96 * buf + 24 * g + 4 * i, mCur);
        if (i == 6 && kind == 0) wv::lds_or32(ts + S_INFO + 32 * buf + 28, mCur);
      }
      wv::wave_sync();
      if (lane == 0) wv::lds_st32(ts + S_FLAGS + F_READY + 4 * buf, gstep + 1);  // the record is complete
      PH(4);  // record for BACK
    }
    if (!claimedNext) nextTask = claim_push(seq + 1);
    TLE(task, 0, TNOW());
    TRACE(6, task + 1u);
  }
  TRACE(7, 0xD0E);
  BAND_DIAG_END();
}

// ==================================================================================================================
// CHROMA wave: chroma residuals, prediction, write-out and hand-off, following the tasks FRONT claims
// ==================================================================================================================
template <bool HAS_I8, bool WIDE>
WV void band_chroma(const KParams& P, const Args& A, const int ldsBase, const int ts) {
  const int lane0 = wv::lane_id();
  BAND_DIAG_BEGIN();
  const int W = P.W, H = P.H, nF = P.n_frames;
  const int nBands = (H + 3) >> 2;
  const int pitchC = W * 8;
  const size_t frameBytes = (size_t)W * H * 384;
  const unsigned offCb = (unsigned)W * H * 256u, offCr = offCb + (unsigned)W * H * 64u;
  // Write-out addresses: their lane-dependent parts never change, so they are computed once per kernel and kept (behind an
  // optimisation barrier: the compiler would otherwise rebuild them from the lane id in every step).
  const int hg = lane0 >> 4, hi = lane0 & 15;
  const int aBot = wv::opaque(ts + S_STC + 16 * CW * hg + 8 * CW * (hi >> 1) + CW * 7 + 4 * (hi & 1));  // + 8 * slot: bottom line of a plane, lanes 0..3
  const int aRing = wv::opaque(ts + S_RINGC + RINGC_ROW * (hg + 1) + 4 * hi);                          // + RINGC_ENT * (x & 3)
  const int aLeftRd = wv::opaque(ts + S_STC + 16 * CW * hg + 8 * CW * (hi >> 3) + CW * (hi & 7) + 7);  // + 8 * slot: column 7
  const int aLeftWr = wv::opaque(ts + S_LEFTC + 16 * hg + 8 * (hi >> 3) + (hi & 7));
  // write-through of the band's bottom lines: + 8 * r0 * pitchC + 8 * s   (8 * x = 8 * s - 16 * g)
  // flush with 64 lanes per macroblock row (NSC = 8): lane = (plane, pixel row, 16-byte segment)
  constexpr int FLR = 8 * NSC;
  const int fw = lane0 % FLR, fpl = fw / (FLR / 2), ffy = (fw / (NSC / 2)) & 7, fseg = fw % (NSC / 2);
  const int aFlush = wv::opaque(ts + S_STC + 8 * CW * fpl + CW * ffy + 16 * fseg);                      // + 16 * CW * row
  const unsigned oFlush = (unsigned)wv::opaque((int)((fpl ? offCr : offCb) + (unsigned)(ffy * pitchC + 16 * fseg)));  // + 8 * (r0 + row) * pitchC + 8 * xp
  const int fseg2 = wv::opaque(2 * fseg);   // 2 * segment
  const unsigned selH0 = (unsigned)wv::opaque((int)(0x0c040c04u + 0x00010001u * (unsigned)(2 * (lane0 >> 5))));   // byte 2 * half of the source in both halves of a pair
  // the band above's bottom lines (lanes 0..1 of a row group Cb, 2..3 Cr): + (8 * r0 - 1) * pitchC + 8 * macroblock

  unsigned gstepC = 0;   // this wave's steps so far (BACK's count of them is in the team's F_FREE words)
  for (unsigned seq = 0;; seq++) {
    team_wait_ge(ts + S_FLAGS + F_HEAD, seq + 1);
    const unsigned task = (unsigned)wv::rfl((int)wv::lds_u32(ts + S_FLAGS + F_TASKS + 4 * (int)(seq & 3u)));
    wv::wave_sync();
    if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_TAILC, seq + 1);
    if (task == TASK_END) break;
    const BandGeo G = band_geo(task, nF, W, H);
    const int r0 = G.r0, nR = G.nR, gl = G.gl, nSteps = G.nSteps;
    const bool hasAbove = G.hasAbove, hasBelow = G.hasBelow;
    uint8_t* const planeY = A.yuv + (size_t)G.f * frameBytes;
    const size_t mbFrame = (size_t)G.f * (size_t)(W * H);
    const uint8_t* const mbsF = (const uint8_t*)(A.mbs + mbFrame);
    const uint8_t* const coefF = (const uint8_t*)(A.coeffs + mbFrame * 384);
    // hand-off records: this band's (if it has a band below) and the band above's
    unsigned* const handMy = A.handoff + ((size_t)G.f * (nBands - 1) + G.b) * (size_t)W * HAND_WORDS;
    const unsigned* const handUp = handMy - (size_t)W * HAND_WORDS;
    unsigned* const recF = A.rowModes + mbFrame * MREC_WORDS;   // the frame's mode records

    // ---- the band's prediction modes, all of them, before its first step: FRONT hands a task over well before the luma
    // waves get to it, and this wave, ahead of them inside a task, has the time (band_modes)
    if (DRYV_BAND_MODES_IN_FRONT(HAS_I8)) {
      team_wait_ge(ts + S_FLAGS + F_MODES, seq + 1);   // (FRONT derives them)
    } else {
      unsigned* const myProgM = A.progM + (size_t)G.f * nBands + G.b;
      wv::setprio<DRYV_BAND_PRIO_MODES>();
      EXP_REP(10) band_modes<HAS_I8>(P, A, G, task, ts, mbsF, recF, myProgM - 1, myProgM);
      wv::setprio<DRYV_BAND_PRIO_CHROMA>();
      if (lane0 == 0) wv::lds_st32(ts + S_FLAGS + F_MODES, seq + 1);
    }
    PH(5);  // mode pre-pass (or the wait for it)
    TLE(task, 3, TNOW());

    // Residuals are computed for two steps at a time, lanes 0..31 the even step's 32 blocks, lanes 32..63 the odd step's
    // (a residual pass costs the same for 32 lanes as for 64); prediction then runs on the half whose step it is.
    // Software pipeline as in FRONT: first record word and coefficients one pair of steps ahead.
    auto mb_index = [&](int step, int g) -> unsigned {  // within the frame
      const int x = wv::clamp3(step - 2 * g, 0, W - 1);
      return (unsigned)(min(r0 + g, H - 1) * W + x);
    };
    // (the record's first word, checked, from the mode record: band_modes)
    auto load_kind = [&](int step) -> unsigned {
      const int l = lane0;
      return wv::ld_sc1(recF + (MREC_WORDS * mb_index(step + (l >> 5), (l >> 3) & 3) + 5u));
    };
    u32x4 cB0, cB1;
    int dcB;
    cB0 = cB1 = u32x4{0, 0, 0, 0};
    dcB = 0;
    auto load_coefs_chroma = [&](int step) {  // `step` even: the pair (step, step + 1)
      const int l = lane0;
      const int cpl = (l >> 2) & 1, cblk = l & 3;
      const unsigned mo = mb_index(step + (l >> 5), (l >> 3) & 3) * 768u;
      const unsigned offc = 2u * (unsigned)(259 + 64 * cpl + 15 * cblk);  // one before the block's 15 AC (trans_chroma.rs:43-58)
      cB0 = wv::ld_u128_a2(coefF + (mo + offc));
      cB1 = wv::ld_u128_a2(coefF + (mo + offc + 16u));
      dcB = *(const int16_t*)(coefF + (mo + 2u * (unsigned)(256 + 64 * cpl + cblk)));
    };
    unsigned kN1 = load_kind(0);  // first record word of this lane's macroblock in the next pair of steps
    unsigned dC = 0;              // ... in the current pair
    unsigned dEven = 0, dOdd = 0; // ... of the even / the odd step of the current pair, on all 64 lanes
    unsigned rB[8];               // this lane's block residual (current pair)
#pragma unroll
    for (int k = 0; k < 8; k++) rB[k] = 0;
    load_coefs_chroma(0);
    PH(0);  // task, prologue loads
    unsigned long long lineN = 0;   // this lane's granule of the band above's hand-off record of macroblock s+1, requested during the previous step

    for (int s = 0; s < nSteps; s++, gstepC++) {
      const bool evenStep = (s & 1) == 0;
      // This wave is the one that finishes a task last (tools/ends_timeline.py: 60 ... 90 us behind BACK, and the launch ends
      // when the last CHROMA does). Priority by how far behind BACK it is (BACK's step count: the record it freed last):
      // one above FRONT's when it trails by more than DRYV_BAND_CHROMA_TRAIL steps, FRONT's otherwise.
      if (DRYV_BAND_CHROMA_TRAIL >= 0 && evenStep) {
        const unsigned bs = max(wv::lds_u32(ts + S_FLAGS + F_FREE), wv::lds_u32(ts + S_FLAGS + F_FREE + 4));
        if ((int)((unsigned)wv::rfl((int)bs) - gstepC) > DRYV_BAND_CHROMA_TRAIL) wv::setprio<DRYV_BAND_PRIO_BACK>();
        else wv::setprio<DRYV_BAND_PRIO_CHROMA>();
      }
      if (evenStep) {
        dC = kN1;   // (checked by band_modes: an unsupported record reads as kind 3 / qp 0 and reconstructs as zero)
      }
      const int lane = wv::opaque(lane0);
      const int g = lane >> 4, i = lane & 15;                                            // write-out organisation: row g
      // residuals: lane = (half of the pair of steps, row gc, plane, block): lanes 32..63 work on the odd step's blocks.
      // prediction: lane = (half of the block, row gc, plane, block) of THIS step: lanes 32..63 predict rows 2, 3 of the
      // block whose rows 0, 1 lane - 32 predicts (8 pixels per lane on all 64 lanes)
      // (row and plane from the plain lane id: what depends on them alone is computed once per kernel; with the block
      // and the half as well the kernel would spill five registers -- it would still be 0.6 % faster, 252 instead of
      // 259 vector instructions per macroblock, but the shipped build keeps to zero spills)
      const int gc = (lane0 >> 3) & 3, cpl = (lane0 >> 2) & 1, cblk = lane & 3;
      const int ccx = cblk & 1, ccy = cblk >> 1, half = lane >> 5;
      const int rC = r0 + gc;
      const bool mbBC = rC > 0;
      const int x = s - 2 * g, xC = s - 2 * gc;
      const bool valid = g < nR && (unsigned)x < (unsigned)W, validC = gc < nR && (unsigned)xC < (unsigned)W;
      const bool mbAC = xC > 0;
      const int slot = x & (NSC - 1), slotC = xC & (NSC - 1);  // staging columns
      const bool needUp = hasAbove && s < W;
      unsigned lineV = 0;
      // dC: the residual lane's macroblock (of its step of the pair); dEven / dOdd: the prediction lane's (of this step)
      const int kindR = (int)(dC & 3u), qpR = (int)(dC >> 24);

      // ---- hand-off traffic. lanes 0..3: the four chroma granules (Cb, Cb, Cr, Cr) of macroblock s+1 of the band above;
      // lanes 16..19: of macroblock 0 at step 0; macroblock s+2 is requested a step early (as in FRONT)
      const int li = lane & 15;
      const int mbx = lane < 16 ? s + 1 : 0;
      const bool fetchLane = needUp && li < 4 && (lane < 16 ? s + 1 < W : (lane < 32 && s == 0));
      if (needUp && !EXP_SKIP(9)) {
        const unsigned long long* const rec = (const unsigned long long*)handUp + ((HAND_WORDS / 2) * mbx + 4 + li);
        unsigned long long g64 = 0;
        if (fetchLane) g64 = (s > 0 && lane < 16) ? lineN : wv::ld_sc1_64(rec);
        g64 = await_granules(rec, g64, fetchLane, A.gen, A.status, task, s, lane);
        lineV = (unsigned)g64;
        if (lane < 4 && s + 2 < W) lineN = wv::ld_sc1_64(rec + HAND_WORDS / 2);
      }
      PH(1);  // hand-off traffic

      // ================= chroma residuals ==========================================================================
      if (evenStep && EXP_SKIP(6)) {
        load_coefs_chroma(s + 2);
        kN1 = load_kind(s + 2);
      } else if (evenStep) {
        // chroma DC 2x2 (8.5.11, trans_chroma.rs:369-415) over the four block lanes of a plane, then the AC pass.
        // The LevelScale table is luma's (quirk Q3).
        const int qc = (int)wv::lds_u8(ldsBase + T_QPC + 52 * cpl + qpR);
        int v = dcB;
        int o = xor1(v);
        v = (cblk & 1) ? o - v : v + o;
        o = xor2(v);
        v = (cblk & 2) ? o - v : v + o;
        const u32x2 qq = wv::lds_u64(ldsBase + T_QP + 32 * qc + 16);   // LevelScale(0,0), qp / 6
        // ((f * LS) << (qp/6)) >> 5   (trans_chroma.rs:413)
        int dcC;
        bool dcHuge;
        long long dcWide = 0;
        if (WIDE) {
          dcWide = (((long long)v * (int)qq.x) * (1ll << qq.y)) >> 5;
          dcHuge = dcWide > (1ll << 26) || dcWide < -(1ll << 26);
          dcC = (int)dcWide;
        } else {
          // |f| < 2^(17 - qp/6) and LevelScale(0,0) < 2^13 keep the shifted product inside 32 bits (as for Intra16x16 DC)
          const unsigned lim = 1u << (17u - qq.y);
          dcHuge = (unsigned)(v + (int)lim - 1) > 2u * lim - 2u;
          dcC = (int)((unsigned)(v * (int)qq.x) << qq.y) >> 5;
        }
        EXP_REP(6) residual_pass<WIDE>(cB0, cB1, ldsBase, qc, true, dcC, dcHuge, dcWide, false, A.status, A.batchSeq, rB, []() {});
        if (wv::any(kindR == 3)) {
#pragma unroll
          for (int k = 0; k < 8; k++) rB[k] = kindR == 3 ? 0u : rB[k];
        }
        // hand the halves over (v_permlane32_swap): lanes 0..31 then hold rows 0, 1 of their even-step block in rB[0..3]
        // and of the odd-step block in rB[4..7], lanes 32..63 rows 2, 3 of the same two blocks; likewise the record words
#pragma unroll
        for (int k = 0; k < 4; k++) wv::swap32(rB[k], rB[k + 4]);
        dEven = dOdd = dC;
        wv::swap32(dEven, dOdd);
        load_coefs_chroma(s + 2);
        kN1 = load_kind(s + 2);
      }
      if (fetchLane) wv::lds_st32(ts + S_RINGC + RINGC_ENT * (mbx & 3) + 4 * li, lineV);
      wv::wave_sync();
      PH(2);  // chroma residuals, prefetch

      // ================= chroma: 8.3.4 (trans_chroma.rs:96-366), lane = (block half, row gc, plane, block) =======
      EXP_REP(7)
      if (!EXP_SKIP(7)) {
        const unsigned dP = evenStep ? dEven : dOdd;
        const int kindC = (int)(dP & 3u), cmode = (int)((dP >> 16) & 3u);
        const int ringP = ts + S_RINGC + RINGC_ROW * gc + 8 * cpl;
        const int leftC = ts + S_LEFTC + 16 * gc + 8 * cpl;
        const unsigned tw = wv::lds_u32(ringP + RINGC_ENT * (xC & 3) + 4 * ccx);
        const unsigned lw = wv::lds_u32(leftC + 4 * ccy);
        // horizontal: the left column (zero without neighbour A: quirk Q4); DC: below; plane: further below
        unsigned src = (cmode == 1 && mbAC) ? lw : 0u;
        if (wv::any(cmode == 0)) {
          // trans_chroma.rs:168-286 incl. quirk Q2 (`> 0` where the spec means "available"):
          //   blocks (0,0),(4,4): both -> 8-sample mean; left only -> left; top only needs every top sample > 0
          //   block (4,0): top, else left if its 4th sample > 0;  block (0,4): left if its 4th sample > 0, else top if ...
          const int st2 = (int)wv::sum4(tw, 2u), sl2 = (int)wv::sum4(lw, 2u);
          const int vT = st2 >> 2, vL = sl2 >> 2;
          int v;
          if (!wv::any(validC && cmode == 0 && !(mbAC && mbBC))) {
            // every DC macroblock of the step has both neighbours (all but the picture's first column and row)
            const int vBL = (lw >> 24) != 0 ? vL : (tw >> 24) != 0 ? vT : 128;
            v = ccx == ccy ? (st2 + sl2) >> 3 : ccx == 1 ? vT : vBL;
          } else {
            const bool nz = (((tw - 0x01010101u) & ~tw) & 0x80808080u) == 0u;
            const bool tAll = mbBC && nz, t3 = mbBC && (tw >> 24) != 0, l3 = mbAC && (lw >> 24) != 0;
            const int vDiag = (mbAC && mbBC) ? (st2 + sl2) >> 3 : mbAC ? vL : tAll ? vT : 128;
            const int vTR = mbBC ? vT : l3 ? vL : 128;
            const int vBL = l3 ? vL : t3 ? vT : 128;
            v = ccx == ccy ? vDiag : ccx == 1 ? vTR : vBL;
          }
          if (cmode == 0) src = (unsigned)v * 0x01010101u;
        }
        // vertical: the top line, pixel pairs (0, 1) and (2, 3) for both of this lane's rows; horizontal and DC: byte y of the
        // source for both pairs of row y = 2 * half + k (one v_perm_b32 per pixel pair either way)
        const bool isV = cmode == 2;
        if (isV) src = mbBC ? tw : 0u;
        unsigned p01[2], p23[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          p01[k] = wv::perm(src, src, isV ? 0x0c010c00u : selH0 + 0x00010001u * k);
          p23[k] = wv::perm(src, src, isV ? 0x0c030c02u : selH0 + 0x00010001u * k);
        }
        if (wv::any(validC && kindC != 3 && cmode == 3)) {
          // plane (:319-363): H = sum (k+1)(T[4+k] - T[2-k]), V likewise on the left column; T[-1] = L[-1] = corner: every lane
          // forms both from the whole top line and left column of its plane, as two byte dot products each
          const u32x2 tt = wv::lds_u64(ringP + RINGC_ENT * (xC & 3)), ll = wv::lds_u64(leftC);
          const unsigned c4 = 4u * wv::lds_u8(ringP + RINGC_ENT * ((xC - 1) & 3) + 7);
          const int hs = (int)wv::dot4(tt.y, 0x04030201u, 0u) - (int)wv::dot4(tt.x, 0x00010203u, c4);
          const int vs = (int)wv::dot4(ll.y, 0x04030201u, 0u) - (int)wv::dot4(ll.x, 0x00010203u, c4);
          if (cmode == 3 && mbAC && mbBC) {
            const int a = 16 * (int)((ll.y >> 24) + (tt.y >> 24));
            const int bq = (34 * hs + 32) >> 6, c = (34 * vs + 32) >> 6;
            // pixel (x, y) of the plane: clip255((a + b (x-3) + c (y-3) + 16) >> 5); all terms fit 16 bits
            const int base = a + bq * (4 * ccx - 3) + c * (4 * ccy + 2 * half - 3) + 16;
            const unsigned b01 = ((unsigned)base & 0xffffu) | ((unsigned)(base + bq) << 16);
            const unsigned step2 = ((unsigned)(2 * bq) & 0xffffu) | ((unsigned)(2 * bq) << 16);
            const unsigned cc = ((unsigned)c & 0xffffu) | ((unsigned)c << 16);
            unsigned q01 = b01, q23 = wv::pk_add(b01, step2);
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
              const unsigned u01 = wv::sat_pk_u8(wv::pk_ashr5(q01)), u23 = wv::sat_pk_u8(wv::pk_ashr5(q23));
              p01[kk] = wv::perm(0u, u01, 0x0c010c00u);
              p23[kk] = wv::perm(0u, u23, 0x0c010c00u);
              q01 = wv::pk_add(q01, cc);
              q23 = wv::pk_add(q23, cc);
            }
          }
        }
        if (wv::any(kindC == 3)) {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            p01[k] = kindC == 3 ? 0u : p01[k];
            p23[k] = kindC == 3 ? 0u : p23[k];
          }
        }
        {
          const int st = ts + S_STC + 16 * CW * gc + 8 * CW * cpl + CW * (4 * ccy + 2 * half) + 8 * slotC + 4 * ccx;
          if (evenStep) {
#pragma unroll
            for (int k = 0; k < 2; k++) wv::lds_st32(st + CW * k, recon_row(p01[k], p23[k], rB[2 * k], rB[2 * k + 1]));
          } else {
#pragma unroll
            for (int k = 0; k < 2; k++) wv::lds_st32(st + CW * k, recon_row(p01[k], p23[k], rB[4 + 2 * k], rB[5 + 2 * k]));
          }
        }
      }

      wv::wave_sync();
      PH(3);  // chroma prediction

      // ================= chroma write-out =========================================================================
      // The prefetched record word and coefficients are "used" here, in front of this step's stores: the compiler then
      // waits for those loads now (they were issued thousands of cycles ago) instead of at the top of the next step,
      // where its vmcnt(0) would also wait for the stores below -- and a write-through store takes thousands of cycles
      // to be acknowledged when the chip is busy. (Measured alternative, not kept: a constant number of stores per step,
      // padded with stores to a dump area, so that the compiler could count them: its waits stayed vmcnt(0) across the
      // loop back-edge and the batch took longer.)
      kN1 = (unsigned)wv::opaque((int)kN1);
      cB0.x = (unsigned)wv::opaque((int)cB0.x); cB1.x = (unsigned)wv::opaque((int)cB1.x); dcB = wv::opaque(dcB);
      // bottom chroma lines for the row below (ring) or the band below (write-through): lanes 0..1 of row g Cb, 2..3 Cr
      EXP_REP(12) {
        unsigned v = 0;
        if (i < 4) v = wv::lds_u32(aBot + 8 * slot);
        if (valid && i < 4 && g < 3 && g < gl) wv::lds_st32(aRing + RINGC_ENT * (x & 3), v);
        if (hasBelow && wv::any(valid && g == gl)) {
          // the band's last row: a granule {4 pixels, tag} per lane into the hand-off record (lanes 0..3 of the row: Cb, Cb, Cr,
          // Cr), each ONE aligned 8-byte write-through store: the tag arrives with the pixels
          if (valid && g == gl && i < 4)
            wv::st_sc1_64((unsigned long long*)handMy + ((HAND_WORDS / 2) * x + 4 + i), (unsigned long long)v | ((unsigned long long)A.gen << 32));
        }
      }
      // left neighbour copy: chroma column 7
      {
        const unsigned c = wv::lds_u8(aLeftRd + 8 * slot);
        wv::lds_st8(aLeftWr, c);
      }
      wv::wave_sync();
      // flush the staged rows: every NSC-th macroblock, or at the end of a row: 8 * NSC contiguous bytes per pixel row.
      EXP_REP(8)
      // (scalar pre-test: a row flushes at x = NSC - 1 mod NSC or x = W - 1, and every row's x has the parity of s)
      if (!EXP_SKIP(8) && ((s & 1) != 0 || ((W - 1 - s) & 1) == 0)) {
        constexpr int LR = 8 * NSC;  // lanes per macroblock row: 2 planes x 8 pixel rows x NSC / 2 segments of 16 bytes
#pragma unroll
        for (int it = 0; it < NSC / 2; it++) {
          // (64 lanes or more per macroblock row: the row and whether it is flushed at all are wave-uniform, decided on
          // the scalar side before any lane work)
          const int fg = LR >= 64 ? (64 * it) / LR : (64 / LR) * it + lane / LR;
          const int w = LR >= 64 ? ((64 * it) % LR) + lane : lane % LR;
          const int fx = s - 2 * fg, xp = fx & ~(NSC - 1);
          const bool rowFlush = fg < nR && fx >= 0 && fx < W && ((fx & (NSC - 1)) == NSC - 1 || fx == W - 1);
          if (LR >= 64 && !rowFlush) continue;
          if (LR == 64) {
            // (everything lane-dependent was prepared before the task loop; the row's part is wave-uniform)
            const int room = fx - xp;   // macroblocks of the segment in front of fx
            const bool ok = (fseg2 & 63) <= room;
            const u32x4 v = wv::lds_u128(aFlush + 16 * CW * fg);
            uint8_t* dst = planeY + (oFlush + (unsigned)(8 * (r0 + fg) * pitchC + 8 * xp));
            if (ok) {
              if ((fseg2 & 63) + 1 <= room) wv::st_g128(dst, v);
              else wv::st_g64(dst, u32x2{v.x, v.y});
            }
            continue;
          }
          const int pl = w / (LR / 2), fy = (w / (NSC / 2)) & 7, seg = w % (NSC / 2);
          const bool ok = rowFlush && xp + 2 * seg <= fx;
          const u32x4 v = wv::lds_u128(ts + S_STC + 16 * CW * fg + 8 * CW * pl + CW * fy + 16 * seg);
          uint8_t* dst = planeY + ((pl ? offCr : offCb) + (unsigned)((8 * (r0 + fg) + fy) * pitchC + 8 * xp + 16 * seg));
          if (ok) {
            if (xp + 2 * seg + 1 <= fx) wv::st_g128(dst, v);
            else wv::st_g64(dst, u32x2{v.x, v.y});
          }
        }
      }
      wv::wave_sync();
      PH(4);  // chroma lines, copies, flush
    }
    TLE(task, 2, TNOW());
  }
  BAND_DIAG_END();
}

// ==================================================================================================================
// BACK wave: luma prediction and write-out, driven by the records FRONT leaves in LDS
// ==================================================================================================================
// top border of macroblock x's luma tile, lane i < 7 of its row group: corner dword of x-1, 16 bytes of x, 8 bytes of x+1
WV void top_border(int ts, int g, int x, int par, int i) {
  const int e = i == 0 ? x - 1 : i < 5 ? x : x + 1;
  const int so = i == 0 ? 12 : i < 5 ? 4 * (i - 1) : 4 * (i - 5);
  const unsigned v = wv::lds_u32(ringy(ts, g, e, par) + so);
  wv::lds_st32(tile_of(ts, g, x) + 4 + 16 * (x & 1) + 4 * i, v);
}

// ---- BACK8 (builds with the 8x8 transform): the Intra8x8 macroblocks of every step ----------------------------------------
// Same lanes as BACK (16 per row group). Per step it waits for FRONT's record and for BACK's write-out of the step before
// (left columns, line rings, tile borders), predicts and reconstructs the step's Intra8x8 macroblocks into the tiles
// while BACK runs the Intra16x16 pass and the Intra4x4 chain on the others (macroblocks of one step never depend on each
// other), and says so; BACK waits for that before it frees the record and writes the step out.
template <bool WIDE>
WV void band_back8(const KParams& P, const Args& A, const int ldsBase, const int ts) {
  const int lane0 = wv::lane_id();
  BAND_DIAG_BEGIN();
  const int W = P.W, H = P.H, nF = P.n_frames;
  BandGeo G = band_geo(0u, nF, W, H);
  int par = 0;
  const int lane = lane0;
  const int g = lane >> 4, i = lane & 15;
  for (unsigned gstep = 0;; gstep++) {
    const int buf = (int)(gstep % (unsigned)NBUF);
    team_wait(ts + S_FLAGS + F_READY + 4 * buf, gstep + 1);
    const unsigned task = (unsigned)wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 16));
    if (task == TASK_END) break;
    const int s = wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 20));
    if (s == 0) {
      G = band_geo(task, nF, W, H);
      par = wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 24));
    }
    const int resBuf = ts + S_RES + RES_BUF * buf;
    const int r = G.r0 + g;
    const bool mbB = r > 0;
    const int x = s - 2 * g;
    const bool valid = g < G.nR && (unsigned)x < (unsigned)W;
    const bool mbA = x > 0;
    const int slot = x & 1;
    const int tile = tile_of(ts, g, x);
    const unsigned info = wv::lds_u32(ts + S_INFO + 32 * buf + 4 * g);
    const int kind = (int)(info & 3u);
    PH(0);  // wait for the record
    if (wv::any(valid && kind == 1)) {
      // the step's Intra8x8 residuals: the passes over the coefficients FRONT left in the record, then the residuals in their place
      // (nothing here needs the step before: it runs while BACK writes that one out)
      {
        unsigned r8[8];
        const unsigned form = (unsigned)wv::rfl((int)(info >> 24));
        residual8x8_run<WIDE>(form, valid && kind == 1, lane, (int)((info >> 16) & 0xffu), ldsBase, ts, ts + S_RES + RES_BUF * buf, r8);
        if (valid && kind == 1) residual8x8_store(form, lane, resBuf, r8);
        wv::wave_sync();
      }
      team_wait(ts + S_F8 + F8_WO, gstep);
      PH(1);  // wait for BACK's write-out of the step before
      if (i < 7 && kind == 1) top_border(ts, g, x, par, i);
      wv::wave_sync();
      // ================= luma, Intra8x8 (8.3.2, pred8x8.rs:152-696): four serial blocks ============================
      // The 16 lanes of a macroblock first build the block's filtered edge E1[0..24] = L7..L0, TL, T0..T15 (8.3.2.2.1,
      // pred8x8.rs:222-288, incl. quirk Q1), two samples per lane with the neighbours exchanged by DPP, as bytes in LDS; then
      // every lane predicts four pixels of one row the way the Intra4x4 chain does: per pixel pair an aligned 8-byte window of
      // the edge, per pixel a byte selector on it and (sum of the four bytes + 2) >> 2 (T_T8S / T_T8O: by table row = mode, or
      // quirk Q4's zero row, which band_modes has put in the mode record), adds the residual and stores them into the tile.
      {
        const bool mine = valid && kind == 1;
        const int e8 = ts + S_E8 + 32 * g;
        const unsigned modes4 = wv::lds_u32(ts + S_MSEQ + 96 * buf + 24 * g);
        const bool mbC = mbB && (x + 1 < P.W);
        const int py = i >> 1, x0 = 4 * (i & 1);
        // the four blocks' residuals depend on nothing the chain below produces: requested up front
        u32x2 rrs[4];
#pragma unroll
        for (int b8 = 0; b8 < 4; b8++)
          rrs[b8] = wv::lds_u64(resBuf + RES_ROW * g + 32 * (4 * (2 * (b8 >> 1) + (py >> 2)) + 2 * (b8 & 1) + (x0 >> 2)) + 8 * (py & 3));
#pragma unroll
        for (int b8 = 0; b8 < 4; b8++) {
          const int bx = b8 & 1, by = b8 >> 1;
          const bool topAv = by > 0 || mbB, leftAv = bx > 0 || mbA, tlAv = topAv && leftAv;
          const bool trAv = b8 == 0 ? mbB : b8 == 1 ? mbC : b8 == 2;
          const int row = min((int)((modes4 >> (8 * b8 + 4)) & 0xfu), T8R_ZERO);
          // the lane's table entry (it does not depend on the edge: on its way while the edge is built)
          const u32x4 sel = wv::lds_u128(ldsBase + T_T8S + T8S_ROW * row + 16 * i);
          const unsigned offs = wv::lds_u32(ldsBase + T_T8O + 64 * row + 4 * i);
          const int org8 = tile + TILE_STRIDE * (8 * by) + 8 + 16 * slot + 8 * bx;  // row y = -1, x = 0 of the block
          // raw edge samples k = i and k = i + 16 (top-right replaced by T7 when unavailable)
          auto eaddr = [&](int k) -> int {
            const int ei = min(k, trAv ? 24 : 16);
            return ei <= 7 ? org8 + TILE_STRIDE * (8 - ei) - 1 : ei == 8 ? org8 - 1 : org8 + ei - 9;
          };
          const int lo = (int)wv::lds_u8(eaddr(i)), hi = (int)wv::lds_u8(eaddr(min(i + 16, 24)));
          // neighbours along the edge: lane 15's right neighbour is lane 0's second sample and vice versa
          int lfLo = wv::dpp<DPP_ROW_SHR(1)>(lo, lo);                               // (k = 0 keeps itself)
          int rtLo = wv::dpp<DPP_ROW_SHL(1)>(wv::dppx<DPP_ROW_ROR(15)>(hi), lo);
          int lfHi = wv::dpp<DPP_ROW_SHR(1)>(wv::dppx<DPP_ROW_ROR(1)>(lo), hi);
          int rtHi = wv::dpp<DPP_ROW_SHL(1)>(hi, hi);
          if (i >= 8) rtHi = hi;                                                    // k >= 24: no right neighbour
          if (i == 8) {                                                             // the corner
            if (!leftAv) lfLo = lo;
            if (!topAv) rtLo = lo;
          }
          if (i == 9 && !tlAv) lfLo = -1;  // Q1: p[-1,-1] = -1 enters the x = 0 filter tap
          if (i == 7 && !tlAv) rtLo = lo;
          const int e1Lo = (lfLo + 2 * lo + rtLo + 2) >> 2, e1Hi = (lfHi + 2 * hi + rtHi + 2) >> 2;
          wv::lds_st8(e8 + i, (unsigned)e1Lo);
          if (i <= 8) wv::lds_st8(e8 + 16 + i, (unsigned)e1Hi);
          wv::wave_sync();
          // four pixels of row py: x0 .. x0 + 3
          // (a window is 4-byte aligned: two dword reads, which the compiler issues as one ds_read2_b32)
          const int qa = e8 + (int)(offs & 0xffffu), qb = e8 + (int)(offs >> 16);
          const unsigned waL = wv::lds_u32(qa), waH = wv::lds_u32(qa + 4), wbL = wv::lds_u32(qb), wbH = wv::lds_u32(qb + 4);
          unsigned p01 = wv::pk_lshr2(wv::sum4_hi(wv::perm(waH, waL, sel.y), wv::sum4(wv::perm(waH, waL, sel.x), 0x00020002u)));
          unsigned p23 = wv::pk_lshr2(wv::sum4_hi(wv::perm(wbH, wbL, sel.w), wv::sum4(wv::perm(wbH, wbL, sel.z), 0x00020002u)));
          if (row == 2) {  // DC (pred8x8.rs:350-394): L7..L0 are bytes 0..7 of the edge, T0..T7 bytes 9..16
            const u32x2 l = wv::lds_u64(e8);
            const unsigned t8 = wv::lds_u32(e8 + 8), t12 = wv::lds_u32(e8 + 12), t16 = wv::lds_u32(e8 + 16);
            const int sumL = (int)wv::dot4(l.x, 0x01010101u, wv::dot4(l.y, 0x01010101u, 0u));
            const int sumT = (int)wv::dot4(t8, 0x01010100u, wv::dot4(t12, 0x01010101u, wv::dot4(t16, 0x00000001u, 0u)));
            const int dc = (topAv && leftAv) ? (sumT + sumL + 8) >> 4 : leftAv ? (sumL + 4) >> 3 : topAv ? (sumT + 4) >> 3 : 128;
            p01 = p23 = (unsigned)dc * 0x10001u;
          }
          const u32x2 rr = rrs[b8];
          const unsigned o = recon_row(p01, p23, rr.x, rr.y);
          if (mine) wv::lds_st32(org8 + TILE_STRIDE * (py + 1) + x0, o);
          wv::wave_sync();
        }
      }

      PH(2);  // Intra8x8
    }
    wv::wave_sync();
    if (lane == 0) wv::lds_st32(ts + S_F8 + F8_DONE, gstep + 1);
  }
  BAND_DIAG_END();
}

template <bool HAS_I8>
WV void band_back(const KParams& P, const Args& A, const int ldsBase, const int ts) {
  const int lane0 = wv::lane_id();
  BAND_DIAG_BEGIN();
  const int W = P.W, H = P.H, nF = P.n_frames;
  const int nBands = (H + 3) >> 2;
  const int pitchY = W * 16;
  const size_t frameBytes = (size_t)W * H * 384;
  // per-task state (set at step 0 of each task)
  BandGeo G = band_geo(0u, nF, W, H);
  uint8_t* planeY = A.yuv;
  unsigned* handMy = A.handoff;   // the band's hand-off records (if it has a band below)
  int par = 0;  // task parity (row 0's luma ring)
  // flush with 64 lanes per macroblock row (NSY = 4): lane = (pixel row, 16-byte segment)
  const unsigned flushLane = (unsigned)wv::opaque(((lane0 / NSY) & 15) * pitchY + 16 * (lane0 % NSY));

  for (unsigned gstep = 0;; gstep++) {
    const int buf = (int)(gstep % (unsigned)NBUF);
    team_wait(ts + S_FLAGS + F_READY + 4 * buf, gstep + 1);
    const unsigned task = (unsigned)wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 16));
    if (task == TASK_END) break;
    const int s = wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 20));
    if (s == 0) {
      G = band_geo(task, nF, W, H);
      par = wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 24));
      TLINE(task, 1, TNOW());
      planeY = A.yuv + (size_t)G.f * frameBytes;
      handMy = A.handoff + ((size_t)G.f * (nBands - 1) + G.b) * (size_t)W * HAND_WORDS;
    }
    const int r0 = G.r0, nR = G.nR, gl = G.gl;
    const bool hasBelow = G.hasBelow;
    const int resBuf = ts + S_RES + RES_BUF * buf;

    const int lane = lane0;  // (BACK has registers to spare: whatever depends on the lane alone is computed once, outside the loop)
    const int g = lane >> 4, i = lane & 15;
    const int ch = (i >> 3) & 1, cp = i & 7;        // block chain: block half, pixel pair
    const int bxp = i & 3, part = i >> 2;           // edge arrays: block column, block row / which part of a top-row block's array
    const int r = r0 + g;
    const bool rowOk = g < nR;
    const bool mbB = r > 0;
    const int x = s - 2 * g;
    const bool valid = rowOk && (unsigned)x < (unsigned)W;
    const bool mbA = x > 0;
    const int slot = x & 1;
    const int tile = tile_of(ts, g, x);
    const int eG = ts + S_EDGE + E_ROW * g;   // the row group's edge arrays: block (bx, by) at E_SLOT * (4 * by + bx)
    const unsigned info = wv::lds_u32(ts + S_INFO + 32 * buf + 4 * g);
    const int kind = (int)(info & 3u), i16mode = (int)((info >> 8) & 3u);
    PH(0);  // wait for the record

    // ================= edge arrays of the step's macroblocks ====================================================
    // Everything a block's prediction reads sits in its 16-byte edge array (E_*). Inside a macroblock the arrays are filled
    // where the pixels are written (the chain's scatter, below); here: what comes from outside the macroblock.
    //   * Intra4x4: the top row's blocks (bx, 0) take T0..T3, T4..T7 and the corner from the line ring (lanes 4 * part + bx:
    //     part 0, 1, 2); the left column's blocks got L3..L0 and their corners at the write-out of the step before.
    //   * Intra16x16 rides the chain too: all sixteen blocks (lane i = block 4 * by + bx) take the macroblock's own top line
    //     (vertical: table row 0) resp. left column (horizontal: T4R_H16) into T0..T3, or -- DC and plane, T4R_PRED16 -- their
    //     sixteen predicted pixels.
    EXP_REP(3)
    {
      const bool i4 = kind == 0, i16 = kind == 2;
      const int de = i4 ? ((part == 1 && bxp == 3) ? 1 : (part == 2 && bxp == 0) ? -1 : 0) : 0;
      const int so = i4 ? (part == 0 ? 4 * bxp : part == 1 ? (bxp < 3 ? 4 * bxp + 4 : 0) : (bxp > 0 ? 4 * bxp - 4 : 12)) : 4 * bxp;
      const unsigned tw = wv::lds_u32(ringy(ts, g, x + de, par) + so);
      const unsigned lw = wv::lds_u32(ts + S_LEFTY + 16 * g + 4 * part);
      const int dst = i4 ? eG + E_SLOT * bxp + (part == 0 ? E_T0 : part == 1 ? E_T4 : E_CORNER) : eG + E_SLOT * i + E_T0;
      if (valid && ((i4 && part < 2) || (i16 && i16mode < 2))) wv::lds_st32(dst, (i16 && i16mode == 1) ? lw : tw);
      if (valid && i4 && part == 2) wv::lds_st8(dst, tw >> 24);
      if (wv::any(valid && i16 && i16mode == 2)) {
        // Intra16x16 DC (pred16x16.rs:291-361): one reduction over the row group of the available sums
        int sm = ((part == 0 && mbB) ? (int)wv::sad4(tw) : 0) + ((bxp == 0 && mbA) ? (int)wv::sad4(lw) : 0);
        sm += xor8(sm);
        sm += xor4(sm, (i & 4) != 0);
        sm += xor2(sm);
        sm += xor1(sm);
        const unsigned v = (unsigned)((mbA && mbB) ? (sm + 16) >> 5 : (mbA || mbB) ? (sm + 8) >> 4 : 128) * 0x01010101u;
        if (valid && i16 && i16mode == 2) wv::lds_st128(eG + E_SLOT * i, u32x4{v, v, v, v});
      }
      if (!EXP_SKIP(3) && wv::any(valid && i16 && i16mode == 3 && mbA && mbB)) {
        // Intra16x16 plane (8.3.3.4, pred16x16.rs:366-424). Every lane of the macroblock forms H, V from the whole top line and
        // left column: H = sum (k + 1) (T[8 + k] - T[6 - k]), T[-1] = the corner, as four byte dot products
        const u32x4 t = wv::lds_u128(ringy(ts, g, x, par)), l = wv::lds_u128(ts + S_LEFTY + 16 * g);
        const unsigned c8 = 8u * wv::lds_u8(ringy(ts, g, x - 1, par) + 15);
        const int hs = (int)wv::dot4(t.z, 0x04030201u, wv::dot4(t.w, 0x08070605u, 0u)) - (int)wv::dot4(t.y, 0x00010203u, wv::dot4(t.x, 0x04050607u, c8));
        const int vs = (int)wv::dot4(l.z, 0x04030201u, wv::dot4(l.w, 0x08070605u, 0u)) - (int)wv::dot4(l.y, 0x00010203u, wv::dot4(l.x, 0x04050607u, c8));
        const int a = 16 * (int)((l.w >> 24) + (t.w >> 24));
        const int bq = (5 * hs + 32) >> 6, c = (5 * vs + 32) >> 6;
        // pixel (x, y) of the macroblock: clip255((a + b (x - 7) + c (y - 7) + 16) >> 5); all terms fit 16 bits. This lane's block:
        // (bxp, part)
        const int base = a + bq * (4 * bxp - 7) + c * (4 * part - 7) + 16;
        const unsigned b01 = ((unsigned)base & 0xffffu) | ((unsigned)(base + bq) << 16);
        const unsigned step2 = ((unsigned)(2 * bq) & 0xffffu) | ((unsigned)(2 * bq) << 16);
        const unsigned cc = ((unsigned)c & 0xffffu) | ((unsigned)c << 16);
        unsigned q01 = b01, q23 = wv::pk_add(b01, step2), row[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          row[kk] = wv::perm(wv::sat_pk_u8(wv::pk_ashr5(q23)), wv::sat_pk_u8(wv::pk_ashr5(q01)), 0x05040100u);
          q01 = wv::pk_add(q01, cc);
          q23 = wv::pk_add(q23, cc);
        }
        if (valid && i16 && i16mode == 3 && mbA && mbB) wv::lds_st128(eG + E_SLOT * i, u32x4{row[0], row[1], row[2], row[3]});
      }
      // no macroblock to the left: the left column reads 128 (DC of a block with no neighbour at all: T4R_DC_LEFT; nothing
      // else reads it there)
      if (wv::any(valid && x == 0 && i4)) {
        if (valid && x == 0 && i4) wv::lds_st8(eG + 4 * E_SLOT * part + E_L3 + 3 - bxp, 128u);
      }
    }
    wv::wave_sync();
    PH(1);  // edge arrays
      PH(2);
      PH(3);
      const bool mine = valid && !(HAS_I8 && kind == 1);
      const bool anyChain = !EXP_SKIP(4) && wv::any(mine);

      // ================= the block chain: Intra4x4 pixels (8.3.1.2, pred4x4.rs:10-360) and all of Intra16x16 (8.3.3) ======
      // lane = (row g, block half ch, pixel pair cp). Rounds with two blocks per macroblock (2..7): pixels
      // (2 * (cp & 1) + {0, 1}, cp >> 1) of block (bx0, by0) for ch = 0, of block (bx0 - 2, by0 + 1) for ch = 1. Rounds with
      // one block (0, 1, 8, 9): all sixteen lanes work on it, one pixel each: (2 * (cp & 1) + ch, cp >> 1).
      // A pixel is (sum of four bytes of the block's edge array + 2) >> 2: the table entry of (row, pixel pair) names an aligned
      // 8-byte window of the array and, per pixel, a byte selector on it (v_perm_b32), e.g. (a, b, b, c) for a three-tap value.
      // An Intra4x4 block's pixels go to the tile and, where they are some block's neighbours, into that block's array:
      // bottom row -> T0..T3 of the block below and T4..T7 of the block below left, right column -> L3..L0 of the block to
      // the right, pixel (3, 3) -> the corner of the block below right. A lane that holds none of these writes to the dump
      // bytes of its own block's array; destinations outside the macroblock are the spare block row 4 or left out per round.
      EXP_REP(4)
      if (anyChain) {
      // (the whole chain under ONE lane mask: a macroblock's lanes take part in all ten rounds or in none, and a mask set and
      // restored around every store of every round is a dozen scalar instructions per round on the wave that paces the band.
      // In the emulator every lane walks through the rounds' barriers: the stores keep their predicate.)
      WV_LANES_IF(mine) {
        const bool i4 = kind == 0;
        const int px = 2 * (cp & 1), py = cp >> 1, xq = px + ch;
        const int seqA = ts + S_MSEQ + 96 * buf + 24 * g + 12 * ch;
        // Everything a round addresses is a per-lane base plus a constant of the round, which the LDS instructions carry as
        // their immediate offset: the second block of a two-block round is always (bx - 2, by + 1) of the first.
        const int orgS = tile + 8 + 16 * slot - 1;                  // (block origin - one row - one column) of block (0, 0)
        const int orgB = orgS + (ch ? 4 * TILE_STRIDE - 8 : 0);     // ... of block (0, 0) / (-2, 1)
        const int stB = orgB + TILE_STRIDE * (py + 1) + 1 + px, stS = orgS + TILE_STRIDE * (py + 1) + 1 + px + ch;
        const int resS = resBuf + RES_ROW * g + 4 * cp + 2 * ch;        // residuals [4 * by + bx][y][x]: this lane's one pixel
        const int resB = resBuf + RES_ROW * g + 4 * cp + (ch ? 64 : 0); // ... this lane's pixel pair
        const int eB = eG + (ch ? 2 * E_SLOT : 0);                   // edge array of block (0, 0) / (-2, 1)
        const int entB = ldsBase + T_T4W + T4W_PAIR * cp, entS = entB + 4 + 4 * ch;   // table entries of the pair; the lane's one pixel's selector
        // scatter destinations (+ the round's E_SLOT * (4 * by0 + bx0)). A lane with nothing to scatter -- by its pixels, or
        // because its macroblock is not Intra4x4 (an Intra16x16 block's array is its own macroblock's, not its neighbours') --
        // writes to the dump bytes of its block's array.
        const int dumpB = eB + E_DUMP, dumpS = eG + E_DUMP;
        const int aW1 = (i4 && py == 3) ? eB + 4 * E_SLOT + E_T0 + px : dumpB;
        const int aW2 = (i4 && py == 3) ? eB + 3 * E_SLOT + E_T4 + px : dumpB, aW2c0 = ch ? dumpB : aW2;
        const int aW3 = (i4 && px == 2) ? eB + E_SLOT + E_L3 + 3 - py : dumpB, aW3c1 = ch ? aW3 : dumpB;
        const int aW4 = (i4 && cp == 7) ? eB + 5 * E_SLOT + E_CORNER : dumpB, aW4c1 = ch ? aW4 : dumpB;
        const int aS1 = (i4 && py == 3) ? eG + 4 * E_SLOT + E_T0 + xq : dumpS, aS2 = (i4 && py == 3) ? eG + 3 * E_SLOT + E_T4 + xq : dumpS;
        const int aS3 = (i4 && xq == 3) ? eG + E_SLOT + E_L3 + 3 - py : dumpS, aS4 = (i4 && xq == 3 && py == 3) ? eG + 5 * E_SLOT + E_CORNER : dumpS;
        // the table rows of all ten rounds (they do not depend on pixels): this lane's block half, and the first half's
        // for the one-block rounds; then entry and residual one round ahead
        const unsigned sq0 = wv::lds_u32(seqA), sq1 = wv::lds_u32(seqA + 4), sq2 = wv::lds_u32(seqA + 8);
        const unsigned sS0 = wv::lds_u32(seqA - 12 * ch), sS2 = wv::lds_u32(seqA - 12 * ch + 8);
        u32x3 en = u32x3{wv::lds_u32(entB + (int)(sS0 & 0xffu)), wv::lds_u32(entS + (int)(sS0 & 0xffu)), 0u};
        unsigned rr = (unsigned)wv::lds_i16(resS);
        const unsigned dcRounds = (unsigned)wv::rfl((int)wv::lds_u32(ts + S_INFO + 32 * buf + 28));
        constexpr unsigned SEL_DC = 0x0d0d0d0du;   // the selector of the row whose value is the block's DC
#define I4_BODY(T, DC)                                                                                           \
        {                                                                                                         \
          unsigned dcv = 0;                                                                                       \
          if (DC) {                                                                                               \
            /* DC (pred4x4.rs:116-167) with both neighbours: (L0 + .. + L3 + T0 + .. + T3 + 4) >> 3 of the block's own array */ \
            const u32x4 ee = wv::lds_u128((two ? eB : eG) + offE);                                                \
            dcv = wv::dot4(ee.x, 0x01000000u, wv::dot4(ee.y, 0x00010101u, wv::dot4(ee.z, 0x01010101u, 4u))) >> 3; \
          }                                                                                                       \
          if (two) {                                                                                              \
            const unsigned pa = wv::perm(hi, lo, en.y), pb = wv::perm(hi, lo, en.z);                              \
            unsigned pr = wv::pk_lshr2(wv::sum4_hi(pb, wv::sum4(pa, 0x00020002u)));                               \
            if (DC) pr = en.y == SEL_DC ? dcv * 0x10001u : pr;                                                    \
            const unsigned o = wv::sat_pk_u8(wv::pk_add_sat(pr, rr));                                             \
            if (mine) {                                                                                           \
              const unsigned o8 = o >> 8;                                                                         \
              wv::lds_st16(stB + offT, o);                                                                        \
              wv::lds_st16(aW1 + offE, o);                                                                        \
              wv::lds_st16((((T) & 1) ? aW2 : aW2c0) + offE, o);                                                  \
              wv::lds_st8((((T) & 1) ? aW3c1 : aW3) + offE, o8);                                                  \
              wv::lds_st8((((T) & 1) ? aW4c1 : aW4) + offE, o8);                                                  \
            }                                                                                                     \
          } else {                                                                                                \
            const unsigned pq = wv::perm(hi, lo, en.y);                                                           \
            int pv = (int)(wv::sum4(pq, 2u) >> 2);                                                                \
            if (DC) pv = en.y == SEL_DC ? (int)dcv : pv;                                                          \
            const int o = wv::med3(pv + (int)rr, 0, 255);                                                         \
            if (mine) {                                                                                           \
              wv::lds_st8(stS + offT, (unsigned)o);                                                               \
              if ((T) < 2) wv::lds_st8(aS1 + offE, (unsigned)o);                                                  \
              if ((T) == 1) wv::lds_st8(aS2 + offE, (unsigned)o);                                                 \
              if ((T) != 9) wv::lds_st8(aS3 + offE, (unsigned)o);                                                 \
              if ((T) < 2) wv::lds_st8(aS4 + offE, (unsigned)o);                                                  \
            }                                                                                                     \
          }                                                                                                       \
        }
#define I4_STEP(T)                                                                                              \
        {                                                                                                         \
          constexpr int by0 = stepByLo(T), bx0 = (T) - 2 * by0;                                                   \
          constexpr bool two = by0 + 1 <= stepByHi(T);                                                            \
          constexpr int TN = (T) < 9 ? (T) + 1 : 9;                                                               \
          constexpr int byN = stepByLo(TN), bxN = TN - 2 * byN;                                                   \
          constexpr bool twoN = byN + 1 <= stepByHi(TN);                                                          \
          constexpr int offT = TILE_STRIDE * 4 * by0 + 4 * bx0;                                                   \
          constexpr int offE = E_SLOT * (4 * by0 + bx0);                                                          \
          /* the window's address: base + table offset behind an optimisation barrier, so that the round's constant */ \
          /* rides in the loads' immediate offset instead of being added to the base once per round */            \
          const int q = wv::opaque((two ? eB : eG) + (int)en.x);                                                  \
          const unsigned lo = wv::lds_u32(q + offE), hi = wv::lds_u32(q + offE + 4);                              \
          /* the next round's entry and residual: requested here, in front of the branch, so that they travel while this */ \
          /* round computes (behind the branch the compiler merges them into the bodies' common tail, after the arithmetic) */ \
          u32x3 enN = en;                                                                                         \
          unsigned rrN;                                                                                           \
          if (twoN) {                                                                                             \
            const unsigned mN = ((TN < 4 ? sq0 : TN < 8 ? sq1 : sq2) >> (8 * (TN & 3))) & 0xffu;                  \
            enN = wv::lds_u96(entB + (int)mN);                                                                    \
            rrN = wv::lds_u32(resB + 32 * (4 * byN + bxN));                                                       \
          } else {                                                                                                \
            const unsigned mN = ((TN < 4 ? sS0 : sS2) >> (8 * (TN & 3))) & 0xffu;                                 \
            enN.x = wv::lds_u32(entB + (int)mN);                                                                  \
            enN.y = wv::lds_u32(entS + (int)mN);                                                                  \
            rrN = (unsigned)wv::lds_i16(resS + 32 * (4 * byN + bxN));                                             \
          }                                                                                                       \
          /* (wave-uniform: the mode pre-pass marked the rounds in which some block of the step takes its own DC) */ \
          if (dcRounds & (1u << (T))) I4_BODY(T, true) else I4_BODY(T, false)                                     \
          en = enN;                                                                                               \
          rr = rrN;                                                                                               \
          wv::wave_sync();                                                                                        \
        }
        I4_STEP(0) I4_STEP(1) I4_STEP(2) I4_STEP(3) I4_STEP(4) I4_STEP(5) I4_STEP(6) I4_STEP(7) I4_STEP(8) I4_STEP(9)
#undef I4_STEP
#undef I4_BODY
      }
      }

      PH(4);  // block chain
      // (builds with the 8x8 transform: the step's Intra8x8 macroblocks are BACK8's, which worked next to the chain)
      // (at least: BACK8 may already be through a next step that has none)
      if (HAS_I8) team_wait_ge(ts + S_F8 + F8_DONE, gstep + 1);

      // the record has been consumed
      wv::wave_sync();
      if (lane == 0) wv::lds_st32(ts + S_FLAGS + F_FREE + 4 * buf, gstep + 1);
      PH(6);  // wait for BACK8, record freed

      // ================= luma write-out ============================================================================
      EXP_REP(11) {
        // bottom line for the row below (ring) or the band below (write-through): lanes 0..3 of row g
        unsigned v = 0;
        if (i < 4) v = wv::lds_u32(tile + TILE_STRIDE * 16 + 8 + 16 * slot + 4 * i);
        if (valid && i < 4 && g < 3 && g < gl) wv::lds_st32(ringy(ts, g + 1, x, par) + 4 * i, v);
        if (hasBelow && wv::any(valid && g == gl)) {
          // the band's last row: a granule {4 pixels, tag} per lane into the hand-off record (lanes 0..3 of the row), each ONE
          // aligned 8-byte write-through store (MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, R2: the data is
          // the flag). The tag travels with the pixels: the band below needs no progress word, and this wave no wait for its
          // stores.
          if (valid && g == gl && i < 4)
            wv::st_sc1_64((unsigned long long*)handMy + ((HAND_WORDS / 2) * x + i), (unsigned long long)v | ((unsigned long long)A.gen << 32));
        }
      }
      // left neighbour copy: luma column 15 -> the left column (Intra16x16), L3..L0 and the corners of the next macroblock's
      // blocks (0, by) (lane i = pixel row: block row i >> 2), and, for BACK8, the tile's x = -1 border when the next macroblock is
      // slot 0
      {
        const unsigned v = wv::lds_u8(tile + TILE_STRIDE * (i + 1) + 8 + 16 * slot + 15);
        wv::lds_st8(ts + S_LEFTY + 16 * g + i, v);
        wv::lds_st8(eG + 4 * E_SLOT * part + E_L3 + 3 - bxp, v);
        wv::lds_st8(eG + ((bxp == 3 && part < 3) ? 4 * E_SLOT * (part + 1) + E_CORNER : E_DUMP), v);
        if (HAS_I8 && slot == 1) wv::lds_st8(tile_of(ts, g, x + 1) + TILE_STRIDE * (i + 1) + 7, v);
      }
      wv::wave_sync();
      // flush the staged rows: every NSY-th macroblock, or at the end of a row: 16 * NSY contiguous bytes per pixel row.
      EXP_REP(5)
      // (scalar pre-test: a row flushes at x = NSY - 1 mod NSY or x = W - 1, and every row's x has the parity of s)
      if (!EXP_SKIP(5) && ((s & 1) != 0 || ((W - 1 - s) & 1) == 0)) {
        constexpr int LR = 16 * NSY;  // lanes per macroblock row: 16 pixel rows x NSY segments of 16 bytes
#pragma unroll
        for (int it = 0; it < NSY; it++) {
          // (64 lanes or more per macroblock row: the row and whether it is flushed at all are wave-uniform)
          const int fg = LR >= 64 ? (64 * it) / LR : (64 / LR) * it + lane / LR;
          const int w = LR >= 64 ? ((64 * it) % LR) + lane : lane % LR;
          const int fx = s - 2 * fg, xp = fx & ~(NSY - 1);
          const bool rowFlush = fg < nR && fx >= 0 && fx < W && ((fx & (NSY - 1)) == NSY - 1 || fx == W - 1);
          if (LR >= 64 && !rowFlush) continue;
          const int fy = (w / NSY) & 15, seg = w % NSY;
          const bool ok = rowFlush && xp + seg <= fx;
          const int src = ts + S_TILE + TILE_BYTES * (NP * fg + (seg >> 1)) + TILE_STRIDE * (fy + 1) + 8 + 16 * (seg & 1);
          const u32x2 lo = wv::lds_u64(src), hi = wv::lds_u64(src + 8);
          // (the lane's part of the address -- its pixel row and segment -- is a constant of the kernel when a macroblock row
          // takes the whole wave: flushLane; the row's part is wave-uniform)
          const unsigned off = LR == 64 ? flushLane + (unsigned)(16 * (r0 + fg) * pitchY + 16 * xp)
                                        : (unsigned)((16 * (r0 + fg) + fy) * pitchY + 16 * (xp + seg));
          if (ok) wv::st_g128(planeY + off, u32x4{lo.x, lo.y, hi.x, hi.y});
        }
      }
      wv::wave_sync();
      if (HAS_I8 && lane == 0) wv::lds_st32(ts + S_F8 + F8_WO, gstep + 1);  // BACK8 may start the next step
      PH(5);  // line, copies, flush
      if (s == G.nSteps - 1) TLINE(task, 2, TNOW());
      if (s == G.nSteps - 1) TLE(task, 1, TNOW());
  }
  BAND_DIAG_END();
}

}  // namespace band
}  // namespace dryv
