// recon_kernel.hip — gfx950 (CDNA4 / MI355X) macroblock reconstruction for dryv's AVC intra path.
//
// Work decomposition
//   * One 64-lane wavefront processes one macroblock ROW of one frame, left to right. Row r may start macroblock x
//     once row r-1 has finished macroblock x (neighbours B and D) and must see macroblock x+1 finished (neighbour
//     C; reference slice/macroblock.rs:455-456, slice/mod.rs:593-598) only before the one Intra4x4 / Intra8x8
//     block that reads its samples.
//   * The unit handed out by the global queue is a BAND: 4 consecutive rows of one frame, one per wave. A
//     512-thread workgroup runs two independent bands that share nothing but the constant tables; the waves of a
//     band synchronise through LDS counters, never through a workgroup barrier. Inside a band, rows hand off
//     through an 8-macroblock LDS ring (bottom pixel line + bottom-row prediction modes) with two-way progress
//     counters (cached in SGPRs: a poll that is already satisfied costs nothing).
//   * Between bands the hand-off goes through L2 (MI355X_MICROARCH.md, "valid forms"): the last row of a band
//     stores its bottom pixel line and modes write-through (sc1), drains vmcnt, then stores its progress counter
//     (sc1); wave 0 of the band below polls that counter and reads the 25+9+9 neighbour samples with sc1 loads
//     only (they bypass the CU's L1, so no acquire/invalidate is needed). Every band comes off ONE queue in the
//     order (band 0 of every frame, band 1 of every frame, ...): the band a task depends on has a smaller number,
//     so it was claimed earlier by a workgroup that is running or done -- no deadlock however few workgroups are
//     resident -- and it is ~4 x (frames x W / waves) macroblocks ahead, so this slow hand-off practically never
//     blocks. Any band slot can take any band: 300 independent frames spread evenly over 256 CUs.
//   * HBM traffic per macroblock: 768 B of coefficients in (one global->LDS DMA of 48 x 16 B, issued a
//     macroblock ahead), a 16 B record, 384 B of pixels out, staged in LDS and stored as whole 32-byte row
//     segments (luma every 2nd, chroma every 4th macroblock); per band boundary row 4 B of modes and a 68 B
//     neighbour window re-read through L2.
//
// Inside a macroblock
//   * residual: 4 lanes per 4x4 block; lane = one row of coefficients (inverse zig-zag is a 4-way LDS gather),
//     table dequantisation, row butterfly in-lane, 4x4 transpose through swizzled LDS slots, column butterfly.
//   * Intra16x16 and chroma: predicted in registers from the neighbour window / left-edge bytes, clip-added,
//     packed 4 pixels per dword, byte-transposed across the quad (v_perm_b32).
//   * Intra4x4: prediction modes by a 7-sweep DPP relaxation over the 4x4 block grid; pixels by a statically
//     unrolled 10-step 2:1 block wavefront, 16 lanes per block: every pixel is (E[p] + 2E[q] + E[r] + 2) >> 2 of
//     three reference samples whose tile offsets come from a per-(mode, pixel) table -- three LDS byte reads, no
//     cross-lane traffic (see I4Lane).
//   * Intra8x8: 64 lanes (one pixel per lane); reference samples on lanes 0..24, the reference's filter (incl.
//     quirk Q1) by lane shuffles, one ds_bpermute picks E/F/G per pixel through a mode table.
//
// Bit-exact with the reference's Rust CPU path (quirks Q1-Q5 of SURVEY.md 8a' included). Arithmetic is int32
// (reference: 64-bit isize): exact whenever every intermediate fits 32 bits -- any conformant 8-bit stream,
// |level| <= 2047 with flat scaling lists at every QP, the full int16 range at qp <= 24.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "recon_kernel.h"

namespace dryv {

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

#define QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define ROW_SHL(n) (0x100 + (n))
#define ROW_SHR(n) (0x110 + (n))

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;

template <int CTRL>
__device__ __forceinline__ int dpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ int xor1(int v) { return dpp<QUAD(1, 0, 3, 2)>(v, v); }
__device__ __forceinline__ int xor2(int v) { return dpp<QUAD(2, 3, 0, 1)>(v, v); }

__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int shfl(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ unsigned sum4(unsigned w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }
__device__ __forceinline__ bool bytes_nonzero(unsigned w) {
  return (((w - 0x01010101u) & ~w) & 0x80808080u) == 0u;
}

// ---- per-wave LDS scratch ---------------------------------------------------------------------
struct WaveScratch {
  int16_t coef[384];  // the macroblock's coefficient lists as they arrive (zig-zag order); DMA target
  union {
    struct {
      union {
        unsigned resTe[256];   // Intra4x4: per pixel [by*4+bx][x*4+y]: prediction entry (see I4Lane)
        int16_t resB[256];     // Intra8x8 residual [y*16+x]
      };
      uint8_t tileY[17 * 32];  // luma tile with borders: row 0 = y -1, byte 3 = x -1, bytes 4..27 = x 0..23
      int16_t resS[256];       // Intra4x4: per pixel [by*4+bx][x*4+y]: residual
    };
    int32_t g8[256];  // 8x8 transform: row-pass output of the four blocks
  };
  uint8_t leftY[16];  // left neighbour's column x = 15 (contiguous copy)
  uint8_t leftC[2][8];
  // neighbour window of the row above, fetched per macroblock: Y x = -4..27, Cb x = -4..11, Cr x = -4..11,
  // then the four bottom-row modes of macroblock B
  uint8_t up[80];
  uint8_t pad[48];
  // Store staging: a macroblock contributes 16 B (luma) / 8 B (chroma) per pixel row, and partial-line stores
  // cost the memory system a full request each. Rows are collected here and stored 32 B at a time.
  uint8_t stageY[16][32];    // luma rows of macroblocks (mx & ~1), (mx | 1)
  uint8_t stageC[2][8][32];  // chroma rows of macroblocks (mx & ~3) .. (mx | 3)
};
static_assert(sizeof(WaveScratch) % 64 == 0, "the transpose swizzle needs 64-byte aligned scratch");

__host__ __device__ constexpr int TY(int x, int y) { return (y + 1) * 32 + (x + 4); }

// 4x4 transpose of (r0..r3) x (4 lanes of a quad): afterwards lane l holds what register l held on
// lanes 0..3. Two exchange stages (lane^1, lane^2), 16 VALU, no LDS.
__device__ __forceinline__ void quad_transpose4(int& r0, int& r1, int& r2, int& r3, bool odd, bool hi) {
  const int s01 = odd ? r0 : r1, s23 = odd ? r2 : r3;
  const int v01 = xor1(s01), v23 = xor1(s23);
  const int a0 = odd ? v01 : r0, a1 = odd ? r1 : v01, a2 = odd ? v23 : r2, a3 = odd ? r3 : v23;
  const int t02 = hi ? a0 : a2, t13 = hi ? a1 : a3;
  const int w02 = xor2(t02), w13 = xor2(t13);
  r0 = hi ? w02 : a0;
  r2 = hi ? a2 : w02;
  r1 = hi ? w13 : a1;
  r3 = hi ? a3 : w13;
}

// 4x4 BYTE transpose across a quad: lane l byte k  <->  lane k byte l.
__device__ __forceinline__ unsigned quad_transpose_bytes(unsigned w, unsigned selA, unsigned selB) {
  const unsigned n = (unsigned)xor1((int)w);
  const unsigned x = __builtin_amdgcn_perm(n, w, selA);
  const unsigned m = (unsigned)xor2((int)x);
  return __builtin_amdgcn_perm(m, x, selB);
}

typedef __attribute__((address_space(3))) const int16_t* lds_i16p;
typedef __attribute__((address_space(3))) const uint16_t* lds_u16p;
typedef __attribute__((address_space(3))) const unsigned* lds_u32p;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* lds_u4p;
typedef __attribute__((address_space(3))) i32x4* lds_i4w;
typedef __attribute__((address_space(3))) const int* lds_i32p;
#define LDSP(T, a) ((T)(uintptr_t)(unsigned)(a))

// ---- 4x4 residual, 4 lanes per block (transform.rs:116-191, 8.5.12) ---------------------------
// In:  this lane is row sq of its block; a0..a3 = LDS byte addresses of c[sq][0..3] in the staged lists;
//      lsAddr = LDS address of the four pre-shifted LevelScale words of (qp, row sq); qi = per-qp
//      (rounding << 8 | right shift). dcLane: c[0][0] arrives already scaled (Intra16x16 / chroma,
//      transform.rs:145-146) and this is the lane holding it.
//      d = (c * (LS << max(qp/6-4, 0)) + rnd) >> max(4-qp/6, 0)  ==  transform.rs:147-152.
// The 4x4 transpose between the row and the column butterfly goes through LDS (trW: this lane's 16 bytes,
// trR: its column in the quad's 64 bytes). Row r of quad q sits in 16-byte chunk r ^ ((q >> 1) & 3) of the
// quad's 64 bytes: with the plain layout the eight quads of a half-wave hit two bank groups (4-way conflicts
// on every column read); with the swizzle they hit eight.
// Out: r[k] = residual of the block at (x = sq, y = k) — the lane now owns COLUMN sq.
__device__ __forceinline__ void residual4x4_quad(int a0, int a1, int a2, int a3, bool dcLane, int dcVal, int lsAddr,
                                                 unsigned qi, int trW, int trR, int r[4]) {
  const int c0 = *LDSP(lds_i16p, a0), c1 = *LDSP(lds_i16p, a1), c2 = *LDSP(lds_i16p, a2), c3 = *LDSP(lds_i16p, a3);
  const u32x4 ls = *LDSP(lds_u4p, lsAddr);
  const int rnd = (int)(qi >> 8), shr = (int)(qi & 0xff);
  int d0 = (__mul24(c0, (int)ls.x) + rnd) >> shr;
  const int d1 = (__mul24(c1, (int)ls.y) + rnd) >> shr;
  const int d2 = (__mul24(c2, (int)ls.z) + rnd) >> shr;
  const int d3 = (__mul24(c3, (int)ls.w) + rnd) >> shr;
  if (dcLane) d0 = dcVal;
  // row butterfly (transform.rs:159-169)
  const int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
  *LDSP(lds_i4w, trW) = i32x4{e0 + e3, e1 + e2, e1 - e2, e0 - e3};
  WAVE_SYNC();
  const int f0 = *LDSP(lds_i32p, trR), f1 = *LDSP(lds_i32p, trR ^ 16), f2 = *LDSP(lds_i32p, trR ^ 32),
            f3 = *LDSP(lds_i32p, trR ^ 48);  // f[0..3][sq]
  WAVE_SYNC();
  // column butterfly (transform.rs:171-181) and rounding (:183-187)
  const int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
  r[0] = (g0 + g3 + 32) >> 6;
  r[1] = (g1 + g2 + 32) >> 6;
  r[2] = (g1 - g2 + 32) >> 6;
  r[3] = (g0 - g3 + 32) >> 6;
}

// 8-point butterfly shared by the row and column pass of 8.5.13 (pred8x8.rs:85-141)
__device__ __forceinline__ void idct8(const int d[8], int o[8]) {
  const int e0 = d[0] + d[4];
  const int e1 = -d[3] + d[5] - d[7] - (d[7] >> 1);
  const int e2 = d[0] - d[4];
  const int e3 = d[1] + d[7] - d[3] - (d[3] >> 1);
  const int e4 = (d[2] >> 1) - d[6];
  const int e5 = -d[1] + d[7] + d[5] + (d[5] >> 1);
  const int e6 = d[2] + (d[6] >> 1);
  const int e7 = d[3] + d[5] + d[1] + (d[1] >> 1);
  const int f0 = e0 + e6;
  const int f1 = e1 + (e7 >> 2);
  const int f2 = e2 + e4;
  const int f3 = e3 + (e5 >> 2);
  const int f4 = e2 - e4;
  const int f5 = (e3 >> 2) - e5;
  const int f6 = e0 - e6;
  const int f7 = e7 - (e1 >> 2);
  o[0] = f0 + f7;
  o[1] = f2 + f5;
  o[2] = f4 + f3;
  o[3] = f6 + f1;
  o[4] = f6 - f1;
  o[5] = f4 - f3;
  o[6] = f2 - f5;
  o[7] = f0 - f7;
}

// Intra4x4 block schedule: step T runs the blocks with bx + 2*by == T (at most two: group 0 and 1)
__host__ __device__ constexpr int stepByLo(int t) { return t < 2 ? 0 : (t - 2) >> 1; }
__host__ __device__ constexpr int stepByHi(int t) { return (t >> 1) < 3 ? (t >> 1) : 3; }
// top-right 4x4 block decoded before this one inside the macroblock (by > 0): bx even, or (1,2)
__host__ __device__ constexpr bool trInside(int bx, int by) { return (bx & 1) == 0 || (bx == 1 && by == 2); }

// ---- Intra4x4 pixel wavefront -------------------------------------------------------------------
// Lanes 0..31, 16 lanes (pixels) per block, two blocks (groups) per step. Every directional mode of 8.3.1.2
// predicts a pixel from at most three neighbouring reference samples p, q, r of the line
//   E = [ L3 L2 L1 L0 | corner | T0..T7 ]            (left column bottom to top, corner, top row + top-right)
// as (E[p] + 2 E[q] + E[r] + 2) >> 2:  E[i] itself = (i, i, i);  the 3-tap (E[i-1] + 2E[i] + E[i+1] + 2) >> 2
// = (i-1, i, i+1);  the 2-tap (E[i] + E[i+1] + 1) >> 1 = (i, i+1, i)  [(2a + 2b + 2) >> 2 == (a + b + 1) >> 1].
// The per-pixel table entry therefore is three byte offsets INTO THE LUMA TILE (relative to the block origin,
// biased by +33) and a right shift: each pixel lane reads its three samples straight from the tile, one LDS round
// trip per step and no cross-lane traffic. Shift 31 gives the all-zero prediction (quirk Q4); DC (bit 5) is
// handled by a wave-uniform branch taken only on steps that contain a DC block.
struct I4Lane {
  int pS;    // LDS byte address of tile sample "offset 0" (block origin - 33) for the group-0 block of step 0
  int pW;    // LDS byte address of this lane's pixel relative to the group-0 block origin
  int pEnt;  // LDS byte address of this lane's entry word relative to the group-0 block's first entry
  int pRes;  // LDS byte address of this lane's residual relative to the group-0 block's first residual
  bool grp1;
};

typedef __attribute__((address_space(3))) const uint8_t* lds_u8p;
typedef __attribute__((address_space(3))) uint8_t* lds_u8w;
typedef __attribute__((address_space(3))) const unsigned* lds_u32cp;
typedef __attribute__((address_space(3))) const int16_t* lds_i16cp;

template <int T>
__device__ __forceinline__ void i4_fetch(const I4Lane& L, unsigned& e, int& r) {
  constexpr int by0 = stepByLo(T), bx0 = T - 2 * by0;
  e = *(lds_u32cp)(uintptr_t)(unsigned)(L.pEnt + (by0 * 4 + bx0) * 64);
  r = *(lds_i16cp)(uintptr_t)(unsigned)(L.pRes + (by0 * 4 + bx0) * 32);
}

// One step of the Intra4x4 block wavefront (8.3.1.2, pred4x4.rs:10-360), statically scheduled. (e, r) = this
// lane's entry and residual for step T, fetched during the previous step; on return they are step T+1's.
// dcBlocks: raster bit mask of the blocks predicted in DC mode (wave-uniform).
// INTERIOR: macroblocks A and B exist, every availability test folds away.
template <int T, bool INTERIOR>
__device__ __forceinline__ void i4_step(const I4Lane& L, unsigned& e, int& r, unsigned dcBlocks, bool mbA, bool mbB) {
  constexpr int by0 = stepByLo(T), bx0 = T - 2 * by0;
  constexpr bool two = by0 + 1 <= stepByHi(T);
  constexpr int bx1 = two ? bx0 - 2 : bx0, by1 = two ? by0 + 1 : by0;  // group-1 block
  constexpr int oTile = TY(4 * bx0, 4 * by0);                          // tile byte offset of the block origin
  constexpr unsigned stepBlocks = (1u << (by0 * 4 + bx0)) | (1u << (by1 * 4 + bx1));
  unsigned eN = 0;
  int rN = 0;
  if (T < 9) i4_fetch<(T < 9 ? T + 1 : 9)>(L, eN, rN);
  if (two || !L.grp1) {
    const int a0 = L.pS + (int)((e >> 8) & 0xff), a1 = L.pS + (int)((e >> 16) & 0xff), a2 = L.pS + (int)(e >> 24);
    const int s0 = *(lds_u8p)(uintptr_t)(unsigned)(a0 + oTile);
    const int s1 = *(lds_u8p)(uintptr_t)(unsigned)(a1 + oTile);
    const int s2 = *(lds_u8p)(uintptr_t)(unsigned)(a2 + oTile);
    int pred = (int)((unsigned)(s0 + 2 * s1 + s2 + 2) >> (e & 31u));
    if (dcBlocks & stepBlocks) {
      // DC (pred4x4.rs:116-167): top row = the aligned dword above the block, left column = four bytes at x = -1
      const unsigned top = *(lds_u32cp)(uintptr_t)(unsigned)(L.pS + 1 + oTile);
      const int l0 = *(lds_u8p)(uintptr_t)(unsigned)(L.pS + 32 + oTile), l1 = *(lds_u8p)(uintptr_t)(unsigned)(L.pS + 64 + oTile);
      const int l2 = *(lds_u8p)(uintptr_t)(unsigned)(L.pS + 96 + oTile), l3 = *(lds_u8p)(uintptr_t)(unsigned)(L.pS + 128 + oTile);
      const int sumT = (int)sum4(top), sumL = l0 + l1 + l2 + l3;
      int dc;
      if (INTERIOR) {
        dc = (sumT + sumL + 4) >> 3;
      } else {
        const bool topAv = L.grp1 ? (by1 > 0 || mbB) : (by0 > 0 || mbB);
        const bool leftAv = L.grp1 ? (bx1 > 0 || mbA) : (bx0 > 0 || mbA);
        dc = (topAv && leftAv) ? (sumT + sumL + 4) >> 3 : leftAv ? (sumL + 2) >> 2 : topAv ? (sumT + 2) >> 2 : 128;
      }
      if (e & 32u) pred = dc;
    }
    *(lds_u8w)(uintptr_t)(unsigned)(L.pW + oTile) = (uint8_t)clip255(pred + r);
  }
  e = eN;
  r = rN;
}

// Intra4x4 macroblock: mode derivation (8.3.1.1), entry/residual packing, the 10-step pixel wavefront.
template <bool INTERIOR>
__device__ __forceinline__ int i4_macroblock(WaveScratch* ws, int t4eAddr, const int rl[4], int lane, int Tb, int Lb,
                                             unsigned prevFlags, unsigned long long remBits, bool mbA, bool mbB,
                                             bool mbC) {
  // mode grid: lanes 0..15 = by*4+bx (raster)
  const int mbx = lane & 3, mby = (lane >> 2) & 3;
  const int mzb = 8 * (mby >> 1) + 4 * (mbx >> 1) + 2 * (mby & 1) + (mbx & 1);
  const int rem = (int)((remBits >> (4 * mzb)) & 7ull);
  const bool prev = ((prevFlags >> mzb) & 1u) != 0;
  const bool unav = INTERIOR ? false : ((mbx == 0 && !mbA) || (mby == 0 && !mbB));
  // ---- Intra4x4PredMode (pred4x4.rs:363-427) as a relaxation over the block grid: after sweep k every
  // block with bx + by <= k is final
  int M = 2;
#pragma unroll
  for (int itr = 0; itr < 7; itr++) {
    int Am = dpp<QUAD(0, 0, 1, 2)>(M, M);
    if (mbx == 0) Am = Lb;
    const int Bm = dpp<ROW_SHR(4)>(Tb, M);  // lanes 0..3 of the row keep Tb
    const int pm = unav ? 2 : min(Am, Bm);
    M = prev ? pm : (rem < pm ? rem : rem + 1);
  }
  int Mp = M;
  if (!INTERIOR) {
    // quirk Q4: a mode whose reference samples are missing leaves the zero-initialised prediction
    const bool topAv = mby > 0 || mbB, leftAv = mbx > 0 || mbA;
    const int have = (topAv ? 1 : 0) | (leftAv ? 2 : 0) | ((topAv && leftAv) ? 4 : 0);
    const int req = (int)((0x217771021ull >> (4 * M)) & 7ull);  // per mode: bit0 top, bit1 left, bit2 corner
    if ((req & ~have) != 0) Mp = 9;
  }
  const unsigned dcBlocks = (unsigned)__builtin_amdgcn_ballot_w64(Mp == 2) & 0xffffu;
  // modes 3 and 7 read the top-right block's samples; without it T4..T7 := T3 (table rows 10 and 11).
  // Top-right exists: inside the macroblock for the blocks decoded after their top-right neighbour, above it
  // from macroblock B (C for the last column).
  {
    const unsigned trBlocks = 0x5750u | (INTERIOR ? 0xFu : ((mbB ? 0x7u : 0u) | (mbC ? 0x8u : 0u)));
    if (!((trBlocks >> (lane & 15)) & 1u)) Mp = Mp == 3 ? 10 : Mp == 7 ? 11 : Mp;
  }
  // ---- entry + residual per pixel. Strip lane (block sb, column sq) owns pixels (x = sq, y = 0..3); both arrays
  // are laid out [block (raster)][x][y], so a lane's four entries / residuals are contiguous
  {
    const int sq = lane & 3, sb = lane >> 2;
    const int sbx = ((sb >> 1) & 2) | (sb & 1), sby = ((sb >> 2) & 2) | ((sb >> 1) & 1);
    const int g = sby * 4 + sbx;
    const int mode = shfl(Mp, g);
    const u32x4 te = *LDSP(lds_u4p, t4eAddr + mode * 64 + sq * 16);
    *(u32x4*)&ws->resTe[g * 16 + sq * 4] = te;
    // (clamped: clip255(pred + r) only sees r through [-255, 255], so 16 bits carry any int32 residual exactly)
    const int r0 = min(max(rl[0], -512), 511), r1 = min(max(rl[1], -512), 511);
    const int r2 = min(max(rl[2], -512), 511), r3 = min(max(rl[3], -512), 511);
    uint2 rp;
    rp.x = __builtin_amdgcn_perm((unsigned)r1, (unsigned)r0, 0x05040100u);
    rp.y = __builtin_amdgcn_perm((unsigned)r3, (unsigned)r2, 0x05040100u);
    *(uint2*)&ws->resS[g * 16 + sq * 4] = rp;
  }
  // ---- per-lane addresses of the pixel organisation
  I4Lane L;
  {
    const int li = lane & 15, g = (lane >> 4) & 1;
    const int px = li & 3, py = li >> 2;
    const int tileBase = (int)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ws->tileY;
    const int rtBase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned*)ws->resTe;
    const int rsBase = (int)(uintptr_t)(__attribute__((address_space(3))) int16_t*)ws->resS;
    // group 1 works on block (bx-2, by+1): +4 tile rows, -8 columns = +120 bytes; +2 raster blocks
    L.pS = tileBase - 33 + 120 * g;
    L.pW = tileBase + py * 32 + px + 120 * g;
    L.pEnt = rtBase + 4 * (px * 4 + py) + 128 * g;
    L.pRes = rsBase + 2 * (px * 4 + py) + 64 * g;
    L.grp1 = g != 0;
  }
  WAVE_SYNC();
#ifndef DRYV_SKIP_CHAIN  // tuning only
  if (lane < 32) {
    unsigned e;
    int r;
    i4_fetch<0>(L, e, r);
    i4_step<0, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<1, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<2, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<3, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<4, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<5, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<6, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<7, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<8, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
    WAVE_SYNC();
    i4_step<9, INTERIOR>(L, e, r, dcBlocks, mbA, mbB);
  }
#endif
  WAVE_SYNC();
  return M;
}

// relaxed agent-scope accesses: global_load/store ... sc1 (served by / written through L2, bypassing L1)
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostic build only (-DDRYV_PHASE_PROFILE, tools/phase_profile.py): per-wave cycle sums per phase,
// written to a buffer of their own. The shipped library contains none of this.
#ifdef DRYV_PHASE_PROFILE
#define PHASE_STAMP(i)                                                   \
  do {                                                                   \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
    phaseAcc[i] += now_ - phaseT;                                        \
    phaseT = now_;                                                       \
  } while (0)
#else
#define PHASE_STAMP(i) do { } while (0)
#endif

// Coefficient prefetch: global -> LDS DMA issued from inline asm (cdna_hip_programming.md §5.7). hipcc treats an
// LDS-DMA it knows about as a pending write to ALL of LDS and waits vmcnt(0) at the next LDS access, which turns
// the prefetch into a synchronous load. Hidden from the compiler it is retired by the counted s_waitcnt at the top
// of the macroblock loop. (vmcnt is in-order: a hidden operation can only make the compiler's own counted waits
// stricter, never looser. Buffer form: scalar resource + scalar offset + one per-lane VGPR offset.)
__device__ __forceinline__ void dma_coefficients(int ldsAddr, unsigned voff, i32x4 rsrc, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(ldsAddr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

// one dword per active lane from a per-lane global address, write-through-coherent (sc1), to LDS ldsAddr + 4*lane;
// hidden from the compiler for the same reason (and because a conditionally issued VGPR load leaves its
// destination "possibly pending" in hipcc's scoreboard, which then waits vmcnt(0) before reusing that register)
__device__ __forceinline__ void dma_window(int ldsAddr, const unsigned* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1" : : "s"(ldsAddr), "v"(src) : "memory", "m0");
}

#define BAND 4     // rows per band = waves that hand rows to each other through LDS
#ifdef DRYV_NO_I8X8  // tuning only: what the Intra8x8 paths cost the other macroblock kinds (registers, code size)
constexpr bool HAS_I8 = false;
#else
constexpr bool HAS_I8 = true;
#endif
#ifndef DRYV_POLL_SLEEP
#define DRYV_POLL_SLEEP 8  // s_sleep argument (x64 clocks) between two polls of a neighbouring wave's LDS counter
#endif
#ifndef DRYV_GPOLL_SLEEP
#define DRYV_GPOLL_SLEEP 2  // the same between two polls of a progress word in L2 (band boundaries)
#endif
#ifndef DRYV_WPS
#define DRYV_WPS 6  // resident waves per SIMD the kernel is compiled for (80 VGPRs: no spills; 8 (64 VGPRs, 4 spilled) measured the same on C3)
#endif
#ifndef WG_BANDS
#define WG_BANDS 2 // independent bands per workgroup: they share nothing but the constant tables (5.7 KB)
#endif
#ifndef RING_K
#define RING_K 8   // macroblocks of bottom line each row keeps in LDS for the row below (power of two)
#endif
struct BandShared {
  unsigned prog[BAND];  // prog[w]: macroblocks of wave w's row finished and present in its ring
  unsigned cons[BAND];  // cons[w]: macroblocks whose neighbour window wave w has copied out of wave w-1's ring
  unsigned task;    // the band being processed
  unsigned seq;     // claim number of `task` (band-local barrier: waves 1..3 wait for it to advance)
  unsigned arrive;  // waves of the band that finished claim number seq-1, cumulative
  unsigned pad;
  uint8_t ringY[BAND][RING_K * 16];     // bottom luma line, macroblock e at (e % RING_K) * 16
  uint8_t ringC[BAND][2][RING_K * 8];   // bottom chroma lines
  unsigned ringM[BAND][RING_K];         // bottom-row prediction modes
};

#define UPY(k) (4 + (k))        // byte index in WaveScratch::up of luma sample x = k of the row above
#define UPC(pl, k) (36 + 16 * (pl) + (k))
#define UPM 64

// per-workgroup LDS tables
#define LT_LSQ 0        // u32 [52][16]  LevelScale4x4(qp%6,i,j) << max(qp/6-4, 0)       (transform.rs:147-152)
#define LT_QINFO 3328   // u16 [52]      rounding << 8 | right shift of the same formula
#define LT_QPC 3440     // u8  [2][52]   QP'c for Cb / Cr as a function of QPY             (transform.rs:194-216)
#define LT_LS8 3552     // u16 [6][64]   LevelScale8x8
#define LT_T4E 4320     // u32 [12][16]  Intra4x4 prediction entries [mode][x][y] (9: zero, 10/11: modes 3/7 without top-right)
#define LT_T8 5088      // u8  [9][64]   Intra8x8 gather table
#define LT_ZZ8 5664     // u8  [64]      raster -> 8x8 zig-zag list index
#define LT_THR 5760      // u16 [52]      KParams::thr_row: int32 exactness bound on |coefficient| per QPY
#define LT_END 5888     // (64-byte aligned: see WaveScratch)

__global__ void __launch_bounds__(64 * BAND * WG_BANDS, DRYV_WPS)  // (threads, waves per SIMD)
recon_kernel(const KParams P, const dryv_mb_desc* __restrict__ mbs, const int16_t* __restrict__ coeffs,
             uint8_t* __restrict__ yuv, unsigned* __restrict__ status, unsigned* __restrict__ rowProg,
             unsigned* __restrict__ rowModes, unsigned* __restrict__ taskCounter
#ifdef DRYV_PHASE_PROFILE
             , unsigned long long* __restrict__ phaseOut
#endif
             ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane0 = threadIdx.x & 63;
  int lane = lane0;
  const int wgWave = rfl(threadIdx.x >> 6);
  const int wave = wgWave & (BAND - 1);  // position in the band
  const int W = P.W, H = P.H;
  // Wave priority = position in the band (s_setprio): a row waits for the row above it, so the consumer goes first when
  // both can issue (the same observation as for the band kernel's roles, recon_band.hip). C3 (100 x 4K, 8x8 transform):
  // no priorities 3.92 ms, the upper rows first 3.82-3.83, the lower rows first 3.77.
#ifndef DRYV_ROW_PRIO
#define DRYV_ROW_PRIO 1
#endif
  if (DRYV_ROW_PRIO) {
    if (wave == 3) __builtin_amdgcn_s_setprio(3);
    else if (wave == 2) __builtin_amdgcn_s_setprio(2);
    else if (wave == 1) __builtin_amdgcn_s_setprio(1);
  }

  // ---- LDS tables (built once per workgroup) -------------------------------------------------------
  unsigned* lsq = (unsigned*)(lds + LT_LSQ);
  uint16_t* qinfo = (uint16_t*)(lds + LT_QINFO);
  uint8_t* qpcT = lds + LT_QPC;
  uint16_t* ls8 = (uint16_t*)(lds + LT_LS8);
  unsigned* t4e = (unsigned*)(lds + LT_T4E);
  uint8_t* t8 = lds + LT_T8;
  uint8_t* zz8i = lds + LT_ZZ8;
  WaveScratch* ws = (WaveScratch*)(lds + LT_END) + wgWave;
  BandShared* bs = (BandShared*)(lds + LT_END + WG_BANDS * BAND * sizeof(WaveScratch)) + (wgWave / BAND);
  if (wave == 0 && lane0 == 0) {
    bs->seq = 0;
    bs->arrive = 0;
  }

  for (int i = threadIdx.x; i < 52 * 16; i += blockDim.x) {
    const int qp = i >> 4, qd = qp / 6, qm = qp - 6 * qd;
    lsq[i] = (unsigned)P.ls4[qm * 16 + (i & 15)] << max(qd - 4, 0);
  }
  for (int i = threadIdx.x; i < 52; i += blockDim.x) {
    const int qd = i / 6;
    qinfo[i] = (uint16_t)((qd < 4 ? (1 << (3 - qd)) << 8 : 0) | max(4 - qd, 0));
  }
  for (int i = threadIdx.x; i < 104; i += blockDim.x) {
    // 8.5.8: qPI = Clip3(0, 51, QPY + offset); QPc = qPI < 30 ? qPI : QPCS[qPI - 30] (table 8-15)
    const int qpi = min(max((i % 52) + (i < 52 ? P.cqo_cb : P.cqo_cr), 0), 51);
    const int k = qpi - 30;  // qPI - QPc for qPI = 30..51, 4 bits each
    const int delta = k < 0 ? 0 : k < 16 ? (int)((0x7765544332221111ull >> (4 * k)) & 15ull) : (int)((0xCBA998u >> (4 * (k - 16))) & 15u);
    qpcT[i] = (uint8_t)(qpi - delta);
  }
  for (int i = threadIdx.x; i < 384; i += blockDim.x) ls8[i] = P.ls8[i];
  for (int i = threadIdx.x; i < 192; i += blockDim.x) {
    // t4e[mode][x][y] = shift | DC flag << 5 | three tile offsets (relative to the block origin, +33) << 8/16/24.
    // P.t4 names a sample by its index j on the line E (0..3 = L3..L0, 4 = corner, 5..12 = T0..T7) and says
    // whether the pixel is E[j], the 3-tap or the 2-tap value there (ends replicated: pred4x4.rs "3*a + b" cases)
    const int m = i >> 4, x = (i >> 2) & 3, y = i & 3;
    const int mode = m < 10 ? m : (m == 10 ? 3 : 7), jmax = m < 10 ? 12 : 8;
    unsigned v;
    if (mode == 9) v = 31u;
    else if (mode == 2) v = 31u | 32u;
    else {
      const int en = P.t4[mode * 16 + y * 4 + x], j = en & 31, sel = en >> 5;
      const int ja = sel == 1 ? j - 1 : j, jb = sel == 2 ? j + 1 : j, jc = sel == 1 ? j + 1 : j;
      auto pos = [jmax](int k) { k = min(max(k, 0), jmax); return (unsigned)(k <= 3 ? (4 - k) * 32 : k - 4); };
      v = 2u | (pos(ja) << 8) | (pos(jb) << 16) | (pos(jc) << 24);
    }
    t4e[i] = v;
  }
  for (int i = threadIdx.x; i < 576; i += blockDim.x) t8[i] = P.t8[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) zz8i[i] = P.zz8i[i];
  for (int i = threadIdx.x; i < 52; i += blockDim.x) ((uint16_t*)(lds + LT_THR))[i] = P.thr_row[i];
  __syncthreads();  // the only workgroup-level synchronisation: waves are independent from here on

  const size_t frameBytes = (size_t)W * H * 384;
  const int pitchY = W * 16, pitchC = W * 8;
#ifdef DRYV_PHASE_PROFILE
  unsigned long long phaseAcc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long phaseT = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif

  // ---- per-lane values kept for the whole kernel ---------------------------------------------------
  const int ldsBase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int wsAddr = (int)(uintptr_t)(__attribute__((address_space(3))) WaveScratch*)ws;
  const unsigned bandsPerFrame = (unsigned)(H + BAND - 1) / BAND;
  const unsigned totalBands = (unsigned)P.n_frames * bandsPerFrame;
  // The waves of a band synchronise among themselves only (LDS counters, no workgroup barrier: the other band of
  // the workgroup runs on its own schedule). Claim k: every wave bumps `arrive` when it is done with claim k-1;
  // wave 0 waits for all four, resets the hand-off counters, takes the next band off the queue and publishes it
  // with seq = k; the others wait for seq.
  unsigned claimNo = 0;
  for (;;) {
    claimNo++;
    if (wave == 0) {
      if (lane0 == 0) {
        while (__hip_atomic_load(&bs->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < BAND * (claimNo - 1))
          __builtin_amdgcn_s_sleep(1);
      }
      WAVE_SYNC();
      if (lane0 < 2 * BAND) bs->prog[lane0] = 0;  // prog[] and cons[] are adjacent
      // Every band, the first one included, comes off the one queue: a band's predecessor (same frame, band
      // above) has a smaller number, so it was claimed earlier by a workgroup that is running or done -- no
      // deadlock however few workgroups the device keeps resident.
      if (lane0 == 0) bs->task = atomicAdd(taskCounter, 1u);
      WAVE_SYNC();
      if (lane0 == 0) __hip_atomic_store(&bs->seq, claimNo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (lane0 == 0) {
        while (__hip_atomic_load(&bs->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < claimNo)
          __builtin_amdgcn_s_sleep(1);
      }
      WAVE_SYNC();
    }
    const unsigned task = (unsigned)rfl((int)bs->task);  // wave-uniform: everything derived from it stays scalar
    PHASE_STAMP(0);  // claim
    if (task >= totalBands) break;
    const int bandRow = (int)(task / (unsigned)P.n_frames);
    const int f = (int)(task - (unsigned)bandRow * (unsigned)P.n_frames);
    const int r = bandRow * BAND + wave;
#ifdef DRYV_PHASE_PROFILE
    const unsigned long long rowT0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same on every XCD
#endif
    if (r < H) {
    // where the row above comes from, and who reads this row's bottom line
    const bool upLds = wave > 0;                       // row r-1 is wave-1 of this band: LDS ring
    const bool upGlobal = wave == 0 && r > 0;          // row r-1 is the last row of the band above: through L2
    const bool toLds = wave + 1 < BAND && r + 1 < H;   // row r+1 is the next wave of this band
    const bool toGlobal = wave + 1 == BAND && r + 1 < H;
    const size_t mbBase = (size_t)f * W * H + (size_t)r * W;
    uint8_t* planeY = yuv + (size_t)f * frameBytes;
    unsigned* myProg = rowProg + (size_t)f * H + r;
    const unsigned* upProg = myProg - 1;
    unsigned* myModes = rowModes + mbBase;
    const bool mbB = r > 0;
    unsigned upDone = 0;  // what we know of the row above's progress
    int consKnown = 0;    // ... and of the row below's consumption of our ring

    // ---- per-row per-lane pointers, advanced by a constant stride per macroblock -----------------
    // luma store: lane (block sb, row sq of the block) owns pixels x = 4*sbx..+3 of pixel row 16r + 4*sby + sq
    // chroma store (lanes 0..31): plane cpl, pixel row 8r + 4*ccy + sq, x = 4*ccx..+3
    // neighbour window: lanes 0..7 Y x = -4..27 of pixel row 16r-1; 8..11 Cb x = -4..11 of row 8r-1; 12..15 Cr; 16 modes
    // (32-bit byte offsets from the frame's Y plane: a frame is < 4 GB; the modes live in their own buffer)
    unsigned yOff, cOff, wOff;
    {
      int lr = lane0;
      asm volatile("" : "+v"(lr));  // recompute per row rather than keep a dozen per-lane constants alive
      const int sq = lr & 3, sb = lr >> 2;
      const int sbx = ((sb >> 1) & 2) | (sb & 1), sby = ((sb >> 2) & 2) | ((sb >> 1) & 1);
      const int cpl = (lr >> 4) & 1, ccb = (lr >> 2) & 3;
      const unsigned offCb = (unsigned)W * H * 256u, offCr = offCb + (unsigned)W * H * 64u;
      yOff = (unsigned)(r * 16 + 4 * sby + sq) * pitchY + 4 * sbx;
      cOff = (cpl ? offCr : offCb) + (unsigned)(r * 8 + 4 * (ccb >> 1) + sq) * pitchC + 4 * (ccb & 1);
      const unsigned wY = (unsigned)(r * 16 - 1) * pitchY - 4 + 4 * lr;
      const unsigned wC = (lr < 12 ? offCb : offCr) + (unsigned)(r * 8 - 1) * pitchC - 4 + 4 * (lr & 3);
      wOff = lr < 8 ? wY : wC;
    }

    // coefficients of macroblock 0 of the row: global -> LDS DMA, 48 lanes x 16 B
    const uint8_t* crow = (const uint8_t*)(coeffs + mbBase * 384);
    // buffer resource over this row's coefficients (raw buffer, 32-bit data format: dword 3 = 0x00020000 on gfx9)
    const unsigned long long crowBits = (unsigned long long)(uintptr_t)crow;
    const i32x4 crsrc = {(int)(unsigned)crowBits, (int)(unsigned)((crowBits >> 32) & 0xffffull), W * 768, 0x00020000};
    if (lane0 < 48) dma_coefficients(wsAddr, (unsigned)lane0 * 16u, crsrc, 0);  // WaveScratch::coef is the first member
    uint4 desc = *(const uint4*)(mbs + mbBase);

    int Mprev = 2;  // derived modes of the macroblock to the left, on the 4x4 grid (lanes 0..15)
    int nStores = 0;  // plain stores issued after the previous macroblock's coefficient DMA (wave-uniform)
    const unsigned offCbS = (unsigned)W * H * 256u, offCrS = offCbS + (unsigned)W * H * 64u;

    for (int mx = 0; mx < W; mx++) {
      // Per-lane constants are recomputed per macroblock from an opaque lane id: keeping them (and the
      // dozens of LDS addresses derived from them) live across the loop costs ~90 VGPRs, i.e. half the
      // occupancy; recomputing costs a few dozen VALU.
      asm volatile("" : "+v"(lane));
      // "strip" organisation: 4 lanes per 4x4 block, sb = blkIdx (z-order), sq = row (then column) in the block
      const int sq = lane & 3, sb = lane >> 2;
      const int sbx = ((sb >> 1) & 2) | (sb & 1), sby = ((sb >> 2) & 2) | ((sb >> 1) & 1);
      const unsigned selA = (lane & 1) ? 0x03070105u : 0x06020400u, selB = (lane & 2) ? 0x03020706u : 0x05040100u;
      // chroma strips: lanes 0..31 = plane*16 + blk*4 + sq (lanes 32..63 mirror them and are never stored)
      const int cpl = (lane >> 4) & 1, ccb = (lane >> 2) & 3, ccx = ccb & 1, ccy = ccb >> 1;
      // LDS transpose slots of the residual passes (they live in the resTe area, free at that point)
      const int trSw = (lane >> 3) & 3;  // (the scratch base is 64-byte aligned: ^ only touches the chunk bits)
      const int trW = wsAddr + 768 + (lane & ~3) * 16 + ((sq ^ trSw) << 4);
      const int trR = wsAddr + 768 + (lane & ~3) * 16 + (trSw << 4) + sq * 4;
      // LDS byte addresses of the four coefficients c[sq][0..3] of an Intra4x4 luma block (list base 16*blk):
      // inverse zig-zag (frame/mod.rs:185-209) as a per-lane gather. Other list layouts differ by a per-lane delta.
      const unsigned zz = (unsigned)((0xFEA9DB83C7426510ull >> (16 * sq)) & 0xffffull);  // list indices of row sq
      const int gaB = wsAddr + 32 * sb;
      const int ga0 = gaB + 2 * (int)(zz & 15u), ga1 = gaB + 2 * (int)((zz >> 4) & 15u);
      const int ga2 = gaB + 2 * (int)((zz >> 8) & 15u), ga3 = gaB + 2 * (int)(zz >> 12);

      const unsigned d0 = rfl(desc.x), d1 = rfl(desc.y), d2 = rfl(desc.z), d3 = rfl(desc.w);
      int kind = d0 & 0xff;
      const int i16mode = (d0 >> 8) & 0xff;
      const int cmode = (d0 >> 16) & 0xff;
      int qp = (d0 >> 24) & 0xff;
      const unsigned prevFlags = d1 & 0xffff;
      const unsigned long long remBits = ((unsigned long long)(d1 >> 16)) | ((unsigned long long)d2 << 16) |
                                         ((unsigned long long)(d3 & 0xffff) << 48);
      if (kind > 2 || qp > 51 || i16mode > 3 || cmode > 3) {
        if (lane == 0) atomicOr(status, 1u);
        kind = 3;
        qp = 0;
      }
      const bool mbA = mx > 0, mbC = mbB && (mx + 1 < W);

      // The DMA of this macroblock's coefficients must have landed. vmcnt is ONE in-order counter for loads,
      // stores and LDS-DMA, so the wait is counted: after the DMA a wave inside a band issued nStores (0..2) plain
      // row-segment stores, and "all but the nStores youngest operations done" covers the DMA without waiting for
      // them. A band's LAST row drains completely instead: its write-through stores of macroblock mx-1 must have
      // retired before that macroblock is published to the band below, and publishing early keeps the ramp of
      // the 27-row wavefront at launch short.
      PHASE_STAMP(1);  // record decode, constants
      // (the builtin, not inline asm: hipcc's wait-count pass then accounts for it)
      if (toGlobal || nStores == 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      else if (nStores == 1) __builtin_amdgcn_s_waitcnt(0x0F71);        // vmcnt(1)
      else __builtin_amdgcn_s_waitcnt(0x0F72);                           // vmcnt(2)
      asm volatile("" ::: "memory");
      PHASE_STAMP(2);  // wait for the coefficient DMA
      if (toGlobal && mx > 0 && lane == 0) st_sc1(myProg, (unsigned)mx);
      // Band boundary (wave 0 only): the row above lives in another workgroup. Start the poll of its progress
      // now and -- when what we already know of its progress covers this macroblock -- the loads of the neighbour
      // window too; both L2 round trips then hide under the residuals. Lanes 0..15 pixels, lane 16 the modes word
      // (window), lane 17 the progress word; all land in WaveScratch::up.
      // Two levels of "the row above is far enough": need1 = it has finished macroblock mx (neighbour B, and D/A
      // before it): enough for the modes, chroma, Intra16x16 and all of Intra4x4/8x8 except the one block that reads
      // the top-RIGHT macroblock's samples (C), which wants need2 = macroblock mx+1. Waiting for need2 only there
      // (and never for Intra16x16) lets the rows of a frame follow each other at ~1 instead of 2 macroblocks.
#ifdef DRYV_NO_WAIT  // tuning only (tools/sweep.py): the dependency-free bound of the same instruction stream
      const unsigned need1 = 0, need2 = 0;
#else
      const unsigned need1 = (unsigned)min(mx + 1, W), need2 = (unsigned)min(mx + 2, W);
#endif
      bool trDone;  // the window's x = 16..23 (top-right) part is valid
      const unsigned upDoneAtStart = upDone;
      const bool winEarly = upGlobal && upDone >= need1;
      const bool pollEarly = upGlobal && upDone < (unsigned)W;
      if (upGlobal) {
        const unsigned* src = lane < 16 ? (const unsigned*)(planeY + wOff) : lane == 16 ? myModes - W + mx : upProg;
        const bool act = lane < 17 ? winEarly : (lane == 17 && pollEarly);
        if (act) dma_window(wsAddr + (int)offsetof(WaveScratch, up), src);
      }
      WAVE_SYNC();

      // This kernel computes in int32 throughout; the reference in 64-bit isize. A macroblock with a coefficient beyond
      // the bound under which the two provably agree (KParams::thr_row; never reached by a conformant stream) is
      // reported (status bit 1): the library then runs the batch again on the band kernel's wide build.
      {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const unsigned* cw = (const unsigned*)ws->coef;
        const s16x2 a = __builtin_bit_cast(s16x2, cw[lane]), b = __builtin_bit_cast(s16x2, cw[lane + 64]),
                    c = __builtin_bit_cast(s16x2, cw[lane + 128]);
        const s16x2 mx2 = __builtin_elementwise_max(__builtin_elementwise_max(a, b), c);
        const s16x2 mn2 = __builtin_elementwise_min(__builtin_elementwise_min(a, b), c);
        const int hi = max((int)mx2.x, (int)mx2.y), lo = min((int)mn2.x, (int)mn2.y);
        const int thr = (int)((const uint16_t*)(lds + LT_THR))[qp];
        if (thr != 0xFFFF && max(hi, -lo) > thr) atomicOr(status, 2u);
      }

      // ================= residuals (need no neighbour: done before waiting for the row above) =====
      int rl[4] = {0, 0, 0, 0};  // luma residual, column organisation (kinds 0 and 2)
      int rc[4] = {0, 0, 0, 0};  // chroma residual
#ifdef DRYV_SKIP_RESID  // tuning only
      if (kind > 3) {
#else
      if (kind != 3) {
#endif
        // ---- chroma: DC 2x2 (8.5.11, trans_chroma.rs:369-415) on lanes plane*16 + blk*4 (+ sq), then AC
        {
          const int qc = (int)qpcT[cpl * 52 + qp];  // QP'c of this lane's plane
          const unsigned qi = qinfo[qc];
          const int lsA = ldsBase + LT_LSQ + qc * 64;
          int x = ws->coef[256 + cpl * 64 + ccb];
          // (both shifts execute on every lane: a DPP op under a divergent branch would see masked sources)
          int up = dpp<ROW_SHL(4)>(x, x), dn = dpp<ROW_SHR(4)>(x, x);
          int o = (ccb & 1) ? dn : up;  // lane ^ 4
          x = (ccb & 1) ? o - x : x + o;
          up = dpp<ROW_SHL(8)>(x, x);
          dn = dpp<ROW_SHR(8)>(x, x);
          o = (ccb & 2) ? dn : up;  // lane ^ 8
          x = (ccb & 2) ? o - x : x + o;
          // ((f * LS) << qp/6) >> 5 == (f * (LS << max(qp/6-4,0))) >> (max(4-qp/6,0) + 1)   (trans_chroma.rs:413)
          const int dcC = (x * (int)*LDSP(lds_u32p, lsA)) >> ((int)(qi & 0xff) + 1);
          // list of chroma block (plane, blk): DC slot + 15 AC at 256 + 64*plane + 4 + 15*blk - 1
          const int dl = 2 * (259 + 64 * cpl + 15 * ccb) - 32 * sb;
          residual4x4_quad(ga0 + dl, ga1 + dl, ga2 + dl, ga3 + dl, sq == 0, dcC, lsA + sq * 16, qi, trW, trR, rc);
        }
        const unsigned qiY = qinfo[qp];
        const int lsY = ldsBase + LT_LSQ + qp * 64;
        if (kind == 0) {
          residual4x4_quad(ga0, ga1, ga2, ga3, false, 0, lsY + sq * 16, qiY, trW, trR, rl);
        } else if (kind == 2) {
          // Intra16x16 luma DC: 8.5.10 (pred16x16.rs:428-482). The lanes of block (bx,by) load c[by][bx];
          // f = A c A = P (H c H) P^T with H the natural 4-point Hadamard and A row i = H row s(i),
          // s = [0,2,3,1]: block (bx,by) takes g[s(by)][s(bx)].
          const int dcZZ = (int)((0xFEA9DB83C7426510ull >> (4 * (sby * 4 + sbx))) & 15ull);
          const int sxx = (0x1320 >> (4 * sbx)) & 3, syy = (0x1320 >> (4 * sby)) & 3;
          const int dcSrcLane = 4 * (8 * (syy >> 1) + 4 * (sxx >> 1) + 2 * (syy & 1) + (sxx & 1));
          int x = ws->coef[dcZZ];
          int o = shfl(x, lane ^ 4);
          x = (sbx & 1) ? o - x : x + o;
          o = shfl(x, lane ^ 16);
          x = (sbx & 2) ? o - x : x + o;
          o = shfl(x, lane ^ 8);
          x = (sby & 1) ? o - x : x + o;
          o = shfl(x, lane ^ 32);
          x = (sby & 2) ? o - x : x + o;
          const int fv = shfl(x, dcSrcLane);
          // qp >= 36: (f*LS) << (qp/6-6), else (f*LS + 2^(5-qp/6)) >> (6-qp/6)   (pred16x16.rs:465-479)
          //   == (f * (LS << max(qp/6-4,0)) + max(4*rnd, 2)) >> (shr + 2) with (rnd, shr) of the 4x4 formula
          const int dcY = (fv * (int)*LDSP(lds_u32p, lsY) + max(4 * (int)(qiY >> 8), 2)) >> ((int)(qiY & 0xff) + 2);
          // list of luma block blk: DC slot + 15 AC at 16 + 15*blk - 1
          const int dl = 30 - 2 * sb;
          residual4x4_quad(ga0 + dl, ga1 + dl, ga2 + dl, ga3 + dl, sq == 0, dcY, lsY + sq * 16, qiY, trW, trR, rl);
        }
      }

      // 8x8 blocks: 8.5.13 (pred8x8.rs:51-150). lanes 0..31 = (blk8, row) then (blk8, column).
      if (HAS_I8 && kind == 1) {
        const int b8 = (lane >> 3) & 3, i = lane & 7;
        int dd[8], oo[8];
        if (lane < 32) {
          const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int c = ws->coef[b8 * 64 + zz8i[i * 8 + j]];
            const int prod = c * (int)ls8[qm * 64 + i * 8 + j];
            dd[j] = qp >= 36 ? prod * (1 << max(qd - 6, 0)) : ((prod + (1 << max(5 - qd, 0))) >> max(6 - qd, 0));
          }
          idct8(dd, oo);
#pragma unroll
          for (int j = 0; j < 8; j++) ws->g8[b8 * 64 + i * 8 + j] = oo[j];
        }
        WAVE_SYNC();
        if (lane < 32) {
          const int j = i;  // this lane now owns column j
#pragma unroll
          for (int k = 0; k < 8; k++) dd[k] = ws->g8[b8 * 64 + k * 8 + j];
          idct8(dd, oo);
        }
        WAVE_SYNC();  // g8 aliases resB: everything is read before the residuals are written
        if (lane < 32) {
          const int bx = b8 & 1, by = b8 >> 1;
#pragma unroll
          // (clamped: clip255(pred + r) only sees r through [-255, 255], so 16 bits carry any int32 residual exactly)
          for (int k = 0; k < 8; k++)
            ws->resB[(8 * by + k) * 16 + 8 * bx + i] = (int16_t)min(max((oo[k] + 32) >> 6, -512), 511);
        }
      }
      WAVE_SYNC();

      PHASE_STAMP(3);  // residuals
      // the early poll / window loads are drained BEFORE the next coefficient DMA is issued
      if (upGlobal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's hidden window / poll loads have landed
        if (pollEarly) upDone = max(upDone, (unsigned)rfl((int)((const unsigned*)ws->up)[17]));
      }
      asm volatile("" ::: "memory");
      // the coefficient buffer is free again: start the DMA of the next macroblock and fetch its record
      if (mx + 1 < W) {
#ifndef DRYV_SKIP_DMA  // tuning only
        if (lane < 48) dma_coefficients(wsAddr, (unsigned)lane * 16u, crsrc, (mx + 1) * 768);  // (opaque lane: not hoisted/spilled)
#endif
        desc = *(const uint4*)(mbs + mbBase + mx + 1);
      }
      asm volatile("" ::: "memory");

      // ================= wait for the row above, fetch the neighbour window ======================
      if (upLds) {
        // row above = wave-1 of this band: poll its LDS counter, copy the window out of its ring, tell it so
        // (upDone caches the last value read: while the row above is ahead, no LDS round trip is spent on the poll)
        while (upDone < need1) {
          upDone = (unsigned)rfl((int)__hip_atomic_load(&bs->prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (upDone < need1) __builtin_amdgcn_s_sleep(DRYV_POLL_SLEEP);
        }
        trDone = upDone >= need2;
        PHASE_STAMP(4);  // poll the row above
        if (lane < 17) {
          // dword j of the window: Y (lanes 0..7) covers x = 4j-4..4j-1, Cb/Cr (lanes 8..11 / 12..15) likewise with 8-pixel macroblocks
          const int jy = lane, jc = lane & 3;
          const int ey = mx + ((jy + 3) >> 2) - 1, ec = mx + ((jc + 1) >> 1) - 1;
          const uint8_t* src = lane < 8    ? &bs->ringY[wave - 1][(ey & (RING_K - 1)) * 16 + ((jy + 3) & 3) * 4]
                               : lane < 16 ? &bs->ringC[wave - 1][(lane >> 2) & 1][(ec & (RING_K - 1)) * 8 + ((jc + 1) & 1) * 4]
                                           : (const uint8_t*)&bs->ringM[wave - 1][mx & (RING_K - 1)];
          ((unsigned*)ws->up)[lane] = *(const unsigned*)src;
        }
        WAVE_SYNC();
        if (lane == 0) __hip_atomic_store(&bs->cons[wave], (unsigned)(mx + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (upGlobal && !winEarly) {
        while (upDone < need1) {
          unsigned v = 0;
          if (lane == 0) v = ld_sc1(upProg);
          upDone = (unsigned)rfl((int)v);
          if (upDone < need1) __builtin_amdgcn_s_sleep(DRYV_GPOLL_SLEEP);
        }
        trDone = upDone >= need2;
        PHASE_STAMP(4);  // poll the row above
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler only: keep the loads below the poll
        if (lane < 17) {
          const unsigned* src = lane < 16 ? (const unsigned*)(planeY + wOff) : myModes - W + mx;
          ((unsigned*)ws->up)[lane] = ld_sc1(src);
        }
      } else {
        trDone = upDoneAtStart >= need2;  // (early window: valid as far as the row above was known to be then)
      }
      WAVE_SYNC();
      // Intra4x4 / Intra8x8 with a top-right macroblock: make sure its bottom line is there before the luma blocks
      const bool lateTR = (kind == 0 || (HAS_I8 && kind == 1)) && mbC && !trDone;
      wOff += lane < 8 ? 16u : 8u;
      const uint8_t* up = ws->up;
      PHASE_STAMP(5);  // neighbour window fetch

      // ================= chroma: 8.3.4 (trans_chroma.rs:96-366) ===================================
      // lane = plane*16 + blk*4 + column sq; pixels (x = 4*ccx + sq, y = 4*ccy + k)
      unsigned cword;  // after the byte transpose: row sq of the block, 4 pixels
      {
        const uint8_t* upc = up + UPC(cpl, 0);
        const int x = 4 * ccx + sq;
        int pr[4] = {0, 0, 0, 0};
        if (kind != 3) {
          if (cmode == 0) {
            const unsigned tw = *(const unsigned*)&upc[4 * ccx];
            const unsigned lw = *(const unsigned*)&ws->leftC[cpl][4 * ccy];
            const int st = (int)sum4(tw), sl = (int)sum4(lw);
            // trans_chroma.rs:168-286 incl. quirk Q2 (`> 0` where the spec means "available"), as selects:
            //   blocks (0,0),(4,4): both -> 8-sample mean; left only -> left; top only needs every top sample > 0
            //   block (4,0): top, else left if its 4th sample > 0;  block (0,4): left if its 4th sample > 0, else top if ...
            const bool tAll = mbB && bytes_nonzero(tw);
            const bool t3 = mbB && (tw >> 24) != 0, l3 = mbA && (lw >> 24) != 0;
            const int vT = (st + 2) >> 2, vL = (sl + 2) >> 2, vB = (st + sl + 4) >> 3;
            const int vDiag = (mbA && mbB) ? vB : mbA ? vL : tAll ? vT : 128;
            const int vTR = mbB ? vT : l3 ? vL : 128;
            const int vBL = l3 ? vL : t3 ? vT : 128;
            const int v = ccx == ccy ? vDiag : ccx == 1 ? vTR : vBL;
            pr[0] = pr[1] = pr[2] = pr[3] = v;
          } else if (cmode == 1) {  // horizontal
            if (mbA) {
              const unsigned lw = *(const unsigned*)&ws->leftC[cpl][4 * ccy];
#pragma unroll
              for (int k = 0; k < 4; k++) pr[k] = (int)((lw >> (8 * k)) & 0xff);
            }
          } else if (cmode == 2) {  // vertical
            if (mbB) pr[0] = pr[1] = pr[2] = pr[3] = upc[x];
          } else if (mbA && mbB) {  // plane: :319-363
            // lanes (per plane) 0..3: horizontal terms, 4..7: vertical terms
            const int l16 = lane & 15, k = l16 & 3;
            const int ha = upc[4 + k], hb = upc[2 - k];  // 2-k = -1 -> corner
            const int va = ws->leftC[cpl][4 + k], vb = k == 3 ? (int)upc[-1] : (int)ws->leftC[cpl][(2 - k) & 7];
            int term = l16 < 4 ? (k + 1) * (ha - hb) : l16 < 8 ? (k + 1) * (va - vb) : 0;
            term += xor1(term);
            term += xor2(term);
            const int hs = shfl(term, lane & 48), vs = shfl(term, (lane & 48) + 4);
            const int a = 16 * ((int)ws->leftC[cpl][7] + (int)upc[7]);
            const int b = (34 * hs + 32) >> 6, c = (34 * vs + 32) >> 6;
            const int base = a + b * (x - 3) + c * (4 * ccy - 3) + 16;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) pr[k2] = clip255((base + c * k2) >> 5);
          }
        }
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) w |= (unsigned)clip255(pr[k] + rc[k]) << (8 * k);
        cword = quad_transpose_bytes(w, selA, selB);  // lane sq now holds row sq: pixels x = 4*ccx .. +3
      }

      PHASE_STAMP(6);  // chroma prediction
      // ================= luma ======================================================================
      unsigned yword = 0;  // row strip of this lane's block, 4 pixels
      int Mcur = 2;        // derived modes on the 4x4 grid (lanes 0..15)
#ifdef DRYV_SKIP_LUMA  // tuning only
      if (kind == 7) {
#else
      if (kind == 2) {
#endif
        // Intra16x16: 8.3.3 (pred16x16.rs:79-425); lane = block sb, column sq: x = 4*sbx+sq, y = 4*sby+k
        const int x = 4 * sbx + sq;
        int pr[4] = {0, 0, 0, 0};
        if (i16mode == 0) {
          if (mbB) pr[0] = pr[1] = pr[2] = pr[3] = up[UPY(x)];
        } else if (i16mode == 1) {
          if (mbA) {
            const unsigned lw = *(const unsigned*)&ws->leftY[4 * sby];
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = (int)((lw >> (8 * k)) & 0xff);
          }
        } else if (i16mode == 2) {
          // lanes 0..3: top words, 4..7: left words
          const unsigned wt = *(const unsigned*)&up[UPY(4 * (lane & 3))];
          const unsigned wl = *(const unsigned*)&ws->leftY[4 * (lane & 3)];
          int s = (int)sum4(lane < 4 ? wt : wl);
          s += xor1(s);
          s += xor2(s);
          const int st = __builtin_amdgcn_readlane(s, 0), sl = __builtin_amdgcn_readlane(s, 4);
          int v;
          if (mbA && mbB) v = (st + sl + 16) >> 5;
          else if (mbA) v = (sl + 8) >> 4;
          else if (mbB) v = (st + 8) >> 4;
          else v = 128;
          pr[0] = pr[1] = pr[2] = pr[3] = v;
        } else if (mbA && mbB) {
          // plane (:366-424): lanes 0..7 horizontal terms, 8..15 vertical terms
          const int k = lane & 7;
          const int ha = up[UPY(8 + k)], hb = up[UPY(6 - k)];  // 6-k = -1: corner
          const int va = ws->leftY[8 + k], vb = k == 7 ? (int)up[UPY(-1)] : (int)ws->leftY[(6 - k) & 15];
          int term = lane < 8 ? (k + 1) * (ha - hb) : lane < 16 ? (k + 1) * (va - vb) : 0;
          term += xor1(term);
          term += xor2(term);
          term += dpp<ROW_SHR(4)>(0, term);
          const int hs = __builtin_amdgcn_readlane(term, 4), vs = __builtin_amdgcn_readlane(term, 12);
          const int a = 16 * ((int)ws->leftY[15] + (int)up[UPY(15)]);
          const int b = (5 * hs + 32) >> 6, c = (5 * vs + 32) >> 6;
          const int base = a + b * (x - 7) + c * (4 * sby - 7) + 16;
#pragma unroll
          for (int k2 = 0; k2 < 4; k2++) pr[k2] = clip255((base + c * k2) >> 5);
        }
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) w |= (unsigned)clip255(pr[k] + rl[k]) << (8 * k);
        yword = quad_transpose_bytes(w, selA, selB);  // row sq of block sb: x = 4*sbx .. +3, y = 4*sby+sq
#ifdef DRYV_SKIP_LUMA
      } else if (kind == 8) {
#else
      } else if (kind == 0 || (HAS_I8 && kind == 1)) {
#endif
        // mode grid: lanes 0..15 = by*4+bx (raster)
        const int mbx = lane & 3, mby = (lane >> 2) & 3;
        if (lateTR) {
          // fetch x = 16..23 of the row above (window dwords 5 and 6: the bottom line of macroblock mx+1) now
          if (upLds) {
            while (upDone < need2) {
              upDone = (unsigned)rfl((int)__hip_atomic_load(&bs->prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
              if (upDone < need2) __builtin_amdgcn_s_sleep(DRYV_POLL_SLEEP);
            }
            if (lane == 5 || lane == 6)
              ((unsigned*)ws->up)[lane] = *(const unsigned*)&bs->ringY[wave - 1][((mx + 1) & (RING_K - 1)) * 16 + (lane - 5) * 4];
          } else {
            while (upDone < need2) {
              unsigned v = 0;
              if (lane == 0) v = ld_sc1(upProg);
              upDone = (unsigned)rfl((int)v);
              if (upDone < need2) __builtin_amdgcn_s_sleep(DRYV_GPOLL_SLEEP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 5 || lane == 6) ((unsigned*)ws->up)[lane] = ld_sc1((const unsigned*)(planeY + (wOff - 16u)));
          }
          WAVE_SYNC();
        }
        // top border of the tile from the neighbour window; the left border is kept up to date
        if (lane < 25) ws->tileY[TY(lane - 1, -1)] = up[UPY(lane - 1)];
        const unsigned upM = mbB ? *(const unsigned*)&up[UPM] : 0x02020202u;
        const int Tb = (int)((upM >> (8 * mbx)) & 0xff);  // meaningful on lanes with mby == 0
        const int Lb = dpp<ROW_SHL(3)>(Mprev, Mprev);     // meaningful on lanes with mbx == 0: left MB's column 3
        if (!HAS_I8 || kind == 0) {
          if (mbA && mbC) Mcur = i4_macroblock<true>(ws, ldsBase + LT_T4E, rl, lane, Tb, Lb, prevFlags, remBits, true, true, true);
          else Mcur = i4_macroblock<false>(ws, ldsBase + LT_T4E, rl, lane, Tb, Lb, prevFlags, remBits, mbA, mbB, mbC);
        } else {
          // ---- Intra8x8: 8.3.2 (pred8x8.rs:152-764), four serial blocks, one pixel per lane -------
          WAVE_SYNC();
          const int px = lane & 7, py = lane >> 3;
          int M = 2;  // modes on the 4x4 grid (each 8x8 block fills its four positions)
          for (int b8 = 0; b8 < 4; b8++) {
            const int bx = b8 & 1, by = b8 >> 1;
            const bool topAv = by > 0 || mbB, leftAv = bx > 0 || mbA;
            const bool tlAv = topAv && leftAv;
            const bool trAv = b8 == 0 ? mbB : (b8 == 1 ? mbC : b8 == 2);
            // 8.3.2.1 (pred8x8.rs:698-764): A = grid position left of the block's first row, B = above it
            int predMode = 2;
            if (topAv && leftAv) {
              const int mAv = bx == 0 ? shfl(Lb, (2 * by) * 4) : shfl(M, (2 * by) * 4 + 2 * bx - 1);
              const int mBv = by == 0 ? shfl(Tb, 2 * bx) : shfl(M, (2 * by - 1) * 4 + 2 * bx);
              predMode = min(rfl(mAv), rfl(mBv));
            }
            const int rem = (int)((remBits >> (4 * b8)) & 7ull);
            const int mode = ((prevFlags >> b8) & 1u) ? predMode : (rem < predMode ? rem : rem + 1);
            if ((mbx >> 1) == bx && (mby >> 1) == by) M = mode;
            // raw edge E[0..24] = L7..L0, TL, T0..T15 (TR replaced by T7 when unavailable)
            const int ei = min(lane, trAv ? 24 : 16);
            const int ex = ei <= 8 ? 8 * bx - 1 : 8 * bx + ei - 9;
            const int ey = ei <= 7 ? 8 * by + 7 - ei : 8 * by - 1;
            const int E = ws->tileY[TY(ex, ey)];
            // reference sample filtering 8.3.2.2.1 (pred8x8.rs:222-288) incl. quirk Q1
            int Lf = shfl(E, max(lane - 1, 0)), Rt = shfl(E, min(lane + 1, 24));
            if (lane == 8) {
              if (!leftAv) Lf = E;
              if (!topAv) Rt = E;
            }
            if (lane == 9 && !tlAv) Lf = -1;  // Q1: p[-1,-1] = -1 enters the x = 0 filter tap
            if (lane == 7 && !tlAv) Rt = E;
            if (lane >= 24) Rt = E;
            const int E1 = (Lf + 2 * E + Rt + 2) >> 2;
            const int El = shfl(E1, max(lane - 1, 0)), Er = shfl(E1, min(lane + 1, 24));
            const int Er2 = lane >= 24 ? E1 : Er;
            const int F = (El + 2 * E1 + Er2 + 2) >> 2;
            const int Gv = (E1 + Er2 + 1) >> 1;
            const int packed = (E1 & 0xff) | ((F & 0xff) << 8) | ((Gv & 0xff) << 16);
            int s = E1 + shfl(E1, min(lane + 1, 63));
            s += shfl(s, min(lane + 2, 63));
            s += shfl(s, min(lane + 4, 63));
            const int sumL = __builtin_amdgcn_readlane(s, 0), sumT = __builtin_amdgcn_readlane(s, 9);
            const int te = t8[mode * 64 + lane];
            const int got = shfl(packed, te & 31);
            int pred = (got >> (8 * (te >> 5))) & 0xff;
            const int req = (int)((0x217771021ull >> (4 * mode)) & 7ull);
            const int have = (topAv ? 1 : 0) | (leftAv ? 2 : 0) | (tlAv ? 4 : 0);
            if ((req & ~have) != 0) pred = 0;
            if (mode == 2) {
              if (topAv && leftAv) pred = (sumT + sumL + 8) >> 4;
              else if (leftAv) pred = (sumL + 4) >> 3;
              else if (topAv) pred = (sumT + 4) >> 3;
              else pred = 128;
            }
            const int res = ws->resB[(8 * by + py) * 16 + 8 * bx + px];
            ws->tileY[TY(8 * bx + px, 8 * by + py)] = (uint8_t)clip255(pred + res);
            WAVE_SYNC();
          }
          Mcur = M;
        }
        yword = *(const unsigned*)&ws->tileY[TY(4 * sbx, 4 * sby + sq)];
      }

      PHASE_STAMP(7);  // luma prediction
      // ================= write-out ==================================================================
      // The bottom pixel line of the row (luma y = 15, chroma y = 7) and the bottom-row modes are what the row
      // below reads. Inside a band they go to this wave's LDS ring; on a band's last row they are stored
      // write-through (sc1) for the band below. Everything else is a plain write-back store.
      const unsigned m4 = (unsigned)__builtin_amdgcn_readlane(Mcur, 12) | ((unsigned)__builtin_amdgcn_readlane(Mcur, 13) << 8) |
                          ((unsigned)__builtin_amdgcn_readlane(Mcur, 14) << 16) |
                          ((unsigned)__builtin_amdgcn_readlane(Mcur, 15) << 24);
      const int slotK = mx & (RING_K - 1);
      if (toLds) {
        // ring entry mx replaces entry mx-RING_K, which the row below needs until it has copied the window of
        // macroblock mx-RING_K+1
#ifndef DRYV_NO_WAIT
        while (consKnown < mx - RING_K + 2) {  // (cached like upDone)
          consKnown = rfl((int)__hip_atomic_load(&bs->cons[wave + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (consKnown < mx - RING_K + 2) __builtin_amdgcn_s_sleep(DRYV_POLL_SLEEP);
        }
#endif
      }
      const bool lastMb = mx + 1 == W;
      if (lane < 32) {  // chroma: row strip (plane cpl, row 4*ccy+sq, x = 4*ccx..+3)
        const bool bottom = ccy == 1 && sq == 3;
        if (bottom && toGlobal) st_sc1((unsigned*)(planeY + cOff), cword);
        *(unsigned*)&ws->stageC[cpl][4 * ccy + sq][8 * (mx & 3) + 4 * ccx] = cword;
        if (bottom && toLds) *(unsigned*)&bs->ringC[wave][cpl][slotK * 8 + 4 * ccx] = cword;
        if (ccx == 1) ws->leftC[cpl][4 * ccy + sq] = (uint8_t)(cword >> 24);
      }
      cOff += 8;
      {
        const int y = 4 * sby + sq;
        if (y == 15 && toGlobal) st_sc1((unsigned*)(planeY + yOff), yword);
        *(unsigned*)&ws->stageY[y][16 * (mx & 1) + 4 * sbx] = yword;
        if (y == 15 && toLds) *(unsigned*)&bs->ringY[wave][slotK * 16 + 4 * sbx] = yword;
        if (sbx == 3) {
          ws->leftY[y] = (uint8_t)(yword >> 24);
          ws->tileY[TY(-1, y)] = (uint8_t)(yword >> 24);
        }
      }
      yOff += 16;
      // flush the staged rows: lane = (row, 8-byte segment); 32 contiguous bytes per pixel row and store
      nStores = 0;
#ifndef DRYV_SKIP_STORES  // tuning only
      if ((mx & 1) || lastMb) {
        WAVE_SYNC();
        const int row = lane >> 2, seg = lane & 3;
        const uint2 v = *(const uint2*)&ws->stageY[row][8 * seg];
        const unsigned off = (unsigned)(r * 16 + row) * pitchY + 16u * (unsigned)(mx & ~1) + 8u * seg;
        if ((mx & 1) || seg < 2) *(uint2*)(planeY + off) = v;
        nStores++;
      }
      if ((mx & 3) == 3 || lastMb) {
        WAVE_SYNC();
        const int pl = lane >> 5, row = (lane >> 2) & 7, seg = lane & 3;
        const uint2 v = *(const uint2*)&ws->stageC[pl][row][8 * seg];
        const unsigned off = (pl ? offCrS : offCbS) + (unsigned)(r * 8 + row) * pitchC + 8u * (unsigned)((mx & ~3) + seg);
        if (seg <= (mx & 3)) *(uint2*)(planeY + off) = v;
        nStores++;
      }
#endif
      // bottom-row modes (grid lanes 12..15) for the row below; the whole grid for the macroblock to the right
      if (toGlobal && lane == 0) st_sc1(myModes + mx, m4);
      if (toLds) {
        if (lane == 0) bs->ringM[wave][slotK] = m4;
        WAVE_SYNC();
        if (lane == 0) __hip_atomic_store(&bs->prog[wave], (unsigned)(mx + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      Mprev = Mcur;
      WAVE_SYNC();
      PHASE_STAMP(8);  // write-out
    }
    // the row is complete once its last stores have been written through
    if (toGlobal) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane0 == 0) st_sc1(myProg, (unsigned)W);
    }
    PHASE_STAMP(9);  // row tail
#ifdef DRYV_PHASE_PROFILE
    if (lane0 == 0 && task * BAND + wave < 49152u - 16u) {  // per-row timeline (tools/timeline.py), after the per-wave sums
      unsigned long long* tl = phaseOut + (size_t)(16384u + task * BAND + wave) * 10;
      tl[0] = rowT0;
      tl[1] = __builtin_amdgcn_s_memrealtime();
      tl[2] = (unsigned long long)blockIdx.x * 8 + wgWave;
    }
#endif
    }  // r < H
    // this wave is done with the band: its ring and counters may be reused once all four said so
    WAVE_SYNC();
    if (lane0 == 0) __hip_atomic_fetch_add(&bs->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#ifdef DRYV_PHASE_PROFILE
  if (lane0 == 0) {
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wgWave;
    for (int i = 0; i < 10; i++) phaseOut[gw * 10 + i] = phaseAcc[i];
  }
#endif
}

int recon_bands_per_block() { return WG_BANDS; }
int recon_blocks_per_cu() { return DRYV_WPS * 4 / (BAND * WG_BANDS); }

size_t recon_lds_bytes() { return LT_END + (size_t)WG_BANDS * BAND * sizeof(WaveScratch) + WG_BANDS * sizeof(BandShared); }

size_t recon_workspace_bytes(int W, int H, int n_frames) {
  // [task counter | pad to 256] [row progress: n_frames*H u32 | pad to 256] [bottom-row modes: n_mbs u32]
  const size_t prog = (((size_t)n_frames * H * 4) + 255) & ~(size_t)255;
  size_t bytes = 256 + prog + (size_t)n_frames * W * H * 4;
#ifdef DRYV_PHASE_PROFILE
  bytes = ((bytes + 255) & ~(size_t)255) + (size_t)65536 * 10 * 8;  // per-wave phase sums
#endif
  return bytes;
}

static size_t prog_bytes(const KParams& P) { return (((size_t)P.n_frames * P.H * 4) + 255) & ~(size_t)255; }

// the task counter and the row-progress words start every launch at zero (the modes need no reset:
// every word is written before it is read)
hipError_t recon_reset_workspace(const KParams& P, void* d_workspace, int grid, hipStream_t stream) {
  (void)grid;
  return hipMemsetAsync(d_workspace, 0, 256 + prog_bytes(P), stream);
}

hipError_t recon_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                        void* d_workspace, int grid, hipStream_t stream) {
  const int wavesPerBlock = BAND * WG_BANDS;
  const size_t ldsBytes = recon_lds_bytes();
  unsigned char* wsb = (unsigned char*)d_workspace;
  hipLaunchKernelGGL(recon_kernel, dim3(grid), dim3(wavesPerBlock * 64), ldsBytes, stream, P,
                     (const dryv_mb_desc*)d_mbs, (const int16_t*)d_coeffs, (uint8_t*)d_yuv, d_status,
                     (unsigned*)(wsb + 256), (unsigned*)(wsb + 256 + prog_bytes(P)), (unsigned*)wsb
#ifdef DRYV_PHASE_PROFILE
                     , (unsigned long long*)(wsb + (((256 + prog_bytes(P) + (size_t)P.n_frames * P.W * P.H * 4) + 255) & ~(size_t)255))
#endif
                     );
  return hipGetLastError();
}

}  // namespace dryv
