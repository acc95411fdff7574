// deblock_kernel.h — the in-loop deblocking filter (ITU-T H.264 8.7) for gfx950, as wave-level code (wave.h).
//
// Scope: frame macroblocks, 4:2:0, 8 bit, one slice per picture, all macroblocks intra: bS is 4 on
// macroblock edges and 3 inside. dryv has no deblocking (README.md:15), so there is no reference behaviour: the checker is
// oracle/dryv_deblock.c, a restatement of the clause.
//
// Decomposition: the reconstruction kernel's (band_kernel.h). The filter has the same 2:1 dependency wavefront as intra
// prediction -- macroblock (x, r) filters its left and top macroblock edges, so it needs (x-1, r), (x, r-1) and, because
// (x+1, r-1)'s left edge rewrites the three right-most columns of (x, r-1), also (x+1, r-1) -- hence:
//   * a band = 4 macroblock rows of one picture in lockstep, row g at macroblock x = s - 2g, 16 lanes per macroblock:
//     one lane per pixel row for vertical edges, per pixel column for horizontal edges (chroma: 8 + 8 lanes for Cb + Cr);
//   * one wave per band and plane kind: luma and chroma never meet in this filter, so half the waves of a launch filter
//     luma and the other half Cb + Cr, each kind with its own band queue (band-major order, any number resident), progress
//     words and side buffer -- a step's latency is then the longer of the two instead of their sum;
//   * macroblocks live in a two-macroblock-wide LDS tile per row with four rows of the macroblock above on top. A
//     macroblock is final towards the left once its right neighbour's left edge has been filtered, and its bottom four
//     rows only once the macroblock below has filtered its top edge: rows 0..11 are stored one step late by the row
//     itself, rows 12..15 travel through an LDS ring to the row below (patched with columns 12..15 one step late) and are
//     stored by it. Between bands they travel through a side buffer in the workspace (write-through stores, drained,
//     then a progress word: MI355X_MICROARCH.md "valid forms"; a side buffer and not the picture, because the band below
//     also WRITES the final values of those lines into the picture: a line it has written could be served stale from
//     its own L2 when a later macroblock's bytes of the same line are handed over).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"
#include "deblock_kernel_params.h"
#include "wave.h"

namespace dryv {
namespace deblock {

struct Args {
  const dryv_mb_desc* mbs;
  uint8_t* yuv;
  unsigned* status;        // bit 2 (value 4): a band gave up waiting (words 1..3: where)
  // per plane kind (0 luma, 1 chroma):
  unsigned* taskCounter[2];
  unsigned* prog[2];       // [frame][band]: macroblocks of the band's last row whose bottom rows are in the side buffer
  uint8_t* side[2];        // [frame][band][W][64]: 4 luma rows x 16   /   [frame][band][W][32]: Cb rows 6, 7 and Cr rows 6, 7 x 8
};

// per-workgroup tables
constexpr int T_ALPHA = 0, T_BETA = 64, T_TC0 = 128, T_QPC = 192, T_END = 320;
// per-wave scratch
constexpr int LSTR = 36;                      // luma tile row: 2 macroblocks x 16 + 4 (bank spread)
constexpr int S_TILE = 0;                     // u8 [4][20][LSTR]  row j = y + 4
constexpr int CSTR = 20;                      // chroma tile row: 2 x 8 + 4
constexpr int S_CTILE = S_TILE + 4 * 20 * LSTR;           // u8 [4][2][10][CSTR]  row j = y + 2
constexpr int S_RING = S_CTILE + 4 * 2 * 10 * CSTR;       // u8 [4][4 entries][4 rows][16]  bottom luma rows of row g's macroblocks
constexpr int S_RINGC = S_RING + 4 * 256;                 // u8 [4][4 entries][2][2 rows][8]
constexpr int S_BYTES = (S_RINGC + 4 * 128 + 63) & ~63;
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int SIDE_Y = 64, SIDE_C = 32;

WV void build_tables(const DParams& P, int ldsBase, int tid, int nthreads) {
  for (int k = tid; k < 52; k += nthreads) {
    wv::lds_st8(ldsBase + T_ALPHA + k, P.alpha[k]);
    wv::lds_st8(ldsBase + T_BETA + k, P.beta[k]);
    wv::lds_st8(ldsBase + T_TC0 + k, P.tc0[k]);
  }
  for (int k = tid; k < 104; k += nthreads) {  // 8.5.8 (as in band_kernel.h)
    const int qpi = min(max((k % 52) + (k < 52 ? P.cqo_cb : P.cqo_cr), 0), 51);
    const int d = qpi - 30;
    const int delta = d < 0 ? 0 : d < 16 ? (int)((0x7765544332221111ull >> (4 * d)) & 15ull) : (int)((0xCBA998u >> (4 * (d - 16))) & 15u);
    wv::lds_st8(ldsBase + T_QPC + k, (unsigned)(qpi - delta));
  }
}

struct Thr {
  int alpha, beta, tc0;
};
// 8.7.2.2: thresholds of an edge between samples of quantiser qp_p and qp_q
WV Thr thresholds(int ldsBase, int qp_p, int qp_q, int offA, int offB) {
  const int qpav = (qp_p + qp_q + 1) >> 1;
  const int ia = min(max(qpav + offA, 0), 51), ib = min(max(qpav + offB, 0), 51);
  Thr t;
  t.alpha = (int)wv::lds_u8(ldsBase + T_ALPHA + ia);
  t.beta = (int)wv::lds_u8(ldsBase + T_BETA + ib);
  t.tc0 = (int)wv::lds_u8(ldsBase + T_TC0 + ia);
  return t;
}
WV int iabs(int v) { return v < 0 ? -v : v; }

// One line of samples across an edge (8.7.2.3 / 8.7.2.4). STRONG: bS = 4, else bS = 3. `on`: this lane's edge is filtered at all.
template <bool STRONG, bool CHROMA>
WV void filter_line(int& p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int& q3, const Thr t, bool on) {
  on = on && iabs(p0 - q0) < t.alpha && iabs(p1 - p0) < t.beta && iabs(q1 - q0) < t.beta;  // filterSamplesFlag
  if (CHROMA) {
    int np0, nq0;
    if (STRONG) {
      np0 = (2 * p1 + p0 + q1 + 2) >> 2;
      nq0 = (2 * q1 + q0 + p1 + 2) >> 2;
    } else {
      const int tc = t.tc0 + 1;
      const int d = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
      np0 = min(max(p0 + d, 0), 255);
      nq0 = min(max(q0 - d, 0), 255);
    }
    p0 = on ? np0 : p0;
    q0 = on ? nq0 : q0;
    return;
  }
  const bool ap = iabs(p2 - p0) < t.beta, aq = iabs(q2 - q0) < t.beta;
  if (STRONG) {
    const bool small = iabs(p0 - q0) < ((t.alpha >> 2) + 2);
    const bool sp = on && ap && small, sq = on && aq && small;
    const int np0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : (2 * p1 + p0 + q1 + 2) >> 2;
    const int np1 = (p2 + p1 + p0 + q0 + 2) >> 2, np2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
    const int nq0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : (2 * q1 + q0 + p1 + 2) >> 2;
    const int nq1 = (p0 + q0 + q1 + q2 + 2) >> 2, nq2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
    p0 = on ? np0 : p0;
    p1 = sp ? np1 : p1;
    p2 = sp ? np2 : p2;
    q0 = on ? nq0 : q0;
    q1 = sq ? nq1 : q1;
    q2 = sq ? nq2 : q2;
  } else {
    const int tc = t.tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
    const int d = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
    const int avg = (p0 + q0 + 1) >> 1;
    const int np1 = p1 + min(max((p2 + avg - (p1 << 1)) >> 1, -t.tc0), t.tc0);
    const int nq1 = q1 + min(max((q2 + avg - (q1 << 1)) >> 1, -t.tc0), t.tc0);
    p1 = (on && ap) ? np1 : p1;
    q1 = (on && aq) ? nq1 : q1;
    p0 = on ? min(max(p0 + d, 0), 255) : p0;
    q0 = on ? min(max(q0 - d, 0), 255) : q0;
  }
}

WV void unpack4(unsigned w, int v[4]) {
  v[0] = (int)(w & 0xffu);
  v[1] = (int)((w >> 8) & 0xffu);
  v[2] = (int)((w >> 16) & 0xffu);
  v[3] = (int)(w >> 24);
}
WV unsigned pack4(const int v[4]) { return (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24); }

// One wave: claims bands of its plane kind until none are left. ldsBase: the workgroup's tables; ts: this wave's scratch.
template <bool LUMA>
WV void deblock_wave(const DParams& P, const Args& A, const int ldsBase, const int ts) {
  const int lane0 = wv::lane_id();
  const int W = P.W, H = P.H, nF = P.n_frames;
  const int nBands = (H + 3) >> 2;
  const unsigned totalTasks = (unsigned)nF * (unsigned)nBands;
  const int pitchY = 16 * W, pitchC = 8 * W;
  const size_t frameBytes = (size_t)W * H * 384;
  const unsigned offCb = (unsigned)W * H * 256u, offCr = offCb + (unsigned)W * H * 64u;

  for (;;) {
    // (every lane takes part in the claim: see band_kernel.h, "exec-mask hazard")
    const unsigned tsk = wv::atomic_add_task(A.taskCounter[LUMA ? 0 : 1], lane0 == 0 ? 1u : 0u);
    const unsigned task = (unsigned)wv::rfl((int)tsk);
    if (task >= totalTasks) break;
    const int b = (int)(task / (unsigned)nF), f = (int)(task - (unsigned)b * (unsigned)nF);
    const int r0 = 4 * b, nR = min(4, H - r0), gl = nR - 1;
    const int nSteps = W + 1 + 2 * (nR - 1);  // one virtual macroblock x = W per row: finalises macroblock W - 1
    const bool hasAbove = b > 0, hasBelow = r0 + nR < H;
    uint8_t* const plane = A.yuv + (size_t)f * frameBytes;
    const dryv_mb_desc* const mbsF = A.mbs + (size_t)f * W * H;
    constexpr int SIDE_ENTRY = LUMA ? SIDE_Y : SIDE_C;
    unsigned* const myProg = A.prog[LUMA ? 0 : 1] + (size_t)f * nBands + b;
    const unsigned* const upProg = myProg - 1;
    uint8_t* const mySide = A.side[LUMA ? 0 : 1] + ((size_t)f * nBands + b) * (size_t)W * SIDE_ENTRY;
    const uint8_t* const upSide = mySide - (size_t)W * SIDE_ENTRY;

    const int lane = lane0;
    const int g = lane >> 4, i = lane & 15;
    const int r = r0 + g;
    const bool rowOk = g < nR;
    const bool mbB = r > 0;
    const int cpl = i >> 3, crow = i & 7;  // chroma vertical-edge organisation: (plane, row); horizontal: (plane, column)
    const int tile = ts + S_TILE + 20 * LSTR * g;
    const int ctile = ts + S_CTILE + 2 * 10 * CSTR * g + 10 * CSTR * cpl;
    const int qpcT = ldsBase + T_QPC + 52 * cpl;

    // software pipeline: the macroblock's pixels and records are requested one step ahead
    u32x4 rowY = {0, 0, 0, 0};
    u32x2 rowC = {0, 0};
    unsigned dCur = 0, dTop = 0;
    auto prefetch = [&](int x) {
      if (rowOk && x >= 0 && x < W) {
        if (LUMA) rowY = wv::ld_u128_a2(plane + (size_t)(16 * r + i) * pitchY + 16 * x);
        else rowC = *(const u32x2*)(plane + (cpl ? offCr : offCb) + (size_t)(8 * r + crow) * pitchC + 8 * x);
        dCur = *(const unsigned*)(mbsF + (size_t)r * W + x);
        if (mbB) dTop = *(const unsigned*)(mbsF + (size_t)(r - 1) * W + x);
      }
    };
    prefetch(-2 * g);
    int qpLeft = 0;
    unsigned upKnown = 0, flagV = 0;
    bool flagPend = false, haveN = false;
    u32x4 topYn = {0, 0, 0, 0};  // the band above's rows for the NEXT step, requested a step early when it is far enough ahead
    u32x2 topCn = {0, 0};
    bool linePend = false;
    unsigned pubCount = 0;

    for (int s = 0; s < nSteps; s++) {
      const int x = s - 2 * g;
      const bool proc = rowOk && x >= 0 && x < W;        // this lane's row filters macroblock x
      const bool fin = rowOk && x >= 1 && x <= W;         // ... and finalises macroblock x - 1
      const int slot = x & 1, other = slot ^ 1;
      const bool mbA = x > 0;
      const int qp = (int)(dCur >> 24), qpT = (int)(dTop >> 24);
      const bool t8 = (dCur & 0xffu) == 1u;
      const u32x4 curY = rowY;
      const u32x2 curC = rowC;

      // ---- fetch what the band above hands to row 0: its progress word is read one step ahead, and so are the rows themselves
      // whenever it has got that far (the request then has a whole step to come back)
      u32x4 topY = {0, 0, 0, 0};
      u32x2 topC = {0, 0};
      auto fetch_side = [&](int mb, u32x4& y, u32x2& c) {
        const unsigned* e = (const unsigned*)(upSide + (size_t)mb * SIDE_ENTRY);
        if (LUMA && lane < 4) {
          y.x = wv::ld_sc1(e + 4 * lane);
          y.y = wv::ld_sc1(e + 4 * lane + 1);
          y.z = wv::ld_sc1(e + 4 * lane + 2);
          y.w = wv::ld_sc1(e + 4 * lane + 3);
        } else if (!LUMA && lane >= 4 && lane < 8) {
          c.x = wv::ld_sc1(e + 2 * (lane - 4));
          c.y = wv::ld_sc1(e + 2 * (lane - 4) + 1);
        }
      };
      bool haveNext = false;
      if (hasAbove && s < W) {  // row 0 is at macroblock s: the band above must have handed over macroblock s
        if (flagPend) upKnown = max(upKnown, (unsigned)wv::rfl((int)flagV));
        flagPend = false;
        unsigned spins = 0;
        while (upKnown < (unsigned)(s + 1)) {
          const unsigned v = wv::ld_sc1(upProg);
          upKnown = (unsigned)wv::rfl((int)v);
          if (upKnown < (unsigned)(s + 1)) {
            wv::sleep_short();
            if (++spins > SPIN_LIMIT) {
              if (lane == 0) {
                wv::atomic_or(A.status, 4u);
                A.status[1] = task;
                A.status[2] = ((unsigned)s << 16) | (unsigned)(s + 1);
                A.status[3] = upKnown;
              }
              upKnown = (unsigned)W;
            }
          }
        }
        wv::compiler_fence();
        if (haveN) {
          topY = topYn;
          topC = topCn;
        } else {
          fetch_side(s, topY, topC);
        }
        haveNext = s + 1 < W && upKnown >= (unsigned)(s + 2);
        if (haveNext) fetch_side(s + 1, topYn, topCn);
        if (upKnown < (unsigned)W) {
          flagV = wv::ld_sc1(upProg);
          flagPend = true;
        }
      }
      haveN = haveNext;
      // the next step's macroblock (its registers are free: curY / curC hold this step's)
      prefetch(x + 1);

      // ---- tile: the four (two) rows above this macroblock (its own rows go in after the vertical edges, straight from the
      // registers they were loaded into)
      if (proc) {
        if (mbB) {
          if (LUMA && i < 4) {
            u32x4 t = topY;
            if (g > 0) {
              const int rs = ts + S_RING + 256 * (g - 1) + 64 * (x & 3) + 16 * i;
              t = u32x4{wv::lds_u32(rs), wv::lds_u32(rs + 4), wv::lds_u32(rs + 8), wv::lds_u32(rs + 12)};
            }
            const int dt = tile + LSTR * i + 16 * slot;
            wv::lds_st32(dt, t.x);
            wv::lds_st32(dt + 4, t.y);
            wv::lds_st32(dt + 8, t.z);
            wv::lds_st32(dt + 12, t.w);
          } else if (!LUMA && i >= 4 && i < 8) {
            const int pl = (i - 4) >> 1, rw = (i - 4) & 1;
            u32x2 t = topC;
            if (g > 0) {
              const int rs = ts + S_RINGC + 128 * (g - 1) + 32 * (x & 3) + 16 * pl + 8 * rw;
              t = u32x2{wv::lds_u32(rs), wv::lds_u32(rs + 4)};
            }
            const int dt = ts + S_CTILE + 2 * 10 * CSTR * g + 10 * CSTR * pl + CSTR * rw + 8 * slot;
            wv::lds_st32(dt, t.x);
            wv::lds_st32(dt + 4, t.y);
          }
        }
      }

      // ---- vertical edges: lane = pixel row ----------------------------------------------------------------------------
      if (proc && LUMA) {
        const int rowB = tile + LSTR * (4 + i);
        int p[4], q[4][4];
        unpack4(wv::lds_u32(rowB + 16 * other + 12), p);
        unpack4(curY.x, q[0]);
        unpack4(curY.y, q[1]);
        unpack4(curY.z, q[2]);
        unpack4(curY.w, q[3]);
        const Thr tE = thresholds(ldsBase, qpLeft, qp, P.offA, P.offB), tI = thresholds(ldsBase, qp, qp, P.offA, P.offB);
        filter_line<true, false>(p[0], p[1], p[2], p[3], q[0][0], q[0][1], q[0][2], q[0][3], tE, mbA);
#pragma unroll
        for (int e = 1; e < 4; e++)
          filter_line<false, false>(q[e - 1][0], q[e - 1][1], q[e - 1][2], q[e - 1][3], q[e][0], q[e][1], q[e][2], q[e][3], tI,
                                    !(t8 && (e & 1)));
        if (mbA) wv::lds_st32(rowB + 16 * other + 12, pack4(p));
#pragma unroll
        for (int e = 0; e < 4; e++) wv::lds_st32(rowB + 16 * slot + 4 * e, pack4(q[e]));
      }
      if (proc && !LUMA) {
        // chroma: lane = (plane, row); edges at x = 0 (macroblock edge) and x = 4
        const int crowB = ctile + CSTR * (2 + crow);
        const int qc = (int)wv::lds_u8(qpcT + qp), qcL = (int)wv::lds_u8(qpcT + qpLeft);
        int cp[4], c0[4], c1[4];
        unpack4(wv::lds_u32(crowB + 8 * other + 4), cp);
        unpack4(curC.x, c0);
        unpack4(curC.y, c1);
        const Thr cE = thresholds(ldsBase, qcL, qc, P.offA, P.offB), cI = thresholds(ldsBase, qc, qc, P.offA, P.offB);
        filter_line<true, true>(cp[0], cp[1], cp[2], cp[3], c0[0], c0[1], c0[2], c0[3], cE, mbA);
        filter_line<false, true>(c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], cI, true);
        if (mbA) wv::lds_st32(crowB + 8 * other + 4, pack4(cp));
        wv::lds_st32(crowB + 8 * slot, pack4(c0));
        wv::lds_st32(crowB + 8 * slot + 4, pack4(c1));
      }
      wv::wave_sync();

      // ---- horizontal edges: lane = pixel column ------------------------------------------------------------------------
      if (proc && LUMA) {
        const int colB = tile + 16 * slot + i;
        int v[20];
#pragma unroll
        for (int j = 0; j < 20; j++) v[j] = (int)wv::lds_u8(colB + LSTR * j);
        const Thr tE = thresholds(ldsBase, qpT, qp, P.offA, P.offB), tI = thresholds(ldsBase, qp, qp, P.offA, P.offB);
        filter_line<true, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], tE, mbB);
#pragma unroll
        for (int e = 1; e < 4; e++)
          filter_line<false, false>(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3], v[4 * e + 4], v[4 * e + 5], v[4 * e + 6],
                                    v[4 * e + 7], tI, !(t8 && (e & 1)));
#pragma unroll
        for (int j = 1; j < 18; j++)
          if (j >= 4 || mbB) wv::lds_st8(colB + LSTR * j, (unsigned)v[j]);
      }
      if (proc && !LUMA) {
        // chroma: lane = (plane, column); rows -2..7
        const int ccolB = ctile + 8 * slot + crow;
        const int qc = (int)wv::lds_u8(qpcT + qp), qcT = (int)wv::lds_u8(qpcT + qpT);
        int c[10], dmy0 = 0, dmy1 = 0;
#pragma unroll
        for (int j = 0; j < 10; j++) c[j] = (int)wv::lds_u8(ccolB + CSTR * j);
        const Thr cE = thresholds(ldsBase, qcT, qc, P.offA, P.offB), cI = thresholds(ldsBase, qc, qc, P.offA, P.offB);
        filter_line<true, true>(dmy0, dmy1, c[0], c[1], c[2], c[3], dmy0, dmy1, cE, mbB);
        filter_line<false, true>(dmy0, dmy1, c[4], c[5], c[6], c[7], dmy0, dmy1, cI, true);
        if (mbB) wv::lds_st8(ccolB + CSTR * 1, (unsigned)c[1]);
        wv::lds_st8(ccolB + CSTR * 2, (unsigned)c[2]);
        wv::lds_st8(ccolB + CSTR * 5, (unsigned)c[5]);
        wv::lds_st8(ccolB + CSTR * 6, (unsigned)c[6]);
      }
      wv::wave_sync();

      // ---- after the filters: everything that leaves the tile is read first (one LDS round trip), then written / stored ----
      const bool lastRow = r == H - 1;
      const bool ringA = proc && g < gl, ringB = fin && x < W && g < gl;
      const bool topOut = proc && mbB;
      unsigned vA = 0, vB = 0;
      u32x4 vy = {0, 0, 0, 0}, vt = {0, 0, 0, 0};
      u32x2 vc = {0, 0}, vtc = {0, 0};
      if (LUMA) {
        if (ringA) vA = wv::lds_u32(tile + LSTR * (16 + (i >> 2)) + 16 * slot + 4 * (i & 3));   // (row 12 + (i >> 2), dword i & 3)
        if (ringB && i < 4) vB = wv::lds_u32(tile + LSTR * (16 + i) + 16 * other + 12);
        if (fin) {
          const int src = tile + LSTR * (4 + i) + 16 * other;
          vy = u32x4{wv::lds_u32(src), wv::lds_u32(src + 4), wv::lds_u32(src + 8), wv::lds_u32(src + 12)};
        }
        if (topOut && i < 4) {
          const int src = tile + LSTR * i + 16 * slot;
          vt = u32x4{wv::lds_u32(src), wv::lds_u32(src + 4), wv::lds_u32(src + 8), wv::lds_u32(src + 12)};
        }
      } else {
        const int cbase = ts + S_CTILE + 2 * 10 * CSTR * g;
        if (ringA && i < 8)  // (plane i >> 2, row 6 + ((i >> 1) & 1), dword i & 1)
          vA = wv::lds_u32(cbase + 10 * CSTR * (i >> 2) + CSTR * (8 + ((i >> 1) & 1)) + 8 * slot + 4 * (i & 1));
        if (ringB && i >= 4 && i < 8) vB = wv::lds_u32(cbase + 10 * CSTR * ((i - 4) >> 1) + CSTR * (8 + ((i - 4) & 1)) + 8 * other + 4);
        if (fin) {
          const int csrc = ctile + CSTR * (2 + crow) + 8 * other;
          vc = u32x2{wv::lds_u32(csrc), wv::lds_u32(csrc + 4)};
        }
        if (topOut && i >= 4 && i < 8) {
          const int src = cbase + 10 * CSTR * ((i - 4) >> 1) + CSTR * ((i - 4) & 1) + 8 * slot;
          vtc = u32x2{wv::lds_u32(src), wv::lds_u32(src + 4)};
        }
      }
      // bottom rows for the row below: this macroblock's as they are now, the left one's columns 12..15 (4..7) as patched
      if (LUMA) {
        if (ringA) wv::lds_st32(ts + S_RING + 256 * g + 64 * (x & 3) + 4 * i, vA);
        if (ringB && i < 4) wv::lds_st32(ts + S_RING + 256 * g + 64 * ((x - 1) & 3) + 16 * i + 12, vB);
      } else {
        if (ringA && i < 8) wv::lds_st32(ts + S_RINGC + 128 * g + 32 * (x & 3) + 4 * i, vA);
        if (ringB && i >= 4 && i < 8) wv::lds_st32(ts + S_RINGC + 128 * g + 32 * ((x - 1) & 3) + 16 * ((i - 4) >> 1) + 8 * ((i - 4) & 1) + 4, vB);
      }
      // The registers requested for the next step are "used" here, in front of this step's stores: the compiler then waits for
      // those loads now (they were issued a step's worth of cycles ago) instead of at the top of the next step, where its
      // vmcnt(0) would also wait for the stores below to be acknowledged (band_kernel.h has the same construction).
      if (LUMA) {
        rowY.x = (unsigned)wv::opaque((int)rowY.x); rowY.y = (unsigned)wv::opaque((int)rowY.y);
        rowY.z = (unsigned)wv::opaque((int)rowY.z); rowY.w = (unsigned)wv::opaque((int)rowY.w);
        topYn.x = (unsigned)wv::opaque((int)topYn.x); topYn.y = (unsigned)wv::opaque((int)topYn.y);
        topYn.z = (unsigned)wv::opaque((int)topYn.z); topYn.w = (unsigned)wv::opaque((int)topYn.w);
      } else {
        rowC.x = (unsigned)wv::opaque((int)rowC.x); rowC.y = (unsigned)wv::opaque((int)rowC.y);
        topCn.x = (unsigned)wv::opaque((int)topCn.x); topCn.y = (unsigned)wv::opaque((int)topCn.y);
      }
      dCur = (unsigned)wv::opaque((int)dCur);
      dTop = (unsigned)wv::opaque((int)dTop);
      flagV = (unsigned)wv::opaque((int)flagV);
      // ---- publish what the PREVIOUS step handed to the band below: its write-through stores have had this whole step to drain
      if (linePend) {
        wv::wait_vm(0);
        if (lane == 0) wv::st_sc1(myProg, pubCount);
        linePend = false;
      }
      // ---- stores: macroblock x - 1 is final except for its bottom rows; the macroblock above is final ----------------
      if (fin) {
        const int xl = x - 1;
        unsigned* e = (unsigned*)(mySide + (size_t)xl * SIDE_ENTRY);
        const bool toBelow = hasBelow && g == gl;  // the band's last row: its bottom rows go to the band below through the side buffer
        if (LUMA) {
          if (i < 12 || lastRow) wv::st_g128(plane + (size_t)(16 * r + i) * pitchY + 16 * xl, vy);
          if (toBelow && i >= 12) {
            wv::st_sc1(e + 4 * (i - 12), vy.x);
            wv::st_sc1(e + 4 * (i - 12) + 1, vy.y);
            wv::st_sc1(e + 4 * (i - 12) + 2, vy.z);
            wv::st_sc1(e + 4 * (i - 12) + 3, vy.w);
          }
        } else {
          if (crow < 6 || lastRow) wv::st_g64(plane + (cpl ? offCr : offCb) + (size_t)(8 * r + crow) * pitchC + 8 * xl, vc);
          if (toBelow && crow >= 6) {
            wv::st_sc1(e + 4 * cpl + 2 * (crow - 6), vc.x);
            wv::st_sc1(e + 4 * cpl + 2 * (crow - 6) + 1, vc.y);
          }
        }
      }
      if (hasBelow && wv::any(fin && g == gl)) {
        linePend = true;
        pubCount = (unsigned)(s - 2 * gl);  // macroblocks 0 .. x - 1 of the last row are in the side buffer
      }
      if (topOut) {  // the macroblock above: its bottom four (two) rows are final now
        if (LUMA && i < 4) {
          wv::st_g128(plane + (size_t)(16 * (r - 1) + 12 + i) * pitchY + 16 * x, vt);
        } else if (!LUMA && i >= 4 && i < 8) {
          const int pl = (i - 4) >> 1, rw = (i - 4) & 1;
          wv::st_g64(plane + (pl ? offCr : offCb) + (size_t)(8 * (r - 1) + 6 + rw) * pitchC + 8 * x, vtc);
        }
      }
      if (proc) qpLeft = qp;
      wv::wave_sync();
    }
    if (hasBelow) {
      wv::wait_vm(0);
      if (lane0 == 0) wv::st_sc1(myProg, (unsigned)W);
    }
  }
}

}  // namespace deblock
}  // namespace dryv
