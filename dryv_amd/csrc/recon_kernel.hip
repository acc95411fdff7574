// recon_kernel.hip — gfx950 (CDNA4 / MI355X) macroblock reconstruction for dryv's AVC intra path.
//
// One workgroup reconstructs whole frames; each 64-lane wavefront owns one macroblock ROW at a
// time and walks it left to right. Rows of a frame advance as a 2:1 diagonal: row r may process
// macroblock x once row r-1 has finished macroblock x+1 (the top-right neighbour C, reference
// slice/macroblock.rs:455-456, slice/mod.rs:593-598). The bottom pixel line, and the bottom-row
// prediction modes, of every row are staged in LDS ring slots for the row below; progress is
// published through LDS counters (workgroup-scope release/acquire). Coefficients stream in from
// HBM (768 B/MB, coalesced dword loads, next MB prefetched), the Y/Cb/Cr planes stream out
// (384 B/MB). No neighbour sample is ever re-read from HBM.
//
// What is computed (reference file:line in each function): inverse zig-zag, dequantisation,
// 4x4 / 8x8 integer inverse transforms, Intra16x16 DC Hadamard, chroma DC 2x2, Intra4x4 / Intra8x8 /
// Intra16x16 / chroma prediction incl. prediction-mode derivation, clip-add and picture
// construction — bit-exact with the reference's Rust CPU path, quirks Q1-Q5 (SURVEY.md §8a') included.
//
// Arithmetic is int32 (the reference uses 64-bit isize): exact whenever every intermediate fits
// 32 bits, which holds for any conformant 8-bit stream (the standard bounds them to 16 bits) and
// for |level| <= 2047 with flat scaling lists at every QP.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "recon_kernel.h"

namespace dryv {

#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ unsigned sum4(unsigned w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }
// all four bytes of w non-zero
__device__ __forceinline__ bool bytes_nonzero(unsigned w) {
  return (((w - 0x01010101u) & ~w) & 0x80808080u) == 0u;
}

// ---- per-wave LDS scratch ---------------------------------------------------------------------
struct WaveScratch {
  union {
    int16_t coef[384];  // the macroblock's coefficient lists as they arrive (zig-zag order)
    int32_t g8[256];    // 8x8 transform: row-pass output of the four blocks
  };
  int16_t resY[256];      // luma residual r, pixel order [y][x]
  int16_t resC[2][64];    // chroma residual, [plane][y][x]
  int32_t dcY[16];        // Intra16x16 luma DC after Hadamard + scaling, [blkIdx]
  int32_t dcC[2][4];      // chroma DC after 2x2 + scaling, [plane][blkIdx]
  uint8_t tileY[17 * 32]; // luma tile with borders: row 0 = y -1, byte 3 = x -1, bytes 4..27 = x 0..23
  uint8_t tileC[2][9 * 16]; // chroma tiles: row 0 = y -1, byte 3 = x -1, bytes 4..11 = x 0..7
  uint8_t leftY[16];      // left neighbour's column x = 15 (contiguous copy)
  uint8_t leftC[2][8];
  uint8_t mgrid[5][8];    // derived luma modes on the 4x4-block grid, row/col 0 = neighbours
  uint8_t pad[24];
};
static_assert(sizeof(WaveScratch) % 16 == 0, "scratch must keep 16-byte alignment");

__host__ __device__ constexpr int TY(int x, int y) { return (y + 1) * 32 + (x + 4); }
__host__ __device__ constexpr int TC(int x, int y) { return (y + 1) * 16 + (x + 4); }

// raster position (i*4+j) -> index in the zig-zag list (frame/mod.rs:185-209)
__device__ constexpr int ZZ4I[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};

// ---- 4x4 residual: one block per lane (transform.rs:116-191, 8.5.12) -------------------------
// list[k] = cs[ptr + k]; when dc_given the DC comes in already scaled (Intra16x16 / chroma) and is
// not dequantised again (transform.rs:145-146).
__device__ __forceinline__ void residual4x4_lane(const int16_t* cs, int ptr, bool dc_given, int dcval,
                                                 int qp, const uint16_t* ls4, int r[16]) {
  const int qd = (qp * 43) >> 8;  // qp / 6 for 0..51
  const int qm = qp - 6 * qd;
  int d[16];
#pragma unroll
  for (int n = 0; n < 16; n++) {
    const int c = cs[ptr + ZZ4I[n]];
    const int ls = ls4[qm * 16 + n];
    const int prod = c * ls;
    const int hi = prod << max(qd - 4, 0);
    const int lo = (prod + (1 << max(3 - qd, 0))) >> max(4 - qd, 0);
    d[n] = qp >= 24 ? hi : lo;
  }
  if (dc_given) d[0] = dcval;
  int f[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int e0 = d[i * 4 + 0] + d[i * 4 + 2];
    const int e1 = d[i * 4 + 0] - d[i * 4 + 2];
    const int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
    const int e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
    f[i * 4 + 0] = e0 + e3;
    f[i * 4 + 1] = e1 + e2;
    f[i * 4 + 2] = e1 - e2;
    f[i * 4 + 3] = e0 - e3;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int g0 = f[0 + j] + f[8 + j];
    const int g1 = f[0 + j] - f[8 + j];
    const int g2 = (f[4 + j] >> 1) - f[12 + j];
    const int g3 = f[4 + j] + (f[12 + j] >> 1);
    r[0 + j] = (g0 + g3 + 32) >> 6;
    r[4 + j] = (g1 + g2 + 32) >> 6;
    r[8 + j] = (g1 - g2 + 32) >> 6;
    r[12 + j] = (g0 - g3 + 32) >> 6;
  }
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

// table 8-15 (transform.rs:194-216); qpy and offset are wave-uniform
__device__ __forceinline__ int qpc_of(int qpy, int offset) {
  const int qpi = min(max(qpy + offset, 0), 51);
  if (qpi < 30) return qpi;
  // qpi - QPCS[qpi - 30] for qpi = 30..51, 4 bits each
  const unsigned long long dlo = 0x7765544332221111ull;  // qpi 30..45
  const unsigned dhi = 0xCBA998u;                        // qpi 46..51
  const int k = qpi - 30;
  const int delta = k < 16 ? (int)((dlo >> (4 * k)) & 15ull) : (int)((dhi >> (4 * (k - 16))) & 15u);
  return qpi - delta;
}

__global__ void __launch_bounds__(1024)
recon_kernel(const KParams P, const dryv_mb_desc* __restrict__ mbs, const int16_t* __restrict__ coeffs,
             uint8_t* __restrict__ yuv, unsigned* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int NW = blockDim.x >> 6;
  const int W = P.W, H = P.H;

  // ---- LDS carve ------------------------------------------------------------------------------
  uint16_t* ls4 = (uint16_t*)(lds + 0);            // [6][16]
  uint16_t* ls8 = (uint16_t*)(lds + 192);          // [6][64]
  uint8_t* t4 = lds + 960;                         // [9][16]
  uint8_t* t8 = lds + 1104;                        // [9][64]
  uint8_t* zz8i = lds + 1680;                      // [64] raster -> list index
  unsigned* prog = (unsigned*)(lds + 1744);        // [NW] (<= 16 entries)
  const int lineY_stride = W * 16 + 48;            // 16 bytes of slack in front, 32 behind
  const int lineC_stride = W * 8 + 32;             // 16 in front, 16 behind
  uint8_t* lineY = lds + 1808;
  uint8_t* lineC = lineY + NW * lineY_stride;      // [NW][2][lineC_stride]
  unsigned* lineM = (unsigned*)(lineC + NW * 2 * lineC_stride);  // [NW][W]
  WaveScratch* ws = (WaveScratch*)((unsigned char*)(lineM + NW * W)) + wave;

  for (int i = threadIdx.x; i < 96; i += blockDim.x) ls4[i] = P.ls4[i];
  for (int i = threadIdx.x; i < 384; i += blockDim.x) ls8[i] = P.ls8[i];
  for (int i = threadIdx.x; i < 144; i += blockDim.x) t4[i] = P.t4[i];
  for (int i = threadIdx.x; i < 576; i += blockDim.x) t8[i] = P.t8[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) zz8i[i] = P.zz8i[i];
  if (threadIdx.x < 16) prog[threadIdx.x] = 0;
  __syncthreads();

  const size_t frameBytes = (size_t)W * H * 384;
  const int pitchY = W * 16, pitchC = W * 8;
  const int G = gridDim.x;
  const int nfLocal = (P.n_frames - (int)blockIdx.x + G - 1) / G;
  const int totalRows = nfLocal * H;

  for (int Rg = wave; Rg < totalRows; Rg += NW) {
    const int it = Rg / H;
    const int r = Rg - it * H;
    const int f = blockIdx.x + it * G;
    const int slot = Rg % NW;
    const int slotUp = (Rg + NW - 1) % NW;
    const size_t mbBase = (size_t)f * W * H + (size_t)r * W;
    uint8_t* planeY = yuv + (size_t)f * frameBytes;
    uint8_t* planeCb = planeY + (size_t)W * H * 256;
    uint8_t* planeCr = planeCb + (size_t)W * H * 64;
    uint8_t* myLineY = lineY + slot * lineY_stride + 16;
    const uint8_t* upLineY = lineY + slotUp * lineY_stride + 16;
    uint8_t* myLineC = lineC + slot * 2 * lineC_stride + 16;
    const uint8_t* upLineC = lineC + slotUp * 2 * lineC_stride + 16;
    unsigned* myLineM = lineM + slot * W;
    const unsigned* upLineM = lineM + slotUp * W;
    const bool rowTop = r > 0;  // macroblock B exists

    if (lane == 0)
      __hip_atomic_store(&prog[slot], ((unsigned)(Rg + 1) << 12), __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_WORKGROUP);

    // prefetch macroblock 0 of the row
    const uint32_t* cptr = (const uint32_t*)(coeffs + mbBase * 384);
    uint32_t pc0 = cptr[lane], pc1 = cptr[lane + 64], pc2 = cptr[lane + 128];
    uint4 pdesc = *(const uint4*)(mbs + mbBase);

    unsigned leftM = 0;  // kind and right-column modes of the macroblock to the left

    for (int mx = 0; mx < W; mx++) {
      // ---- 1. stage this macroblock's inputs, prefetch the next ----------------------------
      uint32_t* cs32 = (uint32_t*)ws->coef;
      cs32[lane] = pc0;
      cs32[lane + 64] = pc1;
      cs32[lane + 128] = pc2;
      const unsigned d0 = rfl(pdesc.x), d1 = rfl(pdesc.y), d2 = rfl(pdesc.z), d3 = rfl(pdesc.w);
      if (mx + 1 < W) {
        const uint32_t* np = cptr + (size_t)(mx + 1) * 192;
        pc0 = np[lane];
        pc1 = np[lane + 64];
        pc2 = np[lane + 128];
        pdesc = *(const uint4*)(mbs + mbBase + mx + 1);
      }
      int kind = d0 & 0xff;
      const int i16mode = (d0 >> 8) & 0xff;
      const int cmode = (d0 >> 16) & 0xff;
      int qp = (d0 >> 24) & 0xff;
      const unsigned prevFlags = d1 & 0xffff;
      // rem nibbles: 16 nibbles = bytes 6..13 of the record
      const unsigned long long remBits =
          ((unsigned long long)(d1 >> 16)) | ((unsigned long long)d2 << 16) |
          ((unsigned long long)(d3 & 0xffff) << 48);
      const bool bad = kind > 2 || qp > 51 || i16mode > 3 || cmode > 3;
      if (bad) {
        if (lane == 0) atomicOr(status, 1u);
        kind = 3;
        qp = 0;
      }

      // ---- 2. wait for the row above ---------------------------------------------------------
      if (Rg > 0) {
        const unsigned need = ((unsigned)Rg << 12) | (unsigned)min(mx + 2, W);
        while (__hip_atomic_load(&prog[slotUp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
          __builtin_amdgcn_s_sleep(1);
      }

      const bool mbA = mx > 0, mbB = rowTop, mbC = rowTop && (mx + 1 < W);
      const unsigned upM = mbB ? upLineM[mx] : 0u;

      // ---- 3. borders of the tiles: top line from the ring slot of the row above -------------
      if (lane < 25) ws->tileY[TY(lane - 1, -1)] = upLineY[mx * 16 - 1 + lane];
      if (lane >= 32 && lane < 50) {
        const int pl = (lane - 32) / 9, k = (lane - 32) % 9;
        ws->tileC[pl][TC(k - 1, -1)] = upLineC[pl * lineC_stride + mx * 8 - 1 + k];
      }
      WAVE_SYNC();

      if (kind == 3) {
        // unsupported record: zero macroblock, neighbours see an Intra16x16 macroblock of zeros
        if (lane < 64) *(uint32_t*)&ws->tileY[TY((lane & 3) * 4, lane >> 2)] = 0;
        if (lane < 32) *(uint32_t*)&ws->tileC[lane >> 4][TC((lane & 1) * 4, (lane & 15) >> 1)] = 0;
      } else {
        // ---- 4. residuals ---------------------------------------------------------------------
        const int qpc0 = qpc_of(qp, P.cqo_cb), qpc1 = qpc_of(qp, P.cqo_cr);
        // chroma DC: 8.5.11 (trans_chroma.rs:369-415), lanes 0..7 = plane*4 + position
        if (lane < 8) {
          const int pl = lane >> 2, pos = lane & 3;
          const int16_t* c = ws->coef + 256 + pl * 64;
          const int c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
          // f = A c A, A = [[1,1],[1,-1]]; c = [[c0,c1],[c2,c3]]
          const int fv = pos == 0 ? c0 + c1 + c2 + c3
                       : pos == 1 ? c0 - c1 + c2 - c3
                       : pos == 2 ? c0 + c1 - c2 - c3
                                  : c0 - c1 - c2 + c3;
          const int q = pl ? qpc1 : qpc0;
          const int qd = (q * 43) >> 8, qm = q - 6 * qd;
          ws->dcC[pl][pos] = ((fv * (int)ls4[qm * 16]) << qd) >> 5;
        }
        // Intra16x16 luma DC: 8.5.10 (pred16x16.rs:428-482), lanes 16..31 = (i, j)
        if (kind == 2 && lane >= 16 && lane < 32) {
          const int i = (lane >> 2) & 3, j = lane & 3;
          // c[k][l] = inverse zig-zag of the DC list; f[i][j] = sum_k sum_l A[i][k] c[k][l] A[l][j]
          int fv = 0;
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int l = 0; l < 4; l++) {
              const int c = ws->coef[ZZ4I[k * 4 + l]];
              // A = [[1,1,1,1],[1,1,-1,-1],[1,-1,-1,1],[1,-1,1,-1]]: sign bits per row
              // sign nibbles of rows 0..3: bit k set -> A[row][k] = -1 (A is symmetric)
              const int sa = (0xA6C0u >> (4 * i + k)) & 1, sb = (0xA6C0u >> (4 * j + l)) & 1;
              fv += (sa ^ sb) ? -c : c;
            }
          const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
          const int prod = fv * (int)ls4[qm * 16];
          const int v = qp >= 36 ? (prod << max(qd - 6, 0)) : ((prod + (1 << max(5 - qd, 0))) >> max(6 - qd, 0));
          // dc_y_to_luma (pred16x16.rs:27-31): blkIdx of the 4x4 block at (x = j, y = i)
          const int blk = 8 * (i >> 1) + 4 * (j >> 1) + 2 * (i & 1) + (j & 1);
          ws->dcY[blk] = v;
        }
        WAVE_SYNC();

        // 4x4 blocks: lanes 0..15 luma (kinds 0 and 2), lanes 16..23 chroma
        {
          const bool lumaLane = lane < 16 && kind != 1;
          const bool chromaLane = lane >= 16 && lane < 24;
          if (lumaLane || chromaLane) {
            int ptr, dcv = 0, q;
            bool dcg;
            if (lumaLane) {
              q = qp;
              if (kind == 0) {
                ptr = lane * 16;
                dcg = false;
              } else {
                ptr = 16 + lane * 15 - 1;
                dcg = true;
                dcv = ws->dcY[lane];
              }
            } else {
              const int pl = (lane - 16) >> 2, cb = (lane - 16) & 3;
              q = pl ? qpc1 : qpc0;
              ptr = 256 + pl * 64 + 4 + cb * 15 - 1;
              dcg = true;
              dcv = ws->dcC[pl][cb];
            }
            int rr[16];
            residual4x4_lane(ws->coef, ptr, dcg, dcv, q, ls4, rr);
            if (lumaLane) {
              const int bx = ((lane >> 1) & 2) | (lane & 1), by = ((lane >> 2) & 2) | ((lane >> 1) & 1);
#pragma unroll
              for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) ws->resY[(4 * by + i) * 16 + 4 * bx + j] = (int16_t)rr[i * 4 + j];
            } else {
              const int pl = (lane - 16) >> 2, cb = (lane - 16) & 3;
              const int cx = cb & 1, cy = cb >> 1;
#pragma unroll
              for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) ws->resC[pl][(4 * cy + i) * 8 + 4 * cx + j] = (int16_t)rr[i * 4 + j];
            }
          }
        }
        WAVE_SYNC();

        // 8x8 blocks: 8.5.13 (pred8x8.rs:51-150). lanes 0..31 = (blk8, row) then (blk8, column).
        if (kind == 1) {
          const int b8 = (lane >> 3) & 3, i = lane & 7;
          int dd[8], oo[8];
          if (lane < 32) {
            const int qd = (qp * 43) >> 8, qm = qp - 6 * qd;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int c = ws->coef[b8 * 64 + zz8i[i * 8 + j]];
              const int prod = c * (int)ls8[qm * 64 + i * 8 + j];
              dd[j] = qp >= 36 ? (prod << max(qd - 6, 0)) : ((prod + (1 << max(5 - qd, 0))) >> max(6 - qd, 0));
            }
            idct8(dd, oo);
          }
          WAVE_SYNC();  // every coefficient has been read before g8 (aliasing coef) is written
          if (lane < 32) {
#pragma unroll
            for (int j = 0; j < 8; j++) ws->g8[b8 * 64 + i * 8 + j] = oo[j];
          }
          WAVE_SYNC();
          if (lane < 32) {
            const int j = i;  // this lane now owns column j
#pragma unroll
            for (int k = 0; k < 8; k++) dd[k] = ws->g8[b8 * 64 + k * 8 + j];
            idct8(dd, oo);
            const int bx = b8 & 1, by = b8 >> 1;
#pragma unroll
            for (int k = 0; k < 8; k++) ws->resY[(8 * by + k) * 16 + 8 * bx + j] = (int16_t)((oo[k] + 32) >> 6);
          }
          WAVE_SYNC();
        }

        // ---- 5. luma prediction + reconstruction -------------------------------------------
        if (kind == 2) {
          // Intra16x16: 8.3.3 (pred16x16.rs:79-425). lane -> row y, 4 pixels at x0
          const int y = lane >> 2, x0 = (lane & 3) * 4;
          const unsigned tw = *(const unsigned*)&ws->tileY[TY(x0, -1)];
          int pr[4];
          if (i16mode == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = mbB ? (int)((tw >> (8 * k)) & 0xff) : 0;
          } else if (i16mode == 1) {
            const int lv = mbA ? (int)ws->leftY[y] : 0;
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = lv;
          } else if (i16mode == 2) {
            unsigned st = 0, sl = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              st += sum4(*(const unsigned*)&ws->tileY[TY(4 * k, -1)]);
              sl += sum4(*(const unsigned*)&ws->leftY[4 * k]);
            }
            int v;
            if (mbA && mbB) v = (int)(st + sl + 16) >> 5;
            else if (mbA) v = (int)(sl + 8) >> 4;
            else if (mbB) v = (int)(st + 8) >> 4;
            else v = 128;
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = v;
          } else {
            if (mbA && mbB) {
              // lanes 0..7: horizontal terms, lanes 8..15: vertical terms
              const int k = lane & 7;
              int term = 0;
              if (lane < 8) {
                const int a = ws->tileY[TY(8 + k, -1)], b = ws->tileY[TY(6 - k, -1)];  // 6-k = -1 -> corner
                term = (k + 1) * (a - b);
              } else if (lane < 16) {
                const int a = ws->leftY[8 + k];
                const int b = k == 7 ? (int)ws->tileY[TY(-1, -1)] : (int)ws->leftY[6 - k];
                term = (k + 1) * (a - b);
              }
              term += shfl(term, lane ^ 1);
              term += shfl(term, lane ^ 2);
              term += shfl(term, lane ^ 4);
              const int hs = __builtin_amdgcn_readlane(term, 0), vs = __builtin_amdgcn_readlane(term, 8);
              const int a = 16 * ((int)ws->leftY[15] + (int)ws->tileY[TY(15, -1)]);
              const int b = (5 * hs + 32) >> 6, c = (5 * vs + 32) >> 6;
#pragma unroll
              for (int k2 = 0; k2 < 4; k2++) pr[k2] = clip255((a + b * (x0 + k2 - 7) + c * (y - 7) + 16) >> 5);
            } else {
#pragma unroll
              for (int k2 = 0; k2 < 4; k2++) pr[k2] = 0;
            }
          }
          unsigned outw = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) outw |= (unsigned)clip255(pr[k] + (int)ws->resY[y * 16 + x0 + k]) << (8 * k);
          *(unsigned*)&ws->tileY[TY(x0, y)] = outw;
        } else {
          // neighbour modes on the 4x4 grid borders (pred4x4.rs:386-412, pred8x8.rs:723-751)
          if (lane < 4) {
            ws->mgrid[0][1 + lane] = (uint8_t)((upM >> (8 + 4 * lane)) & 0xf);
            ws->mgrid[1 + lane][0] = (uint8_t)((leftM >> (8 + 4 * lane)) & 0xf);
          }
          WAVE_SYNC();
          if (kind == 0) {
            // Intra4x4: 8.3.1 (pred4x4.rs:10-427) as a 10-step 2:1 block wavefront, 16 lanes/block
            const int grp = lane >> 4, li = lane & 15;
            const int px = li & 3, py = li >> 2;
            for (int t = 0; t < 10; t++) {
              const int byLo = max(0, (t - 2) >> 1), byHi = min(3, t >> 1);
              const int by = byLo + grp, bx = t - 2 * by;
              const bool act = grp < 2 && by <= byHi;
              if (act) {
                const int blk = 8 * (by >> 1) + 4 * (bx >> 1) + 2 * (by & 1) + (bx & 1);
                const bool topAv = by > 0 || mbB, leftAv = bx > 0 || mbA;
                const bool tlAv = (bx > 0 || mbA) && (by > 0 || mbB);
                const bool trAv = by > 0 ? ((0x5744u >> blk) & 1u) != 0 : (bx < 3 ? mbB : mbC);
                // mode derivation 8.3.1.1
                int predMode = 2;
                if (topAv && leftAv) predMode = min((int)ws->mgrid[by + 1][bx], (int)ws->mgrid[by][bx + 1]);
                const int rem = (int)((remBits >> (4 * blk)) & 7ull);
                const int mode = ((prevFlags >> blk) & 1u) ? predMode : (rem < predMode ? rem : rem + 1);
                // edge E[0..12] = L3..L0, TL, T0..T7 (TR replaced by T3 when unavailable)
                int ei = min(li, trAv ? 12 : 8);
                const int ex = ei <= 4 ? 4 * bx - 1 : 4 * bx + ei - 5;
                const int ey = ei <= 3 ? 4 * by + 3 - ei : 4 * by - 1;
                const int E = ws->tileY[TY(ex, ey)];
                const int gb = lane & ~15;
                const int El = shfl(E, gb + max(li - 1, 0));
                const int Er = shfl(E, gb + min(li + 1, 15));
                const int F = (El + 2 * E + Er + 2) >> 2;
                const int Gv = (E + Er + 1) >> 1;
                const int packed = E | (F << 8) | (Gv << 16);
                const int s1 = E + Er;
                const int s2 = s1 + shfl(s1, gb + min(li + 2, 15));
                const int sumL = shfl(s2, gb + 0), sumT = shfl(s2, gb + 5);
                const int te = t4[mode * 16 + li];
                const int got = shfl(packed, gb + (te & 31));
                int pred = (got >> (8 * (te >> 5))) & 0xff;
                const int req = (int)((0x217771021ull >> (4 * mode)) & 7ull);  // per mode: bit0 top, bit1 left, bit2 corner
                const int have = (topAv ? 1 : 0) | (leftAv ? 2 : 0) | (tlAv ? 4 : 0);
                if ((req & ~have) != 0) pred = 0;  // reference leaves the zero-initialised samples (Q4)
                if (mode == 2) {
                  if (topAv && leftAv) pred = (sumT + sumL + 4) >> 3;
                  else if (leftAv) pred = (sumL + 2) >> 2;
                  else if (topAv) pred = (sumT + 2) >> 2;
                  else pred = 128;
                }
                const int res = ws->resY[(4 * by + py) * 16 + 4 * bx + px];
                ws->tileY[TY(4 * bx + px, 4 * by + py)] = (uint8_t)clip255(pred + res);
                if (li == 0) ws->mgrid[by + 1][bx + 1] = (uint8_t)mode;
              }
              WAVE_SYNC();
            }
          } else {
            // Intra8x8: 8.3.2 (pred8x8.rs:152-764), four serial blocks, one pixel per lane
            const int px = lane & 7, py = lane >> 3;
            for (int b8 = 0; b8 < 4; b8++) {
              const int bx = b8 & 1, by = b8 >> 1;
              const bool topAv = by > 0 || mbB, leftAv = bx > 0 || mbA;
              const bool tlAv = (bx > 0 || mbA) && (by > 0 || mbB);
              const bool trAv = b8 == 0 ? mbB : (b8 == 1 ? mbC : b8 == 2);
              int predMode = 2;
              if (topAv && leftAv)
                predMode = min((int)ws->mgrid[2 * by + 1][2 * bx], (int)ws->mgrid[2 * by][2 * bx + 1]);
              const int rem = (int)((remBits >> (4 * b8)) & 7ull);
              const int mode = ((prevFlags >> b8) & 1u) ? predMode : (rem < predMode ? rem : rem + 1);
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
              if (lane == 9 && !tlAv) Lf = -1;   // Q1: p[-1,-1] = -1 enters the x = 0 filter tap
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
              const int res = ws->resY[(8 * by + py) * 16 + 8 * bx + px];
              ws->tileY[TY(8 * bx + px, 8 * by + py)] = (uint8_t)clip255(pred + res);
              if (lane < 4) ws->mgrid[2 * by + 1 + (lane >> 1)][2 * bx + 1 + (lane & 1)] = (uint8_t)mode;
              WAVE_SYNC();
            }
          }
        }

        // ---- 6. chroma prediction + reconstruction: 8.3.4 (trans_chroma.rs:96-366) ----------
        {
          const int pl = lane >> 5, ll = lane & 31;
          const int y = ll >> 2, x0 = (ll & 3) * 2;
          const int cx = x0 >> 2, cy = y >> 2;
          const uint8_t* tc = ws->tileC[pl];
          int p0 = 0, p1 = 0;
          if (cmode == 0) {
            const unsigned tw = *(const unsigned*)&tc[TC(4 * cx, -1)];
            const unsigned lw = *(const unsigned*)&ws->leftC[pl][4 * cy];
            const int st = (int)sum4(tw), sl = (int)sum4(lw);
            const bool tAv = mbB, lAv = mbA;
            int v;
            if (cx == cy) {
              // blocks (0,0) and (4,4): trans_chroma.rs:174-226 incl. quirk Q2
              if (tAv && lAv) v = (st + sl + 4) >> 3;
              else if (!tAv && lAv) v = (sl + 2) >> 2;
              else if (tAv && bytes_nonzero(tw)) v = (st + 2) >> 2;  // left missing: top needs all > 0
              else v = 128;
            } else if (cx == 1) {
              // block (4,0): :227-252
              if (tAv) v = (st + 2) >> 2;
              else if (lAv && (lw >> 24) != 0) v = (sl + 2) >> 2;
              else v = 128;
            } else {
              // block (0,4): :253-278
              if (lAv && (lw >> 24) != 0) v = (sl + 2) >> 2;
              else if (tAv && (tw >> 24) != 0) v = (st + 2) >> 2;
              else v = 128;
            }
            p0 = p1 = v;
          } else if (cmode == 1) {
            if (mbA) p0 = p1 = ws->leftC[pl][y];
          } else if (cmode == 2) {
            if (mbB) {
              p0 = tc[TC(x0, -1)];
              p1 = tc[TC(x0 + 1, -1)];
            }
          } else {
            if (mbA && mbB) {
              // plane: :319-363. lanes (per 32-lane half) 0..3 horizontal terms, 4..7 vertical
              const int k = ll & 3;
              int term = 0;
              if (ll < 4) {
                term = (k + 1) * ((int)tc[TC(4 + k, -1)] - (int)tc[TC(2 - k, -1)]);
              } else if (ll < 8) {
                const int b = k == 3 ? (int)tc[TC(-1, -1)] : (int)ws->leftC[pl][2 - k];
                term = (k + 1) * ((int)ws->leftC[pl][4 + k] - b);
              }
              term += shfl(term, lane ^ 1);
              term += shfl(term, lane ^ 2);
              const int hs = shfl(term, lane & 32), vs = shfl(term, (lane & 32) + 4);
              const int a = 16 * ((int)ws->leftC[pl][7] + (int)tc[TC(7, -1)]);
              const int b = (34 * hs + 32) >> 6, c = (34 * vs + 32) >> 6;
              p0 = clip255((a + b * (x0 - 3) + c * (y - 3) + 16) >> 5);
              p1 = clip255((a + b * (x0 + 1 - 3) + c * (y - 3) + 16) >> 5);
            }
          }
          const int r0 = ws->resC[pl][y * 8 + x0], r1 = ws->resC[pl][y * 8 + x0 + 1];
          const unsigned short o = (unsigned short)(clip255(p0 + r0) | (clip255(p1 + r1) << 8));
          *(unsigned short*)&ws->tileC[pl][TC(x0, y)] = o;
        }
      }
      WAVE_SYNC();

      // ---- 7. write-out: planes (HBM), ring slot for the row below, left edges --------------
      {
        const int y = lane >> 2, xw = (lane & 3) * 4;
        const unsigned w = *(const unsigned*)&ws->tileY[TY(xw, y)];
        *(unsigned*)(planeY + (size_t)(r * 16 + y) * pitchY + mx * 16 + xw) = w;
        if (y == 15) *(unsigned*)(myLineY + mx * 16 + xw) = w;
        if (lane < 32) {
          const int pl = lane >> 4, cyy = (lane & 15) >> 1, cxw = (lane & 1) * 4;
          const unsigned cw = *(const unsigned*)&ws->tileC[pl][TC(cxw, cyy)];
          uint8_t* pc = pl ? planeCr : planeCb;
          *(unsigned*)(pc + (size_t)(r * 8 + cyy) * pitchC + mx * 8 + cxw) = cw;
          if (cyy == 7) *(unsigned*)(myLineC + pl * lineC_stride + mx * 8 + cxw) = cw;
        }
      }
      // modes seen by the neighbours: bits 0..7 kind, 8..23 four 4-bit modes
      unsigned bm = 0x2222u, rm = 0x2222u;  // not Intra4x4/8x8 -> DC
      if (kind == 0 || kind == 1) {
        bm = (unsigned)ws->mgrid[4][1] | ((unsigned)ws->mgrid[4][2] << 4) | ((unsigned)ws->mgrid[4][3] << 8) |
             ((unsigned)ws->mgrid[4][4] << 12);
        rm = (unsigned)ws->mgrid[1][4] | ((unsigned)ws->mgrid[2][4] << 4) | ((unsigned)ws->mgrid[3][4] << 8) |
             ((unsigned)ws->mgrid[4][4] << 12);
      }
      if (lane == 0) myLineM[mx] = (unsigned)(kind & 3) | (bm << 8);
      leftM = (unsigned)(kind & 3) | (rm << 8);
      WAVE_SYNC();
      // left edges for the next macroblock of this row
      if (lane < 16) {
        const uint8_t v = ws->tileY[TY(15, lane)];
        ws->leftY[lane] = v;
        ws->tileY[TY(-1, lane)] = v;
      } else if (lane < 32) {
        const int pl = (lane - 16) >> 3, yy = (lane - 16) & 7;
        const uint8_t v = ws->tileC[pl][TC(7, yy)];
        ws->leftC[pl][yy] = v;
        ws->tileC[pl][TC(-1, yy)] = v;
      }
      WAVE_SYNC();
      // ---- 8. publish progress ---------------------------------------------------------------
      if (lane == 0)
        __hip_atomic_store(&prog[slot], ((unsigned)(Rg + 1) << 12) | (unsigned)(mx + 1), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

size_t recon_lds_bytes(int W, int NW) {
  const size_t lineY = (size_t)W * 16 + 48, lineC = (size_t)W * 8 + 32;
  return 1808 + (size_t)NW * lineY + (size_t)NW * 2 * lineC + (size_t)NW * W * 4 + (size_t)NW * sizeof(WaveScratch);
}

hipError_t recon_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                        int NW, int grid, hipStream_t stream) {
  const size_t ldsBytes = recon_lds_bytes(P.W, NW);
  hipError_t e = hipFuncSetAttribute((const void*)recon_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)ldsBytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(recon_kernel, dim3(grid), dim3(NW * 64), ldsBytes, stream, P, (const dryv_mb_desc*)d_mbs,
                     (const int16_t*)d_coeffs, (uint8_t*)d_yuv, d_status);
  return hipGetLastError();
}

}  // namespace dryv
