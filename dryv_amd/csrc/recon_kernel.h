// recon_kernel.h — launch interface between the host API (recon_api.hip) and the gfx950 kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"

namespace dryv {

// Everything the kernel needs that is constant for a submit; passed by value in the kernarg
// segment and copied to LDS once per workgroup.
struct KParams {
  int W, H;         // picture size in macroblocks
  int n_frames;
  int cqo_cb;       // pps.chroma_qp_index_offset
  int cqo_cr;       // second_chroma_qp_index_offset
  uint16_t ls4[96];   // LevelScale4x4[m][i*4+j], scaling list 0 (transform.rs:22-45, quirk Q3)
  uint16_t ls8[384];  // LevelScale8x8[m][i*8+j], scaling list 0 (transform.rs:47-77)
  uint8_t t4[144];    // Intra4x4 gather table [mode][y*4+x]:  idx | sel << 5 (sel 0 E, 1 F, 2 G)
  uint8_t t8[576];    // Intra8x8 gather table [mode][y*8+x]
  uint8_t zz8i[64];   // raster position i*8+j -> index in the 8x8 zig-zag list (frame/mod.rs:212-284)
};

// Device workspace a launch needs (task counter, per-row progress words, bottom-row modes).
size_t recon_workspace_bytes(int W, int H, int n_frames);
hipError_t recon_reset_workspace(const KParams& P, void* d_workspace, int grid, hipStream_t stream);
// Bands (4 consecutive macroblock rows, one per wave) a workgroup works on at a time.
int recon_bands_per_block();
// Workgroups the kernel is compiled to keep resident per CU (register budget).
int recon_blocks_per_cu();
// grid = number of workgroups (recon_bands_per_block() x 4 waves each); any grid >= 1 is correct.
hipError_t recon_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv,
                        unsigned* d_status, void* d_workspace, int grid, hipStream_t stream);

// Dual-frame kernel (each wave works on the same row of two frames): 8-wave workgroups at 4 waves per SIMD.
long long recon_task_count_df(int H, int n_frames);
hipError_t recon_launch_df(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv,
                           unsigned* d_status, void* d_workspace, int grid, hipStream_t stream);

}  // namespace dryv
