// recon_kernel.h — launch interface between the host API (recon_api.hip) and the gfx950 kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"
#include "kparams.h"

namespace dryv {

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

// Band kernel (band_kernel.h): a team of three waves per 4-row band; grid = workgroups of band_teams_per_block() teams.
size_t band_lds_bytes(bool hasI8, int teams);
int band_teams_per_block(bool hasI8, bool wide);
int band_blocks_per_cu();
size_t band_workspace_bytes(const KParams& P);
size_t band_reset_bytes(const KParams& P);      // leading bytes of the workspace a launch needs zeroed
size_t band_profile_offset(const KParams& P);   // diagnostic builds: per-wave phase sums / breadcrumbs behind the workspace
hipError_t band_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                       void* d_workspace, int grid, bool wide, hipStream_t stream);

}  // namespace dryv
