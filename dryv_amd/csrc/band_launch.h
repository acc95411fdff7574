// band_launch.h — launch interface between the host API (recon_api.hip) and the gfx950 band kernel (recon_band.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dryv_recon.h"
#include "kparams.h"

namespace dryv {

// Band kernel (band_kernel.h): a team of three waves (four with the 8x8 transform) per 4-row band; grid = workgroups of band_teams_per_block() teams.
size_t band_lds_bytes(bool hasI8, bool wide, int teams);
int band_teams_per_block(bool hasI8, bool wide);
int band_blocks_per_cu(bool hasI8, bool wide);
size_t band_workspace_bytes(const KParams& P);
size_t band_reset_bytes(const KParams& P);      // leading bytes of the workspace (task counter, progress words): zeroed with the
                                                //   hand-off records, once per workspace layout
size_t band_handoff_offset(const KParams& P);   // the hand-off records between bands: tagged with the launch's generation, zeroed
size_t band_handoff_bytes(const KParams& P);    //   once per workspace layout (never per launch)
size_t band_profile_offset(const KParams& P);   // diagnostic builds: per-wave phase sums / breadcrumbs behind the workspace
// gen: the launch's generation: its low 21 bits not 0, and different from that of every earlier launch on this workspace
// layout. task_base: what the workspace's task counter stands at when the launch starts: the counter is never reset between
// launches -- a launch of `grid` workgroups adds band_claims_per_launch() to it.
hipError_t band_launch(const KParams& P, const void* d_mbs, const void* d_coeffs, void* d_yuv, unsigned* d_status,
                       void* d_workspace, int grid, bool wide, unsigned batch_seq, unsigned gen, unsigned task_base, hipStream_t stream,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);   // (events: the kernel's own start / end)
// every band task is claimed once, and every team's FRONT wave claims once more to learn that none are left
unsigned band_claims_per_launch(const KParams& P, int grid, bool wide);

}  // namespace dryv
