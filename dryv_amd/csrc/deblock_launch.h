// deblock_launch.h — host-visible interface of deblock.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "deblock_kernel_params.h"

namespace dryv {
int deblock_waves_per_block();
int deblock_blocks_per_cu();
// grid = number of workgroups; any grid >= 1 is correct (bands come off one queue). gen: the launch's generation (the tag of
// the hand-off granules in the workspace's side buffers: never 0, never repeated while the side buffers are not zeroed)
hipError_t deblock_launch(const deblock::DParams& P, const void* d_mbs, void* d_yuv, unsigned* d_status, void* d_workspace,
                          int grid, unsigned gen, hipStream_t stream);
}  // namespace dryv
