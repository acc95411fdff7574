// deblock_launch.h — host-visible interface of deblock.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "deblock_kernel_params.h"

namespace dryv {
int deblock_waves_per_block();
int deblock_blocks_per_cu();
// grid = number of workgroups; any grid >= 1 is correct (bands come off one queue)
hipError_t deblock_launch(const deblock::DParams& P, const void* d_mbs, void* d_yuv, unsigned* d_status, void* d_workspace,
                          int grid, hipStream_t stream);
}  // namespace dryv
