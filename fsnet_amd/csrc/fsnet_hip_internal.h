#pragma once
#include "../../include/fsnet_hip.h"
// fs_conv3x3_halo_plan: when set (host, per thread), the 3x3 launch functions fill {kernel, blocks, tile pixels, tile
// channels} instead of launching
inline thread_local int32_t* fs_conv3x3_plan_slot = nullptr;
