#include "common.h"
#include "fsnet_hip_internal.h"
extern "C" int fs_abi_version(void) { return FS_ABI_VERSION; }
extern "C" const char* fs_target_arch(void) { return "gfx950"; }

// Debugging aid (FSNET_AMD_MARKS=1): one thread writes the constant-rate (100 MHz) device clock into *slot — a node
// of the captured step on whatever stream it is issued, so a replayed step reports when each chain reached each
// point WITHOUT a profiler in the way (rocprofv3's packet interception slows submission enough to hide the overlap
// of the graph's branches).
namespace {
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}
extern "C" int fs_debug_timestamp(void* slot, void* stream) {
  if (!slot) return FS_EINVAL;
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
  return fs_launch_status();
}
