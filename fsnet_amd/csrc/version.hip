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

// Where is `stream` in the hipGraph capture it belongs to?  *token = a hash of the graph nodes the stream's next node would
// depend on (hipStreamGetCaptureInfo_v2), 0 when the stream is not capturing.  Two calls return the same token iff nothing was
// captured on the stream in between.  The engine puts hand-overs to other streams off until the chain has captured its next
// kernel (nets.flush_deferred: the HIP graph executor assigns its streams by the order of a node's outgoing edges).
extern "C" int fs_capture_position(void* stream, unsigned long long* token) {
  if (!token) return FS_EINVAL;
  *token = 0;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t n = 0;
  hipError_t e = hipStreamGetCaptureInfo_v2((hipStream_t)stream, &status, &id, &graph, &deps, &n);
  if (e != hipSuccess) { (void)hipGetLastError(); return FS_ELAUNCH; }
  if (status != hipStreamCaptureStatusActive) return FS_OK;
  unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)n;      // order-independent: a sum of mixed pointers
  for (size_t i = 0; i < n; ++i) {
    unsigned long long x = (unsigned long long)(uintptr_t)deps[i];
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    h += x;
  }
  *token = h ? h : 1;
  return FS_OK;
}
