#pragma once
#include "../../include/fsnet_hip.h"
// fs_conv3x3_halo_plan: when set (host, per thread), the 3x3 launch functions fill {kernel, blocks, tile pixels, tile
// channels} instead of launching
inline thread_local int32_t* fs_conv3x3_plan_slot = nullptr;

// Two problems in ONE launch ("lanes"): the depth encoder and the stacked pose encoder run the same layer shapes after
// their stems (resnet.py:199-213, invoked per monodepth2_model.py:24-43), so every post-stem launch of the two networks
// is issued once with two argument sets.  The first nb0 blocks of the leading grid dimension work on a[0] / g[0], the
// rest on a[1] / g[1]; a launch with one problem sets nb0 to the whole grid.  The kernels index the argument pair with a
// block-uniform value, which the compiler turns into scalar loads from the kernarg segment at a computed offset (no
// scratch copy of the structs: tests/test_no_spills_cpu.py).
template <typename A, typename G>
struct FsDual {
  A a[2];
  G g[2];
  int nb0;
  int nprob;
};
struct FsNoGeom { int unused; };
// leading-dimension blocks of a problem rounded up to whole XCD rounds: block b of the grid runs on XCD b % 8, and the
// kernels' tile mappings derive the XCD from the problem-local block index
inline int fs_xcd_round(long blocks) { return (int)((blocks + 7) / 8 * 8); }

// the LDS-staged GEMM form of fs_conv1x1 (conv1x1_gemm.hip); FS_EINVAL = not a case for it (the row-streaming kernel of
// conv1x1.hip then takes the launch).
int fs_conv1x1_gemm(const FsConvArgs& a, hipStream_t st);
