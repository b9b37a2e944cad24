#include "common.h"
#include "fsnet_hip_internal.h"
extern "C" int fs_abi_version(void) { return 2; }
extern "C" const char* fs_target_arch(void) { return "gfx950"; }
