// global -> LDS loads without a register stop (gfx950: buffer_load_dwordx4 ... lds), shared by the GEMM-shaped kernels.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) int i32x4;

// 16 bytes per lane: LDS byte lds_addr + 16 * lane receives the 16 bytes at buffer offset voff (zero when out of range).
// Inline assembly because hipcc orders every later ds_read behind an LDS-DMA it knows about with `s_waitcnt vmcnt(0)` —
// which would land the next stage before the current one is multiplied; the kernels count these loads themselves
// (fs_wait_vm in front of the stage barrier).
__device__ __forceinline__ void glds16(const i32x4 rsrc, int voff, unsigned lds_addr) {
  int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
// raw buffer descriptor over [base, base + bytes): wave-uniform by construction
__device__ __forceinline__ i32x4 make_rsrc(const void* base, long bytes) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int N>
__device__ __forceinline__ void fs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
