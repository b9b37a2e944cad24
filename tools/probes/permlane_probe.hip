// semantics check of v_permlane32_swap as used by conv3x3_t32.hip (build: hipcc --offload-arch=gfx950 -o /tmp/pp permlane_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
