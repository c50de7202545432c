#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int v = threadIdx.x;
  int r = __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false);   // wave_ror:1
  int s = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
  out[threadIdx.x] = r;
  out[64 + threadIdx.x] = s;
}
int main() {
  int* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ror: %d %d %d ... %d\n", h[0], h[1], h[2], h[63]);
  printf("shr: %d %d %d ... %d\n", h[64], h[65], h[66], h[127]);
  return 0;
}
