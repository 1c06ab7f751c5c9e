// hipcc --offload-arch=gfx950 -O2 scripts/probes/permlane.hip -o /tmp/permlane && /tmp/permlane
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* out) {
  uint32_t l = threadIdx.x;
  uint32_t a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  uint32_t u = 1000 + l;
  auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int part = 0; part < 4; ++part) {
    printf("part %d:", part);
    for (int l : {0, 1, 31, 32, 33, 63}) printf(" [%d]=%u", l, h[part * 64 + l]);
    printf("\n");
  }
  return 0;
}
