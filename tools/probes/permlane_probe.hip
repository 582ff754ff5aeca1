// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do to two registers, lane by lane: prints the source lane
// and source register of every destination (a = 1000 + lane, b = 2000 + lane before the swap).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/permlane tools/probes/permlane_probe.hip && /tmp/permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  const unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[lane] = r[0];
  p[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[128 + lane] = q[0];
  p[192 + lane] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* names[4] = {"permlane16_swap result[0] (vdst = a)", "permlane16_swap result[1] (src0 = b)",
                          "permlane32_swap result[0] (vdst = a)", "permlane32_swap result[1] (src0 = b)"};
  for (int t = 0; t < 4; ++t) {
    printf("%s:\n", names[t]);
    for (int l = 0; l < 64; ++l) printf("%5u%s", h[t * 64 + l], (l & 15) == 15 ? "\n" : "");
  }
  return 0;
}
