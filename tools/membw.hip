// Streaming bandwidth probe for the MI355X memory path as a kernel sees it: pure writes, pure reads and
// copies with 16 bytes per lane, by workgroup count / waves per workgroup.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membw tools/membw.hip && /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void k_write(u4* p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const u4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void k_read(const u4* p, size_t n, u4* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  u4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += p[i];
  if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void k_copy(const u4* s, u4* d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
// contiguous chunk per workgroup (a persistent kernel's pattern) instead of a grid-stride sweep
__global__ void k_write_chunk(u4* p, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
  const u4 v = {1u, 2u, 3u, 4u};
  for (size_t i = b + threadIdx.x; i < e; i += blockDim.x) p[i] = v;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  u4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch) {
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
  };
  const int grids[] = {256, 1024, 4096, 16384};
  const int blocks[] = {256, 1024};
  for (int g : grids) for (int bs : blocks) {
    const float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(bs), 0, 0, a, n); });
    const float wc = timeit([&] { hipLaunchKernelGGL(k_write_chunk, dim3(g), dim3(bs), 0, 0, a, n); });
    const float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(bs), 0, 0, a, n, b); });
    const float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(bs), 0, 0, a, b, n); });
    printf("grid %5d x %4d threads: write %.2f TB/s (chunked %.2f)  read %.2f TB/s  copy %.2f TB/s (read+write bytes)\n", g, bs,
           bytes / w * 1e-9, bytes / wc * 1e-9, bytes / r * 1e-9, 2.0 * bytes / c * 1e-9);
  }
  return 0;
}
