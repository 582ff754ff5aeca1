// LDS-DMA throughput of one CU by the number of loader waves and the number of 1-KiB pieces each keeps in flight (256
// workgroups, data beyond L2: 1 GiB swept).  The streaming conv kernels have 2 tile loaders + 1 weight loader; their tile
// requests take ~250 cycles each to issue.  Is that the waves, or the pieces in flight?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dmadepth tools/probes/dmadepth_probe.hip && /tmp/dmadepth
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int U>
__global__ void __launch_bounds__(1024) k_dma(const char* base, size_t wg_stride, int region_bytes, int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* p = base + (size_t)blockIdx.x * wg_stride;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, region_bytes, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_w = blockDim.x >> 6;
  int piece = wv;
  const int n_pieces = region_bytes / 1024;
  char* dst = smem + wv * (U * 1024);
  // rolling window: U pieces in flight; wait for the oldest before issuing the next
#pragma unroll
  for (int u = 0; u < U; ++u) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + u * 1024), 16, lane * 16, piece * 1024, 0, 0);
    piece += n_w; if (piece >= n_pieces) piece -= n_pieces;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U - 1) : "memory");
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + u * 1024), 16, lane * 16, piece * 1024, 0, 0);
      piece += n_w; if (piece >= n_pieces) piece -= n_pieces;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (iters < 0) out[0] = *reinterpret_cast<int*>(smem + threadIdx.x * 16);
}

template <int U>
static void run(const char* a, int* out, int waves, size_t stride, int region, hipEvent_t e0, hipEvent_t e1) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if ((size_t)waves * U * 1024 > 160 * 1024) return;
  const size_t bytes_per_wg = (size_t)(4 << 20);
  const int iters = (int)(bytes_per_wg / ((size_t)waves * U * 1024));
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_dma<U>, dim3(256), dim3(waves * 64), waves * U * 1024, 0, a, stride, region, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  printf("  %d waves x %2d in flight: %6.1f B/clk/CU (%.2f TB/s)\n", waves, U, (double)iters * waves * U * 1024 / (best * 1e-3) / 2.4e9,
         256.0 * iters * waves * U * 1024 / (best * 1e-3) * 1e-12);
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* a; int* out;
  if (hipMalloc(&a, total) != hipSuccess) return 1;
  (void)hipMalloc(&out, 64); (void)hipMemset(a, 1, total);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    const size_t stride = mode == 0 ? (4u << 20) : (64u << 10);
    const int region = mode == 0 ? (4 << 20) : (64 << 10);
    printf("== %s\n", mode == 0 ? "beyond L2 (4 MiB per workgroup, 1 GiB in all)" : "L2-resident (64 KiB per workgroup)");
    for (int waves : {1, 2, 3, 4, 8}) {
      run<4>(a, out, waves, stride, region, e0, e1);
      run<8>(a, out, waves, stride, region, e0, e1);
      run<16>(a, out, waves, stride, region, e0, e1);
      run<32>(a, out, waves, stride, region, e0, e1);
    }
  }
  return 0;
}
