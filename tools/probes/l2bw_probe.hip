// How many bytes per clock can ONE compute unit pull from its XCD's L2 (the question behind the C >= 96 streaming conv:
// 38-88 KiB of halo tiles and weights per 4,400-cycle stage)?  256 workgroups (one per CU), each re-reads its own region
// (regions of an XCD's workgroups sum to <= 2 MiB: L2 hits after the first pass; or 64 MiB apart: HBM), with
//   (a) plain 16-byte loads into registers, U in flight per wave,  (b) LDS-DMA (buffer_load ... lds), U per wave,
// by waves per workgroup.  Reports bytes / clock / CU at the measured kernel time (clock taken as 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2bw tools/probes/l2bw_probe.hip && /tmp/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int U>
__global__ void __launch_bounds__(1024) k_plain(const u4* base, size_t wg_stride16, int region16, int iters, u4* out) {
  const u4* p = base + (size_t)blockIdx.x * wg_stride16;
  u4 acc = {0u, 0u, 0u, 0u};
  const int n_thr = blockDim.x;
  int i = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = p[i];
      i += n_thr;
      if (i >= region16) i -= region16;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if (acc.x == 0x12345678u) out[0] = acc;
}

template <int U>
__global__ void __launch_bounds__(1024) k_dma(const u4* base, size_t wg_stride16, int region16, int iters, u4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* p = reinterpret_cast<const char*>(base + (size_t)blockIdx.x * wg_stride16);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, region16 * 16, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_w = blockDim.x >> 6;
  int piece = wv;                                   // 1-KiB pieces of the region, round robin over the waves
  const int n_pieces = region16 / 64;
  char* dst = smem + wv * (U * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + u * 1024), 16, lane * 16, piece * 1024, 0, 0);
      piece += n_w;
      if (piece >= n_pieces) piece -= n_pieces;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (iters < 0) out[0] = *reinterpret_cast<u4*>(smem + threadIdx.x * 16);
}

int main() {
  const size_t total = (size_t)256 * (64u << 20);     // up to 64 MiB apart
  u4* a; u4* out;
  if (hipMalloc(&a, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64);
  hipMemset(a, 1, total);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<22>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch) {
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
  };
  const double clk = 2.4e9;
  struct Mode { const char* name; size_t stride; int region; } modes[] = {
      {"L2-resident (64 KiB per CU, re-read)", (64u << 10) / 16, (64 << 10) / 16},
      {"HBM (4 MiB per CU, streamed once per pass)", (64u << 20) / 16, (4 << 20) / 16}};
  for (const Mode& m : modes) {
    printf("== %s\n", m.name);
    for (int waves : {2, 4, 8, 16}) {
      const int thr = waves * 64;
      const size_t bytes_per_wg = (size_t)(8 << 20);                      // every workgroup moves 8 MiB
      {
        constexpr int U = 8;
        const int iters = (int)(bytes_per_wg / ((size_t)thr * 16 * U));
        const float ms = timeit([&] { hipLaunchKernelGGL(k_plain<U>, dim3(256), dim3(thr), 0, 0, a, m.stride, m.region, iters, out); });
        printf("  plain loads, %2d waves x %d in flight: %6.1f B/clk/CU  (%.2f TB/s chip)\n", waves, U,
               bytes_per_wg / (ms * 1e-3) / clk, 256.0 * bytes_per_wg / (ms * 1e-3) * 1e-12);
      }
      {
        constexpr int U = 8;
        const int iters = (int)(bytes_per_wg / ((size_t)thr * 16 * U));
        const float ms = timeit([&] { hipLaunchKernelGGL(k_dma<U>, dim3(256), dim3(thr), waves * U * 1024, 0, a, m.stride, m.region, iters, out); });
        printf("  LDS-DMA,     %2d waves x %d in flight: %6.1f B/clk/CU  (%.2f TB/s chip)\n", waves, U,
               bytes_per_wg / (ms * 1e-3) / clk, 256.0 * bytes_per_wg / (ms * 1e-3) * 1e-12);
      }
    }
  }
  {   // 3 loader waves with 22 pieces in flight each: the conv kernel's shape
    constexpr int U = 22;
    const int thr = 192;
    const size_t bytes_per_wg = (size_t)(8 << 20);
    const int iters = (int)(bytes_per_wg / ((size_t)thr * 16 * U));
    const float ms = timeit([&] { hipLaunchKernelGGL(k_dma<U>, dim3(256), dim3(thr), 3 * U * 1024, 0, a, modes[0].stride, modes[0].region, iters, out); });
    printf("== LDS-DMA, 3 waves x 22 in flight, L2-resident: %6.1f B/clk/CU\n", bytes_per_wg / (ms * 1e-3) / clk);
  }
  return 0;
}
