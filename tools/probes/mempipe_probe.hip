// Is the ~10 B/clk a compute unit gets from HBM-resident data a limit of the CU's own memory pipeline or the chip's
// HBM bandwidth divided by 256?  (Question behind the C >= 96 streaming conv, DESIGN.md section 4 "Round 3": a unit moves
// 136-779 KiB through the CU for 6,720-26,880 cycles of MFMAs.)  The same streaming loop on 32 / 64 / 128 / 256 workgroups
// (one per CU, the others idle), for data that is (a) streamed from HBM (1 GiB touched per pass: beyond the 256 MiB
// Infinity Cache), (b) Infinity-Cache resident (128 MiB in all, 16 MiB per XCD: beyond the 4 MiB L2s), (c) L2 resident;
// plain 16-byte loads, LDS-DMA, write-through and plain 16-byte stores.  Reports bytes / clock / CU (clock 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mempipe tools/probes/mempipe_probe.hip && /tmp/mempipe
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
  int piece = wv;
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

template <int WT>
__global__ void __launch_bounds__(1024) k_store(u4* base, size_t wg_stride16, int region16, int iters) {
  u4* p = base + (size_t)blockIdx.x * wg_stride16;
  const int n_thr = blockDim.x;
  int i = threadIdx.x;
  u4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + i), "v"(v) : "memory");
      else p[i] = v;
      i += n_thr;
      if (i >= region16) i -= region16;
    }
  }
}

int main() {
  const size_t total = (size_t)256 * (64u << 20);
  u4* a; u4* out;
  if (hipMalloc(&a, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64);
  hipMemset(a, 1, total);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
  for (int n_wg : {32, 64, 128, 256}) {
    struct Mode { const char* name; size_t stride; int region; size_t bytes_per_wg; } modes[] = {
        {"HBM stream (1 GiB / n_wg per WG per pass, 64 MiB apart)", (64u << 20) / 16, (int)(((1u << 30) / n_wg) / 16), (size_t)(1u << 30) / n_wg},
        {"Infinity-Cache resident (128 MiB / n_wg per WG, re-read)", (size_t)((128u << 20) / n_wg) / 16, (int)(((128u << 20) / n_wg) / 16), (size_t)(4u << 20) * (256 / n_wg) * 2},
        {"L2 resident (64 KiB per WG, re-read)", (64u << 10) / 16, (64 << 10) / 16, (size_t)(8 << 20)}};
    printf("==== %d workgroups (one per CU; %d CUs idle)\n", n_wg, 256 - n_wg);
    for (const Mode& m : modes) {
      printf("== %s\n", m.name);
      for (int waves : {4, 16}) {
        const int thr = waves * 64;
        const int iters = (int)(m.bytes_per_wg / ((size_t)thr * 16 * 8));
        float ms = timeit([&] { hipLaunchKernelGGL(k_plain<8>, dim3(n_wg), dim3(thr), 0, 0, a, m.stride, m.region, iters, out); });
        printf("  plain loads, %2d waves x 8 in flight: %6.1f B/clk/CU  (%.2f TB/s)\n", waves,
               m.bytes_per_wg / (ms * 1e-3) / clk, n_wg * (double)m.bytes_per_wg / (ms * 1e-3) * 1e-12);
        ms = timeit([&] { hipLaunchKernelGGL(k_dma<8>, dim3(n_wg), dim3(thr), waves * 8 * 1024, 0, a, m.stride, m.region, iters, out); });
        printf("  LDS-DMA,     %2d waves x 8 in flight: %6.1f B/clk/CU  (%.2f TB/s)\n", waves,
               m.bytes_per_wg / (ms * 1e-3) / clk, n_wg * (double)m.bytes_per_wg / (ms * 1e-3) * 1e-12);
      }
    }
    {
      const size_t bytes_per_wg = (size_t)(1u << 30) / n_wg;
      const int reg16 = (int)(bytes_per_wg / 16);
      for (int waves : {4, 16}) {
        const int thr = waves * 64;
        const int iters = (int)(bytes_per_wg / ((size_t)thr * 16 * 8));
        float ms = timeit([&] { hipLaunchKernelGGL(k_store<1>, dim3(n_wg), dim3(thr), 0, 0, a, (size_t)(64u << 20) / 16, reg16, iters); });
        printf("== write-through 16-byte stores to HBM, %2d waves: %6.1f B/clk/CU  (%.2f TB/s)\n", waves,
               bytes_per_wg / (ms * 1e-3) / clk, n_wg * (double)bytes_per_wg / (ms * 1e-3) * 1e-12);
        ms = timeit([&] { hipLaunchKernelGGL(k_store<0>, dim3(n_wg), dim3(thr), 0, 0, a, (size_t)(64u << 20) / 16, reg16, iters); });
        printf("== plain 16-byte stores to HBM,         %2d waves: %6.1f B/clk/CU  (%.2f TB/s)\n", waves,
               bytes_per_wg / (ms * 1e-3) / clk, n_wg * (double)bytes_per_wg / (ms * 1e-3) * 1e-12);
      }
    }
  }
  return 0;
}
