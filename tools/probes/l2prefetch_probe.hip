// Does a CHEAP line touch (one dword per 128-byte line: 64 lines per wave-instruction, 256 B of return data) bring lines into
// L2 faster than the payload loads that follow can pull them from beyond L2?  The streaming conv kernels move ~1 KiB per
// ~125 cycles and CU whatever the structure (profiles/r04_stream_mempath_counters.txt); if what bounds that is the RETURN
// path of 1-KiB wave-instructions waiting for lines from the Infinity Cache / HBM, touching the lines of the tile of stage
// s + 2 first turns the LDS-DMA of stage s + 1 into L2 hits.
// Modes, n_wg workgroups of 4 waves (one per CU, the other CUs idle), each sweeping its own region beyond L2 (1 GiB in all):
//   dma       : LDS-DMA of dense 1-KiB pieces, 8 in flight per wave (the reference: mempipe_probe)
//   touch     : only the line touches (what a prefetch costs by itself), reported as the bytes of the lines touched
//   touch+dma : every wave touches the lines of block i + D, then LDS-DMAs block i (D = 2 blocks of 8 KiB per wave ahead)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2pf tools/probes/l2prefetch_probe.hip && /tmp/l2pf
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>   // 0 dma, 1 touch, 2 touch + dma
__global__ void __launch_bounds__(256) k(const char* base, size_t wg_stride, int region_bytes, int iters, int dist, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* p = base + (size_t)blockIdx.x * wg_stride;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, region_bytes, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_blocks = region_bytes / (4 * 8192);          // a block = 8 KiB per wave, 32 KiB per workgroup
  char* dst = smem + wv * 8192;
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int b = it % n_blocks;
    if (MODE >= 1) {                                         // one dword of each of the 64 lines of this wave's 8 KiB of block b + dist
      const int bt = (it + dist) % n_blocks;
      acc += __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 128, (bt * 4 + wv) * 8192, 0);
    }
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + u * 1024), 16, lane * 16, (b * 4 + wv) * 8192 + u * 1024, 0, 0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");       // (the touch of this iteration may stay in flight)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* a; int* out;
  if (hipMalloc(&a, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&out, 64);
  (void)hipMemset(a, 1, total);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](auto launch) {
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
  };
  const double clk = 2.4e9;
  for (int n_wg : {32, 64, 256}) {
    const size_t stride = total / n_wg;
    const int region = (int)(stride > (1u << 30) ? (1u << 30) : stride);
    const int iters = region / (4 * 8192);                   // one sweep of the region
    printf("==== %d workgroups x 4 waves, %d MiB each (beyond L2 and, in all, beyond the Infinity Cache)\n", n_wg, region >> 20);
    const double bytes = (double)iters * 4 * 8192;
    float ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(n_wg), dim3(256), 4 * 8192, 0, a, stride, region, iters, 0, out); });
    printf("  dma only                     : %6.1f B/clk/CU (%.2f TB/s)\n", bytes / (ms * 1e-3) / clk, n_wg * bytes / (ms * 1e-3) * 1e-12);
    ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(n_wg), dim3(256), 4 * 8192, 0, a, stride, region, iters, 0, out); });
    printf("  touch only (lines, as bytes) : %6.1f B/clk/CU (%.2f TB/s)\n", bytes / (ms * 1e-3) / clk, n_wg * bytes / (ms * 1e-3) * 1e-12);
    for (int dist : {1, 2, 4, 8}) {
      ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(n_wg), dim3(256), 4 * 8192, 0, a, stride, region, iters, dist, out); });
      printf("  touch block i+%d, dma block i : %6.1f B/clk/CU (%.2f TB/s)\n", dist, bytes / (ms * 1e-3) / clk, n_wg * bytes / (ms * 1e-3) * 1e-12);
    }
  }
  return 0;
}
