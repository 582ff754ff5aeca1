// How fast does LDS-DMA (buffer_load_dwordx4 ... lds) move a halo-tile ROW PATTERN compared with a linear run?  One
// wave-instruction = 64 lanes x 16 B.  Patterns: (a) linear 1 KiB; (b) "NHWC C=96 chunk": 6 slots of 16 B per pixel, pixels
// 192 B apart (the 48-channel chunk of a 96-channel NHWC row: half of every 128-B line); (c) "NHWC C=192 chunk": pixels
// 384 B apart; (d) "NHWC C=384 chunk": 768 B apart; rows of 22 pixels (132 lanes' worth: 3 instructions per row, the last
// partly masked) as the streaming conv kernels request them.  Source L2-resident (64 KiB per workgroup) or beyond L2
// (4 MiB per workgroup, 1 GiB in all: Infinity Cache / HBM).  4 or 8 waves per workgroup, 8 instructions in flight each.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dmapat tools/probes/dmapattern_probe.hip && /tmp/dmapat
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// pix_stride: bytes between pixels of a row (96 = dense / plane-major); a "row" = 22 pixels x 96 B = 2112 B of payload
__global__ void __launch_bounds__(1024) k_rows(const char* base, size_t wg_stride, int region_bytes, int pix_stride, int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* p = base + (size_t)blockIdx.x * wg_stride;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, region_bytes, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_w = blockDim.x >> 6;
  const int row_span = 22 * pix_stride;                       // bytes of source a row covers
  const int n_rows = region_bytes / row_span;
  int voff[3];
  for (int k = 0; k < 3; ++k) {
    const int q = k * 64 + lane;                              // 16-byte slot of the row: pixel q / 6, slot q % 6
    voff[k] = q < 132 ? (q / 6) * pix_stride + (q % 6) * 16 : (int)0x80000000;
  }
  char* dst = smem + wv * (9 * 1024);
  int row = wv;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {                             // 3 rows = 9 instructions in flight
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + (u * 3 + k) * 1024), 16, voff[k], row * row_span, 0, 0);
      row += n_w;
      if (row >= n_rows) row -= n_rows;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (iters < 0) out[0] = *reinterpret_cast<int*>(smem + threadIdx.x * 16);
}

int main() {
  const size_t total = (size_t)256 * (4u << 20);
  char* a; int* out;
  if (hipMalloc(&a, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&out, 64);
  (void)hipMemset(a, 1, total);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
  struct Src { const char* name; size_t stride; int region; } srcs[] = {
      {"L2-resident (64 KiB per workgroup)", 64u << 10, 64 << 10}, {"beyond L2 (4 MiB per workgroup, 1 GiB in all)", 4u << 20, 4 << 20}};
  struct Pat { const char* name; int pix_stride; } pats[] = {
      {"dense rows (plane-major / C = 48)", 96}, {"NHWC C = 96 chunk (96 of every 192 B)", 192},
      {"NHWC C = 192 chunk (96 of every 384 B)", 384}, {"NHWC C = 384 chunk (96 of every 768 B)", 768}};
  for (const Src& s : srcs) {
    printf("== source: %s\n", s.name);
    for (const Pat& p : pats) {
      for (int waves : {2, 4, 8}) {
        const size_t payload_per_wg = (size_t)(8 << 20);                    // useful bytes per workgroup
        const int iters = (int)(payload_per_wg / ((size_t)waves * 3 * 2112));
        const float ms = timeit([&] { hipLaunchKernelGGL(k_rows, dim3(256), dim3(waves * 64), waves * 9 * 1024, 0, a, s.stride, s.region, p.pix_stride, iters, out); });
        printf("  %-42s %d waves: %6.1f B/clk/CU of payload (%.2f TB/s chip)\n", p.name, waves,
               (double)iters * waves * 3 * 2112 / (ms * 1e-3) / clk, 256.0 * iters * waves * 3 * 2112 / (ms * 1e-3) * 1e-12);
      }
    }
  }
  return 0;
}
