// Why do the 8 row stores of a wave in conv_stream_kernel's epilogue take 4,500-11,000 cycles ("what bounds each
// kernel today")?  The kernel's shape in miniature: one workgroup per CU (LDS-limited), 4 "MFMA" waves that alternate between a
// compute phase (a dependent MFMA chain of ~CYC cycles, no memory traffic) and a store phase (S x 16-byte stores per lane = S
// KiB per wave, contiguous rows, new addresses every unit), optionally one loader wave that keeps K LDS-DMA pieces of 1 KiB in
// flight from a 128-MiB buffer.  Measured per unit and wave: cycles from the first store's issue to the last store's issue
// (the time the wave is blocked), for
//   sync   : every workgroup in the same phase (what the conv kernels do),
//   desync : workgroup b starts (b % 4) quarter-units late,
//   quarter: only every 4th workgroup stores at all (the chip-wide burst is a quarter as large),
// with write-through (sc0 sc1) or plain stores, with or without the loader.  If the store phase shrinks with `desync` /
// `quarter`, the epilogue is bound by the chip-wide burst (all 256 CUs storing at once); if not, by the CU's own pipeline.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/storeburst tools/probes/storeburst_probe.hip && /tmp/storeburst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Args {
  u4* out;                  // store target: [unit][workgroup][wave][S][64 lanes] x 16 B
  const char* src;          // loader source
  unsigned src_bytes;
  unsigned long long* res;  // [0] sum of store-phase cycles, [1] count, [2] sum of compute-phase cycles, [3] max kernel cycles
  int units, S, chain, mode, wt, loader_k;
  int res_loads;            // 16-byte loads per lane issued right BEFORE (1) / right AFTER (2) the stores of a unit (0: none): the
                            // residual rows of the next unit in conv_stream_kernel; consumed one unit later
  const u4* rsrc;           // their source (distinct lines per unit)
};

template <int WT>
__device__ __forceinline__ void store16(u4* p, u4 v) {
  if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int WT>
__global__ void __launch_bounds__(320) burst_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long k0 = __builtin_readcyclecounter();
  if (wv == 4) {
    // loader: K pieces of 1 KiB in flight, walking a region of its own
    if (a.loader_k <= 0) return;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.src), 0, (int)a.src_bytes, 0x00020000);
    const unsigned n_pieces = a.src_bytes / 1024u;
    unsigned piece = (blockIdx.x * 9973u) % n_pieces;
    // about as many pieces per unit as a tile loader of the conv kernel issues (76 KiB per unit)
    const int total = a.units * 76;
    for (int i = 0; i < total; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 65536 + ((i + j) & 31) * 1024), 16, lane * 16, (int)(piece * 1024u), 0, 0);
        piece = piece + 257u < n_pieces ? piece + 257u : piece + 257u - n_pieces;
      }
      if (a.loader_k <= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (a.loader_k <= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_sleep(20);                        // pace: ~76 pieces per ~16k-cycle unit
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // "MFMA" waves
  h8 x = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.5f};
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (a.mode == 1) {                                       // desync: quarter-unit offsets
    const int q = (int)(blockIdx.x >> 3) & 3;
    for (int i = 0; i < q * (a.chain / 4); ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, acc, 0, 0, 0);
  }
  const bool stores_on = a.mode != 2 || ((blockIdx.x >> 3) & 3) == 0;
  unsigned long long t_store = 0, t_comp = 0;
  u4 rr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rr[i] = u4{0u, 0u, 0u, 0u};
  for (int u = 0; u < a.units; ++u) {
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < a.chain; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, acc, 0, 0, 0);
    u4 v = __builtin_bit_cast(u4, acc);
    if (a.res_loads) {                                     // last unit's "residual rows" are consumed here
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) v ^= rr[i];
    }
    asm volatile("" : "+v"(v));
    const unsigned long long c1 = __builtin_readcyclecounter();
    const u4* rp = a.rsrc + ((((size_t)u * gridDim.x + blockIdx.x) * 4 + wv) * 8) * 64 + lane;
    if (a.res_loads == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[i]) : "v"(rp + i * 64) : "memory");
    }
    if (stores_on) {
      u4* p = a.out + ((((size_t)u * gridDim.x + blockIdx.x) * 4 + wv) * a.S) * 64 + lane;
      for (int s = 0; s < a.S; ++s) store16<WT>(p + s * 64, v);
    }
    const unsigned long long c2 = __builtin_readcyclecounter();
    if (a.res_loads == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[i]) : "v"(rp + i * 64) : "memory");
    }
    t_comp += c1 - c0;
    t_store += c2 - c1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && stores_on) {
    atomicAdd(&a.res[0], t_store);
    atomicAdd(&a.res[1], (unsigned long long)a.units);
    atomicAdd(&a.res[2], t_comp);
    atomicMax(&a.res[3], __builtin_readcyclecounter() - k0);
  }
}

int main(int argc, char** argv) {
  const int units = 12, S = 8, wgs = 256;
  const int chain = argc > 1 ? atoi(argv[1]) : 500;        // dependent 16x16x32 MFMAs per compute phase (~16 cycles each)
  Args a{};
  const size_t out_bytes = (size_t)units * wgs * 4 * S * 1024;
  hipMalloc((void**)&a.out, out_bytes);
  a.src_bytes = 128u << 20;
  char* src;
  hipMalloc((void**)&src, a.src_bytes);
  hipMemset(src, 1, a.src_bytes);
  a.src = src;
  hipMalloc((void**)&a.res, 64);
  a.units = units; a.S = S; a.chain = chain;
  u4* rsrc;
  hipMalloc((void**)&rsrc, out_bytes);
  hipMemset(rsrc, 0, out_bytes);
  a.rsrc = rsrc;
  hipFuncSetAttribute(reinterpret_cast<const void*>(burst_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(burst_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const char* mode_name[] = {"sync", "desync", "quarter"};
  printf("# %d workgroups (one per CU: 100 KiB LDS), 4 storing waves, %d units of %d MFMAs (~%d cycles) + %d x 1 KiB stores per wave\n",
         wgs, units, chain, chain * 16, S);
  printf("# mode     stores         loader(K) res-loads  store phase (cycles per unit and wave)   compute phase   kernel (cycles, max)   us\n");
  for (int wt = 1; wt >= 0; --wt)
    for (int lk = 0; lk <= 16; lk += 16)
     for (int rl = 0; rl <= 2; ++rl)
      for (int mode = 0; mode < 3; mode += 2) {
        a.mode = mode; a.wt = wt; a.loader_k = lk; a.res_loads = rl;
        double best_us = 1e30;
        unsigned long long r[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(a.res, 0, 64);
          hipEvent_t e0, e1;
          hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          if (wt) hipLaunchKernelGGL(burst_kernel<1>, dim3(wgs), dim3(320), 100 * 1024, 0, a);
          else hipLaunchKernelGGL(burst_kernel<0>, dim3(wgs), dim3(320), 100 * 1024, 0, a);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms * 1e3 < best_us) { best_us = ms * 1e3; hipMemcpy(r, a.res, 32, hipMemcpyDeviceToHost); }
          hipEventDestroy(e0); hipEventDestroy(e1);
        }
        const double n = r[1] ? (double)r[1] : 1.0;
        printf("  %-8s %-14s %-9d %-10s %-40.0f %-15.0f %-22llu %.1f\n", mode_name[mode], wt ? "write-through" : "plain", lk,
               rl == 0 ? "none" : rl == 1 ? "before" : "after", r[0] / n, r[2] / n, r[3], best_us);
      }
  return 0;
}
