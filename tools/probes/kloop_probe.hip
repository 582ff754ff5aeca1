// GPU probe: cycles per MFMA of an LDS-fed k loop (v_mfma_f32_16x16x32_f16) as a function of the per-wave tile
// (MT x NT MFMAs per k-step), of how many operand fragments come from LDS per k-step, and of the prefetch depth.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/kloop_probe tools/probes/kloop_probe.hip && /tmp/kloop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MT, int NT, int ALDS, int BLDS, int D, int PITCH>
__global__ void __launch_bounds__(512) probe(unsigned long long* out, float* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  const char* wl = smem + lane * 16;                         // A fragments: 1 KiB per (k, m)
  const char* xb = smem + 43008 + ((wv & 3) * 16 + r) * PITCH + g * 16;   // B fragments: pixel pitch PITCH
  float4v acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) acc[m][n] = float4v{0, 0, 0, 0};
  half8 af[D + 1][MT], bf[D + 1][NT];
  for (int s = 0; s <= D; ++s) {
    for (int m = 0; m < MT; ++m) af[s][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
    for (int n = 0; n < NT; ++n) bf[s][n] = *reinterpret_cast<const half8*>(xb + n * 16 * PITCH);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
      const int cur = kk % (D + 1), nxt = (kk + D) % (D + 1);
      if (ALDS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) af[nxt][m] = *reinterpret_cast<const half8*>(wl + (((kk + D) % 14) * MT + m) * 1024);
      }
      if (BLDS) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[nxt][n] = *reinterpret_cast<const half8*>(xb + n * 16 * PITCH + ((kk + D) % 14) * 64);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][n], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < (ALDS ? MT : 0) + (BLDS ? NT : 0); ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int MT, int NT, int ALDS, int BLDS, int D, int PITCH>
void run(const char* name, int waves, unsigned long long* d_out, float* d_sink) {
  const int reps = 2000;
  auto k = probe<MT, NT, ALDS, BLDS, D, PITCH>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, d_out, d_sink, reps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, d_out, d_sink, reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int b = 0; b < 256; ++b) sum += (double)h[b * 8];
  const double cyc = sum / 256 / (reps * 14.0 * MT * NT);
  const double mfmas = (double)reps * 14 * MT * NT * waves * 256;
  printf("%-50s waves/CU %d: %6.1f ticks per MFMA per wave | %7.1f us, %6.1f TFLOP/s, %.2f ticks/ns\n", name, waves, cyc,
         ms * 1e3, mfmas * 16384 / (ms * 1e-3) * 1e-12, sum / 256 / (ms * 1e6));
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 256 * 8 * 8);
  hipMalloc(&d_sink, 256 * 512 * 4);
  run<3, 3, 0, 0, 1, 96>("3x3 registers only", 4, d_out, d_sink);
  run<3, 3, 0, 1, 1, 96>("3x3 B from LDS (3 reads), depth 1", 4, d_out, d_sink);
  run<3, 3, 1, 1, 1, 96>("3x3 A+B from LDS (6 reads), depth 1", 4, d_out, d_sink);
  run<3, 3, 1, 1, 2, 96>("3x3 A+B from LDS (6 reads), depth 2", 4, d_out, d_sink);
  run<3, 3, 1, 1, 2, 112>("3x3 A+B from LDS (6 reads), depth 2, pitch 112", 4, d_out, d_sink);
  run<3, 5, 1, 1, 1, 96>("3x5 A+B from LDS (8 reads), depth 1", 4, d_out, d_sink);
  run<3, 4, 1, 1, 1, 96>("3x4 A+B from LDS (7 reads), depth 1", 4, d_out, d_sink);
  run<3, 2, 0, 1, 1, 96>("3x2 B from LDS (2 reads), depth 1", 4, d_out, d_sink);
  run<3, 2, 0, 1, 2, 96>("3x2 B from LDS (2 reads), depth 2", 4, d_out, d_sink);
  run<3, 3, 1, 1, 2, 96>("3x3 A+B from LDS (6 reads), depth 2", 8, d_out, d_sink);
  run<3, 2, 0, 1, 2, 96>("3x2 B from LDS (2 reads), depth 2", 8, d_out, d_sink);
  run<3, 3, 0, 0, 1, 96>("3x3 registers only", 8, d_out, d_sink);
  run<6, 2, 1, 1, 1, 96>("6x2 A+B from LDS (8 reads), depth 1", 4, d_out, d_sink);
  run<3, 6, 1, 1, 1, 96>("3x6 A+B from LDS (9 reads), depth 1", 4, d_out, d_sink);
  return 0;
}
