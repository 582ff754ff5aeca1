// GPU probe: two waves per SIMD, one issuing a dense MFMA stream (operands in registers), the other VALU work
// (the BN / convert mix of the conv epilogues).  How much does each slow the other?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/coissue_probe tools/probes/coissue_probe.hip && /tmp/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run VALU; bit 2: the VALU waves also write LDS
__global__ void __launch_bounds__(512) probe(unsigned long long* out, float* sink, int reps, int mode, int prio_valu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float s = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (wv < 4) {
    if (mode & 1) {
      half8 a[3], b[2];
      for (int m = 0; m < 3; ++m) for (int e = 0; e < 8; ++e) a[m][e] = (_Float16)(0.01f * (lane + m + e));
      for (int n = 0; n < 2; ++n) for (int e = 0; e < 8; ++e) b[n][e] = (_Float16)(0.02f * (lane + n - e));
      float4v acc[3][2];
      for (int m = 0; m < 3; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = float4v{0, 0, 0, 0};
      for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int kk = 0; kk < 14; ++kk)
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
      }
      for (int m = 0; m < 3; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][2];
    }
  } else {
    if (mode & 2) {
      if (prio_valu) __builtin_amdgcn_s_setprio(3);
      float4v v[9];
      for (int q = 0; q < 9; ++q) v[q] = float4v{0.1f * lane + q, 0.2f * lane, 0.3f + q, 0.4f * q};
      const float4v al{1.01f, 0.99f, 1.02f, 0.98f}, be{0.1f, -0.1f, 0.05f, -0.05f};
      for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {                        // one epilogue-A quad: 2 cvt_pk, 4 cvt, 2 pk_fma, 2 cvt_pk, 2 pk_max
          float2v lo{v[q][0], v[q][1]}, hi{v[q][2], v[q][3]};
          lo = __builtin_convertvector(__builtin_convertvector(lo, half2v), float2v);
          hi = __builtin_convertvector(__builtin_convertvector(hi, half2v), float2v);
          lo = __builtin_elementwise_fma(lo, float2v{al[0], al[1]}, float2v{be[0], be[1]});
          hi = __builtin_elementwise_fma(hi, float2v{al[2], al[3]}, float2v{be[2], be[3]});
          const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
          if (mode & 4) *reinterpret_cast<half2v*>(smem + (wv * 64 + lane) * 96 + q * 8) = olo;
          const float2v flo = __builtin_convertvector(olo, float2v), fhi = __builtin_convertvector(ohi, float2v);
          v[q] = float4v{flo[0], flo[1], fhi[0], fhi[1]};
        }
      }
      for (int q = 0; q < 9; ++q) s += v[q][0] + v[q][3];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 256 * 8 * 8);
  hipMalloc(&d_sink, 256 * 512 * 4);
  const int reps = 2000;
  std::vector<unsigned long long> h(256 * 8);
  const char* names[] = {"", "MFMA waves only", "VALU waves only", "both", "", "", "VALU (+LDS writes) only", "both, VALU waves write LDS"};
  for (int prio = 0; prio <= 3; prio += 3)
    for (int mode : {1, 2, 3, 6, 7}) {
      for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 64 * 1024, 0, d_out, d_sink, reps, mode, prio);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      double tm = 0, tv = 0;
      for (int b = 0; b < 256; ++b) { tm += (double)h[b * 8]; tv += (double)h[b * 8 + 4]; }
      tm = tm / 256 * 10.0; tv = tv / 256 * 10.0;            // ns
      printf("prio(valu)=%d %-28s MFMA wave: %6.2f ns per MFMA | VALU wave: %6.2f ns per quad (%5.2f ns per instruction of ~14)\n", prio,
             names[mode], tm / (reps * 84.0), tv / (reps * 9.0), tv / (reps * 9.0 * 14));
    }
  return 0;
}
