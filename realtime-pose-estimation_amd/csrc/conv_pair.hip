// Tail of one Bottleneck + head of the next in ONE pass over the 256-channel tensor of layer1
// (pose_higher_hrnet.py:96-116 of the reference: conv3 (1x1, 64 -> 256) + bn3 + residual + ReLU of block i, then conv1
// (1x1, 256 -> 64) + bn1 + ReLU of block i+1, which reads what conv3 has just produced).
//
// Layer1's 160 x 160 x 256 tensor is 419 MB at batch 32.  As two launches it is written by conv3 (206 us: t 105 MB +
// residual 419 MB in, 419 MB out = 4.6 TB/s, the HBM roof of that kernel) and read back by conv1 (110 us).  Here a wave
// that has the 16 x 256 output rows of a pixel tile in its transposition slab - where the direct kernel
// (conv_direct.hip) only stores them - uses the slab as the B operand of the second GEMM: the [pixel][channel] rows ARE
// the MFMA B layout.  One 419 MB read per block boundary is gone (three per forward).
//
// Same math as the two launches: same k order, the 256-channel rows are rounded to fp16, residual-added, ReLU'd and
// stored exactly as before, and the second conv consumes those fp16 values: bit-identical (tests: the network with and
// without the pairing).  Both weight sets (2 x 32 KiB of A fragments) and the BN parameters live in LDS for the kernel's
// lifetime; 8 waves per workgroup, one persistent workgroup per CU; the kernel is HBM-bound (20 KiB per 64 MFMAs).
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kPWaves = 8;
constexpr int kC0 = 64, kC1 = 256, kC2 = 64;          // t -> y -> u channels
constexpr int kKS1 = kC0 / 32, kM1 = kC1 / 16;        // first GEMM: 2 k-steps, 16 row tiles
constexpr int kKS2 = kC1 / 32, kM2 = kC2 / 16;        // second GEMM: 8 k-steps, 4 row tiles
constexpr int kRowB = kC1 * 2 + 16;                   // slab bytes per pixel row (+16: spreads the 8-byte writes over banks)
constexpr int kCH = kC1 / 8;                          // 16-byte pieces per 256-channel row
constexpr int kNIT = 16 * kCH / 64;                   // pieces per lane and pixel tile: 8
constexpr int kOffW1 = 0;                             // conv3's fragments  [row tile][k-step][lane][8]
constexpr int kOffW2 = kM1 * kKS1 * 1024;             // conv1's fragments
constexpr int kOffBn1 = kOffW2 + kM2 * kKS2 * 1024;   // alpha3[256] beta3[256]
constexpr int kOffBn2 = kOffBn1 + 2 * kC1 * 4;        // alpha1[64] beta1[64]
constexpr int kOffSlab = kOffBn2 + 2 * kC2 * 4;
constexpr int kPairLds = kOffSlab + kPWaves * 16 * kRowB;

__device__ __forceinline__ half4 bn_round_p(const float4v v, const float4v al, const float4v be, bool round_conv) {
  if (round_conv) {
    const half2v h0 = __builtin_convertvector(float2v{v[0], v[1]}, half2v);
    const half2v h1 = __builtin_convertvector(float2v{v[2], v[3]}, half2v);
    float r0 = __builtin_fmaf((float)h0[0], al[0], be[0]);
    float r1 = __builtin_fmaf((float)h0[1], al[1], be[1]);
    float r2 = __builtin_fmaf((float)h1[0], al[2], be[2]);
    float r3 = __builtin_fmaf((float)h1[1], al[3], be[3]);
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));     // two roundings, not v_fma_mixlo (conv_stream.hip bn_round)
    const half2v o0 = __builtin_convertvector(float2v{r0, r1}, half2v), o1 = __builtin_convertvector(float2v{r2, r3}, half2v);
    return half4{o0[0], o0[1], o1[0], o1[1]};
  }
  float2v lo{v[0], v[1]}, hi{v[2], v[3]};
  lo = __builtin_elementwise_fma(lo, float2v{al[0], al[1]}, float2v{be[0], be[1]});
  hi = __builtin_elementwise_fma(hi, float2v{al[2], al[3]}, float2v{be[2], be[3]});
  const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
  return half4{olo[0], olo[1], ohi[0], ohi[1]};
}
}  // namespace

struct PairArgs {
  const _Float16* t;        // [P][t_ld]     input of the first conv (64 channels)
  const _Float16* res;      // [P][res_ld]   its residual (256 channels)
  _Float16* y;              // [P][y_ld]     its output = the second conv's input (256 channels)
  _Float16* u;              // [P][u_ld]     the second conv's output (64 channels)
  const _Float16 *w1, *w2;  // packed fragments (conv_pack_weights) of the two convs
  const float *al1, *be1, *al2, *be2;
  unsigned P;
  int t_ld, res_ld, y_ld, u_ld;
  int mt1, mt2;             // row tiles per packed cout block of the two plans
  int round_conv;
  unsigned t_bytes;         // bytes of the input view (buffer bounds)
};

__global__ void __launch_bounds__(kPWaves * 64) conv1x1_pair_kernel(const PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  // both weight sets and the BN parameters into LDS, once: fragment (row tile mg, k-step k) of a plan with `mt` row
  // tiles per cout block sits at ((cb * KS + k) * mt + mi) * 1 KiB of the packed weights, mg = cb * mt + mi
  for (int f = wv; f < kM1 * kKS1; f += kPWaves) {
    const int mg = f / kKS1, k = f - mg * kKS1;
    const int cb = mg / a.mt1, mi = mg - cb * a.mt1;
    *reinterpret_cast<u32x4*>(smem + kOffW1 + f * 1024 + lane * 16) =
        *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.w1) + ((size_t)((cb * kKS1 + k) * a.mt1 + mi) * 64 + lane) * 16);
  }
  for (int f = wv; f < kM2 * kKS2; f += kPWaves) {
    const int mg = f / kKS2, k = f - mg * kKS2;
    const int cb = mg / a.mt2, mi = mg - cb * a.mt2;
    *reinterpret_cast<u32x4*>(smem + kOffW2 + f * 1024 + lane * 16) =
        *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.w2) + ((size_t)((cb * kKS2 + k) * a.mt2 + mi) * 64 + lane) * 16);
  }
  for (int i = tid; i < 2 * kC1; i += kPWaves * 64)
    reinterpret_cast<float*>(smem + kOffBn1)[i] = i < kC1 ? a.al1[i] : a.be1[i - kC1];
  for (int i = tid; i < 2 * kC2; i += kPWaves * 64)
    reinterpret_cast<float*>(smem + kOffBn2)[i] = i < kC2 ? a.al2[i] : a.be2[i - kC2];
  __syncthreads();
  const char* const w1s = smem + kOffW1 + lane * 16;
  const char* const w2s = smem + kOffW2 + lane * 16;
  const float* const bn1 = reinterpret_cast<const float*>(smem + kOffBn1);
  const float* const bn2 = reinterpret_cast<const float*>(smem + kOffBn2);
  char* const slab = smem + kOffSlab + wv * 16 * kRowB;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.t), 0, (int)a.t_bytes, 0x00020000);
  const unsigned n_tiles = (a.P + 15u) / 16u;
  const unsigned stride = gridDim.x * kPWaves;
  unsigned t = blockIdx.x * kPWaves + wv;
  const int t_ld2 = a.t_ld * 2;
  const uint32_t bcol = (uint32_t)(r * t_ld2 + g * 16);
  // 16-byte pieces of this lane in a 16 x 256 tile: piece c = it * 64 + lane -> pixel c / 32, slot c % 32
  const int ppix0 = lane >> 5, pslot = lane & 31;         // piece it: pixel 2 * it + ppix0, the same slot
  auto load_b = [&](unsigned tile, u32x4 (&b)[kKS1]) __attribute__((always_inline)) {
    const uint32_t base = tile * 16u * (uint32_t)t_ld2 + bcol;      // beyond the tensor: the bounds check returns zeros
#pragma unroll
    for (int k = 0; k < kKS1; ++k)
      b[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(base + k * 64), 0, 0));
  };
  u32x4 bcur[kKS1], bnxt[kKS1];
  if (t < n_tiles) load_b(t, bcur);
  for (; t < n_tiles; t += stride) {
    // residual rows of this tile and the next tile's B fragments: in flight during the first GEMM
    u32x4 rr[kNIT];
#pragma unroll
    for (int it = 0; it < kNIT; ++it) {
      const unsigned p = t * 16u + (unsigned)(2 * it + ppix0);
      rr[it] = u32x4{0u, 0u, 0u, 0u};
      if (p < a.P) rr[it] = *reinterpret_cast<const u32x4*>(a.res + (size_t)p * a.res_ld + pslot * 8);
    }
    const unsigned tn = t + stride;
    if (tn < n_tiles) load_b(tn, bnxt);
    // ---- GEMM 1: y[256][16 pixels] = W3 . t ----
    float4v acc[kM1];
#pragma unroll
    for (int m = 0; m < kM1; ++m) acc[m] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kKS1; ++k)
#pragma unroll
      for (int m = 0; m < kM1; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(w1s + (m * kKS1 + k) * 1024),
                                                        __builtin_bit_cast(half8, bcur[k]), acc[m], 0, 0, 0);
    // BN with the wrapper's rounding points -> fp16 rows of the wave's slab
#pragma unroll
    for (int m = 0; m < kM1; ++m) {
      const float4v al = *reinterpret_cast<const float4v*>(bn1 + m * 16 + g * 4);
      const float4v be = *reinterpret_cast<const float4v*>(bn1 + kC1 + m * 16 + g * 4);
      *reinterpret_cast<half4*>(slab + r * kRowB + m * 32 + g * 8) = bn_round_p(acc[m], al, be, a.round_conv != 0);
    }
    // whole 16-byte row pieces: + residual (fp16 add, round-to-nearest-even = the wrapper's add), ReLU, store - and
    // back into the slab: these fp16 values are the second conv's input
#pragma unroll
    for (int it = 0; it < kNIT; ++it) {
      const int pix = 2 * it + ppix0;
      const unsigned p = t * 16u + (unsigned)pix;
      half8 v = *reinterpret_cast<const half8*>(slab + pix * kRowB + pslot * 16);
      v = v + __builtin_bit_cast(half8, rr[it]);
      short8 bsh = __builtin_bit_cast(short8, v);
      bsh = bsh & ~(bsh >> 15);                             // x > 0 ? x : +0 on the sign bits
      v = __builtin_bit_cast(half8, bsh);
      *reinterpret_cast<half8*>(slab + pix * kRowB + pslot * 16) = v;
      if (p < a.P) store16_wt(a.y + (size_t)p * a.y_ld + pslot * 8, v);
    }
    // ---- GEMM 2: u[64][16 pixels] = W1' . y, y straight from the slab (lane (r, g): pixel r, channels 32 k + 8 g ..) ----
    float4v acc2[kM2];
#pragma unroll
    for (int m = 0; m < kM2; ++m) acc2[m] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kKS2; ++k) {
      const half8 b2 = *reinterpret_cast<const half8*>(slab + r * kRowB + k * 64 + g * 16);
#pragma unroll
      for (int m = 0; m < kM2; ++m)
        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(w2s + (m * kKS2 + k) * 1024), b2,
                                                         acc2[m], 0, 0, 0);
    }
    // BN + ReLU -> 8 bytes per lane and row tile: lane (r, g) holds channels 16 m + 4 g .. + 3 of pixel r
    const unsigned pu = t * 16u + (unsigned)r;
#pragma unroll
    for (int m = 0; m < kM2; ++m) {
      const float4v al = *reinterpret_cast<const float4v*>(bn2 + m * 16 + g * 4);
      const float4v be = *reinterpret_cast<const float4v*>(bn2 + kC2 + m * 16 + g * 4);
      const half4 o = bn_round_p(acc2[m], al, be, a.round_conv != 0);
      short4v osh = __builtin_bit_cast(short4v, o);
      osh = osh & ~(osh >> 15);
      if (pu < a.P) *reinterpret_cast<short4v*>(a.u + (size_t)pu * a.u_ld + m * 16 + g * 4) = osh;
    }
#pragma unroll
    for (int k = 0; k < kKS1; ++k) bcur[k] = bnxt[k];
  }
}

bool conv_pair_supports(int cin1, int cout1, int cout2) { return cin1 == kC0 && cout1 == kC1 && cout2 == kC2; }

int conv_pair_launch(const ConvPlan& p1, const ConvArgs& c1, const ConvPlan& p2, const ConvArgs& c2, hipStream_t s) {
  RTPE_REQUIRE(p1.esize == 2 && p2.esize == 2 && p1.tapw == 1 && p2.tapw == 1 && p1.in_mul == 1 && p2.in_mul == 1 &&
                   conv_pair_supports(c1.cin, c1.cout, c2.cout) && c2.cin == kC1,
               "1x1 pair: unsupported layers (%d -> %d -> %d)", c1.cin, c1.cout, c2.cout);
  RTPE_REQUIRE(p1.cc * p1.n_cchunks == kC0 && p1.kc * 32 == p1.cc && p2.cc * p2.n_cchunks == kC1 && p2.kc * 32 == p2.cc,
               "1x1 pair: the packed k-steps must be the 32-channel steps in order");
  RTPE_REQUIRE(c1.y != nullptr && c2.y != nullptr && c1.res != nullptr && c2.res == nullptr && c1.y_nchw == nullptr &&
                   c2.y_nchw == nullptr && c2.x == c1.y && c2.in_ld == c1.out_ld && c1.relu && c2.relu &&
                   c1.round_conv == c2.round_conv,
               "1x1 pair: conv + residual + ReLU into the input of a conv + ReLU expected");
  RTPE_REQUIRE(c1.in_ld % 8 == 0 && c1.out_ld % 8 == 0 && c1.res_ld % 8 == 0 && c2.out_ld % 4 == 0, "1x1 pair: row alignment");
  RTPE_REQUIRE(c1.x_bytes > 0 && c1.x_bytes < 0x80000000ull, "1x1 pair: input view of %zu bytes", (size_t)c1.x_bytes);
  PairArgs a;
  memset(&a, 0, sizeof(a));
  a.t = c1.x; a.res = c1.res; a.y = c1.y; a.u = c2.y;
  a.w1 = c1.w; a.w2 = c2.w; a.al1 = c1.alpha; a.be1 = c1.beta; a.al2 = c2.alpha; a.be2 = c2.beta;
  a.P = (unsigned)c1.N * (unsigned)c1.H_in * (unsigned)c1.W_in;
  a.t_ld = c1.in_ld; a.res_ld = c1.res_ld; a.y_ld = c1.out_ld; a.u_ld = c2.out_ld;
  a.mt1 = p1.mt; a.mt2 = p2.mt; a.round_conv = c1.round_conv; a.t_bytes = (unsigned)c1.x_bytes;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask))
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_pair_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned n_tiles = (a.P + 15u) / 16u;
  unsigned grid = (n_tiles + kPWaves - 1) / kPWaves;
  if (grid > 256u) grid = 256u;                                     // one persistent workgroup per CU: the waves loop
  hipLaunchKernelGGL(conv1x1_pair_kernel, dim3(grid), dim3(kPWaves * 64), kPairLds, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
