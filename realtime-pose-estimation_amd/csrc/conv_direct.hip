// 1x1 convolution without a staged input tile ("direct"): Conv2d(k=1, s=1) + BN (+ residual) (+ ReLU) on NHWC fp16
// tensors - the Bottleneck convs of layer1 (pose_higher_hrnet.py:78-116 of the reference: 64 -> 64, 64 -> 256 + residual,
// 256 -> 64, the 64 -> 256 projection :587-592) and the 1x1 convs of the fuse layers (:201-209).  Same math, same k order
// (channels ascending in 32-wide MFMA k-steps), same rounding points and the same packed weights as conv_mfma.hip:
// bit-identical results.
//
// A 1x1 conv does not see the image: the tensor is a [pixels][Cin] matrix whose rows ARE the MFMA B-operand layout -
// lane (pixel r, k group g) needs 8 consecutive channels of pixel r, 16 bytes straight from global memory.  So:
//   * no LDS staging, no workgroup barrier: a wave owns 16-pixel tiles (pixels are linear in NHWC memory) x a block of
//     16 * MB output channels, loads its B fragments with one buffer_load_dwordx4 per lane and k-step (two k-steps of
//     16 consecutive 128-byte rows cover a contiguous 2 KiB), and keeps the weights (A fragments) of its cout block in
//     REGISTERS for the whole kernel (KS k-steps x MB row tiles x 4 VGPRs <= 144);
//   * the one-workgroup-per-tile kernel ran these layers at 0.2-0.5 of the HBM roof: a workgroup there is 16-32 MFMAs
//     per wave behind a tile staging, two barriers, a tap table and a weight fetch per workgroup (25,600 workgroups
//     for a layer1 conv); here a wave streams: B fragments and residual pieces of tile t+1 are requested before the
//     MFMAs of tile t, and the BN / transposition / store epilogue of a tile runs beside the other waves' loads;
//   * epilogue as in the other kernels: BN with the wrapper's rounding points, transposed through a wave-private LDS
//     slab so that stores and residual loads are whole 16-byte row pieces.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kDWaves = 4;

// conv accumulator -> fp16 (the conv's output tensor) -> BN in fp32 -> fp16, the wrapper's rounding points (see
// conv_stream.hip bn_round: v_fma_mix_f32 on the fp16 value, the final conversion kept apart from the fma)
__device__ __forceinline__ half4 bn_round_d(const float4v v, const float4v al, const float4v be) {
  const half2v h0 = __builtin_convertvector(float2v{v[0], v[1]}, half2v);
  const half2v h1 = __builtin_convertvector(float2v{v[2], v[3]}, half2v);
  float r0 = __builtin_fmaf((float)h0[0], al[0], be[0]);
  float r1 = __builtin_fmaf((float)h0[1], al[1], be[1]);
  float r2 = __builtin_fmaf((float)h1[0], al[2], be[2]);
  float r3 = __builtin_fmaf((float)h1[1], al[3], be[3]);
  asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
  const half2v o0 = __builtin_convertvector(float2v{r0, r1}, half2v), o1 = __builtin_convertvector(float2v{r2, r3}, half2v);
  return half4{o0[0], o0[1], o1[0], o1[1]};
}
}  // namespace

struct DirectArgs {
  const _Float16* x;       // [P][in_ld]
  const _Float16* w;       // packed fragments (conv_pack_weights): [cout block of 16 mt][k-step][mt][lane][8]
  const float* alpha;      // [cout_pad]
  const float* beta;
  const _Float16* res;     // [P][res_ld] or nullptr
  _Float16* y;             // [P][out_ld]
  unsigned P;              // pixels
  int in_ld, out_ld, res_ld;
  int mt;                  // row tiles per packed cout block
  int n_k;                 // k-steps (= KS)
  int cout_store;          // channels written
  int relu, round_conv;
  unsigned x_bytes;        // bytes of the input view (buffer bounds)
};

template <int KS, int MB>
__global__ void __launch_bounds__(kDWaves * 64) conv1x1_direct_kernel(const DirectArgs a) {
  constexpr int ROWB = MB * 32 + 16;                     // slab bytes per pixel row
  constexpr int CH = MB * 2;                             // 16-byte pieces per pixel row
  constexpr int NIT = (16 * CH + 63) / 64;               // pieces per lane
  __shared__ __attribute__((aligned(16))) char smem[kDWaves * 16 * ROWB + 2 * MB * 16 * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int cblk = blockIdx.y * MB * 16;                 // first output channel of this workgroup
  float* const bn = reinterpret_cast<float*>(smem + kDWaves * 16 * ROWB);
  for (int i = tid; i < 2 * MB * 16; i += kDWaves * 64)
    bn[i] = i < MB * 16 ? a.alpha[cblk + i] : a.beta[cblk + i - MB * 16];
  // the weights of the cout block: this lane's 16 bytes of every (row tile, k-step), kept for the whole kernel
  half8 wf[MB][KS];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int mg = cblk / 16 + m;                        // global row tile
    const int cb = mg / a.mt, mi = mg - cb * a.mt;
#pragma unroll
    for (int k = 0; k < KS; ++k)
      wf[m][k] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(a.w) +
                                                 ((size_t)((cb * KS + k) * a.mt + mi) * 64 + lane) * 16);
  }
  __syncthreads();
  char* const slab = smem + wv * 16 * ROWB;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const unsigned n_tiles = (a.P + 15u) / 16u;
  const unsigned stride = gridDim.x * kDWaves;
  unsigned t = blockIdx.x * kDWaves + wv;
  const int in_ld2 = a.in_ld * 2;
  const uint32_t bcol = (uint32_t)(r * in_ld2 + g * 16);  // this lane's pixel row and k group inside a tile

  // row pieces of this lane in a tile: piece c = it * 64 + lane -> pixel c / CH, 16-byte slot c % CH
  int ppix[NIT], pslot[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    ppix[it] = c / CH;
    pslot[it] = c - ppix[it] * CH;
    if (c >= 16 * CH || cblk + pslot[it] * 8 >= a.cout_store) ppix[it] = 1 << 20;   // no such piece
  }
  auto load_b = [&](unsigned tile, u32x4 (&b)[KS]) __attribute__((always_inline)) {
    // (a tile beyond the tensor: offsets beyond the buffer return zeros)
    const uint32_t base = tile * 16u * (uint32_t)in_ld2 + bcol;
#pragma unroll
    for (int k = 0; k < KS; ++k)
      b[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(base + k * 64), 0, 0));
  };
  auto load_res = [&](unsigned tile, u32x4 (&rr)[NIT]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned p = tile * 16u + (unsigned)ppix[it];
      rr[it] = u32x4{0u, 0u, 0u, 0u};
      if (ppix[it] < 16 && p < a.P)
        rr[it] = *reinterpret_cast<const u32x4*>(a.res + (size_t)p * a.res_ld + cblk + pslot[it] * 8);
    }
  };
  u32x4 bcur[KS], bnxt[KS], rcur[NIT];
  if (t < n_tiles) load_b(t, bcur);
  for (; t < n_tiles; t += stride) {
    if (a.res != nullptr) load_res(t, rcur);
    const unsigned tn = t + stride;
    if (tn < n_tiles) load_b(tn, bnxt);
    float4v acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[m][k], __builtin_bit_cast(half8, bcur[k]), acc[m], 0, 0, 0);
    // epilogue: BN with the wrapper's rounding points -> fp16 rows of the wave's slab
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float4v al = *reinterpret_cast<const float4v*>(bn + m * 16 + g * 4);
      const float4v be = *reinterpret_cast<const float4v*>(bn + MB * 16 + m * 16 + g * 4);
      half4 o;
      if (a.round_conv) {
        o = bn_round_d(acc[m], al, be);
      } else {
        float2v lo{acc[m][0], acc[m][1]}, hi{acc[m][2], acc[m][3]};
        lo = __builtin_elementwise_fma(lo, float2v{al[0], al[1]}, float2v{be[0], be[1]});
        hi = __builtin_elementwise_fma(hi, float2v{al[2], al[3]}, float2v{be[2], be[3]});
        const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
        o = half4{olo[0], olo[1], ohi[0], ohi[1]};
      }
      *reinterpret_cast<half4*>(slab + r * ROWB + m * 32 + g * 8) = o;
    }
    // whole 16-byte row pieces: + residual (fp16 add, round-to-nearest-even = the wrapper's add), ReLU, store
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned p = t * 16u + (unsigned)ppix[it];
      if (ppix[it] < 16 && p < a.P) {
        half8 v = *reinterpret_cast<const half8*>(slab + ppix[it] * ROWB + pslot[it] * 16);
        if (a.res != nullptr) v = v + __builtin_bit_cast(half8, rcur[it]);
        if (a.relu) {
          short8 bsh = __builtin_bit_cast(short8, v);
          bsh = bsh & ~(bsh >> 15);
          v = __builtin_bit_cast(half8, bsh);
        }
        store16_wt(a.y + (size_t)p * a.out_ld + cblk + pslot[it] * 8, v);
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) bcur[k] = bnxt[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The two heads (final_layers, pose_higher_hrnet.py:447-483, :674, :683: Conv2d(48 -> 34 / 17, k = 1, bias), fp32 NCHW out,
// head 0 also NHWC into the buffer the transposed conv reads): the same direct scheme for Cin = 48 - two k-steps, the
// second half-padded (its lanes of channels 48..63 read zeros: out-of-range offsets) - with an NCHW epilogue.  A wave takes 32 consecutive pixels (two MFMA column tiles) per step, so that a
// channel's piece of the step is one whole 128-byte line of its fp32 plane; the finished values go channel-major through
// the wave's slab (two pixels per 4-byte write) and come back as one 8-byte read per lane = four pixels of one channel.
// Round 5, verdict item 3a: on the one-workgroup-per-tile kernel these layers took 152 + 85 us for 380 + 177 MB (staging,
// k loop and epilogue of a workgroup are serial); here nothing is staged and nothing waits for a barrier.
struct HeadArgs {
  const _Float16* x;
  const _Float16* w;       // packed fragments of the plan (one cout block of mt row tiles, 2 k-steps)
  const float* alpha;
  const float* beta;
  _Float16* y;             // NHWC output or nullptr
  float* y_nchw;           // fp32 (N, nchw_channels, H, W)
  unsigned P, HW;          // pixels in all, per image (HW % 32 == 0)
  int in_ld, out_ld, cout_store, nchw_channels, relu, round_conv;
  unsigned x_bytes;
};

template <int MB>
__global__ void __launch_bounds__(kDWaves * 64) conv1x1_head_kernel(const HeadArgs a) {
  constexpr int ROWB = MB * 32 + 16;                     // pixel-major slab: bytes per pixel row (NHWC pieces)
  constexpr int CP = 32 * 2 + 8;                         // channel-major slab: bytes per channel row of 32 pixels
  constexpr int SLAB = 32 * ROWB > MB * 16 * CP ? 32 * ROWB : MB * 16 * CP;
  constexpr int CH = MB * 2;
  constexpr int NITR = (32 * CH + 63) / 64;              // NHWC row pieces per lane and step
  __shared__ __attribute__((aligned(16))) char smem[kDWaves * SLAB + 2 * MB * 16 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  float* const bn = reinterpret_cast<float*>(smem + kDWaves * SLAB);
  for (int i = tid; i < 2 * MB * 16; i += kDWaves * 64) bn[i] = i < MB * 16 ? a.alpha[i] : a.beta[i - MB * 16];
  half8 wf[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      wf[m][k] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(a.w) + ((size_t)(k * MB + m) * 64 + lane) * 16);
  __syncthreads();
  char* const slab = smem + wv * SLAB;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const unsigned n_steps = (a.P + 31u) / 32u;
  const unsigned stride = gridDim.x * kDWaves;
  unsigned t = blockIdx.x * kDWaves + wv;
  const int in_ld2 = a.in_ld * 2;
  const uint32_t bcol = (uint32_t)(r * in_ld2 + g * 16);
  auto load_b = [&](unsigned step, u32x4 (&b)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const uint32_t base = (step * 32u + nt * 16u) * (uint32_t)in_ld2 + bcol;
      // k-step 1: channels 32..63, of which 48..63 do not exist.  Those lanes ask for an out-of-range offset (zeros): the bytes
      // behind a pixel's 48 channels are whatever lies there - the next pixel, or, in the 96-channel buffer head 0 reads, the
      // very channels head 0 writes, or stale workspace bytes that may be NaN patterns (NaN x zero weight = NaN)
      b[nt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)base, 0, 0));
      b[nt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, g < 2 ? (int)(base + 64) : (int)0x80000000, 0, 0));
    }
  };
  u32x4 bcur[2][2], bnxt[2][2];
  if (t < n_steps) load_b(t, bcur);
  for (; t < n_steps; t += stride) {
    const unsigned tn = t + stride;
    if (tn < n_steps) load_b(tn, bnxt);
    float4v acc[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[m][k], __builtin_bit_cast(half8, bcur[nt][k]), acc[m][nt], 0, 0, 0);
    // bias / BN with the wrapper's rounding points
    half4 o[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float4v al = *reinterpret_cast<const float4v*>(bn + m * 16 + g * 4);
      const float4v be = *reinterpret_cast<const float4v*>(bn + MB * 16 + m * 16 + g * 4);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (a.round_conv) {
          o[m][nt] = bn_round_d(acc[m][nt], al, be);
        } else {
          float2v lo{acc[m][nt][0], acc[m][nt][1]}, hi{acc[m][nt][2], acc[m][nt][3]};
          lo = __builtin_elementwise_fma(lo, float2v{al[0], al[1]}, float2v{be[0], be[1]});
          hi = __builtin_elementwise_fma(hi, float2v{al[2], al[3]}, float2v{be[2], be[3]});
          const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
          o[m][nt] = half4{olo[0], olo[1], ohi[0], ohi[1]};
        }
      }
    }
    if (a.y != nullptr) {
      // NHWC: pixel-major rows of the slab, whole 16-byte pieces (as conv1x1_direct_kernel)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<half4*>(slab + (nt * 16 + r) * ROWB + m * 32 + g * 8) = o[m][nt];
#pragma unroll
      for (int it = 0; it < NITR; ++it) {
        const int c = it * 64 + lane;
        const int pw = c / CH, slot = c - pw * CH;
        const unsigned p = t * 32u + (unsigned)pw;
        if (c < 32 * CH && slot * 8 < a.cout_store && p < a.P) {
          half8 v = *reinterpret_cast<const half8*>(slab + pw * ROWB + slot * 16);
          if (a.relu) {
            short8 bsh = __builtin_bit_cast(short8, v);
            bsh = bsh & ~(bsh >> 15);
            v = __builtin_bit_cast(half8, bsh);
          }
          store16_wt(a.y + (size_t)p * a.out_ld + slot * 8, v);
        }
      }
    }
    // NCHW fp32: channel-major slab (the wave's own: the pixel-major rows above are read, in program order, before)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<_Float16*>(slab + (m * 16 + g * 4 + j) * CP + (nt * 16 + r) * 2) = o[m][nt][j];
    {
      const unsigned p0 = t * 32u;                       // the step's 32 pixels lie in one image (HW % 32 == 0)
      const unsigned n = p0 / a.HW, pin = p0 - n * a.HW;
      float* const yb = a.y_nchw + (size_t)n * a.nchw_channels * a.HW + pin;
      const int n_pieces = a.nchw_channels * 8;
      for (int i = lane; i < n_pieces; i += 64) {
        const int c = i >> 3, q = i & 7;
        if (p0 + (unsigned)q * 4u < a.P) {
          const half4 h = *reinterpret_cast<const half4*>(slab + c * CP + q * 8);
          float4v out;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = (float)h[e];
            out[e] = a.relu ? (x > 0.f ? x : 0.f) : x;
          }
          *reinterpret_cast<float4v*>(yb + (size_t)c * a.HW + q * 4) = out;
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k = 0; k < 2; ++k) bcur[nt][k] = bnxt[nt][k];
  }
}

// heads this kernel takes: 1x1, 48 input channels (one chunk of 48, two k-steps), at most 48 padded output channels in one
// cout block, fp32 NCHW output, whole 32-pixel steps inside an image
bool conv_head_supports(const ConvPlan& p, const ConvArgs& c) {
  return p.esize == 2 && p.tapw == 1 && p.in_mul == 1 && p.dil == 1 && p.cc == 48 && p.kc == 2 && p.n_cchunks == 1 && p.n_cb == 1 &&
         (p.mt == 2 || p.mt == 3) && c.cin == 48 && c.y_nchw != nullptr && c.nchw_f32 && c.res == nullptr && c.o_mul == 1 &&
         c.n_cls == 0 && ((unsigned)c.H_in * (unsigned)c.W_in) % 32u == 0 && c.x_bytes > 0 && c.x_bytes < 0x80000000ull &&
         c.in_ld % 8 == 0 && (c.y == nullptr || c.out_ld % 8 == 0) && c.nchw_channels <= p.mt * 16 &&
         c.in_cs == p.cc && ((uintptr_t)c.y_nchw & 15) == 0;
}

template <int MB>
static int launch_head(const HeadArgs& a, hipStream_t s) {
  const unsigned n_steps = (a.P + 31u) / 32u;
  unsigned gx = (n_steps + kDWaves - 1) / kDWaves;
  if (gx > 256u * 8u) gx = 256u * 8u;                    // ~8 workgroups per CU: the waves loop
  hipLaunchKernelGGL((conv1x1_head_kernel<MB>), dim3(gx), dim3(kDWaves * 64), 0, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int conv_head_launch(const ConvPlan& p, const ConvArgs& c, hipStream_t s) {
  RTPE_REQUIRE(conv_head_supports(p, c), "head conv: unsupported layer (cin %d cout %d)", c.cin, c.cout);
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.x = c.x; a.w = c.w; a.alpha = c.alpha; a.beta = c.beta; a.y = c.y; a.y_nchw = reinterpret_cast<float*>(c.y_nchw);
  a.P = (unsigned)c.N * (unsigned)c.H_in * (unsigned)c.W_in;
  a.HW = (unsigned)c.H_in * (unsigned)c.W_in;
  a.in_ld = c.in_ld; a.out_ld = c.out_ld; a.cout_store = c.cout_store; a.nchw_channels = c.nchw_channels;
  a.relu = c.relu; a.round_conv = c.round_conv; a.x_bytes = (unsigned)c.x_bytes;
  return p.mt == 2 ? launch_head<2>(a, s) : launch_head<3>(a, s);
}

template <int KS, int MB>
static int launch_direct(const DirectArgs& a, int n_blocks_y, hipStream_t s) {
  const unsigned n_tiles = (a.P + 15u) / 16u;
  unsigned gx = (n_tiles + kDWaves - 1) / kDWaves;
  const unsigned cap = 256u * 8u / (unsigned)n_blocks_y;          // ~8 workgroups per CU in all: the waves loop
  if (gx > cap) gx = cap < 1 ? 1 : cap;
  hipLaunchKernelGGL((conv1x1_direct_kernel<KS, MB>), dim3(gx, (unsigned)n_blocks_y), dim3(kDWaves * 64), 0, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// cout tiles one wave carries for a layer (0: the layer is not one this kernel takes)
int conv_direct_mb(const ConvPlan& p) {
  if (p.esize != 2 || p.tapw != 1 || p.in_mul != 1 || p.dil != 1 || p.cc % 32 != 0 || p.kc * 32 != p.cc) return 0;
  const int ks = p.n_cchunks * p.kc, tiles = p.cout_pad / 16;       // the packed k-steps are the 32-channel steps, in order
  static const int cand[] = {8, 6, 4, 3, 2, 1};
  for (int mb : cand) {
    if (mb % p.mt != 0 || tiles % mb != 0 || ks * mb * 4 > 144) continue;
    if ((ks == 2 && (mb == 8 || mb == 4)) || (ks == 8 && mb == 4) || (ks == 3 && mb == 3) || (ks == 6 && mb == 3) ||
        (ks == 12 && mb == 3))                                      // instantiated shapes
      return mb;
  }
  return 0;
}

int conv_direct_launch(const ConvPlan& p, const ConvArgs& c, hipStream_t s) {
  const int mb = conv_direct_mb(p);
  RTPE_REQUIRE(mb > 0 && c.cin == p.cc * p.n_cchunks, "direct 1x1 conv: unsupported layer (cin %d cout %d)", c.cin, c.cout);
  RTPE_REQUIRE(c.y != nullptr && c.y_nchw == nullptr && c.o_mul == 1 && c.n_cls == 0, "direct 1x1 conv: NHWC output only");
  RTPE_REQUIRE(c.in_cs == p.cc && c.out_cs == p.mt * 16 && (c.res == nullptr || c.res_cs == p.mt * 16), "direct 1x1 conv: NHWC tensors only");
  RTPE_REQUIRE(c.in_ld % 8 == 0 && c.out_ld % 8 == 0 && (c.res == nullptr || c.res_ld % 8 == 0), "direct 1x1 conv: row alignment");
  RTPE_REQUIRE(c.x_bytes > 0 && c.x_bytes < 0x80000000ull, "direct 1x1 conv: input view of %zu bytes", (size_t)c.x_bytes);
  DirectArgs a;
  memset(&a, 0, sizeof(a));
  a.x = c.x; a.w = c.w; a.alpha = c.alpha; a.beta = c.beta; a.res = c.res; a.y = c.y;
  a.P = (unsigned)c.N * (unsigned)c.H_in * (unsigned)c.W_in;
  a.in_ld = c.in_ld; a.out_ld = c.out_ld; a.res_ld = c.res_ld;
  a.mt = p.mt; a.n_k = c.cin / 32; a.cout_store = c.cout_store; a.relu = c.relu; a.round_conv = c.round_conv;
  a.x_bytes = (unsigned)c.x_bytes;
  const int ny = (p.cout_pad / 16) / mb;
  const int ks = a.n_k;
#define RTPE_D(KSv, MBv) if (ks == KSv && mb == MBv) return launch_direct<KSv, MBv>(a, ny, s);
  RTPE_D(2, 8) RTPE_D(2, 4) RTPE_D(8, 4) RTPE_D(3, 3) RTPE_D(6, 3) RTPE_D(12, 3)
#undef RTPE_D
  set_error("direct 1x1 conv: no kernel variant ks=%d mb=%d", ks, mb);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
