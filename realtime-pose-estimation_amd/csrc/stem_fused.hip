// The teacher's stem as ONE kernel: conv1 (3 -> 64, 3x3 stride 2) + bn1 + relu and conv2 (64 -> 64, 3x3 stride 2) + bn2 +
// relu, pose_higher_hrnet.py:363-368 / :638-643, half wrapper.  The 64-channel half-resolution map between them - 419 MB
// at batch 32 and 640 x 640, written by one kernel and read back by the next: 838 MB of the forward's HBM traffic - stays
// in LDS: a workgroup computes the (17 x 33)-pixel conv1 region that an 8 x 16 tile of conv2 outputs needs (9.6 % more
// than its share), rounds it as the separate kernels do, and multiplies it from there.
//
// Same bits as elementwise.hip's stem_kernel followed by conv_mfma.hip (tests/test_gpu_parity.py):
// * conv1 is the same chain of 27 fp32 multiply-adds per output in (ky, kx, c) order - but on the matrix pipe:
//   v_mfma_f32_16x16x4_f32 is bitwise a chain of four fmaf steps in k order on top of C (MI355X guide, matrix cores),
//   so seven of them (the 28th weight is zero) are the chain for 16 pixels x 16 channels.  The matrix pipe runs fp32 at
//   the rate of the packed vector FMAs (128 multiply-adds per clock and CU), and the vector ALUs stay free for staging
//   and epilogues (on the VALU the chain alone is 72 us at batch 32, which nothing in a fused kernel can hide);
// * conv2 uses the packed weight fragments of the conv op's own plan (mt = 4, one 64-channel chunk, 18 k steps) in the
//   same k order on v_mfma_f32_16x16x32_f16, and the same rounding points (conv output, BatchNorm output).
// The kernel can also stop after conv1 (y1 set: the region's own 16 x 32 pixels go to memory as NHWC rows) - same bits
// again (option "fused_stem" = 2; the executor otherwise runs a stem op that stands alone on the VALU kernel).
//
// Persistent workgroups of 8 waves, one per CU.  Per tile: the input patch of the NEXT tile is requested into registers
// before conv1 of this one starts; conv1 writes the region into LDS with the even columns of a row first (the B-operand
// reads of 16 consecutive conv2 outputs are then consecutive LDS pixels: conflict-free, as for the stride-2 tiles of
// conv_mfma.hip); conv2: wave w multiplies cout tile w % 4 - its 18 weight fragments live in registers for the whole
// kernel - with 4 rows of 16 pixels.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

namespace {
constexpr int kT2H = 8, kT2W = 16;                         // conv2 outputs per tile
constexpr int kC1H = 2 * kT2H + 1, kC1W = 2 * kT2W + 1;    // conv1 outputs they read: 17 x 33
constexpr int kPH = 2 * kC1H + 1, kPW = 2 * kC1W + 1;      // input pixels those read: 35 x 67
constexpr int kPWp = kPW + 1;                              // padded patch row (floats)
constexpr int kPS = 160;                                   // LDS bytes per conv1 pixel: 64 fp16 + 32 (pstride % 64 == 32)
constexpr int kRowB = kC1W * kPS;                          // 5280
constexpr int kNEven = (kC1W + 1) / 2;                     // 17 even columns, then the 16 odd ones
constexpr int kObufRow = 144;                              // epilogue transpose: 64 fp16 + 16 per pixel
constexpr int kPatchBytes = 3 * kPH * kPWp * 4;            // 28,560
constexpr int kAbBytes = 2 * 64 * 4;                       // bn1 alpha, beta
constexpr int kC1Bytes = kC1H * kRowB;                     // 89,760
constexpr int kObufBytes = kT2H * kT2W * kObufRow;         // 18,432
constexpr int kLds = kPatchBytes + kAbBytes + kC1Bytes + kObufBytes;   // 137,264
constexpr int kThreads = 512;
constexpr int kPatch = 3 * kPH * kPW;                      // 7,035 input values per tile
constexpr int kIter = (kPatch + kThreads - 1) / kThreads;  // 14 per thread
constexpr int kC1Px = kC1H * kC1W;                         // 561
constexpr int kC1Tiles = (kC1Px + 15) / 16;                // 36 pixel tiles of 16
constexpr int kTilesPerWave = (kC1Tiles + 7) / 8;          // 5 (waves 0-3) / 4
static_assert(kPatchBytes % 16 == 0 && kC1Bytes % 16 == 0, "LDS sections are 16-byte aligned");

__device__ __forceinline__ float round16(float v) { return (float)(_Float16)v; }
}  // namespace

__global__ void __launch_bounds__(kThreads) stem_fused_kernel(const StemFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* patch = reinterpret_cast<float*>(smem);                               // [3][35][68] fp16 values as fp32
  float* ab1 = reinterpret_cast<float*>(smem + kPatchBytes);                   // alpha1[64], beta1[64]
  char* c1 = smem + kPatchBytes + kAbBytes;                                    // [17][33 (even | odd)][160 B]
  char* obuf = c1 + kC1Bytes;                                                  // [128][144 B]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const bool fused = a.y1 == nullptr;
  const int Ho1 = a.H >> 1, Wo1 = a.W >> 1, Ho2 = a.H >> 2, Wo2 = a.W >> 2;
  const int tiles_x = (Wo2 + kT2W - 1) / kT2W, tiles_y = (Ho2 + kT2H - 1) / kT2H;
  const int total = a.N * tiles_x * tiles_y;
  // an XCD takes a contiguous eighth of the row-major tile list (neighbouring tiles share patch rows and columns in its L2)
  const int per_xcd = (total + 7) >> 3, wg_per_xcd = (int)(gridDim.x >> 3);
  const int xcd = (int)(blockIdx.x & 7u);
  const int t_end = (xcd + 1) * per_xcd < total ? (xcd + 1) * per_xcd : total;
  int t = xcd * per_xcd + (int)(blockIdx.x >> 3);

  // ---- once per workgroup: operands that stay in registers ----
  // conv1 weights as A operands of v_mfma_f32_16x16x4_f32: lane (row r, k slot g) of k group kg and cout tile mm holds
  // w1[4 kg + g][16 mm + r]; the 28th k value is a zero weight; this lane's patch offset of step 4 kg + g
  float w1r[7][4];
  int koff[7];
#pragma unroll
  for (int kg = 0; kg < 7; ++kg) {
    const int k = 4 * kg + g;
    const int kc = k < 27 ? k : 0;
    const int tap = kc / 3, c = kc - tap * 3, ky = tap / 3, kx = tap - ky * 3;
    koff[kg] = (c * kPH + ky) * kPWp + kx;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) w1r[kg][mm] = k < 27 ? (float)a.w1[k * 64 + mm * 16 + r] : 0.f;
  }
  // conv2: this wave's cout tile and half of the tile's rows; its 18 weight fragments
  const int m = wv & 3, hf = wv >> 2;
  uint4 a_res[18];
  if (fused) {
    const uint4* wfrag = reinterpret_cast<const uint4*>(a.w2) + m * 64 + lane;
#pragma unroll
    for (int k = 0; k < 18; ++k) a_res[k] = wfrag[(size_t)k * 4 * 64];
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) a_res[k] = uint4{0u, 0u, 0u, 0u};
  }
  if (tid < 128) ab1[tid] = tid < 64 ? a.alpha1[tid] : a.beta1[tid - 64];
  // this lane's pixel in each of the wave's conv1 pixel tiles (pixel tile wv + 8 i, pixel p = 16 * tile + r)
  int pixoff[kTilesPerWave];    // patch element offset of the pixel's top-left tap
  int c1off[kTilesPerWave];     // byte offset of the lane's first 8 bytes in the conv1 tile; a pixel beyond the region
                                // (15 lanes of the last pixel tile) writes into the idle transpose buffer instead: no
                                // branch in the epilogue, which the scheduler interleaves with the next tile's MFMAs
  int cyx[kTilesPerWave];       // cy | cx << 8
#pragma unroll
  for (int i = 0; i < kTilesPerWave; ++i) {
    const int p = (wv + 8 * i) * 16 + r;
    const int pc = p < kC1Px ? p : kC1Px - 1;
    const int cy = pc / kC1W, cx = pc - cy * kC1W;
    pixoff[i] = 2 * cy * kPWp + 2 * cx;
    c1off[i] = p < kC1Px ? cy * kRowB + ((cx & 1) * kNEven + (cx >> 1)) * kPS + g * 8 : kC1Bytes + lane * 128;
    cyx[i] = cy | (cx << 8);
  }
  float4v al2 = float4v{0.f, 0.f, 0.f, 0.f}, be2 = al2;
  if (fused) {
    al2 = *reinterpret_cast<const float4v*>(a.alpha2 + m * 16 + g * 4);
    be2 = *reinterpret_cast<const float4v*>(a.beta2 + m * 16 + g * 4);
  }
  const int bbase = 2 * (hf * 4) * kRowB + r * kPS + g * 16;

  auto tile_origin = [&](int tt, int* n, int* oy0, int* ox0) {
    const int nn = tt / (tiles_x * tiles_y);
    const int rem = tt - nn * tiles_x * tiles_y;
    const int ty = rem / tiles_x;
    *n = nn; *oy0 = ty * kT2H; *ox0 = (rem - ty * tiles_x) * kT2W;
  };
  // input patch of tile tt -> registers, as raw bits: nothing looks at a value before the LDS write of the next
  // iteration, so all loads of a thread stay in flight over the work on the current tile.  Buffer loads: an element
  // outside the image gets an out-of-range offset and comes back as zero.  Element i = tid + 512 k of the patch is
  // (row = c * 35 + py, px): 512 = 7 * 67 + 43, so the position moves by (7 rows, 43 columns) per step.
  uint32_t v[kIter];
  auto load_patch = [&](int tt) {
    int n, oy0, ox0;
    tile_origin(tt, &n, &oy0, &ox0);
    const int es = a.x_f32 ? 4 : 2;
    const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * 3 * a.H * a.W * es;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, 3 * a.H * a.W * es, 0x00020000);
    const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 3;
    int tid_l = tid;
    asm volatile("" : "+v"(tid_l));        // positions are recomputed per tile (14 x 3 registers are not worth holding)
    int row = tid_l / kPW, px = tid_l - row * kPW;
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
      const int c = row >= 2 * kPH ? 2 : (row >= kPH ? 1 : 0);
      const int py = row - c * kPH;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = row < 3 * kPH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)(((c * a.H + iy) * a.W + ix) * es) : 0x80000000u;
      if (a.x_f32) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
      else v[k] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)off, 0, 0);
      px += kThreads - 7 * kPW; row += 7;
      if (px >= kPW) { px -= kPW; row += 1; }
    }
  };
  if (t < t_end) load_patch(t);

  for (; t < t_end; t += wg_per_xcd) {
    int n, oy0, ox0;
    tile_origin(t, &n, &oy0, &ox0);
    // ---- patch registers -> LDS, rounded to fp16 values (the tofp16 of the half wrapper); element i lands at i + i / 67 ----
    {
      int tid_w = tid;
      asm volatile("" : "+v"(tid_w));
#pragma unroll
      for (int k = 0; k < kIter; ++k) {
        const int i = tid_w + k * kThreads;
        if (i < kPatch) {
          const unsigned short hbits = (unsigned short)v[k];
          patch[i + i / kPW] = a.x_f32 ? round16(__uint_as_float(v[k])) : (float)__builtin_bit_cast(_Float16, hbits);
        }
      }
    }
    __syncthreads();                                   // B1: patch (and, the first time, ab1) visible; c1 / obuf of the previous tile free
    if (t + wg_per_xcd < t_end && !(a.ablate & 4)) load_patch(t + wg_per_xcd);

    // ---- conv1 + bn1 + relu -> LDS: 7 chained k = 4 MFMAs per (16 pixels, 16 channels) ----
    // Software-pipelined inside the wave: the BatchNorm / rounding / ReLU of pixel tile i - 1 is scheduled between the
    // MFMAs of pixel tile i (one MFMA, five vector instructions).  Measured at batch 32: chains alone 88 us (the matrix
    // pipe's fp32 rate), epilogue alone 54 us, together 142-170 us - with two waves per SIMD both in the same phase the
    // two do not overlap, interleaved or not (profiles/r04_stem_fused_ablation.txt); kept because it costs nothing.
    if (!(a.ablate & 1)) {
      const int c1y0 = 2 * oy0 - 1, c1x0 = 2 * ox0 - 1;
      struct Acc4 { float4v t[4]; };
      auto chain = [&](int i) {
        const float* pb = patch + pixoff[i];
        Acc4 acc;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) acc.t[mm] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kg = 0; kg < 7; ++kg) {
          const float x = pb[koff[kg]];
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) acc.t[mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[kg][mm], x, acc.t[mm], 0, 0, 0);
        }
        return acc;
      };
      auto epi = [&](int i, const Acc4& acc) {
        const int cy = cyx[i] & 255, cx = cyx[i] >> 8;
        // outside the conv1 map: conv2's zero padding, not relu(bn(0))
        const bool inside = (unsigned)(c1y0 + cy) < (unsigned)Ho1 && (unsigned)(c1x0 + cx) < (unsigned)Wo1;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          const float4v al = *reinterpret_cast<const float4v*>(ab1 + mm * 16 + g * 4);
          const float4v be = *reinterpret_cast<const float4v*>(ab1 + 64 + mm * 16 + g * 4);
          _Float16 o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = round16(acc.t[mm][j]);                               // conv output
            x = __builtin_fmaf(x, al[j], be[j]);                           // BN output
            asm("" : "+v"(x));                                             // (kept apart from the cast: stem_kernel rounds twice; not volatile: free to move)
            x = round16(x);
            o[j] = (_Float16)((inside && x > 0.f) ? x : 0.f);
          }
          unsigned long long raw;
          __builtin_memcpy(&raw, o, 8);
          *reinterpret_cast<unsigned long long*>(c1 + c1off[i] + mm * 32) = raw;
        }
      };
      auto interleave = [&]() {            // one MFMA, then vector work, 28 times
#pragma unroll
        for (int k = 0; k < 28; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
      };
      static_assert(kTilesPerWave == 5 && 3 + 8 * 3 < kC1Tiles, "pixel tiles 0..3 of every wave exist, tile 4 of waves 0..3");
      Acc4 acc0 = chain(0);
      Acc4 acc1 = chain(1);
      epi(0, acc0);
      interleave();
      acc0 = chain(2);
      epi(1, acc1);
      interleave();
      acc1 = chain(3);
      epi(2, acc0);
      interleave();
      if (wv + 32 < kC1Tiles) {                                            // (uniform per wave)
        acc0 = chain(4);
        epi(3, acc1);
        interleave();
        epi(4, acc0);
      } else {
        epi(3, acc1);
      }
    }
    __syncthreads();                                   // B2: the conv1 region is complete; the patch is free

    if (!fused) {
      // ---- conv1-only mode: the region's own pixels (rows 1..16, columns 1..32) -> NHWC rows of the /2 map ----
#pragma unroll
      for (int k = 0; k < 16 * 32 * 8 / kThreads; ++k) {
        const int i = tid + k * kThreads;
        const int pw = i >> 3, slot = i & 7;
        const int cy = 1 + (pw >> 5), cx = 1 + (pw & 31);
        const int oy = 2 * oy0 - 1 + cy, ox = 2 * ox0 - 1 + cx;
        if (oy >= Ho1 || ox >= Wo1) continue;
        const uint4 raw = *reinterpret_cast<const uint4*>(c1 + cy * kRowB + ((cx & 1) * kNEven + (cx >> 1)) * kPS + slot * 16);
        store16_wt(a.y1 + (((size_t)n * Ho1 + oy) * Wo1 + ox) * a.out_ld + slot * 8, raw);
      }
      continue;                                        // (the next iteration's B1 separates these reads from its conv1 writes)
    }

    // ---- conv2 on the matrix cores: k = (tap, channel), 18 steps of 32 ----
    float4v acc2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc2[nt] = float4v{0.f, 0.f, 0.f, 0.f};
    if (!(a.ablate & 2))
#pragma unroll
    for (int kci = 0; kci < 18; ++kci) {
      const int tap = kci >> 1, ty = tap / 3, tx = tap - ty * 3;
      const int ko = ty * kRowB + ((tx & 1) * kNEven + (tx >> 1)) * kPS + (kci & 1) * 64;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const uint4 b = *reinterpret_cast<const uint4*>(c1 + bbase + 2 * nt * kRowB + ko);
        acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_res[kci]), __builtin_bit_cast(half8, b),
                                                          acc2[nt], 0, 0, 0);
      }
      if (kci % 3 == 2) __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists all 72 operand reads)
    }
    // ---- bn2 (+ the conv output's own rounding) -> transpose buffer ----
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      _Float16 o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = acc2[nt][j];
        if (a.round_conv) x = round16(x);
        float tt = __builtin_fmaf(x, al2[j], be2[j]);
        asm volatile("" : "+v"(tt));                   // (no fma + cast fusion: two roundings, conv_mfma.hip)
        o[j] = (_Float16)tt;
      }
      unsigned long long raw;
      __builtin_memcpy(&raw, o, 8);
      *reinterpret_cast<unsigned long long*>(obuf + ((hf * 4 + nt) * 16 + r) * kObufRow + (m * 16 + g * 4) * 2) = raw;
    }
    __syncthreads();                                   // B3: all 64 channels of every pixel are in the buffer
    // ---- NHWC rows: 16 bytes per lane, 8 lanes per pixel ----
#pragma unroll
    for (int k = 0; k < kT2H * kT2W * 8 / kThreads; ++k) {
      const int i = tid + k * kThreads;
      const int pw = i >> 3, slot = i & 7;
      const int oy = oy0 + (pw >> 4), ox = ox0 + (pw & 15);
      if (oy >= Ho2 || ox >= Wo2 || (a.ablate & 8)) continue;
      uint4 raw = *reinterpret_cast<const uint4*>(obuf + pw * kObufRow + slot * 16);
      _Float16 hv[8];
      __builtin_memcpy(hv, &raw, 16);
      if (a.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = hv[j] > (_Float16)0.f ? hv[j] : (_Float16)0.f;
      }
      __builtin_memcpy(&raw, hv, 16);
      store16_wt(a.y + (((size_t)n * Ho2 + oy) * Wo2 + ox) * a.out_ld + slot * 8, raw);
    }
  }
}

bool stem_fused_supports(int H, int W) { return H % 4 == 0 && W % 4 == 0 && H >= 32 && W >= 32; }

// a.y1 == nullptr: fused (conv1 + conv2 -> a.y, the /4 map); a.y1 != nullptr: conv1 only -> a.y1, the /2 map
int stem_fused_launch(const StemFusedArgs& a, hipStream_t s) {
  RTPE_REQUIRE(stem_fused_supports(a.H, a.W) && a.out_ld % 8 == 0 && a.out_ld >= 64, "stem: H=%d W=%d out_ld=%d", a.H, a.W, a.out_ld);
  RTPE_REQUIRE((size_t)3 * a.H * a.W * 4 < 0x7fffffffull, "stem: image of 3 x %d x %d elements", a.H, a.W);
  RTPE_REQUIRE(((uintptr_t)a.w1 & 15) == 0 && ((uintptr_t)a.w2 & 15) == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.y1 & 15) == 0,
               "stem: alignment");
  RTPE_REQUIRE((a.y1 != nullptr) != (a.y != nullptr && a.w2 != nullptr), "stem: either the /2 output or conv2's operands");
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask))
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  const int Ho2 = a.H / 4, Wo2 = a.W / 4;
  const long tiles = (long)a.N * ((Wo2 + kT2W - 1) / kT2W) * ((Ho2 + kT2H - 1) / kT2H);
  const long per_xcd = (tiles + 7) / 8;
  // workgroups per XCD: 32 = one per CU, each with 1/256 of the tiles.  RTPE_STEM_WGS (default 1) times as many, each
  // with fewer tiles: a CU that is busy with another stream's kernels (the decode of the previous batch runs beside the
  // start of a forward) then takes fewer of them
  static const int wgs = env_int("RTPE_STEM_WGS", 1);
  const long cap = 32L * (wgs > 0 ? wgs : 1);
  const long g = per_xcd < cap ? per_xcd : cap;
  StemFusedArgs b = a;
  static const int abl = RTPE_DIAG_ENV_INT("RTPE_STEM_ABL", 0);
  b.ablate = abl;
  hipLaunchKernelGGL(stem_fused_kernel, dim3((unsigned)(8 * g)), dim3(kThreads), kLds, s, b);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
