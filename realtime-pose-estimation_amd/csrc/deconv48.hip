// ConvTranspose2d(k4, s2, p1) + BN + ReLU to 48 channels (the deconv layer of HigherHRNet's final stage,
// pose_higher_hrnet.py:447-483 / _make_deconv_layers: 82 = 48 + 17 + 17 input channels in a 96-channel row) as ONE
// persistent kernel over all four sub-pixel classes, on the structure of conv64.hip.
//
// The transposed conv is four 2 x 2-tap convs (one per output parity class (a, b): output pixel (2y + a, 2x + b) reads the
// input pixels (y + lo_y .. + 1, x + lo_x .. + 1), lo in {-1, 0}; conv_mfma.hip packs each class's weights as its own plan).
// On the one-workgroup-per-tile kernel the four classes ran as one grid, but every workgroup staged its own halo tile (the
// same input pixels four times), fetched its 36 KiB of weight fragments from L2 in every k step, and stored 96-byte pixel
// pieces 192 bytes apart (the other parity's pixels lie between them): 254 us at batch 32 for a layer whose HBM floor is
// ~80 us.  Here:
//   * persistent workgroups of 8 waves, one per CU, walking tiles of 8 x 16 INPUT positions (16 x 32 output pixels);
//   * multiplier wave k IS class k: the 36 (24 for a 48-channel input) weight fragments of its class stay in registers for
//     the whole kernel; it reads the ONE halo tile (10 x 18 pixels, all input channels of a pixel contiguous) all four
//     classes share, in four passes of 2 rows of 16 positions: 2 B-operand reads (one k step ahead) and 6 MFMAs per k
//     step, k order of the class's plan (chunk, tap, channel: bit-identical to conv_mfma.hip, tests/test_gpu_parity.py);
//   * the four classes' results meet in a transpose buffer laid out as the 16 x 32 OUTPUT pixels of the tile, so the 4
//     mover waves store whole output rows (32 pixels x 96 bytes contiguous) and request the next tile's halo by LDS-DMA
//     while this one is multiplied.
// LDS: 2 x 40,320 B halo tiles + 57,344 B transpose buffer = 135 KiB (96-channel rows).
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

#ifndef RTPE_D48_ORDER
#define RTPE_D48_ORDER 1
#endif

namespace {
constexpr int kTH = 8, kTW = 16;                           // input positions per tile
constexpr int kHH = kTH + 2, kHW = kTW + 2;                // halo tile: 10 x 18 (offsets -1 .. +1 around every position)
constexpr int kOP = 112;                                   // transpose buffer: bytes per output pixel (96 + 16)
constexpr int kObufBytes = 4 * kTH * kTW * kOP;            // 57,344
constexpr int kThreads = 512;
constexpr int kMulWaves = 4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int NCC>
struct Shape {
  static constexpr int kPS = NCC * 96 + (NCC == 2 ? 32 : 64);   // LDS bytes per halo pixel (pstride % 64 == 32)
  static constexpr int kSlots = kPS / 16, kData = NCC * 6;      // 16-byte slots per pixel, of them data
  static constexpr int kRowB = kHW * kPS;
  static constexpr int kBufBytes = kHH * kRowB;
  static constexpr int kTileSlots = kBufBytes / 16;
  static constexpr int kDmaIter = (kTileSlots + 255) / 256;       // wave-instructions per mover wave and tile
  static constexpr int kLds = 2 * kBufBytes + kObufBytes;
  static constexpr int kK = NCC * 6;                            // k steps of 32
};

__device__ __forceinline__ float round16(float v) { return (float)(_Float16)v; }
}  // namespace

template <int NCC, bool ROUND>
__global__ void __launch_bounds__(kThreads) deconv48_kernel(const ConvArgs a) {
  using S = Shape<NCC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufs = smem;
  char* const obuf = smem + 2 * S::kBufBytes;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  const int total = a.N * tiles_x * tiles_y;
  // an XCD takes a contiguous eighth of the row-major tile list (neighbouring tiles share halo rows and columns in its L2)
  const int per_xcd = (total + 7) >> 3, wg_per_xcd = (int)(gridDim.x >> 3);
  const int xcd = (int)(blockIdx.x & 7u);
  const int t_end = (xcd + 1) * per_xcd < total ? (xcd + 1) * per_xcd : total;
  const int t0 = xcd * per_xcd + (int)(blockIdx.x >> 3);
  if (t0 >= t_end) return;

  auto tile_origin = [&](int tt, int* n, int* py0, int* px0) {
    const uint32_t nn = fdiv((uint32_t)tt, a.div_tiles_xy);
    const uint32_t rem = (uint32_t)tt - nn * (uint32_t)(tiles_x * tiles_y);
    const uint32_t ty = fdiv(rem, a.div_tiles_x);
    *n = (int)nn; *py0 = (int)ty * kTH; *px0 = (int)(rem - ty * tiles_x) * kTW;
  };

  // Two barriers per tile, passed by all 8 waves (conv64.hip):
  //   T(u): the halo tile u has landed and the transpose buffer holds the output rows of tile u - 1;
  //   O(u): the movers have read those rows out (the multipliers may overwrite the buffer with tile u's).
  if (wv >= kMulWaves) {
    // ------------------------------------ movers: 4 waves ------------------------------------
    // Per tile: the output rows of the previous tile out of the transpose buffer (registers before O, stores behind it) and
    // the halo tile of the next one by LDS-DMA.  What a lane moves in instruction k is the same 16 bytes of the same pixel
    // of every tile: LDS address, byte offset from the tile's first pixel and the pixel's tile coordinates are computed
    // once (computed per tile - divisions, 64-bit addresses, launch arguments re-read from the kernel-argument segment -
    // an instruction cost ~200 cycles); loads and stores go through buffer descriptors, a piece outside the image
    // (ragged tiles, the zero border) or beyond the stored channels gets an out-of-range offset: zeros / dropped
    const int mw = wv - kMulWaves, mt = tid - kMulWaves * 64;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.buf_bytes, 0x00020000);
    const int H_in = a.H_in, W_in = a.W_in, in_ld = a.in_ld;
    const int H_full = a.H_full, W_full = a.W_full, out_ld = a.out_ld;
    int rel[S::kDmaIter], hyx[S::kDmaIter];
#pragma unroll
    for (int k = 0; k < S::kDmaIter; ++k) {
      const int sl = (mw + 4 * k) * 64 + lane;
      const int pix = sl / S::kSlots, slot = sl - pix * S::kSlots;
      const int hy = pix / kHW, hx = pix - hy * kHW;
      const bool data = sl < S::kTileSlots && slot < S::kData;
      rel[k] = ((hy * W_in + hx) * in_ld + slot * 8) * 2;
      hyx[k] = data ? (hy | (hx << 8)) : 0x4000;             // padding: a row no image has
    }
    auto request_tile = [&](int tt, char* buf) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      const int iy0 = py0 - 1, ix0 = px0 - 1;
      const int base = ((n * H_in + iy0) * W_in + ix0) * in_ld * 2;   // (may be negative: only added to lanes inside the image)
#pragma unroll
      for (int k = 0; k < S::kDmaIter; ++k) {
        const int s0 = (mw + 4 * k) * 64;
        if (s0 >= S::kTileSlots) break;
        const int iy = iy0 + (hyx[k] & 0x40ff), ix = ix0 + ((hyx[k] >> 8) & 0x3f);
        const bool ok = (unsigned)iy < (unsigned)H_in && (unsigned)ix < (unsigned)W_in;
        const uint32_t voff = ok ? (uint32_t)(base + rel[k]) : 0x80000000u;
        if (s0 + 64 <= S::kTileSlots || s0 + lane < S::kTileSlots)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + s0 * 16), 16, (int)voff, 0, 0, 0);
      }
    };
    constexpr int kPieces = 4 * kTH * kTW * 6 / 256;         // 12 per mover lane: 16 bytes, 6 lanes per output pixel
    int lds_off[kPieces], orel[kPieces], oyx[kPieces];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
      const int i = mt + k * 256;
      const int pw = i / 6, slot = i - pw * 6;
      lds_off[k] = pw * kOP + slot * 16;
      orel[k] = (((pw >> 5) * W_full + (pw & 31)) * out_ld + slot * 8) * 2;
      oyx[k] = slot * 8 < a.cout_store ? ((pw >> 5) | ((pw & 31) << 8)) : 0x4000;
    }
    const bool relu = a.relu != 0;
    uint4 raw[kPieces];
    auto read_rows = [&]() {
#pragma unroll
      for (int k = 0; k < kPieces; ++k) raw[k] = *reinterpret_cast<const uint4*>(obuf + lds_off[k]);
    };
    auto store_rows = [&](int tt) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      const int base = ((n * H_full + 2 * py0) * W_full + 2 * px0) * out_ld * 2;
      const int ly = H_full - 2 * py0, lx = W_full - 2 * px0;     // rows / columns of the tile inside the image
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {
        half8 hv = __builtin_bit_cast(half8, raw[k]);
        if (relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
        const bool ok = (oyx[k] & 0x40ff) < ly && (oyx[k] >> 8 & 0x3f) < lx;
        const uint32_t voff = ok ? (uint32_t)(base + orel[k]) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), ysrc, (int)voff, 0, 0);
      }
    };
    request_tile(t0, bufs);
    int cur = 0, prev = -1;
    for (int t = t0; t < t_end; t += wg_per_xcd) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed (its older stores too)
      __syncthreads();                                      // T: the multipliers are done with the other buffer (tile t - 1)
      if (RTPE_D48_ORDER == 0 && t + wg_per_xcd < t_end && !(a.ablate & 4)) request_tile(t + wg_per_xcd, bufs + (cur ^ 1) * S::kBufBytes);
      if (prev >= 0) read_rows();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the rows are in registers
      __syncthreads();                                      // O
      if (RTPE_D48_ORDER == 1 && t + wg_per_xcd < t_end && !(a.ablate & 4)) request_tile(t + wg_per_xcd, bufs + (cur ^ 1) * S::kBufBytes);
      if (prev >= 0 && !(a.ablate & 2)) store_rows(prev);
      if (RTPE_D48_ORDER == 2 && t + wg_per_xcd < t_end && !(a.ablate & 4)) request_tile(t + wg_per_xcd, bufs + (cur ^ 1) * S::kBufBytes);
      prev = t;
      cur ^= 1;
    }
    __syncthreads();                                        // the last tile's rows are in the transpose buffer
    read_rows();
    if (!(a.ablate & 2)) store_rows(prev);
    return;
  }

  // ---------------------------------- multipliers: wave k = class k ----------------------------------
  const int r = lane & 15, g = lane >> 4;
  const int lo_y = a.lo_yc[wv], lo_x = a.lo_xc[wv], pa = a.oy_c[wv], pb = a.ox_c[wv];
  uint4 a_res[S::kK][3];
  {
    const uint4* wfrag = reinterpret_cast<const uint4*>(a.w_c[wv]) + lane;
#pragma unroll
    for (int k = 0; k < S::kK; ++k)
#pragma unroll
      for (int m = 0; m < 3; ++m) a_res[k][m] = wfrag[(size_t)(k * 3 + m) * 64];
  }
  float4v al[3], be[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    al[m] = *reinterpret_cast<const float4v*>(a.alpha + m * 16 + g * 4);
    be[m] = *reinterpret_cast<const float4v*>(a.beta + m * 16 + g * 4);
  }
  // k step kk of a chunk, lane group g: flat k = kk * 32 + g * 8 = tap * 48 + channel (2 x 2 taps)
  int toff[6];
#pragma unroll
  for (int kk = 0; kk < 6; ++kk) {
    const int k = kk * 32 + g * 8, tap = k / 48, c = k - tap * 48;
    toff[kk] = (tap >> 1) * S::kRowB + (tap & 1) * S::kPS + c * 2;
  }
  // halo pixel of position (0, r), tap (0, 0): the halo tile starts at (-1, -1)
  const int bbase = (1 + lo_y) * S::kRowB + (r + 1 + lo_x) * S::kPS;
  // transpose buffer: output pixel (2 * py + pa, 2 * r + pb) of the tile, channels g * 4 .. + 3 of cout tile m
  const int obase = (pa * 32 + 2 * r + pb) * kOP + g * 8;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (int t = t0; t < t_end; t += wg_per_xcd) {
    const char* tb = bufs + cur * S::kBufBytes + bbase;
    __syncthreads();                                        // T
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      float4v acc[3][2];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
      if (!(a.ablate & 1)) {
        const char* th = tb + h * 2 * S::kRowB;
        uint4 bf[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bf[0][nt] = *reinterpret_cast<const uint4*>(th + nt * S::kRowB + toff[0]);
#pragma unroll
        for (int k = 0; k < S::kK; ++k) {
          const int cb = k & 1, nb = cb ^ 1;
          if (k + 1 < S::kK) {
            const int ko = ((k + 1) / 6) * 96;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bf[nb][nt] = *reinterpret_cast<const uint4*>(th + nt * S::kRowB + toff[(k + 1) % 6] + ko);
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int m = 0; m < 3; ++m)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_res[k][m]),
                                                                  __builtin_bit_cast(half8, bf[cb][nt]), acc[m][nt], 0, 0, 0);
          if (k + 1 < S::kK) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (h == 0) __syncthreads();                          // O: the transpose buffer is free
      // BN (+ the conv output's own fp16 rounding) on pairs of values: packed conversions and packed fp32 FMAs - with 12 k
      // steps per accumulator tile the epilogue's vector instructions are a fifth of the multipliers' time
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          half2v o[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float2v x = {acc[m][nt][2 * j], acc[m][nt][2 * j + 1]};
            if (ROUND) x = __builtin_convertvector(__builtin_convertvector(x, half2v), float2v);
            float2v tt = __builtin_elementwise_fma(x, float2v{al[m][2 * j], al[m][2 * j + 1]}, float2v{be[m][2 * j], be[m][2 * j + 1]});
            asm volatile("" : "+v"(tt));                     // (no fma + cast fusion: two roundings, conv_mfma.hip)
            o[j] = __builtin_convertvector(tt, half2v);
          }
          unsigned long long raw;
          __builtin_memcpy(&raw, o, 8);
          *reinterpret_cast<unsigned long long*>(obuf + obase + (2 * (2 * h + nt)) * 32 * kOP + m * 32) = raw;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur ^= 1;
  }
  __syncthreads();                                          // the last tile's rows are in the transpose buffer
}

// `p`: the plan all four classes share (the engine merges them only when they agree), `a`: the merged argument block
// (ConvArgs::n_cls == 4, w_c / lo_yc / lo_xc / oy_c / ox_c per class)
bool deconv48_supports(const ConvPlan& p, const ConvArgs& a) {
  if (!(p.esize == 2 && p.dil == 1 && p.tapw == 2 && p.in_mul == 1 && p.mt == 3 && p.cc == 48 && p.kc == 6 && p.n_cb == 1 &&
        p.cout_pad == 48 && (p.n_cchunks == 1 || p.n_cchunks == 2)))
    return false;
  if (!(a.n_cls == 4 && a.o_mul == 2 && a.res == nullptr && a.y != nullptr && a.y_nchw == nullptr && a.in_cs == 48 &&
        a.in_ld >= 48 * p.n_cchunks && a.out_cs == 48 && a.out_ld >= a.cout_store && a.cout_store % 8 == 0 && a.cout_store <= 48 &&
        a.x_bytes > 0 && a.x_bytes < 0x80000000ull && (size_t)a.N * a.H_full * a.W_full * a.out_ld * 2 < 0x80000000ull &&
        a.H_pos == a.H_in && a.W_pos == a.W_in && a.H_full == 2 * a.H_in &&
        a.W_full == 2 * a.W_in))
    return false;
  for (int k = 0; k < 4; ++k)
    if (a.w_c[k] == nullptr || a.oy_c[k] != (k >> 1) || a.ox_c[k] != (k & 1) || a.lo_yc[k] < -1 || a.lo_yc[k] > 0 ||
        a.lo_xc[k] < -1 || a.lo_xc[k] > 0)
      return false;
  return true;
}

int deconv48_launch(const ConvPlan& p, const ConvArgs& a_in, hipStream_t s) {
  RTPE_REQUIRE(deconv48_supports(p, a_in), "deconv48: unsupported layer");
  ConvArgs a = a_in;
  a.buf_bytes = (int)((size_t)a.N * a.H_full * a.W_full * a.out_ld * 2);   // (the output view's bytes: the storers' buffer window)
  a.tiles_x = (a.W_in + kTW - 1) / kTW;
  a.tiles_y = (a.H_in + kTH - 1) / kTH;
  a.div_tiles_x = make_fastdiv((uint32_t)a.tiles_x);
  a.div_tiles_xy = make_fastdiv((uint32_t)(a.tiles_x * a.tiles_y));
  const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
  RTPE_REQUIRE(tiles < (1l << 20), "deconv48: %ld tiles", tiles);
  const long per_xcd = (tiles + 7) / 8;
  const unsigned grid = (unsigned)(8 * (per_xcd < 32 ? per_xcd : 32));   // one workgroup per CU
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(deconv48_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Shape<1>::kLds));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(deconv48_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Shape<1>::kLds));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(deconv48_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Shape<2>::kLds));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(deconv48_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Shape<2>::kLds));
  }
  const dim3 g(grid), b(kThreads);
  if (p.n_cchunks == 2) {
    if (a.round_conv) hipLaunchKernelGGL((deconv48_kernel<2, true>), g, b, Shape<2>::kLds, s, a);
    else hipLaunchKernelGGL((deconv48_kernel<2, false>), g, b, Shape<2>::kLds, s, a);
  } else {
    if (a.round_conv) hipLaunchKernelGGL((deconv48_kernel<1, true>), g, b, Shape<1>::kLds, s, a);
    else hipLaunchKernelGGL((deconv48_kernel<1, false>), g, b, Shape<1>::kLds, s, a);
  }
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
