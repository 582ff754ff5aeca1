// 3x3 stride-2 conv + BN (+ReLU) from 48 input channels to 48 / 96 / 192 / 384 output channels: the downsampling convs of
// HigherHRNet's fuse layers (pose_higher_hrnet.py:213-230: branch 0's map towards branches 1-3; 25 launches per forward), on
// the structure of conv64.hip / deconv48.hip.
//
// On the streaming kernel (conv_stream.hip, weights resident in LDS) these layers were bound by the LDS array: a stride-2 halo
// tile is 4x its output tile, so a workgroup's tile is 64-128 output pixels - every wave re-read all MT weight fragments of a k
// step from LDS for one or two pixel tiles (mt 6, nt 1: 6 KiB of A operands per 6 MFMAs and wave = 256 B per clock and CU
// against the array's 128), and the B-operand reads of 16 consecutive output pixels, 192 bytes apart in a tile stored in input
// order, hit every bank four times.  Here:
//   * persistent workgroups of 8 waves, one per CU, walking (tile, 48 GW-channel cout block) units; a workgroup keeps its block;
//   * multiplier wave w owns ONE group of 48 output channels - its 42 weight fragments (14 k steps x 3 cout tiles, from the conv
//     op's own packed plan) stay in registers for the whole kernel - and 1 / (4 / GW) of the tile's pixels: GW = 1, 2 or 4 groups
//     per workgroup (48, 96, 192 / 384 output channels); per k step 2 B-operand reads (one step ahead) and 6 MFMAs, k order of
//     the plan (tap, channel: bit-identical to the other kernels, tests/test_gpu_parity.py);
//   * the halo tile keeps the even input columns of a row first, then the odd ones (as conv_mfma.hip's stride-2 tiles): the 16
//     pixels of a B-operand read are consecutive 96-byte LDS pixels, conflict-free;
//   * 2 requester waves keep the halo tiles of the next kD units in flight (LDS-DMA into kNB = kD + 1 buffers; a lane moves the
//     same bytes of the same tile pixel in every unit: offsets computed once, deconv48.hip) and 2 storer waves move the previous
//     unit's output rows out of a transpose buffer.  The layers are memory-bound with little arithmetic per byte (2.7k cycles
//     of MFMAs for 28 KiB of halo): with ONE tile in flight per CU (8 x 16 tiles, two buffers: the first form of this kernel)
//     a unit took a memory round trip, 3.4 TB/s chip-wide, no faster than the streaming kernel.
// LDS: 8 x 8 output tiles: kNB x 28,288 B of halo tiles + 7-25.6 KiB transpose buffer.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

namespace {
constexpr int kK = 14;                                      // k steps of 32 over 9 taps x 48 channels (the last one half padding)
constexpr int kPS = 96;                                     // LDS bytes per halo pixel (pstride % 64 == 32)
constexpr int kThreads = 512;
constexpr int kMulWaves = 4;
constexpr int kTH = 8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef RTPE_C48S2_DEPTH
#define RTPE_C48S2_DEPTH 3
#endif
constexpr int kD = RTPE_C48S2_DEPTH;                        // halo tiles in flight ahead of the one being multiplied
constexpr int kNB = kD + 1;
constexpr int kTW = 8;                                      // output tile: 8 x 8
constexpr int kHH = 2 * kTH + 1, kHW = 2 * kTW + 1;         // halo tile: 17 x 17 input pixels
constexpr int kNEven = (kHW + 1) / 2;                       // even input columns first, then the odd ones
// a pixel tile of 16 is two output rows; the second row's reads are conflict-free beside the first's when 2 * row pitch - 8 * 96
// is a multiple of the 256-byte bank period (conv_row_pitch's rule)
constexpr int kRowB = 1664;
constexpr int kBufBytes = kHH * kRowB;                      // 28,288
constexpr int kTileSlots = kBufBytes / 16;                  // 1,768
constexpr int kRowSlots = kRowB / 16;
constexpr int kDmaIter = (kTileSlots + 127) / 128;          // 14 wave-instructions per requester wave and tile
static_assert(kRowB >= kHW * kPS && kTileSlots > (2 * (kDmaIter - 1) + 1) * 64, "both requesters issue kDmaIter instructions per tile");
static_assert(kD >= 1 && kD <= 3 && kD * kDmaIter < 64, "vmcnt is a 6-bit counter");

template <int GW>
struct Shape {
  static constexpr int kPxWave = kTH * kTW * GW / 4;            // output pixels per multiplier wave: 16 / 32 / 64
  static constexpr int kNT = kPxWave >= 32 ? 2 : 1;             // pixel tiles per pass
  static constexpr int kPass = kPxWave / (16 * kNT);            // 1 / 1 / 2
  static constexpr int kOP = GW * 96 + 16;                      // transpose buffer: bytes per output pixel
  static constexpr int kObufBytes = kTH * kTW * kOP;
  static constexpr int kPieces = kTH * kTW * GW * 6 / 128;      // 16-byte output pieces per storer lane: 3 / 6 / 12
  static constexpr int kLds = kNB * kBufBytes + kObufBytes;
  static_assert(kLds <= 160 * 1024, "LDS layout");
};
}  // namespace

// One group of 48 output channels of a workgroup's block: wave slot gl of the block takes group grp0 + cb * GW of the layer
// the entry describes.  A launch for ONE layer has GW entries of that layer (grp0 = 0 .. GW - 1); a launch for several
// layers that read the same input (conv48s2_launch_group: the first downsampling convs from branch 0 of a fuse layer) has
// the groups of all of them, n_cb = 1.
struct S2Group {
  const _Float16* w;      // the layer's packed weights (plan: mt_pack cout tiles per block)
  const float* alpha;     // its BN scale / shift by output channel
  const float* beta;
  _Float16* y;            // its NHWC output view
  int mt_pack, grp0, out_ld, cout_store, relu;
  unsigned y_bytes;       // bytes of the view (the storers' buffer window)
};
struct S2Args {
  const _Float16* x;
  unsigned x_bytes;
  int N, H_in, W_in, in_ld, H_out, W_out;
  int tiles_x, tiles_y, n_cb, ablate;
  FastDiv div_tiles_x, div_tiles_xy, div_ncb;
  S2Group grp[4];
};

template <int GW, bool ROUND>
__global__ void __launch_bounds__(kThreads) conv48s2_kernel(const S2Args a) {
  using S = Shape<GW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufs = smem;
  char* const obuf = smem + kNB * kBufBytes;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, n_cb = a.n_cb;
  // units = (tile, cout block), the block fastest; an XCD takes a contiguous eighth of the list (a multiple of n_cb units), its
  // workgroups every (grid / 8)-th unit of it: grid / 8 is a multiple of n_cb, so a workgroup keeps its cout block
  const int total = a.N * tiles_x * tiles_y * n_cb;
  const int wg_per_xcd = (int)(gridDim.x >> 3);
  int per_xcd = (total + 7) >> 3;
  per_xcd = (per_xcd + n_cb - 1) / n_cb * n_cb;
  const int xcd = (int)(blockIdx.x & 7u);
  const int u_end = (xcd + 1) * per_xcd < total ? (xcd + 1) * per_xcd : total;
  const int u0 = xcd * per_xcd + (int)(blockIdx.x >> 3);
  if (u0 >= u_end) return;
  const int cb = u0 % n_cb;                                  // this workgroup's block of GW x 48 output channels

  auto tile_origin = [&](int u, int* n, int* py0, int* px0) {
    const uint32_t tt = fdiv((uint32_t)u, a.div_ncb);
    const uint32_t nn = fdiv(tt, a.div_tiles_xy);
    const uint32_t rem = tt - nn * (uint32_t)(tiles_x * tiles_y);
    const uint32_t ty = fdiv(rem, a.div_tiles_x);
    *n = (int)nn; *py0 = (int)ty * kTH; *px0 = (int)(rem - ty * tiles_x) * kTW;
  };

  // Two barriers per unit, passed by all 8 waves (conv64.hip):
  //   T(u): the halo tile u has landed and the transpose buffer holds the output rows of unit u - 1;
  //   O(u): the storers have read those rows out (the multipliers may overwrite the buffer with unit u's).
  // Unit number i of this workgroup uses halo buffer i % kNB.
  if (wv >= kMulWaves + 2) {
    // ------------------------------------ storers: 2 waves ------------------------------------
    // output rows: 16 bytes per lane, GW * 6 lanes per pixel; in registers before O, stored behind it and never waited for
    const int st = tid - (kMulWaves + 2) * 64;
    const int H_out = a.H_out, W_out = a.W_out;
    // pieces in group-major order: instruction k of a storer moves pieces of group k / 3 only (64 pixels x 6 pieces = 3 x 128
    // lanes), so its buffer descriptor, row pitch and ReLU flag are uniform
    constexpr int kPerGroup = 3;
    static_assert(kTH * kTW * 6 == kPerGroup * 128 && S::kPieces == GW * kPerGroup, "three store instructions per group");
    __amdgpu_buffer_rsrc_t ysrc[GW];
    int lds_off[S::kPieces], orel[S::kPieces], oyx[S::kPieces];
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) {
      const S2Group& e = a.grp[gi];
      ysrc[gi] = __builtin_amdgcn_make_buffer_rsrc(e.y, 0, (int)e.y_bytes, 0x00020000);
      const int ch0 = (e.grp0 + cb * GW) * 48;
#pragma unroll
      for (int kk = 0; kk < kPerGroup; ++kk) {
        const int k = gi * kPerGroup + kk;
        const int j = st + kk * 128;
        const int pw = j / 6, slot = j - pw * 6;
        const int oy = pw / kTW, ox = pw - oy * kTW;
        lds_off[k] = pw * S::kOP + gi * 96 + slot * 16;
        orel[k] = ((oy * W_out + ox) * e.out_ld + ch0 + slot * 8) * 2;
        oyx[k] = ch0 + slot * 8 < e.cout_store ? (oy | (ox << 8)) : 0x4000;
      }
    }
    uint4 raw[S::kPieces];
    auto read_rows = [&]() {
#pragma unroll
      for (int k = 0; k < S::kPieces; ++k) raw[k] = *reinterpret_cast<const uint4*>(obuf + lds_off[k]);
    };
    auto store_rows = [&](int u) {
      int n, py0, px0;
      tile_origin(u, &n, &py0, &px0);
      const int pix0 = (n * H_out + py0) * W_out + px0;
      const int ly = H_out - py0, lx = W_out - px0;          // rows / columns of the tile inside the map
#pragma unroll
      for (int k = 0; k < S::kPieces; ++k) {
        const S2Group& e = a.grp[k / kPerGroup];
        half8 hv = __builtin_bit_cast(half8, raw[k]);
        if (e.relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
        const bool ok = (oyx[k] & 0x40ff) < ly && (oyx[k] >> 8 & 0x3f) < lx;
        const uint32_t voff = ok ? (uint32_t)(pix0 * e.out_ld * 2 + orel[k]) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), ysrc[k / kPerGroup], (int)voff, 0, 0);
      }
    };
    int prev = -1;
    for (int u = u0; u < u_end; u += wg_per_xcd) {
      __syncthreads();                                      // T
      if (prev >= 0) read_rows();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the rows are in registers
      __syncthreads();                                      // O
      if (prev >= 0 && !(a.ablate & 2)) store_rows(prev);
      prev = u;
    }
    __syncthreads();                                        // the last unit's rows are in the transpose buffer
    read_rows();
    if (!(a.ablate & 2)) store_rows(prev);
    return;
  }
  if (wv >= kMulWaves) {
    // ------------------------------------ requesters: 2 waves ------------------------------------
    const int mw = wv - kMulWaves;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const int H_in = a.H_in, W_in = a.W_in, in_ld = a.in_ld;
    // slot L of the tile image: row hy, LDS column lc (input column 2 lc for the even ones, 2 (lc - kNEven) + 1 for the odd
    // ones), 16-byte slot s of the pixel; row padding and pixels outside the image: out-of-range offset, zeros
    int rel[kDmaIter], hyx[kDmaIter];
#pragma unroll
    for (int k = 0; k < kDmaIter; ++k) {
      const int sl = (mw + 2 * k) * 64 + lane;
      const int hy = sl / kRowSlots, rem = sl - hy * kRowSlots;
      const int lc = rem / 6, s = rem - lc * 6;
      const int col = lc < kNEven ? 2 * lc : 2 * (lc - kNEven) + 1;
      const bool data = sl < kTileSlots && lc < kHW;
      rel[k] = ((hy * W_in + col) * in_ld + s * 8) * 2;
      hyx[k] = data ? (hy | (col << 8)) : 0x4000;            // padding: a row no image has
    }
    // every call issues exactly kDmaIter instructions (the counted waits below rely on it)
    auto request_tile = [&](int u, char* buf) {
      int n, py0, px0;
      tile_origin(u, &n, &py0, &px0);
      const int iy0 = 2 * py0 - 1, ix0 = 2 * px0 - 1;
      const int base = ((n * H_in + iy0) * W_in + ix0) * in_ld * 2;   // (may be negative: only added to lanes inside the image)
#pragma unroll
      for (int k = 0; k < kDmaIter; ++k) {
        const int s0 = (mw + 2 * k) * 64;
        const int iy = iy0 + (hyx[k] & 0x40ff), ix = ix0 + ((hyx[k] >> 8) & 0x3f);
        const bool ok = (unsigned)iy < (unsigned)H_in && (unsigned)ix < (unsigned)W_in;
        // (lanes behind the tile image in its last instruction: offset and LDS slot of the image's last piece would be wrong
        // places - they are switched off; the instruction still counts)
        const uint32_t voff = ok ? (uint32_t)(base + rel[k]) : 0x80000000u;
        if (s0 + 64 <= kTileSlots || s0 + lane < kTileSlots)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + s0 * 16), 16, (int)voff, 0, 0, 0);
      }
    };
    int i_req = 0;                                          // units requested so far
    for (int u = u0; i_req < kD && u < u_end; u += wg_per_xcd, ++i_req) request_tile(u, bufs + i_req * kBufBytes);
    int u_req = u0 + kD * wg_per_xcd, b_req = kD % kNB;      // next unit to request and its buffer
    for (int u = u0; u < u_end; u += wg_per_xcd) {
      // this wave's pieces of unit u's tile have landed: everything but the requests made after it (loads complete in order)
      const int after = (u_end - 1 - u) / wg_per_xcd;       // units of this workgroup behind u
      if (kD >= 3 && after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kDmaIter) : "memory");
      else if (kD >= 2 && after >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaIter) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                      // T: the multipliers are done with unit u - 1's buffer
      __syncthreads();                                      // O
      if (u_req < u_end && !(a.ablate & 4)) request_tile(u_req, bufs + b_req * kBufBytes);   // into the buffer of unit u - 1
      u_req += wg_per_xcd;
      b_req = b_req + 1 == kNB ? 0 : b_req + 1;
    }
    __syncthreads();
    return;
  }

  // ---------------------------------- multipliers: 4 waves ----------------------------------
  constexpr int kWavesPerGroup = 4 / GW;                     // waves that share a group's pixels
  const int gl = wv / kWavesPerGroup, part = wv - gl * kWavesPerGroup;
  const int r = lane & 15, g = lane >> 4;
  const S2Group& eg = a.grp[gl];
  if (eg.w == nullptr) {                                     // an empty slot (three groups in a block of four): barriers only
    for (int u = u0; u < u_end; u += wg_per_xcd) {
      __syncthreads();                                      // T
      __syncthreads();                                      // O
    }
    __syncthreads();
    return;
  }
  const int grp = eg.grp0 + cb * GW;                         // this wave's group of 48 output channels of its layer
  const int mt_pack = eg.mt_pack;                            // cout tiles per packed block of the layer's plan (3 or 6)
  uint4 a_res[kK][3];
  {
    const int blk = grp * 3 / mt_pack, m0 = grp * 3 - blk * mt_pack;
    const uint4* wfrag = reinterpret_cast<const uint4*>(eg.w) + ((size_t)blk * kK * mt_pack + m0) * 64 + lane;
#pragma unroll
    for (int k = 0; k < kK; ++k)
#pragma unroll
      for (int m = 0; m < 3; ++m) a_res[k][m] = wfrag[(size_t)(k * mt_pack + m) * 64];
  }
  // k step kk, lane group g: flat k = kk * 32 + g * 8 = tap * 48 + channel; beyond 9 x 48 the weights are zero (any finite
  // in-tile data: the tile image is wholly written by every request, zeros where there is no input)
  int toff[kK];
#pragma unroll
  for (int kk = 0; kk < kK; ++kk) {
    int k = kk * 32 + g * 8;
    if (k >= 9 * 48) k -= 9 * 48;
    const int tap = k / 48, c = k - tap * 48;
    const int ty = tap / 3, tx = tap - ty * 3;
    // input column 2 ox + tx: the even columns ox (tx 0) and ox + 1 (tx 2), the odd column ox (tx 1)
    const int cs = tx == 0 ? 0 : tx == 1 ? kNEven : 1;
    toff[kk] = ty * kRowB + cs * kPS + c * 2;
  }
  // pixel tile T of the tile (16 output pixels): output rows 2 T and 2 T + 1, column r & 7
  const int lane_off = 2 * (r >> 3) * kRowB + (r & 7) * kPS;
  constexpr int kTileStep = 4 * kRowB;                       // LDS bytes between the input rows of two pixel tiles
  constexpr int NT = S::kNT;
  const int t_first = part * (S::kPxWave / 16);
  const float* const alp = eg.alpha + grp * 48 + g * 4;
  const float* const bep = eg.beta + grp * 48 + g * 4;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (int u = u0; u < u_end; u += wg_per_xcd) {
    const char* tb = bufs + cur * kBufBytes + lane_off;
    __syncthreads();                                        // T
#pragma unroll 1
    for (int h = 0; h < S::kPass; ++h) {
      float4v acc[3][NT];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
      const int t0 = t_first + NT * h;
      if (!(a.ablate & 1)) {
        const char* th = tb + t0 * kTileStep;
        uint4 bf[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const uint4*>(th + nt * kTileStep + toff[0]);
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const int cbuf = k & 1, nb = cbuf ^ 1;
          if (k + 1 < kK) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nb][nt] = *reinterpret_cast<const uint4*>(th + nt * kTileStep + toff[k + 1]);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < 3; ++m)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_res[k][m]),
                                                                  __builtin_bit_cast(half8, bf[cbuf][nt]), acc[m][nt], 0, 0, 0);
          if (k + 1 < kK) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (h == 0) __syncthreads();                          // O: the transpose buffer is free
      // BN (+ the conv output's own fp16 rounding) on pairs of values (deconv48.hip); the BN parameters come from L1 / L2 in
      // every pass: 42 weight fragments leave no registers to keep them
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const float4v al = *reinterpret_cast<const float4v*>(alp + m * 16), be = *reinterpret_cast<const float4v*>(bep + m * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          half2v o[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float2v x = {acc[m][nt][2 * j], acc[m][nt][2 * j + 1]};
            if (ROUND) x = __builtin_convertvector(__builtin_convertvector(x, half2v), float2v);
            float2v tt = __builtin_elementwise_fma(x, float2v{al[2 * j], al[2 * j + 1]}, float2v{be[2 * j], be[2 * j + 1]});
            asm volatile("" : "+v"(tt));                     // (no fma + cast fusion: two roundings, conv_mfma.hip)
            o[j] = __builtin_convertvector(tt, half2v);
          }
          unsigned long long raw;
          __builtin_memcpy(&raw, o, 8);
          *reinterpret_cast<unsigned long long*>(obuf + ((t0 + nt) * 16 + r) * S::kOP + gl * 96 + m * 32 + g * 8) = raw;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur = cur + 1 == kNB ? 0 : cur + 1;
  }
  __syncthreads();                                          // the last unit's rows are in the transpose buffer
}

// cout groups per workgroup for a layer of `cout` output channels (0: not one of this kernel's)
static int conv48s2_gw(int cout) { return cout == 48 ? 1 : cout == 96 ? 2 : (cout == 192 || cout == 384) ? 4 : 0; }

bool conv48s2_supports(const ConvPlan& p, const ConvArgs& a) {
  if (!(p.esize == 2 && p.dil == 1 && p.tapw == 3 && p.in_mul == 2 && p.cc == 48 && p.n_cchunks == 1 && p.kc == kK &&
        p.cin == 48 && (p.mt == 3 || p.mt == 6) && conv48s2_gw(p.cout) != 0 && p.cout_pad == p.cout))
    return false;
  return a.n_cls == 0 && a.o_mul == 1 && a.res == nullptr && a.y != nullptr && a.y_nchw == nullptr && a.in_cs == 48 &&
         a.out_cs == p.mt * 16 && a.in_ld >= 48 && a.out_ld >= p.cout && a.cout_store % 8 == 0 && a.cout_store <= p.cout &&
         a.x_bytes > 0 && a.x_bytes < 0x80000000ull && (size_t)a.N * a.H_full * a.W_full * a.out_ld * 2 < 0x80000000ull &&
         a.H_pos == a.H_full && a.W_pos == a.W_full && a.H_pos == a.H_in / 2 && a.W_pos == a.W_in / 2 && a.H_in % 2 == 0 &&
         a.W_in % 2 == 0 && a.lo_y == -1 && a.lo_x == -1;
}

template <int GW>
static int conv48s2_launch_gw(S2Args& a, bool round_conv, hipStream_t s) {
  using S = Shape<GW>;
  a.tiles_x = (a.W_out + kTW - 1) / kTW;
  a.tiles_y = (a.H_out + kTH - 1) / kTH;
  a.div_tiles_x = make_fastdiv((uint32_t)a.tiles_x);
  a.div_tiles_xy = make_fastdiv((uint32_t)(a.tiles_x * a.tiles_y));
  a.div_ncb = make_fastdiv((uint32_t)a.n_cb);
  const long units = (long)a.N * a.tiles_x * a.tiles_y * a.n_cb;
  RTPE_REQUIRE(units < (1l << 20), "conv48s2: %ld units", units);
  long per_xcd = (units + 7) / 8;
  per_xcd = (per_xcd + a.n_cb - 1) / a.n_cb * a.n_cb;
  long G = per_xcd < 32 ? per_xcd : 32;                      // one workgroup per CU; a multiple of n_cb (1 or 2)
  G = (G + a.n_cb - 1) / a.n_cb * a.n_cb;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv48s2_kernel<GW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, S::kLds));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv48s2_kernel<GW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S::kLds));
  }
  if (round_conv) hipLaunchKernelGGL((conv48s2_kernel<GW, true>), dim3((unsigned)(8 * G)), dim3(kThreads), S::kLds, s, a);
  else hipLaunchKernelGGL((conv48s2_kernel<GW, false>), dim3((unsigned)(8 * G)), dim3(kThreads), S::kLds, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// layers[0 .. n): convs that read the SAME input view (the caller checks that) and together have 1, 2 or 4 groups of 48 output
// channels - one layer of any supported width, or several of 48 / 96 channels (three groups leave the fourth slot empty): one
// launch, the input read once
int conv48s2_launch_group(const ConvPlan* const* plans, const ConvArgs* layers, int n, hipStream_t s) {
  RTPE_REQUIRE(n >= 1 && n <= 4, "conv48s2: %d layers", n);
  S2Args a;
  memset(&a, 0, sizeof(a));
  const ConvArgs& l0 = layers[0];
  a.x = l0.x; a.x_bytes = (unsigned)l0.x_bytes;
  a.N = l0.N; a.H_in = l0.H_in; a.W_in = l0.W_in; a.in_ld = l0.in_ld; a.H_out = l0.H_full; a.W_out = l0.W_full;
  a.ablate = l0.ablate;
  int ng = 0;
  for (int i = 0; i < n; ++i) {
    const ConvPlan& p = *plans[i];
    const ConvArgs& l = layers[i];
    RTPE_REQUIRE(conv48s2_supports(p, l), "conv48s2: unsupported layer");
    RTPE_REQUIRE(l.x == l0.x && l.x_bytes == l0.x_bytes && l.N == l0.N && l.H_in == l0.H_in && l.W_in == l0.W_in && l.in_ld == l0.in_ld &&
                 l.round_conv == l0.round_conv && (n == 1 || p.cout <= 96), "conv48s2: the layers of a launch read one input");
    const int gw = n == 1 ? conv48s2_gw(p.cout) : p.cout / 48;
    RTPE_REQUIRE(ng + gw <= 4, "conv48s2: more than four groups in a launch");
    for (int k = 0; k < gw; ++k, ++ng) {
      S2Group& e = a.grp[ng];
      e.w = l.w; e.alpha = l.alpha; e.beta = l.beta; e.y = l.y;
      e.mt_pack = p.mt; e.grp0 = k; e.out_ld = l.out_ld; e.cout_store = l.cout_store; e.relu = l.relu;
      e.y_bytes = (unsigned)((size_t)l.N * l.H_full * l.W_full * l.out_ld * 2);
    }
    if (n == 1) a.n_cb = p.cout / (48 * gw);
  }
  if (n > 1) a.n_cb = 1;
  if (ng == 3) ng = 4;                                       // (the fourth slot stays empty: its wave only passes the barriers)
  switch (ng) {
    case 1: return conv48s2_launch_gw<1>(a, l0.round_conv != 0, s);
    case 2: return conv48s2_launch_gw<2>(a, l0.round_conv != 0, s);
    default: return conv48s2_launch_gw<4>(a, l0.round_conv != 0, s);
  }
}

int conv48s2_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s) {
  const ConvPlan* pp = &p;
  return conv48s2_launch_group(&pp, &a, 1, s);
}

}  // namespace rtpe
