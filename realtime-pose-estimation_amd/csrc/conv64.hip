// 3x3 stride-1 conv + BN (+ReLU) with 64 input and 64 output channels (conv2 of layer1's Bottlenecks,
// pose_higher_hrnet.py:78-116): the pipelined kernel for 64-channel chunks that the one-workgroup-per-tile kernel is not
// (ConvTile::kind == 5).  There a workgroup stages its halo tile, waits, multiplies, stores - staging and the epilogue
// were 36 % + 28 % of a wave's cycles on this layer (tools/tile_stamps.sh) and the weight fragments came from L2 in every
// k step.  Here, on the structure of stem_fused.hip's second conv:
//   * persistent workgroups of 8 waves, one per CU, walking 16 x 16 output tiles: 4 waves multiply, 4 move data - the
//     halo tile of the NEXT tile (18 x 18 pixels x 160 bytes) by LDS-DMA into the other of two buffers, and the rows of
//     the PREVIOUS tile out of a transpose buffer as 16-byte NHWC pieces (plain stores: write-through ones measured 12 %
//     slower here) - while this tile is multiplied;
//   * multiplier w owns two cout tiles - their 36 weight fragments (the conv op's own packed plan: mt 4, one 64-channel
//     chunk) stay in registers for the whole kernel - and 8 of the tile's 16 rows of 16 pixels in two passes of 4: 4
//     B-operand reads (one k step ahead) and 8 MFMAs per k step, same k order as conv_mfma.hip (bit-identical,
//     tests/test_gpu_parity.py).
// LDS: 2 x 51,840 B of halo tiles + 36,864 B of transpose buffer = 137 KiB.
// Measured at batch 32 (160 x 160 maps, tools/conv64_abl.sh): 63 us against 88-94 us on the one-workgroup-per-tile kernel
// (HBM floor of the layer ~47 us); with parts switched off: no stores 56 us, no stores and no tile requests 55 us, nothing but
// barriers and epilogue arithmetic 22 us.  Steps on the way: all 8 waves multiplying AND moving data 116 us (a wave that
// issues memory instructions issues no MFMAs meanwhile: requests 18 + stores 23 + k loops 50 us added up); 4 multipliers + 4
// movers 98 us; the movers' eight LDS reads of a tile before their first store, plain stores 81 us; two cout tiles per
// multiplier (half the B-operand reads) 82 us - no change: the k loops were not bound by the LDS array; the movers' stores
// BEFORE their requests for the next tile (the stores gate barrier O, the requests have until the next T) 63 us.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int kT = 16;                                     // output tile: 16 x 16
constexpr int kHW = kT + 2;                                // halo tile: 18 x 18
constexpr int kPS = 160;                                   // LDS bytes per pixel: 64 fp16 + 32 (pstride % 64 == 32)
constexpr int kSlots = kPS / 16;                           // 10 16-byte slots per pixel, 8 of them data
constexpr int kRowB = kHW * kPS;                           // 2,880
constexpr int kBufBytes = kHW * kRowB;                     // 51,840
constexpr int kTileSlots = kBufBytes / 16;                 // 3,240
constexpr int kObufRow = 144;
constexpr int kObufBytes = kT * kT * kObufRow;             // 36,864
constexpr int kLds = 2 * kBufBytes + kObufBytes;           // 140,544
constexpr int kThreads = 512;
constexpr int kMulWaves = 4;                               // waves that multiply; the other 4 move data
constexpr int kDmaIter = (kTileSlots + 255) / 256;         // 13 wave-instructions per mover wave and tile
static_assert(kBufBytes % 1024 == 640, "the last DMA instruction of a tile is partial");

__device__ __forceinline__ float round16(float v) { return (float)(_Float16)v; }
typedef __attribute__((address_space(3))) void* lds_ptr_t;
}  // namespace

__global__ void __launch_bounds__(kThreads) conv64_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufs = smem;
  char* const obuf = smem + 2 * kBufBytes;

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
    *n = (int)nn; *py0 = (int)ty * kT; *px0 = (int)(rem - ty * tiles_x) * kT;
  };

  // Two barriers per tile, passed by all 8 waves:
  //   T(u): the halo tile u has landed and the transpose buffer holds the rows of tile u - 1;
  //   O(u): the movers have read those rows out (the multipliers may overwrite the buffer with tile u's).
  if (wv >= kMulWaves) {
    // ------------------------------------ movers: 4 waves ------------------------------------
    // request the halo tile of the next tile by LDS-DMA, store the rows of the previous one: no memory instruction is
    // issued by a wave that multiplies (in the first form of this kernel all 8 waves did both: the requests, the stores
    // and the k loops of a tile added up - 18 + 23 + 50 us of a 116 us layer)
    const int mw = wv - kMulWaves, mt = tid - kMulWaves * 64;      // mover wave / thread index (0..255)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    // a wave-instruction fills 64 consecutive 16-byte slots of the tile image (a pixel = 8 data slots + 2 of padding;
    // padding, pixels outside the image: out-of-range offset, the bounds check writes zeros)
    auto request_tile = [&](int tt, char* buf) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      const int iy0 = py0 - 1, ix0 = px0 - 1;
#pragma unroll
      for (int k = 0; k < kDmaIter; ++k) {
        const int s0 = (mw + 4 * k) * 64;                     // first slot of this wave-instruction (uniform)
        if (s0 >= kTileSlots) break;
        const int sl = s0 + lane;
        const int pix = sl / kSlots, slot = sl - pix * kSlots;
        const int hy = pix / kHW, hx = pix - hy * kHW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = sl < kTileSlots && slot < 8 && (unsigned)iy < (unsigned)a.H_in && (unsigned)ix < (unsigned)a.W_in;
        const uint32_t voff = ok ? (uint32_t)((((n * a.H_in + iy) * a.W_in + ix) * a.in_ld + slot * 8) * 2) : 0x80000000u;
        if (sl < kTileSlots)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + s0 * 16), 16, (int)voff, 0, 0, 0);
      }
    };
    // NHWC rows of a finished tile from the transpose buffer: 16 bytes per lane, 8 lanes per pixel
    auto store_rows = [&](int tt) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      constexpr int kPieces = kT * kT * 8 / 256;           // 8 per mover lane
      uint4 raw[kPieces];
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {                  // all reads first: one LDS round trip per tile, not eight
        const int i = mt + k * 256;
        raw[k] = *reinterpret_cast<const uint4*>(obuf + (i >> 3) * kObufRow + (i & 7) * 16);
      }
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {
        const int i = mt + k * 256;
        const int pw = i >> 3, slot = i & 7;
        const int oy = py0 + (pw >> 4), ox = px0 + (pw & 15);
        _Float16 hv[8];
        __builtin_memcpy(hv, &raw[k], 16);
        if (a.relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[j] = hv[j] > (_Float16)0.f ? hv[j] : (_Float16)0.f;
        }
        uint4 out;
        __builtin_memcpy(&out, hv, 16);
        if (oy < a.H_pos && ox < a.W_pos && slot * 8 < a.cout_store) {
          _Float16* dst = a.y + (((size_t)n * a.H_full + oy) * a.W_full + ox) * a.out_ld + slot * 8;
          if (a.ablate & 8) store16_wt(dst, out);            // (profiling: write-through instead of plain stores)
          else *reinterpret_cast<uint4*>(dst) = out;
        }
      }
    };
    request_tile(t0, bufs);
    int cur = 0, prev = -1;
    for (int t = t0; t < t_end; t += wg_per_xcd) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of tile t have landed (its older stores too)
      __syncthreads();                                    // T
      // the stores first: they gate O (the multipliers' epilogue); the requests for the next tile have until the next T
      if (prev >= 0 && !(a.ablate & 2)) store_rows(prev);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are in registers / on their way
      __syncthreads();                                    // O
      if (t + wg_per_xcd < t_end && !(a.ablate & 4)) request_tile(t + wg_per_xcd, bufs + (cur ^ 1) * kBufBytes);
      prev = t;
      cur ^= 1;
    }
    __syncthreads();                                      // the last tile's rows are in the transpose buffer
    if (!(a.ablate & 2)) store_rows(prev);
    return;
  }

  // ---------------------------------- multipliers: 4 waves ----------------------------------
  // wave w owns TWO cout tiles (2 * (w & 1), + 1: their 36 weight fragments stay in registers) and 8 of the tile's 16 rows of
  // 16 pixels (half w >> 1), in two passes of 4 rows: a B fragment feeds two MFMAs, so the LDS array serves half the
  // operand reads of one cout tile per wave (which ran the k loops at the LDS array's rate: 57 us of an 81 us layer);
  // the B operands of the next k step are requested before the MFMAs of the current one
  const int r = lane & 15, g = lane >> 4;
  const int m0 = 2 * (wv & 1), hf = wv >> 1;
  uint4 a_res[2][18];
#pragma unroll
  for (int mm = 0; mm < 2; ++mm) {
    const uint4* wfrag = reinterpret_cast<const uint4*>(a.w) + (m0 + mm) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 18; ++k) a_res[mm][k] = wfrag[(size_t)k * 4 * 64];
  }
  float4v al[2], be[2];
#pragma unroll
  for (int mm = 0; mm < 2; ++mm) {
    al[mm] = *reinterpret_cast<const float4v*>(a.alpha + (m0 + mm) * 16 + g * 4);
    be[mm] = *reinterpret_cast<const float4v*>(a.beta + (m0 + mm) * 16 + g * 4);
  }
  const int bbase = (hf * 8) * kRowB + r * kPS + g * 16;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (int t = t0; t < t_end; t += wg_per_xcd) {
    const char* tb = bufs + cur * kBufBytes + bbase;
    __syncthreads();                                      // T
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4v acc[2][4];                                  // [cout tile][row]
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mm][nt] = float4v{0.f, 0.f, 0.f, 0.f};
      if (!(a.ablate & 1)) {                              // (profiling ablations: RTPE_STREAM_ABL in diagnostic builds)
        const char* th = tb + h * 4 * kRowB;
        uint4 bf[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[0][nt] = *reinterpret_cast<const uint4*>(th + nt * kRowB);
#pragma unroll
        for (int kci = 0; kci < 18; ++kci) {
          const int cb = kci & 1, nb = cb ^ 1;
          if (kci + 1 < 18) {
            const int tap = (kci + 1) >> 1, ty = tap / 3, tx = tap - ty * 3;
            const int ko = ty * kRowB + tx * kPS + ((kci + 1) & 1) * 64;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nb][nt] = *reinterpret_cast<const uint4*>(th + nt * kRowB + ko);
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
              acc[mm][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_res[mm][kci]),
                                                                   __builtin_bit_cast(half8, bf[cb][nt]), acc[mm][nt], 0, 0, 0);
          if (kci + 1 < 18) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // O sits behind the first pass: by then the movers have long read the previous tile's rows out of the transpose
      // buffer, and the accumulators of a pass can leave right away (both passes' would not fit the registers)
      if (h == 0) __syncthreads();                        // O: the transpose buffer is free
      // ---- BN (+ the conv output's own rounding) -> transpose buffer ----
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          // (BatchNorm stays here: done by the movers on the row pieces they store - built and measured - it delays their
          // stores, which gate O: 63 -> 71 us)
          _Float16 o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = acc[mm][nt][j];
            if (a.round_conv) x = round16(x);
            float tt = __builtin_fmaf(x, al[mm][j], be[mm][j]);
            asm volatile("" : "+v"(tt));                   // (no fma + cast fusion: two roundings, conv_mfma.hip)
            o[j] = (_Float16)tt;
          }
          unsigned long long raw;
          __builtin_memcpy(&raw, o, 8);
          *reinterpret_cast<unsigned long long*>(obuf + ((hf * 8 + h * 4 + nt) * 16 + r) * kObufRow + ((m0 + mm) * 16 + g * 4) * 2) = raw;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur ^= 1;
  }
  __syncthreads();                                        // the last tile's rows are in the transpose buffer
}

bool conv64_supports(const ConvPlan& p) {
  // a real 64 -> 64 layer only: cin 49..63 / cout 49..63 give the same padded plan (cc 64, one chunk, cout_pad 64) but
  // rows that this kernel's fixed 128-byte pixel pieces do not describe - they stay on the one-workgroup-per-tile kernel
  return p.esize == 2 && p.dil == 1 && p.tapw == 3 && p.in_mul == 1 && p.mt == 4 && p.cc == 64 && p.n_cchunks == 1 && p.kc == 18 &&
         p.n_cb == 1 && p.cout_pad == 64 && p.cin == 64 && p.cout == 64;
}

size_t conv64_lds() { return kLds; }

int conv64_grid(int N, int H_pos, int W_pos) {
  const long tiles = (long)N * ((H_pos + kT - 1) / kT) * ((W_pos + kT - 1) / kT);
  const long per_xcd = (tiles + 7) / 8;
  return (int)(8 * (per_xcd < 32 ? per_xcd : 32));      // one workgroup per CU
}

int conv64_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(conv64_supports(p) && a.cin == 64, "conv64: unsupported plan");
  RTPE_REQUIRE(a.res == nullptr && a.y != nullptr && a.y_nchw == nullptr && a.o_mul == 1 && a.n_cls == 0 && a.in_cs == p.cc &&
               a.out_cs == 64, "conv64: NHWC in and out, no residual");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "conv64: input view of %zu bytes", (size_t)a.x_bytes);
  RTPE_REQUIRE(a.th == kT && a.tw == kT && a.H_in == a.H_pos && a.W_in == a.W_pos && a.out_ld >= 64 && a.cout_store % 8 == 0,
               "conv64: tile %dx%d, map %dx%d -> %dx%d", a.th, a.tw, a.H_in, a.W_in, a.H_pos, a.W_pos);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && t.lds_bytes >= (size_t)kLds, "conv64: launch shape");
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask))
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  hipLaunchKernelGGL(conv64_kernel, dim3((unsigned)t.grid), dim3(kThreads), kLds, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
