// 3x3 stride-1 conv + BN (+residual) (+ReLU) with 96 input and 96 output channels: the BasicBlock convs of the 96-channel
// branch (pose_higher_hrnet.py:46-75; 64 launches per forward, 2.8 ms at batch 32 on the streaming kernel) with ALL 96
// output channels per workgroup (ConvTile::kind == 6).  The streaming kernel keeps the weights of one 48-cout block in LDS,
// so every input tile goes through the CU twice (once per cout block) - and the CU's memory pipeline is what bounds those
// layers (DESIGN.md section 4 "Round 4").  166 KiB of weights do not fit LDS, but they fit REGISTERS: a multiplier wave owns
// one 16-channel cout tile and keeps its 28 weight fragments (2 channel chunks x 14 k steps, the conv op's own packed plan)
// for the whole kernel - six multipliers, 112 registers each.  Structure of conv64.hip:
//   * persistent workgroups of 8 waves, one per CU, walking 8 x 16 output tiles; 6 waves multiply (one cout tile each, the
//     tile's 8 rows of 16 pixels in two passes of 4, B operands one k step ahead), 2 move data: the halo tile of the NEXT
//     tile (10 x 18 pixels x 96 channels, both 48-channel chunks of an NHWC or plane-major tensor) by LDS-DMA into the
//     other of two buffers, the residual rows of THIS tile into registers, and the rows of the PREVIOUS tile out of a
//     transpose buffer (+ residual, ReLU) as 16-byte pieces;
//   * the k order of conv_stream.hip: chunk 0's 14 k steps, then chunk 1's, [tap][channel] inside a chunk with its
//     zero-weight padding; same rounding points - bit-identical (tests/test_gpu_parity.py).
// LDS: 3 x 40,320 B of halo tiles (224 bytes per pixel: 192 + 32, pstride % 64 == 32) + 26,624 B of transpose buffer.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int kTH = 8, kTW = 16;                           // output tile: 8 rows of 16 pixels
constexpr int kHH = kTH + 2, kHW = kTW + 2;                // halo tile: 10 x 18
constexpr int kPS = 224;                                   // LDS bytes per pixel: 96 fp16 + 32
constexpr int kSlots = kPS / 16;                           // 14 16-byte slots per pixel, 12 of them data (6 per chunk)
constexpr int kRowB = kHW * kPS;                           // 4,032
constexpr int kBufBytes = kHH * kRowB;                     // 40,320
constexpr int kTileSlots = kBufBytes / 16;                 // 2,520
constexpr int kObufRow = 208;                              // 96 fp16 + 16
constexpr int kObufBytes = kTH * kTW * kObufRow;           // 26,624
constexpr int kNBuf = 3;                                   // halo tile buffers: a tile is requested two tiles ahead
constexpr int kScratch = 1024;                             // target of the no-op requests behind the last tile
constexpr int kLds = kNBuf * kBufBytes + kObufBytes + kScratch;   // 148,608
constexpr int kThreads = 512;
constexpr int kMul = 6;                                    // multiplier waves; the other 2 move data
constexpr int kMovers = 2;
constexpr int kDmaIter = (kTileSlots + 64 * kMovers - 1) / (64 * kMovers);   // 20 wave-instructions per mover and tile
constexpr int kPieces = kTH * kTW * 12 / (64 * kMovers);                     // 12 16-byte row pieces per mover lane
static_assert(kTH * kTW * 12 % (64 * kMovers) == 0, "row pieces divide evenly among the mover lanes");
constexpr int kKC = 14;                                    // k steps per 48-channel chunk (9 x 48 = 432 values, padded to 448)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// conv accumulator -> fp16 (the conv's output tensor) -> BN in fp32 -> fp16: the rounding points of the wrapper, as in
// conv_stream.hip / conv_block.hip (bn_round there)
__device__ __forceinline__ half4 bn_round96(const float4v v, const float4v al, const float4v be, bool round_conv) {
  float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  if (round_conv) {
    const half2v h0 = __builtin_convertvector(float2v{x0, x1}, half2v), h1 = __builtin_convertvector(float2v{x2, x3}, half2v);
    x0 = (float)h0[0]; x1 = (float)h0[1]; x2 = (float)h1[0]; x3 = (float)h1[1];
  }
  float r0 = __builtin_fmaf(x0, al[0], be[0]), r1 = __builtin_fmaf(x1, al[1], be[1]);
  float r2 = __builtin_fmaf(x2, al[2], be[2]), r3 = __builtin_fmaf(x3, al[3], be[3]);
  asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));   // (no fma + cast fusion: two roundings)
  const half2v o0 = __builtin_convertvector(float2v{r0, r1}, half2v), o1 = __builtin_convertvector(float2v{r2, r3}, half2v);
  return half4{o0[0], o0[1], o1[0], o1[1]};
}
}  // namespace

__global__ void __launch_bounds__(kThreads) conv96_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufs = smem;
  char* const obuf = smem + kNBuf * kBufBytes;

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

  // Two barriers per tile, passed by all 8 waves:
  //   T(u): the halo tile u has landed and the transpose buffer holds the rows of tile u - 1;
  //   O(u): the movers have read those rows out (the multipliers may overwrite the buffer with tile u's).
  if (wv >= kMul) {
    // ------------------------------------ movers: 2 waves ------------------------------------
    const int mw = wv - kMul, mt = tid - kMul * 64;          // mover wave / thread index (0..127)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    // a wave-instruction fills 64 consecutive 16-byte slots of the tile image: a pixel = 6 slots of chunk 0, 6 of chunk 1,
    // 2 of padding (padding, pixels outside the image: out-of-range offset, the bounds check writes zeros)
    auto request_tile = [&](int tt, char* buf) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      const int iy0 = py0 - 1, ix0 = px0 - 1;
      int lane_l = lane;
      asm volatile("" : "+v"(lane_l));                       // positions are recomputed per tile (20 x 4 registers are not worth holding)
#pragma unroll
      for (int k = 0; k < kDmaIter; ++k) {
        const int s0 = (mw + kMovers * k) * 64;              // first slot of this wave-instruction (uniform)
        if (s0 >= kTileSlots) break;
        const int sl = s0 + lane_l;
        const int pix = sl / kSlots, slot = sl - pix * kSlots;
        const int hy = pix / kHW, hx = pix - hy * kHW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = sl < kTileSlots && slot < 12 && (unsigned)iy < (unsigned)a.H_in && (unsigned)ix < (unsigned)a.W_in;
        const int chunk = slot >= 6 ? 1 : 0, s6 = slot - 6 * chunk;
        const long long e = (long long)((n * a.H_in + iy) * a.W_in + ix) * a.in_ld + chunk * a.in_cs + s6 * 8;
        const uint32_t voff = ok ? (uint32_t)(e * 2) : 0x80000000u;
        if (sl < kTileSlots)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + s0 * 16), 16, (int)voff, 0, 0, 0);
      }
    };
    // this lane's 12 row pieces of a tile: piece i = mt + 128 k -> pixel i / 12 of the tile, 16-byte slot i % 12 (cout
    // block slot / 6); fixed for the whole kernel: position in the tile and element offsets from the tile's first pixel
    int ppos[kPieces], po[kPieces], pr[kPieces];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
      const int i = mt + k * 64 * kMovers;
      const int pw = i / 12, slot = i - pw * 12;
      const int oyl = pw >> 4, oxl = pw & 15;
      const int cbk = slot >= 6 ? 1 : 0, s6 = slot - 6 * cbk;
      ppos[k] = (oyl << 8) | oxl;
      po[k] = (oyl * a.W_full + oxl) * a.out_ld + (int)(cbk * a.out_cs) + s6 * 8;
      pr[k] = (oyl * a.W_full + oxl) * a.res_ld + (int)(cbk * a.res_cs) + s6 * 8;
    }
    const bool use_res = a.res != nullptr;
    uint4 rres[kPieces];                                   // residual pieces of the tile whose rows leave next
#pragma unroll
    for (int k = 0; k < kPieces; ++k) rres[k] = uint4{0u, 0u, 0u, 0u};
    auto load_res = [&](int tt) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      const _Float16* rb = a.res + (((size_t)n * a.H_full + py0) * a.W_full + px0) * a.res_ld;
      const int hy = a.H_pos - py0, hx = a.W_pos - px0;
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {
        const bool ok = (ppos[k] >> 8) < hy && (ppos[k] & 255) < hx;
        rres[k] = *reinterpret_cast<const uint4*>(rb + (ok ? pr[k] : 0));     // (the tile's first pixel where the piece does not exist)
      }
    };
    auto store_rows = [&](int tt) {
      int n, py0, px0;
      tile_origin(tt, &n, &py0, &px0);
      _Float16* yb = a.y + (((size_t)n * a.H_full + py0) * a.W_full + px0) * a.out_ld;
      const int hy = a.H_pos - py0, hx = a.W_pos - px0;
      uint4 raw[kPieces];
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {                  // all reads first: one LDS round trip per tile
        const int i = mt + k * 64 * kMovers;
        raw[k] = *reinterpret_cast<const uint4*>(obuf + (i / 12) * kObufRow + (i % 12) * 16);
      }
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {
        half8 v = __builtin_bit_cast(half8, raw[k]);
        if (use_res) v = v + __builtin_bit_cast(half8, rres[k]);   // fp16 add, round-to-nearest-even = the wrapper's add
        if (a.relu) {                                      // x > 0 ? x : +0, on the sign bits
          short8 b = __builtin_bit_cast(short8, v);
          b = b & ~(b >> 15);
          v = __builtin_bit_cast(half8, b);
        }
        const bool ok = (ppos[k] >> 8) < hy && (ppos[k] & 255) < hx;
        if (ok && !(a.ablate & 2)) *reinterpret_cast<half8*>(yb + po[k]) = v;
      }
    };
    // A tile's 3.3 us are too short for a request made half a tile ahead (two buffers: 52 us per launch, the requests'
    // latency in the open): three buffers, tile u + 2 is requested during tile u.  In program order a mover issues per
    // tile: stores (<= 12), residual loads (12 or 0), tile requests (20); its wait in front of T(u) lets the youngest
    // residual loads + requests stay in flight - everything older, the requests for tile u included, has landed.
    request_tile(t0, bufs);
    if (t0 + wg_per_xcd < t_end) request_tile(t0 + wg_per_xcd, bufs + kBufBytes);
    if (use_res) load_res(t0);
    int nb = 2 % kNBuf, prev = -1;                        // buffer of the tile requested next
    bool first = true;
    for (int t = t0; t < t_end; t += wg_per_xcd) {
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (use_res) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces + kDmaIter) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaIter) : "memory");
      __syncthreads();                                    // T
      // the stores first: they gate O (the multipliers' epilogue); the requests have until the T after next
      if (prev >= 0) store_rows(prev);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are in registers / on their way
      __syncthreads();                                    // O
      if (use_res && !first) load_res(t);                 // (tile t0's were requested in front of the loop)
      if (t + 2 * wg_per_xcd < t_end && !(a.ablate & 4)) request_tile(t + 2 * wg_per_xcd, bufs + nb * kBufBytes);
      else {
        // (no tile left to request: the wait above counts on kDmaIter younger requests - issue them as no-ops)
#pragma unroll
        for (int k = 0; k < kDmaIter; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(obuf + kObufBytes), 16, (int)0x80000000u, 0, 0, 0);
      }
      prev = t;
      nb = nb + 1 == kNBuf ? 0 : nb + 1;
      first = false;
    }
    __syncthreads();                                      // the last tile's rows are in the transpose buffer
    store_rows(prev);
    return;
  }

  // ---------------------------------- multipliers: 6 waves ----------------------------------
  const int r = lane & 15, g = lane >> 4;
  const int cb = wv / 3, m = wv - 3 * cb;                  // cout tile wv = tile m of cout block cb
  uint4 a_res[2][kKC];
#pragma unroll
  for (int cci = 0; cci < 2; ++cci) {
    const uint4* wfrag = reinterpret_cast<const uint4*>(a.w) + ((size_t)(cb * 2 + cci) * kKC * 3 + m) * 64 + lane;
#pragma unroll
    for (int k = 0; k < kKC; ++k) a_res[cci][k] = wfrag[(size_t)k * 3 * 64];
  }
  const float4v al = *reinterpret_cast<const float4v*>(a.alpha + wv * 16 + g * 4);
  const float4v be = *reinterpret_cast<const float4v*>(a.beta + wv * 16 + g * 4);
  // LDS byte offset of this lane's 8 channels in k step k of a chunk (flat [tap][channel] order, conv_stream.hip's toff)
  int baddr[kKC];
#pragma unroll
  for (int k = 0; k < kKC; ++k) {
    int kk = k * 32 + g * 8;
    if (kk >= 9 * 48) kk -= 9 * 48;                        // zero-weight k padding: any finite in-tile data
    const int tap = kk / 48, c = kk - tap * 48;
    const int ty = tap / 3, tx = tap - ty * 3;
    baddr[k] = ty * kRowB + (tx + r) * kPS + c * 2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int cur = 0;
  for (int t = t0; t < t_end; t += wg_per_xcd) {
    const char* tb = bufs + cur * kBufBytes;
    __syncthreads();                                      // T
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4v acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = float4v{0.f, 0.f, 0.f, 0.f};
      if (!(a.ablate & 1)) {                              // (profiling ablations: RTPE_STREAM_ABL in diagnostic builds)
        const char* th = tb + h * 4 * kRowB;
        uint4 bf[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[0][nt] = *reinterpret_cast<const uint4*>(th + baddr[0] + nt * kRowB);
#pragma unroll
        for (int s = 0; s < 2 * kKC; ++s) {                // chunk 0's k steps, then chunk 1's
          const int cci = s / kKC, k = s - cci * kKC;
          const int cbuf = s & 1, nbuf = cbuf ^ 1;
          if (s + 1 < 2 * kKC) {
            const int c1 = (s + 1) / kKC, k1 = (s + 1) - c1 * kKC;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nbuf][nt] = *reinterpret_cast<const uint4*>(th + baddr[k1] + nt * kRowB + c1 * 96);
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_res[cci][k]), __builtin_bit_cast(half8, bf[cbuf][nt]),
                                                             acc[nt], 0, 0, 0);
          if (s + 1 < 2 * kKC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // O sits behind the first pass: by then the movers have long read the previous tile's rows out of the transpose buffer
      if (h == 0) __syncthreads();                        // O: the transpose buffer is free
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const half4 o = bn_round96(acc[nt], al, be, a.round_conv != 0);
        *reinterpret_cast<half4*>(obuf + ((h * 4 + nt) * 16 + r) * kObufRow + (wv * 16 + g * 4) * 2) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur = cur + 1 == kNBuf ? 0 : cur + 1;
  }
  __syncthreads();                                        // the last tile's rows are in the transpose buffer
}

bool conv96_supports(const ConvPlan& p) {
  return p.esize == 2 && p.dil == 1 && p.tapw == 3 && p.in_mul == 1 && p.mt == 3 && p.cc == 48 && p.n_cchunks == 2 && p.kc == kKC &&
         p.n_cb == 2 && p.cout_pad == 96;
}

size_t conv96_lds() { return kLds; }

int conv96_grid(int N, int H_pos, int W_pos) {
  const long tiles = (long)N * ((H_pos + kTH - 1) / kTH) * ((W_pos + kTW - 1) / kTW);
  const long per_xcd = (tiles + 7) / 8;
  return (int)(8 * (per_xcd < 32 ? per_xcd : 32));      // one workgroup per CU
}

int conv96_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(conv96_supports(p) && a.cin == 96 && a.cout_store == 96, "conv96: unsupported plan");
  RTPE_REQUIRE(a.y != nullptr && a.y_nchw == nullptr && a.o_mul == 1 && a.n_cls == 0, "conv96: NHWC / plane-major output only");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "conv96: input view of %zu bytes", (size_t)a.x_bytes);
  RTPE_REQUIRE((a.in_ld >= a.cin ? a.in_cs == 48 : a.in_ld == 48 && a.in_cs >= 48) && a.in_ld % 8 == 0 && a.out_ld % 8 == 0 &&
                   a.out_cs % 8 == 0 && (a.res == nullptr || (a.res_ld % 8 == 0 && a.res_cs % 8 == 0)),
               "conv96: views (in_ld %d in_cs %lld out_ld %d out_cs %lld)", a.in_ld, a.in_cs, a.out_ld, a.out_cs);
  RTPE_REQUIRE(a.th == kTH && a.tw == kTW && a.H_in == a.H_pos && a.W_in == a.W_pos, "conv96: tile %dx%d, map %dx%d -> %dx%d", a.th,
               a.tw, a.H_in, a.W_in, a.H_pos, a.W_pos);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && t.lds_bytes >= (size_t)kLds, "conv96: launch shape");
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask))
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv96_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  hipLaunchKernelGGL(conv96_kernel, dim3((unsigned)t.grid), dim3(kThreads), kLds, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
