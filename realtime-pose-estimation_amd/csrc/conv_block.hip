// Fused BasicBlock for the 48-channel branches of the w48 network (pose_higher_hrnet.py:46-75):
//     y = relu( bn2(conv3x3(relu(bn1(conv3x3(x))))) + x )
// in ONE kernel.  The two convolutions of a 48-channel block at 160x160 / 320x320 are the layers
// that sit at the chip's mixed read/write bandwidth ceiling when run one after the other (conv1:
// read x, write mid; conv2: read mid, read x as the residual, write y = 5 tensor passes).  Here
// the intermediate tile never leaves LDS and the residual is the input tile that is already
// there: read x once (+ halo), write y once.
//
// Same math as conv_stream.hip / conv_mfma.hip (same k order, same packed weights, same rounding
// points: conv -> fp16, BN -> fp16, ReLU, (add -> fp16)), so the result is bit-identical to the
// two separate launches.  Per unit = 6x32 output pixels of one image:
//   * x halo tile 10x36 pixels, LDS-DMA'd by two loader waves one unit ahead (2 buffers);
//   * conv1 on the 8x34 region the second conv needs (272 pixels in 5 x 16-pixel tiles per wave;
//     48 slots idle), BN1 + ReLU in registers, written as fp16 rows into the mid tile in LDS
//     (zeros outside the image = conv2's padding);
//   * conv2 on the 6x32 outputs (3 tiles per wave) straight from the mid tile;
//   * BN2, transposition through LDS, + x (16-byte row pieces of the halo tile), ReLU, store;
//   * the weights of both convs (4 half-stage sets of 21 KiB) stream through the 3-slot LDS ring
//     of conv_stream.hip, two half stages ahead, from L2.
// LDS: 63 KiB ring + 2 x 33.75 KiB x tiles + 25.5 KiB mid tile = 156 KiB, one workgroup per CU.
#include <type_traits>

#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// conv accumulator -> fp16 (the conv's output tensor) -> BN in fp32 -> fp16, the wrapper's rounding points.  The
// fp16 values feed the fma directly (v_fma_mix_f32 converts its first operand on the way in: the same fp32 fma on
// the same operands as convert + v_pk_fma_f32, one instruction less per pair); the empty asm keeps the compiler from
// folding the final conversion into v_fma_mixlo_f16, which would round the exact a*b+c once instead of twice.
__device__ __forceinline__ half4 bn_round(const float4v v, const float4v al, const float4v be) {
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
typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTH = 6, kTW = 32;                 // output tile
constexpr int kMH = kTH + 2, kMW = kTW + 2;      // conv1 region = mid tile (8 x 34)
constexpr int kXH = kTH + 4, kXW = kTW + 4;      // x halo tile (10 x 36)
constexpr int kPS = 96;                          // LDS bytes per pixel (48 fp16 channels)
constexpr int kMT = 3;                           // 48 output channels = 3 MFMA row tiles
constexpr int kNT1 = 5, kNT2 = 3;                // 16-pixel tiles per wave: conv1 (320 slots for 272), conv2 (192)
constexpr int kWaves = 4, kLoad = 3;
constexpr int kKH = 7;                           // k-steps per half stage
constexpr int kWSlot = kMT * kKH * 1024;         // 21,504 B
constexpr int kXBytes = kXH * kXW * kPS;         // 34,560 B
constexpr int kMidBytes = kMH * kMW * kPS;       // 26,112 B
constexpr int kRowB = kMT * 32 + 16;             // transposed output row in LDS
constexpr int kNIT = (kNT2 * 16 * 6 + 63) / 64;  // 16-byte output row pieces per lane in epilogue B
constexpr int kLds = 3 * kWSlot + 2 * kXBytes + kMidBytes;

#define RTPE_BBARRIER()                         \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

}  // namespace

struct BlockArgs {
  const _Float16* x;
  _Float16* y;
  const _Float16* w1;       // packed fragments [14 k-steps][3][64 lanes][16 B]
  const _Float16* w2;
  const float* ab1;         // alpha[48], beta[48]
  const float* ab2;
  int N, H, W, in_ld, out_ld;
  int tiles_x, tiles_y;
  FastDiv div_tiles_x, div_tiles_xy;
  int x_bytes;
  int ablate;               // -DRTPE_DIAG builds only (RTPE_BLOCK_ABL): 1 no x-tile reloads, 2 no epilogue A, 4 no output
                            // stores, 8 no k-loops - wrong results, timing only
};

__global__ void __launch_bounds__((kWaves + kLoad) * 64) conv_block_kernel(const BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wring = smem;
  char* const xt = smem + 3 * kWSlot;
  char* const mid = xt + 2 * kXBytes;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD x (= blockIdx % 8) owns a CONTIGUOUS eighth of the tile list (row-major tiles of consecutive
  // images) and its G workgroups walk it with stride G: tiles that share halo rows (t and t +- tiles_x)
  // are multiplied on the same XCD at about the same time, so the overlap comes from that XCD's L2
  const int n_tiles = a.N * a.tiles_x * a.tiles_y;
  const int G = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3);
  const int per_xcd = (n_tiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int tiles_xcd = min(per_xcd, n_tiles - t_begin);               // may be <= 0 for the last XCDs
  const int U = jw < tiles_xcd ? (tiles_xcd - jw + G - 1) / G : 0;      // units of this workgroup
  if (U == 0) return;
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  auto unit_origin = [&](int u, uint32_t* n, int* py0, int* px0) __attribute__((always_inline)) {
    uint32_t t = (uint32_t)(t_begin + jw + u * G);
    *n = fdiv(t, a.div_tiles_xy);
    t -= *n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    *py0 = (int)tyi * kTH;
    *px0 = (int)(t - tyi * a.tiles_x) * kTW;
  };

  if (wv == kWaves) {
    // ----------------------------- weight loader -----------------------------
    // half stage q = 4 u + i : i = 0,1 -> conv1 halves, i = 2,3 -> conv2 halves; ring slot q % 3
    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w1), 0, 2 * kWSlot, 0x00020000);
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w2), 0, 2 * kWSlot, 0x00020000);
    const int voff = lane * 16;
    auto issue = [&](int q) __attribute__((always_inline)) {
      const int i = q & 3;
      char* dst = wring + (q % 3) * kWSlot;
      const int src = (i & 1) * kWSlot;
      if (i < 2) {
#pragma unroll
        for (int p = 0; p < kMT * kKH; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(dst + p * 1024), 16, voff, src + p * 1024, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < kMT * kKH; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr_t)(dst + p * 1024), 16, voff, src + p * 1024, 0, 0);
      }
    };
    const int Q = 4 * U;
    issue(0);
    issue(1);
    for (int q = 0; q < Q; ++q) {
      // half q has landed (only the most recent half may still be in flight), hand it over at the
      // barrier that starts it (M, H1, A2, H2 of the unit), then request half q + 2
      if (q + 1 < Q) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMT * kKH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RTPE_BBARRIER();
      if (q + 2 < Q) issue(q + 2);
      if ((q & 3) == 3) RTPE_BBARRIER();                 // E2 of the unit
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  if (wv > kWaves) {
    // ------------------------------ tile loaders ------------------------------
    const int jl = wv - kWaves - 1;                      // rows [0,5) or [5,10) of every x tile
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, a.x_bytes, 0x00020000);
    constexpr int rowslots = kXW * 6;                    // 216 16-byte slots per halo row
    auto issue = [&](int u) __attribute__((always_inline)) {
      uint32_t n;
      int py0, px0;
      unit_origin(u, &n, &py0, &px0);
      const int iy0 = py0 - 2, ix0 = px0 - 2;
      uint32_t voff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = k * 64 + lane;
        const int hx = q / 6, sl = q - hx * 6;
        const int ix = ix0 + hx;
        voff[k] = (unsigned)ix < (unsigned)a.W ? (uint32_t)(ix * a.in_ld + sl * 8) * 2u : 0x80000000u;
      }
      char* buf = xt + (u & 1) * kXBytes;
      const int img_row0 = (int)n * a.H;
      for (int r = jl * (kXH / 2); r < (jl + 1) * (kXH / 2); ++r) {
        const int iy = iy0 + r;
        const bool row_ok = (unsigned)iy < (unsigned)a.H;
        const int soff = row_ok ? (img_row0 + iy) * a.W * a.in_ld * 2 : 0;
        char* dst = buf + r * (kXW * kPS);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k * 64 + lane < rowslots)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + k * 1024), 16,
                                                     (int)(row_ok ? voff[k] : 0x80000000u), soff, 0, 0);
      }
    };
    issue(0);
    for (int u = 0; u < U; ++u) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this loader's rows of tile u have landed
      RTPE_BBARRIER();                                   // M: x buffer (u+1)&1 is free (its last reader was epilogue B of u-1)
      if (u + 1 < U) issue(u + 1);
      RTPE_BBARRIER();                                   // H1
      RTPE_BBARRIER();                                   // A2
      RTPE_BBARRIER();                                   // H2
      RTPE_BBARRIER();                                   // E2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------- MFMA waves --------------------------------
  const int r = lane & 15, g = lane >> 4;
  // byte offset of this lane group's 8 channels in k-step k: conv1 walks the x tile (36 pixels
  // per row), conv2 the mid tile (34 pixels per row)
  int toff1[2 * kKH], toff2[2 * kKH];
#pragma unroll
  for (int k = 0; k < 2 * kKH; ++k) {
    int kk = k * 32 + g * 8;
    if (kk >= 9 * 48) kk -= 9 * 48;                      // zero-weight k padding: any finite in-tile data
    const int tap = kk / 48, c = kk - tap * 48;
    const int ty = tap / 3, tx = tap - ty * 3;
    toff1[k] = (ty * kXW + tx) * kPS + c * 2;
    toff2[k] = (ty * kMW + tx) * kPS + c * 2;
  }
  int pix1[kNT1], pix2[kNT2];
#pragma unroll
  for (int nt = 0; nt < kNT1; ++nt) {
    int p = (wv * kNT1 + nt) * 16 + r;
    p = p < kMH * kMW ? p : kMH * kMW - 1;               // idle slots recompute the last pixel
    const int my = p / kMW, mx = p - my * kMW;
    pix1[nt] = (my * kXW + mx) * kPS;
  }
#pragma unroll
  for (int nt = 0; nt < kNT2; ++nt) {
    const int p = (wv * kNT2 + nt) * 16 + r;
    pix2[nt] = ((p >> 5) * kMW + (p & 31)) * kPS;
  }
  float4v al1[kMT], be1[kMT], al2[kMT], be2[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int c4 = m * 16 + g * 4;
    al1[m] = *reinterpret_cast<const float4v*>(a.ab1 + c4);
    be1[m] = *reinterpret_cast<const float4v*>(a.ab1 + 48 + c4);
    al2[m] = *reinterpret_cast<const float4v*>(a.ab2 + c4);
    be2[m] = *reinterpret_cast<const float4v*>(a.ab2 + 48 + c4);
  }
#pragma unroll
  for (int m = 0; m < kMT; ++m) asm volatile("" ::"v"(al1[m]), "v"(be1[m]), "v"(al2[m]), "v"(be2[m]));

  // epilogue B, 16-byte row piece `it` of this lane (fixed for the whole kernel): where it is in the
  // wave's transposed slab (eoff) and in the x tile (xoff, the residual), and its position in the output
  // tile (epos = row << 16 | column << 8 | byte offset in the pixel; a piece that does not exist gets a
  // row no tile reaches)
  int eoff[kNIT], xoff[kNIT], epos[kNIT];
#pragma unroll
  for (int it = 0; it < kNIT; ++it) {
    int c = it * 64 + lane;
    const bool exists = c < kNT2 * 16 * 6;
    c = exists ? c : 0;
    const int pw = c / 6, slot = c - pw * 6;
    const int p = wv * kNT2 * 16 + pw;
    const int oy = p >> 5, ox = p & 31;
    eoff[it] = pw * kRowB + slot * 16;
    xoff[it] = ((oy + 2) * kXW + ox + 2) * kPS + slot * 16;
    epos[it] = ((exists ? oy : 0x7fff) << 16) | (ox << 8) | (slot * 16);
  }
  int wsel = 0;                                          // ring slot of the next half stage (q % 3)
  for (int u = 0; u < U; ++u) {
    uint32_t n;
    int py0, px0;
    unit_origin(u, &n, &py0, &px0);
    const char* xb = xt + (u & 1) * kXBytes;

    // ------------------------------- conv1 -------------------------------
    float4v acc1[kMT][kNT1];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int nt = 0; nt < kNT1; ++nt) acc1[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      RTPE_BBARRIER();                                   // M (x tile + half 0) / H1 (half 1)
      const char* wl = wring + wsel * kWSlot + lane * 16;
      wsel = wsel == 2 ? 0 : wsel + 1;
      half8 af[2][kMT], bf[2][kNT1];
#pragma unroll
      for (int m = 0; m < kMT; ++m) af[0][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
#pragma unroll
      for (int nt = 0; nt < kNT1; ++nt) bf[0][nt] = *reinterpret_cast<const half8*>(xb + pix1[nt] + toff1[h * kKH]);
#pragma unroll
      for (int kk = 0; kk < kKH; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < kKH) {
#pragma unroll
          for (int m = 0; m < kMT; ++m) af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 1) * kMT + m) * 1024);
#pragma unroll
          for (int nt = 0; nt < kNT1; ++nt)
            bf[nxt][nt] = *reinterpret_cast<const half8*>(xb + pix1[nt] + toff1[h * kKH + kk + 1]);
        }
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
          for (int nt = 0; nt < kNT1; ++nt)
            acc1[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc1[m][nt], 0, 0, 0);
        if (kk + 1 < kKH) {
#pragma unroll
          for (int i = 0; i < (kMT + kNT1 + 3) / 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue A: BN1 + ReLU -> fp16 rows of the mid tile (the unfused path's HBM tensor) ----
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int re = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
      for (int nt = 0; nt < kNT1; ++nt) {
        const int p = (wv * kNT1 + nt) * 16 + re;
        const int my = p / kMW, mx = p - my * kMW;
        const int iy = py0 - 1 + my, ix = px0 - 1 + mx;
        const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const half4 o_bn = bn_round(acc1[m][nt], al1[m], be1[m]);
          short4v b = __builtin_bit_cast(short4v, o_bn);
          b = b & ~(b >> 15);                                                                // ReLU on the sign bits
          if (!inside) b = b ^ b;                                                            // conv2's zero padding
          if (p < kMH * kMW)
            *reinterpret_cast<short4v*>(mid + p * kPS + m * 32 + ge * 8) = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ------------------------------- conv2 -------------------------------
    float4v acc2[kMT][kNT2];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int nt = 0; nt < kNT2; ++nt) acc2[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      RTPE_BBARRIER();                                   // A2 (mid tile complete + half 2) / H2 (half 3)
      const char* wl = wring + wsel * kWSlot + lane * 16;
      wsel = wsel == 2 ? 0 : wsel + 1;
      half8 af[2][kMT], bf[2][kNT2];
#pragma unroll
      for (int m = 0; m < kMT; ++m) af[0][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
#pragma unroll
      for (int nt = 0; nt < kNT2; ++nt) bf[0][nt] = *reinterpret_cast<const half8*>(mid + pix2[nt] + toff2[h * kKH]);
#pragma unroll
      for (int kk = 0; kk < kKH; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < kKH) {
#pragma unroll
          for (int m = 0; m < kMT; ++m) af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 1) * kMT + m) * 1024);
#pragma unroll
          for (int nt = 0; nt < kNT2; ++nt)
            bf[nxt][nt] = *reinterpret_cast<const half8*>(mid + pix2[nt] + toff2[h * kKH + kk + 1]);
        }
#pragma unroll
        for (int m = 0; m < kMT; ++m)
#pragma unroll
          for (int nt = 0; nt < kNT2; ++nt)
            acc2[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc2[m][nt], 0, 0, 0);
        if (kk + 1 < kKH) {
#pragma unroll
          for (int i = 0; i < (kMT + kNT2 + 3) / 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue B: BN2, transposed through the (now free) mid tile, + x, ReLU, store ----
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int re = lane_e & 15, ge = lane_e >> 4;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RTPE_BBARRIER();                                   // E2: every wave is done reading the mid tile
      char* obuf = mid + wv * (kNT2 * 16 * kRowB);
#pragma unroll
      for (int nt = 0; nt < kNT2; ++nt)
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const half4 o_bn = bn_round(acc2[m][nt], al2[m], be2[m]);
          *reinterpret_cast<half4*>(obuf + (nt * 16 + re) * kRowB + m * 32 + ge * 8) = o_bn;
        }
      half8 ov[kNIT], rv[kNIT];
#pragma unroll
      for (int it = 0; it < kNIT; ++it) {
        ov[it] = *reinterpret_cast<const half8*>(obuf + eoff[it]);
        rv[it] = *reinterpret_cast<const half8*>(xb + xoff[it]);                            // the block input
      }
      // scalar base of the unit + 32-bit lane offset (two 24-bit multiply-adds per piece)
      const int hy = a.H - py0, hx = a.W - px0;
      char* const yb = reinterpret_cast<char*>(a.y + (((size_t)n * a.H + py0) * a.W + px0) * a.out_ld);
      const uint32_t ld2 = (uint32_t)a.out_ld * 2u, row_pix = (uint32_t)a.W & 0xffffffu;
#pragma unroll
      for (int it = 0; it < kNIT; ++it) {
        int e = epos[it];
        asm volatile("" : "+v"(e));                      // lane-only math must not be hoisted out of the unit loop
        half8 v = ov[it] + rv[it];                       // fp16 add, round-to-nearest-even = the wrapper's add
        short8 b = __builtin_bit_cast(short8, v);
        b = b & ~(b >> 15);
        if ((e >> 16) < hy && ((e >> 8) & 255) < hx) {
          const uint32_t pix = __umul24((uint32_t)e >> 16, row_pix) + (((uint32_t)e >> 8) & 255u);
          *reinterpret_cast<short8*>(yb + __umul24(pix, ld2) + ((uint32_t)e & 255u)) = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Variant with RESIDENT weights (the default): the weight fragments of both convs (4 half-stage sets,
// 84 KiB) are loaded into LDS once per workgroup instead of streaming through a 3-slot ring for every
// unit (84 KiB of L2 -> LDS traffic per unit against 34 KiB for the x tile, and two of the five barriers
// of a unit only handed weight slots over).  LDS then holds ONE x tile; to keep prefetching, each MFMA
// wave takes its residual row pieces of the tile into registers before conv1, so the tile buffer is
// free as soon as conv1's last k-step has read it and the next tile streams in underneath conv2 and
// the output epilogue.  All three loader waves move x rows.  Three barriers per unit:
//   M  : x tile u (and, for u = 0, the weights) has landed;
//   A2 : the mid tile is complete - and the x buffer is free: tile u+1 is requested;
//   E2 : every wave is done reading the mid tile (the output transposition reuses it).
// Same k order, same rounding points: bit-identical to the ring variant and to the two-launch path.
// LDS: 84 KiB weights + 33.75 KiB x tile + 25.5 KiB mid tile = 143.25 KiB.
namespace {
constexpr int kLdsRW = 4 * kWSlot + kXBytes + kMidBytes;
}

__global__ void __launch_bounds__((kWaves + kLoad) * 64) conv_block_rw_kernel(const BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wts = smem;                                  // [conv1 half 0][conv1 half 1][conv2 half 0][conv2 half 1]
  char* const xb = smem + 4 * kWSlot;
  char* const mid = xb + kXBytes;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int n_tiles = a.N * a.tiles_x * a.tiles_y;
  const int G = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3);
  const int per_xcd = (n_tiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int tiles_xcd = min(per_xcd, n_tiles - t_begin);
  const int U = jw < tiles_xcd ? (tiles_xcd - jw + G - 1) / G : 0;
  if (U == 0) return;
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  auto unit_origin = [&](int u, uint32_t* n, int* py0, int* px0) __attribute__((always_inline)) {
    uint32_t t = (uint32_t)(t_begin + jw + u * G);
    *n = fdiv(t, a.div_tiles_xy);
    t -= *n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    *py0 = (int)tyi * kTH;
    *px0 = (int)(t - tyi * a.tiles_x) * kTW;
  };

  if (wv >= kWaves) {
    // ------------------------------- loader waves -------------------------------
    const int jl = wv - kWaves;                          // rows [0,4), [4,7), [7,10) of every x tile
    const int r0 = jl == 0 ? 0 : jl == 1 ? 4 : 7, r1 = jl == 0 ? 4 : jl == 1 ? 7 : 10;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, a.x_bytes, 0x00020000);
    constexpr int rowslots = kXW * 6;
    auto issue = [&](int u) __attribute__((always_inline)) {
      uint32_t n;
      int py0, px0;
      unit_origin(u, &n, &py0, &px0);
      const int iy0 = py0 - 2, ix0 = px0 - 2;
      uint32_t voff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = k * 64 + lane;
        const int hx = q / 6, sl = q - hx * 6;
        const int ix = ix0 + hx;
        voff[k] = (unsigned)ix < (unsigned)a.W ? (uint32_t)(ix * a.in_ld + sl * 8) * 2u : 0x80000000u;
      }
      const int img_row0 = (int)n * a.H;
      for (int r = r0; r < r1; ++r) {
        const int iy = iy0 + r;
        const bool row_ok = (unsigned)iy < (unsigned)a.H;
        const int soff = row_ok ? (img_row0 + iy) * a.W * a.in_ld * 2 : 0;
        char* dst = xb + r * (kXW * kPS);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k * 64 + lane < rowslots)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + k * 1024), 16,
                                                     (int)(row_ok ? voff[k] : 0x80000000u), soff, 0, 0);
      }
    };
    // the weights, once: conv1 by loader 0, conv2 by loader 1 (42 KiB each); loader 2 starts on the tile
    if (jl < 2) {
      __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(jl == 0 ? a.w1 : a.w2), 0,
                                                                    2 * kWSlot, 0x00020000);
      char* dst = wts + jl * 2 * kWSlot;
      const int voff = lane * 16;
      for (int p = 0; p < 2 * kMT * kKH; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + p * 1024), 16, voff, p * 1024, 0, 0);
    }
    issue(0);
    for (int u = 0; u < U; ++u) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this loader's rows of tile u (and weights) have landed
      RTPE_BBARRIER();                                   // M
      RTPE_BBARRIER();                                   // A2: conv1 has read the tile, residual pieces are in registers
      if (u + 1 < U && !(a.ablate & 1)) issue(u + 1);
      RTPE_BBARRIER();                                   // E2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------- MFMA waves --------------------------------
  const int r = lane & 15, g = lane >> 4;
  // LDS address of the B operand of k-step k = a per-lane base + a compile-time immediate (koff below): no
  // address arithmetic in the k-loops.  k = 32 k-values of the flat [tap][48 channels] order, 8 per lane
  // group g; 48 = 32 + 16, so the pattern repeats every 3 k-steps: for k % 3 = 0 / 2 all four groups sit in
  // one tap (offset = tap offset + {0, 32} + 16 g); for k % 3 = 1 groups 2, 3 have moved on to the next tap,
  // which is the next pixel of the row (96 bytes on: the same formula) except at the end of a tap row (k = 4:
  // + (row pitch - 3 pixels)) and in the zero-weight k padding (k = 13: any finite in-tile data - tap 0).
  // Three base registers per pixel tile cover the three cases.
  int b1a[kNT1], b1b[kNT1], b1c[kNT1], b2a[kNT2], b2b[kNT2], b2c[kNT2];
  {
    const int hi = g >= 2 ? 1 : 0;
#pragma unroll
    for (int nt = 0; nt < kNT1; ++nt) {
      int p = (wv * kNT1 + nt) * 16 + r;
      p = p < kMH * kMW ? p : kMH * kMW - 1;               // idle slots recompute the last pixel
      const int my = p / kMW, mx = p - my * kMW;
      b1a[nt] = (int)(4 * kWSlot) + (my * kXW + mx) * kPS + g * 16;     // x tile at smem + 4 * kWSlot
      b1b[nt] = b1a[nt] + hi * (kXW - 3) * kPS;
      b1c[nt] = b1a[nt] - hi * ((2 * kXW + 2) * kPS + 96);
    }
#pragma unroll
    for (int nt = 0; nt < kNT2; ++nt) {
      const int p = (wv * kNT2 + nt) * 16 + r;
      b2a[nt] = (int)(4 * kWSlot + kXBytes) + ((p >> 5) * kMW + (p & 31)) * kPS + g * 16;   // mid tile
      b2b[nt] = b2a[nt] + hi * (kMW - 3) * kPS;
      b2c[nt] = b2a[nt] - hi * ((2 * kMW + 2) * kPS + 96);
    }
  }
  float4v al1[kMT], be1[kMT], al2[kMT], be2[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int c4 = m * 16 + g * 4;
    al1[m] = *reinterpret_cast<const float4v*>(a.ab1 + c4);
    be1[m] = *reinterpret_cast<const float4v*>(a.ab1 + 48 + c4);
    al2[m] = *reinterpret_cast<const float4v*>(a.ab2 + c4);
    be2[m] = *reinterpret_cast<const float4v*>(a.ab2 + 48 + c4);
  }
#pragma unroll
  for (int m = 0; m < kMT; ++m) asm volatile("" ::"v"(al1[m]), "v"(be1[m]), "v"(al2[m]), "v"(be2[m]));

  int eoff[kNIT], xoff[kNIT], epos[kNIT];
#pragma unroll
  for (int it = 0; it < kNIT; ++it) {
    int c = it * 64 + lane;
    const bool exists = c < kNT2 * 16 * 6;
    c = exists ? c : 0;
    const int pw = c / 6, slot = c - pw * 6;
    const int p = wv * kNT2 * 16 + pw;
    const int oy = p >> 5, ox = p & 31;
    eoff[it] = pw * kRowB + slot * 16;
    xoff[it] = ((oy + 2) * kXW + ox + 2) * kPS + slot * 16;
    epos[it] = ((exists ? oy : 0x7fff) << 16) | (ox << 8) | (slot * 16);
  }

  // one conv: 14 k-steps straight through (no hand-over in the middle: the weights are resident)
  auto kloop = [&](const char* wbase, const int* ba, const int* bb, const int* bc, auto acc, auto ntc, auto pitch)
      __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value, RW = decltype(pitch)::value;
    const char* wl = wbase + lane * 16;
    // byte offset of k-step k for lane group 0 (see the base registers above)
    auto koff = [](int k) constexpr { const int tap = (32 * k) / 48; return ((tap / 3) * RW + tap % 3) * kPS + ((32 * k) % 48) * 2; };
    auto bptr = [&](int k, int nt) __attribute__((always_inline)) {
      const int base = k == 4 ? bb[nt] : k == 13 ? bc[nt] : ba[nt];
      return reinterpret_cast<const half8*>(smem + base + koff(k));
    };
    half8 af[2][kMT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) af[0][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *bptr(0, nt);
#pragma unroll
    for (int kk = 0; kk < 2 * kKH; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < 2 * kKH) {
#pragma unroll
        for (int m = 0; m < kMT; ++m) af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 1) * kMT + m) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nxt][nt] = *bptr(kk + 1, nt);
      }
#pragma unroll
      for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc[m][nt], 0, 0, 0);
      if (kk + 1 < 2 * kKH) {
#pragma unroll
        for (int i = 0; i < (kMT + NT + 1) / 2; ++i) {                               // two reads per MFMA, early in the step
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int u = 0; u < U; ++u) {
    uint32_t n;
    int py0, px0;
    unit_origin(u, &n, &py0, &px0);

    RTPE_BBARRIER();                                     // M: x tile u (and the weights) are in LDS
    // the residual = the block input: this wave's output row pieces of the x tile, kept in registers so
    // that the tile buffer can take the next tile as soon as conv1 is through
    half8 rv[kNIT];
#pragma unroll
    for (int it = 0; it < kNIT; ++it) rv[it] = *reinterpret_cast<const half8*>(xb + xoff[it]);

    // ------------------------------- conv1 -------------------------------
    float4v acc1[kMT][kNT1];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int nt = 0; nt < kNT1; ++nt) acc1[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    if (!(a.ablate & 8)) kloop(wts, b1a, b1b, b1c, acc1, std::integral_constant<int, kNT1>(), std::integral_constant<int, kXW>());
    // ---- epilogue A: BN1 + ReLU -> fp16 rows of the mid tile ----
    if (!(a.ablate & 2)) {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int re = lane_e & 15, ge = lane_e >> 4;
#pragma unroll
      for (int nt = 0; nt < kNT1; ++nt) {
        const int p = (wv * kNT1 + nt) * 16 + re;
        const int my = p / kMW, mx = p - my * kMW;
        const int iy = py0 - 1 + my, ix = px0 - 1 + mx;
        const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const half4 o_bn = bn_round(acc1[m][nt], al1[m], be1[m]);
          short4v b = __builtin_bit_cast(short4v, o_bn);
          b = b & ~(b >> 15);
          if (!inside) b = b ^ b;
          if (p < kMH * kMW)
            *reinterpret_cast<short4v*>(mid + p * kPS + m * 32 + ge * 8) = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // incl. the rv reads of the x tile
    }
#pragma unroll
    for (int it = 0; it < kNIT; ++it) asm volatile("" : "+v"(rv[it]));   // rv is complete before the buffer is handed back

    // ------------------------------- conv2 -------------------------------
    RTPE_BBARRIER();                                     // A2: mid tile complete, x buffer free
    float4v acc2[kMT][kNT2];
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int nt = 0; nt < kNT2; ++nt) acc2[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    if (!(a.ablate & 8)) kloop(wts + 2 * kWSlot, b2a, b2b, b2c, acc2, std::integral_constant<int, kNT2>(), std::integral_constant<int, kMW>());
    // ---- epilogue B: BN2, transposed through the (now free) mid tile, + x, ReLU, store ----
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int re = lane_e & 15, ge = lane_e >> 4;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RTPE_BBARRIER();                                   // E2: every wave is done reading the mid tile
      char* obuf = mid + wv * (kNT2 * 16 * kRowB);
#pragma unroll
      for (int nt = 0; nt < kNT2; ++nt)
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const half4 o_bn = bn_round(acc2[m][nt], al2[m], be2[m]);
          *reinterpret_cast<half4*>(obuf + (nt * 16 + re) * kRowB + m * 32 + ge * 8) = o_bn;
        }
      half8 ov[kNIT];
#pragma unroll
      for (int it = 0; it < kNIT; ++it) ov[it] = *reinterpret_cast<const half8*>(obuf + eoff[it]);
      const int hy = a.H - py0, hx = a.W - px0;
      char* const yb = reinterpret_cast<char*>(a.y + (((size_t)n * a.H + py0) * a.W + px0) * a.out_ld);
      const uint32_t ld2 = (uint32_t)a.out_ld * 2u, row_pix = (uint32_t)a.W & 0xffffffu;
#pragma unroll
      for (int it = 0; it < kNIT; ++it) {
        int e = epos[it];
        asm volatile("" : "+v"(e));
        half8 v = ov[it] + rv[it];                       // fp16 add, round-to-nearest-even = the wrapper's add
        short8 b = __builtin_bit_cast(short8, v);
        b = b & ~(b >> 15);
        if ((e >> 16) < hy && ((e >> 8) & 255) < hx && !(a.ablate & 4)) {
          const uint32_t pix = __umul24((uint32_t)e >> 16, row_pix) + (((uint32_t)e >> 8) & 255u);
          *reinterpret_cast<short8*>(yb + __umul24(pix, ld2) + ((uint32_t)e & 255u)) = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Producer / consumer variant ("pc").  Counters of the variants above (profiles/r02_pmc_summary.json): the matrix
// pipe is busy 42 % of the kernel and SQ_VALU_MFMA_COEXEC is ~0.7 %: the two epilogues (BN / ReLU / transposition
// / stores, ~45 % of a unit) never run beside matrix work, because the four MFMA waves move through the phases
// together.  Here the workgroup has EIGHT waves, two per SIMD, whose phases are complementary:
//   * producers P0-3, interval i: conv1 of unit i (x tile -> accumulators), request of the x tile of unit i+1
//     (LDS-DMA; there are no loader waves), then conv1's epilogue (BN1 + ReLU -> mid tile);
//   * consumers C0-3, interval i: epilogue of unit i-2 (BN2, transposition, + x, ReLU, stores) - VALU / LDS / memory
//     work while the producers' MFMAs own the matrix pipe - then conv2 of unit i-1 while the producers are in THEIR
//     epilogue.
// ONE workgroup barrier per unit.  What makes it fit the 160 KiB of LDS: 8 x 16 output tiles (conv1 region
// 10 x 18 = 180 pixels = 11.25 MFMA column tiles: 3 per producer wave, 94 % useful), and conv2's weight fragments
// live in the consumer waves' REGISTERS for the whole kernel (14 k-steps x 3 row tiles x 4 VGPRs = 168 of 256) -
// only conv1's 42 KiB sit in LDS:
//   W1 42 KiB | x tile (12 rows x 2,240 B) x 2 = 52.5 KiB | mid tile (10 x 18 px) x 2 = 33.75 KiB | output slabs 14 KiB
//   | residual slabs 12 KiB | BN2 0.4 KiB = 154.6 KiB
// x-tile rows have a 2,240 B pitch (20 pixels = 1,920 B + padding): every tile request is two full-wave LDS-DMA
// instructions per row with no lane masked off (the 8 spare lanes fetch out-of-range -> zeros into the padding):
// straight-line code.  The request's address arithmetic is done while the producer waits at the barrier, the
// requests themselves go out right after the k loop (in flight during the k loop they slow its LDS reads by ~15 %)
// and are waited for (vmcnt(0): a producer has no other memory operations) before the next barrier.
// Buffers alternate by unit parity: in interval i the producers read X[i&1], write MID[i&1] and refill X[(i+1)&1],
// whose last reader (their own conv1 of unit i-1) finished before the barrier; the consumers read MID[(i-1)&1].
// The residual is NOT taken from the x tile (it would have to be held for two intervals): each consumer wave
// re-requests its 3 x 16 B per lane from global memory (an L2 hit: the tile was fetched by the same XCD shortly
// before) by LDS-DMA into a 3 KiB slab of its own one interval ahead - no registers, no exposed latency; BN2's
// parameters are fetched from LDS one interval ahead as well.
// Same k order, same rounding points: bit-identical to the other variants.  Shapes: H % 8 == 0 and W % 16 == 0 (every
// tile complete: no partial-tile handling in the request, mid and store paths); other shapes run the
// resident-weights variant.  A consumer's loop body contains no branch around memory operations: the compiler's
// s_waitcnt pass takes the minimum over paths and would wait for requests that were just issued.
// s_setprio on either role was measured and changes nothing (tools/probes/coissue_probe.hip): not used.
namespace {
constexpr int kPTH = 8, kPTW = 16;                    // output tile
constexpr int kPMH = kPTH + 2, kPMW = kPTW + 2;       // conv1 region = mid tile (10 x 18 = 180 px)
constexpr int kPXH = kPTH + 4, kPXW = kPTW + 4;       // x tile (12 x 20)
constexpr int kPNT1 = 3, kPNT2 = 2;                   // 16-pixel tiles per producer / consumer wave
constexpr int kPXRow = 2240;                          // LDS bytes per x-tile row: >= 2,048 (two full-wave requests) and
                                                      // == 18 px * 96 B (mod 256): a 16-pixel MFMA column tile that wraps to the
                                                      // next mid row keeps the conflict-free bank pattern of consecutive pixels
constexpr int kPMRow = kPMW * kPS;                    // 1,728
constexpr int kPXBytes = kPXH * kPXRow;               // 26,880
constexpr int kPMidBytes = kPMH * kPMRow;             // 17,280
constexpr int kPObuf = kPNT2 * 16 * kRowB;            // 3,584 per consumer wave
constexpr int kPOffX = 2 * kWSlot;
constexpr int kPOffMid = kPOffX + 2 * kPXBytes;
constexpr int kPOffObuf = kPOffMid + 2 * kPMidBytes;
constexpr int kPOffBn = kPOffObuf + 4 * kPObuf;
constexpr int kPOffRes = kPOffBn + 96 * 4;            // residual pieces: 3 KiB per consumer wave
constexpr int kLdsPC = kPOffRes + 4 * 3072;           // 158,336
constexpr int kPNIT = kPNT2 * 16 * 6 / 64;            // 3 pieces of 16 bytes per consumer lane
static_assert(kPXW * 6 <= 128 && kPXRow >= 128 * 16, "an x-tile row is two full-wave requests");
static_assert(kPMH * kPMW <= 4 * kPNT1 * 16, "conv1 region fits the producers' column tiles");
}

#ifdef RTPE_DIAG
// diagnostic build only: shader-clock stamps of one workgroup's waves (kept in spare LDS, copied out at the end)
__device__ unsigned long long g_pc_trace[8 * 64];
#define RTPE_PC_TRACE(slot)                                                                                     \
  do {                                                                                                          \
    if (trace_on && (slot) < 64) {                                                                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                             \
      if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(smem + kLdsPC + (wv * 64 + (slot)) * 8) = t_; \
    }                                                                                                           \
  } while (0)
#define RTPE_PC_TRACE_FLUSH()                                                                                   \
  do {                                                                                                          \
    if (trace_on) g_pc_trace[wv * 64 + lane] = *reinterpret_cast<unsigned long long*>(smem + kLdsPC + (wv * 64 + lane) * 8); \
  } while (0)
#else
#define RTPE_PC_TRACE(slot) do { } while (0)
#define RTPE_PC_TRACE_FLUSH() do { } while (0)
#endif

__global__ void __launch_bounds__(512) conv_block_pc_kernel(const BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int n_tiles = a.N * a.tiles_x * a.tiles_y;
  const int G = (int)(gridDim.x >> 3), xcd = (int)(blockIdx.x & 7), jw = (int)(blockIdx.x >> 3);
  const int per_xcd = (n_tiles + 7) >> 3;
  const int t_begin = xcd * per_xcd;
  const int tiles_xcd = min(per_xcd, n_tiles - t_begin);
  const int U = jw < tiles_xcd ? (tiles_xcd - jw + G - 1) / G : 0;
  if (U == 0) return;
#ifdef RTPE_DIAG
  const bool trace_on = blockIdx.x == 8 * 3 + 2 && a.H == 160;   // one workgroup in the middle of XCD 2
  if (trace_on) *reinterpret_cast<unsigned long long*>(smem + kLdsPC + tid * 8) = 0ull;
#endif
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  auto unit_origin = [&](int u, uint32_t* n, int* py0, int* px0) __attribute__((always_inline)) {
    uint32_t t = (uint32_t)(t_begin + jw + u * G);
    *n = fdiv(t, a.div_tiles_xy);
    t -= *n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    *py0 = (int)tyi * kPTH;
    *px0 = (int)(t - tyi * a.tiles_x) * kPTW;
  };
  const int r = lane & 15, g = lane >> 4;
  const int hi = g >= 2 ? 1 : 0;
  // byte offset of k-step k (32 input channels from channel 32k % 48 of tap 32k / 48) in a tile of row pitch rowb
  auto koff = [](int k, int rowb) constexpr { const int tap = (32 * k) / 48; return (tap / 3) * rowb + (tap % 3) * kPS + ((32 * k) % 48) * 2; };

  // x-tile request: wave quarter wq (= wave & 3) asks for rows [3 wq, 3 wq + 3) of the tile of unit u into buffer
  // `par`: 6 full-wave LDS-DMA instructions, no branches.  Per-lane constants (column, 16-byte slot) are kept.
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, a.x_bytes, 0x00020000);
  const int wq = wv & 3;
  int xcol[2];
  uint32_t xoffs[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = k * 64 + lane;
    const int hx = q / 6, sl = q - hx * 6;
    xcol[k] = q < kPXW * 6 ? hx - 2 : -(1 << 20);          // column relative to the tile's first output column
    xoffs[k] = (uint32_t)((hx - 2) * a.in_ld + sl * 8) * 2u;
  }
  // in two halves, so that the address arithmetic can be done while waiting at the barrier
  struct XReq { uint32_t voff[2][kPXH / 4]; int soff[kPXH / 4]; };
  auto prepare_x = [&](int u, XReq& q) __attribute__((always_inline)) {
    uint32_t n;
    int py0, px0;
    unit_origin(u, &n, &py0, &px0);
    const int iy0 = py0 - 2 + wq * (kPXH / 4);
    uint32_t voff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
      voff[k] = (unsigned)(px0 + xcol[k]) < (unsigned)a.W ? xoffs[k] + (uint32_t)(px0 * a.in_ld) * 2u : 0x80000000u;
    const int img_row0 = (int)n * a.H;
#pragma unroll
    for (int j = 0; j < kPXH / 4; ++j) {
      const int iy = iy0 + j;
      const bool row_ok = (unsigned)iy < (unsigned)a.H;
      q.soff[j] = row_ok ? (img_row0 + iy) * a.W * a.in_ld * 2 : 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) q.voff[k][j] = row_ok ? voff[k] : 0x80000000u;
    }
  };
  auto fire_x = [&](const XReq& q, int par) __attribute__((always_inline)) {
    char* buf = smem + kPOffX + par * kPXBytes + wq * (kPXH / 4) * kPXRow;
#pragma unroll
    for (int j = 0; j < kPXH / 4; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(buf + j * kPXRow + k * 1024), 16, (int)q.voff[k][j],
                                                 q.soff[j], 0, 0);
  };
  auto issue_x = [&](int u, int par) __attribute__((always_inline)) {
    XReq q;
    prepare_x(u, q);
    fire_x(q, par);
  };

  if (wv < 4) {
    // =========================== producers: conv1 + BN1 + ReLU -> mid tile ===========================
    int ba[kPNT1], bb[kPNT1], bc[kPNT1];
#pragma unroll
    for (int nt = 0; nt < kPNT1; ++nt) {
      int p = (wv * kPNT1 + nt) * 16 + r;
      p = p < kPMH * kPMW ? p : kPMH * kPMW - 1;           // the 12 idle columns of the last tile recompute a pixel
      const int my = p / kPMW, mx = p - my * kPMW;
      ba[nt] = kPOffX + my * kPXRow + mx * kPS + g * 16;
      bb[nt] = ba[nt] + hi * (kPXRow - 3 * kPS);           // k-step 4: lanes 32-63 are in tap (1,0), not "(0,3)"
      bc[nt] = ba[nt] - hi * (2 * kPXRow + 2 * kPS + 96);  // k-step 13: lanes 32-63 = zero-weight padding, tap (0,0)
    }
    // epilogue: accumulator column r of tile nt is mid pixel p (the idle columns of the last tile recompute
    // pixel 179 and store the same bits to the same place); this lane's 4 channels start at g * 4
    int moff[kPNT1], myx[kPNT1];
#pragma unroll
    for (int nt = 0; nt < kPNT1; ++nt) {
      int p = (wv * kPNT1 + nt) * 16 + r;
      p = p < kPMH * kPMW ? p : kPMH * kPMW - 1;
      moff[nt] = p * kPS + g * 8;
      myx[nt] = ((p / kPMW) << 8) | (p % kPMW);
    }
    float4v al1[kMT], be1[kMT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
      const int c4 = m * 16 + g * 4;
      al1[m] = *reinterpret_cast<const float4v*>(a.ab1 + c4);
      be1[m] = *reinterpret_cast<const float4v*>(a.ab1 + 48 + c4);
    }
#pragma unroll
    for (int m = 0; m < kMT; ++m) asm volatile("" : "+v"(al1[m]), "+v"(be1[m]));
    const char* wl = smem + lane * 16;                     // conv1's fragments, resident
    for (int i = 0; i < U; ++i) {
      uint32_t n;
      int py0, px0;
      unit_origin(i, &n, &py0, &px0);
      const int xsel = (i & 1) * kPXBytes;
      // the tile of unit i+1 goes into the buffer conv1 of unit i-1 releases at the barrier (the last interval
      // re-requests the last tile: nobody reads it); addresses first, the requests right after the barrier
      XReq xq;
      prepare_x(i + 1 < U ? i + 1 : U - 1, xq);
#pragma unroll
      for (int j = 0; j < kPXH / 4; ++j) asm volatile("" : "+v"(xq.voff[0][j]), "+v"(xq.voff[1][j]), "+s"(xq.soff[j]));
      RTPE_PC_TRACE(i * 5 + 0);
      RTPE_BBARRIER();                                     // B(i): x tile i landed, MID[i&1] free
      RTPE_PC_TRACE(i * 5 + 1);
      RTPE_PC_TRACE(i * 5 + 2);
      float4v acc[kMT][kPNT1];
#pragma unroll
      for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int nt = 0; nt < kPNT1; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
      if (!(a.ablate & 4)) {
        const char* ta[kPNT1];
        const char* tb[kPNT1];
        const char* tc[kPNT1];
#pragma unroll
        for (int nt = 0; nt < kPNT1; ++nt) {
          ta[nt] = smem + ba[nt] + xsel; tb[nt] = smem + bb[nt] + xsel; tc[nt] = smem + bc[nt] + xsel;
        }
        auto bptr = [&](int k, int nt) __attribute__((always_inline)) {
          return reinterpret_cast<const half8*>((k == 4 ? tb[nt] : k == 13 ? tc[nt] : ta[nt]) + koff(k, kPXRow));
        };
        // operands are fetched TWO k-steps ahead (three register sets): with 3 x 3 MFMAs per k-step (144 cycles)
        // and the four producer waves reading in step, a fetch issued one step ahead is not back in time
        half8 af[3][kMT], bf[3][kPNT1];
#pragma unroll
        for (int s0 = 0; s0 < 2; ++s0) {
#pragma unroll
          for (int m = 0; m < kMT; ++m) af[s0][m] = *reinterpret_cast<const half8*>(wl + (s0 * kMT + m) * 1024);
#pragma unroll
          for (int nt = 0; nt < kPNT1; ++nt) bf[s0][nt] = *bptr(s0, nt);
        }
#pragma unroll
        for (int kk = 0; kk < 2 * kKH; ++kk) {
          const int cur = kk % 3, nxt = (kk + 2) % 3;
          if (kk + 2 < 2 * kKH) {
#pragma unroll
            for (int m = 0; m < kMT; ++m) af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 2) * kMT + m) * 1024);
#pragma unroll
            for (int nt = 0; nt < kPNT1; ++nt) bf[nxt][nt] = *bptr(kk + 2, nt);
          }
#pragma unroll
          for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int nt = 0; nt < kPNT1; ++nt)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc[m][nt], 0, 0, 0);
          if (kk + 2 < 2 * kKH) {
#pragma unroll
            for (int q = 0; q < kMT + kPNT1; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#ifdef RTPE_DIAG
      if (!(a.ablate & 32))
#endif
      fire_x(xq, (i + 1) & 1);                             // after the k loop: requests in flight slow its LDS reads by ~15 %
      RTPE_PC_TRACE(i * 5 + 3);
      // ---- BN1 + ReLU -> fp16 rows of MID[i&1] (zeros outside the image = conv2's padding) ----
      if (!(a.ablate & 1)) {
        char* mid = smem + kPOffMid + (i & 1) * kPMidBytes;
        // a tile whose 10 x 18 conv1 region lies inside the image (all but the border tiles) needs no masking
        const bool interior = py0 >= 1 && py0 + kPTH < a.H && px0 >= 1 && px0 + kPTW < a.W;
        auto bn_relu = [&](int m, int nt) __attribute__((always_inline)) {
          const float4v v = acc[m][nt];
          half4 o = bn_round(v, al1[m], be1[m]);
          short4v b = __builtin_bit_cast(short4v, o);
          return b & ~(b >> 15);
        };
        if (interior) {
#pragma unroll
          for (int nt = 0; nt < kPNT1; ++nt)
#pragma unroll
            for (int m = 0; m < kMT; ++m) *reinterpret_cast<short4v*>(mid + moff[nt] + m * 32) = bn_relu(m, nt);
        } else {
#pragma unroll
          for (int nt = 0; nt < kPNT1; ++nt) {
            const int iy = py0 - 1 + (myx[nt] >> 8), ix = px0 - 1 + (myx[nt] & 255);
            const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
            for (int m = 0; m < kMT; ++m) {
              short4v b = bn_relu(m, nt);
              if (!inside) b = b ^ b;
              *reinterpret_cast<short4v*>(mid + moff[nt] + m * 32) = b;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      RTPE_PC_TRACE(i * 5 + 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's rows of tile i+1 have landed
    }
    RTPE_PC_TRACE(U * 5 < 60 ? U * 5 : 60);
    RTPE_BBARRIER();                                       // B(U): the consumers' last conv2
    RTPE_PC_TRACE_FLUSH();
    return;
  }

  // ========== consumers: x-tile requests; BN2 + x + ReLU + stores of unit i-2; conv2 (weights in registers) of unit i-1 ==========
  const int cw = wv - 4;                                   // output rows 2 cw, 2 cw + 1 of the tile
  // prologue: conv1's weights into LDS (a quarter per consumer wave), BN2 parameters, this wave's part of x tile 0
  {
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w1), 0, 2 * kWSlot, 0x00020000);
    const int voff = lane * 16;
    for (int p = cw; p < 2 * kMT * kKH; p += 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(smem + p * 1024), 16, voff, p * 1024, 0, 0);
    if (cw == 0 && lane < 24) reinterpret_cast<float4*>(smem + kPOffBn)[lane] = reinterpret_cast<const float4*>(a.ab2)[lane];
  }
  issue_x(0, 0);
  // conv2's weight fragments: this lane's 16 bytes of every (k-step, row tile), kept for the whole kernel
  half8 w2[2 * kKH][kMT];
#pragma unroll
  for (int k = 0; k < 2 * kKH; ++k)
#pragma unroll
    for (int m = 0; m < kMT; ++m)
      w2[k][m] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(a.w2) + ((k * kMT + m) * 64 + lane) * 16);
  int ca[kPNT2];
#pragma unroll
  for (int nt = 0; nt < kPNT2; ++nt)
    ca[nt] = kPOffMid + ((cw * kPNT2 + nt) * kPMW + r) * kPS + g * 16;   // output pixel (row 2 cw + nt, column r)
  // k-steps 4 and 13: lanes 32-63 are in tap (1,0) / in the zero-weight padding (see the producers' bb, bc)
  const int hi4 = hi * (kPMRow - 3 * kPS), hi13 = -hi * (2 * kPMRow + 2 * kPS + 96);
  // 16-byte pieces of this wave's two output rows: piece c = it * 64 + lane -> slab pixel c / 6 (row c / 96 of
  // the pair, column (c / 6) % 16), slot c % 6.  Offsets in the slab, in x and in y (bytes from the pair's first pixel)
  uint32_t ooffs[kPNIT], xoffr[kPNIT], yoffr[kPNIT];
#pragma unroll
  for (int it = 0; it < kPNIT; ++it) {
    const int c = it * 64 + lane;
    const int pw = c / 6, slot = c - pw * 6;
    ooffs[it] = (uint32_t)(pw * kRowB + slot * 16);
    xoffr[it] = (uint32_t)(((pw >> 4) * a.W + (pw & 15)) * a.in_ld * 2 + slot * 16);
    yoffr[it] = (uint32_t)(((pw >> 4) * a.W + (pw & 15)) * a.out_ld * 2 + slot * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's prologue requests: its quarter of W1, BN2, its part of x tile 0, its w2 registers
  // tell the compiler too: otherwise it keeps the 42 loads "pending" at the loop header and counts them down
  // with s_waitcnt vmcnt(N) inside the k loop - N so small for the last ones that the wait would also cover the
  // x-tile requests issued a moment earlier
#pragma unroll
  for (int k = 0; k < 2 * kKH; ++k)
#pragma unroll
    for (int m = 0; m < kMT; ++m) asm volatile("" : "+v"(w2[k][m]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  float4v acc[kMT][kPNT2];
  auto conv2 = [&](int v) __attribute__((always_inline)) {
    const int msel = (v & 1) * kPMidBytes;
#pragma unroll
    for (int m = 0; m < kMT; ++m)
#pragma unroll
      for (int nt = 0; nt < kPNT2; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    if (a.ablate & 8) return;
    const char* ta[kPNT2];
#pragma unroll
    for (int nt = 0; nt < kPNT2; ++nt) ta[nt] = smem + ca[nt] + msel;
    auto bptr = [&](int k, int nt) __attribute__((always_inline)) {
      return reinterpret_cast<const half8*>(ta[nt] + (k == 4 ? hi4 : k == 13 ? hi13 : 0) + koff(k, kPMRow));
    };
    half8 bf[3][kPNT2];                                    // fetched two k-steps ahead, as in conv1
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
      for (int nt = 0; nt < kPNT2; ++nt) bf[s0][nt] = *bptr(s0, nt);
#pragma unroll
    for (int kk = 0; kk < 2 * kKH; ++kk) {
      const int cur = kk % 3, nxt = (kk + 2) % 3;
      if (kk + 2 < 2 * kKH) {
#pragma unroll
        for (int nt = 0; nt < kPNT2; ++nt) bf[nxt][nt] = *bptr(kk + 2, nt);
      }
#pragma unroll
      for (int m = 0; m < kMT; ++m)
#pragma unroll
        for (int nt = 0; nt < kPNT2; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[kk][m], bf[cur][nt], acc[m][nt], 0, 0, 0);
      if (kk + 2 < 2 * kKH) {
#pragma unroll
        for (int q = 0; q < kPNT2; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // BN2 parameters of this lane's channels (m * 16 + g * 4 ... + 3): fetched from LDS one interval ahead
  struct Bn2 { float4v al[kMT], be[kMT]; };
  auto load_bn = [&](Bn2& bn) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
      bn.al[m] = *reinterpret_cast<const float4v*>(smem + kPOffBn + (m * 16 + g * 4) * 4);
      bn.be[m] = *reinterpret_cast<const float4v*>(smem + kPOffBn + (48 + m * 16 + g * 4) * 4);
    }
  };
  // residual request for unit v: this wave's 192 pieces (3 full-wave LDS-DMA instructions) into its slab, piece
  // it * 64 + lane at it * 1024 + lane * 16; returns the address of the pair's first output pixel
  char* const res = smem + kPOffRes + cw * 3072;
  auto request_residual = [&](int v) __attribute__((always_inline)) -> char* {
    uint32_t n;
    int py0, px0;
    unit_origin(v, &n, &py0, &px0);
    const int pix0 = ((int)n * a.H + py0 + cw * kPNT2) * a.W + px0;
#pragma unroll
    for (int it = 0; it < kPNIT; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(res + it * 1024), 16, (int)xoffr[it], pix0 * a.in_ld * 2, 0, 0);
    return reinterpret_cast<char*>(a.y + (size_t)pix0 * a.out_ld);
  };
  // BN2 of the accumulators, transposed through this wave's slab, + x, ReLU, store
  auto epilogue = [&](const Bn2& bn, char* yb) __attribute__((always_inline)) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int re = lane_e & 15, ge = lane_e >> 4;
    char* obuf = smem + kPOffObuf + cw * kPObuf;
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
      const float4v al = bn.al[m], be = bn.be[m];
#pragma unroll
      for (int nt = 0; nt < kPNT2; ++nt) {
        *reinterpret_cast<half4*>(obuf + (nt * 16 + re) * kRowB + m * 32 + ge * 8) = bn_round(acc[m][nt], al, be);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the residual pieces have landed (requested an interval ago)
#pragma unroll
    for (int it = 0; it < kPNIT; ++it) {
      const half8 ov = *reinterpret_cast<const half8*>(obuf + ooffs[it]);
      const half8 rv = *reinterpret_cast<const half8*>(res + it * 1024 + lane_e * 16);
      half8 o = ov + rv;                                   // fp16 add, round-to-nearest-even = the wrapper's add
      short8 b = __builtin_bit_cast(short8, o);
      b = b & ~(b >> 15);
      store16_wt(yb + yoffr[it], b);                       // every tile is complete
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slab is read: the next request may overwrite it
  };

  RTPE_BBARRIER();                                         // B(0)
  Bn2 bn;
  char* yb_next = nullptr;
  // interval 1: conv2 of unit 0 only (no finished unit yet)
  if (U > 1) {
    RTPE_BBARRIER();                                       // B(1): MID[0] complete
    yb_next = request_residual(0);
    conv2(0);
    load_bn(bn);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  for (int i = 2; i <= U; ++i) {
    RTPE_PC_TRACE((i - 2) * 5 + 0);
    RTPE_BBARRIER();                                       // B(i): MID[(i-1)&1] complete
    RTPE_PC_TRACE((i - 2) * 5 + 1);
    char* const yb = yb_next;
    epilogue(bn, yb);                                      // unit i-2
    RTPE_PC_TRACE((i - 2) * 5 + 2);
    yb_next = request_residual(i - 1);
    RTPE_PC_TRACE((i - 2) * 5 + 3);
    conv2(i - 1);
    RTPE_PC_TRACE((i - 2) * 5 + 4);
    load_bn(bn);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  {
    // the last unit (U == 1: its conv2 has not run yet - MID[0] is complete after B(1))
    if (U == 1) {
      RTPE_BBARRIER();                                     // B(1)
      yb_next = request_residual(0);
      conv2(0);
      load_bn(bn);
    }
    epilogue(bn, yb_next);
  }
  RTPE_PC_TRACE_FLUSH();
}


bool conv_block_supports(int cin, int cout, int H, int W) { return cin == 48 && cout == 48 && H >= kTH && W >= 16; }

int conv_block_launch(const _Float16* x, int in_ld, size_t x_bytes, _Float16* y, int out_ld, const _Float16* w1,
                      const float* ab1, const _Float16* w2, const float* ab2, int N, int H, int W, hipStream_t s) {
  RTPE_REQUIRE(x && y && w1 && w2 && ab1 && ab2 && N > 0, "basic block: null argument");
  RTPE_REQUIRE(in_ld % 8 == 0 && out_ld % 8 == 0 && in_ld >= 48 && out_ld >= 48 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
               "basic block: NHWC views must be 16-byte aligned (in_ld=%d out_ld=%d)", in_ld, out_ld);
  RTPE_REQUIRE(x_bytes > 0 && x_bytes < 0x80000000ull, "basic block: input view of %zu bytes", x_bytes);
  // option "block_ring" (or RTPE_BLOCK_RING=1): the variant that streams the weights through a 3-slot ring
  // (bit-identical results; kept for A/B measurements in one process)
  const int ring = get_option(kOptBlockRing);
  // option "block_pc" (default 1; RTPE_BLOCK_PC=0 or rtpe_set_option("block_pc", 0) to disable): the producer /
  // consumer variant wherever the shape allows it (complete 8 x 16 tiles); bit-identical results
  const bool pc = get_option(kOptBlockPC) != 0 && !ring && H % kPTH == 0 && W % kPTW == 0;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_block_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_block_rw_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_block_pc_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  BlockArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.w1 = w1; a.w2 = w2; a.ab1 = ab1; a.ab2 = ab2;
  a.N = N; a.H = H; a.W = W; a.in_ld = in_ld; a.out_ld = out_ld;
  a.tiles_x = pc ? W / kPTW : (W + kTW - 1) / kTW;
  a.tiles_y = pc ? H / kPTH : (H + kTH - 1) / kTH;
  a.div_tiles_x = make_fastdiv(a.tiles_x);
  a.div_tiles_xy = make_fastdiv(a.tiles_x * a.tiles_y);
  a.x_bytes = (int)x_bytes;
  static const int abl = RTPE_DIAG_ENV_INT("RTPE_BLOCK_ABL", 0);
  a.ablate = abl;
  const long tiles = (long)N * a.tiles_x * a.tiles_y;
  static const int g_env = env_int("RTPE_PERSIST_G", 32);
  long G = g_env;                                         // one workgroup per CU
  if (G > (tiles + 7) / 8) G = (tiles + 7) / 8;
  if (pc)
#ifdef RTPE_DIAG
    hipLaunchKernelGGL(conv_block_pc_kernel, dim3((unsigned)(8 * G)), dim3(512), kLdsPC + 8 * 64 * 8, s, a);
#else
    hipLaunchKernelGGL(conv_block_pc_kernel, dim3((unsigned)(8 * G)), dim3(512), kLdsPC, s, a);
#endif
  else if (ring)
    hipLaunchKernelGGL(conv_block_kernel, dim3((unsigned)(8 * G)), dim3((kWaves + kLoad) * 64), kLds, s, a);
  else
    hipLaunchKernelGGL(conv_block_rw_kernel, dim3((unsigned)(8 * G)), dim3((kWaves + kLoad) * 64), kLdsRW, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe

#ifdef RTPE_DIAG
extern "C" int rtpe_diag_pc_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rtpe::g_pc_trace), sizeof(unsigned long long) * 8 * 64) == hipSuccess ? 0 : -1;
}
#endif
