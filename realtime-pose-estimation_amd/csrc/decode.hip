// Heatmap -> keypoint decode on gfx950: bilinear upsample, 5x5 max-pool NMS,
// top-K per (image, joint), tag gather, quarter-pixel adjust, tag-penalised
// arg-max refine.  Replaces validate_hhrnet.py:94-98 and the device side of
// rtpe/third_party/group.py:125-287 of the reference.
//
// Everything here is HBM/latency-bound integer + fp32 compare work, so the
// design goal is to touch the low-resolution network outputs once and never
// materialise the upsampled (h, w) maps (fused path): a sampler evaluates
// PyTorch-CPU's exact bilinear formula on the fly.  Bit-exactness rules:
//   * compiled with -ffp-contract=off; the fused multiply-adds that PyTorch's
//     CPU kernel performs are written explicitly (T = fma(v0,l0,v1*l1)),
//   * ties in top-K and arg-max go to the lowest flat index (np.argmax /
//     first-occurrence semantics); keys are (order-preserving value bits << 32
//     | ~index) so one unsigned 64-bit max does both.
#include "rtpe_common.h"

namespace rtpe {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------
// samplers
// ---------------------------------------------------------------------------
struct DirectMap {          // dense (planes, h, w)
  const float* p;
  int h, w;
  __device__ __forceinline__ float at(int plane, int y, int x) const {
    return p[((size_t)plane * h + y) * w + x];
  }
};

struct Axis {               // one axis of F.interpolate(bilinear, align_corners=True)
  float scale;              // float(in-1)/float(out-1)
  int n_in, same;
  __device__ __forceinline__ void at(int o, int* i0, int* i1, float* l0, float* l1) const {
    if (same) { *i0 = *i1 = o; *l0 = 1.f; *l1 = 0.f; return; }
    const float real = scale * (float)o;
    int a = (int)real;
    a = a < n_in - 1 ? a : n_in - 1;
    *i0 = a;
    *i1 = a + (a < n_in - 1 ? 1 : 0);
    float l = real - (float)a;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
    *l1 = l;
    *l0 = 1.f - l;
  }
};

struct BilinearMap {        // low-res planes sampled at (oh, ow) resolution
  const float* p;
  int sh, sw, J;
  long long img_stride;     // elements between images; plane j of image n at n*img_stride + j*sh*sw
  Axis ay, ax;
  __device__ __forceinline__ float at(int plane, int y, int x) const {
    const int n = plane / J, j = plane - n * J;
    const float* b = p + (size_t)n * img_stride + (size_t)j * sh * sw;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    ay.at(y, &y0, &y1, &ly0, &ly1);
    ax.at(x, &x0, &x1, &lx0, &lx1);
    const float v00 = b[y0 * sw + x0], v01 = b[y0 * sw + x1];
    const float v10 = b[y1 * sw + x0], v11 = b[y1 * sw + x1];
    const float t0 = __builtin_fmaf(v00, lx0, v01 * lx1);
    const float t1 = __builtin_fmaf(v10, lx0, v11 * lx1);
    return __builtin_fmaf(t0, ly0, t1 * ly1);
  }
};

static Axis make_axis(int n_in, int n_out) {
  Axis a;
  a.n_in = n_in;
  a.same = n_in == n_out;
  a.scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
  return a;
}

__device__ __forceinline__ unsigned order_bits(float v) {   // monotone float -> uint
  if (v == 0.f) v = 0.f;                                     // -0 == +0
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float v, unsigned idx) {
  return ((u64)order_bits(v) << 32) | (u64)(0xffffffffu - idx);
}

__device__ __forceinline__ u64 wave_max(u64 k) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)k, o), hi = __shfl_xor((unsigned)(k >> 32), o);
    const u64 other = ((u64)hi << 32) | lo;
    k = other > k ? other : k;
  }
  return k;
}

// block-wide max of a key; red must hold blockDim/64 entries; all threads get the result
__device__ __forceinline__ u64 block_max(u64 k, u64* red) {
  k = wave_max(k);
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = k;
  __syncthreads();
  u64 r = red[0];
  for (int i = 1; i < nw; ++i) r = red[i] > r ? red[i] : r;
  return r;
}

// ---------------------------------------------------------------------------
// plain bilinear upsample and NMS (API parity with F.interpolate / parser.nms)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bilinear_kernel(BilinearMap m, int planes, int oh, int ow, float* dst) {
  const size_t total = (size_t)planes * oh * ow;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % ow);
    const size_t r = i / ow;
    const int y = (int)(r % oh), pl = (int)(r / oh);
    dst[i] = m.at(pl, y, x);
  }
}

constexpr int kTH = 32, kTW = 64;   // NMS / top-k tile (2048 px, 8 per thread)
constexpr int kMaxPad = 4;          // supports nms kernels up to 9x9
constexpr int kMaxCand = 512;       // local maxima of a tile handled by the one-pass rank sort

// per-tile samples of the (virtual) full-resolution map.  For the bilinear sampler the two
// axis computations are done once per tile row / column (tables in LDS) instead of once per
// sample; the per-sample arithmetic is exactly BilinearMap::at's.
struct AxisTab {
  int i0[kTW + 2 * kMaxPad], i1[kTW + 2 * kMaxPad];
  float l0[kTW + 2 * kMaxPad], l1[kTW + 2 * kMaxPad];
};

// returns false (and leaves `raw` unfilled) only when the caller passed `pos_flag` and no source value under
// the tile is positive
constexpr int kTRows = 24;          // source rows of a tile kept as horizontally interpolated rows (separable sampling)

template <class Map>
__device__ __forceinline__ bool fill_raw(const Map& m, int plane, int h, int w, int y0, int x0, int pad, float* raw,
                                         AxisTab*, AxisTab*, float*, int, int* = nullptr, float* = nullptr) {
  const int PW = kTW + 2 * pad, PH = kTH + 2 * pad;
  for (int i = threadIdx.x; i < PH * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int y = y0 - pad + py, x = x0 - pad + px;
    raw[i] = ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? m.at(plane, y, x) : -INFINITY;
  }
  return true;
}

template <>
__device__ __forceinline__ bool fill_raw<BilinearMap>(const BilinearMap& m, int plane, int h, int w, int y0, int x0,
                                                      int pad, float* raw, AxisTab* ty, AxisTab* tx, float* stage,
                                                      int stage_floats, int* pos_flag, float* tbuf) {
  const int PW = kTW + 2 * pad, PH = kTH + 2 * pad;
  for (int i = threadIdx.x; i < PH + PW; i += 256) {
    const bool isy = i < PH;
    const int k = isy ? i : i - PH;
    const int o = (isy ? y0 : x0) - pad + k;
    AxisTab* t = isy ? ty : tx;
    const int lim = isy ? h : w;
    int a0 = -1, a1 = -1;
    float f0 = 0.f, f1 = 0.f;
    if ((unsigned)o < (unsigned)lim) (isy ? m.ay : m.ax).at(o, &a0, &a1, &f0, &f1);
    t->i0[k] = a0; t->i1[k] = a1; t->l0[k] = f0; t->l1[k] = f1;
  }
  __syncthreads();
  const int n = plane / m.J, j = plane - n * m.J;
  const float* b = m.p + (size_t)n * m.img_stride + (size_t)j * m.sh * m.sw;
  // The source pixels a tile needs form a small rectangle (about half the tile per axis when the map
  // is upsampled 2x): stage it in LDS once (in `stage`, the row-max buffer, free at this point) and take
  // the four taps of every sample from there instead of from L1; same arithmetic, same result.
  const int ky0 = max(0, pad - y0), ky1 = min(PH - 1, h - 1 - (y0 - pad));
  const int kx0 = max(0, pad - x0), kx1 = min(PW - 1, w - 1 - (x0 - pad));
  const int sr0 = ty->i0[ky0], sr1 = ty->i1[ky1], sc0 = tx->i0[kx0], sc1 = tx->i1[kx1];
  const int er = sr1 - sr0 + 1, ec = sc1 - sc0 + 1;
  const bool staged = stage != nullptr && ky0 <= ky1 && kx0 <= kx1 && er > 0 && ec > 0 && er * ec <= stage_floats;
  if (staged) {
    bool pos = false;
    // four loads in flight per thread before the first LDS write (the plain loop compiled to one load + vmcnt(0) per
    // element: one memory round trip per 256 source values; a tile has ~700)
    for (int i0 = threadIdx.x; i0 < er * ec; i0 += 4 * 256) {
      float v4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 256;
        const int ii = i < er * ec ? i : 0;               // (no branch around the load)
        const int r = ii / ec, c = ii - r * ec;
        v4[k] = b[(sr0 + r) * m.sw + sc0 + c];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 256;
        if (i < er * ec) {
          stage[i] = v4[k];
          pos |= v4[k] > 0.f;
        }
      }
    }
    if (pos_flag != nullptr && pos) *pos_flag = 1;         // (the caller zeroed it before the axis-table barrier)
    __syncthreads();
    // every sample is a combination of these values with weights >= 0: none positive here means no positive
    // sample, i.e. no candidate in the tile (the top-k only takes positive local maxima)
    if (pos_flag != nullptr && *pos_flag == 0) return false;
  }
  // sample (py, px): the four taps and three multiply-adds of F.interpolate(align_corners=True), in its order
  auto sample = [&](int r0, int r1, float ly0, float ly1, int c0, int c1, float lx0, float lx1) -> float {
    if (r0 < 0 || c0 < 0) return -INFINITY;               // outside the image
    float v00, v01, v10, v11;
    if (staged) {
      const float* s0 = stage + (r0 - sr0) * ec - sc0;
      const float* s1 = stage + (r1 - sr0) * ec - sc0;
      v00 = s0[c0]; v01 = s0[c1]; v10 = s1[c0]; v11 = s1[c1];
    } else {
      v00 = b[r0 * m.sw + c0]; v01 = b[r0 * m.sw + c1];
      v10 = b[r1 * m.sw + c0]; v11 = b[r1 * m.sw + c1];
    }
    const float t0 = __builtin_fmaf(v00, lx0, v01 * lx1);
    const float t1 = __builtin_fmaf(v10, lx0, v11 * lx1);
    return __builtin_fmaf(t0, ly0, t1 * ly1);
  };
  if (tbuf != nullptr && staged && er <= kTRows && (PW & 3) == 0) {
    // Separable form (5x5 window path): T(r, px) = fma(v(r, c0), lx0, v(r, c1) * lx1) depends on the SOURCE row r and
    // the output column only, and every source row serves ~4 output rows (as their upper or lower row): the
    // horizontal step is done once per (source row, column), the vertical step fma(T(r0), ly0, T(r1) * ly1) takes
    // 4 columns per thread with 16-byte LDS accesses.  The same operations on the same operands as
    // BilinearMap::at: bit-equal.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
      const int c0 = tx->i0[lane], c1 = tx->i1[lane];
      const float lx0 = tx->l0[lane], lx1 = tx->l1[lane];
      for (int r = wv; r < er; r += 4) {
        const float* srow = stage + r * ec - sc0;
        tbuf[r * PW + lane] = c0 < 0 ? 0.f : __builtin_fmaf(srow[c0], lx0, srow[c1] * lx1);
      }
      const int extra = PW - 64;                           // 2 * pad columns
      for (int i = threadIdx.x; i < er * extra; i += 256) {
        const int r = i / extra, px = 64 + i - r * extra;
        const int e0 = tx->i0[px], e1 = tx->i1[px];
        const float* srow = stage + r * ec - sc0;
        tbuf[r * PW + px] = e0 < 0 ? 0.f : __builtin_fmaf(srow[e0], tx->l0[px], srow[e1] * tx->l1[px]);
      }
    }
    __syncthreads();
    const int groups = PW >> 2;                            // 4 columns per thread and row
    for (int i = threadIdx.x; i < PH * groups; i += 256) {
      const int py = i / groups, cg = i - py * groups;
      const int r0 = ty->i0[py];
      float4 o = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (r0 >= 0) {
        const int r1 = ty->i1[py];
        const float ly0 = ty->l0[py], ly1 = ty->l1[py];
        const float4 a = *reinterpret_cast<const float4*>(tbuf + (r0 - sr0) * PW + 4 * cg);
        const float4 bq = *reinterpret_cast<const float4*>(tbuf + (r1 - sr0) * PW + 4 * cg);
        const int x = x0 - pad + 4 * cg;                   // columns outside the image stay -inf
        if ((unsigned)(x + 0) < (unsigned)w) o.x = __builtin_fmaf(a.x, ly0, bq.x * ly1);
        if ((unsigned)(x + 1) < (unsigned)w) o.y = __builtin_fmaf(a.y, ly0, bq.y * ly1);
        if ((unsigned)(x + 2) < (unsigned)w) o.z = __builtin_fmaf(a.z, ly0, bq.z * ly1);
        if ((unsigned)(x + 3) < (unsigned)w) o.w = __builtin_fmaf(a.w, ly0, bq.w * ly1);
      }
      *reinterpret_cast<float4*>(raw + py * PW + 4 * cg) = o;
    }
    __syncthreads();                                       // `stage` becomes the row-max buffer again
    return true;
  }
  // lane = column (its axis entry stays in registers), wave = every 4th row (its axis entry is wave-uniform):
  // per sample only the four taps and the result touch LDS - with one table look-up per sample and axis the
  // kernel was bound by the LDS instruction rate.  Columns 64.. of the padded tile go in one extra pass.
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = tx->i0[lane], c1 = tx->i1[lane];
    const float lx0 = tx->l0[lane], lx1 = tx->l1[lane];
    for (int py = wv; py < PH; py += 4) {
      const int r0 = __builtin_amdgcn_readfirstlane(ty->i0[py]), r1 = __builtin_amdgcn_readfirstlane(ty->i1[py]);
      const float ly0 = ty->l0[py], ly1 = ty->l1[py];
      raw[py * PW + lane] = sample(r0, r1, ly0, ly1, c0, c1, lx0, lx1);
    }
    const int extra = PW - 64;                             // 2 * pad columns
    for (int i = threadIdx.x; i < PH * extra; i += 256) {
      const int py = i / extra, px = 64 + i - py * extra;
      raw[py * PW + px] = sample(ty->i0[py], ty->i1[py], ty->l0[py], ty->l1[py], tx->i0[px], tx->i1[px], tx->l0[px],
                                 tx->l1[px]);
    }
  }
  if (staged) __syncthreads();                          // `stage` becomes the row-max buffer again
  return true;
}

template <class Map>
__device__ __forceinline__ bool nms_tile(const Map& m, int plane, int h, int w, int y0, int x0, int pad,
                                         float* raw, float* rowmax, AxisTab* ty, AxisTab* tx, int* pos_flag = nullptr,
                                         int rowmax_floats = (kTH + 2 * kMaxPad) * kTW, float* tbuf = nullptr) {
  // raw: (kTH+2p) x (kTW+2p) samples (-inf outside the image); rowmax: horizontal window max
  const int PW = kTW + 2 * pad, PH = kTH + 2 * pad;
  if (!fill_raw(m, plane, h, w, y0, x0, pad, raw, ty, tx, rowmax, rowmax_floats, pos_flag, tbuf)) return false;
  __syncthreads();
  if (pad == 2 && tbuf != nullptr) {
    // 5-wide window, 8 outputs per thread from 12 inputs (three 16-byte reads, two 16-byte writes) instead of
    // five 4-byte reads per output; max is exact, so any grouping gives the same bits
    for (int i = threadIdx.x; i < PH * (kTW / 8); i += 256) {
      const int py = i / (kTW / 8), g8 = i - py * (kTW / 8);
      const float4* src = reinterpret_cast<const float4*>(raw + py * PW + 8 * g8);
      const float4 q0 = src[0], q1 = src[1], q2 = src[2];
      const float x[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
      float pr[11], o[8];
#pragma unroll
      for (int j = 0; j < 11; ++j) pr[j] = fmaxf(x[j], x[j + 1]);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaxf(pr[j], pr[j + 2]), x[j + 4]);
      float4* dst = reinterpret_cast<float4*>(rowmax + py * kTW + 8 * g8);
      dst[0] = make_float4(o[0], o[1], o[2], o[3]);
      dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();
    return true;
  }
  for (int i = threadIdx.x; i < PH * kTW; i += 256) {
    const int py = i / kTW, px = i - py * kTW;
    float v = raw[py * PW + px];
    for (int d = 1; d <= 2 * pad; ++d) v = fmaxf(v, raw[py * PW + px + d]);
    rowmax[i] = v;
  }
  __syncthreads();
  return true;
}

template <class Map>
__global__ void __launch_bounds__(256) nms_kernel(Map m, int h, int w, int pad, float* out) {
  __shared__ float raw[(kTH + 2 * kMaxPad) * (kTW + 2 * kMaxPad)];
  __shared__ float rowmax[(kTH + 2 * kMaxPad) * kTW];
  const int tiles_x = (w + kTW - 1) / kTW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, plane = blockIdx.y;
  const int y0 = ty * kTH, x0 = tx * kTW;
  __shared__ AxisTab taby, tabx;
  nms_tile(m, plane, h, w, y0, x0, pad, raw, rowmax, &taby, &tabx);
  const int PW = kTW + 2 * pad;
  for (int i = threadIdx.x; i < kTH * kTW; i += 256) {
    const int ly = i / kTW, lx = i - ly * kTW;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= h || x >= w) continue;
    float mx = rowmax[ly * kTW + lx];
    for (int d = 1; d <= 2 * pad; ++d) mx = fmaxf(mx, rowmax[(ly + d) * kTW + lx]);
    const float v = raw[(ly + pad) * PW + lx + pad];
    out[((size_t)plane * h + y) * w + x] = v * (mx == v ? 1.f : 0.f);   // det * (maxm == det).float()
  }
}

// ---------------------------------------------------------------------------
// top-K, phase 1: per tile, the K best positive local maxima as sorted keys
// ---------------------------------------------------------------------------
template <class Map, int PAD>     // PAD >= 0: the NMS padding as a compile-time constant (divisions by PW become shifts/muls)
__global__ void __launch_bounds__(256) topk_tile_kernel(Map m, int h, int w, int pad_rt, int K, u64* cand, int fast,
                                                        int planes, int tiles) {
  const int pad = PAD >= 0 ? PAD : pad_rt;
  // Workgroups are dealt to the 8 XCDs round robin (blockIdx.x % 8).  An XCD takes whole planes and walks their
  // tiles in row-major order: horizontally and vertically adjacent tiles share the 128-byte lines at the edges of
  // their source rectangles (36 floats per row of a 2x upsampled map: 2-3 lines for 1.1 lines of payload), and with
  // (tile, plane) as the grid they were fetched once per XCD's L2 - 758 MB per batch for 279 MB of maps (PMC, round 2).
  const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
  const int plane_l = slot / tiles;
  const int tile = slot - plane_l * tiles;
  const int plane = plane_l * 8 + xcd;
  if (plane >= planes) return;                             // grid padding (whole workgroup)
  constexpr int kP = PAD >= 0 ? PAD : kMaxPad;           // the common 5x5 window needs 19 KiB of tiles, not 21.8: one more block per CU
  __shared__ __attribute__((aligned(16))) float raw[(kTH + 2 * kP) * (kTW + 2 * kP)];
  __shared__ __attribute__((aligned(16))) float rowmax[(kTH + 2 * kP) * kTW];
  constexpr bool kFast = PAD == 2;                       // 5x5 window: separable sampling, register-blocked max passes
  __shared__ __attribute__((aligned(16))) float tbuf_s[kFast ? kTRows * (kTW + 4) : 4];
  float* const tbuf = kFast && fast ? tbuf_s : nullptr;
  __shared__ u64 red[4];
  const int tiles_x = (w + kTW - 1) / kTW;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * kTH, x0 = tx * kTW;
  __shared__ AxisTab taby, tabx;
  __shared__ u64 clist[kMaxCand];
  __shared__ int ccount, any_positive;
  if (threadIdx.x == 0) { ccount = 0; any_positive = 0; }
  if (!nms_tile(m, plane, h, w, y0, x0, pad, raw, rowmax, &taby, &tabx, &any_positive, (kTH + 2 * kP) * kTW, tbuf)) {
    u64* outp0 = cand + ((size_t)plane * tiles + tile) * K;                 // nothing positive under this tile
    for (int r = threadIdx.x; r < K; r += 256) outp0[r] = 0;
    return;
  }
  const int PW = kTW + 2 * pad;
  u64 mine[8];   // this thread's 8 pixels as keys (0 = not a positive local maximum)
  if (tbuf != nullptr) {
    // thread = 2 rows x 4 columns: six 16-byte reads of the row maxima give both vertical windows
    const int ry = threadIdx.x >> 4, cg = threadIdx.x & 15;
    const int ly = 2 * ry, lx = 4 * cg;
    float4 rm[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) rm[d] = *reinterpret_cast<const float4*>(rowmax + (ly + d) * kTW + lx);
    const float mid[4] = {fmaxf(fmaxf(rm[1].x, rm[2].x), fmaxf(rm[3].x, rm[4].x)), fmaxf(fmaxf(rm[1].y, rm[2].y), fmaxf(rm[3].y, rm[4].y)),
                          fmaxf(fmaxf(rm[1].z, rm[2].z), fmaxf(rm[3].z, rm[4].z)), fmaxf(fmaxf(rm[1].w, rm[2].w), fmaxf(rm[3].w, rm[4].w))};
    const float top[4] = {rm[0].x, rm[0].y, rm[0].z, rm[0].w}, bot[4] = {rm[5].x, rm[5].y, rm[5].z, rm[5].w};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const float2* c2 = reinterpret_cast<const float2*>(raw + (ly + rr + 2) * PW + lx + 2);   // 8-byte aligned
      const float2 va = c2[0], vb = c2[1];
      const float v4[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float mx = fmaxf(mid[j], rr == 0 ? top[j] : bot[j]);
        const int y = y0 + ly + rr, x = x0 + lx + j;
        u64 key = 0;
        if (y < h && x < w && mx == v4[j] && v4[j] > 0.f) key = make_key(v4[j], (unsigned)(y * w + x));
        mine[rr * 4 + j] = key;
        if (key != 0) {
          const int pos = atomicAdd(&ccount, 1);
          if (pos < kMaxCand) clist[pos] = key;
        }
      }
    }
  } else {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = threadIdx.x + q * 256;
    const int ly = i / kTW, lx = i - ly * kTW;
    const int y = y0 + ly, x = x0 + lx;
    u64 key = 0;
    if (y < h && x < w) {
      float mx = rowmax[ly * kTW + lx];
      for (int d = 1; d <= 2 * pad; ++d) mx = fmaxf(mx, rowmax[(ly + d) * kTW + lx]);
      const float v = raw[(ly + pad) * PW + lx + pad];
      if (mx == v && v > 0.f) key = make_key(v, (unsigned)(y * w + x));
    }
    mine[q] = key;
    if (key != 0) {                 // compact the (few) local maxima of the tile
      const int pos = atomicAdd(&ccount, 1);
      if (pos < kMaxCand) clist[pos] = key;
    }
  }
  }
  __syncthreads();
  u64* outp = cand + ((size_t)plane * tiles + tile) * K;
  const int nc = ccount;
  if (nc <= kMaxCand) {
    // rank sort: keys are unique (the pixel index is part of the key), so the rank of a key is
    // the number of larger ones; one pass, no further barriers
    for (int t = threadIdx.x; t < nc; t += 256) {
      const u64 key = clist[t];
      int rank = 0;
      for (int j = 0; j < nc; ++j) rank += clist[j] > key ? 1 : 0;
      if (rank < K) outp[rank] = key;
    }
    for (int r = nc + threadIdx.x; r < K; r += 256) outp[r] = 0;
    return;
  }
  // plateau-heavy tile (more local maxima than the list holds): K rounds of a block-wide max
  for (int k = 0; k < K; ++k) {
    u64 best = mine[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) best = mine[q] > best ? mine[q] : best;
    best = block_max(best, red);
    if (threadIdx.x == 0) outp[k] = best;
    if (best == 0) {              // exhausted: the rest of the list is empty
      for (int r = k + 1 + threadIdx.x; r < K; r += 256) outp[r] = 0;
      break;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (mine[q] == best) mine[q] = 0;
  }
}

// ---------------------------------------------------------------------------
// top-K, phase 2: merge the per-tile lists of one plane, gather tags, pad with
// zero-valued pixels in index order (what a stable top-k of the NMS map gives)
// ---------------------------------------------------------------------------
template <class Map>
__device__ float nms_value_at(const Map& m, int plane, int h, int w, int pad, int y, int x) {
  const float v = m.at(plane, y, x);
  float mx = v;
  for (int dy = -pad; dy <= pad; ++dy)
    for (int dx = -pad; dx <= pad; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) mx = fmaxf(mx, m.at(plane, yy, xx));
    }
  return v * (mx == v ? 1.f : 0.f);
}

template <class Map, class TagMap>
__global__ void __launch_bounds__(256) topk_merge_kernel(Map m, TagMap tm, int tag_shared_joints, int D,
                                                         int h, int w, int pad, int K, int tiles,
                                                         u64* cand, float* val_k, int* ind_k, float* tag_k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  u64* red = reinterpret_cast<u64*>(smem_raw);          // 4 entries
  u64* lkeys = red + 4;
  const int plane = blockIdx.x;
  const int n = tiles * K;
  u64* gk = cand + (size_t)plane * n;
  constexpr int kOwn = 8;
  const bool merge_heads = tiles <= kOwn * 256;
  const bool use_lds = !merge_heads && (size_t)n * 8 + 32 <= 150 * 1024;
  u64* keys = gk;                                         // flat pointer: LDS copy when it fits
  if (use_lds) {
    for (int i = threadIdx.x; i < n; i += 256) lkeys[i] = gk[i];
    __syncthreads();
    keys = lkeys;
  }
  const int tag_plane = tag_shared_joints > 0 ? plane / tag_shared_joints : plane;
  int found = 0;
  if (merge_heads) {
    // every tile list is sorted (largest first, zeros behind): only the heads compete.  A thread keeps the
    // head of each of its (<= kOwn) tiles in registers and re-reads one key after it won a round
    u64 head[kOwn];
    int pos[kOwn];
#pragma unroll
    for (int o = 0; o < kOwn; ++o) {
      const int t = threadIdx.x + o * 256;
      pos[o] = 0;
      head[o] = t < tiles ? gk[(size_t)t * K] : 0;
    }
    for (int k = 0; k < K; ++k) {
      u64 mine = head[0];
#pragma unroll
      for (int o = 1; o < kOwn; ++o) mine = head[o] > mine ? head[o] : mine;
      const u64 best = block_max(mine, red);
      if (best == 0) break;
#pragma unroll
      for (int o = 0; o < kOwn; ++o)
        if (head[o] == best) {                               // keys are unique: exactly one owner
          ++pos[o];
          head[o] = pos[o] < K ? gk[(size_t)(threadIdx.x + o * 256) * K + pos[o]] : 0;
        }
      if (threadIdx.x == 0) {
        const unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
        val_k[(size_t)plane * K + k] = unorder_bits((unsigned)(best >> 32));
        ind_k[(size_t)plane * K + k] = (int)idx;
      }
      ++found;
      __syncthreads();
    }
  } else
  for (int k = 0; k < K; ++k) {
    u64 best = 0;
    for (int i = threadIdx.x; i < n; i += 256) best = keys[i] > best ? keys[i] : best;
    best = block_max(best, red);
    if (best == 0) break;
    for (int i = threadIdx.x; i < n; i += 256)
      if (keys[i] == best) keys[i] = 0;
    if (threadIdx.x == 0) {
      const unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffu);
      val_k[(size_t)plane * K + k] = unorder_bits((unsigned)(best >> 32));
      ind_k[(size_t)plane * K + k] = (int)idx;
    }
    ++found;
    __syncthreads();
  }
  // zero padding: the first K - found pixels, in index order, whose NMS value is zero.  256 pixels per round,
  // one per thread (a round almost always suffices); each thread's rank among the zero-valued ones comes from
  // a ballot + the wave totals.  (One thread walking the pixels cost 200 us per batch: 25 bilinear samples per
  // pixel, one after the other, in every plane with fewer than K positive maxima - most planes.)
  {
    int* wtot = reinterpret_cast<int*>(red);               // 4 wave totals (red is free here)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int total_px = h * w;
    __syncthreads();
    for (int base = 0; found < K && base < total_px; base += 256) {
      const int idx = base + threadIdx.x;
      bool z = false;
      if (idx < total_px) z = nms_value_at(m, plane, h, w, pad, idx / w, idx - (idx / w) * w) == 0.f;
      const unsigned long long mask = __ballot(z);
      if (lane == 0) wtot[wv] = __popcll(mask);
      __syncthreads();
      int before = __popcll(mask & ((1ull << lane) - 1ull));
      int total = 0;
      for (int i = 0; i < 4; ++i) { before += i < wv ? wtot[i] : 0; total += wtot[i]; }
      const int k = found + before;
      if (z && k < K) {
        val_k[(size_t)plane * K + k] = 0.f;
        ind_k[(size_t)plane * K + k] = idx;
      }
      found = found + total < K ? found + total : K;
      __syncthreads();
    }
    for (int k = found + threadIdx.x; k < K; k += 256) { val_k[(size_t)plane * K + k] = 0.f; ind_k[(size_t)plane * K + k] = 0; }
  }
  __syncthreads();
  __threadfence_block();
  for (int i = threadIdx.x; i < K * D; i += 256) {
    const int k = i / D, d = i - k * D;
    const int idx = ind_k[(size_t)plane * K + k];
    tag_k[((size_t)plane * K + k) * D + d] = tm.at(tag_plane, idx / w, idx - (idx / w) * w, d);
  }
}

struct DirectTag {          // (planes, h, w, D)
  const float* p;
  int h, w, D;
  __device__ __forceinline__ float at(int plane, int y, int x, int d) const {
    return p[(((size_t)plane * h + y) * w + x) * D + d];
  }
};
struct BilinearTag {        // D == 1
  BilinearMap m;
  __device__ __forceinline__ float at(int plane, int y, int x, int) const { return m.at(plane, y, x); }
};

// ---------------------------------------------------------------------------
// adjust + refine: one workgroup per (person, joint)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float pairwise8_sum(const float* a, int n) {   // numpy contiguous f32 add.reduce
  if (n < 8) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
  float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s = s + a[i];
  return s;
}

constexpr int kMaxJ = 32, kMaxD = 32;
constexpr int kRefineGroup = 16;     // people handled per pass over the pixels
constexpr int kRefineStripes = 10;   // row stripes per (image, joint) map (refine scan)
constexpr int kArgmaxStripes = 20;   // plane-maximum pass: half the rows per block = half the staged source rows (24 KiB:
                                     // six blocks per CU instead of three)

// (1) one block per person: copy rows, adjust detected joints (group.py:181-200),
//     mean tag of the detected joints (group.py:214-222), score (group.py:272)
template <class Map>
__global__ void __launch_bounds__(64) adjust_prepare_kernel(Map m, int J, int h, int w, int D,
                                                            const float* ans_in, float* ans_out,
                                                            const int* person_img, int do_adjust,
                                                            float* scores, float* mean_tag) {
  const int p = blockIdx.x, j = threadIdx.x;
  const int row_len = 3 + D;
  const float* kp = ans_in + (size_t)p * J * row_len;
  const int img = person_img ? person_img[p] : 0;
  if (j < J) {
    float* out = ans_out + ((size_t)p * J + j) * row_len;
    for (int c = 0; c < row_len; ++c) out[c] = kp[j * row_len + c];
    if (do_adjust && kp[j * row_len + 2] > 0.f) {
      const int plane = img * J + j;
      float cx = kp[j * row_len + 0], cy = kp[j * row_len + 1];
      const int col = (int)cx, row = (int)cy;
      const int cr = col + 1 < w - 1 ? col + 1 : w - 1, cl = col - 1 > 0 ? col - 1 : 0;
      const int rd = row + 1 < h - 1 ? row + 1 : h - 1, ru = row - 1 > 0 ? row - 1 : 0;
      cx += m.at(plane, row, cr) > m.at(plane, row, cl) ? 0.25f : -0.25f;
      cy += m.at(plane, rd, col) > m.at(plane, ru, col) ? 0.25f : -0.25f;
      out[0] = cx + 0.5f;
      out[1] = cy + 0.5f;
    }
  }
  if (j == 0) {
    float v[kMaxJ];
    if (scores) {
      for (int q = 0; q < J; ++q) v[q] = kp[q * row_len + 2];
      scores[p] = pairwise8_sum(v, J) / (float)J;
    }
    for (int d = 0; d < D; ++d) {
      int n = 0;
      for (int q = 0; q < J; ++q)
        if (kp[q * row_len + 2] > 0.f) v[n++] = kp[q * row_len + 3 + d];
      float s;
      if (D == 1) {
        s = pairwise8_sum(v, n);
      } else {
        s = 0.f;
        for (int i = 0; i < n; ++i) s = s + v[i];
      }
      mean_tag[(size_t)p * D + d] = s / (float)n;
    }
  }
}

// (1b) shortcut for the arg-max of refine (group.py:229-233).  The score of person p at pixel x is
//      det(x) - round(|tag(x) - mean_p|) <= det(x).  Let q* be the FIRST pixel attaining max det of the
//      plane: if the penalty of p at q* is 0, then q* is exactly np.argmax of p's score map (no pixel
//      can score higher than max det, and one that ties it is a later det maximum).  Only the
//      (person, joint) pairs whose penalty at q* is not 0 need the full scan below.
template <class Map>
__device__ __forceinline__ void argmax_rows(const Map& m, int plane, int w, int y_begin, int y_end, float* bv,
                                            unsigned* bi, float*, int) {
  const int npix = (y_end - y_begin) * w;
  int y = y_begin + (int)(threadIdx.x / (unsigned)w), x = (int)(threadIdx.x % (unsigned)w);
  const int dy = 256 / w, dx = 256 - dy * w;
  for (int i = threadIdx.x; i < npix; i += 256) {
    const float dv = m.at(plane, y, x);
    const bool up = dv > *bv || *bi == 0xffffffffu;          // increasing index order: '>' keeps the first maximum
    *bv = up ? dv : *bv;
    *bi = up ? (unsigned)(y * w + x) : *bi;
    x += dx; y += dy;
    if (x >= w) { x -= w; y += 1; }
  }
}

// bilinear maps: a thread owns up to 8 fixed columns (their x-axis taps and weights stay in registers),
// the y-axis taps of a row are the same for the whole block; per pixel 4 loads + 5 flops remain
template <>
__device__ __forceinline__ void argmax_rows<BilinearMap>(const BilinearMap& m, int plane, int w, int y_begin,
                                                         int y_end, float* bv, unsigned* bi, float* lds,
                                                         int lds_floats) {
  constexpr int kCols = 8;
  if (w > 256 * kCols) {                                      // very wide maps: the generic walk
    const int npix = (y_end - y_begin) * w;
    int y = y_begin + (int)(threadIdx.x / (unsigned)w), x = (int)(threadIdx.x % (unsigned)w);
    const int dy = 256 / w, dx = 256 - dy * w;
    for (int i = threadIdx.x; i < npix; i += 256) {
      const float dv = m.at(plane, y, x);
      const bool up = dv > *bv || *bi == 0xffffffffu;
      *bv = up ? dv : *bv;
      *bi = up ? (unsigned)(y * w + x) : *bi;
      x += dx; y += dy;
      if (x >= w) { x -= w; y += 1; }
    }
    return;
  }
  const int n = plane / m.J, j = plane - n * m.J;
  const float* b = m.p + (size_t)n * m.img_stride + (size_t)j * m.sh * m.sw;
  // the source rows of the stripe are contiguous in memory: one coalesced copy into LDS, then the four
  // taps of every pixel come from there (the gather of 4-byte taps through L1 was the whole cost)
  int sr0, sr1, t0i, t1i;
  float tf0, tf1;
  m.ay.at(y_begin, &sr0, &t0i, &tf0, &tf1);
  m.ay.at(y_end - 1, &t1i, &sr1, &tf0, &tf1);
  const int nsrc = (sr1 - sr0 + 1) * m.sw;
  const bool staged = lds != nullptr && nsrc > 0 && nsrc <= lds_floats;
  if (staged) {
    const float* g = b + sr0 * m.sw;
    for (int i = threadIdx.x; i < nsrc; i += 256) lds[i] = g[i];
    __syncthreads();
  }
  int c0[kCols], c1[kCols];
  float lx0[kCols], lx1[kCols];
#pragma unroll
  for (int k = 0; k < kCols; ++k) {
    const int x = threadIdx.x + k * 256;
    c0[k] = c1[k] = 0; lx0[k] = lx1[k] = 0.f;
    if (x < w) m.ax.at(x, &c0[k], &c1[k], &lx0[k], &lx1[k]);
  }
  for (int y = y_begin; y < y_end; ++y) {
    int r0, r1;
    float ly0, ly1;
    m.ay.at(y, &r0, &r1, &ly0, &ly1);
    const int o0 = (staged ? r0 - sr0 : r0) * m.sw, o1 = (staged ? r1 - sr0 : r1) * m.sw;
#pragma unroll
    for (int k = 0; k < kCols; ++k) {
      const int x = threadIdx.x + k * 256;
      if (x < w) {
        float v00, v01, v10, v11;
        if (staged) {                                         // uniform branch: LDS taps
          v00 = lds[o0 + c0[k]]; v01 = lds[o0 + c1[k]]; v10 = lds[o1 + c0[k]]; v11 = lds[o1 + c1[k]];
        } else {
          v00 = b[o0 + c0[k]]; v01 = b[o0 + c1[k]]; v10 = b[o1 + c0[k]]; v11 = b[o1 + c1[k]];
        }
        const float t0 = __builtin_fmaf(v00, lx0[k], v01 * lx1[k]);
        const float t1 = __builtin_fmaf(v10, lx0[k], v11 * lx1[k]);
        const float dv = __builtin_fmaf(t0, ly0, t1 * ly1);
        const bool up = dv > *bv || *bi == 0xffffffffu;      // a thread's pixels come in increasing index order
        *bv = up ? dv : *bv;
        *bi = up ? (unsigned)(y * w + x) : *bi;
      }
    }
  }
}

template <class Map>
__global__ void __launch_bounds__(256) plane_argmax_kernel(Map m, int h, int w, u64* plane_key, int lds_floats,
                                                           const unsigned char* known) {
  extern __shared__ __attribute__((aligned(16))) float argmax_src[];
  __shared__ u64 red[4];
  const int plane = blockIdx.y;
  if (known != nullptr && known[plane]) return;            // its maximum came with the top-k table
  const int rows = (h + gridDim.x - 1) / gridDim.x;
  const int y_begin = blockIdx.x * rows, y_end = min(h, y_begin + rows);
  if (y_begin >= y_end) return;
  float bv = -INFINITY;
  unsigned bi = 0xffffffffu;
  argmax_rows(m, plane, w, y_begin, y_end, &bv, &bi, lds_floats > 0 ? argmax_src : nullptr, lds_floats);
  const u64 k = block_max(bi == 0xffffffffu ? 0 : make_key(bv, bi), red);
  if (threadIdx.x == 0 && k != 0) atomicMax(&plane_key[plane], k);
}

// the first pixel that attains a plane's maximum, taken from the top-k table of the same map (possibly in pinned
// host memory: one read per plane here, not one per missing joint): its first entry - the best positive 5x5
// local maximum, ties by lowest index - IS that pixel whenever the maximum is positive; otherwise 0 = full scan
__global__ void __launch_bounds__(256) plane_key_from_topk_kernel(const float* topk_val, const int* topk_ind, int K,
                                                                  int n_planes, u64* plane_key, unsigned char* known) {
  const int plane = blockIdx.x * 256 + threadIdx.x;
  if (plane >= n_planes) return;
  const float v = topk_val[(size_t)plane * K];
  plane_key[plane] = v > 0.f ? make_key(v, (unsigned)topk_ind[(size_t)plane * K]) : 0;
  known[plane] = v > 0.f;                                  // the others (nowhere positive) still need the pass
}

template <class TagMap>
__global__ void __launch_bounds__(256) refine_shortcut_kernel(TagMap tm, int J, int w, int D, const float* ans_in,
                                                              const int* person_img, int P, const float* mean_tag,
                                                              const u64* plane_key, u64* best_key,
                                                              unsigned char* need_scan) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P * J) return;
  const int p = i / J, j = i - p * J;
  const int row_len = 3 + D;
  need_scan[i] = 0;
  if (ans_in[(size_t)i * row_len + 2] != 0.f) return;       // only missing joints are refined
  const int plane = (person_img ? person_img[p] : 0) * J + j;
  const u64 key = plane_key[plane];
  if (key == 0) { need_scan[i] = 1; return; }
  const int idx = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
  const int y = idx / w, x = idx - y * w;
  float ss;                                                  // the exact expressions of refine_scan_kernel
  if (D == 1) {
    const float d0 = tm.at(plane, y, x, 0) - mean_tag[p];
    ss = d0 * d0;
  } else if (D < 8) {
    const float d0 = tm.at(plane, y, x, 0) - mean_tag[(size_t)p * D];
    ss = d0 * d0;
    for (int d = 1; d < D; ++d) { const float dd = tm.at(plane, y, x, d) - mean_tag[(size_t)p * D + d]; ss = ss + dd * dd; }
  } else {
    float sq[kMaxD];
    for (int d = 0; d < D; ++d) { const float dd = tm.at(plane, y, x, d) - mean_tag[(size_t)p * D + d]; sq[d] = dd * dd; }
    ss = pairwise8_sum(sq, D);
  }
  if (rintf(sqrtf(ss)) == 0.f) best_key[i] = key;            // score = det - 0: the key of the plane maximum
  else need_scan[i] = 1;
}

// (2) one block per (image, joint, row stripe): every person of the image that
//     misses this joint is scored against every pixel of the stripe; the map is
//     sampled once per pixel for all of them.  arg-max keys are merged with
//     atomicMax (value bits << 32 | ~index: the first maximum wins, as np.argmax)
template <class Map, class TagMap, bool kD1>
__global__ void __launch_bounds__(256) refine_scan_kernel(Map m, TagMap tm, int J, int h, int w, int D,
                                                          const float* ans_in, const int* person_img, int P,
                                                          const float* mean_tag, u64* best_key,
                                                          const unsigned char* need_scan) {
  __shared__ int lo_hi[2];
  __shared__ int n_need;
  __shared__ int need[512];
  __shared__ u64 red[kRefineGroup];
  __shared__ float gmean[kRefineGroup * kMaxD];
  const int plane = blockIdx.y, img = plane / J, j = plane - img * J;
  const int row_len = 3 + D;
  if (threadIdx.x == 0) {
    int lo = 0, hi = P;                      // persons are sorted by image: [lo, hi) = this image
    if (person_img) {
      int a = 0, b = P;
      while (a < b) { const int c = (a + b) >> 1; if (person_img[c] < img) a = c + 1; else b = c; }
      lo = a; b = P;
      while (a < b) { const int c = (a + b) >> 1; if (person_img[c] <= img) a = c + 1; else b = c; }
      hi = a;
    } else if (img != 0) {
      hi = 0;
    }
    lo_hi[0] = lo; lo_hi[1] = hi;
    n_need = 0;
  }
  __syncthreads();
  const int lo = lo_hi[0], hi = lo_hi[1];
  const int rows = (h + gridDim.x - 1) / gridDim.x;
  const int y_begin = blockIdx.x * rows, y_end = min(h, y_begin + rows);
  if (y_begin >= y_end) return;
  const int npix = (y_end - y_begin) * w;
  for (int base = lo; base < hi; base += 512) {          // people needing joint j, 512 at a time
    __syncthreads();
    if (threadIdx.x == 0) n_need = 0;
    __syncthreads();
    for (int p = base + threadIdx.x; p < min(hi, base + 512); p += 256)
      if (ans_in[((size_t)p * J + j) * row_len + 2] == 0.f && (need_scan == nullptr || need_scan[(size_t)p * J + j]))
        need[atomicAdd(&n_need, 1)] = p;
    __syncthreads();
    const int nn = n_need;
    for (int g0 = 0; g0 < nn; g0 += kRefineGroup) {
      const int gn = min(kRefineGroup, nn - g0);
      __syncthreads();
      if (threadIdx.x < kRefineGroup) red[threadIdx.x] = 0;
      for (int i = threadIdx.x; i < gn * D; i += 256)
        gmean[(i / D) * kMaxD + i % D] = mean_tag[(size_t)need[g0 + i / D] * D + i % D];
      __syncthreads();
      // per-thread running arg-max: pixels are visited in increasing index order, so a
      // strict '>' keeps the first maximum (np.argmax, group.py:233)
      float bscore[kRefineGroup];
      unsigned bidx[kRefineGroup];
#pragma unroll
      for (int q = 0; q < kRefineGroup; ++q) { bscore[q] = -INFINITY; bidx[q] = 0xffffffffu; }
      if (kD1) {
        // D == 1 (what validate_hhrnet.py passes): branch-free over the 16 slots of the
        // group; unused slots (>= gn) carry mean 0 and are dropped afterwards
        float gm[kRefineGroup];
#pragma unroll
        for (int q = 0; q < kRefineGroup; ++q) gm[q] = q < gn ? gmean[q * kMaxD] : 0.f;
        int y = y_begin + (int)(threadIdx.x / (unsigned)w), x = (int)(threadIdx.x % (unsigned)w);
        const int dy = 256 / w, dx = 256 - dy * w;          // advance of 256 pixels in (y, x)
        for (int i = threadIdx.x; i < npix; i += 256) {
          const float dv = m.at(plane, y, x);
          const float tv = tm.at(plane, y, x, 0);
          const unsigned idx = (unsigned)(y * w + x);
          bool slow = false;
          float kk[kRefineGroup];
#pragma unroll
          for (int q = 0; q < kRefineGroup; ++q) {
            const float a = fabsf(tv - gm[q]);
            kk[q] = rintf(a);
            // round(sqrt(fl(d*d))) == round(|d|) unless |d| sits within a few ulp of a
            // half-integer (sqrt(fl(a*a)) is within 1.5*2^-24 relative of a)
            slow |= fabsf(fabsf(a - kk[q]) - 0.5f) <= a * 4e-7f;
          }
          if (slow) {                                         // rare: the exact expression
#pragma unroll
            for (int q = 0; q < kRefineGroup; ++q) {
              const float d0 = tv - gm[q];
              kk[q] = rintf(sqrtf(d0 * d0));
            }
          }
#pragma unroll
          for (int q = 0; q < kRefineGroup; ++q) {
            const float score = dv - kk[q];
            const bool up = score > bscore[q] || bidx[q] == 0xffffffffu;
            bscore[q] = up ? score : bscore[q];
            bidx[q] = up ? idx : bidx[q];
          }
          x += dx; y += dy;
          if (x >= w) { x -= w; y += 1; }
        }
      } else {
        for (int i = threadIdx.x; i < npix; i += 256) {
          const int yy = i / w, x = i - yy * w, y = y_begin + yy;
          const float dv = m.at(plane, y, x);
          const unsigned idx = (unsigned)(y * w + x);
          float tv[kMaxD];
          for (int d = 0; d < D; ++d) tv[d] = tm.at(plane, y, x, d);
          for (int q = 0; q < gn; ++q) {
            float ss;
            if (D < 8) {
              const float d0 = tv[0] - gmean[q * kMaxD];
              ss = d0 * d0;
              for (int d = 1; d < D; ++d) { const float dd = tv[d] - gmean[q * kMaxD + d]; ss = ss + dd * dd; }
            } else {
              float sq[kMaxD];
              for (int d = 0; d < D; ++d) { const float dd = tv[d] - gmean[q * kMaxD + d]; sq[d] = dd * dd; }
              ss = pairwise8_sum(sq, D);
            }
            const float score = dv - rintf(sqrtf(ss));
#pragma unroll
            for (int u = 0; u < kRefineGroup; ++u)            // static register indexing
              if (u == q && (score > bscore[u] || bidx[u] == 0xffffffffu)) { bscore[u] = score; bidx[u] = idx; }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kRefineGroup; ++q) {
        if (q < gn) {
          const u64 mine = bidx[q] == 0xffffffffu ? 0 : make_key(bscore[q], bidx[q]);
          const u64 k = wave_max(mine);
          if ((threadIdx.x & 63) == 0) atomicMax(&red[q], k);
        }
      }
      __syncthreads();
      if (threadIdx.x < gn)
        atomicMax(&best_key[(size_t)need[g0 + threadIdx.x] * J + j], red[threadIdx.x]);
    }
  }
}

// (3) fill the joints that were missing and whose arg-max has a positive value (group.py:233-262)
template <class Map>
__global__ void __launch_bounds__(256) refine_finalize_kernel(Map m, int J, int h, int w, int D,
                                                              const float* ans_in, float* ans_out,
                                                              const int* person_img, int P, const u64* best_key) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P * J) return;
  const int p = i / J, j = i - p * J;
  const int row_len = 3 + D;
  if (ans_in[(size_t)i * row_len + 2] != 0.f) return;
  const u64 key = best_key[i];
  if (key == 0) return;
  const int plane = (person_img ? person_img[p] : 0) * J + j;
  const int idx = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
  const int y = idx / w, x = idx - y * w;
  const float v = m.at(plane, y, x);
  if (v > 0.f) {
    const int xr = x + 1 < w - 1 ? x + 1 : w - 1, xl = x - 1 > 0 ? x - 1 : 0;
    const int yd = y + 1 < h - 1 ? y + 1 : h - 1, yu = y - 1 > 0 ? y - 1 : 0;
    float* out = ans_out + (size_t)i * row_len;
    out[0] = (float)x + 0.5f + (m.at(plane, y, xr) > m.at(plane, y, xl) ? 0.25f : -0.25f);
    out[1] = (float)y + 0.5f + (m.at(plane, yd, x) > m.at(plane, yu, x) ? 0.25f : -0.25f);
    out[2] = v;
  }
}

constexpr int kShortcutPlanes = 1 << 16;   // plane maxima kept for the arg-max shortcut (more planes: full scans only)
static size_t refine_scratch(int P, int J, int D) {
  return (size_t)P * J * sizeof(u64) + (size_t)P * D * sizeof(float) + 256 + (size_t)kShortcutPlanes * sizeof(u64) +
         (size_t)P * J + kShortcutPlanes;                  // + need_scan flags + per-plane "maximum known" flags
}

template <class Map, class TagMap>
static int adjust_refine_run(const Map& m, const TagMap& tm, int n_img, int J, int h, int w, int D,
                             const float* ans_in, float* ans_out, const int* person_img, int P, int do_adjust,
                             int do_refine, float* scores, void* scratch, size_t scratch_bytes, hipStream_t s,
                             const float* topk_val = nullptr, const int* topk_ind = nullptr, int topk_k = 0) {
  RTPE_REQUIRE(scratch && scratch_bytes >= refine_scratch(P, J, D), "adjust_refine: scratch too small");
  u64* best_key = reinterpret_cast<u64*>(scratch);
  float* mean_tag = reinterpret_cast<float*>(best_key + (size_t)P * J);
  hipLaunchKernelGGL((adjust_prepare_kernel<Map>), dim3(P), dim3(64), 0, s, m, J, h, w, D, ans_in, ans_out,
                     person_img, do_adjust, scores, mean_tag);
  RTPE_HIP_CHECK(hipGetLastError());
  if (!do_refine) return RTPE_OK;
  RTPE_HIP_CHECK(hipMemsetAsync(best_key, 0, (size_t)P * J * sizeof(u64), s));
  unsigned char* need_scan = nullptr;
  if (n_img * J <= kShortcutPlanes) {
    char* tail = reinterpret_cast<char*>(mean_tag + (size_t)P * D);
    u64* plane_key = reinterpret_cast<u64*>(tail + ((256 - ((uintptr_t)tail & 255)) & 255));
    need_scan = reinterpret_cast<unsigned char*>(plane_key + kShortcutPlanes);
    unsigned char* known = nullptr;
    if (topk_val != nullptr) {
      known = need_scan + (size_t)P * J;
      hipLaunchKernelGGL(plane_key_from_topk_kernel, dim3((n_img * J + 255) / 256), dim3(256), 0, s, topk_val, topk_ind,
                         topk_k, n_img * J, plane_key, known);
    } else {
      RTPE_HIP_CHECK(hipMemsetAsync(plane_key, 0, (size_t)n_img * J * sizeof(u64), s));
    }
    hipLaunchKernelGGL((plane_argmax_kernel<Map>), dim3(kArgmaxStripes, n_img * J), dim3(256), 24 * 1024, s, m, h, w,
                       plane_key, 24 * 1024 / 4, known);
    RTPE_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((refine_shortcut_kernel<TagMap>), dim3((P * J + 255) / 256), dim3(256), 0, s, tm, J, w, D, ans_in,
                       person_img, P, mean_tag, plane_key, best_key, need_scan);
    RTPE_HIP_CHECK(hipGetLastError());
  }
  if (D == 1)
    hipLaunchKernelGGL((refine_scan_kernel<Map, TagMap, true>), dim3(kRefineStripes, n_img * J), dim3(256), 0, s,
                       m, tm, J, h, w, D, ans_in, person_img, P, mean_tag, best_key, need_scan);
  else
    hipLaunchKernelGGL((refine_scan_kernel<Map, TagMap, false>), dim3(kRefineStripes, n_img * J), dim3(256), 0, s,
                       m, tm, J, h, w, D, ans_in, person_img, P, mean_tag, best_key, need_scan);
  RTPE_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL((refine_finalize_kernel<Map>), dim3((P * J + 255) / 256), dim3(256), 0, s, m, J, h, w, D,
                     ans_in, ans_out, person_img, P, best_key);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// ---------------------------------------------------------------------------
// host wrappers
// ---------------------------------------------------------------------------
static BilinearMap make_bilinear(const float* p, int sh, int sw, long long img_stride, int J, int oh, int ow) {
  BilinearMap m;
  m.p = p; m.sh = sh; m.sw = sw; m.J = J; m.img_stride = img_stride;
  m.ay = make_axis(sh, oh);
  m.ax = make_axis(sw, ow);
  return m;
}

static size_t topk_scratch(int planes, int h, int w, int K) {
  const size_t tiles = (size_t)((h + kTH - 1) / kTH) * ((w + kTW - 1) / kTW);
  return (size_t)planes * tiles * K * sizeof(u64);
}

template <class Map, class TagMap>
static int topk_run(const Map& m, const TagMap& tm, int planes, int tag_shared_joints, int D, int h, int w,
                    int K, int ksize, int pad, float* val_k, int* ind_k, float* tag_k, void* scratch,
                    size_t scratch_bytes, hipStream_t s) {
  RTPE_REQUIRE(ksize == 2 * pad + 1 && pad >= 0 && pad <= kMaxPad, "nms: ksize=%d pad=%d unsupported", ksize, pad);
  RTPE_REQUIRE(planes > 0 && h > 0 && w > 0 && K > 0 && (size_t)h * w < 0x7fffffffu, "topk: bad shape");
  RTPE_REQUIRE(scratch_bytes >= topk_scratch(planes, h, w, K), "topk: scratch too small");
  const int tiles = ((h + kTH - 1) / kTH) * ((w + kTW - 1) / kTW);
  u64* cand = reinterpret_cast<u64*>(scratch);
  // RTPE_TOPK_FAST=0: the 5x5 path without the separable sampling / register-blocked max passes (same bits)
  static const int fast = getenv("RTPE_TOPK_FAST") ? atoi(getenv("RTPE_TOPK_FAST")) : 1;
  const size_t grid = (size_t)((planes + 7) / 8) * 8 * tiles;                     // whole planes per XCD
  RTPE_REQUIRE(grid < 0x7fffffffu, "topk: %d planes x %d tiles do not fit one grid", planes, tiles);
  if (pad == 2)
    hipLaunchKernelGGL((topk_tile_kernel<Map, 2>), dim3((unsigned)grid), dim3(256), 0, s, m, h, w, pad, K, cand, fast, planes, tiles);
  else
    hipLaunchKernelGGL((topk_tile_kernel<Map, -1>), dim3((unsigned)grid), dim3(256), 0, s, m, h, w, pad, K, cand, 0, planes, tiles);
  RTPE_HIP_CHECK(hipGetLastError());
  size_t lds = 32 + (size_t)tiles * K * 8;
  if (lds > 150 * 1024 || tiles <= 8 * 256) lds = 32;     // the head merge (tiles <= kOwn * 256) needs no copy of the lists
  auto kern = topk_merge_kernel<Map, TagMap>;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kern, dim3(planes), dim3(256), lds, s, m, tm, tag_shared_joints, D, h, w, pad, K, tiles,
                     cand, val_k, ind_k, tag_k);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe

using namespace rtpe;

extern "C" int rtpe_bilinear_upsample(const float* src, int32_t planes, int32_t h, int32_t w, float* dst,
                                      int32_t oh, int32_t ow, void* stream) {
  RTPE_REQUIRE(src && dst && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "bilinear: bad argument");
  BilinearMap m = make_bilinear(src, h, w, 0, planes, oh, ow);   // one "image" of `planes` planes
  const size_t total = (size_t)planes * oh * ow;
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     m, planes, oh, ow, dst);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

extern "C" int rtpe_nms(const float* det, int32_t planes, int32_t h, int32_t w, int32_t ksize, int32_t pad,
                        float* out, void* stream) {
  RTPE_REQUIRE(det && out && planes > 0 && h > 0 && w > 0, "nms: bad argument");
  RTPE_REQUIRE(ksize == 2 * pad + 1 && pad >= 0 && pad <= kMaxPad, "nms: ksize=%d pad=%d unsupported", ksize, pad);
  DirectMap m{det, h, w};
  const int tiles = ((h + kTH - 1) / kTH) * ((w + kTW - 1) / kTW);
  hipLaunchKernelGGL((nms_kernel<DirectMap>), dim3(tiles, planes), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), m, h, w, pad, out);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

extern "C" int rtpe_topk_scratch_bytes(int32_t planes, int32_t h, int32_t w, int32_t K, size_t* bytes) {
  RTPE_REQUIRE(bytes && planes > 0 && h > 0 && w > 0 && K > 0, "topk_scratch_bytes: bad argument");
  *bytes = topk_scratch(planes, h, w, K);
  return RTPE_OK;
}

extern "C" int rtpe_topk(const float* det, const float* tag, int32_t planes, int32_t joints,
                         int32_t tag_per_joint, int32_t h, int32_t w, int32_t D, int32_t K, int32_t nms_ksize,
                         int32_t nms_pad, float* val_k, int32_t* ind_k, float* tag_k, void* scratch,
                         size_t scratch_bytes, void* stream) {
  RTPE_REQUIRE(det && tag && val_k && ind_k && tag_k && scratch && D > 0 && joints > 0, "topk: bad argument");
  DirectMap m{det, h, w};
  DirectTag tm{tag, h, w, D};
  return topk_run(m, tm, planes, tag_per_joint ? 0 : joints, D, h, w, K, nms_ksize, nms_pad, val_k, ind_k, tag_k,
                  scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rtpe_topk_fused(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride, const float* tg,
                               int32_t th, int32_t tw, int64_t tg_img_stride, int32_t N, int32_t J, int32_t oh,
                               int32_t ow, int32_t K, int32_t nms_ksize, int32_t nms_pad, float* val_k,
                               int32_t* ind_k, float* tag_k, void* scratch, size_t scratch_bytes, void* stream) {
  RTPE_REQUIRE(hm && tg && val_k && ind_k && tag_k && scratch && N > 0 && J > 0, "topk_fused: bad argument");
  BilinearMap m = make_bilinear(hm, hh, hw, hm_img_stride, J, oh, ow);
  BilinearTag tm{make_bilinear(tg, th, tw, tg_img_stride, J, oh, ow)};
  return topk_run(m, tm, N * J, 0, 1, oh, ow, K, nms_ksize, nms_pad, val_k, ind_k, tag_k, scratch, scratch_bytes,
                  reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rtpe_adjust_refine_scratch_bytes(int32_t P, int32_t J, int32_t D, size_t* bytes) {
  RTPE_REQUIRE(bytes && P >= 0 && J > 0 && D > 0, "adjust_refine_scratch_bytes: bad argument");
  *bytes = refine_scratch(P, J, D);
  return RTPE_OK;
}

extern "C" int rtpe_adjust_refine(const float* det, const float* tag, int32_t N, int32_t J, int32_t h, int32_t w,
                                  int32_t D, const float* ans_in, float* ans_out, const int32_t* person_img,
                                  int32_t P, int32_t do_adjust, int32_t do_refine, float* scores, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  RTPE_REQUIRE(det && tag && ((ans_in && ans_out && ans_in != ans_out) || P == 0) && N > 0 && J > 0 &&
                   J <= kMaxJ && D > 0 && D <= kMaxD,
               "adjust_refine: bad argument (J<=%d, D<=%d)", kMaxJ, kMaxD);
  if (P <= 0) return RTPE_OK;
  DirectMap m{det, h, w};
  DirectTag tm{tag, h, w, D};
  return adjust_refine_run(m, tm, N, J, h, w, D, ans_in, ans_out, person_img, P, do_adjust, do_refine, scores,
                           scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rtpe_adjust_refine_fused(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride,
                                        const float* tg, int32_t th, int32_t tw, int64_t tg_img_stride, int32_t N,
                                        int32_t J, int32_t oh, int32_t ow, const float* ans_in, float* ans_out,
                                        const int32_t* person_img, int32_t P, int32_t do_adjust,
                                        int32_t do_refine, float* scores, void* scratch, size_t scratch_bytes,
                                        void* stream) {
  RTPE_REQUIRE(hm && tg && ((ans_in && ans_out && ans_in != ans_out) || P == 0) && N > 0 && J > 0 && J <= kMaxJ,
               "adjust_refine_fused: bad argument");
  if (P <= 0) return RTPE_OK;
  BilinearMap m = make_bilinear(hm, hh, hw, hm_img_stride, J, oh, ow);
  BilinearTag tm{make_bilinear(tg, th, tw, tg_img_stride, J, oh, ow)};
  return adjust_refine_run(m, tm, N, J, oh, ow, 1, ans_in, ans_out, person_img, P, do_adjust, do_refine, scores,
                           scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rtpe_adjust_refine_fused_topk(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride,
                                             const float* tg, int32_t th, int32_t tw, int64_t tg_img_stride, int32_t N,
                                             int32_t J, int32_t oh, int32_t ow, const float* ans_in, float* ans_out,
                                             const int32_t* person_img, int32_t P, int32_t do_adjust,
                                             int32_t do_refine, float* scores, const float* topk_val,
                                             const int32_t* topk_ind, int32_t K, void* scratch, size_t scratch_bytes,
                                             void* stream) {
  RTPE_REQUIRE(hm && tg && ((ans_in && ans_out && ans_in != ans_out) || P == 0) && N > 0 && J > 0 && J <= kMaxJ &&
               topk_val && topk_ind && K > 0, "adjust_refine_fused_topk: bad argument");
  if (P <= 0) return RTPE_OK;
  BilinearMap m = make_bilinear(hm, hh, hw, hm_img_stride, J, oh, ow);
  BilinearTag tm{make_bilinear(tg, th, tw, tg_img_stride, J, oh, ow)};
  return adjust_refine_run(m, tm, N, J, oh, ow, 1, ans_in, ans_out, person_img, P, do_adjust, do_refine, scores,
                           scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream), topk_val, topk_ind, K);
}
