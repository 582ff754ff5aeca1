// Memory-bound pieces of the forward pass: the Cin=3 stem conv (NCHW fp32 in,
// NHWC fp16 out, the tofp16 cast folded into the load) and the multi-resolution
// fuse sum.  Both are HBM-bound: 16-byte vector accesses, one pass.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float round16(float v) { return (float)(_Float16)v; }

// ---------------------------------------------------------------------------
// HighResolutionModule.forward fuse sum, pose_higher_hrnet.py:245-254:
//   y = t0; y = y + t1; ...; relu(y)   with one fp16 rounding per add.
// Terms at a lower resolution are read with nearest-neighbour upsampling
// (nn.Upsample(scale_factor=2**(j-i), mode='nearest'), :209), i.e. the
// upsampled tensor is never written to HBM.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fuse_kernel(const FuseArgs a) {
  constexpr int EPS = 16 / (int)sizeof(T);       // elements per 16-byte access
  const int c8 = a.C / EPS;
  const size_t total = (size_t)a.N * a.H * a.W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cs = (int)(i % c8);
    size_t pix = i / c8;
    const int x = (int)(pix % a.W);
    pix /= a.W;
    const int y = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    float acc[EPS];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= a.n_terms) break;
      const int u = a.term_up[t];
      const int hs = a.H >> u, ws = a.W >> u;
      const uint4 raw = *reinterpret_cast<const uint4*>(
          reinterpret_cast<const T*>(a.term[t]) + (((size_t)n * hs + (y >> u)) * ws + (x >> u)) * a.term_ld[t] + cs * EPS);
      T v[EPS];
      __builtin_memcpy(v, &raw, 16);
#pragma unroll
      for (int j = 0; j < EPS; ++j) acc[j] = t == 0 ? (float)v[j] : (float)(T)(acc[j] + (float)v[j]);
    }
    T o[EPS];
#pragma unroll
    for (int j = 0; j < EPS; ++j) o[j] = (T)((acc[j] > 0.f || !a.relu) ? acc[j] : 0.f);
    uint4 oraw;
    __builtin_memcpy(&oraw, o, 16);
    store16_wt(reinterpret_cast<T*>(a.y) + (((size_t)n * a.H + y) * a.W + x) * a.out_ld + cs * EPS, oraw);
  }
}

// The same sum with one workgroup row per output row (blockIdx.y = n * H + y) and a thread per 16-byte piece of it: 32-bit
// indices and one division by a constant per thread, where the grid-stride form above spends three 64-bit divisions per piece
// (the sums of the 160 x 160 maps ran at 0.43-0.57 of the HBM rate: they were bound by that arithmetic).  Same adds in the same
// order: same bits.
template <typename T>
__global__ void __launch_bounds__(256) fuse_rows_kernel(const FuseArgs a, const FastDiv div_c8, const FastDiv div_h) {
  constexpr int EPS = 16 / (int)sizeof(T);
  const int c8 = a.C / EPS;
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= (uint32_t)(a.W * c8)) return;
  const uint32_t x = fdiv(p, div_c8), cs = p - x * (uint32_t)c8;
  const uint32_t n = fdiv(blockIdx.y, div_h), y = blockIdx.y - n * (uint32_t)a.H;
  uint4 raw[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {                            // all loads first
    if (t >= a.n_terms) break;
    const int u = a.term_up[t];
    const uint32_t hs = (uint32_t)a.H >> u, ws = (uint32_t)a.W >> u;
    raw[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.term[t]) +
                                             ((size_t)(n * hs + (y >> u)) * ws + (x >> u)) * a.term_ld[t] + cs * EPS);
  }
  float acc[EPS];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= a.n_terms) break;
    T v[EPS];
    __builtin_memcpy(v, &raw[t], 16);
#pragma unroll
    for (int j = 0; j < EPS; ++j) acc[j] = t == 0 ? (float)v[j] : (float)(T)(acc[j] + (float)v[j]);
  }
  T o[EPS];
#pragma unroll
  for (int j = 0; j < EPS; ++j) o[j] = (T)((acc[j] > 0.f || !a.relu) ? acc[j] : 0.f);
  uint4 oraw;
  __builtin_memcpy(&oraw, o, 16);
  store16_wt(reinterpret_cast<T*>(a.y) + ((size_t)(n * (uint32_t)a.H + y) * a.W + x) * a.out_ld + cs * EPS, oraw);
}

int fuse_launch(const FuseArgs& a, hipStream_t s) {
  const int eps = a.f32 ? 4 : 8;
  RTPE_REQUIRE(a.C % eps == 0 && a.n_terms >= 1 && a.n_terms <= 4, "fuse: C=%d terms=%d", a.C, a.n_terms);
  static const int rows = env_int("RTPE_FUSE_ROWS", 1);
  const long row_pieces = (long)a.W * (a.C / eps), n_rows = (long)a.N * a.H;
  if (rows && n_rows <= 65535 && n_rows < (1l << 20) && row_pieces < (1l << 20) && (long)a.N * a.H * a.W < (1l << 31)) {
    const dim3 grid((unsigned)((row_pieces + 255) / 256), (unsigned)n_rows);
    const FastDiv dc = make_fastdiv((uint32_t)(a.C / eps)), dh = make_fastdiv((uint32_t)a.H);
    if (a.f32) hipLaunchKernelGGL(fuse_rows_kernel<float>, grid, dim3(256), 0, s, a, dc, dh);
    else hipLaunchKernelGGL(fuse_rows_kernel<_Float16>, grid, dim3(256), 0, s, a, dc, dh);
    RTPE_HIP_CHECK(hipGetLastError());
    return RTPE_OK;
  }
  const size_t total = (size_t)a.N * a.H * a.W * (a.C / eps);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (a.f32)
    hipLaunchKernelGGL(fuse_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(fuse_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// ---------------------------------------------------------------------------
// Stem conv1 + bn1 + relu, pose_higher_hrnet.py:363-365 / :638-640, with the
// wrapper's input.half() (fp16util.py:50-51) applied at load time.
// One thread = one output pixel x 64 channels; the 3 x 17 x 65 input patch of an
// 8 x 32 output tile and the 27 x 64 weights sit in LDS.
// ---------------------------------------------------------------------------
constexpr int kStemTH = 8, kStemTW = 32, kStemCO = 64;
constexpr int kStemPH = 2 * kStemTH + 1, kStemPW = 2 * kStemTW + 1;

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

template <typename T>
__global__ void __launch_bounds__(256) stem_kernel(const StemArgs a) {
  constexpr bool kHalf = sizeof(T) == 2;
  __shared__ float patch[3][kStemPH][kStemPW + 1];
  __shared__ __attribute__((aligned(16))) float wl[27][kStemCO];
  const int Ho = a.H >> 1, Wo = a.W >> 1;
  const int tiles_x = (Wo + kStemTW - 1) / kStemTW, tiles_y = (Ho + kStemTH - 1) / kStemTH;
  // Workgroups are dealt to the 8 XCDs round robin (blockIdx.x % 8): an XCD takes a CONTIGUOUS eighth of the row-major
  // tile list, so that neighbouring tiles - whose 65-float patch rows share their first / last 128-byte line and
  // whose patches share a halo row - meet in one L2 (tile t on XCD t % 8: 242 MB fetched for 157 MB of input, PMC)
  const int per_xcd = (int)(gridDim.x >> 3);
  int t = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (t >= a.N * tiles_x * tiles_y) return;                // grid padding (whole workgroup)
  const int n = t / (tiles_x * tiles_y);
  t -= n * tiles_x * tiles_y;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int oy0 = ty * kStemTH, ox0 = tx * kStemTW;
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
  const int tid = threadIdx.x;

  // Staging: ALL global loads of a thread are issued before the first LDS write.  (The plain loops compiled to one
  // load + s_waitcnt vmcnt(0) per iteration: ~20 dependent memory round trips per workgroup, which - not the
  // 864 packed FMAs per thread - set the kernel's time: 2 TB/s, 4x its VALU floor.)
  {
    float* wflat = &wl[0][0];
    uint4v wq[2];
    constexpr int kWVec = kHalf ? 27 * kStemCO / 8 : 27 * kStemCO / 4;    // 16-byte pieces of the weights: 216 / 432
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + k * 256 < kWVec) wq[k] = reinterpret_cast<const uint4v*>(a.w)[tid + k * 256];
    constexpr int kPatch = 3 * kStemPH * kStemPW, kIter = (kPatch + 255) / 256;
    auto locate = [&](int k, int* dst, bool* ok, int* src) __attribute__((always_inline)) {
      const int i = tid + k * 256;
      const int c = i < kPatch ? i / (kStemPH * kStemPW) : 0;
      const int rem = i < kPatch ? i - c * kStemPH * kStemPW : 0;
      const int py = rem / kStemPW, px = rem - py * kStemPW;
      const int iy = iy0 + py, ix = ix0 + px;
      *dst = i < kPatch ? (c * kStemPH + py) * (kStemPW + 1) + px : -1;
      *ok = i < kPatch && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      // (no branch around the load: an element outside the image reads a clamped pixel and is zeroed below)
      const int cy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      *src = ((n * 3 + c) * a.H + cy) * a.W + cx;             // < 2^31 elements (host-checked)
    };
    float v[kIter];
    if (a.x_f32) {
#pragma unroll
      for (int k = 0; k < kIter; ++k) {
        int dst, src; bool ok;
        locate(k, &dst, &ok, &src);
        v[k] = reinterpret_cast<const float*>(a.x)[src];
      }
    } else {
#pragma unroll
      for (int k = 0; k < kIter; ++k) {
        int dst, src; bool ok;
        locate(k, &dst, &ok, &src);
        v[k] = (float)reinterpret_cast<const _Float16*>(a.x)[src];
      }
    }
    if (kHalf) {                                          // 27 * 64 fp16 = 216 x 16 bytes
      if (tid < kWVec) {
        const _Float16* h = reinterpret_cast<const _Float16*>(&wq[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) wflat[tid * 8 + j] = (float)h[j];
      }
    } else {                                              // 27 * 64 fp32 = 432 x 16 bytes
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (tid + k * 256 < kWVec) reinterpret_cast<uint4v*>(wflat)[tid + k * 256] = wq[k];
    }
    float* pflat = &patch[0][0][0];
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
      int dst, src; bool ok;
      int tid_l = tid;
      asm volatile("" : "+v"(tid_l));                     // recompute the positions: 13 x 3 registers are not worth holding
      (void)tid_l;
      locate(k, &dst, &ok, &src);
      const float x = ok ? v[k] : 0.f;
      if (dst >= 0) pflat[dst] = kHalf ? round16(x) : x;    // tofp16 of the half wrapper
    }
  }
  __syncthreads();

  // thread = 8 output pixels x 8 output channels.  Lane = (pixel group g = lane / 8, channel group lane % 8); a
  // wave owns two tile rows (64 pixels), the thread's p-th pixel is number 8 p + g of them, so the 8 lanes of
  // a pixel write its 128-byte NHWC row together and one store instruction covers 8 consecutive pixels = 1 KiB
  // (a CU drains partial-line writes at ~2 B/clk: one pixel x 64 channels per thread, 16 bytes per lane at a
  // 128-byte stride, ran the kernel at 1.7 TB/s).  A weight fragment read from LDS feeds 32 FMAs.  Same
  // accumulation order per output (ky, kx, c), so the bits do not change.
  const int lane = tid & 63, wv = tid >> 6;
  const int cg = lane & 7, g = lane >> 3;
  float acc[8][8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[p][co] = 0.f;
  // (the tap loops stay loops: fully unrolled, the scheduler hoists every weight read and spills)
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float xv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int idx = p * 8 + g;
          xv[p] = patch[c][2 * (2 * wv + (idx >> 5)) + ky][2 * (idx & 31) + kx];
        }
        const float4* wr = reinterpret_cast<const float4*>(&wl[(ky * 3 + kx) * 3 + c][cg * 8]);
        const float4 w0 = wr[0], w1 = wr[1];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          acc[p][0] = __builtin_fmaf(xv[p], w0.x, acc[p][0]);
          acc[p][1] = __builtin_fmaf(xv[p], w0.y, acc[p][1]);
          acc[p][2] = __builtin_fmaf(xv[p], w0.z, acc[p][2]);
          acc[p][3] = __builtin_fmaf(xv[p], w0.w, acc[p][3]);
          acc[p][4] = __builtin_fmaf(xv[p], w1.x, acc[p][4]);
          acc[p][5] = __builtin_fmaf(xv[p], w1.y, acc[p][5]);
          acc[p][6] = __builtin_fmaf(xv[p], w1.z, acc[p][6]);
          acc[p][7] = __builtin_fmaf(xv[p], w1.w, acc[p][7]);
        }
      }
  constexpr int EPS = 16 / (int)sizeof(T);
  float al[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { al[j] = a.alpha[cg * 8 + j]; be[j] = a.beta[cg * 8 + j]; }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = p * 8 + g;
    const int oy = oy0 + 2 * wv + (idx >> 5), ox = ox0 + (idx & 31);
    if (oy >= Ho || ox >= Wo) continue;
    T* dst = reinterpret_cast<T*>(a.y) + (((size_t)n * Ho + oy) * Wo + ox) * a.out_ld + cg * 8;
#pragma unroll
    for (int q = 0; q < 8 / EPS; ++q) {
      T o[EPS];
#pragma unroll
      for (int j = 0; j < EPS; ++j) {
        const int co = q * EPS + j;
        float v = kHalf ? round16(acc[p][co]) : acc[p][co];                          // conv output
        v = __builtin_fmaf(v, al[co], be[co]);                                       // BN output
        if (kHalf) v = round16(v);
        o[j] = (T)(v > 0.f ? v : 0.f);
      }
      uint4v raw;
      __builtin_memcpy(&raw, o, 16);
      store16_wt(dst + q * EPS, raw);
    }
  }
}

int stem_launch(const StemArgs& a, hipStream_t s) {
  RTPE_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0 && a.out_ld % 8 == 0, "stem: H=%d W=%d", a.H, a.W);
  RTPE_REQUIRE((size_t)a.N * 3 * a.H * a.W < 0x7fffffffull, "stem: input of %d x 3 x %d x %d elements", a.N, a.H, a.W);
  const int Ho = a.H / 2, Wo = a.W / 2;
  const int tiles = ((Wo + kStemTW - 1) / kStemTW) * ((Ho + kStemTH - 1) / kStemTH);
  const unsigned grid = (unsigned)(((size_t)tiles * a.N + 7) / 8 * 8);            // an eighth of the tiles per XCD
  if (a.f32)
    hipLaunchKernelGGL(stem_kernel<float>, dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(stem_kernel<_Float16>, dim3(grid), dim3(256), 0, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe
