// Multi-scale / flip test aggregation on the GPU: the tensor arithmetic of the upstream
// HigherHRNet `core/inference.py` (`get_multi_stage_outputs`, `aggregate_results`), which the reference's
// legacy/valid_ae1dim.py:166-207 calls (the module itself is not part of the reference repository).
// Every step of those functions is one of
//     dst = [dst +] resize( flip_w( src[:, channel_map] ) )  [/ div]
// with resize = F.interpolate(mode="bilinear", align_corners=False) (or the identity when the sizes agree) and
// flip_w = torch.flip(., [3]) applied AFTER the resize, as upstream does.  One kernel does exactly that, with
// PyTorch-CPU's arithmetic so that results are bit-equal to the torch ops (compiled with -ffp-contract=off):
//   * source index  real = fma(scale, o + 0.5, -0.5) (ATen's CPU build contracts the expression; found by
//     experiment against F.interpolate), clamped at 0, scale = float(in) / float(out);
//     i0 = floor(real), i1 = i0 + (i0 < in - 1), l1 = real - i0, l0 = 1 - l1   (area_pixel_compute_source_index);
//   * value  T = fma(v0, lx0, v1 * lx1) per row, out = fma(T0, ly0, T1 * ly1)   (the order the decode kernels
//     pin for align_corners=True, csrc/decode.hip);
//   * dst + v and the TRUE division by div as separate fp32 operations (x / 3.0 is not x * (1/3)).
// HBM-bound: one read of the (smaller) source planes through L2, one read-modify-write of dst.
#include "rtpe_common.h"

namespace rtpe {

constexpr int kMaxMap = 64;

struct ResizeArgs {
  const float* src;     // (N, C_src, h, w)
  float* dst;           // (N, C_dst, oh, ow)
  int N, C_src, C_dst, h, w, oh, ow;
  int cmap[kMaxMap];    // source channel of every destination channel
  float sy, sx;         // float(in) / float(out)
  int flip, accumulate;
  float div;            // 1: no division
};

__device__ __forceinline__ void axis_nc(float scale, int n_in, int n_out, int o, int* i0, int* i1, float* l0, float* l1) {
  if (n_in == n_out) { *i0 = *i1 = o; *l0 = 1.f; *l1 = 0.f; return; }
  float real = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);   // ATen's build contracts scale * (o + 0.5) - 0.5
  real = real < 0.f ? 0.f : real;
  int a = (int)real;
  a = a < n_in - 1 ? a : n_in - 1;
  *i0 = a;
  *i1 = a + (a < n_in - 1 ? 1 : 0);
  float l = real - (float)a;
  l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);             // guard_index_and_lambda
  *l1 = l;
  *l0 = 1.f - l;
}

__global__ void __launch_bounds__(256) resize_combine_kernel(const ResizeArgs a) {
  const size_t plane = (size_t)a.oh * a.ow;
  const size_t total = (size_t)a.N * a.C_dst * plane;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % a.ow);
    const size_t r = i / a.ow;
    const int y = (int)(r % a.oh);
    const size_t pc = r / a.oh;
    const int c = (int)(pc % a.C_dst), n = (int)(pc / a.C_dst);
    const float* b = a.src + ((size_t)n * a.C_src + a.cmap[c]) * a.h * a.w;
    const int xs = a.flip ? a.ow - 1 - x : x;          // torch.flip of the RESIZED map
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    axis_nc(a.sy, a.h, a.oh, y, &y0, &y1, &ly0, &ly1);
    axis_nc(a.sx, a.w, a.ow, xs, &x0, &x1, &lx0, &lx1);
    float v;
    if (a.h == a.oh && a.w == a.ow) {
      v = b[(size_t)y0 * a.w + x0];
    } else {
      const float v00 = b[(size_t)y0 * a.w + x0], v01 = b[(size_t)y0 * a.w + x1];
      const float v10 = b[(size_t)y1 * a.w + x0], v11 = b[(size_t)y1 * a.w + x1];
      const float t0 = __builtin_fmaf(v00, lx0, v01 * lx1);
      const float t1 = __builtin_fmaf(v10, lx0, v11 * lx1);
      v = __builtin_fmaf(t0, ly0, t1 * ly1);
    }
    if (a.accumulate) v = a.dst[i] + v;
    if (a.div != 1.f) v = v / a.div;
    a.dst[i] = v;
  }
}

}  // namespace rtpe

using namespace rtpe;

extern "C" int rtpe_resize_combine(const float* src, int32_t N, int32_t C_src, int32_t h, int32_t w,
                                   const int32_t* channel_map, int32_t C_dst, int32_t flip_w, float* dst,
                                   int32_t oh, int32_t ow, int32_t accumulate, float div, void* stream) {
  RTPE_REQUIRE(src && dst && N > 0 && C_src > 0 && C_dst > 0 && h > 0 && w > 0 && oh > 0 && ow > 0,
               "resize_combine: bad argument");
  RTPE_REQUIRE(C_dst <= kMaxMap, "resize_combine: at most %d destination channels", kMaxMap);
  RTPE_REQUIRE(div != 0.f, "resize_combine: division by zero");
  ResizeArgs a;
  memset(&a, 0, sizeof(a));
  a.src = src; a.dst = dst;
  a.N = N; a.C_src = C_src; a.C_dst = C_dst; a.h = h; a.w = w; a.oh = oh; a.ow = ow;
  for (int c = 0; c < C_dst; ++c) {
    a.cmap[c] = channel_map ? channel_map[c] : c;
    RTPE_REQUIRE(a.cmap[c] >= 0 && a.cmap[c] < C_src, "resize_combine: channel map entry %d out of range", c);
  }
  a.sy = (float)h / (float)oh;
  a.sx = (float)w / (float)ow;
  a.flip = flip_w != 0;
  a.accumulate = accumulate != 0;
  a.div = div;
  const size_t total = (size_t)N * C_dst * oh * ow;
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(resize_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}
