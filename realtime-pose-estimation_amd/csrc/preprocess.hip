// Pre-processing on the GPU: the step in front of the hot path.
// Replaces, for one uint8 HWC image, transforms.py:181-192 `resize_align_multi_scale`
// (cv2.warpAffine with the 2x3 matrix of get_affine_transform :59-93, rot = 0) followed by
// torchvision ToTensor + Normalize (validate_hhrnet.py:63-67, teacher_inference.py:70-73):
// uint8 HWC in, normalised NCHW fp32 out, one pass, no intermediate image.
//
// Sampling convention (cv2 is not available to pin its INTER_LINEAR fixed-point scheme, so this
// is the documented convention of THIS implementation): destination pixel (x, y) samples the
// source at  M_inv * (x, y, 1)  (pixel centres at integer coordinates, as warpAffine),
// bilinear weights in fp32 (cv2 quantises them to 1/32), out-of-image taps contribute 0
// (BORDER_CONSTANT).  With round_u8 the interpolated value is rounded to a grey level
// (floor(v + 0.5), clamped to [0, 255]) before ToTensor, because the reference's warp returns a
// uint8 image (transforms.py:185-190); ToTensor and Normalize are the true divisions torchvision
// performs (x / 255, then (x - mean) / std).  HBM-bound: 3 B in, 12 B out per pixel.
#include "rtpe_common.h"

namespace rtpe {

struct WarpArgs {
  const unsigned char* src;   // (h, w, 3) uint8, row stride `stride` bytes
  float* dst;                 // (3, oh, ow) fp32
  int h, w, stride, oh, ow;
  float m[6];                 // dst -> src:  sx = m0*x + m1*y + m2,  sy = m3*x + m4*y + m5
  float mean[3], stdev[3];
  int round_u8;
};

__global__ void __launch_bounds__(256) warp_normalize_kernel(const WarpArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.ow || y >= a.oh) return;
  const float sx = __builtin_fmaf(a.m[0], (float)x, __builtin_fmaf(a.m[1], (float)y, a.m[2]));
  const float sy = __builtin_fmaf(a.m[3], (float)x, __builtin_fmaf(a.m[4], (float)y, a.m[5]));
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float lx = sx - fx, ly = sy - fy;
  float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = x0 + dx, yy = y0 + dy;
      const float wgt = (dx ? lx : 1.f - lx) * (dy ? ly : 1.f - ly);
      if ((unsigned)xx < (unsigned)a.w && (unsigned)yy < (unsigned)a.h) {
        const unsigned char* p = a.src + (size_t)yy * a.stride + xx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = __builtin_fmaf(wgt, (float)p[c], v[c]);
      }
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {   // [uint8 image] -> ToTensor (/255) -> Normalize ((t - mean) / std)
    float g = v[c];
    if (a.round_u8) g = fminf(fmaxf(floorf(g + 0.5f), 0.f), 255.f);
    a.dst[((size_t)c * a.oh + y) * a.ow + x] = (g / 255.f - a.mean[c]) / a.stdev[c];
  }
}

}  // namespace rtpe

extern "C" int rtpe_warp_normalize(const void* src_hwc_u8, int32_t h, int32_t w, int32_t stride_bytes,
                                   const float* m_dst_to_src, const float* mean, const float* stdev, void* dst_chw_f32,
                                   int32_t oh, int32_t ow, int32_t round_u8, void* stream) {
  using namespace rtpe;
  RTPE_REQUIRE(src_hwc_u8 && dst_chw_f32 && m_dst_to_src && mean && stdev, "warp_normalize: null argument");
  RTPE_REQUIRE(h > 0 && w > 0 && oh > 0 && ow > 0 && stride_bytes >= 3 * w, "warp_normalize: h=%d w=%d stride=%d oh=%d ow=%d",
               h, w, stride_bytes, oh, ow);
  WarpArgs a;
  a.src = reinterpret_cast<const unsigned char*>(src_hwc_u8);
  a.dst = reinterpret_cast<float*>(dst_chw_f32);
  a.h = h; a.w = w; a.stride = stride_bytes; a.oh = oh; a.ow = ow;
  a.round_u8 = round_u8 != 0;
  for (int i = 0; i < 6; ++i) a.m[i] = m_dst_to_src[i];
  for (int c = 0; c < 3; ++c) {
    RTPE_REQUIRE(stdev[c] > 0.f, "warp_normalize: std must be positive");
    a.mean[c] = mean[c];
    a.stdev[c] = stdev[c];
  }
  hipLaunchKernelGGL(warp_normalize_kernel, dim3((ow + 63) / 64, (oh + 3) / 4), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}
