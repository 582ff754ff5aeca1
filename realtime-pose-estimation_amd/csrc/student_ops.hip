// The small fp32 NHWC ops of the dual-head student (config 5: AttentionStudent,
// rtpe/students.py:595-771 of the reference, built from ContextAwareModule
// :145-201 and SELayer :118-142).  All are single-pass, HBM/latency-bound;
// channel counts are multiples of 4 (16-byte rows pieces).
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// fp16 NHWC -> fp32 NHWC (the tofp32 at the end of the half-wrapped stem, fp16util.py:64-68)
__global__ void __launch_bounds__(256) cast_kernel(const _Float16* x, int in_ld, float* y, int out_ld, int C,
                                                   size_t pixels) {
  const int c4 = C >> 2;
  const size_t total = pixels * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const half4v v = *reinterpret_cast<const half4v*>(x + p * in_ld + c);
    *reinterpret_cast<float4*>(y + p * out_ld + c) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
}

// AvgPool2d(kernel_size=3, stride=2, padding=1, count_include_pad=False), students.py:657-666
__global__ void __launch_bounds__(256) avgpool_kernel(const float* x, int in_ld, float* y, int out_ld, int C, int N,
                                                      int H, int W) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    size_t p = i / c4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + iy) * W + ix) * in_ld + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        ++cnt;
      }
    }
    const float d = (float)cnt;
    *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * out_ld + c) =
        make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
  }
}

// SELayer, students.py:137-142: gate[n, c] = sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2)
// one workgroup per image; w: fc1 (hid, C) | b1 (hid) | fc2 (C, hid) | b2 (C), fp32
constexpr int kSeMaxC = 512, kSeMaxHid = 128;
__global__ void __launch_bounds__(256) se_kernel(const float* x, int in_ld, int C, int hid, int HW, const float* w,
                                                 float* gate, int gate_ld) {
  __shared__ float mean[kSeMaxC];
  __shared__ float part[4][kSeMaxC];
  __shared__ float hbuf[kSeMaxHid];
  const int n = blockIdx.x;
  const float* xi = x + (size_t)n * HW * in_ld;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // each wave sums a quarter of the pixels; lanes stride over channels
  for (int c = lane; c < C; c += 64) {
    float s = 0.f;
    for (int p = wv; p < HW; p += 4) s += xi[(size_t)p * in_ld + c];
    part[wv][c] = s;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) mean[c] = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) / (float)HW;
  __syncthreads();
  const float* w1 = w;
  const float* b1 = w1 + (size_t)hid * C;
  const float* w2 = b1 + hid;
  const float* b2 = w2 + (size_t)C * hid;
  for (int j = threadIdx.x; j < hid; j += 256) {
    float s = b1[j];
    for (int c = 0; c < C; ++c) s = __builtin_fmaf(w1[(size_t)j * C + c], mean[c], s);
    hbuf[j] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = b2[c];
    for (int j = 0; j < hid; ++j) s = __builtin_fmaf(w2[(size_t)c * hid + j], hbuf[j], s);
    gate[(size_t)n * gate_ld + c] = 1.f / (1.f + expf(-s));
  }
  // padding channels of the gate row (C not a multiple of 4): finite zeros, the combine op multiplies them in
  for (int c = C + threadIdx.x; c < gate_ld; c += 256) gate[(size_t)n * gate_ld + c] = 0.f;
}

// ContextAwareModule tail, students.py:199-200: relu(residual + hdc * gate[n, c])
__global__ void __launch_bounds__(256) cam_combine_kernel(const float* hdc, int hdc_ld, const float* res, int res_ld,
                                                          const float* gate, int gate_ld, float* y, int out_ld,
                                                          int C, int N, size_t pix_per_image) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * pix_per_image * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const int n = (int)(p / pix_per_image);
    const float4 h = *reinterpret_cast<const float4*>(hdc + p * hdc_ld + c);
    const float4 r = *reinterpret_cast<const float4*>(res + p * res_ld + c);
    const float4 g = *reinterpret_cast<const float4*>(gate + (size_t)n * gate_ld + c);
    float4 o;
    o.x = r.x + h.x * g.x; o.y = r.y + h.y * g.y; o.z = r.z + h.z * g.z; o.w = r.w + h.w * g.w;
    o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    *reinterpret_cast<float4*>(y + p * out_ld + c) = o;
  }
}

// AttentionStudent.forward, students.py:755-756: att = sigmoid(att / 20); stem_out + att (broadcast over C);
// also emits att as the first NCHW output (N,1,H,W)
__global__ void __launch_bounds__(256) sigmoid_add_kernel(const float* att, int att_ld, const float* x, int x_ld,
                                                          float* y, int out_ld, int C, size_t pixels,
                                                          float* att_out) {
  const int c4 = C >> 2;
  const size_t total = pixels * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const float a = 1.f / (1.f + expf(-(att[p * att_ld] / 20.f)));
    if (c == 0 && att_out) att_out[p] = a;
    const float4 v = *reinterpret_cast<const float4*>(x + p * x_ld + c);
    *reinterpret_cast<float4*>(y + p * out_ld + c) = make_float4(v.x + a, v.y + a, v.z + a, v.w + a);
  }
}

// AttentionStudentSteps.forward, students.py:1012-1040: att = sigmoid(att [/ att_divisor]); stem_out * att
// (broadcast over the channels); also emits att as the first NCHW output (N,1,H,W)
__global__ void __launch_bounds__(256) gate_mul_kernel(const float* att, int att_ld, const float* x, int x_ld, float* y,
                                                       int out_ld, int C, size_t pixels, float div, float* att_out) {
  const int c4 = C >> 2;
  const size_t total = pixels * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    float l = att[p * att_ld];
    if (div != 0.f) l = l / div;
    const float a = 1.f / (1.f + expf(-l));
    if (c == 0 && att_out) att_out[p] = a;
    const float4 v = *reinterpret_cast<const float4*>(x + p * x_ld + c);
    *reinterpret_cast<float4*>(y + p * out_ld + c) = make_float4(v.x * a, v.y * a, v.z * a, v.w * a);
  }
}

// the second network input: (N,3,H,W) fp32 NCHW -> NHWC rows of 4 (r, g, b, 0)
__global__ void __launch_bounds__(256) aux_pack_kernel(const float* a, float* y, int out_ld, int N, int H, int W) {
  const size_t plane = (size_t)H * W, total = (size_t)N * plane;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / plane, q = i - n * plane;
    const float* b = a + n * 3 * plane + q;
    *reinterpret_cast<float4*>(y + i * out_ld) = make_float4(b[0], b[plane], b[2 * plane], 0.f);
  }
}

// F.interpolate(mode="bilinear", align_corners=False) on NHWC rows of C (multiple of 4) channels; PyTorch-CPU's
// arithmetic (see csrc/aggregate.hip): real = scale * (o + 0.5) - 0.5 clamped at 0, T = fma(v0, l0, v1 * l1)
__device__ __forceinline__ void axis_half_pixel(float scale, int n_in, int n_out, int o, int* i0, int* i1, float* l0, float* l1) {
  if (n_in == n_out) { *i0 = *i1 = o; *l0 = 1.f; *l1 = 0.f; return; }
  float real = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);   // ATen's build contracts scale * (o + 0.5) - 0.5
  real = real < 0.f ? 0.f : real;
  int a = (int)real;
  a = a < n_in - 1 ? a : n_in - 1;
  *i0 = a;
  *i1 = a + (a < n_in - 1 ? 1 : 0);
  float l = real - (float)a;
  l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
  *l1 = l;
  *l0 = 1.f - l;
}

__global__ void __launch_bounds__(256) resize_nhwc_kernel(const float* x, int in_ld, int Hi, int Wi, float* y, int out_ld,
                                                          int Ho, int Wo, int C, int N, float sy, float sx) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    size_t p = i / c4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    axis_half_pixel(sy, Hi, Ho, oy, &y0, &y1, &ly0, &ly1);
    axis_half_pixel(sx, Wi, Wo, ox, &x0, &x1, &lx0, &lx1);
    const float* b = x + (size_t)n * Hi * Wi * in_ld + c;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * Wi + x0) * in_ld);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * Wi + x1) * in_ld);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * Wi + x0) * in_ld);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * Wi + x1) * in_ld);
    auto one = [&](float a00, float a01, float a10, float a11) {
      const float t0 = __builtin_fmaf(a00, lx0, a01 * lx1);
      const float t1 = __builtin_fmaf(a10, lx0, a11 * lx1);
      return __builtin_fmaf(t0, ly0, t1 * ly1);
    };
    *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * out_ld + c) =
        make_float4(one(v00.x, v01.x, v10.x, v11.x), one(v00.y, v01.y, v10.y, v11.y), one(v00.z, v01.z, v10.z, v11.z),
                    one(v00.w, v01.w, v10.w, v11.w));
  }
}

// skimage.color.rgb2lab / rgb2hsv on (N,3,H,W) fp32 in [0,1] (rtpe/dataloaders.py:352-356 of the reference)
__global__ void __launch_bounds__(256) rgb_to_alt_kernel(const float* src, float* dst, int N, size_t plane, int mode) {
  const size_t total = (size_t)N * plane;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / plane, q = i - n * plane;
    const float* b = src + n * 3 * plane + q;
    float* o = dst + n * 3 * plane + q;
    const float r = b[0], g = b[plane], bl = b[2 * plane];
    float o0, o1, o2;
    if (mode == 0) {
      // sRGB -> linear -> XYZ (D65) -> Lab; constants of skimage.color.colorconv
      auto lin = [](float v) { return v > 0.04045f ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f; };
      const float R = lin(r), G = lin(g), B = lin(bl);
      float X = 0.412453f * R + 0.357580f * G + 0.180423f * B;
      float Y = 0.212671f * R + 0.715160f * G + 0.072169f * B;
      float Z = 0.019334f * R + 0.119193f * G + 0.950227f * B;
      X /= 0.95047f; Z /= 1.08883f;
      auto f = [](float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; };
      const float fx = f(X), fy = f(Y), fz = f(Z);
      o0 = 116.f * fy - 16.f;
      o1 = 500.f * (fx - fy);
      o2 = 200.f * (fy - fz);
    } else {
      const float mx = fmaxf(r, fmaxf(g, bl)), mn = fminf(r, fminf(g, bl));
      const float d = mx - mn;
      float h = 0.f;
      if (d != 0.f) {
        if (mx == r) h = (g - bl) / d;
        else if (mx == g) h = 2.f + (bl - r) / d;
        else h = 4.f + (r - g) / d;
        h = h / 6.f;
        h = h - floorf(h);                                   // (h / 6) % 1
      }
      o0 = h;
      o1 = d == 0.f ? 0.f : d / mx;
      o2 = mx;
    }
    o[0] = o0; o[plane] = o1; o[2 * plane] = o2;
  }
}

static unsigned grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)(b ? b : 1);
}

int cast_launch(const _Float16* x, int in_ld, float* y, int out_ld, int C, size_t pixels, hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "cast: channel counts must be multiples of 4");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, x, in_ld, y, out_ld, C, pixels);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int avgpool_launch(const float* x, int in_ld, float* y, int out_ld, int C, int N, int H, int W, hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "avgpool: channel counts must be multiples of 4");
  const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, in_ld, y, out_ld, C, N, H, W);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int se_launch(const float* x, int in_ld, int C, int hid, int N, int HW, const float* w, float* gate, int gate_ld,
              hipStream_t s) {
  RTPE_REQUIRE(C <= kSeMaxC && hid <= kSeMaxHid && C > 0 && hid > 0, "se: C=%d hidden=%d unsupported", C, hid);
  hipLaunchKernelGGL(se_kernel, dim3(N), dim3(256), 0, s, x, in_ld, C, hid, HW, w, gate, gate_ld);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int cam_combine_launch(const float* hdc, int hdc_ld, const float* res, int res_ld, const float* gate, int gate_ld,
                       float* y, int out_ld, int C, int N, size_t pix_per_image, hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0 && gate_ld % 4 == 0, "cam_combine: C must be a multiple of 4");
  hipLaunchKernelGGL(cam_combine_kernel, dim3(grid_for((size_t)N * pix_per_image * (C / 4))), dim3(256), 0, s, hdc,
                     hdc_ld, res, res_ld, gate, gate_ld, y, out_ld, C, N, pix_per_image);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int sigmoid_add_launch(const float* att, int att_ld, const float* x, int x_ld, float* y, int out_ld, int C,
                       size_t pixels, float* att_out, hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0, "sigmoid_add: C must be a multiple of 4");
  hipLaunchKernelGGL(sigmoid_add_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, att, att_ld, x, x_ld, y,
                     out_ld, C, pixels, att_out);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int gate_mul_launch(const float* att, int att_ld, const float* x, int x_ld, float* y, int out_ld, int C, size_t pixels,
                    float div, float* att_out, hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0 && x_ld % 4 == 0 && out_ld % 4 == 0, "gate_mul: channel counts must be multiples of 4");
  hipLaunchKernelGGL(gate_mul_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, s, att, att_ld, x, x_ld, y, out_ld,
                     C, pixels, div, att_out);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int aux_pack_launch(const float* aux_nchw, float* y, int out_ld, int N, int H, int W, hipStream_t s) {
  RTPE_REQUIRE(aux_nchw != nullptr && out_ld >= 4 && out_ld % 4 == 0, "aux_pack: bad argument");
  hipLaunchKernelGGL(aux_pack_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, s, aux_nchw, y, out_ld, N, H, W);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int resize_nhwc_launch(const float* x, int in_ld, int Hi, int Wi, float* y, int out_ld, int Ho, int Wo, int C, int N,
                       hipStream_t s) {
  RTPE_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "resize: channel counts must be multiples of 4");
  hipLaunchKernelGGL(resize_nhwc_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, s, x, in_ld, Hi, Wi,
                     y, out_ld, Ho, Wo, C, N, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

}  // namespace rtpe

extern "C" int rtpe_rgb_to_alt(const float* src_nchw, int32_t N, int32_t H, int32_t W, int32_t mode, float* dst_nchw,
                               void* stream) {
  using namespace rtpe;
  RTPE_REQUIRE(src_nchw && dst_nchw && N > 0 && H > 0 && W > 0 && (mode == 0 || mode == 1), "rgb_to_alt: bad argument");
  hipLaunchKernelGGL(rgb_to_alt_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src_nchw, dst_nchw, N, (size_t)H * W, mode);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}
