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

}  // namespace rtpe
