// Internal helpers shared by the HIP translation units of librtpe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "rtpe_hip.h"

namespace rtpe {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define RTPE_HIP_CHECK(expr)                                               \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) return ::rtpe::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define RTPE_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::rtpe::set_error(__VA_ARGS__);           \
      return RTPE_E_INVALID;                    \
    }                                           \
  } while (0)

// Tuning switches (results are bit-identical whatever they are set to: RTPE_CONV_STREAM, RTPE_FUSE_BLOCKS,
// RTPE_PLANE_MAJOR, RTPE_CONV_LDS_CAP) are read in every build.  DIAGNOSTIC switches - profiling ablations
// that skip work (wrong results by construction) and forced launch shapes - exist only in a build made with
// -DRTPE_DIAG (RTPE_BUILD_DEFS=-DRTPE_DIAG): an inherited environment variable cannot corrupt the product.
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#ifdef RTPE_DIAG
#define RTPE_DIAG_ENV_INT(name, dflt) ::rtpe::env_int(name, dflt)
#else
#define RTPE_DIAG_ENV_INT(name, dflt) (dflt)
#endif

// run-time tuning options (rtpe_set_option): every setting gives bit-identical results
enum { kOptBlockRing = 0, kOptBlockPC = 1, kOptDirect1x1 = 2, kOptLanes = 3, kOptTileDma = 4, kOptPair1x1 = 5, kOptFusedStem = 6, kOptConv64 = 7, kOptHeadDirect = 8, kOptDeconv48 = 9, kOptConv48s2 = 10, kNumOptions = 11 };
int get_option(int key);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: true the first time a kernel's
// launcher runs on the CURRENT device (one mask per kernel template instance)
inline bool first_use_on_device(unsigned long long* mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
  const unsigned long long bit = 1ull << dev;
  const unsigned long long old = __atomic_fetch_or(mask, bit, __ATOMIC_RELAXED);
  return !(old & bit);
}

// makes `device` current for the lifetime of the object and restores the caller's device
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// exact unsigned division by a runtime constant: q = umulhi(n, mul) for
// n * d < 2^32 (all uses here: n < 2^20, d < 2^12)
struct FastDiv {
  uint32_t d, mul;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.mul = d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) {
  return f.d <= 1 ? n : __umulhi(n, f.mul);
}

// 16-byte store of an activation row piece, WRITE-THROUGH (sc0 sc1: system scope): the bytes leave the L2 while the kernel
// runs instead of sitting there dirty until the release at the kernel's end writes them back - every kernel here writes a
// tensor the NEXT kernel reads from memory anyway (the L2s are per XCD and invalidated at kernel boundaries).  Measured on
// the streaming conv kernel: 96 -> 96 47.1 -> 46.0 / 44.0 -> 42.6 us, 384 -> 384 42.6 -> 41.1 (diagnostic switch, one box).
// s_nop 1: a store of more than 8 bytes reads its data registers for two more wait states (the compiler pads its own
// stores, not an asm statement).  RTPE_WT_STORES=0 at build time: plain stores.
#ifndef RTPE_WT_STORES
#define RTPE_WT_STORES 1
#endif
typedef int rtpe_i32x4 __attribute__((ext_vector_type(4)));
template <class V>
__device__ __forceinline__ void store16_wt(void* p, const V& v) {
  static_assert(sizeof(V) == 16, "store16_wt: 16-byte values");
#if RTPE_WT_STORES
  const rtpe_i32x4 d = __builtin_bit_cast(rtpe_i32x4, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
#else
  *reinterpret_cast<V*>(p) = v;
#endif
}

// ---- conv launch description (conv_mfma.hip) ------------------------------
struct ConvArgs {
  const _Float16* x;     // input view base (channel offset folded in)
  const _Float16* w;     // packed weights for this layer (device)
  const float* alpha;    // [cout_pad]
  const float* beta;     // [cout_pad]
  const _Float16* res;   // residual view base or nullptr
  _Float16* y;           // NHWC output view base or nullptr
  void* y_nchw;          // NCHW output (fp32 or fp16) or nullptr
  int N, H_in, W_in, in_ld;
  int H_pos, W_pos;      // grid of output positions computed by this launch
  int H_full, W_full;    // full output spatial size (addressing)
  int o_mul, oy_add, ox_add;  // output pixel = pos * o_mul + add
  int in_mul;            // input base = pos * in_mul
  int out_ld, res_ld;
  // element offset of 48-channel chunk c from a view's base: c * cs.  NHWC: cs = channels of a chunk, ld = channels
  // of the tensor; plane-major [C/48][N][H][W][48] (streaming kernel only): cs = N*H*W*48, ld = 48.  0 = NHWC default
  long long in_cs, out_cs, res_cs;
  int cin, cout;         // logical
  int cout_store;        // channels written to NHWC (multiple of 4 groups masked)
  int nchw_channels;     // channel count of the NCHW output
  int nchw_f32;          // 1: fp32 NCHW, 0: fp16
  int tapw, ntaps;       // taps form a tapw x tapw grid: (dy,dx) = (lo_y+ty*dil, lo_x+tx*dil)
  int dil;               // dilation
  int lo_y, lo_x;        // min tap offsets
  int halo_h, halo_w;    // staged input tile (pixels)
  int rowb;              // LDS bytes per staged tile row (>= halo_w * pstride: conv_row_pitch)
  int cc;                // input channels staged per chunk
  int n_cchunks;         // ceil(cin / cc)
  int kc;                // 32-wide k chunks per channel chunk = ceil(ntaps*cc/32)
  int pstride;           // LDS bytes per staged pixel
  int th, tw;            // output tile (positions); th*tw == 16*NT*WAVES
  int tiles_x, tiles_y;
  int relu, round_conv;
  FastDiv div_tw, div_slots, div_rowslots, div_cc, div_tiles_x, div_tiles_xy;
  FastDiv div_rowb16, div_ps16;   // rowb / 16, pstride / 16 (16-byte slots of the LDS tile image)
  size_t x_bytes;        // bytes from x to the end of its tensor (buffer bounds of the LDS-DMA path)
  size_t img_bytes;      // one-workgroup-per-tile kernel: bytes of one image of the input view (its buffer bounds)
  int dma_stage;         // one-workgroup-per-tile kernel: 1 = the halo tiles are staged by LDS-DMA (option "tile_dma")
  int n_cb;              // cout blocks
  // several sub-launches in ONE grid (one-workgroup-per-tile kernel): the 4 sub-pixel classes of the transposed
  // conv.  Class c uses w_c[c], lo_yc/lo_xc[c], oy_c/ox_c[c]; everything else is common.  0 = a plain launch
  int n_cls;
  const _Float16* w_c[4];
  int lo_yc[4], lo_xc[4], oy_c[4], ox_c[4];
  int buf_bytes;         // streaming kernel: bytes of one LDS tile buffer
  int n_bufs;            // streaming kernel: halo tile buffers (2 or 3)
  int n_wslots;          // streaming kernel: weight half-stage slots in LDS (3: ring, 2*n_cchunks: resident)
  int deint;             // one-workgroup-per-tile kernel, stride 2: the staged tile keeps the even input columns of a row
                         // first, then the odd ones (n_even = number of even columns): consecutive OUTPUT pixels of a tap are
                         // then consecutive LDS pixels, as with stride 1, and the B-operand reads are bank-conflict free
  int n_even;
  int wsplit;            // one-workgroup-per-tile kernel: 2 = the shares of a packed cout block go to the two halves of a workgroup's
                         // waves (ConvTile::mrun < 0) instead of to two workgroups; 0 / 1 = every wave multiplies all of the workgroup's tiles
  int m_split;           // one-workgroup-per-tile kernel: workgroups that share one packed cout block (ConvTile::mrun cout
                         // tiles each); 0 / 1 = one workgroup computes the whole block
  int ablate;            // profiling ablations (RTPE_STREAM_ABL): 1 skip MFMA k-loops, 2 skip residual loads + output stores, 4 skip halo DMA
  unsigned long long* dbg;  // diagnostic builds only (-DRTPE_CONV_STAMPS): per-segment cycle sums
};

struct ConvPlan {       // weight-layout half of the plan (fixed at create time)
  int mt;               // cout tiles (x16) per wave = per workgroup
  int cc, kc, n_cchunks, pstride;
  int tapw, in_mul, lo_y, lo_x;
  int esize, dil;       // bytes per element (2 fp16 / 4 fp32), dilation
  int cout_pad, n_cb;   // cout rounded up to 16*mt; number of cout blocks
  int cin, cout;        // the layer's logical channel counts (kernels for ONE layer shape test these, not the padded ones)
  size_t packed_bytes;  // bytes of packed weights (all cout blocks)
};

struct ConvTile {       // launch-shape half of the plan (depends on N, H, W)
  int nt, waves;        // pixel tiles (x16) per wave, MFMA waves per workgroup
  int th, tw;
  size_t lds_bytes;
  int kind;             // 0: one workgroup per tile (conv_mfma.hip), 2: streaming, weights and halos by LDS-DMA
                        // (conv_stream.hip), 4: direct 1x1
                        // (conv_direct.hip), 5: 64 -> 64 3x3 on persistent workgroups (conv64.hip)
  int grid;             // streaming: number of workgroups
  int buf_bytes;        // streaming: one LDS tile buffer
  int n_bufs;           // streaming: halo tile buffers
  int n_wslots;         // streaming: weight half-stage slots
  int mrun;             // kind 0: cout tiles (x16) per workgroup, a divisor of ConvPlan::mt (0 = all mt of them): the
                        // packed cout block is shared out to mt / mrun workgroups (small grids)
                        // < 0: -mrun cout tiles per HALF of the workgroup's waves (two halves on one staged tile of
                        // th * tw = 16 * nt * waves / 2 pixels): the block is shared out to mt / (2 * -mrun) workgroups
};

struct ConvGeom {       // logical layer, independent of the batch
  int cin, cout, ksize, stride;
  int deconv_class;     // -1: plain conv; 0..3: k4 s2 p1 transposed-conv parity class (a*2+b)
  int esize;            // 2 (fp16, default when 0) or 4 (fp32)
  int dil;              // dilation (default 1 when 0); padding = dil * (ksize / 2)
};

ConvPlan conv_make_plan(const ConvGeom& g);
// choose the tile for a position grid (N images of H_pos x W_pos positions)
// (allow_direct = false: never the direct 1x1 kernel, whose launch needs a < 2 GiB input view and a plain NHWC output)
// (allow_conv64 = false: never the persistent 64 -> 64 kernel, which takes no residual and writes plain NHWC rows)
ConvTile conv_make_tile(const ConvPlan& p, int N, int H_pos, int W_pos, bool allow_direct = true, bool allow_conv64 = true);
// LDS bytes per row of the staged input tile of a tw-wide output tile: halo_w * pstride, padded (fp16) so that a
// 16-pixel MFMA column tile which wraps to the next output row keeps the bank pattern of consecutive pixels
int conv_row_pitch(const ConvPlan& p, int tw, int kind);
bool conv_deint(const ConvPlan& p, int kind);
void conv_enum_tiles(const ConvPlan& p, int N, int H_pos, int W_pos, std::vector<ConvTile>* out);
// pack fp16 weights (host) into fragment order; w is OIHW (IOHW 4x4 for deconv classes)
void conv_pack_weights(const ConvGeom& g, const ConvPlan& p, const void* w, void* packed);
// fill geometry / divisors of `a` (pointers, sizes and flags are the caller's)
void conv_fill_args(const ConvGeom& g, const ConvPlan& p, const ConvTile& t, ConvArgs* a);
int conv_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s);
bool conv_stream_supports(const ConvPlan& p);
size_t conv_stream_lds(const ConvPlan& p, int buf_bytes, int n_bufs, int n_wslots);
int conv_stream_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s);
// 1x1 convs without a staged tile (conv_direct.hip, ConvTile::kind == 4): B fragments straight from global memory,
// weights in registers; conv_direct_mb = cout tiles per wave, 0 when the layer is not one the kernel takes
// conv 1x1 64 -> 256 + residual + ReLU and the conv 1x1 256 -> 64 + ReLU that reads its output, as one kernel (conv_pair.hip)
bool conv_pair_supports(int cin1, int cout1, int cout2);
int conv_pair_launch(const ConvPlan& p1, const ConvArgs& c1, const ConvPlan& p2, const ConvArgs& c2, hipStream_t s);
int conv_direct_mb(const ConvPlan& p);
// 3x3 stride-1 64 -> 64 convs without residual on persistent workgroups with double-buffered LDS-DMA halo tiles and
// register-resident weights (conv64.hip, ConvTile::kind == 5: 16 x 16 tiles, 8 waves)
bool conv64_supports(const ConvPlan& p);
size_t conv64_lds();
int conv64_grid(int N, int H_pos, int W_pos);
int conv64_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s);
int conv_direct_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s);
// the two heads (1x1, 48 input channels, fp32 NCHW output, head 0 also NHWC) on the direct scheme (conv_direct.hip)
bool conv_head_supports(const ConvPlan& p, const ConvArgs& a);
int conv_head_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s);
// deconv48.hip: the four sub-pixel classes of the k4 s2 transposed conv to 48 channels on one persistent kernel
bool deconv48_supports(const ConvPlan& p, const ConvArgs& merged);
int deconv48_launch(const ConvPlan& p, const ConvArgs& merged, hipStream_t s);
// conv48s2.hip: 3x3 stride-2 convs from 48 input channels on persistent workgroups with register-resident weights
bool conv48s2_supports(const ConvPlan& p, const ConvArgs& a);
int conv48s2_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s);
// several such layers that read the same input view (48 / 96 output channels each, 2 or 4 groups of 48 together) as one launch
int conv48s2_launch_group(const ConvPlan* const* plans, const ConvArgs* layers, int n, hipStream_t s);
// fused BasicBlock of the 48-channel branches (conv_block.hip): y = relu(bn2(conv(relu(bn1(conv(x))))) + x)
bool conv_block_supports(int cin, int cout, int H, int W);
int conv_block_launch(const _Float16* x, int in_ld, size_t x_bytes, _Float16* y, int out_ld, const _Float16* w1,
                      const float* ab1, const _Float16* w2, const float* ab2, int N, int H, int W, hipStream_t s);

// ---- elementwise / stem (elementwise.hip) ---------------------------------
struct FuseArgs {
  const _Float16* term[4];
  int term_ld[4], term_up[4];
  int n_terms;
  _Float16* y;
  int out_ld, C;
  int N, H, W;  // output spatial size
  int f32;      // 1: fp32 tensors (pointers are reinterpreted), 0: fp16
  int relu;     // final ReLU (HighResolutionModule) or plain sum (student heads)
};
int fuse_launch(const FuseArgs& a, hipStream_t s);

struct StemArgs {
  const void* x;  // NCHW (N,3,H,W), fp32 or fp16
  int x_f32;
  const _Float16* w;  // [27][64] fp16, k = (ky*3+kx)*3 + c
  const float* alpha;
  const float* beta;
  _Float16* y;  // NHWC (N,H/2,W/2,64)
  int N, H, W, out_ld;
  int f32;      // 1: fp32 weights / output, no intermediate rounding
};
int stem_launch(const StemArgs& a, hipStream_t s);

// the stem of a half-precision program on the matrix cores (stem_fused.hip): conv1 + bn1 + relu alone (y1 set), or with
// conv2 + bn2 + relu behind it as one kernel (y1 null; y, w2, alpha2, beta2 set)
struct StemFusedArgs {
  const void* x;          // NCHW (N,3,H,W), fp32 or fp16
  int x_f32;
  const _Float16* w1;     // [27][64] fp16, k = (ky*3+kx)*3 + c (the stem op's weights)
  const float* alpha1;
  const float* beta1;
  const _Float16* w2;     // the conv op's packed fragments (plan: mt 4, one 64-channel chunk, 18 k steps, one cout block)
  const float* alpha2;
  const float* beta2;
  _Float16* y;            // fused: NHWC (N,H/4,W/4,out_ld >= 64)
  _Float16* y1;           // conv1 only: NHWC (N,H/2,W/2,out_ld >= 64)
  int N, H, W, out_ld;
  int relu, round_conv;   // of the conv op
  int ablate;             // profiling: 1 skip conv1's FMAs, 2 skip conv2's k loop, 4 skip the patch prefetch, 8 skip the stores
};
bool stem_fused_supports(int H, int W);
int stem_fused_launch(const StemFusedArgs& a, hipStream_t s);

// ---- student ops (student_ops.hip), fp32 NHWC ------------------------------
int cast_launch(const _Float16* x, int in_ld, float* y, int out_ld, int C, size_t pixels, hipStream_t s);
int avgpool_launch(const float* x, int in_ld, float* y, int out_ld, int C, int N, int H, int W, hipStream_t s);
int se_launch(const float* x, int in_ld, int C, int hid, int N, int HW, const float* w, float* gate, int gate_ld,
              hipStream_t s);
int cam_combine_launch(const float* hdc, int hdc_ld, const float* res, int res_ld, const float* gate, int gate_ld,
                       float* y, int out_ld, int C, int N, size_t pix_per_image, hipStream_t s);
int sigmoid_add_launch(const float* att, int att_ld, const float* x, int x_ld, float* y, int out_ld, int C,
                       size_t pixels, float* att_out, hipStream_t s);
// y = x * sigmoid(att / div) (div == 0: no division), channels [0, C) of a row
int gate_mul_launch(const float* att, int att_ld, const float* x, int x_ld, float* y, int out_ld, int C,
                    size_t pixels, float div, float* att_out, hipStream_t s);
int aux_pack_launch(const float* aux_nchw, float* y, int out_ld, int N, int H, int W, hipStream_t s);
int resize_nhwc_launch(const float* x, int in_ld, int Hi, int Wi, float* y, int out_ld, int Ho, int Wo, int C, int N,
                       hipStream_t s);

}  // namespace rtpe
