// Device-side helpers of the streaming conv kernel (conv_stream.hip) and the kernels built on its structure.
#pragma once
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
typedef int int4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCC = 48;          // channels per staged chunk
constexpr int kSlots = 6;        // 16-byte slots per staged pixel
constexpr int kPStride = 96;     // LDS bytes per staged pixel
constexpr int kKC = 14;          // k-steps per stage (9 taps x 48 channels, padded to 448)
constexpr int kKH = 7;           // k-steps per half stage
constexpr int kLoaders = 3;

__device__ __forceinline__ float round16s(float v) { return (float)(_Float16)v; }

#define RTPE_SBARRIER()                         \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction only takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define RTPE_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
  switch (n) {
    RTPE_W(1) RTPE_W(2) RTPE_W(3) RTPE_W(4) RTPE_W(5) RTPE_W(6) RTPE_W(7) RTPE_W(8) RTPE_W(9) RTPE_W(10)
    RTPE_W(11) RTPE_W(12) RTPE_W(13) RTPE_W(14) RTPE_W(15) RTPE_W(16) RTPE_W(17) RTPE_W(18) RTPE_W(19) RTPE_W(20)
    RTPE_W(21) RTPE_W(22) RTPE_W(23) RTPE_W(24) RTPE_W(25) RTPE_W(26) RTPE_W(27) RTPE_W(28) RTPE_W(29) RTPE_W(30)
    RTPE_W(31) RTPE_W(32) RTPE_W(33) RTPE_W(34) RTPE_W(35) RTPE_W(36) RTPE_W(37) RTPE_W(38) RTPE_W(39) RTPE_W(40)
    RTPE_W(41) RTPE_W(42) RTPE_W(43) RTPE_W(44) RTPE_W(45) RTPE_W(46) RTPE_W(47) RTPE_W(48) RTPE_W(49) RTPE_W(50)
    RTPE_W(51) RTPE_W(52) RTPE_W(53) RTPE_W(54) RTPE_W(55) RTPE_W(56) RTPE_W(57) RTPE_W(58) RTPE_W(59) RTPE_W(60)
    RTPE_W(61) RTPE_W(62) RTPE_W(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // 0, or more than the counter holds
  }
#undef RTPE_W
}

#ifdef RTPE_CONV_STAMPS
#define SSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SSTAMP(v)
#endif

// XCD x (= blockIdx % 8) owns tiles x, x+8, ...; its G workgroups walk that XCD's
// (tile, cout block) sequence with stride G (G % n_cb == 0: a workgroup keeps its
// cout block, and the cout blocks of a tile share its halo through that XCD's L2).
struct Units {
  int n_tiles, n_cb, G, xcd, j, count;
  uint32_t cb_mul;                                       // seq / n_cb = umulhi(seq, cb_mul) (seq * n_cb < 2^32)
  __device__ __forceinline__ void init(int n_tiles_, int n_cb_) {
    n_tiles = n_tiles_; n_cb = n_cb_;
    cb_mul = n_cb_ <= 1 ? 0u : (uint32_t)((0x100000000ull + (unsigned)n_cb_ - 1) / (unsigned)n_cb_);
    G = (int)(gridDim.x >> 3);
    xcd = (int)(blockIdx.x & 7);
    j = (int)(blockIdx.x >> 3);
    const int tiles_xcd = (n_tiles - xcd + 7) >> 3;
    const int n_seq = tiles_xcd * n_cb;
    count = j < n_seq ? (n_seq - j + G - 1) / G : 0;
  }
  __device__ __forceinline__ void get(int i, int* tile, int* cb) const {
    const int seq = j + i * G;
    const int tq = n_cb <= 1 ? seq : (int)__umulhi((uint32_t)seq, cb_mul);
    *cb = seq - tq * n_cb;
    *tile = xcd + 8 * tq;
  }
};

}  // namespace
}  // namespace rtpe

namespace rtpe {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int sgpr4 __attribute__((ext_vector_type(4)));
namespace {
// a raw buffer descriptor over [p, p + bytes), built from wave-uniform values (four SGPRs for the asm forms below)
__device__ __forceinline__ sgpr4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a64 = reinterpret_cast<uint64_t>(p);
  return sgpr4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(a64 >> 32)),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
}  // namespace
}  // namespace rtpe
