// Streaming implicit-GEMM 3x3 convolution for the 48-channel-chunked layers of the
// w48 network (Cin in {48, 96, 192, 384}, stride 1: every BasicBlock conv, i.e. ~85 %
// of the forward FLOPs; pose_higher_hrnet.py:46-75 of the reference; and the stride-2
// convs of the fuse layers with Cin = 48, :213-230, whose (2 th + 1) x (2 tw + 1) halo
// tiles still fit twice beside the resident weights).  Same math, same
// k order, same epilogue rounding points and the same packed-weight format as
// conv_mfma.hip, so results are bit-identical; what changes is how operands reach the
// MFMAs.  Measured on the one-workgroup-per-tile kernel: every layer class sat at
// 450-500 TFLOP/s whatever its shape, because (i) each wave fetched its weight
// fragments from L1/L2 (4 waves x 3 KiB per k-step = the whole 64 B/clk of a CU's
// vector cache at MFMA rate), (ii) staging and k-loop of a workgroup were serial.
//
//   * ONE persistent workgroup per CU: WAVES MFMA waves + 3 loader waves, walking an
//     XCD-aware list of (tile, cout block) units;
//   * loader waves move bytes with LDS-DMA (`buffer_load ... lds`, 1 KiB per
//     instruction, no VGPRs):  one streams the WEIGHT fragments of the next half
//     stage (7 k-steps, MT x 7 KiB) into a 3-slot LDS ring, so all MFMA waves share
//     one copy through LDS instead of 4-5 copies through L1; two alternate on the
//     input HALO tiles (up to 3 LDS buffers: tile s+2 is requested while s is being
//     multiplied).  Halo rows are issued row by row: the per-lane column offsets are
//     computed once per tile, the row base is a scalar, so a DMA instruction costs no
//     VALU work.  Out-of-image pixels use an out-of-range buffer offset: the bounds
//     check returns zeros (= the conv padding);
//   * each loader owns its own vmcnt stream, so every wait is a constant
//     `s_waitcnt vmcnt(N)`; hand-over is by workgroup barriers (M: stage start,
//     H: mid-stage, E: before the epilogue reuses the tile buffer);
//   * MFMA waves: k-loop fully unrolled per half stage, A (weights) and B (pixels)
//     fragments both read from LDS one k-step ahead of the MFMAs that use them.
//   * input, output and residual may each be NHWC or plane-major ([C/48][N][H][W][48]: ConvArgs::in_cs /
//     out_cs / res_cs = element offset of a 48-channel chunk); the engine keeps the inner tensors of a
//     BasicBlock chain plane-major so that a chunk row is contiguous in memory;
//   * the epilogue transposes through LDS, adds the residual rows (requested one unit ahead into the fixed
//     register window v[224:255]) and stores whole 16-byte row pieces from a scalar base + lane offset.
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

// The residual row pieces of the next unit are in flight while the k-loop runs.  They live in a fixed
// register window v[224:255] that the compiler never allocates (amdgpu_num_vgpr below): the loads and
// the adds name the registers in their asm text.  An asm OUTPUT operand would be an ordinary value to
// the register allocator, which is free to copy it (live-range splitting, loop-carried values) right
// after the load was issued, i.e. before the data has arrived.
constexpr int kResReg0 = 224;
#define RTPE_RES_LOAD(K, R0, R1, R2, R3)                                                             \
  if (it == K)                                                                                       \
    asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, %1" ::"v"(off), "s"(rb)               \
                 : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3);
#define RTPE_RES_ADD(K, R0, R1, R2, R3)                                                              \
  if (it == K)                                                                                       \
    asm volatile("v_pk_add_f16 %0, %0, v" #R0 "\n\tv_pk_add_f16 %1, %1, v" #R1 "\n\tv_pk_add_f16 %2, %2, v" #R2 \
                 "\n\tv_pk_add_f16 %3, %3, v" #R3                                                   \
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#define RTPE_RES_ALL(X)                                                                              \
  X(0, 224, 225, 226, 227) X(1, 228, 229, 230, 231) X(2, 232, 233, 234, 235) X(3, 236, 237, 238, 239) \
  X(4, 240, 241, 242, 243) X(5, 244, 245, 246, 247) X(6, 248, 249, 250, 251) X(7, 252, 253, 254, 255)

template <int MT, int NT, int WAVES>
__global__ void __launch_bounds__((WAVES + kLoaders) * 64) __attribute__((amdgpu_num_vgpr(kResReg0)))
conv_stream_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WSLOT = MT * kKH * 1024;               // weight fragments of one half stage
  char* const wring = smem;
  const int NWS = a.n_wslots;                           // weight half-stage slots in LDS
  char* const tiles = smem + NWS * WSLOT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  Units um;
  um.init(a.N * a.tiles_x * a.tiles_y, a.n_cb);
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  const int ncc = a.n_cchunks;                          // power of two (host-checked)
  const int sh = __builtin_ctz((unsigned)ncc);
  const int S = um.count << sh;                         // stages of this workgroup
  const int NB = a.n_bufs;                              // halo tile buffers (2 or 3)
  // a workgroup keeps its cout block: when all 2*ncc weight halves of it fit, they are loaded once
  const bool resident = NWS == 2 * ncc;
  if (S == 0) return;

  if (wv == WAVES) {
    // ----------------------------- weight loader -----------------------------
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(a.w), 0, a.n_cb * ncc * kKC * MT * 1024, 0x00020000);
    const int voff = lane * 16;
    auto issue = [&](int q) {                            // half stage q = 2 s + h
      const int s = q >> 1, h = q & 1;
      int tile, cb;
      um.get(s >> sh, &tile, &cb);
      const int cci = s & (ncc - 1);
      const int src = ((cb * ncc + cci) * kKC + h * kKH) * MT * 1024;
      char* dst = wring + (resident ? q : q % 3) * WSLOT;
#pragma unroll
      for (int p = 0; p < MT * kKH; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + p * 1024), 16, voff, src + p * 1024, 0, 0);
    };
#ifdef RTPE_CONV_STAMPS
    unsigned long long w0, w1, w2, w3, w4, wwait = 0, wissue = 0;
#endif
    if (resident) {
      for (int q = 0; q < 2 * ncc; ++q) issue(q);        // stages 0..ncc-1 of unit 0 = every (cci, half)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int s = 0; s < S; ++s) {
        RTPE_SBARRIER();                                 // M(s)
        if ((s & (ncc - 1)) == ncc - 1) RTPE_SBARRIER(); // E(s)
      }
      return;
    }
    const int Q = 2 * S;
    issue(0);
    issue(1);
    for (int s = 0; s < S; ++s) {
      SSTAMP(w0);
      // everything but the most recent half (MT*7 instructions) has landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * kKH) : "memory");
      SSTAMP(w1);
      RTPE_SBARRIER();                                   // M(s)
      SSTAMP(w2);
      const bool more0 = 2 * s + 2 < Q;
      if (more0) issue(2 * s + 2);
      SSTAMP(w3);
      if (more0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * kKH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SSTAMP(w4);
      RTPE_SBARRIER();                                   // H(s)
      if (2 * s + 3 < Q) issue(2 * s + 3);
      if ((s & (ncc - 1)) == ncc - 1) RTPE_SBARRIER();   // E(s)
#ifdef RTPE_CONV_STAMPS
      wwait += (w1 - w0) + (w4 - w3); wissue += w3 - w2;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RTPE_CONV_STAMPS
    if (a.dbg != nullptr && lane == 0) { atomicAdd(&a.dbg[6], wwait); atomicAdd(&a.dbg[7], wissue); atomicAdd(&a.dbg[11], (unsigned long long)S); }
#endif
    return;
  }

  if (wv > WAVES) {
    // ------------------------------ tile loaders ------------------------------
    const int jl = wv - WAVES - 1;                       // 0 or 1
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const int rowslots = a.halo_w * kSlots;
    const int rowbytes = a.rowb;
    // A tile request in two halves: the address arithmetic of stage s + P is done BEFORE this loader waits for
    // its rows of the current tile and for the barrier (it would otherwise sit between the barrier that frees
    // the buffer and the first request, on the path that bounds the stage: a request needs ~2 us to land and
    // can only be made one stage ahead with two buffers); after the barrier nothing but the requests is left.
    struct TileReq {
      uint32_t voff[4];                                  // per-lane column part, one per DMA instruction of a row
      int soff0, soff_row;                               // scalar offset of halo row 0, bytes between rows
      int row_lo, row_hi;                                // halo rows [row_lo, row_hi) lie inside the image
      char* buf;
    };
    auto prepare = [&](int s, TileReq& q) {
      int tile, cb;
      um.get(s >> sh, &tile, &cb);
      const int chunk = s & (ncc - 1);
      const int cbase = chunk * kCC;
      const int chunk_off = (int)(chunk * a.in_cs);       // elements: 48 per chunk (NHWC) or one plane
      uint32_t t = (uint32_t)tile;
      const uint32_t n = fdiv(t, a.div_tiles_xy);
      t -= n * tiles_xy;
      const uint32_t tyi = fdiv(t, a.div_tiles_x);
      const uint32_t txi = t - tyi * a.tiles_x;
      const int iy0 = (int)tyi * a.th * a.in_mul + a.lo_y, ix0 = (int)txi * a.tw * a.in_mul + a.lo_x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int qq = k * 64 + lane;
        const int hx = qq / kSlots, sl = qq - hx * kSlots;
        const int ix = ix0 + hx;
        const bool ok = (unsigned)ix < (unsigned)a.W_in && cbase + sl * 8 < a.cin;
        q.voff[k] = ok ? (uint32_t)(ix * a.in_ld + sl * 8) * 2u : 0x80000000u;
      }
      q.buf = tiles + (s % NB) * a.buf_bytes;
      q.soff_row = a.W_in * a.in_ld * 2;
      q.soff0 = (((int)n * a.H_in + iy0) * a.W_in * a.in_ld + chunk_off) * 2;
      q.row_lo = iy0 < 0 ? -iy0 : 0;
      q.row_hi = a.H_in - iy0;                            // rows r with iy0 + r < H_in
    };
    auto fire = [&](const TileReq& q, int r0, int r1) {  // halo rows [r0, r1)
      for (int r = r0; r < r1; ++r) {
        const bool row_ok = r >= q.row_lo && r < q.row_hi;
        const int soff = row_ok ? q.soff0 + r * q.soff_row : 0;
        char* dst = q.buf + r * rowbytes;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k * 64 < rowslots && k * 64 + lane < rowslots && !(a.ablate & 4))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + k * 1024), 16,
                                                     (int)(row_ok ? q.voff[k] : 0x80000000u), soff, 0, 0);
        }
      }
    };
    auto issue = [&](int s, int r0, int r1) {
      TileReq q;
      prepare(s, q);
      fire(q, r0, r1);
    };
    // both loaders work on EVERY tile (one half of its rows each): two instruction streams feed the
    // memory pipeline, and a tile is requested P = NB - 1 stages before it is multiplied
    const int hh = a.halo_h;
    const int P = NB - 1;
    const int ra = jl == 0 ? 0 : hh / 2, rb = jl == 0 ? hh / 2 : hh;     // this loader's rows
    const int rmid = resident ? rb : (ra + rb + 1) / 2;   // with a mid-stage barrier the issue is split around it
    const int krow = (rowslots + 63) >> 6;
    const int n_part = (rb - ra) * krow;                 // DMA instructions of one part
    for (int t = 0; t < P && t < S; ++t) issue(t, ra, rb);
#ifdef RTPE_CONV_STAMPS
    unsigned long long t0, t1, t2, t3, twait = 0, tissue = 0;
#endif
    for (int s = 0; s < S; ++s) {
      SSTAMP(t0);
      const bool more = s + P < S;
      TileReq q;
      prepare(more ? s + P : S - 1, q);                  // (no memory operations: before the waits)
      // this loader's part of tile s has landed; with P == 2 its part of tile s+1 may still be in flight
      if (P == 2 && s + 1 < S) wait_vmcnt(n_part);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SSTAMP(t1);
      RTPE_SBARRIER();                                   // M(s): buffer (s-1) % NB is free
      SSTAMP(t2);
      if (more) fire(q, ra, rmid);
      SSTAMP(t3);
      if (!resident) {
        RTPE_SBARRIER();                                 // H(s)
        if (more) fire(q, rmid, rb);
      }
      if ((s & (ncc - 1)) == ncc - 1) RTPE_SBARRIER();   // E(s)
#ifdef RTPE_CONV_STAMPS
      twait += t1 - t0; tissue += t3 - t2;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RTPE_CONV_STAMPS
    if (a.dbg != nullptr && lane == 0) { atomicAdd(&a.dbg[8], twait); atomicAdd(&a.dbg[9], tissue); }
#endif
    return;
  }

  // -------------------------------- MFMA waves --------------------------------
  const int r = lane & 15;
  const int g = lane >> 4;
  // LDS byte offset of this lane group's 8 channels in k-step k (flat [tap][channel] order)
  int toff[kKC];
#pragma unroll
  for (int k = 0; k < kKC; ++k) {
    int kk = k * 32 + g * 8;
    if (kk >= 9 * kCC) kk -= 9 * kCC;                   // zero-weight k padding: any finite in-tile data
    const int tap = kk / kCC, c = kk - tap * kCC;
    const int ty = tap / 3, tx = tap - ty * 3;
    toff[k] = ty * a.rowb + tx * kPStride + c * 2;
  }
  int pixbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const uint32_t p = (wv * NT + nt) * 16 + r;
    const uint32_t oy = fdiv(p, a.div_tw);
    const uint32_t ox = p - oy * a.tw;
    pixbase[nt] = (int)(oy * a.in_mul * a.rowb + ox * a.in_mul * kPStride);      // in_mul = conv stride
  }
  constexpr int ROWB = MT * 32 + 16;
  constexpr int CH = MT * 2;
  constexpr int NIT = (NT * 16 * CH + 63) / 64;          // 16-byte row pieces per lane in the epilogue
  // row piece `it` of this lane, fixed for the whole kernel: its 16 bytes in the wave's transposed slab
  // (eoff) and its position (epos = row << 16 | column * o_mul << 8 | 16-byte slot * 16; a piece that
  // does not exist or lies in the channel padding gets a row no tile reaches).  Per unit only a scalar
  // base address, two compares and two 24-bit multiply-adds per piece are left; stores and residual
  // loads use the scalar-base + 32-bit-lane-offset form
  int eoff[NIT], epos[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    const int pw = c / CH, slot = c - pw * CH;
    const uint32_t p = wv * NT * 16 + pw;
    const uint32_t oyt = fdiv(p, a.div_tw);
    const uint32_t oxt = p - oyt * a.tw;
    eoff[it] = c < NT * 16 * CH ? pw * ROWB + slot * 16 : 0;
    epos[it] = (int)((oyt << 16) | ((oxt * a.o_mul) << 8) | (slot * 16));
  }
  const uint32_t row_pix = (uint32_t)(a.W_full * a.o_mul) & 0xffffffu;   // pixels between two tile rows
  auto piece_off = [&](int e, uint32_t ld2) __attribute__((always_inline)) {
    const uint32_t pix = __umul24((uint32_t)e >> 16, row_pix) + (((uint32_t)e >> 8) & 255u);
    return __umul24(pix, ld2) + ((uint32_t)e & 255u);
  };

  float4v acc[MT][NT];
  int bsel = 0;                                          // s % NB
  int wsel = 0;                                          // (2 s) % 3
  const int n_units = um.count;

  // BN / bias parameters: a workgroup keeps its cout block (grid / 8 is a multiple of n_cb)
  float4v al[MT], be[MT];
  int cb0;
  {
    int tile0;
    um.get(0, &tile0, &cb0);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int c4 = (cb0 * MT + m) * 16 + g * 4;
      al[m] = *reinterpret_cast<const float4v*>(a.alpha + c4);
      be[m] = *reinterpret_cast<const float4v*>(a.beta + c4);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)                           // read them here: the compiler's wait for these
    asm volatile("" ::"v"(al[m]), "v"(be[m]));           // loads belongs in front of the unit loop
  const int cblk = cb0 * MT * 16;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    const int slot = c % CH;
    if (c >= NT * 16 * CH || cblk + slot * 8 >= a.cout_store) epos[it] |= 0x7fff0000;
  }
  const bool use_res = a.res != nullptr && !(a.ablate & 2);
#ifdef RTPE_CONV_STAMPS
  unsigned long long st[16] = {0};
  const unsigned long long k_begin = __builtin_readcyclecounter();
#endif

  // one half stage: 7 k-steps, operands of step k+1 are requested before the MFMAs of step k
  auto half_stage = [&](const char* wslot, const char* tilebuf, auto hsel) {
    constexpr int H = decltype(hsel)::value;
    const char* wl = wslot + lane * 16;
    half8 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const half8*>(tilebuf + pixbase[nt] + toff[H * kKH]);
#pragma unroll
    for (int kk = 0; kk < kKH; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < kKH) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 1) * MT + m) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[nxt][nt] = *reinterpret_cast<const half8*>(tilebuf + pixbase[nt] + toff[H * kKH + kk + 1]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc[m][nt], 0, 0, 0);
      // the LDS reads of step k+1 (and their address adds) are spread between the MFMAs of step k:
      // the matrix pipe never waits for a burst of reads to be issued
      if (kk + 1 < kKH) {
#pragma unroll
        for (int i = 0; i < (MT + NT + 3) / 4; ++i) {                              // four reads per MFMA, early in the step
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                       // VALU (address)
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                       // DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  struct UnitPos { uint32_t n; int py0, px0; };
  auto unit_pos = [&](int u) {
    int tile, cb;
    um.get(u, &tile, &cb);
    uint32_t t = (uint32_t)tile;
    UnitPos q;
    q.n = fdiv(t, a.div_tiles_xy);
    t -= q.n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    const uint32_t txi = t - tyi * a.tiles_x;
    q.py0 = (int)tyi * a.th;
    q.px0 = (int)txi * a.tw;
    return q;
  };
  // residual rows of unit u: requested ONE unit ahead (right after the previous unit's rows were
  // added, before its stores), so an HBM round trip and the write acknowledgements queued in front
  // of them (vmcnt completes in order) have a whole k-loop to finish.  The loads are issued from
  // inline asm: the compiler's own s_waitcnt insertion answers a register set that is filled one unit
  // before it is read with vmcnt(0) after every store; here the one wait that is needed is written by
  // hand (run_unit).  One register set and ONE copy of the unit code: the unrolled unit is ~35 KiB of
  // instructions, two alternating copies did not fit the 64 KiB instruction cache two CUs share.
  // Every lane loads (lanes without a valid row piece read the unit's first pixel and never store).
  static_assert(NIT <= 8, "residual register window");
  auto load_res = [&](int u) {
    const UnitPos q = unit_pos(u);
    const int hy = a.H_pos - q.py0, hx = (a.W_pos - q.px0) * a.o_mul;
    const size_t pix0 = ((size_t)q.n * a.H_full + q.py0 * a.o_mul + a.oy_add) * a.W_full + q.px0 * a.o_mul + a.ox_add;
    const _Float16* rb = a.res + pix0 * a.res_ld + (size_t)cb0 * a.res_cs;
    const uint32_t ld2 = (uint32_t)a.res_ld * 2u;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int e = epos[it];
      asm volatile("" : "+v"(e));                        // lane-only math must not be hoisted out of the unit loop
      const bool ok = (e >> 16) < hy && ((e >> 8) & 255) < hx;
      const uint32_t off = ok ? piece_off(e, ld2) : 0u;   // the unit's first pixel is always inside the tensor
      RTPE_RES_ALL(RTPE_RES_LOAD)
    }
  };

  auto run_unit = [&](int u) {
#ifdef RTPE_CONV_STAMPS
    unsigned long long m0, m1, m2, m3, m4;
#endif
    const UnitPos q = unit_pos(u);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    char* tilebuf = nullptr;
    for (int cci = 0; cci < ncc; ++cci) {
      tilebuf = tiles + bsel * a.buf_bytes;
      const int w0 = resident ? 2 * cci : wsel;
      const int w1 = resident ? 2 * cci + 1 : (wsel == 2 ? 0 : wsel + 1);
      SSTAMP(m0);
      RTPE_SBARRIER();                                   // M(s): tile s and weight half 2s are in LDS
      SSTAMP(m1);
      if (!(a.ablate & 1)) half_stage(wring + w0 * WSLOT, tilebuf, std::integral_constant<int, 0>());
      SSTAMP(m2);
      if (!resident) RTPE_SBARRIER();                    // H(s): weight half 2s+1 is in LDS
      SSTAMP(m3);
      if (!(a.ablate & 1)) half_stage(wring + w1 * WSLOT, tilebuf, std::integral_constant<int, 1>());
      SSTAMP(m4);
#ifdef RTPE_CONV_STAMPS
      st[0] += m1 - m0; st[1] += m2 - m1; st[2] += m3 - m2; st[3] += m4 - m3; st[5] += 1;
#endif
      if (cci + 1 < ncc) {
        bsel = bsel + 1 == NB ? 0 : bsel + 1;
        wsel = wsel == 0 ? 2 : wsel - 1;                 // (2 (s+1)) % 3 = (wsel + 2) % 3
      }
    }
    // ---- epilogue: BN/bias (+ residual) (+ ReLU), transposed through LDS ----
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                     // as above: keep the epilogue's lane math in the loop
    const int re = lane_e & 15, ge = lane_e >> 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SSTAMP(m0);
    RTPE_SBARRIER();                                     // E(s): every MFMA wave is done with the tile
    SSTAMP(m1);
    char* obuf = tilebuf + wv * (NT * 16 * ROWB);
    // BN / bias with the wrapper's rounding points, two channels per VALU op where the ISA allows it
    auto bn_to_lds = [&](auto rc, auto nchw) {
      constexpr bool RC = decltype(rc)::value, NCHW = decltype(nchw)::value;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4v v = acc[m][nt];
          half4 o;
          if (RC) {                                      // the conv output is an fp16 tensor
            o = bn_round(v, al[m], be[m]);
          } else {
            float2v lo{v[0], v[1]}, hi{v[2], v[3]};
            lo = __builtin_elementwise_fma(lo, float2v{al[m][0], al[m][1]}, float2v{be[m][0], be[m][1]});
            hi = __builtin_elementwise_fma(hi, float2v{al[m][2], al[m][3]}, float2v{be[m][2], be[m][3]});
            const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
            o = half4{olo[0], olo[1], ohi[0], ohi[1]};
          }
          *reinterpret_cast<half4*>(obuf + (nt * 16 + re) * ROWB + m * 32 + ge * 8) = o;
          if (NCHW) {                                    // heads: NCHW straight from the registers
            const uint32_t p = (wv * NT + nt) * 16 + re;
            const uint32_t oyt = fdiv(p, a.div_tw);
            const uint32_t oxt = p - oyt * a.tw;
            const int py = q.py0 + (int)oyt, px = q.px0 + (int)oxt;
            if (py < a.H_pos && px < a.W_pos) {
              const int oy = py * a.o_mul + a.oy_add, ox = px * a.o_mul + a.ox_add;
              const int c4 = (cb0 * MT + m) * 16 + ge * 4;
#pragma unroll
              for (int jx = 0; jx < 4; ++jx) {
                const int c = c4 + jx;
                if (c < a.nchw_channels) {
                  const float xr = (float)o[jx];
                  const float x = a.relu ? (xr > 0.f ? xr : 0.f) : xr;
                  const size_t oi = (((size_t)q.n * a.nchw_channels + c) * a.H_full + oy) * a.W_full + ox;
                  if (a.nchw_f32)
                    reinterpret_cast<float*>(a.y_nchw)[oi] = x;
                  else
                    reinterpret_cast<_Float16*>(a.y_nchw)[oi] = (_Float16)x;
                }
              }
            }
          }
        }
      }
    };
    // the executed variant is one straight run of instructions (instruction cache, see load_res)
    if (a.y_nchw != nullptr) {
      if (a.round_conv) bn_to_lds(std::true_type(), std::true_type()); else bn_to_lds(std::false_type(), std::true_type());
    } else {
      if (a.round_conv) bn_to_lds(std::true_type(), std::false_type()); else bn_to_lds(std::false_type(), std::false_type());
    }
    SSTAMP(m2);
    if (a.y != nullptr && !(a.ablate & 2)) {
      // the residual rows of THIS unit were requested a k-loop ago; nothing newer than the stores
      // that followed them is outstanding
      if (use_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SSTAMP(m3);
      // all row pieces are read from LDS first (no control flow between the reads), then
      // finished and stored: one LDS latency per unit instead of one per piece
      half8 ov[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) ov[it] = *reinterpret_cast<const half8*>(obuf + eoff[it]);
      const int hy = a.H_pos - q.py0, hx = (a.W_pos - q.px0) * a.o_mul;
      const size_t pix0 = ((size_t)q.n * a.H_full + q.py0 * a.o_mul + a.oy_add) * a.W_full + q.px0 * a.o_mul + a.ox_add;
      char* const yb = reinterpret_cast<char*>(a.y + pix0 * a.out_ld + (size_t)cb0 * a.out_cs);
      const uint32_t ld2 = (uint32_t)a.out_ld * 2u;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (use_res) {                                   // fp16 add, round-to-nearest-even = the wrapper's add
          int4v w = __builtin_bit_cast(int4v, ov[it]);
          RTPE_RES_ALL(RTPE_RES_ADD)
          ov[it] = __builtin_bit_cast(half8, w);
        }
        if (a.relu) {                                    // x > 0 ? x : +0, on the sign bits
          short8 b = __builtin_bit_cast(short8, ov[it]);
          b = b & ~(b >> 15);
          ov[it] = __builtin_bit_cast(half8, b);
        }
      }
      if (use_res && u + 1 < n_units) load_res(u + 1);   // the adds are done: the window is free
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        int e = epos[it];
        asm volatile("" : "+v"(e));
        if ((e >> 16) < hy && ((e >> 8) & 255) < hx) store16_wt(yb + piece_off(e, ld2), ov[it]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's LDS traffic on the buffer is over
    SSTAMP(m4);
#ifdef RTPE_CONV_STAMPS
    { unsigned long long m5; SSTAMP(m5);
      st[12] += m1 - m0; st[13] += m2 - m1; st[4] += m3 - m2; st[14] += m4 - m3; st[10] += m5 - m4; }
#endif
    bsel = bsel + 1 == NB ? 0 : bsel + 1;
    wsel = wsel == 0 ? 2 : wsel - 1;
  };

  // diagnostic (RTPE_STREAM_ABL >> 8): every other workgroup of an XCD starts late, so that the
  // store bursts of the CUs do not coincide
  if ((a.ablate >> 8) && ((blockIdx.x >> 3) & 1))
    for (int i = 0; i < (a.ablate >> 8); ++i) __builtin_amdgcn_s_sleep(16);
  if (use_res) load_res(0);
  for (int u = 0; u < n_units; ++u) run_unit(u);
#ifdef RTPE_CONV_STAMPS
  if (a.dbg != nullptr && lane == 0) {
    st[15] = __builtin_readcyclecounter() - k_begin;
    const int slots[] = {0, 1, 2, 3, 4, 5, 10, 12, 13, 14, 15};
    for (int i : slots) atomicAdd(&a.dbg[i], st[i]);
  }
#endif
}

// =============================================================================================================
// Streaming kernel, second generation ("v2", ConvTile::kind == 3).  Same math, same k order, same rounding points
// and the same packed weights as the kernel above: bit-identical results.  What changed is who waits for what.
// Counters of the kernel above on the C >= 96 layers (profiles/r02_pmc_summary.json): matrix pipe busy 31 %, half
// of all wave cycles parked (SQ_WAIT_ANY 0.48-0.57); without any MFMA and without stores a launch still takes
// 22-25 us of its 41 (profiles/r02_stream_ablation.txt).  Two serial chains set the stage time, not the work:
//   (1) with two halo buffers the LDS-DMA request for tile s+1 can only be made at the barrier that frees the
//       buffer of tile s-1, i.e. ONE stage before the tile is multiplied, and a request needs 2-2.5 us to land
//       under load against 1.6 us of MFMAs per stage (spreading the requests over more loader waves changes
//       nothing, measured: it is a latency chain, not an issue rate);
//   (2) the four MFMA waves run a unit's epilogue (BN, transposition, residual, ReLU, stores: ~3 us of a
//       8.5-12.5 us unit) in lock step, during which the matrix pipe idles and no tile can land in that buffer.
// Here:
//   * tiles are staged THROUGH THE LOADER WAVES' REGISTERS (their ~200 free VGPRs x 64 lanes are 50 KiB of buffer
//     per wave that the LDS does not have): plain buffer loads of tile s+2 are issued as soon as tile s+1 has been
//     written to LDS, so every tile has two stages to arrive while only two LDS buffers exist; the write into
//     the freed buffer is a burst of ds_write_b128 (~0.3 us) with no memory latency behind the barrier;
//   * the MFMA waves' epilogue ends when BN + ReLU-less fp16 rows are in the slab (the freed tile buffer); after
//     one more barrier they start the next unit.  The loaders take the slab from there: residual rows (loaded one
//     unit ahead into their registers), fp16 add, ReLU and the 16-byte row stores run beside the next k loops;
//   * one loader streams the weight ring by LDS-DMA as before (resident weights: all four loaders stage tiles).
// OUTCOME (round 3, in-kernel stamps in profiles/r03_stream2_stamps.txt): bit-identical to the first kernel and NOT
// faster (96 -> 96 at 80 x 80, batch 32: 42.5-44.9 us against 41.5-45.8 us).  The loaders keep up (tile loader 1,600
// cycles per 4,400-cycle stage, drain loader 3,600 per unit), but three of the four MFMA waves then run their k loops
// near the matrix rate and the one whose SIMD partner is the busiest loader takes twice as long - every instruction
// a partner wave issues is taken out of that SIMD's issue port, which an MFMA wave with 15 MFMAs per 8 LDS reads
// already fills.  The layer class is bound by instruction issue per SIMD, whoever issues: the lever is fewer
// instructions per FLOP (v_mfma_f32_32x32x16_f16: a quarter of the matrix instructions' issue share and fewer operand
// reads per FLOP), not another division of labour.  Kept as an opt-in family of launch shapes (option "stream_v2",
// default 0) for configurations with resident weights or stride 2; never selected by default.
// 4 MFMA waves + 4 loader waves = two waves per SIMD.  Tiles may be smaller than 16 * NT * 4 pixels: the surplus
// 16-pixel MFMA tiles recompute the last pixel and are never stored (20 x 20 maps as 7 x 4 = 28 tiles for 25:
// no SIMD carries two MFMA waves, as the 5-wave shape of the first kernel does).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int sgpr4 __attribute__((ext_vector_type(4)));
namespace {
// a raw buffer descriptor over [p, p + bytes), built from wave-uniform values (four SGPRs for the asm forms below)
__device__ __forceinline__ sgpr4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a64 = reinterpret_cast<uint64_t>(p);
  return sgpr4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(a64 >> 32)),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
// 16 bytes per lane from / to descriptor + per-lane byte offset + scalar byte offset.  The destination is a tied
// operand: the variable keeps its register, the data arrives later (the caller waits: s_waitcnt vmcnt).
// The *_first forms start with s_nop 4: the descriptor / scalar offset may have been restored from a spill lane by
// v_readlane_b32 (a VALU write of an SGPR) just in front of the statement; a vector-memory instruction must not read such
// an SGPR for 5 wait states and the compiler's hazard recognizer does not look into asm text.  The statements that
// follow in a run use the same scalar operands, which are in SGPRs by then.
__device__ __forceinline__ void buf_load16_first(u32x4& dst, uint32_t voff, sgpr4 srd, int soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void buf_load16(u32x4& dst, uint32_t voff, sgpr4 srd, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// s_nop 1 behind a store: a store of more than 8 bytes reads its data registers for two more wait states; the compiler
// pads that for its own stores, not behind an asm statement, and it is free to overwrite `v` with the very next VALU
// instruction (seen: the first dword of a 16-byte piece replaced by the next piece's)
__device__ __forceinline__ void buf_store16_first(u32x4 v, uint32_t voff, sgpr4 srd, int soff) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void buf_store16(u32x4 v, uint32_t voff, sgpr4 srd, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// tile loaders of the second-generation kernel: the fewest whose share of a halo tile (rows x pieces per row) fits the
// kMaxTP register pieces of one wave
__host__ __device__ inline int stream2_tile_loaders(int halo_h, int krow) {
  int n = 1;
  while (n < 4 && ((halo_h + n - 1) / n) * krow > 24) ++n;
  return n;
}
constexpr int kLoaders2 = 4;
constexpr int kMaxTP = 24;         // 1-KiB pieces (64 lanes x 16 B) of one halo tile per loader
constexpr int kSyncBytes = 64;     // LDS behind the buffers: the loaders' drain counter

}  // namespace

template <int MT, int NT>
__global__ void __launch_bounds__(512) conv_stream2_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WAVES = 4;
  constexpr int WSLOT = MT * kKH * 1024;               // weight fragments of one half stage
  constexpr int ROWB = MT * 32 + 16;                   // slab bytes per pixel row
  constexpr int CH = MT * 2;                           // 16-byte pieces per pixel row
  constexpr int PW = NT * 16 * CH;                     // pieces of one MFMA wave's slab
  constexpr int NPIECES = WAVES * PW;
  char* const wring = smem;
  const int NWS = a.n_wslots;
  char* const tiles = smem + NWS * WSLOT;
  int* const sync_ctr = reinterpret_cast<int*>(tiles + 2 * a.buf_bytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  Units um;
  um.init(a.N * a.tiles_x * a.tiles_y, a.n_cb);
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  const int ncc = a.n_cchunks;                          // power of two (host-checked)
  const int sh = __builtin_ctz((unsigned)ncc);
  const int S = um.count << sh;                         // stages of this workgroup
  const bool resident = NWS == 2 * ncc;
  if (S == 0) return;
  if (tid == 0) *sync_ctr = 0;
  __syncthreads();
  const int n_px = a.th * a.tw;                         // pixels of a tile (<= 16 * NT * WAVES)

  struct UnitPos { uint32_t n; int py0, px0; };
  auto unit_pos = [&](int u) {
    int tile, cb;
    um.get(u, &tile, &cb);
    uint32_t t = (uint32_t)tile;
    UnitPos q;
    q.n = fdiv(t, a.div_tiles_xy);
    t -= q.n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    const uint32_t txi = t - tyi * a.tiles_x;
    q.py0 = (int)tyi * a.th;
    q.px0 = (int)txi * a.tw;
    return q;
  };
  int cb0;
  {
    int tile0;
    um.get(0, &tile0, &cb0);                            // a workgroup keeps its cout block (grid / 8 is a multiple of n_cb)
  }

  if (wv >= WAVES) {
    const int li = wv - WAVES;
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(a.w), 0, a.n_cb * ncc * kKC * MT * 1024, 0x00020000);
    const int wvoff = lane * 16;
    if (li == 0 && !resident) {
      // ------------------- weight loader: 3-slot ring of half stages, LDS-DMA (as in the first kernel) -------------------
      auto issue = [&](int q) {
        const int s = q >> 1, h = q & 1;
        int tile, cb;
        um.get(s >> sh, &tile, &cb);
        const int cci = s & (ncc - 1);
        const int src = ((cb * ncc + cci) * kKC + h * kKH) * MT * 1024;
        char* dst = wring + (q % 3) * WSLOT;
#pragma unroll
        for (int p = 0; p < MT * kKH; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + p * 1024), 16, wvoff, src + p * 1024, 0, 0);
      };
      const int Q = 2 * S;
      issue(0);
      issue(1);
      for (int s = 0; s < S; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * kKH) : "memory");   // everything but the most recent half
        RTPE_SBARRIER();                                 // M(s)
        const bool more0 = 2 * s + 2 < Q;
        if (more0) issue(2 * s + 2);
        if (more0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * kKH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RTPE_SBARRIER();                                 // H(s)
        if (2 * s + 3 < Q) issue(2 * s + 3);
        if ((s & (ncc - 1)) == ncc - 1) { RTPE_SBARRIER(); RTPE_SBARRIER(); }   // E(s), S(s)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    // ------------- tile loaders (halo tiles through registers) and drain loaders (slab -> + residual, ReLU -> memory) -------------
    // Rules that shape this code.  (a) Every vector-memory operation is issued from inline asm and every wait is
    // written by hand (`issued` counts this wave's operations, a mark is the count up to the last operation of an
    // item, waiting for an item = s_waitcnt vmcnt(issued - mark)): the compiler's own s_waitcnt insertion answers
    // loads that are consumed one loop iteration later, with stores in flight behind them, with vmcnt(0) at every
    // use.  The loaded values live in loop-carried variables that ONLY the tied "+v" operands of the load statements
    // define, one load site per variable and no branch that selects between two (the compiler would merge two
    // definitions with copies that move the old contents before the data has arrived - seen in the ISA).
    // (b) A loader shares its SIMD with an MFMA wave, next to which EVERY instruction it issues costs 12-20 cycles
    // (in-kernel stamps: a first version with ~450 instructions per stage and loader needed 6,000 cycles per stage
    // against the MFMA waves' 4,400, which then waited 3,300 cycles per stage at the barrier).  Hence two kinds of
    // loader with short loops: piece (row, k) indices are compile-time, row addresses are scalar and advance by
    // addition, per-lane offsets are computed once per kernel, slab addresses are immediates, the scalars of a unit
    // are computed once per unit, partial pieces are written under an exec mask that comes from an SGPR pair.
    const int rowslots = a.halo_w * kSlots;
    const int krow = (rowslots + 63) >> 6;                 // 1-KiB pieces per halo row (2..4, host-checked)
    const int hh = a.halo_h;
    const int n_tl = stream2_tile_loaders(hh, krow);       // tile loaders; the other loaders drain
    const int first_tl = resident ? 0 : 1;
    const int n_dl = kLoaders2 - first_tl - n_tl;          // >= 1 (host-checked)
    const bool do_store = a.y != nullptr && !(a.ablate & 2);
    const bool use_res = a.res != nullptr && do_store;
    int* const drain_ctr = sync_ctr;                       // units whose slab every drain loader has read (x n_dl)
    if (resident) {                                        // this loader's share of the cout block's weights, once
      for (int p = li; p < 2 * ncc * MT * kKH; p += kLoaders2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(wring + p * 1024), 16, wvoff,
                                                 cb0 * ncc * kKC * MT * 1024 + p * 1024, 0, 0);
    }

    if (li - first_tl < n_tl) {
      // ======================================= tile loader =======================================
      const int tl = li - first_tl;
      const sgpr4 xsrd = make_srd(a.x, (uint32_t)a.x_bytes);
      const int rowbytes = a.rowb;
      const int ra = hh * tl / n_tl, rb = hh * (tl + 1) / n_tl;
      const int n_rows = (a.ablate & 4) ? 0 : rb - ra;     // n_rows * krow <= kMaxTP (host-checked)
      const int soff_row = a.W_in * a.in_ld * 2;
      auto run = [&](auto krow_c) {
        constexpr int KROW = decltype(krow_c)::value;
        constexpr int MAXR = kMaxTP / KROW;
        uint32_t xbyte[KROW];
        int hxk[KROW];
        unsigned long long wmask = 0;                     // lanes of the last piece of a row that exist
#pragma unroll
        for (int k = 0; k < KROW; ++k) {
          const int qq = k * 64 + lane;
          const int hx = qq / kSlots, sl = qq - hx * kSlots;
          xbyte[k] = (uint32_t)(hx * a.in_ld + sl * 8) * 2u;
          hxk[k] = qq < rowslots ? hx : (1 << 20);        // a column no image has
          if (k == KROW - 1) wmask = __ballot(qq < rowslots);
        }
        const int lane16 = lane * 16;
        u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
        asm volatile("" : "+v"(zero4));                   // (kept in registers: the data of padding rows)
        int pend_lo = 0, pend_hi = 0;                     // rows of the tile in the registers that lie inside the image
        u32x4 treg[kMaxTP];
#pragma unroll
        for (int i = 0; i < kMaxTP; ++i) treg[i] = u32x4{0u, 0u, 0u, 0u};
        auto load_tile = [&](int s) __attribute__((always_inline)) {
          int tile, cb;
          um.get(s >> sh, &tile, &cb);
          const int chunk = s & (ncc - 1);
          uint32_t t = (uint32_t)tile;
          const uint32_t n = fdiv(t, a.div_tiles_xy);
          t -= n * tiles_xy;
          const uint32_t tyi = fdiv(t, a.div_tiles_x);
          const uint32_t txi = t - tyi * a.tiles_x;
          const int iy0 = (int)tyi * a.th * a.in_mul + a.lo_y, ix0 = (int)txi * a.tw * a.in_mul + a.lo_x;
          const int cvalid = a.cin - chunk * kCC;         // channels of this chunk that exist
          uint32_t voff[KROW];
#pragma unroll
          for (int k = 0; k < KROW; ++k) {
            const int sl8 = ((k * 64 + lane) % kSlots) * 8;
            const bool ok = (unsigned)(ix0 + hxk[k]) < (unsigned)a.W_in && sl8 < cvalid;
            voff[k] = ok ? xbyte[k] + (uint32_t)(ix0 * a.in_ld * 2) : 0x80000000u;
          }
          const int row_lo = -(iy0 + ra), row_hi = a.H_in - (iy0 + ra);   // rows rr in [row_lo, row_hi) lie inside the image
          int soff = (((int)n * a.H_in + iy0 + ra) * a.W_in * a.in_ld + (int)(chunk * a.in_cs)) * 2;
          const bool whole = row_lo <= 0 && row_hi >= n_rows;
          // a row above or below the image loads the nearest image row (always a valid address) and is written as zeros
          if (!whole) soff += (row_lo > 0 ? row_lo : 0) * soff_row;
#pragma unroll
          for (int rr = 0; rr < MAXR; ++rr) {
            if (rr < n_rows) {
#pragma unroll
              for (int k = 0; k < KROW; ++k) {
                if (k == 0) buf_load16_first(treg[rr * KROW + k], voff[k], xsrd, soff);
                else buf_load16(treg[rr * KROW + k], voff[k], xsrd, soff);
              }
              if (whole || (rr >= row_lo && rr + 1 < row_hi)) soff += soff_row;
            }
          }
          pend_lo = row_lo;
          pend_hi = row_hi;
        };
        auto write_tile = [&](int s) __attribute__((always_inline)) {
          uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)(tiles + (s & 1) * a.buf_bytes + ra * rowbytes) + lane16;
          const bool whole = pend_lo <= 0 && pend_hi >= n_rows;
#pragma unroll
          for (int rr = 0; rr < MAXR; ++rr) {
            if (rr < n_rows) {
              const bool inside = whole || (rr >= pend_lo && rr < pend_hi);   // (wave-uniform) else: the conv padding
#pragma unroll
              for (int k = 0; k < KROW; ++k) {
                if (k < KROW - 1) {
                  if (inside) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(treg[rr * KROW + k]), "n"(k * 1024) : "memory");
                  else asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(zero4), "n"(k * 1024) : "memory");
                } else {
                  if (inside) asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"
                                           ::"v"(addr), "v"(treg[rr * KROW + k]), "s"(wmask), "n"(k * 1024) : "memory");
                  else asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"
                                    ::"v"(addr), "v"(zero4), "s"(wmask), "n"(k * 1024) : "memory");
                }
              }
              addr += rowbytes;                           // one VALU add per row
            }
          }
        };
        // prologue: tile 0 into LDS, tile 1 into the registers
        load_tile(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (also the resident weights)
        write_tile(0);
        if (1 < S) load_tile(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int ends = 0;                                       // unit ends seen = slabs the drain loaders have to be done with
#ifdef RTPE_CONV_STAMPS
        unsigned long long lt[6] = {0}, t0, t1, t2, t3, t4, t5;
#endif
        for (int s = 0; s < S; ++s) {
          SSTAMP(t0);
          RTPE_SBARRIER();                                 // M(s): tile s is in LDS; buffer (s+1) & 1 has no MFMA reader
          SSTAMP(t1);
          if (!resident) RTPE_SBARRIER();                  // H(s)
          const bool prev_end = s > 0 && ((s - 1) & (ncc - 1)) == ncc - 1;   // its slab lies in buffer (s-1) & 1 = (s+1) & 1
          if (prev_end && do_store) {                      // wait until the drain loaders have read that slab
            ++ends;
            while (__hip_atomic_load(drain_ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ends * n_dl) __builtin_amdgcn_s_sleep(1);
          }
          SSTAMP(t2);
          if (s + 1 < S) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile s+1 has arrived (requested a stage ago; nothing else is in this queue)
            SSTAMP(t3);
            write_tile(s + 1);
          } else { SSTAMP(t3); }
          SSTAMP(t4);
          if (s + 2 < S) load_tile(s + 2);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tile s+1 is in LDS
          SSTAMP(t5);
#ifdef RTPE_CONV_STAMPS
          lt[0] += t1 - t0; lt[2] += t2 - t1; lt[3] += t3 - t2; lt[4] += t4 - t3; lt[5] += t5 - t4;
#endif
          if ((s & (ncc - 1)) == ncc - 1) { RTPE_SBARRIER(); RTPE_SBARRIER(); }   // E(s), S(s)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RTPE_CONV_STAMPS
        if (a.dbg != nullptr && lane == 0 && tl == 0)
          for (int i = 0; i < 6; ++i) atomicAdd(&a.dbg[8 + i], lt[i]);
#endif
      };
      if (krow == 2) run(std::integral_constant<int, 2>());
      else if (krow == 3) run(std::integral_constant<int, 3>());
      else run(std::integral_constant<int, 4>());
      return;
    }

    // ======================================= drain loader =======================================
    // The slab is one array of pixel rows (ROWB bytes each, CH 16-byte pieces of which are data): piece g = pixel g / CH,
    // slot g % CH.  The n_dl drain loaders take the pieces STR at a time, STR = the largest multiple of CH their lanes
    // cover: lane q of the group owns pieces q, q + STR, q + 2 STR, ... - the same slot of pixels PSTEP apart, so the LDS
    // address of piece `it` is that of piece 0 plus a compile-time multiple of PSTEP * ROWB.
    {
      const int dl = li - first_tl - n_tl;
      // descriptors over "the rest of the address space from the view's base": offsets stay below 2^31 (the tensors are
      // smaller), 1 << 31 is out of range: such a load returns zeros, such a store is dropped
      const sgpr4 rsrd = make_srd(a.res, 0x7fffffffu), ysrd = make_srd(a.y, 0x7fffffffu);
      const int cblk = cb0 * MT * 16;
      auto run = [&](auto ndl_c) {
        constexpr int ND = decltype(ndl_c)::value;
        constexpr int PSTEP = 64 * ND / CH, STR = PSTEP * CH;      // pixels / pieces per step
        constexpr int NIT = (NPIECES + STR - 1) / STR;
        const int q = dl * 64 + lane;                     // lane of the group
        const int pw0 = q / CH, slot = q - pw0 * CH;
        const bool lane_ok = q < STR && cblk + slot * 8 < a.cout_store;
        const uint32_t e0 = (uint32_t)(pw0 * ROWB + slot * 16);
        const uint32_t row_pix = (uint32_t)(a.W_full * a.o_mul);
        uint32_t yoff[NIT];
        const bool same_ld = a.res_ld == a.out_ld;
        const uint32_t slot16 = (uint32_t)slot * 16u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const uint32_t p = (uint32_t)(pw0 + it * PSTEP);
          const uint32_t oyt = fdiv(p, a.div_tw);
          const uint32_t oxt = p - oyt * a.tw;
          const bool ok = lane_ok && (int)p < n_px && (int)p < WAVES * NT * 16;
          const uint32_t pix = oyt * row_pix + oxt * a.o_mul;
          yoff[it] = ok ? pix * (uint32_t)a.out_ld * 2u + slot16 : 0x80000000u;
        }
        // (row << 16 | column) of piece `it` in the tile: only tiles that hang over the image border ask (recomputed
        // there: 32 more registers per lane would spill)
        auto eyx = [&](int it) __attribute__((always_inline)) {
          const uint32_t p = (uint32_t)(pw0 + it * PSTEP);
          const uint32_t oyt = fdiv(p, a.div_tw);
          return (int)((oyt << 16) | ((p - oyt * a.tw) * a.o_mul));
        };
        // without a residual the registers keep -0.0 in every half: x + (-0.0) == x bit for bit (also for x = +-0.0), so
        // the add below is unconditional; the ReLU is a signed 16-bit max with 0 (x > 0 ? x : +0 on the sign bits) or
        // with -32768 (no ReLU): no select instruction per piece
        u32x4 rreg[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) rreg[it] = u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        short8 relu_lim;
        {
          const short lim = a.relu ? (short)0 : (short)-32768;
          relu_lim = short8{lim, lim, lim, lim, lim, lim, lim, lim};
          asm volatile("" : "+v"(relu_lim));
        }
        // scalars of a unit: byte offsets of its first pixel in y and in the residual, rows / columns inside the image
        struct UnitS { int ysoff, rsoff, hy, hx; bool whole; };
        auto unit_scalars = [&](int u) __attribute__((always_inline)) {
          const UnitPos qq = unit_pos(u);
          UnitS r;
          r.hy = a.H_pos - qq.py0;
          r.hx = (a.W_pos - qq.px0) * a.o_mul;
          r.whole = r.hy >= a.th && r.hx >= a.tw * a.o_mul;
          const size_t pix0 = ((size_t)qq.n * a.H_full + qq.py0 * a.o_mul + a.oy_add) * a.W_full + qq.px0 * a.o_mul + a.ox_add;
          r.ysoff = (int)((pix0 * a.out_ld + (size_t)cb0 * a.out_cs) * 2);
          r.rsoff = use_res ? (int)((pix0 * a.res_ld + (size_t)cb0 * a.res_cs) * 2) : 0;
          return r;
        };
        auto load_res = [&](const UnitS& us) __attribute__((always_inline)) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            uint32_t vo = yoff[it];
            if (!same_ld && vo != 0x80000000u)            // (rare: a residual tensor with another row length)
              vo = (((uint32_t)eyx(it) >> 16) * row_pix + ((uint32_t)eyx(it) & 0xffffu)) * (uint32_t)a.res_ld * 2u + slot16;
            if (!us.whole) vo = ((eyx(it) >> 16) < us.hy && (eyx(it) & 0xffff) < us.hx) ? vo : 0x80000000u;
            if (it == 0) buf_load16_first(rreg[it], vo, rsrd, us.rsoff);
            else buf_load16(rreg[it], vo, rsrd, us.rsoff);
          }
        };
        auto drain = [&](int u, const UnitS& us) __attribute__((always_inline)) {
          const int s_end = (u << sh) + ncc - 1;
          const __attribute__((address_space(3))) char* slab =
              (const __attribute__((address_space(3))) char*)(uintptr_t)((uint32_t)(uintptr_t)(lds_ptr_t)(tiles + (s_end & 1) * a.buf_bytes) + e0);
          // the slab pieces are read in groups of GS before the first of them is used: a read that waits behind the
          // MFMA waves' operand traffic takes ~500 cycles, and the asm stores (memory clobbers) keep the compiler from
          // hoisting the next read over them - one at a time the 16 pieces of a unit cost 8,800 cycles (stamps), which
          // the MFMA waves spent waiting at the next barrier
          constexpr int GS = 8;
#pragma unroll
          for (int g0 = 0; g0 < NIT; g0 += GS) {
            short8 v[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j)
              if (g0 + j < NIT)
                v[j] = *reinterpret_cast<const __attribute__((address_space(3))) short8*>(slab + (g0 + j) * PSTEP * ROWB);
#pragma unroll
            for (int j = 0; j < GS; ++j) {
              if (g0 + j < NIT) {
                const int it = g0 + j;
                const half8 sum = __builtin_bit_cast(half8, v[j]) + __builtin_bit_cast(half8, rreg[it]);   // fp16 RNE add = the wrapper's add
                const short8 o = __builtin_elementwise_max(__builtin_bit_cast(short8, sum), relu_lim);
                uint32_t vo = yoff[it];
                if (!us.whole) vo = ((eyx(it) >> 16) < us.hy && (eyx(it) & 0xffff) < us.hx) ? vo : 0x80000000u;
                if (j == 0) buf_store16_first(__builtin_bit_cast(u32x4, o), vo, ysrd, us.ysoff);
                else buf_store16(__builtin_bit_cast(u32x4, o), vo, ysrd, us.ysoff);
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab reads are complete
          if (lane == 0) __hip_atomic_fetch_add(drain_ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        UnitS cur = unit_scalars(0);
        if (use_res) load_res(cur);
#ifdef RTPE_CONV_STAMPS
        unsigned long long dt[4] = {0}, d0, d1, d2, d3, d4;
#endif
        for (int s = 0; s < S; ++s) {
          SSTAMP(d0);
          RTPE_SBARRIER();                                 // M(s)
          if (!resident) RTPE_SBARRIER();                  // H(s)
          SSTAMP(d1);
          const bool prev_end = s > 0 && ((s - 1) & (ncc - 1)) == ncc - 1;
          if (prev_end && do_store) {
            const int u_prev = (s - 1) >> sh;
            if (use_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // its residual rows (requested a unit ago) - and the stores before them
            SSTAMP(d2);
            drain(u_prev, cur);
            SSTAMP(d3);
            cur = unit_scalars(u_prev + 1);                // (this stage belongs to unit u_prev + 1)
            if (use_res) load_res(cur);
            SSTAMP(d4);
#ifdef RTPE_CONV_STAMPS
            dt[1] += d2 - d1; dt[2] += d3 - d2; dt[3] += d4 - d3;
#endif
          }
#ifdef RTPE_CONV_STAMPS
          dt[0] += d1 - d0;
#endif
          if ((s & (ncc - 1)) == ncc - 1) { RTPE_SBARRIER(); RTPE_SBARRIER(); }   // E(s), S(s): the slab of this unit is complete
        }
#ifdef RTPE_CONV_STAMPS
        if (a.dbg != nullptr && lane == 0 && dl == 0) {
          atomicAdd(&a.dbg[9], dt[1]);                     // (slot 9 = "drain" column: here the vmcnt wait)
          printf("");
        }
        if (a.dbg != nullptr && lane == 0 && dl == 0 && blockIdx.x == 8)
          printf("drain loader wg8: per stage waitM+H %llu | per unit: vmcnt %llu drain %llu scalars+load_res %llu (units %d)\n",
                 dt[0] / (unsigned long long)S, dt[1] / (unsigned long long)um.count, dt[2] / (unsigned long long)um.count, dt[3] / (unsigned long long)um.count, um.count);
#endif
        if (do_store) {
          if (use_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          drain(um.count - 1, cur);                        // the last unit
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      if (n_dl == 1) run(std::integral_constant<int, 1>());
      else if (n_dl == 2) run(std::integral_constant<int, 2>());
      else run(std::integral_constant<int, 3>());
      return;
    }
  }

  // -------------------------------- MFMA waves --------------------------------
  const int r = lane & 15;
  const int g = lane >> 4;
  int toff[kKC];
#pragma unroll
  for (int k = 0; k < kKC; ++k) {
    int kk = k * 32 + g * 8;
    if (kk >= 9 * kCC) kk -= 9 * kCC;                   // zero-weight k padding: any finite in-tile data
    const int tap = kk / kCC, c = kk - tap * kCC;
    const int ty = tap / 3, tx = tap - ty * 3;
    toff[k] = ty * a.rowb + tx * kPStride + c * 2;
  }
  int pixbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    uint32_t p = (wv * NT + nt) * 16 + r;
    p = (int)p < n_px ? p : (uint32_t)(n_px - 1);       // surplus MFMA tiles recompute the last pixel (never stored)
    const uint32_t oy = fdiv(p, a.div_tw);
    const uint32_t ox = p - oy * a.tw;
    pixbase[nt] = (int)(oy * a.in_mul * a.rowb + ox * a.in_mul * kPStride);
  }
  float4v acc[MT][NT];
  float4v al[MT], be[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int c4 = (cb0 * MT + m) * 16 + g * 4;
    al[m] = *reinterpret_cast<const float4v*>(a.alpha + c4);
    be[m] = *reinterpret_cast<const float4v*>(a.beta + c4);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(al[m]), "v"(be[m]));

  auto half_stage = [&](const char* wslot, const char* tilebuf, auto hsel) {
    constexpr int H = decltype(hsel)::value;
    const char* wl = wslot + lane * 16;
    half8 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const half8*>(wl + m * 1024);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const half8*>(tilebuf + pixbase[nt] + toff[H * kKH]);
#pragma unroll
    for (int kk = 0; kk < kKH; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < kKH) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          af[nxt][m] = *reinterpret_cast<const half8*>(wl + ((kk + 1) * MT + m) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[nxt][nt] = *reinterpret_cast<const half8*>(tilebuf + pixbase[nt] + toff[H * kKH + kk + 1]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cur][m], bf[cur][nt], acc[m][nt], 0, 0, 0);
      if (kk + 1 < kKH) {
#pragma unroll
        for (int i = 0; i < (MT + NT + 3) / 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int bsel = 0, wsel = 0;                                // s & 1, (2 s) % 3
  const int n_units = um.count;
#ifdef RTPE_CONV_STAMPS
  unsigned long long st[8] = {0}, m0, m1, m2, m3, m4;
  const unsigned long long k_begin = __builtin_readcyclecounter();
#endif
  for (int u = 0; u < n_units; ++u) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    char* tilebuf = nullptr;
    for (int cci = 0; cci < ncc; ++cci) {
      tilebuf = tiles + bsel * a.buf_bytes;
      const int w0 = resident ? 2 * cci : wsel;
      const int w1 = resident ? 2 * cci + 1 : (wsel == 2 ? 0 : wsel + 1);
      SSTAMP(m0);
      RTPE_SBARRIER();                                   // M(s)
      SSTAMP(m1);
      if (!(a.ablate & 1)) half_stage(wring + w0 * WSLOT, tilebuf, std::integral_constant<int, 0>());
      SSTAMP(m2);
      if (!resident) RTPE_SBARRIER();                    // H(s)
      SSTAMP(m3);
      if (!(a.ablate & 1)) half_stage(wring + w1 * WSLOT, tilebuf, std::integral_constant<int, 1>());
      SSTAMP(m4);
#ifdef RTPE_CONV_STAMPS
      st[0] += m1 - m0; st[1] += m2 - m1; st[2] += m3 - m2; st[3] += m4 - m3; st[4] += 1;
#endif
      if (cci + 1 < ncc) {
        bsel ^= 1;
        wsel = wsel == 0 ? 2 : wsel - 1;
      }
    }
    // ---- epilogue: BN / bias with the wrapper's rounding points -> fp16 rows of the slab (the tile buffer) ----
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int re = lane_e & 15, ge = lane_e >> 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SSTAMP(m0);
    RTPE_SBARRIER();                                     // E(s): every MFMA wave is done with the tile
    SSTAMP(m1);
    char* obuf = tilebuf + wv * (NT * 16 * ROWB);
    auto bn_to_lds = [&](auto rc, auto nchw) {
      constexpr bool RC = decltype(rc)::value, NCHW = decltype(nchw)::value;
      UnitPos q{0u, 0, 0};
      if (NCHW) q = unit_pos(u);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4v v = acc[m][nt];
          half4 o;
          if (RC) {
            o = bn_round(v, al[m], be[m]);
          } else {
            float2v lo{v[0], v[1]}, hi{v[2], v[3]};
            lo = __builtin_elementwise_fma(lo, float2v{al[m][0], al[m][1]}, float2v{be[m][0], be[m][1]});
            hi = __builtin_elementwise_fma(hi, float2v{al[m][2], al[m][3]}, float2v{be[m][2], be[m][3]});
            const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
            o = half4{olo[0], olo[1], ohi[0], ohi[1]};
          }
          *reinterpret_cast<half4*>(obuf + (nt * 16 + re) * ROWB + m * 32 + ge * 8) = o;
          if (NCHW) {                                    // heads: NCHW straight from the registers
            const uint32_t p = (wv * NT + nt) * 16 + re;
            const uint32_t oyt = fdiv(p, a.div_tw);
            const uint32_t oxt = p - oyt * a.tw;
            const int py = q.py0 + (int)oyt, px = q.px0 + (int)oxt;
            if ((int)p < n_px && py < a.H_pos && px < a.W_pos) {
              const int oy = py * a.o_mul + a.oy_add, ox = px * a.o_mul + a.ox_add;
              const int c4 = (cb0 * MT + m) * 16 + ge * 4;
#pragma unroll
              for (int jx = 0; jx < 4; ++jx) {
                const int c = c4 + jx;
                if (c < a.nchw_channels) {
                  const float xr = (float)o[jx];
                  const float x = a.relu ? (xr > 0.f ? xr : 0.f) : xr;
                  const size_t oi = (((size_t)q.n * a.nchw_channels + c) * a.H_full + oy) * a.W_full + ox;
                  if (a.nchw_f32)
                    reinterpret_cast<float*>(a.y_nchw)[oi] = x;
                  else
                    reinterpret_cast<_Float16*>(a.y_nchw)[oi] = (_Float16)x;
                }
              }
            }
          }
        }
      }
    };
    if (a.y_nchw != nullptr) {
      if (a.round_conv) bn_to_lds(std::true_type(), std::true_type()); else bn_to_lds(std::false_type(), std::true_type());
    } else {
      if (a.round_conv) bn_to_lds(std::true_type(), std::false_type()); else bn_to_lds(std::false_type(), std::false_type());
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's rows are in the slab
    SSTAMP(m2);
    RTPE_SBARRIER();                                     // S(s): the slab is complete, the loaders take it from here
    SSTAMP(m3);
#ifdef RTPE_CONV_STAMPS
    st[5] += m1 - m0; st[6] += m2 - m1; st[7] += m3 - m2;
#endif
    bsel ^= 1;
    wsel = wsel == 0 ? 2 : wsel - 1;
  }
#ifdef RTPE_CONV_STAMPS
  if (a.dbg != nullptr && lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&a.dbg[i], st[i]);
    atomicAdd(&a.dbg[15], __builtin_readcyclecounter() - k_begin);
    atomicAdd(&a.dbg[14], (unsigned long long)n_units);
  }
#endif
}

template <int MT, int NT>
static int launch_stream2(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  static unsigned long long attr_mask = 0;
  auto kern = conv_stream2_kernel<MT, NT>;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)t.grid), dim3(512), t.lds_bytes, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

size_t conv_stream2_lds(const ConvPlan& p, int buf_bytes, int n_wslots) {
  return (size_t)n_wslots * p.mt * kKH * 1024 + (size_t)2 * buf_bytes + kSyncBytes;
}

// the loader waves (4, one of which streams the weight ring unless the weights are resident) must cover the tile with at
// most kMaxTP register pieces each and leave at least one wave for the drain
bool conv_stream2_tile_fits(int halo_h, int halo_w, bool resident) {
  const int krow = (halo_w * kSlots + 63) / 64;
  if (krow < 2 || krow > 4) return false;
  const int n_tl = stream2_tile_loaders(halo_h, krow);
  if (((halo_h + n_tl - 1) / n_tl) * krow > kMaxTP) return false;
  return kLoaders2 - (resident ? 0 : 1) - n_tl >= 1;
}

// the drain loaders keep the residual rows of a whole unit in flight in their registers: at most 16 pieces of 16 bytes per
// lane (more would spill, and a spill of a register whose load is still in flight stores stale data: seen with the
// 32 pieces of a lone drain loader on 320-pixel tiles)
bool conv_stream2_drain_fits(int mt, int nt, int halo_h, int halo_w, bool resident) {
  const int krow = (halo_w * kSlots + 63) / 64;
  const int n_dl = kLoaders2 - (resident ? 0 : 1) - stream2_tile_loaders(halo_h, krow);
  if (n_dl < 1) return false;
  const int ch = mt * 2, pieces = 4 * nt * 16 * ch, str = (64 * n_dl / ch) * ch;
  return (pieces + str - 1) / str <= 16;
}

int conv_stream2_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(conv_stream_supports(p), "streaming conv v2: unsupported plan");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "streaming conv v2: input view of %zu bytes", (size_t)a.x_bytes);
  RTPE_REQUIRE(a.cin % 8 == 0 && (a.in_ld >= a.cin ? a.in_cs == kCC : a.in_ld == kCC && a.in_cs >= kCC),
               "streaming conv v2: cin=%d in_ld=%d chunk stride %lld", a.cin, a.in_ld, a.in_cs);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && (t.grid / 8) % p.n_cb == 0, "streaming conv v2: bad grid %d", t.grid);
  RTPE_REQUIRE(t.waves == 4 && t.n_bufs == 2 && t.buf_bytes % 16 == 0 && (t.n_wslots == 3 || t.n_wslots == 2 * p.n_cchunks) &&
               t.lds_bytes >= conv_stream2_lds(p, t.buf_bytes, t.n_wslots) && t.lds_bytes <= 160 * 1024,
               "streaming conv v2: LDS layout (%d buffers of %d B, %d weight slots, %zu B)", t.n_bufs, t.buf_bytes,
               t.n_wslots, t.lds_bytes);
  RTPE_REQUIRE(t.th * t.tw <= 16 * t.nt * t.waves && t.th * t.tw > 16 * t.nt * (t.waves - 1),
               "streaming conv v2: tile %dx%d for %d MFMA tiles", t.th, t.tw, t.nt * t.waves);
  RTPE_REQUIRE(conv_stream2_tile_fits(a.halo_h, a.halo_w, t.n_wslots != 3) &&
               conv_stream2_drain_fits(p.mt, t.nt, a.halo_h, a.halo_w, t.n_wslots != 3), "streaming conv v2: halo tile %dx%d", a.halo_h, a.halo_w);
  RTPE_REQUIRE(a.rowb >= a.halo_w * kPStride && (size_t)a.halo_h * a.rowb <= (size_t)t.buf_bytes &&
               (size_t)t.waves * t.nt * 16 * (p.mt * 32 + 16) <= (size_t)t.buf_bytes, "streaming conv v2: tile buffer too small");
#define RTPE_S2(MTv, NTv) \
  if (p.mt == MTv && t.nt == NTv) return launch_stream2<MTv, NTv>(t, a, s);
  RTPE_S2(3, 5) RTPE_S2(3, 7) RTPE_S2(3, 4) RTPE_S2(3, 2)
  RTPE_S2(2, 5) RTPE_S2(2, 4) RTPE_S2(1, 5) RTPE_S2(1, 4)
#undef RTPE_S2
  set_error("streaming conv v2: no kernel variant mt=%d nt=%d", p.mt, t.nt);
  return RTPE_E_INVALID;
}

template <int MT, int NT, int WAVES>
static int launch_stream(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  static unsigned long long attr_mask = 0;
  auto kern = conv_stream_kernel<MT, NT, WAVES>;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)t.grid), dim3((WAVES + kLoaders) * 64), t.lds_bytes, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// LDS of one workgroup: n_wslots weight half-stage slots (3 = streaming ring, 2 * n_cchunks =
// resident) + n_bufs halo tile buffers
size_t conv_stream_lds(const ConvPlan& p, int buf_bytes, int n_bufs, int n_wslots) {
  return (size_t)n_wslots * p.mt * kKH * 1024 + (size_t)n_bufs * buf_bytes;
}

bool conv_stream_supports(const ConvPlan& p) {
  return p.esize == 2 && p.dil == 1 && p.cc == kCC && p.pstride == kPStride && p.tapw == 3 && p.kc == kKC &&
         (p.in_mul == 1 || p.in_mul == 2) && p.mt <= 3 && (p.n_cchunks & (p.n_cchunks - 1)) == 0;
}

int conv_stream_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(conv_stream_supports(p), "streaming conv: unsupported plan");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "streaming conv: input view of %zu bytes", (size_t)a.x_bytes);
  // NHWC (chunk stride 48, row of >= cin channels) or plane-major (row of 48, chunk stride = one plane; the two
  // coincide for a single 1x1 map)
  RTPE_REQUIRE(a.cin % 8 == 0 && (a.in_ld >= a.cin ? a.in_cs == kCC : a.in_ld == kCC && a.in_cs >= kCC),
               "streaming conv: cin=%d in_ld=%d chunk stride %lld", a.cin, a.in_ld, a.in_cs);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && (t.grid / 8) % p.n_cb == 0, "streaming conv: bad grid %d", t.grid);
  RTPE_REQUIRE((t.n_bufs == 2 || t.n_bufs == 3) && t.buf_bytes % 16 == 0 &&
               (t.n_wslots == 3 || t.n_wslots == 2 * p.n_cchunks) &&
               t.lds_bytes >= conv_stream_lds(p, t.buf_bytes, t.n_bufs, t.n_wslots) && t.lds_bytes <= 160 * 1024,
               "streaming conv: LDS layout (%d buffers of %d B, %d weight slots, %zu B)", t.n_bufs, t.buf_bytes,
               t.n_wslots, t.lds_bytes);
  RTPE_REQUIRE(a.halo_w * kSlots <= 256, "streaming conv: halo row of %d pixels", a.halo_w);
  RTPE_REQUIRE(a.rowb >= a.halo_w * kPStride && (size_t)a.halo_h * a.rowb <= (size_t)t.buf_bytes &&
               (size_t)t.waves * t.nt * 16 * (p.mt * 32 + 16) <= (size_t)t.buf_bytes, "streaming conv: tile buffer too small");
#define RTPE_S(MTv, NTv, Wv) \
  if (p.mt == MTv && t.nt == NTv && t.waves == Wv) return launch_stream<MTv, NTv, Wv>(t, a, s);
  RTPE_S(3, 4, 4) RTPE_S(3, 5, 4) RTPE_S(3, 5, 5) RTPE_S(3, 2, 4)
  RTPE_S(2, 4, 4) RTPE_S(2, 5, 4) RTPE_S(2, 5, 5)
  RTPE_S(1, 4, 4) RTPE_S(1, 5, 4) RTPE_S(1, 5, 5)
#undef RTPE_S
  set_error("streaming conv: no kernel variant mt=%d nt=%d waves=%d", p.mt, t.nt, t.waves);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
