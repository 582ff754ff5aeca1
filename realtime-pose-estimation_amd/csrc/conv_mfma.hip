// Implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16 in/out, fp32
// accumulate on v_mfma_f32_16x16x32_f16, fused BN/bias + residual + ReLU
// epilogue with the rounding points of the reference's half wrapper.
//
// Replaces the stock nn.Conv2d + nn.BatchNorm2d (+ add, + ReLU) launches of
// rtpe/third_party/pose_higher_hrnet.py (BasicBlock :59-75, Bottleneck :96-116,
// transitions :548-583, fuse convs :200-230, final_layers :460-482, and the four
// parity classes of the k4s2 ConvTranspose2d :513-524).
//
// Mapping (MI355X-first, not a GEMM library call):
//   * one workgroup = WAVES waves = a TH x TW tile of output positions of one
//     image x one block of 16*MT output channels;
//   * the (TH*s+k-1) x (TW*s+k-1) input halo tile of CC channels is staged ONCE
//     into LDS (NHWC, 16-B slots, pixel stride == 32 mod 64 bytes so that the
//     ds_read_b128 lane groups hit distinct bank slots) and re-read by all k*k
//     taps: no im2col buffer ever exists in HBM;
//   * MFMA orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: weights are
//     the A operand, pre-packed on the host in exact fragment order (one 1-KiB
//     coalesced global_load_dwordx4 per fragment, L2-resident), pixels are the
//     B operand (one ds_read_b128 per lane: 8 consecutive channels of one tap);
//     the accumulator then holds 4 consecutive channels of one pixel per lane,
//     i.e. the NHWC store / residual load are 8-byte row pieces;
//   * K order inside a channel chunk is flat [tap][channel], padded to 32, so
//     Cin = 48 needs 14 MFMA k-steps for 9 taps instead of 18.
#include <vector>

#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kTapTableBytes = 2048;      // 4 ints per k chunk: up to 128 k chunks (5x5 taps x 64 fp32 channels = 100)

__device__ __forceinline__ float round16(float v) { return (float)(_Float16)v; }

// storage precision: fp16 (the half wrapper: v_mfma_f32_16x16x32_f16, one fp16 rounding after
// conv, BN and add) or fp32 (plain PoseHigherResolutionNet: v_mfma_f32_16x16x4_f32 = exact fp32
// FMA chains, no intermediate rounding).  Both MFMAs share the C/D layout, hence the epilogue.
template <typename T> struct Prec;
template <> struct Prec<_Float16> {
  static constexpr int EPS = 8;                                   // elements per 16-byte slot
  static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};
template <> struct Prec<float> {
  static constexpr int EPS = 4;
  static __device__ __forceinline__ float rnd(float v) { return v; }
};

template <typename T>
__device__ __forceinline__ float4v mfma_step(const uint4& av, const uint4& bv, float4v acc);
template <>
__device__ __forceinline__ float4v mfma_step<_Float16>(const uint4& av, const uint4& bv, float4v acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, av), __builtin_bit_cast(half8, bv), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ float4v mfma_step<float>(const uint4& av, const uint4& bv, float4v acc) {
  // element j of every lane group forms one k=4 step: channels {j, 4+j, 8+j, 12+j} of the chunk
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av.x), __uint_as_float(bv.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av.y), __uint_as_float(bv.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av.z), __uint_as_float(bv.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av.w), __uint_as_float(bv.w), acc, 0, 0, 0);
  return acc;
}

// k chunks of weight fragments in flight per wave; a step keeps the matrix pipe busy for step_cycles (16 per fp16 tile
// pair, 4 x 32 per fp32 one)
#ifndef RTPE_WEIGHT_RING
#define RTPE_WEIGHT_RING 0
#endif
__host__ __device__ constexpr int conv_weight_ring(int mt, int step_cycles) {
  // ~1000 cycles of matrix work in flight, at most 8 chunks and 48 registers
  int d = (1000 + step_cycles - 1) / step_cycles;
  d = d > 8 ? 8 : d;
  d = d > 12 / mt ? 12 / mt : d;
  d = (step_cycles >= 256 && d > 2) ? 2 : d;     // 16 and more accumulator tiles: the registers are theirs
  return RTPE_WEIGHT_RING ? RTPE_WEIGHT_RING : d < 1 ? 1 : d;
}

template <typename T, int MT, int NT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPS = Prec<T>::EPS;
  constexpr int KCH = 4 * EPS;                 // k values per k chunk: 32 (fp16) / 16 (fp32)
  constexpr int ES = (int)sizeof(T);
#ifdef RTPE_CONV_STAMPS
#define RTPE_STAMP(i)                                   \
  do {                                                  \
    __builtin_amdgcn_sched_barrier(0);                  \
    stamp[i] = __builtin_readcyclecounter();            \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
  unsigned long long stamp[8];
  unsigned long long seg_stage = 0, seg_k = 0;
  RTPE_STAMP(0);
#else
#define RTPE_STAMP(i)
#endif
  int* tapoff = reinterpret_cast<int*>(smem);
  char* tile = smem + kTapTableBytes;

  constexpr int NTHREADS = WAVES * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int r = lane & 15;
  const int g = lane >> 4;

  // block -> (image, tile row, tile col), cout block.  Workgroups are dealt to the 8 XCDs round robin
  // (blockIdx.x % 8): the n_cb cout blocks of one tile take consecutive slots of ONE XCD, so they run at the same
  // time and share the tile's input through that XCD's L2.  (With the cout block as the slow grid dimension every
  // block of a layer re-read the whole input from HBM: 755 MB fetched per launch of the 64 -> 256 1x1 convs of
  // layer1 instead of 440 MB, PMC FETCH_SIZE.)
  // A transposed conv runs its 4 sub-pixel classes in ONE grid (n_cls = 4) as further "blocks" of a tile: their
  // workgroups write interleaved 96-byte pieces of the same output lines at the same time on one XCD, so the
  // pieces meet in that L2 and HBM sees whole lines (4 separate launches: 1.5 GB of traffic per deconv for 0.46 GB
  // of tensors - every partial line was read back - PMC), and they share the input tile.
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  // m_split > 1: a packed cout block (MT * m_split cout tiles) is shared out to m_split workgroups of MT tiles each
  // (small grids - batch 1, the /16 and /32 branches: three times the workgroups, a third of the k chain each; the
  // k order of an output does not change)
  // wsplit == 2: the shares of a packed cout block are dealt to the two HALVES of a workgroup's waves instead of to two
  // workgroups (ConvTile::mrun < 0): both halves read the same staged tile, each wave twice the pixels - the weight fragments a
  // wave pulls through L1 per MFMA halve (stride-2 convs with >= 96 input channels: four waves of MT = 6, NT = 2 each fetched the
  // same 6 KiB per k step, 128 B per clock and CU against the vector cache's 64)
  const uint32_t wsplit = a.wsplit > 1 ? 2u : 1u;
  const int pwv = WAVES / (int)wsplit;                         // waves that share the tile's pixels
  const int wgx = wsplit > 1 ? wv / pwv : 0;                   // this wave's half
  const int wvp = wv - wgx * pwv;                              // its index among the waves of that half
  const uint32_t m_split = a.m_split > 1 ? (uint32_t)a.m_split : 1u;
  const uint32_t wg_split = m_split / wsplit;                 // shares dealt to workgroups
  const uint32_t n_cbs = (uint32_t)a.n_cb * wg_split;
  const uint32_t per_tile = n_cbs * (uint32_t)(a.n_cls > 0 ? a.n_cls : 1);
  const uint32_t tq = slot / per_tile;
  // an XCD owns a CONTIGUOUS eighth of the row-major tile list: neighbouring tiles share their halo rows / columns and
  // the partial 128-byte lines at their edges in one L2 (tile t on XCD t % 8 fetched them once per XCD: 1.13-1.23 x
  // the input on the 256-channel layers, PMC round 2)
  uint32_t t = xcd * (gridDim.x / (8u * per_tile)) + tq;
  if (t >= (uint32_t)a.N * (uint32_t)(a.tiles_x * a.tiles_y)) return;      // grid padding (whole workgroup)
  const uint32_t sub = slot - tq * per_tile;
  const int cls = (int)(sub / n_cbs);
  const _Float16* const w_base = a.n_cls > 0 ? a.w_c[cls] : a.w;
  const int lo_y = a.n_cls > 0 ? a.lo_yc[cls] : a.lo_y, lo_x = a.n_cls > 0 ? a.lo_xc[cls] : a.lo_x;
  const int oy_add = a.n_cls > 0 ? a.oy_c[cls] : a.oy_add, ox_add = a.n_cls > 0 ? a.ox_c[cls] : a.ox_add;
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  const uint32_t n = fdiv(t, a.div_tiles_xy);
  t -= n * tiles_xy;
  const uint32_t tyi = fdiv(t, a.div_tiles_x);
  const uint32_t txi = t - tyi * a.tiles_x;
  const uint32_t cbs = sub - (uint32_t)cls * n_cbs;
  const int cb = (int)(cbs / wg_split);                       // packed cout block
  const int msel = (int)(cbs - (uint32_t)cb * wg_split) * (int)wsplit + wgx;   // this workgroup's (this half's) share of it
  const int mt_pack = MT * (int)m_split;
  const int co_base = (cb * mt_pack + msel * MT) * 16;        // first output channel of this workgroup
  const int py0 = tyi * a.th, px0 = txi * a.tw;
  const int iy0 = py0 * a.in_mul + lo_y, ix0 = px0 * a.in_mul + lo_x;

  // LDS byte offset of (tap, channel) for every (k chunk, lane group)
  const int kvalid = a.ntaps * a.cc;
  for (int i = tid; i < a.kc * 4; i += NTHREADS) {
    int kk = (i >> 2) * KCH + (i & 3) * EPS;
    if (kk >= kvalid) kk -= kvalid;  // zero-weight padding: any finite in-tile data will do
    const int tap = fdiv(kk, a.div_cc);
    const int c = kk - tap * a.cc;
    const int tyy = (a.tapw == 3) ? (tap * 11 >> 5) : (a.tapw == 2 ? (tap >> 1) : (a.tapw == 5 ? (tap * 13 >> 6) : 0));
    const int txx = tap - tyy * a.tapw;
    // (stride 2, de-interleaved columns: tap column txx of output column ox is input column 2 ox + txx = LDS column
    // (txx & 1) * n_even + ox + (txx >> 1))
    const int colslot = a.deint ? (txx & 1) * a.n_even + (txx >> 1) : txx * a.dil;
    tapoff[i] = tyy * a.dil * a.rowb + colslot * a.pstride + c * ES;
  }

  // per-lane pixel of each N tile
  int pixbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const uint32_t p = (wvp * NT + nt) * 16 + r;
    const uint32_t oy = fdiv(p, a.div_tw);
    const uint32_t ox = p - oy * a.tw;
    pixbase[nt] = (int)(oy * a.in_mul * a.rowb + ox * (a.deint ? 1 : a.in_mul) * a.pstride);
  }

  float4v acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};

  const int n_k = a.n_cchunks * a.kc;  // total k chunks
  const uint4* wfrag = reinterpret_cast<const uint4*>(w_base) + ((size_t)cb * n_k * mt_pack + msel * MT) * 64 + lane;

  // Weight fragments come straight from memory (L2 after the first workgroup): a ring of RING k chunks in flight per
  // wave.  One k chunk keeps the matrix pipe busy for MT * NT * 16 (fp16) / * 128 (fp32) cycles, a fragment takes
  // 500-800 cycles to arrive: with one chunk in flight (rounds 1-3) the small tiles - 8 x 8 stride-2 tiles, shared-out
  // cout blocks, everything at batch 1 - waited for their weights most of the time.
  constexpr int RING = conv_weight_ring(MT, MT * NT * (ES == 4 ? 128 : 16));
  uint4 a_ring[RING][MT];
#pragma unroll
  for (int d = 0; d < RING; ++d) {
    const int kd = d < n_k ? d : n_k - 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) a_ring[d][m] = wfrag[(size_t)(kd * mt_pack + m) * 64];
  }
  int phase = 0;                          // slot of the first k chunk of the current channel chunk

  const int slots = a.cc / EPS;              // 16-B slots per staged pixel
  const int rowslots = a.halo_w * slots;     // per halo row
  const int total = a.halo_h * rowslots;
  const T* xin = reinterpret_cast<const T*>(a.x) + (size_t)n * a.H_in * a.W_in * a.in_ld;

  // residual row pieces of this workgroup's tile (the same (pixel, 16-byte slot) walk as the store
  // loop of the epilogue).  All of them are requested together, and as early as the register budget
  // allows, so that one memory round trip is paid per tile instead of one per row piece.
  constexpr int CHG = MT * 16 / EPS;                    // 16-byte chunks per pixel row
  constexpr int NITG = (NT * 16 * CHG + 63) / 64;
  constexpr bool kEarlyRes = NITG <= 4;                 // 16 VGPRs: small tiles only, the others keep their occupancy
  uint4 rpre[kEarlyRes ? NITG : 1];
  auto issue_res = [&]() {
    const T* rin = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int it = 0; it < (kEarlyRes ? NITG : 0); ++it) {
      const int c = it * 64 + lane;
      const int pw = c / CHG, slot = c - pw * CHG;
      const uint32_t p = wvp * NT * 16 + pw;
      const uint32_t oyt = fdiv(p, a.div_tw);
      const uint32_t oxt = p - oyt * a.tw;
      const int py = py0 + (int)oyt, px = px0 + (int)oxt;
      const int ch = co_base + slot * EPS;
      rpre[it] = uint4{0u, 0u, 0u, 0u};
      if (c < NT * 16 * CHG && py < a.H_pos && px < a.W_pos && ch < a.cout_store) {
        const int oy = py * a.o_mul + oy_add, ox = px * a.o_mul + ox_add;
        const size_t pix = ((size_t)n * a.H_full + oy) * a.W_full + ox;
        rpre[it] = *reinterpret_cast<const uint4*>(rin + pix * a.res_ld + ch);
      }
    }
  };
  if (kEarlyRes && a.res != nullptr) issue_res();

  int kf = 0;  // linear k-chunk index over (channel chunk, k chunk)
  RTPE_STAMP(1);
  for (int cci = 0; cci < a.n_cchunks; ++cci) {
    RTPE_STAMP(2);
    if (cci > 0) __syncthreads();
    // ---- stage the halo tile of channels [cci*cc, cci*cc+cc) ----
    const int cbase = cci * a.cc;
    if (a.dma_stage) {
      // LDS-DMA: a wave-instruction fills 64 consecutive 16-byte slots of the tile image (padding slots and pixels
      // outside the image get an out-of-range offset: the bounds check writes zeros), no registers in between and
      // ONE memory round trip per chunk however large the tile (through registers, 8 loads per lane at a time: a
      // 17 x 17 x 64-channel tile is 9.03 of them = two round trips; 22 x 18: two; 33 x 17 x 48: two)
      const int tile_slots = (a.halo_h * a.rowb) >> 4;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xin), 0, (int)a.img_bytes, 0x00020000);
      for (int ib = __builtin_amdgcn_readfirstlane(wv * 64); ib < tile_slots; ib += NTHREADS) {
        const uint32_t L = (uint32_t)(ib + lane);
        const uint32_t hy = fdiv(L, a.div_rowb16);
        const uint32_t rem = L - hy * ((uint32_t)a.rowb >> 4);
        const uint32_t hx = fdiv(rem, a.div_ps16);
        const uint32_t s = rem - hx * ((uint32_t)a.pstride >> 4);
        // LDS column hx holds input column: even ones first, then the odd ones (stride 2) / itself
        const int hxs = a.deint ? ((int)hx < a.n_even ? 2 * (int)hx : 2 * ((int)hx - a.n_even) + 1) : (int)hx;
        const int iy = iy0 + (int)hy, ix = ix0 + hxs;
        const bool ok = (int)hx < a.halo_w && (int)s < slots && (unsigned)iy < (unsigned)a.H_in &&
                        (unsigned)ix < (unsigned)a.W_in && cbase + (int)s * EPS < a.cin;
        const uint32_t voff = ok ? (uint32_t)((iy * a.W_in + ix) * a.in_ld + cbase + (int)s * EPS) * (uint32_t)ES : 0x80000000u;
        if (L < (uint32_t)tile_slots)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + ib * 16), 16, (int)voff, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    // all loads of a batch are issued before the first LDS write: 8 x 16 B in flight per lane
    for (int base = 0; base < total; base += NTHREADS * 8) {
      uint4 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * NTHREADS + tid;
        const uint32_t hy = fdiv(idx, a.div_rowslots);
        const uint32_t q = idx - hy * rowslots;
        const uint32_t hx = fdiv(q, a.div_slots);
        const uint32_t s = q - hx * slots;
        const int iy = iy0 + (int)hy, ix = ix0 + (int)hx;
        const uint32_t hxl = a.deint ? ((hx & 1u) ? (uint32_t)a.n_even + (hx >> 1) : (hx >> 1)) : hx;   // LDS column of input column hx
        dst[u] = idx < total ? (int)(hy * a.rowb + hxl * a.pstride + s * 16) : -1;
        v[u] = make_uint4(0, 0, 0, 0);
        if (idx < total && (unsigned)iy < (unsigned)a.H_in && (unsigned)ix < (unsigned)a.W_in &&
            cbase + (int)s * EPS < a.cin)
          v[u] = *reinterpret_cast<const uint4*>(xin + ((size_t)iy * a.W_in + ix) * a.in_ld + cbase + s * EPS);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) *reinterpret_cast<uint4*>(tile + dst[u]) = v[u];
    }
    __syncthreads();
    RTPE_STAMP(3);

    // ---- k loop over [tap][channel] of this chunk ----
    // slot d of the ring holds k chunk kf + d at the start of a group of RING steps; a step uses its slot and refills
    // it with the chunk RING further on
    uint4 b_nxt[NT];
    int off_nxt;
    auto k_step = [&](int kci, int d) {
      uint4 a_cur[MT];
      const int kn = (kf + RING < n_k) ? kf + RING : n_k - 1;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a_cur[m] = a_ring[d][m];
        a_ring[d][m] = wfrag[(size_t)(kn * mt_pack + m) * 64];
      }
      // the B operands of the NEXT step are requested before this step's MFMAs, its tap offset one step earlier
      // still (two dependent LDS round trips per step otherwise: with one wave per SIMD - small grids - nothing hid them)
      uint4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        b[nt] = b_nxt[nt];
        b_nxt[nt] = *reinterpret_cast<const uint4*>(tile + pixbase[nt] + off_nxt);
      }
      off_nxt = tapoff[(kci + 2 < a.kc ? kci + 2 : a.kc - 1) * 4 + g];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_step<T>(a_cur[m], b[nt], acc[m][nt]);
      ++kf;
    };
    {
      const int off0 = tapoff[g];
      off_nxt = tapoff[(a.kc > 1 ? 1 : 0) * 4 + g];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = *reinterpret_cast<const uint4*>(tile + pixbase[nt] + off0);
    }
    // (k chunk kf lives in slot kf % RING, a compile-time index in every step below: a channel chunk whose kc is not a
    // multiple of RING starts in the middle of a group - `phase` - and runs head, whole groups, tail)
    int kci = 0;
    if constexpr (RING > 1) {
      if (phase != 0) {
#pragma unroll
        for (int d = 1; d < RING; ++d)
          if (d >= phase && kci < a.kc) { k_step(kci, d); ++kci; }
      }
    }
    for (; kci + RING <= a.kc; kci += RING) {
#pragma unroll
      for (int d = 0; d < RING; ++d) k_step(kci + d, d);
    }
    if constexpr (RING > 1) {
#pragma unroll
      for (int d = 0; d < RING - 1; ++d)
        if (kci + d < a.kc) k_step(kci + d, d);
      phase = (phase + a.kc) % RING;
    }
    RTPE_STAMP(4);
#ifdef RTPE_CONV_STAMPS
    seg_stage += stamp[3] - stamp[2];
    seg_k += stamp[4] - stamp[3];
#endif
  }
  RTPE_STAMP(5);

  // ---- epilogue: lane holds channels cbase4..cbase4+3 of pixel (nt, r) ----
  float4v al[MT], be[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int c4 = co_base + m * 16 + g * 4;
    al[m] = *reinterpret_cast<const float4v*>(a.alpha + c4);
    be[m] = *reinterpret_cast<const float4v*>(a.beta + c4);
  }
  // The accumulator layout (4 channels of one pixel per lane) would make the NHWC
  // store and the residual load small pieces scattered over 16 rows per
  // instruction.  Instead each wave transposes its 16*NT pixels x 16*MT channels
  // through its own slice of the (now free) input-tile LDS and then moves whole
  // rows with 16-byte accesses: consecutive lanes cover consecutive bytes of a
  // pixel row, consecutive pixels of a tile row are contiguous in NHWC.
  constexpr int ROWB = MT * 16 * ES + 16;  // LDS bytes per pixel row (+16: spreads the writes over banks)
  constexpr int CH = MT * 16 / EPS;        // 16-byte chunks per pixel row
  __syncthreads();                         // every wave is done reading the input tile
  char* obuf = tile + wv * (NT * 16 * ROWB);
  // NCHW fp32 output as 16-byte row pieces (below) when four consecutive positions are four consecutive output pixels
  const bool nchw_rows = a.y_nchw != nullptr && a.nchw_f32 && a.o_mul == 1 && a.res == nullptr && (a.tw & 3) == 0 &&
                         (a.W_pos & 3) == 0 && (a.W_full & 3) == 0 && (ox_add & 3) == 0 && ((uintptr_t)a.y_nchw & 15) == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float4v v = acc[m][nt];
      T o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = v[j];
        if (a.round_conv) x = Prec<T>::rnd(x);
        // BatchNorm is an fp32 op of the wrapper whose fp32 result is then cast: two roundings.  Keep the fp32
        // value opaque, or the compiler fuses fma + cast into v_fma_mix*_f16, which rounds the exact a*b+c once
        // (1 output in ~40,000 then differs by one fp16 step from the streaming kernel and the reference)
        float t = __builtin_fmaf(x, al[m][j], be[m][j]);
        asm volatile("" : "+v"(t));
        x = Prec<T>::rnd(t);
        v[j] = x;
        o[j] = (T)x;
      }
      acc[m][nt] = v;                                  // (the NCHW pass below re-reads the finished values)
      T* orow = reinterpret_cast<T*>(obuf + (nt * 16 + r) * ROWB) + m * 16 + g * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) orow[j] = o[j];
      if (a.y_nchw != nullptr && !nchw_rows) {          // heads: NCHW fp32/fp16 straight from the registers
        const uint32_t p = (wvp * NT + nt) * 16 + r;
        const uint32_t oyt = fdiv(p, a.div_tw);
        const uint32_t oxt = p - oyt * a.tw;
        const int py = py0 + (int)oyt, px = px0 + (int)oxt;
        if (py < a.H_pos && px < a.W_pos) {
          const int oy = py * a.o_mul + oy_add, ox = px * a.o_mul + ox_add;
          const int c4 = co_base + m * 16 + g * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = c4 + j;
            if (c < a.nchw_channels) {
              const float x = a.relu ? (v[j] > 0.f ? v[j] : 0.f) : v[j];
              const size_t oi = (((size_t)n * a.nchw_channels + c) * a.H_full + oy) * a.W_full + ox;
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
  if (a.y != nullptr) {
    const int cblk = co_base;
    T* yout = reinterpret_cast<T*>(a.y);
    const T* rin = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int it = 0; it < (NT * 16 * CH + 63) / 64; ++it) {
      const int c = it * 64 + lane;                 // chunk index inside this wave's pixels
      if (c >= NT * 16 * CH) continue;
      const int pw = c / CH, slot = c - pw * CH;    // pixel of the wave, 16-byte slot of its row
      const uint32_t p = wvp * NT * 16 + pw;
      const uint32_t oyt = fdiv(p, a.div_tw);
      const uint32_t oxt = p - oyt * a.tw;
      const int py = py0 + (int)oyt, px = px0 + (int)oxt;
      const int ch = cblk + slot * EPS;
      if (py >= a.H_pos || px >= a.W_pos || ch >= a.cout_store) continue;
      const int oy = py * a.o_mul + oy_add, ox = px * a.o_mul + ox_add;
      const size_t pix = ((size_t)n * a.H_full + oy) * a.W_full + ox;
      uint4 raw = *reinterpret_cast<const uint4*>(obuf + pw * ROWB + slot * 16);
      T v[EPS];
      __builtin_memcpy(v, &raw, 16);
      if (rin != nullptr) {
        uint4 rraw;
        if constexpr (kEarlyRes) rraw = rpre[it];
        else rraw = *reinterpret_cast<const uint4*>(rin + pix * a.res_ld + ch);
        T rr[EPS];
        __builtin_memcpy(rr, &rraw, 16);
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = (T)((float)v[j] + (float)rr[j]);
      }
      if (a.relu) {
#pragma unroll
        for (int j = 0; j < EPS; ++j) v[j] = v[j] > (T)0.f ? v[j] : (T)0.f;
      }
      __builtin_memcpy(&raw, v, 16);
      // (the four sub-pixel classes of a transposed conv write interleaved pieces of the same 128-byte lines, which
      // must meet in L2: plain stores there - write-through made the deconv 222 -> 259 us)
      if (a.o_mul == 1) store16_wt(yout + pix * a.out_ld + ch, raw);
      else *reinterpret_cast<uint4*>(yout + pix * a.out_ld + ch) = raw;
    }
  }
  if (nchw_rows) {
    // heads, fp32 NCHW (tofp32 folded in): four consecutive pixels of a channel plane per lane, 16-byte stores - a tile row
    // of a plane is one contiguous piece (straight from the registers a store instruction wrote four 64-byte segments:
    // 176 us for the 17-channel head at 320 x 320).  Round 5: the values come from a CHANNEL-major copy of the wave's
    // slab ([channel][pixel], pitch NT * 16 elements + 8 bytes: the four channel groups of a write land in four different
    // bank octets), so a lane's four pixels are ONE 8-byte (fp16) / 16-byte (fp32) read; reading them as four elements
    // of the pixel-major slab hit two banks with all 64 lanes (counter: 82 % of the LDS cycles were bank conflicts).
    // The slab is the wave's own: its pixel-major rows (read above by the NHWC stores, in program order) are overwritten.
    constexpr int CPITCH = NT * 16 * ES + 8;
    constexpr bool kPlanes = MT * 16 * CPITCH <= NT * 16 * ROWB;      // (not for the 16-pixel waves of the smallest tiles)
    if constexpr (kPlanes) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (2-byte writes: two neighbouring lanes share a dword - a bank conflict each, 17 % of this kernel's LDS cycles.
            // Pairing them with a DPP row shift into one 4-byte write was tried: correct here, but the same code in the
            // direct head kernel gave wrong values for three of four channels - not worth the risk for a store-bound phase)
            *reinterpret_cast<T*>(obuf + (m * 16 + g * 4 + j) * CPITCH + (nt * 16 + r) * ES) = (T)acc[m][nt][j];
          }
    }
    constexpr int QUADS = NT * 4;
    int n_ch = a.nchw_channels - co_base;
    n_ch = n_ch > MT * 16 ? MT * 16 : n_ch;
    for (int idx = lane; idx < QUADS * n_ch; idx += 64) {
      const int c = idx / QUADS, q = idx - c * QUADS;
      const uint32_t p = wvp * NT * 16 + q * 4;
      const uint32_t oyt = fdiv(p, a.div_tw);
      const uint32_t oxt = p - oyt * a.tw;
      const int py = py0 + (int)oyt, px = px0 + (int)oxt;
      if (py >= a.H_pos || px >= a.W_pos) continue;
      T in4[4];
      if constexpr (kPlanes) {
        __builtin_memcpy(in4, obuf + c * CPITCH + q * 4 * ES, 4 * ES);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) in4[i] = *reinterpret_cast<const T*>(obuf + (q * 4 + i) * ROWB + c * ES);
      }
      float4v out;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x = (float)in4[i];
        out[i] = a.relu ? (x > 0.f ? x : 0.f) : x;
      }
      const size_t oi = (((size_t)n * a.nchw_channels + co_base + c) * a.H_full + (py + oy_add)) * a.W_full + px + ox_add;
      *reinterpret_cast<float4v*>(reinterpret_cast<float*>(a.y_nchw) + oi) = out;
    }
  }
#ifdef RTPE_CONV_STAMPS
  RTPE_STAMP(6);
  if (a.dbg != nullptr && lane == 0) {
    atomicAdd(&a.dbg[0], stamp[1] - stamp[0]);   // setup (tap table, pixel bases, first weights)
    atomicAdd(&a.dbg[1], seg_stage);             // staging incl. barriers
    atomicAdd(&a.dbg[2], seg_k);                 // k loops
    atomicAdd(&a.dbg[3], stamp[6] - stamp[5]);   // epilogue
    atomicAdd(&a.dbg[4], stamp[6] - stamp[0]);   // whole wave
    atomicAdd(&a.dbg[5], 1ull);
  }
#endif
}

// (diagnostic stamp flush is emitted at the end of the kernel body above)
// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

ConvPlan conv_make_plan(const ConvGeom& g) {
  ConvPlan p;
  memset(&p, 0, sizeof(p));
  const bool dc = g.deconv_class >= 0;
  p.tapw = dc ? 2 : g.ksize;
  p.in_mul = dc ? 1 : g.stride;
  p.cin = g.cin;
  p.cout = g.cout;
  if (dc) {
    p.lo_y = (g.deconv_class >> 1) == 0 ? -1 : 0;
    p.lo_x = (g.deconv_class & 1) == 0 ? -1 : 0;
  } else {
    p.lo_y = p.lo_x = -(g.ksize / 2);
  }
  p.mt = (g.cout % 48 == 0 || g.cout < 48) ? 3 : 4;
  if (g.cout <= 32) p.mt = (g.cout + 15) / 16;
  // stride-2 3x3 convs with >= 96 input channels (the 256 -> 96 transition, pose_higher_hrnet.py:571-581, and the
  // downsampling convs of the fuse layers :213-230): their workgroups spend their time staging (33 x 17)-pixel halo
  // tiles of 2-4 channel chunks; 96 output channels per workgroup stage them once instead of once per 48-channel
  // block (256 -> 96: 577 -> 368 us; 96 -> 192: 85 -> 63 us; 192 -> 384: 123 -> 99 us)
  static const int wide = env_int("RTPE_CONV_MT6", 1);
  // round 5: also with 48 input channels (48 -> 96 / 192 / 384 of the fuse layers): on the streaming kernel every 48-cout block
  // of a tile pulls the (2 th + 1) x (2 tw + 1) halo through its own compute unit - 54 KiB for 12 KiB of output; 96 couts per
  // workgroup on 8 x 8 tiles halve the halo bytes per output
  static const int wide48 = env_int("RTPE_CONV_MT6_48", 1);
  if (wide && !dc && g.stride == 2 && g.ksize == 3 && (g.esize == 0 || g.esize == 2) && g.cout % 96 == 0 &&
      (g.cin >= 96 || (wide48 && g.cin == 48)))
    p.mt = 6;
  p.cout_pad = round_up(g.cout, 16 * p.mt);
  p.n_cb = p.cout_pad / (16 * p.mt);
  const int eps = 16 / (g.esize ? g.esize : 2);          // elements per 16-byte slot
  const int kch = 4 * eps;                               // k values per k chunk
  p.esize = g.esize ? g.esize : 2;
  p.dil = g.dil > 0 ? g.dil : 1;
  if (!dc) p.lo_y = p.lo_x = -(g.ksize / 2) * p.dil;
  const int cin8 = round_up(g.cin, eps);
  if (p.esize == 4) {
    // fp32: chunks of 48 / 64 / 32 / 16 channels (multiples of the 16-channel k chunk)
    if (cin8 % 48 == 0) p.cc = 48;
    else if (cin8 % 64 == 0) p.cc = 64;
    else if (cin8 % 32 == 0) p.cc = 32;
    else p.cc = round_up(cin8 < 64 ? cin8 : 48, 16);
  } else if (p.tapw == 1) {
    // 1x1: largest chunk <= 128 that divides cin and is a multiple of 32
    p.cc = 0;
    for (int c = 128; c >= 32; c -= 32)
      if (cin8 % c == 0) { p.cc = c; break; }
    if (!p.cc) p.cc = round_up(cin8 < 128 ? cin8 : 64, 16);
  } else {
    if (cin8 % 48 == 0) p.cc = 48;
    else if (cin8 % 64 == 0) p.cc = 64;
    else if (cin8 % 32 == 0) p.cc = 32;
    else p.cc = round_up(cin8 < 64 ? cin8 : 48, 16);
  }
  if (p.tapw >= 5 && p.cc > 32) p.cc = 32;              // 25 taps: a stride-2 halo tile of 64 fp32 channels would not fit the LDS
  p.n_cchunks = (cin8 + p.cc - 1) / p.cc;
  p.kc = (p.tapw * p.tapw * p.cc + kch - 1) / kch;
  p.pstride = p.cc * p.esize;
  while (p.pstride % 64 != 32) p.pstride += 16;
  p.packed_bytes = (size_t)p.n_cb * p.n_cchunks * p.kc * p.mt * 64 * 16;
  return p;
}

void conv_pack_weights(const ConvGeom& g, const ConvPlan& p, const void* w_in, void* packed_out) {
  const int ntaps = p.tapw * p.tapw;
  const bool dc = g.deconv_class >= 0;
  const int ca = dc ? (g.deconv_class >> 1) : 0, cbb = dc ? (g.deconv_class & 1) : 0;
  const int es = p.esize, eps = 16 / es, kch = 4 * eps;
  const char* w = reinterpret_cast<const char*>(w_in);
  char* packed = reinterpret_cast<char*>(packed_out);
  size_t o = 0;
  for (int cb = 0; cb < p.n_cb; ++cb)
    for (int cci = 0; cci < p.n_cchunks; ++cci)
      for (int kci = 0; kci < p.kc; ++kci)
        for (int m = 0; m < p.mt; ++m)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < eps; ++j, o += es) {
              const int co = (cb * p.mt + m) * 16 + (lane & 15);
              const int kk = kci * kch + eps * (lane >> 4) + j;
              const int tap = kk / p.cc, c = cci * p.cc + kk % p.cc;
              memset(packed + o, 0, es);
              if (tap < ntaps && co < g.cout && c < g.cin) {
                const int ty = tap / p.tapw, tx = tap % p.tapw;
                size_t src;
                if (!dc) {
                  src = (((size_t)co * g.cin + c) * g.ksize + ty) * g.ksize + tx;
                } else {
                  // oy = 2*iy - 1 + ky  =>  class a=0: dy=-1 -> ky=3, dy=0 -> ky=1
                  //                         class a=1: dy=0  -> ky=2, dy=+1 -> ky=0
                  const int ky = ca == 0 ? (ty == 0 ? 3 : 1) : (ty == 0 ? 2 : 0);
                  const int kx = cbb == 0 ? (tx == 0 ? 3 : 1) : (tx == 0 ? 2 : 0);
                  src = (((size_t)c * g.cout + co) * 4 + ky) * 4 + kx;  // IOHW
                }
                memcpy(packed + o, w + src * es, es);
              }
            }
}

// ds_read_b128 serves a wave in four groups of 16 lanes over 64 banks of 4 bytes (MI355X guide, LDS section).  The
// 16 lanes r of a B-operand read are 16 consecutive output pixels: with pstride % 64 == 32 bytes they are
// conflict-free as long as they lie in one tile row.  When tw is not a multiple of 16 a column tile wraps to the
// next row, in_mul * rowb - tw * in_mul * pstride bytes further than the next pixel would be: the row pitch is
// padded until that difference is a multiple of the 256-byte bank period (model: 6.4 -> 4.0 LDS cycles per read
// for 20-wide tiles, 4.8 -> 4.0 for 40-wide ones).
int conv_row_pitch(const ConvPlan& p, int tw, int kind) {
  const int hw = (tw - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1;
  int rb = hw * p.pstride;
  static const int pad = getenv("RTPE_CONV_ROWPAD") ? atoi(getenv("RTPE_CONV_ROWPAD")) : 1;
  // (de-interleaved stride-2 tiles of the one-workgroup-per-tile kernel: consecutive output pixels are pstride apart)
  const int step = conv_deint(p, kind) ? p.pstride : p.in_mul * p.pstride;
  if (pad && p.esize == 2 && tw % 16 != 0)
    while ((p.in_mul * rb - tw * step) % 256 != 0) rb += 16;
  return rb;
}

// stride-2 convs on the one-workgroup-per-tile kernel (kind 0) stage their tiles with the even input columns first
// (RTPE_S2_DEINT=0: in input order, the B-operand reads of 16 consecutive output pixels then hit every bank four times)
bool conv_deint(const ConvPlan& p, int kind) {
  static const int on = getenv("RTPE_S2_DEINT") ? atoi(getenv("RTPE_S2_DEINT")) : 1;
  return on && kind == 0 && p.in_mul == 2 && p.dil == 1;
}

// the instantiations of conv_mfma_kernel: (cout tiles, pixel tiles per wave, waves)
#define RTPE_CONV_VARIANTS(V)                                                                           \
  V(3, 8, 4) V(3, 4, 4) V(3, 2, 4) V(3, 5, 4) V(3, 5, 5) V(3, 1, 4) V(4, 1, 4) V(6, 1, 4)               \
  V(4, 4, 4) V(4, 2, 4) V(4, 5, 4) V(4, 5, 5) V(6, 2, 4)                                                \
  V(2, 8, 4) V(2, 4, 4) V(2, 2, 4) V(2, 5, 4) V(2, 5, 5) V(2, 1, 4)                                     \
  V(1, 8, 4) V(1, 4, 4) V(1, 2, 4) V(1, 5, 4) V(1, 5, 5) V(1, 1, 4)                                     \
  V(6, 2, 2) V(6, 4, 1) V(4, 2, 2) V(4, 4, 1) V(3, 2, 2) V(3, 4, 1)

static bool conv_has_variant(int mt, int nt, int waves) {
#define RTPE_V(MTv, NTv, Wv) if (mt == MTv && nt == NTv && waves == Wv) return true;
  RTPE_CONV_VARIANTS(RTPE_V)
#undef RTPE_V
  return false;
}

struct TileCand { int waves, nt, th, tw; };
static const TileCand kCands[] = {
    {4, 8, 16, 32}, {4, 8, 32, 16}, {4, 4, 16, 16}, {4, 4, 8, 32}, {4, 4, 32, 8},
    {4, 2, 8, 16},  {4, 2, 16, 8},  {5, 5, 20, 20}, {5, 5, 10, 40}, {5, 5, 40, 10},
    {4, 5, 16, 20}, {4, 5, 20, 16}, {4, 5, 8, 40},  {4, 5, 40, 8},
    {4, 1, 8, 8},                                    // stride-2 convs: a 17 x 17 halo tile, 2-3 workgroups per CU
    // the same 8 x 8 tile on 2 waves x 2 / 1 wave x 4 pixel tiles: every wave of a workgroup fetches ALL weight
    // fragments of the cout block through L1 (4 waves x MT KiB per k-step - 24 KiB for 96 MFMA cycles at MT = 6);
    // fewer waves with more pixel tiles each fetch them once / twice per 64 pixels
    {2, 2, 8, 8}, {1, 4, 8, 8},
};

static size_t tile_lds(const ConvPlan& p, int th, int tw, int waves, int nt, int mrun = 0) {
  const int hh = (th - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1;
  const size_t in_tile = (size_t)hh * conv_row_pitch(p, tw, 0);
  const size_t out_tile = (size_t)waves * nt * 16 * ((mrun ? (mrun > 0 ? mrun : -mrun) : p.mt) * 16 * p.esize + 16);   // epilogue transpose buffer
  return (size_t)kTapTableBytes + (in_tile > out_tile ? in_tile : out_tile);
}

// streaming kernel (conv_stream.hip): one workgroup per CU, 3-slot weight ring + 2-3 halo buffers
static bool stream_tile(const ConvPlan& p, const TileCand& c, int N, int H_pos, int W_pos, ConvTile* out,
                        bool allow_resident = true) {
  // stride 1: 4 or 5 pixel tiles per wave; stride 2 (4x the halo per output pixel): 2, and only with
  // resident weights (the halo buffers leave no room for the weight ring)
  if (!conv_stream_supports(p) || c.waves < 4) return false;
  // (96-cout workgroups, mt = 6: their 84 KiB of resident weights leave room for the 17 x 17 halo of an 8 x 8 tile only)
  if (p.in_mul == 1 ? (c.nt != 4 && c.nt != 5) : (c.nt != (p.mt == 6 ? 1 : 2) || p.n_cchunks != 1)) return false;
  if (p.mt == 6 && p.in_mul != 2) return false;
  const int hh = (c.th - 1) * p.in_mul + 3, hw = (c.tw - 1) * p.in_mul + 3;
  if (hw * 6 > 256) return false;                        // a halo row is at most 4 DMA instructions
  const size_t in_tile = (size_t)hh * conv_row_pitch(p, c.tw, 2);
  const size_t out_tile = (size_t)c.waves * c.nt * 16 * (p.mt * 32 + 16);
  size_t buf = in_tile > out_tile ? in_tile : out_tile;
  buf = (buf + 255) / 256 * 256;
  // weights resident (loaded once per workgroup) when all halves of a cout block fit beside 2 halo buffers
  int nw = 3, nb = 3;
  if (allow_resident && p.n_cchunks <= 2 && conv_stream_lds(p, (int)buf, 2, 2 * p.n_cchunks) <= 160 * 1024)
    nw = 2 * p.n_cchunks;
  if (conv_stream_lds(p, (int)buf, nb, nw) > 160 * 1024) nb = 2;
  if (conv_stream_lds(p, (int)buf, nb, nw) > 160 * 1024) return false;
  const long tiles = (long)((H_pos + c.th - 1) / c.th) * ((W_pos + c.tw - 1) / c.tw) * N;
  const long units = tiles * p.n_cb;
  static const int g_env = env_int("RTPE_PERSIST_G", 32);
  long G = g_env;                                         // one workgroup per CU
  const long need = (units + 7) / 8;
  if (G > need) G = need;
  G = (G + p.n_cb - 1) / p.n_cb * p.n_cb;
  memset(out, 0, sizeof(*out));
  out->nt = c.nt; out->waves = c.waves; out->th = c.th; out->tw = c.tw;
  out->kind = 2; out->grid = (int)(8 * G); out->buf_bytes = (int)buf; out->n_bufs = nb;
  out->n_wslots = nw;
  out->lds_bytes = conv_stream_lds(p, (int)buf, nb, nw);
  return true;
}

static ConvTile make_stream_tile(const ConvPlan& p, int N, int H_pos, int W_pos) {
  static const int force_nt = RTPE_DIAG_ENV_INT("RTPE_CONV_NT", 0);
  static const int force_waves = RTPE_DIAG_ENV_INT("RTPE_CONV_WAVES", 0);
  static const int force_th = RTPE_DIAG_ENV_INT("RTPE_CONV_TH", 0);
  double best_score = 1e30;
  ConvTile best;
  memset(&best, 0, sizeof(best));
  for (const TileCand& c : kCands) {
    ConvTile t;
    if ((force_nt && c.nt != force_nt) || (force_waves && c.waves != force_waves) || (force_th && c.th != force_th)) continue;
    if (!stream_tile(p, c, N, H_pos, W_pos, &t)) continue;
    const double waste = (double)((H_pos + c.th - 1) / c.th * c.th) * ((W_pos + c.tw - 1) / c.tw * c.tw) /
                         ((double)H_pos * W_pos);
    const long units = (long)((H_pos + c.th - 1) / c.th) * ((W_pos + c.tw - 1) / c.tw) * N * p.n_cb;
    const double rounds = (double)units / t.grid;
    const double imbalance = rounds >= 1.0 ? (double)((long)(rounds + 0.999)) / rounds : 1.0;
    const double halo = (double)((c.th - 1) * p.in_mul + 3) * ((c.tw - 1) * p.in_mul + 3) /
                        ((double)c.th * c.tw * p.in_mul * p.in_mul);
    // 5 MFMA waves leave one SIMD with two of them; resident weights save the per-stage weight stream
    const double score = waste * imbalance * (1.0 + 0.15 * (halo - 1.0)) * (t.n_bufs == 3 ? 1.0 : 1.05) *
                         (c.waves == 5 ? 1.3 : 1.0) * (t.n_wslots == 3 ? 1.0 : 0.8);
    if (score < best_score) { best_score = score; best = t; }
  }
  return best;
}

static bool direct_tile(const ConvPlan& p, ConvTile* out) {
  const int mb = conv_direct_mb(p);
  if (mb == 0 || get_option(kOptDirect1x1) == 0) return false;
  memset(out, 0, sizeof(*out));
  out->kind = 4; out->nt = 1; out->waves = 4; out->th = 8; out->tw = 8;   // (th x tw = 16 nt waves: the common check)
  out->grid = mb;
  return true;
}

static bool conv64_tile(const ConvPlan& p, int N, int H_pos, int W_pos, ConvTile* out) {
  if (!conv64_supports(p) || get_option(kOptConv64) == 0) return false;
  memset(out, 0, sizeof(*out));
  out->kind = 5; out->nt = 2; out->waves = 8; out->th = 16; out->tw = 16;      // 16 x 16 = 16 x 2 x 8: the common check
  out->grid = conv64_grid(N, H_pos, W_pos);
  out->lds_bytes = conv64_lds();
  return true;
}

ConvTile conv_make_tile(const ConvPlan& p, int N, int H_pos, int W_pos, bool allow_direct, bool allow_conv64) {
  if (allow_direct) {
    ConvTile t;
    if (direct_tile(p, &t)) return t;
  }
  if (allow_conv64) {
    ConvTile t;
    if (conv64_tile(p, N, H_pos, W_pos, &t)) return t;
  }
  static const int stream = getenv("RTPE_CONV_STREAM") ? atoi(getenv("RTPE_CONV_STREAM")) : 1;
  if ((stream & 1) && conv_stream_supports(p)) {
    ConvTile t = make_stream_tile(p, N, H_pos, W_pos);
    if (t.nt) return t;
  }
  static const long lds_cap = getenv("RTPE_CONV_LDS_CAP") ? atol(getenv("RTPE_CONV_LDS_CAP")) : 80 * 1024;
  static const int force_nt = RTPE_DIAG_ENV_INT("RTPE_CONV_NT", 0);
  static const int force_waves = RTPE_DIAG_ENV_INT("RTPE_CONV_WAVES", 0);
  double best_score = 1e30;
  ConvTile best;
  memset(&best, 0, sizeof(best));
  for (const TileCand& c : kCands) {
    if (p.mt == 4 && c.nt == 8) continue;  // 128 accumulators + operands: keep 2 waves/SIMD
    if (p.mt == 6 && c.nt > 2) continue;   // the 96-cout variant exists for 1 and 2 pixel tiles per wave
    if (c.nt == 1 || c.waves < 4) continue;  // 8 x 8 tiles are autotuning candidates of stride-2 convs only
    if (force_nt && c.nt != force_nt) continue;
    if (force_waves && c.waves != force_waves) continue;
    const size_t lds = tile_lds(p, c.th, c.tw, c.waves, c.nt);
    if (lds > 160 * 1024 || ((long)lds > lds_cap && !(c.nt == 2))) continue;
    const long tiles = (long)((H_pos + c.th - 1) / c.th) * ((W_pos + c.tw - 1) / c.tw);
    const double waste = (double)tiles * c.th * c.tw / ((double)H_pos * W_pos);
    const long wgs = tiles * N * p.n_cb;
    // cost model: padded work, penalise grids that cannot fill 256 CUs twice,
    // reward pixel-tile reuse of each weight fragment (nt) and small halos
    double score = waste;
    if (wgs < 512) score *= 1.0 + 0.25 * (512.0 - wgs) / 512.0;
    score *= 1.0 + 0.4 / c.nt;
    const double halo = (double)((c.th - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1) *
                        ((c.tw - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1) / ((double)c.th * c.tw * p.in_mul * p.in_mul);
    score *= 1.0 + 0.1 * (halo - 1.0);
    if (score < best_score) {
      best_score = score;
      best.nt = c.nt; best.waves = c.waves; best.th = c.th; best.tw = c.tw; best.lds_bytes = lds;
    }
  }
  // RTPE_CONV_WAVE_HALVES=2: untuned launches of the stride-2 3x3 convs put the two halves of their cout block on the two halves
  // of the workgroup's waves (mrun < 0; the autotuner times that shape by itself: conv_enum_tiles; the switch lets a test run it
  // everywhere)
  static const int force_halves = env_int("RTPE_CONV_WAVE_HALVES", 1);
  if (force_halves == 2 && p.esize == 2 && p.in_mul == 2 && p.tapw == 3 && p.mt % 2 == 0 && conv_has_variant(p.mt / 2, 4, 4)) {
    ConvTile h;
    memset(&h, 0, sizeof(h));
    h.nt = 4; h.waves = 4; h.th = 8; h.tw = 16; h.mrun = -(p.mt / 2);
    h.lds_bytes = tile_lds(p, h.th, h.tw, 4, 4, h.mrun);
    if (h.lds_bytes <= 160 * 1024) return h;
  }
  // RTPE_CONV_MRUN=m: untuned launches share every cout block out to mt / m workgroups where that shape exists (the
  // autotuner times these shapes for small grids by itself: conv_enum_tiles; the switch lets a test run them everywhere)
  static const int force_mrun = env_int("RTPE_CONV_MRUN", 0);
  if (force_mrun > 0 && best.nt && force_mrun < p.mt && p.mt % force_mrun == 0 && conv_has_variant(force_mrun, best.nt, best.waves)) {
    best.mrun = force_mrun;
    best.lds_bytes = tile_lds(p, best.th, best.tw, best.waves, best.nt, force_mrun);
  }
  return best;
}

void conv_fill_args(const ConvGeom& g, const ConvPlan& p, const ConvTile& t, ConvArgs* a) {
  a->cin = g.cin; a->cout = g.cout;
  a->tapw = p.tapw; a->ntaps = p.tapw * p.tapw;
  a->lo_y = p.lo_y; a->lo_x = p.lo_x;
  a->in_mul = p.in_mul;
  a->cc = p.cc; a->n_cchunks = p.n_cchunks; a->kc = p.kc; a->pstride = p.pstride;
  a->th = t.th; a->tw = t.tw;
  a->dil = p.dil;
  a->halo_h = (t.th - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1;
  a->halo_w = (t.tw - 1) * p.in_mul + (p.tapw - 1) * p.dil + 1;
  a->rowb = conv_row_pitch(p, t.tw, t.kind);
  a->deint = conv_deint(p, t.kind) ? 1 : 0;
  a->n_even = (a->halo_w + 1) >> 1;
  a->div_rowb16 = make_fastdiv((uint32_t)a->rowb >> 4);
  a->div_ps16 = make_fastdiv((uint32_t)p.pstride >> 4);
  {   // buffer bounds of one image of the input view for the LDS-DMA staging (the caller has set N, H_in, W_in, in_ld)
    const int eps = 16 / p.esize;
    a->img_bytes = (((size_t)a->H_in * a->W_in - 1) * a->in_ld + (size_t)((g.cin + eps - 1) / eps * eps)) * p.esize;
    a->dma_stage = get_option(kOptTileDma) != 0 && a->img_bytes < 0x80000000ull ? 1 : 0;
  }
  a->tiles_x = (a->W_pos + t.tw - 1) / t.tw;
  a->tiles_y = (a->H_pos + t.th - 1) / t.th;
  a->div_tw = make_fastdiv(t.tw);
  a->div_slots = make_fastdiv(p.cc / (16 / p.esize));
  a->div_rowslots = make_fastdiv(a->halo_w * (p.cc / (16 / p.esize)));
  a->div_cc = make_fastdiv(p.cc);
  a->div_tiles_x = make_fastdiv(a->tiles_x);
  a->div_tiles_xy = make_fastdiv(a->tiles_x * a->tiles_y);
  a->n_cb = p.n_cb;
  a->m_split = (t.kind == 0 && t.mrun != 0) ? p.mt / (t.mrun > 0 ? t.mrun : -t.mrun) : 1;
  a->wsplit = (t.kind == 0 && t.mrun < 0) ? 2 : 0;
  if (a->in_cs == 0) a->in_cs = p.cc;
  if (a->out_cs == 0) a->out_cs = p.mt * 16;
  if (a->res_cs == 0) a->res_cs = p.mt * 16;
  a->buf_bytes = t.buf_bytes;
  a->n_bufs = t.n_bufs;
  a->n_wslots = t.n_wslots;
  static const int abl = RTPE_DIAG_ENV_INT("RTPE_STREAM_ABL", 0);
  a->ablate = abl;
  // ablations for profiling only (-DRTPE_DIAG builds): RTPE_CONV_SKIPK=1 runs the data movement without the k-loops
  static const int skipk = RTPE_DIAG_ENV_INT("RTPE_CONV_SKIPK", 0);
  if (skipk) a->kc = 0;
}

template <typename T, int MT, int NT, int WAVES>
static int launch_variant(const ConvTile& t, const ConvArgs& a, int n_cb, hipStream_t s) {
  static unsigned long long attr_mask = 0;
  auto kern = conv_mfma_kernel<T, MT, NT, WAVES>;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const unsigned n_tiles = (unsigned)(a.N * a.tiles_x * a.tiles_y);
  dim3 grid(((n_tiles + 7u) / 8u) * 8u * (unsigned)n_cb * (unsigned)((a.m_split > 1 ? a.m_split : 1) / (a.wsplit > 1 ? 2 : 1)) *
            (unsigned)(a.n_cls > 0 ? a.n_cls : 1));   // (tile / 8, class, cout block x share, tile % 8 = XCD)
  hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), t.lds_bytes, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// every launch shape worth timing for a layer (the autotuner picks the fastest; all of
// them give bit-identical results because the k order does not depend on the tiling)
void conv_enum_tiles(const ConvPlan& p, int N, int H_pos, int W_pos, std::vector<ConvTile>* out) {
  out->clear();
  double min_waste = 1e30;
  // a grid of fewer than 512 workgroups (batch 1; the /16 and /32 branches of small batches) leaves compute units idle
  // while every workgroup walks the whole k chain of its 48-96 output channels: 8 x 8 pixel tiles and cout blocks
  // shared out to several workgroups (ConvTile::mrun) are timed as well
  auto small_grid = [&](const TileCand& c) {
    return (long)N * p.n_cb * ((H_pos + c.th - 1) / c.th) * ((W_pos + c.tw - 1) / c.tw) < 512;
  };
  auto tiny_ok = [&](const TileCand& c) {
    return !(c.nt == 1 || c.waves < 4) || (p.in_mul == 2 && p.mt >= 3) || (small_grid(c) && c.waves == 4);
  };
  for (const TileCand& c : kCands) {
    if (!tiny_ok(c)) continue;
    const double w = (double)((H_pos + c.th - 1) / c.th * c.th) * ((W_pos + c.tw - 1) / c.tw * c.tw) /
                     ((double)H_pos * W_pos);
    if (w < min_waste) min_waste = w;
  }
  for (const TileCand& c : kCands) {
    // the block's two halves on the two halves of the workgroup's waves (mrun < 0): same tile, twice the pixels per wave, half
    // the weight fragments per wave - the stride-2 convs with two and more channel chunks, whose weights come from L2 in every
    // k step
    static const int wave_halves = env_int("RTPE_CONV_WAVE_HALVES", 1);
    if (wave_halves && p.esize == 2 && p.in_mul == 2 && p.tapw == 3 && p.mt % 2 == 0 && c.waves == 4 && c.nt >= 4 && c.nt % 2 == 0 &&
        conv_has_variant(p.mt / 2, c.nt, 4)) {
      // candidate c describes (waves, nt, th, tw) with th * tw = 16 nt waves: the same waves and nt on HALF the pixels
      ConvTile h;
      memset(&h, 0, sizeof(h));
      h.nt = c.nt; h.waves = 4; h.mrun = -(p.mt / 2);
      h.th = c.th >= c.tw ? c.th / 2 : c.th; h.tw = c.th >= c.tw ? c.tw : c.tw / 2;
      h.lds_bytes = tile_lds(p, h.th, h.tw, 4, c.nt, h.mrun);
      const double hw = (double)((H_pos + h.th - 1) / h.th * h.th) * ((W_pos + h.tw - 1) / h.tw * h.tw) / ((double)H_pos * W_pos);
      bool dup = false;
      for (const ConvTile& o : *out) dup = dup || (o.kind == 0 && o.mrun == h.mrun && o.nt == h.nt && o.th == h.th && o.tw == h.tw);
      if (!dup && h.lds_bytes <= 160 * 1024 && hw <= 1.35 * min_waste) out->push_back(h);
    }
    if (p.mt == 4 && c.nt == 8) continue;
    if (p.mt == 6 && c.nt > 2 && c.waves >= 4) continue;
    if (!tiny_ok(c)) continue;
    const double waste = (double)((H_pos + c.th - 1) / c.th * c.th) * ((W_pos + c.tw - 1) / c.tw * c.tw) /
                         ((double)H_pos * W_pos);
    // (small maps - 20 x 20 at /32 - fit one 20 x 20 tile per image: 32 workgroups for 256 CUs.  Smaller tiles
    // that pad the map by up to 2.6x are timed as well when the tight tiling cannot fill the chip.)
    const long wgs_tight = (long)N * p.n_cb * (long)(min_waste * H_pos * W_pos / 400.0 + 0.5);   // ~ workgroups of a 400-px tiling
    if (waste > (wgs_tight < 256 ? 2.6 : 1.35) * min_waste) continue;
    ConvTile t;
    memset(&t, 0, sizeof(t));
    t.nt = c.nt; t.waves = c.waves; t.th = c.th; t.tw = c.tw;
    t.lds_bytes = tile_lds(p, c.th, c.tw, c.waves, c.nt);
    if (t.lds_bytes <= 160 * 1024 && conv_has_variant(p.mt, c.nt, c.waves)) out->push_back(t);   // one workgroup per tile
    if (small_grid(c))
      for (int m = p.mt / 2; m >= 1; --m) {
        if (p.mt % m != 0 || !conv_has_variant(m, c.nt, c.waves)) continue;
        t.mrun = m;
        t.lds_bytes = tile_lds(p, c.th, c.tw, c.waves, c.nt, m);
        if (t.lds_bytes <= 160 * 1024) out->push_back(t);
      }
    static const int stream = getenv("RTPE_CONV_STREAM") ? atoi(getenv("RTPE_CONV_STREAM")) : 1;
    if (stream & 1) {
      ConvTile st;
      if (stream_tile(p, c, N, H_pos, W_pos, &st)) out->push_back(st);   // streaming, LDS-DMA operands
      // two-chunk layers: resident weights leave room for 2 halo buffers, the weight ring for 3
      if (p.n_cchunks == 2 && st.nt && st.n_wslots != 3 && stream_tile(p, c, N, H_pos, W_pos, &st, false) &&
          st.n_bufs == 3)
        out->push_back(st);
    }
  }
  {
    ConvTile dt;
    if (direct_tile(p, &dt)) out->push_back(dt);                 // 1x1: no staged tile at all (conv_direct.hip)
    if (conv64_tile(p, N, H_pos, W_pos, &dt)) out->push_back(dt);   // 64 -> 64 3x3: persistent, pipelined (conv64.hip)
  }
}

int conv_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(t.th * t.tw * (t.kind == 0 && t.mrun < 0 ? 2 : 1) == 16 * t.nt * t.waves,
               "conv tile %dx%d != 16*%d*%d", t.th, t.tw, t.nt, t.waves);
  RTPE_REQUIRE(t.lds_bytes <= 160 * 1024, "conv tile needs %zu B of LDS", t.lds_bytes);
  // a launch shape belongs to one plan (halo = taps x dilation, pixel stride): never run a foreign one
  RTPE_REQUIRE(t.kind != 0 || t.lds_bytes >= tile_lds(p, t.th, t.tw, t.waves, t.nt, t.mrun),
               "conv tile (%zu B of LDS) was made for another plan (needs %zu B)", t.lds_bytes,
               tile_lds(p, t.th, t.tw, t.waves, t.nt, t.mrun));
  RTPE_REQUIRE(a.kc * 4 * sizeof(int) <= (size_t)kTapTableBytes, "k chunk table overflow (kc=%d)", a.kc);
  const int eps = 16 / p.esize;
  RTPE_REQUIRE(a.in_ld % eps == 0 && (a.y == nullptr || (a.out_ld % eps == 0 && a.cout_store % eps == 0 &&
                                                         ((uintptr_t)a.y & 15) == 0)),
               "conv: NHWC views must be 16-byte aligned with channel counts that are multiples of %d "
               "(in_ld=%d out_ld=%d cout_store=%d y=%p)", eps, a.in_ld, a.out_ld, a.cout_store, (void*)a.y);
  RTPE_REQUIRE(a.res == nullptr || (a.res_ld % eps == 0 && ((uintptr_t)a.res & 15) == 0), "conv: residual view alignment");
  RTPE_REQUIRE(((uintptr_t)a.x & 15) == 0, "conv: input view must be 16-byte aligned");
  if (t.kind == 2) return conv_stream_launch(p, t, a, s);
  if (t.kind == 4) return conv_direct_launch(p, a, s);
  if (t.kind == 5) return conv64_launch(p, t, a, s);
  RTPE_REQUIRE(a.in_cs == p.cc && a.out_cs == p.mt * 16 && a.res_cs == p.mt * 16,
               "conv: the one-workgroup-per-tile kernel reads and writes NHWC only");
  const int mrun = t.mrun > 0 ? t.mrun : t.mrun < 0 ? -t.mrun : p.mt;
  RTPE_REQUIRE(t.mrun >= 0 ? a.wsplit <= 1 : (a.wsplit == 2 && t.waves % 2 == 0 && (p.mt / mrun) % 2 == 0), "conv: wave halves");
  RTPE_REQUIRE(p.mt % mrun == 0 && a.m_split == p.mt / mrun, "conv: %d cout tiles per workgroup do not divide the block of %d",
               mrun, p.mt);
#define RTPE_V(MTv, NTv, Wv)                                                                  \
  if (mrun == MTv && t.nt == NTv && t.waves == Wv)                                            \
    return p.esize == 4 ? launch_variant<float, MTv, NTv, Wv>(t, a, p.n_cb, s)                 \
                        : launch_variant<_Float16, MTv, NTv, Wv>(t, a, p.n_cb, s);
  RTPE_CONV_VARIANTS(RTPE_V)
#undef RTPE_V
  set_error("conv: no kernel variant mt=%d nt=%d waves=%d", mrun, t.nt, t.waves);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
