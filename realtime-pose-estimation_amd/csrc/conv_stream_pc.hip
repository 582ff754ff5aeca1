// Streaming 3x3 convolution, producer / consumer form ("pc", ConvTile::kind == 3), for the BasicBlock convs of the
// 96 / 192 / 384-channel branches (pose_higher_hrnet.py:46-75 of the reference; 144 launches = 40 % of the forward).
// Same math, same k order, same rounding points and the same packed weights as conv_stream.hip / conv_mfma.hip: the
// results are bit-identical; what changes is what runs beside what.
//
// Why.  conv_stream_kernel's four MFMA waves walk through a unit together: k loops (both channel chunks), then BN,
// the transposition through LDS, residual add, ReLU and the row stores - and the stores and residual requests block at
// issue behind the halo-tile requests in the CU's one in-order memory pipeline (profiles/r03_stream_store_phase.txt: a
// unit at C = 96 is 8,800 cycles of k loops + 8,400 cycles of epilogue).  While the k loops run the chip's HBM is
// mostly idle, while the epilogues run the matrix pipes are (profiles/r04_mempipe_probe.txt: a CU alone pulls 19-24
// B/clk from HBM, 9.5 when all 256 ask at once - the epilogue phases of all workgroups coincide).
//
// Here a workgroup has EIGHT waves, two per SIMD, in two groups (waves 0-3, 4-7) that alternate over the
// workgroup's units: while group A multiplies unit u, group B finishes unit u - 1 (BN, residual, ReLU, stores) and
// requests the operands of the stages to come.  No loader waves: the group that is not multiplying issues the LDS-DMA
// requests (halo tile of stage s + 1, weight halves of the ring) right behind the barrier that frees their buffers.
//   * the stage sequence - and with it the LDS budget: weights resident or a 3-slot ring, two halo buffers - is the
//     one of conv_stream_kernel; consecutive units go to alternating groups;
//   * NO transposition slab: the accumulator layout of v_mfma_f32_16x16x32_f16 (lane (r, g): 4 output channels
//     16 m + 4 g .. + 3 of pixel r) becomes 16-byte row pieces in registers with v_permlane16_swap_b32: swapping the
//     odd 16-lane rows of cout tile m with the even rows of tile m + 1 leaves 8 consecutive channels of a pixel in
//     every lane (the four lanes of a pixel hold channels 0-31: 64 contiguous bytes per store).  The third cout tile
//     of the 48-channel block pairs its pixel tiles with each other.  The LDS holds operands only, nothing waits for
//     a tile buffer to double as a slab;
//   * one workgroup barrier per half stage (per stage with resident weights), shared by both groups; every wait for
//     memory is a counted s_waitcnt at the end of an interval: what was requested before the interval began has landed;
//   * residual rows: buffer loads into the fixed register window v[224:255] (as conv_stream_kernel), requested at the
//     start of the finishing group's first interval; stores and loads are branch-free buffer instructions (a lane
//     outside the tensor gets an out-of-range offset), so the number of vector-memory operations of a wave is known.
#include "conv_stream_dev.h"

namespace rtpe {

namespace {

constexpr int kPcResReg0 = 224;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kOor = 0x80000000u;               // buffer offset no tensor reaches: loads return 0, stores are dropped

// residual piece K of the finishing unit -> v[R0:R3]; the add names the registers again (see conv_stream.hip)
#define RTPE_PC_RES_LOAD(K, R0, R1, R2, R3)                                                                  \
  if (it == K)                                                                                               \
    asm volatile("buffer_load_dwordx4 v[" #R0 ":" #R3 "], %0, %1, %2 offen" ::"v"(vo), "s"(rsrd), "s"(rsoff) \
                 : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3);
#define RTPE_PC_RES_ADD(K, R0, R1, R2, R3)                                                                   \
  if (it == K)                                                                                               \
    asm volatile("v_pk_add_f16 %0, %0, v" #R0 "\n\tv_pk_add_f16 %1, %1, v" #R1 "\n\tv_pk_add_f16 %2, %2, v" #R2 \
                 "\n\tv_pk_add_f16 %3, %3, v" #R3                                                            \
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#define RTPE_PC_RES_ALL(X)                                                                           \
  X(0, 224, 225, 226, 227) X(1, 228, 229, 230, 231) X(2, 232, 233, 234, 235) X(3, 236, 237, 238, 239) \
  X(4, 240, 241, 242, 243) X(5, 244, 245, 246, 247) X(6, 248, 249, 250, 251) X(7, 252, 253, 254, 255)

// 16 bytes per lane to descriptor + per-lane byte offset + scalar byte offset, write-through (rtpe_common.h
// store16_wt).  s_nop 4 in front: the descriptor / scalar offset may just have been restored from a spill lane by
// v_readlane_b32, which a vector-memory instruction must not read for 5 wait states (the hazard recognizer does not see
// asm text); s_nop 1 behind: a store of more than 8 bytes reads its data registers for two more wait states.
__device__ __forceinline__ void pc_store16(int4v v, uint32_t voff, sgpr4 srd, int soff, bool write_through) {
  if (write_through)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
  else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

}  // namespace

template <int NT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(kPcResReg0)))
conv_stream_pc_kernel(const ConvArgs a) {
  constexpr int MT = 3;
  constexpr int NP = NT + NT / 2 + (NT & 1);            // 16-byte row pieces per lane and unit
  static_assert(NP <= 8, "residual register window");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WSLOT = MT * kKH * 1024;                // weight fragments of one half stage
  char* const wring = smem;
  const int NWS = a.n_wslots;
  char* const tiles = smem + NWS * WSLOT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wv >> 2, wi = wv & 3;                 // group (0 / 1), wave of the group

  Units um;
  um.init(a.N * a.tiles_x * a.tiles_y, a.n_cb);
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  const int ncc = a.n_cchunks;                          // power of two (host-checked)
  const int sh = __builtin_ctz((unsigned)ncc);
  const int n_units = um.count;
  const int S = n_units << sh;                          // stages of this workgroup
  const bool resident = NWS == 2 * ncc;                 // a workgroup keeps its cout block
  if (S == 0) return;
  int cb0;
  {
    int tile0;
    um.get(0, &tile0, &cb0);
  }

  // ------------------------------------------------ operand requests ------------------------------------------------
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.w), 0, a.n_cb * ncc * kKC * MT * 1024, 0x00020000);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const int lane16 = lane * 16;
  // weight half q = 2 s + h (s: stage): pieces part, part + nparts, ... of its MT * 7; returns how many were issued
  auto issue_w = [&](int q, int part, int nparts) {
    const int s = q >> 1, h = q & 1;
    const int cci = s & (ncc - 1);
    const int src = ((cb0 * ncc + cci) * kKC + h * kKH) * MT * 1024;
    char* dst = wring + (resident ? 2 * cci + h : q % 3) * WSLOT;
    int n = 0;
    for (int p = part; p < MT * kKH; p += nparts, ++n)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + p * 1024), 16, lane16, src + p * 1024, 0, 0);
    return n;
  };
  const int rowslots = a.halo_w * kSlots;
  const int krow = (rowslots + 63) >> 6;                // DMA instructions per halo row (<= 4, host-checked)
  // halo rows part * hh / nparts .. of the tile of stage s -> buffer s & 1; returns the instructions issued
  // (half / nhalves: the first or second half of those rows, or all of them)
  auto issue_tile = [&](int s, int part, int nparts, int half, int nhalves) {
    int tile, cb;
    um.get(s >> sh, &tile, &cb);
    const int chunk = s & (ncc - 1);
    const int cbase = chunk * kCC;
    uint32_t t = (uint32_t)tile;
    const uint32_t n = fdiv(t, a.div_tiles_xy);
    t -= n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    const uint32_t txi = t - tyi * a.tiles_x;
    const int iy0 = (int)tyi * a.th + a.lo_y, ix0 = (int)txi * a.tw + a.lo_x;
    uint32_t voff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qq = k * 64 + lane;
      const int hx = qq / kSlots, sl = qq - hx * kSlots;
      const int ix = ix0 + hx;
      const bool ok = (unsigned)ix < (unsigned)a.W_in && cbase + sl * 8 < a.cin;
      voff[k] = ok ? (uint32_t)(ix * a.in_ld + sl * 8) * 2u : kOor;
    }
    char* const buf = tiles + (s & 1) * a.buf_bytes;
    const int soff_row = a.W_in * a.in_ld * 2;
    const int soff0 = (((int)n * a.H_in + iy0) * a.W_in * a.in_ld + (int)(chunk * a.in_cs)) * 2;
    const int row_lo = iy0 < 0 ? -iy0 : 0, row_hi = a.H_in - iy0;
    const int ra = a.halo_h * part / nparts, rb = a.halo_h * (part + 1) / nparts;
    const int r0 = ra + (rb - ra) * half / nhalves, r1 = ra + (rb - ra) * (half + 1) / nhalves;
    for (int r = r0; r < r1; ++r) {
      const bool row_ok = r >= row_lo && r < row_hi;
      const int soff = row_ok ? soff0 + r * soff_row : 0;
      char* dst = buf + r * a.rowb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < krow) {
          if (k * 64 + lane < rowslots)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + k * 1024), 16,
                                                     (int)(row_ok ? voff[k] : kOor), soff, 0, 0);
        }
      }
    }
    return (r1 - r0) * krow;
  };

  // ------------------------------------------------ k loop (as conv_stream_kernel) ------------------------------------------------
  const int r = lane & 15;
  const int g = lane >> 4;
  int toff[kKC];                                        // LDS byte offset of this lane group's 8 channels in k-step k
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
    const uint32_t p = (wi * NT + nt) * 16 + r;
    const uint32_t oy = fdiv(p, a.div_tw);
    const uint32_t ox = p - oy * a.tw;
    pixbase[nt] = (int)(oy * a.rowb + ox * kPStride);
  }
  float4v acc[MT][NT];
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
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                       // VALU (address)
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                       // DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ------------------------------------------------ epilogue state ------------------------------------------------
  // BN / bias parameters of the workgroup's cout block: 2 x 48 floats behind the tile buffers (16 bytes per lane group
  // g and cout tile m, read back at the start of an epilogue: 24 registers that the k loops do not carry)
  float* const bn_lds = reinterpret_cast<float*>(tiles + 2 * a.buf_bytes);
  if (tid < 2 * MT * 16) {
    const int c = tid < MT * 16 ? tid : tid - MT * 16;
    bn_lds[tid] = (tid < MT * 16 ? a.alpha : a.beta)[cb0 * MT * 16 + c];
  }
  // Row piece `it` of this lane, fixed for the kernel.  Pieces 0 .. NT-1: cout tiles 0 and 1 of pixel tile `it` (the
  // lane holds channels (g & 1) * 16 + (g >> 1) * 8 .. + 7 of pixel r); pieces NT .. : cout tile 2 of the pixel tiles
  // 2 j (even rows g) and 2 j + 1 (odd rows g), channels 32 + (g >> 1) * 8 .. + 7; with an odd NT the last pixel
  // tile pairs with itself and only the even rows store.  eyx: (row << 16 | column) of the piece's pixel in the tile
  // (a row no tile has: the lane has no such piece); the byte offsets follow from it per unit (two 24-bit multiply-adds)
  int eyx[NP];
  const int cblk = cb0 * MT * 16;
  const uint32_t row_pix = (uint32_t)a.W_full;
  const int ch_a = (g & 1) * 16 + (g >> 1) * 8, ch_b = 32 + (g >> 1) * 8;
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    int ntl;
    bool valid = true;
    if (it < NT) ntl = it;
    else if (it < NT + NT / 2) ntl = 2 * (it - NT) + (g & 1);
    else { ntl = NT - 1; valid = !(g & 1); }
    const uint32_t p = (uint32_t)((wi * NT + ntl) * 16 + r);
    const uint32_t oyt = fdiv(p, a.div_tw);
    const uint32_t oxt = p - oyt * a.tw;
    valid = valid && cblk + (it < NT ? ch_a : ch_b) < a.cout_store;
    eyx[it] = valid ? (int)((oyt << 16) | oxt) : 0x7fff0000;
  }
  // byte offset of piece `it` from the unit's first pixel in a view with rows of `ld` elements; kOor outside the image
  auto piece_off = [&](int it, uint32_t ld, int hy, int hx) __attribute__((always_inline)) {
    int e = eyx[it];
    asm volatile("" : "+v"(e));                            // lane-only math stays inside the unit loop
    const uint32_t pix = __umul24((uint32_t)e >> 16, row_pix) + ((uint32_t)e & 0xffffu);
    const uint32_t off = (__umul24(pix, ld) + (uint32_t)(it < NT ? ch_a : ch_b)) * 2u;
    return ((e >> 16) < hy && (e & 0xffff) < hx) ? off : kOor;
  };
  const bool do_store = a.y != nullptr && !(a.ablate & 2);
  const bool use_res = a.res != nullptr && do_store;
  const sgpr4 ysrd = make_srd(a.y, 0x7fffffffu);
  const sgpr4 rsrd = make_srd(use_res ? (const void*)a.res : (const void*)a.y, 0x7fffffffu);

  struct UnitS { int ysoff, rsoff, hy, hx; };
  auto unit_scalars = [&](int u) {
    int tile, cb;
    um.get(u, &tile, &cb);
    uint32_t t = (uint32_t)tile;
    const uint32_t n = fdiv(t, a.div_tiles_xy);
    t -= n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    const uint32_t txi = t - tyi * a.tiles_x;
    const int py0 = (int)tyi * a.th, px0 = (int)txi * a.tw;
    UnitS q;
    q.hy = a.H_pos - py0;
    q.hx = a.W_pos - px0;
    const size_t pix0 = ((size_t)n * a.H_full + py0) * a.W_full + px0;
    q.ysoff = (int)((pix0 * a.out_ld + (size_t)cb0 * a.out_cs) * 2);
    q.rsoff = use_res ? (int)((pix0 * a.res_ld + (size_t)cb0 * a.res_cs) * 2) : 0;
    return q;
  };

  int pend = -1;                                         // the unit whose sums wait in `acc` (this group)
  const bool wt = RTPE_WT_STORES && !(a.pc_flags & 4);   // write-through row stores (rtpe_common.h store16_wt)

  // residual rows of unit u -> the register window; NP buffer loads, nobody waits here
  auto load_res = [&](int u) {
    const UnitS q = unit_scalars(u);
    const int rsoff = q.rsoff;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const uint32_t vo = piece_off(it, (uint32_t)a.res_ld, q.hy, q.hx);
      if (it == 0) asm volatile("s_nop 4" ::: "memory");
      RTPE_PC_RES_ALL(RTPE_PC_RES_LOAD)
    }
  };
  // The whole epilogue of the pending unit: BN (+ residual) (+ ReLU) on the accumulators, row pieces by lane-row swaps,
  // stores.  `younger`: vector-memory operations this wave has issued behind the unit's residual loads.  Returns the
  // stores issued.
  auto epilogue = [&](int younger) {
    const UnitS pus = unit_scalars(pend);
    pend = -1;
    float4v al[MT], be[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      al[m] = *reinterpret_cast<const float4v*>(bn_lds + m * 16 + g * 4);
      be[m] = *reinterpret_cast<const float4v*>(bn_lds + MT * 16 + m * 16 + g * 4);
    }
    auto bn = [&](const float4v v, int m) __attribute__((always_inline)) {
      half4 o;
      if (a.round_conv) {                                // the conv output is an fp16 tensor
        o = bn_round(v, al[m], be[m]);
      } else {
        float2v lo{v[0], v[1]}, hi{v[2], v[3]};
        lo = __builtin_elementwise_fma(lo, float2v{al[m][0], al[m][1]}, float2v{be[m][0], be[m][1]});
        hi = __builtin_elementwise_fma(hi, float2v{al[m][2], al[m][3]}, float2v{be[m][2], be[m][3]});
        const half2v olo = __builtin_convertvector(lo, half2v), ohi = __builtin_convertvector(hi, half2v);
        o = half4{olo[0], olo[1], ohi[0], ohi[1]};
      }
      return __builtin_bit_cast(u32x2, o);
    };
    // x: registers of the tile whose even rows keep their own data, y: of the tile whose odd rows do
    auto swap_pair = [&](const u32x2 x, const u32x2 y) __attribute__((always_inline)) {
      const auto s0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
      return int4v{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
    };
    int4v piece[NP];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) piece[nt] = swap_pair(bn(acc[0][nt], 0), bn(acc[1][nt], 1));
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) piece[NT + j] = swap_pair(bn(acc[2][2 * j], 2), bn(acc[2][2 * j + 1], 2));
    if (NT & 1) {
      const u32x2 o = bn(acc[2][NT - 1], 2);
      piece[NP - 1] = swap_pair(o, o);
    }
    if (use_res) wait_vmcnt(younger);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      int4v w = piece[it];
      if (use_res) {                                     // fp16 add, round-to-nearest-even = the wrapper's add
        RTPE_PC_RES_ALL(RTPE_PC_RES_ADD)
      }
      if (a.relu) {                                      // x > 0 ? x : +0, on the sign bits
        short8 b = __builtin_bit_cast(short8, w);
        b = b & ~(b >> 15);
        w = __builtin_bit_cast(int4v, b);
      }
      if (do_store) pc_store16(w, piece_off(it, (uint32_t)a.out_ld, pus.hy, pus.hx), ysrd, pus.ysoff, wt);
    }
    return do_store ? NP : 0;
  };

#ifdef RTPE_CONV_STAMPS
  unsigned long long st[16] = {0}, c0, c1, c2, c3, c4;
  const unsigned long long k_begin = __builtin_readcyclecounter();
#endif

  // ------------------------------------------------ prologue: stage 0's operands, by all eight waves ------------------------------------------------
  if (!(a.ablate & 4)) issue_tile(0, wv, 8, 0, 1);
  if (resident) {
    for (int q = 0; q < 2 * ncc; ++q) issue_w(q, wv, 8);
  } else {
    issue_w(0, wv, 8);
    if (2 * S > 1) issue_w(1, wv, 8);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: the BN parameters written to LDS above)
  const bool split = !resident && (a.pc_flags & 2);     // the tile of stage s + 1 is requested in two halves, around H(s)
#ifdef RTPE_CONV_STAMPS
  SSTAMP(c0);
  st[12] = c0 - k_begin;
#endif

  for (int u = 0; u < n_units; ++u) {
    if ((u & 1) == grp) {
      // ===================================== multiplying group =====================================
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
      for (int cci = 0; cci < ncc; ++cci) {
        const int s = (u << sh) + cci;
        const char* tilebuf = tiles + (s & 1) * a.buf_bytes;
        const int w0 = resident ? 2 * cci : (2 * s) % 3;
        const int w1 = resident ? 2 * cci + 1 : (2 * s + 1) % 3;
        SSTAMP(c0);
        RTPE_SBARRIER();                                   // M(s): tile s and weight half 2 s are in LDS
        SSTAMP(c1);
        if (!(a.ablate & 1)) half_stage(wring + w0 * WSLOT, tilebuf, std::integral_constant<int, 0>());
        SSTAMP(c2);
        if (!resident) {
          // (the weight pieces this wave requested in its last interval as the finishing group)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          RTPE_SBARRIER();                                 // H(s): weight half 2 s + 1 is in LDS
        }
        SSTAMP(c3);
        if (!(a.ablate & 1)) half_stage(wring + w1 * WSLOT, tilebuf, std::integral_constant<int, 1>());
        SSTAMP(c4);
#ifdef RTPE_CONV_STAMPS
        st[0] += c1 - c0; st[1] += c2 - c1; st[2] += c3 - c2; st[3] += c4 - c3; st[5] += 1;
#endif
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pend = u;
    } else {
      // ===================================== finishing / requesting group =====================================
      for (int cci = 0; cci < ncc; ++cci) {
        const int s = (u << sh) + cci;
        SSTAMP(c0);
        RTPE_SBARRIER();                                   // M(s): buffer (s + 1) & 1 and ring slot (2 s + 2) % 3 are free
        SSTAMP(c1);
        int n_iv = 0, n_t = 0;
        const bool more = s + 1 < S && !(a.ablate & 4);
        if (more) n_t = issue_tile(s + 1, wi, 4, 0, split ? 2 : 1);
        n_iv = n_t;
        if (!resident && 2 * s + 2 < 2 * S) n_iv += issue_w(2 * s + 2, wi, 4);
        if (resident) n_iv = 0;                            // (the tile must have landed at the next barrier: count what follows it)
        SSTAMP(c2);
        if (cci == 0) {
          // finish this group's previous unit
          // (the residual rows are requested HERE, behind the tile of the next stage, not a unit ahead: requested early they
          // sit in the CU's in-order memory pipeline in front of the next tile requests - 44.6 -> 49.6 us at C = 96)
          if (pend >= 0) {
            if (use_res) { load_res(pend); n_iv = 0; }     // (the epilogue waits for them: everything before has landed too)
            n_iv += epilogue(0);
          }
        }
        SSTAMP(c3);
        if (!resident) {
          wait_vmcnt(n_iv);                                // what was requested before M(s) has landed (weight half 2 s + 1)
          RTPE_SBARRIER();                                 // H(s): ring slot (2 s + 3) % 3 is free
          n_iv = 0;
          if (more && split) issue_tile(s + 1, wi, 4, 1, 2);      // (must have landed at M(s + 1): not counted)
          if (2 * s + 3 < 2 * S) n_iv += issue_w(2 * s + 3, wi, 4);
        }
        wait_vmcnt(n_iv);                                  // tile s + 1 (and weight half 2 s + 2) have landed
        SSTAMP(c4);
#ifdef RTPE_CONV_STAMPS
        st[6] += c1 - c0; st[7] += c2 - c1; st[8] += c3 - c2; st[9] += c4 - c3; st[11] += 1;
#endif
      }
    }
  }
  // the last unit's sums (one group)
#ifdef RTPE_CONV_STAMPS
  SSTAMP(c0);
#endif
  if (pend >= 0) {
    if (use_res) load_res(pend);
    epilogue(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RTPE_CONV_STAMPS
  if (a.dbg != nullptr && lane == 0) {
    SSTAMP(c1);
    st[13] = c1 - c0;
    st[15] = c1 - k_begin;
    for (int i = 0; i < 16; ++i) if (st[i]) atomicAdd(&a.dbg[i], st[i]);
  }
#endif
}

template <int NT>
static int launch_stream_pc(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  static unsigned long long attr_mask = 0;
  auto kern = conv_stream_pc_kernel<NT>;
  if (first_use_on_device(&attr_mask)) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)t.grid), dim3(512), t.lds_bytes, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

// LDS of one workgroup: weight half-stage slots (3: ring, 2 * n_cchunks: resident) + two halo tile buffers + 2 x 48 floats
size_t conv_stream_pc_lds(const ConvPlan& p, int buf_bytes, int n_wslots) {
  return (size_t)n_wslots * p.mt * kKH * 1024 + (size_t)2 * buf_bytes + 512;    // + BN parameters of the cout block
}

// the layers this kernel takes: 3x3, stride 1, 48-channel chunks, 48-cout blocks (every BasicBlock conv of the w48 branches)
bool conv_stream_pc_supports(const ConvPlan& p) {
  return conv_stream_supports(p) && p.mt == 3 && p.in_mul == 1;
}

int conv_stream_pc_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(conv_stream_pc_supports(p), "streaming conv (pc): unsupported plan");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "streaming conv (pc): input view of %zu bytes", (size_t)a.x_bytes);
  RTPE_REQUIRE(a.cin % 8 == 0 && (a.in_ld >= a.cin ? a.in_cs == kCC : a.in_ld == kCC && a.in_cs >= kCC),
               "streaming conv (pc): cin=%d in_ld=%d chunk stride %lld", a.cin, a.in_ld, a.in_cs);
  RTPE_REQUIRE(a.y != nullptr && a.y_nchw == nullptr && a.o_mul == 1 && a.oy_add == 0 && a.ox_add == 0 && a.n_cls == 0,
               "streaming conv (pc): plain NHWC / plane-major output only");
  // output and residual are addressed through 2-GiB buffer windows from the view's base
  const size_t out_span = ((size_t)a.N * a.H_full * a.W_full * (size_t)(a.out_ld > 48 ? a.out_ld : 48) +
                           (size_t)(p.n_cb - 1) * (size_t)(a.out_cs > 48 ? a.out_cs : 0)) * 2;
  const size_t res_span = a.res == nullptr ? 0 : ((size_t)a.N * a.H_full * a.W_full * (size_t)(a.res_ld > 48 ? a.res_ld : 48) +
                                                  (size_t)(p.n_cb - 1) * (size_t)(a.res_cs > 48 ? a.res_cs : 0)) * 2;
  RTPE_REQUIRE(out_span < 0x7fffffffull && res_span < 0x7fffffffull, "streaming conv (pc): output view of %zu bytes", out_span);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && (t.grid / 8) % p.n_cb == 0, "streaming conv (pc): bad grid %d", t.grid);
  RTPE_REQUIRE(t.waves == 4 && t.n_bufs == 2 && t.buf_bytes % 16 == 0 && (t.n_wslots == 3 || t.n_wslots == 2 * p.n_cchunks) &&
               t.lds_bytes >= conv_stream_pc_lds(p, t.buf_bytes, t.n_wslots) && t.lds_bytes <= 160 * 1024,
               "streaming conv (pc): LDS layout (%d buffers of %d B, %d weight slots, %zu B)", t.n_bufs, t.buf_bytes,
               t.n_wslots, t.lds_bytes);
  RTPE_REQUIRE(t.th * t.tw == 16 * t.nt * t.waves && t.tw <= 255 && t.th <= 255, "streaming conv (pc): tile %dx%d", t.th, t.tw);
  RTPE_REQUIRE(a.halo_w * kSlots <= 256, "streaming conv (pc): halo row of %d pixels", a.halo_w);
  RTPE_REQUIRE(a.rowb >= a.halo_w * kPStride && (size_t)a.halo_h * a.rowb <= (size_t)t.buf_bytes, "streaming conv (pc): tile buffer too small");
  if (t.nt == 5) return launch_stream_pc<5>(t, a, s);
  if (t.nt == 4) return launch_stream_pc<4>(t, a, s);
  set_error("streaming conv (pc): no kernel variant nt=%d", t.nt);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
