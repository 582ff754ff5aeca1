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
#include "conv_stream_dev.h"

namespace rtpe {


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
         (p.in_mul == 1 || p.in_mul == 2) && (p.mt <= 3 || (p.mt == 6 && p.in_mul == 2 && p.n_cchunks == 1)) &&
         (p.n_cchunks & (p.n_cchunks - 1)) == 0;
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
  RTPE_S(3, 4, 4) RTPE_S(3, 5, 4) RTPE_S(3, 5, 5) RTPE_S(3, 2, 4) RTPE_S(6, 1, 4)
  RTPE_S(2, 4, 4) RTPE_S(2, 5, 4) RTPE_S(2, 5, 5)
  RTPE_S(1, 4, 4) RTPE_S(1, 5, 4) RTPE_S(1, 5, 5)
#undef RTPE_S
  set_error("streaming conv: no kernel variant mt=%d nt=%d waves=%d", p.mt, t.nt, t.waves);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
