// Persistent, software-pipelined implicit-GEMM convolution for the layers whose
// input is staged 48 channels at a time (Cin in {48, 96, 192, 384} and the
// 96-channel concat of the deconv: every BasicBlock, fuse conv and head of the
// w48 network, ~85 % of its FLOPs).  Same math, same epilogue rounding points
// and same packed-weight format as conv_mfma.hip; what changes is how the bytes
// move, because in-kernel stamps showed the non-persistent kernel spending
// 36 % of a wave's life staging and 54 % in the epilogue, 9 % in MFMAs:
//
//   * workgroups are persistent: each walks a strided list of work units
//     (output tile x block of 16*MT output channels), so set-up is paid once;
//   * a dedicated LOADER wave streams the next unit's halo tile HBM -> LDS with
//     `buffer_load ... lds` (LDS-DMA, 1 KiB per instruction, no VGPRs), into the
//     second of two LDS buffers, while the WAVES MFMA waves work on the current
//     one.  vmcnt completes in order, so keeping the DMA queue in its own wave
//     is what lets the MFMA waves wait for their weight fragments and residual
//     rows without draining the prefetch.  Out-of-image halo pixels use an
//     out-of-range buffer offset: the bounds check returns zeros (= padding);
//   * the halo tile is 96 B per pixel, slot-linear (exactly what LDS-DMA writes)
//     and measured bank-conflict free for the ds_read_b128 B-operand reads;
//   * results are transposed through the (then free) input buffer and stored,
//     with the residual rows, as whole 16-byte row pieces;
//   * units are ordered so that the workgroups of one XCD (blockIdx % 8) work on
//     the same few tiles at a time: the cout blocks of a tile share its halo
//     through that XCD's L2 instead of each fetching it from HBM.
#include "rtpe_common.h"

namespace rtpe {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kTapTableBytesP = 512;
constexpr int kCC = 48;          // channels per staged chunk
constexpr int kSlots = 6;        // 16-byte slots per staged pixel
constexpr int kPStride = 96;     // LDS bytes per staged pixel

__device__ __forceinline__ float round16p(float v) { return (float)(_Float16)v; }

#define RTPE_BARRIER()                          \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

// unit id of work item i (0-based) of the workgroup: XCD-aware order.
// XCD x (= blockIdx % 8) owns tiles x, x+8, x+16, ...; its G workgroups walk
// that XCD's (tile, cout block) sequence with stride G (G % n_cb == 0, so a
// workgroup keeps its cout block and the weights stay in its L1/L2).
struct UnitMap {
  int n_tiles, n_cb, G, xcd, j;
  __device__ __forceinline__ bool get(int i, int* tile, int* cb) const {
    const int seq = j + i * G;                       // position in this XCD's sequence
    const int tq = seq / n_cb;
    *cb = seq - tq * n_cb;
    *tile = xcd + 8 * tq;
    return *tile < n_tiles;
  }
};

template <int MT, int NT, int WAVES>
// second launch-bounds argument = waves per SIMD: 3 keeps two workgroups of WAVES+1 waves per CU
__global__ void __launch_bounds__((WAVES + 1) * 64, (NT <= 5 ? 3 : 2)) conv_persist_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);
  char* const bufs = smem + kTapTableBytesP;
  constexpr int NTHREADS = (WAVES + 1) * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int kvalid = a.ntaps * kCC;
  for (int i = tid; i < a.kc * 4; i += NTHREADS) {
    int kk = (i >> 2) * 32 + (i & 3) * 8;
    if (kk >= kvalid) kk -= kvalid;                  // zero-weight k padding: any finite in-tile data
    const int tap = kk / kCC;
    const int c = kk - tap * kCC;
    const int tyy = (a.tapw == 3) ? (tap * 11 >> 5) : (a.tapw == 2 ? (tap >> 1) : 0);
    const int txx = tap - tyy * a.tapw;
    tapoff[i] = (tyy * a.halo_w + txx) * kPStride + c * 2;
  }
  __syncthreads();

  UnitMap um;
  um.n_tiles = a.N * a.tiles_x * a.tiles_y;
  um.n_cb = a.n_cb;
  um.G = gridDim.x >> 3;
  um.xcd = blockIdx.x & 7;
  um.j = blockIdx.x >> 3;
  const uint32_t tiles_xy = (uint32_t)(a.tiles_x * a.tiles_y);
  const int ncc = a.n_cchunks;

  if (wv == WAVES) {
    // ------------------------------ loader wave ------------------------------
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const int rowslots = a.halo_w * kSlots;
    const int total = a.halo_h * rowslots;
    const int pieces = (total + 63) >> 6;            // 1-KiB DMA pieces per stage
    auto issue = [&](int tile, int cci, char* buf) {
      uint32_t t = (uint32_t)tile;
      const uint32_t n = fdiv(t, a.div_tiles_xy);
      t -= n * tiles_xy;
      const uint32_t tyi = fdiv(t, a.div_tiles_x);
      const uint32_t txi = t - tyi * a.tiles_x;
      const int iy0 = (int)tyi * a.th * a.in_mul + a.lo_y, ix0 = (int)txi * a.tw * a.in_mul + a.lo_x;
      const int cbase = cci * kCC;
      const uint32_t img = n * (uint32_t)(a.H_in * a.W_in);
      for (int c = 0; c < pieces; ++c) {
        const uint32_t idx = c * 64 + lane;
        const uint32_t hy = fdiv(idx, a.div_rowslots);
        const uint32_t q = idx - hy * rowslots;
        const uint32_t hx = q / kSlots;
        const uint32_t s = q - hx * kSlots;
        const int iy = iy0 + (int)hy, ix = ix0 + (int)hx;
        const bool ok = idx < (uint32_t)total && (unsigned)iy < (unsigned)a.H_in &&
                        (unsigned)ix < (unsigned)a.W_in && cbase + (int)s * 8 < a.cin;
        // out-of-range offset -> the buffer bounds check returns 0 = zero padding
        const uint32_t voff = ok ? ((img + (uint32_t)iy * a.W_in + ix) * (uint32_t)a.in_ld + cbase + s * 8) * 2u
                                 : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(buf + c * 1024), 16, (int)voff, 0, 0, 0);
      }
    };
    int tile, cb;
    int s = 0;
#ifdef RTPE_CONV_STAMPS
    unsigned long long l0, l1, l2, l3, l4, lw = 0, lb1 = 0, li = 0, lb2 = 0, ln = 0;
#define LSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LSTAMP(v)
#endif
    if (um.get(0, &tile, &cb)) issue(tile, 0, bufs);
    for (int i = 0; um.get(i, &tile, &cb); ++i) {
      for (int cci = 0; cci < ncc; ++cci, ++s) {
        LSTAMP(l0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stage s has landed in LDS
        LSTAMP(l1);
        RTPE_BARRIER();                                      // #1 hand it to the MFMA waves
        LSTAMP(l2);
        int ntile = tile, ncb, ncci = cci + 1;
        bool more = true;
        if (ncci == ncc) { ncci = 0; more = um.get(i + 1, &ntile, &ncb); }
        if (more) issue(ntile, ncci, bufs + ((s + 1) & 1) * a.buf_bytes);
        LSTAMP(l3);
        if (cci == ncc - 1) RTPE_BARRIER();                  // #E (epilogue reuses the buffer)
        RTPE_BARRIER();                                      // #2 buffer s&1 is free again
#ifdef RTPE_CONV_STAMPS
        LSTAMP(l4);
        lw += l1 - l0; lb1 += l2 - l1; li += l3 - l2; lb2 += l4 - l3; ++ln;
#endif
      }
    }
#ifdef RTPE_CONV_STAMPS
    if (a.dbg != nullptr && lane == 0) {
      atomicAdd(&a.dbg[6], lw); atomicAdd(&a.dbg[7], lb1); atomicAdd(&a.dbg[8], li);
      atomicAdd(&a.dbg[9], lb2); atomicAdd(&a.dbg[10], ln);
    }
#endif
    return;
  }

  // ------------------------------- MFMA waves --------------------------------
  const int r = lane & 15;
  const int g = lane >> 4;
  int pixbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const uint32_t p = (wv * NT + nt) * 16 + r;
    const uint32_t oy = fdiv(p, a.div_tw);
    const uint32_t ox = p - oy * a.tw;
    pixbase[nt] = (int)((oy * a.in_mul * a.halo_w + ox * a.in_mul) * kPStride);
  }
  const int n_k = ncc * a.kc;
  constexpr int ROWB = MT * 32 + 16;
  constexpr int CH = MT * 2;

#ifdef RTPE_CONV_STAMPS
#define PSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
  unsigned long long t0, t1, t2, t3, t4, acc_w1 = 0, acc_k = 0, acc_ep = 0, acc_w2 = 0, acc_setup = 0, n_units_done = 0;
#else
#define PSTAMP(v)
#endif
  constexpr int NIT = (NT * 16 * CH + 63) / 64;       // 16-byte row pieces per lane in the epilogue
  int tile, cb;
  int s = 0;
  for (int i = 0; um.get(i, &tile, &cb); ++i) {
    PSTAMP(t0);
    uint32_t t = (uint32_t)tile;
    const uint32_t n = fdiv(t, a.div_tiles_xy);
    t -= n * tiles_xy;
    const uint32_t tyi = fdiv(t, a.div_tiles_x);
    const uint32_t txi = t - tyi * a.tiles_x;
    const int py0 = tyi * a.th, px0 = txi * a.tw;

    float4v acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    const half8* wfrag = reinterpret_cast<const half8*>(a.w) + (size_t)cb * n_k * MT * 64 + lane;
    half8 a_cur[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = wfrag[m * 64];

    half8 rres[NIT];
    float4v al[MT], be[MT];
    int kf = 0;
    for (int cci = 0; cci < ncc; ++cci, ++s) {
      char* tilebuf = bufs + (s & 1) * a.buf_bytes;
      if (cci == ncc - 1) {                                  // BN / bias parameters of this cout block
        int g_p = g;
        asm volatile("" : "+v"(g_p));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int c4 = (cb * MT + m) * 16 + g_p * 4;
          al[m] = *reinterpret_cast<const float4v*>(a.alpha + c4);
          be[m] = *reinterpret_cast<const float4v*>(a.beta + c4);
        }
      }
      if (cci == ncc - 1 && a.res != nullptr) {
        // residual rows of this unit: issued now, they land while the last k-loop runs
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));                     // keep the address math inside the loop
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int c = it * 64 + lane_p;
          const int pw = c / CH, slot = c - pw * CH;
          const uint32_t p = wv * NT * 16 + pw;
          const uint32_t oyt = fdiv(p, a.div_tw);
          const uint32_t oxt = p - oyt * a.tw;
          const int py = py0 + (int)oyt, px = px0 + (int)oxt;
          const int ch = cb * MT * 16 + slot * 8;
          rres[it] = half8{0, 0, 0, 0, 0, 0, 0, 0};
          if (c < NT * 16 * CH && py < a.H_pos && px < a.W_pos && ch < a.cout_store) {
            const int oy = py * a.o_mul + a.oy_add, ox = px * a.o_mul + a.ox_add;
            const size_t pix = ((size_t)n * a.H_full + oy) * a.W_full + ox;
            rres[it] = *reinterpret_cast<const half8*>(a.res + pix * a.res_ld + ch);
          }
        }
      }
      PSTAMP(t1);
      RTPE_BARRIER();                                        // #1 stage s is in LDS
      PSTAMP(t2);
      for (int kci = 0; kci < a.kc; ++kci, ++kf) {
        half8 a_nxt[MT];                                     // weights one k-chunk ahead (L2 hits)
        const int kn = kf + 1 < n_k ? kf + 1 : n_k - 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = wfrag[(size_t)(kn * MT + m) * 64];
        const int off = tapoff[kci * 4 + g];
        half8 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          b[nt] = *reinterpret_cast<const half8*>(tilebuf + pixbase[nt] + off);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[m], b[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
      }
      PSTAMP(t3);
      if (cci == ncc - 1) {
        // ---- epilogue: BN/bias (+ residual) (+ ReLU), transposed through LDS ----
        // the lane id is made opaque here: everything below depends only on the lane and would
        // otherwise be hoisted out of the persistent loop (it cost 200+ spilled VGPRs)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int r = lane_e & 15, g = lane_e >> 4;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RTPE_BARRIER();                                      // #E every MFMA wave is done with the tile
        char* obuf = tilebuf + wv * (NT * 16 * ROWB);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float4v v = acc[m][nt];
            half4 o;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
              float x = v[jx];
              if (a.round_conv) x = round16p(x);
              x = round16p(__builtin_fmaf(x, al[m][jx], be[m][jx]));
              v[jx] = x;
              o[jx] = (_Float16)x;
            }
            *reinterpret_cast<half4*>(obuf + (nt * 16 + r) * ROWB + m * 32 + g * 8) = o;
            if (a.y_nchw != nullptr) {                       // heads: NCHW straight from the registers
              const uint32_t p = (wv * NT + nt) * 16 + r;
              const uint32_t oyt = fdiv(p, a.div_tw);
              const uint32_t oxt = p - oyt * a.tw;
              const int py = py0 + (int)oyt, px = px0 + (int)oxt;
              if (py < a.H_pos && px < a.W_pos) {
                const int oy = py * a.o_mul + a.oy_add, ox = px * a.o_mul + a.ox_add;
                const int c4 = (cb * MT + m) * 16 + g * 4;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                  const int c = c4 + jx;
                  if (c < a.nchw_channels) {
                    const float x = a.relu ? (v[jx] > 0.f ? v[jx] : 0.f) : v[jx];
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
          const int cblk = cb * MT * 16;
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane_e;
            if (c >= NT * 16 * CH) continue;
            const int pw = c / CH, slot = c - pw * CH;
            const uint32_t p = wv * NT * 16 + pw;
            const uint32_t oyt = fdiv(p, a.div_tw);
            const uint32_t oxt = p - oyt * a.tw;
            const int py = py0 + (int)oyt, px = px0 + (int)oxt;
            const int ch = cblk + slot * 8;
            if (py >= a.H_pos || px >= a.W_pos || ch >= a.cout_store) continue;
            const int oy = py * a.o_mul + a.oy_add, ox = px * a.o_mul + a.ox_add;
            const size_t pix = ((size_t)n * a.H_full + oy) * a.W_full + ox;
            half8 v = *reinterpret_cast<const half8*>(obuf + pw * ROWB + slot * 16);
            if (a.res != nullptr) {
              const half8 rr = rres[it];
#pragma unroll
              for (int jx = 0; jx < 8; ++jx) v[jx] = (_Float16)((float)v[jx] + (float)rr[jx]);
            }
            if (a.relu) {
#pragma unroll
              for (int jx = 0; jx < 8; ++jx) v[jx] = v[jx] > (_Float16)0.f ? v[jx] : (_Float16)0.f;
            }
            *reinterpret_cast<half8*>(a.y + pix * a.out_ld + ch) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS traffic on the buffer is over
      PSTAMP(t4);
      RTPE_BARRIER();                                        // #2 the loader may refill buffer s&1
#ifdef RTPE_CONV_STAMPS
      { unsigned long long t5; PSTAMP(t5);
        acc_w1 += t2 - t1; acc_k += t3 - t2; acc_ep += t4 - t3; acc_w2 += t5 - t4;
        if (cci == 0) acc_setup += t1 - t0; }
#endif
    }
#ifdef RTPE_CONV_STAMPS
    ++n_units_done;
#endif
  }
#ifdef RTPE_CONV_STAMPS
  if (a.dbg != nullptr && lane == 0) {
    atomicAdd(&a.dbg[0], acc_setup); atomicAdd(&a.dbg[1], acc_w1); atomicAdd(&a.dbg[2], acc_k);
    atomicAdd(&a.dbg[3], acc_ep); atomicAdd(&a.dbg[4], acc_w2); atomicAdd(&a.dbg[5], n_units_done);
  }
#endif
}

template <int MT, int NT, int WAVES>
static int launch_persist(const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_persist_kernel<MT, NT, WAVES>;
  if (!attr_set) {
    RTPE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)t.grid), dim3((WAVES + 1) * 64), t.lds_bytes, s, a);
  RTPE_HIP_CHECK(hipGetLastError());
  return RTPE_OK;
}

int conv_persist_launch(const ConvPlan& p, const ConvTile& t, const ConvArgs& a, hipStream_t s) {
  RTPE_REQUIRE(p.cc == kCC && p.pstride == kPStride, "persistent conv: plan is not 48-channel chunked");
  RTPE_REQUIRE(a.x_bytes > 0 && a.x_bytes < 0x80000000ull, "persistent conv: input view of %zu bytes", (size_t)a.x_bytes);
  RTPE_REQUIRE(t.grid >= 8 && t.grid % 8 == 0 && (t.grid / 8) % p.n_cb == 0, "persistent conv: bad grid %d", t.grid);
#define RTPE_P(MTv, NTv, Wv) \
  if (p.mt == MTv && t.nt == NTv && t.waves == Wv) return launch_persist<MTv, NTv, Wv>(t, a, s);
  RTPE_P(3, 8, 4) RTPE_P(3, 4, 4) RTPE_P(3, 2, 4) RTPE_P(3, 5, 4) RTPE_P(3, 5, 5)
  RTPE_P(2, 8, 4) RTPE_P(2, 4, 4) RTPE_P(2, 2, 4) RTPE_P(2, 5, 4) RTPE_P(2, 5, 5)
  RTPE_P(1, 8, 4) RTPE_P(1, 4, 4) RTPE_P(1, 2, 4) RTPE_P(1, 5, 4) RTPE_P(1, 5, 5)
  RTPE_P(4, 4, 4) RTPE_P(4, 2, 4) RTPE_P(4, 5, 4) RTPE_P(4, 5, 5)
#undef RTPE_P
  set_error("persistent conv: no kernel variant mt=%d nt=%d waves=%d", p.mt, t.nt, t.waves);
  return RTPE_E_INVALID;
}

}  // namespace rtpe
