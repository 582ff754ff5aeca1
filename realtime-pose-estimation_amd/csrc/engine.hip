// Program executor: runs the flat op list that the Python module tree compiles
// itself into (rtpe/third_party/pose_higher_hrnet.py of this repo) with the
// kernels of conv_mfma.hip / elementwise.hip.  Owns only the packed weights.
#include <stdarg.h>

#include <chrono>
#include <mutex>
#include <map>
#include <tuple>
#include <vector>

#include "rtpe_common.h"

namespace rtpe {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return RTPE_E_HIP;
}

struct OpState {
  rtpe_op_desc d;
  ConvGeom geom[4];   // 1 for CONV, 4 parity classes for DECONV
  ConvPlan plan[4];
  size_t w_dev_off[4];  // byte offsets in the device weight arena
  size_t ab_dev_off;    // alpha then beta, fp32[cout_pad]
  int n_geom;
  int fuse;             // 1: head of a fused BasicBlock (this conv + the next run as one kernel), 2: its second conv
  int pair;             // 1: head of a 1x1 pair (conv_pair.hip: launched with the next op), 2: its tail
  int stem2;            // 1: the stem op whose conv1 runs together with the next op's conv2 (stem_fused.hip), 2: that conv op
  int s2g;              // n > 1: first of n neighbouring 3x3 stride-2 convs from the SAME 48-channel input that run as one launch
                        // (conv48s2.hip: the input is read once), -1: one of the others
};

}  // namespace rtpe

using namespace rtpe;

struct rtpe_hrnet {
  int device;
  std::vector<OpState> ops;
  // autotuned launch shapes: (N, H, W) -> one ConvTile per (op, parity class); nt == 0 = not tuned
  std::map<std::tuple<int, int, int>, std::vector<ConvTile>> tuned;
  std::map<int, std::vector<hipEvent_t>> records;   // per-op events of rtpe_hrnet_forward_record, by slot
  std::vector<rtpe_tensor_desc> tensors;
  // tensors that MAY be kept plane-major ([C/48][N][H][W][48] instead of NHWC): C >= 96 and every op that
  // touches them is a 3x3 stride-1 conv the streaming kernel runs (the inner tensors of a BasicBlock chain).
  // A 48-channel block is then one contiguous row per image row instead of 96 bytes of every 2C: the
  // kernel's stores, residual loads and halo DMA move whole cache lines.  Decided per run (see run()).
  std::vector<char> plane_ok;
  int n_slots;
  char* arena;        // device: packed weights + affine params
  size_t arena_bytes;
  int n_preds, n_refined;
  // parallel regions (rtpe_op_desc::lane / region): internal streams for lanes 1..3 and one event per op whose output
  // another lane reads, created on first use; wait_ops[i] = ops of other lanes whose output op i reads
  // THREADS: the lane streams and events are state of the handle; a forward with lanes on holds lane_mu from the creation
  // of that state (first such forward) to the join of its last region, so two host threads on one handle enqueue their
  // lane regions one after the other (their kernels still overlap on the GPU when they use different streams)
  std::mutex lane_mu;
  hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> op_event;
  std::vector<char> needs_event;
  std::vector<std::vector<int>> wait_ops;
  hipEvent_t fork_event = nullptr;
  bool has_regions = false;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

namespace rtpe {
static int g_options[kNumOptions] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};       // -1: not set, take the environment's value
static const char* const kOptionNames[kNumOptions] = {"block_ring", "block_pc", "direct_1x1", "lanes", "tile_dma", "pair_1x1", "fused_stem", "conv64", "head_direct", "deconv48", "conv48s2"};
static const char* const kOptionEnv[kNumOptions] = {"RTPE_BLOCK_RING", "RTPE_BLOCK_PC", "RTPE_DIRECT_1X1", "RTPE_LANES", "RTPE_TILE_DMA", "RTPE_PAIR_1X1", "RTPE_FUSED_STEM", "RTPE_CONV64", "RTPE_HEAD_DIRECT", "RTPE_DECONV48", "RTPE_CONV48S2"};
static const int kOptionDefault[kNumOptions] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
int get_option(int key) {
  int v = __atomic_load_n(&g_options[key], __ATOMIC_RELAXED);
  if (v < 0) {
    v = kOptionEnv[key][0] ? env_int(kOptionEnv[key], kOptionDefault[key]) : kOptionDefault[key];
    __atomic_store_n(&g_options[key], v, __ATOMIC_RELAXED);
  }
  return v;
}
}  // namespace rtpe

extern "C" int rtpe_set_option(const char* name, int32_t value) {
  RTPE_REQUIRE(name != nullptr && value >= 0, "set_option: bad argument");
  for (int k = 0; k < kNumOptions; ++k)
    if (kOptionNames[k][0] && strcmp(name, kOptionNames[k]) == 0) {
      __atomic_store_n(&g_options[k], (int)value, __ATOMIC_RELAXED);
      return RTPE_OK;
    }
  set_error("set_option: unknown option '%s'", name);
  return RTPE_E_INVALID;
}

extern "C" int rtpe_get_option(const char* name, int32_t* value) {
  RTPE_REQUIRE(name != nullptr && value != nullptr, "get_option: null argument");
  for (int k = 0; k < kNumOptions; ++k)
    if (kOptionNames[k][0] && strcmp(name, kOptionNames[k]) == 0) {
      *value = get_option(k);
      return RTPE_OK;
    }
  set_error("get_option: unknown option '%s'", name);
  return RTPE_E_INVALID;
}

extern "C" const char* rtpe_last_error_string(void) { return g_err.c_str(); }
extern "C" int rtpe_version(void) { return 4; }   // 4: no "stream_pc" / kind 3; 3: 11 integers per tuned record; 2: rtpe_op_desc has lane / region, rtpe_hrnet_forward_flags
extern "C" int rtpe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int rtpe_hrnet_create(const rtpe_op_desc* ops, int32_t n_ops,
                                 const rtpe_tensor_desc* tensors, int32_t n_tensors,
                                 const void* weights, size_t weights_bytes, int32_t device,
                                 rtpe_hrnet** out) {
  RTPE_REQUIRE(ops && tensors && weights && out && n_ops > 0 && n_tensors > 0, "hrnet_create: null argument");
  DeviceGuard guard(device);               // the caller's current device is restored on return
  RTPE_HIP_CHECK(guard.err);
  rtpe_hrnet* h = new rtpe_hrnet();
  h->device = device;
  h->tensors.assign(tensors, tensors + n_tensors);
  h->n_slots = 0;
  for (auto& t : h->tensors) h->n_slots = t.slot + 1 > h->n_slots ? t.slot + 1 : h->n_slots;
  h->n_preds = h->n_refined = 0;
  h->arena = nullptr;
  const char* wb = reinterpret_cast<const char*>(weights);

  // pass 1: plans + arena layout
  size_t off = 0;
  h->ops.resize(n_ops);
  for (int i = 0; i < n_ops; ++i) {
    OpState& o = h->ops[i];
    o.d = ops[i];
    o.n_geom = 0;
    const rtpe_op_desc& d = o.d;
    auto bad_t = [&](int t) { return t < 0 || t >= n_tensors; };
    if (d.kind != RTPE_OP_FUSE && d.kind != RTPE_OP_STEM && d.kind != RTPE_OP_AUX_PACK && bad_t(d.in_t)) {
      set_error("op %d: bad input tensor", i); delete h; return RTPE_E_INVALID;
    }
    if (bad_t(d.out_t) && !(d.flags & RTPE_F_NO_NHWC)) {
      set_error("op %d: bad output tensor", i); delete h; return RTPE_E_INVALID;
    }
    if (d.flags & RTPE_F_OUT_PREDS) h->n_preds = d.cout;
    if (d.flags & RTPE_F_OUT_REFINED) h->n_refined = d.cout;
    if (d.kind == RTPE_OP_CONV || d.kind == RTPE_OP_DECONV) {
      o.n_geom = d.kind == RTPE_OP_DECONV ? 4 : 1;
      for (int k = 0; k < o.n_geom; ++k) {
        o.geom[k] = ConvGeom{d.cin, d.cout, d.ksize, d.stride, d.kind == RTPE_OP_DECONV ? k : -1,
                             (d.flags & RTPE_F_F32) ? 4 : 2, d.reserved[1] > 0 ? d.reserved[1] : 1};
        o.plan[k] = conv_make_plan(o.geom[k]);
        o.w_dev_off[k] = off;
        off = align_up(off + o.plan[k].packed_bytes, 256);
      }
      o.ab_dev_off = off;
      off = align_up(off + 2 * sizeof(float) * o.plan[0].cout_pad, 256);
      const size_t wbytes = (size_t)d.cin * d.cout * d.ksize * d.ksize * ((d.flags & RTPE_F_F32) ? 4 : 2);
      if (d.w_off < 0 || (size_t)d.w_off + wbytes > weights_bytes ||
          d.ab_off < 0 || (size_t)d.ab_off + 8 * (size_t)d.cout > weights_bytes) {
        set_error("op %d: weight offsets out of range", i); delete h; return RTPE_E_INVALID;
      }
    } else if (d.kind == RTPE_OP_STEM) {
      o.w_dev_off[0] = off;
      off = align_up(off + 27 * 64 * 4, 256);
      o.ab_dev_off = off;
      off = align_up(off + 2 * sizeof(float) * 64, 256);
      if (d.cout != 64 || d.cin != 3) { set_error("stem must be 3->64"); delete h; return RTPE_E_INVALID; }
    } else if (d.kind == RTPE_OP_FUSE) {
      if (d.n_terms < 1 || d.n_terms > 4) { set_error("op %d: fuse terms", i); delete h; return RTPE_E_INVALID; }
    } else if (d.kind == RTPE_OP_SE) {
      const size_t wbytes = ((size_t)d.cout * d.cin * 2 + d.cout + d.cin) * 4;
      if (d.w_off < 0 || (size_t)d.w_off + wbytes > weights_bytes) {
        set_error("op %d: SE weights out of range", i); delete h; return RTPE_E_INVALID;
      }
      o.w_dev_off[0] = off;
      off = align_up(off + wbytes, 256);
    } else if (d.kind == RTPE_OP_CAST || d.kind == RTPE_OP_AVGPOOL || d.kind == RTPE_OP_CAM_COMBINE ||
               d.kind == RTPE_OP_SIGMOID_ADD || d.kind == RTPE_OP_RESIZE || d.kind == RTPE_OP_GATE_MUL) {
      if ((d.kind == RTPE_OP_CAM_COMBINE && (bad_t(d.res_t) || bad_t(d.term_t[0]))) ||
          ((d.kind == RTPE_OP_SIGMOID_ADD || d.kind == RTPE_OP_GATE_MUL) && bad_t(d.res_t))) {
        set_error("op %d: missing operand tensor", i); delete h; return RTPE_E_INVALID;
      }
    } else if (d.kind == RTPE_OP_AUX_PACK) {
      if (h->tensors[d.out_t].reserved != 4 || h->tensors[d.out_t].channels < 4) {
        set_error("op %d: the packed second input must be an fp32 tensor of >= 4 channels", i); delete h; return RTPE_E_INVALID;
      }
    } else {
      set_error("op %d: unknown kind %d", i, d.kind); delete h; return RTPE_E_INVALID;
    }
  }
  // BasicBlock fusion (conv_block.hip): conv 48->48 k3 s1 +relu, then conv 48->48 k3 s1 +residual(+relu)
  // whose residual is the first conv's input and whose input is read by nobody else
  static const int fuse_blocks = getenv("RTPE_FUSE_BLOCKS") ? atoi(getenv("RTPE_FUSE_BLOCKS")) : 1;
  for (size_t i = 0; fuse_blocks && i + 2 < h->ops.size(); ++i) {     // (the last op keeps its own event)
    OpState& a1 = h->ops[i];
    OpState& a2 = h->ops[i + 1];
    const rtpe_op_desc& d1 = a1.d;
    const rtpe_op_desc& d2 = a2.d;
    const int plain = RTPE_F_RELU | RTPE_F_ROUND_CONV;
    if (a1.fuse || d1.kind != RTPE_OP_CONV || d2.kind != RTPE_OP_CONV) continue;
    if (d1.cin != 48 || d1.cout != 48 || d2.cin != 48 || d2.cout != 48 || d1.ksize != 3 || d2.ksize != 3 ||
        d1.stride != 1 || d2.stride != 1 || d1.flags != plain || d2.flags != plain || d1.res_t >= 0 ||
        d2.res_t != d1.in_t || d2.res_coff != d1.in_coff || d2.in_t != d1.out_t || d2.in_coff != d1.out_coff ||
        d1.reserved[1] > 1 || d2.reserved[1] > 1 || h->tensors[d1.in_t].reserved == 4 || d2.out_t == d1.in_t)
      continue;
    bool other_reader = false;                       // the intermediate tensor must die inside the block
    for (size_t k = 0; k < h->ops.size() && !other_reader; ++k) {
      if (k == i + 1) continue;
      const rtpe_op_desc& dk = h->ops[k].d;
      if (k > i && (dk.in_t == d1.out_t || dk.res_t == d1.out_t)) other_reader = true;
      for (int t = 0; t < dk.n_terms && k > i; ++t) other_reader |= dk.term_t[t] == d1.out_t;
    }
    if (other_reader) continue;
    a1.fuse = 1;
    a2.fuse = 2;
  }
  // fused stem (stem_fused.hip): the stem op, then conv 64->64 k3 s2 (+relu) that is the only reader of its output
  for (OpState& o : h->ops) o.stem2 = 0;
  for (size_t i = 0; i + 1 < h->ops.size(); ++i) {
    OpState& a1 = h->ops[i];
    OpState& a2 = h->ops[i + 1];
    const rtpe_op_desc &d1 = a1.d, &d2 = a2.d;
    if (d1.kind != RTPE_OP_STEM || (d1.flags & RTPE_F_F32) || d2.kind != RTPE_OP_CONV || a2.fuse || a2.n_geom != 1) continue;
    const ConvPlan& p2 = a2.plan[0];
    if (d2.cin != 64 || d2.cout != 64 || d2.ksize != 3 || d2.stride != 2 || d2.res_t >= 0 || d2.n_terms != 0 ||
        (d2.flags & ~(RTPE_F_RELU | RTPE_F_ROUND_CONV)) || d2.in_t != d1.out_t || d2.in_coff != 0 || d1.out_coff != 0 ||
        d2.reserved[1] > 1 || d2.reserved[2] > 0 || d2.out_t == d1.out_t || h->tensors[d1.out_t].channels != 64 ||
        h->tensors[d2.out_t].channels - d2.out_coff < 64 || d1.lane != d2.lane || d1.region != d2.region ||
        p2.mt != 4 || p2.cc != 64 || p2.n_cchunks != 1 || p2.kc != 18 || p2.n_cb != 1 || p2.esize != 2)
      continue;
    bool other_reader = false;
    for (size_t k = i + 2; k < h->ops.size() && !other_reader; ++k) {
      const rtpe_op_desc& dk = h->ops[k].d;
      if (dk.in_t == d1.out_t || dk.res_t == d1.out_t) other_reader = true;
      for (int t = 0; t < dk.n_terms; ++t) other_reader |= dk.term_t[t] == d1.out_t;
    }
    if (other_reader) continue;
    a1.stem2 = 1;
    a2.stem2 = 2;
  }
  // 1x1 pairs (conv_pair.hip): the flags are the program's promise that the head's input and residual stay alive over the
  // tail; whether the two ops ARE such a pair is checked here (anything else: the flags are ignored)
  for (OpState& o : h->ops) o.pair = 0;
  for (size_t i = 0; i < h->ops.size(); ++i) {
    OpState& a1 = h->ops[i];
    if (!(a1.d.flags & RTPE_F_PAIR_HEAD) || i + 1 >= h->ops.size()) continue;
    const rtpe_op_desc &d1 = a1.d, &d2 = h->ops[i + 1].d;
    const int want1 = RTPE_F_RELU | RTPE_F_ROUND_CONV | RTPE_F_PAIR_HEAD, want2 = RTPE_F_RELU | RTPE_F_ROUND_CONV | RTPE_F_PAIR_TAIL;
    if (d1.kind != RTPE_OP_CONV || d2.kind != RTPE_OP_CONV || d1.flags != want1 || d2.flags != want2 || d1.ksize != 1 ||
        d2.ksize != 1 || d1.stride != 1 || d2.stride != 1 || d1.res_t < 0 || d2.res_t >= 0 || d2.in_t != d1.out_t ||
        d2.in_coff != d1.out_coff || d2.cin != d1.cout || !conv_pair_supports(d1.cin, d1.cout, d2.cout) ||
        h->tensors[d1.in_t].reserved == 4 || d1.reserved[2] > 0 || d2.reserved[2] > 0 || d2.out_t == d1.in_t ||
        d2.out_t == d1.res_t || d2.out_t == d1.out_t || d1.lane != d2.lane || d1.region != d2.region)
      continue;
    a1.pair = 1;
    h->ops[i + 1].pair = 2;
  }
  // neighbouring downsampling convs that read the same 48-channel tensor (the first convs of a fuse layer's chains from branch 0,
  // pose_higher_hrnet.py:213-230: 48 -> 96 to branch 1, 48 -> 48 towards branches 2 and 3): one launch of conv48s2.hip reads the
  // input once for all of them.  Static conditions here, the launch's own (sizes, layouts, option) in run()
  for (OpState& o : h->ops) o.s2g = 0;
  {
    auto s2_static = [&](const OpState& o) {
      const rtpe_op_desc& d = o.d;
      return d.kind == RTPE_OP_CONV && d.ksize == 3 && d.stride == 2 && d.cin == 48 && (d.cout == 48 || d.cout == 96) && d.res_t < 0 &&
             !(d.flags & (RTPE_F_NO_NHWC | RTPE_F_F32 | RTPE_F_OUT_PREDS | RTPE_F_OUT_REFINED)) && o.n_geom == 1 && o.plan[0].esize == 2 &&
             o.plan[0].cout_pad == d.cout && d.reserved[2] <= 0;
    };
    static const int s2_groups = env_int("RTPE_S2_GROUPS", 1);
    for (size_t i = 0; s2_groups && i < h->ops.size(); ++i) {
      if (h->ops[i].s2g != 0 || !s2_static(h->ops[i])) continue;
      const rtpe_op_desc& d0 = h->ops[i].d;
      int n = 1, groups = d0.cout / 48;
      while (i + n < h->ops.size() && n < 3) {
        const OpState& on = h->ops[i + n];
        const rtpe_op_desc& dn = on.d;
        bool ok = s2_static(on) && dn.in_t == d0.in_t && dn.in_coff == d0.in_coff && dn.lane == d0.lane && dn.region == d0.region &&
                  (dn.flags & RTPE_F_ROUND_CONV) == (d0.flags & RTPE_F_ROUND_CONV) && groups + dn.cout / 48 <= 4;
        for (int m = 0; m < n && ok; ++m) ok = h->ops[i + m].d.out_t != dn.out_t && dn.out_t != d0.in_t;
        if (!ok) break;
        groups += dn.cout / 48;
        ++n;
      }
      if (n < 2) continue;
      h->ops[i].s2g = n;
      for (int m = 1; m < n; ++m) h->ops[i + m].s2g = -1;
    }
  }
  h->arena_bytes = off;
  {
    // cross-lane dependencies inside parallel regions: the last writer of every tensor an op reads, when it ran on
    // another lane of the same region (a region starts with a fork from and ends with a join into the caller's stream,
    // so nothing outside it needs an event)
    h->needs_event.assign(n_ops, 0);
    h->wait_ops.assign(n_ops, std::vector<int>());
    std::vector<int> writer(h->tensors.size(), -1);
    std::vector<std::vector<int>> readers(h->tensors.size());     // ops that read a tensor since its last writer
    for (int i = 0; i < n_ops; ++i) {
      const rtpe_op_desc& d = h->ops[i].d;
      if (d.lane < 0 || d.lane > 3 || d.region < 0) { set_error("op %d: lane %d region %d", i, d.lane, d.region); delete h; return RTPE_E_INVALID; }
      if (d.region > 0) h->has_regions = true;
      if (d.region == 0 && d.lane != 0) { set_error("op %d: a lane outside a region", i); delete h; return RTPE_E_INVALID; }
      std::vector<int> reads;
      if (d.kind != RTPE_OP_FUSE && d.in_t >= 0) reads.push_back(d.in_t);
      if (d.res_t >= 0) reads.push_back(d.res_t);
      for (int t = 0; t < d.n_terms; ++t) reads.push_back(d.term_t[t]);
      if (d.out_t >= 0) reads.push_back(d.out_t);          // (a second writer of a shared buffer: ordered behind the first)
      for (int t : reads) {
        const int j = writer[t];
        if (j >= 0 && h->ops[j].d.region == d.region && d.region > 0 && h->ops[j].d.lane != d.lane) {
          h->needs_event[j] = 1;
          h->wait_ops[i].push_back(j);
        }
      }
      // write after read: a lane that overwrites a tensor (an in-place op, a shared slot written through `into=`) waits
      // for the ops of OTHER lanes that read the previous contents
      if (d.out_t >= 0) {
        for (int j : readers[d.out_t]) {
          if (j != i && h->ops[j].d.region == d.region && d.region > 0 && h->ops[j].d.lane != d.lane) {
            h->needs_event[j] = 1;
            h->wait_ops[i].push_back(j);
          }
        }
        readers[d.out_t].clear();
      }
      if (d.kind != RTPE_OP_FUSE && d.in_t >= 0) readers[d.in_t].push_back(i);
      if (d.res_t >= 0) readers[d.res_t].push_back(i);
      for (int t = 0; t < d.n_terms; ++t) readers[d.term_t[t]].push_back(i);
      if (d.out_t >= 0) writer[d.out_t] = i;
    }
  }
  {
    static const int plane_major = getenv("RTPE_PLANE_MAJOR") ? atoi(getenv("RTPE_PLANE_MAJOR")) : 1;
    h->plane_ok.assign(h->tensors.size(), 0);
    for (size_t t = 0; plane_major && t < h->tensors.size(); ++t) {
      const rtpe_tensor_desc& td = h->tensors[t];
      if (td.reserved == 4 || td.channels < 96 || td.channels % 48 != 0) continue;
      bool ok = true, touched = false;
      for (const OpState& o : h->ops) {
        const rtpe_op_desc& d = o.d;
        bool uses = d.in_t == (int)t || d.out_t == (int)t || d.res_t == (int)t;
        for (int k = 0; k < d.n_terms; ++k) uses |= d.term_t[k] == (int)t;
        if (!uses) continue;
        touched = true;
        const int clean = RTPE_F_RELU | RTPE_F_ROUND_CONV;
        if (d.kind != RTPE_OP_CONV || o.n_geom != 1 || d.n_terms != 0 || (d.flags & ~clean) || d.ksize != 3 ||
            d.stride != 1 || !conv_stream_supports(o.plan[0]) || o.plan[0].mt != 3 || d.cout % 48 != 0 ||
            d.in_coff != 0 || d.out_coff != 0 || (d.res_t >= 0 && d.res_coff != 0) || d.reserved[2] != 0 ||
            d.cin != h->tensors[d.in_t].channels || d.cout != h->tensors[d.out_t].channels ||
            (d.res_t >= 0 && h->tensors[d.res_t].channels != d.cout)) {
          ok = false;
          break;
        }
      }
      h->plane_ok[t] = ok && touched;
    }
  }

  // pass 2: pack on the host, one upload
  std::vector<char> host(off, 0);
  for (int i = 0; i < n_ops; ++i) {
    OpState& o = h->ops[i];
    const rtpe_op_desc& d = o.d;
    if (d.kind == RTPE_OP_CONV || d.kind == RTPE_OP_DECONV) {
      for (int k = 0; k < o.n_geom; ++k)
        conv_pack_weights(o.geom[k], o.plan[k], wb + d.w_off, host.data() + o.w_dev_off[k]);
      float* ab = reinterpret_cast<float*>(host.data() + o.ab_dev_off);
      const float* src = reinterpret_cast<const float*>(wb + d.ab_off);
      const int cp = o.plan[0].cout_pad;
      for (int c = 0; c < d.cout; ++c) { ab[c] = src[c]; ab[cp + c] = src[d.cout + c]; }
    } else if (d.kind == RTPE_OP_STEM) {
      const int es = (d.flags & RTPE_F_F32) ? 4 : 2;                 // (64,3,3,3) in the program's precision
      const char* w = wb + d.w_off;
      char* p = host.data() + o.w_dev_off[0];
      for (int co = 0; co < 64; ++co)
        for (int c = 0; c < 3; ++c)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
              memcpy(p + (size_t)(((ky * 3 + kx) * 3 + c) * 64 + co) * es,
                     w + (size_t)(((co * 3 + c) * 3 + ky) * 3 + kx) * es, es);
      memcpy(host.data() + o.ab_dev_off, wb + d.ab_off, 2 * 64 * sizeof(float));
    } else if (d.kind == RTPE_OP_SE) {
      memcpy(host.data() + o.w_dev_off[0], wb + d.w_off, ((size_t)d.cout * d.cin * 2 + d.cout + d.cin) * 4);
    }
  }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->arena), off ? off : 256);
  if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(arena)", __FILE__, __LINE__); }
  e = hipMemcpy(h->arena, host.data(), off, hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(h->arena); delete h; return hip_fail(e, "hipMemcpy(arena)", __FILE__, __LINE__); }
  *out = h;
  return RTPE_OK;
}

extern "C" int rtpe_hrnet_destroy(rtpe_hrnet* h) {
  if (!h) return RTPE_OK;
  DeviceGuard guard(h->device);
  for (auto& kv : h->records)
    for (auto& e : kv.second) hipEventDestroy(e);
  for (auto& e : h->op_event) if (e) hipEventDestroy(e);
  if (h->fork_event) hipEventDestroy(h->fork_event);
  for (int l = 1; l < 4; ++l) if (h->lane_stream[l]) hipStreamDestroy(h->lane_stream[l]);
  if (h->arena) hipFree(h->arena);
  delete h;
  return RTPE_OK;
}

static void slot_layout(const rtpe_hrnet* h, int N, int H, int W, std::vector<size_t>* offs, size_t* total) {
  std::vector<size_t> sz(h->n_slots, 0);
  for (const auto& t : h->tensors) {
    const size_t px = t.ds_log2 < 0 ? 1 : (size_t)(H >> t.ds_log2) * (W >> t.ds_log2);
    const size_t b = (size_t)N * px * t.channels * (t.reserved == 4 ? 4 : 2);
    if (b > sz[t.slot]) sz[t.slot] = b;
  }
  offs->resize(h->n_slots);
  size_t o = 0;
  for (int s = 0; s < h->n_slots; ++s) { (*offs)[s] = o; o += align_up(sz[s], 256); }
  *total = o;
}

extern "C" int rtpe_hrnet_workspace_bytes(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, size_t* bytes) {
  RTPE_REQUIRE(h && bytes && N > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0,
               "workspace_bytes: N=%d H=%d W=%d (H, W must be multiples of 32)", N, H, W);
  std::vector<size_t> offs;
  slot_layout(h, N, H, W, &offs, bytes);
  return RTPE_OK;
}

// a fused BasicBlock pair (OpState::fuse) runs as one kernel at this map size (conv_block.hip: 48 channels, from 6 x 16 maps up).
// (A fused kernel for the 96-channel branch was built and measured in round 5 - slower than the two streaming launches:
// profiles/r05_block96_ablation.txt, tools/experiments/conv_block96.hip.)
static bool fused_block_runs(const rtpe_hrnet* h, const rtpe_op_desc& d, int N, int H, int W) {
  const rtpe_tensor_desc& ti = h->tensors[d.in_t];
  (void)N;
  return conv_block_supports(d.cin, d.cout, H >> ti.ds_log2, W >> ti.ds_log2);
}

// plane-major tensors of one run: the candidates all of whose ops run on the streaming kernel with the launch
// shapes in force (tiny maps fall back to the generic kernel, which only knows NHWC)
static std::vector<char> plane_tensors(const rtpe_hrnet* h, int N, int H, int W, const std::vector<ConvTile>* tuned) {
  std::vector<char> plane = h->plane_ok;
  for (size_t i = 0; i < h->ops.size(); ++i) {
    const OpState& o = h->ops[i];
    const rtpe_op_desc& d = o.d;
    if (d.kind != RTPE_OP_CONV || o.n_geom != 1) continue;
    if (!(plane[d.in_t] || plane[d.out_t] || (d.res_t >= 0 && plane[d.res_t]))) continue;
    const rtpe_tensor_desc& ti = h->tensors[d.in_t];
    const ConvTile t = (tuned && (*tuned)[i * 4].nt) ? (*tuned)[i * 4]
                                                      : conv_make_tile(o.plan[0], N, H >> ti.ds_log2, W >> ti.ds_log2);
    if (t.kind != 2) {
      plane[d.in_t] = plane[d.out_t] = 0;
      if (d.res_t >= 0) plane[d.res_t] = 0;
    }
  }
  return plane;
}

extern "C" int rtpe_hrnet_plane_major_tensors(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, int32_t* count) {
  RTPE_REQUIRE(h != nullptr && count != nullptr, "plane_major_tensors: null argument");
  const std::vector<ConvTile>* tuned = nullptr;
  auto it = h->tuned.find(std::make_tuple((int)N, (int)H, (int)W));
  if (it != h->tuned.end()) tuned = &it->second;
  int n = 0;
  for (char c : plane_tensors(h, N, H, W, tuned)) n += c;
  *count = n;
  return RTPE_OK;
}

// Per-op HIP events.  An op absorbed by the fused launch of its predecessor has none (its time is 0), and a fused
// block that is followed at once by another one has none either: a marker packet between two kernels costs ~3 us
// of stream time (3-4 % of this kernel), so a run of consecutive fused blocks is bracketed as a whole and its
// blocks share the interval equally - the per-launch time then agrees with the kernel trace.
static bool op_has_event(const rtpe_hrnet* h, size_t i, bool fused_mode) {
  if (!fused_mode) return true;
  const std::vector<OpState>& ops = h->ops;
  if (ops[i].fuse == 2) return false;
  return !(ops[i].fuse == 1 && i + 2 < ops.size() && ops[i + 2].fuse == 1);
}

template <class Events>
static int read_op_times(const rtpe_hrnet* h, const Events& ev, bool fused_mode, float* op_ms) {
  const std::vector<OpState>& ops = h->ops;
  for (size_t i = 0; i < ops.size(); ++i) {
    if (fused_mode && ops[i].fuse == 2) { op_ms[i] = 0.f; continue; }
    size_t first = i, last = i;
    if (fused_mode && ops[i].fuse == 1) {
      while (first >= 2 && ops[first - 2].fuse == 1) first -= 2;
      while (last + 2 < ops.size() && ops[last + 2].fuse == 1) last += 2;
    }
    // the interval starts at the event of the op before `first` (a run never starts behind an absorbed op; any
    // other op that follows an absorbed one starts at the event of that block's head)
    const size_t b0 = (fused_mode && first > 0 && ops[first - 1].fuse == 2) ? first - 1 : first;
    float ms;
    RTPE_HIP_CHECK(hipEventElapsedTime(&ms, ev[b0], ev[last + 1]));
    op_ms[i] = ms / (float)((last - first) / 2 + 1);
  }
  return RTPE_OK;
}

static int run(rtpe_hrnet* h, const void* x, int x_dtype, int N, int H, int W, void* preds, void* refined,
               int out_dtype, void* ws, size_t ws_bytes, hipStream_t s_main, float* op_ms, int n_ms,
               int only_op = -1, int only_k = -1, const ConvTile* force = nullptr,
               std::vector<hipEvent_t>* rec = nullptr, const void* aux = nullptr, uint32_t fwd_flags = 0) {
  RTPE_REQUIRE(h && x && ws, "forward: null argument");
  static const int host_prof = env_int("RTPE_HOST_PROF", 0);       // host time of a forward: where it goes (stderr)
  typedef std::chrono::steady_clock hclock;
  const hclock::time_point hp0 = hclock::now();
  double hp_launch = 0.0;
#define RTPE_HP_LAUNCH(stmt)                                                                   \
  do {                                                                                         \
    if (host_prof) {                                                                           \
      const hclock::time_point t_ = hclock::now();                                             \
      stmt;                                                                                    \
      hp_launch += std::chrono::duration<double, std::micro>(hclock::now() - t_).count();      \
    } else {                                                                                   \
      stmt;                                                                                    \
    }                                                                                          \
  } while (0)
  // kernels, events and function attributes act on HIP's CURRENT device: make the handle's device current for
  // the call (the stream and every buffer must belong to it) and give the caller's device back afterwards
  DeviceGuard guard(h->device);
  RTPE_HIP_CHECK(guard.err);
  RTPE_REQUIRE(N > 0 && H % 32 == 0 && W % 32 == 0 && H >= 32 && W >= 32, "forward: N=%d H=%d W=%d", N, H, W);
  RTPE_REQUIRE(x_dtype == RTPE_DTYPE_F16 || x_dtype == RTPE_DTYPE_F32, "forward: x dtype");
  RTPE_REQUIRE(out_dtype == RTPE_DTYPE_F16 || out_dtype == RTPE_DTYPE_F32, "forward: out dtype");
  std::vector<size_t> offs;
  size_t need;
  slot_layout(h, N, H, W, &offs, &need);
  if (need > ws_bytes) { set_error("forward: workspace %zu < %zu", ws_bytes, need); return RTPE_E_NOMEM; }
  RTPE_REQUIRE(((uintptr_t)ws & 255) == 0, "forward: workspace must be 256-byte aligned");
  char* base = reinterpret_cast<char*>(ws);
  auto esz = [&](int t) -> size_t { return h->tensors[t].reserved == 4 ? 4 : 2; };
  auto tptr = [&](int t, int coff) -> _Float16* {      // element type per tensor (fp16 or fp32): byte arithmetic
    return reinterpret_cast<_Float16*>(base + offs[h->tensors[t].slot] + (size_t)coff * esz(t));
  };
  std::vector<hipEvent_t> ev;
  const bool timed = op_ms != nullptr;
  if (timed) {
    RTPE_REQUIRE(n_ms >= (int)h->ops.size(), "forward_timed: op_ms too small");
    ev.resize(h->ops.size() + 1);
    for (auto& e : ev) RTPE_HIP_CHECK(hipEventCreate(&e));
    RTPE_HIP_CHECK(hipEventRecord(ev[0], s_main));
  }
  if (rec) {                         // non-blocking recording into caller-kept events
    if (rec->size() != h->ops.size() + 1) {
      for (auto& e : *rec) hipEventDestroy(e);
      rec->resize(h->ops.size() + 1);
      for (auto& e : *rec) RTPE_HIP_CHECK(hipEventCreate(&e));
    }
    RTPE_HIP_CHECK(hipEventRecord((*rec)[0], s_main));
  }
  const std::vector<ConvTile>* tuned = nullptr;
  {
    auto it = h->tuned.find(std::make_tuple(N, H, W));
    if (it != h->tuned.end()) tuned = &it->second;
  }
  std::vector<char> plane(h->tensors.size(), 0);
  if (only_op < 0 && force == nullptr) plane = plane_tensors(h, N, H, W, tuned);
  // Parallel regions (the branches of a HighResolutionModule, pose_higher_hrnet.py:242-243, and the conversion convs of
  // its fuse layers are independent): with lanes on, the ops of lane k > 0 go to an internal stream that forks from the
  // caller's stream at the region's first op and joins it behind its last; an op waits for the events of the ops of
  // other lanes whose output it reads.  Timed / recorded / single-op runs stay on one stream (one op after another).
  bool lanes_on = false;
  if (h->has_regions && !timed && rec == nullptr && only_op < 0 && force == nullptr && !(fwd_flags & RTPE_FWD_NO_LANES)) {
    const int opt = get_option(kOptLanes);
    lanes_on = opt == 1 || (opt == 2 && (long long)N * H * W <= 4ll * 640 * 640);
  }
  std::unique_lock<std::mutex> lane_lock;
  if (lanes_on) lane_lock = std::unique_lock<std::mutex>(h->lane_mu);   // held until the last region has joined (return)
  if (lanes_on && h->fork_event == nullptr) {
    for (int l = 1; l < 4; ++l) RTPE_HIP_CHECK(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
    h->op_event.assign(h->ops.size() + 4, nullptr);
    for (size_t i = 0; i < h->ops.size(); ++i)
      if (h->needs_event[i]) RTPE_HIP_CHECK(hipEventCreateWithFlags(&h->op_event[i], hipEventDisableTiming));
    for (size_t l = 0; l < 4; ++l) RTPE_HIP_CHECK(hipEventCreateWithFlags(&h->op_event[h->ops.size() + l], hipEventDisableTiming));
    RTPE_HIP_CHECK(hipEventCreateWithFlags(&h->fork_event, hipEventDisableTiming));
  }
  // 1x1 pairs run as one kernel in whole forwards only (timed / recorded ones included: the pair's time is the tail's)
  const bool pairs_on = get_option(kOptPair1x1) != 0 && only_op < 0 && force == nullptr;
  ConvArgs pair_args;
  memset(&pair_args, 0, sizeof(pair_args));
  bool pair_pending = false;                              // the head of a 1x1 pair waits for its tail's launch
  bool stem_pending = false;                              // the fused stem kernel ran at the stem op: the next op (conv2) is done
  int s2_skip = 0;                                        // ops behind the first of a stride-2 group that its launch has done
  // argument block of op j as a plain NHWC conv (conv48s2.hip's layers: no residual, no NCHW output)
  auto s2_layer_args = [&](size_t j, ConvArgs* a) {
    const OpState& oj = h->ops[j];
    const rtpe_op_desc& dj = oj.d;
    const rtpe_tensor_desc& tj = h->tensors[dj.in_t];
    const int Hj = H >> tj.ds_log2, Wj = W >> tj.ds_log2;
    memset(a, 0, sizeof(*a));
    a->x = tptr(dj.in_t, dj.in_coff);
    a->in_ld = tj.channels;
    a->x_bytes = ((size_t)N * Hj * Wj * tj.channels - (size_t)dj.in_coff) * esz(dj.in_t);
    a->w = reinterpret_cast<const _Float16*>(h->arena + oj.w_dev_off[0]);
    a->alpha = reinterpret_cast<const float*>(h->arena + oj.ab_dev_off);
    a->beta = a->alpha + oj.plan[0].cout_pad;
    a->y = tptr(dj.out_t, dj.out_coff);
    a->out_ld = h->tensors[dj.out_t].channels;
    const int room = h->tensors[dj.out_t].channels - dj.out_coff;
    a->cout_store = oj.plan[0].cout_pad < room ? oj.plan[0].cout_pad : room;
    a->N = N; a->H_in = Hj; a->W_in = Wj;
    a->H_full = a->H_pos = Hj / dj.stride; a->W_full = a->W_pos = Wj / dj.stride;
    a->o_mul = 1;
    a->relu = (dj.flags & RTPE_F_RELU) ? 1 : 0;
    a->round_conv = (dj.flags & RTPE_F_ROUND_CONV) ? 1 : 0;
    a->cin = dj.cin; a->cout = dj.cout;
    a->lo_y = oj.plan[0].lo_y; a->lo_x = oj.plan[0].lo_x;
    a->in_cs = oj.plan[0].cc; a->out_cs = oj.plan[0].mt * 16; a->res_cs = oj.plan[0].mt * 16;
  };
  ConvArgs s2_args[3];
  const ConvPlan* s2_plans[3];
  int cur_region = 0;
  bool lane_used[4] = {false, false, false, false};
  auto join_lanes = [&]() -> hipError_t {
    for (int l = 1; l < 4; ++l) {
      if (!lane_used[l]) continue;
      hipEvent_t e = h->op_event[h->ops.size() + l];
      hipError_t er = hipEventRecord(e, h->lane_stream[l]);
      if (er == hipSuccess) er = hipStreamWaitEvent(s_main, e, 0);
      if (er != hipSuccess) return er;
      lane_used[l] = false;
    }
    return hipSuccess;
  };
  for (size_t i = 0; i < h->ops.size(); ++i) {
    if (only_op >= 0 && (int)i != only_op) continue;
    const OpState& o = h->ops[i];
    const rtpe_op_desc& d = o.d;
    int rc = RTPE_OK;
    hipStream_t s = s_main;
    if (lanes_on) {
      if (d.region != cur_region) {
        if (cur_region > 0) RTPE_HIP_CHECK(join_lanes());
        cur_region = d.region;
        if (cur_region > 0) {                            // fork: every lane starts behind what the caller's stream holds
          RTPE_HIP_CHECK(hipEventRecord(h->fork_event, s_main));
          for (int l = 1; l < 4; ++l) RTPE_HIP_CHECK(hipStreamWaitEvent(h->lane_stream[l], h->fork_event, 0));
        }
      }
      if (cur_region > 0) {
        if (d.lane > 0) { s = h->lane_stream[d.lane]; lane_used[d.lane] = true; }
        for (int j : h->wait_ops[i]) RTPE_HIP_CHECK(hipStreamWaitEvent(s, h->op_event[j], 0));
        if ((o.fuse == 1 || o.stem2 == 1) && i + 1 < h->ops.size())   // the block's second conv (the stem's conv2) is launched with this one
          for (int j : h->wait_ops[i + 1]) RTPE_HIP_CHECK(hipStreamWaitEvent(s, h->op_event[j], 0));
        for (int m = 1; m < o.s2g; ++m)                               // the other convs of a stride-2 group are launched with this one
          for (int j : h->wait_ops[i + m]) RTPE_HIP_CHECK(hipStreamWaitEvent(s, h->op_event[j], 0));
      }
    }
    const bool stem_fused_on = force == nullptr && only_op < 0 && get_option(kOptFusedStem) == 1 && stem_fused_supports(H, W);
    if (d.kind == RTPE_OP_STEM && o.stem2 == 1 && stem_fused_on) {
      // conv1 + bn1 + relu + conv2 + bn2 + relu in one kernel: the half-resolution map stays in LDS (stem_fused.hip)
      const OpState& o2 = h->ops[i + 1];
      const rtpe_op_desc& d2 = o2.d;
      StemFusedArgs a;
      memset(&a, 0, sizeof(a));
      a.x = x; a.x_f32 = x_dtype == RTPE_DTYPE_F32;
      a.w1 = reinterpret_cast<const _Float16*>(h->arena + o.w_dev_off[0]);
      a.alpha1 = reinterpret_cast<const float*>(h->arena + o.ab_dev_off);
      a.beta1 = a.alpha1 + 64;
      a.w2 = reinterpret_cast<const _Float16*>(h->arena + o2.w_dev_off[0]);
      a.alpha2 = reinterpret_cast<const float*>(h->arena + o2.ab_dev_off);
      a.beta2 = a.alpha2 + o2.plan[0].cout_pad;
      a.y = tptr(d2.out_t, d2.out_coff);
      a.N = N; a.H = H; a.W = W; a.out_ld = h->tensors[d2.out_t].channels;
      a.relu = (d2.flags & RTPE_F_RELU) ? 1 : 0;
      a.round_conv = (d2.flags & RTPE_F_ROUND_CONV) ? 1 : 0;
      RTPE_HP_LAUNCH(rc = stem_fused_launch(a, s));
      stem_pending = true;      // latched here: the option is process-wide and may change between this op and the next
    } else if (d.kind == RTPE_OP_CONV && o.stem2 == 2 && stem_pending) {
      stem_pending = false;     // done by the launch at the stem op
    } else if (d.kind == RTPE_OP_STEM && !(d.flags & RTPE_F_F32) && get_option(kOptFusedStem) == 2 && stem_fused_supports(H, W)) {
      // option "fused_stem" = 2: the stem op alone on the fused kernel's conv1 code (the chain on the matrix pipe),
      // output to memory - the bit-identity test of that chain against the VALU kernel
      StemFusedArgs a;
      memset(&a, 0, sizeof(a));
      a.x = x; a.x_f32 = x_dtype == RTPE_DTYPE_F32;
      a.w1 = reinterpret_cast<const _Float16*>(h->arena + o.w_dev_off[0]);
      a.alpha1 = reinterpret_cast<const float*>(h->arena + o.ab_dev_off);
      a.beta1 = a.alpha1 + 64;
      a.y1 = tptr(d.out_t, d.out_coff);
      a.N = N; a.H = H; a.W = W; a.out_ld = h->tensors[d.out_t].channels;
      RTPE_HP_LAUNCH(rc = stem_fused_launch(a, s));
    } else if (d.kind == RTPE_OP_STEM) {
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      StemArgs a;
      a.x = x; a.x_f32 = x_dtype == RTPE_DTYPE_F32;
      a.w = reinterpret_cast<const _Float16*>(h->arena + o.w_dev_off[0]);
      a.alpha = reinterpret_cast<const float*>(h->arena + o.ab_dev_off);
      a.beta = a.alpha + 64;
      a.y = tptr(d.out_t, d.out_coff);
      a.N = N; a.H = H; a.W = W; a.out_ld = to.channels;
      a.f32 = (d.flags & RTPE_F_F32) ? 1 : 0;
      RTPE_HP_LAUNCH(rc = stem_launch(a, s));
    } else if (d.kind == RTPE_OP_CONV && o.fuse == 2 && force == nullptr && only_op < 0 && fused_block_runs(h, d, N, H, W)) {
      // second conv of a fused BasicBlock: done by the launch of its head (same test as there: a map too
      // small for the fused kernel runs both convs on their own)
    } else if (d.kind == RTPE_OP_CONV && o.fuse == 1 && force == nullptr && only_op < 0 && fused_block_runs(h, d, N, H, W)) {
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
      const OpState& o2 = h->ops[i + 1];
      const rtpe_tensor_desc& to = h->tensors[o2.d.out_t];
      {
        RTPE_HP_LAUNCH(rc = conv_block_launch(tptr(d.in_t, d.in_coff), ti.channels,
                               ((size_t)N * Hi * Wi * ti.channels - (size_t)d.in_coff) * 2,
                               tptr(o2.d.out_t, o2.d.out_coff), to.channels,
                               reinterpret_cast<const _Float16*>(h->arena + o.w_dev_off[0]),
                               reinterpret_cast<const float*>(h->arena + o.ab_dev_off),
                               reinterpret_cast<const _Float16*>(h->arena + o2.w_dev_off[0]),
                               reinterpret_cast<const float*>(h->arena + o2.ab_dev_off), N, Hi, Wi, s));
      }
    } else if (d.kind == RTPE_OP_CONV && o.s2g < 0 && s2_skip > 0) {
      --s2_skip;                // done by the launch at the group's first op
    } else if (d.kind == RTPE_OP_CONV && o.s2g > 1 && force == nullptr && only_op < 0 && get_option(kOptConv48s2) != 0 && [&]() {
                 for (int m = 0; m < o.s2g; ++m) {
                   const OpState& om = h->ops[i + m];
                   if (plane[om.d.in_t] || plane[om.d.out_t]) return false;
                   s2_layer_args(i + m, &s2_args[m]);
                   s2_plans[m] = &om.plan[0];
                   if (!conv48s2_supports(om.plan[0], s2_args[m])) return false;
                 }
                 return true;
               }()) {
      // the first downsampling convs of a fuse layer's chains from one branch: one launch, the input read once
      RTPE_HP_LAUNCH(rc = conv48s2_launch_group(s2_plans, s2_args, o.s2g, s));
      s2_skip = o.s2g - 1;
    } else if (d.kind == RTPE_OP_CONV || d.kind == RTPE_OP_DECONV) {
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
      const bool dc = d.kind == RTPE_OP_DECONV;
      const int Ho = dc ? Hi * 2 : Hi / d.stride, Wo = dc ? Wi * 2 : Wi / d.stride;
      ConvArgs merged;
      ConvTile merged_tile;
      memset(&merged, 0, sizeof(merged));
      memset(&merged_tile, 0, sizeof(merged_tile));
      static const int merge_deconv = env_int("RTPE_DECONV_MERGE", 1);
      // the 4 sub-pixel classes of a transposed conv run as ONE grid (conv_mfma.hip) on class 0's launch shape
      // when that shape is a one-workgroup-per-tile one and the plans of all classes agree (they differ in the
      // tap offsets and the packed weights only): decided once, before any class is set up
      ConvTile tile0;
      memset(&tile0, 0, sizeof(tile0));
      bool merge = false;
      if (dc && only_k < 0 && merge_deconv) {
        if (force) tile0 = *force;
        else if (tuned && (*tuned)[i * 4].nt) tile0 = (*tuned)[i * 4];
        else tile0 = conv_make_tile(o.plan[0], N, Hi, Wi);
        merge = tile0.kind == 0;
        for (int k = 1; k < o.n_geom && merge; ++k) {
          const ConvPlan &p0 = o.plan[0], &pk = o.plan[k];
          merge = pk.mt == p0.mt && pk.cc == p0.cc && pk.kc == p0.kc && pk.n_cchunks == p0.n_cchunks &&
                  pk.n_cb == p0.n_cb && pk.pstride == p0.pstride && pk.tapw == p0.tapw;
        }
      }
      for (int k = 0; k < o.n_geom && rc == RTPE_OK; ++k) {
        if (only_k >= 0 && k != only_k) continue;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x = tptr(d.in_t, d.in_coff);
        a.in_ld = ti.channels;
        if (plane[d.in_t]) { a.in_ld = 48; a.in_cs = (long long)N * Hi * Wi * 48; }
        a.x_bytes = ((size_t)N * Hi * Wi * ti.channels - (size_t)d.in_coff) * esz(d.in_t);
        a.w = reinterpret_cast<const _Float16*>(h->arena + o.w_dev_off[k]);
        a.alpha = reinterpret_cast<const float*>(h->arena + o.ab_dev_off);
        a.beta = a.alpha + o.plan[0].cout_pad;
        if (d.res_t >= 0) {
          a.res = tptr(d.res_t, d.res_coff);
          a.res_ld = h->tensors[d.res_t].channels;
          if (plane[d.res_t]) { a.res_ld = 48; a.res_cs = (long long)N * Ho * Wo * 48; }
        }
        if (!(d.flags & RTPE_F_NO_NHWC)) {
          a.y = tptr(d.out_t, d.out_coff);
          a.out_ld = h->tensors[d.out_t].channels;
          if (plane[d.out_t]) { a.out_ld = 48; a.out_cs = (long long)N * Ho * Wo * 48; }
          // zero-padded channels up to the allocated row are written too (they
          // are exact zeros: zero weights, zero affine) so that a consumer that
          // reads the padded view sees finite data
          int cs = o.plan[0].cout_pad;
          const int room = h->tensors[d.out_t].channels - d.out_coff;
          a.cout_store = cs < room ? cs : room;
          if (d.reserved[2] > 0 && d.reserved[2] < a.cout_store) a.cout_store = d.reserved[2];
        }
        if (d.flags & RTPE_F_OUT_PREDS) { a.y_nchw = preds; a.nchw_channels = d.cout; }
        if (d.flags & RTPE_F_OUT_REFINED) { a.y_nchw = refined; a.nchw_channels = d.cout; }
        a.nchw_f32 = out_dtype == RTPE_DTYPE_F32;
        if (a.y_nchw == nullptr && (d.flags & (RTPE_F_OUT_PREDS | RTPE_F_OUT_REFINED))) {
          set_error("forward: output pointer missing"); return RTPE_E_INVALID;
        }
        a.N = N; a.H_in = Hi; a.W_in = Wi;
        a.H_full = Ho; a.W_full = Wo;
        if (dc) {
          a.H_pos = Hi; a.W_pos = Wi; a.o_mul = 2; a.oy_add = k >> 1; a.ox_add = k & 1;
        } else {
          a.H_pos = Ho; a.W_pos = Wo; a.o_mul = 1;
        }
        a.relu = (d.flags & RTPE_F_RELU) ? 1 : 0;
        a.round_conv = (d.flags & RTPE_F_ROUND_CONV) ? 1 : 0;
        ConvTile tile;
        if (merge)
          tile = tile0;
        else if (force)
          tile = *force;
        else if (tuned && (*tuned)[i * 4 + k].nt)
          tile = (*tuned)[i * 4 + k];
        else
          tile = conv_make_tile(o.plan[k], N, a.H_pos, a.W_pos);
        // the direct 1x1 kernel and the 1x1 pair address their input through one 2-GiB buffer window and write plain
        // NHWC rows: a larger view (batch >= 164 at 640 x 640 for the 256 -> 64 conv) or another output form takes the
        // one-workgroup-per-tile kernel, whose staging falls back per image
        const bool direct_ok = a.x_bytes < 0x80000000ull && a.y != nullptr && a.y_nchw == nullptr && a.o_mul == 1;
        if (tile.kind == 4 && !direct_ok) tile = conv_make_tile(o.plan[k], N, a.H_pos, a.W_pos, /*allow_direct=*/false);
        // the persistent 64 -> 64 kernel (conv64.hip) takes no residual and writes plain NHWC rows through one buffer window
        // (also when a tuned or imported shape says kind 5 but option "conv64" has been switched off since)
        if (tile.kind == 5 && !(direct_ok && a.res == nullptr && !plane[d.in_t] && !plane[d.out_t] && get_option(kOptConv64) != 0))
          tile = conv_make_tile(o.plan[k], N, a.H_pos, a.W_pos, true, /*allow_conv64=*/false);
        conv_fill_args(o.geom[k], o.plan[k], tile, &a);
        // the heads (1x1, 48 input channels, fp32 NCHW out): the direct scheme with an NCHW epilogue (conv_direct.hip),
        // whatever launch shape was chosen or tuned for the layer (option "head_direct")
        if (!merge && force == nullptr && get_option(kOptHeadDirect) != 0 && conv_head_supports(o.plan[k], a)) {
          RTPE_HP_LAUNCH(rc = conv_head_launch(o.plan[k], a, s));
          continue;
        }
        // the downsampling convs of the fuse layers from 48 input channels: persistent workgroups with register-resident weights
        // (conv48s2.hip, option "conv48s2"), whatever launch shape was chosen or tuned for the layer
        if (!merge && force == nullptr && get_option(kOptConv48s2) != 0 && !plane[d.in_t] && !(a.y != nullptr && plane[d.out_t]) &&
            conv48s2_supports(o.plan[k], a)) {
          RTPE_HP_LAUNCH(rc = conv48s2_launch(o.plan[k], a, s));
          continue;
        }
        if (merge) {
          // class k's weights and offsets go into the argument block of class 0; the launch follows the last class
          if (k == 0) { merged = a; merged_tile = tile; merged.n_cls = 4; }
          merged.w_c[k] = a.w;
          merged.lo_yc[k] = a.lo_y; merged.lo_xc[k] = a.lo_x;
          merged.oy_c[k] = a.oy_add; merged.ox_c[k] = a.ox_add;
          if (k == o.n_geom - 1) {
            // the four classes on one persistent kernel that shares their halo tiles (deconv48.hip, option "deconv48")
            if (force == nullptr && get_option(kOptDeconv48) != 0 && deconv48_supports(o.plan[0], merged))
              RTPE_HP_LAUNCH(rc = deconv48_launch(o.plan[0], merged, s));
            else
              RTPE_HP_LAUNCH(rc = conv_launch(o.plan[0], merged_tile, merged, s));
          }
          continue;
        }
        if (pairs_on && o.pair == 1 && direct_ok) {       // launched together with the next op (conv_pair.hip)
          pair_args = a;
          pair_pending = true;
          continue;
        }
        if (pair_pending) {
          pair_pending = false;
          RTPE_HP_LAUNCH(rc = conv_pair_launch(h->ops[i - 1].plan[0], pair_args, o.plan[0], a, s));
          continue;
        }
        RTPE_HP_LAUNCH(rc = conv_launch(o.plan[k], tile, a, s));
      }
    } else if (d.kind == RTPE_OP_AUX_PACK) {
      if (aux == nullptr) { set_error("forward: this program has a second input (use rtpe_hrnet_forward_aux)"); return RTPE_E_INVALID; }
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      rc = aux_pack_launch(reinterpret_cast<const float*>(aux), reinterpret_cast<float*>(tptr(d.out_t, d.out_coff)),
                           to.channels, N, H >> to.ds_log2, W >> to.ds_log2, s);
    } else if (d.kind == RTPE_OP_RESIZE) {
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      rc = resize_nhwc_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels, H >> ti.ds_log2,
                              W >> ti.ds_log2, reinterpret_cast<float*>(tptr(d.out_t, d.out_coff)), to.channels,
                              H >> to.ds_log2, W >> to.ds_log2, d.cout, N, s);
    } else if (d.kind == RTPE_OP_GATE_MUL) {
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      const size_t pixels = (size_t)N * (H >> ti.ds_log2) * (W >> ti.ds_log2);
      float* att_out = (d.flags & RTPE_F_OUT_PREDS) ? reinterpret_cast<float*>(preds) : nullptr;
      if ((d.flags & RTPE_F_OUT_PREDS) && (!preds || out_dtype != RTPE_DTYPE_F32)) {
        set_error("forward: the sigmoid map output must be a float32 buffer"); return RTPE_E_INVALID;
      }
      float div;
      memcpy(&div, &d.reserved[0], sizeof(float));
      rc = gate_mul_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels,
                           reinterpret_cast<const float*>(tptr(d.res_t, d.res_coff)), h->tensors[d.res_t].channels,
                           reinterpret_cast<float*>(tptr(d.out_t, d.out_coff)), to.channels, d.cout, pixels, div, att_out, s);
    } else if (d.kind == RTPE_OP_CAST || d.kind == RTPE_OP_AVGPOOL || d.kind == RTPE_OP_SE ||
               d.kind == RTPE_OP_CAM_COMBINE || d.kind == RTPE_OP_SIGMOID_ADD) {
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
      const size_t pixels = (size_t)N * Hi * Wi;
      float* yo = reinterpret_cast<float*>(tptr(d.out_t, d.out_coff));
      if (d.kind == RTPE_OP_CAST) {
        rc = cast_launch(tptr(d.in_t, d.in_coff), ti.channels, yo, to.channels, d.cout, pixels, s);
      } else if (d.kind == RTPE_OP_AVGPOOL) {
        rc = avgpool_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels, yo, to.channels,
                            d.cout, N, Hi, Wi, s);
      } else if (d.kind == RTPE_OP_SE) {
        rc = se_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels, d.cin, d.cout, N, Hi * Wi,
                       reinterpret_cast<const float*>(h->arena + o.w_dev_off[0]), yo, to.channels, s);
      } else if (d.kind == RTPE_OP_CAM_COMBINE) {
        rc = cam_combine_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels,
                                reinterpret_cast<const float*>(tptr(d.res_t, d.res_coff)), h->tensors[d.res_t].channels,
                                reinterpret_cast<const float*>(tptr(d.term_t[0], 0)), h->tensors[d.term_t[0]].channels,
                                yo, to.channels, d.cout, N, (size_t)Hi * Wi, s);
      } else {
        float* att_out = (d.flags & RTPE_F_OUT_PREDS) ? reinterpret_cast<float*>(preds) : nullptr;
        if ((d.flags & RTPE_F_OUT_PREDS) && (!preds || out_dtype != RTPE_DTYPE_F32)) {
          set_error("forward: the sigmoid map output must be a float32 buffer"); return RTPE_E_INVALID;
        }
        rc = sigmoid_add_launch(reinterpret_cast<const float*>(tptr(d.in_t, d.in_coff)), ti.channels,
                                reinterpret_cast<const float*>(tptr(d.res_t, d.res_coff)), h->tensors[d.res_t].channels,
                                yo, to.channels, d.cout, pixels, att_out, s);
      }
    } else {  // FUSE
      const rtpe_tensor_desc& to = h->tensors[d.out_t];
      FuseArgs a;
      memset(&a, 0, sizeof(a));
      a.n_terms = d.n_terms;
      for (int t = 0; t < d.n_terms; ++t) {
        a.term[t] = tptr(d.term_t[t], 0);
        a.term_ld[t] = h->tensors[d.term_t[t]].channels;
        a.term_up[t] = d.term_up[t];
      }
      a.y = tptr(d.out_t, d.out_coff);
      a.out_ld = to.channels; a.C = d.cout;
      a.N = N; a.H = H >> to.ds_log2; a.W = W >> to.ds_log2;
      a.f32 = (d.flags & RTPE_F_F32) ? 1 : 0;
      a.relu = (d.flags & RTPE_F_RELU) ? 1 : 0;
      RTPE_HP_LAUNCH(rc = fuse_launch(a, s));
    }
    if (rc != RTPE_OK) return rc;
    if (lanes_on && cur_region > 0) {
      // (a 1x1 pair is ONE kernel, launched at the tail's place: the head's output exists behind that launch)
      if (h->needs_event[i] && !pair_pending) RTPE_HIP_CHECK(hipEventRecord(h->op_event[i], s));
      if (pairs_on && o.pair == 2 && h->needs_event[i - 1]) RTPE_HIP_CHECK(hipEventRecord(h->op_event[i - 1], s));
    }
    const bool has_event = op_has_event(h, i, force == nullptr && only_op < 0);
    if (timed && has_event) RTPE_HIP_CHECK(hipEventRecord(ev[i + 1], s));
    if (rec && has_event) RTPE_HIP_CHECK(hipEventRecord((*rec)[i + 1], s));
  }
  if (lanes_on && cur_region > 0) RTPE_HIP_CHECK(join_lanes());
  if (host_prof) {
    static double sum_total = 0.0, sum_launch = 0.0;
    static int n_calls = 0;
    sum_total += std::chrono::duration<double, std::micro>(hclock::now() - hp0).count();
    sum_launch += hp_launch;
    if (++n_calls % host_prof == 0) {
      fprintf(stderr, "rtpe host profile (N=%d, %zu ops, lanes %d): %.0f us per forward on the host, %.0f us of it inside the kernel "
              "launch helpers (mean of %d calls)\n", N, h->ops.size(), (int)lanes_on, sum_total / n_calls, sum_launch / n_calls, n_calls);
    }
  }
#undef RTPE_HP_LAUNCH
  if (timed) {
    RTPE_HIP_CHECK(hipEventSynchronize(ev.back()));
    const int rc2 = read_op_times(h, ev, force == nullptr && only_op < 0, op_ms);
    for (auto& e : ev) hipEventDestroy(e);
    if (rc2 != RTPE_OK) return rc2;
  }
  return RTPE_OK;
}

extern "C" int rtpe_hrnet_forward(rtpe_hrnet* h, const void* x, int32_t x_dtype, int32_t N, int32_t H,
                                  int32_t W, void* preds, void* refined, int32_t out_dtype, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), nullptr, 0);
}

extern "C" int rtpe_hrnet_forward_flags(rtpe_hrnet* h, const void* x, int32_t x_dtype, int32_t N, int32_t H,
                                        int32_t W, void* preds, void* refined, int32_t out_dtype, void* workspace,
                                        size_t workspace_bytes, void* stream, uint32_t flags) {
  RTPE_REQUIRE((flags & ~(uint32_t)RTPE_FWD_NO_LANES) == 0, "forward_flags: unknown flag bits 0x%x", flags);
  return run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), nullptr, 0, -1, -1, nullptr, nullptr, nullptr, flags);
}

extern "C" int rtpe_hrnet_forward_aux(rtpe_hrnet* h, const void* x, int32_t x_dtype, const void* aux_nchw_f32, int32_t N,
                                      int32_t H, int32_t W, void* preds, void* refined, int32_t out_dtype,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  RTPE_REQUIRE(aux_nchw_f32 != nullptr, "forward_aux: the second input is null");
  return run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), nullptr, 0, -1, -1, nullptr, nullptr, aux_nchw_f32);
}

extern "C" int rtpe_hrnet_forward_timed(rtpe_hrnet* h, const void* x, int32_t x_dtype, int32_t N, int32_t H,
                                        int32_t W, void* preds, void* refined, int32_t out_dtype,
                                        void* workspace, size_t workspace_bytes, void* stream, float* op_ms,
                                        int32_t n_ops) {
  RTPE_REQUIRE(op_ms != nullptr, "forward_timed: op_ms is null");
  return run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), op_ms, n_ops);
}

extern "C" int rtpe_hrnet_op_cost(const rtpe_hrnet* h, int32_t op, int32_t N, int32_t H, int32_t W,
                                  double* flops, double* bytes) {
  RTPE_REQUIRE(h && op >= 0 && op < (int)h->ops.size() && flops && bytes, "op_cost: bad argument");
  const rtpe_op_desc& d = h->ops[op].d;
  *flops = 0; *bytes = 0;
  if (d.kind == RTPE_OP_FUSE) {
    const rtpe_tensor_desc& to = h->tensors[d.out_t];
    const double px = (double)N * (H >> to.ds_log2) * (W >> to.ds_log2);
    *bytes = px * d.cout * 2;
    for (int t = 0; t < d.n_terms; ++t) *bytes += px / (double)(1 << (2 * d.term_up[t])) * d.cout * 2;
    return RTPE_OK;
  }
  const int cin_logical = d.reserved[0] > 0 ? d.reserved[0] : d.cin;
  if (d.kind >= RTPE_OP_CAST) {                      // small fp32 student ops: bytes only
    const rtpe_tensor_desc& ti = h->tensors[d.kind == RTPE_OP_AUX_PACK ? d.out_t : d.in_t];
    const double px = (double)N * (H >> ti.ds_log2) * (W >> ti.ds_log2);
    *bytes = px * ti.channels * 4 * 2;
    return RTPE_OK;
  }
  if (d.kind == RTPE_OP_STEM) {
    const double po = (double)N * (H / 2) * (W / 2);
    *flops = 2.0 * po * 64 * 27;
    *bytes = (double)N * 3 * H * W * 4 + po * 64 * 2;
    return RTPE_OK;
  }
  const rtpe_tensor_desc& ti = h->tensors[d.in_t];
  const double pi = (double)N * (H >> ti.ds_log2) * (W >> ti.ds_log2);
  const bool dc = d.kind == RTPE_OP_DECONV;
  const double po = dc ? pi * 4 : pi / (d.stride * d.stride);
  const double taps = dc ? 4 : d.ksize * d.ksize;
  const double es = (d.flags & RTPE_F_F32) ? 4.0 : 2.0;       // bytes per element of this op's tensors
  *flops = 2.0 * po * d.cout * cin_logical * taps;
  *bytes = pi * cin_logical * es + po * d.cout * es + (d.res_t >= 0 ? po * d.cout * es : 0) +
           (double)cin_logical * d.cout * d.ksize * d.ksize * es;
  return RTPE_OK;
}

// ---- single-layer entry (layer-level parity tests) -------------------------
extern "C" int rtpe_conv2d_nhwc_ex(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin,
                                   const void* w_host, const float* alpha_host, const float* beta_host,
                                   int32_t cout, int32_t ksize, int32_t stride, int32_t dilation, int32_t flags,
                                   const void* res, void* y, void* stream) {
  RTPE_REQUIRE(x && w_host && alpha_host && beta_host && y, "conv2d_nhwc: null argument");
  const int es = (flags & RTPE_F_F32) ? 4 : 2, eps = 16 / es;
  RTPE_REQUIRE((ksize == 1 || ksize == 3 || ksize == 5) && (stride == 1 || stride == 2) && cin % eps == 0 && cout % eps == 0 &&
                   dilation >= 1 && (dilation == 1 || (ksize == 3 && stride == 1)),
               "conv2d_nhwc: k=%d s=%d d=%d cin=%d cout=%d unsupported", ksize, stride, dilation, cin, cout);
  RTPE_REQUIRE(H % stride == 0 && W % stride == 0, "conv2d_nhwc: H, W must be multiples of the stride");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ConvGeom g{cin, cout, ksize, stride, -1, es, dilation};
  ConvPlan p = conv_make_plan(g);
  std::vector<char> packed(p.packed_bytes);
  conv_pack_weights(g, p, w_host, packed.data());
  std::vector<float> ab(2 * p.cout_pad, 0.f);
  for (int c = 0; c < cout; ++c) { ab[c] = alpha_host[c]; ab[p.cout_pad + c] = beta_host[c]; }
  char* dev = nullptr;
  const size_t wb = align_up(p.packed_bytes, 256);
  RTPE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), wb + ab.size() * 4));
  hipError_t e = hipMemcpy(dev, packed.data(), p.packed_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dev + wb, ab.data(), ab.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(dev); return hip_fail(e, "hipMemcpy", __FILE__, __LINE__); }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = reinterpret_cast<const _Float16*>(x); a.in_ld = cin;
  a.x_bytes = (size_t)N * H * W * cin * es;
  a.w = reinterpret_cast<const _Float16*>(dev);
  a.alpha = reinterpret_cast<const float*>(dev + wb); a.beta = a.alpha + p.cout_pad;
  a.res = reinterpret_cast<const _Float16*>(res); a.res_ld = cout;
  a.y = reinterpret_cast<_Float16*>(y); a.out_ld = cout; a.cout_store = cout;
  a.N = N; a.H_in = H; a.W_in = W;
  a.H_full = a.H_pos = H / stride; a.W_full = a.W_pos = W / stride; a.o_mul = 1;
  a.relu = (flags & RTPE_F_RELU) ? 1 : 0;
  a.round_conv = (flags & RTPE_F_ROUND_CONV) ? 1 : 0;
  const bool one_window = a.x_bytes < 0x80000000ull;
  const ConvTile tile = conv_make_tile(p, N, a.H_pos, a.W_pos, /*allow_direct=*/one_window, /*allow_conv64=*/one_window && res == nullptr);
  // diagnostic builds, RTPE_PROBE_PLANE=1: time a streaming launch with plane-major views ([C/48][N][H][W][48], what the
  // engine gives the inner tensors of the C >= 96 block chains) over the same bytes - the values are then meaningless
  static const int probe_plane = RTPE_DIAG_ENV_INT("RTPE_PROBE_PLANE", 0);
  if (probe_plane && tile.kind == 2 && cin % 48 == 0 && cout % 48 == 0 && stride == 1) {
    a.in_ld = 48; a.in_cs = (long long)N * H * W * 48;
    a.out_ld = 48; a.out_cs = (long long)N * H * W * 48;
    if (res != nullptr) { a.res_ld = 48; a.res_cs = (long long)N * H * W * 48; }
  }
  conv_fill_args(g, p, tile, &a);
#ifdef RTPE_CONV_STAMPS
  unsigned long long* dbg = nullptr;
  hipMalloc(reinterpret_cast<void**>(&dbg), 128);
  hipMemset(dbg, 0, 128);
  a.dbg = dbg;
#endif
  int rc = get_option(kOptConv48s2) != 0 && conv48s2_supports(p, a) ? conv48s2_launch(p, a, s) : conv_launch(p, tile, a, s);
  hipError_t es2 = hipStreamSynchronize(s);
#ifdef RTPE_CONV_STAMPS
  unsigned long long hd[16];
  hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost);
  hipFree(dbg);
  if (tile.kind == 2 && hd[5] && (hd[11] = hd[11] ? hd[11] : hd[5] / tile.waves)) {
    const unsigned long long nw = (unsigned long long)tile.grid * tile.waves, units = hd[5] / p.n_cchunks;
    fprintf(stderr, "stream conv %dx%d nt%d w%d nb%d nw%d grid %d | per MFMA wave: total %llu cycles, %llu units | per stage: waitM %llu half0 %llu "
            "waitH %llu half1 %llu | per unit: E-wait %llu bn+transpose %llu res-wait %llu rows+add+store %llu load_res %llu | weight loader/stage: wait %llu issue %llu | "
            "tile loaders/stage: wait %llu issue %llu\n",
            tile.th, tile.tw, tile.nt, tile.waves, tile.n_bufs, tile.n_wslots, tile.grid, hd[15] / nw, units / nw, hd[0] / hd[5], hd[1] / hd[5],
            hd[2] / hd[5], hd[3] / hd[5], hd[12] / units, hd[13] / units, hd[4] / units, hd[14] / units, hd[10] / units, hd[6] / hd[11],
            hd[7] / hd[11], hd[8] / hd[11], hd[9] / hd[11]);
  }
  else if (hd[5])
    fprintf(stderr, "conv stamps (kind %d): n %llu | per wave(-unit) cycles: setup %llu stage/wait1 %llu kloop %llu epilogue %llu total/wait2 %llu\n",
            tile.kind, hd[5], hd[0] / hd[5], hd[1] / hd[5], hd[2] / hd[5], hd[3] / hd[5], hd[4] / hd[5]);
  if (hd[10] && tile.kind != 2)
    fprintf(stderr, "   loader per stage: vmcnt-wait %llu barrier1 %llu issue %llu barriersE+2 %llu (stages %llu)\n",
            hd[6] / hd[10], hd[7] / hd[10], hd[8] / hd[10], hd[9] / hd[10], hd[10]);
#endif
  hipFree(dev);
  if (rc != RTPE_OK) return rc;
  if (es2 != hipSuccess) return hip_fail(es2, "hipStreamSynchronize", __FILE__, __LINE__);
  return RTPE_OK;
}

extern "C" int rtpe_basicblock_nhwc(const void* x, int32_t N, int32_t H, int32_t W, const void* w1_host,
                                    const float* alpha1, const float* beta1, const void* w2_host, const float* alpha2,
                                    const float* beta2, void* y, void* stream) {
  RTPE_REQUIRE(x && w1_host && w2_host && alpha1 && beta1 && alpha2 && beta2 && y, "basicblock_nhwc: null argument");
  RTPE_REQUIRE(conv_block_supports(48, 48, H, W), "basicblock_nhwc: H=%d W=%d unsupported", H, W);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ConvGeom g{48, 48, 3, 1, -1, 2, 1};
  ConvPlan p = conv_make_plan(g);
  const size_t wb = align_up(p.packed_bytes, 256), abb = align_up(2 * 48 * sizeof(float), 256);
  std::vector<char> host(2 * (wb + abb), 0);
  const void* ws[2] = {w1_host, w2_host};
  const float* al[2] = {alpha1, alpha2};
  const float* be[2] = {beta1, beta2};
  for (int k = 0; k < 2; ++k) {
    conv_pack_weights(g, p, ws[k], host.data() + k * (wb + abb));
    float* ab = reinterpret_cast<float*>(host.data() + k * (wb + abb) + wb);
    for (int c = 0; c < 48; ++c) { ab[c] = al[k][c]; ab[48 + c] = be[k][c]; }
  }
  char* dev = nullptr;
  RTPE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), host.size()));
  hipError_t e = hipMemcpy(dev, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(dev); return hip_fail(e, "hipMemcpy", __FILE__, __LINE__); }
  int rc = conv_block_launch(reinterpret_cast<const _Float16*>(x), 48, (size_t)N * H * W * 48 * 2,
                             reinterpret_cast<_Float16*>(y), 48, reinterpret_cast<const _Float16*>(dev),
                             reinterpret_cast<const float*>(dev + wb), reinterpret_cast<const _Float16*>(dev + wb + abb),
                             reinterpret_cast<const float*>(dev + wb + abb + wb), N, H, W, s);
  hipError_t es = hipStreamSynchronize(s);
  hipFree(dev);
  if (rc != RTPE_OK) return rc;
  if (es != hipSuccess) return hip_fail(es, "hipStreamSynchronize", __FILE__, __LINE__);
  return RTPE_OK;
}

extern "C" int rtpe_deconv4x4s2_nhwc(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin,
                                     const void* w_host, const float* alpha_host, const float* beta_host,
                                     int32_t cout, int32_t flags, void* y, void* stream) {
  RTPE_REQUIRE(x && w_host && alpha_host && beta_host && y, "deconv4x4s2_nhwc: null argument");
  RTPE_REQUIRE(cin % 8 == 0 && cout % 8 == 0 && N > 0 && H > 0 && W > 0, "deconv4x4s2_nhwc: cin=%d cout=%d", cin, cout);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = RTPE_OK;
  std::vector<char*> dev_bufs;
  ConvArgs merged;
  ConvTile merged_tile;
  memset(&merged, 0, sizeof(merged));
  memset(&merged_tile, 0, sizeof(merged_tile));
  // as in the forward: the four classes in one grid on class 0's launch shape when that is a one-workgroup-per-
  // tile shape and all plans agree (RTPE_DECONV_MERGE=0: four launches); decided before any class is set up
  static const int merge_deconv = env_int("RTPE_DECONV_MERGE", 1);
  ConvPlan plans[4];
  for (int k = 0; k < 4; ++k) plans[k] = conv_make_plan(ConvGeom{cin, cout, 4, 2, k, 2, 1});
  const ConvTile tile0 = conv_make_tile(plans[0], N, H, W);
  bool merge = merge_deconv && tile0.kind == 0;
  for (int k = 1; k < 4 && merge; ++k)
    merge = plans[k].mt == plans[0].mt && plans[k].cc == plans[0].cc && plans[k].kc == plans[0].kc &&
            plans[k].n_cchunks == plans[0].n_cchunks && plans[k].n_cb == plans[0].n_cb &&
            plans[k].pstride == plans[0].pstride && plans[k].tapw == plans[0].tapw;
  for (int k = 0; k < 4 && rc == RTPE_OK; ++k) {         // the four sub-pixel (parity) classes of the output
    ConvGeom g{cin, cout, 4, 2, k, 2, 1};
    const ConvPlan p = plans[k];
    std::vector<char> packed(p.packed_bytes);
    conv_pack_weights(g, p, w_host, packed.data());
    std::vector<float> ab(2 * p.cout_pad, 0.f);
    for (int c = 0; c < cout; ++c) { ab[c] = alpha_host[c]; ab[p.cout_pad + c] = beta_host[c]; }
    char* dev = nullptr;
    const size_t wb = align_up(p.packed_bytes, 256);
    RTPE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), wb + ab.size() * 4));
    dev_bufs.push_back(dev);
    hipError_t e = hipMemcpy(dev, packed.data(), p.packed_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dev + wb, ab.data(), ab.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpy", __FILE__, __LINE__); break; }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const _Float16*>(x); a.in_ld = cin;
    a.x_bytes = (size_t)N * H * W * cin * 2;
    a.w = reinterpret_cast<const _Float16*>(dev);
    a.alpha = reinterpret_cast<const float*>(dev + wb); a.beta = a.alpha + p.cout_pad;
    a.y = reinterpret_cast<_Float16*>(y); a.out_ld = cout; a.cout_store = cout;
    a.N = N; a.H_in = H; a.W_in = W;
    a.H_full = 2 * H; a.W_full = 2 * W;
    a.H_pos = H; a.W_pos = W; a.o_mul = 2; a.oy_add = k >> 1; a.ox_add = k & 1;
    a.relu = (flags & RTPE_F_RELU) ? 1 : 0;
    a.round_conv = (flags & RTPE_F_ROUND_CONV) ? 1 : 0;
    const ConvTile tile = merge ? tile0 : conv_make_tile(p, N, a.H_pos, a.W_pos);
    conv_fill_args(g, p, tile, &a);
    if (merge) {
      if (k == 0) { merged = a; merged_tile = tile; merged.n_cls = 4; }
      merged.w_c[k] = a.w;
      merged.lo_yc[k] = a.lo_y; merged.lo_xc[k] = a.lo_x;
      merged.oy_c[k] = a.oy_add; merged.ox_c[k] = a.ox_add;
      if (k == 3)
        rc = get_option(kOptDeconv48) != 0 && deconv48_supports(plans[0], merged) ? deconv48_launch(plans[0], merged, s)
                                                                                   : conv_launch(plans[0], merged_tile, merged, s);
      continue;
    }
    rc = conv_launch(p, tile, a, s);
  }
  hipError_t es = hipStreamSynchronize(s);
  for (char* d : dev_bufs) hipFree(d);
  if (rc != RTPE_OK) return rc;
  if (es != hipSuccess) return hip_fail(es, "hipStreamSynchronize", __FILE__, __LINE__);
  return RTPE_OK;
}

extern "C" int rtpe_fuse_nhwc(const void* const* terms, const int32_t* term_up, int32_t n_terms, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t flags, void* y, void* stream) {
  RTPE_REQUIRE(terms && term_up && y && n_terms >= 1 && n_terms <= 4, "fuse_nhwc: bad argument");
  FuseArgs a;
  memset(&a, 0, sizeof(a));
  a.n_terms = n_terms;
  for (int t = 0; t < n_terms; ++t) {
    RTPE_REQUIRE(terms[t] != nullptr && term_up[t] >= 0 && term_up[t] < 8 && H % (1 << term_up[t]) == 0 &&
                     W % (1 << term_up[t]) == 0, "fuse_nhwc: term %d", t);
    a.term[t] = reinterpret_cast<const _Float16*>(terms[t]);
    a.term_ld[t] = C;
    a.term_up[t] = term_up[t];
  }
  a.y = reinterpret_cast<_Float16*>(y);
  a.out_ld = C; a.C = C; a.N = N; a.H = H; a.W = W;
  a.f32 = (flags & RTPE_F_F32) ? 1 : 0;
  a.relu = (flags & RTPE_F_RELU) ? 1 : 0;
  return fuse_launch(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rtpe_conv2d_nhwc(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin,
                                const void* w_host, const float* alpha_host, const float* beta_host,
                                int32_t cout, int32_t ksize, int32_t stride, int32_t flags, const void* res,
                                void* y, void* stream) {
  return rtpe_conv2d_nhwc_ex(x, N, H, W, cin, w_host, alpha_host, beta_host, cout, ksize, stride, 1,
                             flags & ~RTPE_F_F32, res, y, stream);
}

// introspection for bench / tuning: kernel variant and geometry of a conv op
extern "C" int rtpe_hrnet_op_tile(const rtpe_hrnet* h, int32_t op, int32_t N, int32_t H, int32_t W, int32_t* out8) {
  RTPE_REQUIRE(h && out8 && op >= 0 && op < (int)h->ops.size(), "op_tile: bad argument");
  const OpState& o = h->ops[op];
  memset(out8, 0, 8 * sizeof(int32_t));
  if (o.stem2 && get_option(kOptFusedStem) == 1 && stem_fused_supports(H, W)) {   // fused stem (stem_fused.hip): 8 x 16 tiles, 8 waves
    out8[0] = 4; out8[1] = 4; out8[2] = 8; out8[3] = 8; out8[4] = 16; out8[5] = 64; out8[6] = 1;
    out8[7] = o.stem2 == 1 ? -600001 : -600002;
    return RTPE_OK;
  }
  if (o.n_geom == 0) return RTPE_OK;
  const rtpe_op_desc& d = o.d;
  const rtpe_tensor_desc& ti = h->tensors[d.in_t];
  const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
  const bool dc = d.kind == RTPE_OP_DECONV;
  const int Hp = dc ? Hi : Hi / d.stride, Wp = dc ? Wi : Wi / d.stride;
  ConvTile t = conv_make_tile(o.plan[0], N, Hp, Wp);
  {
    auto it = h->tuned.find(std::make_tuple(N, H, W));
    if (it != h->tuned.end() && it->second[op * 4].nt) t = it->second[op * 4];
  }
  if (t.kind == 5 && (d.res_t >= 0 || (d.flags & RTPE_F_NO_NHWC))) t = conv_make_tile(o.plan[0], N, Hp, Wp, true, false);   // as run() does
  if ((d.flags & (RTPE_F_OUT_PREDS | RTPE_F_OUT_REFINED)) && d.kind == RTPE_OP_CONV && d.ksize == 1 && d.cin == 48 && d.res_t < 0 &&
      !(d.flags & RTPE_F_F32) && get_option(kOptHeadDirect) != 0 && o.plan[0].n_cb == 1 && (o.plan[0].mt == 2 || o.plan[0].mt == 3) &&
      ((unsigned)Hi * (unsigned)Wi) % 32u == 0) {       // head on the direct scheme (conv_direct.hip): 32 pixels per wave step
    out8[0] = o.plan[0].mt; out8[1] = 2; out8[2] = 4; out8[3] = 1; out8[4] = 32; out8[5] = 48; out8[6] = 1; out8[7] = -400001;
    return RTPE_OK;
  }
  if (dc && get_option(kOptDeconv48) != 0 && t.kind == 0 && d.res_t < 0 && !(d.flags & RTPE_F_NO_NHWC) && !h->plane_ok[d.in_t] &&
      !h->plane_ok[d.out_t] && o.plan[0].mt == 3 && o.plan[0].n_cb == 1 && o.plan[0].cout_pad == 48 && o.plan[0].n_cchunks <= 2) {
    // the four classes on one persistent kernel (deconv48.hip): 8 x 16 input positions per tile, wave k = class k
    out8[0] = 3; out8[1] = 2; out8[2] = 4; out8[3] = 8; out8[4] = 16; out8[5] = 48; out8[6] = 1; out8[7] = -300001;
    return RTPE_OK;
  }
  if (!dc && get_option(kOptConv48s2) != 0 && d.ksize == 3 && d.stride == 2 && d.cin == 48 && d.res_t < 0 && !(d.flags & (RTPE_F_NO_NHWC | RTPE_F_F32)) &&
      !h->plane_ok[d.in_t] && !h->plane_ok[d.out_t] && (d.cout == 48 || d.cout == 96 || d.cout == 192 || d.cout == 384) &&
      o.plan[0].cout_pad == d.cout && Hi % 2 == 0 && Wi % 2 == 0) {
    // persistent stride-2 kernel (conv48s2.hip): 8 x 8 output tiles, a wave = one group of 48 output channels
    const int gw = d.cout == 48 ? 1 : d.cout == 96 ? 2 : 4;
    out8[0] = 3; out8[1] = 2; out8[2] = 4; out8[3] = 8; out8[4] = 8; out8[5] = 48; out8[6] = d.cout / (48 * gw);
    // -200001: a launch of its own; -20000n (n = 2, 3): first of n convs from one input in ONE launch; -200009: one of the others
    out8[7] = o.s2g > 1 ? -(200000 + o.s2g) : o.s2g < 0 ? -200009 : -200001;
    return RTPE_OK;
  }
  if (o.pair && get_option(kOptPair1x1) != 0) {     // 1x1 pair (conv_pair.hip): 16-pixel tiles per wave, 8 waves
    out8[0] = o.pair == 1 ? 16 : 4; out8[1] = 1; out8[2] = 8; out8[3] = 1; out8[4] = 16; out8[5] = o.pair == 1 ? 64 : 256; out8[6] = 1;
    out8[7] = o.pair == 1 ? -800001 : -800002;
    return RTPE_OK;
  }
  if (o.fuse) {       // fused BasicBlock (conv_block.hip): 6x32 tiles, 5 + 3 pixel tiles per wave
    out8[0] = 3; out8[1] = o.fuse == 1 ? 5 : 3; out8[2] = 4; out8[3] = 6; out8[4] = 32; out8[5] = 48; out8[6] = 1;
    out8[7] = o.fuse == 1 ? -900001 : -900002;
    return RTPE_OK;
  }
  out8[0] = t.kind == 0 && t.mrun ? t.mrun : o.plan[0].mt; out8[1] = t.nt; out8[2] = t.waves; out8[3] = t.th; out8[4] = t.tw;
  out8[5] = o.plan[0].cc; out8[6] = o.plan[0].n_cb; out8[7] = t.kind == 2 ? -(t.grid + 100000 * t.n_bufs) : t.kind == 4 ? -(700000 + t.grid) : t.kind == 5 ? -(500000 + t.grid)
                                                            : (int32_t)t.lds_bytes;
  return RTPE_OK;
}

// Per-shape plan autotuning (the reference runs its GPU path with cudnn.benchmark = True,
// teacher_inference.py:31).  Layers that look alike form a class; in round r every class
// runs its r-th launch shape (conv_enum_tiles) inside complete, per-op-timed forward
// passes, so each shape is measured in the cache state it will really see; the shape with
// the smallest summed time wins for the class.  All shapes give bit-identical results.
// Host-returning.
static int autotune_impl(rtpe_hrnet* h, const void* x, int32_t x_dtype, const void* aux, int32_t N, int32_t H, int32_t W,
                         void* preds, void* refined, int32_t out_dtype, void* workspace, size_t workspace_bytes,
                         void* stream) {
  RTPE_REQUIRE(h != nullptr, "autotune: null handle");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const auto shape = std::make_tuple(N, H, W);
  h->tuned.erase(shape);
  const size_t n_ops = h->ops.size();
  std::vector<float> ms(n_ops);
  int rc = run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes, s, ms.data(), (int)n_ops,
               -1, -1, nullptr, nullptr, aux);
  if (rc != RTPE_OK) return rc;

  typedef std::tuple<int, int, int, int, int, int, int, int, int, int> Key;   // everything a launch shape depends on
  struct Class { std::vector<size_t> ops; std::vector<std::vector<ConvTile>> cands; std::vector<double> t; };
  std::map<Key, Class> classes;
  for (size_t i = 0; i < n_ops; ++i) {
    const OpState& o = h->ops[i];
    const rtpe_op_desc& d = o.d;
    if (o.n_geom == 0) continue;
    const rtpe_tensor_desc& ti = h->tensors[d.in_t];
    const Key key = std::make_tuple(d.cin, d.cout, d.ksize, d.stride, (int)d.kind, (int)ti.ds_log2, d.res_t >= 0 ? 1 : 0,
                                    (int)(d.flags & (RTPE_F_OUT_PREDS | RTPE_F_OUT_REFINED | RTPE_F_NO_NHWC | RTPE_F_F32)),
                                    (int)d.reserved[1] /* dilation: sets the halo */, (int)d.reserved[2]);
    Class& c = classes[key];
    if (c.ops.empty()) {
      const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
      const bool dc = d.kind == RTPE_OP_DECONV;
      const int Hp = dc ? Hi : Hi / d.stride, Wp = dc ? Wi : Wi / d.stride;
      c.cands.resize(o.n_geom);
      for (int k = 0; k < o.n_geom; ++k) conv_enum_tiles(o.plan[k], N, Hp, Wp, &c.cands[k]);
      size_t nc = c.cands[0].size();
      for (int k = 1; k < o.n_geom; ++k) nc = c.cands[k].size() < nc ? c.cands[k].size() : nc;
      c.t.assign(nc, 0.0);
    }
    c.ops.push_back(i);
  }
  size_t rounds = 0;
  for (auto& kv : classes) rounds = kv.second.t.size() > rounds ? kv.second.t.size() : rounds;
  std::vector<ConvTile> trial(n_ops * 4);
  for (size_t r = 0; r < rounds; ++r) {
    for (auto& b : trial) memset(&b, 0, sizeof(b));
    for (auto& kv : classes) {
      Class& c = kv.second;
      if (c.t.empty()) continue;
      const size_t ci = r < c.t.size() ? r : c.t.size() - 1;
      for (size_t i : c.ops)
        for (size_t k = 0; k < c.cands.size(); ++k) trial[i * 4 + k] = c.cands[k][ci];
    }
    h->tuned[shape] = trial;
    std::vector<float> best_ms(n_ops, 1e30f);
    for (int rep = 0; rep < 6; ++rep) {           // first repetition also warms caches for this choice
      rc = run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes, s, ms.data(), (int)n_ops,
               -1, -1, nullptr, nullptr, aux);
      if (rc != RTPE_OK) { h->tuned.erase(shape); return rc; }
      if (rep == 0) continue;
      for (size_t i = 0; i < n_ops; ++i) best_ms[i] = ms[i] < best_ms[i] ? ms[i] : best_ms[i];
    }
    for (auto& kv : classes) {
      Class& c = kv.second;
      if (r < c.t.size())
        for (size_t i : c.ops) c.t[r] += best_ms[i];
    }
  }
  std::vector<ConvTile> best(n_ops * 4);
  for (auto& b : best) memset(&b, 0, sizeof(b));
  for (auto& kv : classes) {
    Class& c = kv.second;
    if (c.t.empty()) continue;
    size_t bi = 0;
    double t_max = 0.0;
    for (size_t j = 0; j < c.t.size(); ++j) t_max = c.t[j] > t_max ? c.t[j] : t_max;
    // a class whose ops never ran on their own in these forwards (the second conv of a fused BasicBlock, the stem's conv2
    // inside the fused stem kernel: their time is the head's, theirs reads 0) has nothing to choose from: it stays
    // untuned, and whoever switches the fusion off gets the default launch shape instead of "the first candidate"
    if (t_max <= 0.0) continue;
    for (size_t j = 1; j < c.t.size(); ++j) if (c.t[j] < c.t[bi]) bi = j;
    for (size_t i : c.ops)
      for (size_t k = 0; k < c.cands.size(); ++k) best[i * 4 + k] = c.cands[k][bi];
  }
  h->tuned[shape] = best;
  return RTPE_OK;
}

extern "C" int rtpe_hrnet_autotune(rtpe_hrnet* h, const void* x, int32_t x_dtype, int32_t N, int32_t H, int32_t W,
                                   void* preds, void* refined, int32_t out_dtype, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return autotune_impl(h, x, x_dtype, nullptr, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes, stream);
}

// the same for a program with a second input (RTPE_OP_AUX_PACK): every timed pass is a rtpe_hrnet_forward_aux
extern "C" int rtpe_hrnet_autotune_aux(rtpe_hrnet* h, const void* x, int32_t x_dtype, const void* aux_nchw_f32, int32_t N,
                                       int32_t H, int32_t W, void* preds, void* refined, int32_t out_dtype,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  RTPE_REQUIRE(aux_nchw_f32 != nullptr, "autotune_aux: the second input is null");
  return autotune_impl(h, x, x_dtype, aux_nchw_f32, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes, stream);
}

// rtpe_hrnet_forward that also records one HIP event per op into `slot` WITHOUT
// synchronising; rtpe_hrnet_read_record(slot) later waits for them and returns the
// per-op times.  Lets bench.py time the ops inside a pipelined timed region.
extern "C" int rtpe_hrnet_forward_record(rtpe_hrnet* h, const void* x, int32_t x_dtype, int32_t N, int32_t H,
                                         int32_t W, void* preds, void* refined, int32_t out_dtype,
                                         void* workspace, size_t workspace_bytes, void* stream, int32_t slot) {
  RTPE_REQUIRE(h != nullptr && slot >= 0 && slot < 64, "forward_record: bad slot");
  return run(h, x, x_dtype, N, H, W, preds, refined, out_dtype, workspace, workspace_bytes,
             reinterpret_cast<hipStream_t>(stream), nullptr, 0, -1, -1, nullptr, &h->records[slot]);
}

extern "C" int rtpe_hrnet_read_record(rtpe_hrnet* h, int32_t slot, float* op_ms, int32_t n_ops) {
  RTPE_REQUIRE(h != nullptr && op_ms != nullptr, "read_record: null argument");
  DeviceGuard guard(h->device);
  auto it = h->records.find(slot);
  RTPE_REQUIRE(it != h->records.end() && it->second.size() == h->ops.size() + 1 && n_ops >= (int)h->ops.size(),
               "read_record: nothing recorded in slot %d", slot);
  RTPE_HIP_CHECK(hipEventSynchronize(it->second.back()));
  return read_op_times(h, it->second, true, op_ms);
}

// ---- tuned launch shapes: export / import (persisted by the caller, e.g. across processes) -------------------
// One record of RTPE_TUNED_INTS int32 per (op, parity class): {nt, waves, th, tw, lds_bytes, kind, grid,
// buf_bytes, n_bufs, n_wslots, mrun}; nt == 0 = not tuned.  Import accepts a record only if it is one of the launch
// shapes conv_enum_tiles offers for that op at this (N, H, W) - a stale or foreign file cannot produce a launch
// the kernels were not built for - and returns RTPE_E_INVALID without changing anything otherwise.
extern "C" int rtpe_hrnet_tuned_ints(const rtpe_hrnet* h, int32_t* count) {
  RTPE_REQUIRE(h != nullptr && count != nullptr, "tuned_ints: null argument");
  *count = (int32_t)(h->ops.size() * 4 * RTPE_TUNED_INTS);
  return RTPE_OK;
}

extern "C" int rtpe_hrnet_export_tuned(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, int32_t* out, int32_t n) {
  RTPE_REQUIRE(h != nullptr && out != nullptr, "export_tuned: null argument");
  RTPE_REQUIRE(n >= (int)(h->ops.size() * 4 * RTPE_TUNED_INTS), "export_tuned: buffer too small");
  auto it = h->tuned.find(std::make_tuple((int)N, (int)H, (int)W));
  RTPE_REQUIRE(it != h->tuned.end(), "export_tuned: shape %dx%dx%d has not been tuned", N, H, W);
  for (size_t i = 0; i < it->second.size(); ++i) {
    const ConvTile& t = it->second[i];
    int32_t* r = out + i * RTPE_TUNED_INTS;
    r[0] = t.nt; r[1] = t.waves; r[2] = t.th; r[3] = t.tw; r[4] = (int32_t)t.lds_bytes; r[5] = t.kind;
    r[6] = t.grid; r[7] = t.buf_bytes; r[8] = t.n_bufs; r[9] = t.n_wslots; r[10] = t.mrun;
  }
  return RTPE_OK;
}

extern "C" int rtpe_hrnet_import_tuned(rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, const int32_t* in, int32_t n) {
  RTPE_REQUIRE(h != nullptr && in != nullptr && N > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0,
               "import_tuned: bad argument");
  const size_t n_ops = h->ops.size();
  RTPE_REQUIRE(n == (int)(n_ops * 4 * RTPE_TUNED_INTS), "import_tuned: %d values for %zu ops", n, n_ops);
  std::vector<ConvTile> tiles(n_ops * 4);
  for (auto& b : tiles) memset(&b, 0, sizeof(b));
  std::vector<ConvTile> cands;
  for (size_t i = 0; i < n_ops; ++i) {
    const OpState& o = h->ops[i];
    const rtpe_op_desc& d = o.d;
    for (int k = 0; k < 4; ++k) {
      const int32_t* r = in + (i * 4 + k) * RTPE_TUNED_INTS;
      if (r[0] == 0) continue;
      RTPE_REQUIRE(k < o.n_geom, "import_tuned: op %zu has no launch shape %d", i, k);
      const rtpe_tensor_desc& ti = h->tensors[d.in_t];
      const int Hi = H >> ti.ds_log2, Wi = W >> ti.ds_log2;
      const bool dc = d.kind == RTPE_OP_DECONV;
      conv_enum_tiles(o.plan[k], N, dc ? Hi : Hi / d.stride, dc ? Wi : Wi / d.stride, &cands);
      bool found = false;
      for (const ConvTile& c : cands) {
        if (c.nt == r[0] && c.waves == r[1] && c.th == r[2] && c.tw == r[3] && (int32_t)c.lds_bytes == r[4] &&
            c.kind == r[5] && c.grid == r[6] && c.buf_bytes == r[7] && c.n_bufs == r[8] && c.n_wslots == r[9] && c.mrun == r[10]) {
          tiles[i * 4 + k] = c;
          found = true;
          break;
        }
      }
      RTPE_REQUIRE(found, "import_tuned: op %zu class %d: not a launch shape of this build", i, k);
    }
  }
  h->tuned[std::make_tuple((int)N, (int)H, (int)W)] = tiles;
  return RTPE_OK;
}
