/*
 * rtpe_hip.h - C ABI of librtpe_hip.so: the MI355X (gfx950) inference path for
 * the HigherHRNet-w48 teacher forward pass and the heatmap->keypoint decode.
 *
 * The reference (andres-fr/realtime-pose-estimation) is pure Python and has no
 * FFI boundary of its own; the Python objects that sit on this ABI keep the
 * reference's import paths and signatures.  Each entry
 * point below names the reference code it replaces (paths relative to the
 * reference repo root).
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device
 *     addresses (e.g. torch.Tensor.data_ptr()), `stream` is a hipStream_t.
 *   - every function returns 0 on success or a negative RTPE_E_* code; nothing
 *     throws across the boundary; rtpe_last_error_string() describes the last
 *     failure of the calling thread.
 *   - the caller owns all inputs, outputs and workspaces.  The library owns
 *     only what rtpe_hrnet_create() allocates (packed weights) and frees it in
 *     rtpe_hrnet_destroy().  No device allocation, no host sync inside
 *     rtpe_hrnet_forward() or any launch function unless documented
 *     ("host-returning").
 *   - all device work is ordered behind what the given stream holds at the
 *     call and in front of what is enqueued on it afterwards.  Forwards of a
 *     program with parallel regions (rtpe_op_desc.lane / region) run the ops of
 *     lanes 1..3 on INTERNAL non-blocking streams that fork from and join into
 *     the given stream by events (option "lanes", default on); those streams
 *     and events are created by the first such forward of a handle (the only
 *     device-resource creation outside rtpe_hrnet_create) and destroyed with
 *     it.  rtpe_hrnet_forward_flags(..., RTPE_FWD_NO_LANES) keeps every launch
 *     on the given stream: use it under stream capture and with stream
 *     priorities that the internal streams must not escape.
 *   - threads: handles are independent (one process per GPU is the intended
 *     shape).  Several host threads may call forwards on ONE handle at the
 *     same time (each with its own stream and workspace): a forward that uses
 *     the lanes holds a per-handle lock while it enqueues, so such calls
 *     enqueue one after the other; create / destroy / autotune of a handle
 *     must not overlap other calls on it.  rtpe_set_option is process-wide.
 *   - devices: functions that take an rtpe_hrnet handle make the handle's
 *     device current for the call and restore the caller's device before they
 *     return (rtpe_hrnet_create included); their buffers and stream must
 *     belong to that device.  The handle-less launch functions (decode,
 *     bilinear, warp, single-layer entries) run on HIP's CURRENT device: the
 *     caller makes the device of the buffers current first (the Python
 *     binding wraps every call in torch.cuda.device(tensor.device)).
 *   - activations at the boundary are NCHW (what the reference's callers pass
 *     and expect); inside they are NHWC fp16.
 */
#ifndef RTPE_HIP_H
#define RTPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTPE_OK 0
#define RTPE_E_INVALID (-1)   /* bad argument / unsupported shape            */
#define RTPE_E_HIP (-2)       /* a HIP runtime call failed                    */
#define RTPE_E_NOMEM (-3)     /* workspace too small / allocation failed      */
#define RTPE_E_NODEVICE (-4)  /* no usable gfx950 device                      */

#define RTPE_DTYPE_F16 1
#define RTPE_DTYPE_F32 2

const char* rtpe_last_error_string(void);
/* ABI revision: 4 = option "head_direct"; option "stream_pc" and ConvTile kind 3 are gone (a tuned-shape
 * file that names kind 3 is refused by rtpe_hrnet_import_tuned); the Python binding
 * refuses a library of another revision (RTPE_LIBRARY);
 * 3 = tuned-shape records have 11 integers (RTPE_TUNED_INTS);
 * 2 = rtpe_op_desc carries lane / region (sizeof 112 -> 120),
 * rtpe_hrnet_forward_flags exists.  A caller built against revision 1 must
 * not pass its descriptors to this library: check before rtpe_hrnet_create. */
int rtpe_version(void);
/* number of visible HIP devices (0 on a CPU-only host); never fails */
int rtpe_device_count(void);

/* ------------------------------------------------------------------------ *
 * Network program.  The Python module tree (same attribute names, hence the
 * same 1810 state-dict keys as rtpe/third_party/pose_higher_hrnet.py:259-444)
 * compiles itself into a flat list of ops over NHWC tensors; the executor
 * below runs it with hand-written kernels.
 * ------------------------------------------------------------------------ */

/* op kinds */
#define RTPE_OP_STEM 0   /* 3x3 s2 conv Cin=3 + BN + ReLU from the NCHW input;
                            pose_higher_hrnet.py:363-365,638-640 + tofp16
                            (fp16_utils/fp16util.py:50-51)                    */
#define RTPE_OP_CONV 1   /* kxk (1|3|5) stride (1|2) conv + BN/bias [+residual]
                            [+ReLU]; BasicBlock :46-75, Bottleneck :78-116,
                            transitions :548-583, fuse convs :200-230,
                            final_layers :460-482                            */
#define RTPE_OP_DECONV 2 /* ConvTranspose2d k4 s2 p1 + BN + ReLU on the channel
                            concat of two sources; :513-524, :680-682        */
#define RTPE_OP_FUSE 3   /* sum of up to 4 terms with nearest upsampling and a
                            final ReLU; HighResolutionModule.forward :245-254 */

#define RTPE_OP_CAST 4        /* fp16 NHWC -> fp32 NHWC: tofp32 after the half-wrapped stem of
                                 the students, fp16util.py:64-68, students.py:732-733 */
#define RTPE_OP_AVGPOOL 5     /* AvgPool2d(3, 2, 1, count_include_pad=False), fp32,
                                 students.py:657-666                            */
#define RTPE_OP_SE 6          /* SELayer gate, students.py:118-142: cin = C, cout =
                                 hidden; w_off -> fc1 w (hid,C), b1, fc2 w (C,hid),
                                 b2 (fp32); output = per-image vector tensor    */
#define RTPE_OP_CAM_COMBINE 7 /* relu(res + in * gate), ContextAwareModule
                                 students.py:199-200; gate = term_t[0]          */
#define RTPE_OP_SIGMOID_ADD 8 /* y = res + sigmoid(in[...,0] / 20) broadcast over
                                 channels, students.py:755-756; the sigmoid map is
                                 NCHW output 0 when RTPE_F_OUT_PREDS is set     */

#define RTPE_OP_AUX_PACK 9    /* the SECOND network input (AttentionStudentSteps' `alt` image in LAB / HSV,
                                 students.py:980-1002): (N,3,H,W) fp32 NCHW -> fp32 NHWC with 4 channels
                                 (one zero pad) at full resolution                */
#define RTPE_OP_RESIZE 10     /* F.interpolate(in, size of out, mode="bilinear") (align_corners=False) of cout
                                 channels, fp32 NHWC -> channels [out_coff, out_coff+cout) of out;
                                 students.py:996-1000                             */
#define RTPE_OP_GATE_MUL 11   /* y[..., :cout] = res[..., :cout] * sigmoid(in[..., 0] / d), d = the fp32 whose bits
                                 are reserved[0] (0: no division), students.py:1012-1040; the sigmoid map is
                                 NCHW output 0 when RTPE_F_OUT_PREDS is set       */

/* op flags */
#define RTPE_F_RELU 1
#define RTPE_F_ROUND_CONV 2 /* round the fp32 accumulator to fp16 before the
                               affine (conv and BN are separate fp16 ops in the
                               reference's half wrapper)                      */
#define RTPE_F_OUT_PREDS 4   /* also write NCHW output 0 (preds)             */
#define RTPE_F_OUT_REFINED 8 /* also write NCHW output 1 (refined)           */
#define RTPE_F_NO_NHWC 16    /* skip the NHWC store (head whose only consumer
                               is the NCHW output)                            */
#define RTPE_F_F32 32        /* the op works on fp32 tensors / weights (plain
                               PoseHigherResolutionNet without the half
                               wrapper; the students' fp32 part)              */
#define RTPE_F_PAIR_HEAD 64  /* conv 1x1 64 -> 256 + residual + ReLU whose output the NEXT op, a conv 1x1
                                256 -> 64 + ReLU flagged RTPE_F_PAIR_TAIL, reads: the two may run as one
                                kernel that writes the 256-channel tensor and never reads it back (layer1's
                                Bottlenecks, pose_higher_hrnet.py:96-116; option "pair_1x1").  The program's
                                slot assignment must keep the head's input and residual alive over the tail */
#define RTPE_F_PAIR_TAIL 128

typedef struct rtpe_tensor_desc {
  int32_t channels; /* allocated channels per pixel (the NHWC row length)     */
  int32_t ds_log2;  /* spatial size = (H >> ds_log2, W >> ds_log2); < 0: one
                       vector of `channels` per image (SE gates)             */
  int32_t slot;     /* workspace slot (tensors with disjoint lifetimes share) */
  int32_t reserved; /* element size in bytes: 4 = fp32, anything else = fp16  */
} rtpe_tensor_desc;

typedef struct rtpe_op_desc {
  int32_t kind;
  int32_t flags;
  int32_t in_t, in_coff;   /* input tensor id and first channel               */
  int32_t out_t, out_coff; /* output tensor id and first channel              */
  int32_t res_t, res_coff; /* residual tensor (or -1)                         */
  int32_t cin, cout;       /* logical channel counts                          */
  int32_t ksize, stride;
  int64_t w_off;           /* byte offset in the raw weight blob: fp16 weights
                              in PyTorch order (OIHW; IOHW for DECONV)         */
  int64_t ab_off;          /* byte offset of fp32 alpha[cout] then beta[cout] */
  int32_t n_terms;         /* FUSE: number of terms                           */
  int32_t term_t[4];       /* FUSE: term tensor ids                           */
  int32_t term_up[4];      /* FUSE: log2 nearest-upsampling factor per term   */
  int32_t reserved[3];     /* [0] logical Cin for cost accounting (0: = cin);
                              [1] dilation of a 3x3 conv (0: 1);
                              [2] channels written to the NHWC output (0: all
                              that fit; used to lay the HDC branches of a
                              ContextAwareModule side by side)                */
  int32_t lane;            /* 0..3: the branches of a HighResolutionModule are independent
                              (pose_higher_hrnet.py:242-243 runs them one after another); ops of
                              different lanes inside one region may run concurrently on internal
                              streams that fork from / join into the caller's stream               */
  int32_t region;          /* 0: outside any parallel region; ops of one region are contiguous      */
} rtpe_op_desc;

typedef struct rtpe_hrnet rtpe_hrnet;

/* Build an executor for a program.  `weights` is a HOST pointer to the raw
 * blob (see rtpe_op_desc.w_off/ab_off); it is re-packed into MFMA fragment
 * order and uploaded to `device`.  Replaces model construction + strict
 * state-dict load of rtpe/helpers.py:32-73.  Host-returning. */
int rtpe_hrnet_create(const rtpe_op_desc* ops, int32_t n_ops,
                      const rtpe_tensor_desc* tensors, int32_t n_tensors,
                      const void* weights, size_t weights_bytes,
                      int32_t device, rtpe_hrnet** out);
int rtpe_hrnet_destroy(rtpe_hrnet* h);

/* workspace bytes for a batch of N images of H x W (multiples of 32) */
int rtpe_hrnet_workspace_bytes(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W,
                               size_t* bytes);

/* PoseHigherResolutionNet.forward (pose_higher_hrnet.py:637-686) under the half
 * wrapper (fp16util.py:87-91).  x: device NCHW (N,3,H,W) of x_dtype.  preds:
 * device NCHW (N,n_preds,H/4,W/4), refined: (N,n_refined,H/2,W/2), both of
 * out_dtype (F32 = the tofp32 cast of fp16util.py:64-68 folded in). */
int rtpe_hrnet_forward(rtpe_hrnet* h, const void* x, int32_t x_dtype,
                       int32_t N, int32_t H, int32_t W,
                       void* preds, void* refined, int32_t out_dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* rtpe_hrnet_forward with per-call switches (flags: RTPE_FWD_*; unknown bits
 * are an error).  RTPE_FWD_NO_LANES: every op on `stream`, whatever the option
 * "lanes" says (callers that overlap whole forwards on streams of their own,
 * rtpe/engine.py TeacherPipeline.stream; graph capture). */
#define RTPE_FWD_NO_LANES 1u
int rtpe_hrnet_forward_flags(rtpe_hrnet* h, const void* x, int32_t x_dtype,
                             int32_t N, int32_t H, int32_t W,
                             void* preds, void* refined, int32_t out_dtype,
                             void* workspace, size_t workspace_bytes, void* stream,
                             uint32_t flags);

/* Same, but brackets every op with HIP events on `stream` and returns the
 * per-op time in ms (op_ms[n_ops]).  Host-returning (synchronises). */
int rtpe_hrnet_forward_timed(rtpe_hrnet* h, const void* x, int32_t x_dtype,
                             int32_t N, int32_t H, int32_t W,
                             void* preds, void* refined, int32_t out_dtype,
                             void* workspace, size_t workspace_bytes, void* stream,
                             float* op_ms, int32_t n_ops);

/* rtpe_hrnet_forward that also records one HIP event per op into `slot`
 * (0..63) WITHOUT synchronising; rtpe_hrnet_read_record() later waits for
 * those events and returns the per-op times in ms (host-returning). */
int rtpe_hrnet_forward_record(rtpe_hrnet* h, const void* x, int32_t x_dtype,
                              int32_t N, int32_t H, int32_t W,
                              void* preds, void* refined, int32_t out_dtype,
                              void* workspace, size_t workspace_bytes, void* stream, int32_t slot);
int rtpe_hrnet_read_record(rtpe_hrnet* h, int32_t slot, float* op_ms, int32_t n_ops);

/* number of activation tensors a forward of (N,H,W) keeps plane-major
 * ([C/48][N][H][W][48]) instead of NHWC: the inner tensors of the BasicBlock
 * chains with C >= 96 (pose_higher_hrnet.py:46-75) when every conv around them
 * runs on the streaming kernel.  Internal layout only, results are identical
 * (RTPE_PLANE_MAJOR=0 turns it off). */
int rtpe_hrnet_plane_major_tensors(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, int32_t* count);

/* algorithmic cost of op i for (N,H,W): flops and HBM bytes of a layer-fused
 * execution (SURVEY.md section 8d accounting). */
int rtpe_hrnet_op_cost(const rtpe_hrnet* h, int32_t op, int32_t N, int32_t H, int32_t W,
                       double* flops, double* bytes);

/* Per-shape plan autotuning (the reference runs with cudnn.benchmark = True,
 * teacher_inference.py:31): runs one forward, then times every launch shape of
 * every conv op on its real buffers and keeps the fastest for (N,H,W).  All
 * shapes give bit-identical results.  Arguments as rtpe_hrnet_forward.
 * Host-returning. */
int rtpe_hrnet_autotune(rtpe_hrnet* h, const void* x, int32_t x_dtype,
                        int32_t N, int32_t H, int32_t W,
                        void* preds, void* refined, int32_t out_dtype,
                        void* workspace, size_t workspace_bytes, void* stream);
/* the same for a program with a second input (arguments as rtpe_hrnet_forward_aux) */
int rtpe_hrnet_autotune_aux(rtpe_hrnet* h, const void* x, int32_t x_dtype, const void* aux_nchw_f32,
                            int32_t N, int32_t H, int32_t W,
                            void* preds, void* refined, int32_t out_dtype,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Process-wide tuning options; every setting gives bit-identical results (they select between kernel
 * variants for A/B measurements in one process).  "block_pc" = 0: the fused BasicBlock never runs on the
 * producer / consumer kernel (default 1: wherever H % 8 == 0 and W % 16 == 0; env RTPE_BLOCK_PC).  "block_ring" = 1:
 * the fused BasicBlock kernel streams its weights through a 3-slot LDS ring instead of keeping them resident
 * (default 0; env RTPE_BLOCK_RING; takes precedence over "block_pc").  "direct_1x1" (env
 * RTPE_DIRECT_1X1): the 1x1 conv kernel without a staged input tile (csrc/conv_direct.hip) is 0 = never used, 1 = one more launch
 * shape for the autotuner and the default of un-tuned launches (default), 2 = as 1 (reserved).  "lanes" (env RTPE_LANES): the
 * independent branches of a parallel region run 0 = one after another on the caller's stream, 1 = concurrently on internal
 * (normal-priority) streams that fork from and join the caller's stream (default: batch 1 at 640 x 640 2.93 -> 2.73 ms, batch
 * 32 14.5 -> 13.7 ms per forward), 2 = concurrently when the batch is small (N * H * W <= 4 * 640 * 640).  Measured caveat: while
 * other work of the process runs on a HIGH-priority stream the lanes more than halve the throughput; use 0 there.  "tile_dma" (env RTPE_TILE_DMA): the one-workgroup-per-tile conv kernel stages its halo tiles 1 = by LDS-DMA (one
 * memory round trip per channel chunk, no staging registers; default), 0 = through registers, eight 16-byte loads per lane at a
 * time.  "pair_1x1" (env RTPE_PAIR_1X1): op pairs flagged RTPE_F_PAIR_HEAD / _TAIL run 1 = as one kernel (csrc/conv_pair.hip;
 * default), 0 = as two launches.  "fused_stem" (env RTPE_FUSED_STEM): the stem op (conv1 + bn1 + relu) and the 64 -> 64
 * stride-2 conv behind it (conv2 + bn2 + relu) of a half-precision program run 1 = as one kernel that keeps the
 * half-resolution map in LDS (csrc/stem_fused.hip; default), 0 = as two launches, 2 = as two launches with the stem op on
 * the fused kernel's conv1 code (its multiply-add chain on the matrix pipe; a test setting).  Same bits in all three.
 * "conv64" (env RTPE_CONV64): the 3x3 stride-1 convs with 64 input and 64 output channels and no residual (conv2 of layer1's
 * Bottlenecks) run 1 = on persistent workgroups with double-buffered halo tiles and register-resident weights
 * (csrc/conv64.hip; default, and one more launch shape for the autotuner), 0 = on the one-workgroup-per-tile kernel.
 * "head_direct" (env RTPE_HEAD_DIRECT): the 1x1 heads with 48 input channels and fp32 NCHW output run 1 = on the direct scheme
 * with an NCHW epilogue (csrc/conv_direct.hip conv1x1_head_kernel; default; maps whose pixel count per image is a multiple of
 * 32), 0 = on the launch shape chosen for the layer (one workgroup per tile).  Same bits.
 * "deconv48" (env RTPE_DECONV48; added within ABI revision 4: an older library answers RTPE_E_INVALID): transposed convs
 * (k4 s2 p1) from 48 or 96 input channels to at most 48 output channels run 1 = with all four sub-pixel classes on one
 * persistent kernel that shares their halo tiles (csrc/deconv48.hip; default), 0 = as four classes in one grid of the
 * one-workgroup-per-tile kernel.  Same bits.
 * "conv48s2" (env RTPE_CONV48S2; added within ABI revision 4): 3x3 stride-2 convs from 48 input channels to 48 / 96 / 192 / 384
 * output channels run 1 = on persistent workgroups with register-resident weights, neighbouring ones that read the same input
 * (the first downsampling convs of a fuse layer) as ONE launch (csrc/conv48s2.hip; default), 0 = each on the launch shape chosen
 * for it.  Same bits. */
int rtpe_set_option(const char* name, int32_t value);
/* The value an option has NOW (set by rtpe_set_option, else the environment's, else the default): what the next
 * launch will use.  bench.py names the kernel it reports from this, not from the environment. */
int rtpe_get_option(const char* name, int32_t* value);

/* rtpe_hrnet_forward for a program with a second input (RTPE_OP_AUX_PACK): aux = (N,3,H,W) fp32 NCHW on the
 * device (AttentionStudentSteps.forward(x, alt=...), students.py:966).  Everything else as rtpe_hrnet_forward. */
int rtpe_hrnet_forward_aux(rtpe_hrnet* h, const void* x, int32_t x_dtype, const void* aux_nchw_f32,
                           int32_t N, int32_t H, int32_t W, void* preds, void* refined, int32_t out_dtype,
                           void* workspace, size_t workspace_bytes, void* stream);

/* RGB -> CIE-LAB (D65, 2 degree observer) or HSV, the two `alt_colorspace` choices of the reference's
 * CocoDistillationDatasetAugmented2 (rtpe/dataloaders.py:314-375, which calls skimage.color.rgb2lab / rgb2hsv on
 * the ToTensor'd image).  src: (N,3,H,W) fp32 in [0,1]; dst: same shape; mode 0 = LAB (L in [0,100]), 1 = HSV
 * (all in [0,1]).  The published scikit-image formulas in fp32 (parity unpinned: skimage is not in the image). */
int rtpe_rgb_to_alt(const float* src_nchw, int32_t N, int32_t H, int32_t W, int32_t mode, float* dst_nchw,
                    void* stream);

/* Tuned launch shapes of (N,H,W) as plain integers, so that a caller can keep them across
 * processes (the reference's cudnn.benchmark has to re-tune in every process).
 * RTPE_TUNED_INTS int32 per (op, parity class), 4 classes per op:
 * {pixel tiles/wave (0 = not tuned), waves, tile_h, tile_w, LDS bytes, kind, workgroups,
 * halo buffer bytes, halo buffers, weight slots, cout tiles per workgroup (0 = the whole
 * packed block)}.  rtpe_hrnet_tuned_ints gives the array length (11 per record since
 * revision 3 of the library, 10 before: size buffers from that call, not from the macro of
 * an older header).  Import validates every record against the launch shapes this build offers for
 * the op at that shape and changes nothing if any record is foreign (RTPE_E_INVALID). */
#define RTPE_TUNED_INTS 11
int rtpe_hrnet_tuned_ints(const rtpe_hrnet* h, int32_t* count);
int rtpe_hrnet_export_tuned(const rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, int32_t* out, int32_t n);
int rtpe_hrnet_import_tuned(rtpe_hrnet* h, int32_t N, int32_t H, int32_t W, const int32_t* in, int32_t n);

/* kernel variant of conv op i for (N,H,W): out8 = {cout tiles/wave, pixel
 * tiles/wave, waves, tile_h, tile_w, channel chunk, cout blocks, v} with
 * v > 0: LDS bytes of the one-workgroup-per-tile kernel; v <= -100000:
 * -(workgroups + 100000 * halo buffers) of the streaming kernel; -900001 / -900002: first /
 * second conv of a BasicBlock that runs as ONE fused kernel (conv_block.hip), launched by the
 * first; -(500000 + workgroups): the persistent 64 -> 64 3x3 kernel (conv64.hip); -600001 / -600002: the stem op / the 64 -> 64 stride-2 conv behind it when both run as
 * one kernel (stem_fused.hip, option "fused_stem"), launched at the stem op */
int rtpe_hrnet_op_tile(const rtpe_hrnet* h, int32_t op, int32_t N, int32_t H, int32_t W, int32_t* out8);

/* ------------------------------------------------------------------------ *
 * Single layers (layer-level parity tests; same kernels the executor runs).
 * ------------------------------------------------------------------------ */

/* One BasicBlock of a 48-channel branch (pose_higher_hrnet.py:46-75) as a single fused kernel:
 * y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x), NHWC fp16 (N,H,W,48) in and out, weights
 * (48,48,3,3) fp16 on the host, BatchNorm as fp32 alpha/beta[48].  Bit-identical to two
 * rtpe_conv2d_nhwc calls (layer-level parity tests).  Host-returning. */
int rtpe_basicblock_nhwc(const void* x, int32_t N, int32_t H, int32_t W, const void* w1_host,
                         const float* alpha1, const float* beta1, const void* w2_host,
                         const float* alpha2, const float* beta2, void* y, void* stream);
/* ConvTranspose2d(k=4, s=2, p=1, no bias) + BatchNorm (+ReLU), pose_higher_hrnet.py:513-524, as the
 * four sub-pixel 2x2 convolutions the executor runs.  x NHWC fp16 (N,H,W,cin); w_host the PyTorch
 * (cin, cout, 4, 4) fp16 weight; y NHWC fp16 (N,2H,2W,cout).  Host-returning (layer-level tests). */
int rtpe_deconv4x4s2_nhwc(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin, const void* w_host,
                          const float* alpha_host, const float* beta_host, int32_t cout, int32_t flags,
                          void* y, void* stream);

/* HighResolutionModule fuse sum, pose_higher_hrnet.py:245-254: y = [relu](((t0 + t1) + t2) + ...) with one
 * rounding per add; term t is a dense NHWC (N, H >> up[t], W >> up[t], C) tensor read with nearest
 * upsampling (nn.Upsample :209).  flags: RTPE_F_RELU, RTPE_F_F32.  Stream-ordered. */
int rtpe_fuse_nhwc(const void* const* terms, const int32_t* term_up, int32_t n_terms, int32_t N, int32_t H,
                   int32_t W, int32_t C, int32_t flags, void* y, void* stream);

/* NHWC fp16 conv: y = act( round16( round16?(conv(x,w)) * alpha + beta ) [+ res] ).
 * w: HOST pointer, fp16 OIHW (cout,cin,k,k); alpha/beta: HOST fp32[cout].
 * x:(N,H,W,cin) y:(N,Ho,Wo,cout) res:(N,Ho,Wo,cout) or NULL, device, dense.
 * Host-returning (packs + uploads the weights, then syncs). */
int rtpe_conv2d_nhwc(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin,
                     const void* w_host, const float* alpha_host, const float* beta_host,
                     int32_t cout, int32_t ksize, int32_t stride, int32_t flags,
                     const void* res, void* y, void* stream);

/* Same with dilation (3x3 stride 1 only; padding = dilation) and, with
 * RTPE_F_F32 in flags, fp32 x / w_host / res / y (no intermediate rounding):
 * the dilated fp32 convs of ContextAwareModule, rtpe/students.py:145-201. */
int rtpe_conv2d_nhwc_ex(const void* x, int32_t N, int32_t H, int32_t W, int32_t cin,
                        const void* w_host, const float* alpha_host, const float* beta_host,
                        int32_t cout, int32_t ksize, int32_t stride, int32_t dilation,
                        int32_t flags, const void* res, void* y, void* stream);

/* ------------------------------------------------------------------------ *
 * Decode: validate_hhrnet.py:94-98 + rtpe/third_party/group.py:125-287.
 * ------------------------------------------------------------------------ */

/* Pre-processing in front of the path: cv2.warpAffine of
 * resize_align_multi_scale (rtpe/third_party/transforms.py:181-192, matrix of
 * get_affine_transform :59-93) + torchvision ToTensor + Normalize
 * (validate_hhrnet.py:63-67) in one pass.  src: (h, w, 3) uint8 on the device, row
 * stride in bytes; m_dst_to_src: the 2x3 matrix that maps destination pixel (x, y)
 * to source coordinates (the inverse of what warpAffine is given); dst: (3, oh, ow)
 * fp32.  Bilinear weights in fp32, zero outside the image (documented convention:
 * cv2's fixed-point interpolation is not pinned).  round_u8 != 0: the interpolated
 * value is rounded to a grey level first (the reference's warp returns a uint8 image);
 * then x / 255 and (x - mean) / std as torchvision computes them.  Stream-ordered. */
int rtpe_warp_normalize(const void* src_hwc_u8, int32_t h, int32_t w, int32_t stride_bytes,
                        const float* m_dst_to_src, const float* mean, const float* stdev,
                        void* dst_chw_f32, int32_t oh, int32_t ow, int32_t round_u8, void* stream);

/* One step of the multi-scale / flip test aggregation (upstream HigherHRNet core/inference.py
 * get_multi_stage_outputs / aggregate_results, called from legacy/valid_ae1dim.py:166-207):
 *     dst = [dst +] resize( flip_w( src[:, channel_map] ) )  [/ div]
 * src (N,C_src,h,w), dst (N,C_dst,oh,ow) fp32 NCHW on the device; resize = F.interpolate(mode="bilinear",
 * align_corners=False) (identity when the sizes agree); channel_map: C_dst HOST ints (NULL = identity; <= 64
 * channels), e.g. the COCO flip index; flip_w: torch.flip(., [3]) of the resized map; accumulate != 0: add to
 * dst; div != 1: true division of the result.  Bit-equal to the torch ops on the CPU.  Stream-ordered. */
int rtpe_resize_combine(const float* src, int32_t N, int32_t C_src, int32_t h, int32_t w,
                        const int32_t* channel_map, int32_t C_dst, int32_t flip_w, float* dst,
                        int32_t oh, int32_t ow, int32_t accumulate, float div, void* stream);

/* F.interpolate(mode="bilinear", align_corners=True), fp32 NCHW planes.
 * validate_hhrnet.py:94-98.  src (planes,h,w) -> dst (planes,oh,ow). */
int rtpe_bilinear_upsample(const float* src, int32_t planes, int32_t h, int32_t w,
                           float* dst, int32_t oh, int32_t ow, void* stream);

/* HeatmapParser.nms, group.py:134-138: det * (maxpool_kxk(det) == det). */
int rtpe_nms(const float* det, int32_t planes, int32_t h, int32_t w,
             int32_t ksize, int32_t pad, float* out, void* stream);

/* HeatmapParser.top_k, group.py:144-179, on already-upsampled maps.
 * det (planes,h,w) f32 with planes = N*J; tag (planes_tag,h,w,D) f32 where
 * planes_tag = planes (tag_per_joint) or N (then `joints` maps are shared);
 * K = max_num_people.  Outputs (device): val_k (planes,K) f32, ind_k (planes,K)
 * i32 flat index, tag_k (planes,K,D) f32.  Ties in value are ordered by
 * ascending index.  scratch: device, rtpe_topk_scratch_bytes(). */
int rtpe_topk(const float* det, const float* tag, int32_t planes, int32_t joints,
              int32_t tag_per_joint, int32_t h, int32_t w, int32_t D, int32_t K,
              int32_t nms_ksize, int32_t nms_pad,
              float* val_k, int32_t* ind_k, float* tag_k,
              void* scratch, size_t scratch_bytes, void* stream);
int rtpe_topk_scratch_bytes(int32_t planes, int32_t h, int32_t w, int32_t K, size_t* bytes);

/* Fused variant: bilinear upsample of the low-res network outputs + NMS +
 * top-k + tag gather, never materialising the (oh,ow) maps (D = 1).
 * Plane p = n*J + j of the heat maps lives at hm + n*hm_img_stride + j*hh*hw
 * (elements), the tag plane at tg + n*tg_img_stride + j*th*tw, so the forward
 * outputs can be passed as they are (refined; preds + 17*th*tw with an image
 * stride of 34*th*tw). */
int rtpe_topk_fused(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride,
                    const float* tg, int32_t th, int32_t tw, int64_t tg_img_stride,
                    int32_t N, int32_t J, int32_t oh, int32_t ow, int32_t K,
                    int32_t nms_ksize, int32_t nms_pad,
                    float* val_k, int32_t* ind_k, float* tag_k,
                    void* scratch, size_t scratch_bytes, void* stream);

/* match_by_tag, group.py:26-97 (+ py_max_match :19-23, Params :100-110) for
 * one image.  HOST function, no GPU needed.  tag_k (J,K,D) f32, ind_k (J,K)
 * i32 flat indices (x = ind % w, y = ind / w), val_k (J,K) f32.  Writes up to
 * max_people_out persons as rows (J, 3+D) f32 (x, y, val, tag...) into `ans`
 * and returns the TOTAL number of persons found in *n_people (may exceed
 * max_people_out; then only the first max_people_out are written). */
int rtpe_match_by_tag(const float* tag_k, const int32_t* ind_k, const float* val_k,
                      int32_t J, int32_t K, int32_t D, int32_t w,
                      int32_t max_num_people, double detection_threshold,
                      double tag_threshold, int32_t use_detection_val,
                      int32_t ignore_too_much,
                      float* ans, int32_t max_people_out, int32_t* n_people);

/* py_max_match, group.py:19-23 (Munkres().compute of the PyPI package
 * `munkres`).  HOST function.  cost: nr x nc row-major; pairs: 2*min(nr,nc)
 * ints out as (row, col); *n_pairs = pairs written. */
int rtpe_munkres(const double* cost, int32_t nr, int32_t nc, int32_t* pairs, int32_t* n_pairs);

/* HeatmapParser.adjust (group.py:181-200) + .refine (group.py:202-264) on
 * the GPU for P persons of N images.  det (N*J,h,w) f32, tag (N*J,h,w,D) f32
 * device maps; ans_in (P,J,3+D) f32 device is read, ans_out (same shape, a
 * different buffer) is written: every detected joint is adjusted (if
 * do_adjust), every undetected one is searched for by the tag-penalised
 * arg-max (if do_refine), the rest is copied.  person_img (P) i32 device =
 * image index of each person, ascending (NULL: all image 0).  scores (P) f32
 * device = mean val per person before refine (group.py:272); may be NULL.
 * scratch: device, rtpe_adjust_refine_scratch_bytes(). */
int rtpe_adjust_refine(const float* det, const float* tag, int32_t N, int32_t J, int32_t h, int32_t w,
                       int32_t D, const float* ans_in, float* ans_out, const int32_t* person_img,
                       int32_t P, int32_t do_adjust, int32_t do_refine, float* scores,
                       void* scratch, size_t scratch_bytes, void* stream);
int rtpe_adjust_refine_scratch_bytes(int32_t P, int32_t J, int32_t D, size_t* bytes);

/* Fused variant working from the low-res maps (bilinear evaluated on the fly,
 * D = 1; addressing as in rtpe_topk_fused). */
int rtpe_adjust_refine_fused(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride,
                             const float* tg, int32_t th, int32_t tw, int64_t tg_img_stride,
                             int32_t N, int32_t J, int32_t oh, int32_t ow,
                             const float* ans_in, float* ans_out, const int32_t* person_img, int32_t P,
                             int32_t do_adjust, int32_t do_refine, float* scores,
                             void* scratch, size_t scratch_bytes, void* stream);

/* The same for a caller that still holds the top-k table rtpe_topk_fused wrote
 * for these maps (topk_val / topk_ind: (N*J, K), device-addressable): refine's
 * exact arg-max shortcut needs the first pixel attaining each plane's maximum,
 * which is the table's first entry whenever that maximum is positive, so the
 * extra pass over every plane is skipped (group.py:202-264 semantics
 * unchanged; planes without a positive maximum take the full scan). */
int rtpe_adjust_refine_fused_topk(const float* hm, int32_t hh, int32_t hw, int64_t hm_img_stride,
                                  const float* tg, int32_t th, int32_t tw, int64_t tg_img_stride,
                                  int32_t N, int32_t J, int32_t oh, int32_t ow,
                                  const float* ans_in, float* ans_out, const int32_t* person_img, int32_t P,
                                  int32_t do_adjust, int32_t do_refine, float* scores,
                                  const float* topk_val, const int32_t* topk_ind, int32_t K,
                                  void* scratch, size_t scratch_bytes, void* stream);

/* match_by_tag for a batch of N images on `n_threads` host threads.  Inputs as
 * rtpe_match_by_tag with a leading image axis.  People of image n follow those
 * of image n-1 in `ans` (max_people_total rows of (J,3+D)); person_img[i] =
 * image of row i; counts[n] = people found in image n (all of them are
 * counted, rows beyond max_people_total are dropped).  HOST function. */
int rtpe_match_by_tag_batch(const float* tag_k, const int32_t* ind_k, const float* val_k,
                            int32_t N, int32_t J, int32_t K, int32_t D, int32_t w,
                            int32_t max_num_people, double detection_threshold,
                            double tag_threshold, int32_t use_detection_val,
                            int32_t ignore_too_much, float* ans, int32_t max_people_total,
                            int32_t* person_img, int32_t* counts, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* RTPE_HIP_H */
