"""ctypes binding of librtpe_hip.so (include/rtpe_hip.h).

The product path has no CPU fallback: if the shared library is missing or
cannot be loaded, importing anything that needs it raises with build
instructions (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int32, c_int64,
                    c_size_t, c_uint32, c_void_p)

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RTPE_LIBRARY: another build of the same ABI (A/B measurements of two builds on one GPU box)
LIB_PATH = os.environ.get("RTPE_LIBRARY") or os.path.join(_PKG_DIR, "librtpe_hip.so")

ABI_VERSION = 4          # rtpe_version() of the library this binding was written against (include/rtpe_hip.h)
RTPE_DTYPE_F16, RTPE_DTYPE_F32 = 1, 2
OP_STEM, OP_CONV, OP_DECONV, OP_FUSE, OP_CAST, OP_AVGPOOL, OP_SE, OP_CAM_COMBINE, OP_SIGMOID_ADD = range(9)
OP_AUX_PACK, OP_RESIZE, OP_GATE_MUL = 9, 10, 11
F_RELU, F_ROUND_CONV, F_OUT_PREDS, F_OUT_REFINED, F_NO_NHWC, F_F32 = 1, 2, 4, 8, 16, 32
F_PAIR_HEAD, F_PAIR_TAIL = 64, 128


class TensorDesc(Structure):
    _fields_ = [("channels", c_int32), ("ds_log2", c_int32), ("slot", c_int32), ("reserved", c_int32)]


class OpDesc(Structure):
    _fields_ = [("kind", c_int32), ("flags", c_int32),
                ("in_t", c_int32), ("in_coff", c_int32),
                ("out_t", c_int32), ("out_coff", c_int32),
                ("res_t", c_int32), ("res_coff", c_int32),
                ("cin", c_int32), ("cout", c_int32),
                ("ksize", c_int32), ("stride", c_int32),
                ("w_off", c_int64), ("ab_off", c_int64),
                ("n_terms", c_int32), ("term_t", c_int32 * 4), ("term_up", c_int32 * 4),
                ("reserved", c_int32 * 3), ("lane", c_int32), ("region", c_int32)]


_SIGS = {
    "rtpe_last_error_string": (c_char_p, []),
    "rtpe_version": (c_int32, []),
    "rtpe_device_count": (c_int32, []),
    "rtpe_hrnet_create": (c_int32, [POINTER(OpDesc), c_int32, POINTER(TensorDesc), c_int32,
                                    c_void_p, c_size_t, c_int32, POINTER(c_void_p)]),
    "rtpe_hrnet_destroy": (c_int32, [c_void_p]),
    "rtpe_hrnet_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "rtpe_hrnet_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "rtpe_hrnet_forward_flags": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                           c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p, c_uint32]),
    "rtpe_hrnet_forward_aux": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "rtpe_rgb_to_alt": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "rtpe_hrnet_forward_timed": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                           c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p,
                                           POINTER(c_float), c_int32]),
    "rtpe_hrnet_forward_record": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p, c_int32]),
    "rtpe_hrnet_read_record": (c_int32, [c_void_p, c_int32, POINTER(c_float), c_int32]),
    "rtpe_hrnet_op_cost": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     POINTER(c_double), POINTER(c_double)]),
    "rtpe_hrnet_autotune": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                      c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "rtpe_hrnet_autotune_aux": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                          c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "rtpe_set_option": (c_int32, [c_char_p, c_int32]),
    "rtpe_get_option": (c_int32, [c_char_p, POINTER(c_int32)]),
    "rtpe_hrnet_tuned_ints": (c_int32, [c_void_p, POINTER(c_int32)]),
    "rtpe_hrnet_export_tuned": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32), c_int32]),
    "rtpe_hrnet_import_tuned": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32), c_int32]),
    "rtpe_hrnet_op_tile": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "rtpe_hrnet_plane_major_tensors": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "rtpe_conv2d_nhwc": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                   POINTER(c_float), POINTER(c_float), c_int32, c_int32, c_int32,
                                   c_int32, c_void_p, c_void_p, c_void_p]),
    "rtpe_conv2d_nhwc_ex": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                      POINTER(c_float), POINTER(c_float), c_int32, c_int32, c_int32,
                                      c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rtpe_deconv4x4s2_nhwc": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(c_float),
                                        POINTER(c_float), c_int32, c_int32, c_void_p, c_void_p]),
    "rtpe_fuse_nhwc": (c_int32, [POINTER(c_void_p), POINTER(c_int32), c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_int32, c_void_p, c_void_p]),
    "rtpe_basicblock_nhwc": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, POINTER(c_float), POINTER(c_float),
                                       c_void_p, POINTER(c_float), POINTER(c_float), c_void_p, c_void_p]),
    "rtpe_warp_normalize": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(c_float),
                                      POINTER(c_float), c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "rtpe_resize_combine": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), c_int32, c_int32,
                                      c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "rtpe_bilinear_upsample": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                         c_int32, c_void_p]),
    "rtpe_nms": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "rtpe_topk": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                            c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_size_t, c_void_p]),
    "rtpe_topk_scratch_bytes": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "rtpe_topk_fused": (c_int32, [c_void_p, c_int32, c_int32, c_int64, c_void_p, c_int32, c_int32, c_int64,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rtpe_match_by_tag": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_double, c_double, c_int32, c_int32, c_void_p, c_int32,
                                    POINTER(c_int32)]),
    "rtpe_munkres": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, POINTER(c_int32)]),
    "rtpe_adjust_refine": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "rtpe_adjust_refine_scratch_bytes": (c_int32, [c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "rtpe_adjust_refine_fused": (c_int32, [c_void_p, c_int32, c_int32, c_int64, c_void_p, c_int32, c_int32,
                                           c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "rtpe_adjust_refine_fused_topk": (c_int32, [c_void_p, c_int32, c_int32, c_int64, c_void_p, c_int32, c_int32,
                                                c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                                c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                c_int32, c_void_p, c_size_t, c_void_p]),
    "rtpe_match_by_tag_batch": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_int32, c_double, c_double, c_int32, c_int32, c_void_p,
                                          c_int32, c_void_p, c_void_p, c_int32]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """the loaded library (raises if it has not been built)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "rtpe: %s not found - the HIP extension is not built and there is no CPU "
                "fallback.  Build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "from the repository root (needs hipcc)." % LIB_PATH)
        # PyTorch brings its own copy of the HIP runtime (torch/lib/libamdhip64.so); this library is linked against the
        # system's.  Whichever is in the process first serves both (same SONAME) - but only if torch's is loaded BEFORE this
        # library: loaded after it, a second runtime instance comes up and one of the two finds "no ROCm-capable device"
        # (seen with `g.build(); g.smoke()` in one process: build() loaded this library before anything imported torch).
        import torch  # noqa: F401
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError("rtpe: cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)       # AttributeError if an export is missing
            fn.restype, fn.argtypes = res, args
        got = L.rtpe_version()
        if got != ABI_VERSION:      # e.g. RTPE_LIBRARY pointing at a build of another round: descriptors and records differ
            raise RuntimeError("rtpe: %s has ABI revision %d, this binding needs %d - rebuild it from this tree "
                               "(`python -c \"import __graft_entry__ as g; g.build()\"`)" % (LIB_PATH, got, ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().rtpe_last_error_string()
        raise RuntimeError("rtpe_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "rtpe: %s runs on the MI355X HIP path only and got a %s tensor; move it to a GPU "
            "(there is deliberately no CPU fallback in the product path)" % (what, t.device))


def stream_ptr(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(t):
    """context manager: the device of tensor ``t`` is HIP's current device inside the block.  The handle-less
    launch functions of the ABI act on the current device (include/rtpe_hip.h, conventions); PyTorch's default
    stream handle is 0 on every device, so without this a tensor on cuda:1 would be processed by kernels
    launched on cuda:0."""
    import torch
    return torch.cuda.device(t.device)


def same_device(what, first, *others):
    """all tensors of one native call must live on one GPU"""
    for t in others:
        if t is not None and t.device != first.device:
            raise RuntimeError("rtpe: %s: tensors on different devices (%s and %s)" % (what, first.device, t.device))


def host_threads(default_cap=16):
    """host threads this process may use: the cores it is allowed to run on, divided among the ranks of the node
    (LOCAL_WORLD_SIZE, set by torch.distributed.run) - 8 ranks x 16 matcher threads + 8 torch threads each is
    exactly the burst that exhausts a container's CPU quota and stalls the kernel launches"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    return max(1, min(default_cap, cores // local_world))
