"""ctypes binding of ``libxt_mi355x.so`` (the C ABI declared in ``include/xt_mi355x.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C xingtian_amd/csrc``.
There is NO CPU fallback: if the shared object is missing or a call fails, a
``RuntimeError`` is raised.  ``import torch`` happens first so that the HIP runtime the
kernels bind to is the one PyTorch-ROCm already loaded (same streams / context).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

import torch  # noqa: F401  (must precede loading the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxt_mi355x.so")

ACT = {"none": 0, None: 0, "relu": 1, "tanh": 2, "sigmoid": 3, "softsign": 4, "softplus": 5, "leaky_relu": 6, "elu": 7,
       "selu": 8, "swish": 9, "gelu": 10}


class ConvGeom(Structure):
    _fields_ = [(n, c_int32) for n in ("H", "W", "C", "KH", "KW", "S", "PT", "PL", "OH", "OW", "N", "act")]


class InputXform(Structure):
    _fields_ = [("is_u8", c_int32), ("mean", c_float), ("std", c_float)]


class LayerDesc(Structure):
    _fields_ = [("g", ConvGeom), ("param_off", c_int64), ("trunk", c_int32)]


ABI_VERSION = 12         # XT_ABI_VERSION of include/xt_mi355x.h
ACTION_TYPE = {"Categorical": 0, "DiagGaussian": 1}


class NetDesc(Structure):
    _fields_ = [("n_layers", c_int32), ("layers", POINTER(LayerDesc)), ("n_trunks", c_int32),
                ("feat", c_int32), ("action_dim", c_int32), ("pi_off", c_int64), ("v_off", c_int64),
                ("n_params", c_int64), ("xf", InputXform), ("in_h", c_int32), ("in_w", c_int32),
                ("in_c", c_int32), ("action_type", c_int32), ("logstd_off", c_int64)]


class PpoCfg(Structure):
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("clip_ratio", c_float), ("ent_coef", c_float), ("vf_clip", c_float),
                ("critic_coef", c_float), ("max_grad_norm", c_float), ("batch_size", c_int32),
                ("num_sgd_iter", c_int32), ("grad_scale", c_float), ("global_batch", c_int32),
                ("shard_rank", c_int32), ("shard_world", c_int32)]


class ImpalaCfg(Structure):
    _fields_ = [("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("grad_norm_clip", c_float), ("gamma", c_float), ("sample_batch_step", c_int32),
                ("grad_scale", c_float), ("opt_type", c_int32), ("rms_decay", c_float), ("rms_eps", c_float),
                ("shard_rank", c_int32), ("shard_world", c_int32)]


class Tuning(Structure):
    """xt_tuning of include/xt_mi355x.h: process-wide kernel-selection knobs (defaults = measured best)."""
    _fields_ = [(n, c_int32) for n in (
        "bf16x6", "dgrad_all_classes", "dgrad_tile64", "dgrad_halo", "bwd_own_instance", "bwd_fit_slots",
        "conv1_bf16x3", "conv1_flat", "conv1_waves", "fwd_two_groups", "direct", "direct_fwd", "direct_dgrad",
        "direct_all", "direct_waves", "direct_max_waves", "direct_tile64_tiles", "fwd_split_target",
        "wgrad_split_target", "reduce_z_lanes", "defer_splitk", "finalize_ticket", "fwd_tiled_valid", "wgrad_rows", "fwd_prefetch_all", "bwd_deep_prefetch", "fwd_four_groups", "reduce_deep_lanes", "fwd_xcd_chunk", "tail_overlap", "tail_fused", "dense_wgrad_x6", "fwd_fuse12")]


class TrainIO(ctypes.Structure):
    """xt_train_io of include/xt_mi355x.h: the runtime calls around one train, folded into the train's C call"""
    _fields_ = [("wait_event", c_void_p), ("consumed_event", c_void_p), ("loss_host", c_void_p), ("loss_event", c_void_p),
                ("publish_dst", c_void_p), ("publish_event", c_void_p), ("wait_loss", c_int32), ("tail_in_graph", c_int32),
                ("wait_dma_ticket", ctypes.c_uint64)]


OPT_TYPE = {"adam": 0, "rmsprop": 1}
XCHG_OVERLAP = 1         # XT_XCHG_OVERLAP
DIRECT_HANDLE_BYTES = 64  # XT_DIRECT_HANDLE_BYTES
DP_TAIL_FLOATS = 32       # XT_DP_TAIL_FLOATS: 2 x 16 slots behind the gradient (rows / loss share of every rank)
DP_ERR_BITS = {1: "a peer's gradient slices never arrived (scatter wait)", 2: "a peer's reduced slice never arrived (reduce wait)",
               4: "the ranks hold different numbers of rows"}
_P = c_void_p
# name -> (restype, argtypes); every symbol include/xt_mi355x.h declares
SIGNATURES = {
    "xt_abi_version": (c_int32, []),
    "xt_last_launch_arith": (c_int32, []),
    "xt_last_error": (c_char_p, []),
    "xt_build_arch": (c_char_p, []),
    "xt_build_sources_sha": (c_char_p, []),
    "xt_gae_f64_ragged": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_double, c_double, _P]),
    "xt_pad_channels": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P]),
    "xt_net_set_rccl": (c_int32, [_P, _P, _P, c_int32]),
    "xt_net_rccl_status": (c_int32, [_P, POINTER(c_int32), POINTER(c_int32)]),
    "xt_tuning_get": (c_int32, [POINTER(Tuning)]),
    "xt_tuning_set": (c_int32, [POINTER(Tuning)]),
    "xt_stage_rows": (c_int32, [_P, _P, c_int64, _P, c_int64, c_int64, c_int32, _P]),
    "xt_stage_tune": (c_int32, [c_int64, POINTER(c_float)]),
    "xt_stage_get": (c_int32, [POINTER(c_int32), POINTER(c_int32)]),
    "xt_stage_set": (c_int32, [c_int32, c_int32]),
    "xt_adv_normalize_f64": (c_int32, [_P, c_int64, c_double, _P, _P]),
    "xt_gae_f64": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_double, c_double, _P]),
    "xt_layer_fwd": (c_int32, [POINTER(ConvGeom), POINTER(InputXform), c_int32, _P, _P, _P, _P, _P, _P, c_int32, _P]),
    "xt_layer_wgrad": (c_int32, [POINTER(ConvGeom), POINTER(InputXform), c_int32, _P, _P, _P, _P, _P, c_int32, _P]),
    "xt_layer_dgrad": (c_int32, [POINTER(ConvGeom), c_int32, _P, _P, _P, c_int32, _P, _P]),
    "xt_heads_fwd": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "xt_ppo_loss": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float,
                              c_float, _P, _P, _P, _P]),
    "xt_ppo_loss_gauss": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float,
                                    c_float, c_float, _P, _P, _P, _P, _P]),
    "xt_ppo_loss_reduce": (c_int32, [_P, c_int32, c_float, c_float, c_float, _P, _P, _P]),
    "xt_impala_loss": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "xt_heads_bwd": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "xt_adam_state_init": (c_int32, [_P, _P]),
    "xt_adam_tf_clip": (c_int32, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "xt_grad_global_norm": (c_int32, [_P, c_int64, c_float, c_float, _P, _P, _P]),
    "xt_net_create": (c_int32, [POINTER(NetDesc), c_int32, POINTER(c_void_p)]),
    "xt_net_destroy": (None, [_P]),
    "xt_net_workspace_bytes": (c_int64, [_P]),
    "xt_net_bind": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int64]),
    "xt_net_forward": (c_int32, [_P, _P, _P, c_int32, _P, _P, _P]),
    "xt_net_ppo_step": (c_int32, [_P, POINTER(PpoCfg), _P, _P, c_int32, _P, _P, _P, _P, _P, c_int32, _P, _P, _P]),
    "xt_net_ppo_train": (c_int32, [_P, POINTER(PpoCfg), _P, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, _P]),
    "xt_net_set_grad_exchange": (c_int32, [_P, _P, _P]),
    "xt_net_set_grad_exchange_ex": (c_int32, [_P, _P, _P, c_int32]),
    "xt_keras_impala_loss": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P]),
    "xt_adam_keras": (c_int32, [_P, _P, _P, _P, c_int32, _P, _P, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    "xt_net_keras_impala_step": (c_int32, [_P, _P, _P, c_int32, _P, _P, _P, c_float, _P, _P, _P]),
    "xt_net_impala_step": (c_int32, [_P, POINTER(ImpalaCfg), _P, c_int32, _P, _P, _P, _P, c_int32, _P, _P, _P]),
    "xt_net_impala_train": (c_int32, [_P, POINTER(ImpalaCfg), _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_int32, _P]),
    "xt_net_apply": (c_int32, [_P, c_float, c_float, c_float, c_float, c_float, c_float, _P]),
    "xt_net_layer_offsets": (c_int32, [_P, c_int32, POINTER(c_int64)]),
    "xt_net_time_layer": (c_int32, [_P, c_int32, c_int32, _P, _P, c_int32, c_int32, POINTER(c_float), _P]),
    "xt_direct_create": (c_int32, [c_int32, c_int32, c_int64, _P, POINTER(c_void_p)]),
    "xt_direct_connect": (c_int32, [_P, _P]),
    "xt_direct_connect_local": (c_int32, [_P, POINTER(c_void_p)]),
    "xt_allreduce_direct": (c_int32, [_P, _P, c_int64, _P]),
    "xt_allreduce_direct_group": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_void_p), c_int64, POINTER(c_void_p)]),
    "xt_direct_exchange_hook": (c_int32, [_P, c_int64, _P, _P]),
    "xt_direct_set_timeout_ms": (c_int32, [_P, c_int32]),
    "xt_direct_set_fused": (c_int32, [_P, c_int32]),
    "xt_direct_status": (c_int32, [_P, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "xt_direct_destroy": (c_int32, [_P]),
    "xt_direct_info": (c_int32, [_P, POINTER(c_int32), POINTER(c_int32)]),
    "xt_direct_read_result": (c_int32, [_P, _P, c_int64]),
    "xt_direct_reset": (c_int32, [_P]),
    "xt_net_set_dp": (c_int32, [_P, c_int32, c_int32, c_float]),
    "xt_net_set_direct": (c_int32, [_P, _P]),
    "xt_net_time_tail": (c_int32, [_P, c_float, c_float, c_int32, POINTER(c_float), _P]),
    "xt_net_impala_train_io": (c_int32, [_P, POINTER(ImpalaCfg), _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_int32,
                                         POINTER(TrainIO), _P]),
    "xt_net_io_times": (c_int32, [_P, POINTER(c_double), POINTER(c_int64), c_int32]),
    "xt_net_io_wait": (c_int32, [_P, _P, _P]),
    "xt_sdma_copy_d2h": (c_int32, [_P, _P, c_int64]),
    "xt_dma_h2d_async": (c_int32, [_P, _P, c_int64, POINTER(ctypes.c_uint64)]),
    "xt_dma_wait_upto": (c_int32, [ctypes.c_uint64, c_int32]),
    "xt_net_io_seq": (ctypes.c_uint32, [_P]),
    "xt_net_io_loss_ready": (c_int32, [_P]),
    "xt_net_io_publish_wait": (c_int32, [_P, ctypes.c_uint32, c_int32]),
}


def dp_error_text(bits):
    """human-readable form of the data-parallel error bits a train leaves in loss_acc[2] / xt_direct_status"""
    bits = int(bits)
    return "; ".join(t for b, t in DP_ERR_BITS.items() if bits & b) or "bits {}".format(bits)

_lib = None


def load():
    """Load the shared object (once) and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "xingtian_amd: HIP library {} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C xingtian_amd/csrc`). There is no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.xt_abi_version() != ABI_VERSION:
        raise RuntimeError("xingtian_amd: ABI version mismatch")
    _lib = lib
    return lib


def get_tuning():
    """dict of the library's current kernel-selection knobs (xt_tuning_get)."""
    t = Tuning()
    check(load().xt_tuning_get(ctypes.byref(t)), "xt_tuning_get")
    return {n: getattr(t, n) for n, _ in Tuning._fields_}


def set_tuning(**knobs):
    """Change knobs by name (xt_tuning_set); returns the previous values of the ones changed.  Set them before
    networks are created: captured hipGraphs are not revisited."""
    t = Tuning()
    lib = load()
    check(lib.xt_tuning_get(ctypes.byref(t)), "xt_tuning_get")
    old = {}
    for k, v in knobs.items():
        if not hasattr(t, k):
            raise KeyError("unknown tuning knob {!r}".format(k))
        old[k] = getattr(t, k)
        setattr(t, k, int(v))
    check(lib.xt_tuning_set(ctypes.byref(t)), "xt_tuning_set")
    return old


def check(rc, what=""):
    if rc != 0:
        msg = load().xt_last_error()
        raise RuntimeError("xingtian_amd: {} failed (rc={}): {}".format(what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None); an int is taken as a raw device-visible address."""
    if t is None:
        return None
    if isinstance(t, int):
        return c_void_p(t)
    return c_void_p(t.data_ptr())


_STREAMS = {}


def current_stream(device=None):
    """``torch.cuda.current_stream(device)`` without its ~6 us of Python per call (four of them sat in every 128-frame
    IMPALA train): the Stream object is cached per device and reused while the raw handle of the current stream
    (``torch._C._cuda_getCurrentRawStream``, ~0.3 us) has not changed"""
    idx = torch.cuda.current_device() if device is None else (device.index if hasattr(device, "index") else int(device))
    if idx is None:
        idx = torch.cuda.current_device()
    try:
        raw = torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:          # (a torch without the private accessor)
        return torch.cuda.current_stream(idx)
    hit = _STREAMS.get(idx)
    if hit is not None and hit[0] == raw:
        return hit[1]
    st = torch.cuda.current_stream(idx)
    _STREAMS[idx] = (raw, st)
    return st


def stream_ptr():
    """The current PyTorch HIP stream as a void* (kernels are enqueued on it)."""
    try:
        return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
    except AttributeError:
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def kernel_sources_sha():
    """sha256 (16 hex digits) over the kernel sources the library is built from: profile artefacts carry it so that
    counters measured on an older build of the kernels are recognised as stale (bench.py, tools/pmc_summary.py)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


_hip = None
H2D, D2H, D2D = 1, 2, 3        # hipMemcpyKind


def _hiplib():
    """the HIP runtime PyTorch already loaded, with the few prototypes the host plumbing calls directly"""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [c_void_p, c_void_p, ctypes.c_size_t, c_int32, c_void_p]
        _hip.hipMemcpyAsync.restype = c_int32
        _hip.hipHostGetDevicePointer.argtypes = [POINTER(c_void_p), c_void_p, ctypes.c_uint]
        _hip.hipHostGetDevicePointer.restype = c_int32
    return _hip


def memcpy_async(dst_ptr, src_ptr, nbytes, kind, stream):
    """hipMemcpyAsync on an explicit stream (a ``torch.cuda.Stream`` or a raw handle) straight through the HIP runtime
    PyTorch already loaded: the per-message copies of the ingest / publish paths without ``with torch.cuda.stream(...)``
    (each enter / exit costs ~10 us of Python and runtime calls: a third of the host time of a 128-frame IMPALA train)."""
    h = stream.cuda_stream if hasattr(stream, "cuda_stream") else stream
    rc = _hiplib().hipMemcpyAsync(c_void_p(int(dst_ptr)), c_void_p(int(src_ptr)), int(nbytes), kind, c_void_p(h))
    if rc != 0:
        raise RuntimeError("hipMemcpyAsync failed with hipError {}".format(rc))


def host_device_ptr(host_addr):
    """device-side address of a page-locked host block (hipHostGetDevicePointer): kernels can then read / write it directly
    over PCIe -- a few KB of labels per train cost one DMA set-up less that way.  -> int, or None when it is not mapped"""
    out = c_void_p()
    rc = _hiplib().hipHostGetDevicePointer(ctypes.byref(out), c_void_p(int(host_addr)), 0)
    return int(out.value) if rc == 0 and out.value else None


def built_sources_sha():
    """The digest the LOADED library was compiled from (xt_build_sources_sha, embedded by csrc/Makefile).  Differs from
    ``kernel_sources_sha()`` when a stale prebuilt ``.so`` sits next to newer sources."""
    v = load().xt_build_sources_sha()
    return v.decode() if v else "unknown"


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("xingtian_amd: no HIP device visible -- the learner update path runs only on the GPU "
                           "(no CPU fallback by design)")
