"""ctypes binding of libpathnet_hip.so (include/pathnet_hip.h).

There is no fallback: if the HIP library has not been built, importing the product path fails
loudly -- results must never silently come from a CPU path.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN_LIB_PATH", os.path.join(HERE, "csrc", "libpathnet_hip.so"))   # PN_LIB_PATH: tuning builds

PN_OK = 0
PN_ERR_ARG, PN_ERR_IO, PN_ERR_FORMAT, PN_ERR_HIP, PN_ERR_EMPTY_TABLE, PN_ERR_NOMEM, PN_ERR_CAPACITY = (
    -1, -2, -3, -4, -5, -6, -7)
DRAW_GLIBC_REPLAY, DRAW_PHILOX = 0, 1
ABI_VERSION = 10              # PN_ABI_VERSION of include/pathnet_hip.h this binding was written against
VARIANT_HETERO, VARIANT_HOMO, VARIANT_PAGG = 0, 1, 2
CELL_DEFAULT, CELL_LSTM, CELL_RNN, CELL_GRU, CELL_MEAN, CELL_SUM = 0, 1, 2, 3, 4, 5
SEQ_MATH_DEFAULT, SEQ_MATH_BF16X3, SEQ_MATH_F16X2 = 0, 1, 2     # pn_pagg_shape.seq_math
COMPACT_AUTO, COMPACT_ON, COMPACT_OFF = 0, 1, 2                 # pn_pagg_shape.compact
LINEAR_SPLIT_MAX = 8          # PN_LINEAR_SPLIT_MAX: workspace floats per output element of pn_linear_forward
LINEAR_BWD_SPLIT_MAX = 32     # PN_LINEAR_BWD_SPLIT_MAX: chunk sums of pn_linear_backward's deterministic weight gradient

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class PnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpathnet_hip error %d: %s" % (code, msg))
        self.code = code


class DeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("arch", ctypes.c_char * 64), ("compute_units", ctypes.c_int32),
                ("lds_bytes_per_block", ctypes.c_int32), ("hbm_bytes", ctypes.c_int64), ("clock_khz", ctypes.c_int32)]


class SamplerTables(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("total", ctypes.c_int64), ("off", vp), ("triples", vp), ("dis", vp),
                ("adj_off", vp), ("adj", vp), ("radj_off", vp), ("radj", vp), ("draws_per_step", ctypes.c_int32),
                ("node_ref", vp)]


class PaggShape(ctypes.Structure):
    """struct pn_pagg_shape.  S_total / group_begin: this call computes the pooling groups [group_begin, +S) of a
    batch of S_total masked nodes (0 = S); batch_groups > 0: internal micro-batches of that many groups."""
    _fields_ = [("variant", ctypes.c_int32), ("N", ctypes.c_int32), ("F", ctypes.c_int32), ("H", ctypes.c_int32),
                ("C", ctypes.c_int32), ("S", ctypes.c_int32), ("W", ctypes.c_int32), ("L", ctypes.c_int32),
                ("S_total", ctypes.c_int32), ("group_begin", ctypes.c_int32), ("batch_groups", ctypes.c_int32),
                ("cell", ctypes.c_int32), ("deterministic", ctypes.c_int32), ("compact", ctypes.c_int32),
                ("seq_math", ctypes.c_int32)]


class PaggArgs(ctypes.Structure):
    _fields_ = ([("shape", PaggShape), ("X", vp), ("ids", vp), ("codes", vp), ("sel", vp)] +
                [(k, vp) for k in ("fc0_w", "fc0_b", "bank_w", "bank_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w",
                                   "att_b", "fc2_w", "fc2_b")] +
                [("p_seq", ctypes.c_float), ("p_cls", ctypes.c_float), ("seed", ctypes.c_uint64), ("mask_seq", vp),
                 ("mask_cls", vp), ("out", vp), ("workspace", vp), ("workspace_bytes", ctypes.c_int64),
                 ("g_out", vp), ("g_X", vp)] +
                [("g_" + k, vp) for k in ("fc0_w", "fc0_b", "bank_w", "bank_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w",
                                          "att_b", "fc2_w", "fc2_b")] +
                [("Xh_in", vp), ("g_Xh", vp), ("Xh_ready", vp), ("g_Xh_ready", vp), ("no_save", ctypes.c_int32),
                 ("reuse_tables", ctypes.c_int32),
                 ("index_rows_local", ctypes.c_int32), ("step_state", vp)])


# name -> (restype, argtypes): every symbol include/pathnet_hip.h declares
class AdamTensor(ctypes.Structure):
    """struct pn_adam_tensor"""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("count", ctypes.c_int64)]


SIGNATURES = {
    "pn_abi_version": (ctypes.c_int, []),
    "pn_last_error": (ctypes.c_char_p, []),
    "pn_device_query": (ctypes.c_int, [ctypes.POINTER(DeviceInfo)]),
    "pn_context_create": (ctypes.c_int, [ctypes.POINTER(vp)]),
    "pn_context_destroy": (ctypes.c_int, [vp]),
    "pn_context_set_knob": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_int32]),
    "pn_context_get_knob": (ctypes.c_int, [vp, ctypes.c_char_p, c_i32p]),
    "pn_clock_probe": (ctypes.c_int, [c_f64p, vp]),
    "pn_step_state_advance": (ctypes.c_int, [vp, vp]),
    "pn_edges_read_text": (ctypes.c_int, [ctypes.c_char_p, c_i32p, c_i64p, c_i32p, c_i32p, c_f64p, ctypes.c_int64]),
    "pn_pairs_read_text": (ctypes.c_int, [ctypes.c_char_p, c_i32p, c_i64p, c_i32p, c_i32p, ctypes.c_int64]),
    "pn_uniform_build": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_i32p, c_i32p, c_i64p, c_i32p, c_i32p, c_i32p,
                                        ctypes.c_int64, c_i64p]),
    "pn_alias_build": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_i32p, c_i32p, c_f64p, c_i64p, c_i32p, c_i32p,
                                      c_f64p, c_u32p, ctypes.c_int64, c_i64p]),
    "pn_alias_pack": (ctypes.c_int, [ctypes.c_int64, c_i32p, c_i32p, c_u32p, c_i32p]),
    "pn_node_ref_pack": (ctypes.c_int, [ctypes.c_int32, c_i64p, c_u32p]),
    "pn_hops_dense": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_i32p, c_i32p, ctypes.c_int32, c_u8p]),
    "pn_csr_build": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, c_i32p, c_i32p, ctypes.c_int32, c_i64p, c_i32p,
                                    ctypes.c_int64, c_i64p]),
    "pn_glibc_draws": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int64, c_i32p]),
    "pn_sample_workspace_bytes": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                                 ctypes.c_int32, c_i64p]),
    "pn_sample_paths": (ctypes.c_int, [vp, ctypes.POINTER(SamplerTables), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                       vp, vp, vp, ctypes.c_int64, vp, vp, vp, vp]),
    "pn_paths_write_text": (ctypes.c_int, [ctypes.c_char_p, c_i32p, c_u8p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_int32]),
    "pn_paths_read_text": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32, c_i32p, c_u8p, ctypes.c_int64, c_i64p]),
    "pn_paths_write_bin": (ctypes.c_int, [ctypes.c_char_p, c_i32p, c_u8p, ctypes.c_int64, ctypes.c_int32]),
    "pn_paths_read_bin": (ctypes.c_int, [ctypes.c_char_p, c_i32p, c_i32p, c_u8p, ctypes.c_int64, c_i64p]),
    "pn_pagg_workspace_bytes": (ctypes.c_int, [ctypes.POINTER(PaggShape), c_i64p]),
    "pn_pagg_shape_info": (ctypes.c_int, [ctypes.POINTER(PaggShape), c_i64p]),
    "pn_pagg_forward": (ctypes.c_int, [vp, ctypes.POINTER(PaggArgs), vp]),
    "pn_pagg_backward": (ctypes.c_int, [vp, ctypes.POINTER(PaggArgs), vp]),
    "pn_pagg_gather": (ctypes.c_int, [vp, ctypes.POINTER(PaggShape), vp, vp, vp, vp, vp]),
    "pn_gemm_f32": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, vp,
                                   ctypes.c_int64, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   vp]),
    "pn_profile_configure": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.c_int32]),
    "pn_profile_stage_count": (ctypes.c_int, []),
    "pn_profile_stage_name": (ctypes.c_char_p, [ctypes.c_int32]),
    "pn_profile_read": (ctypes.c_int, [vp, c_f64p, c_i64p]),
    "pn_linear_forward": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp,
                                         vp, ctypes.c_int64, vp]),
    "pn_linear_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp,
                                          vp, vp, ctypes.c_int64, vp]),
    "pn_pagg_train_step": (ctypes.c_int, [vp, ctypes.POINTER(PaggArgs), vp, ctypes.c_float, vp, vp]),
    "pn_pagg_debug_offsets": (ctypes.c_int, [ctypes.POINTER(PaggShape), c_i64p]),
    "pn_pagg_range_offset": (ctypes.c_int, [ctypes.POINTER(PaggShape), c_i64p]),
    "pn_pagg_paths_stream": (ctypes.c_int, [vp, ctypes.POINTER(PaggShape), vp, ctypes.POINTER(ctypes.c_void_p)]),
    "pn_merw_workspace_bytes": (ctypes.c_int, [ctypes.c_int32, c_i64p]),
    "pn_merw_probabilities": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, vp, vp, vp, vp, vp, c_f64p, ctypes.c_int32,
                                             ctypes.c_double, c_i32p, vp, ctypes.c_int64, vp]),
    "pn_cross_entropy": (ctypes.c_int, [vp, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]),
    "pn_adam_step": (ctypes.c_int, [ctypes.POINTER(AdamTensor), ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int64, vp, vp]),
    "pn_adam_step_advance": (ctypes.c_int, [ctypes.POINTER(AdamTensor), ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp]),
}

_lib = None


def load():
    """Load the library (once).  Raises ImportError if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("pathnet_amd: %s is missing -- build the HIP extension first "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                              % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        if lib.pn_abi_version() != ABI_VERSION:
            raise ImportError("libpathnet_hip.so ABI version mismatch")
        _lib = lib
    return _lib


_contexts = {}


def context(device):
    """This process's pn_context for a CUDA device (created at first use with that device current, destroyed at
    interpreter exit).  The context owns the library's second stream and the pn_profile_* records; pass it to every
    device entry point, inside ``with torch.cuda.device(device)``."""
    import torch
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ctx = _contexts.get(idx)
    if ctx is None:
        lib = load()
        h = vp()
        with torch.cuda.device(idx):
            check(lib.pn_context_create(ctypes.byref(h)))
        ctx = _contexts[idx] = vp(h.value)
        if len(_contexts) == 1:
            import atexit
            atexit.register(_destroy_contexts)
    return ctx


def set_knob(name, value, device="cuda"):
    """pn_context_set_knob on this process's context for `device`: the kernel-selection knobs (PN_EVAL_ZW, PN_SEQ4, ...) are
    read from the environment once, when the context is created; afterwards this is the only way to change them.
    Returns the previous value."""
    old = get_knob(name, device)
    check(load().pn_context_set_knob(context(device), name.encode(), int(value)))
    return old


def get_knob(name, device="cuda"):
    v = ctypes.c_int32()
    check(load().pn_context_get_knob(context(device), name.encode(), ctypes.byref(v)))
    return v.value


def _destroy_contexts():
    if _lib is None:
        return
    for h in _contexts.values():
        try:
            _lib.pn_context_destroy(h)
        except Exception:       # interpreter shutdown: the HIP runtime may already be gone
            pass
    _contexts.clear()


def stream_ptr(device):
    import torch
    return vp(torch.cuda.current_stream(device).cuda_stream)


def check(rc):
    if rc != PN_OK:
        raise PnError(rc, load().pn_last_error().decode(errors="replace"))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def np_ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))
