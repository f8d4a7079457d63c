"""ctypes binding of libfrost_hip.so (include/frost_hip.h).  The product path: there is NO CPU fallback --
if the library is missing or a call fails this raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FROST_HIP_LIB", os.path.join(_HERE, "libfrost_hip.so"))   # override: A/B runs of two builds in one GPU session (dev only)
_lib = None

P, I, L, F = C.c_void_p, C.c_int, C.c_int64, C.c_float

Q_MIN, Q_MAX, Q_SCALE, Q_ZP, Q_FQMIN, Q_FQMAX, Q_INV, Q_QMAX, Q_OBS_EN, Q_FQ_EN, Q_STRIDE = 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12
COEF_ROWS = int(os.environ.get("FROST_COEF_ROWS_ALLOC", "14"))          # FROST_COEF_ROWS_ALLOC: the 8 named rows + three more copies of the S1 / S2 rows (ABI 5: the reduce passes spread their atomics, readers add the copies up)
COEF_ROWS_NAMED = 8
DWQ_NC, DWQ_SPREAD_MAX = int(os.environ.get("FROST_DWQ_NC", "4")), 32768      # FROST_DWQ_NC / FROST_DWQ_SPREAD_MAX: copies of the raw weight-gradient sums of small layers (csrc/frost_common.h)
COEF_A, COEF_B = 0, 1
COEF_M, COEF_R, COEF_K1, COEF_S1, COEF_S2, COEF_VFRAC = 2, 3, 4, 5, 6, 7      # FROST_COEF_* of include/frost_hip.h
STATS_BYTES_PER_CH = 24 * int(os.environ.get("FROST_STATS_TABLES", "4"))      # FROST_STATS_BYTES_PER_CH: four replicated 24-byte-per-channel tables (ABI 5)


class FrostWDesc(C.Structure):
    _fields_ = [("w", P), ("gamma", P), ("rvar", P), ("qrec", P), ("wq_pack", P), ("wsum", P), ("minmax2", P),
                ("wt_pack", P), ("wscale", P), ("wmin", P), ("wmax", P), ("cout", C.c_int32), ("cin_g", C.c_int32), ("kk", C.c_int32), ("kind", C.c_int32),
                ("cpad", C.c_int32), ("kpad", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32)]


class FrostIDesc(C.Structure):
    _fields_ = [("w", P), ("gamma", P), ("beta", P), ("rmean", P), ("rvar", P), ("pack", P), ("biasf", P),
                ("cout", C.c_int32), ("cin_g", C.c_int32), ("kk", C.c_int32), ("kind", C.c_int32),
                ("cpad", C.c_int32), ("kpad", C.c_int32), ("reserved0", C.c_int32), ("reserved1", C.c_int32)]


class FrostFDesc(C.Structure):
    _fields_ = [("w", P), ("gamma", P), ("beta", P), ("rmean", P), ("rvar", P), ("nbt", P), ("pack", P), ("pack_t", P), ("stat", P),
                ("coef", P), ("dgamma", P), ("dbeta", P), ("cout", C.c_int32), ("cin_g", C.c_int32), ("kk", C.c_int32),
                ("kind", C.c_int32), ("cpad", C.c_int32), ("kpad", C.c_int32), ("kpad_t", C.c_int32), ("fp32", C.c_int32)]


ABI_VERSION = 5
TICKET_WORDS = 40      # FROST_TICKET_WORDS: zeroed uint32 words behind every last-workgroup-done ticket (main counter + 32 sub-counters)


class FrostFinDesc(C.Structure):
    _fields_ = [("qrec_w", P), ("gamma", P), ("beta", P), ("rmean", P), ("rvar", P), ("nbt", P), ("coef", P), ("qrec_y", P), ("counter", P),
                ("training", C.c_int32), ("relu", C.c_int32), ("observe", C.c_int32), ("reserved", C.c_int32), ("wscale", P),
                ("cat_qrec_b", P), ("cat_qrec_y", P)]


class FrostBlockLayer(C.Structure):
    _fields_ = [("wq_pack", P), ("wsum", P), ("stats", P), ("fin", FrostFinDesc), ("wt_pack", P), ("dwq", P), ("cout", C.c_int32), ("k", C.c_int32)]


class FrostBlockDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("x", P), ("qrec_x", P),
                ("conv1", FrostBlockLayer), ("conv2", FrostBlockLayer), ("reduce", FrostBlockLayer), ("y1", P), ("y2", P), ("conv_out3", P), ("y3", P)]


class FrostBlockBwd(C.Structure):
    _fields_ = [("gout3", P), ("dc3", P), ("g2", P), ("g1", P), ("dc1", P), ("dx", P), ("side_stream", P)]


class FrostGDesc(C.Structure):
    _fields_ = [("dwq", P), ("w", P), ("gamma", P), ("sigma_r", P), ("qw", P), ("coef", P), ("dw", P), ("dgamma", P), ("dbeta", P),
                ("cout", C.c_int32), ("per", C.c_int32), ("cpad", C.c_int32), ("reserved", C.c_int32), ("wscale", P)]


class FrostOptTensor(C.Structure):
    _fields_ = [("p", P), ("g", P), ("exp_min", P), ("exp_max", P), ("coin", P), ("buf0", P), ("buf1", P), ("buf2", P),
                ("n", C.c_int64), ("weight_decay", F), ("lr", F), ("first_step", C.c_int32), ("pad", C.c_int32)]


class FrostOptHyper(C.Structure):
    _fields_ = [("kind", C.c_int32), ("boost", C.c_int32), ("toss_coin", C.c_int32), ("nesterov", C.c_int32),
                ("amsgrad", C.c_int32), ("centered", C.c_int32), ("beta", F), ("momentum", F), ("dampening", F),
                ("alpha", F), ("eps", F), ("beta1", F), ("beta2", F), ("clip_by", F), ("bc_beta", F),
                ("noise_scale", F), ("bc1", F), ("bc2", F), ("seed", C.c_uint64), ("offset", C.c_uint64)]


_PROTOS = {
    "frost_abi_version": [],
    "frost_add_state_floats": [],
    "frost_ticket_words": [],
    "frost_fin_desc_bytes": [],
    "frost_minmax_f32": [P, L, P, P],
    "frost_fill_minmax": [P, I, P],
    "frost_minmax_input": [P, I, I, I, I, L, L, L, L, P, P],
    "frost_observer_update": [P, P, I, I, I, P],
    "frost_quantize_input": [P, I, I, I, I, L, L, L, L, P, P, I, P],
    "frost_fake_quant_f32": [P, L, P, I, I, P, P, P],
    "frost_fake_quant_bwd_f32": [P, P, L, P, P],
    "frost_dequant_act": [P, L, P, P, P],
    "frost_weight_prep": [P, I, I, I, I, P],
    "frost_step_prologue": [P, I, P, I, P, P, P, P, I, I, P],
    "frost_export_wq": [P, I, L, P, P],
    "frost_mbox_workspace_floats": [],
    "frost_mbox_forward": [P, P, P, P, P, I, I, I, I, F, I, F, F, P, P, P, P, P, P, P, P, P],
    "frost_mbox_backward": [P, P, P, P, P, P, P, P, I, I, I, P, P, P],
    "frost_stats_init_table": [P, P, P, I, P],
    "frost_pw_conv_fwd": [P, P, P, P, L, I, I, I, P, P, P, I, P, P],
    "frost_pw_conv_fwd_fin": [P, P, P, P, L, I, I, P, P, P],
    "frost_dw_conv_fwd_fin": [P, P, P, P, I, I, I, I, I, I, P, P, P],
    "frost_dw_conv_fwd": [P, P, P, P, I, I, I, I, I, I, I, P, P, P, I, P, P],
    "frost_stem_im2col": [P, P, I, I, I, P, P],
    "frost_stem_wgrad_remap": [P, I, I, P, P],
    "frost_conv_finalize": [P, L, I, P, P, P, P, P, P, P, I, I, I, P, P, P, P],
    "frost_cat_observe": [P, P, P, I, P],
    "frost_cat_requant": [P, P, I, P, P, I, L, P, P, P],
    "frost_add_minmax": [P, P, P, P, L, P, P],
    "frost_add_minmax_observe": [P, P, P, P, L, P, P, I, P],
    "frost_add_requant": [P, P, P, P, L, P, P, P],
    "frost_avgpool": [P, P, I, I, I, P, P, P],
    "frost_classifier_fwd": [P, P, P, P, I, I, I, P, P, P],
    "frost_pw_conv_bwd": [P, P, P, P, P, P, L, I, I, I, P, P, I, P, P, P, I, P],
    "frost_pw_wgrad": [P, P, P, L, I, I, P, P],
    "frost_pw_bwd_fused_ok": [L, I, I],
    "frost_pw_conv_int": [P, P, P, P, L, I, I, P, P],
    "frost_pw_ew": [P, L, I, P, P, I, I, P, P, P],
    "frost_pw_ew_emit_add": [P, L, I, P, P, I, P, P, P, P, P, I, P],
    "frost_pw_ew_add_bwd": [P, L, I, P, P, I, I, P, P, P, P, P, P, I, P, P],
    "frost_pw_conv_fwd_keep": [P, P, P, P, L, I, I, P, P, P, P],
    "frost_pwc_bwd_ok": [L, I, I],
    "frost_pwc_conv_bwd": [P, P, P, P, L, I, I, I, P, P, I, P, P, P],
    "frost_pwc_conv_fwd_emit": [P, P, P, P, L, I, I, P, P, P, P],
    "frost_block_supported": [I, I, I, I, I, I],
    "frost_block_fwd": [P, P],
    "frost_block_bwd": [P, P, P],
    "frost_block_expand_dw_stats": [P, P, P, P, P, P, P, I, I, I, I, I, P, P, I, P, P, P],
    "frost_block_dw_stats": [P, P, P, P, I, I, I, I, I, P, P, P],
    "frost_block_dw_reduce_supported": [I, I, I, I, I, I],
    "frost_block_dw_bwd_supported": [I, I, I, I, I],
    "frost_block_dw_bwd_reduce": [P, P, P, P, I, I, I, I, I, P, P, I, P, P],
    "frost_block_dw_bwd": [P, P, P, P, P, P, I, I, I, I, I, P, P, I, P, P, P, P],
    "frost_block_dw_bwd_c1": [P, P, P, P, P, P, I, I, I, I, I, P, P, I, P, P, P, P, P, P, P, P, I, I, P],
    "frost_block_dw_bwd_c1_ok": [I, I, I, I, I, I],
    "frost_block_dw_reduce": [P, P, P, P, P, P, I, P, I, I, I, I, I, P, P, I, P, P, P, P],
    "frost_pw_dgrad_wide_ok": [L, I, I],
    "frost_pw_dgrad_wide": [P, P, P, L, I, I, P, I, P],
    "frost_pw_conv_bwd_fused": [P, P, P, P, P, P, L, I, I, P, P, I, P, P, P, I, P, P],
    "frost_dw_conv_bwd": [P, P, P, P, P, I, I, I, I, I, I, I, P, P, I, P, P, P],
    "frost_dw_conv_bwd_dc_wgrad": [P, P, P, P, P, I, I, I, I, I, I, P, P, I, P, P, P, P],
    "frost_dw_dgrad": [P, P, P, I, I, I, I, I, I, P, I, P, P],
    "frost_dw_wgrad": [P, P, P, I, I, I, I, I, I, P, P],
    "frost_dw_bwd_fused_ok": [I, I, I, I, I],
    "frost_dw_bwd_fused": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, I, P, P, P, P],
    "frost_dw_bwd_fused_c1_ok": [I, I, I, I, I, I],
    "frost_dw_bwd_fused_c1": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, I, P, P, P, P, P, P, P, P, I, I, P],
    "frost_weight_grad_finalize": [P, P, P, P, P, P, I, I, I, I, P, P, P, I, P, P],
    "frost_weight_grad_finalize_table": [P, I, P],
    "frost_infer_weight_prep": [P, I, P],
    "frost_infer_stem_im2col": [P, I, I, I, L, L, L, L, P, P],
    "frost_infer_stem_ok": [I],
    "frost_infer_stem": [P, I, I, I, L, L, L, L, P, P, I, I, P, P],
    "frost_infer_pw": [P, P, P, L, I, I, I, P, P],
    "frost_infer_dw": [P, P, P, I, I, I, I, I, I, I, P, P],
    "frost_infer_cat": [P, I, P, I, L, P, P],
    "frost_infer_add": [P, P, L, P, P],
    "frost_infer_avgpool": [P, I, I, I, P, P],
    "frost_linear_f32": [P, P, P, I, I, I, P, P],
    "frost_float_weight_prep": [P, I, P],
    "frost_float_bn_finalize": [P, I, L, P],
    "frost_float_bn_eval": [P, I, P],
    "frost_float_bwd_finalize": [P, I, L, P],
    "frost_float_pw": [P, P, P, L, I, I, I, I, P, I, P, I, P],
    "frost_float_dw": [P, P, I, I, I, I, I, I, I, I, P, P, P],
    "frost_float_ew": [P, P, L, I, I, I, P, I, P, I, P],
    "frost_float_ew_f32": [P, P, L, I, I, I, P, I, P, I, P],
    "frost_float_dw_dgrad": [P, P, I, I, I, I, I, I, P, P],
    "frost_float_dw_src": [P, P, P, I, I, I, I, I, I, I, I, I, P, P],
    "frost_float_dw_wgrad_src": [P, P, P, I, I, I, I, I, I, I, P, P],
    "frost_float_dw_wgrad": [P, P, I, I, I, I, I, I, P, P],
    "frost_float_pw_wgrad": [P, P, L, I, I, I, P, I, P],
    "frost_float_stem_wscatter": [P, I, P, P],
    "frost_float_grad_merge": [P, P, I, I, P, L, I, P, P],
    "frost_float_avgpool": [P, I, I, I, P, P, P],
    "frost_float_head_bwd": [P, P, P, I, I, I, I, P, P, P, P, P, P],
    "frost_float_pw_f32": [P, P, P, L, I, I, I, I, P, I, P, I, P],
    "frost_float_dw_f32": [P, P, I, I, I, I, I, I, I, I, P, P, P],
    "frost_float_dw_dgrad_f32": [P, P, I, I, I, I, I, I, P, P],
    "frost_float_dw_src_f32": [P, P, P, I, I, I, I, I, I, I, I, I, P, P],
    "frost_float_dw_wgrad_src_f32": [P, P, P, I, I, I, I, I, I, I, P, P],
    "frost_float_dw_wgrad_f32": [P, P, I, I, I, I, I, I, P, P],
    "frost_float_pw_wgrad_f32": [P, P, L, I, I, I, P, I, P],
    "frost_float_grad_merge_f32": [P, P, I, I, P, L, I, P, P],
    "frost_float_avgpool_f32": [P, I, I, I, P, P, P],
    "frost_float_head_bwd_f32": [P, P, P, I, I, I, I, P, P, P, P, P, P],
    "frost_float_cat_f32": [P, I, P, I, L, P, P],
    "frost_float_add_f32": [P, P, L, P, P],
    "frost_float_stem_im2col_f32": [P, I, I, I, L, L, L, L, P, P],
    "frost_infer_block_ok": [I, I, I, I, I, I, I, I, I, I],
    "frost_infer_block_w_ok": [I, I, I, I, I, I, I, I, I],
    "frost_infer_block_w": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P],
    "frost_infer_block": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P],
    "frost_g32_wq": [P, P, P, P, P, I, I, P, P],
    "frost_g32_conv_acc": [P, P, P, I, I, I, I, I, I, I, I, I, P, P],
    "frost_g32_scratch_bytes": [],
    "frost_g32_set_plain": [I],
    "frost_g32_reduce": [P, L, I, P, P, I, P, P, P],
    "frost_g32_dc": [P, L, I, P, P, I, P, P, P],
    "frost_g32_x_ok": [I, L, I, I, I],
    "frost_g32_reduce_x": [P, P, P, I, I, I, I, I, I, I, P, P, I, P, P, P],
    "frost_g32_dc_x": [P, P, P, I, I, I, I, I, I, I, P, P, I, P, P, P],
    "frost_g32_dgrad": [P, P, P, P, I, I, I, I, I, I, I, I, I, P, I, P],
    "frost_g32_wgrad": [P, P, P, I, I, I, I, I, I, I, I, I, P, P, P],
    "frost_g32_cat_bwd": [P, P, P, I, P, P, I, L, P, P, I, P, I, P],
    "frost_g32_add_bwd": [P, P, P, P, P, L, P, P, I, P, I, P],
    "frost_g32_pool_bwd": [P, P, I, I, I, P, P],
    "frost_sq_emit_cat_ok": [I, I],
    "frost_sq_fwd_ok": [L, I, I],
    "frost_sq_fwd_slot_bytes": [L, I],
    "frost_sq_fwd": [P, P, P, P, L, I, I, P, P, P, P, P, P],
    "frost_sq_emit_cat": [P, P, P, P, L, I, I, P, P, P, P, P, I, P],
    "frost_sq_bwd_cat_ok": [I, I],
    "frost_sq_bwd_cat": [P, P, P, P, L, I, I, P, P, I, P, P, P, I, P, I, P],
    "frost_save_sigma": [P, P, I, P],
    "frost_mask_logits": [P, P, P, L, P, P],
    "frost_cat_bwd": [P, P, P, I, P, P, I, L, P, P, I, P, I, P],
    "frost_add_bwd": [P, P, P, P, P, L, P, P, I, P, I, P],
    "frost_head_bwd": [P, P, P, P, I, I, I, I, P, P, P, P, P, P, P],
    "frost_gradboost_step": [P, I, L, P, P, P, P, P],
    "frost_softmax_ce": [P, P, I, I, F, P, P, P],
    "frost_dropout_mask": [P, C.c_uint64, L, F, P, P],
    "frost_conv_finalize_converted": [P, P, P, P, P, P, I, P, P, P],
    "frost_conv_finalize_converted_fb": [P, P, P, P, P, P, I, P, P, P],
    "frost_add_qnnpack": [P, P, P, P, L, P, P, P],
    "frost_avgpool_q": [P, I, I, I, P, P],
    "frost_stem_converted_ok": [I],
    "frost_stem_converted": [P, I, I, I, L, L, L, L, P, P, P, P, P, I, I, P, P],
    "frost_hswish_fwd": [P, P, L, P, P, P, P, I, P, P, P],
    "frost_hswish_bwd": [P, P, L, P, P, I, P],
    "frost_hswish_converted": [P, P, L, P, P, P, P, P],
    "frost_classifier_q": [P, P, P, P, I, I, I, P, P, P, P],
    "frost_classifier_q_fb": [P, P, P, P, I, I, I, P, P, P, P],
}
SYMBOLS = sorted(list(_PROTOS) + ["frost_last_error"])


def load_library():
    """Load (once) and return the ctypes handle. Raises if the shared library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python __graft_entry__.py` (build()) first. "
                           "frostnet_amd has no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int       # (frost_pw_bwd_fused_ok / frost_abi_version return a value, not a status)
    lib.frost_last_error.restype = C.c_char_p
    lib.frost_g32_scratch_bytes.restype = C.c_int64
    lib.frost_sq_fwd_slot_bytes.restype = C.c_int64
    # the sizes a binding must agree on with the library (the ticket buffers grew from 1 to 40 words, FrostFinDesc gained two fields)
    if lib.frost_abi_version() != ABI_VERSION or lib.frost_ticket_words() != TICKET_WORDS or lib.frost_fin_desc_bytes() != C.sizeof(FrostFinDesc):
        raise RuntimeError(f"{LIB_PATH}: ABI mismatch (library abi {lib.frost_abi_version()} / ticket words {lib.frost_ticket_words()} / FrostFinDesc "
                           f"{lib.frost_fin_desc_bytes()} B, binding {ABI_VERSION} / {TICKET_WORDS} / {C.sizeof(FrostFinDesc)} B): rebuild with `python __graft_entry__.py`")
    _lib = lib
    return lib


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Profiler:
    """Per-launch HIP-event timing on the launch stream (bench.py's roofline leg). `only`: restrict to one label."""

    def __init__(self, only=None):
        self.only = only
        self.records = []      # (label, algorithmic_bytes, start_event, end_event)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for label, nbytes, e0, e1 in self.records:
            a = agg.setdefault(label, [0, 0.0, 0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += nbytes
        return {k: dict(launches=v[0], total_ms=v[1], avg_ms=v[1] / v[0], bytes_per_launch=v[2] / v[0]) for k, v in agg.items()}


# Parameter / buffer writes that torch's tensor version counters do NOT see: the optimizer launch and the training forwards update weights and BatchNorm
# running statistics through raw device pointers.  Every such writer calls note_raw_write(); caches keyed on tensor versions (Bf16Inference._prepare_weights)
# compare RAW_WRITE_GEN as well.  A writer captured into a hipGraph is invisible at replay time, so from then on those caches are switched off.
RAW_WRITE_GEN = 0
RAW_WRITES_CAPTURED = False


def note_raw_write():
    global RAW_WRITE_GEN, RAW_WRITES_CAPTURED
    RAW_WRITE_GEN += 1
    if not RAW_WRITES_CAPTURED and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        RAW_WRITES_CAPTURED = True


PROFILER = None
CALL_LOG = None      # tests: set to a list to record the name of every C-ABI entry launched (which kernel family a case engaged)


_ABL_SKIP = frozenset(v for v in os.environ.get("FROST_ABL_SKIP", "").split(",") if v)   # dev, TIMING ONLY: entries not launched at all (wrong results) -- what a family costs inside the real step


def call(name, *args, prof=None):
    """Launch one C-ABI entry on the current stream. prof=(label, algorithmic_bytes) tags it for the Profiler."""
    lib = load_library()
    if _ABL_SKIP and name in _ABL_SKIP:
        return
    p = PROFILER
    if CALL_LOG is not None:
        CALL_LOG.append(name)
    if p is not None and prof is not None and (p.only is None or p.only == prof[0]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        p.records.append((prof[0], prof[1], e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.frost_last_error().decode()}")


def struct_to_tensor(arr, device):
    """ctypes array/struct -> uint8 device tensor (descriptor tables live in device memory)."""
    raw = bytes(memoryview(arr).cast("B"))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
