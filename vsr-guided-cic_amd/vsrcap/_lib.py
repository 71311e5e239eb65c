"""ctypes binding of libvsrcap.so (the C ABI declared in include/vsrcap.h).

The product path has NO fallback: if the HIP library is missing or a symbol is absent this module raises,
and every caller above it fails loudly (no eager-PyTorch or CPU substitute exists in this package).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvsrcap.so")

WEIGHT_FIELDS = [
    # (C field, state_dict key) in the order of struct vsr_weights
    ("embed_weight", "embed.weight"),
    ("W1_is_weight", "W1_is.weight"), ("W1_is_bias", "W1_is.bias"),
    ("W1_hs_weight", "W1_hs.weight"), ("W1_hs_bias", "W1_hs.bias"),
    ("att_va_weight", "att_va.weight"), ("att_ha_weight", "att_ha.weight"), ("att_a_weight", "att_a.weight"),
    ("att_sa_weight", "att_sa.weight"), ("att_s_weight", "att_s.weight"),
    ("lstm1_weight_ih", "lstm_cell_1.weight_ih"), ("lstm1_weight_hh", "lstm_cell_1.weight_hh"),
    ("lstm1_bias_ih", "lstm_cell_1.bias_ih"), ("lstm1_bias_hh", "lstm_cell_1.bias_hh"),
    ("lstm2_weight_ih", "lstm_cell_2.weight_ih"), ("lstm2_weight_hh", "lstm_cell_2.weight_hh"),
    ("lstm2_bias_ih", "lstm_cell_2.bias_ih"), ("lstm2_bias_hh", "lstm_cell_2.bias_hh"),
    ("out_fc_weight", "out_fc.weight"), ("out_fc_bias", "out_fc.bias"),
    ("s_fc_weight", "s_fc.weight"), ("s_fc_bias", "s_fc.bias"),
    ("W1_ig_weight", "W1_ig.weight"), ("W1_ig_bias", "W1_ig.bias"),
    ("W1_hg_weight", "W1_hg.weight"), ("W1_hg_bias", "W1_hg.bias"),
    ("att_ga_weight", "att_ga.weight"), ("att_g_weight", "att_g.weight"),
]


class VsrDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "seq_len", "vocab_size", "bos_idx", "det_feat_size", "input_encoding_size", "rnn_size", "att_size",
        "h2_first_lstm", "img_second_lstm")]


class VsrWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f, _ in WEIGHT_FIELDS]


P = C.c_void_p
I32, I64, U64, SZ = C.c_int32, C.c_int64, C.c_uint64, C.c_size_t

# ---- ordering models (include/vsrcap.h: vsr_ssp_layer / vsr_ssp_weights / vsr_sinkhorn_weights)
SSP_LAYER_FIELDS = ["ln1_w", "ln1_b", "ln2_w", "ln2_b", "ln3_w", "ln3_b", "Wq", "bq", "Wk", "bk", "Wv", "bv", "Wo", "bo", "W1", "b1", "W2", "b2"]


class VsrSspLayer(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in SSP_LAYER_FIELDS]


class VsrSspWeights(C.Structure):
    _fields_ = [("sr_embed", C.c_void_p), ("v_embed", C.c_void_p), ("n_verbs", C.c_int64), ("fc_w", C.c_void_p), ("fc_b", C.c_void_p),
                ("enc", VsrSspLayer * 3), ("enc_ln_w", C.c_void_p), ("enc_ln_b", C.c_void_p), ("dec", VsrSspLayer * 3),
                ("dec_ln_w", C.c_void_p), ("dec_ln_b", C.c_void_p), ("exp_w", C.c_void_p), ("exp_b", C.c_void_p)]


SINKHORN_FIELDS = ["W1_txt_w", "W1_txt_b", "W1_vis_w", "W1_vis_b", "W2_vis_w", "W2_vis_b", "W_fc_pos_w", "W_fc_pos_b", "W_fc_w", "W_fc_b"]


class VsrSinkhornWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in SINKHORN_FIELDS] + [("N", C.c_int32), ("n_iters", C.c_int32), ("tau", C.c_float)]


# name -> (restype, argtypes); must list every symbol include/vsrcap.h declares
SIGNATURES = {
    "vsr_abi_version": (I32, []),
    "vsr_last_error": (C.c_char_p, []),
    "vsr_create": (I32, [C.POINTER(VsrDims), C.POINTER(P)]),
    "vsr_destroy": (None, [P]),
    "vsr_bind_weights": (I32, [P, C.POINTER(VsrWeights)]),
    "vsr_set_verb_table": (I32, [P, P, P, I32]),
    "vsr_decode_cache_floats": (SZ, [P]),
    "vsr_build_decode_cache": (I32, [P, P, SZ, P]),
    "vsr_bf16_weight_bytes": (SZ, [P]),
    "vsr_refresh_bf16_weights": (I32, [P, P, SZ, P]),
    "vsr_set_gemm_mode": (I32, [P, I32]),
    "vsr_h2_weight_bytes": (SZ, [P]),
    "vsr_refresh_h2_weights": (I32, [P, P, SZ, P]),
    "vsr_workspace_bytes": (SZ, [P, I32, I32, I32, I32, I32]),
    "vsr_prepare": (I32, [P, P, I32, I32, P, I32, I32, I32, P, SZ, P]),
    "vsr_workspace_bytes_indexed": (SZ, [P, I32, I32, I32, I32, I32, I32, I32]),
    "vsr_prepare_indexed": (I32, [P, P, I32, I32, P, I32, P, I32, P, I32, I32, I32, P, SZ, P]),
    "vsr_row_mask": (I32, [P, I64, I32, P, P]),
    "vsr_reorder_slots": (I32, [P, P, P, P, P, I32, I32, I32, I32, P, P, P]),
    "vsr_greedy": (I32, [P, P, I32, P, P, P]),
    "vsr_sample": (I32, [P, U64, P, P, P, P, P, P, P]),
    "vsr_beam": (I32, [P, I32, I32, I64, I64, P, I32, P, P, P, P, P, P]),
    "vsr_xe_forward": (I32, [P, P, I32, P, P, P]),
    "vsr_train_workspace_bytes": (SZ, [P, I32, I32]),
    "vsr_train_forward": (I32, [P, P, P, I32, P, P, P, SZ, P]),
    "vsr_train_backward": (I32, [P, P, P, C.POINTER(VsrWeights), P]),
    "vsr_train_generation": (I64, [P]),
    "vsr_train_select": (I32, [P, I64, P]),
    "vsr_train_bucket_map": (I32, [C.POINTER(I32), C.POINTER(I32)]),
    "vsr_train_wait_bucket": (I32, [P, I32, P]),
    "vsr_bad_ids": (I32, [P, C.POINTER(I32), P]),
    "vsr_set_valid_rows_bound": (I32, [P, I64]),
    "vsr_debug_copy": (I32, [P, C.c_char_p, P, SZ, P]),
    "vsr_cider_rewards": (I32, [P, P, P, C.c_double, P, I32, I32, P, I32, I32, I64, I64, P, I32, C.c_double, P, P]),
    "vsr_ssp_create": (I32, [C.POINTER(P)]),
    "vsr_ssp_destroy": (None, [P]),
    "vsr_ssp_bind": (I32, [P, C.POINTER(VsrSspWeights), C.POINTER(VsrSinkhornWeights)]),
    "vsr_ssp_workspace_bytes": (SZ, [I32]),
    "vsr_ssp_generate": (I32, [P, P, P, I32, P, P, P, SZ, P]),
    "vsr_sinkhorn_workspace_bytes": (SZ, [I32, I32]),
    "vsr_sinkhorn_assign": (I32, [P, P, I32, P, P, P, SZ, P]),
    "vsr_profile_begin": (I32, [P]),
    "vsr_profile_begin_sampled": (I32, [P, I32]),
    "vsr_profile_seen": (I64, [P]),
    "vsr_profile_bytes": (C.c_double, [P]),
    "vsr_profile_end": (I32, [P, P, C.POINTER(C.c_double), C.POINTER(I64), C.POINTER(C.c_double)]),
    "vsr_step": (I32, [P, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, P, P, P]),
}

_lib = None


def load():
    """Load libvsrcap.so once; raise if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (libamdhip64): it must be the one already mapped when libvsrcap.so is
    # loaded, because every device pointer and stream handed to the library comes from torch.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvsrcap.so not found at %s: build it with `python vsr-guided-cic_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for this path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    if lib.vsr_abi_version() != 1:
        raise RuntimeError("libvsrcap.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libvsrcap: " + load().vsr_last_error().decode())
