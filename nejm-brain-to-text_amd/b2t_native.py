"""ctypes binding of libb2t_hip.so (the C ABI declared in include/b2t.h).

The library is the product's compute path: there is NO CPU/PyTorch fallback.  `load()` raises if
the shared object is missing or lacks a declared symbol; every wrapper raises RuntimeError with
b2t_last_error() when an entry point reports failure.
"""
from __future__ import annotations

import ctypes as C
import os
import re

# The pipelined execution plan (b2t_ops) runs the GRU layers on 2L+1 HIP streams.  ROCm maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise, which stalls the persistent
# sweeps behind the kernels they wait for.  Must be set before the HIP runtime initialises (first torch.cuda use).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("B2T_LIB") or os.path.join(_HERE, "csrc", "libb2t_hip.so")   # B2T_LIB: A/B builds of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "b2t.h")

_lib = None

c_f32p = C.c_void_p
c_i32p = C.c_void_p
VP = C.c_void_p
LL = C.c_longlong


class GemmDesc(C.Structure):
    """Mirror of b2t_gemm_desc (include/b2t.h)."""
    _fields_ = [
        ("A", VP), ("B", VP), ("C", VP), ("bias", VP),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("Z", C.c_int),
        ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
        ("a_s0", LL), ("a_s1", LL), ("a_div", C.c_int), ("a_sz", LL),
        ("b_s0", LL), ("b_s1", LL), ("b_div", C.c_int), ("b_sz", LL),
        ("c_s0", LL), ("c_s1", LL), ("c_div", C.c_int), ("c_sz", LL),
        ("b_zmap", VP), ("bias_sz", LL),
        ("epilogue", C.c_int), ("accumulate", C.c_int),
        ("splitk", C.c_int), ("c_ks", LL),
        ("a_brk", C.c_int), ("a_gap", C.c_int), ("ep_aux", VP), ("a_sum", VP), ("a_sum_ks", C.c_longlong), ("ks_counters", VP), ("ks_out", VP), ("ks_accumulate", C.c_int),
    ]


MAX_LAYERS = 8
FP = C.c_void_p   # float* fields of the structs below are filled from tensor.data_ptr()


class ModelDesc(C.Structure):
    """Mirror of b2t_model_t (include/b2t.h): parameter or gradient tensors under the reference's names."""
    _fields_ = [("F", C.c_int), ("H", C.c_int), ("D", C.c_int), ("C", C.c_int), ("L", C.c_int), ("patch", C.c_int),
                ("stride", C.c_int), ("day_w", FP), ("day_b", FP), ("day_w_stride", LL), ("day_b_stride", LL),
                ("w_ih", FP * MAX_LAYERS), ("w_hh", FP * MAX_LAYERS), ("b_ih", FP * MAX_LAYERS), ("b_hh", FP * MAX_LAYERS),
                ("out_w", FP), ("out_b", FP), ("h0", FP)]


class PassDesc(C.Structure):
    """Mirror of b2t_pass_t (include/b2t.h)."""
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("chunks", C.c_int), ("fwd_mode", C.c_int), ("bwd_mode", C.c_int),
                ("bf16_gemm", C.c_int), ("save", C.c_int), ("in_drop", C.c_float), ("rnn_drop", C.c_float),
                ("seed", C.c_uint64), ("chunks_bwd", C.c_int), ("wgrad_chunk_mask", C.c_int)]


BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


class WaveDesc(C.Structure):
    """Mirror of b2t_wave_t (include/b2t.h): the GRU stack's sweeps as one launch per direction (the layer wavefront)."""
    _fields_ = [("L", C.c_int), ("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("gi0", FP),
                ("w_hh", FP * MAX_LAYERS), ("b_hh", FP * MAX_LAYERS), ("w_ih", FP * MAX_LAYERS), ("b_ih", FP * MAX_LAYERS),
                ("h_init", FP * MAX_LAYERS), ("out", FP * MAX_LAYERS), ("outd", FP * MAX_LAYERS), ("reserve", FP * MAX_LAYERS),
                ("dY_top", FP), ("dh_last", FP), ("dh_init", FP),
                ("w_hh_t", FP * MAX_LAYERS), ("w_ih_t", FP * MAX_LAYERS), ("dG", FP * MAX_LAYERS),
                ("drop_p", C.c_float), ("seed", C.c_uint64 * MAX_LAYERS), ("elem0", LL)]


class WfstGraph(C.Structure):
    """Mirror of b2t_wfst_graph_t (include/b2t.h)."""
    _fields_ = [(n, VP) for n in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final_cost")] + \
               [("n_states", C.c_int32), ("start", C.c_int32), ("labels", VP), ("weight_f16", VP), ("compact", C.c_int32)]


class WfstOpts(C.Structure):
    """Mirror of b2t_wfst_opts_t (include/b2t.h)."""
    _fields_ = [(n, C.c_float) for n in ("beam", "lattice_beam", "beam_delta", "acoustic_scale", "length_penalty", "blank_skip_thresh")] + \
               [(n, C.c_int32) for n in ("max_active", "min_active", "max_frames", "max_tokens", "max_links", "hash_size")]


class LexLmDesc(C.Structure):
    """Mirror of b2t_lexlm_t (include/b2t.h)."""
    _fields_ = [(n, VP) for n in ("lex_child", "lex_wbeg", "lex_wend", "wlist", "lm_cb", "lm_ce", "lm_ctok", "lm_cnode",
                                  "lm_logp", "lm_bow", "lm_suffix", "lm_nstate")] + \
               [("lm_start_state", C.c_int32), ("lm_eos", C.c_int32), ("sil", C.c_int32),
                ("alpha", C.c_float), ("beta", C.c_float), ("unk_logp", C.c_float)]


_SIGNATURES = {
    "b2t_version": (C.c_int, []),
    "b2t_last_error": (C.c_char_p, []),
    "b2t_augment_smooth_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_uint64, VP, VP, C.POINTER(C.c_float), C.c_int, C.c_int, VP]),
    "b2t_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), VP]),
    "b2t_gemm_bf16_f32": (C.c_int, [C.POINTER(GemmDesc), VP]),
    "b2t_gemm_bf16p_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "b2t_gemm_bf16p_ws_bytes_z": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2t_gemm_bf16p_f32": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "b2t_softsign_bwd_f32": (C.c_int, [VP, VP, LL, VP]),
    "b2t_adjusted_lens_i32": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "b2t_colsum_ws_bytes": (C.c_size_t, [LL, C.c_int]),
    "b2t_colsum_f32": (C.c_int, [VP, LL, C.c_int, LL, VP, C.c_int, VP, C.c_int, LL, LL, VP]),
    "b2t_slab_reduce_f32": (C.c_int, [VP, C.c_int, LL, VP, C.c_int, VP]),
    "b2t_day_reduce_f32": (C.c_int, [VP, VP, C.c_int, LL, VP, LL, VP]),
    "b2t_patch_fold_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "b2t_patch_fold_day_bwd_f32": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, VP]),
    "b2t_dropout_f32": (C.c_int, [VP, VP, LL, C.c_float, C.c_uint64, LL, VP]),
    "b2t_dropout_mask_f32": (C.c_int, [VP, LL, C.c_float, C.c_uint64, LL, VP]),
    "b2t_batch_gather_b32": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "b2t_gru_wave_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2t_gru_wave_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2t_gru_wave_fwd_f32": (C.c_int, [C.POINTER(WaveDesc), VP, VP, VP]),
    "b2t_gru_wave_bwd_f32": (C.c_int, [C.POINTER(WaveDesc), VP, VP, VP]),
    "b2t_gru_sync_bytes": (C.c_size_t, [C.c_int]),
    "b2t_gru_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "b2t_gru_sync_status": (C.c_int, [VP, C.c_int, C.c_int, C.POINTER(C.c_int), VP]),
    "b2t_gru_layer_fwd_f32": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "b2t_gru_layer_fwd_fused_f32": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "b2t_gru_layer_bwd_f32": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int,
                                        VP, VP]),
    "b2t_transpose_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, VP]),
    "b2t_cumsum_add_f32": (C.c_int, [VP, VP, LL, C.c_int, LL, VP]),
    "b2t_broadcast_rows_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, VP]),
    "b2t_exec_create": (C.c_int, [C.c_int, C.POINTER(VP)]),
    "b2t_exec_destroy": (C.c_int, [VP]),
    "b2t_plan_schedule_host": (C.c_int, [C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP]),
    "b2t_plan_admission_host": (C.c_int, [C.c_int, VP, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP]),
    "b2t_exec_sync_bytes": (C.c_size_t, [C.c_int]),
    "b2t_exec_graph_stats": (C.c_int, [VP, VP, VP, VP]),
    "b2t_pass_ws_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.POINTER(PassDesc)]),
    "b2t_model_forward": (C.c_int, [VP, C.POINTER(ModelDesc), C.POINTER(PassDesc), VP, VP, VP, VP, VP, VP, VP, VP]),
    "b2t_copy_segments_b32": (C.c_int, [VP, VP, VP, C.c_int, VP]),
    "b2t_copy_indirect_b32": (C.c_int, [VP, C.c_int, VP]),
    "b2t_stream_supported": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "b2t_stream_ws_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "b2t_stream_sync_bytes": (C.c_size_t, []),
    "b2t_stream_forward_f32": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, C.c_size_t, VP, VP]),
    "b2t_model_backward": (C.c_int, [VP, C.POINTER(ModelDesc), C.POINTER(ModelDesc), C.POINTER(PassDesc), VP, VP, VP,
                                     C.c_int, VP, VP, C.c_int, VP, VP, BUCKET_CB, VP, VP]),
    "b2t_exec_profile": (C.c_int, [VP, C.c_int]),
    "b2t_exec_profile_read": (C.c_int, [VP, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int]),
    "b2t_ctc_loss_f32": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, VP]),
    "b2t_opt_prepare": (C.c_int, [VP, C.c_int, VP, C.c_int, VP, VP]),
    "b2t_grad_norm_clip_f32": (C.c_int, [VP, VP, VP, C.c_int, C.c_float, VP, VP, VP, C.c_int, VP, C.c_int, LL, VP]),
    "b2t_opt_advance": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "b2t_adamw_f32": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, VP, C.c_int, C.POINTER(C.c_float),
                                C.POINTER(C.c_float), C.c_double, C.c_double, C.c_float, VP]),
    "b2t_greedy_decode_f32": (C.c_int, [VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "b2t_edit_distance_i32": (C.c_int, [VP, VP, C.c_int, VP, VP, C.c_int, VP, C.c_int, VP]),
    "b2t_fst_from_arrays": (VP, [C.c_int, C.c_int, LL, VP, VP, VP, VP, VP, VP]),
    "b2t_fst_free": (None, [VP]),
    "b2t_fst_info": (C.c_int, [VP, C.POINTER(LL)]),
    "b2t_fst_to_arrays": (C.c_int, [VP, VP, VP, VP, VP, VP, VP]),
    "b2t_fst_compose": (VP, [VP, VP]),
    "b2t_fst_trim": (VP, [VP]),
    "b2t_fst_determinize_star": (VP, [VP, C.c_int, C.c_float, LL]),
    "b2t_fst_minimize_encoded": (VP, [VP, C.c_float]),
    "b2t_fst_arcsort": (VP, [VP, C.c_int]),
    "b2t_fst_read_openfst": (VP, [C.c_char_p]),
    "b2t_fst_write_openfst": (C.c_int, [VP, C.c_char_p]),
    "b2t_fst_grammar_score": (C.c_double, [VP, VP, C.c_int, C.c_int]),
    "b2t_fst_prepare_lm": (VP, [VP, C.c_int, VP]),
    "b2t_wfst_state_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "b2t_wfst_cluster_size": (C.c_int, [C.c_int]),
    "b2t_wfst_set_cluster": (C.c_int, [C.c_int]),
    "b2t_wfst_reset": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, C.c_int, VP]),
    "b2t_wfst_search_f32": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "b2t_wfst_best_path": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP,
                                    VP, VP, VP]),
    "b2t_wfst_finalize": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, C.c_int, VP]),
    "b2t_wfst_prune": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, C.c_int, C.c_float, C.c_float, VP]),
    "b2t_wfst_lattice": (C.c_int, [C.POINTER(WfstGraph), C.POINTER(WfstOpts), VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP,
                                  VP, VP, VP, VP, VP]),
    "b2t_wfst_state_offsets": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(LL)]),
    "b2t_lattice_nbest_host": (C.c_int, [C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, C.c_int, VP, VP, C.c_int,
                                         C.c_float, VP, VP, C.c_int, VP, VP, C.c_int, VP]),
    "b2t_lattice_rescore_nbest_host": (C.c_int, [C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, C.c_int,
                                                 C.c_int, C.c_float, VP, VP, C.c_int, VP, VP, C.c_int, VP, VP]),
    "b2t_nbest_convert_to_inputs": (C.c_int, [VP, VP, C.c_int, VP, C.c_int, VP, VP, VP, C.c_int]),
    "b2t_lm_prologue_f32": (C.c_int, [VP, VP, C.c_float, VP, C.c_int, C.c_int, VP]),
    "b2t_beam_state_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "b2t_beam_reset": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP]),
    "b2t_prefix_beam_search_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int,
                                             C.c_int, VP, VP, VP, VP, VP, VP]),
    "b2t_beam_overflowed": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), VP]),
    "b2t_prefix_beam_search_lex_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int,
                                                 C.c_int, VP, VP, VP, VP, VP, VP, VP, VP]),
    "b2t_prefix_beam_search_lm_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, C.c_int,
                                                C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_float, C.c_float, VP, VP]),
}


def header_symbols():
    """Every function name declared in include/b2t.h."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(b2t_[a-z0-9_]+)\s*\(", txt)))


def load():
    """Load libb2t_hip.so; fail loudly if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path of this package "
            f"(no CPU fallback). Build it with `python __graft_entry__.py`.")
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library is
    # loaded, so that both bind to the same runtime instance: loading ours first pulls in /opt/rocm's copy, and the
    # device pointers torch hands us then belong to a different runtime ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name in header_symbols():
        if not hasattr(lib, name):
            raise RuntimeError(f"libb2t_hip.so does not export {name} declared in include/b2t.h")
        if name not in _SIGNATURES:
            raise RuntimeError(f"b2t_native has no ctypes signature for {name}")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.b2t_version() != 2:
        raise RuntimeError("libb2t_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error() -> str:
    return load().b2t_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
