"""ctypes loader for liblantern_hip.so (the C-ABI declared in include/lantern_hip.h).

The HIP library IS the product path: if it is missing this module raises -- there is no
CPU or PyTorch fallback anywhere in lantern_amd.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LANTERN_HIP_LIBRARY: a library built elsewhere -- a tuning build beside the default one; the default is the in-tree build)
LIB_PATH = os.environ.get("LANTERN_HIP_LIBRARY") or os.path.join(_HERE, "liblantern_hip.so")


class LanternError(RuntimeError):
    pass


class EpParams(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("V", C.c_int32),
        ("rows_per_seq", C.c_int32), ("mode", C.c_int32), ("syntax_shortcut", C.c_int32),
        ("tok_offset", C.c_int32), ("img_lo", C.c_int32), ("img_hi", C.c_int32),
        ("n_syntax", C.c_int32), ("syntax", C.c_int32 * 8),
        ("lantern", C.c_int32), ("k", C.c_int32),
        ("table_rows", C.c_int32), ("table_cols", C.c_int32),
        ("top_k", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float),
        ("delta", C.c_double),
        ("n_uniforms", C.c_int32), ("R", C.c_int32), ("N", C.c_int32), ("row_index_per_seq", C.c_int32),
    ]


class EpBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "logits", "row_index", "cand", "n_paths", "n_depth", "cart_prob", "orig_prob", "op_off", "p_idx",
        "b_off", "b_idx", "tree_cand", "nn_table", "uniforms", "cursor", "workspace",
        "best", "accept_len", "sample_p", "counters")]


class EpWindow(C.Structure):
    _fields_ = [("win_lo", C.c_int32), ("win_len", C.c_int32), ("row_hot", C.c_void_p),
                ("orig_prob_stride", C.c_int32), ("orig_prob_offset", C.c_int32),
                ("sample_win", C.c_void_p), ("out_tok", C.c_void_p), ("out_mass", C.c_void_p),
                ("u_bonus", C.c_void_p), ("token", C.c_void_p), ("rows_kind", C.c_int32), ("raw_pos_per_seq", C.c_int32),
                ("raw_uncond", C.c_void_p), ("raw_pos_ids", C.c_void_p), ("raw_seq_len", C.c_void_p), ("raw_pos_base", C.c_int64),
                ("raw_cfg", C.c_float), ("raw_top_k", C.c_int32), ("raw_w_latent", C.c_int32), ("raw_h_latent", C.c_int32),
                ("raw_newline_id", C.c_int32), ("raw_eos_id", C.c_int32), ("raw_probs", C.c_void_p), ("raw_pre", C.c_void_p), ("verdict_host", C.c_void_p),
                ("turn", C.c_void_p), ("turn_wait", C.c_int64)]


class EpNodes(C.Structure):
    _fields_ = [("tables", C.c_void_p), ("tables_host", C.c_void_p), ("n_nodes", C.c_int32), ("n_internal", C.c_int32), ("n_children", C.c_int32),
                ("max_children", C.c_int32), ("prefix_siblings", C.c_int32), ("leaf_workgroups", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("reserved", C.c_int32 * 2)]


class StepDynamic(C.Structure):
    """lantern_step_dynamic (include/lantern_hip.h): the EAGLE-2 tree stage of a group of lantern_verify_step."""
    _fields_ = ([(n, C.c_void_p) for n in ("scores", "tokens", "parents")]
                + [(n, C.c_int32) for n in ("n_scores", "n_parents", "top_k", "total_tokens", "sort_rows", "reserved")]
                + [(n, C.c_void_p) for n in ("draft_tokens", "mask", "pos_ids", "retrieve", "n_leaf", "max_depth", "seq_len", "retrieve_pd",
                                             "row_index", "pos_abs")])


TURN_WORDS = lambda n_groups: 16 * (1 + 33 * n_groups)          # LANTERN_TURN_WORDS
STEP_CANDIDATES_READY, STEP_PREPARED, STEP_FUSED_PREPARE = 1, 2, 4          # lantern_step_group.flags (include/lantern_hip.h)


class StepGreedy(C.Structure):
    """lantern_step_greedy (include/lantern_hip.h): the greedy / TVD accept as the O8 stage of a group of lantern_verify_step."""
    _fields_ = [("logits", C.c_void_p), ("row_index", C.c_void_p), ("row_index_per_seq", C.c_int32), ("lantern", C.c_int32), ("k", C.c_int32),
                ("delta", C.c_double), ("tok_offset", C.c_int32), ("nn_table", C.c_void_p), ("table_rows", C.c_int32), ("table_cols", C.c_int32),
                ("win_lo", C.c_int32), ("win_len", C.c_int32), ("ok_scratch", C.c_void_p), ("out_row", C.c_void_p), ("token", C.c_void_p)]


class StepDense(C.Structure):
    """lantern_step_dense (include/lantern_hip.h): the dense kernel set as the O7 / O8 stages of a group of lantern_verify_step."""
    _fields_ = [("logits", C.c_void_p), ("sample_p", C.c_void_p), ("u_bonus", C.c_void_p), ("token", C.c_void_p)]


class StepGroup(C.Structure):
    """lantern_step_group (include/lantern_hip.h): one group of sequences of lantern_verify_step."""
    _fields_ = ([("stream", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("ss_token", "ss_prob", "sample_token", "tree_indices", "retrieve")]
                + [(n, C.c_int32) for n in ("B", "n_flat", "N", "P", "D", "reserved0")]
                + [(n, C.c_void_p) for n in ("tree_cand", "cand", "cart_prob", "cond", "uncond")]
                + [("dtype", C.c_int32), ("V", C.c_int32), ("cfg", C.c_float), ("model", C.c_int32), ("pos_ids", C.c_void_p),
                   ("pos_base", C.c_int64)]
                + [(n, C.c_int32) for n in ("w_latent", "h_latent", "img_lo", "img_hi", "newline_id", "eos_id", "top_k", "win_lo", "win_len",
                                            "out_kind")]
                + [("seq_len", C.c_void_p), ("out_win", C.c_void_p), ("row_hot", C.c_void_p), ("temperature", C.c_float), ("top_p", C.c_float),
                   ("ep", EpParams), ("ep_buf", EpBuffers), ("ep_win", EpWindow), ("nodes", C.POINTER(EpNodes))]
                + [(n, C.c_void_p) for n in ("slab_ptrs", "slab_seq", "slab_prev", "new_len")]
                + [("n_slabs", C.c_int32), ("elem_bytes", C.c_int32), ("outer", C.c_int64), ("S_max", C.c_int64), ("d", C.c_int64)]
                + [(n, C.c_void_p) for n in ("hidden", "out_hidden", "accepted_tokens")]
                + [(n, C.c_int32) for n in ("hid_elem_bytes", "hid_groups", "H", "reserved1")]
                + [("node_list", C.c_void_p), ("n_list", C.c_int32), ("flags", C.c_int32)]
                + [("hidden_uncond", C.c_void_p), ("ids_buf", C.c_void_p), ("ids_stride", C.c_int64), ("ids_len", C.c_void_p), ("prepare_next", C.c_void_p)]
                + [("turn", C.c_void_p), ("turn_group", C.c_int32), ("turn_groups", C.c_int32), ("turn_wait", C.c_int64)]
                + [("dyn", C.POINTER(StepDynamic)), ("greedy", C.POINTER(StepGreedy)), ("dense", C.POINTER(StepDense))]
                + [("row_ready", C.c_void_p), ("row_epoch", C.c_int32), ("reserved2", C.c_int32)])


class DraftDepthArgs(C.Structure):
    """lantern_draft_depth_args (include/lantern_hip.h): one drafting depth of the EAGLE-2 drafter."""
    _fields_ = ([("stream", C.c_void_p), ("layer_kind", C.c_int32)]
                + [(n, C.c_int32) for n in ("B", "T", "H", "n_q_heads", "n_kv_heads", "head_dim", "inter", "vocab")]
                + [(n, C.c_float) for n in ("eps1", "eps2", "embed_scale", "cfg")]
                + [(n, C.c_void_p) for n in ("ids", "hidden_in", "embed", "fc_w", "fc_b")]
                + [("fc_packed", C.c_int32), ("layer_packed", C.c_int32)]
                + [(n, C.c_void_p) for n in ("ln1_w", "qkv_w", "qkv_b", "o_w", "o_b", "ln2_w", "gate_up_w", "gate_up_b", "down_w", "down_b",
                                             "qn_w", "qn_b", "kn_w", "kn_b")]
                + [("model_parallel", C.c_int32), ("table_rows", C.c_int32)]
                + [(n, C.c_void_p) for n in ("cos_table", "sin_table", "freqs", "position_ids")]
                + [("positions_per_batch_row", C.c_int32), ("reserved0", C.c_int32), ("k_slab", C.c_void_p), ("v_slab", C.c_void_p),
                   ("kv_rows", C.c_int32), ("kv_row0", C.c_int32), ("tree_bits", C.c_void_p), ("t1", C.c_int32), ("reserved1", C.c_int32),
                   ("kv_start", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p)]
                + [(n, C.c_int32) for n in ("head_packed", "row_lo", "n_cols", "model")]
                + [("head_pos", C.c_void_p), ("pos_base", C.c_int64)]
                + [(n, C.c_int32) for n in ("w_latent", "h_latent", "newline_id", "eos_id", "top_k_filter", "top_k")]
                + [(n, C.c_void_p) for n in ("scores_in", "topk_index", "cu_scores", "topk_cs_index", "scores_out", "hidden_next", "ids_next", "parents_next")]
                + [("parent_bias_next", C.c_int64)]
                + [(n, C.c_void_p) for n in ("x", "xn", "qkv", "q", "attn", "h1", "hn", "act", "out", "head_ws", "sk_ws", "ta_ws")]
                + [("sk_ws_bytes", C.c_size_t), ("ta_ws_bytes", C.c_size_t)]
                # static trees (n_draw > 0): sample instead of expand, next inputs through the tree's tables
                + [("n_draw", C.c_int32), ("T_next", C.c_int32)]
                + [(n, C.c_void_p) for n in ("draw_u", "draw_idx", "probs_out", "ss_token", "ss_prob", "next_gather", "next_rep")]
                # the input stage through the tree's tables (tokens / parent rows one level up read in place)
                + [("in_gather", C.c_void_p), ("in_rep", C.c_void_p), ("in_src_T", C.c_int32), ("in_n_flat", C.c_int32)])


_lib = None


def build(force: bool = False) -> str:
    """Compile liblantern_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j8"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LanternError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  lantern_amd has no CPU fallback.")
        # torch brings its own HIP runtime: it has to be initialised BEFORE this library is loaded into the process (loaded first, the
        # library's launches fail with "no ROCm-capable device is detected" once torch has initialised afterwards -- seen on the GPU box
        # with build() followed by smoke() in one process).  No GPU (the build container): nothing to initialise, the library still loads.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.lantern_last_error.restype = C.c_char_p
        _lib.lantern_evaluate_posterior_workspace.restype = C.c_size_t
        _lib.lantern_tree_attention_workspace.restype = C.c_size_t
        _lib.lantern_evaluate_posterior_nodes_workspace.restype = C.c_size_t
        _lib.lantern_head_expand_workspace.restype = C.c_size_t
        _lib.lantern_linear_rows_streamk_workspace.restype = C.c_size_t
        _lib.lantern_pack_linear_weight_bytes.restype = C.c_size_t
        _lib.lantern_tuning_name.restype = C.c_char_p
    return _lib


def set_tuning(name: str, value: int) -> None:
    """lantern_tuning_set: override a kernel-instance / launch-shape choice for a measurement (include/lantern_hip.h lists the names).  The library
    itself reads no environment variable."""
    check(lib().lantern_tuning_set(name.encode(), int(value)), "tuning_set")


def get_tuning(name: str) -> int:
    v = C.c_int(0)
    check(lib().lantern_tuning_get(name.encode(), C.byref(v)), "tuning_get")
    return v.value


def tuning_names():
    out, i = [], 0
    while True:
        n = lib().lantern_tuning_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def tuning_from_env(environ=None) -> dict:
    """For tools/ and tests only: apply LANTERN_<NAME>=<int> variables of the CALLER's environment as explicit lantern_tuning_set calls (the older
    measurement scripts under tools/run/ pass their settings that way).  Returns what was applied."""
    environ = os.environ if environ is None else environ
    applied = {}
    for n in tuning_names():
        v = environ.get("LANTERN_" + n.upper())
        if v is not None and v.lstrip("-").isdigit():
            set_tuning(n, int(v))
            applied[n] = int(v)
    return applied


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().lantern_last_error()
        raise LanternError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


EXPORTS = [
    "lantern_version", "lantern_last_error", "lantern_tree_static_sizes", "lantern_tree_static_build",
    "lantern_tree_drafter_sizes", "lantern_tree_drafter_build", "lantern_tree_dynamic_finalize", "lantern_tree_dynamic_candidates",
    "lantern_expand_dynamic", "lantern_gather_candidates", "lantern_cfg_mask_topk",
    "lantern_evaluate_posterior_workspace", "lantern_evaluate_posterior", "lantern_evaluate_posterior_greedy",
    "lantern_kv_gather", "lantern_accept_gather", "lantern_sample_static", "lantern_drafter_fc",
    "lantern_build_vq_table", "lantern_cfg_mask_topk_window", "lantern_evaluate_posterior_window",
    "lantern_window_to_dense", "lantern_pack_vq_table", "lantern_update_inference_inputs", "lantern_profile_next_launch", "lantern_drafter_attention_mask", "lantern_linear_rows",
    "lantern_tree_attention_workspace", "lantern_tree_attention",
    "lantern_tree_node_tables_size", "lantern_tree_node_tables", "lantern_evaluate_posterior_nodes_workspace",
    "lantern_evaluate_posterior_nodes", "lantern_verify_step", "lantern_gather_candidates_dynamic", "lantern_head_expand_workspace", "lantern_head_expand", "lantern_prepare_step",
    "lantern_linear_rows_epilogue", "lantern_linear_rows_packed", "lantern_linear_rows_splitk", "lantern_linear_rows_streamk_workspace", "lantern_linear_rows_streamk", "lantern_pack_linear_weight_bytes", "lantern_pack_linear_weight", "lantern_drafter_fc_streamk", "lantern_head_expand_streamk", "lantern_rmsnorm_rows", "lantern_qk_norm_rope", "lantern_qk_rope_pairs", "lantern_draft_depth", "lantern_head_sample", "lantern_draft_static_inputs", "lantern_mask_left_padding",
    "lantern_tuning_set", "lantern_tuning_get", "lantern_tuning_name", "lantern_tuning_reset",
]
