// verify_step.cpp -- lantern_verify_step: one verify step of G groups of sequences enqueued by ONE host call
// (include/lantern_hip.h).  Pure launch sequencing over the library's own entry points -- no kernel of its own.
//
// Reference: the body of EaLumina_mGPT.generate's decode loop, models/ea_model_lumina_mgpt.py:936-998.
#include "../../include/lantern_hip.h"

namespace lantern {
void set_error(const char *fmt, ...);
}

extern "C" int lantern_verify_step(const lantern_step_group *groups, int n_groups) {
    if (!groups || n_groups < 0) {
        lantern::set_error("verify_step: bad arguments");
        return LANTERN_E_INVALID;
    }
    // stage by stage across the groups: every stream receives its next kernel before any stream receives the one after
    // (the streams then advance side by side instead of one group running a whole step ahead of the others)
    int rc;
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.node_list && s.n_list > 0) {          // candidates + the likely rows in one launch
            rc = lantern_prepare_step(&s);
            if (rc) return rc;
            continue;
        }
        rc = lantern_gather_candidates(s.ss_token, s.ss_prob, s.sample_token, s.tree_indices, s.retrieve, s.B, s.n_flat, s.N, s.P, s.D,
                                       s.tree_cand, s.cand, s.cart_prob, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (!s.out_win || (s.node_list && s.n_list > 0)) continue;          // LANTERN_ROWS_RAW_BF16: evaluate_posterior post-processes the rows it visits itself
        rc = lantern_cfg_mask_topk_window(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent,
                                          s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k, s.seq_len, s.N, s.win_lo,
                                          s.win_len, s.out_win, s.row_hot, s.out_kind, s.temperature, s.top_p, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        rc = s.nodes ? lantern_evaluate_posterior_nodes(&s.ep, &s.ep_buf, &s.ep_win, s.nodes, s.stream)
                     : lantern_evaluate_posterior_window(&s.ep, &s.ep_buf, &s.ep_win, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (!s.slab_ptrs) continue;
        rc = lantern_update_inference_inputs(s.slab_ptrs, s.slab_seq, s.slab_prev, s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d,
                                             s.retrieve, 0, s.P, s.D, s.ep_buf.best, s.ep_buf.accept_len, s.new_len, s.hidden,
                                             s.hid_elem_bytes, s.B, s.hid_groups, s.N, s.H, s.cand, s.out_hidden, s.accepted_tokens,
                                             s.stream);
        if (rc) return rc;
    }
    return LANTERN_OK;
}
