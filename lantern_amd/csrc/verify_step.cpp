// verify_step.cpp -- lantern_verify_step: one verify step of G groups of sequences enqueued by ONE host call
// (include/lantern_hip.h).  Pure launch sequencing over the library's own entry points -- no kernel of its own.
//
// Reference: the body of EaLumina_mGPT.generate's decode loop, models/ea_model_lumina_mgpt.py:936-998 (static trees), and the same loop
// behind the EAGLE-2 drafter (cnets_lumina_mgpt.py:1337-1420 / cnets_llamagen.py:826-912: per-sequence trees).
#include "prep_dev.h"
#include "gather_dev.h"
#include <cstdio>

namespace lantern {
void set_error(const char *fmt, ...);
const char *last_error();
int launch_update_inference_inputs(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev, int n_slabs, int elem_bytes, int64_t outer,
                                   int64_t S_max, int64_t d, const int64_t *retrieve, int retrieve_per_seq, int P, int D, const int32_t *best,
                                   const int32_t *accept_len, int64_t *new_len, const void *hidden, int hid_elem_bytes, int B, int G, int N, int H,
                                   const int64_t *cand, void *out_hidden, int64_t *accepted_tokens, const int32_t *counters, void *stream,
                                   const void *hidden_g1 = nullptr, int64_t *ids_buf = nullptr, int64_t ids_stride = 0, const int64_t *ids_len = nullptr,
                                   const int64_t *bonus = nullptr, const PrepArgs *prep = nullptr, const TurnArgs *turn = nullptr);
int prepare_step_args(const lantern_step_group *g, PrepArgs *out);
int evaluate_posterior_window_fused(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win, const PrepArgs &prep, int32_t *ready,
                                    int32_t epoch, void *stream);
}

namespace {
// the failing entry point's message, prefixed with where in the step it was raised
int fail(int g, const char *stage, int rc) {
    char msg[400];
    snprintf(msg, sizeof msg, "%s", lantern::last_error());
    lantern::set_error("verify_step: group %d, %s: %s", g, stage, msg);
    return rc;
}
}  // namespace

extern "C" int lantern_verify_step(const lantern_step_group *groups, int n_groups) {
    if (!groups || n_groups < 0) {
        lantern::set_error("verify_step: bad arguments");
        return LANTERN_E_INVALID;
    }
    // stage by stage across the groups: every stream receives its next kernel before any stream receives the one after
    // (the streams then advance side by side instead of one group running a whole step ahead of the others)
    int rc;
    for (int g = 0; g < n_groups; ++g) {          // (before anything is launched: a walk that waits for a turn nobody will release spins out its bound)
        const lantern_step_group &s = groups[g];
        if (s.dense) {
            const lantern_step_dense &q = *s.dense;
            if (!q.logits || !q.sample_p || (q.token != nullptr) != (q.u_bonus != nullptr) || s.nodes || s.greedy || (s.node_list && s.n_list > 0) || s.prepare_next ||
                s.turn || s.ep_win.verdict_host || (s.flags & LANTERN_STEP_PREPARED)) {
                lantern::set_error("dense step: logits / sample_p, token and u_bonus together, and none of nodes / greedy / node_list / prepare_next / turn / "
                                   "ep_win.verdict_host / LANTERN_STEP_PREPARED");
                return fail(g, "dense", LANTERN_E_INVALID);
            }
        }
        if ((s.flags & LANTERN_STEP_FUSED_PREPARE) &&
            (s.dyn || s.nodes || s.greedy || s.dense || s.prepare_next || !s.node_list || s.n_list < 1 || !s.row_ready || (s.flags & (LANTERN_STEP_PREPARED | LANTERN_STEP_CANDIDATES_READY)))) {
            lantern::set_error("LANTERN_STEP_FUSED_PREPARE: a static-tree group with a node list (the root first) and row_ready, on the chain kernel, and none of dyn / nodes / "
                               "greedy / dense / prepare_next / LANTERN_STEP_PREPARED / LANTERN_STEP_CANDIDATES_READY");
            return fail(g, "fused prepare", LANTERN_E_INVALID);
        }
        if (s.turn && (s.turn_groups <= 0 || s.turn_group < 0 || s.turn_group >= s.turn_groups || !s.slab_ptrs)) {
            lantern::set_error("commit turn-taking: turn_group in [0, turn_groups) and the group's KV slabs (its commit launch releases the turn)");
            return fail(g, "turn", LANTERN_E_INVALID);
        }
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.flags & LANTERN_STEP_FUSED_PREPARE) continue;          // (the prepare stage rides in the chain launch below)
        if (s.flags & LANTERN_STEP_PREPARED) {          // the previous call's commit launch prepared this step (prepare_next)
            // only what prepare_next can have prepared: a static-tree group with a node list (an EAGLE-2 group would skip its tree build).  A
            // failed call or a non-zero walk status in the previous step invalidates the preparation: the caller clears the flag and the step
            // prepares itself (include/lantern_hip.h)
            if (s.dyn || !s.node_list || s.n_list <= 0) {
                lantern::set_error("LANTERN_STEP_PREPARED: only a static-tree group with a node list can have been prepared by prepare_next");
                return fail(g, "prepare_step", LANTERN_E_INVALID);
            }
            continue;
        }
        if (s.dyn && s.node_list && s.n_list > 0) { // EAGLE-2 tree + candidates + the likely rows in one launch
            rc = lantern_prepare_step(&s);
            if (rc) return fail(g, "prepare_step", rc);
        } else if (s.dyn) {                         // EAGLE-2: every sequence's own tree out of this step's drafter scores
            const lantern_step_dynamic &d = *s.dyn;
            rc = lantern_tree_dynamic_candidates(d.scores, d.tokens, d.parents, s.sample_token, s.B, d.n_scores, d.n_parents, d.top_k,
                                                 d.total_tokens, d.sort_rows, d.draft_tokens, d.mask, d.pos_ids, d.retrieve, d.n_leaf,
                                                 d.max_depth, d.seq_len, s.P, s.D, s.cand, d.retrieve_pd, d.row_index, d.pos_abs, s.stream);
            if (rc) return fail(g, "tree_dynamic_candidates", rc);
        } else if (s.node_list && s.n_list > 0) {   // candidates + the likely rows in one launch
            rc = lantern_prepare_step(&s);
            if (rc) return fail(g, "prepare_step", rc);
        } else if (s.flags & LANTERN_STEP_CANDIDATES_READY) {
            // the caller assembled the candidates itself (a tree that came with its token list, ea_model_llamagen.py:1125-1131; or O6 called before the
            // target forward): said explicitly, and what the later stages read must be there
            if (s.ss_token || !s.cand || !s.retrieve || s.P <= 0 || s.D <= 0 || s.B < 0) {
                lantern::set_error("LANTERN_STEP_CANDIDATES_READY needs cand [B,P,D] and retrieve [P,D] with P, D > 0 and ss_token NULL");
                return fail(g, "gather_candidates", LANTERN_E_INVALID);
            }
        } else {
            // (a zero-initialised or half-filled group lands here: lantern_gather_candidates refuses its NULL buffers instead of the step
            // evaluating whatever `cand` happens to hold)
            rc = lantern_gather_candidates(s.ss_token, s.ss_prob, s.sample_token, s.tree_indices, s.retrieve, s.B, s.n_flat, s.N, s.P,
                                           s.D, s.tree_cand, s.cand, s.cart_prob, s.stream);
            if (rc) return fail(g, "gather_candidates", rc);
        }
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.greedy) {          // greedy decoding: CFG + model mask of every row, dense f32 (no processor list: ea_model_llamagen.py:930)
            const lantern_step_greedy &q = *s.greedy;
            if (!q.logits || !q.row_index || !q.ok_scratch || !q.out_row || !q.token || s.dyn || s.nodes || s.prepare_next || s.ep_win.verdict_host) {
                lantern::set_error("greedy step: logits / row_index / ok_scratch / out_row / token, and none of dyn / nodes / prepare_next / ep_win.verdict_host "
                                   "(no greedy kernel writes the pinned verdict record)");
                return fail(g, "greedy", LANTERN_E_INVALID);
            }
            rc = lantern_cfg_mask_topk(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent, s.h_latent, s.img_lo,
                                       s.img_hi, s.newline_id, s.eos_id, 0, s.seq_len, s.N, q.logits, s.stream);
            if (rc) return fail(g, "cfg_mask_topk", rc);
            continue;
        }
        if (s.dense) {           // the dense kernel set: every row at the full vocabulary (the reference's own intermediate tensor, ea_model_lumina_mgpt.py:597-605)
            if (!s.cond) continue;          // (the caller's rows are already in dense->logits)
            rc = lantern_cfg_mask_topk(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent, s.h_latent, s.img_lo,
                                       s.img_hi, s.newline_id, s.eos_id, s.top_k, s.seq_len, s.N, s.dense->logits, s.stream);
            if (rc) return fail(g, "cfg_mask_topk", rc);
            continue;
        }
        if (!s.out_win || (s.node_list && s.n_list > 0)) continue;          // LANTERN_ROWS_RAW_BF16: evaluate_posterior post-processes the rows it visits itself
        rc = lantern_cfg_mask_topk_window(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent,
                                          s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k, s.seq_len, s.N, s.win_lo,
                                          s.win_len, s.out_win, s.row_hot, s.out_kind, s.temperature, s.top_p, s.stream);
        if (rc) return fail(g, "cfg_mask_topk_window", rc);
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.greedy) {
            const lantern_step_greedy &q = *s.greedy;
            rc = lantern_evaluate_posterior_greedy(q.logits, q.row_index, s.cand, s.B, s.P, s.D, s.V, s.N, q.row_index_per_seq, q.lantern, q.k, q.delta,
                                                   q.tok_offset, q.nn_table, q.table_rows, q.table_cols, q.win_lo, q.win_len, q.ok_scratch, s.ep_buf.best,
                                                   s.ep_buf.accept_len, q.out_row, s.stream);
            if (rc) return fail(g, "evaluate_posterior_greedy", rc);
            // the bonus token: argmax of the accepted row, first maximum (torch.argmax; ea_model_llamagen.py:664)
            rc = lantern_accept_gather(nullptr, 0, s.B, 1, s.N, 0, s.retrieve, 0, s.P, s.D, nullptr, s.ep_buf.best, s.ep_buf.accept_len, q.out_row, s.V, nullptr,
                                       nullptr, nullptr, q.token, s.stream);
            if (rc) return fail(g, "bonus argmax", rc);
            continue;
        }
        if (s.dense) {
            const lantern_step_dense &q = *s.dense;
            lantern_ep_buffers b = s.ep_buf;
            b.logits = q.logits;
            b.sample_p = q.sample_p;
            rc = lantern_evaluate_posterior(&s.ep, &b, s.stream);
            if (rc) return fail(g, "evaluate_posterior (dense)", rc);
            if (q.token) {          // the bonus token: inverse CDF of sample_p at this step's uniform (ea_model_lumina_mgpt.py:779-790)
                rc = lantern_accept_gather(nullptr, 0, s.B, 1, s.N, 0, s.dyn ? s.dyn->retrieve_pd : s.retrieve, s.dyn ? 1 : 0, s.P, s.D, nullptr, s.ep_buf.best,
                                           s.ep_buf.accept_len, q.sample_p, s.V, q.u_bonus, nullptr, nullptr, q.token, s.stream);
                if (rc) return fail(g, "bonus draw", rc);
            }
            continue;
        }
        if (s.flags & LANTERN_STEP_FUSED_PREPARE) {          // candidates + listed rows + walk in ONE launch
            lantern::PrepArgs pa{};
            rc = lantern::prepare_step_args(&s, &pa);
            if (rc) return fail(g, "fused prepare", rc);
            lantern_ep_window w = s.ep_win;
            if (s.turn && s.slab_ptrs) {
                w.turn = s.turn;
                w.turn_wait = s.turn_wait;
            }
            rc = lantern::evaluate_posterior_window_fused(&s.ep, &s.ep_buf, &w, pa, s.row_ready, s.row_epoch, s.stream);
            if (rc) return fail(g, "evaluate_posterior (fused prepare)", rc);
            continue;
        }
        if (s.turn && !s.nodes && s.slab_ptrs) {          // commit turn-taking: the chain kernel ends when it is this group's turn to commit
            lantern_ep_window w = s.ep_win;
            w.turn = s.turn;
            w.turn_wait = s.turn_wait;
            rc = lantern_evaluate_posterior_window(&s.ep, &s.ep_buf, &w, s.stream);
        } else
        rc = s.nodes ? lantern_evaluate_posterior_nodes(&s.ep, &s.ep_buf, &s.ep_win, s.nodes, s.stream)
                     : lantern_evaluate_posterior_window(&s.ep, &s.ep_buf, &s.ep_win, s.stream);
        if (rc) return fail(g, "evaluate_posterior", rc);
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (!s.slab_ptrs) {
            if (s.turn) {          // (nothing would ever release the turn)
                lantern::set_error("commit turn-taking needs the group's commit launch: slab_ptrs is NULL");
                return fail(g, "update_inference_inputs", LANTERN_E_INVALID);
            }
            if (s.prepare_next) {          // (the next call would skip its preparation for nothing)
                lantern::set_error("prepare_next rides in the commit launch: the group needs its KV slabs (slab_ptrs)");
                return fail(g, "update_inference_inputs", LANTERN_E_INVALID);
            }
            continue;
        }
        // the next step's preparation in the same launch (prepare_next): its lengths are this step's + what this commit adds
        lantern::PrepArgs pa{};
        const lantern::PrepArgs *prep = nullptr;
        if (s.prepare_next) {
            const lantern_step_group &nx = *s.prepare_next;
            if (nx.dyn || !nx.node_list || nx.n_list <= 0 || nx.B != s.B || nx.cand == s.cand) {
                lantern::set_error("prepare_next: a static-tree group with a node list, the same sequences and its OWN candidate buffer");
                return fail(g, "update_inference_inputs", LANTERN_E_INVALID);
            }
            rc = lantern::prepare_step_args(&nx, &pa);
            if (rc) return fail(g, "prepare_next", rc);
            pa.seq_len = s.seq_len;          // THIS step's lengths (the commit beside these blocks is still writing the next ones) ...
            pa.len_alen = s.ep_buf.accept_len;          // ... + what it adds
            pa.len_cnt = s.ep_buf.counters;
            prep = &pa;
        }
        const lantern::TurnArgs ta{s.turn, s.turn_group, s.turn_groups};
        // (a sequence whose walk reported a status commits nothing: its KV rows and lengths stay as the forward left them, its out_hidden rows are
        // zero-filled and its accepted_tokens are -1 -- the caller retries the step and commits it itself; tests/test_gpu_loop.py pins this gate)
        rc = lantern::launch_update_inference_inputs(s.slab_ptrs, s.slab_seq, s.slab_prev, s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d,
                                                     s.dyn ? s.dyn->retrieve_pd : s.retrieve, s.dyn ? 1 : 0, s.P, s.D, s.ep_buf.best,
                                                     s.ep_buf.accept_len, s.new_len, s.hidden, s.hid_elem_bytes, s.B, s.hid_groups, s.N, s.H,
                                                     s.cand, s.out_hidden, s.accepted_tokens, s.ep_buf.counters, s.stream, s.hidden_uncond, s.ids_buf,
                                                     s.ids_stride, s.ids_len, s.ids_buf ? (s.greedy ? s.greedy->token : (s.dense ? s.dense->token : s.ep_win.token)) : nullptr, prep,
                                                     s.turn ? &ta : nullptr);
        if (rc) return fail(g, "update_inference_inputs", rc);
    }
    return LANTERN_OK;
}
