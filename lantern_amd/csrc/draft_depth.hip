// draft_depth.hip -- lantern_draft_depth: one drafting depth of the EAGLE-2 drafter enqueued by ONE host call (include/lantern_hip.h).
// Launch sequencing over the library's own kernels + one small kernel that prepares the next depth's inputs.
//
// Reference: the loop body of topK_genrate -- models/drafters/cnets_lumina_mgpt.py:1271-1320 (tree_type "dynamic"), cnets_llamagen.py:783-821,
// cnets_anole.py:841-903 -- around Model.forward (cnets_lumina_mgpt.py:1052-1146) and the single decoder layer.
#include "common.h"

namespace lantern {
int launch_qk_norm_rope(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const void *q_weight, const void *q_bias,
                        const void *k_weight, const void *k_bias, int model_parallel, const void *cos_table, const void *sin_table, int table_rows,
                        const int64_t *position_ids, void *q_out, int q_rows, int q_row0, void *k_out, void *v_out, int kv_rows, int kv_row0, void *stream);
int launch_qk_rope_pairs(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const float *freqs, int table_rows,
                         const int64_t *position_ids, int positions_per_batch_row, void *q_out, int q_rows, int q_row0, void *k_out, void *v_out,
                         int kv_rows, int kv_row0, void *stream);
int launch_linear_rows_streamk_seg(const void *A, int a_seg_rows, long long a_seg_stride, const void *W, const void *bias, int M, int K, int n_rows, void *out,
                                   int epilogue, const void *aux, int aux_stride, int pair_rows, int packed, void *workspace, size_t workspace_bytes,
                                   hipStream_t st);
int launch_drafter_fc_streamk_tables(const int64_t *ids, int n_flat, const int32_t *in_gather, const int32_t *in_rep, const void *hidden, int src_T, int rows_per_b,
                                     const void *embed, const void *W, const void *bias, int M, int H, int vocab, float embed_scale, void *out, int packed,
                                     void *workspace, size_t workspace_bytes, hipStream_t st);
const char *last_error();
int launch_static_next_inputs(const int64_t *ss_token, int n_flat, const int32_t *gather, const int32_t *rep, const void *out_hidden, int B, int T, int H, int T_next,
                              void *hidden_next, int64_t *ids_next, hipStream_t st);

constexpr int DD_ROWS = 64;          // tree keys a drafting call can hold (one ancestor word per row)

// The next depth's inputs from this depth's outputs (cnets_llamagen.py:812-820): the best top_k of the T * top_k cumulative scores are
// topk_cs_index; parent row out_ids = cs / top_k, token = topk_index.flat[cs]; hidden rows of the parents for both batch rows; `parents` =
// cs + bias; ancestor word of new tree row j = the parent's word | its own bit.
__global__ __launch_bounds__(256) void draft_next_inputs_kernel(const int64_t *__restrict__ cs, const int64_t *__restrict__ topk_index, const uint16_t *__restrict__ out_hidden,
                                                               int B, int T, int H, int top_k, int t1, int64_t bias, uint16_t *__restrict__ hidden_next,
                                                               int64_t *__restrict__ ids_next, int64_t *__restrict__ parents_next, uint64_t *__restrict__ tree_bits) {
    const int j = blockIdx.x % top_k, b = blockIdx.x / top_k;          // one workgroup per (batch row, new token)
    const int64_t c = cs[j];
    int par = (int)(c / top_k);
    par = par < 0 ? 0 : (par >= T ? T - 1 : par);
    const uint4 *src = reinterpret_cast<const uint4 *>(out_hidden + ((size_t)b * T + par) * H);
    uint4 *dst = reinterpret_cast<uint4 *>(hidden_next + ((size_t)b * top_k + j) * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) {
        const int64_t tok = topk_index[(c >= 0 && c < (int64_t)T * top_k) ? c : 0];
        ids_next[(size_t)b * top_k + j] = tok;
        if (b == 0) {
            parents_next[j] = c + bias;
            if (t1 + j < DD_ROWS) tree_bits[t1 + j] = tree_bits[t1 - T + par] | (1ull << (t1 + j));
        }
    }
}
}  // namespace lantern

using namespace lantern;

namespace {
int fail(const char *stage, int rc) {
    char msg[400];
    snprintf(msg, sizeof msg, "%s", lantern::last_error());
    lantern::set_error("draft_depth: %s: %s", stage, msg);
    return rc;
}
}  // namespace

extern "C" int lantern_draft_depth(const lantern_draft_depth_args *ap) {
    LANTERN_CHECK_ARG(ap, "draft_depth: null arguments");
    const lantern_draft_depth_args &a = *ap;
    const int M = a.B * a.T, d = a.head_dim, nq = a.n_q_heads, nk = a.n_kv_heads, H = a.H;
    LANTERN_CHECK_ARG(a.B == 2 && a.T > 0 && M <= 32 && H > 0 && H % 64 == 0 && nq * d == H && (d == 64 || d == 128) && a.inter > 0 && a.inter % 64 == 0,
                      "draft_depth: B = 2 rows (cond, uncond), B * T <= 32, hidden %% 64 == 0, head_dim 64 or 128");
    LANTERN_CHECK_ARG(a.layer_kind == 0 || a.layer_kind == 1, "draft_depth: layer_kind %d", a.layer_kind);
    const bool is_static = a.n_draw > 0;          // a static tree: sample instead of expand, next inputs through the tree's tables
    LANTERN_CHECK_ARG(a.ids && a.hidden_in && a.embed && a.fc_w && a.qkv_w && a.o_w && a.ln2_w && a.gate_up_w && a.down_w && a.position_ids && a.k_slab && a.v_slab &&
                          a.tree_bits && a.head_w && (is_static || (a.topk_index && a.cu_scores && a.topk_cs_index && a.scores_out)),
                      "draft_depth: null weight / state / output buffer");
    if (is_static) {
        LANTERN_CHECK_ARG((a.draw_u || a.draw_idx) && a.probs_out && a.ss_token && a.ss_prob && a.T <= 16, "draft_depth: static tree: draws (uniforms or indices) and the probs_out / ss_token / ss_prob outputs, T <= 16");
        LANTERN_CHECK_ARG(a.T_next >= 0 && a.B * a.T_next <= 32 && (a.T_next == 0 || (a.next_gather && a.next_rep && a.hidden_next && a.ids_next)),
                          "draft_depth: static tree: next-depth tables / buffers (T_next = %d)", a.T_next);
    }
    LANTERN_CHECK_ARG(a.x && a.xn && a.qkv && a.q && a.attn && a.h1 && a.hn && a.act && a.out && a.head_ws && a.sk_ws && a.ta_ws, "draft_depth: null work buffer");
    LANTERN_CHECK_ARG(a.t1 >= a.T && a.t1 <= DD_ROWS && a.kv_row0 >= a.t1 - a.T && a.kv_rows >= a.kv_row0 + a.T,
                      "draft_depth: t1 = %d tree keys (this depth's %d included, at most %d), cache rows [%d, +%d) of %d", a.t1, a.T, DD_ROWS, a.kv_row0, a.T, a.kv_rows);
    if (a.layer_kind == 0) LANTERN_CHECK_ARG(a.qn_w && a.qn_b && a.kn_w && a.kn_b && a.cos_table && a.sin_table && a.model_parallel > 0, "draft_depth: Chameleon head-stage tables missing");
    else LANTERN_CHECK_ARG(a.freqs, "draft_depth: Llama head stage needs the freqs rows");
    if (a.hidden_next && !is_static) LANTERN_CHECK_ARG(a.ids_next && a.parents_next && a.top_k == a.T, "draft_depth: next-depth buffers (top_k == T)");
    hipStream_t st = (hipStream_t)a.stream;
    int rc;
    // ---- input stage
    if (a.in_rep) {
        LANTERN_CHECK_ARG(is_static && a.in_gather && a.in_src_T > 0 && a.in_n_flat > 0, "draft_depth: in_rep needs a static tree, in_gather, in_src_T and in_n_flat");
        rc = launch_drafter_fc_streamk_tables(a.ids, a.in_n_flat, a.in_gather, a.in_rep, a.hidden_in, a.in_src_T, a.T, a.embed, a.fc_w, a.fc_b, M, H, a.vocab,
                                              a.embed_scale, a.x, a.fc_packed, a.sk_ws, a.sk_ws_bytes, st);
    } else
        rc = lantern_drafter_fc_streamk(a.ids, a.hidden_in, a.embed, a.fc_w, a.fc_b, M, H, a.vocab, a.embed_scale, a.x, a.fc_packed, a.sk_ws, a.sk_ws_bytes, a.stream);
    if (rc) return fail("input stage", rc);
    // ---- decoder layer
    const void *xn = a.x;
    if (a.ln1_w) {
        rc = lantern_rmsnorm_rows(a.x, a.ln1_w, M, H, a.eps1, a.xn, a.stream);
        if (rc) return fail("input norm", rc);
        xn = a.xn;
    }
    const int nqkv = (nq + 2 * nk) * d;
    rc = launch_linear_rows_streamk_seg(xn, 0, 0, a.qkv_w, a.qkv_b, M, H, nqkv, a.qkv, 0, nullptr, 0, 0, a.layer_packed, a.sk_ws, a.sk_ws_bytes, st);
    if (rc) return fail("q/k/v projection", rc);
    const int q_row0 = a.t1 - a.T;
    if (a.layer_kind == 0)
        rc = launch_qk_norm_rope(a.qkv, a.B, a.T, nq, nk, d, a.qn_w, a.qn_b, a.kn_w, a.kn_b, a.model_parallel, a.cos_table, a.sin_table, a.table_rows,
                                 a.position_ids, a.q, DD_ROWS, q_row0, a.k_slab, a.v_slab, a.kv_rows, a.kv_row0, a.stream);
    else
        rc = launch_qk_rope_pairs(a.qkv, a.B, a.T, nq, nk, d, a.freqs, a.table_rows, a.position_ids, a.positions_per_batch_row, a.q, DD_ROWS, q_row0,
                                  a.k_slab, a.v_slab, a.kv_rows, a.kv_row0, a.stream);
    if (rc) return fail("head stage", rc);
    // q [B, nq, 64, d], attn [B, 64, H]: the t1 tree rows are the queries (rows in front of this depth's: earlier depths', their outputs unused)
    const int64_t kv_len = (int64_t)a.kv_row0 + a.T;
    rc = lantern_tree_attention(a.q, a.k_slab, a.v_slab, a.attn, a.B, nq, nk, a.t1, d, (int64_t)nq * DD_ROWS * d, (int64_t)d, (int64_t)DD_ROWS * d,
                                (int64_t)nk * a.kv_rows * d, (int64_t)a.kv_rows * d, (int64_t)DD_ROWS * H, (int64_t)H, nullptr, a.kv_start, kv_len,
                                a.tree_bits, 0, 1.0f / sqrtf((float)d), a.ta_ws, a.ta_ws_bytes, a.stream);
    if (rc) return fail("tree attention", rc);
    rc = launch_linear_rows_streamk_seg((const uint16_t *)a.attn + (size_t)q_row0 * H, a.T, (long long)DD_ROWS * H, a.o_w, a.o_b, M, H, H, a.h1, LANTERN_EPI_RESIDUAL,
                                        a.x, H, 0, a.layer_packed, a.sk_ws, a.sk_ws_bytes, st);
    if (rc) return fail("o_proj", rc);
    rc = lantern_rmsnorm_rows(a.h1, a.ln2_w, M, H, a.eps2, a.hn, a.stream);
    if (rc) return fail("post-attention norm", rc);
    rc = launch_linear_rows_streamk_seg(a.hn, 0, 0, a.gate_up_w, a.gate_up_b, M, H, a.inter, a.act, LANTERN_EPI_SILU_MUL, nullptr, 0, a.inter, a.layer_packed,
                                        a.sk_ws, a.sk_ws_bytes, st);
    if (rc) return fail("gate / up projection", rc);
    rc = launch_linear_rows_streamk_seg(a.act, 0, 0, a.down_w, a.down_b, M, a.inter, H, a.out, LANTERN_EPI_RESIDUAL, a.h1, H, 0, a.layer_packed, a.sk_ws,
                                        a.sk_ws_bytes, st);
    if (rc) return fail("down projection", rc);
    if (is_static) {
        // ---- head + sample (Model.sample on this depth's rows), then the next depth's tokens / hidden rows through the tree's tables
        rc = lantern_head_sample(a.out, a.head_w, a.head_b, a.T, H, a.row_lo, a.n_cols, a.vocab, a.cfg, a.model, a.head_pos, a.pos_base, a.w_latent, a.h_latent,
                                 a.newline_id, a.eos_id, a.top_k_filter, a.n_draw, a.draw_u, a.draw_idx, a.head_ws, a.probs_out, a.ss_token, a.ss_prob,
                                 a.head_packed, a.sk_ws, a.sk_ws_bytes, a.stream);
        if (rc) return fail("head sample", rc);
        if (a.T_next > 0) {
            rc = launch_static_next_inputs(a.ss_token, a.T * a.n_draw, a.next_gather, a.next_rep, a.out, a.B, a.T, H, a.T_next, a.hidden_next, a.ids_next, st);
            if (rc) return fail("next inputs", rc);
        }
        return LANTERN_OK;
    }
    // ---- head + expansion
    rc = lantern_head_expand_streamk(a.out, a.head_w, a.head_b, a.T, H, a.row_lo, a.n_cols, a.vocab, a.cfg, a.model, a.head_pos, a.pos_base, a.w_latent, a.h_latent,
                                     a.newline_id, a.eos_id, a.top_k_filter, a.scores_in, a.top_k, a.head_ws, a.topk_index, a.cu_scores, a.topk_cs_index,
                                     a.scores_out, a.head_packed, a.sk_ws, a.sk_ws_bytes, a.stream);
    if (rc) return fail("head expansion", rc);
    // ---- the next depth's inputs
    if (a.hidden_next) {
        hipLaunchKernelGGL(draft_next_inputs_kernel, dim3(a.B * a.top_k), dim3(256), 0, st, a.topk_cs_index, a.topk_index, (const uint16_t *)a.out, a.B, a.T, H, a.top_k,
                           a.t1, a.parent_bias_next, (uint16_t *)a.hidden_next, a.ids_next, a.parents_next, a.tree_bits);
        LANTERN_CHECK_LAUNCH("draft_depth");
    }
    return LANTERN_OK;
}
