/*
 * lantern_hip.h -- C-ABI of liblantern_hip.so: the MI355X (gfx950) implementation of
 * LANTERN's relaxed speculative-decoding verify/accept loop.
 *
 * The reference (jadohu/LANTERN) is pure Python/PyTorch and has no FFI: its seam is the
 * method surface of EaModel / EaLumina_mGPT.  Each entry point below replaces
 * one group of torch-op sequences + Python loops of that surface; the reference site it
 * replaces is cited as path:line relative to the upstream checkout.  The Python host
 * mirror (lantern_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Every function returns 0 on success, <0 = LANTERN_E_*; lantern_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - No allocation, no ownership transfer: every buffer is caller-allocated.  Pointers
 *     marked [dev] are device memory (torch storage), [host] host memory.
 *   - Device entry points are asynchronous on `stream` (a hipStream_t passed as void*),
 *     never synchronise, and keep no global state: they are re-entrant across streams and
 *     capturable into a hipGraph.
 *   - All device entry points are BATCHED over `B` independent sequences (one workgroup
 *     or more per sequence); the reference's B=1 call is the B=1 case.
 */
#ifndef LANTERN_HIP_H
#define LANTERN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LANTERN_VERSION 200

enum {
    LANTERN_OK = 0,
    LANTERN_E_INVALID = -1,    /* bad argument / unsupported shape */
    LANTERN_E_LAUNCH = -2,     /* HIP launch failure */
    LANTERN_E_UNSUPPORTED = -3 /* feature not built into this kernel set */
};

enum { LANTERN_MODE_DYNAMIC = 0, LANTERN_MODE_STATIC_LUMINA = 1, LANTERN_MODE_STATIC_LG = 2 };
enum { LANTERN_MODEL_PLAIN = 0, LANTERN_MODEL_LUMINA = 1, LANTERN_MODEL_ANOLE = 2 };
enum { LANTERN_F32 = 0, LANTERN_BF16 = 1 };

/* per-sequence status written to counters[5] by lantern_evaluate_posterior */
enum {
    LANTERN_ST_OK = 0,
    LANTERN_ST_TOKEN_OOB = 1,     /* candidate token outside [0,V) */
    LANTERN_ST_UNIFORMS = 2,      /* uniform stream exhausted */
    LANTERN_ST_TABLE_OOB = 3,     /* token - tok_offset outside the neighbour table */
    LANTERN_ST_SYNTAX_REJECT = 4, /* reference assert, ea_model_lumina_mgpt.py:694 */
    LANTERN_ST_NO_PREFIX = 5
};

int lantern_version(void);
const char *lantern_last_error(void);

/* Tuning values: kernel-instance / launch-shape choices a MEASUREMENT may override (tools/, a few tests).  The library reads no environment
 * variable; every product path runs the defaults.  Names (defaults): epw_tp (5), epw_tp4 (1), epw_tp_raw (256), epw_spec (2), epw_occ2 (-1),
 * o7_nt (0), prep_nt (0), kv_u (0), kv_ks (4), kv_variant (0), gemm_tiled_from (129), sk_groups (0), sk_whole_mb (40), sk_nt_min_mb (80),
 * ta_splits (0), ta_min_tiles (2), epw_tp_lg (1), epw_fused_helpers (1) -- meanings beside `enum Tuning` in lantern_amd/csrc/common.h.  Process-wide, atomic ints; set before the
 * launches they should affect.  The reference has no counterpart (it has no kernels to choose between). */
int lantern_tuning_set(const char *name, int value);
int lantern_tuning_get(const char *name, int *value);
const char *lantern_tuning_name(int index);   /* NULL past the last slot */
int lantern_tuning_reset(void);

/* ------------------------------------------------------------------------------------
 * O1  static target-tree buffers (HOST; once per tree shape).
 * Replaces generate_tree_buffers: models/ea_model_lumina_mgpt.py:140-277,
 * models/ea_model_llamagen.py:283-420, models/ea_model_anole.py:280-417.
 * choices: tree_choices flattened, choice_off[n_choices+1].
 * Outputs [host]: mask [N,N] f32 (1 = attend), tree_indices [N], pos_ids [N],
 * retrieve [P,D] (-1 pad, rows sorted), p_idx [P,D], b_off [P*D+1] + b_idx (CSR of the
 * earlier-sibling node ids per retrieve cell).
 */
int lantern_tree_static_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices,
                              int *N, int *P, int *D, int *b_total);
int lantern_tree_static_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k,
                              float *mask, int64_t *tree_indices, int64_t *pos_ids, int64_t *retrieve,
                              int32_t *p_idx, int32_t *b_off, int32_t *b_idx);

/* O2  drafter-side static buffers over non-leaf nodes (HOST).
 * Replaces models/drafters/utils_c.py:100-179 (copies cnets_lumina_mgpt.py:106-174).
 * level_counts[l] = non-leaf nodes of depth l+1.  masks_concat: per level [n_l, cum_l]. */
int lantern_tree_drafter_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices,
                               int *n_levels, int *level_counts);
int lantern_tree_drafter_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k,
                               float *masks_concat, int64_t *tree_indices_concat,
                               int32_t *repeat_nums_concat, int32_t *repeat_off);

/* ------------------------------------------------------------------------------------
 * O4  dynamic (EAGLE-2) tree finalise, one wavefront per sequence.
 * Replaces the tail of Model.topK_genrate: models/drafters/cnets_llamagen.py:831-912,
 * cnets_lumina_mgpt.py:1330-1393, cnets_anole.py:913-993.
 * [dev] scores [B,n_scores] f32, tokens [B,n_scores] i64, parents [B,n_parents] i64,
 *       sample_token [B] i64.  T = total_tokens (<= 63), N = T+1.
 * Out [dev]: draft_tokens [B,N] i64, mask [B,N,N] f32, pos_ids [B,N] i64,
 *            retrieve [B,N,N] i64 (row stride N, -1 pad; rows >= n_leaf are -1),
 *            n_leaf [B] i32, max_depth [B] i32.
 * Ties in the top-T selection break towards the lower flat index.
 */
int lantern_tree_dynamic_finalize(const float *scores, const int64_t *tokens, const int64_t *parents,
                                  const int64_t *sample_token, int B, int n_scores, int n_parents,
                                  int top_k, int total_tokens, int sort_rows, int64_t *draft_tokens,
                                  float *mask, int64_t *pos_ids, int64_t *retrieve, int32_t *n_leaf,
                                  int32_t *max_depth, void *stream);

/* O4 + O6-dynamic in ONE launch: lantern_tree_dynamic_finalize followed by lantern_gather_candidates_dynamic (below) for the same
 * sequences -- the workgroup that built a sequence's tree also writes its candidates [B,P,D], compact retrieve rows, row map and
 * absolute positions (same values as the two calls; lantern_verify_step uses this for dynamic groups). */
int lantern_tree_dynamic_candidates(const float *scores, const int64_t *tokens, const int64_t *parents,
                                    const int64_t *sample_token, int B, int n_scores, int n_parents, int top_k,
                                    int total_tokens, int sort_rows, int64_t *draft_tokens, float *mask, int64_t *pos_ids,
                                    int64_t *retrieve, int32_t *n_leaf, int32_t *max_depth, const int64_t *seq_len, int P, int D,
                                    int64_t *cand, int64_t *retrieve_pd, int32_t *row_index, int64_t *pos_abs, void *stream);

/* O3  one EAGLE-2 expansion depth: log_softmax rows -> top_k per row -> cumulative scores
 * -> top_k of the flattened n_rows*top_k.  Replaces cnets_llamagen.py:798-820,
 * cnets_lumina_mgpt.py:1303-1318.
 * [dev] logits [B,n_rows,V] f32 (already CFG'd/processed), scores_in [B,n_rows] f32 or NULL.
 * Out [dev]: topk_index [B,n_rows,top_k] i64, cu_scores [B,n_rows,top_k] f32,
 *            topk_cs_index [B,top_k] i64, scores_out [B,top_k] f32. */
int lantern_expand_dynamic(const float *logits, const float *scores_in, int B, int n_rows, int V, int top_k,
                           int64_t *topk_index, float *cu_scores, int64_t *topk_cs_index,
                           float *scores_out, void *stream);

/* ------------------------------------------------------------------------------------
 * O6  candidate assembly.  Replaces generate_candidates:
 * models/ea_model_lumina_mgpt.py:525-554, models/ea_model_llamagen.py:676-706.
 * [dev] ss_token [B,n_flat] i64, ss_prob [B,n_flat] f32 or NULL, sample_token [B] i64,
 *       tree_indices [N] i64, retrieve [P,D] i64 (shared by all sequences).
 * Out [dev]: tree_cand [B,N] i64, cand [B,P,D] i64, cart_prob [B,P,D] f32 (or NULL). */
int lantern_gather_candidates(const int64_t *ss_token, const float *ss_prob, const int64_t *sample_token,
                              const int64_t *tree_indices, const int64_t *retrieve, int B, int n_flat, int N,
                              int P, int D, int64_t *tree_cand, int64_t *cand, float *cart_prob,
                              void *stream);

/* O6, dynamic (EAGLE-2) trees: every sequence has its own tree (the outputs of lantern_tree_dynamic_finalize).
 * Replaces generate_candidates with the per-call buffers of topK_genrate: models/ea_model_llamagen.py:676-706,
 * models/drafters/cnets_llamagen.py:905-912 (`candidates = cat(draft_tokens, -1)[retrieve_indices]`), and the position
 * arithmetic of tree_decoding (ea_model_lumina_mgpt.py:559,601: tree_position_ids + len(input_ids) + 1).
 * [dev] draft_tokens [B,N] i64, retrieve [B,N,N] i64 (row stride N, -1 pad), pos_ids [B,N] i64 or NULL, seq_len [B] i64 or NULL.
 * Out [dev]: cand [B,P,D] i64 (-1 pad), retrieve_pd [B,P,D] i64 (-1 pad) or NULL, row_index [B,P,D] i32 (a -1 wraps to
 * node N-1, as torch's indexing does) or NULL, pos_abs [B,N] i64 = pos_ids + seq_len + 1 or NULL.  P, D <= N. */
int lantern_gather_candidates_dynamic(const int64_t *draft_tokens, const int64_t *retrieve, const int64_t *pos_ids,
                                      const int64_t *seq_len, int B, int N, int P, int D, int64_t *cand, int64_t *retrieve_pd,
                                      int32_t *row_index, int64_t *pos_abs, void *stream);

/* ------------------------------------------------------------------------------------
 * O7  tree-logit post-process: CFG combine + model mask + top-k threshold, one pass.
 * Replaces tree_decoding's epilogue: models/ea_model_lumina_mgpt.py:597-605 (with
 * MultiModalLogitsProcessor :45-86 and InterleavedTopKLogitsWarper :106-112),
 * models/ea_model_anole.py:930-931, models/ea_model_llamagen.py:930.
 * [dev] cond/uncond [rows,V] in `dtype`; bf16 input reproduces torch's per-op bf16
 * rounding of u + s*(c-u).  uncond == NULL: `cond` is already combined (mask / top-k only:
 * the standalone MultiModalLogitsProcessor / InterleavedTopKLogitsWarper calls of the drafter).  pos_ids i64 = the value the reference passes as
 * `position_ids=` (tree_position_ids + len(input_ids) + 1);
 * num_generated_image_tokens = pos - pos_base (Lumina only).  Two forms:
 *   seq_len == NULL: pos_ids is [rows].
 *   seq_len != NULL: batched sequences; pos_ids is [rows_per_seq] (shared tree position ids
 *   + 1) and seq_len [rows / rows_per_seq] i64 holds each sequence's len(input_ids), read on
 *   the device: pos = pos_ids[row % rows_per_seq] + seq_len[row / rows_per_seq].
 * Out [dev]: out [rows,V] f32.  V % 4 == 0, V <= 65536.
 */
int lantern_cfg_mask_topk(const void *cond, const void *uncond, int dtype, int rows, int V, float cfg,
                          int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent,
                          int img_lo, int img_hi, int newline_id, int eos_id, int top_k,
                          const int64_t *seq_len, int rows_per_seq, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * O8  relaxed tree rejection sampling (the north-star kernel).
 * Replaces the sampling branch of evaluate_posterior:
 *   models/ea_model_lumina_mgpt.py:610-726 (eagle_version 1 and 2),
 *   models/ea_model_llamagen.py:709-787 / models/ea_model_anole.py:709-788 (dynamic),
 *   models/ea_model_llamagen.py:597-669 / models/ea_model_anole.py:597-669 (static, _v1).
 */
typedef struct lantern_ep_params {
    int32_t B;               /* sequences */
    int32_t P, D;            /* array extents of cand/row_index (strides) */
    int32_t V;
    int32_t rows_per_seq;    /* logits rows per sequence */
    int32_t mode;            /* LANTERN_MODE_* */
    int32_t syntax_shortcut; /* Lumina: syntax token -> px=1, non-image -> px=0 (:654-659) */
    int32_t tok_offset;      /* image-token offset into the neighbour table */
    int32_t img_lo, img_hi;
    int32_t n_syntax;
    int32_t syntax[8];
    int32_t lantern, k;
    int32_t table_rows, table_cols;
    int32_t top_k;           /* per-level HF processors (LlamaGen/Anole): <=0 off */
    float temperature;       /* <=1e-5 or 1 -> off */
    float top_p;             /* TopPLogitsWarper in [1e-8, 1) (between temperature and top_k): the dense kernel and LANTERN_ROWS_RAW_BF16; else off */
    double delta;            /* <=1: delta mode; >1: lambda mode, tau = (delta-1)*px */
    int32_t n_uniforms;      /* uniforms per sequence (row stride) */
    int32_t R;               /* orig_prob rows per sequence (static) */
    int32_t N;               /* tree_cand entries per sequence (static) */
    int32_t row_index_per_seq; /* 1: row_index is [B,P,D]; 0: [P,D] shared */
} lantern_ep_params;

typedef struct lantern_ep_buffers {
    const float *logits;      /* [dev] [B,rows_per_seq,V] f32 (processed node logits, or the
                                 materialised [P*D,V] of the reference with row_index = arange) */
    const int32_t *row_index; /* [dev] (path,depth) -> logits row */
    const int64_t *cand;      /* [dev] [B,P,D], -1 pad */
    const int32_t *n_paths;   /* [dev] [B] valid rows of cand, or NULL (= P) */
    const int32_t *n_depth;   /* [dev] [B] valid columns, or NULL (= D) */
    /* static trees only (NULL in dynamic mode) */
    const float *cart_prob;   /* [dev] [B,P,D] */
    const float *orig_prob;   /* [dev] [B,R,V] drafter distributions, levels concatenated */
    const int32_t *op_off;    /* [dev] [D-1] first row of drafter level d */
    const int32_t *p_idx;     /* [dev] [P,D] */
    const int32_t *b_off;     /* [dev] [P*D+1] */
    const int32_t *b_idx;     /* [dev] */
    const int64_t *tree_cand; /* [dev] [B,N] */
    const uint16_t *nn_table; /* [dev] [table_rows,table_cols] uint16 (NULL if !lantern) */
    const double *uniforms;   /* [dev] [B,n_uniforms]: the Python random.random() stream */
    int32_t *cursor;          /* [dev] [B] in/out read position in uniforms, or NULL (0) */
    float *workspace;         /* [dev] lantern_evaluate_posterior_workspace() bytes */
    /* outputs */
    int32_t *best;            /* [dev] [B] */
    int32_t *accept_len;      /* [dev] [B] */
    float *sample_p;          /* [dev] [B,V] */
    int32_t *counters;        /* [dev] [B,6]: levels, tried, rejected, uniforms used,
                                 final-from-residual, status (LANTERN_ST_*) */
} lantern_ep_buffers;

size_t lantern_evaluate_posterior_workspace(const lantern_ep_params *prm);
int lantern_evaluate_posterior(const lantern_ep_params *prm, const lantern_ep_buffers *buf, void *stream);

/* ------------------------------------------------------------------------------------
 * Windowed (v2) form of O7 + O8 for models whose processed rows are -inf outside one id range
 * (Lumina/Anole: image ids [4,8196); LlamaGen: the whole vocabulary).  Same results as the dense
 * entry points; the rows cross HBM as `win_len` floats instead of V, the residual distribution
 * lives in LDS, and the bonus token is drawn in the epilogue so the dense sample_p never has to
 * exist.  A processed row is either a window row (finite values only inside the window) or a
 * one-hot row (`row_hot[row]` = its token id, e.g. the forced newline / end-of-image rows of
 * MultiModalLogitsProcessor); row_hot = -1 for window rows.
 */

/* What a window row holds.  LOGITS: processed logits (-inf = removed); evaluate_posterior_window applies
 * prm->temperature / prm->top_k and the softmax per VISITED row, as the reference does
 * (ea_model_llamagen.py:725,785).  PROBS: cfg_mask_topk_window already applied temperature -> top-k -> softmax to
 * EVERY row (one workgroup per row, all in parallel), so the serial per-sequence chain of O8 only copies the row
 * into LDS; same arithmetic, same bits (the row's distribution does not depend on when it is computed). */
#define LANTERN_ROWS_LOGITS 0
#define LANTERN_ROWS_PROBS 1
/* RAW_BF16 (lantern_evaluate_posterior_window only): buf->logits is the target model's CONDITIONAL head output [B, rows_per_seq, V]
 * in bf16 and win->raw_uncond the unconditional one; the whole post-process of tree_decoding (ea_model_lumina_mgpt.py:597-607: CFG
 * combination, MultiModalLogitsProcessor's row classes from the positions, InterleavedTopKLogitsWarper, softmax) runs inside the
 * kernel, for the rows the walk visits only (~3.7 of a 26-node tree's rows per step) -- lantern_cfg_mask_topk_window is not
 * launched at all.  Same arithmetic as that kernel, so the same bits.  Lumina windows of exactly 8192 ids on the packed table. */
#define LANTERN_ROWS_RAW_BF16 2

/* O7 windowed: same arguments as lantern_cfg_mask_topk, output [rows, win_len] f32 + row_hot [rows].
 * LANTERN_MODEL_PLAIN requires win_lo = 0, win_len = V.  `temperature` (> 1e-5; 1.0 = none) divides the row
 * first, then `top_p` in [1e-8, 1) removes the low tail whose cumulative probability is <= 1 - top_p
 * (TopPLogitsWarper; 1.0 = off), then top_k -- the HF order Temperature -> TopP -> TopK of
 * drafters/utils.py:36-52; out_kind selects what is stored (LANTERN_ROWS_*). */
int lantern_cfg_mask_topk_window(const void *cond, const void *uncond, int dtype, int rows, int V, float cfg,
                                 int model, const int64_t *pos_ids, int64_t pos_base, int w_latent,
                                 int h_latent, int img_lo, int img_hi, int newline_id, int eos_id, int top_k,
                                 const int64_t *seq_len, int rows_per_seq, int win_lo, int win_len,
                                 float *out_win, int32_t *row_hot, int out_kind, float temperature, float top_p,
                                 void *stream);

typedef struct lantern_ep_window {
    int32_t win_lo, win_len;      /* window = token ids [win_lo, win_lo+win_len); win_len % 4 == 0 */
    const int32_t *row_hot;       /* [dev] [B*rows_per_seq] or NULL (all window rows) */
    int32_t orig_prob_stride;     /* elements between drafter rows of buf->orig_prob (V for the dense
                                     [B,R,V] layout, win_len for a windowed pool); row r of sequence b
                                     starts at orig_prob + (b*R + r)*stride + orig_prob_offset */
    int32_t orig_prob_offset;     /* win_lo for the dense layout, 0 for a windowed pool.  Precondition:
                                     drafter rows are zero outside the window (they are masked like the
                                     target rows: cnets_lumina_mgpt.py:1220-1224,1294-1298) */
    float *sample_win;            /* [dev] [B,win_len] out or NULL: sample_p restricted to the window */
    int32_t *out_tok;             /* [dev] [B] out: token id carrying mass outside the window, or -1 */
    float *out_mass;              /* [dev] [B] out: its probability */
    const double *u_bonus;        /* [dev] [B] or NULL: uniform for the bonus-token draw */
    int64_t *token;               /* [dev] [B] out (with u_bonus): inverse-CDF bonus token */
    int32_t rows_kind;            /* LANTERN_ROWS_LOGITS | LANTERN_ROWS_PROBS | LANTERN_ROWS_RAW_BF16: what buf->logits rows hold */
    int32_t raw_pos_per_seq;      /* LANTERN_ROWS_RAW_BF16: 0 = one tree for all sequences (raw_pos_ids [rows_per_seq] relative, + raw_seq_len[b]);
                                     1 = per-sequence trees (EAGLE-2): raw_pos_ids [B, rows_per_seq] ABSOLUTE positions, raw_seq_len unused */
    /* LANTERN_ROWS_RAW_BF16 only (ignored otherwise) */
    const void *raw_uncond;       /* [dev] [B, rows_per_seq, V] bf16 */
    const int64_t *raw_pos_ids;   /* [dev] i64: tree_position_ids + 1 (shared tree), or absolute positions per sequence (raw_pos_per_seq) */
    const int64_t *raw_seq_len;   /* [dev] [B] i64: len(input_ids) of every sequence (NULL with raw_pos_per_seq) */
    int64_t raw_pos_base;         /* num_generated_image_tokens = pos - pos_base */
    float raw_cfg;                /* guidance scale */
    int32_t raw_top_k;            /* InterleavedTopKLogitsWarper image_top_k (0 = off) */
    int32_t raw_w_latent, raw_h_latent, raw_newline_id, raw_eos_id;
    /* optional: rows some earlier launch already post-processed (lantern_prepare_step: the nodes the walk most likely visits) */
    const float *raw_probs;       /* [dev] [B, rows_per_seq, win_len] f32 probabilities of the listed rows, or NULL */
    const uint8_t *raw_pre;       /* [dev] [rows_per_seq]: 0 = post-process the node's row on demand; 1 + d = it is in raw_probs, prepared for a node
                                     at depth d (static trees: the depth is not checked, write 1; raw_pos_per_seq: used only when the
                                     sequence's node really sits at depth d, i.e. raw_pos_ids[node] - raw_pos_ids[0] == d) */
    /* optional: the verdict of every sequence ALSO written where the host can read it without a copy or a stream synchronisation -- pinned
     * (host-coherent) memory, 16 int32 per sequence: [0] best, [1] accept_len, [2..7] counters, [8..9] the bonus token (int64), [10] = 1 written LAST
     * behind a system-scope fence.  The caller zeroes word 10 before the launch and polls it: the record is visible as soon as the walk ends,
     * while the commit kernel behind it still runs (the reference reads these values with ~6 `.item()` syncs per tried candidate). */
    int32_t *verdict_host;
    /* optional: commit turn-taking between stream groups (lantern_step_group.turn, below): the chain kernel's workgroups do not END before
     * turn[0] >= turn_wait, so that the commit launched behind them on the stream starts when it is this group's turn.  NULL: off. */
    const int64_t *turn;
    int64_t turn_wait;
} lantern_ep_window;

/* O8 windowed.  buf->logits is [B, rows_per_seq, win_len]; buf->sample_p may be NULL (if given, the dense
 * [B,V] distribution is also written); buf->workspace unused.  counters[5] == LANTERN_ST_NEEDS_DENSE
 * marks the (measure-zero) residual `gtp.sum()==0 -> ones` case, which only the dense kernel represents. */
#define LANTERN_ST_NEEDS_DENSE 6
/* counters[5] == LANTERN_ST_TREE_LIMIT: a static tree beyond what the windowed kernel stages in LDS (a node with more than 16
 * earlier siblings, or more than 1024 entries in b_idx) -- reported, never silently truncated; the dense kernel has no such limit. */
#define LANTERN_ST_TREE_LIMIT 7
int lantern_evaluate_posterior_window(const lantern_ep_params *prm, const lantern_ep_buffers *buf,
                                      const lantern_ep_window *win, void *stream);

/* ------------------------------------------------------------------------------------
 * O8 node-parallel (v3): the same relaxed rejection sampling, one workgroup per INTERNAL TREE NODE instead of one
 * serial chain per sequence.  While sibling tokens are distinct (sampling without replacement / top-k -- every tree the
 * drafters build) the candidates the reference tries at a level (ea_model_lumina_mgpt.py:628-713) are the children of the
 * accepted node, in the order of their first path, and everything a child's test reads -- the parent's row, the drafter
 * row, the earlier siblings, its cart_candidates_prob cell, the position of its uniform in the random.random() stream
 * (one draw per tried candidate from the root down) -- depends on the node alone.  So every node's accept / reject chain
 * (k-neighbour cumulative mass, residual distribution, bonus-token draw when all children are rejected) runs concurrently
 * in its own workgroup, and a short second kernel walks root -> accepted child -> ... over the per-node results to produce
 * exactly the outputs of lantern_evaluate_posterior_window (best, accept_len, counters, cursor, bonus token, optional
 * sample_p).  Nodes the walk never reaches were computed in vain: B*n_internal workgroups fill the GPU where B chains
 * occupy B compute units, and the launch lasts as long as the widest node's chain, not as long as the unluckiest
 * sequence's whole walk.  A node whose children carry duplicate (or -1) tokens cannot be decomposed this way: a sequence
 * whose walk meets one reports LANTERN_ST_NEEDS_CHAIN and is re-run by the caller on lantern_evaluate_posterior_window.
 *
 * lantern_tree_node_tables (HOST, once per tree shape): retrieve [P,D] i64 (-1 pad) and, for static trees, p_idx [P,D],
 * b_off [P*D+1], b_idx, op_off [D-1] (NULL for dynamic trees) -> packed int32 tables (lantern_tree_node_tables_size ints) that
 * the caller uploads; the header words out[0..3] = {N, n_internal, n_children, max_children}, out[6] = 1 when every child's
 * earlier-sibling list (b_idx) is exactly the children tried before it -- true for every tree generate_tree_buffers builds,
 * and required by the kernel (LANTERN_E_UNSUPPORTED otherwise: use lantern_evaluate_posterior_window).
 * Restrictions of this build: window rows are probabilities (LANTERN_ROWS_PROBS), k + 1 <= 1024, win_len <= 16384, a
 * node has <= 32 children (else LANTERN_ST_TREE_LIMIT), N <= 128, one tree for all B sequences.
 */
#define LANTERN_ST_NEEDS_CHAIN 8
typedef struct lantern_ep_nodes {
    const int32_t *tables;   /* [dev] packed tables from lantern_tree_node_tables */
    const int32_t *tables_host; /* [host] the same tables (the launch copies the per-node header into the kernel arguments) */
    int32_t n_nodes, n_internal, n_children, max_children; /* = tables[0..3] (the host sizes the launch with them) */
    int32_t prefix_siblings; /* = tables[6] */
    int32_t leaf_workgroups; /* 1: leaves get workgroups that pre-draw their bonus token (small batches: nothing after the node kernel but
                                a table walk); 0: the walk kernel draws it for the one leaf a walk ends on; -1: chosen by batch size */
    void *workspace;         /* [dev] lantern_evaluate_posterior_nodes_workspace() bytes, 16-byte aligned */
    size_t workspace_bytes;
    int32_t reserved[2];
} lantern_ep_nodes;

int lantern_tree_node_tables_size(int N, int P, int D);
int lantern_tree_node_tables(const int64_t *retrieve, const int32_t *p_idx, const int32_t *b_off, const int32_t *b_idx,
                             const int32_t *op_off, int N, int P, int D, int32_t *out, int out_ints);
/* want_dist != 0 when win->sample_win or buf->sample_p is requested (every node then parks its final distribution). */
size_t lantern_evaluate_posterior_nodes_workspace(const lantern_ep_params *prm, const lantern_ep_window *win, int n_internal,
                                                  int want_dist);
int lantern_evaluate_posterior_nodes(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win,
                                     const lantern_ep_nodes *nodes, void *stream);

/* a9 inside the one-call step: the greedy / TVD accept of a step whose caller decodes without a processor list (temperature <= 1e-5:
 * models/ea_model_llamagen.py:789-905, ea_model_anole.py:790-905; the loop that reaches it: ea_model_llamagen.py:1109-1169).  With
 * lantern_step_group.greedy set, lantern_verify_step runs: [candidates] -> lantern_cfg_mask_topk (the CFG combination and the model mask of
 * every tree row in the logits' dtype, no top-k: ea_model_llamagen.py:930 / ea_model_anole.py:930-931) into `logits` -> lantern_evaluate_posterior_greedy
 * (best / accept_len -> ep_buf.best / ep_buf.accept_len, the accepted row -> out_row) -> the bonus token = argmax(out_row), first maximum
 * (`token`) -> the KV / hidden / token commit.  No uniforms, no counters (ep_buf.counters may be NULL), ep / ep_win are not read.  No greedy kernel writes the
 * pinned verdict record: a greedy group with ep_win.verdict_host set is refused (LANTERN_E_INVALID), its caller reads best / accept_len / token. */
typedef struct lantern_step_greedy {
    float *logits;                /* [dev] [B, N, V] f32, written by the O7 stage */
    const int32_t *row_index;     /* [dev] [P, D] (or [B, P, D] with row_index_per_seq): tree row of every (path, depth) */
    int row_index_per_seq;
    int lantern, k;               /* LANTERN relaxation: sum the k nearest codes' probability (0: exact greedy match) */
    double delta;
    int tok_offset;               /* id of code 0 */
    const uint16_t *nn_table;     /* [dev] [table_rows, table_cols] (NULL when !lantern) */
    int table_rows, table_cols;
    int win_lo, win_len;          /* ids read: LlamaGen (0, V); Anole the image range */
    int32_t *ok_scratch;          /* [dev] B * P * (D - 1) */
    float *out_row;               /* [dev] [B, V] */
    int64_t *token;               /* [dev] [B] */
} lantern_step_greedy;

/* The DENSE kernel set inside the one-call step (round 6): what a caller falls back to when the windowed kernels report a state only the dense kernel
 * represents (counters[5] = LANTERN_ST_NEEDS_DENSE: the residual vanished and the reference samples uniformly over all V ids, ea_model_lumina_mgpt.py:712-714;
 * uniforms or a tree beyond the windowed kernels' staged sizes), or whose tree never fits them (P > 64).  With lantern_step_group.dense set,
 * lantern_verify_step runs: [candidates] -> lantern_cfg_mask_topk over all N rows at the full vocabulary (the group's O7 fields; skipped when cond is
 * NULL: `logits` then already holds the processed rows) into `logits` -> lantern_evaluate_posterior on those rows (ep, ep_buf with logits / sample_p
 * replaced by the two buffers below; ep_win is not read) -> the bonus token = inverse CDF of sample_p at u_bonus (lantern_accept_gather's draw,
 * ea_model_lumina_mgpt.py:779-790) -> the KV / hidden / token commit.  Not combinable with nodes, greedy, node_list, prepare_next, turn or
 * ep_win.verdict_host (refused). */
typedef struct lantern_step_dense {
    float *logits;                /* [dev] [B, N, V] f32: O7's output, O8's input */
    float *sample_p;              /* [dev] [B, V] out: the distribution the bonus token is drawn from */
    const double *u_bonus;        /* [dev] [B] the bonus draw's uniform; NULL (with token NULL): no bonus token */
    int64_t *token;               /* [dev] [B] out */
} lantern_step_dense;

/* ------------------------------------------------------------------------------------
 * One verify step of G independent groups of sequences in ONE call: for every group, on the group's own stream,
 *   O6 lantern_gather_candidates -> O7 lantern_cfg_mask_topk_window -> O8 lantern_evaluate_posterior_nodes (nodes != NULL) or
 *   lantern_evaluate_posterior_window -> O9 + O10 lantern_update_inference_inputs
 * i.e. the body of the reference's decode loop between the target forward and the next drafter call
 * (models/ea_model_lumina_mgpt.py:936-998: generate_candidates, tree_decoding's post-process, evaluate_posterior,
 * update_inference_inputs), with nothing returning to the host.  Groups are independent sequences, so their streams are not
 * ordered against each other: one group's latency-bound evaluate_posterior overlaps the other groups' bandwidth-bound kernels.
 * Every field has the meaning of the same-named argument of the entry point it is passed to; the host loop only patches the
 * step-dependent pointers (where this step's outputs go) between calls.  slab_ptrs == NULL skips O9 + O10; out_win == NULL skips O7
 * (ep_win.rows_kind == LANTERN_ROWS_RAW_BF16: evaluate_posterior reads the raw cond / uncond logits itself).
 * Pointers inside a group (nodes, dyn, node_list) are read during the call only.  An error names the group and the stage
 * (lantern_last_error(): "verify_step: group 2, evaluate_posterior: ...").
 *
 * Dynamic (EAGLE-2) trees: dyn != NULL replaces the O6 stage by O4 + O6-dynamic on the group's stream, one launch
 * (lantern_tree_dynamic_candidates) --
 *   tree_dynamic_finalize(dyn->scores, dyn->tokens, dyn->parents, sample_token, ...) -> draft_tokens / mask / pos_ids / retrieve
 *   gather_candidates_dynamic(draft_tokens, retrieve, pos_ids, dyn->seq_len, ...) -> cand / retrieve_pd / row_index / pos_abs
 * (models/drafters/cnets_lumina_mgpt.py:1337-1420 and ea_model_lumina_mgpt.py:559,601 for every sequence of the group) -- and O9 + O10
 * then read the per-sequence paths dyn->retrieve_pd.  The caller points pos_ids at dyn->pos_abs (seq_len = NULL) and
 * ep_buf.row_index at dyn->row_index.
 */
typedef struct lantern_step_dynamic {
    const float *scores; const int64_t *tokens; const int64_t *parents;   /* [B,n_scores] f32, [B,n_scores] i64, [B,n_parents] i64 */
    int32_t n_scores, n_parents, top_k, total_tokens, sort_rows, reserved;
    int64_t *draft_tokens; float *mask; int64_t *pos_ids; int64_t *retrieve; int32_t *n_leaf; int32_t *max_depth;   /* O4 outputs */
    const int64_t *seq_len;                                               /* [B] tokens in front of the tree */
    int64_t *retrieve_pd; int32_t *row_index; int64_t *pos_abs;           /* O6-dynamic outputs ([B,P,D], [B,P,D], [B,N]) */
} lantern_step_dynamic;
typedef struct lantern_step_group {
    void *stream;
    /* O6 */
    const int64_t *ss_token; const float *ss_prob; const int64_t *sample_token; const int64_t *tree_indices; const int64_t *retrieve;
    int32_t B, n_flat, N, P, D, reserved0;
    int64_t *tree_cand; int64_t *cand; float *cart_prob;
    /* O7 (windowed) */
    const void *cond; const void *uncond; int32_t dtype, V; float cfg; int32_t model; const int64_t *pos_ids; int64_t pos_base;
    int32_t w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k, win_lo, win_len, out_kind;
    const int64_t *seq_len; float *out_win; int32_t *row_hot; float temperature, top_p;
    /* O8 */
    lantern_ep_params ep; lantern_ep_buffers ep_buf; lantern_ep_window ep_win;
    const lantern_ep_nodes *nodes;        /* NULL: the chain kernel (lantern_evaluate_posterior_window) */
    /* O9 + O10 */
    void *const *slab_ptrs; const int32_t *slab_seq; const int64_t *slab_prev; int64_t *new_len;
    int32_t n_slabs, elem_bytes; int64_t outer, S_max, d;
    const void *hidden; void *out_hidden; int64_t *accepted_tokens; int32_t hid_elem_bytes, hid_groups, H, reserved1;
    /* with ep_win.rows_kind == LANTERN_ROWS_RAW_BF16: the nodes whose rows are post-processed up front, together with the candidate
     * assembly, in ONE launch (lantern_prepare_step) -- the root and the most likely children; NULL / 0: none (all rows on demand) */
    const int32_t *node_list; int32_t n_list, flags;       /* dynamic groups: [2 * n_list] = the nodes, then the depth each is assumed to sit at; flags: LANTERN_STEP_* */
    /* O10 extras (all optional, NULL = off) -- what the reference's loop body does with torch ops around update_inference_inputs:
     *  hidden_uncond: the unconditional pass's hidden rows [B, N, H] as their own pointer (then `hidden` is the conditional pass's [B, N, H] and
     *    hid_groups must be 2): the two passes' outputs are not stacked into one [B, 2, N, H] tensor first (ea_model_lumina_mgpt.py:748-750);
     *  ids_buf [B, ids_stride] i64 + ids_len [B]: `input_ids = cat(input_ids, accepted tokens)` (:763-767) as an in-place append -- the accept_len + 1
     *    tokens of the chosen path go to ids_buf[b][ids_len[b] ...], and the bonus token (ep_win.token) behind them, where the drafter's
     *    `cat(input_ids, token)` (:781-785) expects it.  A sequence whose walk reported a status appends nothing. */
    const void *hidden_uncond; int64_t *ids_buf; int64_t ids_stride; const int64_t *ids_len;
    /* The NEXT step's preparation inside THIS step's commit launch (static trees with node_list, i.e. lantern_prepare_step's form): the workgroups of
     * step s + 1's candidate assembly + likely rows ride in the launch that moves step s's KV rows; they take step s's verdict from the walk (its bonus
     * token = the next root, its accepted length = the next positions).
     * PRECONDITION: step s + 1's cond / uncond / ss_token / ss_prob must be FINAL when step s's commit launches.  That only holds for a caller whose rows
     * exist ahead of time (the synthetic harness's pools: lantern_amd/harness.py `merge_prepare`, a bench extra).  In a real decode loop those tensors are
     * produced by the drafter and the target forward that run AFTER commit(s) (models/ea_model_lumina_mgpt.py:984-998 then :948-955 of the next
     * iteration), so the mirrors never set this field and bench.py's headline does not use it.
     * prepare_next: the group the next lantern_verify_step call will pass for these sequences (its seq_len must be THIS step's lengths -- the kernel adds
     * the accepted tokens itself; its cand buffer must not be the one this step's commit reads); that call then carries LANTERN_STEP_PREPARED.  A call
     * that failed, or a sequence whose walk reported a non-zero status in step s (the caller retries and commits it itself), invalidates the preparation:
     * clear LANTERN_STEP_PREPARED and let step s + 1 prepare itself.  NULL: off. */
    const struct lantern_step_group *prepare_next;
    /* Commit turn-taking between the stream groups of one caller (optional, NULL = off; round 6).  Groups that run the same step loop side by side on
     * their own streams fall into lock-step: all of them move their KV rows at the same time (one bandwidth-bound phase of ~30 us during which no
     * latency-bound walk runs) instead of one group's commit overlapping the others' walks.  With `turn` set the groups take turns without any
     * cross-stream event: `turn` [dev] int64 [LANTERN_TURN_WORDS(turn_groups)], zeroed once by the caller and never reset -- word 0 counts the commits
     * COMPLETED by all groups; behind it, one 128-byte line per counter, each group's "workgroups of my running commit launch that have finished"
     * counters in two levels (32 first-level lines per group, then one: thousands of workgroups adding to one word serialise at the memory side); the
     * workgroup that fills a counter puts it back to zero, the launch's last workgroup increments word 0.  This step's evaluate_posterior (chain kernel)
     * holds its last instructions until turn[0] >= turn_wait -- bounded: after ~40 ms it proceeds anyway, turn-taking is scheduling, never correctness
     * -- and the commit launched behind it releases the turn when it is done.  turn_wait: the commits that must have completed before this one starts
     * (ticket - (window - 1) for `window` commits in flight; the harness issues tickets step * n_groups + group).
     * Independent sequences: any order is correct; the reference runs one sequence per process and has no counterpart. */
    int64_t *turn;
    int32_t turn_group, turn_groups;        /* this group's index, and how many groups share `turn` */
    int64_t turn_wait;
    const lantern_step_dynamic *dyn;      /* NULL: a static tree (ss_token / tree_indices / retrieve), or -- flags & LANTERN_STEP_CANDIDATES_READY, ss_token
                                             NULL -- candidates the caller assembled itself: `cand` [B,P,D] and `retrieve` [P,D] are taken as they are
                                             (a tree that came with its token list, ea_model_llamagen.py:1125-1131; or lantern_gather_candidates called
                                             before the target forward).  A static group with neither is an error, not a silent skip. */
    const lantern_step_greedy *greedy;    /* NULL: relaxed rejection sampling (evaluate_posterior); else the greedy / TVD accept above */
    const lantern_step_dense *dense;      /* NULL: the windowed kernels; else the dense kernel set above */
    /* flags & LANTERN_STEP_FUSED_PREPARE (round 6): the prepare stage rides INSIDE the chain launch -- two launches per step (walk, commit) instead of three.
     * Static Lumina or Anole trees on raw rows with a node list (lantern_prepare_step's form; node_list[0] is the root), chain kernel, at most 256 sequences.  The
     * launch's first B * (n_list - 1) workgroups post-process the listed rows but the root's into out_win (= ep_win.raw_probs) and publish each by storing
     * row_epoch into row_ready[b * N + node]; the sequence workgroups assemble their own candidates (written to cand / tree_cand / cart_prob for the commit
     * launch), post-process the root's row themselves, and read a listed row only once its word carries this step's epoch -- otherwise they post-process it
     * themselves: same bits, so the race decides timing, never the result.  row_ready [dev] int32 [B, N], zeroed once; row_epoch: a value no earlier step of
     * these buffers used (the harness: step + 1).  Any other configuration is refused (LANTERN_E_UNSUPPORTED), not run in three launches. */
    int32_t *row_ready;
    int32_t row_epoch, reserved2;
} lantern_step_group;
#define LANTERN_TURN_WORDS(n_groups) (16 * (1 + 33 * (n_groups)))   /* int64 words of lantern_step_group.turn */
#define LANTERN_STEP_CANDIDATES_READY 1   /* lantern_step_group.flags: skip the O6 stage, `cand` / `retrieve` (/ `cart_prob`, `tree_cand`) are final */
#define LANTERN_STEP_FUSED_PREPARE 4       /* the prepare stage inside the chain launch (row_ready / row_epoch above) */
#define LANTERN_STEP_PREPARED 2           /* the previous call's commit launch already ran this group's lantern_prepare_step (prepare_next); only valid on a
                                             static-tree group with a node_list, refused otherwise */
int lantern_verify_step(const lantern_step_group *groups, int n_groups);
/* O6 + O7 restricted to s->node_list in one launch (bf16 Lumina rows, 8192-id window): candidates -> s->tree_cand / cand / cart_prob,
 * probabilities of the listed rows -> s->out_win, their classes -> s->row_hot.  Called by lantern_verify_step when node_list is set.
 * s->dyn != NULL: the tree workgroups run O4 + O6-dynamic (lantern_tree_dynamic_candidates) beside the row workgroups; the listed
 * rows are prepared for the depth given in node_list[n_list + i] (the root; node 1 = the drafter's best first token at depth 1). */
int lantern_prepare_step(const lantern_step_group *s);
/* window -> dense [B,V] (API compatibility with callers that want the reference's sample_p[V]). */
int lantern_window_to_dense(const float *win, const int32_t *out_tok, const float *out_mass, int B, int V,
                            int win_lo, int win_len, float *dense, void *stream);

/* a9  greedy / TVD branch (temperature <= 1e-5): models/ea_model_llamagen.py:789-905,
 * models/ea_model_anole.py:790-905.  logits [dev] [B,rows_per_seq,V] f32 (dense rows); only ids
 * [win_lo, win_lo+win_len) are read -- LlamaGen: (0, V); Anole: the image range (4, 8192), the rest is
 * finfo.min after CFG (ea_model_anole.py:931).  win_len <= 16384, P <= 64.
 * ok_scratch [dev] B*P*(D-1) int32.  Out: best/accept_len [B] i32, out_row [B,V] = logits[best, accept_len]. */
int lantern_evaluate_posterior_greedy(const float *logits, const int32_t *row_index, const int64_t *cand,
                                      int B, int P, int D, int V, int rows_per_seq, int row_index_per_seq,
                                      int lantern, int k, double delta, int tok_offset,
                                      const uint16_t *nn_table, int table_rows, int table_cols,
                                      int win_lo, int win_len, int32_t *ok_scratch,
                                      int32_t *best, int32_t *accept_len, float *out_row, void *stream);

/* ------------------------------------------------------------------------------------
 * O9  KV-cache index gather, all slabs / layers / heads in one launch.
 * Replaces the slab loop of update_inference_inputs: models/ea_model_lumina_mgpt.py:741-746,
 * :763-767; models/ea_model_llamagen.py:961-970; KVCache.copy models/drafters/kv_cache.py:38-50:
 *   slab[..., prev:prev+a+1, :] <- slab[..., retrieve[best,:a+1] + prev, :]
 * slab_ptrs [dev] [n_slabs] device pointers; each slab is [outer, S_max, d] elements of
 * elem_bytes (row bytes d*elem_bytes % 16 == 0).  slab_seq [dev] [n_slabs]: sequence whose
 * best/accept_len the slab follows; slab_prev [dev] [n_slabs] i64 previous length.
 * retrieve [dev] [B,P,D] (retrieve_per_seq) or [P,D]; best/accept_len [dev] [B] i32 (the
 * outputs of lantern_evaluate_posterior -- no host round trip).
 * Out: new_len [dev] [n_slabs] i64 = prev + a + 1 (the reference's current_length).
 */
int lantern_kv_gather(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev,
                      int n_slabs, int elem_bytes, int64_t outer, int64_t S_max, int64_t d,
                      const int64_t *retrieve, int retrieve_per_seq, int P, int D, const int32_t *best,
                      const int32_t *accept_len, int64_t *new_len, void *stream);

/* O10 accepted-hidden gather + token append + bonus-token draw.
 * Replaces models/ea_model_lumina_mgpt.py:748-750,773-785; models/ea_model_llamagen.py:957-984.
 * hidden [dev] [B,G,N,H] (G = cond/uncond groups) elem_bytes each; out_hidden [dev]
 * [B,G,D,H] (rows > accept_len zero-filled).  accepted_tokens [dev] [B,D] i64 (-1 pad).
 * Bonus token: inverse-CDF of sample_p [B,V] at u [B] (double): smallest i with
 * cumsum(p)[i] > u*sum(p); greedy (u == NULL): argmax.  token [dev] [B] i64.
 */
int lantern_accept_gather(const void *hidden, int elem_bytes, int B, int G, int N, int H,
                          const int64_t *retrieve, int retrieve_per_seq, int P, int D,
                          const int64_t *cand, const int32_t *best, const int32_t *accept_len,
                          const float *sample_p, int V, const double *u, void *out_hidden,
                          int64_t *accepted_tokens, int64_t *token, void *stream);

/* O5  static-tree drafter sampling epilogue with injected multinomial indices.
 * Replaces sample(): models/drafters/cnets_lumina_mgpt.py:936-955, cnets_llamagen.py:924-940.
 * probs [dev] [R,V] f32 (softmaxed), idx [dev] [R,k] i64 -> out_prob [dev] [R,k] f32. */
int lantern_sample_static(const float *probs, const int64_t *idx, int R, int V, int k, float *out_prob,
                          void *stream);

/* O9 + O10 in ONE launch -- the reference does both in update_inference_inputs
 * (models/ea_model_lumina_mgpt.py:731-799): KV rows of every slab move to their final positions and the
 * accepted hidden rows / tokens are gathered, from the same (best, accept_len).  Arguments as in
 * lantern_kv_gather followed by those of lantern_accept_gather (copy form: no bonus-token draw -- the
 * windowed evaluate_posterior draws it; D <= 8). */
int lantern_update_inference_inputs(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev,
                                    int n_slabs, int elem_bytes, int64_t outer, int64_t S_max, int64_t d,
                                    const int64_t *retrieve, int retrieve_per_seq, int P, int D,
                                    const int32_t *best, const int32_t *accept_len, int64_t *new_len,
                                    const void *hidden, int hid_elem_bytes, int B, int G, int N, int H,
                                    const int64_t *cand, void *out_hidden, int64_t *accepted_tokens, void *stream);

/* O11 drafter input contraction on MFMA: out = fc(cat(embed[ids]*scale, hidden)) (+bias).
 * Replaces Model.forward's input stage: models/drafters/cnets_lumina_mgpt.py:1071,1095-1098;
 * cnets_llamagen.py:642,679-680.  bf16 operands, f32 accumulate, bf16 output.
 * ids [dev] [M] i64, hidden [dev] [M,H] bf16, embed [dev] [vocab,H] bf16,
 * W [dev] [H,2H] bf16 row-major (nn.Linear weight), bias [dev] [H] bf16 or NULL. */
int lantern_drafter_fc(const int64_t *ids, const void *hidden, const void *embed, const void *W,
                       const void *bias, int M, int H, int vocab, float embed_scale, void *out,
                       void *stream);

/* a5  the reductions a drafting call makes over its attention mask, in one launch.  Replaces `position_ids = attention_mask.long().cumsum(-1) - 1`,
 * `len_posi = position_ids[:, -1] + 1` (models/drafters/cnets_lumina_mgpt.py:1180-1186) for the calls behind a cached prefix, and the first-visible-key
 * index the tree-attention path takes per row.  mask [dev] [B, S] (row_stride entries apart) of elem_bytes 1 (bool / uint8) or 8 (int64).
 * Out (each may be NULL): first [dev] [B] i64 = index of the first non-zero entry (0 for an all-zero row, as torch.argmax); count [dev] [B] i64 = number of
 * non-zero entries; bad [dev] [B] i64 = 1 when the row has a zero behind a one (not left padding), else 0. */
int lantern_mask_left_padding(const void *mask, int elem_bytes, int B, int64_t S, int64_t row_stride, int64_t *first, int64_t *count, int64_t *bad,
                              void *stream);

/* a5  the drafter's additive attention mask in one launch.  Replaces Model._prepare_decoder_attention_mask
 * (models/drafters/cnets_lumina_mgpt.py:1014-1050; cnets_llamagen.py:592-621): causal (when T > 1) + padding from the
 * boolean mask attn [dev] [B, attn_len] (bytes; columns >= attn_len count as attended, as the reference pads) or NULL,
 * then finfo(f32).min wherever tree_mask [dev] [tree_batch,1,t0,t1] f32 (tree_batch 1 = shared, or B) is zero in the
 * last t0 rows x last t1 columns.  out [dev] [B,1,T,past+T] f32. */
int lantern_drafter_attention_mask(const uint8_t *attn, int attn_len, const float *tree_mask, int tree_batch, int t0,
                                   int t1, int B, int T, int past, float *out, void *stream);

/* 8f-2 (next row, first half)  lm_head restricted to the vocabulary rows the model's mask can let through:
 *   out[m, out_col0 + n] = sum_k A[m,k] * W[row_lo + n, k] (+ bias[row_lo + n]),  n < n_rows,  bf16 in / f32 accumulate / bf16 out.
 * Replaces `head(out_hidden)` in the drafter (cnets_lumina_mgpt.py:1212,1287; cnets_anole.py:835,876) for models
 * whose drafted rows are masked to the image-token ids right after it: 8192 of Lumina's / Anole's 65536 rows of W are
 * read (64 MiB instead of 512 MiB per call).  A [dev] [M,K] bf16 (M <= 128), W [dev] [vocab,K] bf16 (nn.Linear layout),
 * bias [dev] [vocab] bf16 or NULL, out [dev] [M, out_stride] bf16 (only columns [out_col0, out_col0+n_rows) are written). */
int lantern_linear_rows(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                        int out_stride, int out_col0, void *stream);

/* 8f-2 (next row: the drafter's decoder layer at its decode shape, M = 2 x top_k rows)  Pieces of ChameleonDecoderLayer
 * (models/drafters/cnets_lumina_mgpt.py:769-843; lantern_amd/drafters/decoder_layer.py sequences them):
 *  - lantern_linear_rows_epilogue: lantern_linear_rows with the layer's element-wise tail in the GEMM epilogue (M <= 32):
 *      LANTERN_EPI_RESIDUAL  out = bf16(bf16(A W^T + bias) + aux[m, n])      `residual + o_proj(x)`, `residual + down_proj(x)` (:827, :832)
 *      LANTERN_EPI_SILU_MUL  out = bf16(silu(bf16(A Wg^T)) * bf16(A Wu^T))   ChameleonMLP :371-373; W holds gate rows [row_lo, +n_rows) and the
 *                            matching up rows pair_rows further down (one concatenated [2I, K] weight): the [M, 2I] intermediate never exists
 *  - lantern_rmsnorm_rows: ChameleonRMSNorm (:209-223) of M bf16 rows
 *  - lantern_qk_norm_rope: the head stage of ChameleonAttention (:481-499) on the fused q/k/v projection [B*T, (nq + 2 nk) d]: per-head
 *    ChameleonLayerNorm (:375-396; weights [model_parallel, d]), rotary at position_ids [B, T] from cos / sin tables [table_rows, d] bf16,
 *    outputs q [B, nq, T, d] and k / v written at rows [kv_row0, kv_row0 + T) of [B, nk, kv_rows, d] buffers, bf16 (d = 64 or 128): kv_rows = T,
 *    kv_row0 = 0 for fresh tensors, or a preallocated cache appended in place (no torch.cat of the whole cache per call). */
#define LANTERN_EPI_RESIDUAL 1
#define LANTERN_EPI_SILU_MUL 2
int lantern_linear_rows_epilogue(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                                 int out_stride, int out_col0, int epilogue, const void *aux, int aux_stride, int pair_rows, void *stream);
/* The same products for ANY number of rows on a weight packed by lantern_pack_linear_weight (the layout the stream-K kernels read; a gate / up pair
 * packed with pair_rows = rows of the gate half) -- the drafter's prompt prefill, where the layer sees hundreds of rows at once
 * (cnets_lumina_mgpt.py:1066-1098, first call of topK_generate).  epilogue 0 (bias only), LANTERN_EPI_RESIDUAL, LANTERN_EPI_SILU_MUL;
 * out [dev] [M, out_stride >= n_rows] bf16; K % 64 == 0.  Row blocks of 128 (64 for the gate / up pair) re-stream the weight tile. */
int lantern_linear_rows_packed(const void *A, const void *W_packed, const void *bias, int M, int K, int n_rows, void *out, int out_stride,
                               int epilogue, const void *aux, int aux_stride, int pair_rows, void *stream);
/* The same product for narrow outputs, K split over `ksplit` workgroups per 32-column tile (o_proj / down_proj have 128 tiles for 256 CUs):
 * out = bf16(A W^T + bias) (+ residual, rounded again); workspace [dev] ksplit * M * n_rows floats; two launches, deterministic sums. */
int lantern_linear_rows_splitk(const void *A, const void *W, const void *bias, int M, int K, int n_rows, void *out, int out_stride,
                               const void *residual, int residual_stride, int ksplit, float *workspace, void *stream);
/* The same products (epilogue 0: bias only, LANTERN_EPI_RESIDUAL, LANTERN_EPI_SILU_MUL) in stream-K form, ONE launch, deterministic: the (tile, K)
 * space is cut into equal contiguous shares, one per workgroup, one workgroup per CU of the current device (hipDeviceAttributeMultiprocessorCount;
 * the LANTERN_SK_GROUPS environment variable overrides the count for tuning runs only), a tile's partial sums meet in the workspace and the
 * workgroup that completes a tile adds them in K order and runs the epilogue; two trips of loads in flight per wave.  The form the decoder layer
 * uses at its decode shape (M <= 32).  workspace: [dev] lantern_linear_rows_streamk_workspace(n_rows) bytes, 16-byte aligned, zero-filled ONCE
 * by the caller (the kernel leaves its counters zeroed); one launch at a time per workspace. */
size_t lantern_linear_rows_streamk_workspace(int n_rows);
int lantern_linear_rows_streamk(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                                int out_stride, int out_col0, int epilogue, const void *aux, int aux_stride, int pair_rows,
                                int packed, void *workspace, size_t workspace_bytes, void *stream);
/* packed != 0: W is the output of lantern_pack_linear_weight -- the same values re-laid out ONCE (at weight-load time) in the order the
 * kernel's waves consume them: per 32-row tile and 64-element K block one 4 KB brick [fragment 0..3][lane 0..63][8 bf16] (gate / up pairs:
 * the gate brick, then the up brick), so that a wave's load instruction is 1 KB contiguous and a workgroup's share one contiguous byte range
 * (row-major nn.Linear weights make every load 64 separate 16-byte pieces 8 - 22 KB apart).  K % 64 == 0; n_rows is padded to a multiple
 * of 32 with zero rows; pair_rows > 0: rows [0, n_rows) and [pair_rows, pair_rows + n_rows) interleaved per brick.  out: [dev]
 * lantern_pack_linear_weight_bytes(n_rows, K, pair_rows) bytes, 16-byte aligned. */
/* The same kernel behind two more callers at the drafting shape:
 *  - lantern_drafter_fc_streamk: O11 (lantern_drafter_fc) for M <= 32 rows, H % 64 == 0; W [H, 2H] row-major or packed (n_rows = H, K = 2H);
 *  - lantern_head_expand_streamk: lantern_head_expand with the head's window GEMM + CFG epilogue in stream-K form; W row-major [V, K] or the
 *    packed rows [row_lo, row_lo + n_cols). */
int lantern_drafter_fc_streamk(const int64_t *ids, const void *hidden, const void *embed, const void *W, const void *bias, int M, int H,
                               int vocab, float embed_scale, void *out, int packed, void *workspace, size_t workspace_bytes, void *stream);
int lantern_head_expand_streamk(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg,
                                int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id,
                                int eos_id, int top_k_filter, const float *scores_in, int top_k, void *workspace, int64_t *topk_index,
                                float *cu_scores, int64_t *topk_cs_index, float *scores_out, int packed, void *sk_workspace,
                                size_t sk_workspace_bytes, void *stream);
size_t lantern_pack_linear_weight_bytes(int n_rows, int K, int pair_rows);
int lantern_pack_linear_weight(const void *W, int n_rows, int K, int pair_rows, void *out, void *stream);
int lantern_rmsnorm_rows(const void *x, const void *weight, int M, int H, float eps, void *out, void *stream);
int lantern_qk_norm_rope(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const void *q_weight, const void *q_bias,
                         const void *k_weight, const void *k_bias, int model_parallel, const void *cos_table, const void *sin_table,
                         int table_rows, const int64_t *position_ids, void *q_out, void *k_out, void *v_out, int kv_rows, int kv_row0,
                         void *stream);
/* The head stage of LlamaAttention (the LlamaGen drafter's layer, models/drafters/cnets_llamagen.py:315-323, apply_rotary_emb :67-77) on the fused
 * q/k/v projection [B*T, (nq + 2 nk) d] bf16: no per-head norm; rotary on adjacent pairs (x[2p], x[2p+1]) -> (x0 c - x1 s, x1 c + x0 s) with
 * (c, s) = freqs[position][p], freqs [table_rows, d/2, 2] f32 [dev] (the model's table: LlamaGen's 2-D one, precompute_freqs_cis_2d :47-64), in
 * f32, rounded to bf16 once; position_ids [dev] int64 [B, T] (positions_per_batch_row != 0) or [T] shared by the batch rows (the reference's
 * `freqs_cis[position_ids].squeeze(0)`, :663).  Outputs as lantern_qk_norm_rope: q [B, nq, T, d], k / v at rows [kv_row0, kv_row0 + T) of
 * [B, nk, kv_rows, d] buffers (d = 64 or 128). */
int lantern_qk_rope_pairs(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const float *freqs, int table_rows,
                          const int64_t *position_ids, int positions_per_batch_row, void *q_out, void *k_out, void *v_out, int kv_rows, int kv_row0,
                          void *stream);

/* One drafting depth of the EAGLE-2 drafter from ONE host call (what lantern_verify_step is to the accept side): the loop body of topK_genrate
 * (models/drafters/cnets_lumina_mgpt.py:1271-1320, cnets_llamagen.py:783-821, cnets_anole.py:841-903) for the cond / uncond pair of one sequence --
 *   input stage  fc(cat(embed[ids] * scale, hidden))                                  (lantern_drafter_fc_streamk)
 *   the decoder layer at its decode shape (B = 2 rows x T = top_k tokens, B * T <= 32): [RMSNorm] -> fused q/k/v GEMM -> head stage (Chameleon: per-head
 *     layer norm + rotary; Llama: pair rotary from the freqs rows) writing k / v into the cache slabs in place -> tree attention over the cache with the
 *     tree block as ancestor words -> o_proj + residual -> RMSNorm -> gate/up GEMM with silu * up -> down_proj + residual
 *   the head on the window rows with the CFG mix, the model's processors, log-softmax, top-k, cumulative scores, best top_k of T * top_k
 *     (lantern_head_expand_streamk)
 *   the next depth's inputs: hidden rows of the chosen parents, their tokens, the parents' indices into the score list, the ancestor words of the new
 *     tree rows --
 * every launch enqueued on `stream` by this call, every buffer caller-allocated, nothing returned to the host.  Weights: bf16, the GEMM weights as
 * lantern_pack_linear_weight laid them out (`*_packed` != 0) or row-major.  B must be 2 (cond, then uncond). */
typedef struct lantern_draft_depth_args {
    void *stream;
    int32_t layer_kind;                /* 0: Chameleon layer (Lumina-mGPT / Anole), 1: Llama layer (LlamaGen) */
    int32_t B, T, H, n_q_heads, n_kv_heads, head_dim, inter, vocab;
    float eps1, eps2, embed_scale, cfg;
    /* input stage */
    const int64_t *ids;                /* [dev] [B * T] token ids (the T tree tokens, repeated per row) */
    const void *hidden_in;             /* [dev] [B, T, H] bf16 */
    const void *embed, *fc_w, *fc_b;   /* embed [vocab, H]; fc_w [H, 2H] (packed: lantern_pack_linear_weight(fc_w, H, 2H, 0)); fc_b [H] or NULL */
    int32_t fc_packed, layer_packed;
    /* layer */
    const void *ln1_w;                 /* input RMSNorm weight [H], or NULL (EAGLE's layer 0 of the Llama drafter has none) */
    const void *qkv_w, *qkv_b;         /* fused q/k/v [(nq + 2 nk) d, H], bias or NULL */
    const void *o_w, *o_b, *ln2_w, *gate_up_w, *gate_up_b, *down_w, *down_b;      /* gate_up: cat(gate, up) rows, pair_rows = inter */
    /* head stage */
    const void *qn_w, *qn_b, *kn_w, *kn_b;       /* Chameleon: per-head layer-norm rows [model_parallel, d] */
    int32_t model_parallel, table_rows;
    const void *cos_table, *sin_table; /* Chameleon: [table_rows, d] bf16 */
    const float *freqs;                /* Llama: [table_rows, d/2, 2] f32 */
    const int64_t *position_ids;       /* [B, T] (positions_per_batch_row != 0) or [T] */
    int32_t positions_per_batch_row, reserved0;
    /* cache: k / v slabs [B, nk, kv_rows, d] bf16, this depth's rows written at [kv_row0, kv_row0 + T) */
    void *k_slab, *v_slab;
    int32_t kv_rows, kv_row0;
    /* tree */
    uint64_t *tree_bits;               /* [dev] [64] ancestor words of the tree's rows (row r: bit j set = key j of the tree block is visible); rows
                                        * [t1 - T, t1) are this depth's queries; the call writes rows [t1, t1 + T) for the next depth */
    int32_t t1, reserved1;             /* tree keys incl. this depth's T (t1 + T <= 64 when a next depth follows) */
    const int64_t *kv_start;           /* [B] first visible key per row (left padding) or NULL */
    /* head */
    const void *head_w, *head_b;       /* [vocab, H] row-major, or the packed rows [row_lo, row_lo + n_cols) */
    int32_t head_packed, row_lo, n_cols, model;          /* model: LANTERN_MODEL_LUMINA / _ANOLE / _PLAIN (window = the whole vocabulary) */
    const int64_t *head_pos;           /* Lumina: [T] positions of the rows for the grammar mask, or NULL */
    int64_t pos_base;
    int32_t w_latent, h_latent, newline_id, eos_id, top_k_filter, top_k;
    const float *scores_in;            /* [T] cumulative scores of this depth's tokens */
    int64_t *topk_index;               /* out [T, top_k] */
    float *cu_scores;                  /* out [T, top_k] */
    int64_t *topk_cs_index;            /* out [top_k] */
    float *scores_out;                 /* out [top_k] */
    /* next depth (NULL hidden_next: the last depth) */
    void *hidden_next;                 /* out [B, top_k, H] bf16: rows of this depth's output at the chosen parents */
    int64_t *ids_next;                 /* out [B * top_k] */
    int64_t *parents_next;             /* out [top_k]: topk_cs_index + parent_bias_next (the reference's `parents`, cnets_llamagen.py:798-803) */
    int64_t parent_bias_next;
    /* work buffers [dev], bf16 unless noted: x, xn, h1, hn, out [B*T, H]; qkv [B*T, (nq + 2 nk) d]; q [B, nq, 64, d] (zero-filled once by the caller);
     * attn [B, 64, H]; act [B*T, inter]; head_ws: lantern_head_expand_workspace(T, n_cols) bytes; sk_ws: lantern_linear_rows_streamk_workspace
     * (zero-filled once); ta_ws: lantern_tree_attention_workspace(B, nq, 64, d, kv_rows) bytes */
    void *x, *xn, *qkv, *q, *attn, *h1, *hn, *act, *out, *head_ws, *sk_ws, *ta_ws;
    size_t sk_ws_bytes, ta_ws_bytes;
    /* STATIC trees (EAGLE v1 / LANTERN++: the loop bodies of topK_generate(tree_type "static"), cnets_lumina_mgpt.py:1245-1328, and topK_genrate_v1,
     * cnets_llamagen.py:944-1023 / cnets_anole.py:1056-1171): n_draw > 0 replaces the expansion stage by lantern_head_sample -- this depth's T rows get
     * their distributions (probs_out [T, vocab] f32: the verify side's original_prob rows), n_draw draws without replacement each (ss_token / ss_prob
     * [T, n_draw]) -- and the next depth's inputs come from the tree's static tables: token j = ss_token.flat[next_gather[j]] (`idx.view(-1)[tree_indices]`),
     * hidden row j = this depth's output row next_rep[j] (`repeat_hidden`), T_next of them (hidden_next [B, T_next, H], ids_next [B * T_next]; T_next = 0:
     * the last depth).  tree_bits holds the whole tree's ancestor words (static: written once by the caller, rows [t1 - T, t1) are this depth's);
     * topk_index / cu_scores / topk_cs_index / scores_out / scores_in / parents_next are not used. */
    int32_t n_draw, T_next;
    const double *draw_u;              /* [T, n_draw] uniforms of the draws, or NULL with draw_idx */
    const int64_t *draw_idx;           /* [T, n_draw] injected draws (tests, recorded runs), or NULL */
    float *probs_out;                  /* out [T, vocab] f32 */
    int64_t *ss_token;                 /* out [T, n_draw] */
    float *ss_prob;                    /* out [T, n_draw] */
    const int32_t *next_gather, *next_rep;   /* [T_next] */
    /* The input stage through the tables (no materialised next inputs): in_rep != NULL makes THIS depth's input stage read token j of both batch rows
     * as ids[in_gather[j]] (`ids` = the flat draws one level up, [in_n_flat]) and hidden row (b, j) as hidden_in[(b * in_src_T + in_rep[j])] (`hidden_in` =
     * the rows one level up, [B, in_src_T, H]: the previous depth's `out`, or the prefill's last hidden row with in_src_T = 1).  A caller that chains depths
     * this way passes T_next = 0 everywhere: nothing is copied between depths. */
    const int32_t *in_gather, *in_rep;       /* [T] */
    int32_t in_src_T, in_n_flat;
} lantern_draft_depth_args;
int lantern_draft_depth(const lantern_draft_depth_args *args);

/* The static drafter's head stage alone (also the first sample of a drafting call, on the prefill's last hidden row: cnets_lumina_mgpt.py:1234-1243):
 * head(hidden) on the window rows with the CFG mix (A [2n, K] bf16: n conditional rows, then n unconditional; W / bias / packed / sk_workspace as
 * lantern_head_expand_streamk) -> the model's processors (Lumina grammar rows at pos_ids, InterleavedTopKLogitsWarper / top-k threshold top_k_filter) ->
 * softmax -> probs_out [n, V] f32 (zero outside the window, one-hot for a forced row) -> Model.sample (cnets_lumina_mgpt.py:936-955): n_draw draws without
 * replacement per row and their conditional probabilities p_i / (1 - sum_{j<i} p_j).  The draws come from injected uniforms draw_u [n, n_draw] f64 --
 * draw j = inverse CDF in token-id order of the row with the j tokens already drawn removed (the distribution of torch.multinomial(replacement=False);
 * torch's device RNG is not reproducible across devices) -- or are taken from draw_idx [n, n_draw] as given.  workspace: lantern_head_expand_workspace(n, n_cols). */
/* The inputs of a static tree's next depth from the draws of the rows above it (cnets_lumina_mgpt.py:1258-1262): ids_next[b, j] = ss_token.flat[gather[j]]
 * (`idx.view(-1)[tree_indices[i]]`), hidden_next[b, j] = out_hidden[b, rep[j]] (`repeat_hidden(out_hidden, repeat_nums[i])`), bf16 rows of H;
 * out_hidden [B, T, H], hidden_next [B, T_next, H], ids_next [B * T_next], gather / rep [T_next] i32 [dev].  lantern_draft_depth does this at the end of
 * a static depth; the first depth of a drafting call (behind lantern_head_sample on the prefill's last row) calls it directly. */
int lantern_draft_static_inputs(const int64_t *ss_token, int n_flat, const int32_t *gather, const int32_t *rep, const void *out_hidden, int B, int T, int H,
                                int T_next, void *hidden_next, int64_t *ids_next, void *stream);
int lantern_head_sample(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg, int model,
                        const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id, int eos_id, int top_k_filter,
                        int n_draw, const double *draw_u, const int64_t *draw_idx, void *workspace, float *probs_out, int64_t *ss_token,
                        float *ss_prob, int packed, void *sk_workspace, size_t sk_workspace_bytes, void *stream);

/* 8f-2 (next row, second half)  One drafter expansion depth from the hidden states to the top-k in two small launches, the head's
 * logits never in HBM:  head(hidden) restricted to the id window the model's mask lets through -> CFG combination in the GEMM's
 * epilogue (`uncond + cfg * (cond - uncond)` with torch's bf16 roundings) -> [n, n_cols] bf16 window (16 KB per row) -> per row:
 * grammar rows (Lumina newline / end of image), InterleavedTopKLogitsWarper threshold, log-softmax, top_k, + the parents'
 * cumulative scores -> best top_k of the n * top_k.  Replaces models/drafters/cnets_lumina_mgpt.py:1271-1320 (per depth:
 * lm_head on [2, n, H], CFG, MultiModalLogitsProcessor, InterleavedTopKLogitsWarper, log_softmax, topk, cu_scores, topk) and
 * the same lines of cnets_anole.py:876-905; same outputs as lantern_linear_rows + lantern_cfg_mask_topk + lantern_expand_dynamic.
 * A [dev] [2n, K] bf16: the n conditional rows, then the n unconditional ones (n <= 16); W [dev] [vocab, K] bf16, bias [dev] [vocab]
 * bf16 or NULL; window = ids [row_lo, row_lo + n_cols) (n_cols % 8 == 0, <= 8192); pos_ids [dev] [n] i64 (Lumina: position of the
 * token the row predicts, as for lantern_cfg_mask_topk) or NULL; top_k_filter: InterleavedTopKLogitsWarper's image_top_k (0 = off);
 * scores_in [dev] [n] f32 or NULL; workspace [dev] lantern_head_expand_workspace(n, n_cols) bytes, 16-byte aligned.
 * Out [dev]: topk_index [n, top_k] i64 (token ids), cu_scores [n, top_k] f32, topk_cs_index [top_k] i64, scores_out [top_k] f32. */
size_t lantern_head_expand_workspace(int n, int n_cols);
int lantern_head_expand(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg,
                        int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id, int eos_id,
                        int top_k_filter, const float *scores_in, int top_k, void *workspace, int64_t *topk_index,
                        float *cu_scores, int64_t *topk_cs_index, float *scores_out, void *stream);

/* 8f-3 (next row)  Attention of the target model's tree-verify forward over the KV cache in place:
 *   out[b, n, h*d + :] = softmax_f32(q[b,n,h,:] . K[b, h/(Hq/Hkv), key, :] * scale + mask[b][n][key]).bf16 @ V
 * replacing the eager matmul / additive-mask / softmax / matmul of the reference's attention
 * (models/kv_variants/modeling_lumina_mgpt_kv.py:433-442, modeling_llamagen_kv.py and modeling_anole_kv.py alike) and the
 * [B,1,N,S] f32 mask `_prepare_decoder_attention_mask` builds for it (modeling_lumina_mgpt_kv.py:1508-1546).  The mask is
 * implied: query node n of batch row b sees key j iff  kv_start[b] <= j < kv_len[b]-N  (cached prefix behind the left
 * padding) or  j = kv_len[b]-N+t with bit t of tree_bits[n] set (its ancestors and itself among the N tree keys, which the
 * caller has already appended to the cache -- KVCache.cat, kv_cache.py:52-66).
 *   q [dev] bf16, element strides (q_stride_b, q_stride_n, q_stride_h), last dim contiguous -- [B,N,Hq,d] as q_proj leaves
 *   it, or the transposed [B,Hq,N,d] view;  k_cache / v_cache [dev] bf16 [B,Hkv,S_max,d] slices of the reference's slab
 *   (element strides kv_stride_b, kv_stride_h; rows of d);  out [dev] bf16 [B,N,Hq*d] (strides out_stride_b, out_stride_n);
 *   kv_len [dev] [B] keys per batch row INCLUDING the N tree keys (NULL: max_kv_len for every row), kv_start [dev] [B] first
 *   visible key (NULL: 0);  max_kv_len: host-side upper bound of kv_len (sizes the launch, clamps kv_len);
 *   tree_bits [dev] u64 [N] (bits_per_row = 0) or [B,N] (bits_per_row = 1);  N <= 64, d in {64,128}, Hq % Hkv == 0;
 *   workspace [dev]: lantern_tree_attention_workspace(...) bytes (0 when B*Hq alone fills the GPU), 16-byte aligned.
 * Floating point: f32 scores and accumulation, probabilities rounded to bf16 before the V product like the reference. */
size_t lantern_tree_attention_workspace(int B, int Hq, int N, int d, int64_t max_kv_len);
int lantern_tree_attention(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int Hq, int Hkv, int N,
                           int d, int64_t q_stride_b, int64_t q_stride_n, int64_t q_stride_h, int64_t kv_stride_b,
                           int64_t kv_stride_h, int64_t out_stride_b, int64_t out_stride_n, const int64_t *kv_len,
                           const int64_t *kv_start, int64_t max_kv_len, const uint64_t *tree_bits, int bits_per_row,
                           float scale, void *workspace, size_t workspace_bytes, void *stream);

/* 8f-1 VQ-distance neighbour table: cdist + per-row ascending order, self excluded.
 * Replaces entrypoints/generate_codebook.py:53-65.  codebook [dev] [K,C] f32 ->
 * table [dev] [K,K-1] u16, K <= 16384 (LlamaGen).  workspace: unused (NULL). */
int lantern_build_vq_table(const float *codebook, int K, int C, uint16_t *table, void *workspace,
                           void *stream);

/* HBM layout for the hot path: the reference's table rows are K-1 = odd many uint16 (2-byte aligned, 16 KB apart)
 * of which only the first k+1 are ever read (ea_model_lumina_mgpt.py:662).  This copies columns [0, min(src_cols,
 * dst_cols)) into rows of dst_cols ids (zero padded); with dst_cols % 8 == 0 and a 16-byte aligned dst,
 * lantern_evaluate_posterior_window stages a level's neighbour ids with 16-byte loads (8192 x 1024 ids = 16 MiB
 * instead of 128 MiB for Lumina/Anole).  Pass the packed table with table_cols = dst_cols; k <= dst_cols - 1. */
int lantern_pack_vq_table(const uint16_t *src, int rows, int src_cols, uint16_t *dst, int dst_cols, void *stream);

/* Measurement aid.  Arms a (start, stop) hipEvent_t pair on the calling thread: the NEXT launch of
 * lantern_cfg_mask_topk_window / lantern_evaluate_posterior_window / lantern_kv_gather /
 * lantern_update_inference_inputs from this thread records them at kernel begin and kernel end
 * (hipExtLaunchKernelGGL), then disarms.  hipEventElapsedTime(start, stop) is then the kernel-only duration -- the
 * figure rocprofv3 --kernel-trace reports -- without the dispatch gap a hipEventRecord bracket includes. */
int lantern_profile_next_launch(void *start_event, void *stop_event);

#ifdef __cplusplus
}
#endif
#endif /* LANTERN_HIP_H */
