// tree_dynamic.hip -- O3 (EAGLE-2 expansion step) and O4 (dynamic tree finalise).
//
// O4 runs ONE 64-lane wavefront per sequence: every tree in this path has <= 64 nodes, so
// a node is a lane, its ancestor set is one uint64, depth = popcount-1, leaves come from a
// ballot, and retrieve rows are parent-pointer walks -- the reference's .tolist() syncs and
// Python loops (cnets_llamagen.py:845-906) become register/LDS work with no host trip.
//
// Reference: models/drafters/cnets_llamagen.py:798-820 (O3), :831-912 (O4);
// cnets_lumina_mgpt.py:1303-1318, :1330-1393; cnets_anole.py:913-993.
#include "common.h"

namespace lantern {

constexpr int TD_MAX_SCORES = 2048;
constexpr int TD_EPL = TD_MAX_SCORES / 64;  // elements per lane (blocked layout)

__global__ __launch_bounds__(64) void tree_dynamic_finalize_kernel(
    const float *__restrict__ scores_, const int64_t *__restrict__ tokens_, const int64_t *__restrict__ parents_,
    const int64_t *__restrict__ sample_token, int n_scores, int n_parents, int top_k, int T, int sort_rows,
    int64_t *__restrict__ draft_tokens, float *__restrict__ mask, int64_t *__restrict__ pos_ids,
    int64_t *__restrict__ retrieve, int32_t *__restrict__ n_leaf, int32_t *__restrict__ max_depth) {
    __shared__ int s_sel[64];
    __shared__ int s_par[64];
    __shared__ int s_flag[64];
    __shared__ signed char s_rows[64][64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = T + 1;
    const float *scores = scores_ + (size_t)b * n_scores;
    const int64_t *tokens = tokens_ + (size_t)b * n_scores;
    const int64_t *parents = parents_ + (size_t)b * n_parents;

    // ---- top-T by score (ties -> lower flat index), kept in ascending index order
    const int E = (n_scores + 63) / 64;  // blocked: lane owns [lane*E, lane*E+E)
    uint32_t key[TD_EPL];
#pragma unroll
    for (int j = 0; j < TD_EPL; ++j) {
        const int idx = lane * E + j;
        key[j] = (j < E && idx < n_scores) ? float_key(scores[idx]) : 0u;  // 0 < key of any float
    }
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < TD_EPL; ++j) c += (j < E) && key[j] >= trial;
        if (wave_sum(c) >= T) prefix = trial;
    }
    int c_gt = 0, c_eq = 0;
#pragma unroll
    for (int j = 0; j < TD_EPL; ++j) {
        c_gt += (j < E) && key[j] > prefix;
        c_eq += (j < E) && key[j] == prefix;
    }
    const int need_eq = T - wave_sum(c_gt);  // how many threshold-valued entries to take
    // exclusive lane prefixes
    int inc_eq = c_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc_eq, o, 64);
        if (lane >= o) inc_eq += t;
    }
    int eq_before = inc_eq - c_eq;
    int c_sel = 0;
    uint32_t selbits = 0;  // which of my elements are selected
#pragma unroll
    for (int j = 0; j < TD_EPL; ++j) {
        if (j >= E) continue;
        bool s = key[j] > prefix;
        if (key[j] == prefix) {
            s = eq_before < need_eq;
            ++eq_before;
        }
        if (s) {
            selbits |= 1u << j;
            ++c_sel;
        }
    }
    int inc_sel = c_sel;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc_sel, o, 64);
        if (lane >= o) inc_sel += t;
    }
    int pos = inc_sel - c_sel;
#pragma unroll
    for (int j = 0; j < TD_EPL; ++j)
        if (j < E && (selbits >> j) & 1u) s_sel[pos++] = lane * E + j;
    __syncthreads();

    // ---- node = lane (0 = root); parent via searchsorted over the selected flat indices
    int par = 0;
    if (lane == 0) {
        draft_tokens[(size_t)b * N] = sample_token[b];
    } else if (lane < N) {
        const int flat = s_sel[lane - 1];
        draft_tokens[(size_t)b * N + lane] = tokens[flat];
        const int64_t dp = parents[flat / top_k];
        if (dp != 0) {
            const int64_t keyv = dp - 1;
            int p = 0;
            while (p < T && s_sel[p] < keyv) ++p;
            par = p + 1;
        }
    }
    s_par[lane] = par;
    s_flag[lane] = 0;
    __syncthreads();
    if (lane >= 1 && lane < N) s_flag[par] = 1;  // non-leaf marks
    // ancestor set: walk the parent pointers
    unsigned long long anc = 1ull;
    if (lane < N) {
        int cur = lane;
        for (int guard = 0; cur > 0 && guard < 64; ++guard) {
            anc |= 1ull << cur;
            cur = s_par[cur];
        }
    }
    const int depth = lane < N ? __popcll(anc) - 1 : 0;
    int md = depth;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) md = max(md, __shfl_xor(md, o, 64));
    const int MD = md + 1;
    __syncthreads();
    if (lane < N) {
        pos_ids[(size_t)b * N + lane] = depth;
        float *mrow = mask + ((size_t)b * N + lane) * N;
        for (int j = 0; j < N; ++j) mrow[j] = (float)((anc >> j) & 1ull);
    }
    // ---- leaves -> rows
    const bool leaf = lane < N && !s_flag[lane];
    const unsigned long long leafmask = __ballot(leaf);
    const int nl = __popcll(leafmask);
    const int rid = __popcll(leafmask & ((1ull << lane) - 1ull));
    if (leaf) {
        for (int j = 0; j < 64; ++j) s_rows[rid][j] = -1;
        int cur = lane;
        for (int j = depth; j >= 0; --j) {
            s_rows[rid][j] = (signed char)cur;
            cur = cur > 0 ? s_par[cur] : 0;
        }
    }
    __syncthreads();
    int out_row = rid;
    if (leaf && sort_rows) {
        // rank among rows, key = entries with -1 -> T+5 (always larger than any node id)
        int rank = 0;
        for (int o = 0; o < nl; ++o) {
            if (o == rid) continue;
            int cmp = 0;
            for (int j = 0; j < MD && cmp == 0; ++j) {
                const int a = s_rows[o][j] < 0 ? T + 5 : s_rows[o][j];
                const int c = s_rows[rid][j] < 0 ? T + 5 : s_rows[rid][j];
                cmp = (a < c) ? -1 : (a > c ? 1 : 0);
            }
            rank += (cmp < 0) || (cmp == 0 && o < rid);
        }
        out_row = rank;
    }
    int64_t *rbase = retrieve + (size_t)b * N * N;
    if (leaf)
        for (int j = 0; j < N; ++j) rbase[(size_t)out_row * N + j] = (j < MD) ? (int64_t)s_rows[rid][j] : -1;
    if (lane < N && lane >= nl)
        for (int j = 0; j < N; ++j) rbase[(size_t)lane * N + j] = -1;
    if (lane == 0) {
        n_leaf[b] = nl;
        max_depth[b] = MD;
    }
}

// ----------------------------------------------------------------------------- O3
constexpr int EX_THREADS = 256;
constexpr int EX_NW = EX_THREADS / 64;
constexpr int EX_MAX_K = 16;

// one workgroup per (sequence,row): log_softmax stats + iterative top-k (k <= 16)
__global__ __launch_bounds__(EX_THREADS) void expand_rows_kernel(const float *__restrict__ logits, const float *__restrict__ scores_in,
                                                                 int n_rows, int V, int top_k, int64_t *__restrict__ topk_index,
                                                                 float *__restrict__ cu_scores) {
    __shared__ float s_redf[2 * EX_NW];
    __shared__ double s_redd[2 * EX_NW];
    __shared__ float s_bv[EX_NW];
    __shared__ int s_bi[EX_NW];
    __shared__ int s_taken[EX_MAX_K];
    const int rowid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *x = logits + (size_t)rowid * V;
    const float NEG_INF = -__builtin_inff();
    int ph = 0;
    float m = NEG_INF;
    for (int i = tid; i < V; i += EX_THREADS) m = fmaxf(m, x[i]);
    m = block_max<EX_NW>(m, s_redf, ph);
    double s = 0.0;
    for (int i = tid; i < V; i += EX_THREADS) s += (double)expf(x[i] - m);
    const float ls = logf((float)block_sum<double, EX_NW>(s, s_redd, ph));
    const float sc = scores_in ? scores_in[rowid] : 0.0f;
    for (int t = 0; t < top_k; ++t) {
        float bv = NEG_INF;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += EX_THREADS) {
            bool taken = false;
            for (int q = 0; q < t; ++q) taken |= (s_taken[q] == i);
            const float v = x[i];
            if (!taken && (v > bv || (v == bv && i < bi))) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_bv[wave] = bv;
            s_bi[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            float v = s_bv[0];
            int i = s_bi[0];
            for (int w = 1; w < EX_NW; ++w)
                if (s_bv[w] > v || (s_bv[w] == v && s_bi[w] < i)) {
                    v = s_bv[w];
                    i = s_bi[w];
                }
            s_taken[t] = i;
            topk_index[(size_t)rowid * top_k + t] = i;
            cu_scores[(size_t)rowid * top_k + t] = ((v - m) - ls) + sc;
        }
        __syncthreads();
    }
}

// one wavefront per sequence: top_k of the flattened n_rows*top_k cumulative scores
__global__ __launch_bounds__(64) void expand_merge_kernel(const float *__restrict__ cu_scores, int nf, int top_k,
                                                          int64_t *__restrict__ topk_cs_index, float *__restrict__ scores_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *cu = cu_scores + (size_t)b * nf;
    unsigned long long taken_lo = 0, taken_hi = 0, taken_2 = 0, taken_3 = 0;  // up to 256 entries
    for (int t = 0; t < top_k; ++t) {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int i = lane; i < nf; i += 64) {
            const int w = i >> 6;
            const unsigned long long tk = w == 0 ? taken_lo : (w == 1 ? taken_hi : (w == 2 ? taken_2 : taken_3));
            const bool taken = (tk >> (i & 63)) & 1ull;
            const float v = cu[i];
            if (!taken && (v > bv || (v == bv && i < bi))) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        const int w = bi >> 6;
        const unsigned long long bitm = 1ull << (bi & 63);
        if (w == 0) taken_lo |= bitm;
        else if (w == 1) taken_hi |= bitm;
        else if (w == 2) taken_2 |= bitm;
        else taken_3 |= bitm;
        if (lane == 0) {
            topk_cs_index[(size_t)b * top_k + t] = bi;
            scores_out[(size_t)b * top_k + t] = bv;
        }
    }
}

}  // namespace lantern

using namespace lantern;

extern "C" int lantern_tree_dynamic_finalize(const float *scores, const int64_t *tokens, const int64_t *parents,
                                             const int64_t *sample_token, int B, int n_scores, int n_parents, int top_k,
                                             int total_tokens, int sort_rows, int64_t *draft_tokens, float *mask,
                                             int64_t *pos_ids, int64_t *retrieve, int32_t *n_leaf, int32_t *max_depth,
                                             void *stream) {
    LANTERN_CHECK_ARG(scores && tokens && parents && sample_token && draft_tokens && mask && pos_ids && retrieve && n_leaf && max_depth,
                      "tree_dynamic_finalize: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && top_k > 0 && total_tokens >= 1 && total_tokens <= 63, "tree_dynamic_finalize: total_tokens=%d must be in [1,63]",
                      total_tokens);
    LANTERN_CHECK_ARG(n_scores >= total_tokens && n_scores <= TD_MAX_SCORES, "tree_dynamic_finalize: n_scores=%d out of range", n_scores);
    LANTERN_CHECK_ARG(n_parents * top_k >= n_scores, "tree_dynamic_finalize: n_parents=%d too small for n_scores=%d", n_parents, n_scores);
    if (B == 0) return LANTERN_OK;
    hipLaunchKernelGGL(tree_dynamic_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, scores, tokens, parents, sample_token,
                       n_scores, n_parents, top_k, total_tokens, sort_rows, draft_tokens, mask, pos_ids, retrieve, n_leaf, max_depth);
    LANTERN_CHECK_LAUNCH("tree_dynamic_finalize");
    return LANTERN_OK;
}

extern "C" int lantern_expand_dynamic(const float *logits, const float *scores_in, int B, int n_rows, int V, int top_k,
                                      int64_t *topk_index, float *cu_scores, int64_t *topk_cs_index, float *scores_out,
                                      void *stream) {
    LANTERN_CHECK_ARG(logits && topk_index && cu_scores && topk_cs_index && scores_out, "expand_dynamic: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && n_rows > 0 && V > 0 && top_k > 0 && top_k <= EX_MAX_K && top_k <= V, "expand_dynamic: bad sizes");
    LANTERN_CHECK_ARG(n_rows * top_k <= 256, "expand_dynamic: n_rows*top_k=%d > 256", n_rows * top_k);
    if (B == 0) return LANTERN_OK;
    hipLaunchKernelGGL(expand_rows_kernel, dim3(B * n_rows), dim3(EX_THREADS), 0, (hipStream_t)stream, logits, scores_in, n_rows, V,
                       top_k, topk_index, cu_scores);
    hipLaunchKernelGGL(expand_merge_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cu_scores, n_rows * top_k, top_k,
                       topk_cs_index, scores_out);
    LANTERN_CHECK_LAUNCH("expand_dynamic");
    return LANTERN_OK;
}
