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
#include "window_dev.h"

namespace lantern {

constexpr int TD_MAX_SCORES = 2048;
constexpr int TD_EPL = TD_MAX_SCORES / 64;  // elements per lane (blocked layout)

// Four wavefronts per sequence (see tree_dynamic_dev.h)
constexpr int TD_WAVES = 4;
#ifdef TD_TRACE
__device__ unsigned long long g_td_trace[32];
#define TD_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_td_trace[i] = __builtin_readcyclecounter(); } while (0)
#else
#define TD_STAMP(i) do { } while (0)
#endif
}  // namespace lantern
#include "tree_dynamic_dev.h"
namespace lantern {
template <int EPL>
__global__ __launch_bounds__(64 * TD_WAVES) void tree_dynamic_finalize_kernel(const TdArgs ta) {
    td_finalize_body<EPL, TD_WAVES>(ta, blockIdx.x);
}

// ----------------------------------------------------------------------------- O3
constexpr int EX_THREADS = 256;
constexpr int EX_NW = EX_THREADS / 64;
constexpr int EX_MAX_K = 16;

// One workgroup per (sequence, row): log_softmax statistics + the top_k (<= 16) entries, ties towards the lower index.
// Two reads of the row (the second one from L2) instead of top_k + 2:
//   pass 1  per-thread maximum.  The top_k-th largest of the 256 thread maxima is a lower bound of the row's top_k-th
//           largest value (they are 256 distinct elements), so only entries >= that bound can be in the answer;
//   pass 2  sum of exp(x - max) (same per-thread order as before) and the candidates >= bound appended to an LDS list
//           (a few dozen entries on ordinary rows); then top_k rounds of a block-wide argmax over that list.
// Rows with fewer than top_k finite entries (bound = -inf) or more than EX_CAND candidates (heavy ties) take the plain
// iterative path over the row itself.
constexpr int EX_CAND = 2048;
__global__ __launch_bounds__(EX_THREADS) void expand_rows_kernel(const float *__restrict__ logits, const float *__restrict__ scores_in,
                                                                 int n_rows, int V, int top_k, int64_t *__restrict__ topk_index,
                                                                 float *__restrict__ cu_scores) {
    __shared__ float s_redf[2 * EX_NW];
    __shared__ double s_redd[2 * EX_NW];
    __shared__ float s_bv[EX_NW];
    __shared__ int s_bi[EX_NW], s_bs[EX_NW];
    __shared__ int s_taken[EX_MAX_K];
    __shared__ float s_tmax[EX_THREADS];
    __shared__ float s_cv[EX_CAND];
    __shared__ int s_ci[EX_CAND];
    __shared__ float s_bound;
    __shared__ int s_n;
    const int rowid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *x = logits + (size_t)rowid * V;
    const float NEG_INF = -__builtin_inff();
    int ph = 0;
    float tm = NEG_INF;
    for (int i = tid; i < V; i += EX_THREADS) tm = fmaxf(tm, x[i]);
    s_tmax[tid] = tm;
    if (tid == 0) {
        s_bound = NEG_INF;
        s_n = 0;
    }
    const float m = block_max<EX_NW>(tm, s_redf, ph);          // (barrier inside: s_tmax is visible afterwards)
    {
        int rank = 0;                                           // thread maxima ahead of mine in (value desc, thread asc) order
        for (int t = 0; t < EX_THREADS; ++t) {
            const float o = s_tmax[t];
            rank += (o > tm) || (o == tm && t < tid);
        }
        if (rank == top_k - 1) s_bound = tm;
    }
    __syncthreads();
    const float bound = s_bound;
    const bool collect = bound > NEG_INF;
    double s = 0.0;
    for (int i = tid; i < V; i += EX_THREADS) {
        const float v = x[i];
        s += (double)expf(v - m);
        if (collect && v >= bound) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < EX_CAND) {
                s_cv[slot] = v;
                s_ci[slot] = i;
            }
        }
    }
    const float ls = logf((float)block_sum<double, EX_NW>(s, s_redd, ph));
    const float sc = scores_in ? scores_in[rowid] : 0.0f;
    __syncthreads();
    const int n_cand = s_n;
    const bool fast = collect && n_cand <= EX_CAND;            // n_cand >= top_k whenever collect
    for (int t = 0; t < top_k; ++t) {
        float bv = NEG_INF;
        int bi = 0x7fffffff, bs = -1;
        if (fast) {
            for (int c = tid; c < n_cand; c += EX_THREADS) {
                const int i = s_ci[c];
                const float v = s_cv[c];
                if (i >= 0 && (v > bv || (v == bv && i < bi))) {
                    bv = v;
                    bi = i;
                    bs = c;
                }
            }
        } else {
            for (int i = tid; i < V; i += EX_THREADS) {
                bool taken = false;
                for (int q = 0; q < t; ++q) taken |= (s_taken[q] == i);
                const float v = x[i];
                if (!taken && (v > bv || (v == bv && i < bi))) {
                    bv = v;
                    bi = i;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const int os = __shfl_xor(bs, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
                bs = os;
            }
        }
        if (lane == 0) {
            s_bv[wave] = bv;
            s_bi[wave] = bi;
            s_bs[wave] = bs;
        }
        __syncthreads();
        if (tid == 0) {
            float v = s_bv[0];
            int i = s_bi[0], sl = s_bs[0];
            for (int w = 1; w < EX_NW; ++w)
                if (s_bv[w] > v || (s_bv[w] == v && s_bi[w] < i)) {
                    v = s_bv[w];
                    i = s_bi[w];
                    sl = s_bs[w];
                }
            s_taken[t] = i;
            if (fast && sl >= 0) s_ci[sl] = -1;                 // taken
            topk_index[(size_t)rowid * top_k + t] = i;
            cu_scores[(size_t)rowid * top_k + t] = ((v - m) - ls) + sc;
        }
        __syncthreads();
    }
}

// 8f-2 (second half, phase 2): what expand_rows_kernel does, on the CFG-combined bf16 WINDOW of a drafter row (the output of
// linear_rows_cfg_kernel) with the model's processors applied on the way in: the Lumina grammar rows (newline / end of image:
// one-hot), InterleavedTopKLogitsWarper's threshold, then log-softmax statistics and the top_k entries (ties to the lower id).
// One 256-thread workgroup per row, the row in registers (32 values per thread at W = 8192): 16 KB read once, against the
// 2 x 256 KB of the dense f32 row expand_rows_kernel reads (cnets_lumina_mgpt.py:1287-1318).
constexpr int EXW_C8 = 4;            // 4 chunks of 8 ids per thread: W <= 8192 on 256 threads (the Chameleon image window), <= 16384 on 512 (LlamaGen's vocabulary)
template <int EXW_NT>
__global__ __launch_bounds__(EXW_NT) void expand_window_kernel(const uint16_t *__restrict__ win, int W, int win_lo, int V, int model,
                                                               const int64_t *__restrict__ pos_ids, int64_t pos_base, int w_latent,
                                                               int h_latent, int newline_id, int eos_id, int top_k_filter,
                                                               const float *__restrict__ scores_in, int top_k,
                                                               int64_t *__restrict__ topk_index, float *__restrict__ cu_scores) {
    constexpr int NW = EXW_NT / 64, NV4 = 2 * EXW_C8;
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[2 * NW];
    __shared__ double s_redd[2 * NW];
    __shared__ float s_bv[NW];
    __shared__ int s_bi[NW];
    __shared__ int s_taken[EX_MAX_K];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float NEG_INF = -__builtin_inff();
    const float sc = scores_in ? scores_in[row] : 0.0f;
    int hot = -1;
    if (model == LANTERN_MODEL_LUMINA && pos_ids) {
        const int64_t n1 = pos_ids[row] - pos_base + 1;
        if (n1 == ((int64_t)w_latent + 1) * h_latent + 1) hot = eos_id;
        else if (py_mod64(n1, (int64_t)w_latent + 1) == 0) hot = newline_id;
    }
    if (hot >= 0) {          // forced row: probability 1 on `hot`, the other top_k - 1 places go to the lowest ids at -inf
        if (tid == 0) {
            topk_index[(size_t)row * top_k] = hot;
            cu_scores[(size_t)row * top_k] = sc;
            int next = 0;
            for (int t = 1; t < top_k; ++t) {
                if (next == hot) ++next;
                topk_index[(size_t)row * top_k + t] = next++;
                cu_scores[(size_t)row * top_k + t] = NEG_INF;
            }
        }
        return;
    }
    float4 r[NV4];
#pragma unroll
    for (int it = 0; it < EXW_C8; ++it) {
        const int ch = tid + it * EXW_NT;
        uint4 q = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);          // -inf pairs
        if (ch * 8 < W) q = *reinterpret_cast<const uint4 *>(win + (size_t)row * W + ch * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __uint_as_float((j & 1) ? (w[j >> 1] & 0xffff0000u) : (w[j >> 1] << 16));
        r[2 * it] = make_float4(o[0], o[1], o[2], o[3]);
        r[2 * it + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (top_k_filter > 0 && top_k_filter < V && top_k_filter <= W) {
        const float thr = kth_largest_hist_bf16<EXW_NT, NV4>(r, top_k_filter, s_hist);
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x; r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z; r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    int ph = 0;
    float tm = NEG_INF;
#pragma unroll
    for (int it = 0; it < NV4; ++it) tm = fmaxf(fmaxf(tm, fmaxf(r[it].x, r[it].y)), fmaxf(r[it].z, r[it].w));
    const float m = block_max<NW>(tm, s_redf, ph);
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < NV4; ++it)
        s += (double)expf(r[it].x - m) + (double)expf(r[it].y - m) + (double)expf(r[it].z - m) + (double)expf(r[it].w - m);
    const float ls = logf((float)block_sum<double, NW>(s, s_redd, ph));
    // ---- the top_k entries.  Fast form: the top_k-th largest value by one more radix select over the register tile; the few entries at or above
    // it (top_k, plus ties at the threshold) go to an LDS list; one wavefront ranks them (value, then the lower id) and lane `rank` writes slot `rank`.
    // Same order, same score arithmetic as the round-by-round arg-max below, which stays for the rows the list cannot hold (fewer finite entries
    // than top_k: the -inf tail; more than 64 entries at the threshold).  10 rows of the 7B head: 35 -> 12 us per launch.
    {
        __shared__ float s_cv[64];
        __shared__ int s_ci[64];
        __shared__ int s_cn;
        if (tid == 0) s_cn = 0;
        const float thr = kth_largest_hist_bf16<EXW_NT, NV4>(r, top_k, s_hist);          // (its barriers order the counter's reset)
        if (thr > NEG_INF) {
#pragma unroll
            for (int it = 0; it < NV4; ++it) {
                const float v[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (v[c] >= thr) {
                        const int slot = atomicAdd(&s_cn, 1);
                        if (slot < 64) {
                            s_cv[slot] = v[c];
                            s_ci[slot] = win_lo + ((tid + (it >> 1) * EXW_NT) * 8 + (it & 1) * 4 + c);
                        }
                    }
            }
            __syncthreads();
            const int cn = s_cn;
            if (cn <= 64) {          // (workgroup-uniform)
                if (wave == 0 && lane < cn) {
                    const float v = s_cv[lane];
                    const int id = s_ci[lane];
                    int rank = 0;
                    for (int j = 0; j < cn; ++j) {
                        const float ov = s_cv[j];
                        const int oi = s_ci[j];
                        rank += (ov > v || (ov == v && oi < id)) ? 1 : 0;
                    }
                    if (rank < top_k) {
                        topk_index[(size_t)row * top_k + rank] = id;
                        cu_scores[(size_t)row * top_k + rank] = ((v - m) - ls) + sc;
                    }
                }
                return;
            }
            __syncthreads();
        }
    }
    // top_k rounds of a block-wide arg-max (value, then the lower id); a thread's 32 entries carry a taken mask
    unsigned taken = 0u;
    int low_next = 0;                     // next id of the -inf tail (thread 0)
    for (int t = 0; t < top_k; ++t) {
        float bv = NEG_INF;
        int bi = 0x7fffffff;
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            const float v[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int slot = it * 4 + c;
                const int id = win_lo + ((tid + (it >> 1) * EXW_NT) * 8 + (it & 1) * 4 + c);
                if (!((taken >> slot) & 1u) && v[c] > NEG_INF && (v[c] > bv || (v[c] == bv && id < bi))) {
                    bv = v[c];
                    bi = id;
                }
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
        float v = s_bv[0];
        int i = s_bi[0];
        for (int w = 1; w < NW; ++w)
            if (s_bv[w] > v || (s_bv[w] == v && s_bi[w] < i)) {
                v = s_bv[w];
                i = s_bi[w];
            }
        if (v > NEG_INF) {               // mark it taken in its owner's registers
            const int wi = i - win_lo, chn = wi >> 3;
            if ((chn % EXW_NT) == tid) taken |= 1u << ((chn / EXW_NT) * 8 + (wi & 7));
        }
        if (tid == 0) {
            if (!(v > NEG_INF)) {        // fewer finite entries than top_k: the lowest ids not yet listed, at -inf
                bool again = true;
                while (again) {
                    again = false;
                    for (int q = 0; q < t; ++q)
                        if (s_taken[q] == low_next) {
                            ++low_next;
                            again = true;
                        }
                }
                i = low_next++;
            }
            s_taken[t] = i;
            topk_index[(size_t)row * top_k + t] = i;
            cu_scores[(size_t)row * top_k + t] = ((v - m) - ls) + sc;
        }
        __syncthreads();
    }
}

// one wavefront per sequence: top_k of the flattened n_rows*top_k cumulative scores (<= 256: four per lane, in registers -- the rounds of the arg-max
// re-read them from memory before: 11.6 -> 3 us per launch)
__global__ __launch_bounds__(64) void expand_merge_kernel(const float *__restrict__ cu_scores, int nf, int top_k,
                                                          int64_t *__restrict__ topk_cs_index, float *__restrict__ scores_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *cu = cu_scores + (size_t)b * nf;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (lane + 64 * q < nf) ? cu[lane + 64 * q] : -__builtin_inff();
    unsigned taken = 0u;          // bit q: this lane's entry lane + 64 q is listed
    for (int t = 0; t < top_k; ++t) {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < nf && !((taken >> q) & 1u) && (v[q] > bv || (v[q] == bv && i < bi))) {
                bv = v[q];
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
        if ((bi & 63) == lane && bi != 0x7fffffff) taken |= 1u << (bi >> 6);
        if (lane == 0) {
            topk_cs_index[(size_t)b * top_k + t] = bi;
            scores_out[(size_t)b * top_k + t] = bv;
        }
    }
}

}  // namespace lantern

using namespace lantern;

#ifdef TD_TRACE
extern "C" int lantern_debug_td_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lantern::g_td_trace), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

static int td_launch(const float *scores, const int64_t *tokens, const int64_t *parents, const int64_t *sample_token, int B, int n_scores,
                     int n_parents, int top_k, int total_tokens, int sort_rows, int64_t *draft_tokens, float *mask, int64_t *pos_ids,
                     int64_t *retrieve, int32_t *n_leaf, int32_t *max_depth, const TdCand &cd, void *stream) {
    LANTERN_CHECK_ARG(scores && tokens && parents && sample_token && draft_tokens && mask && pos_ids && retrieve && n_leaf && max_depth,
                      "tree_dynamic_finalize: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && top_k > 0 && total_tokens >= 1 && total_tokens <= 63, "tree_dynamic_finalize: total_tokens=%d must be in [1,63]",
                      total_tokens);
    LANTERN_CHECK_ARG(n_scores >= total_tokens && n_scores <= TD_MAX_SCORES, "tree_dynamic_finalize: n_scores=%d out of range", n_scores);
    LANTERN_CHECK_ARG(n_parents * top_k >= n_scores, "tree_dynamic_finalize: n_parents=%d too small for n_scores=%d", n_parents, n_scores);
    if (B == 0) return LANTERN_OK;
    const TdArgs ta{scores, tokens, parents, sample_token, n_scores, n_parents, top_k, total_tokens, sort_rows, draft_tokens, mask, pos_ids, retrieve,
                    n_leaf, max_depth, cd};
    if (n_scores <= 8 * 64) hipLaunchKernelGGL(tree_dynamic_finalize_kernel<8>, dim3(B), dim3(64 * TD_WAVES), 0, (hipStream_t)stream, ta);
    else hipLaunchKernelGGL(tree_dynamic_finalize_kernel<TD_EPL>, dim3(B), dim3(64 * TD_WAVES), 0, (hipStream_t)stream, ta);
    LANTERN_CHECK_LAUNCH("tree_dynamic_finalize");
    return LANTERN_OK;
}

extern "C" int lantern_tree_dynamic_finalize(const float *scores, const int64_t *tokens, const int64_t *parents,
                                             const int64_t *sample_token, int B, int n_scores, int n_parents, int top_k,
                                             int total_tokens, int sort_rows, int64_t *draft_tokens, float *mask,
                                             int64_t *pos_ids, int64_t *retrieve, int32_t *n_leaf, int32_t *max_depth,
                                             void *stream) {
    return td_launch(scores, tokens, parents, sample_token, B, n_scores, n_parents, top_k, total_tokens, sort_rows, draft_tokens, mask, pos_ids,
                     retrieve, n_leaf, max_depth, TdCand{}, stream);
}

extern "C" int lantern_tree_dynamic_candidates(const float *scores, const int64_t *tokens, const int64_t *parents,
                                               const int64_t *sample_token, int B, int n_scores, int n_parents, int top_k,
                                               int total_tokens, int sort_rows, int64_t *draft_tokens, float *mask, int64_t *pos_ids,
                                               int64_t *retrieve, int32_t *n_leaf, int32_t *max_depth, const int64_t *seq_len, int P, int D,
                                               int64_t *cand, int64_t *retrieve_pd, int32_t *row_index, int64_t *pos_abs, void *stream) {
    const int N = total_tokens + 1;
    LANTERN_CHECK_ARG(cand && P > 0 && P <= N && D > 0 && D <= N, "tree_dynamic_candidates: cand missing or bad sizes (P, D <= N)");
    return td_launch(scores, tokens, parents, sample_token, B, n_scores, n_parents, top_k, total_tokens, sort_rows, draft_tokens, mask, pos_ids,
                     retrieve, n_leaf, max_depth, TdCand{seq_len, cand, retrieve_pd, pos_abs, row_index, P, D}, stream);
}

namespace lantern {
int launch_linear_rows_cfg(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, float cfg, void *win,
                           hipStream_t st);
int launch_linear_rows_cfg_streamk(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, float cfg, void *win, int packed,
                                   void *workspace, size_t workspace_bytes, hipStream_t st);
}

extern "C" size_t lantern_head_expand_workspace(int n, int n_cols) { return (size_t)n * (size_t)n_cols * 2; }

static int head_expand_impl(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg,
                            int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id, int eos_id,
                            int top_k_filter, const float *scores_in, int top_k, void *workspace, int64_t *topk_index,
                            float *cu_scores, int64_t *topk_cs_index, float *scores_out, int packed, void *sk_workspace, size_t sk_workspace_bytes,
                            void *stream) {
    LANTERN_CHECK_ARG(A && W && workspace && topk_index && cu_scores && topk_cs_index && scores_out, "head_expand: null buffer");
    LANTERN_CHECK_ARG(n > 0 && n <= 16 && K > 0 && K % 16 == 0, "head_expand: n=%d drafter rows (<= 16 cond + 16 uncond), K=%d (multiple of 16)", n, K);
    LANTERN_CHECK_ARG(row_lo >= 0 && n_cols > 0 && n_cols % 8 == 0 && n_cols <= 8 * 512 * EXW_C8 && row_lo + n_cols <= V,
                      "head_expand: window [%d,+%d) must be a multiple of 8 ids, <= %d wide, inside V", row_lo, n_cols, 8 * 512 * EXW_C8);
    LANTERN_CHECK_ARG(top_k > 0 && top_k <= EX_MAX_K && top_k <= V && n * top_k <= 256, "head_expand: bad top_k");
    LANTERN_CHECK_ARG(model == LANTERN_MODEL_LUMINA || model == LANTERN_MODEL_ANOLE || (model == LANTERN_MODEL_PLAIN && row_lo == 0 && n_cols == V),
                      "head_expand: for models whose drafted rows are masked to one id window (Lumina, Anole), or an unmasked model whose window is the "
                      "whole vocabulary (LlamaGen: LANTERN_MODEL_PLAIN, row_lo = 0, n_cols = V <= 16384)");
    if (model == LANTERN_MODEL_LUMINA && pos_ids)
        LANTERN_CHECK_ARG(w_latent > 0 && h_latent > 0 && newline_id >= 0 && newline_id < V && eos_id >= 0 && eos_id < V, "head_expand: Lumina needs latent dims and syntax ids");
    LANTERN_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "head_expand: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (sk_workspace) {
        const int rc = launch_linear_rows_cfg_streamk(A, W, bias, n, K, row_lo, n_cols, cfg, workspace, packed, sk_workspace, sk_workspace_bytes, st);
        if (rc) return rc;
    } else {
        LANTERN_CHECK_ARG(!packed, "head_expand: a packed weight needs the stream-K workspace");
        launch_linear_rows_cfg(A, W, bias, n, K, row_lo, n_cols, cfg, workspace, st);
    }
    if (n_cols <= 8 * 256 * EXW_C8)
        hipLaunchKernelGGL(expand_window_kernel<256>, dim3(n), dim3(256), 0, st, (const uint16_t *)workspace, n_cols, row_lo, V, model, pos_ids, pos_base,
                           w_latent, h_latent, newline_id, eos_id, top_k_filter, scores_in, top_k, topk_index, cu_scores);
    else
        hipLaunchKernelGGL(expand_window_kernel<512>, dim3(n), dim3(512), 0, st, (const uint16_t *)workspace, n_cols, row_lo, V, model, pos_ids, pos_base,
                           w_latent, h_latent, newline_id, eos_id, top_k_filter, scores_in, top_k, topk_index, cu_scores);
    hipLaunchKernelGGL(expand_merge_kernel, dim3(1), dim3(64), 0, st, cu_scores, n * top_k, top_k, topk_cs_index, scores_out);
    LANTERN_CHECK_LAUNCH("head_expand");
    return LANTERN_OK;
}

extern "C" int lantern_head_expand(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg,
                                   int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id, int eos_id,
                                   int top_k_filter, const float *scores_in, int top_k, void *workspace, int64_t *topk_index,
                                   float *cu_scores, int64_t *topk_cs_index, float *scores_out, void *stream) {
    return head_expand_impl(A, W, bias, n, K, row_lo, n_cols, V, cfg, model, pos_ids, pos_base, w_latent, h_latent, newline_id, eos_id, top_k_filter,
                            scores_in, top_k, workspace, topk_index, cu_scores, topk_cs_index, scores_out, 0, nullptr, 0, stream);
}

// the same with the head's window GEMM in stream-K form (lantern_linear_rows_streamk's kernel with the CFG epilogue): W row-major [V, K], or --
// packed != 0 -- lantern_pack_linear_weight of its rows [row_lo, row_lo + n_cols); sk_workspace as for lantern_linear_rows_streamk
extern "C" int lantern_head_expand_streamk(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg,
                                           int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id,
                                           int eos_id, int top_k_filter, const float *scores_in, int top_k, void *workspace, int64_t *topk_index,
                                           float *cu_scores, int64_t *topk_cs_index, float *scores_out, int packed, void *sk_workspace,
                                           size_t sk_workspace_bytes, void *stream) {
    LANTERN_CHECK_ARG(sk_workspace, "head_expand_streamk: null stream-K workspace");
    return head_expand_impl(A, W, bias, n, K, row_lo, n_cols, V, cfg, model, pos_ids, pos_base, w_latent, h_latent, newline_id, eos_id, top_k_filter,
                            scores_in, top_k, workspace, topk_index, cu_scores, topk_cs_index, scores_out, packed, sk_workspace, sk_workspace_bytes, stream);
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
