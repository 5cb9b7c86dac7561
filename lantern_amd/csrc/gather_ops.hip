// gather_ops.hip -- O6 candidate assembly, O9 KV-cache index gather, O10 accepted-hidden
// gather + token append + bonus-token draw, O5 static-tree sampling epilogue.
// All are index/byte movers: coalesced 16-byte accesses, indices read from device memory
// (the outputs of evaluate_posterior) so the step never round-trips to the host.
#include "common.h"
#include "gather_dev.h"
#include "prep_dev.h"
#include <cstdlib>

namespace lantern {

// ------------------------------------------------------------------------- O6
// models/ea_model_lumina_mgpt.py:525-554; models/ea_model_llamagen.py:676-706
constexpr int GC_MAX_N = 256, GC_PER = 2;
__global__ __launch_bounds__(256) void gather_candidates_kernel(const int64_t *__restrict__ ss_token, const float *__restrict__ ss_prob,
                                         const int64_t *__restrict__ sample_token, const int64_t *__restrict__ tree_indices,
                                         const int64_t *__restrict__ retrieve, int n_flat, int N, int PD,
                                         int64_t *__restrict__ tree_cand, int64_t *__restrict__ cand,
                                         float *__restrict__ cart_prob) {
    const int b = blockIdx.x;
    const int64_t *tok = ss_token + (size_t)b * n_flat;
    const float *prb = ss_prob ? ss_prob + (size_t)b * n_flat : nullptr;
    // tree_indices and retrieve are tiny and shared by every sequence: one round of loads into LDS, so the chain is
    // (tables, sample token) -> token/prob gather instead of retrieve -> tree_indices -> token
    __shared__ int s_ti[GC_MAX_N];
    const bool staged = N <= GC_MAX_N;
    if (staged)
        for (int n = threadIdx.x; n < N; n += blockDim.x) s_ti[n] = (int)tree_indices[n];
    const int64_t st = sample_token[b];
    int64_t rr[GC_PER];
#pragma unroll
    for (int u = 0; u < GC_PER; ++u) {
        const int i = threadIdx.x + u * 256;
        rr[u] = i < PD ? retrieve[i] : -1;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int64_t ti = staged ? (int64_t)s_ti[n] : tree_indices[n];
        tree_cand[(size_t)b * N + n] = (ti <= 0 || ti > n_flat) ? st : tok[ti - 1];      // index 0 = the sample token; out-of-range ids read nothing
    }
#pragma unroll
    for (int u = 0; u < GC_PER; ++u) {
        const int i = threadIdx.x + u * 256;
        if (i < PD) {
            const int64_t r = rr[u];
            int64_t c = -1;
            float p = 1.0f;
            if (r >= 0 && r < N) {
                const int64_t ti = staged ? (int64_t)s_ti[r] : tree_indices[r];
                const bool root = ti <= 0 || ti > n_flat;
                c = root ? st : tok[ti - 1];
                if (prb) p = root ? 1.0f : prb[ti - 1];
            }
            cand[(size_t)b * PD + i] = c;
            if (cart_prob) cart_prob[(size_t)b * PD + i] = p;
        }
    }
    for (int i = threadIdx.x + GC_PER * 256; i < PD; i += blockDim.x) {      // P*D beyond GC_PER*256 (not met by the reference's trees)
        const int64_t r = retrieve[i];
        int64_t c = -1;
        float p = 1.0f;
        if (r >= 0 && r < N) {
            const int64_t ti = tree_indices[r];
            const bool root = ti <= 0 || ti > n_flat;
            c = root ? st : tok[ti - 1];
            if (prb) p = root ? 1.0f : prb[ti - 1];
        }
        cand[(size_t)b * PD + i] = c;
        if (cart_prob) cart_prob[(size_t)b * PD + i] = p;
    }
}

template <int MAXSEL, int U>
__global__ __launch_bounds__(256) void kv_gather_kernel(void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                        const int64_t *__restrict__ slab_prev, int64_t outer, int64_t S_max,
                                                        int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                        int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                        const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len) {
    kv_gather_body<MAXSEL, U>(blockIdx.x, gridDim.x, blockIdx.y, slab_ptrs, slab_seq, slab_prev, outer, S_max, chunks_per_row, retrieve,
                              retrieve_per_seq, P, D, best, accept_len, new_len);
}

__global__ __launch_bounds__(256) void accept_copy_kernel(const uint4 *__restrict__ hidden, int G, int N, int cpr,
                                                          const int64_t *__restrict__ retrieve, int retrieve_per_seq, int P, int D,
                                                          const int64_t *__restrict__ cand, const int32_t *__restrict__ best,
                                                          const int32_t *__restrict__ accept_len, uint4 *__restrict__ out_hidden,
                                                          int64_t *__restrict__ accepted_tokens) {
    accept_copy_body(blockIdx.x, blockIdx.y, hidden, G, N, cpr, retrieve, retrieve_per_seq, P, D, cand, best, accept_len, out_hidden,
                     accepted_tokens);
}

// O9 + O10 in one launch (the two halves of the reference's update_inference_inputs): grid.y < n_slabs moves KV rows,
// the rows above it carry the accepted-hidden copy, one (sequence, cond/uncond, depth) row per workgroup.
template <int MAXSEL, int U, int MODE = 0>
__global__ __launch_bounds__(256) void update_inputs_kernel(void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                            const int64_t *__restrict__ slab_prev, int n_slabs, int64_t outer,
                                                            int64_t S_max, int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                            int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                            const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len,
                                                            const uint4 *__restrict__ hidden, int B, int G, int N, int hid_cpr,
                                                            const int64_t *__restrict__ cand, uint4 *__restrict__ out_hidden,
                                                            int64_t *__restrict__ accepted_tokens, const int32_t *__restrict__ counters,
                                                            const CommitExtras ex) {
    if ((int)blockIdx.y < n_slabs) {
        kv_gather_body<MAXSEL, U, MODE>(blockIdx.x, gridDim.x, blockIdx.y, slab_ptrs, slab_seq, slab_prev, outer, S_max, chunks_per_row, retrieve,
                                        retrieve_per_seq, P, D, best, accept_len, new_len, counters);
    } else {
        const int lin = ((int)blockIdx.y - n_slabs) * gridDim.x + blockIdx.x;
        const int per_seq = G * D;
        if (lin < B * per_seq)
            accept_copy_body(lin % per_seq, lin / per_seq, hidden, G, N, hid_cpr, retrieve, retrieve_per_seq, P, D, cand, best, accept_len,
                             out_hidden, accepted_tokens, counters, ex);
    }
    commit_release(ex);
}

// the same with KS slabs per workgroup (kv_gather_slabs): blockIdx.x < ceil(n_slabs / KS) moves KV rows, the workgroups above carry the hidden copy
template <int MAXSEL, int KS>
__global__ __launch_bounds__(256) void update_inputs_slabs_kernel(void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                                  const int64_t *__restrict__ slab_prev, int n_slabs, int64_t outer,
                                                                  int64_t S_max, int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                                  int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                                  const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len,
                                                                  const uint4 *__restrict__ hidden, int B, int G, int N, int hid_cpr,
                                                                  const int64_t *__restrict__ cand, uint4 *__restrict__ out_hidden,
                                                                  int64_t *__restrict__ accepted_tokens, const int32_t *__restrict__ counters,
                                                                  const CommitExtras ex) {
    const int n_kv = (n_slabs + KS - 1) / KS;
    if ((int)blockIdx.x < n_kv) {
        kv_gather_slabs<MAXSEL, KS, 256>((int)blockIdx.x * KS, n_slabs, slab_ptrs, slab_seq, slab_prev, outer, S_max, chunks_per_row, retrieve,
                                         retrieve_per_seq, P, D, best, accept_len, new_len, counters);
    } else {
        const int lin = (int)blockIdx.x - n_kv;
        const int per_seq = G * D;
        if (lin < B * per_seq)
            accept_copy_body(lin % per_seq, lin / per_seq, hidden, G, N, hid_cpr, retrieve, retrieve_per_seq, P, D, cand, best, accept_len,
                             out_hidden, accepted_tokens, counters, ex);
    }
    commit_release(ex);
}

constexpr int AG_THREADS = 1024;
constexpr int AG_NW = AG_THREADS / 64;

__global__ __launch_bounds__(AG_THREADS) void accept_gather_kernel(const void *__restrict__ hidden, int elem_bytes, int G, int N,
                                                                   int H, const int64_t *__restrict__ retrieve,
                                                                   int retrieve_per_seq, int P, int D,
                                                                   const int64_t *__restrict__ cand,
                                                                   const int32_t *__restrict__ best,
                                                                   const int32_t *__restrict__ accept_len,
                                                                   const float *__restrict__ sample_p, int V,
                                                                   const double *__restrict__ u, void *__restrict__ out_hidden,
                                                                   int64_t *__restrict__ accepted_tokens,
                                                                   int64_t *__restrict__ token) {
    __shared__ double s_tot[AG_NW];
    __shared__ float s_redf[2 * AG_NW];
    __shared__ int s_redi[2 * AG_NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bst = best[b];
    int n_sel = accept_len[b] + 1;
    if (n_sel > D) n_sel = D;
    const int64_t *rrow = retrieve + (retrieve_per_seq ? (size_t)b * P * D : 0) + (size_t)bst * D;
    int ph = 0;

    if (accepted_tokens && cand)
        for (int t = tid; t < D; t += AG_THREADS)
            accepted_tokens[(size_t)b * D + t] = t < n_sel ? cand[(size_t)b * P * D + (size_t)bst * D + t] : -1;

    if (hidden && out_hidden) {
        const int rowb = H * elem_bytes;  // multiple of 16 (checked on host)
        const int cpr = rowb / 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(hidden);
        uint4 *dst = reinterpret_cast<uint4 *>(out_hidden);
        const int total = G * D * cpr;
        for (int w = tid; w < total; w += AG_THREADS) {
            const int gi = w / (D * cpr);
            const int rem = w - gi * D * cpr;
            const int t = rem / cpr, c = rem - t * cpr;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t < n_sel) {
                int64_t r = rrow[t];
                if (r < 0) r += N;
                r = r < 0 ? 0 : (r >= N ? N - 1 : r);
                v = src[(((size_t)b * G + gi) * N + r) * cpr + c];
            }
            dst[(((size_t)b * G + gi) * D + t) * cpr + c] = v;
        }
    }

    if (!sample_p || !token) return;
    const float *p = sample_p + (size_t)b * V;
    if (u == nullptr) {  // greedy: argmax, lowest index on ties (torch.argmax)
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += AG_THREADS) {
            const float v = p[i];
            if (v > bv) {
                bv = v;
                bi = i;
            }
        }
        const float mx = block_max<AG_NW>(bv, s_redf, ph);
        int cnd = (bv == mx) ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnd = min(cnd, __shfl_xor(cnd, o, 64));
        int *buf = s_redi + (ph & 1) * AG_NW;
        ph ^= 1;
        if (lane == 0) buf[wave] = cnd;
        __syncthreads();
        if (tid == 0) {
            int r = buf[0];
            for (int w = 1; w < AG_NW; ++w) r = min(r, buf[w]);
            token[b] = r;
        }
        return;
    }
    // inverse CDF: thread t owns the contiguous chunk [t*cs, (t+1)*cs)
    const int cs = (V + AG_THREADS - 1) / AG_THREADS;
    const int lo = tid * cs, hi = min(V, lo + cs);
    double loc = 0.0;
    for (int i = lo; i < hi; ++i) loc += (double)p[i];
    double inc = wave_scan_incl(loc);
    if (lane == 63) s_tot[wave] = inc;
    __syncthreads();
    double woff = 0.0, total = 0.0;
#pragma unroll
    for (int w = 0; w < AG_NW; ++w) {
        const double t = s_tot[w];
        woff += (w < wave) ? t : 0.0;
        total += t;
    }
    const double excl = inc - loc + woff;
    const double tgt = u[b] * total;
    // first index whose inclusive prefix exceeds tgt (and has p > 0)
    int found = 0x7fffffff, last_pos = -1;
    double acc = excl;
    for (int i = lo; i < hi; ++i) {
        const float v = p[i];
        acc += (double)v;
        if (v > 0.0f) {
            last_pos = i;
            if (acc > tgt && found == 0x7fffffff) found = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        found = min(found, __shfl_xor(found, o, 64));
        last_pos = max(last_pos, __shfl_xor(last_pos, o, 64));
    }
    int *bf = s_redi + (ph & 1) * AG_NW;
    ph ^= 1;
    __shared__ int s_last[AG_NW];
    if (lane == 0) {
        bf[wave] = found;
        s_last[wave] = last_pos;
    }
    __syncthreads();
    if (tid == 0) {
        int f = bf[0], l = s_last[0];
        for (int w = 1; w < AG_NW; ++w) {
            f = min(f, bf[w]);
            l = max(l, s_last[w]);
        }
        token[b] = f != 0x7fffffff ? f : l;
    }
}

// The commit launch that ALSO prepares the next step (lantern_step_group.prepare_next): the blocks behind the commit's run lantern_prepare_step's
// body for step s + 1 -- candidate assembly and the likely rows (prep_dev.h), 256 threads x 4 chunks per row -- which needs step s's verdict only.
// One kernel boundary less in front of the latency-bound evaluate_posterior of the next step; the rows are produced while the KV rows move.
template <int MAXSEL, int U, bool NUCLEUS>
__global__ __launch_bounds__(256) void update_inputs_prep_kernel(void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                                 const int64_t *__restrict__ slab_prev, int n_slabs, int64_t outer,
                                                                 int64_t S_max, int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                                 int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                                 const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len,
                                                                 const uint4 *__restrict__ hidden, int B, int G, int N, int hid_cpr,
                                                                 const int64_t *__restrict__ cand, uint4 *__restrict__ out_hidden,
                                                                 int64_t *__restrict__ accepted_tokens, const int32_t *__restrict__ counters,
                                                                 const CommitExtras ex, int prep_y, const PrepArgs pa) {
    // the preparation's blocks come FIRST in the grid: they are few and latency-bound (a row's post-process is one long dependent chain), so they must be
    // dispatched before the thousands of bandwidth-bound commit blocks, not queue behind them (behind them the launch lasted commit + preparation)
    const int y = (int)blockIdx.y - prep_y;
    if (y < 0) {
        const int lin = (int)blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < pa.B * pa.n_list + pa.B) prep_rows_body<256, 4, NUCLEUS>(pa, lin);
    } else if (y < n_slabs) {
        kv_gather_body<MAXSEL, U, 0>(blockIdx.x, gridDim.x, y, slab_ptrs, slab_seq, slab_prev, outer, S_max, chunks_per_row, retrieve,
                                     retrieve_per_seq, P, D, best, accept_len, new_len, counters);
    } else {
        const int lin = (y - n_slabs) * gridDim.x + blockIdx.x;
        const int per_seq = G * D;
        if (lin < B * per_seq)
            accept_copy_body(lin % per_seq, lin / per_seq, hidden, G, N, hid_cpr, retrieve, retrieve_per_seq, P, D, cand, best, accept_len,
                             out_hidden, accepted_tokens, counters, ex);
    }
    commit_release(ex);
}

template <int MAXSEL, int KS, bool NUCLEUS>
__global__ __launch_bounds__(256) void update_inputs_slabs_prep_kernel(void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                                       const int64_t *__restrict__ slab_prev, int n_slabs, int64_t outer,
                                                                       int64_t S_max, int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                                       int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                                       const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len,
                                                                       const uint4 *__restrict__ hidden, int B, int G, int N, int hid_cpr,
                                                                       const int64_t *__restrict__ cand, uint4 *__restrict__ out_hidden,
                                                                       int64_t *__restrict__ accepted_tokens, const int32_t *__restrict__ counters,
                                                                       const CommitExtras ex, int n_prep, const PrepArgs pa) {
    const int n_kv = (n_slabs + KS - 1) / KS;
    const int x = (int)blockIdx.x - n_prep;          // (the preparation's blocks first: see update_inputs_prep_kernel)
    if (x < 0) {
        prep_rows_body<256, 4, NUCLEUS>(pa, (int)blockIdx.x);
    } else if (x < n_kv) {
        kv_gather_slabs<MAXSEL, KS, 256>(x * KS, n_slabs, slab_ptrs, slab_seq, slab_prev, outer, S_max, chunks_per_row, retrieve,
                                         retrieve_per_seq, P, D, best, accept_len, new_len, counters);
    } else {
        const int lin = x - n_kv;
        const int per_seq = G * D;
        if (lin < B * per_seq)
            accept_copy_body(lin % per_seq, lin / per_seq, hidden, G, N, hid_cpr, retrieve, retrieve_per_seq, P, D, cand, best, accept_len,
                             out_hidden, accepted_tokens, counters, ex);
    }
    commit_release(ex);
}

// ------------------------------------------------------------------------- O5
// models/drafters/cnets_lumina_mgpt.py:936-955: p_i / (1 - sum_{j<i} p_j), inf/nan -> -1, clamp [0,1]
__global__ void sample_static_kernel(const float *__restrict__ probs, const int64_t *__restrict__ idx, int R, int V, int k,
                                     float *__restrict__ out_prob) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double acc = 0.0;
    float prev_c = 0.0f;
    for (int i = 0; i < k; ++i) {
        int64_t t = idx[(size_t)r * k + i];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);
        const float p = probs[(size_t)r * V + t];
        float v = p / (1.0f - prev_c);
        if (isinf(v) || isnan(v)) v = -1.0f;
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        out_prob[(size_t)r * k + i] = v;
        acc += (double)p;
        prev_c = (float)acc;
    }
}

// O6 for the EAGLE-2 dynamic tree: candidates = cat(draft_tokens, -1)[retrieve_indices] of every sequence's own tree
// (models/ea_model_llamagen.py:676-706 with the per-call buffers of topK_genrate, cnets_llamagen.py:905-912), plus what the
// next kernels index with: the (path, depth) -> row map (a -1 wraps to the last node row, as torch's indexing does), the
// compact [P, D] retrieve rows, and the absolute position of every node (tree_position_ids + len(input_ids) + 1).
__global__ __launch_bounds__(256) void gather_candidates_dynamic_kernel(const int64_t *__restrict__ draft, const int64_t *__restrict__ retrieve,
                                                                        const int64_t *__restrict__ pos_ids, const int64_t *__restrict__ seq_len,
                                                                        int N, int P, int D, int64_t *__restrict__ cand,
                                                                        int64_t *__restrict__ retrieve_pd, int32_t *__restrict__ row_index,
                                                                        int64_t *__restrict__ pos_abs) {
    const int b = blockIdx.x;
    const int64_t *dr = draft + (size_t)b * N, *rt = retrieve + (size_t)b * N * N;
    for (int i = threadIdx.x; i < P * D; i += blockDim.x) {
        const int p = i / D, d = i - p * D;
        const int64_t r = rt[(size_t)p * N + d];
        const bool ok = r >= 0 && r < N;
        cand[(size_t)b * P * D + i] = ok ? dr[r] : -1;
        if (retrieve_pd) retrieve_pd[(size_t)b * P * D + i] = ok ? r : -1;
        if (row_index) row_index[(size_t)b * P * D + i] = ok ? (int32_t)r : N - 1;
    }
    if (pos_abs && pos_ids) {
        const int64_t base = seq_len ? seq_len[b] + 1 : 0;
        for (int n = threadIdx.x; n < N; n += blockDim.x) pos_abs[(size_t)b * N + n] = pos_ids[(size_t)b * N + n] + base;
    }
}

}  // namespace lantern

using namespace lantern;

extern "C" int lantern_gather_candidates_dynamic(const int64_t *draft_tokens, const int64_t *retrieve, const int64_t *pos_ids,
                                                 const int64_t *seq_len, int B, int N, int P, int D, int64_t *cand, int64_t *retrieve_pd,
                                                 int32_t *row_index, int64_t *pos_abs, void *stream) {
    LANTERN_CHECK_ARG(draft_tokens && retrieve && cand, "gather_candidates_dynamic: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && N > 0 && P > 0 && P <= N && D > 0 && D <= N, "gather_candidates_dynamic: bad sizes (P, D <= N)");
    if (B == 0) return LANTERN_OK;
    hipLaunchKernelGGL(gather_candidates_dynamic_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, draft_tokens, retrieve, pos_ids, seq_len, N,
                       P, D, cand, retrieve_pd, row_index, pos_abs);
    LANTERN_CHECK_LAUNCH("gather_candidates_dynamic");
    return LANTERN_OK;
}

extern "C" int lantern_gather_candidates(const int64_t *ss_token, const float *ss_prob, const int64_t *sample_token,
                                         const int64_t *tree_indices, const int64_t *retrieve, int B, int n_flat, int N, int P,
                                         int D, int64_t *tree_cand, int64_t *cand, float *cart_prob, void *stream) {
    LANTERN_CHECK_ARG(ss_token && sample_token && tree_indices && retrieve && tree_cand && cand, "gather_candidates: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && n_flat > 0 && N > 0 && P > 0 && D > 0, "gather_candidates: bad sizes");
    if (B == 0) return LANTERN_OK;
    hipLaunchKernelGGL(gather_candidates_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ss_token, ss_prob, sample_token,
                       tree_indices, retrieve, n_flat, N, P * D, tree_cand, cand, cart_prob);
    LANTERN_CHECK_LAUNCH("gather_candidates");
    return LANTERN_OK;
}

extern "C" int lantern_kv_gather(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev, int n_slabs,
                                 int elem_bytes, int64_t outer, int64_t S_max, int64_t d, const int64_t *retrieve,
                                 int retrieve_per_seq, int P, int D, const int32_t *best, const int32_t *accept_len,
                                 int64_t *new_len, void *stream) {
    LANTERN_CHECK_ARG(slab_ptrs && slab_seq && slab_prev && retrieve && best && accept_len, "kv_gather: null buffer");
    LANTERN_CHECK_ARG(n_slabs >= 0 && outer > 0 && S_max > 0 && d > 0 && P > 0 && D > 0, "kv_gather: bad sizes");
    LANTERN_CHECK_ARG((d * elem_bytes) % 16 == 0, "kv_gather: row bytes %lld must be a multiple of 16", (long long)(d * elem_bytes));
    LANTERN_CHECK_ARG(D <= KV_MAXSEL, "kv_gather: D=%d > %d", D, KV_MAXSEL);
    if (n_slabs == 0) return LANTERN_OK;
    const int cpr = (int)(d * elem_bytes / 16);
    const int64_t total = outer * cpr;
    LANTERN_CHECK_ARG(total < (1ll << 31), "kv_gather: outer * row chunks = %lld does not fit 31 bits", (long long)total);
    const int u_knob = tuning(TUNE_KV_U);
    const int U = u_knob ? u_knob : 2;
    int gx = (int)((total + 256 * U - 1) / (256 * U));
    if (gx > 4096) gx = 4096;
#define KV_LAUNCH(MS_, U_)                                                                                                               \
    LANTERN_LAUNCH((kv_gather_kernel<MS_, U_>), dim3(gx, n_slabs), dim3(256), 0, (hipStream_t)stream, slab_ptrs, slab_seq, slab_prev, \
                       outer, S_max, cpr, retrieve, retrieve_per_seq, P, D, best, accept_len, new_len)
    if (D <= 8) {
        if (U == 1) KV_LAUNCH(8, 1);
        else if (U == 4) KV_LAUNCH(8, 4);
        else KV_LAUNCH(8, 2);
    } else {
        gx = (int)((total + 255) / 256);
        if (gx > 4096) gx = 4096;
        KV_LAUNCH(16, 1);
    }
#undef KV_LAUNCH
    LANTERN_CHECK_LAUNCH("kv_gather");
    return LANTERN_OK;
}

extern "C" int lantern_accept_gather(const void *hidden, int elem_bytes, int B, int G, int N, int H, const int64_t *retrieve,
                                     int retrieve_per_seq, int P, int D, const int64_t *cand, const int32_t *best,
                                     const int32_t *accept_len, const float *sample_p, int V, const double *u,
                                     void *out_hidden, int64_t *accepted_tokens, int64_t *token, void *stream) {
    LANTERN_CHECK_ARG(retrieve && best && accept_len, "accept_gather: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && P > 0 && D > 0, "accept_gather: bad sizes");
    if (hidden) LANTERN_CHECK_ARG(out_hidden && G > 0 && N > 0 && H > 0 && (H * elem_bytes) % 16 == 0,
                                  "accept_gather: hidden row bytes must be a multiple of 16");
    if (sample_p) LANTERN_CHECK_ARG(token && V > 0, "accept_gather: sampling needs token and V");
    if (B == 0) return LANTERN_OK;
    if (!sample_p || !token) {     // copy only
        const int g = hidden ? G : 1;
        hipLaunchKernelGGL(accept_copy_kernel, dim3(g * D, B), dim3(256), 0, (hipStream_t)stream, (const uint4 *)hidden, g, N,
                           hidden ? H * elem_bytes / 16 : 0, retrieve, retrieve_per_seq, P, D, cand, best, accept_len, (uint4 *)out_hidden,
                           accepted_tokens);
        LANTERN_CHECK_LAUNCH("accept_gather");
        return LANTERN_OK;
    }
    hipLaunchKernelGGL(accept_gather_kernel, dim3(B), dim3(AG_THREADS), 0, (hipStream_t)stream, hidden, elem_bytes, G, N, H,
                       retrieve, retrieve_per_seq, P, D, cand, best, accept_len, sample_p, V, u, out_hidden, accepted_tokens, token);
    LANTERN_CHECK_LAUNCH("accept_gather");
    return LANTERN_OK;
}

extern "C" int lantern_sample_static(const float *probs, const int64_t *idx, int R, int V, int k, float *out_prob, void *stream) {
    LANTERN_CHECK_ARG(probs && idx && out_prob && R >= 0 && V > 0 && k > 0, "sample_static: bad arguments");
    if (R == 0) return LANTERN_OK;
    hipLaunchKernelGGL(sample_static_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, probs, idx, R, V, k, out_prob);
    LANTERN_CHECK_LAUNCH("sample_static");
    return LANTERN_OK;
}

namespace lantern {
// `counters` [B, 6] (evaluate_posterior's) or NULL: a sequence whose walk reported a status (counters[b][5] != 0) moves no KV row, copies no hidden
// row, lists no token and keeps its lengths -- lantern_verify_step commits a step only where the walk succeeded
int launch_update_inference_inputs(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev, int n_slabs, int elem_bytes, int64_t outer,
                                   int64_t S_max, int64_t d, const int64_t *retrieve, int retrieve_per_seq, int P, int D, const int32_t *best,
                                   const int32_t *accept_len, int64_t *new_len, const void *hidden, int hid_elem_bytes, int B, int G, int N, int H,
                                   const int64_t *cand, void *out_hidden, int64_t *accepted_tokens, const int32_t *counters, void *stream,
                                   const void *hidden_g1 = nullptr, int64_t *ids_buf = nullptr, int64_t ids_stride = 0, const int64_t *ids_len = nullptr,
                                   const int64_t *bonus = nullptr, const PrepArgs *prep = nullptr, const TurnArgs *turn = nullptr);
}

extern "C" int lantern_update_inference_inputs(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev, int n_slabs,
                                               int elem_bytes, int64_t outer, int64_t S_max, int64_t d, const int64_t *retrieve,
                                               int retrieve_per_seq, int P, int D, const int32_t *best, const int32_t *accept_len,
                                               int64_t *new_len, const void *hidden, int hid_elem_bytes, int B, int G, int N, int H,
                                               const int64_t *cand, void *out_hidden, int64_t *accepted_tokens, void *stream) {
    return lantern::launch_update_inference_inputs(slab_ptrs, slab_seq, slab_prev, n_slabs, elem_bytes, outer, S_max, d, retrieve, retrieve_per_seq, P, D, best,
                                                   accept_len, new_len, hidden, hid_elem_bytes, B, G, N, H, cand, out_hidden, accepted_tokens, nullptr, stream);
}

int lantern::launch_update_inference_inputs(void *const *slab_ptrs, const int32_t *slab_seq, const int64_t *slab_prev, int n_slabs, int elem_bytes,
                                            int64_t outer, int64_t S_max, int64_t d, const int64_t *retrieve, int retrieve_per_seq, int P, int D,
                                            const int32_t *best, const int32_t *accept_len, int64_t *new_len, const void *hidden, int hid_elem_bytes, int B,
                                            int G, int N, int H, const int64_t *cand, void *out_hidden, int64_t *accepted_tokens, const int32_t *counters,
                                            void *stream, const void *hidden_g1, int64_t *ids_buf, int64_t ids_stride, const int64_t *ids_len,
                                            const int64_t *bonus, const PrepArgs *prep, const TurnArgs *turn) {
    LANTERN_CHECK_ARG(slab_ptrs && slab_seq && slab_prev && retrieve && best && accept_len, "update_inference_inputs: null buffer");
    if (hidden_g1) LANTERN_CHECK_ARG(hidden && G == 2, "update_inference_inputs: hidden_uncond needs the conditional rows in `hidden` and hid_groups == 2");
    if (ids_buf) LANTERN_CHECK_ARG(ids_len && cand && ids_stride > 0, "update_inference_inputs: ids_buf needs ids_len, the candidates and ids_stride > 0");
    CommitExtras ex{(const uint4 *)hidden_g1, ids_buf, ids_stride, ids_len, bonus};
    // commit turn-taking: the launch's last workgroup releases the turn (the grid size of whichever form is launched below gives the target)
    auto arm_turn = [&](long long n_workgroups) {
        if (turn && turn->turn) {
            ex.turn = (unsigned long long *)turn->turn;
            ex.turn_group = turn->group;
            ex.turn_groups = turn->groups;
            ex.turn_nwg = (unsigned int)n_workgroups;
        }
    };
    if (turn && turn->turn) LANTERN_CHECK_ARG(turn->group >= 0 && turn->group < turn->groups, "update_inference_inputs: turn_group in [0, turn_groups)");
    LANTERN_CHECK_ARG(n_slabs > 0 && outer > 0 && S_max > 0 && d > 0 && P > 0 && D > 0 && B > 0, "update_inference_inputs: bad sizes");
    LANTERN_CHECK_ARG((d * elem_bytes) % 16 == 0, "update_inference_inputs: KV row bytes %lld must be a multiple of 16", (long long)(d * elem_bytes));
    LANTERN_CHECK_ARG(D <= 8, "update_inference_inputs: D=%d > 8 (use lantern_kv_gather + lantern_accept_gather)", D);
    if (hidden) LANTERN_CHECK_ARG(out_hidden && G > 0 && N > 0 && H > 0 && (H * hid_elem_bytes) % 16 == 0,
                                  "update_inference_inputs: hidden row bytes must be a multiple of 16");
    const int cpr = (int)(d * elem_bytes / 16);
    const int64_t total = outer * cpr;
    LANTERN_CHECK_ARG(total < (1ll << 31), "update_inference_inputs: outer * row chunks = %lld does not fit 31 bits", (long long)total);
    LANTERN_CHECK_ARG(S_max < (1ll << 31), "update_inference_inputs: S_max = %lld does not fit 31 bits", (long long)S_max);
    const int ks_knob = tuning(TUNE_KV_KS);          // slabs per workgroup, 0 = one workgroup tile per slab
    // Slab blocks (a workgroup walks KS whole slabs, two row groups per trip) only for slabs of at most 1024 chunks -- the mirrors' per-layer caches (32 heads x
    // 16 chunks = 512): above that a workgroup's trips are a serial chain of load / store round trips (LlamaGen-B's [24, 1, 12, S, 64] slabs, 2304 chunks:
    // 18 trips per workgroup, 22 us per commit of 16 sequences against 4.8 us on the tiled mover; round 6)
    constexpr long long KV_SLAB_BLOCK_MAX = 1024;
    const bool prep_nucleus = prep && prep->top_p >= 1e-8f && prep->top_p < 1.0f;
    const int n_prep = prep ? prep->B * prep->n_list + prep->B : 0;
    if (prep) LANTERN_CHECK_ARG(prep->W == 8192 && prep->B >= 0, "update_inference_inputs: the next step's preparation rides on the 8192-id window only");
    if (prep && n_prep > 0 && total <= KV_SLAB_BLOCK_MAX) {          // small slabs + the next step's preparation
        const int g = hidden ? G : 1;
        const int n_commit_x = (n_slabs + 3) / 4 + B * g * D;
        arm_turn((long long)n_commit_x + n_prep);
#define UISP_LAUNCH(NUC_)                                                                                                                      \
    LANTERN_LAUNCH((update_inputs_slabs_prep_kernel<8, 4, NUC_>), dim3(n_commit_x + n_prep), dim3(256), 0, (hipStream_t)stream, slab_ptrs, slab_seq, \
                   slab_prev, n_slabs, outer, S_max, cpr, retrieve, retrieve_per_seq, P, D, best, accept_len, new_len, (const uint4 *)hidden, B, g, N, \
                   hidden ? H * hid_elem_bytes / 16 : 0, cand, (uint4 *)out_hidden, accepted_tokens, counters, ex, n_prep, *prep)
        if (prep_nucleus) UISP_LAUNCH(true);
        else UISP_LAUNCH(false);
#undef UISP_LAUNCH
        LANTERN_CHECK_LAUNCH("update_inference_inputs");
        return LANTERN_OK;
    }
    if (prep && n_prep > 0) {                            // slab tiles + the next step's preparation
        int gx = (int)((total + 2 * 256 - 1) / (2 * 256));
        if (gx > 4096) gx = 4096;
        const int g = hidden ? G : 1;
        const int extra = (B * g * D + gx - 1) / gx, n_commit_y = n_slabs + extra, prep_y = (n_prep + gx - 1) / gx;
        arm_turn((long long)gx * (n_commit_y + prep_y));
#define UIP_LAUNCH(NUC_)                                                                                                                       \
    LANTERN_LAUNCH((update_inputs_prep_kernel<8, 2, NUC_>), dim3(gx, n_commit_y + prep_y), dim3(256), 0, (hipStream_t)stream, slab_ptrs, slab_seq, \
                   slab_prev, n_slabs, outer, S_max, cpr, retrieve, retrieve_per_seq, P, D, best, accept_len, new_len, (const uint4 *)hidden, B, g, N, \
                   hidden ? H * hid_elem_bytes / 16 : 0, cand, (uint4 *)out_hidden, accepted_tokens, counters, ex, prep_y, *prep)
        if (prep_nucleus) UIP_LAUNCH(true);
        else UIP_LAUNCH(false);
#undef UIP_LAUNCH
        LANTERN_CHECK_LAUNCH("update_inference_inputs");
        return LANTERN_OK;
    }
    if (ks_knob > 0 && total <= KV_SLAB_BLOCK_MAX) {
        // small slabs (the 7B geometry: 32 heads x 16 chunks): a workgroup covers whole slabs, KS of them
        const int g = hidden ? G : 1;
        const int ks = ks_knob >= 8 ? 8 : (ks_knob >= 4 ? 4 : (ks_knob >= 2 ? 2 : 1));
        const int gridx = (n_slabs + ks - 1) / ks + B * g * D;
        arm_turn(gridx);
#define UIS_LAUNCH(KS_)                                                                                                                  \
    LANTERN_LAUNCH((update_inputs_slabs_kernel<8, KS_>), dim3(gridx), dim3(256), 0, (hipStream_t)stream, slab_ptrs, slab_seq, slab_prev,  \
                   n_slabs, outer, S_max, cpr, retrieve, retrieve_per_seq, P, D, best, accept_len, new_len, (const uint4 *)hidden, B, g, N, \
                   hidden ? H * hid_elem_bytes / 16 : 0, cand, (uint4 *)out_hidden, accepted_tokens, counters, ex)
        if (ks == 8) UIS_LAUNCH(8);
        else if (ks == 4) UIS_LAUNCH(4);
        else if (ks == 2) UIS_LAUNCH(2);
        else UIS_LAUNCH(1);
#undef UIS_LAUNCH
        LANTERN_CHECK_LAUNCH("update_inference_inputs");
        return LANTERN_OK;
    }
    const int variant = tuning(TUNE_KV_VARIANT);          // 10*U + mode
    const int uu = variant / 10 ? variant / 10 : 2, mode = variant % 10;
    int gx = (int)((total + uu * 256 - 1) / (uu * 256));
    if (gx > 4096) gx = 4096;
    const int g = hidden ? G : 1;
    const int extra = (B * g * D + gx - 1) / gx;
    arm_turn((long long)gx * (n_slabs + extra));
#define UI_LAUNCH(U_, M_)                                                                                                                \
    LANTERN_LAUNCH((update_inputs_kernel<8, U_, M_>), dim3(gx, n_slabs + extra), dim3(256), 0, (hipStream_t)stream, slab_ptrs, slab_seq, \
                   slab_prev, n_slabs, outer, S_max, cpr, retrieve, retrieve_per_seq, P, D, best, accept_len, new_len,                   \
                   (const uint4 *)hidden, B, g, N, hidden ? H * hid_elem_bytes / 16 : 0, cand, (uint4 *)out_hidden, accepted_tokens, counters, ex)
    if (uu == 1 && mode == 0) UI_LAUNCH(1, 0);
    else if (uu == 4 && mode == 0) UI_LAUNCH(4, 0);
    else if (uu == 2 && mode == 1) UI_LAUNCH(2, 1);
    else if (uu == 2 && mode == 2) UI_LAUNCH(2, 2);
    else if (uu == 2 && mode == 3) UI_LAUNCH(2, 3);
    else if (uu == 4 && mode == 3) UI_LAUNCH(4, 3);
    else UI_LAUNCH(2, 0);
#undef UI_LAUNCH
    LANTERN_CHECK_LAUNCH("update_inference_inputs");
    return LANTERN_OK;
}
