// prep_dev.h -- the row post-process of one tree row on a workgroup (cfg_window_bf16_row: CFG combination, top-k threshold, softmax, window store) and
// the candidate assembly + likely rows of lantern_prepare_step as a device body (prep_rows_body), shared by window_kernels.hip (prep_rows_kernel,
// cfg_window_bf16_kernel) and gather_ops.hip (the commit launch that ALSO prepares the next step: update_inputs_prep_kernel).  Moved verbatim out of
// window_kernels.hip.
//
// Reference: models/ea_model_lumina_mgpt.py:525-554 (generate_candidates), :597-607 with :45-112 (tree_decoding post-process).
#pragma once
#include "common.h"
#include "window_dev.h"

namespace lantern {

// Workgroup -> row map of the O7 kernels.  Consecutive workgroup ids land on consecutive XCDs (8 of them, each with its
// own L2); evaluate_posterior's workgroup b (one per sequence) lands on XCD b % 8.  With rows grouped by sequence
// (rows_per_seq > 0) the rows of sequence b are therefore produced by workgroups whose id is congruent to b modulo 8, so that
// O8 finds them in ITS XCD's L2 instead of fetching them across the fabric.  Placement is only a speed hint: any map that
// is a permutation of the rows is correct.
__device__ __forceinline__ int o7_row_of_block(int x, int rows, int rows_per_seq) {
    if (rows_per_seq <= 0) return x;
    const unsigned rps = (unsigned)rows_per_seq;                             // (unsigned divisions: half the scalar instructions of the signed ones)
    const unsigned n_seq = (unsigned)rows / rps, full = n_seq & ~7u;         // sequences that fill whole groups of 8
    const unsigned n_full_rows = full * rps;
    if ((unsigned)x >= n_full_rows) return x;                                // the ragged tail keeps the identity map
    const unsigned xcd = (unsigned)x & 7u, idx = (unsigned)x >> 3;           // idx-th workgroup of this XCD
    const unsigned q = idx / rps;
    return (int)((xcd + 8u * q) * rps + (idx - q * rps));
}

// ---- O7 windowed, bf16 logits, 16-byte loads.  Thread t owns E8 chunks of 8 consecutive window ids (chunk index
// t + it*NT), so cond and uncond arrive as one global_load_dwordx4 each per chunk (the window start need only be
// 4-aligned: the loads are then 8-byte aligned, which gfx950 global loads accept).  The first radix pass -- sign + 7 exponent bits, where a logit row
// concentrates in a handful of bins -- uses a 16-way replicated LDS histogram (copy = lane % 16: at most 4 lanes of an
// instruction on one word, window_dev.h); the second pass (7 mantissa bits + 1 exponent bit inside the chosen bin) is spread
// out by nature and uses a single copy.
// One row of the windowed O7 on a workgroup: CFG combination, top-k threshold, softmax, window store (the body of cfg_window_bf16_kernel,
// shared with the merged launch below).  `cls`: 0 = grid row, 1 = forced newline, 2 = forced end of image.
// 16 bytes stored / loaded as two relaxed agent-scope 64-bit atomics (global_store / global_load ... sc1: through to the device's coherence point, past the
// L2 of the XCD that runs the wave): what a row needs when a workgroup of the SAME launch, possibly on another XCD, reads it behind a flag -- without the L2
// write-back / invalidate of an agent-scope fence (measured: the fused chain launch 64 us with fences against 32 us for the walk alone)
__device__ __forceinline__ void store_f4_agent(float *p, const float4 v) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    __hip_atomic_store(q, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_f4_agent(const float *p) {
    unsigned long long *q = const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(p));
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}

template <int NT, int E8, bool FULL, bool NUCLEUS = false, bool COH = false>
__device__ __forceinline__ void cfg_window_bf16_row(int row, int cls, const uint16_t *__restrict__ cond, const uint16_t *__restrict__ uncond, int V,
                                                    float cfg, int model, int img_lo, int img_hi, int newline_id, int eos_id, int top_k, int win_lo,
                                                    int W, float *__restrict__ out_win, int32_t *__restrict__ row_hot, int out_kind, int *s_hist,
                                                    float *s_redf, double *s_redd, float top_p = 1.0f, int *s_redi = nullptr) {
    const int tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    if (cls != 0) {
        if (tid == 0) row_hot[row] = cls == 2 ? eos_id : newline_id;   // one-hot row: its window is never read
        return;
    }
    if (tid == 0) row_hot[row] = -1;
    const bool lumina = model == LANTERN_MODEL_LUMINA;
    const bool masked = model != LANTERN_MODEL_PLAIN;
    const float fill = lumina ? NEG_INF : __uint_as_float(0xff7f0000u);
    const bool need_mask = masked && (win_lo < img_lo || win_lo + W > img_hi);     // window inside the image range: nothing to mask
    const int e_base = win_lo;                            // first id of chunk 0
    const uint16_t *crow = cond + (size_t)row * V + e_base;
    const uint16_t *urow = uncond ? uncond + (size_t)row * V + e_base : nullptr;
    float *out = out_win + (size_t)row * W;
    float4 r[2 * E8];
    Bf16x8 cb[E8], ub[E8];
    // every load of the row in flight before the first use: the unconditional rows under ONE wave-uniform branch (as a per-chunk select with the
    // conditional chunk as its fallback the compiler waited for each conditional chunk -- vmcnt(0) -- and then fetched the unconditional one dword by
    // dword under four more branches: four serial HBM round trips per row and 16 + 4 load instructions instead of 8)
#pragma unroll
    for (int it = 0; it < E8; ++it) {
        const int ch = tid + it * NT;
        const bool in = FULL || ch * 8 < W;       // FULL: W == 8 * NT * E8, every chunk is inside the window
        cb[it] = in ? *reinterpret_cast<const Bf16x8 *>(crow + ch * 8) : Bf16x8{make_uint2(0, 0), make_uint2(0, 0)};
    }
    if (urow) {
#pragma unroll
        for (int it = 0; it < E8; ++it) {
            const int ch = tid + it * NT;
            const bool in = FULL || ch * 8 < W;
            ub[it] = in ? *reinterpret_cast<const Bf16x8 *>(urow + ch * 8) : Bf16x8{make_uint2(0, 0), make_uint2(0, 0)};
        }
    } else {
#pragma unroll
        for (int it = 0; it < E8; ++it) ub[it] = Bf16x8{make_uint2(0, 0), make_uint2(0, 0)};
    }
#pragma unroll
    for (int it = 0; it < E8; ++it) {
        const int e0 = e_base + (tid + it * NT) * 8;
        const bool in_chunk = FULL || (tid + it * NT) * 8 < W;
        const uint32_t cw[4] = {cb[it].a.x, cb[it].a.y, cb[it].b.x, cb[it].b.y};
        const uint32_t uw[4] = {ub[it].a.x, ub[it].a.y, ub[it].b.x, ub[it].b.y};
        float o[8];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {      // two ids per packed word
            f32x2_t t2;
            if (uncond) {
                t2 = cfg_mix_bf16x2(cw[q2], uw[q2], cfg);
            } else {
                t2.x = __uint_as_float(cw[q2] << 16);
                t2.y = __uint_as_float(cw[q2] & 0xffff0000u);
            }
            o[2 * q2] = t2.x;
            o[2 * q2 + 1] = t2.y;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float t = o[q];
            const int e = e0 + q;
            if (need_mask) t = (e < img_lo || e >= img_hi) ? fill : t;
            o[q] = in_chunk ? t : NEG_INF;      // chunks are whole: W % 8 == 0
        }
        r[2 * it] = make_float4(o[0], o[1], o[2], o[3]);
        r[2 * it + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    if constexpr (NUCLEUS) {          // TopPLogitsWarper in front of the top-k (the mass bins in the front of the histogram buffer)
        if (top_p >= 1e-8f && top_p < 1.0f && s_redi) {
            int php = 0;
            top_p_tile<NT, 2 * E8, true>(r, top_p, reinterpret_cast<double *>(s_hist), s_redf, s_redd, s_redi, php);
        }
    }
    if (top_k > 0 && top_k < V) {
        // k-th largest of the FULL row = k-th largest of the window whenever >= k window entries beat the fill
        // value; otherwise the threshold is the fill value (or lower) and nothing inside the window is removed.
        // (ids outside the window sit in r as -inf: they never count.)
        const float thr = (top_k <= W) ? kth_largest_hist_bf16<NT, 2 * E8>(r, top_k, s_hist) : NEG_INF;
#pragma unroll
        for (int it = 0; it < 2 * E8; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x;
            r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z;
            r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    if (out_kind == LANTERN_ROWS_PROBS) {
        int ph = 0;
        softmax_tile<NT, 2 * E8>(r, s_redf, s_redd, ph);
    }
#pragma unroll
    for (int it = 0; it < E8; ++it) {
        const int w0 = e_base + (tid + it * NT) * 8 - win_lo;     // window index of the chunk's first id (multiple of 4)
        if constexpr (COH) {          // (read by another workgroup of this launch)
            if (FULL || (w0 >= 0 && w0 < W)) store_f4_agent(out + w0, r[2 * it]);
            if (FULL || (w0 + 4 >= 0 && w0 + 4 < W)) store_f4_agent(out + w0 + 4, r[2 * it + 1]);
        } else {
            if (FULL || (w0 >= 0 && w0 < W)) *reinterpret_cast<float4 *>(out + w0) = r[2 * it];
            if (FULL || (w0 + 4 >= 0 && w0 + 4 < W)) *reinterpret_cast<float4 *>(out + w0 + 4) = r[2 * it + 1];
        }
    }
}

__device__ __forceinline__ int lumina_row_class(int64_t pos, int64_t pos_base, int w_latent, int h_latent) {
    const int64_t n1 = pos - pos_base + 1;
    if (n1 == ((int64_t)w_latent + 1) * h_latent + 1) return 2;
    // (a 64-bit modulo is ~150 scalar instructions that every wave of the workgroup repeats: the 32-bit form whenever it applies)
    if (n1 >= 0 && n1 < (1ll << 31) && w_latent >= 0 && w_latent < (1 << 30)) return ((uint32_t)n1 % (uint32_t)(w_latent + 1)) == 0 ? 1 : 0;
    return py_mod64(n1, (int64_t)w_latent + 1) == 0 ? 1 : 0;
}

// O6 + O7 of a verify step in ONE launch: workgroups [0, B * n_list) post-process the LISTED rows of every sequence (the nodes the
// walk is most likely to visit -- the root always; LANTERN_ROWS_RAW_BF16 handles the others on demand inside evaluate_posterior),
// workgroups [B * n_list, B * n_list + B) assemble the candidates (generate_candidates, ea_model_lumina_mgpt.py:525-554).  One
// kernel boundary instead of two in front of the latency-bound evaluate_posterior.
struct PrepArgs {
    const uint16_t *cond, *uncond;
    int V;
    float cfg;
    const int64_t *pos_ids;
    int64_t pos_base;
    int w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k;
    const int64_t *seq_len;
    int rows_per_seq, win_lo, W;
    float *out_win;
    int32_t *row_hot;
    const int32_t *node_list;
    int n_list, B;
    const int64_t *ss_token;
    const float *ss_prob;
    const int64_t *sample_token, *tree_indices, *retrieve;
    int n_flat, N, PD;
    int64_t *tree_cand, *cand;
    float *cart_prob;
    float top_p;
    // the launch that prepares step s + 1 inside step s's commit (lantern_step_group.prepare_next): seq_len still holds step s's lengths; the walk's
    // verdict says what the commit running beside these blocks is adding (a walk that reported a status adds nothing)
    const int32_t *len_alen, *len_cnt;
};

// `bx`: the block's index among the B * n_list row blocks followed by the B candidate-assembly blocks
template <int NT, int E8, bool NUCLEUS = false>
__device__ __forceinline__ void prep_rows_body(const PrepArgs &a, const int bx) {
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[32];
    __shared__ double s_redd[32];
    __shared__ int s_redi[32];
    const int n_rows = a.B * a.n_list;
    if (bx < n_rows) {
        const int x = o7_row_of_block(bx, n_rows, a.n_list);            // rows of sequence b on XCD b % 8, where its chain runs
        const int b = x / a.n_list, node = a.node_list[x % a.n_list];
        const int row = b * a.rows_per_seq + node;
        // (w_latent == 0: a model without grammar rows -- Anole; the window is its image-token range, so no id needs the model's mask)
        int64_t len_b = a.w_latent > 0 ? a.seq_len[b] : 0;
        if (a.w_latent > 0 && a.len_alen) len_b += (a.len_cnt && a.len_cnt[(size_t)b * 6 + 5] != 0) ? 0 : (int64_t)a.len_alen[b] + 1;
        const int cls = a.w_latent > 0 ? lumina_row_class(a.pos_ids[node] + len_b, a.pos_base, a.w_latent, a.h_latent) : 0;
        cfg_window_bf16_row<NT, E8, true, NUCLEUS>(row, cls, a.cond, a.uncond, a.V, a.cfg, LANTERN_MODEL_LUMINA, a.img_lo, a.img_hi, a.newline_id, a.eos_id,
                                                   a.top_k, a.win_lo, a.W, a.out_win, a.row_hot, LANTERN_ROWS_PROBS, s_hist, s_redf, s_redd, a.top_p, s_redi);
        return;
    }
    // ---- candidate assembly of sequence b (same arithmetic as gather_candidates_kernel)
    const int b = bx - n_rows, N = a.N, PD = a.PD, n_flat = a.n_flat;
    const int64_t *tok = a.ss_token + (size_t)b * n_flat;
    const float *prb = a.ss_prob ? a.ss_prob + (size_t)b * n_flat : nullptr;
    const int64_t st = a.sample_token[b];
    for (int n = threadIdx.x; n < N; n += NT) {
        const int64_t ti = a.tree_indices[n];
        a.tree_cand[(size_t)b * N + n] = (ti <= 0 || ti > n_flat) ? st : tok[ti - 1];
    }
    for (int i = threadIdx.x; i < PD; i += NT) {
        const int64_t r = a.retrieve[i];
        int64_t c = -1;
        float p = 1.0f;
        if (r >= 0 && r < N) {
            const int64_t ti = a.tree_indices[r];
            const bool root = ti <= 0 || ti > n_flat;
            c = root ? st : tok[ti - 1];
            if (prb) p = root ? 1.0f : prb[ti - 1];
        }
        a.cand[(size_t)b * PD + i] = c;
        if (a.cart_prob) a.cart_prob[(size_t)b * PD + i] = p;
    }
}

}  // namespace lantern
