// gather_dev.h -- device bodies of the O9 KV-row move and the O10 accepted-hidden copy (launched by gather_ops.hip).
#pragma once
#include "common.h"

namespace lantern {

// ------------------------------------------------------------------------- O9
// models/ea_model_lumina_mgpt.py:741-746,763-767; models/drafters/kv_cache.py:38-50.
// grid.x = tiles of `outer` rows, grid.y = slab.  A thread owns one 16-byte column chunk of
// one (layer,batch,head) row-group: it loads the <= MAXSEL selected chunks into registers,
// then stores them at prev..prev+a, so the in-place move cannot race with itself.
constexpr int KV_MAXSEL = 16;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// MAXSEL = rows a step can accept (D); U = row groups a thread moves per trip, all loads of the U groups in flight
// before the first store (few, fat workgroups: a thread that moves one 16-byte chunk of ~1.3 rows has too little in
// flight to cover HBM latency, and 12k tiny workgroups per launch are dispatch-bound).
// the move with the verdict (best path, rows to keep) handed in by the caller
template <int MAXSEL, int U, int MODE = 0>
__device__ __forceinline__ void kv_gather_rows(int bx, int nbx, int s, int seq, int bst, int n_sel, void *const *__restrict__ slab_ptrs,
                                               const int64_t *__restrict__ slab_prev, int64_t outer, int64_t S_max, int chunks_per_row,
                                               const int64_t *__restrict__ retrieve, int retrieve_per_seq, int P, int D,
                                               int64_t *__restrict__ new_len) {
    const int64_t prev = slab_prev[s];
    if (n_sel > MAXSEL) n_sel = MAXSEL;
    const int64_t *rrow = retrieve + (retrieve_per_seq ? (size_t)seq * P * D : 0) + (size_t)bst * D;
    if (bx == 0 && threadIdx.x == 0 && new_len) new_len[s] = prev + n_sel;

    // rows already in place (tree node t sits at position prev + t: always the root, and every accepted first child) are
    // not touched -- copying a row onto itself is the identity, so the result equals the reference's index_select + copy_
    unsigned move = 0u;
    int64_t srcrow[MAXSEL];
#pragma unroll
    for (int t = 0; t < MAXSEL; ++t) {
        srcrow[t] = 0;
        if (t < n_sel) {
            const int64_t r = rrow[t];
            if (r != t && prev + t < S_max) move |= 1u << t;
            const int64_t src = r + prev;
            srcrow[t] = src < 0 ? 0 : (src >= S_max ? S_max - 1 : src);
        }
    }
    if (move == 0u) return;
    // the slab address comes out of a pointer table: tell the compiler it is global memory (global_load/store, not flat)
    typedef __attribute__((address_space(1))) u32x4_t gvec_t;
    gvec_t *base = (gvec_t *)(uintptr_t)slab_ptrs[s];
    const unsigned total = (unsigned)(outer * chunks_per_row), cpr = (unsigned)chunks_per_row;
    const unsigned stride = (unsigned)nbx * blockDim.x;
    for (unsigned w0 = (unsigned)bx * blockDim.x + threadIdx.x; w0 < total; w0 += U * stride) {
        u32x4_t v[U][MAXSEL];
        gvec_t *rowbase[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned w = w0 + u * stride;
            const unsigned o = w / cpr, c = w - o * cpr;
            rowbase[u] = base + (size_t)o * S_max * cpr + c;
            if (w < total) {
#pragma unroll
                for (int t = 0; t < MAXSEL; ++t)
                    if ((move >> t) & 1u) v[u][t] = (MODE & 1) ? rowbase[u][srcrow[t] * cpr] : __builtin_nontemporal_load(&rowbase[u][srcrow[t] * cpr]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned w = w0 + u * stride;
            if (w < total) {
#pragma unroll
                for (int t = 0; t < MAXSEL; ++t)
                    if ((move >> t) & 1u) {
                        if (MODE & 2) rowbase[u][(prev + t) * cpr] = v[u][t];
                        else __builtin_nontemporal_store(v[u][t], &rowbase[u][(prev + t) * cpr]);
                    }
            }
        }
    }
}

template <int MAXSEL, int U, int MODE = 0>
__device__ __forceinline__ void kv_gather_body(int bx, int nbx, int s, void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                        const int64_t *__restrict__ slab_prev, int64_t outer, int64_t S_max,
                                                        int chunks_per_row, const int64_t *__restrict__ retrieve,
                                                        int retrieve_per_seq, int P, int D, const int32_t *__restrict__ best,
                                                        const int32_t *__restrict__ accept_len, int64_t *__restrict__ new_len,
                                                        const int32_t *__restrict__ counters = nullptr) {
    const int seq = slab_seq[s];
    const int bst = best[seq];
    int n_sel = accept_len[seq] + 1;
    if (n_sel > D) n_sel = D;
    if (counters && counters[(size_t)seq * 6 + 5] != 0) n_sel = 0;          // a walk that reported a status commits nothing (the caller retries the step)
    kv_gather_rows<MAXSEL, U, MODE>(bx, nbx, s, seq, bst, n_sel, slab_ptrs, slab_prev, outer, S_max, chunks_per_row, retrieve, retrieve_per_seq, P, D, new_len);
}

// The same move with KS consecutive slabs per workgroup (small slabs: one workgroup covers a whole slab).  Everything a slab's move depends on
// -- its sequence, previous length, base address, then the sequence's verdict, then the path's rows -- is fetched for all KS slabs at once, so the
// three dependent round trips are paid once per workgroup instead of once per slab, and a launch has KS times fewer workgroups: the commits of
// several groups in flight then fit the chip in one or two rounds (per-slab workgroups: 8 064 of them for 63 sequences of the 7B geometry,
// six rounds of ~5 us).
template <int MAXSEL, int KS, int NT>
__device__ __forceinline__ void kv_gather_slabs(int s0, int n_slabs, void *const *__restrict__ slab_ptrs, const int32_t *__restrict__ slab_seq,
                                                const int64_t *__restrict__ slab_prev, int64_t outer, int64_t S_max, int chunks_per_row,
                                                const int64_t *__restrict__ retrieve, int retrieve_per_seq, int P, int D,
                                                const int32_t *__restrict__ best, const int32_t *__restrict__ accept_len,
                                                int64_t *__restrict__ new_len, const int32_t *__restrict__ counters) {
    static_assert(KS * MAXSEL <= 64, "the slabs' path rows are resolved by the lanes of one wave");
    typedef __attribute__((address_space(1))) u32x4_t gvec_t;
    __shared__ int s_src[KS][MAXSEL];
    __shared__ unsigned s_move[KS];
    __shared__ long long s_prev[KS];
    __shared__ unsigned long long s_ptr[KS];
    const int tid = threadIdx.x;
    if (tid < 64) {          // lane (i, t): path row t of slab i -- the dependent loads of all slabs side by side
        const int i = tid / MAXSEL, t = tid % MAXSEL;
        const bool valid = i < KS && s0 + i < n_slabs;
        const int s = valid ? s0 + i : n_slabs - 1;
        const int seq = slab_seq[s];
        const long long prev = slab_prev[s];
        const int bst = best[seq];
        int n = accept_len[seq] + 1;
        if (n > D) n = D;
        if (counters && counters[(size_t)seq * 6 + 5] != 0) n = 0;          // a walk that reported a status commits nothing
        if (n > MAXSEL) n = MAXSEL;
        const bool live = valid && t < n;
        const long long r = live ? retrieve[(retrieve_per_seq ? (size_t)seq * P * D : 0) + (size_t)bst * D + t] : (long long)t;
        // rows already in place (tree node t at position prev + t: the root, every accepted first child) are not touched
        const bool mv = live && r != t && prev + t < S_max;
        const unsigned long long mask = __ballot(mv);
        if (i < KS) {
            const long long sr = r + prev;
            s_src[i][t] = (int)(sr < 0 ? 0 : (sr >= S_max ? S_max - 1 : sr));
            if (t == 0) {
                s_move[i] = (unsigned)((mask >> (i * MAXSEL)) & ((1u << MAXSEL) - 1u));
                s_prev[i] = prev;
                s_ptr[i] = (unsigned long long)(uintptr_t)slab_ptrs[s];
                if (valid && new_len) new_len[s] = prev + n;
            }
        }
    }
    __syncthreads();
    const unsigned total = (unsigned)(outer * chunks_per_row), cpr = (unsigned)chunks_per_row;
#pragma unroll 1
    for (int i = 0; i < KS; ++i) {
        const unsigned move = __builtin_amdgcn_readfirstlane(s_move[i]);
        if (move == 0u) continue;
        gvec_t *base = (gvec_t *)(uintptr_t)s_ptr[i];
        const long long prev = s_prev[i];
        int src[MAXSEL];
#pragma unroll
        for (int t = 0; t < MAXSEL; ++t) src[t] = s_src[i][t];
        for (unsigned w0 = tid; w0 < total; w0 += 2 * NT) {
            u32x4_t v[2][MAXSEL];
            gvec_t *rowbase[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned w = w0 + u * NT;
                const unsigned o = w / cpr, c = w - o * cpr;
                rowbase[u] = base + (size_t)o * S_max * cpr + c;
                if (w < total) {
#pragma unroll
                    for (int t = 0; t < MAXSEL; ++t)
                        if ((move >> t) & 1u) v[u][t] = __builtin_nontemporal_load(&rowbase[u][(size_t)src[t] * cpr]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned w = w0 + u * NT;
                if (w < total) {
#pragma unroll
                    for (int t = 0; t < MAXSEL; ++t)
                        if ((move >> t) & 1u) __builtin_nontemporal_store(v[u][t], &rowbase[u][(size_t)(prev + t) * cpr]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------ O10
// models/ea_model_lumina_mgpt.py:748-750,773-785.
// Copy-only form (no bonus-token draw: the windowed evaluate_posterior draws it): one workgroup per (sequence, cond/uncond,
// depth) row so the 2*D rows of every sequence move in parallel across the chip instead of through one CU.
// Optional extras of the commit (lantern_step_group: hidden_uncond, ids_buf / ids_stride / ids_len, the bonus token): all NULL = off.
struct CommitExtras {
    const uint4 *hidden_g1;          // the second group's rows [B, N, H] as their own pointer (then `hidden` is [B, N, H] too)
    int64_t *ids_buf;                // [B, ids_stride]: the accepted tokens (+ the bonus token) appended in place
    int64_t ids_stride;
    const int64_t *ids_len;          // [B] tokens already in ids_buf
    const int64_t *bonus;            // [B] or NULL
    // commit turn-taking (lantern_step_group.turn; layout in include/lantern_hip.h): the launch's last workgroup releases the turn.  "Last" is found
    // in two levels -- 32 first-level counters per group, one 128-byte line each, then one second-level counter -- because thousands of workgroups
    // adding to ONE word serialise at the memory side (measured: +150 us per step with a single counter per group)
    unsigned long long *turn = nullptr;
    int turn_group = 0, turn_groups = 0;
    unsigned int turn_nwg = 0;
};

constexpr int TURN_LINE = 16, TURN_SLOTS = 32;          // int64 words per 128-byte line; first-level counters per group

// host side of the same (lantern_step_group.turn / turn_group / turn_groups)
struct TurnArgs {
    int64_t *turn;
    int group, groups;
};

// last statement of every commit kernel's workgroup
__device__ __forceinline__ void commit_release(const CommitExtras &ex) {
    if (!ex.turn) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (every counter is zero when a launch starts: the workgroup that fills one puts it back to zero -- no other workgroup of this launch touches it
        // again, and the group's next commit launch is ordered behind this one on its stream -- so launches of different grid sizes can follow each other)
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, slot = lin & (TURN_SLOTS - 1);
        const unsigned long long n_slot = (ex.turn_nwg - slot + TURN_SLOTS - 1) / TURN_SLOTS;          // workgroups of this launch that count in `slot`
        unsigned long long *c1 = ex.turn + (size_t)TURN_LINE * (1 + ex.turn_groups + ex.turn_group * TURN_SLOTS + slot);
        if (atomicAdd(c1, 1ull) + 1 == n_slot) {
            atomicExch(c1, 0ull);
            const unsigned long long used = ex.turn_nwg < (unsigned)TURN_SLOTS ? ex.turn_nwg : TURN_SLOTS;
            unsigned long long *c2 = ex.turn + (size_t)TURN_LINE * (1 + ex.turn_group);
            if (atomicAdd(c2, 1ull) + 1 == used) {
                atomicExch(c2, 0ull);
                atomicAdd(ex.turn, 1ull);
            }
        }
    }
}

__device__ __forceinline__ void accept_copy_row(int bx, int b, int bst, int n_sel, const uint4 *__restrict__ hidden, int G, int N, int cpr,
                                                const int64_t *__restrict__ retrieve, int retrieve_per_seq, int P, int D,
                                                const int64_t *__restrict__ cand, uint4 *__restrict__ out_hidden,
                                                int64_t *__restrict__ accepted_tokens, const CommitExtras ex = CommitExtras{}) {
    const int gi = bx / D, t = bx % D, tid = threadIdx.x;
    if (bx == 0 && accepted_tokens && cand && tid < D)
        accepted_tokens[(size_t)b * D + tid] = tid < n_sel ? cand[(size_t)b * P * D + (size_t)bst * D + tid] : -1;
    if (bx == 0 && ex.ids_buf && ex.ids_len && cand && n_sel > 0 && tid <= n_sel && tid <= D) {
        // input_ids = cat(input_ids, accepted tokens) in place, the bonus token behind them (never beyond the buffer)
        const int64_t at = ex.ids_len[b] + tid;
        if (at >= 0 && at < ex.ids_stride) {
            if (tid < n_sel) ex.ids_buf[(size_t)b * ex.ids_stride + at] = cand[(size_t)b * P * D + (size_t)bst * D + tid];
            else if (ex.bonus) ex.ids_buf[(size_t)b * ex.ids_stride + at] = ex.bonus[b];
        }
    }
    if (!hidden || !out_hidden) return;
    uint4 *dst = out_hidden + (((size_t)b * G + gi) * D + t) * cpr;
    if (t < n_sel) {
        int64_t r = retrieve[(retrieve_per_seq ? (size_t)b * P * D : 0) + (size_t)bst * D + t];
        if (r < 0) r += N;
        r = r < 0 ? 0 : (r >= N ? N - 1 : r);
        const uint4 *src = ex.hidden_g1 ? ((gi == 0 ? hidden : ex.hidden_g1) + ((size_t)b * N + r) * cpr) : hidden + (((size_t)b * G + gi) * N + r) * cpr;
        for (int c = tid; c < cpr; c += blockDim.x) dst[c] = src[c];
    } else {
        for (int c = tid; c < cpr; c += blockDim.x) dst[c] = make_uint4(0, 0, 0, 0);
    }
}

__device__ __forceinline__ void accept_copy_body(int bx, int b, const uint4 *__restrict__ hidden, int G, int N, int cpr,
                                                 const int64_t *__restrict__ retrieve, int retrieve_per_seq, int P, int D,
                                                 const int64_t *__restrict__ cand, const int32_t *__restrict__ best,
                                                 const int32_t *__restrict__ accept_len, uint4 *__restrict__ out_hidden,
                                                 int64_t *__restrict__ accepted_tokens, const int32_t *__restrict__ counters = nullptr,
                                                 const CommitExtras ex = CommitExtras{}) {
    int n_sel = accept_len[b] + 1;
    if (n_sel > D) n_sel = D;
    if (counters && counters[(size_t)b * 6 + 5] != 0) n_sel = 0;
    accept_copy_row(bx, b, best[b], n_sel, hidden, G, N, cpr, retrieve, retrieve_per_seq, P, D, cand, out_hidden, accepted_tokens, ex);
}

}  // namespace lantern
