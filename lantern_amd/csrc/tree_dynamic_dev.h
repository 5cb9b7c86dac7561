// tree_dynamic_dev.h -- device body of O4 (dynamic tree finalise) + O6-dynamic (candidates of that tree), shared by tree_dynamic.hip (its own
// launches) and window_kernels.hip (lantern_prepare_step for dynamic groups: the tree workgroups run beside the row workgroups).
//
// ONE 64-lane wavefront holds a sequence's tree: every tree in this path has <= 64 nodes, so a node is a lane, its ancestor set one
// uint64, depth = popcount - 1, leaves come from a ballot, retrieve rows are parent-pointer walks.  NW wavefronts compute the whole
// (cheap, latency-bound) tree redundantly, each in its own LDS slice -- no cross-wave dependency -- and share only the stores.
// Reference: models/drafters/cnets_llamagen.py:831-912; cnets_lumina_mgpt.py:1330-1393; cnets_anole.py:913-993 (O4);
// models/ea_model_llamagen.py:676-706 (O6 with the per-call buffers).
#pragma once
#include "common.h"

#ifndef TD_STAMP
#define TD_STAMP(i) do { } while (0)
#endif

namespace lantern {

// what O6-dynamic adds when the same launch assembles the candidates (lantern_tree_dynamic_candidates); cand == NULL: finalize only
struct TdCand {
    const int64_t *seq_len;
    int64_t *cand, *retrieve_pd, *pos_abs;
    int32_t *row_index;
    int P, D;
};

struct TdArgs {
    const float *scores;
    const int64_t *tokens, *parents, *sample_token;
    int n_scores, n_parents, top_k, T, sort_rows;
    int64_t *draft_tokens;
    float *mask;
    int64_t *pos_ids, *retrieve;
    int32_t *n_leaf, *max_depth;
    TdCand cd;
};

// EPL: score elements per lane (8 covers the reference's 10 + 100 * depth <= 512 scores, 32 the general case); NW: wavefronts of the workgroup
template <int EPL, int NW>
__device__ __forceinline__ void td_finalize_body(const TdArgs &ta, const int b) {
    const float *__restrict__ scores_ = ta.scores;
    const int64_t *__restrict__ tokens_ = ta.tokens, *__restrict__ parents_ = ta.parents, *__restrict__ sample_token = ta.sample_token;
    const int n_scores = ta.n_scores, n_parents = ta.n_parents, top_k = ta.top_k, T = ta.T, sort_rows = ta.sort_rows;
    int64_t *__restrict__ draft_tokens = ta.draft_tokens, *__restrict__ pos_ids = ta.pos_ids, *__restrict__ retrieve = ta.retrieve;
    float *__restrict__ mask = ta.mask;
    int32_t *__restrict__ n_leaf = ta.n_leaf, *__restrict__ max_depth = ta.max_depth;
    const TdCand &cd = ta.cd;
    (void)n_parents;
    __shared__ long long s_tok_[NW][64];
    __shared__ int s_sel_[NW][64];
    __shared__ int s_par_[NW][64];
    __shared__ int s_flag_[NW][64];
    __shared__ signed char s_rows_[NW][64][64];
    __shared__ unsigned long long s_anc_[NW][64], s_key_[NW][64];
    __shared__ int s_slot_[NW][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *const s_sel = s_sel_[wave], *const s_par = s_par_[wave], *const s_flag = s_flag_[wave], *const s_slot = s_slot_[wave];
    signed char(*const s_rows)[64] = s_rows_[wave];
    unsigned long long *const s_anc = s_anc_[wave], *const s_key = s_key_[wave];
    const int N = T + 1;
    const float *scores = scores_ + (size_t)b * n_scores;
    const int64_t *tokens = tokens_ + (size_t)b * n_scores;
    const int64_t *parents = parents_ + (size_t)b * n_parents;

    TD_STAMP(0);
    // ---- top-T by score (ties -> lower flat index), kept in ascending index order
    const int E = (n_scores + 63) / 64;  // blocked: lane owns [lane*E, lane*E+E)
    uint32_t key[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int idx = lane * E + j;
        key[j] = (j < E && idx < n_scores) ? float_key(scores[idx]) : 0u;  // 0 < key of any float
    }
    TD_STAMP(1);
    // threshold key = T-th largest: bitwise search, wave-wide counts on the scalar unit (one compare + one s_bcnt1 per element
    // slot, no cross-lane chain).  Only wave 0 searches -- four waves doing it at once queue up on the CU's scalar unit -- and
    // hands the result over through LDS (measured the same either way: ~400 cycles per bit, 13 k cycles of the kernel's 36 k).
    __shared__ uint32_t s_prefix;
    if (wave == 0) {
        uint32_t pf = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = pf | (1u << bit);
            int c = 0;
#pragma unroll
            for (int j = 0; j < EPL; ++j) c += __popcll(__ballot((j < E) && key[j] >= trial));
            if (c >= T) pf = trial;
        }
        if (lane == 0) s_prefix = pf;
    }
    __syncthreads();
    const uint32_t prefix = s_prefix;
    TD_STAMP(2);
    int c_gt = 0, c_eq = 0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
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
    for (int j = 0; j < EPL; ++j) {
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
    for (int j = 0; j < EPL; ++j)
        if (j < E && (selbits >> j) & 1u) s_sel[pos++] = lane * E + j;
    __syncthreads();

    TD_STAMP(3);
    // ---- node = lane (0 = root); parent via searchsorted over the selected flat indices
    int par = 0;
    long long tok_l = -1;
    if (lane == 0) {
        tok_l = sample_token[b];
        if (wave == 0) draft_tokens[(size_t)b * N] = tok_l;
    } else if (lane < N) {
        const int flat = s_sel[lane - 1];
        tok_l = tokens[flat];
        if (wave == 0) draft_tokens[(size_t)b * N + lane] = tok_l;
        const int64_t dp = parents[flat / top_k];
        if (dp != 0) {
            const int64_t keyv = dp - 1;
            int lo_ = 0, hi_ = T;                  // searchsorted(left) over the ascending selected indices
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_) >> 1;
                if (s_sel[mid] < keyv) lo_ = mid + 1;
                else hi_ = mid;
            }
            par = lo_ + 1;
        }
    }
    TD_STAMP(4);
    s_par[lane] = par;
    s_flag[lane] = 0;
    s_tok_[wave][lane] = tok_l;
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
    TD_STAMP(5);
    const int MD = md + 1;
    s_anc[lane] = lane < N ? anc : 0ull;
    __syncthreads();
    if (lane < N && wave == 0) pos_ids[(size_t)b * N + lane] = depth;
    {   // mask [N,N]: lanes run along the row-major output (coalesced 256-byte stores), bits from the ancestor words in LDS
        float *mb = mask + (size_t)b * N * N;
        const int dq = (64 * NW) / N, dr = (64 * NW) - dq * N;       // index step in (row, column) form
        int r = tid / N, c = tid - r * N;          // one division, then (r, c) advance with the index
        for (int idx = tid; idx < N * N; idx += 64 * NW) {
            mb[idx] = (float)((s_anc[r] >> c) & 1ull);
            r += dq, c += dr;
            if (c >= N) c -= N, ++r;
        }
    }
    TD_STAMP(6);
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
    TD_STAMP(7);
    int out_row = rid;
    if (sort_rows) {
        // rank among rows, key = entries with -1 -> T+5 (always larger than any node id).  Up to 8 columns the row is one
        // 64-bit big-endian key (one byte per column): 58 independent broadcast LDS reads instead of a compare loop per pair
        if (MD <= 8) {
            unsigned long long k64 = 0ull;
            if (leaf)
                for (int j = 0; j < 8; ++j) {
                    const int e = (j < MD && s_rows[rid][j] >= 0) ? s_rows[rid][j] : T + 5;
                    k64 = (k64 << 8) | (unsigned long long)(e & 255);
                }
            if (leaf) s_key[rid] = k64;
            __syncthreads();
            if (leaf) {
                int rank = 0;
                for (int o = 0; o < nl; ++o) {
                    const unsigned long long ko = s_key[o];
                    rank += (ko < k64) || (ko == k64 && o < rid);
                }
                out_row = rank;
            }
        } else if (leaf) {
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
    }
    TD_STAMP(8);
    if (leaf) s_slot[out_row] = rid;           // output row -> staged row
    __syncthreads();
    {   // retrieve [N,N] i64, -1 padded: coalesced 512-byte stores
        int64_t *rbase = retrieve + (size_t)b * N * N;
        const int dq = (64 * NW) / N, dr = (64 * NW) - dq * N;
        int r = tid / N, c = tid - r * N;
        for (int idx = tid; idx < N * N; idx += 64 * NW) {
            rbase[idx] = (r < nl && c < MD) ? (int64_t)s_rows[s_slot[r]][c] : -1;
            r += dq, c += dr;
            if (c >= N) c -= N, ++r;
        }
    }
    TD_STAMP(9);
    if (tid == 0) {
        n_leaf[b] = nl;
        max_depth[b] = MD;
    }
    // ---- O6, dynamic (ea_model_llamagen.py:676-706 with this tree; gather_candidates_dynamic_kernel's arithmetic): the candidates by
    // (path, depth), the compact retrieve rows, the row map (a -1 wraps to the last node's row) and every node's absolute position
    if (cd.cand) {
        const int PD = cd.P * cd.D;
        for (int i = tid; i < PD; i += 64 * NW) {
            const int p = i / cd.D, d = i - p * cd.D;
            const int r = (p < nl && d < MD) ? (int)s_rows[s_slot[p]][d] : -1;
            const bool ok = r >= 0 && r < N;
            cd.cand[(size_t)b * PD + i] = ok ? (int64_t)s_tok_[wave][r] : -1;
            if (cd.retrieve_pd) cd.retrieve_pd[(size_t)b * PD + i] = ok ? r : -1;
            if (cd.row_index) cd.row_index[(size_t)b * PD + i] = ok ? r : N - 1;
        }
        if (cd.pos_abs && lane < N && wave == 0) cd.pos_abs[(size_t)b * N + lane] = (int64_t)depth + (cd.seq_len ? cd.seq_len[b] + 1 : 0);
    }
}

}  // namespace lantern
