// draft_static.hip -- the static-tree (EAGLE v1 / LANTERN++) drafter's head stage on the device: head window GEMM + CFG -> the model's
// processors -> softmax -> the row's distribution (the verify side's `original_prob` row) -> `n_draw` draws WITHOUT replacement + their conditional
// probabilities, and the next depth's inputs gathered through the static tree's index tables.  What lantern_head_expand is to the EAGLE-2 loop.
//
// Reference: Model.sample, models/drafters/cnets_lumina_mgpt.py:936-955 (softmax -> torch.multinomial(k, replacement=False) -> p_i / (1 - sum_{j<i} p_j));
// the static loop bodies cnets_lumina_mgpt.py:1245-1328 (topK_generate, tree_type "static"), cnets_llamagen.py:944-1023 / cnets_anole.py:1056-1171
// (topK_genrate_v1); the tables behind `tree_indices[i]` / `repeat_nums[i]`: models/drafters/utils_c.py:100-179.
//
// RNG contract: torch.multinomial's device RNG is not reproducible across devices, so the draws come from INJECTED uniforms --
// draw j of a row is the inverse CDF (token-id order, f64 running sum: lo_sample_inverse_cdf) of the row's distribution with the j tokens already
// drawn removed, at uniform u[row][j].  Successive inverse-CDF draws without replacement have exactly the distribution of torch.multinomial(...,
// replacement=False) (both are Plackett-Luce); `draw_idx` instead of `draw_u` takes the indices as given (tests, recorded runs).
#include "common.h"
#include "window_dev.h"

namespace lantern {
int launch_linear_rows_cfg_streamk(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, float cfg, void *win, int packed,
                                   void *workspace, size_t workspace_bytes, hipStream_t st);

// SW_C8 chunks of 8 ids per thread.  Round 5: 1024 threads x 1 chunk (W <= 8192; x 2: LlamaGen's 16384) -- one row per workgroup and one to four rows per
// launch, so the kernel is a latency chain: a quarter of the per-thread work in the select / softmax / store phases (21.5 -> see EXPERIMENTS.md).
constexpr int SW_MAX_DRAW = 16;

// One workgroup per drafter row.  `win`: the CFG-combined bf16 window logits of the row ([n, W], the head GEMM's epilogue output).
template <int NT, int SW_C8>
__global__ __launch_bounds__(NT) void sample_window_kernel(const uint16_t *__restrict__ win, int W, int win_lo, int V, int model,
                                                           const int64_t *__restrict__ pos_ids, int64_t pos_base, int w_latent, int h_latent,
                                                           int newline_id, int eos_id, int top_k_filter, int n_draw,
                                                           const double *__restrict__ draw_u, const int64_t *__restrict__ draw_idx,
                                                           float *__restrict__ probs_out, int64_t *__restrict__ ss_token, float *__restrict__ ss_prob) {
    constexpr int NV4 = 2 * SW_C8;
    extern __shared__ float4 sw_dyn[];
    float *g = reinterpret_cast<float *>(sw_dyn);          // [W] the row's distribution, entries of drawn tokens zeroed as the draws go
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[2 * 16];
    __shared__ double s_redd[2 * 16];
    __shared__ double s_seg[(NT * SW_C8 * 8) / 32];          // f64 sums of the row's 32-id segments
    __shared__ int s_tok[SW_MAX_DRAW];
    __shared__ float s_p[SW_MAX_DRAW];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    float *out = probs_out + (size_t)row * V;
    int hot = -1;
    if (model == LANTERN_MODEL_LUMINA && pos_ids) {
        const int64_t n1 = pos_ids[row] - pos_base + 1;
        if (n1 == ((int64_t)w_latent + 1) * h_latent + 1) hot = eos_id;
        else if (py_mod64(n1, (int64_t)w_latent + 1) == 0) hot = newline_id;
    }
    // ---- the dense row outside the window: zero (masked ids carry no mass); a forced row: one 1.0
    for (int i4 = tid; i4 * 4 < V; i4 += NT) {
        const int e = i4 * 4;
        if (hot < 0 && e >= win_lo && e + 4 <= win_lo + W) continue;          // the window part is written from the registers below
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hot >= e && hot < e + 4) set_comp(v, hot - e, 1.0f);
        reinterpret_cast<float4 *>(out)[i4] = v;
    }
    if (hot >= 0) {
        // softmax of a row that is -inf everywhere but `hot`; the draws behind the first find no mass: the reference's multinomial returns arbitrary
        // zero-probability ids there -- here the lowest window ids, in order (their conditional probability is 0: the verify side skips them)
        if (tid == 0) {
            double acc = 0.0;
            float prev_c = 0.0f;
            int next = win_lo;
            for (int j = 0; j < n_draw; ++j) {
                int64_t t;
                if (draw_idx) t = draw_idx[(size_t)row * n_draw + j];
                else if (j == 0) t = hot;
                else {
                    if (next == hot) ++next;
                    t = next++;
                }
                const float p = (t == hot) ? 1.0f : 0.0f;
                float v = p / (1.0f - prev_c);
                if (isinf(v) || isnan(v)) v = -1.0f;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                ss_token[(size_t)row * n_draw + j] = t;
                ss_prob[(size_t)row * n_draw + j] = v;
                acc += (double)p;
                prev_c = (float)acc;
            }
        }
        return;
    }
    float4 r[NV4];
#pragma unroll
    for (int it = 0; it < SW_C8; ++it) {
        const int ch = tid + it * NT;
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
        const float thr = kth_largest_hist_bf16<NT, NV4>(r, top_k_filter, s_hist);
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x; r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z; r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    int ph = 0;
    softmax_tile<NT, NV4>(r, s_redf, s_redd, ph);
#pragma unroll
    for (int it = 0; it < SW_C8; ++it) {
        const int w0 = (tid + it * NT) * 8;
        if (w0 < W) {
            *reinterpret_cast<float4 *>(g + w0) = r[2 * it];
            *reinterpret_cast<float4 *>(g + w0 + 4) = r[2 * it + 1];
            // (win_lo % 4 == 0: 16-byte stores into the dense row)
            *reinterpret_cast<float4 *>(out + win_lo + w0) = r[2 * it];
            *reinterpret_cast<float4 *>(out + win_lo + w0 + 4) = r[2 * it + 1];
        }
    }
    __syncthreads();
    // ---- the draws.  Inverse CDF in token-id order with removal (oracle: sample_draws), without a workgroup barrier per draw: every thread leaves the f64
    // sum of its EPT consecutive ids in LDS once; wave 0 then walks the draws alone -- per draw one DPP scan over the segment sums (SPL per lane), a ballot
    // for the segment that holds the crossing, one DPP scan over that segment's EPT entries, the entry zeroed and its segment's sum rebuilt.  (Round 5: the
    // barrier form -- bonus_draw_lds per draw, the whole row re-read and four barriers each -- cost ~2 us per draw: 20 of the kernel's 32 us.)
    constexpr int EPT = 32, NSEG = (NT * SW_C8 * 8) / EPT, SPL = NSEG / 64;          // a segment = 32 consecutive ids = 32 lanes of wave 0
    static_assert(NSEG % 64 == 0 && NSEG <= NT, "segment sums: one per thread of the first NSEG");
    if (tid < NSEG) {
        double sseg = 0.0;
        const int e0 = tid * EPT;
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q) {
            const float4 v = (e0 + 4 * q < W) ? *reinterpret_cast<const float4 *>(g + e0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            sseg += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        }
        s_seg[tid] = sseg;
    }
    __syncthreads();
    if (tid < 64) {
        const int lane = tid;
        double sv[SPL];
#pragma unroll
        for (int q = 0; q < SPL; ++q) sv[q] = s_seg[lane * SPL + q];
        int toks[SW_MAX_DRAW];          // (wave-uniform)
        int fallback = win_lo;
#pragma unroll 1
        for (int j = 0; j < n_draw; ++j) {
            int tok = -1;
            float pv = 0.0f;
            if (draw_idx) {
                const int64_t t = draw_idx[(size_t)row * n_draw + j];
                tok = (int)(t < 0 ? 0 : (t >= V ? V - 1 : t));
                if (tok >= win_lo && tok < win_lo + W) {
                    pv = g[tok - win_lo];          // 0 once it has been drawn before (injected duplicates)
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) g[tok - win_lo] = 0.0f;
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                double ls = 0.0;
#pragma unroll
                for (int q = 0; q < SPL; ++q) ls += sv[q];
                const double inc = wave_scan_incl_dpp(ls);
                const double total = readlane63(inc);
                const double tgt = draw_u[(size_t)row * n_draw + j] * total;
                int seg = -1;
                if (total > 0.0) {
                    // the first segment whose inclusive prefix passes tgt
                    double pre = inc - ls, presel = 0.0;
                    int mine = -1;
#pragma unroll
                    for (int q = 0; q < SPL; ++q) {
                        const double nx = pre + sv[q];
                        if (mine < 0 && sv[q] > 0.0 && nx > tgt) { mine = lane * SPL + q; presel = pre; }
                        pre = nx;
                    }
                    unsigned long long who = __ballot(mine >= 0);
                    int el = -1;
                    if (who) {
                        const int src = __ffsll((long long)who) - 1;
                        seg = __builtin_amdgcn_readlane(mine, src);
                        const double prefix = rdlane(presel, src);
                        const float v = (lane < EPT && seg * EPT + lane < W) ? g[seg * EPT + lane] : 0.0f;
                        const double acc = prefix + wave_scan_incl_dpp((double)v);
                        const unsigned long long hit = __ballot(lane < EPT && v > 0.0f && acc > tgt);
                        if (hit) el = __ffsll((long long)hit) - 1;
                        else {          // rounding at the segment's end: the first positive entry behind it
                            int nxt = -1;
#pragma unroll
                            for (int q = 0; q < SPL; ++q)
                                if (nxt < 0 && lane * SPL + q > seg && sv[q] > 0.0) nxt = lane * SPL + q;
                            const unsigned long long w2 = __ballot(nxt >= 0);
                            seg = w2 ? __builtin_amdgcn_readlane(nxt, __ffsll((long long)w2) - 1) : -1;
                            if (seg >= 0) {
                                const float v2 = (lane < EPT && seg * EPT + lane < W) ? g[seg * EPT + lane] : 0.0f;
                                const unsigned long long pos = __ballot(lane < EPT && v2 > 0.0f);
                                el = pos ? __ffsll((long long)pos) - 1 : -1;
                            }
                        }
                    }
                    if (el < 0) {          // u ~ 1 and the sums' rounding left no crossing: the last positive entry (lo_sample_inverse_cdf)
                        int lastseg = -1;
#pragma unroll
                        for (int q = 0; q < SPL; ++q)
                            if (sv[q] > 0.0) lastseg = lane * SPL + q;
                        const unsigned long long w3 = __ballot(lastseg >= 0);
                        seg = w3 ? __builtin_amdgcn_readlane(lastseg, 63 - __clzll((long long)w3)) : -1;
                        if (seg >= 0) {
                            const float v3 = (lane < EPT && seg * EPT + lane < W) ? g[seg * EPT + lane] : 0.0f;
                            const unsigned long long pos = __ballot(lane < EPT && v3 > 0.0f);
                            el = pos ? 63 - __clzll((long long)pos) : -1;
                        }
                    }
                    if (seg >= 0 && el >= 0) {
                        const int wi = seg * EPT + el;
                        tok = win_lo + wi;
                        const float v = (lane < EPT && seg * EPT + lane < W) ? g[seg * EPT + lane] : 0.0f;
                        pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), el));
                        // remove it: the entry in LDS, and the segment's sum rebuilt from the 31 that stay
                        __builtin_amdgcn_wave_barrier();
                        if (lane == el) g[wi] = 0.0f;
                        __builtin_amdgcn_wave_barrier();
                        const double ns = readlane63(wave_scan_incl_dpp((double)((lane == el) ? 0.0f : v)));
#pragma unroll
                        for (int q = 0; q < SPL; ++q)
                            if (lane * SPL + q == seg) sv[q] = ns;
                    }
                }
                if (tok < 0) {          // no mass left (fewer positive entries than draws): the lowest window ids not drawn yet
                    bool again = true;
                    while (again) {
                        again = false;
                        for (int q = 0; q < j; ++q)
                            if (toks[q] == fallback) {
                                ++fallback;
                                again = true;
                            }
                    }
                    tok = fallback++;
                }
            }
            toks[j] = tok;
            if (lane == 0) {
                s_tok[j] = tok;
                s_p[j] = pv;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        float prev_c = 0.0f;
        for (int j = 0; j < n_draw; ++j) {          // Model.sample's conditional probabilities (lantern_sample_static's arithmetic)
            const float p = s_p[j];
            float v = p / (1.0f - prev_c);
            if (isinf(v) || isnan(v)) v = -1.0f;
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            ss_token[(size_t)row * n_draw + j] = s_tok[j];
            ss_prob[(size_t)row * n_draw + j] = v;
            acc += (double)p;
            prev_c = (float)acc;
        }
    }
}

// The next depth's inputs of a static tree (cnets_lumina_mgpt.py:1258-1262): token j = this depth's draws, flattened, at tree_indices[i + 1][j]
// (`idx.view(-1)[tree_indices]`); hidden row j of both batch rows = this depth's output row of node j's parent (`repeat_hidden`: parent p repeated
// repeat_nums[p] times -> the table `rep`).  One workgroup per (batch row, new token).
__global__ __launch_bounds__(256) void static_next_inputs_kernel(const int64_t *__restrict__ ss_token, int n_flat, const int32_t *__restrict__ gather,
                                                                const int32_t *__restrict__ rep, const uint16_t *__restrict__ out_hidden, int B, int T, int H,
                                                                int T_next, uint16_t *__restrict__ hidden_next, int64_t *__restrict__ ids_next) {
    const int j = blockIdx.x % T_next, b = blockIdx.x / T_next;
    int par = rep[j];
    par = par < 0 ? 0 : (par >= T ? T - 1 : par);
    const uint4 *src = reinterpret_cast<const uint4 *>(out_hidden + ((size_t)b * T + par) * H);
    uint4 *dst = reinterpret_cast<uint4 *>(hidden_next + ((size_t)b * T_next + j) * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) {
        int gi = gather[j];
        gi = gi < 0 ? 0 : (gi >= n_flat ? n_flat - 1 : gi);
        ids_next[(size_t)b * T_next + j] = ss_token[gi];
    }
}

int launch_static_next_inputs(const int64_t *ss_token, int n_flat, const int32_t *gather, const int32_t *rep, const void *out_hidden, int B, int T, int H, int T_next,
                              void *hidden_next, int64_t *ids_next, hipStream_t st) {
    hipLaunchKernelGGL(static_next_inputs_kernel, dim3(B * T_next), dim3(256), 0, st, ss_token, n_flat, gather, rep, (const uint16_t *)out_hidden, B, T, H, T_next,
                       (uint16_t *)hidden_next, ids_next);
    LANTERN_CHECK_LAUNCH("static_next_inputs");
    return LANTERN_OK;
}
}  // namespace lantern

using namespace lantern;

extern "C" int lantern_head_sample(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, int V, float cfg, int model,
                                   const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int newline_id, int eos_id, int top_k_filter,
                                   int n_draw, const double *draw_u, const int64_t *draw_idx, void *workspace, float *probs_out, int64_t *ss_token,
                                   float *ss_prob, int packed, void *sk_workspace, size_t sk_workspace_bytes, void *stream) {
    LANTERN_CHECK_ARG(A && W && workspace && probs_out && ss_token && ss_prob && sk_workspace, "head_sample: null buffer");
    LANTERN_CHECK_ARG(draw_u || draw_idx, "head_sample: the draws need uniforms (draw_u [n, n_draw] f64) or injected indices (draw_idx [n, n_draw] i64)");
    LANTERN_CHECK_ARG(n > 0 && n <= 16 && K > 0 && K % 16 == 0, "head_sample: n=%d drafter rows (<= 16 cond + 16 uncond), K=%d (multiple of 16)", n, K);
    LANTERN_CHECK_ARG(row_lo >= 0 && row_lo % 4 == 0 && n_cols > 0 && n_cols % 8 == 0 && n_cols <= 16384 && row_lo + n_cols <= V && V % 4 == 0,
                      "head_sample: window [%d,+%d) must start on a multiple of 4, be a multiple of 8 ids and <= %d wide, inside V (V %% 4 == 0)", row_lo, n_cols,
                      16384);
    LANTERN_CHECK_ARG(n_draw > 0 && n_draw <= SW_MAX_DRAW && n_draw <= n_cols, "head_sample: n_draw=%d (1..%d)", n_draw, SW_MAX_DRAW);
    LANTERN_CHECK_ARG(model == LANTERN_MODEL_LUMINA || model == LANTERN_MODEL_ANOLE || (model == LANTERN_MODEL_PLAIN && row_lo == 0 && n_cols == V),
                      "head_sample: models whose drafted rows are masked to one id window (Lumina, Anole), or LANTERN_MODEL_PLAIN with the window = the vocabulary");
    if (model == LANTERN_MODEL_LUMINA && pos_ids)
        LANTERN_CHECK_ARG(w_latent > 0 && h_latent > 0 && newline_id >= 0 && newline_id < V && eos_id >= 0 && eos_id < V, "head_sample: Lumina needs latent dims and syntax ids");
    LANTERN_CHECK_ARG(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)probs_out & 15) == 0, "head_sample: workspace / probs_out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_linear_rows_cfg_streamk(A, W, bias, n, K, row_lo, n_cols, cfg, workspace, packed, sk_workspace, sk_workspace_bytes, st);
    if (rc) return rc;
    const size_t lds = (size_t)n_cols * 4;
    if (n_cols <= 8 * 1024)
        hipLaunchKernelGGL((sample_window_kernel<1024, 1>), dim3(n), dim3(1024), lds, st, (const uint16_t *)workspace, n_cols, row_lo, V, model, pos_ids, pos_base, w_latent,
                           h_latent, newline_id, eos_id, top_k_filter, n_draw, draw_u, draw_idx, probs_out, ss_token, ss_prob);
    else
        hipLaunchKernelGGL((sample_window_kernel<1024, 2>), dim3(n), dim3(1024), lds, st, (const uint16_t *)workspace, n_cols, row_lo, V, model, pos_ids, pos_base, w_latent,
                           h_latent, newline_id, eos_id, top_k_filter, n_draw, draw_u, draw_idx, probs_out, ss_token, ss_prob);
    LANTERN_CHECK_LAUNCH("head_sample");
    return LANTERN_OK;
}

extern "C" int lantern_draft_static_inputs(const int64_t *ss_token, int n_flat, const int32_t *gather, const int32_t *rep, const void *out_hidden, int B, int T, int H,
                                           int T_next, void *hidden_next, int64_t *ids_next, void *stream) {
    LANTERN_CHECK_ARG(ss_token && gather && rep && out_hidden && hidden_next && ids_next, "draft_static_inputs: null buffer");
    LANTERN_CHECK_ARG(n_flat > 0 && B > 0 && T > 0 && H > 0 && H % 8 == 0 && T_next > 0, "draft_static_inputs: bad sizes");
    return launch_static_next_inputs(ss_token, n_flat, gather, rep, out_hidden, B, T, H, T_next, hidden_next, ids_next, (hipStream_t)stream);
}
