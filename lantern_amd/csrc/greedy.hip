// greedy.hip -- a9: greedy / TVD branch of evaluate_posterior (temperature <= 1e-5).
//
// The reference materialises softmax over [P, D, V] plus [P, D-1, 2k] temporaries
// (models/ea_model_llamagen.py:789-905; Anole adds the image-token offset, ea_model_anole.py:811-821).
// Here every (path, depth) cell is an independent 256-thread workgroup that keeps the unnormalised
// exp row in LDS (window <= 16384 ids: LlamaGen's whole vocabulary, Anole's image range), gathers the
// k neighbour masses, runs the two cumulative sums of the TVD expression in f64 and decides
// `argmax(gtp) == candidate` from two block maxima -- nothing is materialised.  A one-wavefront
// kernel then takes cumprod/sum per path, the max / first-argmax over paths, and copies the answer row.
//
// The float expression for tvd is kept term by term (0.5*|px-(px+c)| + cumsum(0.5*|nb|)) so that
// threshold decisions match the reference's (SURVEY 8a numerical caveat 1).
#include "common.h"

namespace lantern {

constexpr int GR_THREADS = 256;
constexpr int GR_NW = GR_THREADS / 64;

template <int E4>
__global__ __launch_bounds__(GR_THREADS) void greedy_cell_kernel(const float *__restrict__ logits, const int32_t *__restrict__ row_index,
                                                                 const int64_t *__restrict__ cand, int P, int D, int V, int rows_per_seq,
                                                                 int row_index_per_seq, int lantern, int k, double delta, int tok_offset,
                                                                 const uint16_t *__restrict__ nn_table, int table_rows, int table_cols,
                                                                 int win_lo, int W, int32_t *__restrict__ ok_out) {
    extern __shared__ float4 dyn_lds[];
    float *e = reinterpret_cast<float *>(dyn_lds);   // unnormalised exp row (window)
    __shared__ double s_redd[2 * GR_NW];
    __shared__ float s_redf[2 * GR_NW];
    __shared__ double s_tot[2][GR_NW];
    const int b = blockIdx.y, cell = blockIdx.x, p = cell / (D - 1), d = cell % (D - 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float NEG_INF = -__builtin_inff();
    int ph = 0;
    const int64_t x64 = cand[((size_t)b * P + p) * D + d + 1];
    int32_t *okp = ok_out + ((size_t)b * P + p) * (D - 1) + d;
    if (x64 == -1 || x64 < 0 || x64 >= V) {   // invalid position: posterior_mask = 0 (:491-499)
        if (tid == 0) *okp = 0;
        return;
    }
    const int x = (int)x64;
    const int rid = row_index[(row_index_per_seq ? (size_t)b * P * D : 0) + (size_t)p * D + d];
    const float *rowp = logits + ((size_t)b * rows_per_seq + rid) * V + win_lo;
    const bool x_in = x >= win_lo && x < win_lo + W;
    float4 r[E4];
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * GR_THREADS;
        r[it] = (i4 * 4 < W) ? reinterpret_cast<const float4 *>(rowp)[i4] : make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
    }
    // maxima before / after the candidate's index (torch.argmax returns the FIRST maximum)
    float m_before = NEG_INF, m_after = NEG_INF, m_all = NEG_INF;
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int e0 = win_lo + (tid + it * GR_THREADS) * 4;
        const float vv[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int id = e0 + c;
            m_all = fmaxf(m_all, vv[c]);
            if (id < x) m_before = fmaxf(m_before, vv[c]);
            if (id > x) m_after = fmaxf(m_after, vv[c]);
        }
    }
    m_before = block_max_fast<GR_NW>(m_before, s_redf, ph);
    m_after = block_max_fast<GR_NW>(m_after, s_redf, ph);
    const float lx = x_in ? rowp[x - win_lo] : NEG_INF;
    if (!lantern) {
        if (tid == 0) *okp = (lx > m_before && lx >= m_after) ? 1 : 0;
        return;
    }
    m_all = fmaxf(fmaxf(m_before, m_after), lx);
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * GR_THREADS;
        r[it].x = expf(r[it].x - m_all); r[it].y = expf(r[it].y - m_all);
        r[it].z = expf(r[it].z - m_all); r[it].w = expf(r[it].w - m_all);
        s += (double)r[it].x + (double)r[it].y + (double)r[it].z + (double)r[it].w;
        if (i4 * 4 < W) reinterpret_cast<float4 *>(e)[i4] = r[it];
    }
    const float sf = (float)block_sum_fast<double, GR_NW>(s, s_redd, ph);   // barrier inside: e[] visible
    const float px = x_in ? e[x - win_lo] / sf : 0.0f;
    const int trow = x - tok_offset;
    float px_adj = px;
    if (trow >= 0 && trow < table_rows) {
        const uint16_t *nb = nn_table + (size_t)trow * table_cols;
        const float tau = delta > 1.0 ? (float)(delta - 1.0) * px : (float)delta;
        double carry_c = 0.0, carry_t = 0.0;
        float best = -__builtin_inff();
        for (int base = 0; base < k; base += GR_THREADS * 4) {
            const int i0 = base + tid * 4;
            double vc[4], vt[4], lc = 0.0, lt = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float nbp = 0.0f;
                if (i0 + c < k) {
                    const int t = (int)nb[i0 + c] + tok_offset - win_lo;
                    nbp = (t >= 0 && t < W) ? e[t] / sf : 0.0f;
                }
                lc += (double)nbp;
                lt += (double)(0.5f * fabsf(nbp - 0.0f));
                vc[c] = lc;
                vt[c] = lt;
            }
            const double ic = wave_scan_incl_dpp(lc), itv = wave_scan_incl_dpp(lt);
            if (lane == 63) {
                s_tot[0][wave] = ic;
                s_tot[1][wave] = itv;
            }
            __syncthreads();
            double oc = carry_c, ot = carry_t, tc = 0.0, tt = 0.0;
#pragma unroll
            for (int w = 0; w < GR_NW; ++w) {
                const double a0 = s_tot[0][w], a1 = s_tot[1][w];
                oc += (w < wave) ? a0 : 0.0;
                ot += (w < wave) ? a1 : 0.0;
                tc += a0;
                tt += a1;
            }
            oc += ic - lc;
            ot += itv - lt;
            float mx = -__builtin_inff();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float cs = (float)(oc + vc[c]);           // cumsum_nearest_probs
                const float approx = px + cs;                     // approx_p
                const float tvd = 0.5f * fabsf(px - approx) + (float)(ot + vt[c]);
                const bool ok = (i0 + c < k) && tvd <= tau;
                mx = ok ? fmaxf(mx, approx) : mx;                 // tvd and approx are non-decreasing: last ok = max ok
            }
            best = fmaxf(best, block_max_fast<GR_NW>(mx, s_redf, ph));
            carry_c += tc;
            carry_t += tt;
        }
        if (best > -__builtin_inff()) px_adj = best;
    }
    // argmax(gtp with gtp[x] = px_adj) == x ; p_j = e_j / sf is monotone in e_j
    float eb = NEG_INF, ea = NEG_INF;
    for (int i4 = tid; i4 * 4 < W; i4 += GR_THREADS) {
        const float4 v = reinterpret_cast<const float4 *>(e)[i4];
        const int e0 = win_lo + i4 * 4;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (e0 + c < x) eb = fmaxf(eb, vv[c]);
            if (e0 + c > x) ea = fmaxf(ea, vv[c]);
        }
    }
    eb = block_max_fast<GR_NW>(eb, s_redf, ph);
    ea = block_max_fast<GR_NW>(ea, s_redf, ph);
    const float pb = eb > NEG_INF ? eb / sf : NEG_INF, pa = ea > NEG_INF ? ea / sf : NEG_INF;
    bool ok = px_adj > pb && px_adj >= pa;
    // ids in front of the window hold probability 0 and come first in argmax order
    if (win_lo > 0 && !(px_adj > 0.0f)) ok = false;
    if (tid == 0) *okp = ok ? 1 : 0;
}

__global__ __launch_bounds__(64) void greedy_finalize_kernel(const int32_t *__restrict__ ok, const float *__restrict__ logits,
                                                             const int32_t *__restrict__ row_index, int P, int D, int V, int rows_per_seq,
                                                             int row_index_per_seq, int32_t *__restrict__ best_out,
                                                             int32_t *__restrict__ alen_out, float *__restrict__ out_row) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int alen = 0;
    if (lane < P) {
        int run = 1;
        for (int d = 0; d < D - 1; ++d) {
            run &= ok[((size_t)b * P + lane) * (D - 1) + d];
            alen += run;
        }
    }
    int mx = alen;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    const unsigned long long m = __ballot(lane < P && alen == mx);
    const int best = mx == 0 ? 0 : (__ffsll((long long)m) - 1);   // torch.argmax: first maximum
    if (lane == 0) {
        best_out[b] = best;
        alen_out[b] = mx;
    }
    const int rid = row_index[(row_index_per_seq ? (size_t)b * P * D : 0) + (size_t)best * D + mx];
    const float4 *src = reinterpret_cast<const float4 *>(logits + ((size_t)b * rows_per_seq + rid) * V);
    float4 *dst = reinterpret_cast<float4 *>(out_row + (size_t)b * V);
    for (int i4 = lane; i4 * 4 < V; i4 += 64) dst[i4] = src[i4];
}

}  // namespace lantern

using namespace lantern;

extern "C" int lantern_evaluate_posterior_greedy(const float *logits, const int32_t *row_index, const int64_t *cand, int B, int P, int D,
                                                 int V, int rows_per_seq, int row_index_per_seq, int lantern, int k, double delta,
                                                 int tok_offset, const uint16_t *nn_table, int table_rows, int table_cols, int win_lo,
                                                 int win_len, int32_t *ok_scratch, int32_t *best, int32_t *accept_len, float *out_row,
                                                 void *stream) {
    LANTERN_CHECK_ARG(logits && row_index && cand && ok_scratch && best && accept_len && out_row, "evaluate_posterior_greedy: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && P > 0 && P <= 64 && D >= 2 && V > 0 && V % 4 == 0, "evaluate_posterior_greedy: bad sizes (P <= 64, D >= 2, V %% 4 == 0)");
    LANTERN_CHECK_ARG(win_lo >= 0 && win_lo % 4 == 0 && win_len > 0 && win_len % 4 == 0 && win_lo + win_len <= V && win_len <= 16384,
                      "evaluate_posterior_greedy: window [%d,+%d) must be 4-aligned, inside V and <= 16384 wide", win_lo, win_len);
    if (lantern) LANTERN_CHECK_ARG(nn_table && k >= 1 && k <= table_cols && table_rows > 0, "evaluate_posterior_greedy: lantern needs nn_table, 1<=k<=cols");
    if (B == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(P * (D - 1), B);
    const size_t lds = (size_t)win_len * 4;
#define GR_ARGS logits, row_index, cand, P, D, V, rows_per_seq, row_index_per_seq, lantern, k, delta, tok_offset, nn_table, table_rows, table_cols, win_lo, win_len, ok_scratch
    if (win_len <= 1024) hipLaunchKernelGGL((greedy_cell_kernel<1>), grid, dim3(GR_THREADS), lds, st, GR_ARGS);
    else if (win_len <= 2048) hipLaunchKernelGGL((greedy_cell_kernel<2>), grid, dim3(GR_THREADS), lds, st, GR_ARGS);
    else if (win_len <= 4096) hipLaunchKernelGGL((greedy_cell_kernel<4>), grid, dim3(GR_THREADS), lds, st, GR_ARGS);
    else if (win_len <= 8192) hipLaunchKernelGGL((greedy_cell_kernel<8>), grid, dim3(GR_THREADS), lds, st, GR_ARGS);
    else hipLaunchKernelGGL((greedy_cell_kernel<16>), grid, dim3(GR_THREADS), lds, st, GR_ARGS);
#undef GR_ARGS
    hipLaunchKernelGGL(greedy_finalize_kernel, dim3(B), dim3(64), 0, st, ok_scratch, logits, row_index, P, D, V, rows_per_seq,
                       row_index_per_seq, best, accept_len, out_row);
    LANTERN_CHECK_LAUNCH("evaluate_posterior_greedy");
    return LANTERN_OK;
}
