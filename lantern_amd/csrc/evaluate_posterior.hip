// evaluate_posterior.hip -- O8: relaxed tree rejection sampling on gfx950.
//
// One 1024-thread workgroup (16 wavefronts) per sequence.  The level loop and the
// candidate loop are inherently sequential (each acceptance test depends on the residual
// distribution left by the previous rejection), so parallelism is (a) across the V-wide
// row operations inside a sequence and (b) across sequences (grid = B).
//
// Data layout in HBM
//   logits    [B, rows_per_seq, V] f32   read once per visited level (float4, coalesced)
//   sample_p  [B, V] f32                 doubles as the residual distribution `gtp`:
//                                        written once per level, patched on rejection,
//                                        and already holds the answer when the step ends
//                                        in a rejection (no extra copy)
//   nn_table  [K, K-1] u16               only the first k(+1) columns of one row per tried
//                                        candidate are touched (2-byte coalesced reads)
//   uniforms  [B, n_uniforms] f64        Python's random.random() stream, consumed in the
//                                        reference's order through a per-sequence cursor
//
// Numerics follow torch-CPU where an integer outcome depends on them: the neighbour
// cumsum is accumulated in f64 and rounded to f32 per prefix (torch.cumsum), sums are f64
// -> f32, `r <= acp` compares (float)r, tau = f32(delta-1)*px.
//
// Reference: models/ea_model_lumina_mgpt.py:610-726, models/ea_model_llamagen.py:597-669,
// :709-787, models/ea_model_anole.py (same lines + image-token offset).
#include "common.h"
#include "window_dev.h"          // top_p_tile (TopPLogitsWarper on a register tile)

namespace lantern {

constexpr int EP_THREADS = 1024;
constexpr int EP_NW = EP_THREADS / 64;
constexpr int EP_MAX_P = 128;
constexpr int EP_MAX_D = 16;
constexpr int EP_MAX_PD = 1024;

struct EpShared {
    int cand[EP_MAX_PD];
    int row[EP_MAX_PD];
    int acc[EP_MAX_D];
    int tried[EP_MAX_P];
    int eq[EP_MAX_P];
    double redd[2 * EP_NW];
    float redf[2 * EP_NW];
    int redi[2 * EP_NW];
    double scan_tot[EP_NW];
    double mass[256];          // top_p_tile's probability mass per radix bin
    int fi;
};

// k-th largest of the per-thread register tile (bitwise bisection on order-preserving
// keys: 32 counting passes, one barrier each).  Entries past V are -inf padding.
template <int VI>
__device__ float kth_largest_regs(const float4 (&r)[VI], int k, EpShared &S, int &ph) {
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            c += float_key(r[it].x) >= trial;
            c += float_key(r[it].y) >= trial;
            c += float_key(r[it].z) >= trial;
            c += float_key(r[it].w) >= trial;
        }
        const int tot = block_sum<int, EP_NW>(c, S.redi, ph);
        if (tot >= k) prefix = trial;
    }
    return key_float(prefix);
}

// softmax(processors(row)) -> g, the row held in registers between the passes:
// one HBM read of the row, one write of g.
template <int VI, bool NUCLEUS>
__device__ __attribute__((noinline)) void softmax_to_g(const float *__restrict__ row, float *__restrict__ g, int V, float temperature,
                             int top_k, float top_p, EpShared &S, int &ph) {
    const int tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    float4 r[VI];
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int i4 = tid + it * EP_THREADS;
        r[it] = (i4 * 4 < V) ? reinterpret_cast<const float4 *>(row)[i4] : make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
    }
    if (temperature > 1e-5f && temperature != 1.0f) {
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            r[it].x = r[it].x / temperature;
            r[it].y = r[it].y / temperature;
            r[it].z = r[it].z / temperature;
            r[it].w = r[it].w / temperature;
        }
    }
    if constexpr (NUCLEUS) {          // TopPLogitsWarper between the temperature and the top-k (prepare_logits_processor's order, drafters/utils.py:36-52)
        if (top_p >= 1e-8f && top_p < 1.0f) top_p_tile<EP_THREADS, VI>(r, top_p, S.mass, S.redf, S.redd, S.redi, ph);
    }
    if (top_k > 0) {
        const float thr = kth_largest_regs<VI>(r, top_k < V ? top_k : V, S, ph);
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x;
            r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z;
            r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    float m = NEG_INF;
#pragma unroll
    for (int it = 0; it < VI; ++it) m = fmaxf(fmaxf(m, fmaxf(r[it].x, r[it].y)), fmaxf(r[it].z, r[it].w));
    m = block_max<EP_NW>(m, S.redf, ph);
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        r[it].x = expf(r[it].x - m);
        r[it].y = expf(r[it].y - m);
        r[it].z = expf(r[it].z - m);
        r[it].w = expf(r[it].w - m);
        s += (double)r[it].x + (double)r[it].y + (double)r[it].z + (double)r[it].w;
    }
    const float sf = (float)block_sum<double, EP_NW>(s, S.redd, ph);
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int i4 = tid + it * EP_THREADS;
        if (i4 * 4 < V)
            reinterpret_cast<float4 *>(g)[i4] = make_float4(r[it].x / sf, r[it].y / sf, r[it].z / sf, r[it].w / sf);
    }
    __syncthreads();
}

__device__ __forceinline__ double sum_row_f64(const float *__restrict__ p, int V) {
    double loc = 0.0;
    for (int i4 = threadIdx.x; i4 * 4 < V; i4 += EP_THREADS) {
        const float4 v = reinterpret_cast<const float4 *>(p)[i4];
        loc += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
    return loc;
}

__device__ __forceinline__ void scale_row(float *__restrict__ p, int V, float denom) {
    for (int i4 = threadIdx.x; i4 * 4 < V; i4 += EP_THREADS) {
        float4 v = reinterpret_cast<float4 *>(p)[i4];
        v.x = v.x / denom;
        v.y = v.y / denom;
        v.z = v.z / denom;
        v.w = v.w / denom;
        reinterpret_cast<float4 *>(p)[i4] = v;
    }
}

__device__ __forceinline__ void fill_row(float *__restrict__ p, int V, float val) {
    for (int i4 = threadIdx.x; i4 * 4 < V; i4 += EP_THREADS)
        reinterpret_cast<float4 *>(p)[i4] = make_float4(val, val, val, val);
}

template <int VI, bool NUCLEUS>
__global__ __launch_bounds__(EP_THREADS) void ep_kernel(const lantern_ep_params prm, const lantern_ep_buffers buf) {
    __shared__ EpShared S;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ps = prm.P, Ds = prm.D, V = prm.V;
    const int P = buf.n_paths ? buf.n_paths[b] : Ps;
    const int D = buf.n_depth ? buf.n_depth[b] : Ds;
    const int k = prm.k, off = prm.tok_offset;
    const bool is_static = prm.mode != LANTERN_MODE_DYNAMIC;
    const float NEG_INF = -__builtin_inff();
    int ph = 0;

    const int64_t *cand_g = buf.cand + (size_t)b * Ps * Ds;
    const int32_t *row_g = buf.row_index + (prm.row_index_per_seq ? (size_t)b * Ps * Ds : 0);
    for (int t = tid; t < Ps * Ds; t += EP_THREADS) {
        S.cand[t] = (int)cand_g[t];
        const int rw = row_g[t];
        S.row[t] = rw < 0 ? 0 : (rw >= prm.rows_per_seq ? prm.rows_per_seq - 1 : rw);   // a bad row map must not read outside the batch
    }
    const float *logits = buf.logits + (size_t)b * prm.rows_per_seq * V;
    float *g = buf.sample_p + (size_t)b * V;
    float *qw = buf.workspace ? buf.workspace + (size_t)b * V : nullptr;
    const double *uni = buf.uniforms + (size_t)b * prm.n_uniforms;
    int ucur = buf.cursor ? buf.cursor[b] : 0;
    const int u0 = ucur;
    __syncthreads();
    if (tid == 0) S.acc[0] = S.cand[0];
    __syncthreads();

    int a = 1, best = 0, adjust = 0, status = LANTERN_ST_OK;
    int n_levels = 0, n_tried = 0, n_rej = 0;

    for (int i = 1; i < D && status == LANTERN_ST_OK; ++i) {
        if (i != a) break;
        adjust = 0;
        ++n_levels;
        // rows whose first `a` tokens equal the accepted prefix; fi = first of them
        if (tid < P) {
            int eq = 1;
            for (int t = 0; t < a; ++t) eq &= (S.cand[tid * Ds + t] == S.acc[t]);
            S.eq[tid] = eq;
        }
        __syncthreads();
        if (tid == 0) {
            int fi = -1;
            for (int j = 0; j < P; ++j)
                if (S.eq[j]) {
                    fi = j;
                    break;
                }
            S.fi = fi;
        }
        __syncthreads();
        const int fi = S.fi;
        if (fi < 0) {
            status = LANTERN_ST_NO_PREFIX;
            break;
        }
        softmax_to_g<VI, NUCLEUS>(logits + (size_t)S.row[fi * Ds + (i - 1)] * V, g, V, prm.temperature, prm.top_k, prm.top_p, S, ph);

        int nset = 0;
        for (int j = 0; j < P; ++j) {
            if (!S.eq[j]) continue;
            const int x = S.cand[j * Ds + i];
            if (x == -1) continue;
            bool dup = false;
            for (int t = 0; t < nset; ++t) dup |= (S.tried[t] == x);
            if (dup) continue;
            if (tid == 0) S.tried[nset] = x;
            ++nset;
            __syncthreads();
            if (x < 0 || x >= V) {
                status = LANTERN_ST_TOKEN_OOB;
                break;
            }
            if (ucur >= prm.n_uniforms) {
                status = LANTERN_ST_UNIFORMS;
                break;
            }
            const double r = uni[ucur++];
            ++n_tried;

            float px = g[x];
            int m = 0;
            bool is_syn = false;
            const bool in_img = (x >= prm.img_lo && x < prm.img_hi);
            if (prm.syntax_shortcut)
                for (int t = 0; t < prm.n_syntax; ++t) is_syn |= (x == prm.syntax[t]);
            const uint16_t *nb = nullptr;
            if (prm.syntax_shortcut && is_syn) {
                px = 1.0f;
            } else if (prm.syntax_shortcut && !in_img) {
                px = 0.0f;
            } else if (prm.lantern) {
                const int trow = x - off;
                if (trow < 0 || trow >= prm.table_rows) {
                    status = LANTERN_ST_TABLE_OOB;
                    break;
                }
                nb = buf.nn_table + (size_t)trow * prm.table_cols;
                const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                float csm1 = 0.0f;
                double carry = 0.0;
                for (int base = 0; base < k; base += EP_THREADS) {
                    const int idx = base + tid;
                    double v = 0.0;
                    if (idx < k) {
                        const int id = (int)nb[idx] + off;          // a table id beyond the vocabulary carries no mass
                        v = id < V ? (double)g[id] : 0.0;
                    }
                    double inc = wave_scan_incl(v);
                    if (lane == 63) S.scan_tot[wave] = inc;
                    __syncthreads();
                    double woff = 0.0, total = 0.0;
#pragma unroll
                    for (int w = 0; w < EP_NW; ++w) {
                        const double t = S.scan_tot[w];
                        woff += (w < wave) ? t : 0.0;
                        total += t;
                    }
                    inc += woff + carry;
                    const float cs = (float)inc;
                    const bool ok = idx < k && cs <= tau;
                    const int cnt = block_sum<int, EP_NW>(ok ? 1 : 0, S.redi, ph);
                    const float mx = block_max<EP_NW>(ok ? cs : NEG_INF, S.redf, ph);
                    if (cnt > 0) {
                        m += cnt;
                        csm1 = mx;
                    }
                    carry += total;
                    const int chunk = (k - base) < EP_THREADS ? (k - base) : EP_THREADS;
                    if (cnt < chunk) break;
                }
                if (m > 0) px = px + csm1;
            }

            float qx = 1.0f;
            if (is_static) {
                qx = buf.cart_prob[(size_t)b * Ps * Ds + j * Ds + i];
                if (qx <= 0.0f) continue;
            }
            const float acp = px / qx;
            if ((float)r <= acp) {
                if (tid == 0) S.acc[a] = x;
                ++a;
                best = j;
                __syncthreads();
                break;
            }
            // ------------------------------------------------ rejection: residual
            ++n_rej;
            if (prm.syntax_shortcut && is_syn) {
                status = LANTERN_ST_SYNTAX_REJECT;
                break;
            }
            const bool zero_nb = prm.lantern && m > 0 && (!prm.syntax_shortcut || in_img);
            const int nz = (k + 1 < prm.table_cols) ? k + 1 : prm.table_cols;
            if (!is_static) {
                if (tid == 0) g[x] = 0.0f;
                if (zero_nb)
                    for (int t = tid; t < nz; t += EP_THREADS) {
                        const int id = (int)nb[t] + off;
                        if (id < V) g[id] = 0.0f;
                    }
                __syncthreads();
            } else {
                int qrow = buf.op_off[i - 1] + buf.p_idx[j * Ds + i];
                qrow = qrow < 0 ? 0 : (qrow >= prm.R ? prm.R - 1 : qrow);
                const float *qsrc = buf.orig_prob + ((size_t)b * prm.R + qrow) * (size_t)V;
                const int b0 = buf.b_off[j * Ds + i], b1 = buf.b_off[j * Ds + i + 1];
                for (int i4 = tid; i4 * 4 < V; i4 += EP_THREADS)
                    reinterpret_cast<float4 *>(qw)[i4] = reinterpret_cast<const float4 *>(qsrc)[i4];
                __syncthreads();
                if (b1 > b0) {
                    for (int t = b0 + tid; t < b1; t += EP_THREADS) {
                        const int64_t tok = buf.tree_cand[(size_t)b * prm.N + buf.b_idx[t]];
                        if (tok >= 0 && tok < V) qw[tok] = 0.0f;
                    }
                    __syncthreads();
                    const float qs = (float)block_sum<double, EP_NW>(sum_row_f64(qw, V), S.redd, ph);
                    scale_row(qw, V, qs);
                    __syncthreads();
                }
                if (zero_nb) {
                    float *tgt = (prm.mode == LANTERN_MODE_STATIC_LUMINA) ? g : qw;
                    for (int t = tid; t < nz; t += EP_THREADS) {
                        const int id = (int)nb[t] + off;
                        if (id < V) tgt[id] = 0.0f;
                    }
                    __syncthreads();
                }
                for (int i4 = tid; i4 * 4 < V; i4 += EP_THREADS) {
                    float4 gv = reinterpret_cast<float4 *>(g)[i4];
                    const float4 qv = reinterpret_cast<const float4 *>(qw)[i4];
                    float d;
                    d = gv.x - qv.x; gv.x = d < 0.0f ? 0.0f : d;
                    d = gv.y - qv.y; gv.y = d < 0.0f ? 0.0f : d;
                    d = gv.z - qv.z; gv.z = d < 0.0f ? 0.0f : d;
                    d = gv.w - qv.w; gv.w = d < 0.0f ? 0.0f : d;
                    reinterpret_cast<float4 *>(g)[i4] = gv;
                }
                __syncthreads();
            }
            float gs = (float)block_sum<double, EP_NW>(sum_row_f64(g, V), S.redd, ph);
            if (gs == 0.0f) {
                fill_row(g, V, 1.0f);
                gs = (float)V;
            }
            __syncthreads();
            scale_row(g, V, gs);
            __syncthreads();
            adjust = 1;
        }
    }

    const int from_residual = (adjust && a != D) ? 1 : 0;
    if (status == LANTERN_ST_OK && !from_residual)
        softmax_to_g<VI, NUCLEUS>(logits + (size_t)S.row[best * Ds + (a - 1)] * V, g, V, prm.temperature, prm.top_k, prm.top_p, S, ph);
    if (tid == 0) {
        buf.best[b] = best;
        buf.accept_len[b] = a - 1;
        int32_t *c = buf.counters + (size_t)b * 6;
        c[0] = n_levels;
        c[1] = n_tried;
        c[2] = n_rej;
        c[3] = ucur - u0;
        c[4] = from_residual;
        c[5] = status;
        if (buf.cursor) buf.cursor[b] = ucur;
    }
}

}  // namespace lantern

using namespace lantern;

extern "C" size_t lantern_evaluate_posterior_workspace(const lantern_ep_params *prm) {
    if (!prm || prm->mode == LANTERN_MODE_DYNAMIC) return 0;
    return (size_t)prm->B * (size_t)prm->V * sizeof(float);
}

extern "C" int lantern_evaluate_posterior(const lantern_ep_params *prm, const lantern_ep_buffers *buf, void *stream) {
    LANTERN_CHECK_ARG(prm && buf, "evaluate_posterior: null params");
    const lantern_ep_params &p = *prm;
    LANTERN_CHECK_ARG(p.B >= 0 && p.P > 0 && p.D > 0 && p.V > 0, "evaluate_posterior: bad B/P/D/V");
    if (p.B == 0) return LANTERN_OK;
    LANTERN_CHECK_ARG(p.V % 4 == 0, "evaluate_posterior: V=%d must be a multiple of 4", p.V);
    LANTERN_CHECK_ARG(p.V <= 4096 * 16, "evaluate_posterior: V=%d > 65536 unsupported", p.V);
    LANTERN_CHECK_ARG(p.P <= EP_MAX_P && p.D <= EP_MAX_D && p.P * p.D <= EP_MAX_PD,
                      "evaluate_posterior: P=%d D=%d exceed limits (%d,%d,%d)", p.P, p.D, EP_MAX_P, EP_MAX_D, EP_MAX_PD);
    LANTERN_CHECK_ARG(p.n_syntax >= 0 && p.n_syntax <= 8, "evaluate_posterior: n_syntax");
    LANTERN_CHECK_ARG(buf->logits && buf->row_index && buf->cand && buf->uniforms && buf->best && buf->accept_len &&
                          buf->sample_p && buf->counters,
                      "evaluate_posterior: null required buffer");
    LANTERN_CHECK_ARG(p.mode >= 0 && p.mode <= 2, "evaluate_posterior: bad mode %d", p.mode);
    if (p.mode != LANTERN_MODE_DYNAMIC)
        LANTERN_CHECK_ARG(buf->cart_prob && buf->orig_prob && buf->op_off && buf->p_idx && buf->b_off && buf->b_idx &&
                              buf->tree_cand && buf->workspace && p.R > 0 && p.N > 0,
                          "evaluate_posterior: static mode needs cart_prob/orig_prob/op_off/p_idx/b_off/b_idx/tree_cand/workspace");
    if (p.lantern)
        LANTERN_CHECK_ARG(buf->nn_table && p.k >= 1 && p.k <= p.table_cols && p.table_rows > 0,
                          "evaluate_posterior: lantern needs nn_table and 1 <= k=%d <= table_cols=%d", p.k, p.table_cols);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.B), block(EP_THREADS);
    const bool nucleus = p.top_p >= 1e-8f && p.top_p < 1.0f;          // its own instances: the default ones keep their register budget
#define EP_LAUNCH(VI_)                                                               \
    do {                                                                             \
        if (nucleus) hipLaunchKernelGGL((ep_kernel<VI_, true>), grid, block, 0, st, p, *buf); \
        else hipLaunchKernelGGL((ep_kernel<VI_, false>), grid, block, 0, st, p, *buf);        \
    } while (0)
    if (p.V <= 4096) EP_LAUNCH(1);
    else if (p.V <= 4096 * 4) EP_LAUNCH(4);
    else if (p.V <= 4096 * 8) EP_LAUNCH(8);
    else EP_LAUNCH(16);
#undef EP_LAUNCH
    LANTERN_CHECK_LAUNCH("evaluate_posterior");
    return LANTERN_OK;
}
