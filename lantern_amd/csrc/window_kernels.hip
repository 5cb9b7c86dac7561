// window_kernels.hip -- v2 of the verify hot path: windowed O7 and O8 + window->dense.
//
// For Lumina / Anole every processed row is -inf outside the image-token range [4, 8196) (or is a
// forced one-hot row), so a row is 8192 floats = 32 KB, not 65536 floats = 256 KB: it fits in LDS.
//   cfg_window_kernel : one 256-thread workgroup per node row; reads only the window of cond/uncond,
//                       k-th largest from registers, writes the 32 KB window (+ one int for one-hot rows).
//   epw_kernel        : one 256-thread workgroup per sequence; the residual distribution `gtp` lives in
//                       LDS for the whole step; per level one 32 KB row read, per tried candidate one
//                       2 KB table-row read + an LDS gather + an f64 wave/block scan, per rejection one
//                       32 KB drafter-row read (static trees); the bonus token is drawn from LDS in the
//                       epilogue (inverse CDF), so the dense sample_p[V] never exists unless asked for.
// Same arithmetic, same order of operations as the dense kernels (evaluate_posterior.hip).
//
// Reference: models/ea_model_lumina_mgpt.py:597-605 (O7), :610-726 (O8), :781 (bonus token);
// models/ea_model_llamagen.py:597-669,709-787.
#include "common.h"
#include <type_traits>

namespace lantern {
// In-kernel phase stamps for diagnosis (tools/ep_trace.py builds a separate .so with -DEPW_TRACE);
// the shipped library compiles EPW_STAMP to nothing.
#ifdef EPW_TRACE
// every workgroup stamps into its own LDS (thread 0: one s_memtime + two LDS accesses per stamp) and dumps at exit
constexpr int EPW_TR_MAX = 256, EPW_TR_BLOCKS = 64;
__device__ unsigned long long g_epw_trace[EPW_TR_BLOCKS][EPW_TR_MAX];
__device__ int g_epw_trace_n[EPW_TR_BLOCKS];
__shared__ unsigned long long s_epw_tr[EPW_TR_MAX];
__shared__ int s_epw_trn;
#define EPW_STAMP(id)                                                                                          \
    do {                                                                                                       \
        if (threadIdx.x == 0) {                                                                                \
            const int n__ = s_epw_trn;                                                                         \
            if (n__ < EPW_TR_MAX) {                                                                            \
                s_epw_tr[n__] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
                s_epw_trn = n__ + 1;                                                                           \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#if EPW_TRACE >= 2
#define EPW_STAMPF(id) EPW_STAMP(id)
#else
#define EPW_STAMPF(id) do { } while (0)
#endif
#if EPW_TRACE >= 3
#define EPW_STAMPG(id) EPW_STAMP(id)
#else
#define EPW_STAMPG(id) do { } while (0)
#endif
#else
#define EPW_STAMP(id) do { } while (0)
#define EPW_STAMPF(id) do { } while (0)
#define EPW_STAMPG(id) do { } while (0)
#endif


}  // namespace lantern

#include "window_dev.h"
#include "tree_dynamic_dev.h"
#include "epw_body.h"
#include "prep_dev.h"

namespace lantern {


// ------------------------------------------------------------------------------- O7 windowed
template <int NT, int E4, bool BF16>
__global__ __launch_bounds__(NT) void cfg_window_kernel(const void *__restrict__ cond_, const void *__restrict__ uncond_, int V, float cfg,
                                                        int model, const int64_t *__restrict__ pos_ids, int64_t pos_base, int w_latent,
                                                        int h_latent, int img_lo, int img_hi, int newline_id, int eos_id, int top_k,
                                                        const int64_t *__restrict__ seq_len, int rows_per_seq, int win_lo, int W,
                                                        float *__restrict__ out_win, int32_t *__restrict__ row_hot, int out_kind,
                                                        float temperature, float top_p) {
    __shared__ int s_hist[256];
    __shared__ float s_redf[32];
    __shared__ double s_redd[32];
    __shared__ int s_redi[32];
    __shared__ double s_mass[256];
    const int row = o7_row_of_block(blockIdx.x, gridDim.x, seq_len ? rows_per_seq : 0), tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    float *out = out_win + (size_t)row * W;
    int cls = 0;
    if (model == LANTERN_MODEL_LUMINA) {
        const int64_t pos = seq_len ? pos_ids[row % rows_per_seq] + seq_len[row / rows_per_seq] : pos_ids[row];
        const int64_t n1 = pos - pos_base + 1;
        if (n1 == ((int64_t)w_latent + 1) * h_latent + 1)
            cls = 2;
        else if (py_mod64(n1, (int64_t)w_latent + 1) == 0)
            cls = 1;
    }
    if (cls != 0) {
        if (tid == 0) row_hot[row] = cls == 2 ? eos_id : newline_id;   // one-hot row: its window is never read
        return;
    }
    if (tid == 0) row_hot[row] = -1;
    const bool lumina = model == LANTERN_MODEL_LUMINA;
    const bool masked = model != LANTERN_MODEL_PLAIN;
    const float fill = lumina ? NEG_INF : (BF16 ? __uint_as_float(0xff7f0000u) : -3.4028234663852886e38f);
    float4 r[E4];
    // every load of the row in flight before the first use, the unconditional ones under ONE wave-uniform branch (a per-chunk `uncond ? load : cond`
    // made the compiler wait for each conditional chunk and fetch the unconditional one element by element: a serial HBM round trip per chunk)
    typedef typename std::conditional<BF16, ushort4, float4>::type Raw;
    Raw craw[E4], uraw[E4];
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        craw[it] = Raw{};
        if (i4 * 4 < W) craw[it] = reinterpret_cast<const Raw *>(cond_)[((size_t)row * V + win_lo + i4 * 4) / 4];
    }
    if (uncond_) {
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int i4 = tid + it * NT;
            uraw[it] = Raw{};
            if (i4 * 4 < W) uraw[it] = reinterpret_cast<const Raw *>(uncond_)[((size_t)row * V + win_lo + i4 * 4) / 4];
        }
    } else {
#pragma unroll
        for (int it = 0; it < E4; ++it) uraw[it] = craw[it];
    }
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        float4 v = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        if (i4 * 4 < W) {
            const int e = win_lo + i4 * 4;
            float c[4], u[4];
            if constexpr (BF16) {
                const ushort4 cb = craw[it], ub = uraw[it];
                c[0] = bf16_bits_to_f32(cb.x); c[1] = bf16_bits_to_f32(cb.y); c[2] = bf16_bits_to_f32(cb.z); c[3] = bf16_bits_to_f32(cb.w);
                u[0] = bf16_bits_to_f32(ub.x); u[1] = bf16_bits_to_f32(ub.y); u[2] = bf16_bits_to_f32(ub.z); u[3] = bf16_bits_to_f32(ub.w);
            } else {
                const float4 cf = craw[it], uf = uraw[it];
                c[0] = cf.x; c[1] = cf.y; c[2] = cf.z; c[3] = cf.w;
                u[0] = uf.x; u[1] = uf.y; u[2] = uf.z; u[3] = uf.w;
            }
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = c[q];
                if (uncond_) {
                    t = c[q] - u[q];
                    if (BF16) t = round_bf16(t);
                    t = cfg * t;
                    if (BF16) t = round_bf16(t);
                    t = u[q] + t;
                    if (BF16) t = round_bf16(t);
                }
                o[q] = (masked && (e + q < img_lo || e + q >= img_hi)) ? fill : t;
            }
            v = make_float4(o[0], o[1], o[2], o[3]);
        }
        r[it] = v;
    }
    const bool scaled = temperature > 1e-5f && temperature != 1.0f;      // TemperatureLogitsWarper (drafters/utils.py:36-52)
    if (scaled) {
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            r[it].x = r[it].x / temperature; r[it].y = r[it].y / temperature;
            r[it].z = r[it].z / temperature; r[it].w = r[it].w / temperature;
        }
    }
    const bool nucleus = top_p >= 1e-8f && top_p < 1.0f;      // TopPLogitsWarper (HF order: Temperature -> TopP -> TopK)
    if (nucleus) {
        int ph = 0;
        top_p_tile<NT, E4>(r, top_p, s_mass, s_redf, s_redd, s_redi, ph);
    }
    if (top_k > 0 && top_k < V) {
        // k-th largest of the FULL row = k-th largest of the window whenever >= k window entries beat the fill
        // value; otherwise the threshold is the fill value (or lower) and nothing inside the window is removed.
        float thr = NEG_INF;
        if (top_k <= W) thr = (BF16 && !scaled) ? kth_largest_hist<NT, E4, BF16>(r, top_k, s_hist) : kth_largest_hist<NT, E4, false>(r, top_k, s_hist);
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x;
            r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z;
            r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    if (out_kind == LANTERN_ROWS_PROBS) {
        int ph = 0;
        softmax_tile<NT, E4>(r, s_redf, s_redd, ph);
    }
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        if (i4 * 4 < W) reinterpret_cast<float4 *>(out)[i4] = r[it];
    }
}

// (cfg_window_bf16_row, lumina_row_class, PrepArgs and prep_rows_body live in prep_dev.h: the commit launch of gather_ops.hip shares them)

template <int NT, int E8, bool FULL>
__global__ __launch_bounds__(NT) void cfg_window_bf16_kernel(const uint16_t *__restrict__ cond, const uint16_t *__restrict__ uncond, int V,
                                                             float cfg, int model, const int64_t *__restrict__ pos_ids, int64_t pos_base,
                                                             int w_latent, int h_latent, int img_lo, int img_hi, int newline_id, int eos_id,
                                                             int top_k, const int64_t *__restrict__ seq_len, int rows_per_seq, int win_lo,
                                                             int W, float *__restrict__ out_win, int32_t *__restrict__ row_hot,
                                                             int out_kind) {
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[32];
    __shared__ double s_redd[32];
    const int row = o7_row_of_block(blockIdx.x, gridDim.x, seq_len ? rows_per_seq : 0);
    int cls = 0;
    if (model == LANTERN_MODEL_LUMINA)
        cls = lumina_row_class(seq_len ? pos_ids[row % rows_per_seq] + seq_len[row / rows_per_seq] : pos_ids[row], pos_base, w_latent, h_latent);
    cfg_window_bf16_row<NT, E8, FULL>(row, cls, cond, uncond, V, cfg, model, img_lo, img_hi, newline_id, eos_id, top_k, win_lo, W, out_win, row_hot,
                                      out_kind, s_hist, s_redf, s_redd);
}

template <int NT, int E8, bool NUCLEUS = false>
__global__ __launch_bounds__(NT) void prep_rows_kernel(const PrepArgs a) {
    prep_rows_body<NT, E8, NUCLEUS>(a, (int)blockIdx.x);
}

// The same for a group of DYNAMIC trees (lantern_step_group.dyn): workgroups [0, B * n_list) post-process the listed rows -- a listed node's
// position is assumed (node_list[n_list + i] = its depth: the root, and node 1 = the drafter's best first token, always at depth 1 in an
// EAGLE-2 tree), so the rows do not wait for the tree -- while workgroups [B * n_list, B * n_list + B) build the sequence's tree and its
// candidates (O4 + O6-dynamic, tree_dynamic_dev.h).  evaluate_posterior uses a prepared row only when the node really sits at that depth.
struct DynPrepArgs {
    const uint16_t *cond, *uncond;
    int V;
    float cfg;
    int64_t pos_base;
    int w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k;
    int rows_per_seq, win_lo, W;
    float *out_win;
    int32_t *row_hot;
    const int32_t *node_list;
    int n_list, B;
    TdArgs td;
    float top_p;
};

template <int NT, int E8, bool NUCLEUS = false>
__global__ __launch_bounds__(NT) void dyn_prep_kernel(const DynPrepArgs a) {
    const int n_rows = a.B * a.n_list;
    if ((int)blockIdx.x < n_rows) {
        __shared__ alignas(16) int s_hist[O7_HIST_INTS];
        __shared__ float s_redf[32];
        __shared__ double s_redd[32];
        __shared__ int s_redi[32];
        const int x = o7_row_of_block(blockIdx.x, n_rows, a.n_list);
        const int b = x / a.n_list, i = x % a.n_list;
        const int node = a.node_list[i], depth = a.node_list[a.n_list + i];
        const int row = b * a.rows_per_seq + node;
        // (= pos_abs of a node at that depth; w_latent == 0: a model without grammar rows -- LlamaGen)
        const int cls = a.w_latent > 0 ? lumina_row_class(a.td.cd.seq_len[b] + 1 + depth, a.pos_base, a.w_latent, a.h_latent) : 0;
        cfg_window_bf16_row<NT, E8, true, NUCLEUS>(row, cls, a.cond, a.uncond, a.V, a.cfg, LANTERN_MODEL_LUMINA, a.img_lo, a.img_hi, a.newline_id, a.eos_id,
                                                   a.top_k, a.win_lo, a.W, a.out_win, a.row_hot, LANTERN_ROWS_PROBS, s_hist, s_redf, s_redd, a.top_p, s_redi);
        return;
    }
    td_finalize_body<8, NT / 64>(a.td, blockIdx.x - n_rows);
}


__global__ void window_to_dense_kernel(const float *__restrict__ winp, const int32_t *__restrict__ out_tok,
                                       const float *__restrict__ out_mass, int V, int lo, int W, float *__restrict__ dense) {
    const int b = blockIdx.y;
    const int ot = out_tok ? out_tok[b] : -1;
    const float om = out_mass ? out_mass[b] : 0.0f;
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < V; i4 += gridDim.x * blockDim.x) {
        const int e = i4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e >= lo && e < lo + W) v = reinterpret_cast<const float4 *>(winp + (size_t)b * W)[(e - lo) / 4];
        if (ot >= e && ot < e + 4) set_comp(v, ot - e, om);
        reinterpret_cast<float4 *>(dense + (size_t)b * V)[i4] = v;
    }
}

}  // namespace lantern

using namespace lantern;

template <int NT, int E4>
static void launch_cfgw(bool bf16, int rows, hipStream_t st, const void *cond, const void *uncond, int V, float cfg, int model,
                        const int64_t *pos_ids, int64_t pos_base, int w, int h, int img_lo, int img_hi, int nl, int eos, int top_k,
                        const int64_t *seq_len, int rps, int win_lo, int W, float *out, int32_t *hot, int out_kind, float temperature,
                        float top_p) {
    if (bf16)
        hipLaunchKernelGGL((cfg_window_kernel<NT, E4, true>), dim3(rows), dim3(NT), 0, st, cond, uncond, V, cfg, model, pos_ids, pos_base, w,
                           h, img_lo, img_hi, nl, eos, top_k, seq_len, rps, win_lo, W, out, hot, out_kind, temperature, top_p);
    else
        hipLaunchKernelGGL((cfg_window_kernel<NT, E4, false>), dim3(rows), dim3(NT), 0, st, cond, uncond, V, cfg, model, pos_ids, pos_base, w,
                           h, img_lo, img_hi, nl, eos, top_k, seq_len, rps, win_lo, W, out, hot, out_kind, temperature, top_p);
}

extern "C" int lantern_cfg_mask_topk_window(const void *cond, const void *uncond, int dtype, int rows, int V, float cfg, int model,
                                            const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int img_lo,
                                            int img_hi, int newline_id, int eos_id, int top_k, const int64_t *seq_len,
                                            int rows_per_seq, int win_lo, int win_len, float *out_win, int32_t *row_hot, int out_kind,
                                            float temperature, float top_p, void *stream) {
    LANTERN_CHECK_ARG(cond && out_win && row_hot, "cfg_mask_topk_window: null buffer");
    LANTERN_CHECK_ARG(out_kind == LANTERN_ROWS_LOGITS || out_kind == LANTERN_ROWS_PROBS, "cfg_mask_topk_window: bad out_kind %d", out_kind);
    LANTERN_CHECK_ARG(temperature > 1e-5f, "cfg_mask_topk_window: temperature %g (the greedy branch has its own kernel)", (double)temperature);
    LANTERN_CHECK_ARG(rows >= 0 && V > 0 && V % 4 == 0, "cfg_mask_topk_window: bad rows=%d V=%d", rows, V);
    LANTERN_CHECK_ARG(win_lo >= 0 && win_lo % 4 == 0 && win_len > 0 && win_len % 4 == 0 && win_lo + win_len <= V && win_len <= 16384,
                      "cfg_mask_topk_window: window [%d,+%d) must be 4-aligned, inside V and <= 16384 wide", win_lo, win_len);
    LANTERN_CHECK_ARG(dtype == LANTERN_F32 || dtype == LANTERN_BF16, "cfg_mask_topk_window: bad dtype");
    LANTERN_CHECK_ARG(model >= 0 && model <= 2, "cfg_mask_topk_window: bad model");
    if (model == LANTERN_MODEL_PLAIN)
        LANTERN_CHECK_ARG(win_lo == 0 && win_len == V, "cfg_mask_topk_window: an unmasked model needs the window to be the whole vocabulary");
    else
        LANTERN_CHECK_ARG(img_lo >= win_lo && img_hi <= win_lo + win_len && img_lo < img_hi,
                          "cfg_mask_topk_window: image range [%d,%d) must lie inside the window", img_lo, img_hi);
    if (model == LANTERN_MODEL_LUMINA)
        LANTERN_CHECK_ARG(pos_ids && w_latent > 0 && h_latent > 0 && newline_id >= 0 && newline_id < V && eos_id >= 0 && eos_id < V,
                          "cfg_mask_topk_window: Lumina needs pos_ids, latent dims and syntax ids");
    if (seq_len) LANTERN_CHECK_ARG(rows_per_seq > 0 && rows % rows_per_seq == 0, "cfg_mask_topk_window: rows %% rows_per_seq != 0");
    if (rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = dtype == LANTERN_BF16;
#define CW_ARGS bf, rows, st, cond, uncond, V, cfg, model, pos_ids, pos_base, w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k, seq_len, rows_per_seq, win_lo, win_len, out_win, row_hot, out_kind, temperature, top_p
    const bool nucleus = top_p >= 1e-8f && top_p < 1.0f;
    if (bf && win_len % 8 == 0 && win_len >= 2048 && temperature == 1.0f && !nucleus) {
        const int chunks = win_len / 8;
        const uint16_t *c16 = (const uint16_t *)cond, *u16 = (const uint16_t *)uncond;
        const int nt_knob = tuning(TUNE_O7_NT);
#define CW16_ARGS c16, u16, V, cfg, model, pos_ids, pos_base, w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k, seq_len, rows_per_seq, win_lo, win_len, out_win, row_hot, out_kind
#define CW16(NT_, E8_)                                                                                                       \
    do {                                                                                                                     \
        if (chunks == NT_ * E8_) LANTERN_LAUNCH((cfg_window_bf16_kernel<NT_, E8_, true>), dim3(rows), dim3(NT_), 0, st, CW16_ARGS);  \
        else LANTERN_LAUNCH((cfg_window_bf16_kernel<NT_, E8_, false>), dim3(rows), dim3(NT_), 0, st, CW16_ARGS);               \
    } while (0)
        if (chunks <= 256 * 2) CW16(256, 2);
        else if (chunks <= 256 * 4 && nt_knob == 256) CW16(256, 4);
        else if (chunks <= 512 * 2 && nt_knob != 1024) CW16(512, 2);
        else if (chunks <= 1024 * 1) CW16(1024, 1);
        else CW16(1024, 2);
#undef CW16
#undef CW16_ARGS
        LANTERN_CHECK_LAUNCH("cfg_mask_topk_window");
        return LANTERN_OK;
    }
    if (win_len <= 1024) launch_cfgw<256, 1>(CW_ARGS);
    else if (win_len <= 2048) launch_cfgw<256, 2>(CW_ARGS);
    else if (win_len <= 4096) launch_cfgw<512, 2>(CW_ARGS);
    else if (win_len <= 8192) launch_cfgw<512, 4>(CW_ARGS);
    else launch_cfgw<1024, 4>(CW_ARGS);
#undef CW_ARGS
    LANTERN_CHECK_LAUNCH("cfg_mask_topk_window");
    return LANTERN_OK;
}

static int prepare_step_dynamic(const lantern_step_group *g) {
    const lantern_step_dynamic &d = *g->dyn;
    const int N = d.total_tokens + 1;
    LANTERN_CHECK_ARG(d.scores && d.tokens && d.parents && g->sample_token && d.draft_tokens && d.mask && d.pos_ids && d.retrieve && d.n_leaf && d.max_depth &&
                          d.seq_len && g->cand, "prepare_step: dynamic-tree buffers missing");
    LANTERN_CHECK_ARG(d.top_k > 0 && d.total_tokens >= 1 && d.total_tokens <= 63 && d.n_scores >= d.total_tokens && d.n_scores <= 512 &&
                          d.n_parents * d.top_k >= d.n_scores && g->N == N && g->P > 0 && g->P <= N && g->D > 0 && g->D <= N,
                      "prepare_step: dynamic tree sizes (n_scores <= 512, total_tokens <= 63, P, D <= N = total_tokens + 1)");
    if (g->B == 0) return LANTERN_OK;
    DynPrepArgs a{(const uint16_t *)g->cond, (const uint16_t *)g->uncond, g->V, g->cfg, g->pos_base, g->w_latent, g->h_latent, g->img_lo, g->img_hi,
                  g->newline_id, g->eos_id, g->top_k, g->N, g->win_lo, g->win_len, g->out_win, g->row_hot, g->node_list, g->n_list, g->B,
                  TdArgs{d.scores, d.tokens, d.parents, g->sample_token, d.n_scores, d.n_parents, d.top_k, d.total_tokens, d.sort_rows, d.draft_tokens, d.mask,
                         d.pos_ids, d.retrieve, d.n_leaf, d.max_depth, TdCand{d.seq_len, g->cand, d.retrieve_pd, d.pos_abs, d.row_index, g->P, g->D}},
                  g->top_p};
    const bool nucleus = g->top_p >= 1e-8f && g->top_p < 1.0f;
    const dim3 grid(g->B * g->n_list + g->B);
    if (g->win_len == 16384) {          // LlamaGen: the whole vocabulary
        if (nucleus) LANTERN_LAUNCH((dyn_prep_kernel<512, 4, true>), grid, dim3(512), 0, (hipStream_t)g->stream, a);
        else LANTERN_LAUNCH((dyn_prep_kernel<512, 4>), grid, dim3(512), 0, (hipStream_t)g->stream, a);
    } else if (nucleus) LANTERN_LAUNCH((dyn_prep_kernel<512, 2, true>), grid, dim3(512), 0, (hipStream_t)g->stream, a);
    else LANTERN_LAUNCH((dyn_prep_kernel<512, 2>), grid, dim3(512), 0, (hipStream_t)g->stream, a);
    LANTERN_CHECK_LAUNCH("prepare_step");
    return LANTERN_OK;
}

// validation + argument block of the static-tree form of lantern_prepare_step (also used by the commit launch that prepares the next step)
namespace lantern {
int prepare_step_args(const lantern_step_group *g, PrepArgs *out) {
    LANTERN_CHECK_ARG(g && g->node_list && g->n_list > 0 && g->n_list <= g->N, "prepare_step: needs a node list");
    const bool no_grammar = g->w_latent == 0 && g->h_latent == 0;          // Anole, LlamaGen: every row an ordinary distribution
    // the three forms: Lumina (static or dynamic trees), Anole static trees on the same Chameleon image window, LlamaGen dynamic trees on its whole vocabulary
    const bool chameleon_window = g->win_len == 8192 && g->win_lo == g->img_lo && g->win_lo + g->win_len == g->img_hi && g->win_lo % 4 == 0 &&
                                  (g->model == LANTERN_MODEL_LUMINA || (g->model == LANTERN_MODEL_ANOLE && no_grammar && !g->dyn));
    const bool llamagen_rows = g->model == LANTERN_MODEL_PLAIN && no_grammar && g->dyn && g->win_lo == 0 && g->win_len == 16384 && g->V == 16384 &&
                               g->img_lo == 0 && g->img_hi == 16384;
    LANTERN_CHECK_ARG(g->cond && g->uncond && g->out_win && g->row_hot && (g->dyn || no_grammar || (g->seq_len && g->pos_ids)) && g->dtype == LANTERN_BF16 &&
                          (chameleon_window || llamagen_rows) && g->V % 8 == 0 &&
                          g->out_kind == LANTERN_ROWS_PROBS && g->temperature == 1.0f,
                      "prepare_step: bf16 rows, probability output, temperature 1: Lumina (or Anole static trees, w_latent = h_latent = 0) on the 8192-id image window, or "
                      "LlamaGen dynamic trees (LANTERN_MODEL_PLAIN, w_latent = h_latent = 0) on its 16384 ids");
    if (g->dyn) return LANTERN_OK;          // (the caller takes the dynamic form)
    LANTERN_CHECK_ARG(g->ss_token && g->sample_token && g->tree_indices && g->retrieve && g->tree_cand && g->cand && g->B >= 0 && g->n_flat > 0 && g->N > 0 &&
                          g->P > 0 && g->D > 0, "prepare_step: candidate-assembly buffers missing");
    *out = PrepArgs{(const uint16_t *)g->cond, (const uint16_t *)g->uncond, g->V, g->cfg, g->pos_ids, g->pos_base, g->w_latent, g->h_latent, g->img_lo, g->img_hi,
                    g->newline_id, g->eos_id, g->top_k, g->seq_len, g->N, g->win_lo, g->win_len, g->out_win, g->row_hot, g->node_list, g->n_list, g->B,
                    g->ss_token, g->ss_prob, g->sample_token, g->tree_indices, g->retrieve, g->n_flat, g->N, g->P * g->D, g->tree_cand, g->cand, g->cart_prob,
                    g->top_p, nullptr, nullptr};
    return LANTERN_OK;
}
}  // namespace lantern

extern "C" int lantern_prepare_step(const lantern_step_group *g) {
    PrepArgs a{};
    const int rc = lantern::prepare_step_args(g, &a);
    if (rc) return rc;
    if (g->dyn) return prepare_step_dynamic(g);
    if (g->B == 0) return LANTERN_OK;
    const int nt_knob = tuning(TUNE_PREP_NT);
    if (g->top_p >= 1e-8f && g->top_p < 1.0f) LANTERN_LAUNCH((prep_rows_kernel<512, 2, true>), dim3(g->B * g->n_list + g->B), dim3(512), 0, (hipStream_t)g->stream, a);
    else if (nt_knob == 1024) LANTERN_LAUNCH((prep_rows_kernel<1024, 1>), dim3(g->B * g->n_list + g->B), dim3(1024), 0, (hipStream_t)g->stream, a);
    else LANTERN_LAUNCH((prep_rows_kernel<512, 2>), dim3(g->B * g->n_list + g->B), dim3(512), 0, (hipStream_t)g->stream, a);
    LANTERN_CHECK_LAUNCH("prepare_step");
    return LANTERN_OK;
}

// argument rules of the windowed chain kernel
static int epw_check(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win) {
    LANTERN_CHECK_ARG(prm && buf && win, "evaluate_posterior_window: null params");
    const lantern_ep_params &p = *prm;
    LANTERN_CHECK_ARG(p.B >= 0 && p.P > 0 && p.D > 0 && p.V > 0 && p.V % 4 == 0, "evaluate_posterior_window: bad B/P/D/V");
    if (p.B == 0) return LANTERN_OK;
    LANTERN_CHECK_ARG(p.P <= EW_MAX_P && p.D <= EW_MAX_D && p.P * p.D <= EW_MAX_PD, "evaluate_posterior_window: P=%d D=%d exceed limits (64 paths: one lane per path); use the dense kernel", p.P, p.D);
    if (p.mode != LANTERN_MODE_DYNAMIC) LANTERN_CHECK_ARG(p.N <= EW_MAX_N, "evaluate_posterior_window: N=%d > %d", p.N, EW_MAX_N);
    // one uniform per tried candidate, EW_UNI of them staged per step: a tree that allows more tries than that must take the dense kernel
    // (said here, on the host, instead of a spurious LANTERN_ST_UNIFORMS from inside the launch)
    if (p.mode != LANTERN_MODE_DYNAMIC && p.N - 1 > EW_UNI) {
        set_error("evaluate_posterior_window: a tree of %d nodes allows %d tries per step, %d uniforms are staged: use the dense kernel", p.N, p.N - 1, EW_UNI);
        return LANTERN_E_UNSUPPORTED;
    }
    LANTERN_CHECK_ARG(win->win_lo >= 0 && win->win_lo % 4 == 0 && win->win_len > 0 && win->win_len % 4 == 0 &&
                          win->win_lo + win->win_len <= p.V && win->win_len <= 16384,
                      "evaluate_posterior_window: window [%d,+%d) must be 4-aligned, inside V and <= 16384 wide", win->win_lo, win->win_len);
    LANTERN_CHECK_ARG(p.n_syntax >= 0 && p.n_syntax <= 8 && p.mode >= 0 && p.mode <= 2, "evaluate_posterior_window: bad mode/n_syntax");
    LANTERN_CHECK_ARG(buf->logits && buf->row_index && buf->cand && buf->uniforms && buf->best && buf->accept_len && buf->counters,
                      "evaluate_posterior_window: null required buffer");
    if (p.mode != LANTERN_MODE_DYNAMIC)
        LANTERN_CHECK_ARG(buf->cart_prob && buf->orig_prob && buf->op_off && buf->p_idx && buf->b_off && buf->b_idx && buf->tree_cand &&
                              p.R > 0 && p.N > 0 && win->orig_prob_stride >= win->win_len && win->orig_prob_offset % 4 == 0 &&
                              win->orig_prob_stride % 4 == 0,
                          "evaluate_posterior_window: static mode needs cart_prob/orig_prob(+4-aligned stride/offset)/op_off/p_idx/b_off/b_idx/tree_cand");
    if (p.lantern)
        LANTERN_CHECK_ARG(buf->nn_table && p.k >= 1 && p.k <= p.table_cols && p.table_rows > 0, "evaluate_posterior_window: lantern needs nn_table, 1<=k<=cols");
    if (win->u_bonus) LANTERN_CHECK_ARG(win->token, "evaluate_posterior_window: u_bonus needs token");
    LANTERN_CHECK_ARG(win->rows_kind == LANTERN_ROWS_LOGITS || win->rows_kind == LANTERN_ROWS_PROBS || win->rows_kind == LANTERN_ROWS_RAW_BF16,
                      "evaluate_posterior_window: bad rows_kind");
    const bool raw = win->rows_kind == LANTERN_ROWS_RAW_BF16;
    if (raw) {
        const bool plain = win->raw_w_latent == 0 && win->raw_h_latent == 0;          // no grammar rows (LlamaGen): positions are not needed
        LANTERN_CHECK_ARG(win->raw_uncond && (plain || (win->raw_pos_ids && (win->raw_seq_len || win->raw_pos_per_seq) && win->raw_w_latent > 0 && win->raw_h_latent > 0 &&
                                                        win->raw_newline_id >= 0 && win->raw_newline_id < p.V && win->raw_eos_id >= 0 && win->raw_eos_id < p.V)),
                          "evaluate_posterior_window: raw rows need the unconditional logits and -- Lumina -- positions, sequence lengths and the grammar ids "
                          "(raw_w_latent = raw_h_latent = 0: a model without grammar rows)");
        LANTERN_CHECK_ARG(p.top_k <= 0 && p.temperature == 1.0f && p.rows_per_seq <= EW_MAX_N && win->win_lo % 4 == 0 && p.V % 8 == 0 &&
                              win->win_lo == p.img_lo && win->win_lo + win->win_len == p.img_hi,
                          "evaluate_posterior_window: raw rows: the window is the image-token range, the processors are the Lumina ones (raw_top_k)");
        const bool lumina_form = win->win_len == 8192 && p.lantern && p.table_cols % 8 == 0 && ((uintptr_t)buf->nn_table & 15) == 0 &&
                                 ((p.k + 1 < p.table_cols ? p.k + 1 : p.table_cols) <= EW_PF_K);
        const bool llamagen_form = win->win_len == 16384 && !p.lantern && plain && p.mode == LANTERN_MODE_DYNAMIC && !p.syntax_shortcut;
        if (!lumina_form && !llamagen_form) {
            set_error("evaluate_posterior_window: raw rows are built for the 8192-id window on the packed neighbour table (k + 1 <= %d) and for the "
                      "16384-id window of LlamaGen's standard verify (LANTERN off, dynamic trees)", EW_PF_K);
            return LANTERN_E_UNSUPPORTED;
        }
    }
    if (win->rows_kind == LANTERN_ROWS_PROBS)
        LANTERN_CHECK_ARG(p.top_k <= 0 && p.temperature == 1.0f, "evaluate_posterior_window: probability rows are final -- apply temperature/top-k where they are produced (cfg_mask_topk_window)");
    if (p.top_p > 0.0f && p.top_p < 1.0f && win->rows_kind == LANTERN_ROWS_PROBS) {
        set_error("evaluate_posterior_window: top_p=%g with probability rows: they are final -- apply it where the rows are produced "
                  "(lantern_cfg_mask_topk_window), or hand the rows over as logits / raw bf16 rows", (double)p.top_p);
        return LANTERN_E_UNSUPPORTED;
    }
    if (p.top_k > win->win_len && p.top_k < p.V) {
        set_error("evaluate_posterior_window: top_k=%d wider than the window (%d) needs the dense kernel", p.top_k, win->win_len);
        return LANTERN_E_UNSUPPORTED;
    }
    return LANTERN_OK;
}

static size_t epw_lds_bytes(const lantern_ep_params &p, const lantern_ep_window *win) {
    const bool nucleus_logits = win->rows_kind == LANTERN_ROWS_LOGITS && p.top_p >= 1e-8f && p.top_p < 1.0f;      // top_p_tile's 256 f64 mass bins
    return epw_shared_offset(win->win_len) + sizeof(EwShared) + (size_t)6 * epw_pd_cap(p.P, p.D) * 4 +
           (win->rows_kind == LANTERN_ROWS_RAW_BF16 ? (size_t)O7_HIST_INTS * 4 : (nucleus_logits ? (size_t)256 * 8 : 0));
}

extern "C" int lantern_evaluate_posterior_window(const lantern_ep_params *prm, const lantern_ep_buffers *buf,
                                                 const lantern_ep_window *win, void *stream) {
    const int rc = epw_check(prm, buf, win);
    if (rc) return rc;
    const lantern_ep_params &p = *prm;
    if (p.B == 0) return LANTERN_OK;
    const bool raw = win->rows_kind == LANTERN_ROWS_RAW_BF16;
    hipStream_t st = (hipStream_t)stream;
    const int W = win->win_len;
    const size_t lds = epw_lds_bytes(p, win);
    dim3 grid(p.B);
    const int nz = (p.k + 1 < p.table_cols) ? p.k + 1 : p.table_cols;
    const bool lds_ids = !p.lantern || nz <= EW_PF_K;
    const EpwArgs args{p, *buf, *win};
    const int idmode = !lds_ids ? 0 : ((p.lantern && p.table_cols % 8 == 0 && ((uintptr_t)buf->nn_table & 15) == 0) ? 2 : 1);
    const EpwLaunch L{grid, lds, st};
    // the headline shape gets its own instance (SPEC 1: mode / LANTERN / syntax-shortcut flags are compile-time constants there)
    const int spec_knob = tuning(TUNE_EPW_SPEC);          // 0 = the generic instance, 1 = no fixed tree
    const bool chameleon = spec_knob != 0 && p.lantern && p.V == 65536 && p.img_lo == 4 && p.img_hi == 8196 && p.tok_offset == 4 && p.table_rows == 8192 &&
                           win->win_lo == 4 && W == 8192 && p.rows_per_seq <= EW_MAX_N && (raw || win->rows_kind == LANTERN_ROWS_PROBS) &&
                           (!raw || win->raw_w_latent == 0 || (win->raw_eos_id == 8196 && win->raw_newline_id == 8803));
    const bool lumina_syntax = p.syntax_shortcut && p.n_syntax == 4 && p.syntax[0] == 8196 && p.syntax[1] == 8197 && p.syntax[2] == 8803 && p.syntax[3] == 8828;
    const bool lumina_static = chameleon && p.mode == LANTERN_MODE_STATIC_LUMINA && lumina_syntax && !buf->n_paths && !buf->n_depth && (!raw || !win->raw_pos_per_seq);
    const bool lumina_dynamic = chameleon && p.mode == LANTERN_MODE_DYNAMIC && lumina_syntax && buf->n_paths && buf->n_depth && (!raw || win->raw_pos_per_seq);
    const bool anole_static = chameleon && p.mode == LANTERN_MODE_STATIC_LG && !p.syntax_shortcut && !buf->n_paths && !buf->n_depth &&
                              (!raw || (win->raw_w_latent == 0 && !win->raw_pos_per_seq));
    // the fixed-tree instance is picked by SIZES (P, D, N of mc_sim_7b_63) and stages at most 4 candidates per level ahead; a tree of the same sizes
    // with a wider fan-out still runs correctly (the restage path, tests: test_window_two_workgroups_per_cu_build[six_wide])
    const bool default_tree = spec_knob >= 2 && p.P == 15 && p.D == 6 && p.N == 26 && p.rows_per_seq == 26;
    const int occ_knob = tuning(TUNE_EPW_OCC2);
    const bool many = occ_knob >= 0 ? occ_knob != 0 : p.B > 256;          // more sequences than CUs: the throughput forms (epw_throughput.hip)
    // Throughput form of the fixed-configuration instances (the shape BASELINE's roofline target is assessed on): 256 threads x 8 float4 per thread, three
    // workgroups per CU (53 KB of LDS each, <= 168 VGPRs at 3 waves per SIMD), drafter rows requested only once a rejection is known.  At saturation
    // the kernel is bound by instruction ISSUE (profiles/r04_ep_sweep_pmc.txt), so half the waves per sequence is what pays: 4096 sequences per
    // launch 293 us (generic, 512 threads, two per CU) -> 228 (fixed configuration, 512 threads) -> 194 (owner-wave sibling zeroing) -> 163 us.
    // LANTERN_EPW_TP=0: the generic two-per-CU instance; 1: the 512-thread fixed-configuration instance (diagnostic).
    const int tp_knob = tuning(TUNE_EPW_TP);
    const bool nucleus = p.top_p >= 1e-8f && p.top_p < 1.0f;
    bool ok = true;
    if (raw && nucleus) {          // raw rows with a nucleus filter: the generic raw instances with the filter compiled in
        if (W == 16384) LANTERN_LAUNCH((epw_kernel<1024, 4, 1, 1, true, true, 0, 2>), grid, dim3(1024), lds, st, args);
        else LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 0, 2>), grid, dim3(512), lds, st, args);
    }
    else if (raw && W == 16384) {          // LlamaGen standard verify on raw rows (the form epw_check admitted)
        const bool lg_dynamic = spec_knob != 0 && p.V == 16384 && p.img_lo == 0 && p.img_hi == 16384 && p.tok_offset == 0 && win->win_lo == 0 && buf->n_paths &&
                                buf->n_depth && p.rows_per_seq <= EW_MAX_N;
        if (lg_dynamic) LANTERN_LAUNCH((epw_kernel<1024, 4, 1, 1, true, true, 5>), grid, dim3(1024), lds, st, args);
        else LANTERN_LAUNCH((epw_kernel<1024, 4, 1, 1, true, true>), grid, dim3(1024), lds, st, args);
    }
    else if (raw) {
        const bool tp_raw = many && tp_knob >= 5;
        if (tp_raw && lumina_static && default_tree) ok = epw_launch_throughput(EPW_TP_RAW_LUMINA_DEFAULT_TREE, L, args);
        else if (tp_raw && lumina_static) ok = epw_launch_throughput(EPW_TP_RAW_LUMINA_STATIC, L, args);
        else if (tp_raw && lumina_dynamic) ok = epw_launch_throughput(EPW_TP_RAW_LUMINA_DYNAMIC, L, args);
        else if (tp_raw && anole_static) ok = epw_launch_throughput(EPW_TP_RAW_ANOLE_STATIC, L, args);
        else if (many) ok = epw_launch_throughput(EPW_TP_RAW_GENERIC, L, args);
        else if (lumina_static && default_tree) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 2>), grid, dim3(512), lds, st, args);
        else if (lumina_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 1>), grid, dim3(512), lds, st, args);
        else if (lumina_dynamic) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 3>), grid, dim3(512), lds, st, args);
        else if (anole_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 4>), grid, dim3(512), lds, st, args);
        else LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true>), grid, dim3(512), lds, st, args);
    }
    else if (nucleus) ok = epw_launch_generic(W, idmode, true, L, args);          // rows that arrive as logits, TopPLogitsWarper in the kernel
    else if (W > 4096 && W <= 8192) {
        const bool tp_form = many && W == 8192 && idmode == 2 && tp_knob >= 5;
        if (tp_form && lumina_static && default_tree) ok = epw_launch_throughput(EPW_TP_LUMINA_DEFAULT_TREE, L, args);
        else if (tp_form && lumina_static) ok = epw_launch_throughput(EPW_TP_LUMINA_STATIC, L, args);
        else if (tp_form && lumina_dynamic) ok = epw_launch_throughput(EPW_TP_LUMINA_DYNAMIC, L, args);
        else if (tp_form && anole_static) ok = epw_launch_throughput(EPW_TP_ANOLE_STATIC, L, args);
        else if (many && W == 8192 && idmode == 2 && lumina_static && default_tree && tp_knob >= 1) ok = epw_launch_throughput(EPW_TP_512_DEFAULT_TREE, L, args);
        else if (many && W == 8192 && idmode == 2) ok = epw_launch_throughput(EPW_TP_512_PACKED, L, args);
        else if (many) ok = epw_launch_throughput(idmode == 2 ? EPW_TP_512_ID2 : (idmode == 1 ? EPW_TP_512_ID1 : EPW_TP_512_ID0), L, args);
        else if (W == 8192 && idmode == 2) {
            if (lumina_static && default_tree) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 2>), grid, dim3(512), lds, st, args);
            else if (lumina_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 1>), grid, dim3(512), lds, st, args);
            else if (lumina_dynamic) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 3>), grid, dim3(512), lds, st, args);
            else if (anole_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 4>), grid, dim3(512), lds, st, args);
            else LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true>), grid, dim3(512), lds, st, args);   // the Lumina / Anole image window on the packed table
        }
        else ok = epw_launch_generic(W, idmode, false, L, args);
    }
    else if (W == 16384 && many && tuning(TUNE_EPW_TP_LG) != 0 && spec_knob != 0 && !p.lantern && p.mode == LANTERN_MODE_DYNAMIC && !p.syntax_shortcut && p.V == 16384 &&
             p.img_lo == 0 && p.img_hi >= 16384 && p.tok_offset == 0 && win->win_lo == 0 && buf->n_paths && buf->n_depth && win->rows_kind == LANTERN_ROWS_PROBS &&
             p.rows_per_seq <= EwSharedLite::kMaxN &&
             epw_shared_offset(W, false) + sizeof(EwSharedLite) + (size_t)6 * epw_pd_cap(p.P, p.D) * 4 <= (size_t)80 * 1024) {
        // LlamaGen's standard verify at more sequences than CUs: two workgroups per CU (epw_throughput.hip)
        const EpwLaunch Llg{grid, epw_shared_offset(W, false) + sizeof(EwSharedLite) + (size_t)6 * epw_pd_cap(p.P, p.D) * 4, st};
        ok = epw_launch_throughput(EPW_TP_LLAMAGEN_DYNAMIC, Llg, args);
    }
    else ok = epw_launch_generic(W, idmode, false, L, args);
    if (!ok) {
        set_error("evaluate_posterior_window: no kernel instance for window %d, id mode %d%s", W, idmode, nucleus ? ", top_p inside the kernel" : "");
        return LANTERN_E_UNSUPPORTED;
    }
    LANTERN_CHECK_LAUNCH("evaluate_posterior_window");
    return LANTERN_OK;
}

// The chain launch with its prepare stage inside (lantern_step_group flags & LANTERN_STEP_FUSED_PREPARE, verify_step.cpp): the static-tree latency instances (Lumina,
// Anole) on raw rows, at most 256 sequences.  Anything else is refused -- the caller asked for a form that does not exist, it is not quietly run in three launches.
namespace lantern {
int evaluate_posterior_window_fused(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win, const PrepArgs &prep, int32_t *ready,
                                    int32_t epoch, void *stream) {
    const int rc = epw_check(prm, buf, win);
    if (rc) return rc;
    const lantern_ep_params &p = *prm;
    if (p.B == 0) return LANTERN_OK;
    const int W = win->win_len;
    const bool raw = win->rows_kind == LANTERN_ROWS_RAW_BF16;
    const bool nucleus = p.top_p >= 1e-8f && p.top_p < 1.0f;
    const bool lumina_syntax = p.syntax_shortcut && p.n_syntax == 4 && p.syntax[0] == 8196 && p.syntax[1] == 8197 && p.syntax[2] == 8803 && p.syntax[3] == 8828;
    // the Chameleon image window on the packed table, one tree for all sequences (the conditions under which lantern_evaluate_posterior_window takes a SPEC instance)
    const bool chameleon = raw && !nucleus && p.lantern && p.V == 65536 && p.img_lo == 4 && p.img_hi == 8196 && p.tok_offset == 4 && p.table_rows == 8192 && win->win_lo == 4 &&
                           W == 8192 && p.rows_per_seq <= EW_MAX_N && !buf->n_paths && !buf->n_depth && !win->raw_pos_per_seq && p.table_cols % 8 == 0 &&
                           ((uintptr_t)buf->nn_table & 15) == 0 && ((p.k + 1 < p.table_cols ? p.k + 1 : p.table_cols) <= EW_PF_K) && p.B <= 256;
    const bool lumina = chameleon && p.mode == LANTERN_MODE_STATIC_LUMINA && lumina_syntax && (win->raw_w_latent == 0 || (win->raw_eos_id == 8196 && win->raw_newline_id == 8803));
    const bool anole = chameleon && p.mode == LANTERN_MODE_STATIC_LG && !p.syntax_shortcut && win->raw_w_latent == 0;
    const bool form = lumina || anole;
    if (!form || !ready || !win->raw_probs || !win->raw_pre || prep.n_list < 1 || prep.B != p.B || prep.PD != p.P * p.D || prep.N != p.N || prep.rows_per_seq != p.rows_per_seq ||
        prep.out_win != win->raw_probs || prep.W != W || !prep.cand || !prep.tree_cand || (prep.top_p >= 1e-8f && prep.top_p < 1.0f)) {
        set_error("fused prepare: Lumina or Anole static trees on raw bf16 rows (8192-id window, packed table, k + 1 <= %d, top_p off), at most 256 sequences, a node list "
                  "whose rows go to ep_win.raw_probs, and the row_ready words", EW_PF_K);
        return LANTERN_E_UNSUPPORTED;
    }
    const size_t lds = epw_lds_bytes(p, win);
    const EpwArgs args{p, *buf, *win};
    int per = tuning(TUNE_EPW_FUSED_HELPERS);          // helper workgroups per sequence (each takes its listed rows one after the other)
    if (per < 1) per = 1;
    if (per > prep.n_list - 1) per = prep.n_list - 1;          // (a list of the root alone: no helper, the sequences post-process every row themselves)
    const EpwFused fz{prep, ready, epoch, p.B * per};
    const dim3 grid(fz.n_helpers + p.B);
    const bool default_tree = tuning(TUNE_EPW_SPEC) >= 2 && p.P == 15 && p.D == 6 && p.N == 26 && p.rows_per_seq == 26;
    hipStream_t st = (hipStream_t)stream;
    if (anole) LANTERN_LAUNCH((epw_kernel_fused<512, 4, 2, 1, true, true, 4, 0>), grid, dim3(512), lds, st, args, fz);
    else if (default_tree) LANTERN_LAUNCH((epw_kernel_fused<512, 4, 2, 1, true, true, 2, 0>), grid, dim3(512), lds, st, args, fz);
    else LANTERN_LAUNCH((epw_kernel_fused<512, 4, 2, 1, true, true, 1, 0>), grid, dim3(512), lds, st, args, fz);
    LANTERN_CHECK_LAUNCH("evaluate_posterior_window (fused prepare)");
    return LANTERN_OK;
}
}  // namespace lantern

#ifdef EPW_TRACE
// host_out: [EPW_TR_BLOCKS][EPW_TR_MAX] stamps (id << 56 | cycles), counts: [EPW_TR_BLOCKS]
extern "C" int lantern_debug_epw_trace(unsigned long long *host_out, int *counts) {
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_epw_trace_n), sizeof(int) * EPW_TR_BLOCKS) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_epw_trace), sizeof(unsigned long long) * EPW_TR_BLOCKS * EPW_TR_MAX) != hipSuccess) return -1;
    return EPW_TR_MAX;
}
#endif

extern "C" int lantern_window_to_dense(const float *winp, const int32_t *out_tok, const float *out_mass, int B, int V, int win_lo,
                                       int win_len, float *dense, void *stream) {
    LANTERN_CHECK_ARG(winp && dense && B >= 0 && V > 0 && V % 4 == 0 && win_lo % 4 == 0 && win_len % 4 == 0 && win_lo + win_len <= V,
                      "window_to_dense: bad arguments");
    if (B == 0) return LANTERN_OK;
    int gx = (V / 4 + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(window_to_dense_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, winp, out_tok, out_mass, V, win_lo, win_len, dense);
    LANTERN_CHECK_LAUNCH("window_to_dense");
    return LANTERN_OK;
}
