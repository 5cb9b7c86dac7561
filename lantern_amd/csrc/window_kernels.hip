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
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        float4 v = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        if (i4 * 4 < W) {
            const int e = win_lo + i4 * 4;
            const size_t g4 = ((size_t)row * V + e) / 4;
            float c[4], u[4];
            if (BF16) {
                const ushort4 cb = reinterpret_cast<const ushort4 *>(cond_)[g4];
                const ushort4 ub = uncond_ ? reinterpret_cast<const ushort4 *>(uncond_)[g4] : cb;
                c[0] = bf16_bits_to_f32(cb.x); c[1] = bf16_bits_to_f32(cb.y); c[2] = bf16_bits_to_f32(cb.z); c[3] = bf16_bits_to_f32(cb.w);
                u[0] = bf16_bits_to_f32(ub.x); u[1] = bf16_bits_to_f32(ub.y); u[2] = bf16_bits_to_f32(ub.z); u[3] = bf16_bits_to_f32(ub.w);
            } else {
                const float4 cf = reinterpret_cast<const float4 *>(cond_)[g4];
                const float4 uf = uncond_ ? reinterpret_cast<const float4 *>(uncond_)[g4] : cf;
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

// ---- O7 windowed, bf16 logits, 16-byte loads.  Thread t owns E8 chunks of 8 consecutive window ids (chunk index
// t + it*NT), so cond and uncond arrive as one global_load_dwordx4 each per chunk (the window start need only be
// 4-aligned: the loads are then 8-byte aligned, which gfx950 global loads accept).  The first radix pass -- sign + 7 exponent bits, where a logit row
// concentrates in a handful of bins -- uses a 16-way replicated LDS histogram (copy = lane % 16: at most 4 lanes of an
// instruction on one word, window_dev.h); the second pass (7 mantissa bits + 1 exponent bit inside the chosen bin) is spread
// out by nature and uses a single copy.
// One row of the windowed O7 on a workgroup: CFG combination, top-k threshold, softmax, window store (the body of cfg_window_bf16_kernel,
// shared with the merged launch below).  `cls`: 0 = grid row, 1 = forced newline, 2 = forced end of image.
template <int NT, int E8, bool FULL, bool NUCLEUS = false>
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
#pragma unroll
    for (int it = 0; it < E8; ++it) {
        const int ch = tid + it * NT;
        const bool in = FULL || ch * 8 < W;       // FULL: W == 8 * NT * E8, every chunk is inside the window
        cb[it] = in ? *reinterpret_cast<const Bf16x8 *>(crow + ch * 8) : Bf16x8{make_uint2(0, 0), make_uint2(0, 0)};
        ub[it] = (in && urow) ? *reinterpret_cast<const Bf16x8 *>(urow + ch * 8) : cb[it];
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
            top_p_tile<NT, 2 * E8>(r, top_p, reinterpret_cast<double *>(s_hist), s_redf, s_redd, s_redi, php);
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
        if (FULL || (w0 >= 0 && w0 < W)) *reinterpret_cast<float4 *>(out + w0) = r[2 * it];
        if (FULL || (w0 + 4 >= 0 && w0 + 4 < W)) *reinterpret_cast<float4 *>(out + w0 + 4) = r[2 * it + 1];
    }
}

__device__ __forceinline__ int lumina_row_class(int64_t pos, int64_t pos_base, int w_latent, int h_latent) {
    const int64_t n1 = pos - pos_base + 1;
    if (n1 == ((int64_t)w_latent + 1) * h_latent + 1) return 2;
    // (a 64-bit modulo is ~150 scalar instructions that every wave of the workgroup repeats: the 32-bit form whenever it applies)
    if (n1 >= 0 && n1 < (1ll << 31) && w_latent >= 0 && w_latent < (1 << 30)) return ((uint32_t)n1 % (uint32_t)(w_latent + 1)) == 0 ? 1 : 0;
    return py_mod64(n1, (int64_t)w_latent + 1) == 0 ? 1 : 0;
}

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
};

template <int NT, int E8, bool NUCLEUS = false>
__global__ __launch_bounds__(NT) void prep_rows_kernel(const PrepArgs a) {
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[32];
    __shared__ double s_redd[32];
    __shared__ int s_redi[32];
    const int n_rows = a.B * a.n_list;
    if ((int)blockIdx.x < n_rows) {
        const int x = o7_row_of_block(blockIdx.x, n_rows, a.n_list);            // rows of sequence b on XCD b % 8, where its chain runs
        const int b = x / a.n_list, node = a.node_list[x % a.n_list];
        const int row = b * a.rows_per_seq + node;
        // (w_latent == 0: a model without grammar rows -- Anole; the window is its image-token range, so no id needs the model's mask)
        const int cls = a.w_latent > 0 ? lumina_row_class(a.pos_ids[node] + a.seq_len[b], a.pos_base, a.w_latent, a.h_latent) : 0;
        cfg_window_bf16_row<NT, E8, true, NUCLEUS>(row, cls, a.cond, a.uncond, a.V, a.cfg, LANTERN_MODEL_LUMINA, a.img_lo, a.img_hi, a.newline_id, a.eos_id,
                                                   a.top_k, a.win_lo, a.W, a.out_win, a.row_hot, LANTERN_ROWS_PROBS, s_hist, s_redf, s_redd, a.top_p, s_redi);
        return;
    }
    // ---- candidate assembly of sequence b (same arithmetic as gather_candidates_kernel)
    const int b = blockIdx.x - n_rows, N = a.N, PD = a.PD, n_flat = a.n_flat;
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

// ------------------------------------------------------------------------------- O8 windowed
//
// Structure (v3).  A 512-thread workgroup owns one sequence.  The serial part of the algorithm -- walking the
// candidates of a level, the k-neighbour cumulative-mass scan, the accept test -- is executed by WAVE 0 ONLY
// (one lane per path for the prefix masks, 16 neighbours per lane for the scan, DPP scans); the other waves
// wait at a barrier and join for the W-wide passes (row softmax, residual update, renormalisation), whose
// cross-wave reductions combine <= 16 partials with one DPP row.  All decisions travel through one LDS word,
// so control flow stays workgroup-uniform.  The neighbour ids of every candidate of a level are fetched in one
// round at level start (they depend only on the accepted prefix), so the per-candidate work is LDS + ALU only.

template <int NT, int E4, bool FULLW = false, typename Hook = NoHook>
__device__ __forceinline__ void row_softmax_to_lds(float4 (&r)[E4], int hot, bool probs, int win_lo, int W, float temperature, int top_k,
                                                   int V, float *g, int &out_tok, float &out_mass, EwShared &S, int &ph,
                                                   const Hook &pre_barrier = Hook()) {
    const int tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    out_tok = -1;
    out_mass = 0.0f;
    if (hot >= 0) {
        const bool inside = hot >= win_lo && hot < win_lo + W;
        for (int i4 = tid; i4 * 4 < W; i4 += NT) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e = win_lo + i4 * 4;
            if (hot >= e && hot < e + 4) set_comp(v, hot - e, 1.0f);
            reinterpret_cast<float4 *>(g)[i4] = v;
        }
        if (!inside) {
            out_tok = hot;
            out_mass = 1.0f;
        }
        if (tid == 0) g[W + EW_G_OUT] = out_mass;
        pre_barrier();
        __syncthreads();
        return;
    }
    if (!probs) {       // rows arrive as logits: processors + softmax here; LANTERN_ROWS_PROBS rows are final (O7 did both)
        if (temperature > 1e-5f && temperature != 1.0f) {
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                r[it].x = r[it].x / temperature; r[it].y = r[it].y / temperature;
                r[it].z = r[it].z / temperature; r[it].w = r[it].w / temperature;
            }
        }
        if (top_k > 0 && top_k < V && top_k <= W) {
            const float thr = kth_largest_tile<NT, E4, false>(r, top_k, S.redi, ph);
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                r[it].x = r[it].x < thr ? NEG_INF : r[it].x; r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
                r[it].z = r[it].z < thr ? NEG_INF : r[it].z; r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
            }
        }
        softmax_tile<NT, E4>(r, S.redf, S.redd, ph);
    }
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(g)[i4] = r[it];
    }
    if (tid == 0) g[W + EW_G_OUT] = 0.0f;
    pre_barrier();
    __syncthreads();
}

// LDSIDS: every candidate's neighbour ids are staged in LDS (k + 1 <= EW_PF_K, or LANTERN off), so the serial wave-0
// section contains no vector-memory instruction -- the compiler then has no reason to drain vmcnt inside it and the
// drafter-row / id loads issued before it stay in flight across the scan.  !LDSIDS (k > 1023) reads ids from HBM.
//
// Scalar registers are the scarce resource of this kernel (three parameter blocks + the walk's state): everything the
// epilogue alone needs (output pointers, the bonus-draw inputs) is re-read from the kernarg segment there instead of
// being held in SGPRs across the whole walk.
struct EpwArgs {
    lantern_ep_params prm;
    lantern_ep_buffers buf;
    lantern_ep_window win;
};
typedef const __attribute__((address_space(4))) EpwArgs *EpwArgsK;

// IDMODE 0: ids from HBM in the scan (k > 1023).  1: ids staged in LDS, table rows at any (2-byte) alignment -- the
// reference's [K, K-1] layout.  2: ids staged in LDS from a table whose row stride is a multiple of 8 ids and whose base
// is 16-byte aligned (lantern_pack_vq_table): one 16-byte load brings 8 ids, 2 loads per thread cover a whole level.
// WPE (waves per SIMD the register allocation must allow): 1 = no constraint -- the latency-optimal build (162 VGPRs, one
// workgroup per CU) used while every sequence of the launch gets a CU to itself; 4 (512-thread workgroups) = 128 VGPRs so
// that TWO workgroups share a CU once the batch exceeds the CU count (67 KB of LDS each): +41 % sequences/s at saturation,
// -7 % at 48 sequences (a few spills), hence selected by batch size.
// FULLW: the window is exactly the workgroup's register tile (W == 4 * NT * E4, the Lumina / Anole 8192-id image range on
// 512 x 4): no per-chunk bounds predicate, so the four chunks of a pass are one basic block and their LDS reads go out together.
// RAW: rows are the target model's raw cond / uncond bf16 logits (LANTERN_ROWS_RAW_BF16; W == 8 * 2 * NT, packed table).
template <int NT, int E4, int IDMODE, int WPE, bool FULLW = false, bool RAW = false, int SPEC = 0, int TPO = 0>
__device__ __forceinline__ int epw_body(const EpwArgs &args, const int b) {
    constexpr bool NUCLEUS = (TPO & 2) != 0;         // raw rows: TopPLogitsWarper (prm.top_p) in front of the top-k of the rows the walk post-processes
    constexpr bool LATE_Q = (TPO & 1) != 0;          // throughput builds: a candidate's drafter row is requested once its rejection is known (an accepted
                                                     // candidate -- 0.65 of the first tries -- then costs no row request at all; the latency is another workgroup's problem)
    static_assert(!RAW || (FULLW && E4 == 4), "raw rows: the 8192-id window on 512 threads");
    constexpr bool LDSIDS = IDMODE != 0;
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    constexpr int NW = NT / 64;
    // one dynamic LDS region (16-byte aligned base): [ g : W f32 | nbmask : W bits | EwShared ]
    extern __shared__ float4 dyn_lds[];
    float *g = reinterpret_cast<float *>(dyn_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Compile-time instances of the reference's configurations (SPEC 0: everything from the argument block).  They all share the Chameleon
    // vocabulary of Lumina-mGPT / Anole (V = 65536, image ids = window = [4, 8196), table offset 4, 8192 table rows):
    //   1  Lumina static tree (LANTERN_MODE_STATIC_LUMINA), LANTERN on, syntax shortcut with Lumina's four syntax ids
    //   2  = 1 + the reference's default Lumina tree mc_sim_7b_63 (26 nodes, 15 paths of depth <= 6; run.sh / generate_images.py)
    //   3  Lumina dynamic (EAGLE-2) trees: LANTERN_MODE_DYNAMIC, LANTERN on, syntax shortcut, per-sequence paths / depths / positions
    //   4  Anole static tree (LANTERN_MODE_STATIC_LG: the neighbour set zeroes q), LANTERN on, no syntax shortcut
    //   5  LlamaGen dynamic (EAGLE-2) trees, standard verify (BASELINE config 2): LANTERN off, no syntax shortcut, V = window = 16384 ids from 0
    // The mode / flag tests below fold away, and with them the scalar registers that carried them through the whole walk; the host
    // dispatches to an instance only when the argument block says exactly that.
    constexpr bool SL = SPEC >= 1;                               // one of the fixed configurations
    constexpr bool S_STATIC = SPEC == 1 || SPEC == 2 || SPEC == 4, S_DYN = SPEC == 3 || SPEC == 5, S_SYN = SPEC >= 1 && SPEC <= 3;
    constexpr bool S_LG = SPEC == 5;                             // LlamaGen's vocabulary instead of Chameleon's
    const int Ps = (SPEC == 2) ? 15 : prm.P, Ds = (SPEC == 2) ? 6 : prm.D, V = SL ? (S_LG ? 16384 : 65536) : prm.V, W = SL ? (S_LG ? 16384 : 8192) : win.win_len,
              lo = SL ? (S_LG ? 0 : 4) : win.win_lo;
    uint32_t *nbmask = reinterpret_cast<uint32_t *>(g + W + EW_G_EXT);  // W bits: neighbour set (static LlamaGen/Anole: zeroing hits q)
    EwShared &S = *reinterpret_cast<EwShared *>(reinterpret_cast<char *>(g) + epw_shared_offset(W));
    int *const Scand = reinterpret_cast<int *>(reinterpret_cast<char *>(&S) + sizeof(EwShared));
    const int pd_cap = epw_pd_cap(Ps, Ds);
    int *const Srow = Scand + pd_cap, *const Spidx = Srow + pd_cap, *const Sboff = Spidx + pd_cap;
    float *const Scart = reinterpret_cast<float *>(Sboff + pd_cap);
    int *const Sflag = reinterpret_cast<int *>(Scart + pd_cap);       // per (path, depth): bit 1 image token, bit 0 syntax token
    int *const Shist = Sflag + pd_cap;                                // RAW: the radix-select histograms of the row post-process
    const int k = prm.k, off = SL ? (S_LG ? 0 : 4) : prm.tok_offset;
    const int p_mode = !SL ? prm.mode : (SPEC == 4 ? (int)LANTERN_MODE_STATIC_LG : (S_DYN ? (int)LANTERN_MODE_DYNAMIC : (int)LANTERN_MODE_STATIC_LUMINA));
    const bool p_lantern = SL ? !S_LG : prm.lantern != 0;
    const bool p_syntax = SL ? S_SYN : prm.syntax_shortcut != 0;
    const int p_nsyn = SL ? (S_SYN ? 4 : 0) : prm.n_syntax;
    const int p_rows = (SPEC == 2) ? 26 : prm.rows_per_seq, p_N = (SPEC == 2) ? 26 : prm.N;
    const int p_img_lo = SL ? (S_LG ? 0 : 4) : prm.img_lo, p_img_hi = SL ? (S_LG ? 16384 : 8196) : prm.img_hi, p_trows = SL ? (S_LG ? 0 : 8192) : prm.table_rows;
    auto p_syn = [&](int q) -> int { return SL ? (q == 0 ? 8196 : (q == 1 ? 8197 : (q == 2 ? 8803 : 8828))) : prm.syntax[q]; };
    const bool is_static = SL ? S_STATIC : p_mode != LANTERN_MODE_DYNAMIC;
    const int P = S_DYN ? buf.n_paths[b] : ((!SL && buf.n_paths) ? buf.n_paths[b] : Ps);
    const int D = S_DYN ? buf.n_depth[b] : ((!SL && buf.n_depth) ? buf.n_depth[b] : Ds);
    const float NEG_INF = -__builtin_inff();
    const int nz = (k + 1 < prm.table_cols) ? k + 1 : prm.table_cols;   // ids touched per candidate (k summed, k+1 zeroed)
    const bool can_prefetch = LDSIDS && p_lantern;          // (SPEC 1: true at compile time)
    const bool hot_in_lds = SL ? true : p_rows <= EW_MAX_N;
    const bool rows_probs = (SL && !RAW) ? true : win.rows_kind == LANTERN_ROWS_PROBS;
    int ph = 0;
#ifdef EPW_TRACE
    if (tid == 0) s_epw_trn = 0;
    EPW_STAMP(0);
#endif

    // ---- stage every small per-step table in LDS: two rounds of global loads (everything independent first, then what
    // needs the uniform cursor / the sibling count / the first row id), all issued before the first wait
    const float *logits = buf.logits + (size_t)b * p_rows * W;
    const uint16_t *raw_c = RAW ? reinterpret_cast<const uint16_t *>(buf.logits) + (size_t)b * p_rows * V + lo : nullptr;
    const uint16_t *raw_u = RAW ? reinterpret_cast<const uint16_t *>(win.raw_uncond) + (size_t)b * p_rows * V + lo : nullptr;
    const float *raw_p = (RAW && win.raw_probs) ? win.raw_probs + (size_t)b * p_rows * W : nullptr;
    const bool root_pre = RAW && raw_p && win.raw_pre && win.raw_pre[0] != 0;     // (the level-1 row is requested before the tables are staged)
    bool rp_probs = false;       // what rp holds: probabilities of a pre-processed row, or raw cond / uncond chunks
    const int32_t *hot_g = win.row_hot ? win.row_hot + (size_t)b * p_rows : nullptr;
    const int ucur0 = buf.cursor ? buf.cursor[b] : 0;
    float4 rp[E4];              // prefetched row (registers) and the row id it holds
    int rp_rid = -1;
    {
        constexpr int PD_PER = (EW_MAX_PD + NT - 1) / NT, B_PER = (EW_MAX_B + NT - 1) / NT, N_PER = (EW_MAX_N + NT - 1) / NT;
        const int npd = Ps * Ds;
        const int64_t *cand_g = buf.cand + (size_t)b * npd;
        const int32_t *row_g = buf.row_index + (prm.row_index_per_seq ? (size_t)b * npd : 0);
        const int nb_total = is_static ? buf.b_off[npd] : 0;
        const int rid1 = row_g[0];          // level 1: every path shares the root, the first matching path is path 0
        int64_t c_[PD_PER];
        int r_[PD_PER], pi_[PD_PER], bo_[PD_PER], tc_[N_PER], hot_[N_PER], oo_ = 0;
        float ct_[PD_PER];
#pragma unroll
        for (int u = 0; u < PD_PER; ++u) {
            const int t = tid + u * NT;
            const bool in = t < npd;
            c_[u] = in ? cand_g[t] : 0;
            r_[u] = in ? row_g[t] : 0;
            ct_[u] = (in && is_static) ? buf.cart_prob[(size_t)b * npd + t] : 0.0f;
            pi_[u] = (in && is_static) ? buf.p_idx[t] : 0;
            bo_[u] = (in && is_static) ? buf.b_off[t] : 0;
        }
#pragma unroll
        for (int u = 0; u < N_PER; ++u) {
            const int t = tid + u * NT;
            tc_[u] = (is_static && t < p_N && t < EW_MAX_N) ? (int)buf.tree_cand[(size_t)b * p_N + t] : 0;
            hot_[u] = (!RAW && hot_g && hot_in_lds && t < p_rows) ? hot_g[t] : -1;          // (raw rows: the class comes from the position, below; row_hot is not read)
            if (RAW && t < p_rows) {
                // raw_pre[t] = 1 + the depth the row was prepared for; with per-sequence trees the node has to sit there (its position says so)
                int pre = (win.raw_pre && win.raw_probs) ? (int)win.raw_pre[t] : 0;
                if (pre && (SL ? (int)S_DYN : win.raw_pos_per_seq)) {
                    const int64_t *pp = win.raw_pos_ids + (size_t)b * p_rows;
                    if (pp[t] - pp[0] != pre - 1) pre = 0;
                }
                S.pre[t] = pre;
            }
            // the row's class from its position (MultiModalLogitsProcessor, ea_model_lumina_mgpt.py:45-86); raw_w_latent == 0: a model without
            // grammar rows (LlamaGen: every row is an ordinary distribution)
            if (RAW && !S_LG && t < p_rows && win.raw_w_latent > 0) {
                const int64_t n1 = ((SL ? (int)S_DYN : win.raw_pos_per_seq) ? win.raw_pos_ids[(size_t)b * p_rows + t] : win.raw_pos_ids[t] + win.raw_seq_len[b]) - win.raw_pos_base + 1;
                // (a 64-bit modulo is ~150 instructions: the 32-bit form whenever the operands fit -- always, for real image sizes)
                const bool fits = n1 >= 0 && n1 < (1ll << 31) && win.raw_w_latent >= 0 && win.raw_w_latent < (1 << 30);
                const bool nl = (fits ? ((uint32_t)n1 % (uint32_t)(win.raw_w_latent + 1)) == 0u : py_mod64(n1, (int64_t)win.raw_w_latent + 1) == 0);
                hot_[u] = (n1 == ((int64_t)win.raw_w_latent + 1) * win.raw_h_latent + 1) ? (SL ? 8196 : win.raw_eos_id)
                          : (nl ? (SL ? 8803 : win.raw_newline_id) : -1);
            }
        }
        if (is_static && tid < Ds - 1) oo_ = buf.op_off[tid];
        double ub_ = 0.0;                   // read by the epilogue from LDS: a global load there sits on the chain with its full latency
        if (tid == 0 && win.u_bonus) ub_ = win.u_bonus[b];
        // round 2
        const double *uni = buf.uniforms + (size_t)b * prm.n_uniforms;
        double un_ = 2.0;                   // never drawn: guarded below
        if (tid < EW_UNI && ucur0 + tid < prm.n_uniforms) un_ = uni[ucur0 + tid];
        int bi_[B_PER];
#pragma unroll
        for (int u = 0; u < B_PER; ++u) {
            const int t = tid + u * NT;
            bi_[u] = (t < nb_total && t < EW_MAX_B) ? buf.b_idx[t] : 0;
        }
        if (rid1 >= 0 && rid1 < p_rows) {
            if constexpr (RAW) {
                rp_probs = root_pre && rid1 == 0;
                if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid1 * W, W, rp);
                else raw_row_load<NT>(raw_c + (size_t)rid1 * V, raw_u + (size_t)rid1 * V, rp);
            } else row_load<NT, E4, FULLW>(logits + (size_t)rid1 * W, W, rp);
            rp_rid = rid1;
        }
        // LDS stores
#pragma unroll
        for (int u = 0; u < PD_PER; ++u) {
            const int t = tid + u * NT;
            if (t < npd) {
                const int tok = (int)c_[u];
                int fl = (tok >= p_img_lo && tok < p_img_hi) ? 2 : 0;
                if (p_syntax)
                    for (int q = 0; q < p_nsyn; ++q) fl |= (tok == p_syn(q)) ? 1 : 0;
                Sflag[t] = fl;
                Scand[t] = tok;
                Srow[t] = r_[u];
                if (is_static) {
                    Scart[t] = ct_[u];
                    Spidx[t] = pi_[u];
                    Sboff[t] = bo_[u];
                }
            }
        }
        if (is_static) {
            if (tid == 0) Sboff[npd] = nb_total;
#pragma unroll
            for (int u = 0; u < B_PER; ++u) {
                const int t = tid + u * NT;
                if (t < nb_total && t < EW_MAX_B) S.bidx[t] = (unsigned short)bi_[u];
            }
#pragma unroll
            for (int u = 0; u < N_PER; ++u) {
                const int t = tid + u * NT;
                if (t < p_N && t < EW_MAX_N) S.tcand[t] = tc_[u];
            }
            if (tid < Ds - 1) S.opoff[tid] = oo_;
        }
        if ((hot_g || RAW) && hot_in_lds) {
#pragma unroll
            for (int u = 0; u < N_PER; ++u) {
                const int t = tid + u * NT;
                if (t < p_rows) S.hot[t] = hot_[u];
            }
        }
        if (tid < EW_UNI) S.uni[tid] = un_;
        if (tid == 0) {
            g[W + EW_G_ZERO] = 0.0f;
            g[W + EW_G_HUGE] = 3.0e38f;
            g[W + EW_G_OUT] = 0.0f;
            S.ubonus[0] = ub_;
        }
    }
    EPW_STAMP(1);
    __syncthreads();
    EPW_STAMP(2);
    // paths sharing the root token (the reference compares candidates[:, :1] with candidates[0, :1])
    unsigned long long eq_mask = __ballot(lane < P && Scand[(lane < P ? lane : 0) * Ds] == Scand[0]);

    int a = 1, best = 0, adjust = 0, status = LANTERN_ST_OK;
    int n_levels = 0, n_tried = 0, n_rej = 0, n_used = 0;
    int out_tok = -1;
    float out_mass = 0.0f;

    for (int i = 1; i < D && status == LANTERN_ST_OK; ++i) {
        if (i != a) break;
        adjust = 0;
        ++n_levels;
        // prefix mask, one lane per path (P <= 64; every wave holds the same mask): kept across levels -- a path matches
        // the accepted prefix of length a iff it matched at a-1 and carries the token accepted there
        if (eq_mask == 0ull) {
            status = LANTERN_ST_NO_PREFIX;
            break;
        }
        EPW_STAMPG(16);
        const int fi = __ffsll((long long)eq_mask) - 1;
        // everything a candidate needs from its path at this level, one lane per path, fetched once per level
        const int pl = (lane < P ? lane : 0) * Ds + i;
        const int x_lane = (lane < P) ? Scand[pl] : -1;
        float cart_lane = 1.0f;
        int qrow_lane = 0, b0_lane = 0, b1_lane = 0;
        if (is_static) {
            cart_lane = Scart[pl];
            qrow_lane = S.opoff[i - 1] + Spidx[pl];
            b0_lane = Sboff[pl];
            b1_lane = Sboff[pl + 1];
        }
        const int flag_lane = (lane < P) ? Sflag[pl] : 0;     // bit 1: image token, bit 0: syntax token (classified once, at staging)
        const unsigned long long todo0 = eq_mask & __ballot(x_lane != -1);
        EPW_STAMPG(17);
        // neighbour ids of the level's candidates: their HBM reads are issued first, the row's loads second; both are in
        // flight together and the ids are written to LDS under the row's last barrier (one exposed latency per level)
        constexpr int PF_PER = (EW_PF_K + NT - 1) / NT;
        constexpr int PF_LIST = SPEC == 2 ? 4 : EW_PF_C;          // candidates per level staged ahead
        // position t of a neighbour list -> index into g for the scan (out_tok is final before the ids are staged)
        auto plain_addr = [&](int id) -> unsigned short {
            const int e = id + off;
            if (e >= lo && e < lo + W) return (unsigned short)(e - lo);
            return (unsigned short)(W + (e == out_tok ? EW_G_OUT : EW_G_ZERO));
        };
        auto gather_addr = [&](int id, int t) -> unsigned short { return t >= k ? (unsigned short)(W + EW_G_HUGE) : plain_addr(id); };
        unsigned short idv[PF_LIST][PF_PER];
        constexpr int CH_PER_C = EW_PF_K / 8;                              // 16-byte chunks per candidate
        constexpr int PF16_PER = (PF_LIST * CH_PER_C + NT - 1) / NT;
        uint4 idq[PF16_PER];
        int ncand = 0;
        if (can_prefetch && IDMODE == 2) {
            // candidate list first (scalar work only): lane c of xs_lane holds the c-th unique candidate token
            // (PF_LIST: the reference's default tree has at most 4 children under a node; a tree of the same sizes with more takes the restage path)
            unsigned long long td = todo0;
            int xs_lane = -1;
#pragma unroll
            for (int c = 0; c < PF_LIST; ++c) {
                const bool have = td != 0ull;
                const int j = have ? __ffsll((long long)td) - 1 : 0;
                const int x = rdlane(x_lane, j);
                td &= ~__ballot(have && x_lane == x);
                if (lane == c) xs_lane = have ? x : -1;
                ncand += have ? 1 : 0;
            }
            EPW_STAMPG(18);
            // chunk ch = 8 ids of candidate ch / 128: a wave works on one candidate at a time (128 chunks = 2 waves)
#pragma unroll
            for (int u = 0; u < PF16_PER; ++u) {
                const int ch = tid + u * NT;
                const int c = __builtin_amdgcn_readfirstlane(ch / CH_PER_C);
                const int t0 = (ch % CH_PER_C) * 8;
                const int x = c < PF_LIST ? rdlane(xs_lane, c < PF_LIST ? c : 0) : -1;
                const int trow = x - off;
                const bool lookup = c < ncand && trow >= 0 && trow < p_trows && !(p_syntax && !(x >= p_img_lo && x < p_img_hi));
                idq[u] = make_uint4(0u, 0u, 0u, 0u);
                if (lookup && t0 < nz)
                    idq[u] = *reinterpret_cast<const uint4 *>(buf.nn_table + (size_t)trow * prm.table_cols + t0);
            }
        } else if (can_prefetch) {
            unsigned long long td = todo0;
#pragma unroll
            for (int c = 0; c < PF_LIST; ++c) {
                const bool have = td != 0ull;
                const int j = have ? __ffsll((long long)td) - 1 : 0;
                const int x = rdlane(x_lane, j);
                td &= ~__ballot(have && x_lane == x);
                const int trow = x - off;
                const bool lookup = have && trow >= 0 && trow < p_trows && !(p_syntax && !(x >= p_img_lo && x < p_img_hi));
                const uint16_t *nbp = buf.nn_table + (size_t)(lookup ? trow : 0) * prm.table_cols;
#pragma unroll
                for (int u = 0; u < PF_PER; ++u) {
                    const int t = tid + u * NT;
                    idv[c][u] = (lookup && t < nz) ? nbp[t] : (unsigned short)0;
                }
                ncand += have ? 1 : 0;
            }
        }
        {
            int rid = Srow[fi * Ds + (i - 1)];
            rid = rid < 0 ? 0 : (rid >= p_rows ? p_rows - 1 : rid);     // a bad row map must not read outside the batch
            const int hot = RAW ? S.hot[rid] : (!hot_g ? -1 : (hot_in_lds ? S.hot[rid] : hot_g[rid]));
            EPW_STAMP(10);
            if (hot < 0 && rp_rid != rid) {
                if constexpr (RAW) {
                    rp_probs = raw_p && S.pre[rid] != 0;
                    if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid * W, W, rp);
                    else raw_row_load<NT>(raw_c + (size_t)rid * V, raw_u + (size_t)rid * V, rp);
                } else row_load<NT, E4, FULLW>(logits + (size_t)rid * W, W, rp);
            }
            rp_rid = -1;
            auto stage_ids = [&]() {
                if (can_prefetch && IDMODE == 2) {
#pragma unroll
                    for (int u = 0; u < PF16_PER; ++u) {
                        const int ch = tid + u * NT;
                        const int c = ch / CH_PER_C, t0 = (ch % CH_PER_C) * 8;
                        if (c < ncand) {
                            uint32_t w[4] = {idq[u].x, idq[u].y, idq[u].z, idq[u].w}, ad[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const uint32_t i0 = (t0 + 2 * q < nz) ? (w[q] & 0xffffu) : 0u, i1 = (t0 + 2 * q + 1 < nz) ? (w[q] >> 16) : 0u;
                                ad[q] = (uint32_t)gather_addr((int)i0, t0 + 2 * q) | ((uint32_t)gather_addr((int)i1, t0 + 2 * q + 1) << 16);
                                if (t0 + 2 * q == k) S.nbk[c] = plain_addr((int)i0);                 // (k < nz: the id is real; else nothing reads nbk)
                                if (t0 + 2 * q + 1 == k) S.nbk[c] = plain_addr((int)i1);
                            }
                            *reinterpret_cast<uint4 *>(&S.nbaddr[c][t0]) = make_uint4(ad[0], ad[1], ad[2], ad[3]);
                        }
                    }
                } else if (can_prefetch) {
#pragma unroll
                    for (int c = 0; c < PF_LIST; ++c)
#pragma unroll
                        for (int u = 0; u < PF_PER; ++u) {
                            const int t = tid + u * NT;
                            if (c < ncand && t < EW_PF_K) {
                                S.nbaddr[c][t] = gather_addr((int)idv[c][u], t);
                                if (t == k) S.nbk[c] = plain_addr((int)idv[c][u]);
                            }
                        }
                }
            };
            if constexpr (RAW) {
                if (!rp_probs) raw_row_to_lds<NT, decltype(stage_ids), NUCLEUS>(rp, hot, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, Shist, ph, stage_ids, prm.top_p, S.redi);
                else row_softmax_to_lds<NT, E4, FULLW>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, stage_ids);
            } else row_softmax_to_lds<NT, E4, FULLW>(rp, hot, rows_probs, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, stage_ids);
            EPW_STAMP(11);
        }
        unsigned long long todo = todo0;
        int cidx = -1;
        while (todo != 0ull) {
            const int j = __ffsll((long long)todo) - 1;
            const int x = rdlane(x_lane, j);
            const unsigned long long same = __ballot(x_lane == x);
            todo &= ~same;                      // this path and every later path carrying the same token
            ++cidx;
            if (x < 0 || x >= V) {
                status = LANTERN_ST_TOKEN_OOB;
                break;
            }
            if (n_used >= EW_UNI || ucur0 + n_used >= prm.n_uniforms) {
                status = LANTERN_ST_UNIFORMS;
                break;
            }
            const double r = S.uni[n_used++];
            ++n_tried;
            EPW_STAMP(20);
            const bool x_in = (x >= lo && x < lo + W);
            const int flags = rdlane(flag_lane, j);
            const bool in_img = (flags & 2) != 0;
            const bool is_syn = (flags & 1) != 0;
            const int slot = cidx % PF_LIST;
            const int trow = x - off;
            const uint16_t *nb = (p_lantern && trow >= 0 && trow < p_trows) ? buf.nn_table + (size_t)trow * prm.table_cols : nullptr;
            int *dec = S.dec[n_tried & 1];
            if (LDSIDS && can_prefetch && cidx >= PF_LIST) {
                // more unique candidates than prefetch slots (rare): stage this one's ids now, reusing a finished slot
                __syncthreads();
                for (int t = tid; t < EW_PF_K; t += NT) {
                    const unsigned short id = (nb && t < nz) ? nb[t] : (unsigned short)0;
                    S.nbaddr[slot][t] = gather_addr((int)id, t);
                    if (t == k) S.nbk[slot] = plain_addr((int)id);
                }
                __syncthreads();
            }
            // static trees: start the drafter-row read now; it lands while wave 0 runs the neighbour scan and is
            // simply dropped if the candidate is accepted (one 32 KB row, L2/MALL-resident for the next try)
            float4 q[E4];
#pragma unroll
            for (int it = 0; it < E4; ++it)
                if constexpr (WPE == 1) q[it] = make_float4(0.f, 0.f, 0.f, 0.f);   // defined on every path: no value carried around the loop
            const float *qsrc = nullptr;
            if (is_static) {
                int qrow = rdlane(qrow_lane, j);
                qrow = qrow < 0 ? 0 : (qrow >= prm.R ? prm.R - 1 : qrow);                           // same for the drafter-row index
                qsrc = buf.orig_prob + ((size_t)b * prm.R + qrow) * (size_t)win.orig_prob_stride + win.orig_prob_offset;
            }
            // wave 0 (the serial worker) would sit behind the other waves' loads in the CU's address unit before it can enter
            // the scan: it fetches its own 4 KB share only once a rejection is known
            if (is_static && !LATE_Q && (wave != 0 || WPE != 1)) {     // (the throughput build has a second workgroup to hide the queueing)
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    q[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(qsrc)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            // ---------------- serial section: wave 0 only
            if (wave == 0) {
                EPW_STAMPF(27);
                float px = x_in ? g[x - lo] : (x == out_tok ? out_mass : 0.0f);
                int code = 0, mflag = 0;
                if (p_syntax && is_syn) {
                    px = 1.0f;
                } else if (p_syntax && !in_img) {
                    px = 0.0f;
                } else if (p_lantern) {
                    if (nb == nullptr) {
                        code = 3;   // LANTERN_ST_TABLE_OOB
                    } else {
                        const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                        float best_cs = NEG_INF;
                        if constexpr (LDSIDS) {
                            // 16 consecutive neighbours per lane, one round (k <= 1023).  The gather indices were resolved
                            // when the ids were staged (window index, or a sentinel slot: 0 outside the window, 3e38 at
                            // positions >= k so that they can never pass `<= tau`): 16 plain LDS reads, no predicate
                            const uint4 a = *reinterpret_cast<const uint4 *>(&S.nbaddr[slot][lane * 16]);
                            const uint4 bq = *reinterpret_cast<const uint4 *>(&S.nbaddr[slot][lane * 16 + 8]);
                            const uint32_t w[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
                            double v[16], loc = 0.0;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                loc += (double)g[w[c] & 0xffffu];
                                v[2 * c] = loc;
                                loc += (double)g[w[c] >> 16];
                                v[2 * c + 1] = loc;
                            }
                            // exclusive prefix = inclusive scan of the lane totals shifted up by one lane (NOT inc - loc: a
                            // lane whose own total holds a 3e38 sentinel would cancel its true prefix away)
                            const double excl = wave_scan_incl_dpp(dpp_mov<0x138>(loc));   // wave_shr:1, lane 0 gets 0
                            float mx = NEG_INF;
#pragma unroll
                            for (int c = 0; c < 16; ++c) {
                                const float cs = (float)(excl + v[c]);
                                mx = (cs <= tau) ? cs : mx;      // cs is non-decreasing in c: the last ok one is the largest
                            }
                            best_cs = wave_max(mx);
                        } else {
                        double carry = 0.0;
                        // 16 consecutive neighbours per lane, 1024 per round, ids from HBM; every gather is clamped + masked
                        for (int base = 0; base < k; base += 1024) {
                            const int i0 = base + lane * 16;
                            int ids[16];
#pragma unroll
                            for (int c = 0; c < 16; ++c) ids[c] = (i0 + c < k) ? (int)nb[i0 + c] : 0;
                            double v[16], loc = 0.0;
#pragma unroll
                            for (int c = 0; c < 16; ++c) {
                                const int t = ids[c] + off - lo;
                                const bool inw = (i0 + c < k) && t >= 0 && t < W;
                                float gv = g[inw ? t : 0];
                                gv = inw ? gv : 0.0f;
                                if (out_tok >= 0 && (i0 + c < k) && ids[c] + off == out_tok) gv = out_mass;   // hot token outside the window
                                loc += (double)gv;
                                v[c] = loc;
                            }
                            const double inc = wave_scan_incl_dpp(loc);
                            const double excl = carry + (inc - loc);
                            float mx = NEG_INF;
                            int nok = 0;
#pragma unroll
                            for (int c = 0; c < 16; ++c) {
                                const float cs = (float)(excl + v[c]);
                                const bool ok = (i0 + c < k) && cs <= tau;
                                mx = ok ? fmaxf(mx, cs) : mx;
                                nok += ok ? 1 : 0;
                            }
                            mx = wave_max(mx);
                            best_cs = fmaxf(best_cs, mx);
                            carry += readlane63(inc);
                            if (k - base <= 1024) break;
                            const int tot_ok = wave_sum(nok);
                            if (tot_ok < 1024) break;   // the cumulative mass is non-decreasing: the ok set is a prefix
                        }
                        }
                        if (best_cs > NEG_INF) {
                            mflag = 1;
                            px = px + best_cs;
                        }
                    }
                }
                if (code == 0) {
                    float qx = 1.0f;
                    bool skip = false;
                    if (is_static) {
                        qx = rdlane(cart_lane, j);
                        skip = qx <= 0.0f;
                    }
                    if (skip)
                        code = 0;
                    else
                        code = ((float)r <= px / qx) ? 1 : 2;
                }
                if (lane == 0) {
                    dec[0] = code;
                    dec[1] = mflag;
                }
                EPW_STAMPF(26);
            }
            __syncthreads();
            const int code = dec[0];
            const int m = dec[1];
            EPW_STAMP(21);
            if (code == 3) {
                status = LANTERN_ST_TABLE_OOB;
                break;
            }
            if (code == 0) continue;
            if (code == 1) {
                ++a;
                best = j;
                eq_mask &= same;                // paths that also carry the accepted token at this depth
                break;
            }
            // ------------------------------------------------ rejection: residual, all waves, all in LDS
            ++n_rej;
            if (p_syntax && is_syn) {
                status = LANTERN_ST_SYNTAX_REJECT;
                break;
            }
            if (is_static && (LATE_Q || (wave == 0 && WPE == 1))) {
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    q[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(qsrc)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const bool zero_nb = p_lantern && m > 0 && (!p_syntax || in_img);
            double loc = 0.0;
            float4 gn[E4];           // the unnormalised residual stays in registers until the sum is known
            if (!is_static) {
                if (tid == 0 && x_in) g[x - lo] = 0.0f;
                if (!x_in && x == out_tok) out_mass = 0.0f;
                if (zero_nb) {
                    bool hit = false;
                    for (int t = tid; t < nz; t += NT) {
                        if constexpr (LDSIDS) {
                            const int ad = t < k ? S.nbaddr[slot][t] : S.nbk[slot];
                            if (ad < W) g[ad] = 0.0f;
                            hit |= (ad == W + EW_G_OUT);
                        } else {
                            const int id = (int)nb[t] + off;
                            if (id >= lo && id < lo + W) g[id - lo] = 0.0f;
                            hit |= (id == out_tok);
                        }
                    }
                    if (out_tok >= 0 && block_sum_fast<int, NW>(hit ? 1 : 0, S.redi, ph) > 0) out_mass = 0.0f;
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    gn[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(g)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    loc += (double)gn[it].x + (double)gn[it].y + (double)gn[it].z + (double)gn[it].w;
                }
            } else {
                const int b0 = rdlane(b0_lane, j), b1 = rdlane(b1_lane, j);
                const int nsib = b1 - b0;
                if (nsib > EW_MAX_SIB || b1 > EW_MAX_B) {       // beyond the staged tables: say so instead of truncating the list
                    status = LANTERN_ST_TREE_LIMIT;
                    break;
                }
                // window indices of the earlier siblings' tokens, straight from the staged tables (every thread reads the same
                // LDS words: broadcast, no barrier); the first four live in registers, longer sibling lists loop over LDS
                auto sib_at = [&](int t) -> int {
                    const int node = (b0 + t < EW_MAX_B) ? (int)S.bidx[b0 + t] : 0;
                    const int tok = (node >= 0 && node < EW_MAX_N) ? S.tcand[node] : -1;
                    return (tok >= lo && tok < lo + W) ? (tok - lo) : -1;
                };
                int sib_r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) sib_r[t] = t < nsib ? sib_at(t) : -1;
                const bool lg_nb = zero_nb && p_mode == LANTERN_MODE_STATIC_LG;
                if (lg_nb) {
                    for (int t = tid; t < (W + 31) / 32; t += NT) nbmask[t] = 0u;
                    __syncthreads();
                    for (int t = tid; t < nz; t += NT) {
                        const int id = LDSIDS ? (int)(t < k ? S.nbaddr[slot][t] : S.nbk[slot]) : (int)nb[t] + off - lo;      // (a sentinel slot is >= W)
                        if (id >= 0 && id < W) atomicOr(&nbmask[id >> 5], 1u << (id & 31));
                    }
                }
                if (zero_nb && p_mode == LANTERN_MODE_STATIC_LUMINA)
                    for (int t = tid; t < nz; t += NT) {
                        const int id = LDSIDS ? (int)(t < k ? S.nbaddr[slot][t] : S.nbk[slot]) : (int)nb[t] + off - lo;
                        if (id >= 0 && id < W) g[id] = 0.0f;
                    }
                EPW_STAMPG(31);
                double qs_loc = 0.0;
                if (nsib > 0) {          // a level's first candidate has no earlier sibling: q is used as it is (qs = 1)
                    // window entry sidx lives in thread (sidx / 4) % NT, chunk (sidx / 4) / NT, component sidx % 4: only the WAVE that holds it
                    // runs the component selects (a wave-uniform branch), the other seven pay one compare per sibling
                    auto zero_q_at = [&](int sidx) {
                        const bool mine = sidx >= 0 && ((sidx >> 2) & (NT - 1)) == tid;
                        if (__ballot(mine) != 0ull) {
                            const int itx = mine ? (sidx >> 2) / NT : -1, c = sidx & 3;
#pragma unroll
                            for (int it = 0; it < E4; ++it)
                                if (it == itx) set_comp(q[it], c, 0.0f);
                        }
                    };
#pragma unroll
                    for (int t = 0; t < 4; ++t) zero_q_at(sib_r[t]);
                    for (int t = 4; t < nsib; ++t) zero_q_at(sib_at(t));
#pragma unroll
                    for (int it = 0; it < E4; ++it) qs_loc += (double)q[it].x + (double)q[it].y + (double)q[it].z + (double)q[it].w;
                }
                EPW_STAMPG(32);
                float qs = 1.0f;
                if (nsib > 0)
                    qs = (float)block_sum_fast<double, NW>(qs_loc, S.redd, ph);
                else
                    __syncthreads();   // neighbour zeroing / mask visible (block_sum_fast carries the barrier otherwise)
                EPW_STAMPG(33);
                const FastDiv dq(qs);
                if (nsib > 0) {      // one uniform branch around all chunks (not one inside each): the residual loop below stays one block
#pragma unroll
                    for (int it = 0; it < E4; ++it) q[it] = dq(q[it]);      // chunks beyond the window hold zeros: 0 / qs = 0
                }
                if (lg_nb) {         // likewise: the neighbour mask of the LlamaGen / Anole static mode, all chunks under one branch
#pragma unroll
                    for (int it = 0; it < E4; ++it) {
                        const int i4 = tid + it * NT;
                        if (FULLW || i4 * 4 < W) {
                            const int e = i4 * 4;
                            const uint32_t bits = nbmask[e >> 5] >> (e & 31);
                            if (bits & 1u) q[it].x = 0.f;
                            if (bits & 2u) q[it].y = 0.f;
                            if (bits & 4u) q[it].z = 0.f;
                            if (bits & 8u) q[it].w = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    gn[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (FULLW || i4 * 4 < W) {
                        const float4 qv = q[it];
                        float4 gv = reinterpret_cast<float4 *>(g)[i4];
                        float d;
                        d = gv.x - qv.x; gv.x = d < 0.0f ? 0.0f : d;
                        d = gv.y - qv.y; gv.y = d < 0.0f ? 0.0f : d;
                        d = gv.z - qv.z; gv.z = d < 0.0f ? 0.0f : d;
                        d = gv.w - qv.w; gv.w = d < 0.0f ? 0.0f : d;
                        gn[it] = gv;
                        loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                    }
                }
                // out-of-window mass: the drafter is zero there (precondition): max(out_mass - 0, 0) = out_mass
            }
            EPW_STAMPG(34);
            double tot = block_sum_fast<double, NW>(loc, S.redd, ph);
            EPW_STAMPG(35);
            tot += (double)out_mass;
            const float gs = (float)tot;
            if (gs == 0.0f) {
                status = LANTERN_ST_NEEDS_DENSE;   // `gtp.sum()==0 -> ones`: uniform over all V, only the dense kernel holds it
                break;
            }
            const FastDiv dg(gs);
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W)
                    reinterpret_cast<float4 *>(g)[i4] = dg(gn[it]);
            }
            out_mass = out_mass / gs;
            if (tid == 0) g[W + EW_G_OUT] = out_mass;
            __syncthreads();
            EPW_STAMP(30);
            adjust = 1;
        }
    }

    const int from_residual = (adjust && a != D) ? 1 : 0;
    if (status == LANTERN_ST_OK && !from_residual) {
        int rid = Srow[best * Ds + (a - 1)];
        rid = rid < 0 ? 0 : (rid >= p_rows ? p_rows - 1 : rid);
        const int hot = RAW ? S.hot[rid] : (!hot_g ? -1 : (hot_in_lds ? S.hot[rid] : hot_g[rid]));
        if (hot < 0 && rp_rid != rid) {
            if constexpr (RAW) {
                rp_probs = raw_p && S.pre[rid] != 0;
                if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid * W, W, rp);
                else raw_row_load<NT>(raw_c + (size_t)rid * V, raw_u + (size_t)rid * V, rp);
            } else row_load<NT, E4, FULLW>(logits + (size_t)rid * W, W, rp);
        }
        if constexpr (RAW) {
            if (!rp_probs) raw_row_to_lds<NT, NoHook, NUCLEUS>(rp, hot, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, Shist, ph, NoHook(), prm.top_p, S.redi);
            else row_softmax_to_lds<NT, E4, FULLW>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph);
        } else row_softmax_to_lds<NT, E4, FULLW>(rp, hot, rows_probs, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph);
    }
    // ---------------------------------------------------------------- epilogue: outputs from LDS
    EPW_STAMP(40);
    const EpwArgsK ka = (EpwArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    float *const k_sample_win = ka->win.sample_win, *const k_sample_p = ka->buf.sample_p;
    const double *const k_u_bonus = ka->win.u_bonus;
    int64_t *const k_token = ka->win.token;
    float4 p[E4];
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = tid + it * NT;
        p[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(g)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (k_sample_win) {
        float *sw = k_sample_win + (size_t)b * W;
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int i4 = tid + it * NT;
            if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(sw)[i4] = p[it];
        }
    }
    if (k_sample_p) {   // optional dense copy (API compatibility)
        float *sp = k_sample_p + (size_t)b * V;
        for (int i4 = tid; i4 * 4 < V; i4 += NT) {
            const int e = i4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e >= lo && e < lo + W) v = reinterpret_cast<const float4 *>(g)[(e - lo) / 4];
            if (out_tok >= e && out_tok < e + 4) set_comp(v, out_tok - e, out_mass);
            reinterpret_cast<float4 *>(sp)[i4] = v;
        }
    }
    if (k_u_bonus && k_token && status == LANTERN_ST_OK) {
        // inverse CDF in token-id order.  Register tile order (it, tid, component) IS ascending token id.
        const bool out_before = out_tok >= 0 && out_tok < lo;
        double s4[E4], inc[E4];
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            s4[it] = (double)p[it].x + (double)p[it].y + (double)p[it].z + (double)p[it].w;
            inc[it] = wave_scan_incl_dpp(s4[it]);
            if (lane == 63) S.samp_tot[wave * E4 + it] = inc[it];
        }
        __syncthreads();
        // segment (it, wave) = ids [lo + 4*(it*NT + 64*wave), +256): token order is it-major.  One DPP scan over the
        // E4*NW segment totals (lane q = it*NW + w) gives every segment's start; each thread then picks its E4 starts
        // with uniform-lane reads.
        static_assert(E4 * NW <= 64, "segment totals fit one wave");
        const int q_it = lane / NW, q_w = lane % NW;
        const double seg = (lane < E4 * NW) ? S.samp_tot[q_w * E4 + (q_it < E4 ? q_it : 0)] : 0.0;
        const double seg_excl = wave_scan_incl_dpp(dpp_mov<0x138>(seg));      // exclusive: scan of the totals shifted up one lane
        const double front = out_before ? (double)out_mass : 0.0;   // mass in front of the window
        double total = front + (readlane63(seg_excl) + readlane63(seg));       // lanes >= E4*NW hold 0
        double excl[E4];
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int q = it * NW + wave;                       // wave-uniform
            const long long bits = __double_as_longlong(seg_excl);
            const double start = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), q) << 32) |
                                                      (unsigned int)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), q));
            excl[it] = front + start + (inc[it] - s4[it]);
        }
        if (out_tok >= 0 && !out_before) total += (double)out_mass;
        const double tgt = S.ubonus[0] * total;
        int found = 0x7fffffff, last_pos = -1;
        if (out_before && out_mass > 0.0f) {
            last_pos = out_tok;
            if ((double)out_mass > tgt) found = out_tok;
        }
        // The crossing is the smallest id whose running sum exceeds tgt: a 4-id chunk whose sum range [excl, excl + s4] lies wholly below tgt
        // cannot hold it, and a chunk wholly above tgt can only offer an id larger than the crossing chunk's -- so only chunks whose range
        // comes within a guard band of tgt run the element loop (same additions in the same order as before, hence the same token): one or two
        // threads of the workgroup instead of all 512 x 16 elements in f64.
        const double band = 1e-9 * total;
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            if (excl[it] - band <= tgt && excl[it] + s4[it] + band >= tgt) {
                const int e = lo + (tid + it * NT) * 4;
                double acc = excl[it];
                const float pv[4] = {p[it].x, p[it].y, p[it].z, p[it].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc += (double)pv[c];
                    if (pv[c] > 0.0f && acc > tgt) found = min(found, e + c);
                }
            }
        }
        if (out_tok >= 0 && !out_before && out_mass > 0.0f && total > tgt) found = min(found, out_tok);   // only wins when nothing inside the window crossed
        found = wave_min_i(found);
        __syncthreads();
        if (lane == 0) S.redi[wave] = found;
        __syncthreads();
        int f = S.redi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) f = min(f, S.redi[w]);
        if (f == 0x7fffffff) {
            // nothing crossed (tgt landed on the rounding of the total): the reference's inverse CDF then yields the LAST id with mass -- found
            // by a second pass, off the common path
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int e = lo + (tid + it * NT) * 4;
                const float pv[4] = {p[it].x, p[it].y, p[it].z, p[it].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (pv[c] > 0.0f) last_pos = max(last_pos, e + c);
            }
            if (out_tok >= 0 && !out_before && out_mass > 0.0f) last_pos = max(last_pos, out_tok);
            last_pos = wave_max_i(last_pos);
            __syncthreads();
            if (lane == 0) S.redi[16 + wave] = last_pos;
            __syncthreads();
            f = S.redi[16];
#pragma unroll
            for (int w = 1; w < NW; ++w) f = max(f, S.redi[16 + w]);
        }
        if (tid == 0) k_token[b] = f;
    }
    EPW_STAMP(50);
#ifdef EPW_TRACE
    if (tid == 0 && b < EPW_TR_BLOCKS) {
        const int n = s_epw_trn;
        for (int t = 0; t < n; ++t) g_epw_trace[b][t] = s_epw_tr[t];
        g_epw_trace_n[b] = n;
    }
#endif
    if (tid == 0) {
        ka->buf.best[b] = best;
        ka->buf.accept_len[b] = a - 1;
        int32_t *c = ka->buf.counters + (size_t)b * 6;
        c[0] = n_levels;
        c[1] = n_tried;
        c[2] = n_rej;
        c[3] = n_used;
        c[4] = from_residual;
        c[5] = status;
        if (ka->buf.cursor) ka->buf.cursor[b] = ucur0 + n_used;
        if (ka->win.out_tok) ka->win.out_tok[b] = out_tok;
        if (ka->win.out_mass) ka->win.out_mass[b] = out_mass;
    }
    return (best << 8) | a;          // the verdict (uniform): best path, rows kept = accept_len + 1
}

template <int NT, int E4, int IDMODE, int WPE, bool FULLW = false, bool RAW = false, int SPEC = 0, int TPO = 0>
__global__ __launch_bounds__(NT, WPE) void epw_kernel(const EpwArgs args) {
    epw_body<NT, E4, IDMODE, WPE, FULLW, RAW, SPEC, TPO>(args, blockIdx.x);
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
        static const int nt_knob = getenv("LANTERN_O7_NT") ? atoi(getenv("LANTERN_O7_NT")) : 0;   // tuning knob (diagnostic)
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

extern "C" int lantern_prepare_step(const lantern_step_group *g) {
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
    if (g->dyn) return prepare_step_dynamic(g);
    LANTERN_CHECK_ARG(g->ss_token && g->sample_token && g->tree_indices && g->retrieve && g->tree_cand && g->cand && g->B >= 0 && g->n_flat > 0 && g->N > 0 &&
                          g->P > 0 && g->D > 0, "prepare_step: candidate-assembly buffers missing");
    if (g->B == 0) return LANTERN_OK;
    PrepArgs a{(const uint16_t *)g->cond, (const uint16_t *)g->uncond, g->V, g->cfg, g->pos_ids, g->pos_base, g->w_latent, g->h_latent, g->img_lo, g->img_hi,
               g->newline_id, g->eos_id, g->top_k, g->seq_len, g->N, g->win_lo, g->win_len, g->out_win, g->row_hot, g->node_list, g->n_list, g->B,
               g->ss_token, g->ss_prob, g->sample_token, g->tree_indices, g->retrieve, g->n_flat, g->N, g->P * g->D, g->tree_cand, g->cand, g->cart_prob,
               g->top_p};
    static const int nt_knob = getenv("LANTERN_PREP_NT") ? atoi(getenv("LANTERN_PREP_NT")) : 0;   // tuning knob (diagnostic)
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
    if (p.top_p > 0.0f && p.top_p < 1.0f && !raw) {
        set_error("evaluate_posterior_window: top_p=%g inside the kernel is built for LANTERN_ROWS_RAW_BF16 rows only (probability / logit rows: apply it "
                  "where the rows are produced, lantern_cfg_mask_topk_window)", (double)p.top_p);
        return LANTERN_E_UNSUPPORTED;
    }
    if (p.top_k > win->win_len && p.top_k < p.V) {
        set_error("evaluate_posterior_window: top_k=%d wider than the window (%d) needs the dense kernel", p.top_k, win->win_len);
        return LANTERN_E_UNSUPPORTED;
    }
    return LANTERN_OK;
}

static size_t epw_lds_bytes(const lantern_ep_params &p, const lantern_ep_window *win) {
    return epw_shared_offset(win->win_len) + sizeof(EwShared) + (size_t)6 * epw_pd_cap(p.P, p.D) * 4 +
           (win->rows_kind == LANTERN_ROWS_RAW_BF16 ? (size_t)O7_HIST_INTS * 4 : 0);
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
#define EPW_LAUNCH_W(NT_, E4_, WPE_)                                                                                        \
    do {                                                                                                                  \
        if (idmode == 2) LANTERN_LAUNCH((epw_kernel<NT_, E4_, 2, WPE_>), grid, dim3(NT_), lds, st, args);                  \
        else if (idmode == 1) LANTERN_LAUNCH((epw_kernel<NT_, E4_, 1, WPE_>), grid, dim3(NT_), lds, st, args);             \
        else LANTERN_LAUNCH((epw_kernel<NT_, E4_, 0, WPE_>), grid, dim3(NT_), lds, st, args);                              \
    } while (0)
#define EPW_LAUNCH(NT_, E4_) EPW_LAUNCH_W(NT_, E4_, 1)
    // the headline shape gets its own instance (SPEC 1: mode / LANTERN / syntax-shortcut flags are compile-time constants there)
    static const int spec_knob = getenv("LANTERN_EPW_SPEC") ? atoi(getenv("LANTERN_EPW_SPEC")) : 2;   // tuning knob (diagnostic): 0 = the generic instance, 1 = no fixed tree
    const bool chameleon = spec_knob != 0 && p.lantern && p.V == 65536 && p.img_lo == 4 && p.img_hi == 8196 && p.tok_offset == 4 && p.table_rows == 8192 &&
                           win->win_lo == 4 && W == 8192 && p.rows_per_seq <= EW_MAX_N && (raw || win->rows_kind == LANTERN_ROWS_PROBS) &&
                           (!raw || win->raw_w_latent == 0 || (win->raw_eos_id == 8196 && win->raw_newline_id == 8803));
    const bool lumina_syntax = p.syntax_shortcut && p.n_syntax == 4 && p.syntax[0] == 8196 && p.syntax[1] == 8197 && p.syntax[2] == 8803 && p.syntax[3] == 8828;
    const bool lumina_static = chameleon && p.mode == LANTERN_MODE_STATIC_LUMINA && lumina_syntax && !buf->n_paths && !buf->n_depth && (!raw || !win->raw_pos_per_seq);
    const bool lumina_dynamic = chameleon && p.mode == LANTERN_MODE_DYNAMIC && lumina_syntax && buf->n_paths && buf->n_depth && (!raw || win->raw_pos_per_seq);
    const bool anole_static = chameleon && p.mode == LANTERN_MODE_STATIC_LG && !p.syntax_shortcut && !buf->n_paths && !buf->n_depth &&
                              (!raw || (win->raw_w_latent == 0 && !win->raw_pos_per_seq));
    const bool default_tree = spec_knob >= 2 && p.P == 15 && p.D == 6 && p.N == 26 && p.rows_per_seq == 26;
    static const int occ_knob = getenv("LANTERN_EPW_OCC2") ? atoi(getenv("LANTERN_EPW_OCC2")) : -1;   // tuning knob (diagnostic)
    const bool two_per_cu = occ_knob >= 0 ? occ_knob != 0 : p.B > 256;
    // Throughput form of the fixed-configuration instances (more sequences than CUs; the shape BASELINE's roofline target is assessed on): 256
    // threads x 8 float4 per thread, three workgroups per CU (53 KB of LDS each, <= 168 VGPRs at 3 waves per SIMD), drafter rows requested only once
    // a rejection is known.  At saturation the kernel is bound by instruction ISSUE (profiles/r04_ep_sweep_pmc.txt: the SIMDs' arbiters busy 0.93 of
    // the launch, a third of it scalar work every wave of a sequence repeats), so half the waves per sequence is what pays: 4096 sequences per
    // launch 293 us (generic, 512 threads, two per CU) -> 228 (fixed configuration, 512 threads) -> 194 (owner-wave sibling zeroing) -> 163 us.
    // LANTERN_EPW_TP=0: the generic two-per-CU instance; 1: the 512-thread fixed-configuration instance (diagnostic).
    static const int tp_knob = getenv("LANTERN_EPW_TP") ? atoi(getenv("LANTERN_EPW_TP")) : 5;   // tuning knob (diagnostic)
    if (W <= 1024) EPW_LAUNCH(256, 1);
    else if (W <= 2048) EPW_LAUNCH(256, 2);
    else if (W <= 4096) EPW_LAUNCH(512, 2);
    else if (raw && p.top_p >= 1e-8f && p.top_p < 1.0f) {          // raw rows with a nucleus filter: the generic raw instances with the filter compiled in
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
        if (two_per_cu) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 4, true, true>), grid, dim3(512), lds, st, args);
        else if (lumina_static && default_tree) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 2>), grid, dim3(512), lds, st, args);
        else if (lumina_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 1>), grid, dim3(512), lds, st, args);
        else if (lumina_dynamic) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 3>), grid, dim3(512), lds, st, args);
        else if (anole_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true, 4>), grid, dim3(512), lds, st, args);
        else LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, true>), grid, dim3(512), lds, st, args);
    }
    else if (W <= 8192) {
        const bool tp_form = two_per_cu && W == 8192 && idmode == 2 && tp_knob >= 5;
        if (tp_form && lumina_static && default_tree) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 3, true, false, 2, 1>), grid, dim3(256), lds, st, args);
        else if (tp_form && lumina_static) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 3, true, false, 1, 1>), grid, dim3(256), lds, st, args);
        else if (tp_form && lumina_dynamic) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 3, true, false, 3, 1>), grid, dim3(256), lds, st, args);
        else if (tp_form && anole_static) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 3, true, false, 4, 1>), grid, dim3(256), lds, st, args);
        else if (two_per_cu && W == 8192 && idmode == 2 && lumina_static && default_tree && tp_knob >= 1) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 4, true, false, 2>), grid, dim3(512), lds, st, args);
        else if (two_per_cu && W == 8192 && idmode == 2) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 4, true>), grid, dim3(512), lds, st, args);
        else if (two_per_cu) EPW_LAUNCH_W(512, 4, 4);
        else if (W == 8192 && idmode == 2) {
            if (lumina_static && default_tree) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 2>), grid, dim3(512), lds, st, args);
            else if (lumina_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 1>), grid, dim3(512), lds, st, args);
            else if (lumina_dynamic) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 3>), grid, dim3(512), lds, st, args);
            else if (anole_static) LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true, false, 4>), grid, dim3(512), lds, st, args);
            else LANTERN_LAUNCH((epw_kernel<512, 4, 2, 1, true>), grid, dim3(512), lds, st, args);   // the Lumina / Anole image window on the packed table
        }
        else EPW_LAUNCH(512, 4);
    }
    else EPW_LAUNCH(1024, 4);
#undef EPW_LAUNCH_W
#undef EPW_LAUNCH
    LANTERN_CHECK_LAUNCH("evaluate_posterior_window");
    return LANTERN_OK;
}

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
