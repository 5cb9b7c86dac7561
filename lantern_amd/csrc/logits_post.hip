// logits_post.hip -- O7: CFG combine + model mask + top-k threshold in one pass.
//
// One 1024-thread workgroup per tree-node row.  The row lives in registers between the
// CFG/mask stage and the k-th-largest selection, so HBM sees one read of cond+uncond and
// one write of the processed f32 row.  Lumina rows are classified from their position:
// newline / end-of-image rows are one-hot and read nothing; grid rows read only the image
// token range [img_lo, img_hi) (everything else is -inf by construction).
//
// Reference: models/ea_model_lumina_mgpt.py:597-605, :45-86, :106-112;
// models/ea_model_anole.py:930-931; models/ea_model_llamagen.py:26-29, :930.
#include "common.h"
#include <type_traits>

namespace lantern {

constexpr int LP_THREADS = 1024;
constexpr int LP_NW = LP_THREADS / 64;

struct LpShared {
    int redi[2 * LP_NW];
};

__device__ __forceinline__ int64_t py_mod(int64_t a, int64_t b) {
    int64_t r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? r + b : r;
}

template <int VI, bool BF16>
__global__ __launch_bounds__(LP_THREADS) void cfg_mask_topk_kernel(const void *__restrict__ cond_, const void *__restrict__ uncond_,
                                                                  int V, float cfg, int model,
                                                                  const int64_t *__restrict__ pos_ids, int64_t pos_base,
                                                                  int w_latent, int h_latent, int img_lo, int img_hi,
                                                                  int newline_id, int eos_id, int top_k,
                                                                  const int64_t *__restrict__ seq_len, int rows_per_seq,
                                                                  float *__restrict__ out_) {
    __shared__ LpShared S;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float NEG_INF = -__builtin_inff();
    float *out = out_ + (size_t)row * V;
    int ph = 0;

    // row class (Lumina): 0 = grid position, 1 = newline, 2 = end of image
    int cls = 0;
    if (model == LANTERN_MODEL_LUMINA) {
        const int64_t pos = seq_len ? pos_ids[row % rows_per_seq] + seq_len[row / rows_per_seq] : pos_ids[row];
        const int64_t n1 = pos - pos_base + 1;
        if (n1 == ((int64_t)w_latent + 1) * h_latent + 1)
            cls = 2;
        else if (py_mod(n1, (int64_t)w_latent + 1) == 0)
            cls = 1;
    }
    if (cls != 0) {
        const int hot = cls == 2 ? eos_id : newline_id;
        for (int i4 = tid; i4 * 4 < V; i4 += LP_THREADS) {
            float4 v = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            const int e = i4 * 4;
            if (hot >= e && hot < e + 4) set_comp(v, hot - e, 0.0f);
            reinterpret_cast<float4 *>(out)[i4] = v;
        }
        return;  // top-k of a one-hot row removes nothing
    }

    const bool lumina = model == LANTERN_MODEL_LUMINA;
    const float fill = lumina ? NEG_INF
                              : (BF16 ? __uint_as_float(0xff7f0000u) /* finfo(bf16).min */
                                      : -3.4028234663852886e38f /* finfo(f32).min */);
    const bool masked = model != LANTERN_MODEL_PLAIN;
    float4 r[VI];
    // the row's tile, with the `uncond == NULL` decision taken ONCE (a wave-uniform branch around the whole loop): as a per-chunk
    // `uncond ? load : cond` the compiler waited for each conditional chunk before fetching the unconditional one, element by element
    auto load_tile = [&](auto has_u) {
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            const int i4 = tid + it * LP_THREADS;
            const int e = i4 * 4;
            float4 v = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            if (e < V) {
                const bool need = !masked || (e + 4 > img_lo && e < img_hi);
                if (need) {
                    float c[4], u[4];
                    if (BF16) {
                        const ushort4 cb = reinterpret_cast<const ushort4 *>((const uint16_t *)cond_ + (size_t)row * V)[i4];
                        ushort4 ub = cb;
                        if constexpr (decltype(has_u)::value) ub = reinterpret_cast<const ushort4 *>((const uint16_t *)uncond_ + (size_t)row * V)[i4];
                        c[0] = bf16_bits_to_f32(cb.x); c[1] = bf16_bits_to_f32(cb.y);
                        c[2] = bf16_bits_to_f32(cb.z); c[3] = bf16_bits_to_f32(cb.w);
                        u[0] = bf16_bits_to_f32(ub.x); u[1] = bf16_bits_to_f32(ub.y);
                        u[2] = bf16_bits_to_f32(ub.z); u[3] = bf16_bits_to_f32(ub.w);
                    } else {
                        const float4 cf = reinterpret_cast<const float4 *>((const float *)cond_ + (size_t)row * V)[i4];
                        float4 uf = cf;
                        if constexpr (decltype(has_u)::value) uf = reinterpret_cast<const float4 *>((const float *)uncond_ + (size_t)row * V)[i4];
                        c[0] = cf.x; c[1] = cf.y; c[2] = cf.z; c[3] = cf.w;
                        u[0] = uf.x; u[1] = uf.y; u[2] = uf.z; u[3] = uf.w;
                    }
                    float o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = c[q];
                        if constexpr (decltype(has_u)::value) {   // uncond == NULL: logits are already combined, mask / top-k only
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
                } else {
                    v = make_float4(fill, fill, fill, fill);
                }
            }
            r[it] = v;
        }
    };
    if (uncond_) load_tile(std::true_type{});
    else load_tile(std::false_type{});

    if (top_k > 0) {
        // k-th largest over the register tile; iterations wholly outside the finite window
        // of a masked row hold only `fill` and are skipped while counting.
        int it_lo = 0, it_hi = VI;
        if (masked && lumina) {
            it_lo = (img_lo / 4) / LP_THREADS;
            it_hi = ((img_hi + 3) / 4 + LP_THREADS - 1) / LP_THREADS;
            if (it_hi > VI) it_hi = VI;
        }
        const int kk = top_k < V ? top_k : V;
        // elements outside the counted window are all -inf: they can only matter when fewer
        // than kk finite values exist, in which case the threshold is -inf and nothing changes.
        uint32_t prefix = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int it = 0; it < VI; ++it) {
                if (it < it_lo || it >= it_hi) continue;
                c += float_key(r[it].x) >= trial;
                c += float_key(r[it].y) >= trial;
                c += float_key(r[it].z) >= trial;
                c += float_key(r[it].w) >= trial;
            }
            const int tot = block_sum<int, LP_NW>(c, S.redi, ph);
            if (tot >= kk) prefix = trial;
        }
        // prefix == 0 means fewer than kk values counted: threshold below every float
        const float thr = prefix == 0 ? NEG_INF : key_float(prefix);
#pragma unroll
        for (int it = 0; it < VI; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x;
            r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z;
            r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
#pragma unroll
    for (int it = 0; it < VI; ++it) {
        const int i4 = tid + it * LP_THREADS;
        if (i4 * 4 < V) reinterpret_cast<float4 *>(out)[i4] = r[it];
    }
}

}  // namespace lantern

using namespace lantern;

template <int VI>
static void launch_lp(bool bf16, dim3 grid, hipStream_t st, const void *cond, const void *uncond, int V, float cfg, int model,
                      const int64_t *pos_ids, int64_t pos_base, int w, int h, int img_lo, int img_hi, int nl, int eos,
                      int top_k, const int64_t *seq_len, int rows_per_seq, float *out) {
    if (bf16)
        hipLaunchKernelGGL((cfg_mask_topk_kernel<VI, true>), grid, dim3(LP_THREADS), 0, st, cond, uncond, V, cfg, model, pos_ids,
                           pos_base, w, h, img_lo, img_hi, nl, eos, top_k, seq_len, rows_per_seq, out);
    else
        hipLaunchKernelGGL((cfg_mask_topk_kernel<VI, false>), grid, dim3(LP_THREADS), 0, st, cond, uncond, V, cfg, model, pos_ids,
                           pos_base, w, h, img_lo, img_hi, nl, eos, top_k, seq_len, rows_per_seq, out);
}

extern "C" int lantern_cfg_mask_topk(const void *cond, const void *uncond, int dtype, int rows, int V, float cfg, int model,
                                     const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent, int img_lo,
                                     int img_hi, int newline_id, int eos_id, int top_k, const int64_t *seq_len,
                                     int rows_per_seq, float *out, void *stream) {
    LANTERN_CHECK_ARG(cond && out, "cfg_mask_topk: null buffer");
    LANTERN_CHECK_ARG(rows >= 0 && V > 0 && V % 4 == 0 && V <= 4096 * 16, "cfg_mask_topk: bad rows=%d V=%d", rows, V);
    LANTERN_CHECK_ARG(dtype == LANTERN_F32 || dtype == LANTERN_BF16, "cfg_mask_topk: bad dtype %d", dtype);
    LANTERN_CHECK_ARG(model >= 0 && model <= 2, "cfg_mask_topk: bad model %d", model);
    if (model == LANTERN_MODEL_LUMINA)
        LANTERN_CHECK_ARG(pos_ids && w_latent > 0 && h_latent > 0 && newline_id >= 0 && newline_id < V && eos_id >= 0 && eos_id < V,
                          "cfg_mask_topk: Lumina needs pos_ids, latent dims and syntax ids inside [0,V)");
    if (seq_len) LANTERN_CHECK_ARG(rows_per_seq > 0 && rows % rows_per_seq == 0, "cfg_mask_topk: rows=%d not a multiple of rows_per_seq=%d", rows, rows_per_seq);
    if (model != LANTERN_MODEL_PLAIN)
        LANTERN_CHECK_ARG(img_lo >= 0 && img_lo < img_hi && img_hi <= V, "cfg_mask_topk: bad image range [%d,%d)", img_lo, img_hi);
    if (rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(rows);
    const bool bf = dtype == LANTERN_BF16;
#define LP_ARGS bf, grid, st, cond, uncond, V, cfg, model, pos_ids, pos_base, w_latent, h_latent, img_lo, img_hi, newline_id, eos_id, top_k, seq_len, rows_per_seq, out
    if (V <= 4096)
        launch_lp<1>(LP_ARGS);
    else if (V <= 4096 * 4)
        launch_lp<4>(LP_ARGS);
    else if (V <= 4096 * 8)
        launch_lp<8>(LP_ARGS);
    else
        launch_lp<16>(LP_ARGS);
#undef LP_ARGS
    LANTERN_CHECK_LAUNCH("cfg_mask_topk");
    return LANTERN_OK;
}
