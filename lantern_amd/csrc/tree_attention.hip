// tree_attention.hip -- SURVEY 8f row 3: the attention of the TARGET model's tree-verify forward (the step right before the
// accept path), replacing the reference's eager
//     softmax_f32(Q K^T / sqrt(d) + additive [B,1,N,S] f32 mask).to(bf16) @ V
// (models/kv_variants/modeling_lumina_mgpt_kv.py:433-442; the mask is built per forward by _prepare_decoder_attention_mask,
// :1508-1546: causal + left padding + `tree_mask == 0 -> min` on the last N x N block) with one flash-style pass over the
// KV cache in its own layout (kv_cache.py: [B, Hkv, S_max, d] per layer and K/V).  The mask never exists: a query node sees
// every cached key from kv_start[b] up to the prefix end and, inside the N tree keys, exactly its ancestors -- one 64-bit word
// per node (N <= 64).
//
// Shape of the work: per (batch row, head) the N <= 64 tree queries meet S keys; S*d*2*2 bytes of K and V are streamed once
// (HBM-bound: 1.2 MB per head at S = 2400, d = 128), so the kernel is organised around reading K/V exactly once with 16-byte
// loads and keeping everything else on chip:
//   * swapped product S^T = K Q^T on v_mfma_f32_32x32x16_bf16: the A fragment of K (lane (r,h): K[key0+r][16ks+8h ..+8)) and
//     the B fragment of Q^T (Q[query r][16ks+8h ..+8)) are 16 contiguous bytes of a row, straight from global memory; Q
//     fragments stay in registers for the whole pass;
//   * in S^T's accumulator layout a lane holds 16 keys of ONE query (col = lane&31), so the online softmax is lane-local
//     plus one exchange with lane^32, and the exponentials are already the B fragment of O^T = V^T P^T (keys 8g+4h+j of
//     register group g feed k-step g/2 -- the key order inside the contraction is free as long as V^T agrees);
//   * V^T fragments come from a per-wave LDS image of the [32 keys][d] tile, filled with coalesced 16-byte loads and read with
//     ds_read_b64_tr_b16 (4 keys x 16 columns per 16 lanes, delivered column-major): no register transpose, no workgroup
//     barrier inside the loop (each wave owns its tiles and its LDS slice);
//   * the 4 waves of a workgroup take every 4th key tile; their (m, l, O) are merged once through LDS in wave order (no
//     float atomics: the result does not depend on timing).  When
//     B*Hq alone cannot fill 256 CUs the key range is also split over gridDim.y and merged by a second tiny kernel.
#include "common.h"

namespace lantern {

typedef __bf16 ta_bf16x8_t __attribute__((ext_vector_type(8)));
typedef short ta_s16x4_t __attribute__((ext_vector_type(4)));
typedef float ta_f32x16_t __attribute__((ext_vector_type(16)));

constexpr int TA_WAVES = 4;
constexpr int TA_THREADS = TA_WAVES * 64;
constexpr int TA_TILE = 32;   // keys per wave step

struct TaArgs {
    const uint16_t *q, *k, *v;
    uint16_t *out;
    const int64_t *kv_len, *kv_start;
    const uint64_t *bits;
    float *ws;
    int64_t q_sb, q_sn, q_sh;   // element strides of q[b][n][h][:]
    int64_t kv_sb, kv_sh;       // element strides of the caches (row stride = D)
    int64_t o_sb, o_sn;         // element strides of out[b][n][h*D + :]
    int64_t max_kv_len;
    int B, Hq, Hkv, N, bits_per_row, q_groups;
    float scale_log2;
};

// byte offset of 16-byte chunk `ch` of key row `row` in a wave's V image; the XOR keeps both the 16-byte writes and the
// transposed reads (4 rows x 64 bytes per 32-lane half) off each other's banks (cdna_hip_programming.md T10, image (b))
template <int D>
__device__ __forceinline__ int ta_v_off(int row, int ch) {
    if (D == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    return 128 * row + 16 * (ch ^ (((row >> 1) & 1) << 2));
}

__device__ __forceinline__ ta_bf16x8_t ta_load8(const uint16_t *p) {
    return __builtin_bit_cast(ta_bf16x8_t, *reinterpret_cast<const uint4 *>(p));
}

template <int D, int QT>
__global__ __launch_bounds__(TA_THREADS, 2) void tree_attention_kernel(const TaArgs a) {
    constexpr int KS = D / 16, DT = D / 32, CPR = D / 8, VL = (TA_TILE * CPR) / 64, QR = QT * 32;
    __shared__ __attribute__((aligned(16))) unsigned char s_v[TA_WAVES][TA_TILE * D * 2];
    __shared__ __attribute__((aligned(16))) float s_o[QR][D + 4];
    __shared__ float s_m[TA_WAVES][QR];
    __shared__ float s_l[QR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    // Query groups (N > 32 at d = 128: two workgroups of 32 queries each, so that the 64 accumulator registers of O^T leave
    // room for two waves per SIMD): ids 16g+j and 16g+8+j carry the two groups of (batch row, head) 8g+j -- same XCD (id % 8),
    // dispatched together, so the second reader of a K/V row finds it in that XCD's L2.  (Measured alternatives at 48
    // sequences, N = 59: both groups in one 4-wave workgroup at 473 VGPRs 927 us; one 8-wave workgroup, groups in lockstep on
    // the same CU 943 us; this pairing 796 us.)
    int bh = blockIdx.x, qs = 0;
    if (a.q_groups == 2) {
        const int j = bh & 15;
        bh = (bh >> 4) * 8 + (j & 7);
        qs = j >> 3;
        if (bh >= a.B * a.Hq) return;
    }
    const int q0 = qs * QT * 32;
    const int b = bh / a.Hq, h = bh % a.Hq, hk = h / (a.Hq / a.Hkv);
    const int split = blockIdx.y, nsplit = gridDim.y;
    int64_t len = a.kv_len ? a.kv_len[b] : a.max_kv_len;
    len = len < a.N ? a.N : (len > a.max_kv_len ? a.max_kv_len : len);
    const int64_t prev = len - a.N;
    int64_t start = a.kv_start ? a.kv_start[b] : 0;
    start = start < 0 ? 0 : (start > prev ? prev : start);
    const int t_first = (int)(start >> 5), t_last = (int)((len + TA_TILE - 1) >> 5);
    const int per = (t_last - t_first + nsplit - 1) / nsplit;
    const int tb = t_first + split * per, te = min(t_last, tb + per);
    const uint16_t *kb = a.k + (int64_t)b * a.kv_sb + (int64_t)hk * a.kv_sh;
    const uint16_t *vb = a.v + (int64_t)b * a.kv_sb + (int64_t)hk * a.kv_sh;
    const float NEG = -__builtin_huge_valf();

    // Q^T fragments and each lane's ancestor word (its query column)
    ta_bf16x8_t qf[QT][KS];
    uint64_t anc[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qn = q0 + 32 * qt + r;
        const bool live = qn < a.N;
        const uint16_t *qp = a.q + (int64_t)b * a.q_sb + (int64_t)(live ? qn : 0) * a.q_sn + (int64_t)h * a.q_sh + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ta_bf16x8_t f = ta_load8(qp + 16 * ks);
            if (!live) f = ta_bf16x8_t{};
            qf[qt][ks] = f;
        }
        anc[qt] = live ? a.bits[(a.bits_per_row ? (int64_t)b * a.N : 0) + qn] : ~0ull;
    }

    ta_f32x16_t o[DT][QT];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m[qt] = NEG;
        l[qt] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][qt] = ta_f32x16_t{};
    }

    unsigned char *vimg = s_v[wave];
    // transposed-read addresses: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its 4 x 16 block
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;

    // Software pipeline without extra registers: a tile's V registers are free once written to the LDS image and its K
    // fragments once S^T is computed, so the NEXT tile's V loads are issued right after the image is written and its K
    // loads right after the QK products -- both are in flight during the softmax and the PV products of the current tile.
    ta_bf16x8_t kf[KS];
    uint4 vr[VL];
    auto load_k = [&](int tile) {
        int64_t krow = ((int64_t)tile << 5) + r;
        krow = krow < len ? krow : len - 1;
        const uint16_t *kp = kb + krow * D + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = ta_load8(kp + 16 * ks);      // 16 contiguous bytes per lane and k-step
    };
    auto load_v = [&](int tile) {   // rows past the end are zeroed: 0 * stale-NaN would poison O
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            const int64_t key = ((int64_t)tile << 5) + row;
            vr[i] = key < len ? *reinterpret_cast<const uint4 *>(vb + key * D + 8 * ch) : make_uint4(0, 0, 0, 0);
        }
    };
    int tile = tb + wave;
    if (tile < te) {
        load_k(tile);
        load_v(tile);
    }
    while (tile < te) {
        const int64_t key0 = (int64_t)tile << 5;
        const int next = tile + TA_WAVES;
#pragma unroll
        for (int i = 0; i < VL; ++i) {
            const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
            *reinterpret_cast<uint4 *>(vimg + ta_v_off<D>(row, ch)) = vr[i];
        }
        if (next < te) load_v(next);

        // S^T = K Q^T
        ta_f32x16_t s[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            s[qt] = ta_f32x16_t{};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[qt][ks], s[qt], 0, 0, 0);
        }
        if (next < te) load_k(next);

        const bool full = key0 >= start && key0 + TA_TILE <= prev;   // wave-uniform: whole tile inside the visible prefix
        ta_bf16x8_t pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = NEG;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float x = s[qt][reg] * a.scale_log2;
                if (!full) {
                    const int64_t key = key0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    const int64_t j = key - prev;
                    const bool in_tree = j >= 0 && key < len && ((anc[qt] >> (j & 63)) & 1ull);
                    const bool vis = key >= start && (key < prev || in_tree);
                    x = vis ? x : NEG;
                }
                s[qt][reg] = x;
                mx = fmaxf(mx, x);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m[qt], mx);
            const float m_use = m_new == NEG ? 0.0f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m[qt] - m_use);
            m[qt] = m_new;
            float sum = 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float p = __builtin_amdgcn_exp2f(s[qt][reg] - m_use);
                s[qt][reg] = p;
                sum += p;
            }
            l[qt] = l[qt] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[qt][t][jj] = (__bf16)s[qt][8 * t + jj];
        }

        // O^T += V^T P^T
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int ch = 4 * dt + 2 * tg + (tp >> 1);
                const int row0 = 16 * t + 4 * hh + tq;
                typedef __attribute__((address_space(3))) ta_s16x4_t *lds_v4;
                const ta_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vimg + ta_v_off<D>(row0, ch) + 8 * (tp & 1)));
                const ta_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vimg + ta_v_off<D>(row0 + 8, ch) + 8 * (tp & 1)));
                struct { ta_s16x4_t lo, hi; } pair{lo, hi};
                const ta_bf16x8_t vf = __builtin_bit_cast(ta_bf16x8_t, pair);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][t], o[dt][qt], 0, 0, 0);
            }
        }
        tile = next;
    }

    // ---- merge the four waves: common max per query, rescale, add into one LDS tile
    float lf[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        lf[qt] = l[qt] + __shfl_xor(l[qt], 32);
        if (hh == 0) s_m[wave][32 * qt + r] = m[qt];
    }
    __syncthreads();
    // fixed wave order (plain read-modify-write, one owner lane per element and wave): the result does not depend on timing
    float fw[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qn = 32 * qt + r;
        float mt = s_m[0][qn];
#pragma unroll
        for (int w = 1; w < TA_WAVES; ++w) mt = fmaxf(mt, s_m[w][qn]);
        const float mu = mt == NEG ? 0.0f : mt;
        fw[qt] = __builtin_amdgcn_exp2f(m[qt] - mu);
    }
    for (int w = 0; w < TA_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int qn = 32 * qt + r;
                const float f = fw[qt];
                if (hh == 0) s_l[qn] = (w ? s_l[qn] : 0.0f) + lf[qt] * f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float *p = &s_o[qn][32 * dt + (reg & 3) + 8 * (reg >> 2) + 4 * hh];
                        *p = (w ? *p : 0.0f) + o[dt][qt][reg] * f;
                    }
            }
        }
        __syncthreads();
    }

    const int n_here = min(QR, a.N - q0);      // live query rows of this workgroup (>= 1: the host sizes q_groups from N)
    if (nsplit == 1) {
        for (int idx = tid; idx < n_here * CPR; idx += TA_THREADS) {
            const int qn = idx / CPR, ch = idx % CPR;
            const float inv = 1.0f / s_l[qn];
            ta_bf16x8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (__bf16)(s_o[qn][8 * ch + e] * inv);
            *reinterpret_cast<uint4 *>(a.out + (int64_t)b * a.o_sb + (int64_t)(q0 + qn) * a.o_sn + (int64_t)h * D + 8 * ch) =
                __builtin_bit_cast(uint4, ov);
        }
    } else {
        float *w = a.ws + (((int64_t)bh * a.q_groups + qs) * nsplit + split) * (int64_t)(QR * (D + 2));
        for (int idx = tid; idx < n_here * (D / 4); idx += TA_THREADS) {
            const int qn = idx / (D / 4), c4 = idx % (D / 4);
            *reinterpret_cast<float4 *>(w + qn * D + 4 * c4) = *reinterpret_cast<const float4 *>(&s_o[qn][4 * c4]);
        }
        if (tid < n_here) {
            float mt = s_m[0][tid];
#pragma unroll
            for (int wv = 1; wv < TA_WAVES; ++wv) mt = fmaxf(mt, s_m[wv][tid]);
            w[QR * D + tid] = mt;
            w[QR * D + QR + tid] = s_l[tid];
        }
    }
}

// second pass when the keys were split over gridDim.y (a last-arriver merge inside the main kernel was measured at 77 us
// against 20 + 9 us for this second launch: each workgroup's agent-scope fence is an L2 write-back + invalidate on gfx950; with
// device-coherent `sc1` stores / loads and an arrival counter instead of the fence -- round 3 -- the one kernel takes 20.9 us against
// 15.3 + 5.9 for the two and the drafter layer 136 against 134 us: the last workgroup's merge is serial where this launch is wide): grid (batch row x head x query group, chunks of 256/(D/4) query rows),
// one thread per (query row, 4 output columns); the loop over splits is a fixed order (deterministic) of independent loads
__global__ __launch_bounds__(256) void tree_attention_merge_kernel(const float *__restrict__ ws, uint16_t *__restrict__ out, int nsplit, int N,
                                                                   int D, int QR, int q_groups, int Hq, int64_t o_sb, int64_t o_sn) {
    const int bh = blockIdx.x / q_groups, qs = blockIdx.x % q_groups, b = bh / Hq, h = bh % Hq;
    const int q0 = qs * QR, n_here = min(QR, N - q0);
    const int per_row = D / 4, rows = 256 / per_row;
    const int qn = blockIdx.y * rows + (int)threadIdx.x / per_row, c4 = (int)threadIdx.x % per_row;
    if (qn >= n_here) return;
    const int64_t stride = (int64_t)QR * (D + 2);
    const float *base = ws + (int64_t)blockIdx.x * nsplit * stride;
    const float NEG = -__builtin_huge_valf();
    float mt = NEG;
    for (int s = 0; s < nsplit; ++s) mt = fmaxf(mt, base[s * stride + QR * D + qn]);
    const float mu = mt == NEG ? 0.0f : mt;
    float L = 0.0f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int s = 0; s < nsplit; ++s) {
        const float *w = base + s * stride;
        const float f = __builtin_amdgcn_exp2f(w[QR * D + qn] - mu);      // exp2(-inf) = 0: an empty split adds nothing
        const float lv = w[QR * D + QR + qn];
        const float4 v = *reinterpret_cast<const float4 *>(w + qn * D + 4 * c4);
        L += lv * f;
        acc.x += v.x * f, acc.y += v.y * f, acc.z += v.z * f, acc.w += v.w * f;
    }
    const float inv = 1.0f / L;
    uint16_t *op = out + (int64_t)b * o_sb + (int64_t)(q0 + qn) * o_sn + (int64_t)h * D + 4 * c4;
    const __bf16 r0 = (__bf16)(acc.x * inv), r1 = (__bf16)(acc.y * inv), r2 = (__bf16)(acc.z * inv), r3 = (__bf16)(acc.w * inv);
    uint2 pk;
    pk.x = (uint32_t)__builtin_bit_cast(uint16_t, r0) | ((uint32_t)__builtin_bit_cast(uint16_t, r1) << 16);
    pk.y = (uint32_t)__builtin_bit_cast(uint16_t, r2) | ((uint32_t)__builtin_bit_cast(uint16_t, r3) << 16);
    *reinterpret_cast<uint2 *>(op) = pk;
}

// launch shape: QT query tiles per workgroup x q_groups workgroups per (batch row, head)
struct TaShape {
    int QT, q_groups, QR;
};
static TaShape ta_shape(int N, int d) {
    TaShape s;
    // always 32 queries per workgroup; N > 32 -> two workgroups paired on one XCD.  (64 queries in one workgroup: 473 VGPRs at
    // d = 128; at d = 64 it fits 256 with 7 spills and measures 68 us against 57 us for the pair, 48 sequences.)
    (void)d;
    s.QT = 1, s.q_groups = N <= 32 ? 1 : 2;
    s.QR = 32 * s.QT;
    return s;
}

static int ta_splits(int B, int Hq, int q_groups, int64_t max_kv_len) {
    const int forced = tuning(TUNE_TA_SPLITS);
    const int64_t tiles = (max_kv_len + TA_TILE - 1) / TA_TILE;
    const int64_t wgs = (int64_t)B * Hq * q_groups;
    int64_t want = forced > 0 ? forced : 512 / wgs;                                          // one round of two resident workgroups per CU
    const int min_tiles = tuning(TUNE_TA_MIN_TILES) > 0 ? tuning(TUNE_TA_MIN_TILES) : 2;          // key tiles per wave and split below which no further split is made
    const int64_t cap = tiles / (min_tiles * TA_WAVES) > 1 ? tiles / (min_tiles * TA_WAVES) : 1;              // >= min_tiles tiles per wave and split
    if (want > cap) want = cap;
    if (want > 64) want = 64;
    return (int)(want < 1 ? 1 : want);
}
}  // namespace lantern

extern "C" size_t lantern_tree_attention_workspace(int B, int Hq, int N, int d, int64_t max_kv_len) {
    if (B <= 0 || Hq <= 0 || N <= 0 || d <= 0) return 0;
    const lantern::TaShape sh = lantern::ta_shape(N, d);
    const int ns = lantern::ta_splits(B, Hq, sh.q_groups, max_kv_len);
    if (ns == 1) return 0;
    return (size_t)B * Hq * sh.q_groups * ns * sh.QR * (d + 2) * sizeof(float);
}

extern "C" int lantern_tree_attention(const void *q, const void *k_cache, const void *v_cache, void *out, int B, int Hq, int Hkv, int N, int d,
                                      int64_t q_stride_b, int64_t q_stride_n, int64_t q_stride_h, int64_t kv_stride_b, int64_t kv_stride_h,
                                      int64_t out_stride_b, int64_t out_stride_n, const int64_t *kv_len, const int64_t *kv_start,
                                      int64_t max_kv_len, const uint64_t *tree_bits, int bits_per_row, float scale, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    using namespace lantern;
    if (B == 0) return LANTERN_OK;
    LANTERN_CHECK_ARG(q && k_cache && v_cache && out && tree_bits, "tree_attention: null pointer");
    LANTERN_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "tree_attention: bad head counts (B=%d Hq=%d Hkv=%d)", B, Hq, Hkv);
    LANTERN_CHECK_ARG(N >= 1 && N <= 64, "tree_attention: N=%d tree nodes, need 1..64 (one mask word per node)", N);
    LANTERN_CHECK_ARG(d == 64 || d == 128, "tree_attention: head_dim %d not built (64 or 128)", d);
    LANTERN_CHECK_ARG(max_kv_len >= N, "tree_attention: max_kv_len=%lld < N=%d (the N tree keys must already be appended)",
                      (long long)max_kv_len, N);
    LANTERN_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_n % 8 == 0 && q_stride_h % 8 == 0 && kv_stride_b % 8 == 0 && kv_stride_h % 8 == 0 &&
                          out_stride_b % 8 == 0 && out_stride_n % 8 == 0,
                      "tree_attention: strides must be multiples of 8 elements (16-byte rows)");
    LANTERN_CHECK_ARG(((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) % 16 == 0, "tree_attention: pointers must be 16-byte aligned");
    LANTERN_CHECK_ARG(kv_stride_h >= max_kv_len * d, "tree_attention: kv_stride_h=%lld smaller than max_kv_len*d", (long long)kv_stride_h);
    const TaShape sh = ta_shape(N, d);
    const int ns = ta_splits(B, Hq, sh.q_groups, max_kv_len);
    if (ns > 1) {
        const size_t need = (size_t)B * Hq * sh.q_groups * ns * sh.QR * (d + 2) * sizeof(float);
        LANTERN_CHECK_ARG(workspace && workspace_bytes >= need, "tree_attention: workspace of %zu bytes needed (lantern_tree_attention_workspace)", need);
        LANTERN_CHECK_ARG((uintptr_t)workspace % 16 == 0, "tree_attention: workspace must be 16-byte aligned");
    }
    TaArgs a;
    a.q = (const uint16_t *)q, a.k = (const uint16_t *)k_cache, a.v = (const uint16_t *)v_cache, a.out = (uint16_t *)out;
    a.kv_len = kv_len, a.kv_start = kv_start, a.bits = tree_bits, a.ws = (float *)workspace;
    a.q_sb = q_stride_b, a.q_sn = q_stride_n, a.q_sh = q_stride_h, a.kv_sb = kv_stride_b, a.kv_sh = kv_stride_h;
    a.o_sb = out_stride_b, a.o_sn = out_stride_n, a.max_kv_len = max_kv_len;
    a.B = B, a.Hq = Hq, a.Hkv = Hkv, a.N = N, a.bits_per_row = bits_per_row ? 1 : 0, a.q_groups = sh.q_groups;
    a.scale_log2 = scale * 1.4426950408889634f;
    const int gx = sh.q_groups == 2 ? ((B * Hq + 7) / 8) * 16 : B * Hq;
    const dim3 grid(gx, ns), block(TA_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (d == 128) LANTERN_LAUNCH((tree_attention_kernel<128, 1>), grid, block, 0, st, a);
    else LANTERN_LAUNCH((tree_attention_kernel<64, 1>), grid, block, 0, st, a);
    LANTERN_CHECK_LAUNCH("tree_attention");
    if (ns > 1) {
        const int rows = 256 / (d / 4);
        hipLaunchKernelGGL(tree_attention_merge_kernel, dim3(B * Hq * sh.q_groups, (sh.QR + rows - 1) / rows), dim3(256), 0, st, (const float *)workspace, (uint16_t *)out, ns,
                           N, d, sh.QR, sh.q_groups, Hq, out_stride_b, out_stride_n);
        LANTERN_CHECK_LAUNCH("tree_attention_merge");
    }
    return LANTERN_OK;
}
