// drafter_fc.hip -- O11: the drafter's input contraction on the matrix cores,
//   out[m, n] = sum_k cat(embed[ids[m]] * scale, hidden[m])[k] * W[n, k] + bias[n]      (k < 2H)
// replacing nn.Embedding + torch.cat + nn.Linear (models/drafters/cnets_lumina_mgpt.py:1071,1095-1098;
// cnets_llamagen.py:642,679-680).  The only MFMA-shaped op on the verify path.
//
// Shape: M = B*T is tiny (2..~120 rows), N = H, K = 2H: a weight-streaming GEMV-like GEMM bound by the
// 2*H*2H bytes of W (64 MiB bf16 for H = 4096).  Each wave owns 32 output columns and a K slice and feeds
// v_mfma_f32_32x32x16_bf16 STRAIGHT FROM GLOBAL MEMORY: lane (r = l&31, h = l>>5) needs
// A[row r][8h..8h+8) = 16 contiguous bytes of an activation row and B[8h..8h+8)[col r] = 16 contiguous bytes
// of W's row (n0 + r) -- the nn.Linear layout [out, in] is already the B fragment layout, so W is read
// exactly once, 16 bytes per lane, with no LDS staging and no transpose.  The embedding gather and the
// concat are folded into the A-fragment address (k < H -> embed row, else hidden row).  The 8 waves of a
// workgroup split K and add their 32x32 tiles into one LDS tile in wave order; bias + bf16 rounding in the
// epilogue.  C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "common.h"

namespace lantern {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int FC_WAVES = 8;
constexpr int FC_THREADS = FC_WAVES * 64;

__device__ __forceinline__ bf16x8_t load_frag(const uint16_t *p) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    return __builtin_bit_cast(bf16x8_t, v);
}

// the same load with the non-temporal hint (global_load_dwordx4 ... nt): a weight stream that should not displace what the caches hold
__device__ __forceinline__ bf16x8_t load_frag_nt(const uint16_t *p) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// COLS = output columns per workgroup: 32 fills the 32x32 MFMA tile; 16 (half of the tile's columns idle -- the matrix
// cores are nowhere near the limit of this weight-streaming kernel) doubles the workgroup count when H / 32 would leave
// CUs without work (H = 4096: 128 -> 256 workgroups).
template <int MT, int COLS>
__global__ __launch_bounds__(FC_THREADS) void drafter_fc_kernel(const int64_t *__restrict__ ids, const uint16_t *__restrict__ hidden,
                                                                const uint16_t *__restrict__ embed, const uint16_t *__restrict__ Wt,
                                                                const uint16_t *__restrict__ bias, int M, int H, int vocab,
                                                                float embed_scale, uint16_t *__restrict__ out) {
    __shared__ float tile[MT][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * COLS;
    const bool bcol = r < COLS;        // lanes of the tile's idle columns stream nothing
    const int K = 2 * H;
    for (int t = tid; t < MT * 32 * 33; t += FC_THREADS) (&tile[0][0][0])[t] = 0.0f;

    const int ksteps = K / 16;
    const int ks0 = (int)((long long)ksteps * wave / FC_WAVES), ks1 = (int)((long long)ksteps * (wave + 1) / FC_WAVES);
    const int ncol = n0 + r;
    const uint16_t *wrow = Wt + (size_t)(ncol < H ? ncol : H - 1) * K;
    const uint16_t *erow[MT];
    const uint16_t *hrow[MT];
    bool live[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 32 + r;
        live[mt] = row < M;
        int64_t id = live[mt] ? ids[row] : 0;
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        erow[mt] = embed + (size_t)id * H;
        hrow[mt] = hidden + (size_t)(live[mt] ? row : 0) * H;
    }
    f32x16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;

    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    // A fragment for 8 consecutive k starting at k0 of row-tile mt: embedding row (scaled, re-rounded to bf16) for k < H,
    // hidden row otherwise (k0 and H are multiples of 8: a fragment never straddles the seam)
    auto a_frag = [&](int mt, int k0) -> bf16x8_t {
        if (!live[mt]) return zero;
        if (k0 >= H) return load_frag(hrow[mt] + (k0 - H));
        bf16x8_t afrag = load_frag(erow[mt] + k0);
        if (embed_scale > 1.0f) {   // inputs_embeds * embed_upscale, rounded to bf16 (cnets_lumina_mgpt.py:1096-1097)
            uint4 v = __builtin_bit_cast(uint4, afrag);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint16_t lo = f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(w[i] & 0xffffu)) * embed_scale);
                const uint16_t hi = f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(w[i] >> 16)) * embed_scale);
                w[i] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
            afrag = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
        }
        return afrag;
    };
    // four K steps per trip with the k order permuted so that a lane streams 64 contiguous bytes of its weight row (see
    // linear_rows_kernel); the remainder in natural order
    int ks = ks0;
    for (; ks + 3 < ks1; ks += 4) {
        const int kb = ks * 16 + 32 * h;
        bf16x8_t bw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = bcol ? load_frag(wrow + kb + 8 * q) : zero;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8_t aw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) aw[q] = a_frag(mt, kb + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], bw[q], acc[mt], 0, 0, 0);
        }
    }
    for (; ks < ks1; ++ks) {
        const int k0 = ks * 16 + 8 * h;
        const bf16x8_t bfrag = bcol ? load_frag(wrow + k0) : zero;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_frag(mt, k0), bfrag, acc[mt], 0, 0, 0);
    }
    // combine the K slices in wave order (deterministic f32 sum; 8 short rounds)
    for (int w = 0; w < FC_WAVES; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    tile[mt][row][r] += acc[mt][reg];
                }
        }
    }
    __syncthreads();
    for (int t = tid; t < MT * 32 * 32; t += FC_THREADS) {
        const int mt = t / 1024, row = (t / 32) % 32, col = t % 32;
        const int m = mt * 32 + row, n = n0 + col;
        if (m < M && n < H && col < COLS) {
            float v = tile[mt][row][col];
            if (bias) v += bf16_bits_to_f32(bias[n]);
            out[(size_t)m * H + n] = f32_to_bf16_rne(v);
        }
    }
}

// 8f-2 (first half): the drafter's lm_head restricted to the vocabulary rows that can survive the model's mask.
//   out[m, col0 + n] = sum_k A[m, k] * W[row_lo + n, k]   (n < n_rows)
// For Lumina-mGPT / Anole every drafted row is masked to the image-token ids [4, 8196) right after the head
// (cnets_lumina_mgpt.py:1216-1224,1291-1298; cnets_anole.py:837,878), so 8192 of the 65536 rows of W are ever used:
// 64 MiB of weights per call instead of 512 MiB.  Same wave layout as drafter_fc_kernel: a wave feeds
// v_mfma_f32_32x32x16_bf16 straight from global memory (nn.Linear's [out, in] layout is the B-fragment layout), the
// 8 waves of a workgroup split K and add their tiles in LDS in wave order; bf16 output like nn.Linear in bf16.
// EPI 0: out = bf16(acc + bias).  EPI 1 (residual): out = bf16(bf16(acc + bias) + aux[m, n]) -- `residual + o_proj(x)` / `residual + down_proj(x)`
// with torch's two roundings.  EPI 2 (gated pair): the workgroup also contracts weight row `pair_rows` further down (up_proj under gate_proj
// in one concatenated weight) and writes bf16(silu(bf16 gate) * bf16 up) -- ChameleonMLP's `act_fn(gate_proj(x)) * up_proj(x)`, the [M, 2I]
// intermediate never in HBM.
template <int MT, int EPI = 0>
__global__ __launch_bounds__(FC_THREADS) void linear_rows_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Wt,
                                                                 const uint16_t *__restrict__ bias, int M, int K, int row_lo, int n_rows,
                                                                 uint16_t *__restrict__ out, int out_stride, int out_col0,
                                                                 const uint16_t *__restrict__ aux, int aux_stride, int pair_rows) {
    __shared__ float tile[MT][32][33];
    __shared__ float tile2[EPI == 2 ? MT : 1][EPI == 2 ? 32 : 1][EPI == 2 ? 33 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    for (int t = tid; t < MT * 32 * 33; t += FC_THREADS) (&tile[0][0][0])[t] = 0.0f;
    if constexpr (EPI == 2)
        for (int t = tid; t < MT * 32 * 33; t += FC_THREADS) (&tile2[0][0][0])[t] = 0.0f;
    const int ksteps = K / 16;
    const int ks0 = (int)((long long)ksteps * wave / FC_WAVES), ks1 = (int)((long long)ksteps * (wave + 1) / FC_WAVES);
    const int ncol = n0 + r;
    const uint16_t *wrow = Wt + (size_t)(row_lo + (ncol < n_rows ? ncol : n_rows - 1)) * K;
    const uint16_t *wrow2 = wrow + (size_t)pair_rows * K;
    const uint16_t *arow[MT];
    bool live[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 32 + r;
        live[mt] = row < M;
        arow[mt] = A + (size_t)(live[mt] ? row : 0) * K;
    }
    f32x16_t acc[MT], acc2[EPI == 2 ? MT : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < (EPI == 2 ? MT : 1); ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[mt][i] = 0.0f;
    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    // Four K steps (64 elements) per trip.  The order of k inside the contraction is free as long as A and B agree, so lane
    // (r, h) takes the CONTIGUOUS 64 bytes W[row r][k0 + 32h, +32) -- four back-to-back 16-byte loads of one half cache
    // line -- and feeds MFMA step s with its s-th 16 bytes (k = k0 + 32h + 8s ...), the A fragment taken at the same k.
    // A row's 128-byte line is then consumed whole by the two lanes that own it within one trip (instead of 32 bytes per
    // trip over four trips), and all weight loads of the trip are in flight before the first MFMA.
    int ks = ks0;
    for (; ks + 3 < ks1; ks += 4) {
        const int kb = ks * 16 + 32 * h;
        bf16x8_t bw[4], bw2[EPI == 2 ? 4 : 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = load_frag(wrow + kb + 8 * q);
        if constexpr (EPI == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bw2[q] = load_frag(wrow2 + kb + 8 * q);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8_t aw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) aw[q] = live[mt] ? load_frag(arow[mt] + kb + 8 * q) : zero;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], bw[q], acc[mt], 0, 0, 0);
            if constexpr (EPI == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], bw2[q], acc2[mt], 0, 0, 0);
            }
        }
    }
    for (; ks < ks1; ++ks) {
        const int k0 = ks * 16 + 8 * h;
        const bf16x8_t b0 = load_frag(wrow + k0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8_t a0 = live[mt] ? load_frag(arow[mt] + k0) : zero;
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[mt], 0, 0, 0);
            if constexpr (EPI == 2) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, load_frag(wrow2 + k0), acc2[mt], 0, 0, 0);
        }
    }
    for (int w = 0; w < FC_WAVES; ++w) {      // combine the K slices in wave order (deterministic f32 sum)
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    tile[mt][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] += acc[mt][reg];
                    if constexpr (EPI == 2) tile2[mt][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] += acc2[mt][reg];
                }
        }
    }
    __syncthreads();
    for (int t = tid; t < MT * 32 * 32; t += FC_THREADS) {
        const int mt = t / 1024, row = (t / 32) % 32, col = t % 32;
        const int m = mt * 32 + row, n = n0 + col;
        if (m < M && n < n_rows) {
            float v = tile[mt][row][col];
            if (bias) v += bf16_bits_to_f32(bias[row_lo + n]);
            uint16_t o = f32_to_bf16_rne(v);
            if constexpr (EPI == 1) o = f32_to_bf16_rne(bf16_bits_to_f32(aux[(size_t)m * aux_stride + n]) + bf16_bits_to_f32(o));
            if constexpr (EPI == 2) {
                float u = tile2[mt][row][col];
                if (bias) u += bf16_bits_to_f32(bias[row_lo + pair_rows + n]);
                const float gb = bf16_bits_to_f32(o), ub = bf16_bits_to_f32(f32_to_bf16_rne(u));
                const float sg = bf16_bits_to_f32(f32_to_bf16_rne(gb / (1.0f + expf(-gb))));       // torch: silu in f32 on the bf16 value, rounded to bf16
                o = f32_to_bf16_rne(sg * ub);
            }
            out[(size_t)m * out_stride + out_col0 + n] = o;
        }
    }
}

// ChameleonRMSNorm (cnets_lumina_mgpt.py:209-223) of M rows: f32 statistics, the normalised value rounded to the input dtype, then
// times the bf16 weight (rounded again) -- one workgroup per row.
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ w, int H, float eps,
                                                           uint16_t *__restrict__ out) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const uint16_t *xr = x + (size_t)row * H;
    constexpr int VMAX = 4;                       // 16-byte chunks a thread keeps in registers: rows up to 8192 elements are read once
    const bool vec = H % 8 == 0 && H <= 8 * 256 * VMAX && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0;
    uint4 xv[VMAX], wv[VMAX];
    float ss = 0.0f;
    if (vec) {
        // every load of the row and of the weight goes out before the first use (one memory round instead of 2 x H / 256 dependent ones)
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int i = (tid + j * 256) * 8;
            xv[j] = make_uint4(0u, 0u, 0u, 0u);
            wv[j] = make_uint4(0u, 0u, 0u, 0u);
            if (i < H) {
                xv[j] = *reinterpret_cast<const uint4 *>(xr + i);
                wv[j] = *reinterpret_cast<const uint4 *>(w + i);
            }
        }
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const uint32_t u[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {          // same element order as the scalar loop would visit within a thread; the sum is a tree anyway
                const float lo = bf16_bits_to_f32((uint16_t)(u[c] & 0xffffu)), hi = bf16_bits_to_f32((uint16_t)(u[c] >> 16));
                ss += lo * lo;
                ss += hi * hi;
            }
        }
    } else {
        for (int i = tid; i < H; i += 256) {
            const float v = bf16_bits_to_f32(xr[i]);
            ss += v * v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)H;
    const float rstd = rsqrtf(var + eps);
    if (vec) {
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int i = (tid + j * 256) * 8;
            if (i < H) {
                const uint32_t u[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, g[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w};
                uint32_t o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float nl = bf16_bits_to_f32(f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(u[c] & 0xffffu)) * rstd));
                    const float nh = bf16_bits_to_f32(f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(u[c] >> 16)) * rstd));
                    o[c] = (uint32_t)f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(g[c] & 0xffffu)) * nl) |
                           ((uint32_t)f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(g[c] >> 16)) * nh) << 16);
                }
                *reinterpret_cast<uint4 *>(out + (size_t)row * H + i) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        return;
    }
    for (int i = tid; i < H; i += 256) {
        const float n = bf16_bits_to_f32(f32_to_bf16_rne(bf16_bits_to_f32(xr[i]) * rstd));
        out[(size_t)row * H + i] = f32_to_bf16_rne(bf16_bits_to_f32(w[i]) * n);
    }
}

// ChameleonAttention's head stage for decode-sized inputs (cnets_lumina_mgpt.py:481-499): per (token, head) layer-norm over head_dim
// (ChameleonLayerNorm: one gamma / beta row per model-parallel shard), rotary embedding at the token's position, written in the
// [B, heads, T, d] layout attention consumes; V only changes layout.  qkv [B*T, (nq + 2 nk) d] bf16 (the fused projection);
// cos / sin [max_pos, d] bf16 tables; one wave per (token, head), lane i holds elements i and i + 64 (d = 128) or i (d = 64).
template <int D>
__global__ __launch_bounds__(64) void qk_norm_rope_kernel(const uint16_t *__restrict__ qkv, int T, int nq, int nk, const uint16_t *__restrict__ qw,
                                                          const uint16_t *__restrict__ qb, const uint16_t *__restrict__ kw, const uint16_t *__restrict__ kb,
                                                          int q_heads_per_mp, int k_heads_per_mp, const uint16_t *__restrict__ cos_t,
                                                          const uint16_t *__restrict__ sin_t, const int64_t *__restrict__ pos, uint16_t *__restrict__ q_out,
                                                          uint16_t *__restrict__ k_out, uint16_t *__restrict__ v_out, int kv_rows, int kv_row0, int table_rows,
                                                          int q_rows, int q_row0) {
    constexpr int E = D / 64;
    const int tok = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;          // head in [0, nq + 2 nk)
    const int b = tok / T, t = tok % T;
    const uint16_t *src = qkv + (size_t)tok * (nq + 2 * nk) * D + (size_t)head * D;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = bf16_bits_to_f32(src[lane + 64 * e]);
    if (head >= nq + nk) {          // V: layout only
        const int hv = head - nq - nk;
        uint16_t *dst = v_out + (((size_t)b * nk + hv) * kv_rows + kv_row0 + t) * D;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[lane + 64 * e] = src[lane + 64 * e];
        return;
    }
    const bool is_q = head < nq;
    const int hh = is_q ? head : head - nq;
    // layer norm over head_dim: f32 statistics (torch.layer_norm on bf16 computes in f32), output rounded to bf16
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += v[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) ss += (v[e] - mean) * (v[e] - mean);
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss / (float)D + 1e-5f);
    const uint16_t *gw = (is_q ? qw : kw) + (size_t)(hh / (is_q ? q_heads_per_mp : k_heads_per_mp)) * D;
    const uint16_t *gb = (is_q ? qb : kb) + (size_t)(hh / (is_q ? q_heads_per_mp : k_heads_per_mp)) * D;
    float n[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float ln = bf16_bits_to_f32(f32_to_bf16_rne((v[e] - mean) * rstd));
        const float sc = bf16_bits_to_f32(f32_to_bf16_rne(ln * bf16_bits_to_f32(gw[lane + 64 * e])));
        n[e] = bf16_bits_to_f32(f32_to_bf16_rne(sc + bf16_bits_to_f32(gb[lane + 64 * e])));
    }
    // rotary: x * cos + rotate_half(x) * sin, rotate_half(x)[i] = -x[i + d/2] (i < d/2), x[i - d/2] otherwise; every product and the sum in bf16
    int64_t p = pos[(size_t)b * T + t];
    p = p < 0 ? 0 : (p >= table_rows ? table_rows - 1 : p);          // (the reference would raise on a position beyond its tables; never read outside ours)
    const uint16_t *cr = cos_t + (size_t)p * D, *sr = sin_t + (size_t)p * D;
    uint16_t *dst = (is_q ? q_out + (((size_t)b * nq + hh) * q_rows + q_row0 + t) * D : k_out + (((size_t)b * nk + hh) * kv_rows + kv_row0 + t) * D);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        float other;          // x[(i + d/2) mod d]
        if constexpr (E == 2) other = n[e ^ 1];                       // d = 128: the partner element sits in the same lane's other slot
        else other = __shfl_xor(n[0], 32, 64);                        // d = 64: lane i <-> lane i ^ 32
        const float rot = (i < D / 2) ? -other : other;
        const float a = bf16_bits_to_f32(f32_to_bf16_rne(n[e] * bf16_bits_to_f32(cr[i])));
        const float c = bf16_bits_to_f32(f32_to_bf16_rne(rot * bf16_bits_to_f32(sr[i])));
        dst[i] = f32_to_bf16_rne(a + c);
    }
}

// LlamaAttention's head stage for decode-sized inputs (cnets_llamagen.py:315-323): no per-head norm; rotary on ADJACENT pairs
// (x[2p], x[2p+1]) -> (x0 c - x1 s, x1 c + x0 s) with (c, s) = freqs[pos][p] from the model's precomputed f32 table (LlamaGen: the 2-D table of
// precompute_freqs_cis_2d, :47-64 -- whatever the table holds, the kernel only indexes it by position), computed in f32 on the bf16 inputs and
// rounded to bf16 once (`x.float() ... type_as(x)`), written in the [B, heads, T, d] layout attention consumes; V only changes layout.
// qkv [B*T, (nq + 2 nk) d] bf16 (the fused projection); one wave per (token, head), lane i holds elements i and i + 64 (d = 128) or i (d = 64).
template <int D>
__global__ __launch_bounds__(64) void qk_rope_pairs_kernel(const uint16_t *__restrict__ qkv, int T, int nq, int nk, const float *__restrict__ freqs,
                                                           const int64_t *__restrict__ pos, int pos_per_batch, uint16_t *__restrict__ q_out,
                                                           uint16_t *__restrict__ k_out, uint16_t *__restrict__ v_out, int kv_rows, int kv_row0, int table_rows,
                                                           int q_rows, int q_row0) {
    constexpr int E = D / 64;
    const int tok = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;          // head in [0, nq + 2 nk)
    const int b = tok / T, t = tok % T;
    const uint16_t *src = qkv + (size_t)tok * (nq + 2 * nk) * D + (size_t)head * D;
    if (head >= nq + nk) {          // V: layout only
        const int hv = head - nq - nk;
        uint16_t *dst = v_out + (((size_t)b * nk + hv) * kv_rows + kv_row0 + t) * D;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[lane + 64 * e] = src[lane + 64 * e];
        return;
    }
    const bool is_q = head < nq;
    const int hh = is_q ? head : head - nq;
    int64_t p = pos[pos_per_batch ? (size_t)b * T + t : (size_t)t];
    p = p < 0 ? 0 : (p >= table_rows ? table_rows - 1 : p);          // (the reference would raise on a position beyond its table; never read outside ours)
    const float *fr = freqs + (size_t)p * D;                         // [d/2][2]
    uint16_t *dst = (is_q ? q_out + (((size_t)b * nq + hh) * q_rows + q_row0 + t) * D : k_out + (((size_t)b * nk + hh) * kv_rows + kv_row0 + t) * D);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        const float x = bf16_bits_to_f32(src[i]);
        const float y = __shfl_xor(x, 1, 64);                        // the pair's other element (lane ^ 1: pairs are adjacent elements)
        const float c = fr[(i >> 1) * 2], sn = fr[(i >> 1) * 2 + 1];
        // even element: x0 c - x1 s; odd element: x1 c + x0 s (two products, one sum, each rounded to f32 like torch's separate ops)
        const float a = x * c, bq = y * sn;
        dst[i] = f32_to_bf16_rne((i & 1) ? a + bq : a - bq);
    }
}

// 8f-2 (second half, phase 1): the same contraction for the drafter's cond + uncond rows, with the CFG combination as the
// epilogue -- rows 0..n-1 of A are the conditional hidden states, rows n..2n-1 the unconditional ones
// (cnets_lumina_mgpt.py:1271-1320: `out = uncond + cfg_scale * (cond - uncond)` on the head's bf16 logits).  The head's
// [2, n, V] logits never exist: what leaves the kernel is the combined window [n, n_cols] in bf16 (16 KB per row for the
// 8192 image ids), every rounding where torch rounds (the nn.Linear output, then each step of the bf16 arithmetic).
__global__ __launch_bounds__(FC_THREADS) void linear_rows_cfg_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Wt,
                                                                     const uint16_t *__restrict__ bias, int n, int K, int row_lo, int n_cols,
                                                                     float cfg, uint16_t *__restrict__ win) {
    __shared__ float tile[32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, M = 2 * n;
    for (int t = tid; t < 32 * 33; t += FC_THREADS) (&tile[0][0])[t] = 0.0f;
    const int ksteps = K / 16;
    const int ks0 = (int)((long long)ksteps * wave / FC_WAVES), ks1 = (int)((long long)ksteps * (wave + 1) / FC_WAVES);
    const int ncol = n0 + r;
    const uint16_t *wrow = Wt + (size_t)(row_lo + (ncol < n_cols ? ncol : n_cols - 1)) * K;
    const bool live = r < M;
    const uint16_t *arow = A + (size_t)(live ? r : 0) * K;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    int ks = ks0;
    for (; ks + 3 < ks1; ks += 4) {          // same K order as linear_rows_kernel: the logits are the same bits
        const int kb = ks * 16 + 32 * h;
        bf16x8_t bw[4], aw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = load_frag(wrow + kb + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) aw[q] = live ? load_frag(arow + kb + 8 * q) : zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], bw[q], acc, 0, 0, 0);
    }
    for (; ks < ks1; ++ks) {
        const int k0 = ks * 16 + 8 * h;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(live ? load_frag(arow + k0) : zero, load_frag(wrow + k0), acc, 0, 0, 0);
    }
    for (int w = 0; w < FC_WAVES; ++w) {      // combine the K slices in wave order (deterministic f32 sum)
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) tile[(reg & 3) + 8 * (reg >> 2) + 4 * h][r] += acc[reg];
        }
    }
    __syncthreads();
    for (int t = tid; t < n * 32; t += FC_THREADS) {
        const int i = t / 32, col = t % 32, nn = n0 + col;
        if (nn < n_cols) {
            const float b = bias ? bf16_bits_to_f32(bias[row_lo + nn]) : 0.0f;
            const float c = bf16_bits_to_f32(f32_to_bf16_rne(tile[i][col] + b)), u = bf16_bits_to_f32(f32_to_bf16_rne(tile[i + n][col] + b));
            const float o = round_bf16(u + round_bf16(cfg * round_bf16(c - u)));
            win[(size_t)i * n_cols + nn] = (uint16_t)(__float_as_uint(o) >> 16);
        }
    }
}

int launch_linear_rows_cfg(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, float cfg, void *win,
                           hipStream_t st) {
    LANTERN_LAUNCH(linear_rows_cfg_kernel, dim3((n_cols + 31) / 32), dim3(FC_THREADS), 0, st, (const uint16_t *)A, (const uint16_t *)W,
                   (const uint16_t *)bias, n, K, row_lo, n_cols, cfg, (uint16_t *)win);
    return 0;
}

// a5: Model._prepare_decoder_attention_mask (cnets_lumina_mgpt.py:1014-1050, cnets_llamagen.py:592-621) in one launch:
// The reductions a drafting call makes over its attention mask, in one launch: first[b] = index of the first non-zero entry (0 for an all-zero row:
// torch.argmax), count[b] = number of non-zero entries (= position_ids[:, -1] + 1 of `mask.cumsum(-1) - 1`, cnets_lumina_mgpt.py:1180-1186),
// bad[b] = 1 when a zero follows a one in row b (not left padding), else 0.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void mask_left_padding_kernel(const T *__restrict__ mask, int64_t S, int64_t row_stride, int64_t *__restrict__ first,
                                                                int64_t *__restrict__ count, int64_t *__restrict__ bad) {
    __shared__ long long s_first[4], s_cnt[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T *row = mask + (size_t)b * row_stride;
    long long f = 0x7fffffffffffffffll, c = 0;
    for (int64_t j = tid; j < S; j += 256)
        if (row[j] != T(0)) {
            f = f < j ? f : j;
            ++c;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long of = __shfl_xor(f, o, 64), oc = __shfl_xor(c, o, 64);
        f = f < of ? f : of;
        c += oc;
    }
    if (lane == 0) {
        s_first[wave] = f;
        s_cnt[wave] = c;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            f = f < s_first[w] ? f : s_first[w];
            c += s_cnt[w];
        }
        const bool none = c == 0;
        if (first) first[b] = none ? 0 : f;
        if (count) count[b] = c;
        if (bad) bad[b] = (!none && c != S - f) ? 1 : 0;
    }
}

// out[b,0,i,j] = padding(b,j) + causal(i,j) with padding = 0 / finfo.min from the boolean mask (columns beyond its length
// count as attended), causal = finfo.min for j - past > i when T > 1 (the reference ADDS the two, so a position masked by
// both is -inf), then finfo.min wherever the tree mask (last t0 rows x last t1 columns) is zero.
__global__ void drafter_mask_kernel(const uint8_t *__restrict__ attn, int attn_len, const float *__restrict__ tree, int tree_batch,
                                    int t0, int t1, int B, int T, int past, float *__restrict__ out) {
    const int S = past + T;
    const float FMIN = -3.4028234663852886e38f;
    const size_t total = (size_t)B * T * S;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(w % S), i = (int)((w / S) % T), b = (int)(w / ((size_t)S * T));
        float v = 0.0f;
        bool have = false;
        if (T > 1) {
            v = (j - past > i) ? FMIN : 0.0f;
            have = true;
        }
        if (attn) {
            const bool keep = j >= attn_len || attn[(size_t)b * attn_len + j] != 0;
            const float a = keep ? 0.0f : FMIN;
            v = have ? a + v : a;
        }
        if (tree && i >= T - t0 && j >= S - t1) {
            const int tb = tree_batch > 1 ? b : 0;
            if (tree[((size_t)tb * t0 + (i - (T - t0))) * t1 + (j - (S - t1))] == 0.0f) v = FMIN;
        }
        out[w] = v;
    }
}

}  // namespace lantern

using namespace lantern;

extern "C" int lantern_mask_left_padding(const void *mask, int elem_bytes, int B, int64_t S, int64_t row_stride, int64_t *first, int64_t *count, int64_t *bad,
                                         void *stream) {
    LANTERN_CHECK_ARG(mask && (elem_bytes == 1 || elem_bytes == 8) && B >= 0 && S >= 0 && row_stride >= S, "mask_left_padding: mask [B, S] of 1- or 8-byte entries");
    if (B == 0) return LANTERN_OK;
    if (elem_bytes == 1)
        hipLaunchKernelGGL(lantern::mask_left_padding_kernel<uint8_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)mask, S, row_stride, first, count, bad);
    else
        hipLaunchKernelGGL(lantern::mask_left_padding_kernel<int64_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const int64_t *)mask, S, row_stride, first, count, bad);
    LANTERN_CHECK_LAUNCH("mask_left_padding");
    return LANTERN_OK;
}

extern "C" int lantern_drafter_attention_mask(const uint8_t *attn, int attn_len, const float *tree_mask, int tree_batch, int t0, int t1,
                                              int B, int T, int past, float *out, void *stream) {
    LANTERN_CHECK_ARG(out && B > 0 && T > 0 && past >= 0, "drafter_attention_mask: bad sizes");
    LANTERN_CHECK_ARG(attn || T > 1, "drafter_attention_mask: no padding mask and a single query row: the reference builds no mask");
    if (tree_mask) LANTERN_CHECK_ARG(t0 > 0 && t1 > 0 && t0 <= T && t1 <= past + T && (tree_batch == 1 || tree_batch == B),
                                     "drafter_attention_mask: tree mask [%d,1,%d,%d] does not fit [%d,1,%d,%d]", tree_batch, t0, t1, B, T, past + T);
    const size_t total = (size_t)B * T * (past + T);
    int gx = (int)((total + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(drafter_mask_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, attn, attn_len, tree_mask, tree_batch, t0, t1, B, T,
                       past, out);
    LANTERN_CHECK_LAUNCH("drafter_attention_mask");
    return LANTERN_OK;
}

extern "C" int lantern_drafter_fc(const int64_t *ids, const void *hidden, const void *embed, const void *W, const void *bias, int M, int H,
                                  int vocab, float embed_scale, void *out, void *stream) {
    LANTERN_CHECK_ARG(ids && hidden && embed && W && out, "drafter_fc: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 128, "drafter_fc: M=%d must be <= 128 rows (B*T of one drafter call)", M);
    LANTERN_CHECK_ARG(H > 0 && H % 16 == 0 && vocab > 0, "drafter_fc: H=%d must be a multiple of 16", H);
    if (M == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = (H + 31) / 32 < 256;          // fewer column tiles than CUs: 16 columns per workgroup
    dim3 grid(narrow ? (H + 15) / 16 : (H + 31) / 32), block(FC_THREADS);
    const uint16_t *h = (const uint16_t *)hidden, *e = (const uint16_t *)embed, *w = (const uint16_t *)W, *bi = (const uint16_t *)bias;
    uint16_t *o = (uint16_t *)out;
    if (M <= 32) {
        if (narrow) hipLaunchKernelGGL((drafter_fc_kernel<1, 16>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
        else hipLaunchKernelGGL((drafter_fc_kernel<1, 32>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
    }
    else if (M <= 64) {
        if (narrow) hipLaunchKernelGGL((drafter_fc_kernel<2, 16>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
        else hipLaunchKernelGGL((drafter_fc_kernel<2, 32>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
    }
    else if (M <= 96) {
        if (narrow) hipLaunchKernelGGL((drafter_fc_kernel<3, 16>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
        else hipLaunchKernelGGL((drafter_fc_kernel<3, 32>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
    }
    else {
        if (narrow) hipLaunchKernelGGL((drafter_fc_kernel<4, 16>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
        else hipLaunchKernelGGL((drafter_fc_kernel<4, 32>), grid, block, 0, st, ids, h, e, w, bi, M, H, vocab, embed_scale, o);
    }
    LANTERN_CHECK_LAUNCH("drafter_fc");
    return LANTERN_OK;
}

// Split-K form for narrow outputs (o_proj / down_proj: 4096 columns = 128 workgroups of 32 columns on 256 CUs, each streaming 0.25 - 0.7 MB
// alone): grid (column tiles, ksplit), workgroup (x, y) contracts K slice y of its 32 columns into workspace[y][M][n_rows] (f32, plain
// stores: every element has one writer), then a small pass adds the slices in order, the bias, the residual -- deterministic.
__global__ __launch_bounds__(FC_THREADS) void linear_rows_splitk_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Wt, int M, int K,
                                                                        int n_rows, int ksplit, float *__restrict__ ws) {
    __shared__ float tile[32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, ky = blockIdx.y;
    for (int t = tid; t < 32 * 33; t += FC_THREADS) (&tile[0][0])[t] = 0.0f;
    const int ksteps = K / 16;
    const int kb0 = (int)((long long)ksteps * ky / ksplit), kb1 = (int)((long long)ksteps * (ky + 1) / ksplit), kn = kb1 - kb0;
    const int ks0 = kb0 + (int)((long long)kn * wave / FC_WAVES), ks1 = kb0 + (int)((long long)kn * (wave + 1) / FC_WAVES);
    const int ncol = n0 + r;
    const uint16_t *wrow = Wt + (size_t)(ncol < n_rows ? ncol : n_rows - 1) * K;
    const bool live = r < M;
    const uint16_t *arow = A + (size_t)(live ? r : 0) * K;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    int ks = ks0;
    for (; ks + 3 < ks1; ks += 4) {
        const int kb = ks * 16 + 32 * h;
        bf16x8_t bw[4], aw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = load_frag(wrow + kb + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) aw[q] = live ? load_frag(arow + kb + 8 * q) : zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], bw[q], acc, 0, 0, 0);
    }
    for (; ks < ks1; ++ks) {
        const int k0 = ks * 16 + 8 * h;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(live ? load_frag(arow + k0) : zero, load_frag(wrow + k0), acc, 0, 0, 0);
    }
    for (int w = 0; w < FC_WAVES; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) tile[(reg & 3) + 8 * (reg >> 2) + 4 * h][r] += acc[reg];
        }
    }
    __syncthreads();
    for (int t = tid; t < 32 * 32; t += FC_THREADS) {
        const int row = t / 32, col = t % 32, n = n0 + col;
        if (row < M && n < n_rows) ws[((size_t)ky * M + row) * n_rows + n] = tile[row][col];
    }
}

__global__ __launch_bounds__(256) void linear_rows_splitk_finish_kernel(const float *__restrict__ ws, const uint16_t *__restrict__ bias,
                                                                        const uint16_t *__restrict__ residual, int residual_stride, int M, int n_rows,
                                                                        int ksplit, uint16_t *__restrict__ out, int out_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * n_rows) return;
    const int m = i / n_rows, n = i % n_rows;
    float v = 0.0f;
    for (int y = 0; y < ksplit; ++y) v += ws[((size_t)y * M + m) * n_rows + n];
    if (bias) v += bf16_bits_to_f32(bias[n]);
    uint16_t o = f32_to_bf16_rne(v);
    if (residual) o = f32_to_bf16_rne(bf16_bits_to_f32(residual[(size_t)m * residual_stride + n]) + bf16_bits_to_f32(o));
    out[(size_t)m * out_stride + n] = o;
}

extern "C" int lantern_linear_rows_splitk(const void *A, const void *W, const void *bias, int M, int K, int n_rows, void *out, int out_stride,
                                          const void *residual, int residual_stride, int ksplit, float *workspace, void *stream) {
    LANTERN_CHECK_ARG(A && W && out && workspace, "linear_rows_splitk: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 32 && K > 0 && K % 16 == 0 && n_rows >= 0 && out_stride >= n_rows && ksplit >= 1 && ksplit <= 16 && K / 16 >= ksplit,
                      "linear_rows_splitk: M=%d <= 32, K=%d %% 16 == 0, 1 <= ksplit=%d <= 16", M, K, ksplit);
    if (residual) LANTERN_CHECK_ARG(residual_stride >= n_rows, "linear_rows_splitk: residual [M, stride >= n_rows]");
    if (M == 0 || n_rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    LANTERN_LAUNCH(linear_rows_splitk_kernel, dim3((n_rows + 31) / 32, ksplit), dim3(FC_THREADS), 0, st, (const uint16_t *)A, (const uint16_t *)W, M, K,
                   n_rows, ksplit, workspace);
    LANTERN_CHECK_LAUNCH("linear_rows_splitk");
    hipLaunchKernelGGL(linear_rows_splitk_finish_kernel, dim3((M * n_rows + 255) / 256), dim3(256), 0, st, (const float *)workspace,
                       (const uint16_t *)bias, (const uint16_t *)residual, residual_stride, M, n_rows, ksplit, (uint16_t *)out, out_stride);
    LANTERN_CHECK_LAUNCH("linear_rows_splitk_finish");
    return LANTERN_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Stream-K form of the same product for the drafting shape (M <= 32): the decoder layer's four weight matrices are 34 - 180 MB each
// and nothing else of the layer moves more than a few hundred KB, so the layer runs at the speed the weights stream.  Two things
// bound the per-tile kernels above: (1) a workgroup owns a 32-column tile, so 128 / 344 / 384 tiles land unevenly on 256 CUs, and the
// split-K variant pays a second launch; (2) one trip of loads in flight per wave (64 B per lane) is ~32 KB per CU -- at ~2.5 us of
// loaded HBM latency that is 3.3 TB/s for the whole GPU, which is what they measure.  Here
//  * the (tile, K) space is cut into `cpt` chunks of SK_CHUNK K-elements per tile, linearised, and workgroup g of G gets the contiguous
//    range [total g / G, total (g + 1) / G): every workgroup streams the same number of bytes whatever the tile count (G = one workgroup per CU: sk_groups; 320 - 768 measured slower with the non-temporal weight streams, round 5);
//  * a range that ends inside a tile leaves an f32 partial tile in the workspace (at most two per workgroup: the tail of its first
//    tile, the head of its last) and adds its chunk count to the tile's counter; the workgroup that completes the count sums the
//    tile's partials in K order -- always the same order, whoever arrives last -- and runs the epilogue: ONE launch, deterministic;
//  * every wave keeps TWO trips in flight (the next trip's 64 B of W and of A per lane are requested before the current trip's MFMAs).
//  * PACKED: the weight re-laid out once, at load time, in the order the waves consume it (lantern_pack_linear_weight: per 32-row tile
//    and 64-element K block a 4 KB brick [fragment q][lane][8 bf16]) -- a load instruction of a wave is 1 KB contiguous and a workgroup's
//    whole share one contiguous byte range, where nn.Linear's [out, in] rows make every load 64 separate 16-byte pieces 8 - 22 KB apart
//    (measured 3.3 TB/s at best, whatever the occupancy or the prefetch depth: the DRAM pages, not the latency, were the limit).
// Epilogues as in linear_rows_kernel (EPI 0 bias, 1 + residual, 2 gate / up pair with silu * mul; torch's bf16 roundings).
constexpr int SK_CHUNK = 256;

struct SkArgs {
    const uint16_t *A, *W, *bias, *aux;
    uint16_t *out;
    float *ws;            // [G][2][NSET][32 * 32] partial tiles
    uint32_t *cnt;        // [n_tiles] chunks accumulated (zero before the launch; left zero)
    int M, K, row_lo, n_rows, out_stride, out_col0, aux_stride, pair_rows, n_tiles, cpt, G;
    // activation rows in segments: row m lives at A + (m / a_seg_rows) * a_seg_stride + (m % a_seg_rows) * K (elements); a_seg_rows = 0: one [M, K] matrix.
    // (lantern_draft_depth: the attention output of the cond / uncond rows sits in two [64, H] slots, the last T rows of each are this depth's)
    int a_seg_rows;
    long long a_seg_stride;
    // GATHER (the drafter's input stage, lantern_drafter_fc): A[m] = cat(embed[ids[m]] * embed_scale, hidden[m]) with K = 2 * hsplit; `A` is hidden
    const int64_t *ids;
    const uint16_t *embed;
    int vocab, hsplit;
    // GATHER through a static tree's tables (lantern_draft_depth in_rep): row r = (b, j) of rows_per_b takes token ids[in_gather[j]] and hidden row b * src_T + in_rep[j]
    const int32_t *in_gather, *in_rep;
    int rows_per_b, src_T, n_flat;
    int w_stream;         // 1: the weights are read with the non-temporal hint (sk_run: matrices of 80 MB and more)
    float embed_scale, cfg;          // cfg: EPI 3 (rows [0, M/2) conditional, [M/2, M) unconditional -> uncond + cfg * (cond - uncond) in bf16 steps)
};

template <int EPI, bool PACKED, int NBUF, bool GATHER = false, bool WNT = false>
__global__ __launch_bounds__(FC_THREADS) void linear_rows_streamk_kernel(const SkArgs a) {
    constexpr int NSET = EPI == 2 ? 2 : 1;
    __shared__ float red[NSET][FC_WAVES][32][33];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int K = a.K, cpt = a.cpt, G = a.G, wg = blockIdx.x;
    const long long total = (long long)a.n_tiles * cpt;
    long long c0 = total * wg / G;
    const long long c1 = total * (wg + 1) / G;
    const int first_tile = (int)(c0 / cpt);
    const bool live = r < a.M;
    const int seg = (a.a_seg_rows > 0 && live) ? r / a.a_seg_rows : 0;
    const uint16_t *arow = a.A + (size_t)seg * a.a_seg_stride + (size_t)(live ? r - seg * a.a_seg_rows : 0) * (GATHER ? a.hsplit : K);
    const uint16_t *erow = nullptr;
    if constexpr (GATHER) {
        int idx = r;
        if (a.in_rep && live) {          // tokens / parent rows one level up, read through the tree's tables
            const int b = r / a.rows_per_b, j = r - b * a.rows_per_b;
            int par = a.in_rep[j], gi = a.in_gather[j];
            par = par < 0 ? 0 : (par >= a.src_T ? a.src_T - 1 : par);
            idx = gi < 0 ? 0 : (gi >= a.n_flat ? a.n_flat - 1 : gi);
            arow = a.A + ((size_t)b * a.src_T + par) * a.hsplit;
        }
        int64_t id = live ? a.ids[idx] : 0;
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        erow = a.embed + (size_t)id * a.hsplit;
    }
    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    // 8 consecutive k of this lane's activation row (GATHER: the embedding row, scaled and re-rounded to bf16 as
    // cnets_lumina_mgpt.py:1096-1097 does, for k < hsplit, the hidden row behind it; hsplit % 64 == 0: a trip never straddles the seam)
    auto a_frag = [&](int k0) -> bf16x8_t {
        if (!live) return zero;
        if constexpr (!GATHER) {
            return load_frag(arow + k0);
        } else {
            if (k0 >= a.hsplit) return load_frag(arow + (k0 - a.hsplit));
            bf16x8_t af = load_frag(erow + k0);
            if (a.embed_scale > 1.0f) {
                const uint4 v = __builtin_bit_cast(uint4, af);
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint16_t lo = f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(w[i] & 0xffffu)) * a.embed_scale);
                    const uint16_t hi = f32_to_bf16_rne(bf16_bits_to_f32((uint16_t)(w[i] >> 16)) * a.embed_scale);
                    w[i] = (uint32_t)lo | ((uint32_t)hi << 16);
                }
                af = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
            }
            return af;
        }
    };
    while (c0 < c1) {
        const int t = (int)(c0 / cpt);
        const int cb = (int)(c0 - (long long)t * cpt);
        const long long tile_end = (long long)(t + 1) * cpt;
        const int ce = (int)((c1 < tile_end ? c1 : tile_end) - (long long)t * cpt);
        const int k_lo = cb * SK_CHUNK, k_hi = ce * SK_CHUNK < K ? ce * SK_CHUNK : K;
        // waves split the segment's K range.  Row-major weights: contiguous sub-ranges at any 16-element step boundary.  Packed weights: the
        // 64-element bricks round-robin (wave w takes bricks w, w + 8, ...), so the eight waves walk ONE contiguous byte range together
        // (32 KB per round) instead of eight separate ones: 256 sequential streams on the GPU, not 2048.
        const int ksteps = (k_hi - k_lo) / 16;
        const int ks0 = PACKED ? (k_lo / 64 + wave) * 4 : k_lo / 16 + (int)((long long)ksteps * wave / FC_WAVES);
        const int ks1 = PACKED ? (k_hi / 64) * 4 : k_lo / 16 + (int)((long long)ksteps * (wave + 1) / FC_WAVES);
        constexpr int KSTRIDE = PACKED ? 4 * FC_WAVES : 4;          // steps from one trip of a wave to its next
        const int ncol = t * 32 + r;
        const uint16_t *wrow = a.W + (size_t)(a.row_lo + (ncol < a.n_rows ? ncol : a.n_rows - 1)) * K;
        const uint16_t *wrow2 = wrow + (size_t)a.pair_rows * K;
        f32x16_t acc[NSET];
#pragma unroll
        for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s_][i] = 0.0f;
        // ---- the K range of this wave, four MFMA steps (64 elements, 64 contiguous bytes per lane and row) per trip, NBUF trips in flight
        // (a ring of register buffers with compile-time indices: a buffer is refilled as soon as its MFMAs have read it)
        bf16x8_t wb[NBUF][NSET][4], ab[NBUF][4];
        auto load_trip = [&](int ks, bf16x8_t (&wv)[NSET][4], bf16x8_t (&av)[4]) {
            const int kb = ks * 16 + 32 * h;
            if constexpr (PACKED) {          // brick (tile t, K block ks / 4, set): [q][lane][8]
                const uint16_t *brick = a.W + ((size_t)((long long)t * (K / 64) + (ks >> 2)) * NSET) * 2048 + lane * 8;
#pragma unroll
                for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                    for (int q = 0; q < 4; ++q) wv[s_][q] = WNT ? load_frag_nt(brick + s_ * 2048 + q * 512) : load_frag(brick + s_ * 2048 + q * 512);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[0][q] = load_frag(wrow + kb + 8 * q);
                if constexpr (NSET == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) wv[1][q] = load_frag(wrow2 + kb + 8 * q);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = a_frag(kb + 8 * q);
        };
        auto mfma_trip = [&](const bf16x8_t (&wv)[NSET][4], const bf16x8_t (&av)[4]) {
#pragma unroll
            for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q], wv[s_][q], acc[s_], 0, 0, 0);
        };
        int ks = ks0;
#pragma unroll
        for (int i = 0; i < NBUF; ++i)
            if (ks + KSTRIDE * i + 3 < ks1) load_trip(ks + KSTRIDE * i, wb[i], ab[i]);
        while (ks + 3 < ks1) {
#pragma unroll
            for (int i = 0; i < NBUF; ++i) {
                if (ks + 3 < ks1) {
                    mfma_trip(wb[i], ab[i]);
                    if (ks + KSTRIDE * NBUF + 3 < ks1) load_trip(ks + KSTRIDE * NBUF, wb[i], ab[i]);
                    ks += KSTRIDE;
                }
            }
        }
        if constexpr (!PACKED) {
            for (; ks < ks1; ++ks) {
                const int k0 = ks * 16 + 8 * h;
                const bf16x8_t av = a_frag(k0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, load_frag(wrow + k0), acc[0], 0, 0, 0);
                if constexpr (NSET == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, load_frag(wrow2 + k0), acc[1], 0, 0, 0);
            }
        }
        // ---- the eight waves' K slices: one LDS slot each, summed in wave order by the element's owner
#pragma unroll
        for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[s_][wave][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] = acc[s_][reg];
        __syncthreads();
        float v[NSET][2];
#pragma unroll
        for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = tid + j * FC_THREADS, row = e >> 5, col = e & 31;
                float x = 0.0f;
#pragma unroll
                for (int w = 0; w < FC_WAVES; ++w) x += red[s_][w][row][col];
                v[s_][j] = x;
            }
        bool finish = cb == 0 && ce == cpt;          // the whole K range of the tile was ours
        if (!finish) {
            const int slot = t == first_tile ? 0 : 1;
            // The partial tile leaves with device-coherent stores (written through this XCD's L2) and is read back with device-coherent
            // loads: no L2 write-back / invalidate of everything else (`__threadfence()` costs exactly that on a part with one L2 per XCD
            // -- measured: the kernel 10x slower).  Every thread waits for its own stores, the barrier collects the workgroup, then one
            // thread publishes the chunk count.
            float *wsp = a.ws + ((size_t)(wg * 2 + slot) * NSET) * 1024;
#pragma unroll
            for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                for (int j = 0; j < 2; ++j) __hip_atomic_store(wsp + s_ * 1024 + tid + j * FC_THREADS, v[s_][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // A workgroup-scope release fence emits no wait on gfx950 (and its barrier is a back-off barrier that inserts none either): the
            // explicit vmcnt(0) is what makes every thread's sc1 stores acknowledged by the L2 before the barrier lets thread 0 publish
            // the count -- the count and the partials live in different L2 channels and nothing else orders them.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const uint32_t mine = (uint32_t)(ce - cb);
                s_last = (__hip_atomic_fetch_add(&a.cnt[t], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine == (uint32_t)cpt) ? 1u : 0u;
            }
            __syncthreads();
            finish = s_last != 0u;
            if (finish) {          // every other segment of the tile is in the workspace: add them in K order
                const long long tb = (long long)t * cpt;
                const int g0 = (int)(((tb + 1) * G - 1) / total), g1 = (int)(((tb + cpt) * G - 1) / total);
#pragma unroll
                for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                    for (int j = 0; j < 2; ++j) v[s_][j] = 0.0f;
                for (int g = g0; g <= g1; ++g) {
                    const int gslot = (int)((total * g / G) / cpt) == t ? 0 : 1;
                    const float *src = a.ws + ((size_t)(g * 2 + gslot) * NSET) * 1024;
#pragma unroll
                    for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                        for (int j = 0; j < 2; ++j) v[s_][j] += __hip_atomic_load(src + s_ * 1024 + tid + j * FC_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == 0) __hip_atomic_store(&a.cnt[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // the next launch finds the counter at zero
            }
        }
        if constexpr (EPI == 3) {
            // the CFG combination pairs row i with row i + M / 2: through the (free) LDS tile
            if (finish) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = tid + j * FC_THREADS;
                    red[0][0][e >> 5][e & 31] = v[0][j];
                }
                __syncthreads();
                const int n_half = a.M / 2;
                for (int e = tid; e < n_half * 32; e += FC_THREADS) {
                    const int i = e >> 5, col = e & 31, n = t * 32 + col;
                    if (n < a.n_rows) {
                        const float b = a.bias ? bf16_bits_to_f32(a.bias[a.row_lo + n]) : 0.0f;
                        const float c = bf16_bits_to_f32(f32_to_bf16_rne(red[0][0][i][col] + b)), u = bf16_bits_to_f32(f32_to_bf16_rne(red[0][0][i + n_half][col] + b));
                        const float o = round_bf16(u + round_bf16(a.cfg * round_bf16(c - u)));
                        a.out[(size_t)i * a.out_stride + a.out_col0 + n] = (uint16_t)(__float_as_uint(o) >> 16);
                    }
                }
            }
        } else if (finish) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = tid + j * FC_THREADS, m = e >> 5, n = t * 32 + (e & 31);
                if (m < a.M && n < a.n_rows) {
                    float x = v[0][j];
                    if (a.bias) x += bf16_bits_to_f32(a.bias[a.row_lo + n]);
                    uint16_t o = f32_to_bf16_rne(x);
                    if constexpr (EPI == 1) o = f32_to_bf16_rne(bf16_bits_to_f32(a.aux[(size_t)m * a.aux_stride + n]) + bf16_bits_to_f32(o));
                    if constexpr (EPI == 2) {
                        float u = v[NSET - 1][j];
                        if (a.bias) u += bf16_bits_to_f32(a.bias[a.row_lo + a.pair_rows + n]);
                        const float gb = bf16_bits_to_f32(o), ub = bf16_bits_to_f32(f32_to_bf16_rne(u));
                        const float sg = bf16_bits_to_f32(f32_to_bf16_rne(gb / (1.0f + expf(-gb))));
                        o = f32_to_bf16_rne(sg * ub);
                    }
                    a.out[(size_t)m * a.out_stride + a.out_col0 + n] = o;
                }
            }
        }
        __syncthreads();          // red / s_last are reused by the next segment
        c0 = (long long)t * cpt + ce;
    }
}

// The same products for ANY number of rows (the drafter's prompt prefill: hundreds to a few thousand rows) on the PACKED weights the decoder layer
// keeps for the stream-K kernels: workgroup (tile, row block) contracts MT x 32 rows with the tile's 32 columns (EPI 2: and the paired 32), the
// eight waves take the tile's 64-element K bricks round-robin (a wave's load = 1 KB contiguous), the next brick's weights are requested before the
// current brick's MFMAs, the K slices meet in LDS in wave order (deterministic f32 sums).  A weight tile is streamed once per row block: from HBM
// for the first, from the Infinity Cache for the others (the largest matrix, 180 MB, fits it).  Epilogues as everywhere in this file.
template <int MT, int EPI>
__global__ __launch_bounds__(FC_THREADS) void linear_rows_packed_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Wp,
                                                                        const uint16_t *__restrict__ bias, int M, int K, int n_rows,
                                                                        uint16_t *__restrict__ out, int out_stride, const uint16_t *__restrict__ aux,
                                                                        int aux_stride, int pair_rows) {
    constexpr int NSET = EPI == 2 ? 2 : 1;
    __shared__ float tile[NSET][MT][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.x, n0 = t * 32, m0 = blockIdx.y * (MT * 32);
    for (int i = tid; i < NSET * MT * 32 * 33; i += FC_THREADS) (&tile[0][0][0][0])[i] = 0.0f;
    const uint16_t *arow[MT];
    bool live[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 32 + r;
        live[mt] = row < M;
        arow[mt] = A + (size_t)(live[mt] ? row : 0) * K + 32 * h;
    }
    f32x16_t acc[NSET][MT];
#pragma unroll
    for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s_][mt][i] = 0.0f;
    const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    const int nb = K / 64;
    const uint16_t *tbase = Wp + (size_t)t * nb * NSET * 2048 + lane * 8;
    bf16x8_t wb[2][NSET][4];
    auto load_w = [&](int kb, bf16x8_t (&wv)[NSET][4]) {
        const uint16_t *brick = tbase + (size_t)kb * NSET * 2048;
#pragma unroll
        for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[s_][q] = load_frag(brick + s_ * 2048 + q * 512);
    };
    auto step = [&](int kb, const bf16x8_t (&wv)[NSET][4]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8_t aw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) aw[q] = live[mt] ? load_frag(arow[mt] + kb * 64 + 8 * q) : zero;
#pragma unroll
            for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s_][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q], wv[s_][q], acc[s_][mt], 0, 0, 0);
        }
    };
    int kb = wave;
    if (kb < nb) load_w(kb, wb[0]);
    while (kb < nb) {
        if (kb + FC_WAVES < nb) load_w(kb + FC_WAVES, wb[1]);
        step(kb, wb[0]);
        kb += FC_WAVES;
        if (kb >= nb) break;
        if (kb + FC_WAVES < nb) load_w(kb + FC_WAVES, wb[0]);
        step(kb, wb[1]);
        kb += FC_WAVES;
    }
    for (int w = 0; w < FC_WAVES; ++w) {      // combine the K slices in wave order (deterministic f32 sum)
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int s_ = 0; s_ < NSET; ++s_)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) tile[s_][mt][(reg & 3) + 8 * (reg >> 2) + 4 * h][r] += acc[s_][mt][reg];
        }
    }
    __syncthreads();
    for (int i = tid; i < MT * 32 * 32; i += FC_THREADS) {
        const int mt = i / 1024, row = (i / 32) % 32, col = i % 32;
        const int m = m0 + mt * 32 + row, n = n0 + col;
        if (m < M && n < n_rows) {
            float v = tile[0][mt][row][col];
            if (bias) v += bf16_bits_to_f32(bias[n]);
            uint16_t o = f32_to_bf16_rne(v);
            if constexpr (EPI == 1) o = f32_to_bf16_rne(bf16_bits_to_f32(aux[(size_t)m * aux_stride + n]) + bf16_bits_to_f32(o));
            if constexpr (EPI == 2) {
                float u = tile[NSET - 1][mt][row][col];
                if (bias) u += bf16_bits_to_f32(bias[pair_rows + n]);
                const float gb = bf16_bits_to_f32(o), ub = bf16_bits_to_f32(f32_to_bf16_rne(u));
                const float sg = bf16_bits_to_f32(f32_to_bf16_rne(gb / (1.0f + expf(-gb))));
                o = f32_to_bf16_rne(sg * ub);
            }
            out[(size_t)m * out_stride + n] = o;
        }
    }
}

// The same products as a tiled MFMA GEMM for MANY rows (prompt prefills: hundreds to thousands): workgroup = 128 rows x 128 output columns
// (gate / up pair: 128 x 64 of each), four waves of 64 x 64 each (2 x 2 tiles of v_mfma_f32_32x32x16_bf16; the pair: 2 row tiles x {gate, up} of
// one 32-column tile, so silu(gate) * up is element-wise in registers).  Per 64-element K block the A tile (128 x 64 bf16) goes global ->
// registers -> LDS (rows padded to 144 bytes: the 16 lanes of a ds_read_b128 phase cover all 64 banks once), double-buffered, the next block's
// global loads in flight under this block's MFMAs; the weights come straight from the packed bricks (already in B-fragment order: 16 bytes
// per lane and MFMA step, no LDS), next block's prefetched into a second register set.  The row-blocked kernel above (kept for 33 - 128 rows) re-reads A from L2 for
// every 32-column tile (15 GB at 1200 rows of the 7B layer); here A crosses L2 -> LDS once per 128 columns.
constexpr int TG_BM = 128, TG_THREADS = 256, TG_LDA = 72;          // LDS row stride in bf16 elements (64 + 8 pad)

template <int EPI>
__global__ __launch_bounds__(TG_THREADS, 2) void linear_rows_tiled_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ Wp,
                                                                       const uint16_t *__restrict__ bias, int M, int K, int n_rows,
                                                                       uint16_t *__restrict__ out, int out_stride, const uint16_t *__restrict__ aux,
                                                                       int aux_stride, int pair_rows) {
    constexpr int NSET = EPI == 2 ? 2 : 1;
    __shared__ alignas(16) uint16_t sA[2][TG_BM * TG_LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;          // the wave's 64-row half and 64-column (pair: 32-column) half
    const int m0 = blockIdx.x * TG_BM;
    const int nb = K / 64;
    // the two 32-column weight tiles (EPI 2: the gate and the up set of ONE tile) this wave contracts with
    const int n_tiles = (n_rows + 31) / 32;
    int tile[2];
    size_t boff[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        int t = EPI == 2 ? (int)blockIdx.y * 2 + wn : (int)blockIdx.y * 4 + wn * 2 + s_;
        tile[s_] = t;
        if (t >= n_tiles) t = n_tiles - 1;          // (a surplus tile contracts the last one again and stores nothing)
        boff[s_] = ((size_t)t * nb * NSET + (EPI == 2 ? s_ : 0)) * 2048 + lane * 8;
    }
    // A: thread -> four 16-byte chunks of the 128 x 64 block (chunk c of row i / 8 ... coalesced 128-byte rows)
    const uint16_t *ag[4];
    bool alive[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = tid + i * TG_THREADS, row = ch >> 3, c = ch & 7;
        alive[i] = m0 + row < M;
        ag[i] = A + (size_t)(alive[i] ? m0 + row : 0) * K + c * 8;
        lds_off[i] = row * TG_LDA + c * 8;
    }
    f32x16_t acc[2][2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s_][mt][i] = 0.0f;
    uint4 areg[4];
    bf16x8_t w0[2][4], w1[2][4];          // (two named sets, not an indexed array: a run-time index would put them in scratch)
    auto load_a = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) areg[i] = alive[i] ? *reinterpret_cast<const uint4 *>(ag[i] + (size_t)kb * 64) : make_uint4(0, 0, 0, 0);
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(&sA[buf][lds_off[i]]) = areg[i];
    };
    auto load_w = [&](int kb, bf16x8_t (&w)[2][4]) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[s_][q] = load_frag(Wp + boff[s_] + (size_t)kb * NSET * 2048 + q * 512);
    };
    auto compute = [&](int buf, const bf16x8_t (&w)[2][4]) {
        // the four accumulators take turns (an MFMA that adds to the accumulator of the one before it waits out its whole latency)
        bf16x8_t af[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const uint16_t *arow = &sA[buf][(wm * 64 + mt * 32 + r) * TG_LDA + 32 * h];
#pragma unroll
            for (int q = 0; q < 4; ++q) af[mt][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(arow + 8 * q));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) acc[s_][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][q], w[s_][q], acc[s_][mt], 0, 0, 0);
    };
    load_a(0);
    load_w(0, w0);
    store_a(0);
    __syncthreads();
    for (int kb = 0; kb < nb; kb += 2) {
        if (kb + 1 < nb) {
            load_a(kb + 1);
            load_w(kb + 1, w1);
        }
        compute(0, w0);
        if (kb + 1 < nb) store_a(1);          // (the other buffer: its readers finished before the previous barrier)
        __syncthreads();
        if (kb + 1 >= nb) break;
        if (kb + 2 < nb) {
            load_a(kb + 2);
            load_w(kb + 2, w0);
        }
        compute(1, w1);
        if (kb + 2 < nb) store_a(0);
        __syncthreads();
    }
    // epilogue from the accumulators: element (row = (reg & 3) + 8 (reg >> 2) + 4 h, column r) of each 32 x 32 tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = m0 + wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (m >= M) continue;
            if constexpr (EPI == 2) {
                const int n = tile[0] * 32 + r;
                if (tile[0] < n_tiles && n < n_rows) {
                    float g = acc[0][mt][reg], u = acc[1][mt][reg];
                    if (bias) { g += bf16_bits_to_f32(bias[n]); u += bf16_bits_to_f32(bias[pair_rows + n]); }
                    const float gb = bf16_bits_to_f32(f32_to_bf16_rne(g)), ub = bf16_bits_to_f32(f32_to_bf16_rne(u));
                    const float sg = bf16_bits_to_f32(f32_to_bf16_rne(gb / (1.0f + expf(-gb))));
                    out[(size_t)m * out_stride + n] = f32_to_bf16_rne(sg * ub);
                }
            } else {
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const int n = tile[s_] * 32 + r;
                    if (tile[s_] < n_tiles && n < n_rows) {
                        float v = acc[s_][mt][reg];
                        if (bias) v += bf16_bits_to_f32(bias[n]);
                        uint16_t o = f32_to_bf16_rne(v);
                        if constexpr (EPI == 1) o = f32_to_bf16_rne(bf16_bits_to_f32(aux[(size_t)m * aux_stride + n]) + bf16_bits_to_f32(o));
                        out[(size_t)m * out_stride + n] = o;
                    }
                }
            }
        }
}

extern "C" int lantern_linear_rows_packed(const void *A, const void *W_packed, const void *bias, int M, int K, int n_rows, void *out, int out_stride,
                                          int epilogue, const void *aux, int aux_stride, int pair_rows, void *stream) {
    LANTERN_CHECK_ARG(A && W_packed && out, "linear_rows_packed: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && K > 0 && K % 64 == 0 && n_rows >= 0 && out_stride >= n_rows, "linear_rows_packed: K=%d must be a multiple of 64, out rows hold n_rows", K);
    LANTERN_CHECK_ARG(epilogue == 0 || epilogue == LANTERN_EPI_RESIDUAL || epilogue == LANTERN_EPI_SILU_MUL, "linear_rows_packed: epilogue %d", epilogue);
    if (epilogue == LANTERN_EPI_RESIDUAL) LANTERN_CHECK_ARG(aux && aux_stride >= n_rows, "linear_rows_packed: the residual [M, aux_stride >= n_rows] is missing");
    if (epilogue == LANTERN_EPI_SILU_MUL) LANTERN_CHECK_ARG(pair_rows > 0, "linear_rows_packed: pair_rows = rows of the gate half of the packed gate / up pair");
    if (M == 0 || n_rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t *a = (const uint16_t *)A, *w = (const uint16_t *)W_packed, *bi = (const uint16_t *)bias, *ax = (const uint16_t *)aux;
    uint16_t *o = (uint16_t *)out;
    const int tiles = (n_rows + 31) / 32;
    LANTERN_CHECK_ARG((M + 63) / 64 <= 65535, "linear_rows_packed: M=%d rows exceed the launch grid", M);
    const int tiled_from = tuning(TUNE_GEMM_TILED_FROM);
    if (M >= tiled_from) {          // many rows: the LDS-tiled form (128 x 128 per workgroup; row blocks fastest so that a weight tile's readers are neighbours)
        const dim3 grid((M + TG_BM - 1) / TG_BM, epilogue == LANTERN_EPI_SILU_MUL ? (tiles + 1) / 2 : (tiles + 3) / 4);
        if (epilogue == LANTERN_EPI_SILU_MUL) LANTERN_LAUNCH((linear_rows_tiled_kernel<2>), grid, dim3(TG_THREADS), 0, st, a, w, bi, M, K, n_rows, o, out_stride, ax, aux_stride, pair_rows);
        else if (epilogue == LANTERN_EPI_RESIDUAL) LANTERN_LAUNCH((linear_rows_tiled_kernel<1>), grid, dim3(TG_THREADS), 0, st, a, w, bi, M, K, n_rows, o, out_stride, ax, aux_stride, 0);
        else LANTERN_LAUNCH((linear_rows_tiled_kernel<0>), grid, dim3(TG_THREADS), 0, st, a, w, bi, M, K, n_rows, o, out_stride, ax, aux_stride, 0);
        LANTERN_CHECK_LAUNCH("linear_rows_packed");
        return LANTERN_OK;
    }
#define LRP_LAUNCH(MT_, E_)                                                                                                                         \
    LANTERN_LAUNCH((linear_rows_packed_kernel<MT_, E_>), dim3(tiles, (M + MT_ * 32 - 1) / (MT_ * 32)), dim3(FC_THREADS), 0, st, a, w, bi, M, K, n_rows, o, \
                   out_stride, ax, aux_stride, epilogue == LANTERN_EPI_SILU_MUL ? pair_rows : 0)
    if (epilogue == LANTERN_EPI_SILU_MUL) {
        if (M <= 32) LRP_LAUNCH(1, 2);
        else LRP_LAUNCH(2, 2);
    } else if (epilogue == LANTERN_EPI_RESIDUAL) {
        if (M <= 32) LRP_LAUNCH(1, 1);
        else if (M <= 64) LRP_LAUNCH(2, 1);
        else LRP_LAUNCH(4, 1);
    } else {
        if (M <= 32) LRP_LAUNCH(1, 0);
        else if (M <= 64) LRP_LAUNCH(2, 0);
        else LRP_LAUNCH(4, 0);
    }
#undef LRP_LAUNCH
    LANTERN_CHECK_LAUNCH("linear_rows_packed");
    return LANTERN_OK;
}

// workgroups of a launch: one per CU of the current device (the register buffers of the trips in flight leave room for one 512-thread
// workgroup per CU).  lantern_tuning_set("sk_groups", g) overrides it (measurement runs).
static int sk_groups(int) {
    const int env_g = tuning(TUNE_SK_GROUPS);
    if (env_g > 0) return env_g;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}


constexpr size_t SK_PARTIAL_BYTES = (size_t)1024 * 2 * 2 * 1024 * sizeof(float);      // <= 1024 workgroups x 2 segments x 2 tile sets x 32 x 32 f32

extern "C" size_t lantern_linear_rows_streamk_workspace(int n_rows) {
    if (n_rows <= 0) return 0;
    return SK_PARTIAL_BYTES + (size_t)((n_rows + 31) / 32) * sizeof(uint32_t) + 256;      // + the tile counters
}

// [N, K] row-major -> bricks: out[((tile * K/64 + kb) * sets + set) * 2048 + q * 512 + lane * 8 + e] = W[set * pair_rows + tile * 32 + (lane & 31)][kb * 64 + 32 (lane >> 5) + 8 q + e]
// (rows beyond n_rows: zeros)
__global__ void pack_linear_weight_kernel(const uint16_t *__restrict__ W, int n_rows, int K, int sets, int pair_rows, uint16_t *__restrict__ out, long long n_frag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte fragment per thread
    if (i >= n_frag) return;
    const int lane = (int)(i & 63), q = (int)((i >> 6) & 3);
    long long b = i >> 8;
    const int set = (int)(b % sets);
    b /= sets;
    const int kb = (int)(b % (K / 64)), tile = (int)(b / (K / 64));
    const int row = tile * 32 + (lane & 31);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < n_rows) v = *reinterpret_cast<const uint4 *>(W + (size_t)(set * pair_rows + row) * K + kb * 64 + 32 * (lane >> 5) + 8 * q);
    reinterpret_cast<uint4 *>(out)[i] = v;
}

extern "C" size_t lantern_pack_linear_weight_bytes(int n_rows, int K, int pair_rows) {
    if (n_rows <= 0 || K <= 0 || K % 64) return 0;
    return (size_t)((n_rows + 31) / 32) * 32 * (size_t)K * 2 * (pair_rows > 0 ? 2 : 1);
}

extern "C" int lantern_pack_linear_weight(const void *W, int n_rows, int K, int pair_rows, void *out, void *stream) {
    LANTERN_CHECK_ARG(W && out && n_rows > 0 && K > 0 && K % 64 == 0 && pair_rows >= 0, "pack_linear_weight: K=%d must be a multiple of 64", K);
    const int sets = pair_rows > 0 ? 2 : 1;
    const long long n_frag = (long long)((n_rows + 31) / 32) * (K / 64) * sets * 256;
    hipLaunchKernelGGL(pack_linear_weight_kernel, dim3((unsigned)((n_frag + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)W, n_rows, K, sets,
                       pair_rows, (uint16_t *)out, n_frag);
    LANTERN_CHECK_LAUNCH("pack_linear_weight");
    return LANTERN_OK;
}

// shared launch: shares of the (tile, K chunk) space over G workgroups, fixed workspace layout (partials, then counters)
static int sk_run(SkArgs a, int epilogue, bool packed, bool gather, void *workspace, size_t workspace_bytes, hipStream_t st, const char *who) {
    LANTERN_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= lantern_linear_rows_streamk_workspace(a.n_rows),
                      "%s: workspace of lantern_linear_rows_streamk_workspace(n_rows) bytes, 16-byte aligned, zero-filled once", who);
    if (packed) LANTERN_CHECK_ARG(a.K % 64 == 0, "%s: a packed weight needs K %% 64 == 0 (K=%d)", who, a.K);
    a.n_tiles = (a.n_rows + 31) / 32;
    a.cpt = (a.K + SK_CHUNK - 1) / SK_CHUNK;
    const long long total = (long long)a.n_tiles * a.cpt;
    int G = sk_groups(epilogue);
    if (G > 1024) G = 1024;
    if ((long long)G > total) G = (int)total;
    // Small matrices (LlamaGen-size drafters: 3 - 20 MB per product) are bound by the launch's fixed costs, not by the stream: a range that ends inside a tile
    // costs its workgroup a partial-tile round trip and the tile's finisher another, so they get WHOLE tiles per workgroup (the largest divisor of the tile
    // count that fits the grid: no partial tiles at all) even where that leaves CUs idle (LlamaGen EAGLE-2 cycle 848 - 863 -> 822 - 831 us; 50 MB: slower).  lantern_tuning_set("sk_whole_mb", ..): the
    // size limit in MB (default 40: the 7B drafter's o_proj, 34 MB in 128 tiles, is still better off on 128 workgroups without partial tiles -- Lumina static
    // cycle 884 - 889 -> 865 - 868 us; 70, which takes in the 67 MB input stage: 878; 0 = never).
    const int whole_mb = tuning(TUNE_SK_WHOLE_MB);
    if (whole_mb > 0 && (long long)a.n_tiles * 32 * a.K * 2 * (epilogue == LANTERN_EPI_SILU_MUL ? 2 : 1) <= (long long)whole_mb * 1000000) {
        int best = 1;
        for (int g = 1; g <= G && g <= a.n_tiles; ++g)
            if (a.n_tiles % g == 0) best = g;
        // (a tile count without a large divisor -- prime, or just above the grid: 257, 2 x 131 -- would collapse to one or two workgroups for the
        // whole product: whole tiles only when they keep at least half of the grid busy, else the stream-K split stays -- ADVICE round 5)
        if (2 * best >= G || 2 * best >= a.n_tiles) G = best;
    }
    a.G = G;
    // fixed layout whatever the shape (launches of different shapes share one workspace): the partial tiles first, the tile counters behind them
    a.ws = (float *)workspace;
    a.cnt = (uint32_t *)((char *)workspace + SK_PARTIAL_BYTES);
    // Weight matrices of 80 MB and more are read with the non-temporal hint (`global_load_dwordx4 ... nt`, the WNT instances): a stream that size is gone
    // from the 256 MB Infinity Cache before the next drafting pass comes back to it anyway, and read that way it neither displaces what the smaller
    // products (input stage, o_proj, head window: 167 MB at 7B size) and the K / V rows leave there nor pays for allocating its lines.  Measured on the 7B
    // drafter (tools/draft_bench.py lumina_static): threshold 200 MB (none) 1001 us per cycle, 120 (gate / up) 981, 95 (+ q/k/v) 931, 80 (+ down) 887-900,
    // 30 (all but o_proj) 901, 0 (all) 912.  lantern_tuning_set("sk_nt_min_mb", ..): negative = never.
    const int nt_min_mb = tuning(TUNE_SK_NT_MIN_MB);
    a.w_stream = nt_min_mb >= 0 && (long long)a.n_tiles * 32 * a.K * 2 * (epilogue == LANTERN_EPI_SILU_MUL ? 2 : 1) >= (long long)nt_min_mb * 1000000 ? 1 : 0;
    // two trips in flight per wave: three and four measured the same (24.4 - 24.6 / 21.5 - 21.9 us for the 100 / 90 MB matrices) -- the kernel is
    // at the read bandwidth the part delivers (4.1 - 4.6 TB/s; torch's read-only reductions reach 3.8 - 4.0, its copy 5.2 read + write)
#define SK_LAUNCH(E_, GA_)                                                                                                   \
    do {                                                                                                                  \
        if (packed && a.w_stream) LANTERN_LAUNCH((linear_rows_streamk_kernel<E_, true, 2, GA_, true>), dim3(G), dim3(FC_THREADS), 0, st, a);   \
        else if (packed) LANTERN_LAUNCH((linear_rows_streamk_kernel<E_, true, 2, GA_>), dim3(G), dim3(FC_THREADS), 0, st, a);   \
        else LANTERN_LAUNCH((linear_rows_streamk_kernel<E_, false, 2, GA_>), dim3(G), dim3(FC_THREADS), 0, st, a);         \
    } while (0)
    if (gather) SK_LAUNCH(0, true);
    else if (epilogue == LANTERN_EPI_SILU_MUL) SK_LAUNCH(2, false);
    else if (epilogue == 3) SK_LAUNCH(3, false);
    else if (epilogue == 0) SK_LAUNCH(0, false);
    else SK_LAUNCH(1, false);
#undef SK_LAUNCH
    LANTERN_CHECK_LAUNCH(who);
    return LANTERN_OK;
}

extern "C" int lantern_linear_rows_streamk(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                                           int out_stride, int out_col0, int epilogue, const void *aux, int aux_stride, int pair_rows,
                                           int packed, void *workspace, size_t workspace_bytes, void *stream) {
    LANTERN_CHECK_ARG(A && W && out && workspace, "linear_rows_streamk: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 32, "linear_rows_streamk: M=%d must be <= 32 rows (the drafter's decode shape)", M);
    LANTERN_CHECK_ARG(K > 0 && K % 16 == 0 && row_lo >= 0 && n_rows >= 0 && out_col0 >= 0 && out_stride >= out_col0 + n_rows,
                      "linear_rows_streamk: K=%d must be a multiple of 16, the output row must hold [col0, col0 + n_rows)", K);
    LANTERN_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "linear_rows_streamk: epilogue %d", epilogue);
    if (epilogue == LANTERN_EPI_RESIDUAL) LANTERN_CHECK_ARG(aux && aux_stride >= n_rows, "linear_rows_streamk: the residual [M, aux_stride >= n_rows] is missing");
    if (epilogue == LANTERN_EPI_SILU_MUL) LANTERN_CHECK_ARG(pair_rows > 0, "linear_rows_streamk: pair_rows = distance (in weight rows) from a gate row to its up row");
    if (M == 0 || n_rows == 0) return LANTERN_OK;
    SkArgs a{};
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = (const uint16_t *)bias; a.aux = (const uint16_t *)aux; a.out = (uint16_t *)out;
    a.M = M; a.K = K; a.row_lo = row_lo; a.n_rows = n_rows; a.out_stride = out_stride; a.out_col0 = out_col0; a.aux_stride = aux_stride;
    a.pair_rows = epilogue == LANTERN_EPI_SILU_MUL ? pair_rows : 0;
    return sk_run(a, epilogue, packed != 0, false, workspace, workspace_bytes, (hipStream_t)stream, "linear_rows_streamk");
}

// O11 in stream-K form (M <= 32 rows: the drafting shape): the embedding gather, its scaling and the concat stay folded into the A-fragment
// address; W [H, 2H] row-major or packed (lantern_pack_linear_weight(W, H, 2H, 0)).
extern "C" int lantern_drafter_fc_streamk(const int64_t *ids, const void *hidden, const void *embed, const void *W, const void *bias, int M, int H,
                                          int vocab, float embed_scale, void *out, int packed, void *workspace, size_t workspace_bytes, void *stream) {
    LANTERN_CHECK_ARG(ids && hidden && embed && W && out && workspace, "drafter_fc_streamk: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 32 && H > 0 && H % 64 == 0 && vocab > 0, "drafter_fc_streamk: M=%d <= 32 rows, H=%d a multiple of 64", M, H);
    if (M == 0) return LANTERN_OK;
    SkArgs a{};
    a.A = (const uint16_t *)hidden; a.W = (const uint16_t *)W; a.bias = (const uint16_t *)bias; a.out = (uint16_t *)out;
    a.M = M; a.K = 2 * H; a.n_rows = H; a.out_stride = H;
    a.ids = ids; a.embed = (const uint16_t *)embed; a.vocab = vocab; a.hsplit = H; a.embed_scale = embed_scale;
    return sk_run(a, 0, packed != 0, true, workspace, workspace_bytes, (hipStream_t)stream, "drafter_fc_streamk");
}

namespace lantern {
// the same with the rows read through a static tree's tables (lantern_draft_depth, in_rep): M = B * rows_per_b rows; ids [n_flat]; hidden [B, src_T, H]
int launch_drafter_fc_streamk_tables(const int64_t *ids, int n_flat, const int32_t *in_gather, const int32_t *in_rep, const void *hidden, int src_T, int rows_per_b,
                                     const void *embed, const void *W, const void *bias, int M, int H, int vocab, float embed_scale, void *out, int packed,
                                     void *workspace, size_t workspace_bytes, hipStream_t st) {
    LANTERN_CHECK_ARG(ids && in_gather && in_rep && hidden && embed && W && out && workspace && n_flat > 0 && src_T > 0 && rows_per_b > 0 && M > 0 && M <= 32 &&
                          M % rows_per_b == 0 && H > 0 && H % 64 == 0 && vocab > 0,
                      "drafter_fc_streamk (tables): bad arguments");
    SkArgs a{};
    a.A = (const uint16_t *)hidden; a.W = (const uint16_t *)W; a.bias = (const uint16_t *)bias; a.out = (uint16_t *)out;
    a.M = M; a.K = 2 * H; a.n_rows = H; a.out_stride = H;
    a.ids = ids; a.embed = (const uint16_t *)embed; a.vocab = vocab; a.hsplit = H; a.embed_scale = embed_scale;
    a.in_gather = in_gather; a.in_rep = in_rep; a.rows_per_b = rows_per_b; a.src_T = src_T; a.n_flat = n_flat;
    return sk_run(a, 0, packed != 0, true, workspace, workspace_bytes, st, "drafter_fc_streamk");
}
}  // namespace lantern

namespace lantern {
// the drafter head's window GEMM with the CFG epilogue (lantern_head_expand), stream-K form: W row-major [V, K] (rows row_lo.. used) or the
// packed rows [row_lo, row_lo + n_cols)
int launch_linear_rows_cfg_streamk(const void *A, const void *W, const void *bias, int n, int K, int row_lo, int n_cols, float cfg, void *win, int packed,
                                   void *workspace, size_t workspace_bytes, hipStream_t st) {
    SkArgs a{};
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = (const uint16_t *)bias; a.out = (uint16_t *)win;
    a.M = 2 * n; a.K = K; a.row_lo = row_lo; a.n_rows = n_cols; a.out_stride = n_cols; a.cfg = cfg;
    return sk_run(a, 3, packed != 0, false, workspace, workspace_bytes, st, "head_expand");
}
}  // namespace lantern

extern "C" int lantern_linear_rows_epilogue(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                                            int out_stride, int out_col0, int epilogue, const void *aux, int aux_stride, int pair_rows, void *stream) {
    LANTERN_CHECK_ARG(A && W && out, "linear_rows_epilogue: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 32, "linear_rows_epilogue: M=%d must be <= 32 rows (the drafter's decode shape)", M);
    LANTERN_CHECK_ARG(K > 0 && K % 16 == 0 && row_lo >= 0 && n_rows >= 0 && out_col0 >= 0 && out_stride >= out_col0 + n_rows,
                      "linear_rows_epilogue: K=%d must be a multiple of 16, the output row must hold [col0, col0 + n_rows)", K);
    LANTERN_CHECK_ARG(epilogue == LANTERN_EPI_RESIDUAL || epilogue == LANTERN_EPI_SILU_MUL, "linear_rows_epilogue: epilogue %d", epilogue);
    if (epilogue == LANTERN_EPI_RESIDUAL) LANTERN_CHECK_ARG(aux && aux_stride >= n_rows, "linear_rows_epilogue: the residual [M, aux_stride >= n_rows] is missing");
    if (epilogue == LANTERN_EPI_SILU_MUL) LANTERN_CHECK_ARG(pair_rows > 0, "linear_rows_epilogue: pair_rows = distance (in weight rows) from a gate row to its up row");
    if (M == 0 || n_rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n_rows + 31) / 32), block(FC_THREADS);
    const uint16_t *a = (const uint16_t *)A, *w = (const uint16_t *)W, *bi = (const uint16_t *)bias, *ax = (const uint16_t *)aux;
    uint16_t *o = (uint16_t *)out;
    if (epilogue == LANTERN_EPI_RESIDUAL) LANTERN_LAUNCH((linear_rows_kernel<1, 1>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, ax, aux_stride, 0);
    else LANTERN_LAUNCH((linear_rows_kernel<1, 2>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, ax, 0, pair_rows);
    LANTERN_CHECK_LAUNCH("linear_rows_epilogue");
    return LANTERN_OK;
}

extern "C" int lantern_rmsnorm_rows(const void *x, const void *weight, int M, int H, float eps, void *out, void *stream) {
    LANTERN_CHECK_ARG(x && weight && out && M >= 0 && H > 0, "rmsnorm_rows: bad arguments");
    if (M == 0) return LANTERN_OK;
    LANTERN_LAUNCH(rmsnorm_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, (const uint16_t *)weight, H, eps, (uint16_t *)out);
    LANTERN_CHECK_LAUNCH("rmsnorm_rows");
    return LANTERN_OK;
}

namespace lantern {
int launch_qk_norm_rope(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const void *q_weight, const void *q_bias,
                        const void *k_weight, const void *k_bias, int model_parallel, const void *cos_table, const void *sin_table,
                        int table_rows, const int64_t *position_ids, void *q_out, int q_rows, int q_row0, void *k_out, void *v_out, int kv_rows, int kv_row0,
                        void *stream) {
    LANTERN_CHECK_ARG(qkv && q_weight && q_bias && k_weight && k_bias && cos_table && sin_table && position_ids && q_out && k_out && v_out,
                      "qk_norm_rope: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && T >= 0 && n_q_heads > 0 && n_kv_heads > 0 && model_parallel > 0 && n_q_heads % model_parallel == 0 &&
                          n_kv_heads % model_parallel == 0 && table_rows > 0 && kv_row0 >= 0 && kv_rows >= kv_row0 + T && q_row0 >= 0 && q_rows >= q_row0 + T,
                      "qk_norm_rope: bad sizes (k / v are [B, nk, kv_rows, d] slabs written at rows kv_row0 .. kv_row0 + T)");
    LANTERN_CHECK_ARG(head_dim == 128 || head_dim == 64, "qk_norm_rope: head_dim %d (64 or 128)", head_dim);
    if (B * T == 0) return LANTERN_OK;
    dim3 grid(B * T, n_q_heads + 2 * n_kv_heads);
#define QKNR(D_) LANTERN_LAUNCH((qk_norm_rope_kernel<D_>), grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t *)qkv, T, n_q_heads, n_kv_heads,     \
                                (const uint16_t *)q_weight, (const uint16_t *)q_bias, (const uint16_t *)k_weight, (const uint16_t *)k_bias,                 \
                                n_q_heads / model_parallel, n_kv_heads / model_parallel, (const uint16_t *)cos_table, (const uint16_t *)sin_table,           \
                                position_ids, (uint16_t *)q_out, (uint16_t *)k_out, (uint16_t *)v_out, kv_rows, kv_row0, table_rows, q_rows, q_row0)
    if (head_dim == 128) QKNR(128);
    else QKNR(64);
#undef QKNR
    LANTERN_CHECK_LAUNCH("qk_norm_rope");
    return LANTERN_OK;
}

int launch_qk_rope_pairs(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const float *freqs, int table_rows,
                         const int64_t *position_ids, int positions_per_batch_row, void *q_out, int q_rows, int q_row0, void *k_out, void *v_out,
                         int kv_rows, int kv_row0, void *stream) {
    LANTERN_CHECK_ARG(qkv && freqs && position_ids && q_out && k_out && v_out, "qk_rope_pairs: null buffer");
    LANTERN_CHECK_ARG(B >= 0 && T >= 0 && n_q_heads > 0 && n_kv_heads > 0 && table_rows > 0 && kv_row0 >= 0 && kv_rows >= kv_row0 + T && q_row0 >= 0 &&
                          q_rows >= q_row0 + T,
                      "qk_rope_pairs: bad sizes (k / v are [B, nk, kv_rows, d] slabs written at rows kv_row0 .. kv_row0 + T)");
    LANTERN_CHECK_ARG(head_dim == 128 || head_dim == 64, "qk_rope_pairs: head_dim %d (64 or 128)", head_dim);
    if (B * T == 0) return LANTERN_OK;
    dim3 grid(B * T, n_q_heads + 2 * n_kv_heads);
#define QKRP(D_) LANTERN_LAUNCH((qk_rope_pairs_kernel<D_>), grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t *)qkv, T, n_q_heads, n_kv_heads, freqs,       \
                                position_ids, positions_per_batch_row, (uint16_t *)q_out, (uint16_t *)k_out, (uint16_t *)v_out, kv_rows, kv_row0, table_rows,    \
                                q_rows, q_row0)
    if (head_dim == 128) QKRP(128);
    else QKRP(64);
#undef QKRP
    LANTERN_CHECK_LAUNCH("qk_rope_pairs");
    return LANTERN_OK;
}
}  // namespace lantern

extern "C" int lantern_qk_norm_rope(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const void *q_weight, const void *q_bias,
                                    const void *k_weight, const void *k_bias, int model_parallel, const void *cos_table, const void *sin_table,
                                    int table_rows, const int64_t *position_ids, void *q_out, void *k_out, void *v_out, int kv_rows, int kv_row0,
                                    void *stream) {
    return lantern::launch_qk_norm_rope(qkv, B, T, n_q_heads, n_kv_heads, head_dim, q_weight, q_bias, k_weight, k_bias, model_parallel, cos_table, sin_table,
                                        table_rows, position_ids, q_out, T, 0, k_out, v_out, kv_rows, kv_row0, stream);
}

extern "C" int lantern_qk_rope_pairs(const void *qkv, int B, int T, int n_q_heads, int n_kv_heads, int head_dim, const float *freqs, int table_rows,
                                     const int64_t *position_ids, int positions_per_batch_row, void *q_out, void *k_out, void *v_out, int kv_rows,
                                     int kv_row0, void *stream) {
    return lantern::launch_qk_rope_pairs(qkv, B, T, n_q_heads, n_kv_heads, head_dim, freqs, table_rows, position_ids, positions_per_batch_row, q_out, T, 0,
                                         k_out, v_out, kv_rows, kv_row0, stream);
}

namespace lantern {
// lantern_linear_rows_streamk with the activation rows in segments (lantern_draft_depth): row m at A + (m / seg_rows) * seg_stride + (m % seg_rows) * K
int launch_linear_rows_streamk_seg(const void *A, int a_seg_rows, long long a_seg_stride, const void *W, const void *bias, int M, int K, int n_rows, void *out,
                                   int epilogue, const void *aux, int aux_stride, int pair_rows, int packed, void *workspace, size_t workspace_bytes,
                                   hipStream_t st) {
    LANTERN_CHECK_ARG(A && W && out && workspace && M > 0 && M <= 32 && K > 0 && K % 16 == 0 && n_rows > 0 && epilogue >= 0 && epilogue <= 2,
                      "linear_rows_streamk: bad arguments (M <= 32 rows, K a multiple of 16)");
    SkArgs a{};
    a.A = (const uint16_t *)A; a.W = (const uint16_t *)W; a.bias = (const uint16_t *)bias; a.aux = (const uint16_t *)aux; a.out = (uint16_t *)out;
    a.M = M; a.K = K; a.n_rows = n_rows; a.out_stride = n_rows; a.aux_stride = aux_stride;
    a.pair_rows = epilogue == LANTERN_EPI_SILU_MUL ? pair_rows : 0;
    a.a_seg_rows = a_seg_rows; a.a_seg_stride = a_seg_stride;
    return sk_run(a, epilogue, packed != 0, false, workspace, workspace_bytes, st, "linear_rows_streamk");
}
}  // namespace lantern

extern "C" int lantern_linear_rows(const void *A, const void *W, const void *bias, int M, int K, int row_lo, int n_rows, void *out,
                                   int out_stride, int out_col0, void *stream) {
    LANTERN_CHECK_ARG(A && W && out, "linear_rows: null buffer");
    LANTERN_CHECK_ARG(M >= 0 && M <= 128, "linear_rows: M=%d must be <= 128 rows", M);
    LANTERN_CHECK_ARG(K > 0 && K % 16 == 0 && row_lo >= 0 && n_rows >= 0 && out_col0 >= 0 && out_stride >= out_col0 + n_rows,
                      "linear_rows: K=%d must be a multiple of 16, the output row must hold [col0, col0 + n_rows)", K);
    if (M == 0 || n_rows == 0) return LANTERN_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n_rows + 31) / 32), block(FC_THREADS);
    const uint16_t *a = (const uint16_t *)A, *w = (const uint16_t *)W, *bi = (const uint16_t *)bias;
    uint16_t *o = (uint16_t *)out;
    if (M <= 32) LANTERN_LAUNCH((linear_rows_kernel<1, 0>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, (const uint16_t *)nullptr, 0, 0);
    else if (M <= 64) LANTERN_LAUNCH((linear_rows_kernel<2, 0>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, (const uint16_t *)nullptr, 0, 0);
    else if (M <= 96) LANTERN_LAUNCH((linear_rows_kernel<3, 0>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, (const uint16_t *)nullptr, 0, 0);
    else LANTERN_LAUNCH((linear_rows_kernel<4, 0>), grid, block, 0, st, a, w, bi, M, K, row_lo, n_rows, o, out_stride, out_col0, (const uint16_t *)nullptr, 0, 0);
    LANTERN_CHECK_LAUNCH("linear_rows");
    return LANTERN_OK;
}
