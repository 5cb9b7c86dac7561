// window_dev.h -- device helpers shared by the windowed kernel sets (window_kernels.hip: O7 windowed + the per-sequence
// chain kernel of O8; node_kernels.hip: the node-parallel O8).  Moved verbatim out of window_kernels.hip.
#pragma once
#include "common.h"

// diagnosis: window_kernels.hip's trace build (-DEPW_TRACE=3) defines this before including the header; a no-op everywhere else
#ifndef EPW_STAMPG
#define EPW_STAMPG(id) do { } while (0)
#endif

namespace lantern {

__device__ __forceinline__ int64_t py_mod64(int64_t a, int64_t b) {
    int64_t r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? r + b : r;
}

// k-th largest over a register tile of E4 float4 held by NT threads: bitwise bisection on order-preserving
// keys, one counting pass + one barrier per bit.  HIGH16: every value is bf16-representable (low 16 bits of
// the float are zero), so the low half of the key is a function of the sign and 16 passes decide the key.
template <int NT, int E4, bool HIGH16>
__device__ __forceinline__ float kth_largest_tile(const float4 (&r)[E4], int k, int *redi, int &ph) {
    uint32_t prefix = 0;
    for (int bit = 31; bit >= (HIGH16 ? 16 : 0); --bit) {
        const uint32_t trial = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            c += float_key(r[it].x) >= trial;
            c += float_key(r[it].y) >= trial;
            c += float_key(r[it].z) >= trial;
            c += float_key(r[it].w) >= trial;
        }
        const int tot = block_sum<int, NT / 64>(c, redi, ph);
        if (tot >= k) prefix = trial;
    }
    if (prefix == 0) return -__builtin_inff();
    if (HIGH16 && !(prefix & 0x80000000u)) prefix |= 0xffffu;   // negative float: key = ~bits, low half all ones
    return key_float(prefix);
}

// k-th largest by radix-256 histograms in LDS: 2 passes for bf16-representable values (HIGH16), 4 for f32.
// Per pass: one no-return LDS atomic per live value, then every wave resolves the digit from the 256 counters with
// 4 bins per lane + one DPP scan (no serial loop, no extra barrier for a broadcast).  `hist` = 256 ints.
template <int NT, int E4, bool HIGH16>
__device__ __forceinline__ float kth_largest_hist(const float4 (&r)[E4], int k, int *hist) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t prefix = 0, mask = 0;
    int krem = k;
#pragma unroll 1
    for (int pass = 0; pass < (HIGH16 ? 2 : 4); ++pass) {
        const int shift = 24 - 8 * pass;
        for (int t = tid; t < 256; t += NT) hist[t] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const uint32_t kk[4] = {float_key(r[it].x), float_key(r[it].y), float_key(r[it].z), float_key(r[it].w)};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((kk[c] & mask) == prefix) atomicAdd(&hist[(kk[c] >> shift) & 255u], 1);
        }
        __syncthreads();
        const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const int s4 = c0 + c1 + c2 + c3;
        const int incl = wave_scan_incl_dpp(s4);
        const int total = readlane63(incl);
        if (total < krem) return -__builtin_inff();     // fewer than k values: the k-th largest is below everything
        int above = total - incl;                        // values in bins of higher lanes
        int digit = -1, kn = 0;
        // bins of this lane, high to low
        if (above < krem && above + c3 >= krem) { digit = 4 * lane + 3; kn = krem - above; }
        above += c3;
        if (digit < 0 && above < krem && above + c2 >= krem) { digit = 4 * lane + 2; kn = krem - above; }
        above += c2;
        if (digit < 0 && above < krem && above + c1 >= krem) { digit = 4 * lane + 1; kn = krem - above; }
        above += c1;
        if (digit < 0 && above < krem && above + c0 >= krem) { digit = 4 * lane; kn = krem - above; }
        const unsigned long long who = __ballot(digit >= 0);
        const int src = __ffsll((long long)who) - 1;
        digit = __shfl(digit, src, 64);
        krem = __shfl(kn, src, 64);
        prefix |= (uint32_t)digit << shift;
        mask |= 255u << shift;
        __syncthreads();   // everyone has read hist before the next pass clears it
    }
    if (HIGH16 && !(prefix & 0x80000000u)) prefix |= 0xffffu;   // negative float: key = ~bits, low half all ones
    return key_float(prefix);
}

// exp(x) for x <= 0 (softmax arguments): n = rint(x*log2e), r = x - n*ln2 (two-term), 2^(r*log2e) on the
// hardware exp unit, ldexp.  7 VALU ops, < 1 ulp like the libm/ocml routine it replaces (which costs ~20).
__device__ __forceinline__ float exp_nonpos(float x) {
    x = fmaxf(x, -104.0f);                              // exp(-104) is below half the smallest subnormal: rounds to 0 (also -inf)
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(-n, 0.693145751953125f, x);          // ln2 high part (exact product for |n| < 2^11)
    r = fmaf(-n, 1.42860682030941723e-6f, r);           // ln2 low part
    return ldexpf(__builtin_amdgcn_exp2f(r * 1.44269504088896341f), (int)n);
}

// x / d for many x and one d: reciprocal refined once (Newton), then q = fma(fma(-q0, d, x), r, q0) -- the quotient
// correction step of the IEEE division expansion without its per-element scaling / fix-up instructions.  Correctly
// rounded for the normal-range operands met here (d in [2^-20, 2^14], x in [0, 1]); 3 VALU ops instead of ~11.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct FastDiv {
    float d, r;
    __device__ __forceinline__ explicit FastDiv(float den) : d(den) {
        float r0 = __builtin_amdgcn_rcpf(den);
        r = fmaf(fmaf(-den, r0, 1.0f), r0, r0);
    }
    __device__ __forceinline__ float operator()(float x) const {
        const float q0 = x * r;
        return fmaf(fmaf(-q0, d, x), r, q0);
    }
    // two quotients per instruction (v_pk_mul_f32 / v_pk_fma_f32): same three operations per element
    __device__ __forceinline__ f32x2_t operator()(f32x2_t x) const {
        const f32x2_t q0 = x * r;
        return __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, (f32x2_t)(d), x), (f32x2_t)(r), q0);
    }
    __device__ __forceinline__ float4 operator()(float4 v) const {
        const f32x2_t a = (*this)((f32x2_t){v.x, v.y}), b = (*this)((f32x2_t){v.z, v.w});
        return make_float4(a.x, a.y, b.x, b.y);
    }
};

// max(a - b, 0) for a, b in [0, 1] (probabilities; no NaN), two elements per instruction: v_pk_add_f32 with the second operand negated and the
// output clamp, which bounds a VOP3P float result to [0, 1] -- the upper bound never binds (a - b <= 1), the lower one IS the max with zero.  Same
// bits as `d = a - b; d < 0 ? 0 : d` (a - b is never -0 for non-negative operands).  The residual update `max(gtp - q, 0)` of evaluate_posterior
// (ea_model_lumina_mgpt.py:703-705) was a subtract + compare + select per element (10 instructions per float4); this is 2.
__device__ __forceinline__ f32x2_t sub_clamp0_pk(f32x2_t a, f32x2_t b) {
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float4 sub_clamp0(float4 a, float4 b) {
    const f32x2_t lo = sub_clamp0_pk((f32x2_t){a.x, a.y}, (f32x2_t){b.x, b.y}), hi = sub_clamp0_pk((f32x2_t){a.z, a.w}, (f32x2_t){b.z, b.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// softmax of a register tile over the workgroup: max, exp, f64 sum, one division per element -- the arithmetic
// of the reference's torch.softmax(row) restated (oracle: lo_softmax_row); shared by O7 (rows emitted as
// probabilities) and O8 (rows arriving as logits) so that both produce the same bits.
// max of three without the NaN-quieting pre-pass fmaxf() compiles to (v_max x, x, x per operand): logits are never NaN here
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int NT, int NV4>
__device__ __forceinline__ void softmax_tile(float4 (&r)[NV4], float *redf, double *redd, int &ph) {
    constexpr int NW = NT / 64;
    float m = -__builtin_inff();
#pragma unroll
    for (int it = 0; it < NV4; ++it) m = max3_raw(m, max3_raw(r[it].x, r[it].y, r[it].z), r[it].w);
    m = block_max_fast<NW>(m, redf, ph);
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < NV4; ++it) {
        r[it].x = exp_nonpos(r[it].x - m); r[it].y = exp_nonpos(r[it].y - m);
        r[it].z = exp_nonpos(r[it].z - m); r[it].w = exp_nonpos(r[it].w - m);
        s += (double)r[it].x + (double)r[it].y + (double)r[it].z + (double)r[it].w;
    }
    const float sf = (float)block_sum_fast<double, NW>(s, redd, ph);
    const FastDiv dv(sf);
#pragma unroll
    for (int it = 0; it < NV4; ++it) r[it] = dv(r[it]);
}

// TopPLogitsWarper on a register tile (HF order: after the temperature, before top-k; drafters/utils.py:36-52 ->
// transformers TopPLogitsWarper): sort ascending, softmax, cumsum, remove every entry whose cumulative probability is
// <= 1 - top_p, never the last (largest) one.  No sort here: the removed set is a prefix of the ascending order, so a
// WEIGHTED radix select over the 32-bit order keys finds its end -- per 8-bit digit, probability mass per bin
// (LDS f64 atomics), bins walked in ascending order until the running mass (rounded to f32 like torch.cumsum's
// output) exceeds 1 - top_p.  Equal values at the boundary go in index order (a stable ascending sort), counted with a
// block scan; that path and the keep-the-last rule only run when they apply.
// CHUNK8: the tile holds 8-id chunks (r[2j], r[2j+1] = ids (tid + j*NT)*8 .. +7: the bf16 row kernels' layout) instead of 4-id chunks
// (r[it] = ids (tid + it*NT)*4 .. +3): the index order that breaks ties at the boundary, and the "last entry" of the keep-one rule, follow it.
template <int NT, int NV4, bool CHUNK8 = false>
__device__ void top_p_tile(float4 (&r)[NV4], float top_p, double *mass, float *redf, double *redd, int *redi, int &ph) {
    static_assert(!CHUNK8 || NV4 % 2 == 0, "8-id chunks are pairs of float4");
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float NEG_INF = -__builtin_inff();
    auto first_index = [&](int it) -> int { return CHUNK8 ? (tid + (it >> 1) * NT) * 8 + (it & 1) * 4 : (tid + it * NT) * 4; };
    float4 p[NV4];
#pragma unroll
    for (int it = 0; it < NV4; ++it) p[it] = r[it];
    softmax_tile<NT, NV4>(p, redf, redd, ph);
    const float thr = (float)(1.0 - (double)top_p);
    uint32_t prefix = 0, kmask = 0;
    double below = 0.0;
    bool none_cross = false;
#pragma unroll 1
    for (int pass = 0; pass < 4 && !none_cross; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int t = tid; t < 256; t += NT) mass[t] = 0.0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            const float rv[4] = {r[it].x, r[it].y, r[it].z, r[it].w}, pv[4] = {p[it].x, p[it].y, p[it].z, p[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t kk = float_key(rv[c]);
                if (rv[c] != NEG_INF && (kk & kmask) == prefix) atomicAdd(&mass[(kk >> shift) & 255u], (double)pv[c]);
            }
        }
        __syncthreads();
        const double c0 = mass[4 * lane], c1 = mass[4 * lane + 1], c2 = mass[4 * lane + 2], c3 = mass[4 * lane + 3];
        const double s4 = (c0 + c1) + (c2 + c3);
        const double before = below + wave_scan_incl_dpp(dpp_mov<0x138>(s4));      // mass of all lower bins (exclusive scan)
        int digit = -1;
        double upto = before;                                                       // mass below the chosen bin
        if ((float)(before + c0) > thr) digit = 4 * lane;
        else if ((float)(before + c0 + c1) > thr) { digit = 4 * lane + 1; upto = before + c0; }
        else if ((float)(before + c0 + c1 + c2) > thr) { digit = 4 * lane + 2; upto = before + c0 + c1; }
        else if ((float)(before + c0 + c1 + c2 + c3) > thr) { digit = 4 * lane + 3; upto = before + c0 + c1 + c2; }
        const unsigned long long who = __ballot(digit >= 0);
        if (who == 0ull) {
            none_cross = true;          // the whole row's mass stays <= 1 - top_p: everything but the last entry goes
        } else {
            const int src = __ffsll((long long)who) - 1;
            digit = __builtin_amdgcn_readlane(digit, src);
            const long long ub = __double_as_longlong(upto);
            below = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(ub >> 32), src) << 32) |
                                         (unsigned int)__builtin_amdgcn_readlane((int)(ub & 0xffffffffll), src));
            prefix |= (uint32_t)digit << shift;
            kmask |= 255u << shift;
        }
        __syncthreads();
    }
    if (none_cross) {
        // keep only the last entry of the ascending order: the largest key, highest index among equals
        uint32_t kmax = 0;
        int imax = -1;
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            const float rv[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t kk = float_key(rv[c]);
                const int idx = first_index(it) + c;
                if (kk > kmax || (kk == kmax && idx > imax)) { kmax = kk; imax = idx; }
            }
        }
        // block arg-max of (key, index): two integer reductions
        int *buf = redi + (ph & 1) * NW;
        ph ^= 1;
        const int kw = wave_max_i((int)(kmax >> 1));                     // order-preserving 31-bit compare first
        if (lane == 0) buf[wave] = kw;
        __syncthreads();
        int kb = buf[0];
        for (int w = 1; w < NW; ++w) kb = max(kb, buf[w]);
        // (the dropped low bit: resolve among the candidates by the full key)
        int *buf2 = redi + (ph & 1) * NW;
        ph ^= 1;
        const int full = ((int)(kmax >> 1) == kb) ? (int)(kmax & 1u) : -1;
        const int fw = wave_max_i(full);
        if (lane == 0) buf2[wave] = fw;
        __syncthreads();
        int fb = buf2[0];
        for (int w = 1; w < NW; ++w) fb = max(fb, buf2[w]);
        const uint32_t kbest = ((uint32_t)kb << 1) | (uint32_t)fb;
        int *buf3 = redi + (ph & 1) * NW;
        ph ^= 1;
        const int iw = wave_max_i(kmax == kbest ? imax : -1);
        if (lane == 0) buf3[wave] = iw;
        __syncthreads();
        int ib = buf3[0];
        for (int w = 1; w < NW; ++w) ib = max(ib, buf3[w]);
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
            const int i0 = first_index(it);
            r[it].x = (i0 == ib) ? r[it].x : NEG_INF;
            r[it].y = (i0 + 1 == ib) ? r[it].y : NEG_INF;
            r[it].z = (i0 + 2 == ib) ? r[it].z : NEG_INF;
            r[it].w = (i0 + 3 == ib) ? r[it].w : NEG_INF;
        }
        __syncthreads();
        return;
    }
    // boundary value = key `prefix`; its holders all carry the same probability p*
    int tie_cnt[NV4], my_ties = 0;
    float pstar = 0.0f;
#pragma unroll
    for (int it = 0; it < NV4; ++it) {
        const float rv[4] = {r[it].x, r[it].y, r[it].z, r[it].w}, pv[4] = {p[it].x, p[it].y, p[it].z, p[it].w};
        tie_cnt[it] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (rv[c] != NEG_INF && float_key(rv[c]) == prefix) {
                ++tie_cnt[it];
                pstar = pv[c];
            }
        my_ties += tie_cnt[it];
    }
    const int group = block_sum_fast<int, NW>(my_ties, redi, ph);
    pstar = block_max_fast<NW>(pstar, redf, ph);
    // how many of the equal entries still fit under the threshold, taken one by one like the cumsum does
    int m = 0;
    {
        double acc = below;
        for (int j = 1; j < group; ++j) {          // the group's last entry is the one that crossed: at most group-1 go
            acc += (double)pstar;
            if ((float)acc <= thr) m = j;
            else break;
        }
    }
    int base[NV4];
    if (m > 0) {        // rank of every boundary-valued entry in index order: a slice = the ids one pass of the workgroup covers (ascending with the thread
                        // id), 4 per thread or -- CHUNK8 -- 8 per thread in two float4
        constexpr int SL = CHUNK8 ? 2 : 1, NS = NV4 / SL;
        __shared__ int s_tie_tot[NV4][NW];
        int incl[NS], cnt[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            cnt[sl] = tie_cnt[SL * sl] + (CHUNK8 ? tie_cnt[SL * sl + SL - 1] : 0);
            incl[sl] = wave_scan_incl_dpp(cnt[sl]);
            if (lane == 63) s_tie_tot[sl][wave] = incl[sl];
        }
        __syncthreads();
        int run = 0;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            int woff = 0, tot = 0;
            for (int w = 0; w < NW; ++w) {
                const int t = s_tie_tot[sl][w];
                woff += (w < wave) ? t : 0;
                tot += t;
            }
            base[SL * sl] = run + woff + (incl[sl] - cnt[sl]);
            if (CHUNK8) base[SL * sl + SL - 1] = base[SL * sl] + tie_cnt[SL * sl];
            run += tot;
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < NV4; ++it) {
        float rv[4] = {r[it].x, r[it].y, r[it].z, r[it].w};
        int rank = (m > 0) ? base[it] : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (rv[c] == NEG_INF) continue;
            const uint32_t kk = float_key(rv[c]);
            if (kk < prefix) rv[c] = NEG_INF;
            else if (kk == prefix) {
                if (rank < m) rv[c] = NEG_INF;
                ++rank;
            }
        }
        r[it] = make_float4(rv[0], rv[1], rv[2], rv[3]);
    }
}

constexpr int EW_MAX_P = 64, EW_MAX_D = 16, EW_MAX_PD = 1024, EW_MAX_SIB = 16, EW_MAX_N = 128, EW_MAX_B = 1024, EW_UNI = 64;
constexpr int EW_PF_C = 6;        // candidates per level whose neighbour ids are prefetched into LDS
constexpr int EW_PF_K = 1024;     // ... when k + 1 <= EW_PF_K; otherwise ids are read from HBM on demand

// Fixed-size part of the workgroup's LDS; the five per-(path, depth) tables (cand, row, cart, pidx, boff) follow it, sized
// by the launch's actual P*D (epw_pd_cap) instead of the 64 x 16 worst case: 20 KB -> ~2 KB for the reference's trees, which
// is what lets two workgroups share a CU at saturating batch sizes.
// PF_C / MAX_B / MAX_N: the staged sizes -- the defaults hold every tree the windowed kernels accept; the COMPACT throughput instance of the reference's
// default tree (26 nodes, 26 sibling entries, fan-out <= 4) takes the smallest that tree needs, which is what lets FOUR workgroups share a CU.
template <int PF_C = EW_PF_C, int MAX_B = EW_MAX_B, int MAX_N = EW_MAX_N, int N_UNI = EW_UNI, int N_SAMP = 64>
struct alignas(16) EwSharedT {
    static constexpr int kPfC = PF_C, kMaxB = MAX_B, kMaxN = MAX_N, kUni = N_UNI;
    unsigned short bidx[MAX_B];             // earlier-sibling node ids (< MAX_N)
    int tcand[MAX_N];
    int opoff[EW_MAX_D];
    double uni[N_UNI];
    double redd[2 * 16];
    float redf[2 * 16];
    int redi[2 * 16];
    double samp_tot[N_SAMP];                // [wave * E4 + it]: per-wave totals of the bonus draw's segments (NW * E4 <= N_SAMP)
    int dec[2][4];                          // decision words of wave 0: {code, m>0, csm1 bits, -}
    double ubonus[2];                       // [0]: the bonus draw's uniform, fetched with the prologue's first round of loads
    int hot[MAX_N];                         // row_hot of this sequence's rows (when rows_per_seq <= MAX_N)
    int pre[MAX_N];                         // LANTERN_ROWS_RAW_BF16: 1 = the row was post-processed up front (win.raw_probs)
    // Prefetched neighbours of a level's candidates as gather indices into g: window index, or a sentinel slot (0 outside the window, the
    // out-of-window one-hot token's mass, 3e38 at positions >= k so that they can never pass `<= tau`).  The raw table ids are not kept: the
    // zeroing after a rejection needs "window index or not" and "is it the out-of-window token", both of which the gather index says; the
    // one id it zeroes beyond the k it sums (position k, whose scan slot is the 3e38 sentinel) keeps its plain index in nbk.  12 KB less LDS per
    // workgroup than with both forms -- what lets three workgroups share a CU.
    unsigned short nbaddr[PF_C][EW_PF_K];
    unsigned short nbk[8];                  // per slot: plain gather index of neighbour k (the k+1-th, zeroed but never summed)
};
typedef EwSharedT<> EwShared;
typedef EwSharedT<2, 128, 32, 32, 32> EwSharedCompact;          // (256 threads x 8 float4: NW * E4 = 32 segment totals)
typedef EwSharedT<1, 128, 64, 64, 64> EwSharedLite;             // LlamaGen's 16384-id window at two workgroups per CU: trees of <= 64 nodes, LANTERN off (one unused neighbour slot)
static_assert(EW_PF_C <= 8, "nbk slots");

// g[W + EW_G_ZERO] = 0 (neighbour outside the window), g[W + EW_G_HUGE] = 3e38 (position >= k: never under tau),
// g[W + EW_G_OUT] = out_mass (neighbour == the one-hot token outside the window): gather targets of the scan
constexpr int EW_G_ZERO = 0, EW_G_HUGE = 1, EW_G_OUT = 2, EW_G_EXT = 4;
__host__ __device__ inline int epw_pd_cap(int P, int D) { return (P * D + 1 + 3) & ~3; }      // boff has P*D + 1 entries
__host__ __device__ inline size_t epw_shared_offset(int W, bool with_nbmask = true) {
    size_t o = (size_t)(W + EW_G_EXT) * 4 + (with_nbmask ? (size_t)((W + 31) / 32) * 4 : 0);      // (nbmask: LANTERN_MODE_STATIC_LG only)
    return (o + 15) & ~(size_t)15;
}

// softmax(processors(row)) -> g (LDS); one-hot rows put their mass in (out_tok,out_mass) when the hot token
// lies outside the window.
// value of lane `l` (wave-uniform index) without the LDS crossbar
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// `pre_barrier` runs after the row's global loads have landed and before the last barrier: LDS writes of data whose
// loads were issued BEFORE this call ride on the row's latency (vmcnt retires in issue order) and on its barrier.
//
// The row arrives in registers (`r`, loaded by row_load -- possibly long before, so that its HBM latency hides behind
// other work); one-hot rows ignore `r`.
template <int NT, int E4, bool FULLW = false>
__device__ __forceinline__ void row_load(const float *__restrict__ rowp, int W, float4 (&r)[E4]) {
    const float NEG_INF = -__builtin_inff();
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const int i4 = threadIdx.x + it * NT;
        r[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(rowp)[i4] : make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
    }
}

struct alignas(8) Bf16x8 {
    uint2 a, b;
};

// Histogram layout: [256 bins x O7_REP copies | 64 spill slots | 256 bins].  First pass: copy = lane % O7_REP.  An LDS atomic costs ~4.2 cycles
// per wave instruction plus ~2 cycles for every further lane that hits the same word (tools/probe/lds_atomic_probe.hip: 15 cycles at 8 lanes per
// word, 63 at 32), and a logit row puts most of its 8192 values into a handful of the 256 sign + exponent bins: with 8 copies (up to 8 lanes of an
// instruction on one word) the pass ran at ~17 cycles per instruction, with 16 copies at most 4 lanes share a word (O7 at 1664 rows 33.8 -> 32.1 us);
// 32 copies (32 KB) measure the same again -- clearing and merging them costs what the atomics save.
// Second pass: values outside the chosen top bin add into a spill slot (their own lane's, mostly) instead of being skipped under a branch: every
// atomic is unconditional (no exec-mask juggling per element) and the spill slots are conflict-free.  O7 keeps its VALU arbiters busy 0.71 of
// the launch (profiles/r03_pmc_sq.txt), so the address arithmetic of the 2 x 16 atomics per thread is written for instruction count: -inf
// (key 0x007f) counts in the lowest bin like any value (no compare + select per element: O7 31.5 -> 30.0 us, the raw-row walk 35.9 -> 34.9 us).
// The second pass counts into its own 256 words (O7_HIST1), cleared together with the first pass's copies: nothing has to be cleared -- and no
// barrier taken -- between reading the merged first-pass counts and adding the second pass's.
// Layout (32-bit words): [256 bins x O7_REP copies | 256 merged first-pass counts (O7_HIST2) | 256 second-pass counts (O7_HIST1) | 64 spill slots].
constexpr int O7_REP = 16;
constexpr int O7_HIST2 = 256 * O7_REP, O7_HIST1 = O7_HIST2 + 256, O7_SPILL = O7_HIST1 + 256, O7_HIST_INTS = O7_SPILL + 64;
static_assert(O7_HIST2 % 4 == 0 && O7_HIST1 % 4 == 0 && O7_REP % 4 == 0, "16-byte clears / merges");

template <int NT, int NV4>
__device__ __forceinline__ float kth_largest_hist_bf16(const float4 (&r)[NV4], int k, int *h) {
    const int tid = threadIdx.x, lane = tid & 63;
    int *hist = h + O7_HIST2;
    // order keys once: bf16-exact values -> the upper 16 key bits carry everything
    uint32_t key[NV4][4];
#pragma unroll
    for (int it = 0; it < NV4; ++it) {
        key[it][0] = float_key(r[it].x) >> 16; key[it][1] = float_key(r[it].y) >> 16;
        key[it][2] = float_key(r[it].z) >> 16; key[it][3] = float_key(r[it].w) >> 16;
    }
    // ---- pass 0: top 8 bits, replicated histogram
    for (int t = tid; t < O7_HIST2 / 4 + 64; t += NT)          // the copies, and the second pass's 256 words (what the spill slots hold is never read)
        reinterpret_cast<int4 *>(h)[t < O7_HIST2 / 4 ? t : O7_HIST1 / 4 + (t - O7_HIST2 / 4)] = make_int4(0, 0, 0, 0);
    __syncthreads();
    EPW_STAMPG(82);
    static_assert(O7_REP == 16, "byte offset of (bin, copy) below");
    char *const hb = reinterpret_cast<char *>(h);
    const uint32_t rep4 = (uint32_t)(lane & (O7_REP - 1)) << 2;
#pragma unroll
    for (int it = 0; it < NV4; ++it)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // byte offset of word (bin * 16 + copy), bin = key >> 8: two operations per value.  (-inf, key 0x007f, counts in the lowest bin like
            // any value: if the k-th largest is -inf the two passes find exactly that key)
            atomicAdd(reinterpret_cast<int *>(hb + (((key[it][c] >> 2) & 0x3fc0u) | rep4)), 1);
        }
    __syncthreads();
    EPW_STAMPG(83);
    for (int t = tid; t < 256; t += NT) {
        int sum = 0;
#pragma unroll
        for (int q = 0; q < O7_REP / 4; ++q) {
            const int4 a = *reinterpret_cast<const int4 *>(&h[t * O7_REP + 4 * q]);
            sum += a.x + a.y + a.z + a.w;
        }
        hist[t] = sum;
    }
    __syncthreads();
    EPW_STAMPG(84);
    uint32_t prefix = 0;
    int krem = k;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int *hp = pass == 0 ? hist : h + O7_HIST1;
        const int c0 = hp[4 * lane], c1 = hp[4 * lane + 1], c2 = hp[4 * lane + 2], c3 = hp[4 * lane + 3];
        const int s4 = c0 + c1 + c2 + c3;
        const int incl = wave_scan_incl_dpp(s4);
        const int total = readlane63(incl);
        if (total < krem) return -__builtin_inff();     // fewer than k values: the k-th largest is below everything
        int above = total - incl;                        // values in bins of higher lanes
        int digit = -1, kn = 0;
        if (above < krem && above + c3 >= krem) { digit = 4 * lane + 3; kn = krem - above; }
        above += c3;
        if (digit < 0 && above < krem && above + c2 >= krem) { digit = 4 * lane + 2; kn = krem - above; }
        above += c2;
        if (digit < 0 && above < krem && above + c1 >= krem) { digit = 4 * lane + 1; kn = krem - above; }
        above += c1;
        if (digit < 0 && above < krem && above + c0 >= krem) { digit = 4 * lane; kn = krem - above; }
        const unsigned long long who = __ballot(digit >= 0);
        const int src = __ffsll((long long)who) - 1;
        digit = __builtin_amdgcn_readlane(digit, src);
        krem = __builtin_amdgcn_readlane(kn, src);
        prefix |= (uint32_t)digit << (8 - 8 * pass);
        if (pass == 1) break;
        // ---- pass 1: low 8 bits among the values of the chosen top bin, single histogram in its own (already cleared) words.  A value of the bin
        // sits at key - (top << 8) in 0..255; anything else wraps to >= 256 and is capped at 256 + lane: the spill slots follow the 256 counts
        // (a value just above the bin may land in another lane's slot -- nobody reads them).  Subtract, min, shift: three operations per value.
        const uint32_t base = prefix & 0xff00u, cap = 256u + (uint32_t)lane;
#pragma unroll
        for (int it = 0; it < NV4; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t d = key[it][c] - base;
                atomicAdd(reinterpret_cast<int *>(hb + O7_HIST1 * 4 + ((d < cap ? d : cap) << 2)), 1);
            }
        __syncthreads();
        EPW_STAMPG(85);
    }
    prefix <<= 16;
    if (!(prefix & 0x80000000u)) prefix |= 0xffffu;   // negative float: key = ~bits, low half all ones
    return key_float(prefix);
}


// ---- LANTERN_ROWS_RAW_BF16: the row arrives as the target model's raw cond / uncond logits (bf16) and the whole post-process
// of tree_decoding (CFG combination, top-k threshold, softmax: ea_model_lumina_mgpt.py:597-607) runs HERE, for the rows the walk
// actually visits -- 2.7 + 1 of a tree's 26 rows per step -- instead of for every row in a separate launch.  Same code as
// cfg_window_bf16_kernel (same tile layout, same reductions), so the probabilities are the same bits.  The 16 row registers
// carry two 16-byte chunks of cond and two of uncond per thread until they are needed.
// CH: 16-byte chunks (8 ids) per thread and operand -- W = 8 * CH * NT: 2 on 512 threads (the latency instances), 4 on 256 threads (the
// throughput instances: half the waves per sequence).  rp[0..CH) hold cond, rp[CH..2 CH) uncond.
template <int NT, int CH = 2>
__device__ __forceinline__ void raw_row_load(const uint16_t *__restrict__ crow, const uint16_t *__restrict__ urow, float4 (&rp)[2 * CH]) {
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int ch = threadIdx.x + it * NT;
        const Bf16x8 c = *reinterpret_cast<const Bf16x8 *>(crow + ch * 8), u = *reinterpret_cast<const Bf16x8 *>(urow + ch * 8);
        rp[it] = make_float4(__uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.b.x), __uint_as_float(c.b.y));
        rp[CH + it] = make_float4(__uint_as_float(u.a.x), __uint_as_float(u.a.y), __uint_as_float(u.b.x), __uint_as_float(u.b.y));
    }
}

// `top_p` in (0, 1): TopPLogitsWarper between the CFG mix and the top-k threshold (HF order Temperature -> TopP -> TopK, drafters/utils.py:36-52:
// LlamaGen / Anole take top_p from generate()); its 256 f64 mass bins live in the front of the histogram buffer (the top-k select clears it afterwards).
// (NUCLEUS is a template parameter: the filter keeps a second copy of the row in registers, which the instances without it must not pay for --
// 162 -> 217 VGPRs for the chain kernel, 51 -> 94 for the row preparation when it was a run-time branch.)
template <int NT, typename Hook = NoHook, bool NUCLEUS = false, int CH = 2>
__device__ __forceinline__ void raw_row_to_lds(const float4 (&rp)[2 * CH], int hot, float cfg, int top_k, int V, int win_lo, int W, float *g,
                                               int &out_tok, float &out_mass, float *redf, double *redd, int *hist, int &ph, const Hook &pre_barrier = Hook(),
                                               float top_p = 1.0f, int *redi = nullptr) {
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
    float4 r[2 * CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const uint32_t cw[4] = {__float_as_uint(rp[it].x), __float_as_uint(rp[it].y), __float_as_uint(rp[it].z), __float_as_uint(rp[it].w)};
        const uint32_t uw[4] = {__float_as_uint(rp[CH + it].x), __float_as_uint(rp[CH + it].y), __float_as_uint(rp[CH + it].z), __float_as_uint(rp[CH + it].w)};
        float o[8];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            const f32x2_t t2 = cfg_mix_bf16x2(cw[q2], uw[q2], cfg);
            o[2 * q2] = t2.x;
            o[2 * q2 + 1] = t2.y;
        }
        r[2 * it] = make_float4(o[0], o[1], o[2], o[3]);
        r[2 * it + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    EPW_STAMPG(81);
    if constexpr (NUCLEUS) {
        if (top_p >= 1e-8f && top_p < 1.0f && redi) top_p_tile<NT, 2 * CH, true>(r, top_p, reinterpret_cast<double *>(hist), redf, redd, redi, ph);
    }
    if (top_k > 0 && top_k < V) {
        const float thr = (top_k <= W) ? kth_largest_hist_bf16<NT, 2 * CH>(r, top_k, hist) : NEG_INF;
#pragma unroll
        for (int it = 0; it < 2 * CH; ++it) {
            r[it].x = r[it].x < thr ? NEG_INF : r[it].x; r[it].y = r[it].y < thr ? NEG_INF : r[it].y;
            r[it].z = r[it].z < thr ? NEG_INF : r[it].z; r[it].w = r[it].w < thr ? NEG_INF : r[it].w;
        }
    }
    EPW_STAMPG(86);
    softmax_tile<NT, 2 * CH>(r, redf, redd, ph);
    EPW_STAMPG(88);
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        float *dst = g + (size_t)(tid + it * NT) * 8;
        *reinterpret_cast<float4 *>(dst) = r[2 * it];
        *reinterpret_cast<float4 *>(dst + 4) = r[2 * it + 1];
    }
    if (tid == 0) g[W + EW_G_OUT] = 0.0f;
    pre_barrier();
    __syncthreads();
}

// loads through the constant address space: uniform addresses become scalar loads (s_load), off the vector-memory queue.
// Only for memory no kernel of the same launch writes (tables, candidates, the uniform stream, cursors).
template <typename T>
__device__ __forceinline__ T ldc(const T *p) {
    return *(const __attribute__((address_space(4))) T *)(p);
}

__device__ __forceinline__ double rdlane(double v, int l) {
    const long long bits = __double_as_longlong(v);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), l) << 32) |
                                (unsigned int)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), l));
}

// Bonus token by inverse CDF in token-id order over the distribution in LDS (g = its window, + an optional (out_tok,
// out_mass) pair outside it): smallest id whose cumulative f64 mass exceeds u * total; the last positive id if rounding
// leaves none (lo_sample_inverse_cdf in the oracle).  Every thread takes 16 CONSECUTIVE ids, so one wave scan + the wave
// totals locate the thread that holds the crossing; only that thread looks at single entries.
template <int NT, int E4>
__device__ __forceinline__ int bonus_draw_lds(const float *g, int W, int lo, int out_tok, float out_mass, double u, double *wtot,
                                              int *bonus, int *redi, bool lazy = false, const FastDiv dgc = FastDiv(1.0f)) {
    constexpr int NW = NT / 64, EPT = 4 * E4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool out_before = out_tok >= 0 && out_tok < lo;
    float4 p[E4];
#pragma unroll
    for (int j = 0; j < E4; ++j) {
        const int e = tid * EPT + 4 * j;
        p[j] = (e < W) ? *reinterpret_cast<const float4 *>(g + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (lazy) p[j] = dgc(p[j]);          // g holds an unnormalised residual (see epn_kernel)
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < E4; ++j) s += (double)p[j].x + (double)p[j].y + (double)p[j].z + (double)p[j].w;
    const double inc = wave_scan_incl_dpp(s);
    if (lane == 63) wtot[wave] = inc;
    if (tid == 0) bonus[0] = 0x7fffffff;
    __syncthreads();
    double pre = 0.0, all = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const double t = wtot[w];
        pre += (w < wave) ? t : 0.0;
        all += t;
    }
    const double front = out_before ? (double)out_mass : 0.0;
    double total = front + all;
    if (out_tok >= 0 && !out_before) total += (double)out_mass;
    const double tgt = u * total;
    const double excl = front + pre + (inc - s);
    if (excl <= tgt && excl + s > tgt) {          // the crossing lies among this thread's ids
        double acc = excl;
        int found = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < E4; ++j) {
            const float v[4] = {p[j].x, p[j].y, p[j].z, p[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc += (double)v[c];
                found = min(found, (v[c] > 0.0f && acc > tgt) ? lo + tid * EPT + 4 * j + c : 0x7fffffff);
            }
        }
        if (found != 0x7fffffff) atomicMin(bonus, found);
    }
    __syncthreads();
    int token = bonus[0];
    if (out_before && out_mass > 0.0f && (double)out_mass > tgt) token = out_tok;
    if (token != 0x7fffffff) return token;
    // rare tail (u ~ 1, rounding at a thread boundary, or the mass sits behind the window): the exhaustive search
    int found = 0x7fffffff, last_pos = -1;
    if (out_before && out_mass > 0.0f) last_pos = out_tok;
    {
        double acc = excl;
#pragma unroll
        for (int j = 0; j < E4; ++j) {
            const float v[4] = {p[j].x, p[j].y, p[j].z, p[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc += (double)v[c];
                const bool pos = v[c] > 0.0f;
                const int id = lo + tid * EPT + 4 * j + c;
                last_pos = max(last_pos, pos ? id : -1);
                found = min(found, (pos && acc > tgt) ? id : 0x7fffffff);
            }
        }
    }
    if (out_tok >= 0 && !out_before && out_mass > 0.0f) {
        last_pos = max(last_pos, out_tok);
        if (total > tgt) found = min(found, out_tok);
    }
    found = wave_min_i(found);
    last_pos = wave_max_i(last_pos);
    if (lane == 0) {
        redi[wave] = found;
        redi[16 + wave] = last_pos;
    }
    __syncthreads();
    const int f = wave_min_i(lane < NW ? redi[lane] : 0x7fffffff);
    const int l = wave_max_i(lane < NW ? redi[16 + lane] : -1);
    return f != 0x7fffffff ? f : l;
}

}  // namespace lantern
