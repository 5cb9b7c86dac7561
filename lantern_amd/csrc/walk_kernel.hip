// walk_kernel.hip -- O8, "fast walk" form (lantern_evaluate_posterior_nodes with nodes->serial == 2).
//
// The relaxed rejection sampling of evaluate_posterior (models/ea_model_lumina_mgpt.py:610-726, eagle_version 1: static
// Lumina trees with LANTERN on) for the launch shape the serving loop runs: the 8192-id image window on the packed
// neighbour table, one 512-thread workgroup per sequence that walks root -> accepted child -> ... over the host-built
// node tables (lantern_tree_node_tables).  Same arithmetic, operation for operation, as epw_kernel / epn_serial_kernel
// (shared helpers in window_dev.h), so the three produce the same bits; what differs is WHEN things are computed:
//
//   * everything a mode / model switch would decide at run time is fixed at compile time (one mode, one window, one
//     table layout, rows as probabilities or as raw bf16 logits): the kernel carries no flag tests and a third of
//     the scalar state of the general chain kernel;
//   * the waves have ROLES and work ahead of the verdict instead of waiting for it.  Wave 0 is the serial engine (the
//     k-neighbour cumulative-mass scan and the accept test of candidate t, :650-684) and meets the others at ONE
//     barrier per candidate.  Meanwhile the PASS waves (the second half of the workgroup, owners of the residual)
//     compute max(gtp - q, 0) for the whole window into registers -- what a rejection needs (:695-707) is ready when
//     the verdict arrives; the candidate's neighbours are zeroed afterwards by the idle waves, their mass subtracted
//     from the sum -- and the LOAD waves request the candidate's own row and the neighbour ids of ITS first children
//     into a second LDS buffer: an acceptance (:685-689) swaps two pointers;
//   * the residual stays unnormalised after a rejection (its sum travels as a scalar; readers divide with the same
//     per-entry division `gtp /= gtp.sum()` performs), so the normalise pass, its LDS write and its barrier leave the chain.
//
// A tree whose sibling tokens are not distinct, or a launch outside this shape, belongs to the general kernels:
// LANTERN_ST_NEEDS_CHAIN per sequence / LANTERN_E_UNSUPPORTED per launch, never a guess.
#include "common.h"
#include "window_dev.h"

namespace lantern {

constexpr int FW_SLOTS = 4;          // children of a node whose neighbour ids are staged ahead (more children: staged on demand)
constexpr int FW_MAX_N = 128, FW_MAX_CH = 32, FW_UNI = 64, FW_INFO = 16;

#ifdef EPF_TRACE
constexpr int EPF_TR_MAX = 256, EPF_TR_BLOCKS = 64;
__device__ unsigned long long g_epf_trace[EPF_TR_BLOCKS][EPF_TR_MAX];
__device__ int g_epf_trace_n[EPF_TR_BLOCKS];
__shared__ unsigned long long s_epf_tr[EPF_TR_MAX];
__shared__ int s_epf_trn;
#define EPF_STAMP(id)                                                                                          \
    do {                                                                                                       \
        if (threadIdx.x == 0) {                                                                                \
            const int n__ = s_epf_trn;                                                                         \
            if (n__ < EPF_TR_MAX) {                                                                            \
                s_epf_tr[n__] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
                s_epf_trn = n__ + 1;                                                                           \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#else
#define EPF_STAMP(id) do { } while (0)
#endif

struct alignas(16) FwShared {
    int nd_kids[FW_MAX_N];                    // by node: child-list offset | children << 16 (0 for a leaf)
    int nd_qrow[FW_MAX_N];                    // by node: drafter row of its children
    int nd_info[FW_MAX_N];                    // by node: first path | depth << 8
    int2 child[FW_MAX_N];                     // child lists: {node, cell}
    int tok[FW_MAX_N];                        // tree_candidates by node (-2: outside [0, V))
    int hot[FW_MAX_N];                        // one-hot class of the node's row (-1: a window row)
    int pre[FW_MAX_N];                        // raw rows: 1 = post-processed up front (win.raw_probs)
    int fl[FW_MAX_N];                         // class of the node's token: bit 1 image token, bit 0 syntax token
    float cart[EW_MAX_PD];                    // cart_candidates_prob by cell
    double red[2][3][16];                     // [parity][residual sum | zeroed neighbours' mass | drafter-row sum][wave]
    float redf[2 * 16];
    double redd[2 * 16];
    int redi[2 * 16];
    double wtot[16];
    int bonus[4];
    int dec[8];                               // wave 0's words: [0] neighbours under tau exist (zero them on a rejection), [1] verdict
    int nbk[2][FW_SLOTS];                     // per staged child: byte offset into g of its (k+1)-th neighbour (zeroed, never summed)
    unsigned short nbaddr[2][FW_SLOTS][EW_PF_K];   // per staged child: its k neighbours as byte offsets into g (outside the window / beyond k: the zero slot)
};

struct FwArgs {
    lantern_ep_params prm;
    lantern_ep_buffers buf;
    lantern_ep_window win;
    const int32_t *tables;
    int32_t n_nodes, n_internal, n_children;
    int32_t root_qrow, root_nch;              // the root's facts, known on the host: its requests go out with the first instructions
    int32_t root_child[FW_SLOTS];             // node ids of its first children
};

// total of <= 16 per-wave partials held by the first lanes (one DPP row), in every lane
__device__ __forceinline__ double fw_row_total(double x) {
    x += dpp_mov<0x111>(x);
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    const long long bits = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 15), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 15);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// NT threads, window W = 16 * NT ids (512 -> the 8192-id Lumina / Anole image range; 64 * 16 = 1024 ids on 256 threads is the
// reduced-size window of the reference vectors: W = 4 * NT there).  RAW (8192 only): rows arrive as raw cond / uncond bf16 logits.
template <int NT, int E4, bool RAW>
__global__ __launch_bounds__(NT) void epf_kernel(const FwArgs args) {
    constexpr int NW = NT / 64, W = 4 * NT * E4;
    constexpr int NL = NW >= 8 ? 3 : 1;                 // load waves 1 .. NL
    constexpr int PW0 = 1 + NL, NP = NW - PW0;          // pass waves PW0 .. NW-1
    constexpr int PT = NP * 64, PC = (W / 4) / PT;      // pass threads; float4 chunks of the window per pass thread
    constexpr int RP = W * 4 / 1024, LC = (RP + NL - 1) / NL;      // 1 KB pieces of a row; pieces per load wave (register path)
    constexpr int DC = (RP + NW - 2) / (NW - 1);                    // ... per non-scan wave (LDS-DMA path)
    constexpr int SPW = (FW_SLOTS + NL - 1) / NL;       // id slots per load wave
    constexpr int ZOFF = (W + EW_G_ZERO) * 4;           // byte offset of the zero slot
    static_assert((W / 4) % PT == 0 && PC <= 8 && ZOFF < 65536, "pass tile");
    static_assert(!RAW || (NT == 512 && E4 == 4), "raw rows: the 8192-id window on 512 threads");
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    extern __shared__ float4 dyn_lds[];
    float *g = reinterpret_cast<float *>(dyn_lds);                       // the distribution the walk is testing against
    float *gs = g + (W + EW_G_EXT);                                      // staging: the row of the candidate being tested
    float *qc = gs + (W + EW_G_EXT);                                     // the drafter row of the current node's children (what its rejections subtract)
    float *qs = qc + W;                                                  // staging: the drafter row one level down
    FwShared &S = *reinterpret_cast<FwShared *>(reinterpret_cast<char *>(g) + ((size_t)2 * (W + EW_G_EXT) + (size_t)2 * W) * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_scan = wave == 0, is_load = wave >= 1 && wave <= NL, is_pass = wave >= PW0;
    const int pt = tid - PW0 * 64;                                       // pass thread index
    const int b = blockIdx.x;
    const int lo = win.win_lo, V = prm.V, D = prm.D, N = args.n_nodes, npd = prm.P * prm.D, k = prm.k, off = prm.tok_offset;
    const int rps = prm.rows_per_seq;
    const int nz = (k + 1 < prm.table_cols) ? k + 1 : prm.table_cols;
    const int n_int = args.n_internal;
    const int32_t *tb = args.tables;
    const int32_t *chl = tb + 8 + FW_INFO * n_int, *nodeinfo = chl + 4 * args.n_children;
    int ph = 0;
#ifdef EPF_TRACE
    if (tid == 0) s_epf_trn = 0;
    EPF_STAMP(0);
#endif

    // ------------------------------------------------------------------------------------------------ prologue
    const int ucur0 = buf.cursor ? ldc(buf.cursor + b) : 0;
    const double ub = win.u_bonus ? ldc(win.u_bonus + b) : 0.0;
    const float *probs = (RAW ? win.raw_probs : buf.logits);
    if (probs) probs += (size_t)b * rps * W;
    const uint16_t *raw_c = RAW ? reinterpret_cast<const uint16_t *>(buf.logits) + (size_t)b * rps * V + lo : nullptr;
    const uint16_t *raw_u = RAW ? reinterpret_cast<const uint16_t *>(win.raw_uncond) + (size_t)b * rps * V + lo : nullptr;
    const float *qbase = buf.orig_prob + (size_t)b * prm.R * (size_t)win.orig_prob_stride + win.orig_prob_offset;
    // ---- role helpers ------------------------------------------------------------------------------------------------
    // LOAD waves: neighbour ids of the first FW_SLOTS children of node `pn` (slot s belongs to load wave 1 + s % NL; 2 x 16 bytes per lane)
    uint4 idq[SPW][2];
    auto ids_request = [&](int pn, int first) {
        const int kd = S.nd_kids[pn & (FW_MAX_N - 1)];
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            idq[j][0] = make_uint4(0u, 0u, 0u, 0u);
            idq[j][1] = make_uint4(0u, 0u, 0u, 0u);
            const int slot = (wave - 1) + j * NL, ci = first + slot;
            if (slot < FW_SLOTS && ci < (kd >> 16)) {
                const int cn = S.child[((kd & 0xffff) + ci) & (FW_MAX_N - 1)].x;
                const int x = S.tok[cn & (FW_MAX_N - 1)];
                const int trow = x - off;
                if (x >= prm.img_lo && x < prm.img_hi && trow >= 0 && trow < prm.table_rows) {
                    const uint16_t *row = buf.nn_table + (size_t)trow * prm.table_cols;
                    if (lane * 8 < nz) idq[j][0] = *reinterpret_cast<const uint4 *>(row + lane * 8);
                    if ((lane + 64) * 8 < nz) idq[j][1] = *reinterpret_cast<const uint4 *>(row + (lane + 64) * 8);
                }
            }
        }
    };
    // ... parked as byte offsets into g: positions < k inside the window -> 4 * id, everything else -> the zero slot; the
    // (k+1)-th neighbour (zeroed on a rejection, never summed) apart
    auto ids_store = [&](int set) {
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int slot = (wave - 1) + j * NL;
            if (slot < FW_SLOTS) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int t0 = (lane + 64 * h) * 8;
                    const uint32_t w[4] = {idq[j][h].x, idq[j][h].y, idq[j][h].z, idq[j][h].w};
                    uint32_t ad[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t i0 = w[q] & 0xffffu, i1 = w[q] >> 16;
                        const uint32_t a0 = (t0 + 2 * q < k && i0 < (uint32_t)W) ? i0 * 4u : (uint32_t)ZOFF;
                        const uint32_t a1 = (t0 + 2 * q + 1 < k && i1 < (uint32_t)W) ? i1 * 4u : (uint32_t)ZOFF;
                        ad[q] = a0 | (a1 << 16);
                        if (t0 + 2 * q == k) S.nbk[set][slot] = (k < nz && i0 < (uint32_t)W) ? (int)(i0 * 4u) : ZOFF;
                        if (t0 + 2 * q + 1 == k) S.nbk[set][slot] = (k < nz && i1 < (uint32_t)W) ? (int)(i1 * 4u) : ZOFF;
                    }
                    *reinterpret_cast<uint4 *>(&S.nbaddr[set][slot][t0]) = make_uint4(ad[0], ad[1], ad[2], ad[3]);
                }
                if (k >= EW_PF_K && lane == 0) S.nbk[set][slot] = ZOFF;      // (no staged position k)
            }
        }
    };
    // LOAD waves: a 4 * W byte row (probabilities of a node, or a drafter row) from HBM straight into an LDS buffer: LDS-DMA, 1 KB
    // pieces (16 bytes per lane), no registers in between; fw_dma_wait() before the barrier that publishes the buffer
    auto row_dma = [&](const float *src, float *dst) {          // (called by the load waves)
#pragma unroll
        for (int i = 0; i < LC; ++i) {
            const int p = (wave - 1) + i * NL;
            if (p < RP)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
        }
    };
    auto row_dma_all = [&](const float *src, float *dst) {      // (called by every wave but wave 0)
#pragma unroll
        for (int i = 0; i < DC; ++i) {
            const int p = (wave - 1) + i * (NW - 1);
            if (p < RP)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
        }
    };
    auto fw_dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // raw rows that were not post-processed up front: the cond / uncond bf16 windows (8-byte aligned: through registers)
    float4 rr[RAW ? LC : 1];
    auto raw_request = [&](int cn) {
        if constexpr (RAW) {
#pragma unroll
            for (int i = 0; i < LC; ++i) {
                const int p = (wave - 1) + i * NL;
                rr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < RP) {          // pieces 0 .. RP/2-1: the cond window, then the uncond window
                    const uint16_t *src = (p < RP / 2 ? raw_c : raw_u) + (size_t)cn * V + (size_t)(p % (RP / 2)) * 512 + lane * 8;
                    const Bf16x8 c = *reinterpret_cast<const Bf16x8 *>(src);
                    rr[i] = make_float4(__uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.b.x), __uint_as_float(c.b.y));
                }
            }
        }
    };
    auto raw_store = [&](float *dst) {
        if constexpr (RAW) {
#pragma unroll
            for (int i = 0; i < LC; ++i) {
                const int p = (wave - 1) + i * NL;
                if (p < RP) *reinterpret_cast<float4 *>(dst + p * 256 + lane * 4) = rr[i];
            }
        }
    };
    auto q_src = [&](int pn) -> const float * {
        int qr = S.nd_qrow[pn & (FW_MAX_N - 1)];
        qr = qr < 0 ? 0 : (qr >= prm.R ? prm.R - 1 : qr);
        return qbase + (size_t)qr * (size_t)win.orig_prob_stride;
    };

    // ---- the root's requests first: everything whose address the kernel arguments give.  Its row (probability form) and its
    // drafter row by LDS-DMA straight into g / qc; raw rows: the raw chunks too (used when the root was not post-processed up
    // front); its first children's tokens -> their table rows (two dependent rounds, started now)
    float4 rp[E4];
    int out_tok = -1;
    float out_mass = 0.0f;
    if (!is_scan) {
        if (probs) row_dma_all(probs, g);
        if (args.root_nch > 0) {
            int qr = args.root_qrow;
            qr = qr < 0 ? 0 : (qr >= prm.R ? prm.R - 1 : qr);
            row_dma_all(qbase + (size_t)qr * (size_t)win.orig_prob_stride, qc);
        }
    }
    if constexpr (RAW) raw_row_load<NT>(raw_c, raw_u, rp);
    if (is_load) {
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            idq[j][0] = make_uint4(0u, 0u, 0u, 0u);
            idq[j][1] = make_uint4(0u, 0u, 0u, 0u);
            const int slot = (wave - 1) + j * NL;
            if (slot < FW_SLOTS && slot < args.root_nch) {
                const int64_t t64 = buf.tree_cand[(size_t)b * prm.N + (args.root_child[slot] & (FW_MAX_N - 1))];
                const int x = (t64 < 0 || t64 >= V) ? -2 : (int)t64;
                const int trow = x - off;
                if (x >= prm.img_lo && x < prm.img_hi && trow >= 0 && trow < prm.table_rows) {
                    const uint16_t *row = buf.nn_table + (size_t)trow * prm.table_cols;
                    if (lane * 8 < nz) idq[j][0] = *reinterpret_cast<const uint4 *>(row + lane * 8);
                    if ((lane + 64) * 8 < nz) idq[j][1] = *reinterpret_cast<const uint4 *>(row + (lane + 64) * 8);
                }
            }
        }
    }
    EPF_STAMP(3);
    // wave 0 keeps the step's uniforms in registers (lane l: the l-th draw from the cursor on): nobody else reads them
    double un_l = 2.0;
    if (is_scan && ucur0 + lane < prm.n_uniforms) un_l = buf.uniforms[(size_t)b * prm.n_uniforms + ucur0 + lane];
    EPF_STAMP(4);
    // ---- the sequence's small facts -> LDS
    {
        int4 ni = make_int4(0, 0, -1, -1), e03 = make_int4(0, 0, 0, 0), chv = make_int4(0, 0, 0, 0);
        int tok_ = -1, hot_ = -1, pre_ = 0, e5 = 0;
        if (tid < N) {
            ni = *reinterpret_cast<const int4 *>(nodeinfo + 4 * tid);
            const int64_t t64 = buf.tree_cand[(size_t)b * prm.N + tid];
            tok_ = (t64 < -1 || t64 >= V) ? -2 : (int)t64;
        }
        if (tid < rps && tid < FW_MAX_N) {
            if constexpr (RAW) {      // the row's class from its position (MultiModalLogitsProcessor, ea_model_lumina_mgpt.py:45-86)
                const int64_t n1 = (win.raw_pos_per_seq ? win.raw_pos_ids[(size_t)b * rps + tid] : win.raw_pos_ids[tid] + win.raw_seq_len[b]) - win.raw_pos_base + 1;
                hot_ = (n1 == ((int64_t)win.raw_w_latent + 1) * win.raw_h_latent + 1) ? win.raw_eos_id
                       : (py_mod64(n1, (int64_t)win.raw_w_latent + 1) == 0 ? win.raw_newline_id : -1);
                pre_ = (win.raw_pre && win.raw_probs) ? (int)win.raw_pre[tid] : 0;
            } else if (win.row_hot) {
                hot_ = win.row_hot[(size_t)b * rps + tid];
            }
        }
        if (tid < n_int) {
            e03 = *reinterpret_cast<const int4 *>(tb + 8 + FW_INFO * tid);          // {node, child offset, children, depth}
            e5 = tb[8 + FW_INFO * tid + 5];                                         // drafter row
        }
        if (tid < args.n_children) chv = *reinterpret_cast<const int4 *>(chl + 4 * tid);     // {node, cell, ., .}
        constexpr int CT = (EW_MAX_PD + NT - 1) / NT;
        float ct_[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int t = tid + u * NT;
            ct_[u] = t < npd ? buf.cart_prob[(size_t)b * npd + t] : 1.0f;
        }
        EPF_STAMP(5);
        // ---- park them
        if (tid < FW_MAX_N) {
            S.nd_info[tid] = (ni.x & 255) | ((ni.y & 255) << 8);
            if (ni.z < 0) {               // a leaf (internal nodes are written through their rank entry below)
                S.nd_kids[tid] = 0;
                S.nd_qrow[tid] = 0;
            }
            S.tok[tid] = tok_;
            int fl_ = (tok_ >= prm.img_lo && tok_ < prm.img_hi) ? 2 : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) fl_ |= (q < prm.n_syntax && tok_ == prm.syntax[q]) ? 1 : 0;
            S.fl[tid] = fl_;
            S.hot[tid] = hot_;
            S.pre[tid] = pre_;
        }
        if (tid < n_int) {
            S.nd_kids[e03.x & (FW_MAX_N - 1)] = (e03.y & 0xffff) | (e03.z << 16);
            S.nd_qrow[e03.x & (FW_MAX_N - 1)] = e5;
        }
        if (tid < args.n_children && tid < FW_MAX_N) S.child[tid] = make_int2(chv.x, chv.y);
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int t = tid + u * NT;
            if (t < EW_MAX_PD) S.cart[t] = ct_[u];
        }
        if (tid == 0) {
            g[W + EW_G_ZERO] = 0.0f;
            g[W + EW_G_HUGE] = 3.0e38f;
            g[W + EW_G_OUT] = 0.0f;
            gs[W + EW_G_ZERO] = 0.0f;
            gs[W + EW_G_HUGE] = 3.0e38f;
            gs[W + EW_G_OUT] = 0.0f;
        }
    }
    EPF_STAMP(6);
    if (is_load) ids_store(0);
    if (!is_scan) fw_dma_wait();
    __syncthreads();
    EPF_STAMP(1);
    // ---- the root's row as the walk needs it (g holds its probability form if there is one)
    {
        const int hot = S.hot[0];
        if (hot >= 0) {
            const bool inside = hot >= lo && hot < lo + W;
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int e = lo + i4 * 4;
                if (hot >= e && hot < e + 4) set_comp(v, hot - e, 1.0f);
                reinterpret_cast<float4 *>(g)[i4] = v;
            }
            if (!inside) {
                out_tok = hot;
                out_mass = 1.0f;
            }
            __syncthreads();
        } else if constexpr (RAW) {
            if (!(probs && S.pre[0] != 0))
                raw_row_to_lds<NT>(rp, -1, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, reinterpret_cast<int *>(qs), ph);
        }
    }

    EPF_STAMP(2);

    int node = 0, status = LANTERN_ST_OK, idset = 0;
    int n_tried = 0, n_rej = 0, n_used = 0, rej_here = 0;
    float gsum = 1.0f;
    bool lazy = false, q_have = true;

    // ------------------------------------------------------------------------------------------------ the walk
    for (;;) {
        // ---- arrival at `node`: its row is in g, its children's neighbour ids in id set `idset`
        const int kd = S.nd_kids[node];
        const int coff = kd & 0xffff, nch_all = kd >> 16;
        const int nch = nch_all < FW_MAX_CH ? nch_all : FW_MAX_CH;
        if (nch_all > FW_MAX_CH) status = LANTERN_ST_TREE_LIMIT;
        // lane t of every wave: child t (node, token, class, cart_candidates_prob)
        int node_l = 0, tok_l = -3 - lane, fl_l = 0;
        float qx_l = 1.0f;
        if (lane < nch) {
            const int2 ce = S.child[(coff + lane) & (FW_MAX_N - 1)];
            node_l = ce.x & (FW_MAX_N - 1);
            tok_l = S.tok[node_l];
            qx_l = S.cart[(ce.y >= 0 && ce.y < EW_MAX_PD) ? ce.y : 0];
            fl_l = S.fl[node_l];
        }
        {       // duplicate / missing sibling tokens: the node view does not hold here (the chain kernel's case)
            bool dup = lane < nch && tok_l == -1;
            for (int u = 0; u + 1 < nch; ++u) dup |= lane > u && lane < nch && tok_l == rdlane(tok_l, u);
            if (__ballot(dup) != 0ull) status = LANTERN_ST_NEEDS_CHAIN;
        }
        // the drafter row of this node's children and q[child tokens]
        if (nch > 0 && !q_have) {        // (the accepted child was not the first one tried: its drafter row was not requested ahead)
            if (is_load) {
                row_dma(q_src(node), qc);
                fw_dma_wait();
            }
            __syncthreads();
        }
        float qv_l = 0.0f;               // q[token of child `lane`]: what the later siblings' q.sum() loses
        if (lane < nch && tok_l >= lo && tok_l < lo + W) qv_l = qc[tok_l - lo];
        gsum = 1.0f;
        lazy = false;
        rej_here = 0;
        EPF_STAMP(10);

        unsigned zmask = 0;              // pass threads: which of their entries of the drafter row belong to earlier siblings
        double rem = 0.0, sq = 0.0;
        bool sq_ready = false;
        int acc_t = -1, staged_lo = 0;   // children [staged_lo, staged_lo + FW_SLOTS) have their ids in the slots

        for (int t = 0; t < nch && status == LANTERN_ST_OK; ++t) {
            if (t > 0) {                 // child t - 1 is now an earlier sibling (q[siblings] = 0; q /= q.sum(), :696-700)
                const int xp = rdlane(tok_l, t - 1) - lo;
                if (is_pass && xp >= 0 && xp < W && ((xp >> 2) % PT) == pt) zmask |= 1u << ((((xp >> 2) / PT) << 2) | (xp & 3));
                rem += (double)rdlane(qv_l, t - 1);
            }
            const int x = rdlane(tok_l, t);
            if (x == -2) {
                status = LANTERN_ST_TOKEN_OOB;
                break;
            }
            if (n_used >= FW_UNI || ucur0 + n_used >= prm.n_uniforms) {
                status = LANTERN_ST_UNIFORMS;
                break;
            }
            const double rr_u = rdlane(un_l, n_used & 63);      // (wave 0's lanes hold the draws; the other waves take its verdict)
            ++n_used;
            ++n_tried;
            const int cnode = rdlane(node_l, t);
            const int fl = rdlane(fl_l, t);
            const bool in_img = (fl & 2) != 0, is_syn = prm.syntax_shortcut && (fl & 1) != 0;
            const float qx = rdlane(qx_l, t);
            EPF_STAMP(20);
            if (qx <= 0.0f) continue;                     // skipped (:680-682): the draw is spent, nothing else happens
            const bool scan = !is_syn && in_img;
            const int trow = x - off;
            if (scan && !(trow >= 0 && trow < prm.table_rows)) {
                status = LANTERN_ST_TABLE_OOB;
                break;
            }
            if (t >= staged_lo + FW_SLOTS) {              // more children than staging slots (rare): the next FW_SLOTS of them, synchronously
                __syncthreads();
                if (is_load) {
                    ids_request(node, t);
                    ids_store(idset);
                }
                staged_lo = t;
                __syncthreads();
            }
            if (t > 0 && !sq_ready) {                     // (every earlier child was skipped: S_q has not been reduced yet)
                double sl = 0.0;
                if (is_pass) {
#pragma unroll
                    for (int it = 0; it < PC; ++it) {
                        const float4 qv = reinterpret_cast<const float4 *>(qc)[pt + it * PT];
                        sl += (double)qv.x + (double)qv.y + (double)qv.z + (double)qv.w;
                    }
                }
                sq = block_sum_fast<double, NW>(sl, S.redd, ph);
                sq_ready = true;
            }
            const int slot = t - staged_lo;
            const int ckd = S.nd_kids[cnode];
            const int chot = S.hot[cnode];
            const bool c_probs = !RAW || (probs && S.pre[cnode] != 0);
            int code = is_syn ? (((float)rr_u <= 1.0f / qx) ? 1 : 2) : (((float)rr_u <= 0.0f / qx) ? 1 : 2);      // px = 1 / px = 0 (:654-659); scanned below
            int m0 = 0;
            float4 gn[PC];
            double loc = 0.0;
            const FastDiv dgc(gsum);

            if (is_scan) {
                // ---------------- wave 0: the k-neighbour cumulative mass (:661-677), 16 consecutive neighbours per lane
                if (scan) {
                    const char *gb = reinterpret_cast<const char *>(g);
                    float px = g[x - lo];
                    const uint4 a = *reinterpret_cast<const uint4 *>(&S.nbaddr[idset][slot][lane * 16]);
                    const uint4 bq = *reinterpret_cast<const uint4 *>(&S.nbaddr[idset][slot][lane * 16 + 8]);
                    const uint32_t w[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
                    float f[16];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        f[2 * c] = *reinterpret_cast<const float *>(gb + (w[c] & 0xffffu));
                        f[2 * c + 1] = *reinterpret_cast<const float *>(gb + (w[c] >> 16));
                    }
                    if (lazy) {             // the window holds an unnormalised residual (the zero slot: 0 / gsum = 0)
                        px = dgc(px);
#pragma unroll
                        for (int c = 0; c < 16; ++c) f[c] = dgc(f[c]);
                    }
                    const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                    m0 = (rdlane(f[0], 0) <= tau) ? 1 : 0;      // neighbours under tau exist iff the first one is (the cumulative mass does not decrease)
                    double v[16], l = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        l += (double)f[c];
                        v[c] = l;
                    }
                    const double excl = wave_scan_incl_dpp(dpp_mov<0x138>(l));      // exclusive: scan of the lane totals shifted up one lane
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float cs = (float)(excl + v[c]);
                        mx = (cs <= tau) ? cs : mx;                     // non-decreasing: the last one under tau is the largest
                    }
                    const float best_cs = wave_max(mx);
                    if (best_cs > -__builtin_inff()) px = px + best_cs;
                    code = ((float)rr_u <= px / qx) ? 1 : 2;
                }
                if (lane == 0) {
                    S.dec[0] = m0;
                    S.dec[1] = code;
                }
                EPF_STAMP(24);
            } else if (is_load) {
                // ---------------- load waves: "this candidate is accepted" -- its row and its first children's neighbour ids
                // (stored into the staging buffers before the verdict's barrier: an acceptance finds them in LDS)
                const bool want_q = t == 0 && (ckd >> 16) > 0;       // the drafter row one level down, for the first (likeliest) child only
                if (chot < 0) {
                    if (c_probs) row_dma(probs + (size_t)cnode * W, gs);
                    else raw_request(cnode);
                }
                ids_request(cnode, 0);
                if (want_q) row_dma(q_src(cnode), qs);
                if (chot < 0 && !c_probs) raw_store(gs);
                ids_store(idset ^ 1);
                fw_dma_wait();
            } else {
                // ---------------- pass waves: "this candidate is rejected" -- max(gtp - q, 0) of the whole window, into registers
                const FastDiv dq(t > 0 ? (float)(sq - rem) : 1.0f);
#pragma unroll
                for (int it = 0; it < PC; ++it) {
                    float4 qv = reinterpret_cast<const float4 *>(qc)[pt + it * PT];
                    if (t > 0) {
                        const unsigned z = zmask >> (4 * it);
                        qv.x = (z & 1u) ? 0.f : qv.x; qv.y = (z & 2u) ? 0.f : qv.y;
                        qv.z = (z & 4u) ? 0.f : qv.z; qv.w = (z & 8u) ? 0.f : qv.w;
                        qv = dq(qv);
                    }
                    float4 gv = reinterpret_cast<const float4 *>(g)[pt + it * PT];
                    if (lazy) gv = dgc(gv);
                    float d;
                    d = gv.x - qv.x; gv.x = d < 0.0f ? 0.0f : d;
                    d = gv.y - qv.y; gv.y = d < 0.0f ? 0.0f : d;
                    d = gv.z - qv.z; gv.z = d < 0.0f ? 0.0f : d;
                    d = gv.w - qv.w; gv.w = d < 0.0f ? 0.0f : d;
                    gn[it] = gv;
                    loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                }
            }
            __syncthreads();                                    // the verdict
            code = S.dec[1];
            m0 = S.dec[0];
            EPF_STAMP(21);
            if (code == 1) {
                // ------------------------------------------------ acceptance (:685-689): on to the child
                acc_t = t;
                const bool q_have_next = t == 0 && (ckd >> 16) > 0;
                out_tok = -1;
                out_mass = 0.0f;
                if (chot >= 0) {          // a one-hot row: written in place (g is dead)
                    const bool inside = chot >= lo && chot < lo + W;
#pragma unroll
                    for (int it = 0; it < E4; ++it) {
                        const int i4 = tid + it * NT;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        const int e = lo + i4 * 4;
                        if (chot >= e && chot < e + 4) set_comp(v, chot - e, 1.0f);
                        reinterpret_cast<float4 *>(g)[i4] = v;
                    }
                    if (!inside) {
                        out_tok = chot;
                        out_mass = 1.0f;
                    }
                    __syncthreads();
                } else if (c_probs) {      // the staged probabilities become the distribution: swap the buffers
                    float *tmp = g;
                    g = gs;
                    gs = tmp;
                } else if constexpr (RAW) {      // the staged raw chunks -> CFG, top-k, softmax -> g (every wave)
                    const char *sb = reinterpret_cast<const char *>(gs);
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int ch = tid + it * NT;
                        rp[it] = *reinterpret_cast<const float4 *>(sb + (size_t)ch * 16);
                        rp[2 + it] = *reinterpret_cast<const float4 *>(sb + (size_t)W * 2 + (size_t)ch * 16);
                    }
                    // (the histograms live in the drafter-row buffer this acceptance retires)
                    raw_row_to_lds<NT>(rp, -1, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd,
                                       reinterpret_cast<int *>(q_have_next ? qc : qs), ph);
                }
                idset ^= 1;
                node = cnode;
                q_have = q_have_next;
                if (q_have) {             // the staged drafter row becomes the current one
                    float *tmp = qc;
                    qc = qs;
                    qs = tmp;
                }
                break;
            }
            // ------------------------------------------------ rejection (:690-713)
            ++n_rej;
            ++rej_here;
            if (is_syn) {
                status = LANTERN_ST_SYNTAX_REJECT;
                break;
            }
            double(*red)[16] = S.red[ph & 1];
            ph ^= 1;
            // the unnormalised residual goes back to LDS (the pass threads own it)
            if (is_pass) {
#pragma unroll
                for (int it = 0; it < PC; ++it) reinterpret_cast<float4 *>(g)[pt + it * PT] = gn[it];
                loc = wave_sum(loc);
            }
            if (lane == 0) red[0][wave] = is_pass ? loc : 0.0;
            __syncthreads();
            // gtp[neighbours] = 0, k + 1 of them (:702-704): the waves that did not run the pass zero them in the residual and
            // take their mass off its sum; the pass waves reduce S_q meanwhile (a later child's q.sum() is S_q minus its
            // earlier siblings' entries)
            double rml = 0.0, sl = 0.0;
            if (!is_pass) {
                if (scan && m0) {
                    char *gb = reinterpret_cast<char *>(g);
                    for (int p = tid; p <= k && p < EW_PF_K + 1; p += PW0 * 64) {
                        const int ao = p < k ? (int)S.nbaddr[idset][slot][p] : S.nbk[idset][slot];
                        if (ao != ZOFF) {
                            rml += (double)*reinterpret_cast<const float *>(gb + ao);
                            *reinterpret_cast<float *>(gb + ao) = 0.0f;
                        }
                    }
                    rml = wave_sum(rml);
                }
            } else if (!sq_ready) {
#pragma unroll
                for (int it = 0; it < PC; ++it) {
                    const float4 qv = reinterpret_cast<const float4 *>(qc)[pt + it * PT];
                    sl += (double)qv.x + (double)qv.y + (double)qv.z + (double)qv.w;
                }
                sl = wave_sum(sl);
            }
            if (lane == 0) {
                red[1][wave] = rml;
                red[2][wave] = sl;
            }
            __syncthreads();
            double tot = fw_row_total(lane < NW ? red[0][lane] : 0.0) - fw_row_total(lane < NW ? red[1][lane] : 0.0);
            if (!sq_ready) {
                sq = fw_row_total(lane < NW ? red[2][lane] : 0.0);
                sq_ready = true;
            }
            tot += (double)out_mass;
            const float gs_ = (float)tot;
            if (gs_ == 0.0f) {
                status = LANTERN_ST_NEEDS_DENSE;      // `gtp.sum()==0 -> ones`: uniform over all V, only the dense kernel holds it
                break;
            }
            gsum = gs_;
            lazy = true;
            out_mass = out_mass / gs_;
            EPF_STAMP(30);
        }
        if (acc_t < 0 || status != LANTERN_ST_OK) break;
    }
    EPF_STAMP(40);

    // ------------------------------------------------------------------------------------------------ the end of the walk
    // g holds the last node's own row if nothing was rejected there, else the (unnormalised) residual
    const int info = S.nd_info[node];
    const int depth = (info >> 8) & 255, best = info & 255;
    const int from_res = (rej_here > 0 && depth + 1 != D) ? 1 : 0;
    int token = -1;
    if (status == LANTERN_ST_OK) {
        const FastDiv dgc(gsum);
        if (win.sample_win || buf.sample_p) {
            float4 p[E4];
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                p[it] = reinterpret_cast<const float4 *>(g)[tid + it * NT];
                if (lazy) p[it] = dgc(p[it]);
            }
            if (win.sample_win) {
                float *sw_ = win.sample_win + (size_t)b * W;
#pragma unroll
                for (int it = 0; it < E4; ++it) reinterpret_cast<float4 *>(sw_)[tid + it * NT] = p[it];
            }
            if (buf.sample_p) {
                float *sp = buf.sample_p + (size_t)b * V;
                for (int i4 = tid; i4 * 4 < V; i4 += NT) {
                    const int e = i4 * 4;
                    if (e + 4 <= lo || e >= lo + W) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (out_tok >= e && out_tok < e + 4) set_comp(v, out_tok - e, out_mass);
                        reinterpret_cast<float4 *>(sp)[i4] = v;
                    }
                }
#pragma unroll
                for (int it = 0; it < E4; ++it) reinterpret_cast<float4 *>(sp + lo)[tid + it * NT] = p[it];
            }
        }
        if (win.u_bonus) token = bonus_draw_lds<NT, E4>(g, W, lo, out_tok, out_mass, ub, S.wtot, S.bonus, S.redi, lazy, dgc);
    }
    EPF_STAMP(50);
#ifdef EPF_TRACE
    if (tid == 0 && b < EPF_TR_BLOCKS) {
        const int n = s_epf_trn;
        for (int t = 0; t < n; ++t) g_epf_trace[b][t] = s_epf_tr[t];
        g_epf_trace_n[b] = n;
    }
#endif
    if (tid == 0) {
        const int a = depth + 1;
        buf.best[b] = best;
        buf.accept_len[b] = depth;
        int32_t *c = buf.counters + (size_t)b * 6;
        c[0] = a < D - 1 ? a : D - 1;
        c[1] = n_tried;
        c[2] = n_rej;
        c[3] = n_used;
        c[4] = from_res;
        c[5] = status;
        if (buf.cursor) buf.cursor[b] = ucur0 + n_used;
        if (win.out_tok) win.out_tok[b] = out_tok;
        if (win.out_mass) win.out_mass[b] = out_mass;
        if (win.u_bonus && win.token && status == LANTERN_ST_OK) win.token[b] = token;
    }
}

}  // namespace lantern

using namespace lantern;

#ifdef EPF_TRACE
extern "C" int lantern_debug_epf_trace(unsigned long long *host_out, int *counts) {
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_epf_trace_n), sizeof(int) * EPF_TR_BLOCKS) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_epf_trace), sizeof(unsigned long long) * EPF_TR_BLOCKS * EPF_TR_MAX) != hipSuccess) return -1;
    return EPF_TR_MAX;
}
#endif

// Called by lantern_evaluate_posterior_nodes (node_kernels.hip) for nodes->serial == 2, after its common argument checks.
// Returns LANTERN_E_UNSUPPORTED (with the reason) for a launch outside the shape this kernel is built for.
int lantern_launch_fast_walk(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win,
                             const lantern_ep_nodes *nodes, void *stream) {
    const lantern_ep_params &p = *prm;
    const bool raw = win->rows_kind == LANTERN_ROWS_RAW_BF16;
    const int nz = (p.k + 1 < p.table_cols) ? p.k + 1 : p.table_cols;
    const char *why = nullptr;
    if (p.mode != LANTERN_MODE_STATIC_LUMINA || !p.lantern || !p.syntax_shortcut) why = "static Lumina trees with LANTERN on and the syntax shortcut";
    else if ((win->win_len != 8192 && win->win_len != 1024) || (raw && win->win_len != 8192) || win->win_lo != p.tok_offset || win->win_lo != p.img_lo || win->win_lo + win->win_len != p.img_hi) why = "an 8192-id (or 1024-id) image window (window = image range, table offset = window start; raw rows: 8192)";
    else if (p.table_cols % 8 != 0 || ((uintptr_t)buf->nn_table & 15) != 0 || nz > EW_PF_K || p.table_rows > win->win_len) why = "the packed neighbour table (lantern_pack_vq_table), k + 1 <= 1024";
    else if (!(raw || win->rows_kind == LANTERN_ROWS_PROBS)) why = "probability rows or raw bf16 rows";
    else if (nodes->n_nodes > FW_MAX_N || p.rows_per_seq > FW_MAX_N || p.P * p.D > EW_MAX_PD || p.P > 255 || p.D > 255 || p.R > 65535) why = "at most 128 nodes / 1024 cells";
    else if (p.top_p > 0.0f && p.top_p < 1.0f) why = "top_p off";
    if (why) {
        set_error("evaluate_posterior_nodes (fast walk): built for %s; use serial = 1 or evaluate_posterior_window", why);
        return LANTERN_E_UNSUPPORTED;
    }
    LANTERN_CHECK_ARG(nodes->tables_host && nodes->tables_host[4] == p.D && nodes->tables_host[5] == p.P, "evaluate_posterior_nodes (fast walk): the node tables were built for another [P, D]");
    LANTERN_CHECK_ARG(buf->cart_prob && buf->orig_prob && buf->tree_cand && buf->uniforms, "evaluate_posterior_nodes (fast walk): null required buffer");
    if (raw) {
        LANTERN_CHECK_ARG(win->raw_uncond && win->raw_pos_ids && (win->raw_seq_len || win->raw_pos_per_seq) && win->raw_w_latent > 0 && win->raw_h_latent > 0 &&
                              win->raw_newline_id >= 0 && win->raw_newline_id < p.V && win->raw_eos_id >= 0 && win->raw_eos_id < p.V,
                          "evaluate_posterior_nodes (fast walk): raw rows need the unconditional logits, positions, sequence lengths and the Lumina grammar ids");
        LANTERN_CHECK_ARG(p.top_k <= 0 && p.temperature == 1.0f && p.V % 8 == 0, "evaluate_posterior_nodes (fast walk): raw rows: the processors are the Lumina ones (raw_top_k)");
    } else {
        LANTERN_CHECK_ARG(p.top_k <= 0 && p.temperature == 1.0f, "evaluate_posterior_nodes (fast walk): probability rows are final");
    }
    FwArgs args{p, *buf, *win, nodes->tables, nodes->n_nodes, nodes->n_internal, nodes->n_children, 0, 0, {0, 0, 0, 0}};
    {           // the root's entry of the node tables (lantern_tree_node_tables: entries by rank, {node, child offset, children, depth, uoff, drafter row, ..})
        const int32_t *th = nodes->tables_host, *it = th + 8, *chh = it + FW_INFO * nodes->n_internal;
        for (int r = 0; r < nodes->n_internal; ++r)
            if (it[FW_INFO * r] == 0) {
                args.root_nch = it[FW_INFO * r + 2];
                args.root_qrow = it[FW_INFO * r + 5];
                for (int t = 0; t < FW_SLOTS && t < args.root_nch; ++t) args.root_child[t] = chh[4 * (it[FW_INFO * r + 1] + t)];
            }
    }
    const size_t lds = ((size_t)2 * (win->win_len + EW_G_EXT) + (size_t)2 * win->win_len) * 4 + sizeof(FwShared);
    hipStream_t st = (hipStream_t)stream;
    if (raw) LANTERN_LAUNCH((epf_kernel<512, 4, true>), dim3(p.B), dim3(512), lds, st, args);
    else if (win->win_len == 8192) LANTERN_LAUNCH((epf_kernel<512, 4, false>), dim3(p.B), dim3(512), lds, st, args);
    else LANTERN_LAUNCH((epf_kernel<256, 1, false>), dim3(p.B), dim3(256), lds, st, args);
    LANTERN_CHECK_LAUNCH("evaluate_posterior_nodes (fast walk)");
    return LANTERN_OK;
}
