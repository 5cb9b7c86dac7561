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
//   * wave 0 is the serial engine (candidate facts, the k-neighbour cumulative-mass scan, the accept test), the other
//     waves work AHEAD of its verdict instead of waiting for it: while wave 0 scans candidate t they (a) zero the
//     candidate's neighbours in the residual and compute max(gtp - q, 0) for their share of the window into registers
//     -- the residual a rejection needs (:695-707) is ready when the verdict arrives, a rejection costs wave 0's own
//     share plus one reduction -- and (b) have requested the candidate's own row, its drafter row's successor and
//     the neighbour ids of ITS first children, so that an acceptance (:685-689) finds the next level's inputs in
//     registers;
//   * the residual stays unnormalised after a rejection (its sum travels as a scalar; readers divide with the same
//     per-entry division `gtp /= gtp.sum()` performs), so the normalise pass, its LDS write and its barrier leave the chain.
//
// A tree whose sibling tokens are not distinct, or a launch outside this shape, belongs to the general kernels:
// LANTERN_ST_NEEDS_CHAIN per sequence / LANTERN_E_UNSUPPORTED per launch, never a guess.
#include "common.h"
#include "window_dev.h"

namespace lantern {

constexpr int FW_SLOTS = 6;          // children of a node whose neighbour ids are staged (worker wave w stages child w - 1; fewer waves: fewer slots)
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
    int nd_par[FW_MAX_N];                     // by node: parent
    int nd_dup[FW_MAX_N];                     // by node: 1 = its children carry duplicate / missing tokens (the node view does not hold)
    int2 child[FW_MAX_N];                     // child lists: {node, cell}
    int tok[FW_MAX_N];                        // tree_candidates by node (-2: outside [0, V))
    int hot[FW_MAX_N];                        // one-hot class of the node's row (-1: a window row)
    int pre[FW_MAX_N];                        // raw rows: 1 = post-processed up front (win.raw_probs)
    float cart[EW_MAX_PD];                    // cart_candidates_prob by cell
    double un[FW_UNI];                        // the step's uniforms from the cursor on
    double redd[2 * 16], redq[2 * 16];
    float redf[2 * 16];
    int redi[2 * 16];
    double wtot[16];
    int bonus[4];
    int dec[8];                               // wave 0's words: [0] neighbours under tau exist (zero them on a rejection), [1] verdict
    unsigned short nbid[FW_SLOTS][EW_PF_K];   // staged neighbour ids (table values; 0 beyond k + 1)
    unsigned short nbaddr[FW_SLOTS][EW_PF_K]; // the same as gather indices into g (positions >= k: the zero slot)
};

struct FwArgs {
    lantern_ep_params prm;
    lantern_ep_buffers buf;
    lantern_ep_window win;
    const int32_t *tables;
    int32_t n_nodes, n_internal, n_children, pad0;
};

// two f64 block sums behind one barrier (same reduction tree as block_sum_fast: DPP wave sum, one LDS slot per wave, DPP row combine)
__device__ __forceinline__ double fw_row_total(double x) {
    x += dpp_mov<0x111>(x);
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    const long long bits = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 15), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 15);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// NT x E4: the window is exactly the workgroup's register tile (W = 4 * NT * E4): 512 x 4 = the 8192-id Lumina / Anole image range,
// 256 x 1 = the 1024-id windows of the reduced-size reference vectors.  RAW (8192 only): rows arrive as raw cond / uncond bf16 logits.
template <int NT, int E4, bool RAW>
__global__ __launch_bounds__(NT) void epf_kernel(const FwArgs args) {
    constexpr int NW = NT / 64, W = 4 * NT * E4, SLOTS = (NW - 1 < FW_SLOTS) ? NW - 1 : FW_SLOTS;
    static_assert(!RAW || (NT == 512 && E4 == 4), "raw rows: the 8192-id window on 512 threads");
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    extern __shared__ float4 dyn_lds[];
    float *g = reinterpret_cast<float *>(dyn_lds);
    FwShared &S = *reinterpret_cast<FwShared *>(reinterpret_cast<char *>(g) + (size_t)(W + EW_G_EXT) * 4);
    int *const Shist = reinterpret_cast<int *>(reinterpret_cast<char *>(&S) + sizeof(FwShared));      // raw rows: radix-select histograms
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // the root's row first (the longest load of the round); raw rows: the probability form if there is one, replaced below
    // by the raw chunks when the root was not post-processed up front
    float4 rp[E4];
    bool rp_probs = true;
    if (probs) row_load<NT, E4, true>(probs, W, rp);
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
        float ct_[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * NT;
            ct_[u] = t < npd ? buf.cart_prob[(size_t)b * npd + t] : 1.0f;
        }
        double un_ = 2.0;
        if (tid < FW_UNI && ucur0 + tid < prm.n_uniforms) un_ = buf.uniforms[(size_t)b * prm.n_uniforms + ucur0 + tid];
        // ---- park them
        if (tid < FW_MAX_N) {
            S.nd_info[tid] = (ni.x & 255) | ((ni.y & 255) << 8);
            S.nd_par[tid] = ni.w;
            S.nd_dup[tid] = 0;
            if (ni.z < 0) {               // a leaf (internal nodes are written through their rank entry below)
                S.nd_kids[tid] = 0;
                S.nd_qrow[tid] = 0;
            }
            S.tok[tid] = tok_;
            S.hot[tid] = hot_;
            S.pre[tid] = pre_;
        }
        if (tid < n_int) {
            S.nd_kids[e03.x & (FW_MAX_N - 1)] = (e03.y & 0xffff) | (e03.z << 16);
            S.nd_qrow[e03.x & (FW_MAX_N - 1)] = e5;
        }
        if (tid < args.n_children) S.child[tid] = make_int2(chv.x, chv.y);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * NT;
            if (t < EW_MAX_PD) S.cart[t] = ct_[u];
        }
        if (tid < FW_UNI) S.un[tid] = un_;
        if (tid == 0) {
            g[W + EW_G_ZERO] = 0.0f;
            g[W + EW_G_HUGE] = 3.0e38f;
            g[W + EW_G_OUT] = 0.0f;
        }
    }
    __syncthreads();
    EPF_STAMP(1);
    // duplicate / missing sibling tokens (the node view does not hold below such a node): one child entry per thread
    if (tid < args.n_children) {
        const int2 ce = S.child[tid];
        const int nd = ce.x & (FW_MAX_N - 1), p = S.nd_par[nd] & (FW_MAX_N - 1);
        const int c0 = S.nd_kids[p] & 0xffff, mt = S.tok[nd];
        bool dup = mt == -1;
        for (int u = c0; u < tid; ++u) dup |= S.tok[S.child[u & (FW_MAX_N - 1)].x & (FW_MAX_N - 1)] == mt;
        if (dup) S.nd_dup[p] = 1;
    }

    // neighbour ids of the first children of node `pn`, requested by the worker waves (wave w: child w - 1; 2 x 16 bytes per lane)
    uint4 idq[2];
    auto ids_request = [&](int pn) {
        idq[0] = make_uint4(0u, 0u, 0u, 0u);
        idq[1] = make_uint4(0u, 0u, 0u, 0u);
        if (wave >= 1 && wave <= SLOTS) {
            const int kd = S.nd_kids[pn & (FW_MAX_N - 1)];
            const int ci = wave - 1;
            if (ci < (kd >> 16)) {
                const int cn = S.child[((kd & 0xffff) + ci) & (FW_MAX_N - 1)].x;
                const int x = S.tok[cn & (FW_MAX_N - 1)];
                const int trow = x - off;
                if (x >= prm.img_lo && x < prm.img_hi && trow >= 0 && trow < prm.table_rows) {
                    const uint16_t *row = buf.nn_table + (size_t)trow * prm.table_cols;
                    if (lane * 8 < nz) idq[0] = *reinterpret_cast<const uint4 *>(row + lane * 8);
                    if ((lane + 64) * 8 < nz) idq[1] = *reinterpret_cast<const uint4 *>(row + (lane + 64) * 8);
                }
            }
        }
    };
    // ... and parked: raw ids (for the zeroing, k + 1 of them) and gather indices (for the scan: positions >= k read the zero slot)
    auto ids_store = [&](int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t0 = (lane + 64 * h) * 8;
            uint32_t w[4] = {idq[h].x, idq[h].y, idq[h].z, idq[h].w}, ad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i0 = (t0 + 2 * q < nz) ? (w[q] & 0xffffu) : 0u, i1 = (t0 + 2 * q + 1 < nz) ? (w[q] >> 16) : 0u;
                w[q] = i0 | (i1 << 16);
                const uint32_t a0 = (t0 + 2 * q < k && i0 < (uint32_t)W) ? i0 : (uint32_t)(W + EW_G_ZERO);
                const uint32_t a1 = (t0 + 2 * q + 1 < k && i1 < (uint32_t)W) ? i1 : (uint32_t)(W + EW_G_ZERO);
                ad[q] = a0 | (a1 << 16);
            }
            *reinterpret_cast<uint4 *>(&S.nbid[slot][t0]) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4 *>(&S.nbaddr[slot][t0]) = make_uint4(ad[0], ad[1], ad[2], ad[3]);
        }
    };
    __syncthreads();          // the duplicate flags
    if constexpr (RAW) {
        if (!(probs && S.pre[0] != 0)) {      // the root was not post-processed up front: its raw chunks
            raw_row_load<NT>(raw_c, raw_u, rp);
            rp_probs = false;
        }
    }
    ids_request(0);
    EPF_STAMP(2);

    int node = 0, status = LANTERN_ST_OK;
    int n_tried = 0, n_rej = 0, n_used = 0, rej_here = 0;
    int out_tok = -1;
    float out_mass = 0.0f, gsum = 1.0f;
    bool lazy = false;

    // ------------------------------------------------------------------------------------------------ the walk
    for (;;) {
        // ---- arrival at `node`: its row is in rp (requested while its parent's verdict was computed), its children's ids in idq
        const int kd = S.nd_kids[node];
        const int coff = kd & 0xffff, nch_all = kd >> 16;
        const int nch = nch_all < FW_MAX_CH ? nch_all : FW_MAX_CH;
        if (nch_all > FW_MAX_CH) status = LANTERN_ST_TREE_LIMIT;
        int qrow = S.nd_qrow[node];
        qrow = qrow < 0 ? 0 : (qrow >= prm.R ? prm.R - 1 : qrow);
        const float *qsrc = qbase + (size_t)qrow * (size_t)win.orig_prob_stride;
        // lane t of every wave: child t (node, token, class, cart_candidates_prob)
        int node_l = 0, tok_l = -3 - lane, fl_l = 0;
        float qx_l = 1.0f;
        if (lane < nch) {
            const int2 ce = S.child[(coff + lane) & (FW_MAX_N - 1)];
            node_l = ce.x & (FW_MAX_N - 1);
            tok_l = S.tok[node_l];
            qx_l = S.cart[(ce.y >= 0 && ce.y < EW_MAX_PD) ? ce.y : 0];
            fl_l = (tok_l >= prm.img_lo && tok_l < prm.img_hi) ? 2 : 0;
            for (int q = 0; q < prm.n_syntax; ++q) fl_l |= (tok_l == prm.syntax[q]) ? 1 : 0;
        }
        // the drafter row of this node's children (speculative: unused when the first child is accepted) and q[child tokens]
        float4 qraw[E4];
        float qv_l = 0.0f;
        if (nch > 0) {
            row_load<NT, E4, true>(qsrc, W, qraw);
            if (lane < nch && tok_l >= lo && tok_l < lo + W) qv_l = qsrc[tok_l - lo];
        } else {
#pragma unroll
            for (int it = 0; it < E4; ++it) qraw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- the node's row -> g
        const int hot = S.hot[node];
        out_tok = -1;
        out_mass = 0.0f;
        gsum = 1.0f;
        lazy = false;
        rej_here = 0;
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
            if (wave >= 1 && wave <= SLOTS && wave - 1 < nch) ids_store(wave - 1);
            __syncthreads();
        } else if (!RAW || rp_probs) {
#pragma unroll
            for (int it = 0; it < E4; ++it) reinterpret_cast<float4 *>(g)[tid + it * NT] = rp[it];
            if (wave >= 1 && wave <= SLOTS && wave - 1 < nch) ids_store(wave - 1);
            __syncthreads();
        } else if constexpr (RAW) {
            auto hook = [&]() {
                if (wave >= 1 && wave <= SLOTS && wave - 1 < nch) ids_store(wave - 1);
            };
            raw_row_to_lds<NT>(rp, -1, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, Shist, ph, hook);
        }
        EPF_STAMP(10);
        if (nch > 0 && S.nd_dup[node]) status = LANTERN_ST_NEEDS_CHAIN;

        // per-thread view of the earlier siblings' entries of the drafter row (q[siblings] = 0; q /= q.sum(), :696-700)
        unsigned zmask = 0;
        double rem = 0.0, sq = 0.0;
        bool sq_ready = false;
        int acc_t = -1;
        int staged_lo = 0;               // children [staged_lo, staged_lo + SLOTS) have their ids in the slots

        for (int t = 0; t < nch && status == LANTERN_ST_OK; ++t) {
            if (t > 0) {                 // child t - 1 is now an earlier sibling
                const int xp = rdlane(tok_l, t - 1) - lo;
                if (xp >= 0 && xp < W && ((xp >> 2) % NT) == tid) zmask |= 1u << ((((xp >> 2) / NT) << 2) | (xp & 3));
                rem += (double)rdlane(qv_l, t - 1);
            }
            if (t > 0 && !sq_ready) {     // (child 0 was skipped, not rejected: S_q has not been reduced yet)
                double sl = 0.0;
#pragma unroll
                for (int it = 0; it < E4; ++it) sl += (double)qraw[it].x + (double)qraw[it].y + (double)qraw[it].z + (double)qraw[it].w;
                sq = block_sum_fast<double, NW>(sl, S.redq, ph);
                sq_ready = true;
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
            const double rr = S.un[n_used++];
            ++n_tried;
            const int cnode = rdlane(node_l, t);
            const int fl = rdlane(fl_l, t);
            const bool in_img = (fl & 2) != 0, is_syn = prm.syntax_shortcut && (fl & 1) != 0;
            const float qx = rdlane(qx_l, t);
            EPF_STAMP(20);
            if (t >= staged_lo + SLOTS) {          // more children than staging slots (rare): the next SLOTS of them, synchronously
                __syncthreads();
                idq[0] = make_uint4(0u, 0u, 0u, 0u);
                idq[1] = make_uint4(0u, 0u, 0u, 0u);
                if (wave >= 1 && wave <= SLOTS && t + wave - 1 < nch) {
                    const int xs = rdlane(tok_l, (t + wave - 1) & 63), trow = xs - off;
                    if (xs >= prm.img_lo && xs < prm.img_hi && trow >= 0 && trow < prm.table_rows) {
                        const uint16_t *row = buf.nn_table + (size_t)trow * prm.table_cols;
                        if (lane * 8 < nz) idq[0] = *reinterpret_cast<const uint4 *>(row + lane * 8);
                        if ((lane + 64) * 8 < nz) idq[1] = *reinterpret_cast<const uint4 *>(row + (lane + 64) * 8);
                    }
                    ids_store(wave - 1);
                }
                staged_lo = t;
                __syncthreads();
            }
            const int slot = t - staged_lo;
            // ---- speculative requests for "this candidate is accepted": its row, and its first children's neighbour ids
            const int chot = S.hot[cnode];
            bool crp_probs = true;
            if (chot < 0) {
                if constexpr (RAW) {
                    crp_probs = probs && S.pre[cnode] != 0;
                    if (crp_probs) row_load<NT, E4, true>(probs + (size_t)cnode * W, W, rp);
                    else raw_row_load<NT>(raw_c + (size_t)cnode * V, raw_u + (size_t)cnode * V, rp);
                } else {
                    row_load<NT, E4, true>(probs + (size_t)cnode * W, W, rp);
                }
            }
            ids_request(cnode);

            int code = 0;
            const bool scan = !(qx <= 0.0f) && !is_syn && in_img;
            if (qx <= 0.0f) {
                code = 0;                                   // skipped (:680-682): the draw is spent, nothing else happens
            } else if (is_syn) {
                code = ((float)rr <= 1.0f / qx) ? 1 : 2;    // px = 1 (:654-656)
            } else if (!in_img) {
                code = ((float)rr <= 0.0f / qx) ? 1 : 2;    // px = 0 (:657-659)
            }
            int m0 = 0;
            float4 gn[E4];
            double loc = 0.0;
            const int trow = x - off;
            if (scan && !(trow >= 0 && trow < prm.table_rows)) {
                status = LANTERN_ST_TABLE_OOB;
                break;
            }
            const FastDiv dgc(gsum), dq(t > 0 ? (float)(sq - rem) : 1.0f);
            // max(gtp - q, 0) of this thread's entries, into registers (q: earlier siblings zeroed, renormalised)
            auto residual_pass = [&]() {
                float4 q[E4];
#pragma unroll
                for (int it = 0; it < E4; ++it) q[it] = qraw[it];
                if (t > 0) {
                    if (zmask) {
#pragma unroll
                        for (int it = 0; it < E4; ++it) {
                            const unsigned z = zmask >> (4 * it);
                            q[it].x = (z & 1u) ? 0.f : q[it].x; q[it].y = (z & 2u) ? 0.f : q[it].y;
                            q[it].z = (z & 4u) ? 0.f : q[it].z; q[it].w = (z & 8u) ? 0.f : q[it].w;
                        }
                    }
#pragma unroll
                    for (int it = 0; it < E4; ++it) q[it] = dq(q[it]);
                }
                loc = 0.0;
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const float4 qv = q[it];
                    float4 gv = reinterpret_cast<const float4 *>(g)[tid + it * NT];
                    if (lazy) gv = dgc(gv);
                    float d;
                    d = gv.x - qv.x; gv.x = d < 0.0f ? 0.0f : d;
                    d = gv.y - qv.y; gv.y = d < 0.0f ? 0.0f : d;
                    d = gv.z - qv.z; gv.z = d < 0.0f ? 0.0f : d;
                    d = gv.w - qv.w; gv.w = d < 0.0f ? 0.0f : d;
                    gn[it] = gv;
                    loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                }
            };
            if (scan) {
                if (wave == 0) {
                    // ---------------- wave 0: the k-neighbour cumulative mass (:661-677), 16 consecutive neighbours per lane
                    __builtin_amdgcn_s_setprio(3);
                    float px = g[x - lo];
                    const uint4 a = *reinterpret_cast<const uint4 *>(&S.nbaddr[slot][lane * 16]);
                    const uint4 bq = *reinterpret_cast<const uint4 *>(&S.nbaddr[slot][lane * 16 + 8]);
                    const uint32_t w[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
                    float f[16];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        f[2 * c] = g[w[c] & 0xffffu];
                        f[2 * c + 1] = g[w[c] >> 16];
                    }
                    if (lazy) {             // the window holds an unnormalised residual (the zero slot: 0 / gsum = 0)
                        px = dgc(px);
#pragma unroll
                        for (int c = 0; c < 16; ++c) f[c] = dgc(f[c]);
                    }
                    const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                    // neighbours under tau exist iff the first one is (the cumulative mass does not decrease)
                    m0 = (rdlane(f[0], 0) <= tau) ? 1 : 0;
                    if (lane == 0) S.dec[0] = m0;
                    __syncthreads();                                    // B1: g has been read; the workers may zero the neighbours
                    double v[16], l = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        l += (double)f[c];
                        v[c] = l;
                    }
                    const double excl = wave_scan_incl_dpp(dpp_mov<0x138>(l));      // exclusive: scan of the lane totals shifted up one lane
                    __syncthreads();                                    // B2: (the workers: zeroing done, residual pass next)
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float cs = (float)(excl + v[c]);
                        mx = (cs <= tau) ? cs : mx;                     // non-decreasing: the last one under tau is the largest
                    }
                    const float best_cs = wave_max(mx);
                    if (best_cs > -__builtin_inff()) px = px + best_cs;
                    code = ((float)rr <= px / qx) ? 1 : 2;
                    if (lane == 0) S.dec[1] = code;
                    __builtin_amdgcn_s_setprio(0);
                    __syncthreads();                                    // B3: the verdict
                } else {
                    // ---------------- workers: the residual this candidate's rejection would need, ahead of the verdict
                    __syncthreads();                                    // B1
                    m0 = S.dec[0];
                    if (m0) {                                           // gtp[neighbours] = 0, k + 1 of them (:702-704)
                        for (int p = tid - 64; p < nz; p += NT - 64) {
                            const int id = (int)S.nbid[slot][p];
                            if (id < W) g[id] = 0.0f;
                        }
                    }
                    __syncthreads();                                    // B2
                    residual_pass();
                    __syncthreads();                                    // B3
                    code = S.dec[1];
                }
                m0 = S.dec[0];
            }
            EPF_STAMP(21);
            if (code == 0) continue;
            if (code == 1) {
                acc_t = t;
                node = cnode;
                rp_probs = crp_probs;
                break;
            }
            // ------------------------------------------------ rejection (:690-713)
            ++n_rej;
            ++rej_here;
            if (is_syn) {
                status = LANTERN_ST_SYNTAX_REJECT;
                break;
            }
            if (!scan) {            // a non-image candidate (px = 0): no neighbours to zero, the pass was not run ahead
                residual_pass();
            } else if (wave == 0) {
                residual_pass();    // wave 0's own share (the neighbours are already zeroed)
            }
            // the unnormalised residual goes back to LDS (every thread owns its entries); the reduction's barrier publishes it
#pragma unroll
            for (int it = 0; it < E4; ++it) reinterpret_cast<float4 *>(g)[tid + it * NT] = gn[it];
            double tot;
            if (!sq_ready) {        // S_q with the same barrier: a later child's q.sum() is S_q minus its earlier siblings' entries
                double sl = 0.0;
#pragma unroll
                for (int it = 0; it < E4; ++it) sl += (double)qraw[it].x + (double)qraw[it].y + (double)qraw[it].z + (double)qraw[it].w;
                loc = wave_sum(loc);
                sl = wave_sum(sl);
                double *b0 = S.redd + (ph & 1) * NW, *b1 = S.redq + (ph & 1) * NW;
                ph ^= 1;
                if (lane == 0) {
                    b0[wave] = loc;
                    b1[wave] = sl;
                }
                __syncthreads();
                tot = fw_row_total(lane < NW ? b0[lane] : 0.0);
                sq = fw_row_total(lane < NW ? b1[lane] : 0.0);
                sq_ready = true;
            } else {
                tot = block_sum_fast<double, NW>(loc, S.redd, ph);
            }
            tot += (double)out_mass;
            const float gs = (float)tot;
            if (gs == 0.0f) {
                status = LANTERN_ST_NEEDS_DENSE;      // `gtp.sum()==0 -> ones`: uniform over all V, only the dense kernel holds it
                break;
            }
            gsum = gs;
            lazy = true;
            out_mass = out_mass / gs;
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
    FwArgs args{p, *buf, *win, nodes->tables, nodes->n_nodes, nodes->n_internal, nodes->n_children, 0};
    const size_t lds = (size_t)(win->win_len + EW_G_EXT) * 4 + sizeof(FwShared) + (raw ? (size_t)O7_HIST_INTS * 4 : 0);
    hipStream_t st = (hipStream_t)stream;
    if (raw) LANTERN_LAUNCH((epf_kernel<512, 4, true>), dim3(p.B), dim3(512), lds, st, args);
    else if (win->win_len == 8192) LANTERN_LAUNCH((epf_kernel<512, 4, false>), dim3(p.B), dim3(512), lds, st, args);
    else LANTERN_LAUNCH((epf_kernel<256, 1, false>), dim3(p.B), dim3(256), lds, st, args);
    LANTERN_CHECK_LAUNCH("evaluate_posterior_nodes (fast walk)");
    return LANTERN_OK;
}
