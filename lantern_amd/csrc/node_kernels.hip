// node_kernels.hip -- O8, node-parallel form (include/lantern_hip.h: lantern_evaluate_posterior_nodes).
//
// The chain kernel (window_kernels.hip: epw_kernel) gives every sequence one workgroup that walks the tree level by level:
// B sequences occupy B compute units and the launch lasts as long as the unluckiest sequence's whole walk.  Here every
// INTERNAL NODE of every sequence gets its own workgroup (B * n_internal of them: the whole GPU), which runs exactly the
// part of the reference's loop that would execute if the walk arrived at that node -- try its children in path order
// against the node's row, relaxed by the k-neighbour cumulative mass, residual update after every rejection, bonus token
// from the residual when all are rejected (ea_model_lumina_mgpt.py:628-713, :781) -- with the uniforms the walk would
// hand it (one draw per tried candidate, root..node: a static offset per node).  epn_walk_kernel then follows
// root -> accepted child -> ... through the per-node records and writes the outputs of the chain kernel.  Same arithmetic
// as epw_kernel, operation for operation (shared helpers: window_dev.h), so both produce the same bits.
#include "common.h"
#include "window_dev.h"

namespace lantern {

constexpr int EN_MAX_CH = 32;     // children per node (one lane each)
constexpr int EN_SLOTS = 4;       // neighbour-id staging slots (children whose table rows are resident in LDS)
constexpr int EN_REC = 8;         // ints per node record: {accepted child or -1, tried, rejected, status, bonus token, out_tok, out_mass bits, flags}
constexpr int EN_INFO = 16;       // ints per internal-node entry of the tables
constexpr int EN_TR = 64;
constexpr int EN_MAX_N = 128;     // nodes per tree

struct alignas(16) EnShared {
    double redd[2 * 16];
    double wtot[16];                                // per-wave totals (neighbour scan, bonus draw)
    float wmax[16];
    int redi[2 * 16];
    int bonus[4];
    unsigned short nbid[EN_SLOTS][EW_PF_K];         // raw neighbour ids of the staged children (table values)
};

// dynamic LDS: [ g : W + 4 f32 | W bits (LlamaGen / Anole static: the neighbour set zeroes q) | EnShared ]
__host__ __device__ inline size_t epn_shared_offset(int W) { return epw_shared_offset(W); }

// Static per-node facts, copied into the kernel-argument segment by the host (2.1 KB): a workgroup learns which node it is,
// where that node's row / drafter row / children are, with no load round in front of the long ones.
struct EpnStatic {
    // per launch rank, four dwords (one s_load_dwordx4): x = node | nch << 8 | depth << 16 | uoff << 24;
    // y = qrow | child0 << 8 | child1 << 16 | child2 << 24; z = child3 | path0 << 8 | path1 << 16 | path2 << 24; w = path3
    // (child t: node id of the t-th child; path t: first path through it -- its cell is path * D + depth + 1)
    uint4 w[EN_MAX_N];
};

struct EpnArgs {
    lantern_ep_params prm;
    lantern_ep_buffers buf;
    lantern_ep_window win;
    const int32_t *tables;
    int32_t *records;      // [B, N, EN_REC]
    float *dist;           // [B, n_internal, W] or NULL
    int32_t n_internal, n_nodes, n_children, leaf_wgs;
    unsigned long long *trace;   // diagnosis (lantern_debug_epn_trace): [grid][EN_TR] phase stamps (id << 56 | cycles), or NULL
    EpnStatic st;
};

#define EPN_STAMP(id)                                                                                                  \
    do {                                                                                                               \
        if (args.trace && threadIdx.x == 0 && tr_n < EN_TR - 1)                                                        \
            args.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EN_TR + 1 + tr_n++] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
    } while (0)

// IDMODE 1: table rows at any 2-byte alignment (the reference's [K, K-1] layout); 2: packed table (lantern_pack_vq_table).
// WIDE: some node of the tree has more than four children (their tokens then come from the child list with two more
// dependent load rounds in the prologue; compiled out otherwise so that the common prologue keeps one wait per round).
// What a node's routine hands back (uniform across the workgroup).
struct EpnVerdict {
    int acc;                          // accepted child (node id), or -1
    int n_tried, n_rej, status, token, out_tok, flags;
    float out_mass;
};

// The routine of ONE node: try its children in order against the node's row (see the header).  `sw`: the node's packed header
// (EpnStatic), `ri`: its rank among the internal nodes (0 for a leaf), `uoff`: uniforms the walk consumed before this node.
// SPEC: compile-time instance of the caller's configuration, as in epw_body (window_kernels.hip): 0 = everything from the argument block,
// 1 = Lumina static tree + LANTERN + syntax shortcut + the Chameleon vocabulary constants, 4 = Anole static tree (neighbours zeroed in q).
template <int NT, int E4, int IDMODE, bool FULLW, bool WIDE, int SPEC = 0>
__device__ __forceinline__ void epn_node(const EpnArgs &args, const int b, const uint4 sw, const bool internal, const int ri, const int uoff,
                                         int &ph, int &tr_n, EpnVerdict &vd) {
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    constexpr int NW = NT / 64;
    constexpr int PP = EW_PF_K / NT;                  // neighbour positions per thread in the scan (2 or 1)
    static_assert(PP == 1 || PP == 2, "512 or 1024 threads");
    extern __shared__ float4 dyn_lds[];
    float *g = reinterpret_cast<float *>(dyn_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool SL = SPEC != 0;
    const int W = SL ? 8192 : win.win_len, lo = SL ? 4 : win.win_lo, V = SL ? 65536 : prm.V;
    uint32_t *nbmask = reinterpret_cast<uint32_t *>(g + W + EW_G_EXT);
    EnShared &S = *reinterpret_cast<EnShared *>(reinterpret_cast<char *>(g) + epn_shared_offset(W));
    const int k = prm.k, off = SL ? 4 : prm.tok_offset;
    const int p_mode = !SL ? prm.mode : (SPEC == 4 ? (int)LANTERN_MODE_STATIC_LG : (int)LANTERN_MODE_STATIC_LUMINA);
    const bool p_lantern = SL ? true : prm.lantern != 0;
    const bool p_syntax = SL ? SPEC == 1 : prm.syntax_shortcut != 0;
    const int p_nsyn = SL ? (SPEC == 1 ? 4 : 0) : prm.n_syntax;
    const int p_img_lo = SL ? 4 : prm.img_lo, p_img_hi = SL ? 8196 : prm.img_hi, p_trows = SL ? 8192 : prm.table_rows;
    auto p_syn = [&](int q) -> int { return SL ? (q == 0 ? 8196 : (q == 1 ? 8197 : (q == 2 ? 8803 : 8828))) : prm.syntax[q]; };
    const bool is_static = SL ? true : p_mode != LANTERN_MODE_DYNAMIC;
    const int nz = (k + 1 < prm.table_cols) ? k + 1 : prm.table_cols;
    const int32_t *tb = args.tables;
    const int n_int = args.n_internal;
    EPN_STAMP(0);

    // ---- node header: from the kernel-argument segment (no load round in front of the row loads)
    const int node = sw.x & 255, qrow = sw.y & 255, depth = (sw.x >> 16) & 255;
    const int nch_all = (internal && depth + 1 < prm.D) ? (int)((sw.x >> 8) & 255) : 0;      // a node on the last level has no level below it
    const int nch = nch_all < EN_MAX_CH ? nch_all : EN_MAX_CH;
    int status = nch_all > EN_MAX_CH ? LANTERN_ST_TREE_LIMIT : LANTERN_ST_OK;
    const int npd = prm.P * prm.D;

    // ---- first round of loads.  The small ones go FIRST (vector loads return in issue order: behind the rows they would
    // arrive with the rows): lane t < 4 of every wave fetches child t's token and cart_candidates_prob (children 0..3 sit in
    // the node's kernel-argument word), every lane the cursor, the row's one-hot flag and the bonus uniform.  Every load is
    // unconditional, from an always-valid address, the choice made afterwards.
    const int32_t *safe = args.tables;                            // >= 8 readable ints
    const uint32_t cw = (sw.y >> 8) | (sw.z << 24), pw = (sw.z >> 8) | (sw.w << 24);       // child node ids / first paths, one byte each
    const int l4 = lane & 3;
    const int cn_v = (int)((cw >> (8 * l4)) & 255u), cc_v = (int)((pw >> (8 * l4)) & 255u) * prm.D + depth + 1;
    const int cn_c = (cn_v < prm.N) ? cn_v : 0, cc_c = (cc_v < npd) ? cc_v : 0;
    const int64_t *p_tok = is_static ? buf.tree_cand + (size_t)b * prm.N + cn_c : buf.cand + (size_t)b * npd + cc_c;
    const float *p_qx = is_static ? buf.cart_prob + (size_t)b * npd + cc_c : reinterpret_cast<const float *>(safe);
    // (addresses made lane-dependent on purpose -- + 0 * lane -- so that these stay VECTOR loads, queued in front of the rows;
    // as scalar loads they would share the wait counter of the kernel-argument fetches and stall the row issue)
    const int32_t *p_cur = (buf.cursor ? buf.cursor + b : safe);
    const int32_t *p_hot = (win.row_hot ? win.row_hot + (size_t)b * prm.rows_per_seq + node : safe);
    const double *p_ub = (win.u_bonus ? win.u_bonus + b : reinterpret_cast<const double *>(safe));
    int64_t tok_ld = -1;
    float qx_ld = 1.0f;
    tok_ld = *p_tok;
    qx_ld = *p_qx;
    const int ucur_ld = p_cur[l4 >> 2], hot_ld = p_hot[l4 >> 2];
    const double ub_ld = p_ub[l4 >> 2];
    __builtin_amdgcn_sched_barrier(0);          // keep the issue order: the scheduler must not sink these behind the rows

    // ---- the node's row and (static trees) the drafter row.  The drafter row is speculative: unused when the first child is
    // accepted.  Both groups of loads are unconditional (a node without a drafter row reads its own row twice): a load under a
    // branch makes every later wait count conservative, i.e. the small loads above would wait for the rows.
    const float *rowp = buf.logits + ((size_t)b * prm.rows_per_seq + node) * W;
    float4 rp[E4];
    row_load<NT, E4, FULLW>(rowp, W, rp);
    float4 qraw[E4];
    const float *qsrc = nullptr;
    {
        const int qr = qrow < 0 ? 0 : (qrow >= prm.R ? prm.R - 1 : qrow);
        const bool has_q = is_static && nch > 0;
        const float *qp = has_q ? buf.orig_prob + ((size_t)b * prm.R + qr) * (size_t)win.orig_prob_stride + win.orig_prob_offset
                                : buf.logits + (size_t)b * prm.rows_per_seq * W;      // (any readable row: the values are never used)
        row_load<NT, E4, FULLW>(qp, W, qraw);
        if constexpr (!FULLW) {
#pragma unroll
            for (int it = 0; it < E4; ++it)
                if ((tid + it * NT) * 4 >= W) qraw[it] = make_float4(0.f, 0.f, 0.f, 0.f);      // beyond the window: no drafter mass
        }
        qsrc = has_q ? qp : nullptr;
    }
    // ---- children: lane t of every wave holds child t (token, flags, cart_candidates_prob, drafter probability of its token, its uniform)
    int tok_l = -1, node_l = 0;
    float qx_l = 1.0f;
    if (nch > 0) {
        int64_t tl = -1;
        if (WIDE && nch > 4) {          // the rest of the child list, with two dependent vector rounds
            const int32_t *child = tb + 8 + EN_INFO * n_int + 4 * ldc(tb + 8 + EN_INFO * ri + 1);
            if (lane >= 4 && lane < nch) {
                const int4 ci = *reinterpret_cast<const int4 *>(child + 4 * lane);
                const int cn_l = (ci.x >= 0 && ci.x < prm.N) ? ci.x : 0, cc_l = (ci.y >= 0 && ci.y < npd) ? ci.y : 0;
                node_l = ci.x;
                tl = is_static ? buf.tree_cand[(size_t)b * prm.N + cn_l] : buf.cand[(size_t)b * npd + cc_l];
                if (is_static) qx_l = buf.cart_prob[(size_t)b * npd + cc_l];
            }
        }
        if (lane < 4) {
            tl = tok_ld;
            qx_l = is_static ? qx_ld : 1.0f;
            node_l = cn_v;
        }
        tok_l = (tl < -1 || tl >= V) ? -2 : (int)tl;      // -2: outside [0,V) (TOKEN_OOB when tried)
        if (lane >= nch) tok_l = -3 - lane;               // no such child
    }
    const int ucur0 = buf.cursor ? __builtin_amdgcn_readfirstlane(ucur_ld) : 0;
    const int hot = !win.row_hot ? -1 : __builtin_amdgcn_readfirstlane(hot_ld);
    const double ub = win.u_bonus ? ub_ld : 0.0;
    // ---- second round of vector loads, queued behind the rows: uniforms, q[child tokens], the first children's table rows
    double un_l = 2.0;
    if (lane < nch && ucur0 + uoff + lane < prm.n_uniforms)
        un_l = buf.uniforms[(size_t)b * prm.n_uniforms + ucur0 + uoff + lane];
    float qv_l = 0.0f;                                        // q[token of child `lane`]: what the later siblings' q.sum() loses
    if (qsrc && lane < nch && tok_l >= lo && tok_l < lo + W) qv_l = qsrc[tok_l - lo];
    // neighbour ids -> LDS as raw table values (addresses are resolved in the scan).  IDMODE 2: 128 threads x 16 bytes cover one
    // child, the workgroup stages min(NT/128, EN_SLOTS) children per round; IDMODE 1: 2-byte loads, one child per round.
    constexpr int GROUPS = (NT / 128 < EN_SLOTS) ? NT / 128 : EN_SLOTS;
    constexpr int ROUND = (IDMODE == 2) ? GROUPS : 1;
    constexpr int IDS_PER = (IDMODE == 2) ? 1 : (EW_PF_K + NT - 1) / NT;
    struct StageRegs {
        uint4 q;
        unsigned short h[IDS_PER];
    };
    auto lookup_row = [&](int x, int &trow) -> bool {
        trow = x - off;
        return p_lantern && x >= 0 && trow >= 0 && trow < p_trows && !(p_syntax && !(x >= p_img_lo && x < p_img_hi));
    };
    auto stage_load = [&](int t_first) -> StageRegs {          // issue the loads ...
        StageRegs sr;
        sr.q = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (IDMODE == 2) {
            const int gidx = tid >> 7, chn = tid & 127, t = t_first + gidx, t0 = chn * 8;      // gidx is wave-uniform (two waves per child)
            const int x = rdlane(tok_l, (t < EN_MAX_CH ? t : 0));
            int trow = 0;
            const bool lookup = gidx < GROUPS && t < nch && lookup_row(x, trow);
            if (lookup && t0 < nz) sr.q = *reinterpret_cast<const uint4 *>(buf.nn_table + (size_t)trow * prm.table_cols + t0);
        } else {
            const int t = t_first;
            const int x = rdlane(tok_l, (t < EN_MAX_CH ? t : 0));
            int trow = 0;
            const bool lookup = t < nch && lookup_row(x, trow);
            const uint16_t *nbp = buf.nn_table + (size_t)(lookup ? trow : 0) * prm.table_cols;
#pragma unroll
            for (int u = 0; u < IDS_PER; ++u) {
                const int p = tid + u * NT;
                sr.h[u] = (lookup && p < nz) ? nbp[p] : (unsigned short)0;
            }
        }
        return sr;
    };
    auto stage_store = [&](int t_first, const StageRegs &sr) { // ... and park them once they are needed
        if constexpr (IDMODE == 2) {
            const int gidx = tid >> 7, chn = tid & 127, t = t_first + gidx;
            if (t < nch && gidx < GROUPS) *reinterpret_cast<uint4 *>(&S.nbid[t % EN_SLOTS][chn * 8]) = sr.q;
        } else {
            const int t = t_first;
            if (t < nch) {
#pragma unroll
                for (int u = 0; u < IDS_PER; ++u) {
                    const int p = tid + u * NT;
                    if (p < EW_PF_K) S.nbid[t % EN_SLOTS][p] = sr.h[u];
                }
            }
        }
    };
    int staged = 0;                                   // children [0, staged) have had their ids staged (slot reuse: t % EN_SLOTS)
    StageRegs sr0;
    sr0.q = make_uint4(0u, 0u, 0u, 0u);
    if (p_lantern && nch > 0) {
        sr0 = stage_load(0);
        staged = ROUND;
    }
    int fl_l = (tok_l >= p_img_lo && tok_l < p_img_hi) ? 2 : 0;
    if (p_syntax)
        for (int q = 0; q < p_nsyn; ++q) fl_l |= (tok_l == p_syn(q)) ? 1 : 0;
    EPN_STAMP(1);
    // ---- the row -> LDS (probabilities; one-hot rows carry their mass in (out_tok, out_mass) when the token lies outside the window)
    int out_tok = -1;
    float out_mass = 0.0f;
    if (hot >= 0) {
        const bool inside = hot >= lo && hot < lo + W;
        for (int i4 = tid; i4 * 4 < W; i4 += NT) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e = lo + i4 * 4;
            if (hot >= e && hot < e + 4) set_comp(v, hot - e, 1.0f);
            reinterpret_cast<float4 *>(g)[i4] = v;
        }
        if (!inside) {
            out_tok = hot;
            out_mass = 1.0f;
        }
    } else {
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int i4 = tid + it * NT;
            if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(g)[i4] = rp[it];
        }
    }
    if (tid == 0) {
        g[W + EW_G_ZERO] = 0.0f;
        g[W + EW_G_HUGE] = 3.0e38f;
        g[W + EW_G_OUT] = out_mass;
    }
    if (p_lantern && nch > 0) stage_store(0, sr0);
    EPN_STAMP(2);
    // S_q: the drafter row's sum (f64); a later child's q.sum() is S_q minus its earlier siblings' entries -- no
    // workgroup reduction on the rejection path
    double sq = 0.0;
    if (is_static && nch > 1) {
#pragma unroll
        for (int it = 0; it < E4; ++it) sq += (double)qraw[it].x + (double)qraw[it].y + (double)qraw[it].z + (double)qraw[it].w;
        sq = block_sum_fast<double, NW>(sq, S.redd, ph);      // carries the barrier
    } else {
        __syncthreads();
    }
    EPN_STAMP(3);

    // duplicate / missing sibling tokens: the node view does not hold (see the header); reported, never guessed
    int flags = 0;
    {
        bool dup = lane < nch && tok_l == -1;
        for (int u = 0; u < nch; ++u) {
            const int x = rdlane(tok_l, u);
            dup |= (lane > u && lane < nch && tok_l == x);
        }
        if (__ballot(dup) != 0ull) flags |= 1;
    }
    EPN_STAMP(4);

    int acc = -1, n_tried = 0, n_rej = 0;
    unsigned zmask = 0;        // which of this thread's 16 entries of the drafter row belong to earlier siblings (bit it*4 + c)
    double rem = 0.0;          // their drafter probabilities, summed
    // Lazy normalisation: after a rejection g keeps the UNNORMALISED residual and `gsum` its sum; every reader divides on the
    // fly with the same division the reference's `gtp /= gtp.sum()` performs per entry -- the W-wide normalise pass, its LDS
    // write and its barrier leave the chain (the residual is written while the sum is still being reduced).
    float gsum = 1.0f;
    bool lazy = false;
    for (int t = 0; t < nch && status == LANTERN_ST_OK && !(flags & 1); ++t) {
        if (t > 0) {           // child t-1 is now an earlier sibling
            const int xp = rdlane(tok_l, t - 1) - lo;
            if (xp >= 0 && xp < W && ((xp >> 2) % NT) == tid) zmask |= 1u << ((((xp >> 2) / NT) << 2) | (xp & 3));
            rem += (double)rdlane(qv_l, t - 1);
        }
        const int x = rdlane(tok_l, t);
        if (x == -2) {
            status = LANTERN_ST_TOKEN_OOB;
            break;
        }
        if (ucur0 + uoff + t >= prm.n_uniforms) {
            status = LANTERN_ST_UNIFORMS;
            break;
        }
        if (p_lantern && t >= staged) {            // more children than staging slots: the next round reuses the slots of finished children
            __syncthreads();
            stage_store(t, stage_load(t));
            staged = t + ROUND;
            __syncthreads();
        }
        const double rr = rdlane(un_l, t);
        ++n_tried;
        EPN_STAMP(10);
        const int fl = rdlane(fl_l, t);
        const bool in_img = (fl & 2) != 0, is_syn = (fl & 1) != 0;
        const bool x_in = (x >= lo && x < lo + W);
        const int slot = t % EN_SLOTS;
        const int trow = x - off;
        const bool has_nb = p_lantern && trow >= 0 && trow < p_trows;
        // ---------------- the k-neighbour cumulative mass, all waves: PP neighbours per thread, f64 DPP scan per wave, wave totals through
        // LDS; every thread then holds the decision (no broadcast round)
        const FastDiv dgc(gsum);
        float px = x_in ? g[x - lo] : (x == out_tok ? out_mass : 0.0f);
        if (lazy && x_in) px = dgc(px);
        int code = 0, m = 0;
        if (p_syntax && is_syn) {
            px = 1.0f;
        } else if (p_syntax && !in_img) {
            px = 0.0f;
        } else if (p_lantern) {
            if (!has_nb) {
                code = 3;
            } else {
                const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                // position t of the list -> a slot of g: window index, or a sentinel (0 outside the window, out_mass for the one-hot
                // token outside it, 3e38 at positions >= k so that they can never pass `<= tau`)
                auto gaddr = [&](int id, int pos) -> int {
                    const int e = id + off - lo;
                    int a = (e >= 0 && e < W) ? e : W + ((id + off == out_tok) ? EW_G_OUT : EW_G_ZERO);
                    return pos >= k ? W + EW_G_HUGE : a;
                };
                double v0, v1;
                if constexpr (PP == 2) {
                    const uint32_t a = *reinterpret_cast<const uint32_t *>(&S.nbid[slot][tid * 2]);
                    const int a0 = gaddr((int)(a & 0xffffu), tid * 2), a1 = gaddr((int)(a >> 16), tid * 2 + 1);
                    float f0 = g[a0], f1 = g[a1];
                    if (lazy) {                       // window entries are unnormalised; the sentinel slots hold final values
                        f0 = a0 < W ? dgc(f0) : f0;
                        f1 = a1 < W ? dgc(f1) : f1;
                    }
                    v0 = (double)f0;
                    v1 = v0 + (double)f1;
                } else {
                    const int a0 = gaddr((int)S.nbid[slot][tid], tid);
                    float f0 = g[a0];
                    if (lazy) f0 = a0 < W ? dgc(f0) : f0;
                    v0 = (double)f0;
                    v1 = v0;
                }
                const double inc = wave_scan_incl_dpp(v1);
                const double excl = dpp_mov<0x138>(inc);          // the lanes below (no subtraction: a 3e38 sentinel must not cancel a true prefix)
                if (lane == 63) S.wtot[wave] = inc;
                __syncthreads();
                double pre = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) pre += (w < wave) ? S.wtot[w] : 0.0;
                const double base = pre + excl;
                const float cs0 = (float)(base + v0), cs1 = (float)(base + v1);
                float mx = (cs1 <= tau) ? cs1 : ((cs0 <= tau) ? cs0 : -__builtin_inff());    // non-decreasing: the last one under tau is the largest
                mx = wave_max(mx);
                if (lane == 0) S.wmax[wave] = mx;
                __syncthreads();
                float best_cs = S.wmax[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) best_cs = fmaxf(best_cs, S.wmax[w]);
                if (best_cs > -__builtin_inff()) {
                    m = 1;
                    px = px + best_cs;
                }
            }
        }
        if (code == 0) {
            float qx = 1.0f;
            bool skip = false;
            if (is_static) {
                qx = rdlane(qx_l, t);
                skip = qx <= 0.0f;
            }
            if (!skip) code = ((float)rr <= px / qx) ? 1 : 2;
        }
        EPN_STAMP(11);
        if (code == 3) {
            status = LANTERN_ST_TABLE_OOB;
            break;
        }
        if (code == 0) continue;
        if (code == 1) {
            acc = rdlane(node_l, t);
            break;
        }
        // ------------------------------------------------ rejection: residual, all waves
        ++n_rej;
        if (p_syntax && is_syn) {
            status = LANTERN_ST_SYNTAX_REJECT;
            break;
        }
        const bool zero_nb = p_lantern && m > 0 && (!p_syntax || in_img);
        const bool lg_nb = zero_nb && p_mode == LANTERN_MODE_STATIC_LG;          // LlamaGen / Anole static: the neighbours are zeroed in q
        double loc = 0.0;
        float4 gn[E4];
        bool need_bar = false;
        if (zero_nb && !lg_nb) {              // gtp[neighbours] = 0 (k + 1 of them): scattered straight into g
            bool hit = false;
            for (int p = tid; p < nz; p += NT) {
                const int id = (int)S.nbid[slot][p] + off;
                if (id >= lo && id < lo + W) g[id - lo] = 0.0f;
                hit |= (id == out_tok);
            }
            if (out_tok >= 0 && block_sum_fast<int, NW>(hit ? 1 : 0, S.redi, ph) > 0) out_mass = 0.0f;
            need_bar = true;
        }
        if (lg_nb) {
            for (int p = tid; p < (W + 31) / 32; p += NT) nbmask[p] = 0u;
            __syncthreads();
            for (int p = tid; p < nz; p += NT) {
                const int id = (int)S.nbid[slot][p] + off - lo;
                if (id >= 0 && id < W) atomicOr(&nbmask[id >> 5], 1u << (id & 31));
            }
            need_bar = true;
        }
        if (!is_static) {
            // gtp[x] = 0 (ea_model_llamagen.py:772)
            if (tid == 0 && x_in) g[x - lo] = 0.0f;
            if (!x_in && x == out_tok) out_mass = 0.0f;
            __syncthreads();
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                gn[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(g)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (lazy) gn[it] = dgc(gn[it]);
                loc += (double)gn[it].x + (double)gn[it].y + (double)gn[it].z + (double)gn[it].w;
            }
        } else {
            float4 q[E4];
#pragma unroll
            for (int it = 0; it < E4; ++it) q[it] = qraw[it];
            if (t > 0) {          // q[earlier siblings] = 0; q /= q.sum()
                if (zmask) {
#pragma unroll
                    for (int it = 0; it < E4; ++it) {
                        const unsigned z = zmask >> (4 * it);
                        q[it].x = (z & 1u) ? 0.f : q[it].x; q[it].y = (z & 2u) ? 0.f : q[it].y;
                        q[it].z = (z & 4u) ? 0.f : q[it].z; q[it].w = (z & 8u) ? 0.f : q[it].w;
                    }
                }
                const FastDiv dq((float)(sq - rem));
#pragma unroll
                for (int it = 0; it < E4; ++it) q[it] = dq(q[it]);
            }
            if (need_bar) __syncthreads();           // the zeroing / the mask is visible
            if (lg_nb) {
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    if (FULLW || i4 * 4 < W) {
                        const int e = i4 * 4;
                        const uint32_t b4 = nbmask[e >> 5] >> (e & 31);
                        q[it].x = (b4 & 1u) ? 0.f : q[it].x; q[it].y = (b4 & 2u) ? 0.f : q[it].y;
                        q[it].z = (b4 & 4u) ? 0.f : q[it].z; q[it].w = (b4 & 8u) ? 0.f : q[it].w;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                gn[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (FULLW || i4 * 4 < W) {
                    const float4 qv = q[it];
                    float4 gv = reinterpret_cast<const float4 *>(g)[i4];
                    if (lazy) gv = dgc(gv);
                    gv = sub_clamp0(gv, qv);          // max(gtp - q, 0), two elements per instruction (window_dev.h)
                    gn[it] = gv;
                    loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                }
            }
            // out-of-window mass: the drafter is zero there (precondition): max(out_mass - 0, 0) = out_mass
        }
        // the unnormalised residual goes back to LDS now (every thread owns its entries); the reduction's barrier publishes it
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int i4 = tid + it * NT;
            if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(g)[i4] = gn[it];
        }
        EPN_STAMP(12);
        double tot = block_sum_fast<double, NW>(loc, S.redd, ph);
        EPN_STAMP(13);
        tot += (double)out_mass;
        const float gs = (float)tot;
        if (gs == 0.0f) {
            status = LANTERN_ST_NEEDS_DENSE;
            break;
        }
        gsum = gs;
        lazy = true;
        out_mass = out_mass / gs;
        if (tid == 0) g[W + EW_G_OUT] = out_mass;       // read by the next scan, behind its first barrier at the latest... see below
        if (out_tok >= 0) __syncthreads();              // (only one-hot rows outside the window ever carry mass there)
        EPN_STAMP(14);
    }
    EPN_STAMP(20);

    // ---- no child accepted: this node would end the walk -- bonus token from what g holds (the node's own row if nothing
    // was rejected, else the residual), and the distribution itself when the caller wants sample_p
    int token = -1;
    if (acc < 0 && status == LANTERN_ST_OK && !(flags & 1)) {
        const FastDiv dgc(gsum);
        if (args.dist && internal) {
            float *dw = args.dist + ((size_t)b * n_int + ri) * W;
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W) {
                    float4 v = reinterpret_cast<const float4 *>(g)[i4];
                    if (lazy) v = dgc(v);
                    reinterpret_cast<float4 *>(dw)[i4] = v;
                }
            }
        }
        if (win.u_bonus) token = bonus_draw_lds<NT, E4>(g, W, lo, out_tok, out_mass, ub, S.wtot, S.bonus, S.redi, lazy, dgc);
    }
    EPN_STAMP(21);
    vd.acc = acc;
    vd.n_tried = n_tried; vd.n_rej = n_rej; vd.status = status; vd.token = token; vd.out_tok = out_tok; vd.flags = flags;
    vd.out_mass = out_mass;
}

// node-parallel launch: workgroup (b, r) runs the routine of the node of launch rank r (internal nodes, longest child lists first;
// then leaves) of sequence b and leaves its record for the walk kernel
template <int NT, int E4, int IDMODE, bool FULLW, bool WIDE, int SPEC = 0>
__global__ __launch_bounds__(NT) void epn_kernel(const EpnArgs args) {
    const int r = blockIdx.y, b = blockIdx.x;
    const bool internal = r < args.n_internal;
    const uint4 sw = args.st.w[r];
    int ph = 0, tr_n = 0;
    EpnVerdict vd;
    epn_node<NT, E4, IDMODE, FULLW, WIDE, SPEC>(args, b, sw, internal, internal ? r : 0, (int)(sw.x >> 24), ph, tr_n, vd);
    const int node = sw.x & 255;
    if (args.trace && threadIdx.x == 0) args.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * EN_TR] = (unsigned long long)tr_n | ((unsigned long long)node << 32);
    if (threadIdx.x == 0) {
        int32_t *rec = args.records + ((size_t)b * args.n_nodes + node) * EN_REC;
        rec[0] = vd.acc;
        rec[1] = vd.n_tried;
        rec[2] = vd.n_rej;
        rec[3] = vd.status;
        rec[4] = vd.token;
        rec[5] = vd.out_tok;
        rec[6] = __float_as_int(vd.out_mass);
        rec[7] = vd.flags;
    }
}

// One workgroup per sequence: follow root -> accepted child -> ... through the node records (staged in LDS by one round of
// loads), then write what the chain kernel writes.  With leaf workgroups the bonus token of every possible end of the walk
// is already in its node's record; without them a walk that ends on a leaf draws it here from the leaf's row.
constexpr int EN_WALK_MAX_N = EN_MAX_N;
template <int NT, int E4, bool FULLW>
__global__ __launch_bounds__(NT) void epn_walk_kernel(const EpnArgs args) {
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    extern __shared__ float4 dyn_lds[];
    float *g = reinterpret_cast<float *>(dyn_lds);        // W + 4 floats, only touched when a leaf's token is drawn here
    __shared__ int s_rec[EN_WALK_MAX_N * EN_REC];
    __shared__ int s_node[EN_WALK_MAX_N * 4];
    __shared__ int s_walk[16];
    __shared__ double s_wtot[16];
    __shared__ int s_redi[32];
    __shared__ int s_bonus[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int W = win.win_len, lo = win.win_lo, V = prm.V;
    const int32_t *tb = args.tables;
    const int N = args.n_nodes, n_int = args.n_internal;
    const int32_t *nodeinfo = tb + 8 + EN_INFO * n_int + 4 * args.n_children;
    const int D = prm.D;
    const int32_t *recs = args.records + (size_t)b * N * EN_REC;
    for (int i = tid; i < N * 4; i += NT) s_node[i] = nodeinfo[i];
    for (int i = tid; i < N * EN_REC; i += NT) {
        const int nd = i / EN_REC;
        const bool has = args.leaf_wgs || nodeinfo[4 * nd + 2] >= 0;       // without leaf workgroups a leaf has no record
        int v = has ? recs[i] : 0;
        if (!has && (i % EN_REC) == 0) v = -1;
        s_rec[i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int node = 0, status = LANTERN_ST_OK, n_tried = 0, n_rej = 0, from_res = 0;
        for (int step = 0; step < N; ++step) {
            const int *rec = s_rec + node * EN_REC;
            if (rec[7] & 1) {
                status = LANTERN_ST_NEEDS_CHAIN;
                break;
            }
            n_tried += rec[1];
            n_rej += rec[2];
            if (rec[3] != LANTERN_ST_OK) {
                status = rec[3];
                break;
            }
            if (rec[0] >= 0 && rec[0] < N) {
                node = rec[0];
                continue;
            }
            from_res = (rec[2] > 0 && s_node[4 * node + 1] + 1 != D) ? 1 : 0;      // every child rejected (or none to try): the walk ends here
            break;
        }
        s_walk[0] = node; s_walk[1] = status; s_walk[2] = n_tried; s_walk[3] = n_rej; s_walk[6] = from_res;
    }
    __syncthreads();
    const int node = s_walk[0], status = s_walk[1], from_res = s_walk[6];
    const int *frec = s_rec + node * EN_REC;
    const int rank = s_node[4 * node + 2];
    const bool no_record = !args.leaf_wgs && rank < 0;                   // the walk ended on a leaf nobody computed
    int token = frec[4], out_tok = frec[5];
    float out_mass = __int_as_float(frec[6]);
    const bool want_dist = win.sample_win || buf.sample_p;
    if ((want_dist || (no_record && win.u_bonus)) && status == LANTERN_ST_OK) {
        float4 p[E4];
#pragma unroll
        for (int it = 0; it < E4; ++it) p[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (frec[2] > 0 && rank >= 0 && args.dist) {            // a residual: parked by the node's workgroup
            const float *dw = args.dist + ((size_t)b * n_int + rank) * W;
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W) p[it] = reinterpret_cast<const float4 *>(dw)[i4];
            }
        } else {                                                 // the node's own row
            const int hot = win.row_hot ? win.row_hot[(size_t)b * prm.rows_per_seq + node] : -1;
            if (no_record) {
                out_tok = -1;
                out_mass = 0.0f;
            }
            if (hot >= 0) {
                if (hot >= lo && hot < lo + W) {
#pragma unroll
                    for (int it = 0; it < E4; ++it) {
                        const int e = lo + (tid + it * NT) * 4;
                        if (hot >= e && hot < e + 4) set_comp(p[it], hot - e, 1.0f);
                    }
                } else if (no_record) {
                    out_tok = hot;
                    out_mass = 1.0f;
                }
            } else {
                row_load<NT, E4, FULLW>(buf.logits + ((size_t)b * prm.rows_per_seq + node) * W, W, p);
#pragma unroll
                for (int it = 0; it < E4; ++it) {            // lanes beyond the window hold row_load's -inf padding: no mass
                    const int i4 = tid + it * NT;
                    if (!(FULLW || i4 * 4 < W)) p[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        if (no_record && win.u_bonus) {
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(g)[i4] = p[it];
            }
            __syncthreads();
            token = bonus_draw_lds<NT, E4>(g, W, lo, out_tok, out_mass, win.u_bonus[b], s_wtot, s_bonus, s_redi);
        }
        if (win.sample_win) {
            float *sw = win.sample_win + (size_t)b * W;
#pragma unroll
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(sw)[i4] = p[it];
            }
        }
        if (buf.sample_p) {
            // dense copy: zero fill outside the window (+ the out-of-window token), then the window
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
            for (int it = 0; it < E4; ++it) {
                const int i4 = tid + it * NT;
                if (FULLW || i4 * 4 < W) reinterpret_cast<float4 *>(sp + lo)[i4] = p[it];
            }
        }
    }
    if (tid == 0) {
        const int depth = s_node[4 * node + 1];
        const int a = depth + 1;
        buf.best[b] = s_node[4 * node];
        buf.accept_len[b] = depth;
        int32_t *c = buf.counters + (size_t)b * 6;
        c[0] = a < D - 1 ? a : D - 1;
        c[1] = s_walk[2];
        c[2] = s_walk[3];
        c[3] = s_walk[2];
        c[4] = from_res;
        c[5] = status;
        if (buf.cursor) buf.cursor[b] = buf.cursor[b] + s_walk[2];
        if (win.out_tok) win.out_tok[b] = out_tok;
        if (win.out_mass) win.out_mass[b] = out_mass;
        if (win.u_bonus && win.token && status == LANTERN_ST_OK) win.token[b] = token;
        if (win.verdict_host) {          // the same verdict where the host polls it (pinned memory): record first, the ready word last
            volatile int32_t *vh = win.verdict_host + (size_t)b * 16;
            const long long tk = (win.u_bonus && win.token && status == LANTERN_ST_OK) ? (long long)token : -1ll;
            vh[0] = s_node[4 * node]; vh[1] = depth; vh[2] = c[0]; vh[3] = c[1]; vh[4] = c[2]; vh[5] = c[3]; vh[6] = c[4]; vh[7] = c[5];
            vh[8] = (int32_t)(tk & 0xffffffffll); vh[9] = (int32_t)(tk >> 32);
            __threadfence_system();
            vh[10] = 1;
        }
    }
}

}  // namespace lantern

using namespace lantern;

// diagnosis: a device buffer of grid * 64 u64 that the NEXT node launches fill with phase stamps (tools/epn_trace.py); NULL disarms
static unsigned long long *g_epn_trace = nullptr;
extern "C" int lantern_debug_epn_trace(void *dev_buf) {
    g_epn_trace = (unsigned long long *)dev_buf;
    return LANTERN_OK;
}

extern "C" size_t lantern_evaluate_posterior_nodes_workspace(const lantern_ep_params *prm, const lantern_ep_window *win, int n_internal,
                                                             int want_dist) {
    if (!prm || !win || n_internal <= 0) return 0;
    size_t rec = ((size_t)prm->B * (size_t)prm->rows_per_seq * EN_REC * sizeof(int32_t) + 255) & ~(size_t)255;
    size_t dist = want_dist ? (size_t)prm->B * n_internal * (size_t)win->win_len * sizeof(float) : 0;
    return rec + dist;
}

extern "C" int lantern_evaluate_posterior_nodes(const lantern_ep_params *prm, const lantern_ep_buffers *buf, const lantern_ep_window *win,
                                                const lantern_ep_nodes *nodes, void *stream) {
    LANTERN_CHECK_ARG(prm && buf && win && nodes, "evaluate_posterior_nodes: null params");
    const lantern_ep_params &p = *prm;
    LANTERN_CHECK_ARG(p.B >= 0 && p.P > 0 && p.D > 0 && p.V > 0 && p.V % 4 == 0, "evaluate_posterior_nodes: bad B/P/D/V");
    if (p.B == 0) return LANTERN_OK;
    LANTERN_CHECK_ARG(nodes->tables && nodes->workspace && nodes->n_nodes > 0 && nodes->n_internal > 0 && nodes->n_internal <= nodes->n_nodes &&
                          nodes->n_nodes <= p.rows_per_seq,
                      "evaluate_posterior_nodes: node tables / workspace missing or inconsistent with rows_per_seq");
    LANTERN_CHECK_ARG(win->win_lo >= 0 && win->win_lo % 4 == 0 && win->win_len > 0 && win->win_len % 4 == 0 &&
                          win->win_lo + win->win_len <= p.V && win->win_len <= 16384,
                      "evaluate_posterior_nodes: window [%d,+%d) must be 4-aligned, inside V and <= 16384 wide", win->win_lo, win->win_len);
    LANTERN_CHECK_ARG(p.n_syntax >= 0 && p.n_syntax <= 8 && p.mode >= 0 && p.mode <= 2, "evaluate_posterior_nodes: bad mode/n_syntax");
    LANTERN_CHECK_ARG(buf->logits && buf->uniforms && buf->best && buf->accept_len && buf->counters, "evaluate_posterior_nodes: null required buffer");
    LANTERN_CHECK_ARG(!p.row_index_per_seq && !buf->n_paths && !buf->n_depth, "evaluate_posterior_nodes: one tree for all sequences in this build (per-sequence trees: evaluate_posterior_window)");
    if (p.mode != LANTERN_MODE_DYNAMIC)
        LANTERN_CHECK_ARG(buf->cart_prob && buf->orig_prob && buf->b_idx && buf->tree_cand && p.R > 0 && p.N > 0 && p.N >= nodes->n_nodes &&
                              win->orig_prob_stride >= win->win_len && win->orig_prob_offset % 4 == 0 && win->orig_prob_stride % 4 == 0,
                          "evaluate_posterior_nodes: static mode needs cart_prob/orig_prob(+4-aligned stride/offset)/b_idx/tree_cand");
    else
        LANTERN_CHECK_ARG(buf->cand, "evaluate_posterior_nodes: dynamic mode needs cand");
    if (p.lantern)
        LANTERN_CHECK_ARG(buf->nn_table && p.k >= 1 && p.k <= p.table_cols && p.table_rows > 0, "evaluate_posterior_nodes: lantern needs nn_table, 1<=k<=cols");
    if (win->u_bonus) LANTERN_CHECK_ARG(win->token, "evaluate_posterior_nodes: u_bonus needs token");
    const int nz = (p.k + 1 < p.table_cols) ? p.k + 1 : p.table_cols;
    if ((win->rows_kind != LANTERN_ROWS_PROBS || (p.lantern && nz > EW_PF_K) || (p.top_p > 0.0f && p.top_p < 1.0f))) {
        set_error("evaluate_posterior_nodes: needs probability rows (LANTERN_ROWS_PROBS) and k + 1 <= %d: use evaluate_posterior_window", EW_PF_K);
        return LANTERN_E_UNSUPPORTED;
    }
    const bool want_dist = win->sample_win || buf->sample_p;
    const size_t need = lantern_evaluate_posterior_nodes_workspace(prm, win, nodes->n_internal, want_dist);
    LANTERN_CHECK_ARG(nodes->workspace_bytes >= need && ((uintptr_t)nodes->workspace & 15) == 0, "evaluate_posterior_nodes: workspace of %zu bytes needed (16-byte aligned), %zu given",
                      need, nodes->workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int W = win->win_len;
    LANTERN_CHECK_ARG(nodes->n_nodes <= EN_WALK_MAX_N && nodes->n_children == nodes->n_nodes - 1, "evaluate_posterior_nodes: N=%d > %d nodes (or not a tree)", nodes->n_nodes, EN_WALK_MAX_N);
    if (!nodes->prefix_siblings && p.mode != LANTERN_MODE_DYNAMIC) {
        set_error("evaluate_posterior_nodes: a child's earlier-sibling list is not the list of children tried before it (tables[6] == 0): use evaluate_posterior_window");
        return LANTERN_E_UNSUPPORTED;
    }
    // leaves as workgroups of the node launch (their bonus token pre-drawn) while the launch is small; beyond ~2 workgroups per
    // compute-unit slot the walk kernel draws the one token a walk needs instead
    const int leaf_wgs = !win->u_bonus ? 0 : (nodes->leaf_workgroups >= 0 ? (nodes->leaf_workgroups != 0) : ((long)p.B * nodes->n_nodes <= 1024));
    LANTERN_CHECK_ARG(nodes->tables_host && nodes->tables_host[0] == nodes->n_nodes && nodes->tables_host[1] == nodes->n_internal,
                      "evaluate_posterior_nodes: tables_host missing or not the tables of this tree");
    EpnArgs args{p, *buf, *win, nodes->tables, (int32_t *)nodes->workspace, nullptr, nodes->n_internal, nodes->n_nodes, nodes->n_children, leaf_wgs, g_epn_trace, {}};
    {
        const int32_t *th = nodes->tables_host;
        const int Nn = nodes->n_nodes, ni = nodes->n_internal;
        const int32_t *ord = th + 8 + EN_INFO * ni + 4 * nodes->n_children + 4 * Nn;
        for (int r = 0; r < Nn; ++r) {
            const int32_t *e = th + 8 + EN_INFO * (r < ni ? r : 0);
            const int nd = ord[r];
            LANTERN_CHECK_ARG(nd >= 0 && nd < Nn && (r >= ni || (e[0] == nd && e[4] < 256 && e[5] >= 0 && e[5] < 256 && e[3] < 256)),
                              "evaluate_posterior_nodes: corrupt node tables (rank %d)", r);
            LANTERN_CHECK_ARG(r >= ni || (e[12] / p.D < 256 && e[13] / p.D < 256 && e[14] / p.D < 256 && e[15] / p.D < 256 && e[8] < 256 && e[9] < 256 && e[10] < 256 && e[11] < 256),
                              "evaluate_posterior_nodes: tree too large for the node tables (rank %d)", r);
            uint4 w = make_uint4((uint32_t)nd, 0u, 0u, 0u);
            if (r < ni) {
                w.x |= (uint32_t)(e[2] > 255 ? 255 : e[2]) << 8 | (uint32_t)e[3] << 16 | (uint32_t)e[4] << 24;
                w.y = (uint32_t)e[5] | (uint32_t)e[8] << 8 | (uint32_t)e[9] << 16 | (uint32_t)e[10] << 24;
                w.z = (uint32_t)e[11] | (uint32_t)(e[12] / p.D) << 8 | (uint32_t)(e[13] / p.D) << 16 | (uint32_t)(e[14] / p.D) << 24;
                w.w = (uint32_t)(e[15] / p.D);
            }
            args.st.w[r] = w;
        }
    }
    if (want_dist)
        args.dist = (float *)((char *)nodes->workspace + (((size_t)p.B * (size_t)p.rows_per_seq * EN_REC * sizeof(int32_t) + 255) & ~(size_t)255));
    const size_t lds = epn_shared_offset(W) + sizeof(EnShared);
    const size_t wlds = (size_t)(W + EW_G_EXT) * 4;
    // every node is a workgroup: internal nodes run their children's chain, leaves only draw the bonus token of a walk that ends on them
    const int n_wg_nodes = leaf_wgs ? nodes->n_nodes : nodes->n_internal;
    const bool packed = p.lantern && p.table_cols % 8 == 0 && ((uintptr_t)buf->nn_table & 15) == 0;
    dim3 grid((unsigned)p.B, (unsigned)n_wg_nodes), wgrid(p.B);
    const bool wide = nodes->max_children > 4;
    // measurement aid: an armed (start, stop) pair brackets BOTH kernels (start at the node kernel's begin, stop at the walk's end)
    void *ev0 = nullptr, *ev1 = nullptr;
    take_launch_events(&ev0, &ev1);
#define EPN_LAUNCH(NT_, E4_, FW_)                                                                                         \
    do {                                                                                                                  \
        if (packed && !wide) hipExtLaunchKernelGGL((epn_kernel<NT_, E4_, 2, FW_, false>), grid, dim3(NT_), lds, st, (hipEvent_t)ev0, nullptr, 0, args); \
        else if (packed) hipExtLaunchKernelGGL((epn_kernel<NT_, E4_, 2, FW_, true>), grid, dim3(NT_), lds, st, (hipEvent_t)ev0, nullptr, 0, args);       \
        else hipExtLaunchKernelGGL((epn_kernel<NT_, E4_, 1, FW_, true>), grid, dim3(NT_), lds, st, (hipEvent_t)ev0, nullptr, 0, args);                   \
        hipExtLaunchKernelGGL((epn_walk_kernel<NT_, E4_, FW_>), wgrid, dim3(NT_), wlds, st, nullptr, (hipEvent_t)ev1, 0, args);         \
    } while (0)
    // compile-time instances for the reference's configurations on the Chameleon vocabulary (see epn_node): Lumina static (1), Anole static (4)
    const int spec_knob = tuning(TUNE_EPW_SPEC);          // 0 = the generic instance
    const bool chameleon = spec_knob != 0 && packed && !wide && W == 8192 && p.lantern && p.V == 65536 && p.img_lo == 4 && p.img_hi == 8196 && p.tok_offset == 4 &&
                           p.table_rows == 8192 && win->win_lo == 4;
    const bool lumina = chameleon && p.mode == LANTERN_MODE_STATIC_LUMINA && p.syntax_shortcut && p.n_syntax == 4 && p.syntax[0] == 8196 && p.syntax[1] == 8197 &&
                        p.syntax[2] == 8803 && p.syntax[3] == 8828;
    const bool anole = chameleon && p.mode == LANTERN_MODE_STATIC_LG && !p.syntax_shortcut;
    if (lumina || anole) {
        if (lumina) hipExtLaunchKernelGGL((epn_kernel<512, 4, 2, true, false, 1>), grid, dim3(512), lds, st, (hipEvent_t)ev0, nullptr, 0, args);
        else hipExtLaunchKernelGGL((epn_kernel<512, 4, 2, true, false, 4>), grid, dim3(512), lds, st, (hipEvent_t)ev0, nullptr, 0, args);
        hipExtLaunchKernelGGL((epn_walk_kernel<512, 4, true>), wgrid, dim3(512), wlds, st, nullptr, (hipEvent_t)ev1, 0, args);
    }
    else if (W == 8192) EPN_LAUNCH(512, 4, true);
    else if (W <= 8192) EPN_LAUNCH(512, 4, false);
    else EPN_LAUNCH(1024, 4, false);
#undef EPN_LAUNCH
    LANTERN_CHECK_LAUNCH("evaluate_posterior_nodes");
    return LANTERN_OK;
}
