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
//   * the waves have ROLES, each its own loop (its own registers: nothing of one role is live in another), and work ahead
//     of the verdict instead of waiting for it.  Wave 0 owns the walk: it publishes one command word per candidate,
//     runs the k-neighbour cumulative-mass scan and the accept test (:650-684) and publishes the verdict.  Meanwhile
//     the PASS waves (the second half of the workgroup, owners of the residual) compute max(gtp - q, 0) for the whole
//     window into registers -- what a rejection needs (:695-707) is ready when the verdict arrives; the candidate's
//     neighbours are zeroed afterwards by the load waves, their mass taken off the sum -- and the LOAD waves bring the
//     candidate's own row (LDS-DMA), its children's drafter row and the neighbour ids of ITS first children into
//     staging buffers: an acceptance (:685-689) flips three selector bits;
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
constexpr int EPF_TR_MAX = 160, EPF_TR_BLOCKS = 64, EPF_TRR_MAX = 96;
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
// the same from the first load wave (thread 64) and the first pass wave (thread 64 * (1 + load waves)): their own timelines
__device__ unsigned long long g_epf_trace_r[2][EPF_TR_BLOCKS][EPF_TRR_MAX];
__device__ int g_epf_trace_rn[2][EPF_TR_BLOCKS];
__shared__ unsigned long long s_epf_trr[2][EPF_TRR_MAX];
__shared__ int s_epf_trrn[2];
#define EPF_STAMPR(r, id)                                                                                      \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                                         \
            const int n__ = s_epf_trrn[r];                                                                     \
            if (n__ < EPF_TRR_MAX) {                                                                            \
                s_epf_trr[r][n__] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
                s_epf_trrn[r] = n__ + 1;                                                                       \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#else
#define EPF_STAMP(id) do { } while (0)
#define EPF_STAMPR(r, id) do { } while (0)
#endif

enum { FW_CAND = 0, FW_STAGE = 1, FW_LOADQ = 2, FW_SQ = 3, FW_EV = 8, FW_EV_HOT = 8, FW_EV_RAW = 9, FW_EV_FINISH = 10 };
// flags of a FW_CAND command
enum { FWF_ROW = 1, FWF_PROBS = 2, FWF_Q = 4, FWF_LAZY = 8, FWF_SIB = 16, FWF_FIRST = 32, FWF_NEED_SQ = 64, FWF_SCAN = 128 };

struct alignas(16) FwCmd {        // written by wave 0 in front of a barrier, read by every other wave behind it
    int kind, node, flags, slot;  // FW_*; CAND: the candidate's node / STAGE, LOADQ: the current node / events: the accepted child; CAND: its id slot, STAGE: first child
    int sel, par, sib_tok, hot;   // current buffers (bit 0: g, bit 1: drafter row, bit 2: id set); reduction parity; token of the child tried before; events: one-hot token
    float gsum, dq, out_mass;     // the residual's sum (FWF_LAZY); q.sum() after zeroing the earlier siblings (FWF_SIB); FINISH: mass outside the window
    int out_tok;
    int code, m0, status, pad;    // the verdict (published in front of the verdict barrier)
};

struct alignas(16) FwShared {
    FwCmd cmd;
    int nd_kids[FW_MAX_N];                    // by node: child-list offset | children << 16 (0 for a leaf)
    int nd_qrow[FW_MAX_N];                    // by node: drafter row of its children
    int nd_info[FW_MAX_N];                    // by node: first path | depth << 8
    int2 child[FW_MAX_N];                     // child lists: {node, cell}
    int tok[FW_MAX_N];                        // tree_candidates by node (-2: outside [0, V))
    int hot[FW_MAX_N];                        // one-hot class of the node's row (-1: a window row)
    int pre[FW_MAX_N];                        // raw rows: 1 = post-processed up front (win.raw_probs)
    int fl[FW_MAX_N];                         // class of the node's token: bit 1 image token, bit 0 syntax token
    float cart[EW_MAX_PD];                    // cart_candidates_prob by cell
    double red[2][3][16];                     // [parity][residual sum | zeroed neighbours' mass | -][wave]
    double sqp[2][4];                         // per drafter-row buffer: its sum, one partial per load wave (S_q: a later child's q.sum() is S_q minus its earlier siblings' entries)
    float redf[2 * 16];
    double redd[2 * 16];
    int redi[2 * 16];
    double wtot[16];
    int bonus[4];
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

// NT threads, window W = 4 * NT * E4 ids (512 x 4 = the 8192-id Lumina / Anole image range; 256 x 1 = the 1024-id window of the
// reduced-size reference vectors).  RAW (8192 only): rows arrive as raw cond / uncond bf16 logits.
template <int NT, int E4, bool RAW>
__global__ __launch_bounds__(NT) void epf_kernel(const FwArgs args) {
    constexpr int NW = NT / 64, W = 4 * NT * E4;
    constexpr int NL = NW >= 8 ? 3 : 1;                 // load waves 1 .. NL
    constexpr int PW0 = 1 + NL, NP = NW - PW0;          // pass waves PW0 .. NW-1
    constexpr int PT = NP * 64, PC = (W / 4) / PT;      // pass threads; float4 chunks of the window per pass thread
    constexpr int RP = W * 4 / 1024, LC = (RP + NL - 1) / NL;      // 1 KB pieces of a row; pieces per load wave
    constexpr int DC = (RP + NW - 2) / (NW - 1);        // ... per non-scan wave (the root's rows)
    constexpr int SPW = (FW_SLOTS + NL - 1) / NL;       // id slots per load wave
    constexpr int ZOFF = (W + EW_G_ZERO) * 4;           // byte offset of the zero slot
    constexpr int GB = W + EW_G_EXT;                    // floats per distribution buffer
    static_assert((W / 4) % PT == 0 && PC <= 8 && ZOFF < 65536, "pass tile");
    static_assert(!RAW || (NT == 512 && E4 == 4), "raw rows: the 8192-id window on 512 threads");
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    extern __shared__ float4 dyn_lds[];
    float *const gbuf = reinterpret_cast<float *>(dyn_lds);              // two distribution buffers: the one the walk tests against, and staging
    float *const qbuf = gbuf + 2 * GB;                                   // two drafter-row buffers: the current node's children's, and staging
    FwShared &S = *reinterpret_cast<FwShared *>(reinterpret_cast<char *>(gbuf) + ((size_t)2 * GB + (size_t)2 * W) * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_scan = wave == 0, is_load = wave >= 1 && wave <= NL;
    const int b = blockIdx.x;
    const int lo = win.win_lo, V = prm.V, N = args.n_nodes;
    int ph = 0;
#ifdef EPF_TRACE
    if (tid == 0) s_epf_trn = 0;
    if (tid < 2) s_epf_trrn[tid] = 0;
    EPF_STAMP(0);
#endif
    const float *probs = (RAW ? win.raw_probs : buf.logits);
    if (probs) probs += (size_t)b * prm.rows_per_seq * W;
    const float *qbase = buf.orig_prob + (size_t)b * prm.R * (size_t)win.orig_prob_stride + win.orig_prob_offset;

    // a 4 * W byte row (probabilities of a node, or a drafter row) from HBM straight into an LDS buffer: LDS-DMA, 1 KB pieces
    // (16 bytes per lane), no registers in between; s_waitcnt vmcnt(0) before the barrier that publishes the buffer
    auto row_dma = [&](const float *src, float *dst, int first, int stride, int count) {
        for (int i = 0; i < count; ++i) {
            const int p = first + i * stride;
            if (p < RP)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
        }
    };

    // ------------------------------------------------------------------------------------------------ prologue (every wave)
    // the root's requests first: everything whose address the kernel arguments give.  Its row (probability form) and its
    // children's drafter row by LDS-DMA into buffers 0; raw rows: the raw chunks too (used when the root was not post-processed
    // up front)
    float4 rp[E4];
    if (!is_scan) {
        if (probs) row_dma(probs, gbuf, wave - 1, NW - 1, DC);
    }
    if constexpr (RAW) {
        const uint16_t *rc = reinterpret_cast<const uint16_t *>(buf.logits) + (size_t)b * prm.rows_per_seq * V + lo;
        const uint16_t *ru = reinterpret_cast<const uint16_t *>(win.raw_uncond) + (size_t)b * prm.rows_per_seq * V + lo;
        raw_row_load<NT>(rc, ru, rp);
    }
    EPF_STAMP(3);
    {
        // ---- the sequence's small facts -> LDS: every load unconditional (clamped index), all in flight before the first use
        const int32_t *tb = args.tables;
        const int n_int = args.n_internal, npd = prm.P * prm.D, rps = prm.rows_per_seq;
        const int32_t *chl = tb + 8 + FW_INFO * n_int, *nodeinfo = chl + 4 * args.n_children;
        const int in_ = tid < N ? tid : N - 1, ir_ = tid < rps ? tid : rps - 1, ii_ = tid < n_int ? tid : n_int - 1;
        const int ic_ = args.n_children > 0 ? (tid < args.n_children ? tid : args.n_children - 1) : 0;
        const int4 ni = *reinterpret_cast<const int4 *>(nodeinfo + 4 * in_);
        const int64_t t64 = buf.tree_cand[(size_t)b * prm.N + in_];
        int hot_ = -1, pre_ = 0;
        int64_t pos_ = 0;
        if constexpr (RAW) {
            pos_ = win.raw_pos_per_seq ? win.raw_pos_ids[(size_t)b * rps + ir_] : win.raw_pos_ids[ir_] + win.raw_seq_len[b];
            pre_ = (win.raw_pre && win.raw_probs) ? (int)win.raw_pre[ir_] : 0;
        } else if (win.row_hot) {
            hot_ = win.row_hot[(size_t)b * rps + ir_];
        }
        const int4 e03 = *reinterpret_cast<const int4 *>(tb + 8 + FW_INFO * ii_);          // {node, child offset, children, depth}
        const int e5 = tb[8 + FW_INFO * ii_ + 5];                                           // drafter row
        const int4 chv = *reinterpret_cast<const int4 *>(chl + 4 * ic_);                    // {node, cell, ., .}
        constexpr int CT = (EW_MAX_PD + NT - 1) / NT;
        float ct_[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const int t = tid + u * NT;
            ct_[u] = buf.cart_prob[(size_t)b * npd + (t < npd ? t : npd - 1)];
        }
        EPF_STAMP(5);
        // ---- park them
        if constexpr (RAW) {      // the row's class from its position (MultiModalLogitsProcessor, ea_model_lumina_mgpt.py:45-86)
            const int64_t n1 = pos_ - win.raw_pos_base + 1;
            hot_ = (n1 == ((int64_t)win.raw_w_latent + 1) * win.raw_h_latent + 1) ? win.raw_eos_id
                   : (py_mod64(n1, (int64_t)win.raw_w_latent + 1) == 0 ? win.raw_newline_id : -1);
        }
        if (tid < FW_MAX_N) {
            const bool in = tid < N;
            const int tok_ = !in ? -1 : ((t64 < -1 || t64 >= V) ? -2 : (int)t64);
            S.nd_info[tid] = in ? ((ni.x & 255) | ((ni.y & 255) << 8)) : 0;
            if (!in || ni.z < 0) {        // a leaf (internal nodes are written through their rank entry below)
                S.nd_kids[tid] = 0;
                S.nd_qrow[tid] = 0;
            }
            S.tok[tid] = tok_;
            int fl_ = (tok_ >= prm.img_lo && tok_ < prm.img_hi) ? 2 : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) fl_ |= (q < prm.n_syntax && tok_ == prm.syntax[q]) ? 1 : 0;
            S.fl[tid] = fl_;
            S.hot[tid] = tid < rps ? hot_ : -1;
            S.pre[tid] = tid < rps ? pre_ : 0;
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
        if (tid < 2 * 3 * 16) (&S.red[0][0][0])[tid] = 0.0;
        if (tid < 2) {
            gbuf[tid * GB + W + EW_G_ZERO] = 0.0f;
            gbuf[tid * GB + W + EW_G_HUGE] = 3.0e38f;
            gbuf[tid * GB + W + EW_G_OUT] = 0.0f;
        }
    }
    EPF_STAMP(6);
    if (!is_scan) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    EPF_STAMP(1);
    // ---- the root's row as the walk needs it (buffer 0 holds its probability form if there is one)
    int out_tok = -1;
    float out_mass = 0.0f;
    auto one_hot_row = [&](float *g, int hot) {
        const bool inside = hot >= lo && hot < lo + W;
#pragma unroll
        for (int it = 0; it < E4; ++it) {
            const int i4 = tid + it * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e = lo + i4 * 4;
            if (hot >= e && hot < e + 4) set_comp(v, hot - e, 1.0f);
            reinterpret_cast<float4 *>(g)[i4] = v;
        }
        out_tok = inside ? -1 : hot;
        out_mass = inside ? 0.0f : 1.0f;
        __syncthreads();
    };
    {
        const int hot = S.hot[0];
        if (hot >= 0) {
            one_hot_row(gbuf, hot);
        } else if constexpr (RAW) {
            if (!(probs && S.pre[0] != 0))
                raw_row_to_lds<NT>(rp, -1, win.raw_cfg, win.raw_top_k, V, lo, W, gbuf, out_tok, out_mass, S.redf, S.redd, reinterpret_cast<int *>(qbuf + W), ph);
        }
    }
    EPF_STAMP(2);

    // the walk's state: wave 0's; the other waves learn what they need from the command word
    int node = 0, status = LANTERN_ST_OK, sel = 0;          // sel: bit 0 g buffer, bit 1 drafter-row buffer, bit 2 id set
    int n_tried = 0, n_rej = 0, n_used = 0, rej_here = 0, par = 0;
    float gsum = 1.0f;
    bool lazy = false;
    int ucur0 = 0;
    bool resume = false;                  // wave 0: re-entering its loop behind an acceptance that needed every wave (one-hot / raw row)

    for (;;) {
        if (is_scan) {
            // ====================================================================================== wave 0: the walk
            const int off = prm.tok_offset;
            if (!resume) ucur0 = buf.cursor ? buf.cursor[b] : 0;
            // the step's uniforms in registers (lane l: the l-th draw from the cursor on): nobody else reads them
            double un_l = 2.0;
            if (ucur0 + lane < prm.n_uniforms) un_l = buf.uniforms[(size_t)b * prm.n_uniforms + ucur0 + lane];
            auto publish = [&](int kind, int nd, int flags, int slot, int sib, int hot, float dq) {
                if (lane == 0) {
                    S.cmd.kind = kind; S.cmd.node = nd; S.cmd.flags = flags; S.cmd.slot = slot;
                    S.cmd.sel = sel; S.cmd.par = par; S.cmd.sib_tok = sib; S.cmd.hot = hot;
                    S.cmd.gsum = gsum; S.cmd.dq = dq; S.cmd.out_mass = out_mass; S.cmd.out_tok = out_tok;
                    S.cmd.status = status;
                }
            };
            if (!resume) __syncthreads();      // the load waves have parked the root's children's ids and drafter row
            int ev = 0;
            bool q_have = true;           // (the root's drafter row was requested by the prologue; after an event: set below)
            if (resume) q_have = (S.cmd.flags & FWF_Q) != 0;
            for (;;) {
                // ---- arrival at `node`: its row is in g, its children's neighbour ids in the current id set
                float *g = gbuf + (sel & 1) * GB;
                const float *qc = qbuf + ((sel >> 1) & 1) * W;
                const int idset = (sel >> 2) & 1;
                const int kd = S.nd_kids[node];
                const int coff = kd & 0xffff, nch_all = kd >> 16;
                const int nch = nch_all < FW_MAX_CH ? nch_all : FW_MAX_CH;
                if (nch_all > FW_MAX_CH) status = LANTERN_ST_TREE_LIMIT;
                // lane t: child t (node, token, class, cart_candidates_prob)
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
                if (nch > 0 && !q_have && status == LANTERN_ST_OK) {        // (the accepted child was not the first one tried: its drafter row was not requested ahead)
                    publish(FW_LOADQ, node, 0, 0, -1, -1, 1.0f);
                    __syncthreads();
                    __syncthreads();
                }
                float qv_l = 0.0f;        // q[token of child `lane`]: what the later siblings' q.sum() loses
                if (lane < nch && tok_l >= lo && tok_l < lo + W) qv_l = qc[tok_l - lo];
                gsum = 1.0f;
                lazy = false;
                rej_here = 0;
                EPF_STAMP(10);
                double rem = 0.0;
                const double sq = (lane < NL ? S.sqp[(sel >> 1) & 1][lane & 3] : 0.0);        // (summed below)
                const double sq_tot = rdlane(sq, 0) + (NL > 1 ? rdlane(sq, 1) : 0.0) + (NL > 2 ? rdlane(sq, 2) : 0.0);
                int acc_t = -1, staged_lo = 0;   // children [staged_lo, staged_lo + FW_SLOTS) have their ids in the slots
                for (int t = 0; t < nch && status == LANTERN_ST_OK; ++t) {
                    if (t > 0) rem += (double)rdlane(qv_l, t - 1);       // child t - 1 is now an earlier sibling (q[siblings] = 0; q /= q.sum(), :696-700)
                    const int x = rdlane(tok_l, t);
                    if (x == -2) {
                        status = LANTERN_ST_TOKEN_OOB;
                        break;
                    }
                    if (n_used >= FW_UNI || ucur0 + n_used >= prm.n_uniforms) {
                        status = LANTERN_ST_UNIFORMS;
                        break;
                    }
                    const double rr_u = rdlane(un_l, n_used & 63);
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
                    if (t >= staged_lo + FW_SLOTS) {              // more children than staging slots (rare): the next FW_SLOTS of them
                        publish(FW_STAGE, node, 0, t, -1, -1, 1.0f);
                        __syncthreads();
                        __syncthreads();
                        staged_lo = t;
                    }
                    const int slot = t - staged_lo;
                    const int ckd = S.nd_kids[cnode], chot = S.hot[cnode];
                    const bool c_probs = !RAW || (probs && S.pre[cnode] != 0);
                    const bool want_q = t == 0 && (ckd >> 16) > 0;       // the drafter row one level down, for the first (likeliest) child only
                    const float dqv = t > 0 ? (float)(sq_tot - rem) : 1.0f;
                    publish(FW_CAND, cnode,
                            (chot < 0 ? FWF_ROW : 0) | (c_probs ? FWF_PROBS : 0) | (want_q ? FWF_Q : 0) | (lazy ? FWF_LAZY : 0) | (t > 0 ? FWF_SIB : 0) |
                                (t == 0 ? FWF_FIRST : 0) | (scan ? FWF_SCAN : 0),
                            slot, t, coff, dqv);          // (sib_tok / hot fields: the candidate's rank among its siblings, the child list's offset)
                    __syncthreads();                                    // the command
                    int code = is_syn ? (((float)rr_u <= 1.0f / qx) ? 1 : 2) : (((float)rr_u <= 0.0f / qx) ? 1 : 2);      // px = 1 / px = 0 (:654-659)
                    int m0 = 0;
                    if (scan) {
                        // ---------------- the k-neighbour cumulative mass (:661-677), 16 consecutive neighbours per lane
                        const FastDiv dgc(gsum);
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
                        S.cmd.code = code;
                        S.cmd.m0 = m0;
                    }
                    EPF_STAMP(24);
                    __syncthreads();                                    // the verdict
                    EPF_STAMP(21);
                    if (code == 1) {
                        // ------------------------------------------------ acceptance (:685-689): on to the child
                        acc_t = t;
                        sel ^= 4;                                       // its children's ids were staged into the other set
                        if (want_q) sel ^= 2;                           // ... and their drafter row into the other buffer
                        q_have = want_q;
                        node = cnode;
                        out_tok = -1;
                        out_mass = 0.0f;
                        if (chot >= 0) ev = FW_EV_HOT;                  // a one-hot row: written by every wave
                        else if (c_probs) sel ^= 1;                     // the staged probabilities become the distribution
                        else ev = FW_EV_RAW;                            // the staged raw chunks -> CFG, top-k, softmax (every wave)
                        if (ev) publish(ev, cnode, want_q ? FWF_Q : 0, 0, -1, chot, 1.0f);
                        break;
                    }
                    // ------------------------------------------------ rejection (:690-713)
                    ++n_rej;
                    ++rej_here;
                    __syncthreads();                                    // the residual is back in LDS, its sum's partials are there
                    if (is_syn) {
                        status = LANTERN_ST_SYNTAX_REJECT;
                        break;
                    }
                    // (its neighbours -- zeroed by the load waves in front of the next barrier -- leave the sum with the mass the load
                    // waves computed ahead)
                    double tot = fw_row_total(lane < NW ? S.red[par][0][lane] : 0.0);
                    if (scan && m0) tot -= fw_row_total(lane < NW ? S.red[par][1][lane] : 0.0);
                    par ^= 1;
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
                if (ev) break;
                if (acc_t < 0 || status != LANTERN_ST_OK) {
                    ev = FW_EV_FINISH;
                    publish(ev, node, lazy ? FWF_LAZY : 0, 0, -1, -1, 1.0f);
                    break;
                }
            }
            __syncthreads();                                            // the event
        } else if (is_load) {
            // ====================================================================================== load waves
            const int k = prm.k, off = prm.tok_offset;
            const int nz = (k + 1 < prm.table_cols) ? k + 1 : prm.table_cols;
            const uint16_t *raw_c = RAW ? reinterpret_cast<const uint16_t *>(buf.logits) + (size_t)b * prm.rows_per_seq * V + lo : nullptr;
            const uint16_t *raw_u = RAW ? reinterpret_cast<const uint16_t *>(win.raw_uncond) + (size_t)b * prm.rows_per_seq * V + lo : nullptr;
            // neighbour ids of up to FW_SLOTS children of node `pn` from child `first` on (slot s belongs to load wave 1 + s % NL;
            // 2 x 16 bytes per lane): requested ...
            uint4 idq[SPW][2];
            auto ids_req = [&](int pn, int first) {
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
            // ... and parked as byte offsets into g: positions < k inside the window -> 4 * id, everything else -> the zero slot; the
            // (k+1)-th neighbour (zeroed on a rejection, never summed) apart
            auto ids_put = [&](int set) {
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
            // a drafter row through registers into buffer `dst`, its sum's partial of this wave into S.sqp[which]
            auto q_fetch = [&](int pn, float4 (&rq)[LC]) {
                int qr = S.nd_qrow[pn & (FW_MAX_N - 1)];
                qr = qr < 0 ? 0 : (qr >= prm.R ? prm.R - 1 : qr);
                const float *src = qbase + (size_t)qr * (size_t)win.orig_prob_stride;
#pragma unroll
                for (int i = 0; i < LC; ++i) {
                    const int p = (wave - 1) + i * NL;
                    rq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p < RP) rq[i] = *reinterpret_cast<const float4 *>(src + p * 256 + lane * 4);
                }
            };
            auto q_put = [&](float *dst, int which, const float4 (&rq)[LC]) {
                double sl = 0.0;
#pragma unroll
                for (int i = 0; i < LC; ++i) {
                    const int p = (wave - 1) + i * NL;
                    if (p < RP) *reinterpret_cast<float4 *>(dst + p * 256 + lane * 4) = rq[i];
                    sl += (double)rq[i].x + (double)rq[i].y + (double)rq[i].z + (double)rq[i].w;
                }
                sl = wave_sum(sl);
                if (lane == 0) S.sqp[which][wave - 1] = sl;
            };
            if (!resume) {
                // the root's requests, started with the kernel: its first children's tokens -> their table rows (two dependent rounds),
                // its children's drafter row; parked before the first command
                float4 rq[LC];
                if (args.root_nch > 0) q_fetch(0, rq);
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    idq[j][0] = make_uint4(0u, 0u, 0u, 0u);
                    idq[j][1] = make_uint4(0u, 0u, 0u, 0u);
                    const int slot = (wave - 1) + j * NL;
                    if (slot < FW_SLOTS && slot < args.root_nch) {
                        const int x = S.tok[args.root_child[slot] & (FW_MAX_N - 1)];
                        const int trow = x - off;
                        if (x >= prm.img_lo && x < prm.img_hi && trow >= 0 && trow < prm.table_rows) {
                            const uint16_t *row = buf.nn_table + (size_t)trow * prm.table_cols;
                            if (lane * 8 < nz) idq[j][0] = *reinterpret_cast<const uint4 *>(row + lane * 8);
                            if ((lane + 64) * 8 < nz) idq[j][1] = *reinterpret_cast<const uint4 *>(row + (lane + 64) * 8);
                        }
                    }
                }
                ids_put(0);
                if (args.root_nch > 0) q_put(qbuf, 0, rq);
                __syncthreads();
            }
            for (;;) {
                if (wave == 1) EPF_STAMPR(0, 60);
                __syncthreads();                                        // a command
                if (wave == 1) EPF_STAMPR(0, 61);
                const int4 c0 = *reinterpret_cast<const int4 *>(&S.cmd.kind), c1 = *reinterpret_cast<const int4 *>(&S.cmd.sel);
                const int kind = c0.x, nd = c0.y, flags = c0.z, cslot = c0.w, csel = c1.x, cpar = c1.y, nsib = c1.z, coff = c1.w;
                if (kind >= FW_EV) break;
                float *g = gbuf + (csel & 1) * GB, *gs = gbuf + ((csel & 1) ^ 1) * GB;
                float *qc = qbuf + ((csel >> 1) & 1) * W, *qs = qbuf + (((csel >> 1) & 1) ^ 1) * W;
                const int idset = (csel >> 2) & 1;
                if (kind == FW_CAND) {
                    // "this candidate is accepted": its first children's neighbour ids (requested first: the shortest loads), its row,
                    // its children's drafter row -- through registers into the staging buffers, before the verdict's barrier: an
                    // acceptance finds them in LDS
                    ids_req(nd, 0);
                    float4 rr[LC], rq[LC];
                    const bool want_row = (flags & FWF_ROW) != 0, as_probs = (flags & FWF_PROBS) != 0, want_q = (flags & FWF_Q) != 0;
#pragma unroll
                    for (int i = 0; i < LC; ++i) {
                        const int p = (wave - 1) + i * NL;
                        rr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p < RP && want_row) {
                            if (!RAW || as_probs) {
                                rr[i] = *reinterpret_cast<const float4 *>(probs + (size_t)nd * W + p * 256 + lane * 4);
                            } else {          // raw rows: pieces 0 .. RP/2-1 the cond window (bf16), then the uncond window
                                const uint16_t *src = (p < RP / 2 ? raw_c : raw_u) + (size_t)nd * V + (size_t)(p % (RP / 2)) * 512 + lane * 8;
                                const Bf16x8 c = *reinterpret_cast<const Bf16x8 *>(src);
                                rr[i] = make_float4(__uint_as_float(c.a.x), __uint_as_float(c.a.y), __uint_as_float(c.b.x), __uint_as_float(c.b.y));
                            }
                        }
                    }
                    if (want_q) q_fetch(nd, rq);
                    if (wave == 1) EPF_STAMPR(0, 62);
                    // while those are in flight: "this candidate is rejected" -- the mass its neighbours (k + 1 of them, :702-704) would
                    // take out of the residual: max(gtp - q, 0) at their positions, the pass waves' arithmetic entry for entry
                    double rml = 0.0;
                    constexpr int FX = (EW_PF_K + 1 + NL * 64 - 1) / (NL * 64);          // neighbours per load thread
                    int ao_[FX];
                    if (flags & FWF_SCAN) {
                        const char *gb = reinterpret_cast<const char *>(g), *qb = reinterpret_cast<const char *>(qc);
                        const FastDiv dgc(S.cmd.gsum), dq(S.cmd.dq);
                        const bool lz = (flags & FWF_LAZY) != 0, sb = (flags & FWF_SIB) != 0;
                        float gv_[FX], qv_[FX];
#pragma unroll
                        for (int u = 0; u < FX; ++u) {
                            const int p = (tid - 64) + u * NL * 64;
                            ao_[u] = p < k ? (int)S.nbaddr[idset][cslot][p < EW_PF_K ? p : 0] : (p == k ? S.nbk[idset][cslot] : ZOFF);
                        }
#pragma unroll
                        for (int u = 0; u < FX; ++u) {
                            gv_[u] = *reinterpret_cast<const float *>(gb + ao_[u]);                      // (the zero slot reads 0)
                            qv_[u] = *reinterpret_cast<const float *>(qb + (ao_[u] != ZOFF ? ao_[u] : 0));
                        }
                        if (sb) {
                            for (int v = 0; v < nsib; ++v) {          // q[earlier siblings] = 0
                                const int so = (S.tok[S.child[(coff + v) & (FW_MAX_N - 1)].x & (FW_MAX_N - 1)] - lo) * 4;
#pragma unroll
                                for (int u = 0; u < FX; ++u) qv_[u] = (ao_[u] == so) ? 0.0f : qv_[u];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < FX; ++u) {
                            float gv = gv_[u], qv = qv_[u];
                            if (lz) gv = dgc(gv);
                            if (sb) qv = dq(qv);
                            const float d = gv - qv;
                            rml += (ao_[u] != ZOFF) ? (double)(d < 0.0f ? 0.0f : d) : 0.0;
                        }
                        rml = wave_sum(rml);
                    }
                    if (lane == 0) S.red[cpar][1][wave] = rml;
                    ids_put(idset ^ 1);
                    if (want_row) {
#pragma unroll
                        for (int i = 0; i < LC; ++i) {
                            const int p = (wave - 1) + i * NL;
                            if (p < RP) *reinterpret_cast<float4 *>(gs + p * 256 + lane * 4) = rr[i];
                        }
                    }
                    if (want_q) q_put(qs, ((csel >> 1) & 1) ^ 1, rq);
                    if (wave == 1) EPF_STAMPR(0, 64);
                    __syncthreads();                                    // the verdict
                    if (wave == 1) EPF_STAMPR(0, 65);
                    if (S.cmd.code == 2) {
                        __syncthreads();                                // the residual is back in LDS
                        if ((flags & FWF_SCAN) && S.cmd.m0) {           // gtp[neighbours] = 0
                            char *gb = reinterpret_cast<char *>(g);
#pragma unroll
                            for (int u = 0; u < FX; ++u)
                                if (ao_[u] != ZOFF) *reinterpret_cast<float *>(gb + ao_[u]) = 0.0f;
                        }
                        if (wave == 1) EPF_STAMPR(0, 66);
                    }
                } else if (kind == FW_STAGE) {
                    ids_req(nd, cslot);
                    ids_put(idset);
                    __syncthreads();
                } else if (kind == FW_LOADQ) {
                    float4 rq[LC];
                    q_fetch(nd, rq);
                    q_put(qc, (csel >> 1) & 1, rq);
                    __syncthreads();
                } else {
                    __syncthreads();
                }
            }
        } else {
            // ====================================================================================== pass waves (owners of the residual)
            const int pt = tid - PW0 * 64;
            unsigned zmask = 0;              // which of this thread's entries of the drafter row belong to earlier siblings
            if (!resume) __syncthreads();    // (the load waves have parked the root's children's ids and drafter row)
            for (;;) {
                if (wave == PW0) EPF_STAMPR(1, 70);
                __syncthreads();                                        // a command
                if (wave == PW0) EPF_STAMPR(1, 71);
                const int4 c0 = *reinterpret_cast<const int4 *>(&S.cmd.kind), c1 = *reinterpret_cast<const int4 *>(&S.cmd.sel);
                const int kind = c0.x, flags = c0.z, csel = c1.x, cpar = c1.y, sib = c1.z;
                if (kind >= FW_EV) break;
                float *g = gbuf + (csel & 1) * GB;
                const float *qc = qbuf + ((csel >> 1) & 1) * W;
                if (kind == FW_CAND) {
                    // "this candidate is rejected": max(gtp - q, 0) of the whole window, into registers (q: earlier siblings zeroed, renormalised)
                    // which of this thread's entries of the drafter row belong to the candidate's earlier siblings (skipped ones included)
                    zmask = 0;
                    for (int u = 0; u < sib; ++u) {
                        const int xp = S.tok[S.child[(c1.w + u) & (FW_MAX_N - 1)].x & (FW_MAX_N - 1)] - lo;
                        if (xp >= 0 && xp < W && ((xp >> 2) % PT) == pt) zmask |= 1u << ((((xp >> 2) / PT) << 2) | (xp & 3));
                    }
                    const FastDiv dgc(S.cmd.gsum), dq(S.cmd.dq);
                    const bool lz = (flags & FWF_LAZY) != 0, sb = (flags & FWF_SIB) != 0;
                    float4 gn[PC];
                    double loc = 0.0;
#pragma unroll
                    for (int it = 0; it < PC; ++it) {
                        float4 qv = reinterpret_cast<const float4 *>(qc)[pt + it * PT];
                        if (sb) {
                            const unsigned z = zmask >> (4 * it);
                            qv.x = (z & 1u) ? 0.f : qv.x; qv.y = (z & 2u) ? 0.f : qv.y;
                            qv.z = (z & 4u) ? 0.f : qv.z; qv.w = (z & 8u) ? 0.f : qv.w;
                            qv = dq(qv);
                        }
                        float4 gv = reinterpret_cast<const float4 *>(g)[pt + it * PT];
                        if (lz) gv = dgc(gv);
                        float d;
                        d = gv.x - qv.x; gv.x = d < 0.0f ? 0.0f : d;
                        d = gv.y - qv.y; gv.y = d < 0.0f ? 0.0f : d;
                        d = gv.z - qv.z; gv.z = d < 0.0f ? 0.0f : d;
                        d = gv.w - qv.w; gv.w = d < 0.0f ? 0.0f : d;
                        gn[it] = gv;
                        loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                    }
                    if (wave == PW0) EPF_STAMPR(1, 72);
                    __syncthreads();                                    // the verdict
                    if (wave == PW0) EPF_STAMPR(1, 73);
                    if (S.cmd.code == 2) {
                        // the unnormalised residual goes back to LDS; S_q with the same barriers (a later child's q.sum() is S_q minus
                        // its earlier siblings' entries)
#pragma unroll
                        for (int it = 0; it < PC; ++it) reinterpret_cast<float4 *>(g)[pt + it * PT] = gn[it];
                        loc = wave_sum(loc);
                        if (lane == 0) S.red[cpar][0][wave] = loc;
                        if (wave == PW0) EPF_STAMPR(1, 74);
                        __syncthreads();
                        if (wave == PW0) EPF_STAMPR(1, 75);
                    }
                } else {
                    __syncthreads();
                }
            }
        }
        // ========================================================================================== every wave: what an event needs
        const int ekind = S.cmd.kind;
        if (ekind == FW_EV_FINISH) break;
        {
            const int csel = S.cmd.sel;         // (already flipped to the accepted child's buffers by wave 0)
            float *g = gbuf + (csel & 1) * GB;
            if (ekind == FW_EV_HOT) {
                one_hot_row(g, S.cmd.hot);
            } else if constexpr (RAW) {         // FW_EV_RAW: the staged raw chunks -> CFG, top-k, softmax -> g
                const char *sb = reinterpret_cast<const char *>(gbuf + ((csel & 1) ^ 1) * GB);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int ch = tid + it * NT;
                    rp[it] = *reinterpret_cast<const float4 *>(sb + (size_t)ch * 16);
                    rp[2 + it] = *reinterpret_cast<const float4 *>(sb + (size_t)W * 2 + (size_t)ch * 16);
                }
                // (the histograms live in the drafter-row buffer that is not the current one)
                raw_row_to_lds<NT>(rp, -1, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd,
                                   reinterpret_cast<int *>(qbuf + (((csel >> 1) & 1) ^ 1) * W), ph);
            }
        }
        resume = true;
    }
    EPF_STAMP(40);

    // ------------------------------------------------------------------------------------------------ the end of the walk (every wave)
    // g holds the last node's own row if nothing was rejected there, else the (unnormalised) residual; wave 0 published the rest
    {
        const int csel = S.cmd.sel;
        lazy = (S.cmd.flags & FWF_LAZY) != 0;
        gsum = S.cmd.gsum;
        out_mass = S.cmd.out_mass;
        out_tok = S.cmd.out_tok;
        status = S.cmd.status;
        node = S.cmd.node;
        const float *g = gbuf + (csel & 1) * GB;
        const double ub = win.u_bonus ? win.u_bonus[b] : 0.0;
        const int D = prm.D;
        const int info = S.nd_info[node & (FW_MAX_N - 1)];
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
            for (int r = 0; r < 2; ++r) {
                const int m = s_epf_trrn[r];
                for (int t = 0; t < m; ++t) g_epf_trace_r[r][b][t] = s_epf_trr[r][t];
                g_epf_trace_rn[r][b] = m;
            }
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
}

}  // namespace lantern

using namespace lantern;

#ifdef EPF_TRACE
extern "C" int lantern_debug_epf_trace_role(int role, unsigned long long *host_out, int *counts) {
    if (role < 0 || role > 1) return -1;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_epf_trace_rn), sizeof(int) * EPF_TR_BLOCKS, sizeof(int) * EPF_TR_BLOCKS * role) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_epf_trace_r), sizeof(unsigned long long) * EPF_TR_BLOCKS * EPF_TRR_MAX,
                            sizeof(unsigned long long) * EPF_TR_BLOCKS * EPF_TRR_MAX * role) != hipSuccess) return -1;
    return EPF_TRR_MAX;
}
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
