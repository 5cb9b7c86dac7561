// epw_body.h -- the windowed chain kernel of evaluate_posterior (O8): one workgroup per sequence (epw_body / epw_kernel), shared by the
// translation units that instantiate it (window_kernels.hip: the latency instances of the fixed configurations; epw_generic.hip: the
// argument-driven instances for any window / table form; epw_throughput.hip: the forms for more sequences per launch than CUs) -- three
// files so that the ~55 instances of this 800-line template compile side by side.
//
// Reference: models/ea_model_lumina_mgpt.py:610-726 (O8), :781 (bonus token); models/ea_model_llamagen.py:597-669,709-787.
#pragma once
#include <type_traits>
#include "window_dev.h"
#include "tree_dynamic_dev.h"
#include "prep_dev.h"

// in-kernel phase stamps: only the diagnostic build of window_kernels.hip defines them (-DEPW_TRACE); nothing everywhere else
#ifndef EPW_STAMP
#define EPW_STAMP(id) do { } while (0)
#endif
#ifndef EPW_STAMPF
#define EPW_STAMPF(id) do { } while (0)
#endif

namespace lantern {

// ------------------------------------------------------------------------------- O8 windowed
//
// Structure (v3).  A 512-thread workgroup owns one sequence.  The serial part of the algorithm -- walking the
// candidates of a level, the k-neighbour cumulative-mass scan, the accept test -- is executed by WAVE 0 ONLY
// (one lane per path for the prefix masks, 16 neighbours per lane for the scan, DPP scans); the other waves
// wait at a barrier and join for the W-wide passes (row softmax, residual update, renormalisation), whose
// cross-wave reductions combine <= 16 partials with one DPP row.  All decisions travel through one LDS word,
// so control flow stays workgroup-uniform.  The neighbour ids of every candidate of a level are fetched in one
// round at level start (they depend only on the accepted prefix), so the per-candidate work is LDS + ALU only.

// NUC: TopPLogitsWarper (top_p in (0, 1)) between the temperature and the top-k of a row that arrives as logits -- HF order Temperature -> TopP ->
// TopK, drafters/utils.py:36-52; `mass`: top_p_tile's 256 f64 bins in LDS.  A template parameter: the filter keeps a second copy of the row in registers.
template <int NT, int E4, bool FULLW = false, typename Hook = NoHook, bool NUC = false, typename SH = EwShared>
__device__ __forceinline__ void row_softmax_to_lds(float4 (&r)[E4], int hot, bool probs, int win_lo, int W, float temperature, int top_k,
                                                   int V, float *g, int &out_tok, float &out_mass, SH &S, int &ph,
                                                   const Hook &pre_barrier = Hook(), float top_p = 1.0f, double *mass = nullptr) {
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
        if constexpr (NUC) {
            if (top_p >= 1e-8f && top_p < 1.0f && mass) top_p_tile<NT, E4>(r, top_p, mass, S.redf, S.redd, S.redi, ph);
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

// COMPACT builds: a probability row goes straight into g by LDS-DMA (global_load_lds_dwordx4: per-lane source address, LDS destination = wave-uniform
// base + lane * 16 -- exactly g's token order), no VGPR staging and no ds_write pass.  The caller waits vmcnt(0) and takes the barrier.
template <int NT, int E4>
__device__ __forceinline__ void row_dma_to_lds(const float *__restrict__ rowp, float *g) {
    const int tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
    for (int it = 0; it < E4; ++it) {
        const float *src = rowp + (size_t)(tid + it * NT) * 4;
        float *dst = g + (size_t)(wave * 64 + it * NT) * 4;          // (wave-uniform)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
}

// a row another workgroup of this launch published (EpwFused): agent-scope atomic loads, the window is the workgroup's tile (FULLW)
template <int NT, int E4>
__device__ __forceinline__ void row_load_agent(const float *__restrict__ rowp, float4 (&r)[E4]) {
#pragma unroll
    for (int it = 0; it < E4; ++it) r[it] = load_f4_agent(rowp + (size_t)(threadIdx.x + it * NT) * 4);
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

// The prepare stage INSIDE the chain launch (lantern_step_group flags & LANTERN_STEP_FUSED_PREPARE; epw_kernel_fused below): the launch's first n_helpers
// workgroups (a whole number per sequence) post-process the listed rows (all but node_list[0], the root) into win.raw_probs -- as agent-scope atomic stores, no
// fence: an agent-scope release / acquire pair writes back and invalidates the XCD's L2, measured 64 us per launch against 32 -- and publish each by storing `epoch`
// into ready[b * rows_per_seq + node]; the sequence workgroups assemble their own candidates (and write them out for the commit launch), post-process the root's
// row themselves -- they would only wait for a helper to do the same -- and take a listed row from raw_probs when its word says this step's epoch, else
// post-process it themselves (same bits either way: which of the two happens is timing, never the result).  One kernel boundary and one launch less on
// every group's chain.
struct EpwFused {
    PrepArgs prep;
    int32_t *ready;
    int32_t epoch, n_helpers;
};

template <int NT, int E8>
__device__ __forceinline__ void epw_helper_row(const EpwFused &f, const int hx) {
    __shared__ alignas(16) int s_hist[O7_HIST_INTS];
    __shared__ float s_redf[32];
    __shared__ double s_redd[32];
    __shared__ int s_redi[32];
    const PrepArgs &a = f.prep;
    // `per` helper workgroups per sequence, helper h of a sequence takes the listed nodes 1 + h, 1 + h + per, ... one after the other: the list is in likelihood
    // order, so every helper publishes its likeliest row first (a helper workgroup costs a CU for ~5 us per row: the launch carries the chain kernel's LDS footprint)
    const int per = f.n_helpers / a.B, n_rows = f.n_helpers;
    const int x = o7_row_of_block(hx, n_rows, per);            // helpers of sequence b on XCD b % 8, where its chain runs (n_helpers % 8 == 0)
    const int b = x / per, h = x % per;
    const int64_t len_b = a.w_latent > 0 ? a.seq_len[b] : 0;
    for (int j = 1 + h; j < a.n_list; j += per) {
        const int node = a.node_list[j];
        const int row = b * a.rows_per_seq + node;
        const int cls = a.w_latent > 0 ? lumina_row_class(a.pos_ids[node] + len_b, a.pos_base, a.w_latent, a.h_latent) : 0;
        cfg_window_bf16_row<NT, E8, true, false, true>(row, cls, a.cond, a.uncond, a.V, a.cfg, LANTERN_MODEL_LUMINA, a.img_lo, a.img_hi, a.newline_id, a.eos_id,
                                                       a.top_k, a.win_lo, a.W, a.out_win, a.row_hot, LANTERN_ROWS_PROBS, s_hist, s_redf, s_redd, a.top_p, s_redi);
        // the row went out as agent-scope atomic stores (store_f4_agent): once every thread's stores are acknowledged ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // (also: the next row may reuse the histogram / reduction scratch)
        if (threadIdx.x == 0) __hip_atomic_store(f.ready + row, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // ... the row's word says so
    }
}

template <int NT, int E4, int IDMODE, int WPE, bool FULLW = false, bool RAW = false, int SPEC = 0, int TPO = 0>
__device__ __forceinline__ int epw_body(const EpwArgs &args, const int b, const EpwFused *fz = nullptr) {
    constexpr bool NUCLEUS = (TPO & 2) != 0;         // raw rows: TopPLogitsWarper (prm.top_p) in front of the top-k of the rows the walk post-processes
    constexpr bool LATE_Q = (TPO & 1) != 0;          // throughput builds: a candidate's drafter row is requested once its rejection is known (an accepted
                                                     // candidate -- 0.65 of the first tries -- then costs no row request at all; the latency is another workgroup's problem)
    static_assert(!RAW || (FULLW && (E4 == 4 || E4 == 8)), "raw rows: W = 4 * E4 * NT ids, E4 / 2 sixteen-byte chunks of cond and of uncond per thread");
    constexpr int CH = E4 / 2 > 0 ? E4 / 2 : 1;          // raw rows: 8-id chunks per thread and operand
    constexpr bool ROTW = (TPO & 8) != 0;            // throughput builds: the wave that runs a sequence's serial section rotates with the sequence (b % NW) instead of
                                                     // always being wave 0 -- four workgroups share a CU, and their serial sections then sit on different SIMDs
    constexpr bool ALLW = (TPO & 16) != 0;           // throughput builds: the k-neighbour cumulative-mass scan on ALL waves (EW_PF_K / NT neighbours per lane, one DPP scan per
                                                     // wave, the waves' totals exchanged through LDS) and the decision computed by every wave -- no serial section at all
    constexpr bool LDS2P = (TPO & 64) != 0;          // the residual of a rejection goes back to LDS unnormalised and is normalised by a second pass over LDS: no 8-float4
                                                     // register copy of the window lives across the reduction (the four-per-CU build sits at its 128-VGPR cap)
    constexpr bool PRIO = (TPO & 128) != 0;          // the serial section runs at raised wave priority (s_setprio): four workgroups share a CU, and a sequence's progress is gated
                                                     // by its one serial wave while the other workgroups' W-wide passes compete for the same SIMD's issue slots
    constexpr bool COMPACT = (TPO & 4) != 0;         // the default tree's throughput build on the smallest staged tables (EwSharedCompact, no neighbour
                                                     // bit mask, 2 prefetch slots): 40 KB of LDS, four workgroups per CU
    static_assert(!COMPACT || (SPEC == 2 && !RAW), "the compact tables are sized for the reference's default tree mc_sim_7b_63 on probability rows");
    constexpr bool LITE = (TPO & 256) != 0;          // LlamaGen's throughput build: the 64 KB window leaves 14 KB for everything else if TWO workgroups are to share a CU --
                                                     // EwSharedLite (trees of <= 64 nodes, no neighbour lists: LANTERN off), no neighbour bit mask
    static_assert(!LITE || (SPEC == 5 && !RAW && !COMPACT), "the lite tables: LlamaGen's standard verify (dynamic trees, LANTERN off) on probability rows");
    constexpr bool DMAROW = COMPACT || (TPO & 512) != 0;          // probability rows land in g by LDS-DMA (row_dma_to_lds): no VGPR staging, no ds_write pass
    static_assert(!DMAROW || (SPEC >= 1 && !RAW && FULLW), "LDS-DMA rows: final probability rows of a fixed configuration whose window is the workgroup's tile");
    constexpr bool FUSED = (TPO & 1024) != 0;        // the prepare stage rides in this launch (EpwFused above): candidates assembled here, listed rows taken when published
    static_assert(!FUSED || (RAW && (SPEC == 1 || SPEC == 2 || SPEC == 4) && WPE == 1), "fused prepare: the static-tree latency instances (Lumina, Anole) on raw rows");
    typedef typename std::conditional<COMPACT, EwSharedCompact, typename std::conditional<LITE, EwSharedLite, EwShared>::type>::type SH;
    constexpr int MAX_B = SH::kMaxB, MAX_N = SH::kMaxN, N_UNI = SH::kUni;
    constexpr bool LDSIDS = IDMODE != 0;
    const lantern_ep_params &prm = args.prm;
    const lantern_ep_buffers &buf = args.buf;
    const lantern_ep_window &win = args.win;
    constexpr int NW = NT / 64;
    static_assert(!ALLW || (IDMODE != 0 && NW <= 8 && EW_PF_K % NT == 0), "the all-wave scan reads the staged gather indices, EW_PF_K / NT per lane");
    const int ww = ROTW ? (b & (NW - 1)) : 0;          // the serial section's wave
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
    uint32_t *nbmask = reinterpret_cast<uint32_t *>(g + W + EW_G_EXT);  // W bits: neighbour set (static LlamaGen/Anole: zeroing hits q); not allocated when COMPACT
    SH &S = *reinterpret_cast<SH *>(reinterpret_cast<char *>(g) + epw_shared_offset(W, !(COMPACT || LITE)));
    int *const Scand = reinterpret_cast<int *>(reinterpret_cast<char *>(&S) + sizeof(SH));
    const int pd_cap = epw_pd_cap(Ps, Ds);
    int *const Srow = Scand + pd_cap, *const Spidx = Srow + pd_cap, *const Sboff = Spidx + pd_cap;
    float *const Scart = reinterpret_cast<float *>(Sboff + pd_cap);
    int *const Sflag = reinterpret_cast<int *>(Scart + pd_cap);       // per (path, depth): bit 1 image token, bit 0 syntax token
    int *const Shist = Sflag + pd_cap;                                // RAW: the radix-select histograms of the row post-process; logit rows with top_p: its 256 f64 mass bins
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
    const bool hot_in_lds = SL ? true : p_rows <= MAX_N;
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
    const bool root_pre = RAW && !FUSED && raw_p && win.raw_pre && win.raw_pre[0] != 0;     // (the level-1 row is requested before the tables are staged)
    // FUSED: is row `rid` of raw_probs published for this step?  One agent-scope load by thread 0, the answer through LDS (workgroup-uniform)
    auto row_ready = [&](int rid) -> bool {
        if constexpr (FUSED) {
            int *flag = reinterpret_cast<int *>(g) + W + 3;          // (the fourth word of g's extension: no gather target)
            if (tid == 0) *flag = (__hip_atomic_load(fz->ready + (size_t)b * p_rows + rid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz->epoch) ? 1 : 0;
            __syncthreads();
            return *flag != 0;          // (the row is then read with agent-scope atomic loads, row_load_agent: no fence, no cache to invalidate)
        } else return true;
    };
    bool rp_probs = false;       // what rp holds: probabilities of a pre-processed row, or raw cond / uncond chunks
    const int32_t *hot_g = win.row_hot ? win.row_hot + (size_t)b * p_rows : nullptr;
    const int ucur0 = buf.cursor ? buf.cursor[b] : 0;
    float4 rp[E4];              // prefetched row (registers) and the row id it holds
    int rp_rid = -1;
    {
        constexpr int PD_PER = ((COMPACT ? 96 : EW_MAX_PD) + NT - 1) / NT, B_PER = (MAX_B + NT - 1) / NT, N_PER = (MAX_N + NT - 1) / NT;
        const int npd = Ps * Ds;
        const int64_t *cand_g = buf.cand + (size_t)b * npd;
        const int32_t *row_g = buf.row_index + (prm.row_index_per_seq ? (size_t)b * npd : 0);
        const int nb_total = is_static ? buf.b_off[npd] : 0;
        const int rid1 = row_g[0];          // level 1: every path shares the root, the first matching path is path 0
        int64_t c_[PD_PER];
        int r_[PD_PER], pi_[PD_PER], bo_[PD_PER], tc_[N_PER], hot_[N_PER], oo_ = 0;
        float ct_[PD_PER];
        if constexpr (FUSED) {          // the candidate assembly of this sequence (prep_rows_body's arithmetic), kept in registers and written out for the commit launch
            const PrepArgs &pa = fz->prep;
            const int n_flat = pa.n_flat;
            const int64_t *tok = pa.ss_token + (size_t)b * n_flat;
            const float *prb = pa.ss_prob ? pa.ss_prob + (size_t)b * n_flat : nullptr;
            const int64_t st = pa.sample_token[b];
#pragma unroll
            for (int u = 0; u < PD_PER; ++u) {
                const int t = tid + u * NT;
                const bool in = t < npd;
                const int64_t r = in ? pa.retrieve[t] : -1;
                int64_t c = -1;
                float pr = 1.0f;
                if (r >= 0 && r < pa.N) {
                    const int64_t ti = pa.tree_indices[r];
                    const bool root = ti <= 0 || ti > n_flat;
                    c = root ? st : tok[ti - 1];
                    if (prb) pr = root ? 1.0f : prb[ti - 1];
                }
                c_[u] = in ? c : 0;
                ct_[u] = in ? pr : 0.0f;
                r_[u] = in ? row_g[t] : 0;
                pi_[u] = in ? buf.p_idx[t] : 0;
                bo_[u] = in ? buf.b_off[t] : 0;
                if (in) {
                    pa.cand[(size_t)b * npd + t] = c;
                    if (pa.cart_prob) pa.cart_prob[(size_t)b * npd + t] = pr;
                }
            }
        } else {
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
        }
#pragma unroll
        for (int u = 0; u < N_PER; ++u) {
            const int t = tid + u * NT;
            if constexpr (FUSED) {
                const PrepArgs &pa = fz->prep;
                tc_[u] = 0;
                if (t < p_N && t < MAX_N) {
                    const int64_t ti = pa.tree_indices[t];
                    const int64_t tc = (ti <= 0 || ti > pa.n_flat) ? pa.sample_token[b] : pa.ss_token[(size_t)b * pa.n_flat + ti - 1];
                    tc_[u] = (int)tc;
                    pa.tree_cand[(size_t)b * p_N + t] = tc;
                }
            } else
            tc_[u] = (is_static && t < p_N && t < MAX_N) ? (int)buf.tree_cand[(size_t)b * p_N + t] : 0;
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
        if (tid < N_UNI && ucur0 + tid < prm.n_uniforms) un_ = uni[ucur0 + tid];
        int bi_[B_PER];
#pragma unroll
        for (int u = 0; u < B_PER; ++u) {
            const int t = tid + u * NT;
            bi_[u] = (t < nb_total && t < MAX_B) ? buf.b_idx[t] : 0;
        }
        if (rid1 >= 0 && rid1 < p_rows) {
            if constexpr (RAW) {
                rp_probs = root_pre && rid1 == 0;
                if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid1 * W, W, rp);
                else raw_row_load<NT, CH>(raw_c + (size_t)rid1 * V, raw_u + (size_t)rid1 * V, rp);
            } else if constexpr (!DMAROW) row_load<NT, E4, FULLW>(logits + (size_t)rid1 * W, W, rp);          // (DMAROW: by LDS-DMA at level 1, once the row's class is known)
            if constexpr (!DMAROW) rp_rid = rid1;
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
                if (t < nb_total && t < MAX_B) S.bidx[t] = (unsigned short)bi_[u];
            }
#pragma unroll
            for (int u = 0; u < N_PER; ++u) {
                const int t = tid + u * NT;
                if (t < p_N && t < MAX_N) S.tcand[t] = tc_[u];
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
        if (tid < N_UNI) S.uni[tid] = un_;
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
        constexpr int PF_LIST = COMPACT ? 2 : (SPEC == 2 ? 4 : EW_PF_C);          // candidates per level staged ahead (the others restage a slot when their turn comes)
        // position t of a neighbour list -> index into g for the scan (out_tok is final before the ids are staged)
        auto plain_addr = [&](int id) -> unsigned short {
            // the Chameleon configurations: the table's ids ARE window indices (ids 0..8191 + offset 4 = the window [4, 8196)): no range test, no
            // out-of-window token can be a neighbour (masked, so that a corrupt table entry still lands inside g)
            if constexpr (SL && !S_LG) return (unsigned short)(id & 8191);
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
            if constexpr (DMAROW) {
                if (hot < 0) row_dma_to_lds<NT, E4>(logits + (size_t)rid * W, g);          // (g is dead here: every reader of the previous level passed its decision barrier)
            } else if (hot < 0 && rp_rid != rid) {
                if constexpr (RAW) {
                    rp_probs = raw_p && S.pre[rid] != 0;
                    if (FUSED && rp_probs) rp_probs = row_ready(rid);
                    if (FUSED && rp_probs) row_load_agent<NT, E4>(raw_p + (size_t)rid * W, rp);
                    else if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid * W, W, rp);
                    else raw_row_load<NT, CH>(raw_c + (size_t)rid * V, raw_u + (size_t)rid * V, rp);
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
                if (!rp_probs) raw_row_to_lds<NT, decltype(stage_ids), NUCLEUS, CH>(rp, hot, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, Shist, ph, stage_ids, prm.top_p, S.redi);
                else row_softmax_to_lds<NT, E4, FULLW, decltype(stage_ids), false, SH>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, stage_ids);
            } else if constexpr (DMAROW) {
                if (hot < 0) {          // the row is landing in g by DMA: nothing to compute (probability rows are final)
                    out_tok = -1;
                    out_mass = 0.0f;
                    if (tid == 0) g[W + EW_G_OUT] = 0.0f;
                    stage_ids();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                } else row_softmax_to_lds<NT, E4, FULLW, decltype(stage_ids), false, SH>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, stage_ids);
            } else row_softmax_to_lds<NT, E4, FULLW, decltype(stage_ids), NUCLEUS, SH>(rp, hot, rows_probs, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, stage_ids, prm.top_p, reinterpret_cast<double *>(Shist));
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
            if (n_used >= N_UNI || ucur0 + n_used >= prm.n_uniforms) {
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
            int code_u = 0, m_u = 0;          // ALLW: the decision, computed by every wave
            if constexpr (ALLW) {
                float px = x_in ? g[x - lo] : (x == out_tok ? out_mass : 0.0f);
                bool scan = false;
                if (p_syntax && is_syn) px = 1.0f;
                else if (p_syntax && !in_img) px = 0.0f;
                else if (p_lantern) {
                    if (nb == nullptr) code_u = 3;          // LANTERN_ST_TABLE_OOB
                    else scan = true;
                }
                if (scan) {          // (workgroup-uniform: x and its flags are)
                    const float tau = prm.delta > 1.0 ? (float)(prm.delta - 1.0) * px : (float)prm.delta;
                    constexpr int NPL = EW_PF_K / NT;          // consecutive neighbours per lane: wave w holds positions [w * 64 * NPL, (w + 1) * 64 * NPL)
                    uint32_t wv[(NPL + 1) / 2];
                    if constexpr (NPL == 4) {
                        const uint2 a2 = *reinterpret_cast<const uint2 *>(&S.nbaddr[slot][tid * 4]);
                        wv[0] = a2.x; wv[1] = a2.y;
                    } else {
#pragma unroll
                        for (int c = 0; c < (NPL + 1) / 2; ++c) wv[c] = *reinterpret_cast<const uint32_t *>(&S.nbaddr[slot][tid * NPL + 2 * c]);
                    }
                    double v[NPL], loc = 0.0;
#pragma unroll
                    for (int c = 0; c < NPL; ++c) {
                        loc += (double)g[(c & 1) ? (wv[c >> 1] >> 16) : (wv[c >> 1] & 0xffffu)];
                        v[c] = loc;
                    }
                    const double excl = wave_scan_incl_dpp(dpp_mov<0x138>(loc));          // exclusive prefix inside the wave (see the one-wave form below)
                    double *xs = S.samp_tot + (n_tried & 1) * 16;                        // (the bonus draw's slots: free until the epilogue)
                    if (lane == 63) xs[wave] = excl + loc;
                    __syncthreads();
                    double offs = 0.0;
                    for (int w = 0; w < wave; ++w) offs += xs[w];                          // (wave-uniform trip count)
                    const double base = offs + excl;
                    float mx = NEG_INF;
#pragma unroll
                    for (int c = 0; c < NPL; ++c) {
                        const float cs = (float)(base + v[c]);
                        mx = (cs <= tau) ? cs : mx;      // cs is non-decreasing in c: the last ok one is the largest
                    }
                    mx = wave_max(mx);
                    if (lane == 0) xs[8 + wave] = (double)mx;
                    __syncthreads();
                    float best_cs = NEG_INF;
#pragma unroll
                    for (int w = 0; w < NW; ++w) best_cs = fmaxf(best_cs, (float)xs[8 + w]);
                    if (best_cs > NEG_INF) {
                        m_u = 1;
                        px = px + best_cs;
                    }
                } else {
                    __syncthreads();          // every wave has read g before a rejection's zeroing can touch it
                }
                if (code_u == 0) {
                    float qx = 1.0f;
                    bool skip = false;
                    if (is_static) {
                        qx = rdlane(cart_lane, j);
                        skip = qx <= 0.0f;
                    }
                    code_u = skip ? 0 : (((float)r <= px / qx) ? 1 : 2);
                }
            }
            // ---------------- serial section: one wave only (wave 0; ROTW: wave b % NW)
            if (!ALLW && wave == ww) {
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
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
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
                EPW_STAMPF(26);
            }
            if constexpr (!ALLW) __syncthreads();
            const int code = ALLW ? code_u : dec[0];
            const int m = ALLW ? m_u : dec[1];
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
                    const float4 gv = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(g)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (!LDS2P) gn[it] = gv;
                    loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                }
            } else {
                const int b0 = rdlane(b0_lane, j), b1 = rdlane(b1_lane, j);
                const int nsib = b1 - b0;
                if (nsib > EW_MAX_SIB || b1 > MAX_B) {       // beyond the staged tables: say so instead of truncating the list
                    status = LANTERN_ST_TREE_LIMIT;
                    break;
                }
                // window indices of the earlier siblings' tokens, straight from the staged tables (every thread reads the same
                // LDS words: broadcast, no barrier); the first four live in registers, longer sibling lists loop over LDS
                auto sib_at = [&](int t) -> int {
                    const int node = (b0 + t < MAX_B) ? (int)S.bidx[b0 + t] : 0;
                    const int tok = (node >= 0 && node < MAX_N) ? S.tcand[node] : -1;
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
                    for (int t = 0; t < 4; ++t)
                        if (t < nsib) zero_q_at(sib_r[t]);          // (uniform: a slot without a sibling costs a compare, not the owner search)
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
                // max(gtp - q, 0): the residual's LDS reads go out together (one wait, not one per chunk), then a packed subtract with the output
                // clamp per two elements (window_dev.h sub_clamp0: same bits as subtract + compare + select, a fifth of the instructions)
                if constexpr (LDS2P) {
#pragma unroll
                    for (int it = 0; it < E4; ++it) {
                        const int i4 = tid + it * NT;
                        if (FULLW || i4 * 4 < W) {
                            const float4 gv = sub_clamp0(reinterpret_cast<const float4 *>(g)[i4], q[it]);
                            reinterpret_cast<float4 *>(g)[i4] = gv;          // (this thread's own slots: no other thread reads them before the normalising pass)
                            loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                        }
                    }
                } else {
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    gn[it] = (FULLW || i4 * 4 < W) ? reinterpret_cast<const float4 *>(g)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int it = 0; it < E4; ++it) {
                    const int i4 = tid + it * NT;
                    if (FULLW || i4 * 4 < W) {
                        const float4 gv = sub_clamp0(gn[it], q[it]);
                        gn[it] = gv;
                        loc += (double)gv.x + (double)gv.y + (double)gv.z + (double)gv.w;
                    }
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
                    reinterpret_cast<float4 *>(g)[i4] = dg(LDS2P ? reinterpret_cast<const float4 *>(g)[i4] : gn[it]);          // (dynamic trees: g already holds the unnormalised residual)
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
        if constexpr (DMAROW) {
            if (hot < 0) row_dma_to_lds<NT, E4>(logits + (size_t)rid * W, g);
        } else if (hot < 0 && rp_rid != rid) {
            if constexpr (RAW) {
                rp_probs = raw_p && S.pre[rid] != 0;
                if (FUSED && rp_probs) rp_probs = row_ready(rid);
                if (FUSED && rp_probs) row_load_agent<NT, E4>(raw_p + (size_t)rid * W, rp);
                else if (rp_probs) row_load<NT, E4, FULLW>(raw_p + (size_t)rid * W, W, rp);
                else raw_row_load<NT, CH>(raw_c + (size_t)rid * V, raw_u + (size_t)rid * V, rp);
            } else row_load<NT, E4, FULLW>(logits + (size_t)rid * W, W, rp);
        }
        if constexpr (RAW) {
            if (!rp_probs) raw_row_to_lds<NT, NoHook, NUCLEUS, CH>(rp, hot, win.raw_cfg, win.raw_top_k, V, lo, W, g, out_tok, out_mass, S.redf, S.redd, Shist, ph, NoHook(), prm.top_p, S.redi);
            else row_softmax_to_lds<NT, E4, FULLW, NoHook, false, SH>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph);
        } else if constexpr (DMAROW) {
            if (hot < 0) {
                out_tok = -1;
                out_mass = 0.0f;
                if (tid == 0) g[W + EW_G_OUT] = 0.0f;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else row_softmax_to_lds<NT, E4, FULLW, NoHook, false, SH>(rp, hot, true, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph);
        } else row_softmax_to_lds<NT, E4, FULLW, NoHook, NUCLEUS, SH>(rp, hot, rows_probs, lo, W, prm.temperature, prm.top_k, V, g, out_tok, out_mass, S, ph, NoHook(), prm.top_p, reinterpret_cast<double *>(Shist));
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
        static_assert(E4 * NW <= 64 && E4 * NW <= (int)(sizeof(S.samp_tot) / sizeof(double)), "segment totals fit one wave and their LDS slots");
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
        if (ka->win.verdict_host) {          // the same verdict where the host polls it (pinned memory): record first, the ready word last
            volatile int32_t *vh = ka->win.verdict_host + (size_t)b * 16;
            const long long tk = (k_u_bonus && k_token && status == LANTERN_ST_OK) ? (long long)k_token[b] : -1ll;
            vh[0] = best; vh[1] = a - 1; vh[2] = n_levels; vh[3] = n_tried; vh[4] = n_rej; vh[5] = n_used; vh[6] = from_residual; vh[7] = status;
            vh[8] = (int32_t)(tk & 0xffffffffll); vh[9] = (int32_t)(tk >> 32);
            __threadfence_system();
            vh[10] = 1;
        }
    }
    // commit turn-taking (lantern_step_group.turn): everything is written; the workgroup's last act is to wait for its group's turn, so that the commit
    // launched behind this kernel on the stream starts then.  Bounded (~40 ms): the turn is scheduling, never correctness.
    if (ka->win.turn) {
        if (tid == 0) {
            const long long need = ka->win.turn_wait;
            for (int spins = 0; spins < 200000; ++spins) {
                if ((long long)__hip_atomic_load(ka->win.turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
                __builtin_amdgcn_s_sleep(16);
            }
        }
    }
    return (best << 8) | a;          // the verdict (uniform): best path, rows kept = accept_len + 1
}

template <int NT, int E4, int IDMODE, int WPE, bool FULLW = false, bool RAW = false, int SPEC = 0, int TPO = 0>
__global__ __launch_bounds__(NT, WPE) void epw_kernel(const EpwArgs args) {
    epw_body<NT, E4, IDMODE, WPE, FULLW, RAW, SPEC, TPO>(args, blockIdx.x);
}

// the chain launch with its prepare stage inside (EpwFused): workgroups [0, n_helpers) post-process the listed rows -- dispatched first, so they run before the
// sequence workgroups that will ask for their rows --, workgroups [n_helpers, n_helpers + B) are the sequences
template <int NT, int E4, int IDMODE, int WPE, bool FULLW, bool RAW, int SPEC, int TPO>
__global__ __launch_bounds__(NT, WPE) void epw_kernel_fused(const EpwArgs args, const EpwFused fz) {
    if ((int)blockIdx.x < fz.n_helpers) {
        epw_helper_row<NT, E4 / 2>(fz, (int)blockIdx.x);
        return;
    }
    epw_body<NT, E4, IDMODE, WPE, FULLW, RAW, SPEC, TPO | 1024>(args, (int)blockIdx.x - fz.n_helpers, &fz);
}

// One launch of the chain kernel: what the instance files need besides the argument block.
struct EpwLaunch {
    dim3 grid;
    size_t lds;
    hipStream_t st;
};
// epw_generic.hip: the argument-driven instances (SPEC 0, one workgroup per CU) by window width / id mode; nucleus: TopPLogitsWarper inside the
// kernel for rows that arrive as logits (LANTERN_ROWS_LOGITS, prm.top_p).  Returns false when no instance fits.
bool epw_launch_generic(int W, int idmode, bool nucleus, const EpwLaunch &l, const EpwArgs &args);
// epw_throughput.hip: the forms for more sequences per launch than CUs (several workgroups per CU).
enum EpwThroughput {
    EPW_TP_LUMINA_DEFAULT_TREE = 0, EPW_TP_LUMINA_STATIC, EPW_TP_LUMINA_DYNAMIC, EPW_TP_ANOLE_STATIC,      // 256 threads x 8 float4, three per CU, LATE_Q
                                                                                                             // (the default tree: COMPACT, four per CU)
    EPW_TP_512_DEFAULT_TREE, EPW_TP_512_PACKED, EPW_TP_512_ID0, EPW_TP_512_ID1, EPW_TP_512_ID2,            // 512 threads at 128 VGPRs, two per CU
    EPW_TP_RAW_GENERIC,                                                                                      // raw rows, 512 threads, two per CU
    EPW_TP_RAW_LUMINA_DEFAULT_TREE, EPW_TP_RAW_LUMINA_STATIC, EPW_TP_RAW_LUMINA_DYNAMIC, EPW_TP_RAW_ANOLE_STATIC,   // raw rows, fixed configurations
    EPW_TP_RAW_LLAMAGEN_DYNAMIC,                                                                             // LlamaGen's 16384-id window, raw rows
    EPW_TP_LLAMAGEN_DYNAMIC                                                                                  // LlamaGen's 16384-id window, probability rows: 512 threads x 8 float4, two per CU
};
bool epw_launch_throughput(int kind, const EpwLaunch &l, const EpwArgs &args);


}  // namespace lantern
