// verify_step.cpp -- lantern_verify_step: one verify step of G groups of sequences enqueued by ONE host call
// (include/lantern_hip.h).  Pure launch sequencing over the library's own entry points -- no kernel of its own.
//
// Reference: the body of EaLumina_mGPT.generate's decode loop, models/ea_model_lumina_mgpt.py:936-998.
#include "../../include/lantern_hip.h"

namespace lantern {
void set_error(const char *fmt, ...);
const char *last_error();
}

extern "C" int lantern_verify_step(const lantern_step_group *groups, int n_groups) {
    if (!groups || n_groups < 0) {
        lantern::set_error("verify_step: bad arguments");
        return LANTERN_E_INVALID;
    }
    // stage by stage across the groups: every stream receives its next kernel before any stream receives the one after
    // (the streams then advance side by side instead of one group running a whole step ahead of the others)
    int rc;
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.node_list && s.n_list > 0) {          // candidates + the likely rows in one launch
            rc = lantern_prepare_step(&s);
            if (rc) return rc;
            continue;
        }
        rc = lantern_gather_candidates(s.ss_token, s.ss_prob, s.sample_token, s.tree_indices, s.retrieve, s.B, s.n_flat, s.N, s.P, s.D,
                                       s.tree_cand, s.cand, s.cart_prob, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (!s.out_win || (s.node_list && s.n_list > 0)) continue;          // LANTERN_ROWS_RAW_BF16: evaluate_posterior post-processes the rows it visits itself
        rc = lantern_cfg_mask_topk_window(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent,
                                          s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k, s.seq_len, s.N, s.win_lo,
                                          s.win_len, s.out_win, s.row_hot, s.out_kind, s.temperature, s.top_p, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (s.fused_ws && s.slab_ptrs && !s.nodes)          // O8 + O9 + O10 in one launch
            rc = lantern_verify_accept(&s);
        else
            rc = s.nodes ? lantern_evaluate_posterior_nodes(&s.ep, &s.ep_buf, &s.ep_win, s.nodes, s.stream)
                         : lantern_evaluate_posterior_window(&s.ep, &s.ep_buf, &s.ep_win, s.stream);
        if (rc) return rc;
    }
    for (int g = 0; g < n_groups; ++g) {
        const lantern_step_group &s = groups[g];
        if (!s.slab_ptrs || (s.fused_ws && !s.nodes)) continue;
        rc = lantern_update_inference_inputs(s.slab_ptrs, s.slab_seq, s.slab_prev, s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d,
                                             s.retrieve, 0, s.P, s.D, s.ep_buf.best, s.ep_buf.accept_len, s.new_len, s.hidden,
                                             s.hid_elem_bytes, s.B, s.hid_groups, s.N, s.H, s.cand, s.out_hidden, s.accepted_tokens,
                                             s.stream);
        if (rc) return rc;
    }
    return LANTERN_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// lantern_step_launcher: the same launch sequence, enqueued by worker threads.
//
// A HIP kernel launch costs the calling thread several microseconds (kernarg copy, AQL packet, doorbell); a step of G groups
// is 3 G launches, so past a handful of groups ONE enqueuing thread, not the GPU, sets the step time.  The groups are
// independent sequences on independent streams: worker t enqueues the groups g = t (mod T), each from a private ring of
// argument-block COPIES (lantern_step_group embeds its O8 blocks by value, so one struct copy is a deep copy).  submit()
// returns as soon as the copies are queued; the caller may patch its blocks for the next step at once.  Stream order is
// kept because one stream is only ever fed by one worker.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>

namespace {

int enqueue_group(const lantern_step_group &s) {
    int rc;
    const bool prep = s.node_list && s.n_list > 0;
    if (prep)
        rc = lantern_prepare_step(&s);
    else
        rc = lantern_gather_candidates(s.ss_token, s.ss_prob, s.sample_token, s.tree_indices, s.retrieve, s.B, s.n_flat, s.N, s.P, s.D,
                                       s.tree_cand, s.cand, s.cart_prob, s.stream);
    if (rc) return rc;
    if (s.out_win && !prep) {
        rc = lantern_cfg_mask_topk_window(s.cond, s.uncond, s.dtype, s.B * s.N, s.V, s.cfg, s.model, s.pos_ids, s.pos_base, s.w_latent,
                                          s.h_latent, s.img_lo, s.img_hi, s.newline_id, s.eos_id, s.top_k, s.seq_len, s.N, s.win_lo,
                                          s.win_len, s.out_win, s.row_hot, s.out_kind, s.temperature, s.top_p, s.stream);
        if (rc) return rc;
    }
    if (s.fused_ws && s.slab_ptrs && !s.nodes) return lantern_verify_accept(&s);          // O8 + O9 + O10 in one launch
    rc = s.nodes ? lantern_evaluate_posterior_nodes(&s.ep, &s.ep_buf, &s.ep_win, s.nodes, s.stream)
                 : lantern_evaluate_posterior_window(&s.ep, &s.ep_buf, &s.ep_win, s.stream);
    if (rc) return rc;
    if (s.slab_ptrs)
        rc = lantern_update_inference_inputs(s.slab_ptrs, s.slab_seq, s.slab_prev, s.n_slabs, s.elem_bytes, s.outer, s.S_max, s.d,
                                             s.retrieve, 0, s.P, s.D, s.ep_buf.best, s.ep_buf.accept_len, s.new_len, s.hidden,
                                             s.hid_elem_bytes, s.B, s.hid_groups, s.N, s.H, s.cand, s.out_hidden, s.accepted_tokens,
                                             s.stream);
    return rc;
}

constexpr unsigned RING = 64;                 // argument blocks in flight per worker (power of two)

struct alignas(64) Worker {
    std::atomic<uint64_t> head{0};            // written by the submitting thread
    char pad0[56];
    std::atomic<uint64_t> tail{0};            // written by the worker
    char pad1[56];
    std::atomic<int> asleep{0};
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
    lantern_step_group ring[RING];
};

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

}  // namespace

struct lantern_step_launcher {
    int device = 0, n_threads = 0;
    std::atomic<int> stop{0};
    std::atomic<int> error{0};                // first non-zero return code of any enqueue
    std::mutex err_mu;
    char err_msg[512] = "";                   // ... and its message (the library's error text is per thread)
    Worker *workers = nullptr;

    void fail(int rc, const char *msg) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (error.load(std::memory_order_relaxed)) return;
        std::strncpy(err_msg, msg, sizeof(err_msg) - 1);
        error.store(rc, std::memory_order_release);
    }

    void run(int t) {
        Worker &w = workers[t];
        bool bound = false;
        unsigned idle = 0;
        for (;;) {
            const uint64_t tl = w.tail.load(std::memory_order_relaxed);
            if (w.head.load(std::memory_order_acquire) == tl) {
                if (stop.load(std::memory_order_acquire)) return;
                if (++idle < 20000) { cpu_relax(); continue; }          // ~0.2-0.5 ms of spinning between steps, then sleep
                std::unique_lock<std::mutex> lk(w.mu);
                w.asleep.store(1, std::memory_order_seq_cst);
                w.cv.wait_for(lk, std::chrono::milliseconds(50), [&] {
                    return w.head.load(std::memory_order_acquire) != tl || stop.load(std::memory_order_acquire);
                });
                w.asleep.store(0, std::memory_order_seq_cst);
                idle = 0;
                continue;
            }
            idle = 0;
            if (!bound) {                     // the device context is per thread
                if (hipSetDevice(device) != hipSuccess) fail(LANTERN_E_LAUNCH, "step_launcher: hipSetDevice failed on a worker thread");
                bound = true;
            }
            if (!error.load(std::memory_order_relaxed)) {
                const int rc = enqueue_group(w.ring[tl & (RING - 1)]);
                if (rc) fail(rc, lantern::last_error());
            }
            w.tail.store(tl + 1, std::memory_order_release);
        }
    }
};

extern "C" int lantern_step_launcher_create(int n_threads, int device, lantern_step_launcher **out) {
    if (!out || n_threads < 1 || n_threads > 64 || device < 0) {
        lantern::set_error("step_launcher_create: n_threads in [1, 64], device >= 0, out != NULL");
        return LANTERN_E_INVALID;
    }
    lantern_step_launcher *L = new (std::nothrow) lantern_step_launcher;
    if (!L) return LANTERN_E_INVALID;
    L->device = device;
    L->n_threads = n_threads;
    L->workers = new (std::nothrow) Worker[n_threads];
    if (!L->workers) { delete L; return LANTERN_E_INVALID; }
    for (int t = 0; t < n_threads; ++t) L->workers[t].th = std::thread([L, t] { L->run(t); });
    *out = L;
    return LANTERN_OK;
}

extern "C" int lantern_step_launcher_wait(lantern_step_launcher *L);

extern "C" int lantern_step_launcher_submit(lantern_step_launcher *L, const lantern_step_group *groups, int n_groups) {
    if (!L || (!groups && n_groups) || n_groups < 0) {
        lantern::set_error("step_launcher_submit: bad arguments");
        return LANTERN_E_INVALID;
    }
    if (L->error.load(std::memory_order_acquire)) return lantern_step_launcher_wait(L);          // surfaces the worker's message
    for (int g = 0; g < n_groups; ++g) {
        Worker &w = L->workers[g % L->n_threads];
        const uint64_t h = w.head.load(std::memory_order_relaxed);
        while (h - w.tail.load(std::memory_order_acquire) >= RING) cpu_relax();          // ring full: the worker is RING blocks behind
        std::memcpy(&w.ring[h & (RING - 1)], &groups[g], sizeof(lantern_step_group));
        w.head.store(h + 1, std::memory_order_seq_cst);
        if (w.asleep.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(w.mu);
            w.cv.notify_one();
        }
    }
    return LANTERN_OK;
}

extern "C" int lantern_step_launcher_wait(lantern_step_launcher *L) {
    if (!L) {
        lantern::set_error("step_launcher_wait: NULL launcher");
        return LANTERN_E_INVALID;
    }
    for (int t = 0; t < L->n_threads; ++t) {
        Worker &w = L->workers[t];
        const uint64_t h = w.head.load(std::memory_order_acquire);
        unsigned spins = 0;
        while (w.tail.load(std::memory_order_acquire) < h)
            if (++spins < 4000) cpu_relax(); else std::this_thread::yield();
    }
    std::lock_guard<std::mutex> lk(L->err_mu);
    const int e = L->error.exchange(0);
    if (e) lantern::set_error("step_launcher (worker thread): %s", L->err_msg);
    return e;
}

extern "C" void lantern_step_launcher_destroy(lantern_step_launcher *L) {
    if (!L) return;
    L->stop.store(1, std::memory_order_release);
    for (int t = 0; t < L->n_threads; ++t) {
        {
            std::lock_guard<std::mutex> lk(L->workers[t].mu);
            L->workers[t].cv.notify_one();
        }
        if (L->workers[t].th.joinable()) L->workers[t].th.join();
    }
    delete[] L->workers;
    delete L;
}
