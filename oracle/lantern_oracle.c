/*
 * lantern_oracle.c -- plain-C CPU restatement of the reference's verify/accept
 * hot path (jadohu/LANTERN, Python/PyTorch).  TEST INFRASTRUCTURE ONLY: see the
 * header of lantern_oracle.h for who may call this.
 *
 * Parity status: PINNED.  the .npz files under tests/golden hold inputs/outputs captured from
 * the reference's own functions (EaLumina_mGPT.evaluate_posterior,
 * EaModel.evaluate_posterior[_v1], generate_tree_buffers, utils_c.generate_tree_buffers,
 * Model.topK_genrate tail, generate_candidates, KV copy, logits processors) run
 * in the build container; tests/test_oracle_golden.py checks every function here
 * against them (integers bit-exact, probabilities <= 1e-6).
 *
 * Numerics follow torch-CPU where it decides an integer outcome:
 *   - torch.cumsum(float32) accumulates in double and rounds each prefix to f32
 *     (verified in-container) -> cumsum_f32().
 *   - `r <= acp` compares a Python double against a 0-dim f32 tensor: the scalar
 *     is cast to f32 first -> (float)r <= acp.
 *   - `(lantern_delta - 1) * px` = f32(double(delta-1)) * px in f32.
 */
#include "lantern_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int lo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ helpers */

static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* round-to-nearest-even f32 -> bf16 (torch's c10::BFloat16 conversion). */
static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0; /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

static double sum_f32(const float *x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)x[i];
    return s;
}

/* Per-thread scratch: the routines below need a few V-sized work rows per call.  malloc()ing them per call (256 KB each:
 * mmap + munmap + first-touch page faults every time) made 63 concurrent baseline threads spend most of their time in the
 * kernel's address-space lock -- a thread's step took 4.6x as long as the same step on an idle host.  Buffers are kept per
 * thread and slot, grown on demand, freed when the thread exits (pthread key destructor). */
#include <pthread.h>
#define LO_SCRATCH_SLOTS 12
typedef struct {
    void *buf[LO_SCRATCH_SLOTS];
    size_t cap[LO_SCRATCH_SLOTS];
} lo_scratch_t;
static pthread_key_t lo_scratch_key;
static pthread_once_t lo_scratch_once = PTHREAD_ONCE_INIT;
static void lo_scratch_free(void *p) {
    lo_scratch_t *s = (lo_scratch_t *)p;
    if (!s) return;
    for (int i = 0; i < LO_SCRATCH_SLOTS; ++i) free(s->buf[i]);
    free(s);
}
static void lo_scratch_make_key(void) { pthread_key_create(&lo_scratch_key, lo_scratch_free); }
static void *lo_scratch(int slot, size_t bytes) {
    pthread_once(&lo_scratch_once, lo_scratch_make_key);
    lo_scratch_t *s = (lo_scratch_t *)pthread_getspecific(lo_scratch_key);
    if (!s) {
        s = (lo_scratch_t *)calloc(1, sizeof(lo_scratch_t));
        if (!s) {
            fprintf(stderr, "lantern_oracle: out of memory (scratch table)\n");
            abort();
        }
        pthread_setspecific(lo_scratch_key, s);
    }
    if (s->cap[slot] < bytes) {
        free(s->buf[slot]);
        s->cap[slot] = 0;
        s->buf[slot] = malloc(bytes);
        if (!s->buf[slot]) { /* the checker must fail loudly, never hand a NULL row to the arithmetic */
            fprintf(stderr, "lantern_oracle: out of memory (%zu scratch bytes, slot %d)\n", bytes, slot);
            abort();
        }
        s->cap[slot] = bytes;
    }
    return s->buf[slot];
}

/* rows of one lo_cfg_mask_topk call shared over this many OpenMP threads of the calling thread's team (the all-core leg of the
 * CPU baseline: 63 sequences are fewer than a 256-core host has cores; a sequence's 26 tree rows are independent) */
#include <stdatomic.h>
static _Atomic int lo_row_threads = 1;
int lo_set_row_threads(int n) {
    const int v = n < 1 ? 1 : (n > 64 ? 64 : n);
    atomic_store(&lo_row_threads, v);
    return v;
}

/* torch.softmax(row, dim=0) for a float32 row (ea_model_lumina_mgpt.py:637). */
static void softmax_row(const float *x, int n, float *out) {
    float m = -INFINITY;
    for (int i = 0; i < n; ++i)
        if (x[i] > m) m = x[i];
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        float e = expf(x[i] - m);
        out[i] = e;
        s += (double)e;
    }
    float sf = (float)s;
    for (int i = 0; i < n; ++i) out[i] = out[i] / sf;
}

/* k-th largest value of x[0..n) (torch.topk(x,k)[0][-1]); k>=1.  Quickselect on a copy
 * (O(n) expected) so that the CPU baseline is not handicapped by a full sort. */
static float kth_largest(const float *x, int n, int k) {
    float *t = (float *)lo_scratch(0, sizeof(float) * (size_t)n);
    memcpy(t, x, sizeof(float) * (size_t)n);
    int lo = 0, hi = n - 1, target = k - 1; /* index in descending order */
    while (lo < hi) {
        /* median of three as pivot */
        int mid = lo + (hi - lo) / 2;
        float a = t[lo], b = t[mid], c = t[hi];
        float pv = (a > b) ? ((b > c) ? b : (a > c ? c : a)) : ((a > c) ? a : (b > c ? c : b));
        int i = lo, j = hi;
        while (i <= j) {
            while (t[i] > pv) ++i;
            while (t[j] < pv) --j;
            if (i <= j) {
                float tmp = t[i];
                t[i] = t[j];
                t[j] = tmp;
                ++i;
                --j;
            }
        }
        if (target <= j)
            hi = j;
        else if (target >= i)
            lo = i;
        else
            break;
    }
    return t[target];
}

typedef struct {
    float v;
    int i;
} fi_pair;
static int cmp_pair_asc(const void *a, const void *b) {
    const fi_pair *pa = (const fi_pair *)a, *pb = (const fi_pair *)b;
    if (pa->v < pb->v) return -1;
    if (pa->v > pb->v) return 1;
    return (pa->i > pb->i) - (pa->i < pb->i);
}

/* HF LogitsProcessorList built by prepare_logits_processor (drafters/utils.py:36-52):
 * Temperature -> TopP -> TopK, applied to one row in place. */
static void apply_processors(float *row, int V, float temperature, float top_p, int top_k) {
    if (temperature > 1e-5f && temperature != 1.0f)
        for (int i = 0; i < V; ++i) row[i] = row[i] / temperature;
    if (top_p >= 1e-8f && top_p < 1.0f) {
        /* TopPLogitsWarper: ascending sort, softmax, cumsum, remove cum <= 1-top_p,
         * always keep the last (largest) entry. */
        fi_pair *pr = (fi_pair *)lo_scratch(1, sizeof(fi_pair) * (size_t)V);
        float *sv = (float *)lo_scratch(2, sizeof(float) * (size_t)V);
        float *sp = (float *)lo_scratch(3, sizeof(float) * (size_t)V);
        for (int i = 0; i < V; ++i) {
            pr[i].v = row[i];
            pr[i].i = i;
        }
        qsort(pr, (size_t)V, sizeof(fi_pair), cmp_pair_asc);
        for (int i = 0; i < V; ++i) sv[i] = pr[i].v;
        softmax_row(sv, V, sp);
        float thr = (float)(1.0 - (double)top_p);
        double acc = 0.0;
        for (int i = 0; i < V - 1; ++i) {
            acc += (double)sp[i];
            if ((float)acc <= thr) row[pr[i].i] = -INFINITY;
        }
    }
    if (top_k > 0) {
        int k = top_k < V ? top_k : V;
        float thr = kth_largest(row, V, k);
        for (int i = 0; i < V; ++i)
            if (row[i] < thr) row[i] = -INFINITY;
    }
}

/* ------------------------------------------------- O8: evaluate_posterior */
/*
 * Restates the sampling branch of
 *   EaLumina_mGPT.evaluate_posterior      models/ea_model_lumina_mgpt.py:610-726
 *   EaModel.evaluate_posterior (dynamic)  models/ea_model_llamagen.py:709-787, ea_model_anole.py:709-788
 *   EaModel.evaluate_posterior_v1 (static) models/ea_model_llamagen.py:597-669
 * selected by prm->mode / prm->syntax_shortcut / prm->tok_offset.
 */
int lo_evaluate_posterior(const lo_ep_params *prm, const float *logits, const int32_t *row_index,
                          const int64_t *cand, const float *cart_prob, const float *orig_prob,
                          const int32_t *op_off, const int32_t *p_idx, const int32_t *b_off,
                          const int32_t *b_idx, const int64_t *tree_cand, const uint16_t *nn_table,
                          const double *uniforms, int32_t n_uniforms, int32_t *best_out,
                          int32_t *accept_len_out, float *sample_p, int32_t *counters) {
    const int P = prm->P, D = prm->D, V = prm->V;
    const int is_static = prm->mode != LO_MODE_DYNAMIC;
    const int k = prm->k, off = prm->tok_offset;
    if (P <= 0 || D <= 0 || V <= 0) return -1;
    if (is_static && (!cart_prob || !orig_prob || !op_off || !p_idx || !b_off || !tree_cand)) return -2;
    if (prm->lantern && (!nn_table || k < 1 || k > prm->table_cols)) return -3;

    float *g = (float *)lo_scratch(4, sizeof(float) * (size_t)V);
    float *q = (float *)lo_scratch(5, sizeof(float) * (size_t)V);
    float *row = (float *)lo_scratch(6, sizeof(float) * (size_t)V);
    float *cs = (float *)lo_scratch(7, sizeof(float) * (size_t)(k > 0 ? k : 1));
    int64_t *acc_tok = (int64_t *)lo_scratch(8, sizeof(int64_t) * (size_t)D);
    int64_t *tried = (int64_t *)lo_scratch(9, sizeof(int64_t) * (size_t)P);
    char *is_eq = (char *)lo_scratch(10, (size_t)P);
    int rc = 0;

    int a = 1, best = 0, adjust = 0, u = 0;
    int n_levels = 0, n_tried = 0, n_rej = 0;
    acc_tok[0] = cand[0];

    for (int i = 1; i < D; ++i) {
        if (i != a) break;
        adjust = 0;
        ++n_levels;
        int fi = -1;
        for (int j = 0; j < P; ++j) {
            int eq = 1;
            for (int t = 0; t < a; ++t)
                if (cand[(size_t)j * D + t] != acc_tok[t]) {
                    eq = 0;
                    break;
                }
            is_eq[j] = (char)eq;
            if (eq && fi < 0) fi = j;
        }
        if (fi < 0) {
            rc = -4;
            goto done;
        }
        memcpy(row, logits + (size_t)row_index[(size_t)fi * D + (i - 1)] * V, sizeof(float) * (size_t)V);
        apply_processors(row, V, prm->temperature, prm->top_p, prm->top_k);
        softmax_row(row, V, g);

        int nset = 0;
        for (int j = 0; j < P; ++j) {
            if (!is_eq[j]) continue;
            int64_t x = cand[(size_t)j * D + i];
            if (x == -1) continue;
            int dup = 0;
            for (int t = 0; t < nset; ++t)
                if (tried[t] == x) {
                    dup = 1;
                    break;
                }
            if (dup) continue;
            tried[nset++] = x;
            if (x < 0 || x >= V) {
                rc = -5;
                goto done;
            }
            if (u >= n_uniforms) {
                rc = -6;
                goto done;
            }
            double r = uniforms[u++];
            ++n_tried;

            float px = g[x];
            int m = 0; /* number of neighbours whose cumulative mass stays <= tau */
            int is_syn = 0, in_img = (x >= prm->img_lo && x < prm->img_hi);
            if (prm->syntax_shortcut)
                for (int t = 0; t < prm->n_syntax; ++t)
                    if (x == prm->syntax[t]) is_syn = 1;
            const uint16_t *nb = NULL;
            if (prm->syntax_shortcut && is_syn) {
                px = 1.0f;
            } else if (prm->syntax_shortcut && !in_img) {
                px = 0.0f;
            } else if (prm->lantern) {
                int64_t trow = x - off;
                if (trow < 0 || trow >= prm->table_rows) {
                    rc = -7;
                    goto done;
                }
                nb = nn_table + (size_t)trow * prm->table_cols;
                double accd = 0.0;
                for (int t = 0; t < k; ++t) {
                    accd += (double)g[(int)nb[t] + off];
                    cs[t] = (float)accd;
                }
                float tau = prm->delta > 1.0 ? (float)(prm->delta - 1.0) * px : (float)prm->delta;
                for (int t = 0; t < k; ++t)
                    if (cs[t] <= tau) m = t + 1;
                if (m > 0) px = px + cs[m - 1];
            }

            float qx = 1.0f;
            if (is_static) {
                qx = cart_prob[(size_t)j * D + i];
                if (qx <= 0.0f) continue;
            }
            float acp = px / qx;
            if ((float)r <= acp) {
                acc_tok[a] = x;
                ++a;
                best = j;
                break;
            }
            /* ---- rejection: residual distribution ---- */
            ++n_rej;
            if (prm->syntax_shortcut && is_syn) {
                rc = -8; /* reference asserts (ea_model_lumina_mgpt.py:694) */
                goto done;
            }
            if (is_static) {
                const float *qsrc = orig_prob + ((size_t)op_off[i - 1] + (size_t)p_idx[(size_t)j * D + i]) * V;
                memcpy(q, qsrc, sizeof(float) * (size_t)V);
                int b0 = b_off[(size_t)j * D + i], b1 = b_off[(size_t)j * D + i + 1];
                if (b1 > b0) {
                    for (int t = b0; t < b1; ++t) q[tree_cand[b_idx[t]]] = 0.0f;
                    float qs = (float)sum_f32(q, V);
                    for (int t = 0; t < V; ++t) q[t] = q[t] / qs;
                }
            }
            int zero_nb = prm->lantern && m > 0 && (!prm->syntax_shortcut || in_img);
            int nz = (k + 1 < prm->table_cols) ? k + 1 : prm->table_cols;
            if (prm->mode == LO_MODE_DYNAMIC) {
                g[x] = 0.0f;
                if (zero_nb)
                    for (int t = 0; t < nz; ++t) g[(int)nb[t] + off] = 0.0f;
            } else {
                if (zero_nb) {
                    float *tgt = (prm->mode == LO_MODE_STATIC_LUMINA) ? g : q;
                    for (int t = 0; t < nz; ++t) tgt[(int)nb[t] + off] = 0.0f;
                }
                for (int t = 0; t < V; ++t) {
                    float d = g[t] - q[t];
                    g[t] = d < 0.0f ? 0.0f : d;
                }
            }
            float gs = (float)sum_f32(g, V);
            if (gs == 0.0f) {
                for (int t = 0; t < V; ++t) g[t] = 1.0f;
                gs = (float)sum_f32(g, V);
            }
            for (int t = 0; t < V; ++t) g[t] = g[t] / gs;
            adjust = 1;
        }
    }

    if (adjust && a != D) {
        memcpy(sample_p, g, sizeof(float) * (size_t)V);
    } else {
        memcpy(row, logits + (size_t)row_index[(size_t)best * D + (a - 1)] * V, sizeof(float) * (size_t)V);
        apply_processors(row, V, prm->temperature, prm->top_p, prm->top_k);
        softmax_row(row, V, sample_p);
    }
    *best_out = best;
    *accept_len_out = a - 1;
    if (counters) {
        counters[0] = n_levels;
        counters[1] = n_tried;
        counters[2] = n_rej;
        counters[3] = u;
        counters[4] = (adjust && a != D) ? 1 : 0;
        counters[5] = 0;
    }
done:
    return rc;
}

/* ------------------------------------------ a9: greedy / TVD evaluate_posterior */
/*
 * Restates the `logits_processor is None` branch of EaModel.evaluate_posterior
 * (models/ea_model_llamagen.py:789-905; Anole ea_model_anole.py:790-905 adds the
 * image-token offset).  The float expression for tvd is kept term by term
 * (0.5*|px-(px+c)| + cumsum(0.5*nb)) so threshold decisions match.
 */
int lo_evaluate_posterior_greedy(int P, int D, int V, const float *logits, const int32_t *row_index,
                                 const int64_t *cand, int lantern, int k, double delta, int tok_offset,
                                 const uint16_t *nn_table, int table_rows, int table_cols,
                                 int32_t *best_out, int32_t *accept_len_out, float *out_row) {
    if (lantern && (!nn_table || k < 1 || k > table_cols)) return -3;
    int *alen = (int *)calloc((size_t)P, sizeof(int));
    float *g = (float *)malloc(sizeof(float) * (size_t)V);
    int rc = 0;
    for (int p = 0; p < P; ++p) {
        int run = 1, cnt = 0;
        for (int d = 0; d < D - 1; ++d) {
            int64_t x = cand[(size_t)p * D + d + 1];
            int valid = (x != -1);
            const float *row = logits + (size_t)row_index[(size_t)p * D + d] * V;
            int ok = 0;
            if (valid) {
                if (!lantern) {
                    int am = 0;
                    for (int t = 1; t < V; ++t)
                        if (row[t] > row[am]) am = t;
                    ok = (am == x);
                } else {
                    softmax_row(row, V, g);
                    float px = g[x];
                    int64_t trow = x - tok_offset;
                    if (trow < 0 || trow >= table_rows) {
                        rc = -7;
                        goto done;
                    }
                    const uint16_t *nb = nn_table + (size_t)trow * table_cols;
                    double acc_c = 0.0, acc_t = 0.0;
                    int m = 0;
                    float px_adj = px;
                    for (int t = 0; t < k; ++t) {
                        float nbp = g[(int)nb[t] + tok_offset];
                        acc_c += (double)nbp;
                        float c = (float)acc_c;          /* cumsum_nearest_probs */
                        float approx = px + c;           /* approx_p */
                        float tvd_px = 0.5f * fabsf(px - approx);
                        float tvd_nb = 0.5f * fabsf(nbp - 0.0f);
                        acc_t += (double)tvd_nb;
                        float tvd = tvd_px + (float)acc_t;
                        float tau = delta > 1.0 ? (float)(delta - 1.0) * px : (float)delta;
                        if (tvd <= tau) {
                            m = t + 1;
                            px_adj = approx;
                        }
                    }
                    (void)m;
                    g[x] = px_adj;
                    int am = 0;
                    for (int t = 1; t < V; ++t)
                        if (g[t] > g[am]) am = t;
                    ok = (am == x);
                }
            }
            run = run && ok;
            cnt += run;
        }
        alen[p] = cnt;
    }
    {
        int mx = 0, arg = 0;
        for (int p = 0; p < P; ++p)
            if (alen[p] > mx) {
                mx = alen[p];
                arg = p;
            }
        *accept_len_out = mx;
        *best_out = (mx == 0) ? 0 : arg;
        memcpy(out_row, logits + (size_t)row_index[(size_t)(*best_out) * D + mx] * V, sizeof(float) * (size_t)V);
    }
done:
    free(alen);
    free(g);
    return rc;
}

/* --------------------------------------------------- O7: cfg + mask + top-k */
/*
 * Restates the logit post-processing of tree_decoding:
 *   Lumina  ea_model_lumina_mgpt.py:597-605 (CFG, MultiModalLogitsProcessor :45-86,
 *           InterleavedTopKLogitsWarper :106-112)
 *   Anole   ea_model_anole.py:930-931 (CFG, non-image -> finfo.min)
 *   LlamaGen ea_model_llamagen.py:930 (CFG only; cfg_logit_process :26-29)
 * bf16 inputs reproduce torch's per-op bf16 rounding of `u + s*(c-u)`.
 * pos_ids[n] is the value the reference passes as `position_ids=` (already +1);
 * num_generated_image_tokens = pos_ids[n] - pos_base.
 */
static int64_t py_mod(int64_t a, int64_t b) {
    int64_t r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? r + b : r;
}

int lo_cfg_mask_topk(const void *cond, const void *uncond, int dtype, int N, int V, float cfg,
                     int model, const int64_t *pos_ids, int64_t pos_base, int w_latent, int h_latent,
                     int img_lo, int img_hi, int newline_id, int eos_id, int top_k, float *out) {
    const int row_threads = atomic_load(&lo_row_threads);
#pragma omp parallel for schedule(dynamic, 1) num_threads(row_threads) if (row_threads > 1)
    for (int n = 0; n < N; ++n) {
        float *o = out + (size_t)n * V;
        for (int v = 0; v < V; ++v) {
            float c, u;
            if (dtype == LO_BF16) {
                c = bf16_to_f32(((const uint16_t *)cond)[(size_t)n * V + v]);
                u = bf16_to_f32(((const uint16_t *)uncond)[(size_t)n * V + v]);
                float t = round_bf16(c - u);
                t = round_bf16(cfg * t);
                o[v] = round_bf16(u + t);
            } else {
                c = ((const float *)cond)[(size_t)n * V + v];
                u = ((const float *)uncond)[(size_t)n * V + v];
                float t = c - u;
                t = cfg * t;
                o[v] = u + t;
            }
        }
        if (model == LO_MODEL_LUMINA) {
            int64_t n1 = pos_ids[n] - pos_base + 1;
            int64_t wrap = py_mod(n1, (int64_t)w_latent + 1);
            if (n1 == ((int64_t)w_latent + 1) * h_latent + 1) {
                for (int v = 0; v < V; ++v) o[v] = -INFINITY;
                o[eos_id] = 0.0f;
            } else if (wrap == 0) {
                for (int v = 0; v < V; ++v) o[v] = -INFINITY;
                o[newline_id] = 0.0f;
            } else {
                for (int v = 0; v < V; ++v)
                    if (v < img_lo || v >= img_hi) o[v] = -INFINITY;
            }
            if (top_k > 0) {
                int kk = top_k < V ? top_k : V;
                float thr = kth_largest(o, V, kk);
                for (int v = 0; v < V; ++v)
                    if (o[v] < thr) o[v] = -INFINITY;
            }
        } else if (model == LO_MODEL_ANOLE) {
            float mn = (dtype == LO_BF16) ? bf16_to_f32(0xff7f) : -3.4028234663852886e38f;
            for (int v = 0; v < V; ++v)
                if (v < img_lo || v >= img_hi) o[v] = mn;
        }
    }
    return 0;
}

/* -------------------------------------------------- O1: static target tree */
/*
 * Restates generate_tree_buffers (models/ea_model_lumina_mgpt.py:140-277; method
 * copies ea_model_llamagen.py:283-420, ea_model_anole.py:280-417).
 */
typedef struct {
    int len;
    const int32_t *p;
    int orig;
} lo_path;

static int cmp_path(const void *a, const void *b) {
    const lo_path *x = (const lo_path *)a, *y = (const lo_path *)b;
    if (x->len != y->len) return x->len - y->len;
    for (int i = 0; i < x->len; ++i)
        if (x->p[i] != y->p[i]) return x->p[i] - y->p[i];
    return 0;
}

static int find_prefix(const lo_path *s, int n, const int32_t *p, int len) {
    for (int i = 0; i < n; ++i)
        if (s[i].len == len && memcmp(s[i].p, p, sizeof(int32_t) * (size_t)len) == 0) return i;
    return -1;
}

static lo_path *sorted_paths(const int32_t *choices, const int32_t *choice_off, int n) {
    lo_path *s = (lo_path *)malloc(sizeof(lo_path) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        s[i].len = choice_off[i + 1] - choice_off[i];
        s[i].p = choices + choice_off[i];
        s[i].orig = i;
    }
    qsort(s, (size_t)n, sizeof(lo_path), cmp_path);
    return s;
}

/* leaf rows in the reference's discovery order (sorted choices walked from the end,
 * skipping anything already covered as a prefix). Returns count; rows[r] = sorted idx. */
static int leaf_rows(const lo_path *s, int n, int *rows) {
    char *covered = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int cnt = 0;
    for (int i = n - 1; i >= 0; --i) {
        if (covered[i]) continue;
        rows[cnt++] = i;
        for (int c = 1; c <= s[i].len; ++c) {
            int id = find_prefix(s, n, s[i].p, c);
            if (id >= 0) covered[id] = 1;
        }
    }
    free(covered);
    return cnt;
}

int lo_tree_static_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices, int *N,
                         int *P, int *D, int *b_total) {
    lo_path *s = sorted_paths(choices, choice_off, n_choices);
    int *rows = (int *)malloc(sizeof(int) * (size_t)(n_choices > 0 ? n_choices : 1));
    int maxlen = 0;
    for (int i = 0; i < n_choices; ++i)
        if (s[i].len > maxlen) maxlen = s[i].len;
    int p = leaf_rows(s, n_choices, rows);
    /* b list of a node = earlier siblings; per retrieve cell */
    int bt = 0;
    for (int r = 0; r < p; ++r) {
        int leaf = rows[r];
        for (int c = 1; c <= s[leaf].len; ++c) {
            int id = find_prefix(s, n_choices, s[leaf].p, c);
            for (int t = 0; t < id; ++t)
                if (s[t].len == c && memcmp(s[t].p, s[id].p, sizeof(int32_t) * (size_t)(c - 1)) == 0) ++bt;
        }
    }
    *N = n_choices + 1;
    *P = p;
    *D = maxlen + 1;
    *b_total = bt;
    free(s);
    free(rows);
    return 0;
}

typedef struct {
    int64_t key[64];
    int64_t val[64];
    int D;
} lo_rrow;
static int cmp_rrow(const void *a, const void *b) {
    const lo_rrow *x = (const lo_rrow *)a, *y = (const lo_rrow *)b;
    for (int i = 0; i < x->D; ++i)
        if (x->key[i] != y->key[i]) return x->key[i] < y->key[i] ? -1 : 1;
    return 0;
}

int lo_tree_static_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k,
                         float *mask, int64_t *tree_indices, int64_t *pos_ids, int64_t *retrieve,
                         int32_t *p_idx_out, int32_t *b_off, int32_t *b_idx) {
    const int n = n_choices, N = n + 1;
    lo_path *s = sorted_paths(choices, choice_off, n);
    int maxlen = 0;
    for (int i = 0; i < n; ++i)
        if (s[i].len > maxlen) maxlen = s[i].len;
    const int D = maxlen + 1;
    if (D > 64) {
        free(s);
        return -1;
    }
    /* attention mask */
    memset(mask, 0, sizeof(float) * (size_t)N * N);
    for (int i = 0; i < N; ++i) {
        mask[(size_t)i * N + i] = 1.0f;
        mask[(size_t)i * N] = 1.0f;
    }
    for (int i = 0; i < n; ++i)
        for (int c = 1; c < s[i].len; ++c) {
            int id = find_prefix(s, n, s[i].p, c);
            if (id < 0) {
                free(s);
                return -2; /* reference raises ValueError from list.index */
            }
            mask[(size_t)(i + 1) * N + (id + 1)] = 1.0f;
        }
    /* tree_indices, per-node in-layer parent ordinal, earlier-sibling lists */
    int *p_node = (int *)calloc((size_t)N, sizeof(int));
    tree_indices[0] = 0;
    pos_ids[0] = 0;
    p_node[0] = -1;
    int bias = 0, start = 0;
    for (int depth = 1; depth <= maxlen; ++depth) {
        int cnt = 0;
        while (start + cnt < n && s[start + cnt].len == depth) ++cnt;
        int inlayer = 0;
        for (int j = 0; j < cnt; ++j) {
            const lo_path *cur = &s[start + j];
            if (j != 0) {
                const lo_path *prv = &s[start + j - 1];
                if (memcmp(cur->p, prv->p, sizeof(int32_t) * (size_t)(depth - 1)) != 0) {
                    ++bias;
                    ++inlayer;
                }
            }
            tree_indices[start + j + 1] = cur->p[depth - 1] + (int64_t)top_k * (depth - 1 + bias) + 1;
            p_node[start + j + 1] = inlayer;
            pos_ids[start + j + 1] = depth;
        }
        start += cnt;
    }
    /* retrieve_indices */
    int *rows = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int P = leaf_rows(s, n, rows);
    lo_rrow *rr = (lo_rrow *)malloc(sizeof(lo_rrow) * (size_t)(P > 0 ? P : 1));
    int64_t maxitem = (int64_t)n + 5; /* retrieve.max() + 5, max id is n */
    {
        int64_t mx = 0;
        for (int r = 0; r < P; ++r) {
            rr[r].D = D;
            rr[r].val[0] = 0;
            for (int c = 1; c < D; ++c) {
                if (c <= s[rows[r]].len)
                    rr[r].val[c] = find_prefix(s, n, s[rows[r]].p, c) + 1;
                else
                    rr[r].val[c] = -1;
                if (rr[r].val[c] > mx) mx = rr[r].val[c];
            }
        }
        maxitem = mx + 5;
        for (int r = 0; r < P; ++r)
            for (int c = 0; c < D; ++c) rr[r].key[c] = rr[r].val[c] >= 0 ? rr[r].val[c] : maxitem;
    }
    qsort(rr, (size_t)P, sizeof(lo_rrow), cmp_rrow);
    int bt = 0;
    for (int r = 0; r < P; ++r)
        for (int c = 0; c < D; ++c) {
            int64_t id = rr[r].val[c];
            retrieve[(size_t)r * D + c] = id;
            /* torch indexing with -1 wraps to the last node (ea_model_lumina_mgpt.py:238) */
            p_idx_out[(size_t)r * D + c] = p_node[id >= 0 ? id : N - 1];
            b_off[(size_t)r * D + c] = bt;
            if (id > 0) {
                int si = (int)id - 1, len = s[si].len;
                for (int t = 0; t < si; ++t)
                    if (s[t].len == len && memcmp(s[t].p, s[si].p, sizeof(int32_t) * (size_t)(len - 1)) == 0)
                        b_idx[bt++] = t + 1;
            }
        }
    b_off[(size_t)P * D] = bt;
    free(rows);
    free(rr);
    free(p_node);
    free(s);
    return 0;
}

/* ------------------------------------------------- O2: drafter-side static tree */
/*
 * Restates drafters/utils_c.py:35-179 (Tree/node + generate_tree_buffers): buffers
 * over the NON-LEAF nodes only.  level l holds the non-leaf nodes of depth l+1.
 * masks_concat: for each level, [n_l, cum_l] row-major, concatenated.
 */
static int has_child(const lo_path *s, int n, int i) {
    for (int t = 0; t < n; ++t)
        if (s[t].len == s[i].len + 1 && memcmp(s[t].p, s[i].p, sizeof(int32_t) * (size_t)s[i].len) == 0) return 1;
    return 0;
}

int lo_tree_drafter_sizes(const int32_t *choices, const int32_t *choice_off, int n_choices,
                          int *n_levels, int *level_counts) {
    lo_path *s = sorted_paths(choices, choice_off, n_choices);
    int maxlen = 0;
    for (int i = 0; i < n_choices; ++i)
        if (s[i].len > maxlen) maxlen = s[i].len;
    int L = maxlen - 1;
    for (int l = 0; l < L; ++l) level_counts[l] = 0;
    for (int i = 0; i < n_choices; ++i)
        if (has_child(s, n_choices, i)) level_counts[s[i].len - 1] += 1;
    *n_levels = L;
    free(s);
    return 0;
}

int lo_tree_drafter_build(const int32_t *choices, const int32_t *choice_off, int n_choices, int top_k,
                          float *masks_concat, int64_t *tree_indices_concat, int32_t *repeat_nums_concat,
                          int32_t *repeat_off) {
    const int n = n_choices;
    lo_path *s = sorted_paths(choices, choice_off, n);
    int *wc = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* sorted idx of non-leaf nodes */
    int *wc_index = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int nw = 0, maxlen = 0;
    for (int i = 0; i < n; ++i) {
        wc_index[i] = -1;
        if (s[i].len > maxlen) maxlen = s[i].len;
        if (has_child(s, n, i)) {
            wc_index[i] = nw;
            wc[nw++] = i;
        }
    }
    const int L = maxlen - 1;
    float *full = (float *)calloc((size_t)(nw > 0 ? nw * nw : 1), sizeof(float));
    for (int a = 0; a < nw; ++a) {
        full[(size_t)a * nw + a] = 1.0f;
        int i = wc[a];
        for (int c = 1; c <= s[i].len; ++c) {
            int id = find_prefix(s, n, s[i].p, c);
            full[(size_t)a * nw + wc_index[id]] = 1.0f;
        }
    }
    int start = 0, cum = 0, rp = 0;
    size_t mo = 0, to = 0;
    for (int l = 0; l < L; ++l) {
        int cnt = 0;
        while (start + cnt < nw && s[wc[start + cnt]].len == l + 1) ++cnt;
        cum += cnt;
        for (int a = 0; a < cnt; ++a)
            for (int b = 0; b < cum; ++b) masks_concat[mo++] = full[(size_t)(start + a) * nw + b];
        int bias = 0, repeat_j = 0;
        repeat_off[l] = rp;
        for (int j = 0; j < cnt; ++j) {
            const lo_path *cur = &s[wc[start + j]];
            if (j != 0) {
                const lo_path *prv = &s[wc[start + j - 1]];
                if (memcmp(cur->p, prv->p, sizeof(int32_t) * (size_t)l) != 0) {
                    ++bias;
                    repeat_nums_concat[rp++] = j - repeat_j;
                    repeat_j = j;
                }
            }
            tree_indices_concat[to++] = cur->p[l] + (int64_t)top_k * bias;
        }
        repeat_nums_concat[rp++] = (cnt - 1) - repeat_j + 1;
        start += cnt;
    }
    repeat_off[L] = rp;
    free(full);
    free(wc);
    free(wc_index);
    free(s);
    return 0;
}

/* ------------------------------------------------- O4: dynamic tree finalise */
/*
 * Restates the tail of Model.topK_genrate (models/drafters/cnets_llamagen.py:831-912;
 * cnets_lumina_mgpt.py:1330-1393; cnets_anole.py:913-993).  Ties in the top-T
 * selection are broken towards the lower flat index (torch leaves it unspecified).
 * retrieve is written row-major with row stride (total_tokens+1).
 */
int lo_tree_dynamic_finalize(const float *scores, const int64_t *tokens, const int64_t *parents,
                             int n_scores, int top_k, int total_tokens, int64_t sample_token,
                             int sort_rows, int64_t *draft_tokens, float *mask, int64_t *pos_ids,
                             int64_t *retrieve, int32_t *n_leaf_out, int32_t *max_depth_out) {
    const int T = total_tokens, N = T + 1;
    if (T > n_scores || N > 64) return -1;
    int *sel = (int *)malloc(sizeof(int) * (size_t)T);
    char *taken = (char *)calloc((size_t)n_scores, 1);
    /* top-T by score, then ascending index */
    for (int t = 0; t < T; ++t) {
        int bi = -1;
        for (int i = 0; i < n_scores; ++i)
            if (!taken[i] && (bi < 0 || scores[i] > scores[bi])) bi = i;
        taken[bi] = 1;
    }
    {
        int c = 0;
        for (int i = 0; i < n_scores; ++i)
            if (taken[i]) sel[c++] = i;
    }
    int *mi = (int *)malloc(sizeof(int) * (size_t)T); /* mask_index (+1 applied) */
    draft_tokens[0] = sample_token;
    for (int t = 0; t < T; ++t) {
        draft_tokens[t + 1] = tokens[sel[t]];
        int64_t dp = parents[sel[t] / top_k];
        if (dp == 0) {
            mi[t] = 0;
        } else {
            int64_t key = dp - 1;
            int pos = 0;
            while (pos < T && sel[pos] < key) ++pos; /* searchsorted(left) */
            mi[t] = pos + 1;
        }
    }
    uint64_t anc[64];
    anc[0] = 1ull;
    for (int t = 0; t < T; ++t) anc[t + 1] = (1ull << (t + 1)) | 1ull | anc[mi[t]];
    int maxd = 0;
    for (int i = 0; i < N; ++i) {
        int pc = 0;
        for (int j = 0; j < N; ++j) {
            int b = (int)((anc[i] >> j) & 1ull);
            mask[(size_t)i * N + j] = (float)b;
            pc += b;
        }
        pos_ids[i] = pc - 1;
        if (pc - 1 > maxd) maxd = pc - 1;
    }
    const int MD = maxd + 1;
    char nonleaf[64];
    memset(nonleaf, 0, sizeof(nonleaf));
    for (int t = 0; t < T; ++t) nonleaf[mi[t]] = 1;
    int rid = 0;
    lo_rrow *rr = (lo_rrow *)malloc(sizeof(lo_rrow) * (size_t)N);
    for (int i = 0; i < N; ++i) {
        if (nonleaf[i]) continue;
        rr[rid].D = MD;
        for (int j = 0; j < MD; ++j) rr[rid].val[j] = -1;
        int cid = i;
        for (int j = (int)pos_ids[i]; j >= 0; --j) {
            rr[rid].val[j] = cid;
            cid = cid > 0 ? mi[cid - 1] : 0;
        }
        for (int j = 0; j < MD; ++j) rr[rid].key[j] = rr[rid].val[j] >= 0 ? rr[rid].val[j] : (int64_t)T + 5;
        ++rid;
    }
    if (sort_rows) qsort(rr, (size_t)rid, sizeof(lo_rrow), cmp_rrow);
    for (int r = 0; r < rid; ++r)
        for (int j = 0; j < MD; ++j) retrieve[(size_t)r * N + j] = rr[r].val[j];
    *n_leaf_out = rid;
    *max_depth_out = MD;
    free(rr);
    free(mi);
    free(sel);
    free(taken);
    return 0;
}

/* ---------------------------------------------------- O6: candidate assembly */
/* Restates generate_candidates (ea_model_lumina_mgpt.py:525-554; ea_model_llamagen.py:676-706). */
int lo_gather_candidates(const int64_t *ss_token, const float *ss_prob, int n_flat, int64_t sample_token,
                         const int64_t *tree_indices, int N, const int64_t *retrieve, int P, int D,
                         int64_t *tree_cand, int64_t *cand, float *cart_prob) {
    for (int n = 0; n < N; ++n) {
        int64_t ti = tree_indices[n];
        if (ti < 0 || ti > n_flat) return -1;
        tree_cand[n] = ti == 0 ? sample_token : ss_token[ti - 1];
    }
    for (int i = 0; i < P * D; ++i) {
        int64_t r = retrieve[i];
        if (r < -1 || r >= N) return -2;
        cand[i] = r == -1 ? -1 : tree_cand[r];
        if (cart_prob) {
            if (r == -1)
                cart_prob[i] = 1.0f;
            else {
                int64_t ti = tree_indices[r];
                cart_prob[i] = ti == 0 ? 1.0f : ss_prob[ti - 1];
            }
        }
    }
    return 0;
}

/* -------------------------------------------------------- O9: KV index gather */
/*
 * Restates the slab update of update_inference_inputs (ea_model_lumina_mgpt.py:741-746;
 * ea_model_llamagen.py:961-967; KVCache.copy kv_cache.py:38-50):
 *   slab[..., prev:prev+n, :] <- slab[..., retrieve_row[:n] + prev, :]
 * slab viewed as [outer, S_max, d] (outer = 2L*B*Hkv).
 */
int lo_kv_gather(void *slab, int elem_bytes, int64_t outer, int64_t S_max, int64_t d,
                 const int64_t *retrieve_row, int n_sel, int64_t prev_len) {
    size_t rowb = (size_t)d * (size_t)elem_bytes;
    char *tmp = (char *)malloc(rowb * (size_t)(n_sel > 0 ? n_sel : 1));
    for (int t = 0; t < n_sel; ++t)
        if (retrieve_row[t] + prev_len < 0 || retrieve_row[t] + prev_len >= S_max || prev_len + t >= S_max) {
            free(tmp);
            return -1;
        }
    for (int64_t o = 0; o < outer; ++o) {
        char *base = (char *)slab + (size_t)o * (size_t)S_max * rowb;
        for (int t = 0; t < n_sel; ++t) memcpy(tmp + (size_t)t * rowb, base + (size_t)(retrieve_row[t] + prev_len) * rowb, rowb);
        memcpy(base + (size_t)prev_len * rowb, tmp, rowb * (size_t)n_sel);
    }
    free(tmp);
    return 0;
}

/* ---------------------------------------- O10: accepted-hidden gather + sampling */
/* hidden[:, retrieve][:, best, :a+1] (ea_model_lumina_mgpt.py:773-777). hidden is [B,N,H]. */
int lo_hidden_gather(const void *hidden, int elem_bytes, int B, int N, int H,
                     const int64_t *retrieve_row, int n_sel, void *out) {
    size_t rowb = (size_t)H * (size_t)elem_bytes;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < n_sel; ++t) {
            int64_t r = retrieve_row[t];
            if (r < 0) r += N;
            if (r < 0 || r >= N) return -1;
            memcpy((char *)out + ((size_t)b * n_sel + t) * rowb, (const char *)hidden + ((size_t)b * N + r) * rowb, rowb);
        }
    return 0;
}

/*
 * Bonus-token draw (torch.multinomial(sample_p, 1), ea_model_lumina_mgpt.py:781) restated
 * as inverse-CDF on an injected uniform: smallest i with cumsum(p)[i] > u * sum(p),
 * cumsum in double.  torch's own device RNG cannot be reproduced across devices
 * (SURVEY 8a RNG contract); parity is on the distribution and on this rule.
 */
int64_t lo_sample_inverse_cdf(const float *p, int V, double u) {
    double tot = sum_f32(p, V);
    double tgt = u * tot, acc = 0.0;
    int64_t last = -1;
    for (int i = 0; i < V; ++i) {
        if (p[i] > 0.0f) last = i;
        acc += (double)p[i];
        if (acc > tgt && p[i] > 0.0f) return i;
    }
    return last;
}

/* --------------------------------------------- O5: static-tree drafter sampling */
/*
 * Restates sample() (cnets_lumina_mgpt.py:936-955; cnets_llamagen.py:924-940) with
 * the multinomial indices injected: probs [R,V] (already softmaxed), idx [R,k] ->
 * out_prob[r][i] = clamp(p_i / (1 - sum_{j<i} p_j)), inf/nan -> -1 -> clamp 0.
 */
int lo_sample_static(const float *probs, int R, int V, const int64_t *idx, int k, float *out_prob) {
    for (int r = 0; r < R; ++r) {
        double acc = 0.0;
        float prev_c = 0.0f;
        for (int i = 0; i < k; ++i) {
            int64_t t = idx[(size_t)r * k + i];
            if (t < 0 || t >= V) return -1;
            float p = probs[(size_t)r * V + t];
            float v = p / (1.0f - prev_c);
            if (isinf(v) || isnan(v)) v = -1.0f;
            if (v < 0.0f) v = 0.0f;
            if (v > 1.0f) v = 1.0f;
            out_prob[(size_t)r * k + i] = v;
            acc += (double)p;
            prev_c = (float)acc;
        }
    }
    return 0;
}

/* ------------------------------------------------ O3: dynamic tree expand step */
/*
 * Restates one depth of the EAGLE-2 expansion (cnets_llamagen.py:798-820;
 * cnets_lumina_mgpt.py:1303-1318): log_softmax rows -> top_k per row -> cumulative
 * scores -> top_k of the flattened n_rows*top_k.  Ties -> lower index first.
 */
int lo_expand_dynamic(const float *logits, int n_rows, int V, int top_k, const float *scores_in,
                      int64_t *topk_index, float *cu_scores, int64_t *topk_cs_index, float *scores_out) {
    float *lp = (float *)malloc(sizeof(float) * (size_t)V);
    char *used = (char *)malloc((size_t)(V > n_rows * top_k ? V : n_rows * top_k));
    for (int r = 0; r < n_rows; ++r) {
        const float *x = logits + (size_t)r * V;
        float m = -INFINITY;
        for (int i = 0; i < V; ++i)
            if (x[i] > m) m = x[i];
        double s = 0.0;
        for (int i = 0; i < V; ++i) s += (double)expf(x[i] - m);
        float ls = logf((float)s);
        for (int i = 0; i < V; ++i) lp[i] = (x[i] - m) - ls;
        memset(used, 0, (size_t)V);
        for (int t = 0; t < top_k; ++t) {
            int bi = -1;
            for (int i = 0; i < V; ++i)
                if (!used[i] && (bi < 0 || lp[i] > lp[bi])) bi = i;
            used[bi] = 1;
            topk_index[(size_t)r * top_k + t] = bi;
            cu_scores[(size_t)r * top_k + t] = lp[bi] + (scores_in ? scores_in[r] : 0.0f);
        }
    }
    int nf = n_rows * top_k;
    memset(used, 0, (size_t)nf);
    for (int t = 0; t < top_k && t < nf; ++t) {
        int bi = -1;
        for (int i = 0; i < nf; ++i)
            if (!used[i] && (bi < 0 || cu_scores[i] > cu_scores[bi])) bi = i;
        used[bi] = 1;
        topk_cs_index[t] = bi;
        scores_out[t] = cu_scores[bi];
    }
    free(lp);
    free(used);
    return 0;
}

/* --------------------------------------------------- O11: drafter input stage */
/*
 * Restates Model.forward's input contraction (cnets_lumina_mgpt.py:1071,1095-1098;
 * cnets_llamagen.py:642,679-680): fc(cat(embed_tokens(ids) * scale, hidden)) with bf16
 * operands, f32 accumulation, f32 output (the caller rounds to bf16).
 * w is [H, 2H] row-major (nn.Linear weight), bias [H] or NULL.
 */
int lo_drafter_fc(const int64_t *ids, const uint16_t *hidden_bf16, const uint16_t *embed_bf16,
                  const uint16_t *w_bf16, const uint16_t *bias_bf16, int M, int H, float embed_scale,
                  float *out_f32) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const uint16_t *e = embed_bf16 + (size_t)ids[m] * H;
        const uint16_t *h = hidden_bf16 + (size_t)m * H;
        for (int o = 0; o < H; ++o) {
            const uint16_t *w = w_bf16 + (size_t)o * 2 * H;
            double acc = 0.0;
            for (int i = 0; i < H; ++i) {
                float ev = bf16_to_f32(e[i]);
                if (embed_scale > 1.0f) ev = round_bf16(ev * embed_scale);
                acc += (double)ev * (double)bf16_to_f32(w[i]);
            }
            for (int i = 0; i < H; ++i) acc += (double)bf16_to_f32(h[i]) * (double)bf16_to_f32(w[H + i]);
            if (bias_bf16) acc += (double)bf16_to_f32(bias_bf16[o]);
            out_f32[(size_t)m * H + o] = (float)acc;
        }
    }
    return 0;
}

/* ------------------------------------------- 8f-1: VQ-distance neighbour table */
/*
 * Restates entrypoints/generate_codebook.py:53-65: cdist (L2) of the codebook,
 * diagonal = inf, ascending order, self excluded -> uint16 [K, K-1].
 * Ties -> lower index first.
 */
typedef struct {
    double d;
    int i;
} di_pair;
static int cmp_di(const void *a, const void *b) {
    const di_pair *x = (const di_pair *)a, *y = (const di_pair *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->i - y->i;
}
int lo_build_vq_table(const float *codebook, int K, int C, uint16_t *table) {
    if (K > 65536) return -1;
#pragma omp parallel
    {
        di_pair *pr = (di_pair *)malloc(sizeof(di_pair) * (size_t)K);
#pragma omp for schedule(dynamic, 16)
        for (int a = 0; a < K; ++a) {
            int c = 0;
            for (int b = 0; b < K; ++b) {
                if (b == a) continue;
                double s = 0.0;
                for (int t = 0; t < C; ++t) {
                    double df = (double)codebook[(size_t)a * C + t] - (double)codebook[(size_t)b * C + t];
                    s += df * df;
                }
                pr[c].d = s;
                pr[c].i = b;
                ++c;
            }
            qsort(pr, (size_t)(K - 1), sizeof(di_pair), cmp_di);
            for (int t = 0; t < K - 1; ++t) table[(size_t)a * (K - 1) + t] = (uint16_t)pr[t].i;
        }
        free(pr);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * lo_verify_loop_mt -- the whole synthetic verify loop of bench.py's CPU baseline in ONE call,
 * on n_threads pthreads (thread t owns sequences t, t + n_threads, ...): per step and sequence
 *   lo_gather_candidates -> lo_cfg_mask_topk (bf16 inputs) -> lo_evaluate_posterior (static Lumina,
 *   dense [R,V] drafter rows rebuilt from the windowed pool) -> lo_kv_gather x 2 -> lo_hidden_gather
 *   -> lo_sample_inverse_cdf
 * i.e. exactly what bench.py::cpu_baseline.run_seq does through the Python wrappers, without the
 * interpreter (and its lock) between the calls.  Test infrastructure / timed baseline only.
 * ------------------------------------------------------------------------------------------- */
#include <pthread.h>

typedef struct lo_loop_args {
    int32_t n_seq, n_steps, pool_steps, n_threads;
    int32_t N, P, D, R, V, W, win_lo, H;
    lo_ep_params prm;
    /* tree (shared) */
    const int64_t *tree_indices, *retrieve, *pos1; /* pos1 = tree_position_ids + 1 */
    const int32_t *row_index, *p_idx, *b_off, *b_idx, *op_off;
    /* pools, layout [pool_steps][n_seq_stride][...] */
    int64_t seq_stride;                 /* sequences per pool step in the pool tensors */
    const int64_t *ss_token;            /* [S, seq_stride, R*10] */
    const float *ss_prob;               /* [S, seq_stride, R*10] */
    const uint16_t *cond, *uncond;      /* [S, seq_stride, N, V] bf16 bits */
    const float *orig_win;              /* [S, seq_stride, R, W] */
    const uint16_t *hidden;             /* [S, seq_stride, 2, N, H] bf16 bits */
    const uint16_t *nn_table;
    const double *uniforms;             /* [n_seq_total, n_uniforms] */
    int64_t n_uniforms;
    const double *u_bonus;              /* [steps, seq_stride] */
    const int64_t *first_token;         /* [seq_stride] */
    float cfg_scale;
    int32_t w_latent, h_latent, newline_id, eos_id, top_k;
    int64_t prompt_len, tokens_per_image;
    /* KV slabs of the host copy: kv_outer * kv_smax * kv_dim bf16 each, 2 per sequence, or NULL */
    uint16_t **slabs;
    int64_t kv_outer, kv_smax, kv_dim;
    /* outputs [n_steps, n_seq] */
    int32_t *best, *alen;
    int64_t *token;
    int32_t *rc;                        /* [n_threads] */
} lo_loop_args;

typedef struct {
    const lo_loop_args *a;
    int tid;
} lo_loop_thread;

static void *lo_loop_worker(void *vp) {
    const lo_loop_thread *th = (const lo_loop_thread *)vp;
    const lo_loop_args *a = th->a;
    const int N = a->N, P = a->P, D = a->D, R = a->R, V = a->V, W = a->W, H = a->H;
    int64_t *tree_cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)P * D);
    float *cart = (float *)malloc(sizeof(float) * (size_t)P * D);
    float *proc = (float *)malloc(sizeof(float) * (size_t)N * V);
    float *orig = (float *)calloc((size_t)R * V, sizeof(float));
    float *sp = (float *)malloc(sizeof(float) * (size_t)V);
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    uint16_t *hout = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)2 * D * H);
    int rc = 0;
    for (int b = th->tid; b < a->n_seq && rc == 0; b += a->n_threads) {
        int64_t lens[2] = {a->prompt_len + 3, 3};
        int64_t tok = a->first_token[b];
        int64_t cursor = 0;
        for (int i = 0; i < a->n_steps && rc == 0; ++i) {
            const int s = i % a->pool_steps;
            const size_t sb = (size_t)s * (size_t)a->seq_stride + (size_t)b;
            rc = lo_gather_candidates(a->ss_token + sb * R * 10, a->ss_prob + sb * R * 10, R * 10, tok, a->tree_indices, N, a->retrieve, P,
                                      D, tree_cand, cand, cart);
            if (rc) break;
            for (int n = 0; n < N; ++n) pos[n] = a->pos1[n] + lens[0];
            rc = lo_cfg_mask_topk(a->cond + sb * N * V, a->uncond + sb * N * V, LO_BF16, N, V, a->cfg_scale, LO_MODEL_LUMINA, pos,
                                  a->prompt_len + 3, a->w_latent, a->h_latent, a->prm.img_lo, a->prm.img_hi, a->newline_id, a->eos_id,
                                  a->top_k, proc);
            if (rc) break;
            for (int r = 0; r < R; ++r) memcpy(orig + (size_t)r * V + a->win_lo, a->orig_win + (sb * R + r) * W, sizeof(float) * (size_t)W);
            int32_t best = 0, alen = 0, cnt[6];
            const int64_t left = a->n_uniforms - cursor;
            rc = lo_evaluate_posterior(&a->prm, proc, a->row_index, cand, cart, orig, a->op_off, a->p_idx, a->b_off, a->b_idx, tree_cand,
                                       a->nn_table, a->uniforms + (size_t)b * a->n_uniforms + cursor, (int32_t)(left < 64 ? left : 64), &best,
                                       &alen, sp, cnt);
            if (rc) break;
            cursor += cnt[3];
            const int64_t *row = a->retrieve + (size_t)best * D;
            if (a->slabs)
                for (int j = 0; j < 2; ++j) lo_kv_gather(a->slabs[2 * b + j], 2, a->kv_outer, a->kv_smax, a->kv_dim, row, alen + 1, lens[j]);
            lo_hidden_gather(a->hidden + sb * 2 * N * H, 2, 2, N, H, row, alen + 1, hout);
            tok = lo_sample_inverse_cdf(sp, V, a->u_bonus[(size_t)i * a->seq_stride + b]);
            lens[0] += alen + 1;
            lens[1] += alen + 1;
            if (lens[1] - 3 >= a->tokens_per_image) {
                lens[0] = a->prompt_len + 3;
                lens[1] = 3;
            }
            a->best[(size_t)i * a->n_seq + b] = best;
            a->alen[(size_t)i * a->n_seq + b] = alen;
            a->token[(size_t)i * a->n_seq + b] = tok;
        }
    }
    a->rc[th->tid] = rc;
    free(tree_cand); free(cand); free(cart); free(proc); free(orig); free(sp); free(pos); free(hout);
    return NULL;
}

int lo_verify_loop_mt(const lo_loop_args *a) {
    if (!a || a->n_threads < 1 || a->n_threads > 1024) return -1;
    pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)a->n_threads);
    lo_loop_thread *th = (lo_loop_thread *)malloc(sizeof(lo_loop_thread) * (size_t)a->n_threads);
    int started = 0;
    for (int i = 0; i < a->n_threads; ++i) {
        th[i].a = a;
        th[i].tid = i;
        if (pthread_create(&t[i], NULL, lo_loop_worker, &th[i]) != 0) break;
        ++started;
    }
    for (int i = 0; i < started; ++i) pthread_join(t[i], NULL);
    int rc = started == a->n_threads ? 0 : -2;
    for (int i = 0; i < started && rc == 0; ++i) rc = a->rc[i];
    free(t);
    free(th);
    return rc;
}
