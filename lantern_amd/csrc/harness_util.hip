// harness_util.hip -- bookkeeping kernel of the synthetic benchmark harness (lantern_amd/harness.py), NOT part of
// the product API in include/lantern_hip.h: it appends one step's results to the device-side logs, feeds the bonus
// token back as the next step's sample token, wraps finished images and advances the device step counter, so that a
// whole verify step is a fixed sequence of kernels with fixed arguments (capturable once into a hipGraph per pool slot).
#include "common.h"

namespace lantern {
__global__ void harness_advance_kernel(int B, int n_lens, int ld, int64_t tokens_per_image, int64_t max_steps, int64_t step_host,
                                       int64_t *__restrict__ step_dev,
                                       const int32_t *__restrict__ st_best, const int32_t *__restrict__ st_alen,
                                       const int32_t *__restrict__ st_cnt, const int64_t *__restrict__ st_token,
                                       int32_t *__restrict__ log_best, int32_t *__restrict__ log_alen, int32_t *__restrict__ log_cnt,
                                       int64_t *__restrict__ log_token, int64_t *__restrict__ sample_token,
                                       int64_t *__restrict__ lens_next, const int64_t *__restrict__ lens_base,
                                       const double *__restrict__ u_bonus, double *__restrict__ u_cur) {
    const int64_t step = step_host >= 0 ? step_host : step_dev[0];   // eager launches know the step; graph replays read the device counter
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (step < max_steps) {
            log_best[step * ld + b] = st_best[b];
            log_alen[step * ld + b] = st_alen[b];
            log_token[step * ld + b] = st_token[b];
            for (int c = 0; c < 6; ++c) log_cnt[(step * ld + b) * 6 + c] = st_cnt[b * 6 + c];
        }
        sample_token[b] = st_token[b];
        if (step + 1 < max_steps) u_cur[b] = u_bonus[(step + 1) * ld + b];
    }
    for (int t = threadIdx.x; t < n_lens; t += blockDim.x)
        if (lens_next[t] - lens_base[t] >= tokens_per_image) lens_next[t] = lens_base[t];
    __syncthreads();
    if (threadIdx.x == 0) step_dev[0] = step + 1;
}
}  // namespace lantern

// B sequences of one group; `ld` = row stride (all sequences) of the step-major logs and of u_bonus
extern "C" int lantern_harness_advance(int B, int n_lens, int ld, int64_t tokens_per_image, int64_t max_steps, int64_t step_host,
                                       int64_t *step_dev,
                                       const int32_t *st_best, const int32_t *st_alen, const int32_t *st_cnt, const int64_t *st_token,
                                       int32_t *log_best, int32_t *log_alen, int32_t *log_cnt, int64_t *log_token, int64_t *sample_token,
                                       int64_t *lens_next, const int64_t *lens_base, const double *u_bonus, double *u_cur, void *stream) {
    hipLaunchKernelGGL(lantern::harness_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, n_lens, ld, tokens_per_image, max_steps,
                       step_host, step_dev, st_best, st_alen, st_cnt, st_token, log_best, log_alen, log_cnt, log_token, sample_token, lens_next,
                       lens_base, u_bonus, u_cur);
    LANTERN_CHECK_LAUNCH("harness_advance");
    return LANTERN_OK;
}
