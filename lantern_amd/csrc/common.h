// common.h -- shared host/device helpers of liblantern_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lantern_hip.h"

namespace lantern {

// ---------------------------------------------------------------- host error state
void set_error(const char *fmt, ...);

// ---------------------------------------------------------------- tuning values (lantern_tuning_set / lantern_tuning_get)
// Kernel-instance and launch-shape choices a measurement may want to override.  They are arguments of an explicit C-ABI call -- the library reads
// NO environment variable (round 5's getenv knobs are gone); the defaults below are what every product path runs.
enum Tuning {
    TUNE_EPW_TP = 0,        // 5: throughput instances of the chain kernel for > 256 sequences per launch (0: generic two-per-CU instance; 1..4: older forms)
    TUNE_EPW_TP4,           // 1: the compact four-per-CU instance for the default tree; 0: round 4's three-per-CU form
    TUNE_EPW_TP_RAW,        // 256: threads per sequence of the raw-row throughput instances (512: the 128-VGPR form)
    TUNE_EPW_SPEC,          // 2: fixed-configuration instances (1: no fixed tree, 0: the generic instance)
    TUNE_EPW_OCC2,          // -1: throughput forms when B > 256 (0 / 1 force them off / on)
    TUNE_O7_NT,             // 0: threads per row of cfg_window_bf16_kernel by window size (256 / 1024 force a form)
    TUNE_PREP_NT,           // 0: prep_rows_kernel<512> (1024: the 1024-thread form)
    TUNE_KV_U,              // 0 -> 2 row groups in flight per thread of the KV mover
    TUNE_KV_KS,             // 4: slabs per workgroup of the small-slab commit kernel (0: one workgroup tile per slab)
    TUNE_KV_VARIANT,        // 0: 10 * U + mode of the KV mover (measurement variants)
    TUNE_GEMM_TILED_FROM,   // 129: rows from which lantern_linear_rows_packed runs the LDS-tiled GEMM
    TUNE_SK_GROUPS,         // 0: stream-K grid = one workgroup per CU (> 0 forces the grid)
    TUNE_SK_WHOLE_MB,       // 40: matrices up to this many MB get whole tiles per workgroup
    TUNE_SK_NT_MIN_MB,      // 80: matrices from this many MB are streamed with non-temporal loads
    TUNE_TA_SPLITS,         // 0: key splits of tree attention by launch size (> 0 forces the split count)
    TUNE_TA_MIN_TILES,      // 2: key tiles per wave and split below which no further split is made
    TUNE_EPW_TP_LG,         // 1: LlamaGen's 16384-id throughput instance (two per CU, rows by LDS-DMA); 2: + second LDS pass for the residual; 3: rows through registers; 4: + raised priority; 0: off
    TUNE_EPW_FUSED_HELPERS, // 1: helper workgroups per sequence of the chain launch that carries its prepare stage (LANTERN_STEP_FUSED_PREPARE)
    TUNE_COUNT
};
int tuning(int t);

// ---------------------------------------------------------------- measurement aid (lantern_profile_next_launch)
// When a (start, stop) event pair is armed on this thread, the next launch through LANTERN_LAUNCH records them at
// kernel begin / end (hipExtLaunchKernelGGL: the dispatch's own timestamps, i.e. the kernel-only duration rocprofv3
// reports) and disarms; otherwise a plain hipLaunchKernelGGL.
void take_launch_events(void **start, void **stop);
#define LANTERN_LAUNCH(kernel, grid, block, lds, st, ...)                                                                  \
    do {                                                                                                                   \
        void *ev0__ = nullptr, *ev1__ = nullptr;                                                                           \
        ::lantern::take_launch_events(&ev0__, &ev1__);                                                                     \
        if (ev0__ && ev1__)                                                                                                \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, (hipEvent_t)ev0__, (hipEvent_t)ev1__, 0, __VA_ARGS__);     \
        else                                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                                 \
    } while (0)

#define LANTERN_CHECK_ARG(cond, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::lantern::set_error(__VA_ARGS__);  \
            return LANTERN_E_INVALID;           \
        }                                       \
    } while (0)

#define LANTERN_CHECK_LAUNCH(what)                                                       \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::lantern::set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return LANTERN_E_LAUNCH;                                                     \
        }                                                                                \
    } while (0)

// ---------------------------------------------------------------- device helpers
constexpr int kWave = 64;

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// round-to-nearest-even f32 -> bf16 -> f32 (torch's per-op bf16 rounding): v_cvt_pk_bf16_f32 + a shift on gfx950
__device__ __forceinline__ float round_bf16(float f) { return (float)(__bf16)f; }

// Two at a time: one v_cvt_pk_bf16_f32 rounds a pair (RNE, the same result as round_bf16 on each), the halves come back with a
// shift and a mask -- against cvt + shift (+ a NaN-quieting v_max the compiler adds around the scalar conversion) per value.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t round_bf16x2(f32x2_t v) {
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const unsigned int w = __builtin_bit_cast(unsigned int, h);
    f32x2_t r;
    r.x = __uint_as_float(w << 16);
    r.y = __uint_as_float(w & 0xffff0000u);
    return r;
}
// torch's bf16 CFG mix  u + s * (c - u)  with a bf16 rounding after every operation, on a packed pair of bf16 values per word
__device__ __forceinline__ f32x2_t cfg_mix_bf16x2(unsigned int cw, unsigned int uw, float cfg) {
    f32x2_t c, u;
    c.x = __uint_as_float(cw << 16);
    c.y = __uint_as_float(cw & 0xffff0000u);
    u.x = __uint_as_float(uw << 16);
    u.y = __uint_as_float(uw & 0xffff0000u);
    f32x2_t t = round_bf16x2(c - u);
    t = round_bf16x2(t * cfg);
    return round_bf16x2(u + t);
}

// order-preserving float -> uint key (ascending)
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t u = __float_as_uint(f);
    // negative: ~u, else u | sign bit -- as one xor with (sign-extended sign | sign bit): three operations, no compare + select
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// v[c] = val for a runtime component index c in 0..3 (no pointer arithmetic across members)
__device__ __forceinline__ void set_comp(float4 &v, int c, float val) {
    v.x = (c == 0) ? val : v.x;
    v.y = (c == 1) ? val : v.y;
    v.z = (c == 2) ? val : v.z;
    v.w = (c == 3) ? val : v.w;
}

// ---- wave64 cross-lane primitives on DPP (row shifts / row broadcasts: VALU-rate, no LDS crossbar).
// ds_bpermute-based __shfl costs ~50-90 cycles per step and serialises on the CU's LDS unit when 16
// waves reduce at once; the DPP forms below are plain VALU moves (MI355X_MICROARCH.md: cross-lane
// without LDS).  ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int v) {
    // full row/bank masks: bound_ctrl makes lanes without a source read 0 by itself, so no `old` register has to be zeroed
    // before every move; with a partial row mask the rows left out keep `old` = 0
    constexpr bool BC = (ROW_MASK == 0xf && BANK_MASK == 0xf);
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, BC);
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v) {
    constexpr bool BC = (ROW_MASK == 0xf && BANK_MASK == 0xf);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BC));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    constexpr bool BC = (ROW_MASK == 0xf && BANK_MASK == 0xf);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, BANK_MASK, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, BANK_MASK, BC);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// the same move with `old` = the value itself: lanes without a valid source (and rows outside the row mask) read their own
// value back, the identity of min / max -- no lane predicate needed
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_self(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_self(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// inclusive scan over the 64 lanes (Kogge-Stone inside 16-lane rows, then row broadcasts)
template <typename T>
__device__ __forceinline__ T wave_scan_incl_dpp(T v) {
    // dpp_mov hands lanes without a valid source (and rows outside the row mask) the `old` operand = 0: adding it is the
    // identity, so the adds need no lane predicate (one v_add with a DPP operand per step for 32-bit types)
    v += dpp_mov<0x111>(v);
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    v += dpp_mov<0x142, 0xa>(v);
    v += dpp_mov<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ int readlane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float readlane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double readlane63(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// full-wave sum / max, result in every lane (scan, then broadcast lane 63)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    return readlane63(wave_scan_incl_dpp(v));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_self<0x111>(v));
    v = fmaxf(v, dpp_self<0x112>(v));
    v = fmaxf(v, dpp_self<0x114>(v));
    v = fmaxf(v, dpp_self<0x118>(v));
    v = fmaxf(v, dpp_self<0x142, 0xa>(v));
    v = fmaxf(v, dpp_self<0x143, 0xc>(v));
    return readlane63(v);
}

// integer min / max over the wave (DPP), result in every lane
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_self<0x111>(v));
    v = min(v, dpp_self<0x112>(v));
    v = min(v, dpp_self<0x114>(v));
    v = min(v, dpp_self<0x118>(v));
    v = min(v, dpp_self<0x142, 0xa>(v));
    v = min(v, dpp_self<0x143, 0xc>(v));
    return readlane63(v);
}
__device__ __forceinline__ int wave_max_i(int v) { return -wave_min_i(-v); }

// Block-wide reductions for blockDim.x = NW*64.  `sm` is a shared scratch of >= 2*NW
// elements; consecutive calls alternate halves (`phase`) so one barrier per call suffices.
template <typename T, int NW>
__device__ __forceinline__ T block_sum(T v, T *sm, int &phase) {
    v = wave_sum(v);
    T *buf = sm + (phase & 1) * NW;
    phase ^= 1;
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    T s = buf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += buf[w];
    return s;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float *sm, int &phase) {
    v = wave_max(v);
    float *buf = sm + (phase & 1) * NW;
    phase ^= 1;
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = buf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = fmaxf(s, buf[w]);
    return s;
}

// Lean block reductions: one DPP wave reduce, one LDS slot per wave, one barrier, then the first lanes of
// every wave combine the <= 16 partials with row-local DPP steps (instead of every thread reading all partials).
template <typename T, int NW>
__device__ __forceinline__ T block_sum_fast(T v, T *sm, int &phase) {
    static_assert(NW <= 16, "one DPP row");
    v = wave_sum(v);
    T *buf = sm + (phase & 1) * NW;
    phase ^= 1;
    const int lane = threadIdx.x & 63;
    if (lane == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    T x = (lane < NW) ? buf[lane] : T(0);
    x += dpp_mov<0x111>(x);      // lanes without a source add the `old` operand = 0 (see wave_scan_incl_dpp)
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    // lane 15 of row 0 holds the total of partials 0..15
    if constexpr (sizeof(T) == 8) {
        const long long b = __double_as_longlong((double)x);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 15), hi = __builtin_amdgcn_readlane((int)(b >> 32), 15);
        return (T)__longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    } else if constexpr (sizeof(T) == 4 && !__is_same(T, float)) {
        return (T)__builtin_amdgcn_readlane((int)x, 15);
    } else {
        return (T)__int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)x), 15));
    }
}
template <int NW>
__device__ __forceinline__ float block_max_fast(float v, float *sm, int &phase) {
    static_assert(NW <= 16, "one DPP row");
    v = wave_max(v);
    float *buf = sm + (phase & 1) * NW;
    phase ^= 1;
    const int lane = threadIdx.x & 63;
    if (lane == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float x = (lane < NW) ? buf[lane] : -__builtin_inff();
    x = fmaxf(x, dpp_self<0x111>(x));
    x = fmaxf(x, dpp_self<0x112>(x));
    x = fmaxf(x, dpp_self<0x114>(x));
    x = fmaxf(x, dpp_self<0x118>(x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 15));
}

// inclusive wave scan (double)
__device__ __forceinline__ double wave_scan_incl(double v) { return wave_scan_incl_dpp(v); }

}  // namespace lantern
