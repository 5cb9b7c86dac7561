// common.h -- shared host/device helpers of liblantern_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lantern_hip.h"

namespace lantern {

// ---------------------------------------------------------------- host error state
void set_error(const char *fmt, ...);

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

// round-to-nearest-even f32 -> bf16 -> f32 (torch's per-op bf16 rounding)
__device__ __forceinline__ float round_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return __uint_as_float(0x7fc00000u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}

// order-preserving float -> uint key (ascending)
__device__ __forceinline__ uint32_t float_key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
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

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

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

// inclusive wave scan (double)
__device__ __forceinline__ double wave_scan_incl(double v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

}  // namespace lantern
