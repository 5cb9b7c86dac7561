// vq_table.hip -- 8f-1: the VQ-distance neighbour table, rows of codes by ascending L2 distance with the
// code itself excluded (entrypoints/generate_codebook.py:53-65: cdist -> diagonal=inf -> topk(K-1, smallest)).
//
// One 1024-thread workgroup per code: squared distances to all K codes in f64 (exact for f32 inputs up to
// rounding of the sum; ties -> lower index, where the reference's f32 cdist leaves the order to chance), kept in
// LDS next to their indices, bitonic-sorted in place, and the first K-1 indices written as uint16.
// K <= 8192 (Lumina / Anole codebooks): 64 KiB keys + 16 KiB indices per workgroup; 8192 < K <= 16384: packed keys (below).
#include "common.h"

namespace lantern {

constexpr int VQ_THREADS = 1024;

__global__ __launch_bounds__(VQ_THREADS) void vq_table_kernel(const float *__restrict__ cb, int K, int C, int Kp2, uint16_t *__restrict__ table) {
    extern __shared__ double vq_lds[];
    double *key = vq_lds;
    uint16_t *idx = reinterpret_cast<uint16_t *>(key + Kp2);
    const int a = blockIdx.x, tid = threadIdx.x;
    const float *ra = cb + (size_t)a * C;
    for (int bb = tid; bb < Kp2; bb += VQ_THREADS) {
        double s = __builtin_inf();
        if (bb < K && bb != a) {
            const float *rb = cb + (size_t)bb * C;
            s = 0.0;
            for (int t = 0; t < C; ++t) {
                const double df = (double)ra[t] - (double)rb[t];
                s += df * df;
            }
        }
        key[bb] = s;
        idx[bb] = (uint16_t)(bb < K ? bb : 0xffff);
    }
    __syncthreads();
    for (int size = 2; size <= Kp2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < Kp2 / 2; t += VQ_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const double kl = key[lo], kh = key[hi];
                const uint16_t il = idx[lo], ih = idx[hi];
                const bool gt = kl > kh || (kl == kh && il > ih);
                if (gt == asc) {
                    key[lo] = kh; key[hi] = kl;
                    idx[lo] = ih; idx[hi] = il;
                }
            }
            __syncthreads();
        }
    for (int t = tid; t < K - 1; t += VQ_THREADS) table[(size_t)a * (K - 1) + t] = idx[t];
}

// K <= 16384 (LlamaGen's codebook): (distance, index) packed into one 64-bit key -- the upper 48 bits of the non-negative
// f64 distance (bit patterns of non-negative doubles order like the values; 36 mantissa bits kept, 1.5e-11 relative)
// over the 16-bit index, which also breaks ties towards the lower index -- so 16384 keys fit 128 KiB of LDS.
__global__ __launch_bounds__(VQ_THREADS) void vq_table_packed_kernel(const float *__restrict__ cb, int K, int C, int Kp2,
                                                                     uint16_t *__restrict__ table) {
    extern __shared__ unsigned long long vq_keys[];
    const int a = blockIdx.x, tid = threadIdx.x;
    const float *ra = cb + (size_t)a * C;
    for (int bb = tid; bb < Kp2; bb += VQ_THREADS) {
        unsigned long long key = ~0ull;
        if (bb < K && bb != a) {
            const float *rb = cb + (size_t)bb * C;
            double s = 0.0;
            for (int t = 0; t < C; ++t) {
                const double df = (double)ra[t] - (double)rb[t];
                s += df * df;
            }
            key = ((unsigned long long)__double_as_longlong(s) & ~0xffffull) | (unsigned long long)bb;
        }
        vq_keys[bb] = key;
    }
    __syncthreads();
    for (int size = 2; size <= Kp2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < Kp2 / 2; t += VQ_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const unsigned long long kl = vq_keys[lo], kh = vq_keys[hi];
                if ((kl > kh) == asc) {
                    vq_keys[lo] = kh;
                    vq_keys[hi] = kl;
                }
            }
            __syncthreads();
        }
    for (int t = tid; t < K - 1; t += VQ_THREADS) table[(size_t)a * (K - 1) + t] = (uint16_t)(vq_keys[t] & 0xffffull);
}

}  // namespace lantern

using namespace lantern;

extern "C" int lantern_build_vq_table(const float *codebook, int K, int C, uint16_t *table, void *workspace, void *stream) {
    (void)workspace;
    LANTERN_CHECK_ARG(codebook && table && K >= 2 && C >= 1, "build_vq_table: bad arguments");
    if (K > 16384) {
        set_error("build_vq_table: K=%d > 16384 (ids no longer fit the packed 64-bit sort keys held in LDS)", K);
        return LANTERN_E_UNSUPPORTED;
    }
    int Kp2 = 1;
    while (Kp2 < K) Kp2 <<= 1;
    if (K > 8192) {
        const size_t lds = (size_t)Kp2 * 8;
        hipError_t e = hipFuncSetAttribute((const void *)vq_table_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("build_vq_table: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return LANTERN_E_LAUNCH;
        }
        hipLaunchKernelGGL(vq_table_packed_kernel, dim3(K), dim3(VQ_THREADS), lds, (hipStream_t)stream, codebook, K, C, Kp2, table);
        LANTERN_CHECK_LAUNCH("build_vq_table");
        return LANTERN_OK;
    }
    const size_t lds = (size_t)Kp2 * 8 + (size_t)Kp2 * 2;
    hipError_t e = hipFuncSetAttribute((const void *)vq_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("build_vq_table: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return LANTERN_E_LAUNCH;
    }
    hipLaunchKernelGGL(vq_table_kernel, dim3(K), dim3(VQ_THREADS), lds, (hipStream_t)stream, codebook, K, C, Kp2, table);
    LANTERN_CHECK_LAUNCH("build_vq_table");
    return LANTERN_OK;
}

namespace lantern {
__global__ void pack_vq_table_kernel(const uint16_t *__restrict__ src, int rows, int src_cols, uint16_t *__restrict__ dst, int dst_cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < dst_cols; c += blockDim.x) dst[(size_t)r * dst_cols + c] = c < src_cols ? src[(size_t)r * src_cols + c] : (uint16_t)0;
}
}  // namespace lantern

extern "C" int lantern_pack_vq_table(const uint16_t *src, int rows, int src_cols, uint16_t *dst, int dst_cols, void *stream) {
    LANTERN_CHECK_ARG(src && dst && rows >= 0 && src_cols > 0 && dst_cols > 0, "pack_vq_table: bad arguments");
    if (rows == 0) return LANTERN_OK;
    hipLaunchKernelGGL(lantern::pack_vq_table_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, src, rows, src_cols, dst, dst_cols);
    LANTERN_CHECK_LAUNCH("pack_vq_table");
    return LANTERN_OK;
}
