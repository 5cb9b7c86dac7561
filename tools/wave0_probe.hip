// Diagnostic (not product): how fast does ONE wave run a dependent VALU/DPP/LDS chain (a) alone in its workgroup,
// (b) with 7 sibling waves parked at s_barrier, (c) with siblings having global loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../lantern_amd/csrc/common.h"
using namespace lantern;
template <int MODE>
__global__ void probe(unsigned long long *out, double *sink, const float4 *src, int iters) {
    __shared__ float g[8192];
    for (int t = threadIdx.x; t < 8192; t += blockDim.x) g[t] = 1.0f / 8192;
    __syncthreads();
    float4 q = make_float4(0, 0, 0, 0);
    if (MODE == 2) q = src[blockIdx.x * 4096 + threadIdx.x];
    double v = threadIdx.x * 1e-3, acc = 0;
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (threadIdx.x < 64) {
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < iters; ++i) {
            v = wave_scan_incl_dpp(v) * 1e-3;
            acc += v;
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    unsigned long long t2 = 0, t3 = 0;
    int idx = threadIdx.x * 37;
    if (threadIdx.x < 64) {
        t2 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            double loc = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) loc += (double)g[(idx + c * 531 + i) & 8191];
            acc += loc;
            idx += (int)(loc * 8192.0);
        }
        t3 = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4] = (t1 - t0) / iters; out[blockIdx.x * 4 + 1] = (r1 - r0); out[blockIdx.x * 4 + 2] = t1 - t0; out[blockIdx.x * 4 + 3] = (t3 - t2) / iters;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + q.x;
}
int main() {
    unsigned long long *d; hipMalloc(&d, 4096 * 8); std::vector<unsigned long long> h(4096);
    double *sink; hipMalloc(&sink, 256 * 1024 * 8);
    float4 *src; hipMalloc(&src, 256 * 4096 * 16);
    for (int nb : {1, 48, 256}) for (int nt : {64, 512}) {
        hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(nt), 0, 0, d, sink, src, 200); hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(nt), 0, 0, d, sink, src, 200); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 32, hipMemcpyDeviceToHost);
        printf("blocks %3d threads %3d: f64 dpp scan %llu cyc/iter; clock %.0f MHz; 16 dependent-issue LDS gathers+f64 adds %llu cyc/iter\n", nb, nt, h[0], (double)h[2] / h[1] * 100.0, h[3]);
        hipLaunchKernelGGL(probe<2>, dim3(nb), dim3(nt), 0, 0, d, sink, src, 200); hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 32, hipMemcpyDeviceToHost);
        printf("   with sibling loads in flight:      scan %llu cyc/iter; gathers %llu\n", h[0], h[3]);
    }
    return 0;
}
