// Diagnostic microbenchmarks (not part of the product): shader clock under light load, cost of
// __syncthreads, f64 wave scan, dependent LDS reads, dependent global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void clk(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x*4+2] = (unsigned long long)a; }
}
__global__ void barrier_cost(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / iters;
}
__global__ void scan_cost(unsigned long long *out, double *sink, int iters) {
    double v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        for (int o = 1; o < 64; o <<= 1) { double t = __shfl_up(v, o, 64); if ((threadIdx.x & 63) >= o) v += t; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / iters;
    sink[threadIdx.x] = v;
}
__global__ void lds_chain(unsigned long long *out, int iters) {
    __shared__ int s[256];
    s[threadIdx.x] = (threadIdx.x * 7 + 1) & 255;
    __syncthreads();
    int j = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) j = s[j];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / iters + (j == 12345);
}
__global__ void gmem_chain(unsigned long long *out, const int *p, int iters) {
    int j = blockIdx.x * 1024;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) j = p[j];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / iters + (j == -5);
}
__global__ void busy(float *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; float a = i; for (int k = 0; k < n; ++k) a = a * 1.0001f + 1.f; x[i] = a; }
int main() {
    unsigned long long *d; hipMalloc(&d, 4096 * 8); std::vector<unsigned long long> h(4096);
    double *sink; hipMalloc(&sink, 4096 * 8);
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1) { float *x; hipMalloc(&x, 256 * 1024 * 4 * 4); for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, 0, x, 20000); hipDeviceSynchronize(); }
        for (int blocks : {1, 32, 256}) {
            hipLaunchKernelGGL(clk, dim3(blocks), dim3(256), 0, 0, d, 100000); hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 32, hipMemcpyDeviceToHost);
            printf("phase %d blocks %3d: memtime %llu realtime(100MHz) %llu -> clock %.0f MHz, cycles/iter %.2f\n", phase, blocks, h[0], h[1], (double)h[0] / h[1] * 100.0, (double)h[0] / 100000);
        }
    }
    hipLaunchKernelGGL(barrier_cost, dim3(32), dim3(256), 0, 0, d, 1000); hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf("__syncthreads (256 thr): %llu cycles\n", h[0]);
    hipLaunchKernelGGL(barrier_cost, dim3(32), dim3(1024), 0, 0, d, 1000); hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf("__syncthreads (1024 thr): %llu cycles\n", h[0]);
    hipLaunchKernelGGL(scan_cost, dim3(32), dim3(256), 0, 0, d, sink, 1000); hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf("f64 wave inclusive scan (6 shfl_up steps): %llu cycles\n", h[0]);
    hipLaunchKernelGGL(lds_chain, dim3(32), dim3(256), 0, 0, d, 1000); hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf("dependent LDS read: %llu cycles\n", h[0]);
    int n = 64 << 20; int *p; hipMalloc(&p, (size_t)n * 4); std::vector<int> hp(n);
    for (int i = 0; i < n; ++i) hp[i] = (int)(((long long)i * 1000003LL + 12345) % n);
    hipMemcpy(p, hp.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gmem_chain, dim3(32), dim3(64), 0, 0, d, p, 200); hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost);
    printf("dependent global load (256MB random chase): %llu cycles\n", h[0]);
    return 0;
}
