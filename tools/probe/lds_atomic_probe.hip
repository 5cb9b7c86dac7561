// Diagnostic: cost of LDS atomic adds per wave instruction as a function of active lanes and same-address conflicts (one workgroup of 512 threads,
// clock64 around an unrolled run of 64 ds_add_u32 per lane).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/lds_atomic_probe tools/probe/lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(int mode, int active, long long *out) {
    __shared__ int h[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < 4096; t += 512) h[t] = 0;
    __syncthreads();
    int idx;
    if (mode == 0) idx = tid;                                  // conflict-free, every lane its own word
    else if (mode == 1) idx = (lane & 7) + 8 * (tid >> 6);      // 8 lanes per word (8-way same-address conflicts)
    else idx = (lane & 1) + 8 * (tid >> 6);                     // 32-way
    const bool on = lane < active;
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (on) atomicAdd(&h[(idx + 64 * (i & 7)) & 4095], 1);
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    if (h[tid] == 123456789) out[1] = 1;
}
int main() {
    long long *d, hst[2];
    hipMalloc(&d, 16);
    for (int mode = 0; mode < 3; ++mode)
        for (int active : {64, 32, 16, 8, 4, 1}) {
            long long best = 1ll << 60;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, mode, active, d);
                hipMemcpy(hst, d, 16, hipMemcpyDeviceToHost);
                if (hst[0] < best) best = hst[0];
            }
            printf("mode %d (%s) active lanes %2d: %lld cycles for 64 atomics x 8 waves = %.1f cycles per wave instruction\n", mode,
                   mode == 0 ? "no conflict" : (mode == 1 ? "8 lanes/word" : "32 lanes/word"), active, best, best / 512.0);
        }
    return 0;
}
