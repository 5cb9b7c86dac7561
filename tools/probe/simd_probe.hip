// Diagnostic (not product): which SIMD does wave w of a 256-thread workgroup land on when four workgroups share a CU (40 KB of LDS each, the compact
// throughput instance's shape)?  HW_REG_HW_ID (gfx9: id 4): wave_id [3:0], simd_id [5:4], cu_id [11:8].  Prints, per wave index, how many workgroups had it on SIMD 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    // keep the workgroup resident for a while so that four really share a CU
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
}
int main() {
    const int nb = 1024;
    unsigned *d; hipMalloc(&d, nb * 4 * 4);
    std::vector<unsigned> h(nb * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 40960, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    int cnt[4][4] = {};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) cnt[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, cnt[w][0], cnt[w][1], cnt[w][2], cnt[w][3]);
    printf("first workgroups (hw_id of waves 0..3): ");
    for (int b = 0; b < 6; ++b) printf("[%x %x %x %x] ", h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    printf("\n");
    return 0;
}
