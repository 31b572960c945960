// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 on gfx950 (what the MLP kernel's roof really is
// on this box).  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int WITH_LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    __shared__ float buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = seed + i;
    __syncthreads();
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const float *row = buf + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            float bb = WITH_LDS ? row[s * 64] : b;
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int WITH_LDS> void run(float *d, int wgs_per_cu, const char *name) {
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, WITH_LDS>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, WITH_LDS>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)256 * wgs_per_cu * 4 * iters * 64 * CHAINS * 4096.0;
    printf("%-40s %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float *d; hipMalloc(&d, sizeof(float) * 256 * 4 * 256);
    run<1, 0>(d, 1, "1 chain, register operands");
    run<2, 0>(d, 1, "2 chains, register operands");
    run<4, 0>(d, 1, "4 chains, register operands");
    run<2, 0>(d, 2, "2 chains, register operands");
    run<2, 1>(d, 1, "2 chains, B operand from LDS");
    run<2, 1>(d, 2, "2 chains, B operand from LDS");
    run<4, 1>(d, 1, "4 chains, B operand from LDS");
    return 0;
}
