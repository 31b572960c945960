// Micro-benchmark: do fp32 MFMAs and fp32 VALU instructions of DIFFERENT waves of one SIMD execute side by side on
// gfx950, or do they share one issue budget?  (MI355X_MICROARCH.md: the f32-input MFMA runs at the f32 VECTOR rate,
// 64 FLOP/clk/SIMD; the guide's "matrix and vector pipes are separate" figures are bf16 measurements.)
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 land pairwise on the four SIMDs.  Modes:
//   M   every wave runs N fp32 MFMAs (16x16x4)            V   every wave runs K fp32 FMAs
//   MV  waves 0-3 run the MFMAs, waves 4-7 the FMAs       M1 / V1: only waves 0-3 work (one wave per SIMD)
// If the pipes co-execute, t(MV) ~ max(t(M1), t(V1)); if they share the issue budget, t(MV) ~ t(M1) + t(V1).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_coexec.hip -o /tmp/ubench_coexec && /tmp/ubench_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mfma_work(int iters, float a, float b) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);   // 64 MFMAs
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    return s;
}
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
__device__ __forceinline__ float mfma_bf16_work(int iters, float a, float b) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (__bf16)(a + i); vb[i] = (__bf16)(b - 0.01f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[c], 0, 0, 0);   // 64 MFMAs
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    return s;
}
__device__ __forceinline__ float valu_work(int iters, float a, float b) {
    float x[8];
    for (int c = 0; c < 8; ++c) x[c] = a + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = __builtin_fmaf(x[c], b, a);   // 512 independent-ish FMAs (8 chains)
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += x[c];
    return s;
}
// mode bits: 1 = waves 0-3 MFMA, 2 = waves 4-7 MFMA, 4 = waves 0-3 VALU, 8 = waves 4-7 VALU, 16 = waves 0-3 bf16 MFMA,
//            32 = waves 4-7 bf16 MFMA
__global__ __launch_bounds__(512) void k(float *out, int mode, int im, int iv, float seed) {
    const int wave = threadIdx.x >> 6;
    const float a = seed + (threadIdx.x & 63) * 1e-3f, b = 0.999f;
    float s = 0.f;
    const int lo = wave < 4;
    if ((lo && (mode & 1)) || (!lo && (mode & 2))) s += mfma_work(im, a, b);
    if ((lo && (mode & 4)) || (!lo && (mode & 8))) s += valu_work(iv, a, b);
    if ((lo && (mode & 16)) || (!lo && (mode & 32))) s += mfma_bf16_work(2 * im, a, b);   // 16 cycles each: the same time
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
static float run(float *d, int mode, int im, int iv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, 4, 4, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, mode, im, iv, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * 512);
    const int im = 2000;            // 128 k MFMAs of 32 cycles per wave
    const int iv = 2000;            // 1 M FMAs per wave
    for (int rep = 0; rep < 2; ++rep) {
        const float m1 = run(d, 1, im, iv), v1 = run(d, 4, im, iv);
        const float m2 = run(d, 3, im, iv), v2 = run(d, 12, im, iv);
        const float mv = run(d, 1 | 8, im, iv);
        const float b1 = run(d, 16, im, iv), b2 = run(d, 48, im, iv), bv = run(d, 16 | 8, im, iv), bm = run(d, 16 | 2, im, iv);
        printf("bf16 MFMA 16x16x32: one wave per SIMD %.3f ms, two %.3f ms; beside fp32 VALU %.3f ms (max %.3f, sum %.3f); "
               "beside fp32 MFMA %.3f ms (sum %.3f)\n", b1, b2, bv, b1 > v1 ? b1 : v1, b1 + v1, bm, b1 + m1);
        printf("one wave per SIMD:  MFMA %.3f ms   VALU %.3f ms\n", m1, v1);
        printf("two waves per SIMD: MFMA+MFMA %.3f ms   VALU+VALU %.3f ms   MFMA beside VALU %.3f ms "
               "(max %.3f, sum %.3f)\n", m2, v2, mv, m1 > v1 ? m1 : v1, m1 + v1);
    }
    return 0;
}
