// Micro-benchmark: LDS read cost on gfx950 for the record-broadcast patterns of the blend kernels.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o /tmp/ubench_lds && /tmp/ubench_lds
// MODE 0: ds_read_b128, one address for the whole wave (blend2 record broadcast)
// MODE 1: ds_read_b128, one address per 16-lane row, rows on random 16-byte slots (blend3)
// MODE 2: ds_read_b128, one address per row, rows on distinct bank groups
// MODE 3: ds_read_b128, every lane its own consecutive 16 bytes
// MODE 4: ds_read_b32, one address for the whole wave
// MODE 5: ds_read_b32, one address per row
// VALU > 0: that many dependent-free fma per read interleaved (overlap test)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 1024
template <int MODE, int VALU>
__global__ __launch_bounds__(256) void k(float *out, int seed) {
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane >> 4;
    float4 acc = make_float4(0, 0, 0, 0);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + lane;
    unsigned s = seed * 2654435761u + (threadIdx.x >> 6) * 40503u;
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            int idx;
            if (MODE == 0 || MODE == 4) idx = (s >> 8) & 1023;
            else if (MODE == 1 || MODE == 5) idx = ((s >> 8) * (row * 2 + 1) + row * 77) & 1023;
            else if (MODE == 2) idx = (((s >> 8) & 127) * 8 + row * 2) & 1023;
            else idx = ((s >> 8) + lane) & 1023;
            if (MODE >= 4) {
                acc.x += reinterpret_cast<float *>(buf)[idx];
            } else {
                float4 t = buf[idx];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
#pragma unroll
            for (int w = 0; w < VALU; ++w) v[w & 7] = v[w & 7] * 1.0001f + 0.5f;
        }
    }
    float r = acc.x + acc.y + acc.z + acc.w;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE, int VALU> void run(float *d, const char *name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL((k<MODE, VALU>), dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, VALU>), dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    // wave-instructions of the read per CU: 8 WG * 4 waves * N * 8
    double reads_per_cu = 8.0 * 4 * N * 8;
    printf("%-44s VALU/read %2d : %.3f ms  -> %.2f ns per read wave-instr per CU (%.1f clk @2.4GHz)\n", name, VALU, ms,
           ms * 1e6 / reads_per_cu, ms * 1e6 / reads_per_cu * 2.4);
}
int main() {
    float *d; hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    run<0, 0>(d, "b128 wave-uniform");
    run<1, 0>(d, "b128 per-row, random slots");
    run<2, 0>(d, "b128 per-row, distinct bank groups");
    run<3, 0>(d, "b128 per-lane consecutive");
    run<4, 0>(d, "b32 wave-uniform");
    run<5, 0>(d, "b32 per-row");
    run<0, 4>(d, "b128 wave-uniform");
    run<1, 4>(d, "b128 per-row, random slots");
    run<0, 16>(d, "b128 wave-uniform");
    run<1, 16>(d, "b128 per-row, random slots");
    run<4, 16>(d, "b32 wave-uniform (VALU reference)");
    return 0;
}
