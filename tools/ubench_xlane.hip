// Micro-benchmark: cost of the cross-lane primitives used by the gradient butterfly on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_xlane.hip -o tools/ubench_xlane && ./tools/ubench_xlane
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f + i; }
    const int lane = threadIdx.x & 63;
    const int addr32 = ((lane ^ 32) << 2), addr16 = ((lane ^ 16) << 2);
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // plain VALU: 2 ops
                a[i] = a[i] * 1.0001f + b[i];
                b[i] = b[i] + a[i];
            } else if (MODE == 1) {     // permlane32_swap + add
                auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a[i]), __builtin_bit_cast(unsigned, b[i]), false, false);
                a[i] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
                b[i] = b[i] + 1.0f;
            } else if (MODE == 2) {     // permlane16_swap + add
                auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a[i]), __builtin_bit_cast(unsigned, b[i]), false, false);
                a[i] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
                b[i] = b[i] + 1.0f;
            } else if (MODE == 3) {     // dpp add (row half mirror)
                a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x141, 0xf, 0xf, true));
                b[i] = b[i] + 1.0f;
            } else if (MODE == 4) {     // bpermute xor32: select + bpermute + add
                float send = (lane & 32) ? a[i] : b[i];
                float keep = (lane & 32) ? b[i] : a[i];
                float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr32, __builtin_bit_cast(int, send)));
                a[i] = keep + got;
                b[i] = b[i] + 1.0f;
            } else if (MODE == 5) {     // ds_swizzle xor16 (bitmode and=0x1f or=0 xor=0x10)
                float send = (lane & 16) ? a[i] : b[i];
                float keep = (lane & 16) ? b[i] : a[i];
                float got = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x401F));
                a[i] = keep + got;
                b[i] = b[i] + 1.0f;
            } else if (MODE == 6) {     // dpp row_ror:8 based xor8 exchange-add (full add, no halving)
                a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x128, 0xf, 0xf, true));
                b[i] = b[i] + 1.0f;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const char *names[] = {"valu x2 (fma+add)", "permlane32_swap+add(+1 add)", "permlane16_swap+add(+1 add)", "dpp add (+1 add)", "sel2+bpermute+add(+1 add)", "sel2+ds_swizzle+add(+1 add)", "dpp row_ror8 add(+1 add)"};
    float t[7] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d)};
    // per SIMD: blocks*4 waves / 1024 SIMDs = 8 waves per SIMD, each N*8 item-iterations
    for (int m = 0; m < 7; ++m) {
        double item_ns = t[m] * 1e6 / (8.0 * N * 8);   // ns per (item-iteration) per SIMD-slot
        printf("%-32s %8.3f ms   %6.2f ns per item per SIMD (8 waves/SIMD)\n", names[m], t[m], item_ns);
    }
    return 0;
}
