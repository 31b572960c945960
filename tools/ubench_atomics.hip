// Micro-benchmark: the ceiling of float atomic adds that go to memory-side (L2) execution on gfx950, in the access
// shapes of the pair backward's flushes: a wave instruction whose 64 lanes cover R rows (of a table of 1 M rows) with
// 64 / R consecutive floats each.  4 096 resident waves (16 per CU), every wave issues ITER instructions back to back;
// rows are pseudo-random per instruction (a hash of wave and iteration), so nothing coalesces across instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_atomics.hip -o /tmp/ubench_atomics && /tmp/ubench_atomics
// Prints instructions/s, 64-byte segments/s and bytes/s per shape.  (The pair backward issues per batch of 16 Gaussians:
// 8 instructions of shape R=4 x 16 floats (feature rows), 4 of R=4 x 8 floats, 4 of R=4 x 6 floats: 64 row segments.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned hashu(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// rows: R per instruction; width: consecutive floats per row; stride: floats per table row
template <int R>
__global__ __launch_bounds__(256) void k(float *table, int nrows, int stride, int width, int iters, int unique) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = 64 / R, r = lane / per, c = lane % per;
    const bool on = c < width;
    for (int it = 0; it < iters; ++it) {
        const unsigned h = hashu(wave * 9781u + it * 6271u + (unique ? r * 977u : 0u) + 1u);
        const size_t row = (h % (unsigned)nrows);
        if (on) atomicAdd(table + (row + (unique ? 0 : r)) % nrows * stride + c, 1.0f);
    }
}
template <int R>
static void run(float *d, int nrows, int stride, int width, const char *what) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<R>, dim3(1024), dim3(256), 0, 0, d, nrows, stride, width, 50, 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<R>, dim3(1024), dim3(256), 0, 0, d, nrows, stride, width, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = 4096.0 * iters, segs = inst * R;
    printf("%-44s %7.3f ms  %6.2f G inst/s  %6.2f G rows/s  %7.1f GB/s of operands\n", what, ms, inst / ms * 1e-6,
           segs / ms * 1e-6, segs * width * 4 / ms * 1e-6);
}
int main() {
    const int nrows = 1 << 20;
    float *d;
    hipMalloc(&d, sizeof(float) * (size_t)nrows * 64);
    hipMemset(d, 0, sizeof(float) * (size_t)nrows * 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(d, nrows, 32, 16, "4 rows x 16 floats (feature half rows)");
        run<2>(d, nrows, 32, 32, "2 rows x 32 floats (whole feature rows)");
        run<1>(d, nrows, 64, 64, "1 row  x 64 floats");
        run<4>(d, nrows, 13, 6,  "4 rows x 6 floats, stride 13 (geometry)");
        run<4>(d, nrows, 13, 13, "4 rows x 13 floats, stride 13 (geometry+tail)");
        run<8>(d, nrows, 13, 8,  "8 rows x 8 floats, stride 13");
        run<16>(d, nrows, 13, 4, "16 rows x 4 floats, stride 13");
    }
    return 0;
}
