// blend_measure.h — measurement-only hooks of the blend kernels (blend2.hip).  Nothing here is compiled into the
// product library: without -DGG_ABLATION (libgg_raster_abl.so, tools/kbench.py, tools/walkstats.py) and -DGG_STAMPS
// (tools/stamps.py) every macro below is empty.
#pragma once
#ifdef GG_ABLATION
// measurement twin only: walk statistics of the forward kernel (tools/walkstats.py)
//   0 list entries staged   1 survivors of the quadrant cull that were walked   2 of those, with >= 1
//   blending pixel   3 (pixel, Gaussian) pairs evaluated   4 pairs passing the alpha test   5 pairs blended
__device__ unsigned long long g_walk_stats[8];
extern "C" int gg_debug_walk_stats(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_walk_stats), sizeof(g_walk_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#ifdef GG_WALK_STATS   // the counters cost 100x the kernel: only tools/walkstats.py builds with them
#define WALK_STAT(i, v) do { if (lane == 0) atomicAdd(&g_walk_stats[i], (unsigned long long)(v)); } while (0)
#else
#define WALK_STAT(i, v) do { } while (0)
#endif
// forward ablation of the pair kernel (tools/kbench.py; template parameter FABL): 1 no MFMAs, 2 also no
// colour-row loads, 3 also no second-array fma, 4 staging only
static int g_fwd_abl = 0;
extern "C" int gg_debug_set_fwd_ablation(int level) {
    const int prev = g_fwd_abl;
    g_fwd_abl = level;
    return prev;
}
#else
#define WALK_STAT(i, v) do { } while (0)
#endif
// GG_STAMPS (tools/stamps.py; a diagnostic build of its own, never timed as a whole): s_memtime stamps around the
// phases of the wide backward.  Every wave adds the cycles it spent per phase into g_stamp_sums:
//   0 prologue  1 staging (list ids, records, cull, queue)  2 batch start: colour-row loads issued and waited for
//   3 D product (MFMAs + slab stores)  4 walk  5 flush MFMAs  6 32-channel colour atomics  7 second-array flush
//   8 queue compaction  9 wave lifetime  10 batches  11 waves
#ifdef GG_STAMPS
__device__ unsigned long long g_stamp_sums[16];
extern "C" int gg_debug_stamps(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_sums), sizeof(g_stamp_sums)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sums), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
__device__ __forceinline__ unsigned long long gg_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP_DECL unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned st_batches = 0; \
                   const unsigned long long st_t0 = gg_stamp(); unsigned long long st_prev = st_t0
#define STAMP(i) do { const unsigned long long st_now = gg_stamp(); st_sum[i] += st_now - st_prev; st_prev = st_now; } while (0)
#define STAMP_BATCH() (++st_batches)
#define STAMP_END() do { st_sum[9] = gg_stamp() - st_t0; if (lane == 0) { \
        for (int q_ = 0; q_ < 10; ++q_) atomicAdd(&g_stamp_sums[q_], st_sum[q_]); \
        atomicAdd(&g_stamp_sums[10], (unsigned long long)st_batches); atomicAdd(&g_stamp_sums[11], 1ull); } } while (0)
#else
#define STAMP_DECL do { } while (0)
#define STAMP(i) do { } while (0)
#define STAMP_BATCH() do { } while (0)
#define STAMP_END() do { } while (0)
#endif
#ifdef GG_STAMPS
#ifndef GG_EPI_SKIP
#define GG_EPI_SKIP 0
#endif
extern "C" int gg_debug_epi_skip() { return GG_EPI_SKIP; }
#endif
