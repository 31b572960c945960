// gg_common.h — shared by the gfx950 kernels of libgg_raster.so (not by the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gg_constants.h"
#include "../../include/gg_raster.h"

#define GG_WAVE 64

void gg_set_error(const char *fmt, ...);

#define GG_REQUIRE(cond, msg)                      \
    do {                                           \
        if (!(cond)) {                             \
            gg_set_error("%s: %s", __func__, msg); \
            return GG_ERR_INVALID_ARG;             \
        }                                          \
    } while (0)

#define GG_CHECK_LAUNCH()                                                   \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            gg_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return GG_ERR_LAUNCH;                                           \
        }                                                                   \
    } while (0)

// measurement hooks (prof.hip); no-ops unless gg_prof_enable(1)
void gg_prof_begin(int kernel_id, hipStream_t s);
void gg_prof_end(int kernel_id, hipStream_t s);
static inline int gg_width_index(int w) {
    return w <= 1 ? 0 : w <= 3 ? 1 : w == 4 ? 2 : w == 8 ? 3 : w == 16 ? 4 : 5;
}

static inline size_t gg_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// gg_fill_async: hipMemsetAsync as an ordinary kernel.  On this runtime a hipMemsetAsync between two kernels
// of a stream leaves ~6 us of idle GPU on either side of it (rocprofv3 kernel trace of a bench view: four
// memsets per view = ~50 us of a 2.1 ms view); a kernel launch does not.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void gg_fill_kernel(uint32_t *__restrict__ p, uint32_t v, size_t nwords,
                                                             int vec) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (vec && i + 3 < nwords) {
        *reinterpret_cast<uint4 *>(p + i) = make_uint4(v, v, v, v);
    } else {
        for (int k = 0; k < 4; ++k)
            if (i + k < nwords) p[i + k] = v;
    }
}
static inline hipError_t gg_fill_async(void *p, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return hipMemsetAsync(p, byte, bytes, s);
    uint32_t v = (uint32_t)byte & 0xffu;
    v |= v << 8;
    v |= v << 16;
    const size_t nwords = bytes / 4;
    const unsigned blocks = (unsigned)((nwords + 1023) / 1024);
    hipLaunchKernelGGL(gg_fill_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t *>(p), v, nwords,
                       (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? 1 : 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gg_expf: the operation sequence documented in gg_constants.h.  Every step is a correctly
// rounded fp32 instruction (v_mul, v_rndne, v_fma, v_add, integer shift), the file is built
// with -ffp-contract=off, so the result is bit-identical to oracle/gg_oracle.c:gg_exp.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gg_expf(float x) {
    float t = x * GG_EXP_LOG2E;
    float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -GG_EXP_LN2_HI, x);
    r = __builtin_fmaf(n, -GG_EXP_LN2_LO, r);
    float p = GG_EXP_P0;
    p = __builtin_fmaf(p, r, GG_EXP_P1);
    p = __builtin_fmaf(p, r, GG_EXP_P2);
    p = __builtin_fmaf(p, r, GG_EXP_P3);
    p = __builtin_fmaf(p, r, GG_EXP_P4);
    p = __builtin_fmaf(p, r, GG_EXP_P5);
    float z = r * r;
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    float s = __builtin_bit_cast(float, (uint32_t)((int)n + 127) << 23);
    float e = y * s;
    return (x < GG_EXP_LO) ? 0.0f : e;
}

// The blend walks' form: same sequence up to y, then y * 2^n as ONE v_ldexp_f32 instead of integer add, shift and
// multiply, and no clamp below GG_EXP_LO (cmp + cndmask): 14 instructions instead of 18.  Bit-identical to gg_expf
// wherever the result is >= 2^-126 (a multiplication by a power of two is exact); below that it returns a
// denormal or 0 where gg_expf returns 0 — the walks only use exp(-sigma) through alpha = opacity * exp >= 1/255,
// so every pair they keep sees the same bits and every pair they drop is dropped by both.
__device__ __forceinline__ float gg_expf_walk(float x) {
#ifdef GG_EXP_PROBE
    // TIMING PROBE ONLY (tools/pairbench.py on a variant build): the hardware's v_exp_f32 —
    // 2 instructions instead of 14, values NOT the oracle's (alpha / stop decisions flip): never in the product library
    return __builtin_amdgcn_exp2f(x * GG_EXP_LOG2E);
#endif
    float t = x * GG_EXP_LOG2E;
    float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -GG_EXP_LN2_HI, x);
    r = __builtin_fmaf(n, -GG_EXP_LN2_LO, r);
    float p = GG_EXP_P0;
    p = __builtin_fmaf(p, r, GG_EXP_P1);
    p = __builtin_fmaf(p, r, GG_EXP_P2);
    p = __builtin_fmaf(p, r, GG_EXP_P3);
    p = __builtin_fmaf(p, r, GG_EXP_P4);
    p = __builtin_fmaf(p, r, GG_EXP_P5);
    float z = r * r;
    float y = __builtin_fmaf(p, z, r);
    y = y + 1.0f;
    return __builtin_amdgcn_ldexpf(y, (int)n);
}

// ---------------------------------------------------------------------------------------------
// fp32 products on the fp16 matrix rate (round 3: the 16-slot blend backward's dense products, csrc/blend2.hip, and
// the feature MLP's fast forward, csrc/mlp.hip)
// ---------------------------------------------------------------------------------------------
// x s = hi + lo with hi = RNE_fp16(x s), lo = RNE_fp16(x s - hi): 2^-24 relative when both pieces are normal fp16 numbers,
// i.e. for elements within 2^-12 of the largest one the scale s (a power of two, exact) was chosen for; below that the
// absolute error stays <= 2^-40 of that largest element.  A[M x 32] * B[32 x N] is then four v_mfma_f32_16x16x32_f16
// (lo lo, lo hi, hi lo, hi hi: 16 cycles each, fp32 accumulation) instead of eight v_mfma_f32_16x16x4_f32 (32 cycles
// each): measured MORE accurate than the fp32 instruction against a double sum (tools/check_f16split.hip:
// 4-5e-8 of sum|terms| against 8-9e-8) because the piece products are exact in fp32.  Six VALU instructions per
// pair of values to split.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &hi, unsigned &lo) {
    const h16x2 h = __builtin_convertvector((f32x2){x0, x1}, h16x2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const h16x2 l = __builtin_convertvector((f32x2){x0 - back[0], x1 - back[1]}, h16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// the power of two that brings |m| into [2^14, 2^15); at most 2^126 (m = 0 or tiny: the pieces are zeros whatever the
// scale), 2^-114 for inf / nan (which stay inf / nan).  Branch-free: three instructions.
__device__ __forceinline__ float pow2_scale(float m) {
    const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;
    return __builtin_bit_cast(float, min(268u - e, 253u) << 23);
}
// 1 / s for a power of two s (exact)
__device__ __forceinline__ float pow2_inv(float s) {
    return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s));
}
#define H8(a, b, c, d) __builtin_bit_cast(h16x8, (u32x4){(a), (b), (c), (d)})

// ---------------------------------------------------------------------------------------------
// Tile bounding box of a projected Gaussian (same float-domain clamp as the oracle).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int gg_clampi_f(float v, int bound) {
    return (int)fminf(fmaxf(v, 0.0f), (float)bound);
}
__device__ __forceinline__ void gg_tile_bbox(float cx, float cy, float radius, int tiles_x,
                                             int tiles_y, int &x0, int &y0, int &x1, int &y1) {
    float tcx = cx / (float)GG_BLOCK, tcy = cy / (float)GG_BLOCK;
    float tr = radius / (float)GG_BLOCK;
    x0 = gg_clampi_f(tcx - tr, tiles_x);
    x1 = gg_clampi_f((tcx + tr) + 1.0f, tiles_x);
    y0 = gg_clampi_f(tcy - tr, tiles_y);
    y1 = gg_clampi_f((tcy + tr) + 1.0f, tiles_y);
}
