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

// ---------------------------------------------------------------------------------------------
// Packed pairs.  gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — two independent IEEE fp32
// operations per lane — at the rate of one scalar-typed VALU instruction, and the blend kernels are
// VALU-issue bound, so everything that is computed for two Gaussians of a group alike is written on
// 2-vectors.  Each component is the same correctly rounded operation sequence as the scalar code:
// results are bit-identical (gg_expf2(x).k == gg_expf(x.k)).
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f gg_splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f gg_fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f gg_expf2(v2f x) {
    const v2f t = x * gg_splat(GG_EXP_LOG2E);
    const v2f n = (v2f){__builtin_rintf(t.x), __builtin_rintf(t.y)};
    v2f r = gg_fma2(n, gg_splat(-GG_EXP_LN2_HI), x);
    r = gg_fma2(n, gg_splat(-GG_EXP_LN2_LO), r);
    v2f p = gg_splat(GG_EXP_P0);
    p = gg_fma2(p, r, gg_splat(GG_EXP_P1));
    p = gg_fma2(p, r, gg_splat(GG_EXP_P2));
    p = gg_fma2(p, r, gg_splat(GG_EXP_P3));
    p = gg_fma2(p, r, gg_splat(GG_EXP_P4));
    p = gg_fma2(p, r, gg_splat(GG_EXP_P5));
    const v2f z = r * r;
    v2f y = gg_fma2(p, z, r);
    y = y + gg_splat(1.0f);
    const v2f s = (v2f){__builtin_bit_cast(float, (uint32_t)((int)n.x + 127) << 23),
                        __builtin_bit_cast(float, (uint32_t)((int)n.y + 127) << 23)};
    const v2f e = y * s;
    return (v2f){(x.x < GG_EXP_LO) ? 0.0f : e.x, (x.y < GG_EXP_LO) ? 0.0f : e.y};
}
// d = centre - pixel and sigma = 0.5*(a dx^2 + c dy^2) + b dx dy of two list records at one pixel, in
// the operation order of the scalar kernels: fma(0.5, fma(a*dx, dx, (c*dy)*dy), (b*dx)*dy)
__device__ __forceinline__ v2f gg_sigma2(const float4 &A0, const float4 &B0, const float4 &A1,
                                        const float4 &B1, float px, float py, v2f &dx, v2f &dy) {
    dx = (v2f){A0.x, A1.x} - gg_splat(px);
    dy = (v2f){A0.y, A1.y} - gg_splat(py);
    const v2f ca = (v2f){B0.x, B1.x}, cb = (v2f){B0.y, B1.y}, cc = (v2f){B0.z, B1.z};
    return gg_fma2(gg_splat(0.5f), gg_fma2(ca * dx, dx, (cc * dy) * dy), (cb * dx) * dy);
}

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
