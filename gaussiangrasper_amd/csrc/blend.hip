// blend.hip — C ABI of the alpha-blending step and the record-packing kernel.
// The blend kernels themselves live in blend2.hip (their design notes are in its header and in
// DESIGN.md §3.4-3.5).
#include <stdlib.h>

#include "blend_common.h"

// ---------------------------------------------------------------------------------------------
// per-call record of the narrow path: {opacity, colour 0..2 of this channel chunk} per Gaussian — the
// one 16-byte gather a surviving list entry makes (32-channel chunks gather the opacity alone)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_prep_kernel(int N, int C, int ch_off, int nch,
                                                         const float *__restrict__ opacity,
                                                         const float *__restrict__ colors,
                                                         float4 *__restrict__ crec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *c = colors + (size_t)i * C + ch_off;
    crec[i] = make_float4(opacity[i], c[0], nch > 1 ? c[1] : 0.0f, nch > 2 ? c[2] : 0.0f);
}

// ---------------------------------------------------------------------------------------------
// tile-sorted geometry stream: geo[e] = geometry of Gaussian ids[e] (blend_common.h, GG_GEO_BYTES)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geo_sort_kernel(long I, const int32_t *__restrict__ ids,
                                                       const float *__restrict__ xys,
                                                       const float *__restrict__ conics,
                                                       float4 *__restrict__ geo) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= I) return;
    const int g = ids[e];
    const float2 xy = reinterpret_cast<const float2 *>(xys)[g];
    const float *cn = conics + 3 * (size_t)g;
    geo[2 * e] = make_float4(xy.x, xy.y, cn[0], cn[1]);
    geo[2 * e + 1] = make_float4(cn[2], __builtin_bit_cast(float, g), 0.0f, 0.0f);
}

extern "C" size_t gg_geo_sort_bytes(int64_t num_intersects) {
    return gg_align_up((size_t)GG_GEO_BYTES * (size_t)(num_intersects > 0 ? num_intersects : 1), 256);
}
extern "C" int gg_geo_sort(int64_t I, const int32_t *ids, const float *xys, const float *conics,
                           void *geo_sorted, gg_stream_t stream) {
    GG_REQUIRE(I >= 0, "num_intersects < 0");
    if (I == 0) return GG_OK;
    GG_REQUIRE(ids && xys && conics && geo_sorted, "null pointer");
    GG_REQUIRE((((uintptr_t)geo_sorted | (uintptr_t)xys) & 7) == 0 && ((uintptr_t)geo_sorted & 15) == 0,
               "geo_sorted must be 16-byte and xys 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_GEO_SORT, s);
    hipLaunchKernelGGL(geo_sort_kernel, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, s, (long)I, ids,
                       xys, conics, (float4 *)geo_sorted);
    gg_prof_end(GG_K_GEO_SORT, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// workspace: [crec: 16 B x N][geometry stream: only used when the caller passes geo_sorted = NULL]
static size_t crec_bytes(int N) { return gg_align_up(16 * (size_t)(N > 0 ? N : 1), 256); }
extern "C" size_t gg_blend_workspace(int num_points, int64_t num_intersects) {
    return crec_bytes(num_points) + gg_geo_sort_bytes(num_intersects);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
void gg_launch_blend2_fwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const float4 *geo, const int2 *bins, const float4 *crec,
                          const float *opacity,
                          const float *colors, const float *background, float *out_img,
                          float *final_Ts, int32_t *final_idx, int write_final, hipStream_t s);
void gg_launch_blend2_bwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const float4 *geo, const int2 *bins, const float4 *crec,
                          const float *opacity,
                          const float *colors, const float *background, const float *final_Ts,
                          const int32_t *final_idx, const float *v_out, float *v_xy, float *v_conic,
                          float *v_colors, float *v_opacity, hipStream_t s);
void gg_launch_blend2_bwd_ablate(int abl, int C, int off, int img_h, int img_w, int tiles_x,
                                 int ntiles, const float4 *geo, const int2 *bins, const float4 *crec,
                          const float *opacity,
                                 const float *colors, const float *background, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_opacity, hipStream_t s);

// Channel chunking: calls with <= 3 channels (rgb / depth / normal) use the narrow kernels with
// the colours inside the LDS record; anything wider is processed in chunks of 32 channels on the
// MFMA kernels (a final partial chunk is zero-padded), each chunk re-walking the tile lists.
// (A 64-channel-per-walk variant — two 32-column MFMA blocks, V_OUT operands streamed per flush —
// was built and measured: 231 VGPRs / occupancy 2 backward, 143 registers forward; the fused
// 39-channel call ran at 98 views/s against 122 with two 32-channel walks, so it was dropped.)
static int chunk_width(int remaining) { return remaining <= 3 ? remaining : 32; }

static int g_ablate = 0;  // measurement only
extern "C" int gg_debug_set_ablation(int level) {
    int prev = g_ablate;
    g_ablate = level;
    return prev;
}

// geometry stream of this call: the caller's (gg_geo_sort, shared by the calls of a view) or one
// built here into the workspace
static int resolve_geo(const char *who, int N, int64_t I, const int32_t *ids, const float *xys,
                       const float *conics, const void *geo_sorted, void *ws, size_t ws_bytes,
                       hipStream_t s, const float4 **geo) {
    const size_t need = crec_bytes(N) + (geo_sorted ? 0 : gg_geo_sort_bytes(I));
    if (ws == nullptr || ws_bytes < need) {
        gg_set_error("%s: workspace too small (%zu < %zu bytes)", who, ws_bytes, need);
        return GG_ERR_WORKSPACE;
    }
    if (geo_sorted) {
        GG_REQUIRE(((uintptr_t)geo_sorted & 15) == 0, "geo_sorted must be 16-byte aligned");
        *geo = (const float4 *)geo_sorted;
        return GG_OK;
    }
    void *mine = (char *)ws + crec_bytes(N);
    *geo = (const float4 *)mine;
    return gg_geo_sort(I, ids, xys, conics, mine, (gg_stream_t)s);
}

static void launch_prep(int N, int C, int off, int n, const float *opacity, const float *colors,
                        float4 *crec, hipStream_t s) {
    gg_prof_begin(GG_K_BLEND_PREP, s);
    hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, C, off, n, opacity,
                       colors, crec);
    gg_prof_end(GG_K_BLEND_PREP, s);
}

extern "C" int gg_blend_fwd(int C, int N, int64_t I, int img_h, int img_w, const int32_t *ids,
                            const int32_t *tile_bins, const float *xys, const float *conics,
                            const float *colors, const float *opacity, const float *background,
                            const void *geo_sorted, float *out_img, float *final_Ts,
                            int32_t *final_idx, void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(C >= 1, "channels < 1");
    GG_REQUIRE(N >= 0 && I >= 0, "num_points or num_intersects < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    GG_REQUIRE(tile_bins && background && out_img && final_Ts && final_idx, "null pointer");
    GG_REQUIRE(N == 0 || (xys && conics && colors && opacity), "null pointer");
    GG_REQUIRE(I == 0 || (N > 0 && (ids || geo_sorted)), "intersections without ids / Gaussians");
    hipStream_t s = (hipStream_t)stream;
    const float4 *geo = nullptr;
    if (I > 0) {
        int rc = resolve_geo("gg_blend_fwd", N, I, ids, xys, conics, geo_sorted, ws, ws_bytes, s, &geo);
        if (rc != GG_OK) return rc;
    }
    float4 *crec = (float4 *)ws;
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    for (int off = 0; off < C;) {
        const int w = chunk_width(C - off);
        const int n = min(w, C - off);
        if (w <= 3 && I > 0) launch_prep(N, C, off, n, opacity, colors, crec, s);
        gg_prof_begin(GG_K_BLEND_FWD + gg_width_index(w), s);
        gg_launch_blend2_fwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, geo, (const int2 *)tile_bins,
                             crec, opacity, colors, background, out_img, final_Ts, final_idx, off == 0, s);
        gg_prof_end(GG_K_BLEND_FWD + gg_width_index(w), s);
        off += n;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_blend_bwd(int C, int N, int64_t I, int img_h, int img_w, const int32_t *ids,
                            const int32_t *tile_bins, const float *xys, const float *conics,
                            const float *colors, const float *opacity, const float *background,
                            const void *geo_sorted, const float *final_Ts, const int32_t *final_idx,
                            const float *v_out, float *v_xy, float *v_conic, float *v_colors,
                            float *v_opacity, void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(C >= 1, "channels < 1");
    GG_REQUIRE(N >= 0 && I >= 0, "num_points or num_intersects < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    if (N == 0) return GG_OK;
    GG_REQUIRE(tile_bins && xys && conics && colors && opacity && background && final_Ts &&
                   final_idx && v_out && v_xy && v_conic && v_colors && v_opacity,
               "null pointer");
    GG_REQUIRE(I == 0 || ids || geo_sorted, "intersections without ids");
    hipStream_t s = (hipStream_t)stream;
    const float4 *geo = nullptr;
    if (I > 0) {
        int rc = resolve_geo("gg_blend_bwd", N, I, ids, xys, conics, geo_sorted, ws, ws_bytes, s, &geo);
        if (rc != GG_OK) return rc;
    }
    float4 *crec = (float4 *)ws;
    // the kernels accumulate with atomics: the four gradient arrays start at zero (one memset when
    // the caller laid them out back to back: v_xy | v_conic | v_opacity | v_colors)
    bool fail;
    if (v_conic == v_xy + 2 * (size_t)N && v_opacity == v_conic + 3 * (size_t)N &&
        v_colors == v_opacity + (size_t)N) {
        fail = hipMemsetAsync(v_xy, 0, sizeof(float) * (6 + (size_t)C) * (size_t)N, s) != hipSuccess;
    } else {
        fail = hipMemsetAsync(v_xy, 0, sizeof(float) * 2 * (size_t)N, s) != hipSuccess;
        fail |= hipMemsetAsync(v_conic, 0, sizeof(float) * 3 * (size_t)N, s) != hipSuccess;
        fail |= hipMemsetAsync(v_colors, 0, sizeof(float) * (size_t)C * (size_t)N, s) != hipSuccess;
        fail |= hipMemsetAsync(v_opacity, 0, sizeof(float) * (size_t)N, s) != hipSuccess;
    }
    if (fail) {
        gg_set_error("gg_blend_bwd: memset failed");
        return GG_ERR_LAUNCH;
    }
    if (I == 0) return GG_OK;
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    for (int off = 0; off < C;) {
        const int w = chunk_width(C - off);
        const int n = min(w, C - off);
        if (w <= 3) launch_prep(N, C, off, n, opacity, colors, crec, s);
        gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(w), s);
        if ((w == 3 && g_ablate > 0 && g_ablate < 10) || (w == 32 && n == 32 && g_ablate > 10))
            gg_launch_blend2_bwd_ablate(g_ablate, C, off, img_h, img_w, tiles_x, ntiles, geo,
                                        (const int2 *)tile_bins, crec, opacity, colors, background,
                                        final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s);
        else
            gg_launch_blend2_bwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, geo,
                                 (const int2 *)tile_bins, crec, opacity, colors, background, final_Ts,
                                 final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s);
        gg_prof_end(GG_K_BLEND_BWD + gg_width_index(w), s);
        off += n;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}
