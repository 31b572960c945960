// blend.hip — C ABI of the alpha-blending step and the record-packing kernel.
// The blend kernels themselves live in blend2.hip (their design notes are in its header and in
// DESIGN.md §3.4-3.5).
#include <stdlib.h>

#include "blend_common.h"

// ---------------------------------------------------------------------------------------------
// prep: pack xys / conics / opacity into 32-byte records (one gather per list entry later)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_prep_kernel(int N, const float *__restrict__ xys,
                                                         const float *__restrict__ conics,
                                                         const float *__restrict__ opacity,
                                                         GRec *__restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    grec_pack(xys[2 * (size_t)i], xys[2 * (size_t)i + 1], opacity[i], conics[3 * (size_t)i], conics[3 * (size_t)i + 1],
              conics[3 * (size_t)i + 2], rec + i);
}

extern "C" size_t gg_blend_workspace(int num_points) {
    return gg_align_up(sizeof(GRec) * (size_t)(num_points > 0 ? num_points : 1), 256);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
void gg_launch_blend2_fwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, float *out_img,
                          float *final_Ts, int32_t *final_idx, int write_final, hipStream_t s);
void gg_launch_blend2_bwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, const float *final_Ts,
                          const int32_t *final_idx, const float *v_out, float *v_xy, float *v_conic,
                          float *v_colors, float *v_opacity, int gstride, int cstride, hipStream_t s,
                          DetSlab det);
size_t gg_sort_pairs_workspace(int64_t n);   // binning.hip: stable LSD radix sort of (key, value) pairs
int gg_sort_pairs(int64_t n, uint32_t *keys, uint32_t *vals, int bits, void *ws, size_t ws_bytes, hipStream_t s);
#ifdef GG_ABLATION
void gg_launch_blend2_bwd_ablate(int abl, int C, int off, int img_h, int img_w, int tiles_x,
                                 int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                                 const float *colors, const float *background, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_opacity, int gstride, int cstride, hipStream_t s);
#endif

void gg_launch_blend2_bwd_pair(int C, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *ids,
                               const int2 *bins, const GRec *rec, const float *colors, const float *background,
                               const float *final_Ts, const int32_t *final_idx, const float *v_out, float *v_xy,
                               float *v_conic, float *v_colors, float *v_opacity, int gstride, int cstride,
                               const float *colors2, int C2, const float *background2,
                               const float *const *v_out2_parts, const int *v_out2_channels, int num_parts,
                               float *v_colors2, int cstride2, hipStream_t s);
void gg_launch_blend2_fwd_pair(int C, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *ids,
                               const int2 *bins, const GRec *rec, const float *colors, const float *background,
                               float *out_img, float *final_Ts, int32_t *final_idx, const float *colors2, int C2,
                               const float *background2, float *out_img2, hipStream_t s, int ncb, bool fast,
                               unsigned bytes1, unsigned bytes2);
void gg_launch_blend2_fwd_blocks(int ncb, int C, int off, int img_h, int img_w, int tiles_x, int ntiles,
                                 const int32_t *ids, const int2 *bins, const GRec *rec, const float *colors,
                                 const float *background, float *out_img, float *final_Ts, int32_t *final_idx,
                                 int write_final, hipStream_t s);

// Channel chunking: calls with <= 3 channels (rgb / depth / normal) use the narrow kernels with
// the colours inside the LDS record; anything wider is processed in chunks of 32 channels on the
// MFMA kernels (a final partial chunk is zero-padded), each chunk re-walking the tile lists.
// (A 64-channel-per-walk variant — two 32-column MFMA blocks, V_OUT operands streamed per flush —
// was built and measured: 231 VGPRs / occupancy 2 backward, 143 registers forward; the fused
// 39-channel call ran at 98 views/s against 122 with two 32-channel walks, so it was dropped.)
// 4..8 remaining channels (the rgb | depth | normal tail of a fused call) go to the 8-wide narrow
// kernels: on the wide kernels such a tail cost as much as a full 32-channel chunk.
static int chunk_width(int remaining) { return remaining <= 3 ? remaining : remaining <= 8 ? 8 : 32; }
// Defaults (tools/fwdblocks_bench.py, 5 M Gaussians, 1920x1080, 128 + 7 channels, forward kernels per view): pair walk of
// one block + one walk of the other three 1.51 ms (three waves per SIMD, 147 registers); all four blocks in the pair walk
// 1.50 (two waves, 201 registers); 2 + 2 1.74; one block per walk 2.0-2.2.  The headline's 32-channel array is one block.
#ifndef GG_FWD_BLOCKS_PAIR
#define GG_FWD_BLOCKS_PAIR 1
#endif
#ifndef GG_FWD_BLOCKS_CHUNK
#define GG_FWD_BLOCKS_CHUNK 3
#endif

// Forward walks over several 32-channel blocks at once (csrc/blend2.hip, NCB; r03): how many blocks the pair walk and
// the walks of the remaining chunks of a wide first array take.  Tuning entry (tools/): returns the previous pair value.
static int g_fwd_blocks_pair = GG_FWD_BLOCKS_PAIR, g_fwd_blocks_chunk = GG_FWD_BLOCKS_CHUNK;
extern "C" int gg_debug_set_fwd_blocks(int pair_blocks, int chunk_blocks) {
    const int prev = g_fwd_blocks_pair;
    g_fwd_blocks_pair = pair_blocks == 4 ? 4 : (pair_blocks == 2 ? 2 : 1);
    g_fwd_blocks_chunk = chunk_blocks >= 2 && chunk_blocks <= 4 ? chunk_blocks : 1;
    return prev;
}
// the forward walks of channels [off, C) of a wide array beyond its first walk: as many 32-channel blocks per walk as
// the policy allows (16-byte aligned rows), then the usual chunks
static void fwd_remaining_chunks(int C, int off, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *ids,
                                 const int32_t *tile_bins, const GRec *rec, const float *colors, const float *background,
                                 float *out_img, float *final_Ts, int32_t *final_idx, bool first_writes_final,
                                 hipStream_t s) {
    const bool aligned = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_img) & 15) == 0);
    bool first = first_writes_final;
    while (off < C) {
        const int blocks = (C - off) / 32;
        const int ncb = aligned && (off % 4 == 0) ? min(blocks, g_fwd_blocks_chunk) : 1;
        if (ncb >= 2) {
            gg_prof_begin(GG_K_BLEND_FWD + gg_width_index(32), s);
            gg_launch_blend2_fwd_blocks(ncb, C, off, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec,
                                        colors, background, out_img, final_Ts, final_idx, first, s);
            gg_prof_end(GG_K_BLEND_FWD + gg_width_index(32), s);
            off += 32 * ncb;
        } else {
            const int w = chunk_width(C - off);
            const int n = min(w, C - off);
            gg_prof_begin(GG_K_BLEND_FWD + gg_width_index(w), s);
            gg_launch_blend2_fwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                                 background, out_img, final_Ts, final_idx, first, s);
            gg_prof_end(GG_K_BLEND_FWD + gg_width_index(w), s);
            off += n;
        }
        first = false;
    }
}

#ifdef GG_ABLATION
// Measurement build only (libgg_raster_abl.so, tools/kbench.py): level > 0 makes the backward run an
// ABLATED kernel (wrong results) so its time can be attributed.  Not declared in gg_raster.h, not
// compiled into libgg_raster.so.
static int g_ablate = 0;
extern "C" int gg_debug_set_ablation(int level) {
    int prev = g_ablate;
    g_ablate = level;
    return prev;
}
#endif

extern "C" int gg_blend_fwd(int C, int N, int img_h, int img_w, const int32_t *ids,
                            const int32_t *tile_bins, const float *xys, const float *conics,
                            const float *colors, const float *opacity, const float *background,
                            float *out_img, float *final_Ts, int32_t *final_idx, void *ws,
                            size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(C >= 1, "channels < 1");
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    GG_REQUIRE(tile_bins && background && out_img && final_Ts && final_idx, "null pointer");
    GG_REQUIRE(N == 0 || (ids && xys && conics && colors && opacity), "null pointer");
    if (ws == nullptr || ws_bytes < gg_blend_workspace(N)) {
        gg_set_error("gg_blend_fwd: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    GRec *rec = (GRec *)ws;
    if (N > 0) {
        gg_prof_begin(GG_K_BLEND_PREP, s);
        hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics,
                           opacity, rec);
        gg_prof_end(GG_K_BLEND_PREP, s);
    }
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    fwd_remaining_chunks(C, 0, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts,
                         final_idx, true, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

static int blend_fwd_pair_impl(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                 const int32_t *tile_bins, const float *xys, const float *conics,
                                 const float *colors, const float *colors2, const float *opacity,
                                 const float *background, const float *background2, float *out_img,
                                 float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                                 size_t ws_bytes, gg_stream_t stream, bool fast, bool records_ready = false) {
    GG_REQUIRE(C >= 32, "the first colour array needs >= 32 channels (its first chunk carries the second array)");
    GG_REQUIRE(C2 >= 1 && C2 <= 8, "the second colour array has 1..8 channels");
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    GG_REQUIRE(tile_bins && background && background2 && out_img && out_img2 && final_Ts && final_idx,
               "null pointer");
    GG_REQUIRE(N == 0 || (ids && colors && colors2 && (records_ready || (xys && conics && opacity))), "null pointer");
    if (ws == nullptr || ws_bytes < gg_blend_workspace(N)) {
        gg_set_error("gg_blend_fwd_pair: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    GRec *rec = (GRec *)ws;
    if (N > 0 && !records_ready) {
        gg_prof_begin(GG_K_BLEND_PREP, s);
        hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics,
                           opacity, rec);
        gg_prof_end(GG_K_BLEND_PREP, s);
    }
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    // the batched kernel (fp32-grade images, not the exact summation order) needs 16-byte aligned image rows; without
    // them the exact-order kernel runs (its scalar-store epilogue takes any layout)
    // ... and reads the colour rows through buffer descriptors with 32-bit offsets (id x row bytes as a 24-bit multiply)
    const uint64_t bytes1 = (uint64_t)(N > 0 ? N : 1) * (uint64_t)C * 4u, bytes2 = (uint64_t)(N > 0 ? N : 1) * (uint64_t)C2 * 4u;
    fast = fast && (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_img) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(background) & 15) == 0) && N < (1 << 24) && C < (1 << 20) &&
           bytes1 < ((uint64_t)1 << 32) && bytes2 < ((uint64_t)1 << 32);
    // the pair walk takes 1, 2 or 4 blocks of the first array (aligned rows)
    int pair_blocks = 1;
    if (!fast && (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_img) & 15) == 0)) {
        if (g_fwd_blocks_pair >= 4 && C >= 128) pair_blocks = 4;
        else if (g_fwd_blocks_pair >= 2 && C >= 64) pair_blocks = 2;
    }
    gg_prof_begin(GG_K_BLEND_FWD_PAIR, s);
    gg_launch_blend2_fwd_pair(C, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                              background, out_img, final_Ts, final_idx, colors2, C2, background2, out_img2, s,
                              pair_blocks, fast, (unsigned)bytes1, (unsigned)bytes2);
    gg_prof_end(GG_K_BLEND_FWD_PAIR, s);
    fwd_remaining_chunks(C, 32 * pair_blocks, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background,
                         out_img, final_Ts, final_idx, false, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_blend_fwd_pair(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                 const int32_t *tile_bins, const float *xys, const float *conics,
                                 const float *colors, const float *colors2, const float *opacity,
                                 const float *background, const float *background2, float *out_img,
                                 float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                                 size_t ws_bytes, gg_stream_t stream) {
    return blend_fwd_pair_impl(C, C2, N, img_h, img_w, ids, tile_bins, xys, conics, colors, colors2, opacity, background,
                               background2, out_img, out_img2, final_Ts, final_idx, ws, ws_bytes, stream, false);
}

extern "C" int gg_blend_fwd_pair_fast(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                      const int32_t *tile_bins, const float *xys, const float *conics,
                                      const float *colors, const float *colors2, const float *opacity,
                                      const float *background, const float *background2, float *out_img,
                                      float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                                      size_t ws_bytes, gg_stream_t stream) {
    return blend_fwd_pair_impl(C, C2, N, img_h, img_w, ids, tile_bins, xys, conics, colors, colors2, opacity, background,
                               background2, out_img, out_img2, final_Ts, final_idx, ws, ws_bytes, stream, true);
}

// gg_blend_fwd_pair / gg_blend_fwd_pair_fast on a workspace that ALREADY holds the packed records of these Gaussians
// (gg_view_fwd's `records` output, or an earlier forward over the same xys / conics / opacity): no packing pass
extern "C" int gg_blend_fwd_pair_packed(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                        const int32_t *tile_bins, const float *colors, const float *colors2,
                                        const float *background, const float *background2, float *out_img,
                                        float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                                        size_t ws_bytes, int fast, gg_stream_t stream) {
    return blend_fwd_pair_impl(C, C2, N, img_h, img_w, ids, tile_bins, nullptr, nullptr, colors, colors2, nullptr,
                               background, background2, out_img, out_img2, final_Ts, final_idx, ws, ws_bytes, stream,
                               fast != 0, true);
}

// ---------------------------------------------------------------------------------------------
// deterministic backward: slab of per-(list entry, quadrant) totals + ordered per-Gaussian sums
// ---------------------------------------------------------------------------------------------
struct DetWs {
    float *slab;              // I * 4 * ks
    uint32_t *keys, *vals;    // I each: Gaussian id / list entry, sorted by id (stable: entries ascend)
    int32_t *seg;             // 2 N: [first, last) of each Gaussian's run in keys
    void *sort_ws;
    size_t sort_bytes, bytes;
};
static int det_chunks(int C) {
    int nc = 0;
    for (int off = 0; off < C; ++nc) off += min(chunk_width(C - off), C - off);
    return nc;
}
static DetWs det_ws_layout(void *ws, int N, int C, int64_t I) {
    DetWs w;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        char *p = ws ? (char *)ws + off : nullptr;
        off += gg_align_up(nbytes, 256);
        return (void *)p;
    };
    const size_t i = (size_t)(I > 0 ? I : 1), n = (size_t)(N > 0 ? N : 1);
    const size_t ks = (size_t)C + 6 * (size_t)det_chunks(C);
    w.slab = (float *)take(sizeof(float) * 4 * ks * i);
    w.keys = (uint32_t *)take(4 * i);
    w.vals = (uint32_t *)take(4 * i);
    w.seg = (int32_t *)take(8 * n);
    w.sort_bytes = gg_sort_pairs_workspace((int64_t)i);
    w.sort_ws = take(w.sort_bytes);
    w.bytes = off;
    return w;
}
extern "C" size_t gg_blend_bwd_deterministic_workspace(int num_points, int channels, int64_t num_intersects) {
    if (channels < 1) return 0;
    return det_ws_layout(nullptr, num_points, channels, num_intersects).bytes;
}

__global__ void det_pairs_kernel(int64_t I, const int32_t *__restrict__ ids, uint32_t *__restrict__ keys,
                                 uint32_t *__restrict__ vals) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= I) return;
    keys[e] = (uint32_t)ids[e];
    vals[e] = (uint32_t)e;
}
__global__ void det_edges_kernel(int64_t I, int N, const uint32_t *__restrict__ keys, int32_t *__restrict__ seg) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= I) return;
    const uint32_t k = keys[e];
    if (k >= (uint32_t)N) return;
    if (e == 0 || keys[e - 1] != k) seg[2 * (size_t)k] = (int32_t)e;
    if (e == I - 1 || keys[e + 1] != k) seg[2 * (size_t)k + 1] = (int32_t)e + 1;
}
// One wave per Gaussian; lane l owns output column l (and l + 64, ...): colour channel c < C or geometry
// component m = column - C.  Every sum runs over the Gaussian's list entries in ascending order, quadrants
// 0..3 inside an entry and (geometry) channel chunks inside a quadrant: one fixed order, no atomics.
__global__ __launch_bounds__(256) void det_reduce_kernel(int N, int C, int nchunks, int ks,
                                                         const float *__restrict__ slab,
                                                         const uint32_t *__restrict__ vals,
                                                         const int32_t *__restrict__ seg, float *v_xy, float *v_conic,
                                                         float *v_colors, float *v_opacity, int gstride, int cstride) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= N) return;
    const int first = seg[2 * (size_t)g], last = seg[2 * (size_t)g + 1];
    if (last <= first) return;
    for (int col = lane; col < C + 6; col += 64) {
        float sum = 0.0f;
        const bool geom = col >= C;
        const int c0 = geom ? C + (col - C) : col;
        for (int r = first; r < last; ++r) {
            const float *row = slab + (size_t)vals[r] * 4 * ks;
            for (int w = 0; w < 4; ++w) {
                if (geom)
                    for (int c = 0; c < nchunks; ++c) sum += row[(size_t)w * ks + c0 + 6 * c];
                else
                    sum += row[(size_t)w * ks + c0];
            }
        }
        float *dst;
        if (!geom) dst = v_colors + (size_t)g * (cstride ? cstride : C) + col;
        else {
            const int m = col - C;
            if (m < 2) dst = v_xy + (size_t)g * (gstride ? gstride : 2) + m;
            else if (m < 5) dst = v_conic + (size_t)g * (gstride ? gstride : 3) + (m - 2);
            else dst = v_opacity + (size_t)g * (gstride ? gstride : 1);
        }
        *dst += sum;   // the arrays start at zero (or hold what earlier calls added: the accumulate flags)
    }
}

static int blend_bwd_impl(int C, int N, int img_h, int img_w, const int32_t *ids,
                          const int32_t *tile_bins, const float *xys, const float *conics,
                          const float *colors, const float *opacity, const float *background,
                          const float *final_Ts, const int32_t *final_idx, const float *v_out,
                          float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                          int geom_stride, int color_stride, void *ws, size_t ws_bytes,
                          int flags, gg_stream_t stream, bool deterministic, int64_t I, void *det_ws,
                          size_t det_ws_bytes) {
    const bool ws_from_forward = (flags & GG_BWD_WS_FROM_FORWARD) != 0;
    const bool acc_colors = (flags & GG_BWD_ACCUMULATE_COLORS) != 0;
    const bool acc_geom = (flags & GG_BWD_ACCUMULATE_GEOM) != 0;
    GG_REQUIRE(C >= 1, "channels < 1");
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    if (N == 0) return GG_OK;
    GG_REQUIRE(ids && tile_bins && xys && conics && colors && opacity && background && final_Ts &&
                   final_idx && v_out && v_xy && v_conic && v_colors && v_opacity,
               "null pointer");
    if (ws == nullptr || ws_bytes < gg_blend_workspace(N)) {
        gg_set_error("gg_blend_bwd: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    GRec *rec = (GRec *)ws;
    if (!ws_from_forward) {
        gg_prof_begin(GG_K_BLEND_PREP, s);
        hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics,
                           opacity, rec);
        gg_prof_end(GG_K_BLEND_PREP, s);
    }
    // the kernels accumulate with atomics: the four gradient arrays start at zero (one memset when
    // the caller laid them out back to back: v_xy | v_conic | v_opacity | v_colors)
    GG_REQUIRE(geom_stride == 0 || geom_stride >= 6, "geom_stride must be 0 (dense) or >= 6");
    GG_REQUIRE(color_stride == 0 || color_stride >= C, "color_stride must be 0 (dense) or >= channels");
    GG_REQUIRE(!acc_colors || !(color_stride == geom_stride && geom_stride > 0 && v_colors == v_xy + 6),
               "GG_BWD_ACCUMULATE_COLORS needs v_colors outside the interleaved geometry record");
    bool fail;
    const size_t n = (size_t)N;
    if (acc_geom) {
        // a later segment of a multi-segment call: the geometry gradients (and colours that live in
        // the same record) keep what the earlier segments added; separate colour arrays are cleared
        GG_REQUIRE(geom_stride == 0 || (v_conic == v_xy + 2 && v_opacity == v_xy + 5),
                   "interleaved geometry gradients: v_conic = v_xy + 2 and v_opacity = v_xy + 5 expected");
        fail = false;
        const bool in_record = geom_stride > 0 && color_stride == geom_stride && v_colors == v_xy + 6;
        if (!in_record && !acc_colors)
            fail = gg_fill_async(v_colors, 0, sizeof(float) * (color_stride ? color_stride : C) * n, s) !=
                   hipSuccess;
    } else if (geom_stride > 0) {
        // interleaved records {xy, conic, opacity[, colours]} of geom_stride floats per Gaussian
        GG_REQUIRE(v_conic == v_xy + 2 && v_opacity == v_xy + 5, "interleaved geometry gradients: "
                   "v_conic = v_xy + 2 and v_opacity = v_xy + 5 expected");
        fail = gg_fill_async(v_xy, 0, sizeof(float) * geom_stride * n, s) != hipSuccess;
        if (!(color_stride == geom_stride && v_colors == v_xy + 6) && !acc_colors)   // colours live elsewhere
            fail |= gg_fill_async(v_colors, 0, sizeof(float) * (color_stride ? color_stride : C) * n, s) !=
                    hipSuccess;
    } else if (!acc_colors && color_stride == 0 && v_conic == v_xy + 2 * n && v_opacity == v_conic + 3 * n &&
               v_colors == v_opacity + n) {
        fail = gg_fill_async(v_xy, 0, sizeof(float) * (6 + (size_t)C) * n, s) != hipSuccess;
    } else {
        fail = gg_fill_async(v_xy, 0, sizeof(float) * 2 * n, s) != hipSuccess;
        fail |= gg_fill_async(v_conic, 0, sizeof(float) * 3 * n, s) != hipSuccess;
        if (!acc_colors)
            fail |= gg_fill_async(v_colors, 0, sizeof(float) * (color_stride ? color_stride : C) * n, s) !=
                    hipSuccess;
        fail |= gg_fill_async(v_opacity, 0, sizeof(float) * n, s) != hipSuccess;
    }
    if (fail) {
        gg_set_error("gg_blend_bwd: memset failed");
        return GG_ERR_LAUNCH;
    }
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    DetSlab det = DetSlab();
    DetWs dw;
    const int nchunks = det_chunks(C);
    if (deterministic) {
        if (I == 0) return GG_OK;   // empty lists: the zeroed (or untouched) arrays are the answer
        dw = det_ws_layout(det_ws, N, C, I);
        det.p = dw.slab;
        det.ks = C + 6 * nchunks;
        if (gg_fill_async(dw.slab, 0, sizeof(float) * 4 * (size_t)det.ks * (size_t)I, s) != hipSuccess ||
            gg_fill_async(dw.seg, 0, 8 * (size_t)N, s) != hipSuccess) {
            gg_set_error("gg_blend_bwd_deterministic: memset failed");
            return GG_ERR_LAUNCH;
        }
    }
    int chunk = 0;
    for (int off = 0; off < C; ++chunk) {
        const int w = chunk_width(C - off);
        const int n = min(w, C - off);
        det.coff = off;
        det.goff = C + 6 * chunk;
        gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(w), s);
#ifdef GG_ABLATION
        if ((w == 3 && g_ablate > 0 && g_ablate < 10) || (w == 32 && n == 32 && g_ablate > 10))
            gg_launch_blend2_bwd_ablate(g_ablate, C, off, img_h, img_w, tiles_x, ntiles, ids,
                                        (const int2 *)tile_bins, rec, colors, background, final_Ts,
                                        final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride,
                                        color_stride, s);
        else
#endif
            gg_launch_blend2_bwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, ids,
                                 (const int2 *)tile_bins, rec, colors, background, final_Ts,
                                 final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride,
                                 color_stride, s, det);
        gg_prof_end(GG_K_BLEND_BWD + gg_width_index(w), s);
        off += n;
    }
    if (deterministic) {
        const unsigned nb = (unsigned)((I + 255) / 256);
        hipLaunchKernelGGL(det_pairs_kernel, dim3(nb), dim3(256), 0, s, I, ids, dw.keys, dw.vals);
        int bits = 1;
        while (((int64_t)1 << bits) < (int64_t)N) ++bits;
        const int rc = gg_sort_pairs(I, dw.keys, dw.vals, bits, dw.sort_ws, dw.sort_bytes, s);
        if (rc != GG_OK) {
            gg_set_error("gg_blend_bwd_deterministic: sort failed");
            return rc;
        }
        hipLaunchKernelGGL(det_edges_kernel, dim3(nb), dim3(256), 0, s, I, N, dw.keys, dw.seg);
        hipLaunchKernelGGL(det_reduce_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, C, nchunks, det.ks, dw.slab,
                           dw.vals, dw.seg, v_xy, v_conic, v_colors, v_opacity, geom_stride, color_stride);
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_blend_bwd(int C, int N, int img_h, int img_w, const int32_t *ids,
                            const int32_t *tile_bins, const float *xys, const float *conics,
                            const float *colors, const float *opacity, const float *background,
                            const float *final_Ts, const int32_t *final_idx, const float *v_out,
                            float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                            int geom_stride, int color_stride, void *ws, size_t ws_bytes,
                            int flags, gg_stream_t stream) {
    return blend_bwd_impl(C, N, img_h, img_w, ids, tile_bins, xys, conics, colors, opacity, background, final_Ts,
                          final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride, color_stride, ws,
                          ws_bytes, flags, stream, false, 0, nullptr, 0);
}

static int blend_bwd_pair_impl(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                 const int32_t *tile_bins, const float *xys, const float *conics,
                                 const float *colors, const float *colors2, const float *opacity,
                                 const float *background, const float *background2, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, const float *const *v_out2_parts,
                                 const int *v_out2_channels, int num_parts, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_colors2, float *v_opacity,
                                 int geom_stride, int color_stride, int color_stride2, void *ws, size_t ws_bytes,
                                 int flags, gg_stream_t stream) {
    const bool ws_from_forward = (flags & GG_BWD_WS_FROM_FORWARD) != 0;
    const bool acc_colors = (flags & GG_BWD_ACCUMULATE_COLORS) != 0;
    GG_REQUIRE((flags & GG_BWD_ACCUMULATE_GEOM) == 0, "gg_blend_bwd_pair writes the geometry gradients itself");
    GG_REQUIRE(C >= 32, "the first colour array needs >= 32 channels (its first chunk carries the second array)");
    GG_REQUIRE(C2 >= 1 && C2 <= 8, "the second colour array has 1..8 channels");
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_h > 0 && img_w > 0, "empty image");
    if (N == 0) return GG_OK;
    GG_REQUIRE(ids && tile_bins && xys && conics && colors && colors2 && opacity && background && background2 &&
                   final_Ts && final_idx && v_out && v_out2_parts && v_out2_channels && v_xy && v_conic && v_colors &&
                   v_colors2 && v_opacity,
               "null pointer");
    GG_REQUIRE(num_parts >= 1 && num_parts <= 3, "the second cotangent comes as 1..3 images");
    {
        int total = 0;
        for (int k = 0; k < num_parts; ++k) {
            GG_REQUIRE(v_out2_parts[k] != nullptr && v_out2_channels[k] >= 1, "empty part of the second cotangent");
            total += v_out2_channels[k];
        }
        GG_REQUIRE(total == C2, "the parts of the second cotangent must add up to channels2");
    }
    GG_REQUIRE(geom_stride == 0 || geom_stride >= 6, "geom_stride must be 0 (dense) or >= 6");
    GG_REQUIRE(color_stride == 0 || color_stride >= C, "color_stride must be 0 (dense) or >= channels");
    GG_REQUIRE(color_stride2 == 0 || color_stride2 >= C2, "color_stride2 must be 0 (dense) or >= channels2");
    GG_REQUIRE(geom_stride == 0 || (v_conic == v_xy + 2 && v_opacity == v_xy + 5),
               "interleaved geometry gradients: v_conic = v_xy + 2 and v_opacity = v_xy + 5 expected");
    if (ws == nullptr || ws_bytes < gg_blend_workspace(N)) {
        gg_set_error("gg_blend_bwd_pair: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    GRec *rec = (GRec *)ws;
    if (!ws_from_forward) {
        gg_prof_begin(GG_K_BLEND_PREP, s);
        hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics, opacity, rec);
        gg_prof_end(GG_K_BLEND_PREP, s);
    }
    const size_t n = (size_t)N;
    const bool in_record = geom_stride > 0 && color_stride2 == geom_stride && v_colors2 == v_xy + 6;
    GG_REQUIRE(!in_record || geom_stride >= 6 + C2, "the record is too short for the second array's gradients");
    bool fail = false;
    if (geom_stride > 0) {
        fail |= gg_fill_async(v_xy, 0, sizeof(float) * geom_stride * n, s) != hipSuccess;
    } else {
        fail |= gg_fill_async(v_xy, 0, sizeof(float) * 2 * n, s) != hipSuccess;
        fail |= gg_fill_async(v_conic, 0, sizeof(float) * 3 * n, s) != hipSuccess;
        fail |= gg_fill_async(v_opacity, 0, sizeof(float) * n, s) != hipSuccess;
    }
    if (!in_record)
        fail |= gg_fill_async(v_colors2, 0, sizeof(float) * (color_stride2 ? color_stride2 : C2) * n, s) != hipSuccess;
    if (!acc_colors)
        fail |= gg_fill_async(v_colors, 0, sizeof(float) * (color_stride ? color_stride : C) * n, s) != hipSuccess;
    if (fail) {
        gg_set_error("gg_blend_bwd_pair: memset failed");
        return GG_ERR_LAUNCH;
    }
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    gg_prof_begin(GG_K_BLEND_BWD_PAIR, s);
    gg_launch_blend2_bwd_pair(C, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors, background,
                              final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride,
                              color_stride, colors2, C2, background2, v_out2_parts, v_out2_channels, num_parts,
                              v_colors2, color_stride2, s);
    gg_prof_end(GG_K_BLEND_BWD_PAIR, s);
    for (int off = 32; off < C;) {   // further chunks of the first array: their own walks, adding to the same arrays
        const int w = chunk_width(C - off);
        const int nn = min(w, C - off);
        gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(w), s);
        gg_launch_blend2_bwd(w, C, off, nn, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                             background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride,
                             color_stride, s, DetSlab());
        gg_prof_end(GG_K_BLEND_BWD + gg_width_index(w), s);
        off += nn;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_blend_bwd_pair(int C, int C2, int N, int img_h, int img_w, const int32_t *ids,
                                 const int32_t *tile_bins, const float *xys, const float *conics,
                                 const float *colors, const float *colors2, const float *opacity,
                                 const float *background, const float *background2, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, const float *const *v_out2_parts,
                                 const int *v_out2_channels, int num_parts, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_colors2, float *v_opacity,
                                 int geom_stride, int color_stride, int color_stride2, void *ws, size_t ws_bytes,
                                 int flags, gg_stream_t stream) {
    return blend_bwd_pair_impl(C, C2, N, img_h, img_w, ids, tile_bins, xys, conics, colors, colors2, opacity, background,
                               background2, final_Ts, final_idx, v_out, v_out2_parts, v_out2_channels, num_parts, v_xy,
                               v_conic, v_colors, v_colors2, v_opacity, geom_stride, color_stride, color_stride2, ws,
                               ws_bytes, flags, stream);
}

extern "C" int gg_blend_bwd_deterministic(int C, int N, int img_h, int img_w, const int32_t *ids,
                                          const int32_t *tile_bins, const float *xys, const float *conics,
                                          const float *colors, const float *opacity, const float *background,
                                          const float *final_Ts, const int32_t *final_idx, const float *v_out,
                                          float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                                          int geom_stride, int color_stride, void *ws, size_t ws_bytes,
                                          int flags, int64_t num_intersects, void *det_ws, size_t det_ws_bytes,
                                          gg_stream_t stream) {
    GG_REQUIRE(C >= 1, "channels < 1");
    GG_REQUIRE(num_intersects >= 0 && num_intersects < ((int64_t)1 << 31), "num_intersects out of range");
    if (num_intersects > 0 &&
        (det_ws == nullptr || det_ws_bytes < gg_blend_bwd_deterministic_workspace(N, C, num_intersects))) {
        gg_set_error("gg_blend_bwd_deterministic: deterministic workspace too small");
        return GG_ERR_WORKSPACE;
    }
    return blend_bwd_impl(C, N, img_h, img_w, ids, tile_bins, xys, conics, colors, opacity, background, final_Ts,
                          final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, geom_stride, color_stride, ws,
                          ws_bytes, flags, stream, true, num_intersects, det_ws, det_ws_bytes);
}
