/*
 * gg_raster.h — C ABI of libgg_raster.so, the MI355X (gfx950) Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for GaussianGrasper's feature-field hot path.  The reference
 * reaches the same functionality through gsplat==0.1.0's private extension module
 * (`gsplat.cuda._C`, bound with pybind11/torch types); the four autograd.Functions the
 * reference model calls — nerfstudio/models/gaussian_splatting.py:699 (ProjectGaussians),
 * :730 (SphericalHarmonics), :735/:759/:773 (RasterizeGaussians), :747 (NDRasterizeGaussians)
 * — sit directly on top of it.  Each entry point below names the binding it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / PyTorch caching allocator); the library
 *     never allocates, frees or keeps caller memory; scratch comes in through (ws, ws_bytes)
 *     whose size the matching *_workspace() query returns;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously, nothing synchronises the host;
 *   - layouts are torch's: row-major array-of-structs, fp32, ids int32;
 *   - return 0 on success; negative on error (GG_ERR_*), message via gg_last_error().
 *
 * Arithmetic contract: SURVEY.md §8a rows a3-a12 with the constants of gg_constants.h;
 * the forward results (radii, tile counts, depth order, tile lists, images, final_T, final_idx)
 * are bit-identical to the CPU oracle (oracle/gg_oracle.c), gradients agree to fp32 summation
 * order.
 */
#ifndef GG_RASTER_H
#define GG_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_OK 0
#define GG_ERR_INVALID_ARG (-1)
#define GG_ERR_LAUNCH (-2)
#define GG_ERR_WORKSPACE (-3)
#define GG_ERR_UNSUPPORTED (-4)

typedef void *gg_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
int gg_abi_version(void);
/* Message for the last error returned on the calling thread ("" if none). */
const char *gg_last_error(void);

/* ---- projection ------------------------------------------------------------------------
 * Replaces gsplat `_C.project_gaussians_forward` (ProjectGaussians.forward; reference call
 * gaussian_splatting.py:699-713).  viewmat: >=12 floats row-major (the caller's viewmat[:3,:]);
 * projmat: 16 floats (projmat @ viewmat).  All six outputs are fully written (zeros for culled
 * Gaussians as the oracle documents), so they may be uninitialised on entry. */
int gg_project_fwd(int num_points, const float *means3d, const float *scales, float glob_scale,
                   const float *quats, const float *viewmat, const float *projmat, float fx,
                   float fy, float cx, float cy, int img_height, int img_width, int tiles_x,
                   int tiles_y, float clip_thresh, float *cov3d, float *xys, float *depths,
                   int32_t *radii, float *conics, int32_t *num_tiles_hit, gg_stream_t stream);

/* gg_project_fwd that also leaves sum(num_tiles_hit) in *num_intersects_out (device, int64) — what gg_count_intersects
 * computes with a launch of its own (reference: `torch.cumsum(num_tiles_hit)[-1].item()` inside every rasterize call).
 * count_ws: gg_project_count_workspace(num_points) bytes of scratch, 4-byte aligned (per-workgroup partial sums, added
 * up by a one-workgroup launch behind the projection). */
size_t gg_project_count_workspace(int num_points);
int gg_project_fwd_count(int num_points, const float *means3d, const float *scales, float glob_scale,
                         const float *quats, const float *viewmat, const float *projmat, float fx, float fy,
                         float cx, float cy, int img_height, int img_width, int tiles_x, int tiles_y,
                         float clip_thresh, float *cov3d, float *xys, float *depths, int32_t *radii, float *conics,
                         int32_t *num_tiles_hit, int64_t *num_intersects_out, void *count_ws, size_t count_ws_bytes,
                         gg_stream_t stream);

/* Replaces gsplat `_C.project_gaussians_backward` (ProjectGaussians.backward).  v_conic uses
 * gsplat's symmetric-matrix convention (v_conic[:,1] is half of dL/d conic.y).  Outputs fully
 * written (zeros where radii<=0). */
int gg_project_bwd(int num_points, const float *means3d, const float *scales, float glob_scale,
                   const float *quats, const float *viewmat, const float *projmat, float fx,
                   float fy, float cx, float cy, int img_height, int img_width,
                   const int32_t *radii, const float *conics, const float *v_xy,
                   const float *v_depth, const float *v_conic, float *v_mean3d, float *v_scale,
                   float *v_quat, gg_stream_t stream);
/* gg_project_bwd_ex: the same with v_xy / v_conic rows v_xy_stride / v_conic_stride floats apart (the blend
 * backward's interleaved gradient record {xy, conic, opacity, ...} is read in place: no compaction copies) and,
 * with accumulate_means != 0, v_mean3d added to instead of written (the caller's gradient buffer of the means,
 * which enter the projection as a leaf). */
int gg_project_bwd_ex(int num_points, const float *means3d, const float *scales, float glob_scale,
                      const float *quats, const float *viewmat, const float *projmat, float fx, float fy, float cx,
                      float cy, int img_height, int img_width, const int32_t *radii, const float *conics,
                      const float *v_xy, int v_xy_stride, const float *v_depth, const float *v_conic,
                      int v_conic_stride, float *v_mean3d, int accumulate_means, float *v_scale, float *v_quat,
                      gg_stream_t stream);

/* ---- spherical harmonics -----------------------------------------------------------------
 * Replace gsplat `_C.compute_sh_forward` / `_C.compute_sh_backward` (SphericalHarmonics;
 * reference call gaussian_splatting.py:730).  coeffs (N, num_bases, 3); num_bases in
 * {1,4,9,16,25}; degrees_to_use <= degree(num_bases). */
int gg_sh_fwd(int num_points, int num_bases, int degrees_to_use, const float *viewdirs,
              const float *coeffs, float *colors, gg_stream_t stream);
int gg_sh_bwd(int num_points, int num_bases, int degrees_to_use, const float *viewdirs,
              const float *v_colors, float *v_coeffs, gg_stream_t stream);
/* v_coeffs += (instead of =): the gradient goes straight into the caller's accumulation buffer (the
 * `.grad` of the SH parameter across the views of an optimizer step) — one read-modify-write of
 * 300 B per Gaussian instead of a 300 B store followed by autograd's separate add. */
int gg_sh_bwd_accumulate(int num_points, int num_bases, int degrees_to_use, const float *viewdirs,
                         const float *v_colors, float *v_coeffs, gg_stream_t stream);

/* gg_shade_tail_fwd / _bwd: the plugin route's 7-channel colour array in one pass each way —
 * tail (N, 7) = [ clamp(SH(viewdirs, coeffs) + 0.5, 0, 1) | depth | normal ], i.e. the reference's
 * `rgbs = torch.clamp(SphericalHarmonics.apply(n, viewdirs, colors) + 0.5, 0.0, 1.0)` (gaussian_splatting.py:
 * 730-731) packed with the depth (:765) and normal (:779) colour arrays it rasterizes next.  clamp_mask (N bytes):
 * bit c set = the gradient of colour c passes the clamp (0 <= x <= 1, torch.clamp's rule).
 * Backward: v_tail rows are v_tail_stride >= 7 floats apart (the interleaved gradient record of the blend
 * backward is read in place); v_coeffs (N, num_bases, 3) is written, or added to when accumulate != 0;
 * v_depths (N,), v_normals (N, 3) are written. */
int gg_shade_tail_fwd(int num_points, int num_bases, int degrees_to_use, const float *viewdirs,
                      const float *coeffs, const float *depths, const float *normals, float *tail,
                      uint8_t *clamp_mask, gg_stream_t stream);
int gg_shade_tail_bwd(int num_points, int num_bases, int degrees_to_use, const float *viewdirs,
                      const float *v_tail, int v_tail_stride, const uint8_t *clamp_mask, float *v_coeffs,
                      int accumulate, float *v_depths, float *v_normals, gg_stream_t stream);

/* Deferred SH gradient over the views of an optimizer step (the views of a step accumulate into one
 * gradient).  gg_shade_tail_bwd_split is the part of gg_shade_tail_bwd that cannot wait — v_depths, v_normals and
 * the clamp-masked colour cotangent v_rgb (N, 3), which is KEPT instead of being expanded; gg_sh_bwd_multi expands
 * the kept cotangents of num_views views in one pass: v_coeffs (N, num_bases, 3) = [v_coeffs +] sum_v Y(viewdirs_v)
 * (x) v_rgb_v, the views summed in order starting from the buffer's value (the bits of adding view after view).  viewdirs /
 * v_colors: host arrays of num_views device pointers to (N, 3) arrays.  One 300-byte write per Gaussian and step
 * instead of a 600-byte read-modify-write per Gaussian and view. */
int gg_shade_tail_bwd_split(int num_points, const float *v_tail, int v_tail_stride, const uint8_t *clamp_mask,
                            float *v_rgb, float *v_depths, float *v_normals, gg_stream_t stream);
int gg_sh_bwd_multi(int num_points, int num_bases, int degrees_to_use, int num_views, const float *const *viewdirs,
                    const float *const *v_colors, float *v_coeffs, int accumulate, gg_stream_t stream);

/* ---- one backward pass over the Gaussians of a view (round 4; ops.ViewGeometry of the plugin route) ------------
 * gg_shade_tail_bwd_split + gg_project_bwd_ex + gg_activate_bwd_ex in ONE kernel, for the case where every parameter has a
 * gradient buffer to add into: reads the blend backward's record of a Gaussian once — rec_stride floats apart,
 * [v_xy 0..1 | v_conic 2..4 | v_opacity 5 | v_rgb 6..8 | v_depth 9 | v_normal 10..12] — keeps the clamp-masked colour
 * cotangent v_rgb (N, 3) for gg_sh_bwd_multi and ADDS the gradients of the means (N, 3), log scales (N, 3), raw
 * quaternions (N, 4) and opacity logits (N) to v_means / v_log_scales / v_quats / v_opacities.  `scales`, `quats_n`,
 * `opac`, `axis` are gg_activate_fwd's outputs, `quats_raw` its input, `radii` / `conics` gg_project_fwd's.  The
 * per-Gaussian operation sequence is the three kernels' (shared device functions): the same bits.
 * Reference: what autograd does behind gaussian_splatting.py:699-731 in ~60 launches per view. */
int gg_view_bwd(int num_points, const float *rec, int rec_stride, const uint8_t *clamp_mask, const float *means,
                const float *scales, float glob_scale, const float *quats_raw, const float *quats_n, const float *opac,
                const int32_t *axis, const float *viewmat, const float *projmat, float fx, float fy, int img_height,
                int img_width, const int32_t *radii, const float *conics, float *v_rgb, float *v_means,
                float *v_log_scales, float *v_quats, float *v_opacities, gg_stream_t stream);

/* gg_activate_fwd + gg_project_fwd (glob_scale 1) + the intersection count in one pass over the Gaussians (round 4,
 * ops.ViewGeometry).  Outputs are those two entries' (bit for bit: shared device code) except that cov3d is not produced.
 * *num_intersects_out (device, int64) receives sum(num_tiles_hit).  `parts`: gg_view_fwd_workspace(num_points) bytes,
 * 4-byte aligned; on return it holds per workgroup of 256 Gaussians [sum of num_tiles_hit | smallest | largest depth bits
 * of its visible Gaussians] (three arrays of ceil(num_points / 256) words) — gg_bin_sort_dev_ex takes the last two
 * instead of running its own pass over depths and radii.  `records` (nullable): gg_blend_workspace(num_points) bytes,
 * 16-byte aligned — filled with the blend kernels' packed per-Gaussian records (xy, opacity, cull threshold, conic), i.e.
 * a blend workspace that gg_blend_fwd_pair_packed takes as it is. */
size_t gg_view_fwd_workspace(int num_points);
int gg_view_fwd(int num_points, const float *means, const float *log_scales, const float *quats, const float *opacities,
                const float *cam_pos, const float *viewmat, const float *projmat, float fx, float fy, float cx, float cy,
                int img_height, int img_width, int tiles_x, int tiles_y, float clip_thresh, float *scales, float *quats_n,
                float *opac, float *viewdirs, float *normals, int32_t *axis, float *xys, float *depths, int32_t *radii,
                float *conics, int32_t *num_tiles_hit, int64_t *num_intersects_out, void *parts, size_t parts_bytes,
                void *records, size_t records_bytes, gg_stream_t stream);

/* ---- quat_to_rotmat ------------------------------------------------------------------------
 * Replace gsplat `_torch_impl.quat_to_rotmat` (differentiable torch code there: ~35 elementwise
 * launches forward, ~70 backward; reference call sites gaussian_splatting.py:516,614 — the
 * normals rendered at :772-784 — and scripts/update.py:204,229).  quats (N,4) wxyz, normalised
 * as q / max(|q|, 1e-12); rot (N,3,3) row-major.  bwd: v_quats = d<rot, v_rot>/d quats, through
 * the normalisation. */
int gg_quat_to_rotmat_fwd(int num_points, const float *quats, float *rot, gg_stream_t stream);
int gg_quat_to_rotmat_bwd(int num_points, const float *quats, const float *v_rot, float *v_quats,
                          gg_stream_t stream);

/* ---- caller-side activations of one view ----------------------------------------
 * One kernel each way for what the reference's get_outputs does with torch elementwise ops before it
 * calls the rasterizer: exp(scales) (gaussian_splatting.py:701), quats / |quats| (:703),
 * sigmoid(opacities) (:742, once instead of four times), view directions (:727-728) and the
 * smallest-axis normals (:605-619 via quat_to_rotmat).  `axis` (N) is the argmin index, kept for the
 * backward.  Gradients: v_log_scales, v_quats (both the normalisation and the normal path), v_opacities
 * (logits); means get no gradient here (viewdirs are detached in the reference). */
int gg_activate_fwd(int num_points, const float *means, const float *log_scales, const float *quats,
                    const float *opacities, const float *cam_pos, float *scales, float *quats_n,
                    float *opac, float *viewdirs, float *normals, int32_t *axis, gg_stream_t stream);
int gg_activate_bwd(int num_points, const float *quats, const float *scales, const float *opac,
                    const int32_t *axis, const float *v_scales, const float *v_quats_n,
                    const float *v_opac, const float *v_normals, float *v_log_scales, float *v_quats,
                    float *v_opacities, gg_stream_t stream);
/* gg_activate_bwd_ex: v_opac entries v_opac_stride floats apart (column 5 of the blend backward's record, read in
 * place); accumulate != 0: v_log_scales, v_quats, v_opacities are added to (registered gradient buffers). */
int gg_activate_bwd_ex(int num_points, const float *quats, const float *scales, const float *opac,
                       const int32_t *axis, const float *v_scales, const float *v_quats_n, const float *v_opac,
                       int v_opac_stride, const float *v_normals, float *v_log_scales, float *v_quats,
                       float *v_opacities, int accumulate, gg_stream_t stream);

/* ---- binning -------------------------------------------------------------------------------
 * Together replace gsplat `compute_cumulative_intersects` + `bin_and_sort_gaussians`
 * (`_C.map_gaussian_to_intersects`, torch.sort, `_C.get_tile_bin_edges`) that every
 * Rasterize*.forward runs (gaussian_splatting.py:735,747,759,773).
 *
 * gg_count_intersects: *num_intersects_out (device int64) = sum(num_tiles_hit). */
size_t gg_count_workspace(int num_points);
int gg_count_intersects(int num_points, const int32_t *num_tiles_hit,
                        int64_t *num_intersects_out, void *ws, size_t ws_bytes,
                        gg_stream_t stream);

/* gg_bin_sort: given I = sum(num_tiles_hit) (read back by the caller), writes
 *   gaussian_ids_sorted (I,)  — Gaussian ids, tile-major, near-to-far, ties by ascending id:
 *                               exactly the order of the reference's sorted int64 keys
 *                               (tile_id << 32 | depth bits);
 *   tile_bins (tiles_x*tiles_y, 2) — [start,end) of each tile in that list, (0,0) if empty;
 *   isect_tile_sorted (I,) optional (may be NULL) — the tile id of every sorted entry.
 * Tile grids up to 1023 x 1023 (images up to 16 368 pixels a side): a Gaussian's tile box travels through the depth
 * sort in one 32-bit word (round 4); larger grids are rejected with GG_ERR_INVALID_ARG.  Workspace: ~44 bytes per Gaussian
 * + 12 bytes per list entry. */
size_t gg_bin_sort_workspace(int num_points, int64_t num_intersects);
int gg_bin_sort(int num_points, int64_t num_intersects, const float *xys, const float *depths,
                const int32_t *radii, const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *isect_tile_sorted,
                void *ws, size_t ws_bytes, gg_stream_t stream);

/* gg_bin_sort_dev: the same without the host knowing the count.  `capacity` sizes the outputs, the
 * workspace (gg_bin_sort_workspace(num_points, capacity)) and the launches; the kernels read the
 * actual count from *num_intersects_dev (what gg_count_intersects wrote, same stream) and process
 * min(count, capacity) entries.  The caller reads the count back later (asynchronously) and, in the
 * rare case count > capacity (lists truncated), calls again with a larger capacity.  Removes the one
 * host<->device round trip per view the reference has at this point (`.item()`). */
/* gg_bin_sort_status: kept for ABI stability.  Up to round 3 the offsets scan waited on other workgroups inside its
 * launch and a wait that gave up was reported here; since round 4 no binning kernel waits on another workgroup (the
 * offsets come out of the depth-bucket kernels), so there is nothing to report: the call synchronises the stream and
 * returns GG_OK. */
int gg_bin_sort_status(int num_points, int64_t num_intersects, const void *ws, size_t ws_bytes, gg_stream_t stream);
int gg_bin_sort_dev(int num_points, int64_t capacity, const int64_t *num_intersects_dev,
                    const float *xys, const float *depths, const int32_t *radii,
                    const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                    int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *isect_tile_sorted,
                    void *ws, size_t ws_bytes, gg_stream_t stream);
/* gg_bin_sort_dev_ex: the same, with the partial minima / maxima of the visible Gaussians' depth bits handed over
 * (range_parts pairs, e.g. gg_view_fwd's `parts` arrays 1 and 2): the depth buckets' own pass over depths / radii is not
 * run.  range_parts 0: gg_bin_sort_dev. */
int gg_bin_sort_dev_ex(int num_points, int64_t capacity, const int64_t *num_intersects_dev, const float *xys,
                       const float *depths, const int32_t *radii, const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                       int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *isect_tile_sorted, void *ws,
                       size_t ws_bytes, const uint32_t *depth_bits_min, const uint32_t *depth_bits_max, int range_parts,
                       gg_stream_t stream);

/* ---- alpha blending ----------------------------------------------------------------------
 * gg_blend_fwd replaces gsplat `_C.rasterize_forward` (C=3) and `_C.nd_rasterize_forward`
 * (any C >= 1).  colors (N,C), opacity (N,) or (N,1), background (C,), out_img (H,W,C),
 * final_Ts (H,W), final_idx (H,W).  ws: gg_blend_workspace(num_points) bytes. */
size_t gg_blend_workspace(int num_points);
int gg_blend_fwd(int channels, int num_points, int img_height, int img_width,
                 const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                 const float *conics, const float *colors, const float *opacity,
                 const float *background, float *out_img, float *final_Ts, int32_t *final_idx,
                 void *ws, size_t ws_bytes, gg_stream_t stream);

/* gg_blend_fwd_pair: gg_blend_fwd of TWO colour arrays over the same Gaussians — colors (N, channels >= 32)
 * into out_img and colors2 (N, channels2 <= 8) into out_img2 — where the second array is blended in the
 * same walk as the first 32-channel chunk of the first (its colours ride in the LDS record, 8 more fma
 * per Gaussian on the VALU while the matrix pipe does the 32 channels).  What `ops.RasterizeSegments`
 * uses for feature | rgb+depth+normal: one forward walk per view instead of two.  Images are bit-identical
 * to two gg_blend_fwd calls. */
int gg_blend_fwd_pair(int channels, int channels2, int num_points, int img_height, int img_width,
                      const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                      const float *conics, const float *colors, const float *colors2,
                      const float *opacity, const float *background, const float *background2,
                      float *out_img, float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                      size_t ws_bytes, gg_stream_t stream);

/* gg_blend_fwd_pair_fast (round 4): the same call on the batched kernel — the survivors of a quadrant's cull queued to
 * batches of 32, alpha T written to a slab, ONE product OUT[64 pixels x 48 channels] += VIS[64 x 32] COL[32 x 48] per
 * batch on v_mfma_f32_16x16x32_f16 with fp16 two-piece operands scaled by powers of two (csrc/blend2.hip
 * blend2_fwd_batch_kernel).  final_Ts, final_idx and every alpha / stop decision are the exact kernel's, bit for bit (the
 * walk's arithmetic is unchanged); the IMAGES agree with gg_blend_fwd_pair to fp32 rounding — |difference| <=
 * ~1e-6 (1 + |value|) for colours of ordinary range; per channel and batch the error is bounded by 2^-22 of the batch's
 * largest |colour| times the pixel's sum of alpha T — not bit for bit (the exact kernel sums in list order).  Needs
 * channels % 4 == 0 and 16-byte aligned out_img / background; otherwise it runs the exact kernel.  What gsplat computes
 * here is a CUDA fma chain with __expf (rasterize_forward / nd_rasterize_forward): no summation order that could be
 * matched bit for bit on either kernel. */
int gg_blend_fwd_pair_fast(int channels, int channels2, int num_points, int img_height, int img_width,
                           const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                           const float *conics, const float *colors, const float *colors2,
                           const float *opacity, const float *background, const float *background2,
                           float *out_img, float *out_img2, float *final_Ts, int32_t *final_idx, void *ws,
                           size_t ws_bytes, gg_stream_t stream);
/* gg_blend_fwd_pair (fast == 0) / gg_blend_fwd_pair_fast (fast != 0) on a workspace that already holds the packed
 * records of these Gaussians — gg_view_fwd's `records`, or the workspace of an earlier forward over the same xys /
 * conics / opacity: the packing pass over the Gaussians is not run.  The backward entries take that workspace as
 * always. */
int gg_blend_fwd_pair_packed(int channels, int channels2, int num_points, int img_height, int img_width,
                             const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *colors,
                             const float *colors2, const float *background, const float *background2, float *out_img,
                             float *out_img2, float *final_Ts, int32_t *final_idx, void *ws, size_t ws_bytes, int fast,
                             gg_stream_t stream);

/* gg_blend_bwd replaces gsplat `_C.rasterize_backward` / `_C.nd_rasterize_backward`.
 * v_xy (N,2), v_conic (N,3), v_colors (N,C), v_opacity (N,) are fully written.
 * ws_from_forward != 0: `ws` is the very workspace the matching gg_blend_fwd call (same xys, conics,
 * opacity) ran with, untouched since — its packed per-Gaussian records are reused instead of being
 * packed again (gsplat's backward re-reads xys/conics/opacities itself; this saves one pass).
 * geom_stride / color_stride: floats between consecutive Gaussians in v_xy, v_conic, v_opacity /
 * in v_colors; 0 = the dense arrays above.  geom_stride >= 6 means ONE interleaved record per
 * Gaussian, {xy.x, xy.y, conic a, b, c, opacity[, colours]}: v_conic = v_xy + 2, v_opacity =
 * v_xy + 5 (and v_colors = v_xy + 6 with color_stride = geom_stride when the colours are part of
 * it).  The kernels add with float atomics; an interleaved record puts all of a Gaussian's
 * atomics of one instruction on one cache line instead of four. */
int gg_blend_bwd(int channels, int num_points, int img_height, int img_width,
                 const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                 const float *conics, const float *colors, const float *opacity,
                 const float *background, const float *final_Ts, const int32_t *final_idx,
                 const float *v_out_img, float *v_xy, float *v_conic, float *v_colors,
                 float *v_opacity, int geom_stride, int color_stride, void *ws, size_t ws_bytes,
                 int flags, gg_stream_t stream);
#define GG_BWD_WS_FROM_FORWARD 1
#define GG_BWD_ACCUMULATE_COLORS 2
#define GG_BWD_ACCUMULATE_GEOM 4

/* gg_blend_bwd_pair: the backward of gg_blend_fwd_pair — colors (N, channels >= 32) with v_out_img and colors2
 * (N, channels2 <= 8) with v_out_img2 — in ONE walk of the tile lists for the first 32 channels and the second
 * array together (alpha, T and the geometry gradients are those of all channels at once; D = <colour, v_out>
 * over 40 channels and both colour-gradient products run on the matrix pipe).  v_xy, v_conic, v_opacity hold
 * the geometry gradients of BOTH arrays (what autograd would sum over two gg_blend_bwd calls); v_colors /
 * v_colors2 the colour gradients, rows color_stride / color_stride2 floats apart (0 = dense).  geom_stride as
 * in gg_blend_bwd; v_colors2 = v_xy + 6 with color_stride2 = geom_stride puts the second array's gradients
 * into the interleaved record.  The second cotangent (H, W, channels2) is handed over as num_parts (1..3) images
 * of v_out_img2_channels[k] consecutive channels each (host arrays) — the caller's rgb | depth | normal cotangents
 * are read where they are, not concatenated first.
 * flags: GG_BWD_WS_FROM_FORWARD, GG_BWD_ACCUMULATE_COLORS (for v_colors). */
int gg_blend_bwd_pair(int channels, int channels2, int num_points, int img_height, int img_width,
                      const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                      const float *conics, const float *colors, const float *colors2, const float *opacity,
                      const float *background, const float *background2, const float *final_Ts,
                      const int32_t *final_idx, const float *v_out_img, const float *const *v_out_img2_parts,
                      const int *v_out_img2_channels, int num_parts, float *v_xy,
                      float *v_conic, float *v_colors, float *v_colors2, float *v_opacity, int geom_stride,
                      int color_stride, int color_stride2, void *ws, size_t ws_bytes, int flags,
                      gg_stream_t stream);

/* gg_blend_bwd_deterministic: gg_blend_bwd with bit-reproducible results.  gsplat's backward
 * (csrc/backward.cu: one atomicAdd per warp per Gaussian) and gg_blend_bwd add in whatever order the
 * hardware schedules; here the kernels store the total of every (tile-list entry, 8x8 quadrant) into a slab
 * and a second pass (stable sort of the list by Gaussian id, one wave per Gaussian) sums each Gaussian's
 * entries in list order, quadrants 0..3 inside an entry.  Same arguments and flags as gg_blend_bwd plus
 * num_intersects (length of gaussian_ids_sorted) and a second workspace of
 * gg_blend_bwd_deterministic_workspace(num_points, channels, num_intersects) bytes
 * (16 (channels + 6 ceil(channels / 32)) bytes per list entry dominate). */
size_t gg_blend_bwd_deterministic_workspace(int num_points, int channels, int64_t num_intersects);
int gg_blend_bwd_deterministic(int channels, int num_points, int img_height, int img_width,
                               const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                               const float *conics, const float *colors, const float *opacity,
                               const float *background, const float *final_Ts, const int32_t *final_idx,
                               const float *v_out_img, float *v_xy, float *v_conic, float *v_colors,
                               float *v_opacity, int geom_stride, int color_stride, void *ws, size_t ws_bytes,
                               int flags, int64_t num_intersects, void *det_ws, size_t det_ws_bytes,
                               gg_stream_t stream);

/* ---- feature up-projection MLP ------------------------------------------------
 * Replaces the forward of the reference's `MLP(32, 512, hidden_list=[128])` module
 * (nerfstudio/models/gaussian_splatting.py:198-213; `self.fea_up`, called on every pixel of the
 * rendered feature image at nerfstudio/pipelines/base_pipeline.py:408):
 *     y = relu(x @ w1^T + b1) @ w2^T + b2,  x (num_rows, in_dim), w1 (128, in_dim), b1 (128),
 *     w2 (out_dim, 128), b2 (out_dim), y (num_rows, out_dim), all fp32 row-major (torch Linear
 *     layout).  hidden_dim must be 128, in_dim 8/16/32/64, out_dim a multiple of 32. */
int gg_mlp_fwd(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
               const float *w1, const float *b1, const float *w2, const float *b2, float *y,
               gg_stream_t stream);

/* Tuning entry (round 3): how many 32-channel blocks of a wide colour array one forward walk takes — in the pair walk
 * of gg_blend_fwd_pair (1, 2 or 4) and in the walks of the remaining chunks / of gg_blend_fwd (1..4).  Images are
 * bit-identical for every setting; returns the previous pair value.  Defaults: see csrc/blend.hip. */
int gg_debug_set_fwd_blocks(int pair_blocks, int chunk_blocks);

/* The same forward four times faster (round 3): both layers as products of fp16 two-piece operands scaled by powers of
 * two on v_mfma_f32_16x16x32_f16 — as accurate against a double-precision sum as the fp32 matrix instruction
 * (tools/check_f16split.hip), but not gg_mlp_fwd's summation order: the two agree to ~1e-6 of the largest output
 * (tests/test_gpu_parity.py), not bit for bit.  in_dim 32 / 64 / 128, out_dim a multiple of 16 (<= 3968: LDS); `ws`:
 * gg_mlp_fwd_fast_workspace(in_dim, 128, out_dim) bytes, 16-byte aligned, rewritten by every call (the packed
 * weights: 0.3 MB at out_dim 512).  What the reference runs here is cuBLAS behind nn.Linear
 * (gaussian_splatting.py:198-213): no summation order to match on that side. */
size_t gg_mlp_fwd_fast_workspace(int in_dim, int hidden_dim, int out_dim);
int gg_mlp_fwd_fast(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
                    const float *w1, const float *b1, const float *w2, const float *b2, float *y,
                    void *ws, size_t ws_bytes, gg_stream_t stream);

/* Backward of the same module (the reference evaluates fea_up on 1000 sampled pixels per training
 * step, gaussian_splatting.py:917): g = dL/dy (num_rows, out_dim) -> v_x (num_rows, in_dim), v_w1 (128,
 * in_dim), v_b1 (128), v_w2 (out_dim, 128), v_b2 (out_dim), all fully written.  in_dim 8..128, out_dim
 * <= 1024.  Sized for 10^3-10^5 rows (one launch, row tiles through LDS, float atomics for the weight
 * gradients); for the 1.9 M-pixel render pass a library GEMM is the right tool. */
int gg_mlp_bwd(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
               const float *w1, const float *b1, const float *w2, const float *g, float *v_x,
               float *v_w1, float *v_b1, float *v_w2, float *v_b2, gg_stream_t stream);

/* ---- cosine-similarity loss --------------------------------------------------------
 * Replace the reference's `cosine_similarity_loss` (gaussian_splatting.py:113-118; contrastive feature
 * loss over 800 pixel pairs per mask :909-914, `up_loss` :917-918) for a, b stored (num_points,
 * channels): sim_m = <a_m, b_m> / (max(|a_m|, 1e-12) max(|b_m|, 1e-12)); *sim_sum = sum_m sim_m (the
 * loss is 1 - sim_sum / num_points).  sim / norm_a / norm_b (num_points,) are kept for the backward,
 * which takes v_loss from device memory (1 float) and writes v_a, v_b (num_points, channels). */
int gg_cosine_loss_fwd(int64_t num_points, int channels, const float *a, const float *b, float *sim,
                       float *norm_a, float *norm_b, float *sim_sum, gg_stream_t stream);
int gg_cosine_loss_bwd(int64_t num_points, int channels, const float *a, const float *b,
                       const float *sim, const float *norm_a, const float *norm_b,
                       const float *v_loss, float *v_a, float *v_b, gg_stream_t stream);

/* ---- image-space main loss (SURVEY 8f-4 tail) ---------------------------------------------------------
 * Replace, in the reference's get_loss_dict (gaussian_splatting.py:882-885, :931; self.ssim :284 =
 * pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3), requirements.txt:199):
 *     Ll1 = abs(gt[valid] - rgb[valid]).mean();  gt[~valid] = 0;  rgb[~valid] = 0;
 *     main_loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * (1 - ssim(gt, rgb))
 * rgb, gt (H, W, 3) fp32, rgb with rgb_pixel_stride >= 3 floats between pixels (a channel slice of a wider image
 * is read in place); valid (H, W) bytes or NULL (all valid).  out3 = {main_loss, Ll1, ssim} (device).
 * The forward leaves what the backward needs (three derivative maps per channel, the valid count) in `ws`
 * (gg_image_loss_workspace bytes); the backward takes the SAME workspace, untouched, and v_main (1 float, device)
 * and writes v_rgb (H, W, 3) = v_main * d main_loss / d rgb, zero at invalid pixels.  H, W >= 11. */
size_t gg_image_loss_workspace(int img_height, int img_width);
int gg_image_loss_fwd(int img_height, int img_width, const float *rgb, int rgb_pixel_stride, const float *gt,
                      const uint8_t *valid, float ssim_lambda, float *out3, void *ws, size_t ws_bytes,
                      gg_stream_t stream);
int gg_image_loss_bwd(int img_height, int img_width, const float *rgb, int rgb_pixel_stride, const float *gt,
                      const uint8_t *valid, float ssim_lambda, const float *v_main, const void *ws, size_t ws_bytes,
                      float *v_rgb, gg_stream_t stream);

/* Depth and normal losses of get_loss_dict (gaussian_splatting.py:879-880) over the pixels where mask != 0:
 *     depth_loss  = F.l1_loss(depth[m], gt_depth[m])
 *     normal_loss = 0.5 F.mse_loss(normal[:, m], gt_normal[:, m]) + 0.5 cosine_similarity_loss(normal[:, m], gt_normal[:, m])
 * depth / gt_depth: element p at base[p * stride]; normal / gt_normal: channel c of pixel p at
 * base[p * pixel_stride + c * channel_stride] (the model's images are pixel-major, the reference's ground truth
 * channel-major).  out3 = {depth_loss, normal_loss, number of masked pixels} (device).  The backward takes the
 * forward's workspace (gg_geom_loss_workspace bytes), the two loss cotangents from device memory (1 float each) and
 * writes dense v_depth (num_pixels,), v_normal (num_pixels, 3), zero outside the mask. */
size_t gg_geom_loss_workspace(void);
int gg_geom_loss_fwd(int64_t num_pixels, const float *depth, int depth_stride, const float *gt_depth,
                     int gt_depth_stride, const float *normal, int normal_pixel_stride, int normal_channel_stride,
                     const float *gt_normal, int gt_normal_pixel_stride, int gt_normal_channel_stride,
                     const uint8_t *mask, float *out3, void *ws, size_t ws_bytes, gg_stream_t stream);
int gg_geom_loss_bwd(int64_t num_pixels, const float *depth, int depth_stride, const float *gt_depth,
                     int gt_depth_stride, const float *normal, int normal_pixel_stride, int normal_channel_stride,
                     const float *gt_normal, int gt_normal_pixel_stride, int gt_normal_channel_stride,
                     const uint8_t *mask, const float *v_depth_loss, const float *v_normal_loss, const void *ws,
                     size_t ws_bytes, float *v_depth, float *v_normal, gg_stream_t stream);

/* ---- densification, culling and the optimizer step ---------------------------------
 * The per-Gaussian optimizer-side work of the reference model, which it does with torch indexing,
 * torch.cat and one torch.optim.Adam per parameter group:
 *   statistics   GaussianSplattingModel.after_train          gaussian_splatting.py:373-393
 *   masks        refinement_after :412-421,:430-431 ; cull_gaussians :485-496
 *   cull         cull_gaussians :497-502 + remove_from_optim :333-350
 *   split / dup  split_gaussians :504-531, dup_gaussians :533-546, torch.cat :434-439, dup_in_optim :352-371
 *   Adam         Optimizers.optimizer_step_all               engine/optimizers.py:158-171
 * Row arrays are (num_rows, row_floats) fp32, contiguous; byte masks are 0 / non-0. */
typedef struct {
    const float *src; /* (num_rows, row_floats) */
    float *dst;       /* destination array (capacity: see each function) */
    int row_floats;
    int kind; /* GG_ROWS_* (gg_densify_rows only; ignored by gg_compact_rows) */
} gg_row_array_t;
#define GG_ROWS_COPY 0     /* appended rows copy their source row                                    */
#define GG_ROWS_MEANS 1    /* split samples: R(q/|q|) (exp(scale) * z) + mean  (:509-516); 3 floats  */
#define GG_ROWS_SCALES 2   /* split sources and samples: log(exp(scale) / size_fac) (:524-526); 3 floats */
#define GG_ROWS_ZERO_NEW 3 /* Adam moments: appended rows are zero (dup_in_optim :352-371)            */

/* bytes of scratch for gg_mask_scan / gg_compact_rows over num_rows rows */
size_t gg_rows_workspace(int num_rows);
/* ranks[i] = number of selected rows before i (selected = mask != 0, or == 0 with invert);
 * *total_out (device int64) = number selected.  One launch (decoupled look-back scan). */
int gg_mask_scan(int num_rows, const uint8_t *mask, int invert, int32_t *ranks, int64_t *total_out,
                 void *ws, size_t ws_bytes, gg_stream_t stream);
/* Stream compaction: rows with deleted_mask == 0 of every array move, in order, to the front of its
 * dst (capacity num_rows rows; src != dst).  Replaces `t[~culls]` on the 6 parameters and the 12
 * moment tensors (<= 24 arrays, one launch).  *num_kept_out (device int64) = rows kept.
 * `arrays` is a HOST array of descriptors. */
int gg_compact_rows(int num_rows, const uint8_t *deleted_mask, int num_arrays,
                    const gg_row_array_t *arrays, int64_t *num_kept_out, void *ws, size_t ws_bytes,
                    gg_stream_t stream);
/* Append rows: dst = [num_rows old | num_samples x num_split split samples (sample-major) | num_dup
 * duplicates]; dst capacity num_rows + num_samples*num_split + num_dup rows.  split_ranks / dup_ranks
 * and the counts come from gg_mask_scan; samples (num_samples*num_split, 3) are the caller's N(0,1)
 * draws (torch.randn, :508); means / scales / quats are the CURRENT parameter arrays. */
int gg_densify_rows(int num_rows, const uint8_t *split_mask, const uint8_t *dup_mask,
                    const int32_t *split_ranks, const int32_t *dup_ranks, int num_split, int num_dup,
                    int num_samples, const float *samples, float size_fac, const float *means,
                    const float *scales, const float *quats, int num_arrays,
                    const gg_row_array_t *arrays, gg_stream_t stream);
/* after_train (:373-393).  first != 0: the accumulators are (re)initialised as the reference does
 * when they are None (grad norms of every Gaussian, counts 1, max_2dsize 0 then max over visible). */
int gg_densify_stats(int num_points, const float *xys_grad, const int32_t *radii, int max_image_dim,
                     int first, float *grad_norm_accum, float *vis_counts, float *max_2dsize,
                     gg_stream_t stream);
/* split / dup masks of refinement_after (:412-421, :430-431); scales are log-scales (N,3).  The duplicate test
 * runs on the scales split_gaussians has already shrunk in place (log(exp(s) / size_fac) for the split rows,
 * :524-526), as the reference's does: a Gaussian can be in both masks. */
int gg_densify_masks(int num_points, const float *grad_norm_accum, const float *vis_counts,
                     const float *max_2dsize, const float *scales, int max_image_dim,
                     float densify_grad_thresh, float densify_size_thresh, float split_screen_size,
                     int use_screen_size, float size_fac, uint8_t *split_mask, uint8_t *dup_mask,
                     gg_stream_t stream);
/* cull mask of cull_gaussians (:485-496); opacities are logits (N,), scales log-scales (N,3). */
int gg_cull_mask(int num_points, const float *opacities, const float *scales, const float *max_2dsize,
                 float cull_alpha_thresh, float cull_scale_thresh, float cull_screen_size,
                 int use_scale, int use_screen_size, uint8_t *deleted_mask, gg_stream_t stream);

/* One Adam step (torch.optim.Adam, amsgrad off) of up to GG_ADAM_MAX_GROUPS parameter arrays in ONE
 * launch; every group has its own hyper-parameters and step count, as the reference's one optimizer
 * per group does (method_configs.py:618-660: eps 1e-15, lr 1.6e-4 ... 0.05).  step is the step
 * number being taken (>= 1).  16-byte aligned arrays take the float4 path.  zero_grad != 0 clears the gradients
 * in the same pass.  `groups` is a HOST array. */
#define GG_ADAM_MAX_GROUPS 8
typedef struct {
    float *param, *grad, *exp_avg, *exp_avg_sq;
    int64_t numel;
    double lr, beta1, beta2, eps, weight_decay;
    int64_t step;
} gg_adam_group_t;
int gg_adam_step(int num_groups, const gg_adam_group_t *groups, int zero_grad, gg_stream_t stream);

/* ---- in-library kernel timing (measurement only; off by default) --------------------------------
 * When enabled, every launch of the kernels below is bracketed by a hipEvent pair recorded on the
 * launch stream, so bench.py can report the average duration of exactly that kernel over its
 * timed region (the number a rocprofv3 --kernel-trace --stats run must agree with).
 * gg_prof_get synchronises on the recorded events.  Kernel ids: */
#define GG_K_PROJECT_FWD 0
#define GG_K_PROJECT_BWD 1
#define GG_K_SH_FWD 2
#define GG_K_SH_BWD 3
#define GG_K_BIN_SORT 4   /* the whole gg_bin_sort launch sequence */
#define GG_K_BLEND_PREP 5
#define GG_K_QUAT_FWD 6
#define GG_K_QUAT_BWD 7
#define GG_K_MLP_FWD 8
#define GG_K_MLP_BWD 9
#define GG_K_BLEND_FWD 10 /* + width index: template widths {1,3,4,8,16,32} -> 0..5 */
#define GG_K_BLEND_BWD 20 /* + width index */
#define GG_K_BLEND_FWD_PAIR 16 /* 32 channels + a second array of <= 8 in one walk */
#define GG_K_BLEND_BWD_PAIR 17
#define GG_K_COMPACT 26
#define GG_K_DENSIFY 27
#define GG_K_ADAM 28
#define GG_K_VIEW_BWD 29      /* gg_view_bwd: the per-Gaussian backward of a view in one kernel */
#define GG_K_ACTIVATE_FWD 30
#define GG_K_ACTIVATE_BWD 31
#define GG_K_COUNT 18         /* gg_count_intersects */
#define GG_K_TAIL_SPLIT 19    /* gg_shade_tail_bwd_split */
#define GG_K_VIEW_FWD 32      /* gg_view_fwd: activations + projection of a view in one kernel */
#define GG_K_IDS 40           /* ids are below this */
#define GG_PROF_NUM_KERNELS 32
int gg_prof_enable(int on);
int gg_prof_reset(void);
int gg_prof_get(int kernel_id, int *launches, double *total_ms);
const char *gg_prof_name(int kernel_id);

/* y[i] = gg_expf(x[i]) on the device — lets the tests pin the GPU exponential bit-for-bit
 * against the oracle's (gg_constants.h documents the operation sequence). */
int gg_expf_array(int n, const float *x, float *y, gg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GG_RASTER_H */
