// densify.hip — SURVEY.md §8 row f3: the per-Gaussian optimizer-side work of the reference model,
// as streaming HIP kernels behind the C ABI.
//
//   after_train statistics            nerfstudio/models/gaussian_splatting.py:373-393
//   densify masks (split / dup)        :412-421, :430-431
//   cull mask                          :480-496
//   cull = stream compaction           :497-502 (6 parameters) + remove_from_optim :333-350 (12 Adam moments)
//   split / dup = append rows          :504-546, :434-443 + dup_in_optim :352-371
//   Adam step per parameter group      nerfstudio/engine/optimizers.py:158-171 (torch.optim.Adam, eps 1e-15)
//
// All of it is HBM-bound row traffic: no MFMA, no LDS tiling beyond the per-block rank table.  The
// compaction is ONE launch: keep-flags -> decoupled look-back prefix sum (scan.h) -> gather of up to
// 24 row arrays (the 6 parameter tensors and their 12 moment tensors together), reads fully
// coalesced, writes in contiguous runs of kept rows.
#include "scan.h"

#define DN_THREADS 256
#define DN_ROWS 1024          // rows per workgroup of the compaction / densify kernels
#define GG_MAX_ROW_ARRAYS 24

struct RowArrays {
    int n;
    gg_row_array_t a[GG_MAX_ROW_ARRAYS];
};

// ---------------------------------------------------------------------------------------------
// mask -> exclusive ranks + total (one launch)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DN_THREADS) void mask_scan_kernel(int N, const uint8_t *__restrict__ mask,
                                                               int invert, int32_t *__restrict__ ranks,
                                                               ScanState *st, int nblocks) {
    __shared__ unsigned int s_slot, s_excl, wsum[4];
    const int bid = scan_ticket(st, &s_slot);
    constexpr int PER = DN_ROWS / DN_THREADS;
    const int base = bid * DN_ROWS + threadIdx.x * PER;
    unsigned int f[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = base + k;
        f[k] = (i < N) ? (unsigned int)((mask[i] != 0) != (invert != 0)) : 0u;
        sum += f[k];
    }
    unsigned int total;
    unsigned int ex = scan_block256(sum, wsum, total);
    ex += scan_lookback(st, bid, nblocks, total, &s_excl);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = base + k;
        if (i < N) ranks[i] = (int32_t)ex;
        ex += f[k];
    }
}

static int dn_blocks(int N) { return (N + DN_ROWS - 1) / DN_ROWS; }

extern "C" size_t gg_rows_workspace(int num_rows) { return gg_scan_state_bytes(dn_blocks(num_rows)); }

extern "C" int gg_mask_scan(int num_rows, const uint8_t *mask, int invert, int32_t *ranks,
                            int64_t *total_out, void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0, "num_rows < 0");
    GG_REQUIRE(total_out != nullptr, "null total_out");
    hipStream_t s = (hipStream_t)stream;
    const int nb = dn_blocks(num_rows);
    if (num_rows == 0) {
        if (hipMemsetAsync(total_out, 0, sizeof(int64_t), s) != hipSuccess) return GG_ERR_LAUNCH;
        return GG_OK;
    }
    GG_REQUIRE(mask && ranks, "null pointer");
    if (ws == nullptr || ws_bytes < gg_scan_state_bytes(nb)) {
        gg_set_error("gg_mask_scan: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    ScanState *st = (ScanState *)ws;
    if (hipMemsetAsync(st, 0, gg_scan_state_bytes(nb), s) != hipSuccess) return GG_ERR_LAUNCH;
    hipLaunchKernelGGL(mask_scan_kernel, dim3(nb), dim3(DN_THREADS), 0, s, num_rows, mask, invert, ranks,
                       st, nb);
    if (hipMemcpyAsync(total_out, &st->total, sizeof(int64_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return GG_ERR_LAUNCH;
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// cull: compaction of many row arrays in one launch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DN_THREADS) void compact_rows_kernel(int N, const uint8_t *__restrict__ deleted,
                                                                  RowArrays arrs, ScanState *st, int nblocks) {
    __shared__ unsigned int s_slot, s_excl, wsum[4];
    __shared__ int s_dst[DN_ROWS];      // destination row of each row of this block, -1 = deleted
    const int bid = scan_ticket(st, &s_slot);
    constexpr int PER = DN_ROWS / DN_THREADS;
    const int base = bid * DN_ROWS;
    const int tbase = base + threadIdx.x * PER;
    unsigned int f[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tbase + k;
        f[k] = (i < N) ? (unsigned int)(deleted[i] == 0) : 0u;
        sum += f[k];
    }
    unsigned int total;
    unsigned int ex = scan_block256(sum, wsum, total);
    ex += scan_lookback(st, bid, nblocks, total, &s_excl);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        s_dst[threadIdx.x * PER + k] = f[k] ? (int)ex : -1;
        ex += f[k];
    }
    __syncthreads();
    const int rows = min(DN_ROWS, N - base);
    for (int a = 0; a < arrs.n; ++a) {
        const int w = arrs.a[a].row_floats;
        const float *__restrict__ src = arrs.a[a].src + (size_t)base * w;
        float *__restrict__ dst = arrs.a[a].dst;
        const int count = rows * w;
        for (int e = threadIdx.x; e < count; e += DN_THREADS) {
            const int r = e / w, c = e - r * w;
            const int d = s_dst[r];
            if (d >= 0) dst[(size_t)d * w + c] = src[e];
        }
    }
}

extern "C" int gg_compact_rows(int num_rows, const uint8_t *deleted_mask, int num_arrays,
                               const gg_row_array_t *arrays, int64_t *num_kept_out, void *ws,
                               size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0, "num_rows < 0");
    GG_REQUIRE(num_arrays >= 0 && num_arrays <= GG_MAX_ROW_ARRAYS, "num_arrays out of range (<= 24)");
    GG_REQUIRE(num_kept_out != nullptr, "null num_kept_out");
    hipStream_t s = (hipStream_t)stream;
    if (num_rows == 0) {
        if (hipMemsetAsync(num_kept_out, 0, sizeof(int64_t), s) != hipSuccess) return GG_ERR_LAUNCH;
        return GG_OK;
    }
    GG_REQUIRE(deleted_mask != nullptr && (num_arrays == 0 || arrays != nullptr), "null pointer");
    RowArrays ra;
    ra.n = num_arrays;
    for (int a = 0; a < num_arrays; ++a) {
        GG_REQUIRE(arrays[a].src && arrays[a].dst && arrays[a].row_floats > 0, "bad row array descriptor");
        GG_REQUIRE(arrays[a].src != arrays[a].dst, "in-place compaction is not supported");
        ra.a[a] = arrays[a];
    }
    const int nb = dn_blocks(num_rows);
    if (ws == nullptr || ws_bytes < gg_scan_state_bytes(nb)) {
        gg_set_error("gg_compact_rows: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    ScanState *st = (ScanState *)ws;
    if (hipMemsetAsync(st, 0, gg_scan_state_bytes(nb), s) != hipSuccess) return GG_ERR_LAUNCH;
    gg_prof_begin(GG_K_COMPACT, s);
    hipLaunchKernelGGL(compact_rows_kernel, dim3(nb), dim3(DN_THREADS), 0, s, num_rows, deleted_mask, ra, st,
                       nb);
    gg_prof_end(GG_K_COMPACT, s);
    if (hipMemcpyAsync(num_kept_out, &st->total, sizeof(int64_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return GG_ERR_LAUNCH;
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// split / dup: append rows
// ---------------------------------------------------------------------------------------------
// Destination layout (reference :434-439): [ N old rows | nsamps x n_split split samples, sample-major
// (`.repeat(samps, 1)`: row N + s*n_split + rank) | n_dup duplicated rows ].
//   GG_ROWS_COPY      old rows and new rows copy the source row
//   GG_ROWS_MEANS     new split rows: R(q/|q|) (exp(scale) * z) + mean (:509-516); dup rows copy
//   GG_ROWS_SCALES    split sources AND their samples: log(exp(scale) / size_fac) (:524-526); dup rows copy
//   GG_ROWS_ZERO_NEW  Adam moments: old rows copy, every appended row is zero (dup_in_optim :352-371)
__device__ __forceinline__ void quat_rot(const float *q, float (&R)[9]) {
    // q / |q| (the caller's normalisation, :513) then gsplat's quat_to_rotmat (F.normalize + matrix)
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    const float n2 = fmaxf(sqrtf(w * w + x * x + y * y + z * z), GG_QUAT_NORM_EPS);
    w /= n2; x /= n2; y /= n2; z /= n2;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

__global__ __launch_bounds__(DN_THREADS) void densify_rows_kernel(
    int N, const uint8_t *__restrict__ split_mask, const uint8_t *__restrict__ dup_mask,
    const int32_t *__restrict__ split_rank, const int32_t *__restrict__ dup_rank, int n_split, int n_dup,
    int nsamps, const float *__restrict__ samples, float size_fac, const float *__restrict__ means,
    const float *__restrict__ scales, const float *__restrict__ quats, RowArrays arrs) {
    __shared__ int s_srank[DN_ROWS], s_drank[DN_ROWS];   // -1 = not selected
    const int base = blockIdx.x * DN_ROWS;
    const int rows = min(DN_ROWS, N - base);
    for (int r = threadIdx.x; r < rows; r += DN_THREADS) {
        const int i = base + r;
        s_srank[r] = (split_mask && split_mask[i]) ? split_rank[i] : -1;
        s_drank[r] = (dup_mask && dup_mask[i]) ? dup_rank[i] : -1;
    }
    __syncthreads();
    const size_t dup_base = (size_t)N + (size_t)nsamps * n_split;
    for (int a = 0; a < arrs.n; ++a) {
        const int w = arrs.a[a].row_floats, kind = arrs.a[a].kind;
        const float *__restrict__ src = arrs.a[a].src + (size_t)base * w;
        float *__restrict__ dst = arrs.a[a].dst;
        const int count = rows * w;
        for (int e = threadIdx.x; e < count; e += DN_THREADS) {
            const int r = e / w, c = e - r * w;
            const int i = base + r;
            const float v = src[e];
            const int sr = s_srank[r], dr = s_drank[r];
            float old_v = v, split_v = v;
            if (kind == GG_ROWS_SCALES && sr >= 0) old_v = split_v = logf(expf(v) / size_fac);
            if (kind == GG_ROWS_ZERO_NEW) split_v = 0.0f;
            dst[(size_t)i * w + c] = old_v;
            if (sr >= 0) {
                float R[9];
                float sc[3];
                if (kind == GG_ROWS_MEANS) {
                    quat_rot(quats + 4 * (size_t)i, R);
                    for (int k = 0; k < 3; ++k) sc[k] = expf(scales[3 * (size_t)i + k]);
                }
                for (int sidx = 0; sidx < nsamps; ++sidx) {
                    const size_t row = (size_t)N + (size_t)sidx * n_split + sr;
                    float out = split_v;
                    if (kind == GG_ROWS_MEANS) {
                        const float *z = samples + 3 * ((size_t)sidx * n_split + sr);
                        const float s0 = sc[0] * z[0], s1 = sc[1] * z[1], s2 = sc[2] * z[2];
                        out = ((R[3 * c] * s0 + R[3 * c + 1] * s1) + R[3 * c + 2] * s2) + means[3 * (size_t)i + c];
                    }
                    dst[row * w + c] = out;
                }
            }
            // (dup_gaussians copies self.scales AFTER split_gaussians shrank the split rows in place: :524-526, :541)
            if (dr >= 0) dst[(dup_base + dr) * w + c] = (kind == GG_ROWS_ZERO_NEW) ? 0.0f : old_v;
        }
    }
}

extern "C" int gg_densify_rows(int num_rows, const uint8_t *split_mask, const uint8_t *dup_mask,
                               const int32_t *split_ranks, const int32_t *dup_ranks, int num_split,
                               int num_dup, int num_samples, const float *samples, float size_fac,
                               const float *means, const float *scales, const float *quats,
                               int num_arrays, const gg_row_array_t *arrays, gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0 && num_split >= 0 && num_dup >= 0 && num_samples >= 0, "negative size");
    GG_REQUIRE(num_arrays >= 0 && num_arrays <= GG_MAX_ROW_ARRAYS, "num_arrays out of range (<= 24)");
    if (num_rows == 0) return GG_OK;
    GG_REQUIRE(num_arrays == 0 || arrays != nullptr, "null pointer");
    GG_REQUIRE(num_split == 0 || (split_mask && split_ranks), "split mask / ranks missing");
    GG_REQUIRE(num_dup == 0 || (dup_mask && dup_ranks), "dup mask / ranks missing");
    GG_REQUIRE(size_fac > 0.0f, "size_fac must be positive");
    RowArrays ra;
    ra.n = num_arrays;
    for (int a = 0; a < num_arrays; ++a) {
        GG_REQUIRE(arrays[a].src && arrays[a].dst && arrays[a].row_floats > 0, "bad row array descriptor");
        GG_REQUIRE(arrays[a].kind >= GG_ROWS_COPY && arrays[a].kind <= GG_ROWS_ZERO_NEW, "bad row kind");
        if (arrays[a].kind == GG_ROWS_MEANS) {
            GG_REQUIRE(arrays[a].row_floats == 3, "GG_ROWS_MEANS rows have 3 floats");
            GG_REQUIRE(num_split == 0 || num_samples == 0 || (samples && means && scales && quats),
                       "split sampling needs samples, means, scales and quats");
        }
        if (arrays[a].kind == GG_ROWS_SCALES) GG_REQUIRE(arrays[a].row_floats == 3, "GG_ROWS_SCALES rows have 3 floats");
        ra.a[a] = arrays[a];
    }
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_DENSIFY, s);
    hipLaunchKernelGGL(densify_rows_kernel, dim3(dn_blocks(num_rows)), dim3(DN_THREADS), 0, s, num_rows,
                       num_split ? split_mask : nullptr, num_dup ? dup_mask : nullptr, split_ranks, dup_ranks,
                       num_split, num_dup, num_samples, samples, size_fac, means, scales, quats, ra);
    gg_prof_end(GG_K_DENSIFY, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// statistics and masks (elementwise)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const float *__restrict__ xys_grad,
                                                            const int32_t *__restrict__ radii,
                                                            float max_dim, int first,
                                                            float *__restrict__ grad_norm,
                                                            float *__restrict__ vis_counts,
                                                            float *__restrict__ max_2dsize) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float gx = xys_grad[2 * (size_t)i], gy = xys_grad[2 * (size_t)i + 1];
    const float g = sqrtf(gx * gx + gy * gy);
    const int rad = radii[i];
    const bool vis = rad > 0;
    if (first) {                     // :378-380: every Gaussian, visible or not
        grad_norm[i] = g;
        vis_counts[i] = 1.0f;
    } else if (vis) {                // :383-384
        vis_counts[i] = vis_counts[i] + 1.0f;
        grad_norm[i] = g + grad_norm[i];
    }
    if (vis) {                       // :389-392 (max_2Dsize starts at zero)
        const float m = first ? 0.0f : max_2dsize[i];
        max_2dsize[i] = fmaxf(m, (float)rad / max_dim);
    } else if (first) {
        max_2dsize[i] = 0.0f;
    }
}

extern "C" int gg_densify_stats(int num_points, const float *xys_grad, const int32_t *radii,
                                int max_image_dim, int first, float *grad_norm_accum, float *vis_counts,
                                float *max_2dsize, gg_stream_t stream) {
    GG_REQUIRE(num_points >= 0 && max_image_dim > 0, "bad size");
    if (num_points == 0) return GG_OK;
    GG_REQUIRE(xys_grad && radii && grad_norm_accum && vis_counts && max_2dsize, "null pointer");
    hipLaunchKernelGGL(densify_stats_kernel, dim3((num_points + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, num_points, xys_grad, radii, (float)max_image_dim, first,
                       grad_norm_accum, vis_counts, max_2dsize);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__device__ __forceinline__ float max_exp3(const float *s) {
    return fmaxf(fmaxf(expf(s[0]), expf(s[1])), expf(s[2]));
}

__global__ __launch_bounds__(256) void densify_masks_kernel(
    int N, const float *__restrict__ grad_norm, const float *__restrict__ vis_counts,
    const float *__restrict__ max_2dsize, const float *__restrict__ scales, float max_dim,
    float grad_thresh, float size_thresh, float split_screen_size, int use_screen, float size_fac,
    uint8_t *__restrict__ split_mask, uint8_t *__restrict__ dup_mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float avg = ((grad_norm[i] / vis_counts[i]) * 0.5f) * max_dim;
    const bool high = avg > grad_thresh;                               // :416-417
    const float *sc = scales + 3 * (size_t)i;
    float smax = max_exp3(sc);
    bool split = smax > size_thresh;                                   // :418
    if (use_screen) split = split || (max_2dsize[i] > split_screen_size);   // :419-420
    split = split && high;                                             // :421
    // split_gaussians has shrunk the scales of the split Gaussians IN PLACE by now (:524-526, called at :423-429),
    // so the duplicate test (:430) sees log(exp(s) / 1.6) for them: a high-gradient Gaussian with
    // thresh < max scale <= 1.6 thresh, or a small one split for its screen size, is split AND duplicated
    if (split) {
        const float sh[3] = {logf(expf(sc[0]) / size_fac), logf(expf(sc[1]) / size_fac), logf(expf(sc[2]) / size_fac)};
        smax = max_exp3(sh);
    }
    const bool dup = (smax <= size_thresh) && high;                    // :430-431
    split_mask[i] = split ? 1 : 0;
    dup_mask[i] = dup ? 1 : 0;
}

extern "C" int gg_densify_masks(int num_points, const float *grad_norm_accum, const float *vis_counts,
                                const float *max_2dsize, const float *scales, int max_image_dim,
                                float densify_grad_thresh, float densify_size_thresh,
                                float split_screen_size, int use_screen_size, float size_fac, uint8_t *split_mask,
                                uint8_t *dup_mask, gg_stream_t stream) {
    GG_REQUIRE(num_points >= 0 && max_image_dim > 0, "bad size");
    GG_REQUIRE(size_fac > 0.0f, "size_fac must be positive");
    if (num_points == 0) return GG_OK;
    GG_REQUIRE(grad_norm_accum && vis_counts && scales && split_mask && dup_mask, "null pointer");
    GG_REQUIRE(!use_screen_size || max_2dsize, "max_2dsize missing");
    hipLaunchKernelGGL(densify_masks_kernel, dim3((num_points + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, num_points, grad_norm_accum, vis_counts, max_2dsize, scales,
                       (float)max_image_dim, densify_grad_thresh, densify_size_thresh, split_screen_size,
                       use_screen_size, size_fac, split_mask, dup_mask);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__global__ __launch_bounds__(256) void cull_mask_kernel(int N, const float *__restrict__ opacities,
                                                        const float *__restrict__ scales,
                                                        const float *__restrict__ max_2dsize,
                                                        float alpha_thresh, float scale_thresh,
                                                        float screen_thresh, int use_scale, int use_screen,
                                                        uint8_t *__restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float sig = 1.0f / (1.0f + expf(-opacities[i]));
    bool cull = sig < alpha_thresh;                                                    // :486
    if (use_scale) {
        cull = cull || (max_exp3(scales + 3 * (size_t)i) > scale_thresh);              // :489-490
        if (use_screen) cull = cull || (max_2dsize[i] > screen_thresh);                // :494
    }
    mask[i] = cull ? 1 : 0;
}

extern "C" int gg_cull_mask(int num_points, const float *opacities, const float *scales,
                            const float *max_2dsize, float cull_alpha_thresh, float cull_scale_thresh,
                            float cull_screen_size, int use_scale, int use_screen_size,
                            uint8_t *deleted_mask, gg_stream_t stream) {
    GG_REQUIRE(num_points >= 0, "num_points < 0");
    if (num_points == 0) return GG_OK;
    GG_REQUIRE(opacities && scales && deleted_mask, "null pointer");
    GG_REQUIRE(!(use_scale && use_screen_size) || max_2dsize, "max_2dsize missing");
    hipLaunchKernelGGL(cull_mask_kernel, dim3((num_points + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       num_points, opacities, scales, max_2dsize, cull_alpha_thresh, cull_scale_thresh,
                       cull_screen_size, use_scale, use_screen_size, deleted_mask);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// fused Adam over all parameter groups (torch.optim.Adam semantics: no amsgrad, no maximize)
// ---------------------------------------------------------------------------------------------
// One launch walks every group (the reference steps one torch.optim.Adam per group:
// engine/optimizers.py:158-171 -> 6 optimizers x ~10 elementwise kernels).  Per element 16 bytes read
// (param, grad, exp_avg, exp_avg_sq) and 12 written (+4 when the gradient is zeroed in the same pass).
// Operation order (the oracle's, checked against torch.optim.Adam on the CPU):
//   g  = grad (+ weight_decay * p)
//   m  = fma(1-beta1, g - m, m)                       exp_avg.lerp_(grad, 1-beta1)
//   v  = fma((1-beta2) * g, g, v * beta2)             exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
//   p  = p + (-step_size * m) / (sqrt(v) / sqrt(bias_correction2) + eps)
// step_size = lr / bias_correction1 and sqrt(bias_correction2) are computed by the host in double, as
// torch does, and rounded to fp32.
struct AdamDev {
    float *param, *grad, *exp_avg, *exp_avg_sq;
    long long numel;
    int vec;   // all four arrays 16-byte aligned: float4 path
    float step_size, bc2_sqrt, w1, w2, beta2, eps, weight_decay;
};
struct AdamGroups {
    int n;
    AdamDev g[GG_ADAM_MAX_GROUPS];
    long long start4[GG_ADAM_MAX_GROUPS + 1];   // prefix of ceil(numel / 4) over the groups
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float wd, float w1, float w2,
                                          float beta2, float step_size, float bc2s, float eps) {
    if (wd != 0.0f) g = __builtin_fmaf(wd, p, g);
    m = __builtin_fmaf(w1, g - m, m);
    v = __builtin_fmaf(w2 * g, g, v * beta2);
    const float denom = sqrtf(v) / bc2s + eps;
    p = p + (-step_size * m) / denom;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamGroups G, int zero_grad) {
    const long long total4 = G.start4[G.n];
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total4;
         q += (long long)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int t = 1; t < GG_ADAM_MAX_GROUPS; ++t)
            if (t < G.n && q >= G.start4[t]) k = t;
        const AdamDev &grp = G.g[k];
        const long long e0 = (q - G.start4[k]) * 4;
        const float wd = grp.weight_decay, w1 = grp.w1, w2 = grp.w2, b2 = grp.beta2, ss = grp.step_size,
                    bc = grp.bc2_sqrt, eps = grp.eps;
        if (grp.vec && e0 + 4 <= grp.numel) {
            float4 p = *reinterpret_cast<float4 *>(grp.param + e0);
            const float4 g = *reinterpret_cast<const float4 *>(grp.grad + e0);
            float4 m = *reinterpret_cast<float4 *>(grp.exp_avg + e0);
            float4 v = *reinterpret_cast<float4 *>(grp.exp_avg_sq + e0);
            adam_elem(p.x, g.x, m.x, v.x, wd, w1, w2, b2, ss, bc, eps);
            adam_elem(p.y, g.y, m.y, v.y, wd, w1, w2, b2, ss, bc, eps);
            adam_elem(p.z, g.z, m.z, v.z, wd, w1, w2, b2, ss, bc, eps);
            adam_elem(p.w, g.w, m.w, v.w, wd, w1, w2, b2, ss, bc, eps);
            *reinterpret_cast<float4 *>(grp.param + e0) = p;
            *reinterpret_cast<float4 *>(grp.exp_avg + e0) = m;
            *reinterpret_cast<float4 *>(grp.exp_avg_sq + e0) = v;
            if (zero_grad) *reinterpret_cast<float4 *>(grp.grad + e0) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long long e = e0; e < grp.numel && e < e0 + 4; ++e) {
                float p = grp.param[e], m = grp.exp_avg[e], v = grp.exp_avg_sq[e];
                adam_elem(p, grp.grad[e], m, v, wd, w1, w2, b2, ss, bc, eps);
                grp.param[e] = p;
                grp.exp_avg[e] = m;
                grp.exp_avg_sq[e] = v;
                if (zero_grad) grp.grad[e] = 0.0f;
            }
        }
    }
}

extern "C" int gg_adam_step(int num_groups, const gg_adam_group_t *groups, int zero_grad,
                            gg_stream_t stream) {
    GG_REQUIRE(num_groups >= 0 && num_groups <= GG_ADAM_MAX_GROUPS, "num_groups out of range (<= 8)");
    if (num_groups == 0) return GG_OK;
    GG_REQUIRE(groups != nullptr, "null groups");
    AdamGroups G;
    G.n = num_groups;
    G.start4[0] = 0;
    for (int k = 0; k < num_groups; ++k) {
        const gg_adam_group_t &g = groups[k];
        GG_REQUIRE(g.numel >= 0 && g.step >= 1, "numel < 0 or step < 1");
        GG_REQUIRE(g.numel == 0 || (g.param && g.grad && g.exp_avg && g.exp_avg_sq), "null pointer");
        GG_REQUIRE((((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 3) == 0,
                   "parameter, gradient and moment arrays must be 4-byte aligned");
        GG_REQUIRE(g.beta1 >= 0.0 && g.beta1 < 1.0 && g.beta2 >= 0.0 && g.beta2 < 1.0, "betas must be in [0,1)");
        const double bc1 = 1.0 - pow(g.beta1, (double)g.step);
        const double bc2 = 1.0 - pow(g.beta2, (double)g.step);
        AdamDev &d = G.g[k];
        d.param = g.param; d.grad = g.grad; d.exp_avg = g.exp_avg; d.exp_avg_sq = g.exp_avg_sq;
        d.numel = g.numel;
        d.vec = (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15) == 0;
        d.step_size = (float)(g.lr / bc1);
        d.bc2_sqrt = (float)sqrt(bc2);
        d.w1 = (float)(1.0 - g.beta1);
        d.w2 = (float)(1.0 - g.beta2);
        d.beta2 = (float)g.beta2;
        d.eps = (float)g.eps;
        d.weight_decay = (float)g.weight_decay;
        G.start4[k + 1] = G.start4[k] + (g.numel + 3) / 4;
    }
    const long long total4 = G.start4[num_groups];
    if (total4 == 0) return GG_OK;
    const long long want = (total4 + 255) / 256;
    const int blocks = (int)(want < 256 * 16 ? want : 256 * 16);   // grid-stride above 16 workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_ADAM, s);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, G, zero_grad);
    gg_prof_end(GG_K_ADAM, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
