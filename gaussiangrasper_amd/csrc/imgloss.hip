// imgloss.hip — the image-space main loss of get_loss_dict (SURVEY 8f-4 tail), forward and backward:
//
//     Ll1 = torch.abs(gt_img[valid_mask, :] - outputs["rgb"][valid_mask, :]).mean()        (gaussian_splatting.py:882)
//     gt_img[~valid_mask, :] = 0.0;  outputs["rgb"][~valid_mask, :] = 0.0                    (:883-884)
//     simloss = 1 - self.ssim(gt_img ..., outputs["rgb"] ...)                                (:885; SSIM :284)
//     main_loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * simloss                            (:931)
//
// self.ssim is pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3): Gaussian window of 11 taps,
// sigma 1.5, `valid` separable filtering of X, Y, X^2, Y^2, XY.  In torch that is ten grouped conv2d launches over
// five 3 x H x W images forward and as many again backward, each through HBM; here the forward is ONE pass over the
// two images (16x16 output tiles, the 26x26 input windows and the row-filtered moments in LDS) that also leaves the
// three partial-derivative maps of the SSIM (d/d mu_rgb, d/d E[rgb^2], d/d E[gt rgb]) for the backward, and the
// backward is ONE pass that filters those maps back (transposed filter = the same symmetric window, zero outside)
// and adds the L1 sign term.  HBM-bound stencil: 24 B/pixel read + 36 B/pixel of maps written (forward), 36 + 24
// read + 12 written (backward); no atomics — per-workgroup partial sums, summed in a fixed order by one workgroup.
//
// The arithmetic follows oracle/gg_oracle.c:image_loss_* operation for operation (filter rows first, then columns;
// -ffp-contract=off), so the maps and the gradient image are bit-identical to the oracle; the two scalar sums are
// double sums in a different order and agree to 1e-7.
#include "gg_common.h"

#define IL_WIN 11
#define IL_T 16                  // tile edge
#define IL_IN (IL_T + IL_WIN - 1)  // 26: input window edge

struct IlWindow {
    float w[IL_WIN];
};

// header of the workspace (doubles): [0] sum of ssim over channels and positions, [1] sum |gt - rgb| over valid
// pixels x 3, [2] number of valid pixels x 3; then per-workgroup partials (3 doubles each); then the maps
#define IL_HEADER 8

__device__ __forceinline__ float il_ssim_point(const float *m, float &g_b, float &g_yy, float &g_xy) {
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    const float a = m[0], b = m[1];
    const float s1 = m[2] - a * a, s2 = m[3] - b * b, s12 = m[4] - a * b;
    const float ln = 2 * a * b + C1, ld = a * a + b * b + C1, cn = 2 * s12 + C2, cd = s1 + s2 + C2;
    const float L = ln / ld, CS = cn / cd;
    const float dL_db = (2 * a * ld - ln * 2 * b) / (ld * ld);
    const float dCS_ds2 = -cn / (cd * cd), dCS_ds12 = 2 / cd;
    g_b = CS * dL_db + L * (dCS_ds2 * (-2 * b) + dCS_ds12 * (-a));
    g_yy = L * dCS_ds2;
    g_xy = L * dCS_ds12;
    return L * CS;
}

__global__ __launch_bounds__(256) void image_loss_fwd_kernel(int H, int W, const float *__restrict__ rgb, int rs,
                                                             const float *__restrict__ gt,
                                                             const uint8_t *__restrict__ valid, IlWindow win,
                                                             double *__restrict__ partials,
                                                             float *__restrict__ maps) {
    __shared__ float sx[3][IL_IN][IL_IN + 1], sy[3][IL_IN][IL_IN + 1];     // gt, rgb (masked)
    __shared__ float hb[5][3][IL_IN][IL_T + 1];                            // row-filtered moments
    __shared__ double red[3][4];
    const int Ho = H - IL_WIN + 1, Wo = W - IL_WIN + 1;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * IL_T, y0 = blockIdx.y * IL_T;
    {   // the six loads of each of a thread's three window points requested together (the validity byte first: it is
        // NOT a condition of the loads, only of what is kept)
        constexpr int NPT = (IL_IN * IL_IN + 255) / 256;
        float vx[NPT][3], vy[NPT][3];
        bool keep[NPT];
        int rr[NPT], cc[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx / IL_IN, c = idx - r * IL_IN;
            rr[k] = r; cc[k] = c;
            const int y = y0 + r, x = x0 + c;
            const bool in = idx < IL_IN * IL_IN && y < H && x < W;
            const size_t p = in ? (size_t)y * W + x : 0;
            keep[k] = in && (!valid || valid[p]);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { vx[k][ch] = gt[3 * p + ch]; vy[k][ch] = rgb[(size_t)rs * p + ch]; }
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (tid + 256 * k < IL_IN * IL_IN) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    sx[ch][rr[k]][cc[k]] = keep[k] ? vx[k][ch] : 0.0f;
                    sy[ch][rr[k]][cc[k]] = keep[k] ? vy[k][ch] : 0.0f;
                }
            }
    }
    // L1 over this tile's own 16 x 16 input pixels (the masked images are zero where invalid, but so is the
    // difference only if both are masked: count validity explicitly)
    double l1 = 0.0, cnt = 0.0;
    {
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W) {
            const size_t p = (size_t)y * W + x;
            if (!valid || valid[p]) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) l1 += (double)fabsf(gt[3 * p + ch] - rgb[(size_t)rs * p + ch]);
                cnt = 3.0;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * IL_IN * IL_T; idx += 256) {
        const int ch = idx / (IL_IN * IL_T), rem = idx - ch * (IL_IN * IL_T);
        const int r = rem / IL_T, c = rem - r * IL_T;
        float rx = 0.f, ry = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
#pragma unroll
        for (int b = 0; b < IL_WIN; ++b) {
            const float x = sx[ch][r][c + b], y = sy[ch][r][c + b], w = win.w[b];
            rx += w * x; ry += w * y; rxx += w * (x * x); ryy += w * (y * y); rxy += w * (x * y);
        }
        hb[0][ch][r][c] = rx; hb[1][ch][r][c] = ry; hb[2][ch][r][c] = rxx; hb[3][ch][r][c] = ryy;
        hb[4][ch][r][c] = rxy;
    }
    __syncthreads();
    double ss = 0.0;
    const int i = y0 + ty, j = x0 + tx;
    if (i < Ho && j < Wo) {
        const size_t plane = (size_t)Ho * Wo, o = (size_t)i * Wo + j;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < IL_WIN; ++a) {
                const float w = win.w[a];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += w * hb[q][ch][ty + a][tx];
            }
            float gb, gyy, gxy;
            ss += (double)il_ssim_point(m, gb, gyy, gxy);
            maps[(3 * ch + 0) * plane + o] = gb;
            maps[(3 * ch + 1) * plane + o] = gyy;
            maps[(3 * ch + 2) * plane + o] = gxy;
        }
    }
    // workgroup sums, fixed order: lanes of a wave by shuffle, the four waves through LDS
    double v[3] = {ss, l1, cnt};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) red[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
        for (int k = 0; k < 3; ++k) partials[3 * blk + k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    }
}

// one workgroup: the partials in index order -> header sums and the three scalars
__global__ __launch_bounds__(256) void image_loss_finish_kernel(int nblocks, int Ho, int Wo, float ssim_lambda,
                                                                const double *__restrict__ partials,
                                                                double *__restrict__ header,
                                                                float *__restrict__ out3) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double v[3] = {0.0, 0.0, 0.0};
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] += partials[3 * (size_t)b + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) red[k][tid] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        header[0] = red[0][0];
        header[1] = red[1][0];
        header[2] = red[2][0];
        const float Ll1 = (float)(red[1][0] / red[2][0]);
        const float ssim = (float)(red[0][0] / (3.0 * Ho * Wo));
        out3[0] = (1 - ssim_lambda) * Ll1 + ssim_lambda * (1 - ssim);
        out3[1] = Ll1;
        out3[2] = ssim;
    }
}

__global__ __launch_bounds__(256) void image_loss_bwd_kernel(int H, int W, const float *__restrict__ rgb, int rs,
                                                             const float *__restrict__ gt,
                                                             const uint8_t *__restrict__ valid, IlWindow win,
                                                             float ssim_lambda, const float *__restrict__ v_main,
                                                             const double *__restrict__ header,
                                                             const float *__restrict__ maps,
                                                             float *__restrict__ v_rgb) {
    // one workgroup per (tile, colour channel) — blockIdx.z: its three maps are 13.7 KB of LDS instead of 41 KB for all
    // nine, so eight workgroups fit a CU instead of three (r03: occupancy 33 %, VALUBusy 41 % with the nine together)
    __shared__ float sg[3][IL_IN][IL_IN + 1];          // the channel's maps around the tile (zero outside their domain)
    __shared__ float hb[3][IL_IN][IL_T + 1];
    const int Ho = H - IL_WIN + 1, Wo = W - IL_WIN + 1;
    const size_t plane = (size_t)Ho * Wo;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * IL_T, y0 = blockIdx.y * IL_T, ch = blockIdx.z;
    // pixel (y, x) gathers the outputs (y - a, x - b), a, b in 0..10: window origin (y0 - 10, x0 - 10)
    {   // all loads of a thread's three window points in flight before the first is used (r03: one point at a time
        // was three L2 round trips per workgroup)
        constexpr int NPT = (IL_IN * IL_IN + 255) / 256;
        float tv[NPT][3];
        int rr[NPT], cc[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int idx = tid + 256 * k;
            const int r = idx / IL_IN, c = idx - r * IL_IN;
            rr[k] = r; cc[k] = c;
            const int i = y0 - (IL_WIN - 1) + r, j = x0 - (IL_WIN - 1) + c;
            const bool in = idx < IL_IN * IL_IN && i >= 0 && i < Ho && j >= 0 && j < Wo;
            const size_t o = in ? (size_t)i * Wo + j : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) tv[k][q] = in ? maps[(3 * ch + q) * plane + o] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (tid + 256 * k < IL_IN * IL_IN) {
#pragma unroll
                for (int q = 0; q < 3; ++q) sg[q][rr[k]][cc[k]] = tv[k][q];
            }
    }
    __syncthreads();
    // rows: h[r][tx] = sum_b w[b] g[r][x - b]  (window column of x - b: tx + 10 - b)
    for (int idx = tid; idx < 3 * IL_IN * IL_T; idx += 256) {
        const int q = idx / (IL_IN * IL_T), rem = idx - q * (IL_IN * IL_T);
        const int r = rem / IL_T, c = rem - r * IL_T;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < IL_WIN; ++b) {
            const int j = x0 + c - b;                       // outputs outside their domain are skipped by the
            if (j >= 0 && j < Wo) acc += win.w[b] * sg[q][r][c + (IL_WIN - 1) - b];   // oracle, not added as zeros
        }
        hb[q][r][c] = acc;
    }
    __syncthreads();
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) return;
    const size_t p = (size_t)y * W + x;
    const bool ok = !valid || valid[p];
    const float vm = v_main[0];
    const float ks = -ssim_lambda * vm / (float)(3.0 * Ho * Wo);
    const float kl = (1 - ssim_lambda) * vm / (float)header[2];
    if (!ok) { v_rgb[3 * p + ch] = 0.0f; return; }
    float G[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < IL_WIN; ++a) {
        const int i = y - a;
        if (i < 0 || i >= Ho) continue;
        const float w = win.w[a];
#pragma unroll
        for (int k = 0; k < 3; ++k) G[k] += w * hb[k][ty + (IL_WIN - 1) - a][tx];
    }
    const float Y = rgb[(size_t)rs * p + ch], X = gt[3 * p + ch];
    const float d = Y - X;
    const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
    v_rgb[3 * p + ch] = ks * (G[0] + 2 * Y * G[1] + X * G[2]) + kl * sgn;
}

static IlWindow il_window() {
    IlWindow win;
    float sum = 0.f;
    for (int k = 0; k < IL_WIN; ++k) {
        const float c = (float)(k - IL_WIN / 2);
        win.w[k] = (float)exp(-(double)(c * c) / (2.0 * 1.5 * 1.5));
        sum += win.w[k];
    }
    for (int k = 0; k < IL_WIN; ++k) win.w[k] /= sum;
    return win;
}
static size_t il_blocks(int H, int W) {
    return (size_t)((H + IL_T - 1) / IL_T) * (size_t)((W + IL_T - 1) / IL_T);
}

extern "C" size_t gg_image_loss_workspace(int img_height, int img_width) {
    if (img_height < IL_WIN || img_width < IL_WIN) return 0;
    const size_t Ho = (size_t)(img_height - IL_WIN + 1), Wo = (size_t)(img_width - IL_WIN + 1);
    return gg_align_up(sizeof(double) * (IL_HEADER + 3 * il_blocks(img_height, img_width)), 256) +
           gg_align_up(sizeof(float) * 9 * Ho * Wo, 256);
}

extern "C" int gg_image_loss_fwd(int H, int W, const float *rgb, int rgb_pixel_stride, const float *gt,
                                 const uint8_t *valid, float ssim_lambda, float *out3, void *ws, size_t ws_bytes,
                                 gg_stream_t stream) {
    GG_REQUIRE(H >= IL_WIN && W >= IL_WIN, "image smaller than the 11 x 11 SSIM window");
    GG_REQUIRE(rgb_pixel_stride >= 3, "rgb pixels hold 3 values");
    GG_REQUIRE(rgb && gt && out3, "null pointer");
    if (ws == nullptr || ws_bytes < gg_image_loss_workspace(H, W)) {
        gg_set_error("gg_image_loss_fwd: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    double *header = (double *)ws;
    double *partials = header + IL_HEADER;
    const size_t nblk = il_blocks(H, W);
    float *maps = (float *)((char *)ws + gg_align_up(sizeof(double) * (IL_HEADER + 3 * nblk), 256));
    const dim3 grid((W + IL_T - 1) / IL_T, (H + IL_T - 1) / IL_T);
    hipLaunchKernelGGL(image_loss_fwd_kernel, grid, dim3(256), 0, s, H, W, rgb, rgb_pixel_stride, gt, valid,
                       il_window(), partials, maps);
    hipLaunchKernelGGL(image_loss_finish_kernel, dim3(1), dim3(256), 0, s, (int)nblk, H - IL_WIN + 1,
                       W - IL_WIN + 1, ssim_lambda, partials, header, out3);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_image_loss_bwd(int H, int W, const float *rgb, int rgb_pixel_stride, const float *gt,
                                 const uint8_t *valid, float ssim_lambda, const float *v_main, const void *ws,
                                 size_t ws_bytes, float *v_rgb, gg_stream_t stream) {
    GG_REQUIRE(H >= IL_WIN && W >= IL_WIN, "image smaller than the 11 x 11 SSIM window");
    GG_REQUIRE(rgb_pixel_stride >= 3, "rgb pixels hold 3 values");
    GG_REQUIRE(rgb && gt && v_main && v_rgb, "null pointer");
    if (ws == nullptr || ws_bytes < gg_image_loss_workspace(H, W)) {
        gg_set_error("gg_image_loss_bwd: workspace too small (it must be the forward's)");
        return GG_ERR_WORKSPACE;
    }
    const double *header = (const double *)ws;
    const size_t nblk = il_blocks(H, W);
    const float *maps = (const float *)((const char *)ws + gg_align_up(sizeof(double) * (IL_HEADER + 3 * nblk), 256));
    const dim3 grid((W + IL_T - 1) / IL_T, (H + IL_T - 1) / IL_T, 3);
    hipLaunchKernelGGL(image_loss_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, rgb, rgb_pixel_stride,
                       gt, valid, il_window(), ssim_lambda, v_main, header, maps, v_rgb);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// =============================================================================================
// depth and normal losses of get_loss_dict (gaussian_splatting.py:879-880) over the masked pixels:
//     normal_loss = 0.5 mse(normal, gt_normal) + 0.5 cosine_similarity_loss(normal, gt_normal);  depth_loss = L1
// The reference gathers the masked pixels with boolean indexing (a host round trip for the count and five gathered
// copies of up to 1.9 M pixels each way), then runs ~15 elementwise / reduction launches.  One streaming pass each
// way here: 37 B/pixel read forward, + 16 B/pixel written backward; per-workgroup partial sums in double, summed
// in index order (reproducible).  The arithmetic follows oracle/gg_oracle.c:geom_loss_* (gradients bit-identical).
// =============================================================================================
struct GeomLossArgs {
    const float *depth, *gt_depth, *normal, *gt_normal;
    const uint8_t *mask;
    int d_stride, gd_stride, n_pstride, n_cstride, g_pstride, g_cstride;
};
#define GL_EPS 1e-12f

__global__ __launch_bounds__(256) void geom_loss_fwd_kernel(long P, GeomLossArgs a, double *__restrict__ partials) {
    __shared__ double red[4][4];
    double v[4] = {0.0, 0.0, 0.0, 0.0};   // l1, mse, cosine sum, count
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += stride) {
        if (a.mask && !a.mask[p]) continue;
        v[0] += (double)fabsf(a.depth[p * a.d_stride] - a.gt_depth[p * a.gd_stride]);
        float dot = 0.f, sa = 0.f, sb = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float u = a.normal[p * a.n_pstride + c * a.n_cstride];
            const float g = a.gt_normal[p * a.g_pstride + c * a.g_cstride];
            const float d = u - g;
            v[1] += (double)(d * d);
            dot = __builtin_fmaf(u, g, dot);
            sa = __builtin_fmaf(u, u, sa);
            sb = __builtin_fmaf(g, g, sb);
        }
        const float n1 = sqrtf(sa), n2 = sqrtf(sb);
        v[2] += (double)(dot / (fmaxf(n1, GL_EPS) * fmaxf(n2, GL_EPS)));
        v[3] += 1.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            partials[4 * (size_t)blockIdx.x + k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    }
}
__global__ __launch_bounds__(256) void geom_loss_finish_kernel(int nblocks, const double *__restrict__ partials,
                                                               double *__restrict__ header,
                                                               float *__restrict__ out3) {
    __shared__ double red[4][256];
    const int tid = threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += partials[4 * (size_t)b + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][tid] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = red[3][0];
        header[0] = cnt;
        out3[0] = (float)(red[0][0] / cnt);
        out3[1] = 0.5f * (float)(red[1][0] / (3.0 * cnt)) + 0.5f * (1.0f - (float)(red[2][0] / cnt));
        out3[2] = (float)cnt;
    }
}
__global__ __launch_bounds__(256) void geom_loss_bwd_kernel(long P, GeomLossArgs a, const double *__restrict__ header,
                                                            const float *__restrict__ v_depth_loss,
                                                            const float *__restrict__ v_normal_loss,
                                                            float *__restrict__ v_depth, float *__restrict__ v_normal) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (a.mask && !a.mask[p]) {
        v_depth[p] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) v_normal[3 * p + c] = 0.0f;
        return;
    }
    const float M = (float)header[0];
    const float vdl = v_depth_loss[0], vnl = v_normal_loss[0];
    const float kd = vdl / M;
    const float km = 0.5f * vnl * 2 / (3 * M);
    const float kc = -(0.5f * vnl) / M;
    const float d = a.depth[p * a.d_stride] - a.gt_depth[p * a.gd_stride];
    v_depth[p] = kd * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f));
    float dot = 0.f, sa = 0.f, sb = 0.f, u3[3], g3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        u3[c] = a.normal[p * a.n_pstride + c * a.n_cstride];
        g3[c] = a.gt_normal[p * a.g_pstride + c * a.g_cstride];
        dot = __builtin_fmaf(u3[c], g3[c], dot);
        sa = __builtin_fmaf(u3[c], u3[c], sa);
        sb = __builtin_fmaf(g3[c], g3[c], sb);
    }
    const float n1 = sqrtf(sa), n2 = sqrtf(sb);
    const float ca = fmaxf(n1, GL_EPS), cb = fmaxf(n2, GL_EPS);
    const float sim = dot / (ca * cb);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ua = u3[c] / ca, vb = g3[c] / cb;
        const float dsim = (vb - (n1 > GL_EPS ? sim * ua : 0.0f)) / ca;
        v_normal[3 * p + c] = km * (u3[c] - g3[c]) + kc * dsim;
    }
}

#define GL_BLOCKS 2048
extern "C" size_t gg_geom_loss_workspace(void) { return sizeof(double) * (IL_HEADER + 4 * GL_BLOCKS); }

static GeomLossArgs geom_args(const float *depth, int d_stride, const float *gt_depth, int gd_stride,
                              const float *normal, int n_pstride, int n_cstride, const float *gt_normal, int g_pstride,
                              int g_cstride, const uint8_t *mask) {
    GeomLossArgs a;
    a.depth = depth; a.gt_depth = gt_depth; a.normal = normal; a.gt_normal = gt_normal; a.mask = mask;
    a.d_stride = d_stride; a.gd_stride = gd_stride; a.n_pstride = n_pstride; a.n_cstride = n_cstride;
    a.g_pstride = g_pstride; a.g_cstride = g_cstride;
    return a;
}

extern "C" int gg_geom_loss_fwd(int64_t num_pixels, const float *depth, int depth_stride, const float *gt_depth,
                                int gt_depth_stride, const float *normal, int normal_pixel_stride,
                                int normal_channel_stride, const float *gt_normal, int gt_normal_pixel_stride,
                                int gt_normal_channel_stride, const uint8_t *mask, float *out3, void *ws,
                                size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(num_pixels >= 1, "num_pixels < 1");
    GG_REQUIRE(depth && gt_depth && normal && gt_normal && out3, "null pointer");
    if (ws == nullptr || ws_bytes < gg_geom_loss_workspace()) {
        gg_set_error("gg_geom_loss_fwd: workspace too small");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    double *header = (double *)ws, *partials = header + IL_HEADER;
    const int nb = (int)((num_pixels + 255) / 256 < GL_BLOCKS ? (num_pixels + 255) / 256 : GL_BLOCKS);
    const GeomLossArgs a = geom_args(depth, depth_stride, gt_depth, gt_depth_stride, normal, normal_pixel_stride,
                                     normal_channel_stride, gt_normal, gt_normal_pixel_stride,
                                     gt_normal_channel_stride, mask);
    hipLaunchKernelGGL(geom_loss_fwd_kernel, dim3(nb), dim3(256), 0, s, (long)num_pixels, a, partials);
    hipLaunchKernelGGL(geom_loss_finish_kernel, dim3(1), dim3(256), 0, s, nb, partials, header, out3);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_geom_loss_bwd(int64_t num_pixels, const float *depth, int depth_stride, const float *gt_depth,
                                int gt_depth_stride, const float *normal, int normal_pixel_stride,
                                int normal_channel_stride, const float *gt_normal, int gt_normal_pixel_stride,
                                int gt_normal_channel_stride, const uint8_t *mask, const float *v_depth_loss,
                                const float *v_normal_loss, const void *ws, size_t ws_bytes, float *v_depth,
                                float *v_normal, gg_stream_t stream) {
    GG_REQUIRE(num_pixels >= 1, "num_pixels < 1");
    GG_REQUIRE(depth && gt_depth && normal && gt_normal && v_depth_loss && v_normal_loss && v_depth && v_normal,
               "null pointer");
    if (ws == nullptr || ws_bytes < gg_geom_loss_workspace()) {
        gg_set_error("gg_geom_loss_bwd: workspace too small (it must be the forward's)");
        return GG_ERR_WORKSPACE;
    }
    const GeomLossArgs a = geom_args(depth, depth_stride, gt_depth, gt_depth_stride, normal, normal_pixel_stride,
                                     normal_channel_stride, gt_normal, gt_normal_pixel_stride,
                                     gt_normal_channel_stride, mask);
    hipLaunchKernelGGL(geom_loss_bwd_kernel, dim3((unsigned)((num_pixels + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (long)num_pixels, a, (const double *)ws, v_depth_loss, v_normal_loss,
                       v_depth, v_normal);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
