// project.hip — per-Gaussian projection, frustum cull, EWA covariance, conic, radius, tile
// count (forward) and its VJP (backward), plus the SH colour kernels.  O(N) streaming kernels:
// one lane per Gaussian, HBM-bound.
//
// The forward follows oracle/gg_oracle.c:project_fwd operation for operation (fixed association,
// no contraction: the file is compiled with -ffp-contract=off, div/sqrt are IEEE-correct), so
// radii, num_tiles_hit, depth bits, xys and conics are bit-identical to the oracle and with
// them the sort keys and the per-tile lists.
#include <stdarg.h>
#include <stdio.h>

#include "blend_common.h"

// ---- error plumbing (one thread-local message, SURVEY §8b "gg_last_error") ------------------
static thread_local char g_err[512] = "";
void gg_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gg_last_error(void) { return g_err; }
extern "C" int gg_abi_version(void) { return 4; }

__device__ __forceinline__ void quat_to_R(const float4 q, float *R, float *qn, float &inv) {
    float nn = ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w;  // (w,x,y,z) stored in .x.y.z.w
    inv = 1.0f / sqrtf(nn);
    float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * (x * y - w * z);
    R[2] = 2.0f * (x * z + w * y);
    R[3] = 2.0f * (x * y + w * z);
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = 2.0f * (y * z - w * x);
    R[6] = 2.0f * (x * z - w * y);
    R[7] = 2.0f * (y * z + w * x);
    R[8] = 1.0f - 2.0f * (x * x + y * y);
    qn[0] = w;
    qn[1] = x;
    qn[2] = y;
    qn[3] = z;
}

// projection of one Gaussian (mean p, normalised-or-not quaternion q, scales s already times glob_scale): shared by
// project_fwd_kernel and view_fwd_kernel (one operation sequence, one result)
__device__ __forceinline__ void project_fwd_point(
    const float px, const float py, const float pz, const float4 q, const float s0, const float s1, const float s2,
    const float (&V)[12], const float (&P)[16], const float fx, const float fy, const float cx, const float cy,
    const int img_h, const int img_w, const int tiles_x, const int tiles_y, const float clip_thresh, float (&o_c3)[6],
    float (&o_con)[3], float &o_x, float &o_y, float &o_d, int &o_r, int &o_n) {
    float tx = ((V[0] * px + V[1] * py) + V[2] * pz) + V[3];
    float ty = ((V[4] * px + V[5] * py) + V[6] * pz) + V[7];
    float tz = ((V[8] * px + V[9] * py) + V[10] * pz) + V[11];
    if (tz > clip_thresh) {
        float R[9], qn[4], inv;
        quat_to_R(q, R, qn, inv);
        float M[9] = {R[0] * s0, R[1] * s1, R[2] * s2, R[3] * s0, R[4] * s1,
                      R[5] * s2, R[6] * s0, R[7] * s1, R[8] * s2};
        float c3[6];
        c3[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
        c3[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
        c3[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
        c3[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
        c3[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
        c3[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
#pragma unroll
        for (int k = 0; k < 6; ++k) o_c3[k] = c3[k];

        float tan_fovx = (0.5f * (float)img_w) / fx;
        float tan_fovy = (0.5f * (float)img_h) / fy;
        float lim_x = GG_FOV_LIM * tan_fovx, lim_y = GG_FOV_LIM * tan_fovy;
        float txc = tz * fminf(lim_x, fmaxf(-lim_x, tx / tz));
        float tyc = tz * fminf(lim_y, fmaxf(-lim_y, ty / tz));
        float rz = 1.0f / tz;
        float rz2 = rz * rz;
        float J00 = fx * rz, J02 = (-(fx * txc)) * rz2;
        float J11 = fy * rz, J12 = (-(fy * tyc)) * rz2;
        float T00 = J00 * V[0] + J02 * V[8], T01 = J00 * V[1] + J02 * V[9],
              T02 = J00 * V[2] + J02 * V[10];
        float T10 = J11 * V[4] + J12 * V[8], T11 = J11 * V[5] + J12 * V[9],
              T12 = J11 * V[6] + J12 * V[10];
        float A00 = (T00 * c3[0] + T01 * c3[1]) + T02 * c3[2];
        float A01 = (T00 * c3[1] + T01 * c3[3]) + T02 * c3[4];
        float A02 = (T00 * c3[2] + T01 * c3[4]) + T02 * c3[5];
        float A10 = (T10 * c3[0] + T11 * c3[1]) + T12 * c3[2];
        float A11 = (T10 * c3[1] + T11 * c3[3]) + T12 * c3[4];
        float A12 = (T10 * c3[2] + T11 * c3[4]) + T12 * c3[5];
        float a = ((A00 * T00 + A01 * T01) + A02 * T02) + GG_BLUR;
        float b = (A00 * T10 + A01 * T11) + A02 * T12;
        float c = ((A10 * T10 + A11 * T11) + A12 * T12) + GG_BLUR;

        float det = a * c - b * b;
        if (det != 0.0f) {
            float inv_det = 1.0f / det;
            o_con[0] = c * inv_det;
            o_con[1] = (-b) * inv_det;
            o_con[2] = a * inv_det;
            float bm = 0.5f * (a + c);
            float sq = sqrtf(fmaxf(GG_EIG_FLOOR, bm * bm - det));
            float v1 = bm + sq, v2 = bm - sq;
            float radius = ceilf(GG_RADIUS_SIGMA * sqrtf(fmaxf(v1, v2)));

            float hx = ((P[0] * px + P[1] * py) + P[2] * pz) + P[3];
            float hy = ((P[4] * px + P[5] * py) + P[6] * pz) + P[7];
            float hw = ((P[12] * px + P[13] * py) + P[14] * pz) + P[15];
            float rw = 1.0f / (hw + GG_W_EPS);
            float ndx = hx * rw, ndy = hy * rw;
            float ux = ((0.5f * (float)img_w) * ndx + cx) - GG_PIX_OFFSET;
            float uy = ((0.5f * (float)img_h) * ndy + cy) - GG_PIX_OFFSET;
            int x0, y0, x1, y1;
            gg_tile_bbox(ux, uy, radius, tiles_x, tiles_y, x0, y0, x1, y1);
            int area = (x1 - x0) * (y1 - y0);
            if (area > 0) {
                o_n = area;
                o_d = tz;
                o_r = (int)radius;
                o_x = ux;
                o_y = uy;
            }
        }
    }
}
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ scales, float glob_scale,
    const float *__restrict__ quats, const float *__restrict__ viewmat,
    const float *__restrict__ projmat, float fx, float fy, float cx, float cy, int img_h,
    int img_w, int tiles_x, int tiles_y, float clip_thresh, float *__restrict__ cov3d,
    float *__restrict__ xys, float *__restrict__ depths, int32_t *__restrict__ radii,
    float *__restrict__ conics, int32_t *__restrict__ num_tiles_hit,
    unsigned *__restrict__ count_ws = nullptr) {
    // count_ws != nullptr (gg_project_fwd_count): every workgroup leaves the sum of its num_tiles_hit in count_ws[block]
    // and a one-workgroup kernel adds them up (r04: instead of count_kernel's pass over num_tiles_hit + an 8-byte fill,
    // 18 us per view.  A ticket with the LAST workgroup adding up was tried first: 3 907 returning atomics on one word
    // made the projection 0.20 ms instead of 0.02)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N && count_ws == nullptr) return;
    const bool live = i < N;
    if (!live) i = N - 1;          // (a padding thread of the last workgroup: computes, stores nothing, counts nothing)
    float V[12], P[16];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = viewmat[k];  // uniform -> scalar loads
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = projmat[k];

    float o_c3[6] = {0, 0, 0, 0, 0, 0};
    float o_con[3] = {0, 0, 0};
    float o_x = 0, o_y = 0, o_d = 0;
    int o_r = 0, o_n = 0;

    const float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
    project_fwd_point(px, py, pz, reinterpret_cast<const float4 *>(quats)[i], glob_scale * scales[3 * i],
                      glob_scale * scales[3 * i + 1], glob_scale * scales[3 * i + 2], V, P, fx, fy, cx, cy, img_h, img_w,
                      tiles_x, tiles_y, clip_thresh, o_c3, o_con, o_x, o_y, o_d, o_r, o_n);
    if (live) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov3d[6 * (size_t)i + k] = o_c3[k];
        xys[2 * (size_t)i] = o_x;
        xys[2 * (size_t)i + 1] = o_y;
        depths[i] = o_d;
        radii[i] = o_r;
#pragma unroll
        for (int k = 0; k < 3; ++k) conics[3 * (size_t)i + k] = o_con[k];
        num_tiles_hit[i] = o_n;
    }
    if (count_ws == nullptr) return;
    __shared__ unsigned s_part[4];
    unsigned mine = live ? (unsigned)o_n : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) count_ws[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}
// one workgroup: the partial sums of project_fwd_kernel's workgroups -> *count_out
__global__ __launch_bounds__(1024) void count_finish_kernel(int nparts, const unsigned *__restrict__ parts,
                                                            unsigned long long *__restrict__ count_out) {
    __shared__ unsigned long long s_tot[16];
    unsigned long long acc = 0;
    for (int b = threadIdx.x; b < nparts; b += 1024) acc += parts[b];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_tot[w];
        *count_out = t;
    }
}

// the backward of one visible Gaussian: cotangents (v_xy, v_depth, v_conic) -> vm += d/d mean, vs = d/d scale,
// vq4 = d/d quat.  Shared by project_bwd_kernel and view_bwd_kernel (one operation sequence, one result)
__device__ __forceinline__ void project_bwd_point(
    const int i, const float *__restrict__ means, const float *__restrict__ scales, const float glob_scale,
    const float *__restrict__ quats, const float *__restrict__ viewmat, const float *__restrict__ projmat,
    const float fx, const float fy, const int img_h, const int img_w, const float *__restrict__ conics,
    const float vxy0, const float vxy1, const float vdepth, const float ga, const float gb, const float gc,
    float (&vm)[3], float (&vs)[3], float (&vq4)[4]) {
    {
        float V[12], P[16];
#pragma unroll
        for (int k = 0; k < 12; ++k) V[k] = viewmat[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) P[k] = projmat[k];
        float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];

        // (1) pixel centre, exact VJP through the homogeneous divide
        float hx = ((P[0] * px + P[1] * py) + P[2] * pz) + P[3];
        float hy = ((P[4] * px + P[5] * py) + P[6] * pz) + P[7];
        float hw = ((P[12] * px + P[13] * py) + P[14] * pz) + P[15];
        float rw = 1.0f / (hw + GG_W_EPS);
        float vnx = (0.5f * (float)img_w) * vxy0;
        float vny = (0.5f * (float)img_h) * vxy1;
        float vhx = vnx * rw, vhy = vny * rw;
#if GG_VJP_GSPLAT_COMPAT
        float vhw = 0.0f * hx * hy;   // compat: the homogeneous-w path is dropped
#else
        float vhw = -((vnx * hx + vny * hy) * (rw * rw));
#endif
#pragma unroll
        for (int j = 0; j < 3; ++j) vm[j] += (P[j] * vhx + P[4 + j] * vhy) + P[12 + j] * vhw;
        // (2) depth
        float vz = vdepth;
#pragma unroll
        for (int j = 0; j < 3; ++j) vm[j] += V[8 + j] * vz;
        // (3) conic -> cov2d
        float ca = conics[3 * i], cb = conics[3 * i + 1], cc = conics[3 * i + 2];
        float xg00 = ca * ga + cb * gb, xg01 = ca * gb + cb * gc;
        float xg10 = cb * ga + cc * gb, xg11 = cb * gb + cc * gc;
        float s00 = -(xg00 * ca + xg01 * cb), s01 = -(xg00 * cb + xg01 * cc);
        float s10 = -(xg10 * ca + xg11 * cb), s11 = -(xg10 * cb + xg11 * cc);
        float v_a = s00, v_b = s01 + s10, v_c = s11;

        float tx = ((V[0] * px + V[1] * py) + V[2] * pz) + V[3];
        float ty = ((V[4] * px + V[5] * py) + V[6] * pz) + V[7];
        float tz = ((V[8] * px + V[9] * py) + V[10] * pz) + V[11];
        float R[9], qn[4], inv_norm;
        float4 q = reinterpret_cast<const float4 *>(quats)[i];
        quat_to_R(q, R, qn, inv_norm);
        float s[3] = {glob_scale * scales[3 * i], glob_scale * scales[3 * i + 1],
                      glob_scale * scales[3 * i + 2]};
        float M[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) M[3 * r + k] = R[3 * r + k] * s[k];
        float C3[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                C3[3 * r + k] = (M[3 * r] * M[3 * k] + M[3 * r + 1] * M[3 * k + 1]) +
                                M[3 * r + 2] * M[3 * k + 2];
        float tan_fovx = (0.5f * (float)img_w) / fx, tan_fovy = (0.5f * (float)img_h) / fy;
        float lim_x = GG_FOV_LIM * tan_fovx, lim_y = GG_FOV_LIM * tan_fovy;
        float rx = tx / tz, ry = ty / tz;
#if GG_VJP_GSPLAT_COMPAT
        // compat: the Jacobian of the backward is rebuilt at the unclamped point, no clamp derivative
        float sgx = 0.0f * rx * lim_x, sgy = 0.0f * ry * lim_y;
        float txc = tx, tyc = ty;
#else
        float sgx = (rx > lim_x) ? 1.0f : ((rx < -lim_x) ? -1.0f : 0.0f);
        float sgy = (ry > lim_y) ? 1.0f : ((ry < -lim_y) ? -1.0f : 0.0f);
        float txc = tz * fminf(lim_x, fmaxf(-lim_x, rx));
        float tyc = tz * fminf(lim_y, fmaxf(-lim_y, ry));
#endif
        float rz = 1.0f / tz, rz2 = rz * rz, rz3 = rz2 * rz;
        float J00 = fx * rz, J02 = (-(fx * txc)) * rz2, J11 = fy * rz, J12 = (-(fy * tyc)) * rz2;
        float Tm[6] = {J00 * V[0] + J02 * V[8], J00 * V[1] + J02 * V[9], J00 * V[2] + J02 * V[10],
                       J11 * V[4] + J12 * V[8], J11 * V[5] + J12 * V[9], J11 * V[6] + J12 * V[10]};
        // (4)
        float g00 = v_a, g01 = 0.5f * v_b, g11 = v_c;
        float GT[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            GT[k] = g00 * Tm[k] + g01 * Tm[3 + k];
            GT[3 + k] = g01 * Tm[k] + g11 * Tm[3 + k];
        }
        float G3[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) G3[3 * r + k] = Tm[r] * GT[k] + Tm[3 + r] * GT[3 + k];
        float vT[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                vT[3 * r + k] = 2.0f * ((GT[3 * r] * C3[k] + GT[3 * r + 1] * C3[3 + k]) +
                                        GT[3 * r + 2] * C3[6 + k]);
        // (5)
        float vJ00 = (vT[0] * V[0] + vT[1] * V[1]) + vT[2] * V[2];
        float vJ02 = (vT[0] * V[8] + vT[1] * V[9]) + vT[2] * V[10];
        float vJ11 = (vT[3] * V[4] + vT[4] * V[5]) + vT[5] * V[6];
        float vJ12 = (vT[3] * V[8] + vT[4] * V[9]) + vT[5] * V[10];
        float v_txc = (-(fx * rz2)) * vJ02;
        float v_tyc = (-(fy * rz2)) * vJ12;
        float v_tz = ((-(fx * rz2)) * vJ00 - (fy * rz2) * vJ11) +
                     (2.0f * fx * txc * rz3) * vJ02 + (2.0f * fy * tyc * rz3) * vJ12;
        float v_tx = (sgx != 0.0f) ? 0.0f : v_txc;
        float v_ty = (sgy != 0.0f) ? 0.0f : v_tyc;
        v_tz += (sgx * lim_x) * v_txc + (sgy * lim_y) * v_tyc;
#pragma unroll
        for (int j = 0; j < 3; ++j) vm[j] += (V[j] * v_tx + V[4 + j] * v_ty) + V[8 + j] * v_tz;
        // (6)
        float vM[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                vM[3 * r + k] = 2.0f * ((G3[3 * r] * M[k] + G3[3 * r + 1] * M[3 + k]) +
                                        G3[3 * r + 2] * M[6 + k]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            vs[k] = glob_scale * ((R[k] * vM[k] + R[3 + k] * vM[3 + k]) + R[6 + k] * vM[6 + k]);
        float G[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) G[3 * r + k] = vM[3 * r + k] * s[k];
        float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
        float vq[4];
        vq[0] = 2.0f * ((x * (G[7] - G[5]) + y * (G[2] - G[6])) + z * (G[3] - G[1]));
        vq[1] = 2.0f * (((-2.0f * x) * (G[4] + G[8]) + y * (G[1] + G[3])) +
                        (z * (G[2] + G[6]) + w * (G[7] - G[5])));
        vq[2] = 2.0f * ((x * (G[1] + G[3]) + (-2.0f * y) * (G[0] + G[8])) +
                        (z * (G[5] + G[7]) + w * (G[2] - G[6])));
        vq[3] = 2.0f * ((x * (G[2] + G[6]) + y * (G[5] + G[7])) +
                        ((-2.0f * z) * (G[0] + G[4]) + w * (G[3] - G[1])));
        float dotp = ((qn[0] * vq[0] + qn[1] * vq[1]) + qn[2] * vq[2]) + qn[3] * vq[3];
#pragma unroll
#if GG_VJP_GSPLAT_COMPAT
        for (int k = 0; k < 4; ++k) vq4[k] = vq[k] + 0.0f * dotp * inv_norm;   // w.r.t. the normalised q
#else
        for (int k = 0; k < 4; ++k) vq4[k] = (vq[k] - qn[k] * dotp) * inv_norm;
#endif
    }
}
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ scales, float glob_scale,
    const float *__restrict__ quats, const float *__restrict__ viewmat,
    const float *__restrict__ projmat, float fx, float fy, int img_h, int img_w,
    const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ v_xy, const float *__restrict__ v_depth,
    const float *__restrict__ v_conic, float *__restrict__ v_mean3d, float *__restrict__ v_scale,
    float *__restrict__ v_quat, int xy_stride = 2, int conic_stride = 3, int acc_means = 0) {
    // xy_stride / conic_stride: floats between the rows of v_xy / v_conic (gg_project_bwd_ex: the blend backward's
    // interleaved gradient record is read in place); acc_means: v_mean3d += (a registered gradient buffer)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float vm[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, vq4[4] = {0, 0, 0, 0};
    if (radii[i] > 0) {
        const float *vcn = v_conic + (size_t)conic_stride * i;
        project_bwd_point(i, means, scales, glob_scale, quats, viewmat, projmat, fx, fy, img_h, img_w, conics,
                          v_xy[(size_t)xy_stride * i], v_xy[(size_t)xy_stride * i + 1], v_depth[i], vcn[0], vcn[1], vcn[2],
                          vm, vs, vq4);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v_mean3d[3 * (size_t)i + k] = acc_means ? v_mean3d[3 * (size_t)i + k] + vm[k] : vm[k];
        v_scale[3 * (size_t)i + k] = vs[k];
    }
    reinterpret_cast<float4 *>(v_quat)[i] = make_float4(vq4[0], vq4[1], vq4[2], vq4[3]);
}

// ---- spherical harmonics ---------------------------------------------------------------------
__device__ __forceinline__ void sh_basis(int deg, float dx, float dy, float dz, float *Y) {
    Y[0] = GG_SH_C0;
    if (deg < 1) return;
    float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
    float x = dx / norm, y = dy / norm, z = dz / norm;
    Y[1] = GG_SH_C1 * (-y);
    Y[2] = GG_SH_C1 * z;
    Y[3] = GG_SH_C1 * (-x);
    if (deg < 2) return;
    float xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
    Y[4] = GG_SH_C2_0 * xy;
    Y[5] = GG_SH_C2_1 * yz;
    Y[6] = GG_SH_C2_2 * ((2.0f * zz - xx) - yy);
    Y[7] = GG_SH_C2_3 * xz;
    Y[8] = GG_SH_C2_4 * (xx - yy);
    if (deg < 3) return;
    Y[9] = (GG_SH_C3_0 * y) * (3.0f * xx - yy);
    Y[10] = (GG_SH_C3_1 * xy) * z;
    Y[11] = (GG_SH_C3_2 * y) * ((4.0f * zz - xx) - yy);
    Y[12] = (GG_SH_C3_3 * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
    Y[13] = (GG_SH_C3_4 * x) * ((4.0f * zz - xx) - yy);
    Y[14] = (GG_SH_C3_5 * z) * (xx - yy);
    Y[15] = (GG_SH_C3_6 * x) * (xx - 3.0f * yy);
    if (deg < 4) return;
    Y[16] = (GG_SH_C4_0 * xy) * (xx - yy);
    Y[17] = (GG_SH_C4_1 * yz) * (3.0f * xx - yy);
    Y[18] = (GG_SH_C4_2 * xy) * (7.0f * zz - 1.0f);
    Y[19] = (GG_SH_C4_3 * yz) * (7.0f * zz - 3.0f);
    Y[20] = GG_SH_C4_4 * (zz * (35.0f * zz - 30.0f) + 3.0f);
    Y[21] = (GG_SH_C4_5 * xz) * (7.0f * zz - 3.0f);
    Y[22] = (GG_SH_C4_6 * (xx - yy)) * (7.0f * zz - 1.0f);
    Y[23] = (GG_SH_C4_7 * xz) * (xx - 3.0f * yy);
    Y[24] = GG_SH_C4_8 * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}
__host__ __device__ static inline int sh_nbases(int deg) {
    return deg >= 4 ? 25 : (deg + 1) * (deg + 1);
}

// One lane per Gaussian.  The (N,K,3) coefficient rows are 12K bytes apart, so a direct
// per-lane walk would touch 64 cache lines per load; instead each wave copies its 64 rows
// (64*3K contiguous floats) through LDS with fully coalesced dword loads and every lane then
// reads its own row from LDS (row stride 3K words is odd -> bank-conflict free).
// TAIL (gg_shade_tail_fwd): the plugin route's 7-channel array rgb | depth | normal in one pass — colours =
// clamp(sh + 0.5, 0, 1) (reference gaussian_splatting.py:731) written to columns 0..2 of a (N, 7) row, depth to
// column 3, the normal to 4..6, and one byte per Gaussian saying which of the three clamps let the gradient
// through (torch.clamp: min <= x <= max) for the backward.
#ifndef GG_SH_NT
#define GG_SH_NT 1   // the 300 B/Gaussian coefficient stream with non-temporal loads: 78 -> 70 us at 1 M Gaussians (r03)
#endif
template <int K, bool TAIL = false>
__global__ __launch_bounds__(256) void sh_fwd_kernel(int N, int deg,
                                                     const float *__restrict__ viewdirs,
                                                     const float *__restrict__ coeffs,
                                                     float *__restrict__ colors,
                                                     const float *__restrict__ depths = nullptr,
                                                     const float *__restrict__ normals = nullptr,
                                                     uint8_t *__restrict__ mask = nullptr) {
    // SH_ROWS Gaussians per wave: 32 halves the LDS per workgroup (38 KB at K = 25), which doubles
    // the resident waves and with them the loads in flight — the kernel only moves bytes
    // (K = 25, N = 1 M: 0.112 -> 0.092 ms = 3.4 TB/s; 16 rows per wave: no further change)
    constexpr int ROW = 3 * K, SH_ROWS = (K >= 16) ? 32 : 64;
    __shared__ float stage[4][SH_ROWS * ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (blockIdx.x * 4 + wave) * SH_ROWS;  // first Gaussian of this wave
    if (base >= N) return;
    const int nrows = min(SH_ROWS, N - base);
    const float *src = coeffs + (size_t)base * ROW;
    float *st = stage[wave];
    if (nrows == SH_ROWS) {   // (16-byte copies measure the same: 0.093 ms)
#pragma unroll
        for (int e = 0; e < SH_ROWS * ROW; e += 64)
            if (e + lane < SH_ROWS * ROW) st[e + lane] = GG_SH_NT ? __builtin_nontemporal_load(src + e + lane) : src[e + lane];
    } else {
        for (int e = lane; e < nrows * ROW; e += 64) st[e] = src[e];
    }
    __builtin_amdgcn_wave_barrier();
    const int i = base + lane;
    if (lane >= nrows) return;
    float Y[GG_SH_MAX_BASES];
    sh_basis(deg, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
    const int nb = min(sh_nbases(deg), K);
    const float *cf = st + lane * ROW;
    unsigned open_bits = 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = Y[0] * cf[c];
#pragma unroll
        for (int k = 1; k < K; ++k)
            if (k < nb) acc = __builtin_fmaf(Y[k], cf[3 * k + c], acc);
        if (TAIL) {
            const float x = acc + 0.5f;
            if (x >= 0.0f && x <= 1.0f) open_bits |= 1u << c;
            // torch.clamp: NaN stays NaN
            colors[7 * (size_t)i + c] = (x != x) ? x : fminf(fmaxf(x, 0.0f), 1.0f);
        } else {
            colors[3 * (size_t)i + c] = acc;
        }
    }
    if (TAIL) {
        colors[7 * (size_t)i + 3] = depths[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) colors[7 * (size_t)i + 4 + c] = normals[3 * (size_t)i + c];
        mask[i] = (uint8_t)open_bits;
    }
}

// ACC: add to v_coeffs instead of overwriting it (gg_sh_bwd_accumulate: the caller's gradient buffer)
// TAIL (gg_shade_tail_bwd): v_colors is a (N, >= 7) array with rows vstride floats apart (the interleaved gradient
// record of the blend backward); the colour cotangent passes the clamp where the forward's mask says so, the depth
// and normal cotangents are copied out.
template <int K, bool ACC = false, bool TAIL = false>
__global__ __launch_bounds__(256) void sh_bwd_kernel(int N, int deg,
                                                     const float *__restrict__ viewdirs,
                                                     const float *__restrict__ v_colors,
                                                     float *__restrict__ v_coeffs, int vstride = 3,
                                                     const uint8_t *__restrict__ mask = nullptr,
                                                     float *__restrict__ v_depths = nullptr,
                                                     float *__restrict__ v_normals = nullptr) {
    constexpr int ROW = 3 * K;
    __shared__ float stage[4][64 * ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (blockIdx.x * 4 + wave) * 64;
    if (base >= N) return;
    const int nrows = min(64, N - base);
    float *st = stage[wave];
    const int i = base + lane;
    if (lane < nrows) {
        float Y[GG_SH_MAX_BASES];
        sh_basis(deg, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        const int nb = min(sh_nbases(deg), K);
        float vc[3];
        if (TAIL) {
            const float *vt = v_colors + (size_t)i * vstride;
            const unsigned m = mask[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) vc[c] = ((m >> c) & 1u) ? vt[c] : 0.0f;
            v_depths[i] = vt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v_normals[3 * (size_t)i + c] = vt[4 + c];
        } else {
            vc[0] = v_colors[3 * i]; vc[1] = v_colors[3 * i + 1]; vc[2] = v_colors[3 * i + 2];
        }
        float *row = st + lane * ROW;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) row[3 * k + c] = (k < nb) ? Y[k] * vc[c] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    float *dst = v_coeffs + (size_t)base * ROW;
    for (int e = lane; e < nrows * ROW; e += 64) dst[e] = ACC ? dst[e] + st[e] : st[e];
}

// ---- C ABI -----------------------------------------------------------------------------------
extern "C" int gg_project_fwd(int N, const float *means3d, const float *scales, float glob_scale,
                              const float *quats, const float *viewmat, const float *projmat,
                              float fx, float fy, float cx, float cy, int img_height,
                              int img_width, int tiles_x, int tiles_y, float clip_thresh,
                              float *cov3d, float *xys, float *depths, int32_t *radii,
                              float *conics, int32_t *num_tiles_hit, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_height > 0 && img_width > 0, "empty image");
    GG_REQUIRE(tiles_x == (img_width + GG_BLOCK - 1) / GG_BLOCK &&
                   tiles_y == (img_height + GG_BLOCK - 1) / GG_BLOCK,
               "tile_bounds must be ceil(W/16), ceil(H/16)");
    if (N == 0) return GG_OK;
    GG_REQUIRE(means3d && scales && quats && viewmat && projmat && cov3d && xys && depths &&
                   radii && conics && num_tiles_hit,
               "null pointer");
    gg_prof_begin(GG_K_PROJECT_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(project_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, N, means3d, scales, glob_scale, quats, viewmat,
                       projmat, fx, fy, cx, cy, img_height, img_width, tiles_x, tiles_y,
                       clip_thresh, cov3d, xys, depths, radii, conics, num_tiles_hit);
    gg_prof_end(GG_K_PROJECT_FWD, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" size_t gg_project_count_workspace(int N) { return sizeof(unsigned) * (size_t)(N > 0 ? (N + 255) / 256 : 1); }
extern "C" int gg_project_fwd_count(int N, const float *means3d, const float *scales, float glob_scale,
                                    const float *quats, const float *viewmat, const float *projmat,
                                    float fx, float fy, float cx, float cy, int img_height,
                                    int img_width, int tiles_x, int tiles_y, float clip_thresh,
                                    float *cov3d, float *xys, float *depths, int32_t *radii,
                                    float *conics, int32_t *num_tiles_hit, int64_t *num_intersects_out,
                                    void *count_ws, size_t count_ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_height > 0 && img_width > 0, "empty image");
    GG_REQUIRE(tiles_x == (img_width + GG_BLOCK - 1) / GG_BLOCK &&
                   tiles_y == (img_height + GG_BLOCK - 1) / GG_BLOCK,
               "tile_bounds must be ceil(W/16), ceil(H/16)");
    GG_REQUIRE(num_intersects_out != nullptr, "null num_intersects_out");
    if (N == 0) {
        if (gg_fill_async(num_intersects_out, 0, sizeof(int64_t), (hipStream_t)stream) != hipSuccess) {
            gg_set_error("gg_project_fwd_count: memset failed");
            return GG_ERR_LAUNCH;
        }
        return GG_OK;
    }
    GG_REQUIRE(means3d && scales && quats && viewmat && projmat && cov3d && xys && depths &&
                   radii && conics && num_tiles_hit,
               "null pointer");
    if (count_ws == nullptr || count_ws_bytes < gg_project_count_workspace(N) || ((uintptr_t)count_ws & 3)) {
        gg_set_error("gg_project_fwd_count: count workspace of gg_project_count_workspace() bytes expected");
        return GG_ERR_WORKSPACE;
    }
    gg_prof_begin(GG_K_PROJECT_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(project_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, N, means3d, scales, glob_scale, quats, viewmat,
                       projmat, fx, fy, cx, cy, img_height, img_width, tiles_x, tiles_y,
                       clip_thresh, cov3d, xys, depths, radii, conics, num_tiles_hit, (unsigned *)count_ws);
    gg_prof_end(GG_K_PROJECT_FWD, (hipStream_t)stream);
    gg_prof_begin(GG_K_COUNT, (hipStream_t)stream);
    hipLaunchKernelGGL(count_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (N + 255) / 256,
                       (const unsigned *)count_ws, (unsigned long long *)num_intersects_out);
    gg_prof_end(GG_K_COUNT, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_project_bwd_ex(int N, const float *means3d, const float *scales, float glob_scale,
                                 const float *quats, const float *viewmat, const float *projmat,
                                 float fx, float fy, float cx, float cy, int img_height,
                                 int img_width, const int32_t *radii, const float *conics,
                                 const float *v_xy, int v_xy_stride, const float *v_depth, const float *v_conic,
                                 int v_conic_stride, float *v_mean3d, int accumulate_means, float *v_scale,
                                 float *v_quat, gg_stream_t stream);
extern "C" int gg_project_bwd(int N, const float *means3d, const float *scales, float glob_scale,
                              const float *quats, const float *viewmat, const float *projmat,
                              float fx, float fy, float cx, float cy, int img_height,
                              int img_width, const int32_t *radii, const float *conics,
                              const float *v_xy, const float *v_depth, const float *v_conic,
                              float *v_mean3d, float *v_scale, float *v_quat,
                              gg_stream_t stream) {
    return gg_project_bwd_ex(N, means3d, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy, img_height,
                             img_width, radii, conics, v_xy, 2, v_depth, v_conic, 3, v_mean3d, 0, v_scale, v_quat,
                             stream);
}
extern "C" int gg_project_bwd_ex(int N, const float *means3d, const float *scales, float glob_scale,
                                 const float *quats, const float *viewmat, const float *projmat,
                                 float fx, float fy, float cx, float cy, int img_height,
                                 int img_width, const int32_t *radii, const float *conics,
                                 const float *v_xy, int v_xy_stride, const float *v_depth, const float *v_conic,
                                 int v_conic_stride, float *v_mean3d, int accumulate_means, float *v_scale,
                                 float *v_quat, gg_stream_t stream) {
    (void)cx;
    (void)cy;
    GG_REQUIRE(v_xy_stride >= 2 && v_conic_stride >= 3, "v_xy rows hold 2 values, v_conic rows 3");
    GG_REQUIRE(N >= 0, "num_points < 0");
    if (N == 0) return GG_OK;
    GG_REQUIRE(means3d && scales && quats && viewmat && projmat && radii && conics && v_xy &&
                   v_depth && v_conic && v_mean3d && v_scale && v_quat,
               "null pointer");
    gg_prof_begin(GG_K_PROJECT_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(project_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, N, means3d, scales, glob_scale, quats, viewmat,
                       projmat, fx, fy, img_height, img_width, radii, conics, v_xy, v_depth,
                       v_conic, v_mean3d, v_scale, v_quat, v_xy_stride, v_conic_stride, accumulate_means);
    gg_prof_end(GG_K_PROJECT_BWD, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int K>
static void launch_sh(bool fwd, bool acc, int N, int deg, const float *viewdirs, const float *in, float *out,
                      hipStream_t s) {
    dim3 grid((N + 255) / 256), block(256);
    const int rows_fwd = 4 * ((K >= 16) ? 32 : 64);   // Gaussians per forward workgroup
    if (fwd)
        hipLaunchKernelGGL(sh_fwd_kernel<K>, dim3((N + rows_fwd - 1) / rows_fwd), block, 0, s, N, deg,
                           viewdirs, in, out);
    else if (acc)
        hipLaunchKernelGGL((sh_bwd_kernel<K, true>), grid, block, 0, s, N, deg, viewdirs, in, out);
    else
        hipLaunchKernelGGL((sh_bwd_kernel<K, false>), grid, block, 0, s, N, deg, viewdirs, in, out);
}

static int sh_dispatch(bool fwd, bool acc, int N, int K, int deg, const float *viewdirs, const float *in,
                       float *out, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(K == 1 || K == 4 || K == 9 || K == 16 || K == 25, "num_bases must be 1,4,9,16,25");
    GG_REQUIRE(deg >= 0 && sh_nbases(deg) <= K, "degrees_to_use exceeds stored bases");
    if (N == 0) return GG_OK;
    GG_REQUIRE(viewdirs && in && out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(fwd ? GG_K_SH_FWD : GG_K_SH_BWD, s);
    switch (K) {
        case 1: launch_sh<1>(fwd, acc, N, deg, viewdirs, in, out, s); break;
        case 4: launch_sh<4>(fwd, acc, N, deg, viewdirs, in, out, s); break;
        case 9: launch_sh<9>(fwd, acc, N, deg, viewdirs, in, out, s); break;
        case 16: launch_sh<16>(fwd, acc, N, deg, viewdirs, in, out, s); break;
        default: launch_sh<25>(fwd, acc, N, deg, viewdirs, in, out, s); break;
    }
    gg_prof_end(fwd ? GG_K_SH_FWD : GG_K_SH_BWD, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
extern "C" int gg_sh_fwd(int N, int K, int deg, const float *viewdirs, const float *coeffs,
                         float *colors, gg_stream_t stream) {
    return sh_dispatch(true, false, N, K, deg, viewdirs, coeffs, colors, stream);
}
extern "C" int gg_sh_bwd(int N, int K, int deg, const float *viewdirs, const float *v_colors,
                         float *v_coeffs, gg_stream_t stream) {
    return sh_dispatch(false, false, N, K, deg, viewdirs, v_colors, v_coeffs, stream);
}
extern "C" int gg_sh_bwd_accumulate(int N, int K, int deg, const float *viewdirs, const float *v_colors,
                                    float *v_coeffs, gg_stream_t stream) {
    return sh_dispatch(false, true, N, K, deg, viewdirs, v_colors, v_coeffs, stream);
}

// ---- deferred SH gradient: expansion of several views' colour cotangents in one pass --------------------------
// Over the views of an optimizer step the SH gradient is sum_v Y(dir_v) (x) v_rgb_v: added view by view it is a
// read-modify-write of 600 B per Gaussian per view (0.157 ms each at 1 M Gaussians); kept as its factors — 12 B
// of masked colour cotangent per Gaussian and view, the view directions are there anyway — it is expanded ONCE per
// step: 24 B read per view + 300 B written (+ 300 B read when the buffer already holds something).  Sums run over
// the views in order, in registers, starting from the buffer's value: the bits of adding view after view.
#define GG_SH_MULTI_VIEWS 16
struct ShMultiArgs {
    const float *viewdirs[GG_SH_MULTI_VIEWS];   // (N, 3) each
    const float *v_colors[GG_SH_MULTI_VIEWS];   // (N, 3) each, already masked by the clamp
    int num_views;
};
template <int K, bool ACC>
__global__ __launch_bounds__(256) void sh_bwd_multi_kernel(int N, int deg, ShMultiArgs a,
                                                           float *__restrict__ v_coeffs) {
    constexpr int ROW = 3 * K;
    __shared__ float stage[4][64 * ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (blockIdx.x * 4 + wave) * 64;
    if (base >= N) return;
    const int nrows = min(64, N - base);
    float *st = stage[wave];
    const int i = base + lane;
    float *dst = v_coeffs + (size_t)base * ROW;
    if (ACC) {   // the sums continue from what the buffer holds: exactly the bits of adding view after view
        for (int e = lane; e < nrows * ROW; e += 64) st[e] = dst[e];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < nrows) {
        float acc[ROW];
        float *row0 = st + lane * ROW;
#pragma unroll
        for (int e = 0; e < ROW; ++e) acc[e] = ACC ? row0[e] : 0.0f;
        const int nb = min(sh_nbases(deg), K);
        for (int v = 0; v < a.num_views; ++v) {
            const float *vd = a.viewdirs[v] + 3 * (size_t)i;
            const float *vc = a.v_colors[v] + 3 * (size_t)i;
            float Y[GG_SH_MAX_BASES];
            sh_basis(deg, vd[0], vd[1], vd[2], Y);
            const float c0 = vc[0], c1 = vc[1], c2 = vc[2];
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < nb) {
                    acc[3 * k + 0] = acc[3 * k + 0] + Y[k] * c0;
                    acc[3 * k + 1] = acc[3 * k + 1] + Y[k] * c1;
                    acc[3 * k + 2] = acc[3 * k + 2] + Y[k] * c2;
                }
        }
        float *row = st + lane * ROW;
#pragma unroll
        for (int e = 0; e < ROW; ++e) row[e] = acc[e];
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < nrows * ROW; e += 64) dst[e] = st[e];
}
// the part of the tail backward that cannot wait: masked colour cotangent (kept for the expansion), depth and
// normal cotangents (needed by the projection / activation backward of this view)
__global__ __launch_bounds__(256) void tail_split_kernel(int N, const float *__restrict__ v_tail, int vstride,
                                                         const uint8_t *__restrict__ mask,
                                                         float *__restrict__ v_rgb, float *__restrict__ v_depths,
                                                         float *__restrict__ v_normals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *vt = v_tail + (size_t)i * vstride;
    const unsigned m = mask[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) v_rgb[3 * (size_t)i + c] = ((m >> c) & 1u) ? vt[c] : 0.0f;
    v_depths[i] = vt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v_normals[3 * (size_t)i + c] = vt[4 + c];
}
extern "C" int gg_shade_tail_bwd_split(int N, const float *v_tail, int v_tail_stride, const uint8_t *clamp_mask,
                                       float *v_rgb, float *v_depths, float *v_normals, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(v_tail_stride >= 7, "v_tail rows hold 7 values");
    if (N == 0) return GG_OK;
    GG_REQUIRE(v_tail && clamp_mask && v_rgb && v_depths && v_normals, "null pointer");
    gg_prof_begin(GG_K_TAIL_SPLIT, (hipStream_t)stream);
    hipLaunchKernelGGL(tail_split_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, v_tail,
                       v_tail_stride, clamp_mask, v_rgb, v_depths, v_normals);
    gg_prof_end(GG_K_TAIL_SPLIT, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
template <int K>
static void launch_sh_multi(bool acc, int N, int deg, const ShMultiArgs &a, float *out, hipStream_t s) {
    dim3 grid((N + 255) / 256), block(256);
    if (acc) hipLaunchKernelGGL((sh_bwd_multi_kernel<K, true>), grid, block, 0, s, N, deg, a, out);
    else hipLaunchKernelGGL((sh_bwd_multi_kernel<K, false>), grid, block, 0, s, N, deg, a, out);
}
extern "C" int gg_sh_bwd_multi(int N, int K, int deg, int num_views, const float *const *viewdirs,
                               const float *const *v_colors, float *v_coeffs, int accumulate, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(K == 1 || K == 4 || K == 9 || K == 16 || K == 25, "num_bases must be 1,4,9,16,25");
    GG_REQUIRE(deg >= 0 && sh_nbases(deg) <= K, "degrees_to_use exceeds stored bases");
    GG_REQUIRE(num_views >= 1, "num_views < 1");
    if (N == 0) return GG_OK;
    GG_REQUIRE(viewdirs && v_colors && v_coeffs, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_SH_BWD, s);
    for (int first = 0; first < num_views; first += GG_SH_MULTI_VIEWS) {
        ShMultiArgs a;
        a.num_views = min(GG_SH_MULTI_VIEWS, num_views - first);
        for (int v = 0; v < GG_SH_MULTI_VIEWS; ++v) {
            const int src = first + (v < a.num_views ? v : 0);
            GG_REQUIRE(viewdirs[src] && v_colors[src], "null view pointer");
            a.viewdirs[v] = viewdirs[src];
            a.v_colors[v] = v_colors[src];
        }
        const bool acc = accumulate != 0 || first > 0;
        switch (K) {
            case 1: launch_sh_multi<1>(acc, N, deg, a, v_coeffs, s); break;
            case 4: launch_sh_multi<4>(acc, N, deg, a, v_coeffs, s); break;
            case 9: launch_sh_multi<9>(acc, N, deg, a, v_coeffs, s); break;
            case 16: launch_sh_multi<16>(acc, N, deg, a, v_coeffs, s); break;
            default: launch_sh_multi<25>(acc, N, deg, a, v_coeffs, s); break;
        }
    }
    gg_prof_end(GG_K_SH_BWD, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int K>
static void launch_tail(bool fwd, bool acc, int N, int deg, const float *viewdirs, const float *in, float *out,
                        const float *depths, const float *normals, uint8_t *mask, int vstride, float *v_depths,
                        float *v_normals, hipStream_t s) {
    dim3 grid((N + 255) / 256), block(256);
    const int rows_fwd = 4 * ((K >= 16) ? 32 : 64);
    if (fwd)
        hipLaunchKernelGGL((sh_fwd_kernel<K, true>), dim3((N + rows_fwd - 1) / rows_fwd), block, 0, s, N, deg,
                           viewdirs, in, out, depths, normals, mask);
    else if (acc)
        hipLaunchKernelGGL((sh_bwd_kernel<K, true, true>), grid, block, 0, s, N, deg, viewdirs, in, out, vstride,
                           mask, v_depths, v_normals);
    else
        hipLaunchKernelGGL((sh_bwd_kernel<K, false, true>), grid, block, 0, s, N, deg, viewdirs, in, out, vstride,
                           mask, v_depths, v_normals);
}
static int tail_dispatch(bool fwd, bool acc, int N, int K, int deg, const float *viewdirs, const float *in,
                         float *out, const float *depths, const float *normals, uint8_t *mask, int vstride,
                         float *v_depths, float *v_normals, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(K == 1 || K == 4 || K == 9 || K == 16 || K == 25, "num_bases must be 1,4,9,16,25");
    GG_REQUIRE(deg >= 0 && sh_nbases(deg) <= K, "degrees_to_use exceeds stored bases");
    if (N == 0) return GG_OK;
    GG_REQUIRE(viewdirs && in && out && mask, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(fwd ? GG_K_SH_FWD : GG_K_SH_BWD, s);
    switch (K) {
        case 1: launch_tail<1>(fwd, acc, N, deg, viewdirs, in, out, depths, normals, mask, vstride, v_depths, v_normals, s); break;
        case 4: launch_tail<4>(fwd, acc, N, deg, viewdirs, in, out, depths, normals, mask, vstride, v_depths, v_normals, s); break;
        case 9: launch_tail<9>(fwd, acc, N, deg, viewdirs, in, out, depths, normals, mask, vstride, v_depths, v_normals, s); break;
        case 16: launch_tail<16>(fwd, acc, N, deg, viewdirs, in, out, depths, normals, mask, vstride, v_depths, v_normals, s); break;
        default: launch_tail<25>(fwd, acc, N, deg, viewdirs, in, out, depths, normals, mask, vstride, v_depths, v_normals, s); break;
    }
    gg_prof_end(fwd ? GG_K_SH_FWD : GG_K_SH_BWD, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
extern "C" int gg_shade_tail_fwd(int N, int K, int deg, const float *viewdirs, const float *coeffs,
                                 const float *depths, const float *normals, float *tail, uint8_t *clamp_mask,
                                 gg_stream_t stream) {
    GG_REQUIRE(N == 0 || (depths && normals), "null pointer");
    return tail_dispatch(true, false, N, K, deg, viewdirs, coeffs, tail, depths, normals, clamp_mask, 7, nullptr,
                         nullptr, stream);
}
extern "C" int gg_shade_tail_bwd(int N, int K, int deg, const float *viewdirs, const float *v_tail,
                                 int v_tail_stride, const uint8_t *clamp_mask, float *v_coeffs, int accumulate,
                                 float *v_depths, float *v_normals, gg_stream_t stream) {
    GG_REQUIRE(v_tail_stride >= 7, "v_tail rows hold 7 values");
    GG_REQUIRE(N == 0 || (v_depths && v_normals), "null pointer");
    return tail_dispatch(false, accumulate != 0, N, K, deg, viewdirs, v_tail, v_coeffs, nullptr, nullptr,
                         const_cast<uint8_t *>(clamp_mask), v_tail_stride, v_depths, v_normals, stream);
}

// ------------------------------------------------------------------------------------------------
// quat_to_rotmat (gsplat._torch_impl; reference gaussian_splatting.py:516,614): one lane per
// quaternion, the (N,9) side staged through LDS so that global traffic stays coalesced.
// Operation order is the oracle's (ggo_quat_to_rotmat_fwd): bit-exact forward.
// ------------------------------------------------------------------------------------------------
#define QR_THREADS 256
__device__ __forceinline__ float quat_normalise(float4 q, float &w, float &x, float &y, float &z) {
    float nn = ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w;
    float d = fmaxf(sqrtf(nn), GG_QUAT_NORM_EPS);
    w = q.x / d;
    x = q.y / d;
    y = q.z / d;
    z = q.w / d;
    return d;
}
__global__ void __launch_bounds__(QR_THREADS) quat_to_rotmat_fwd_kernel(int N, const float4 *quats,
                                                                       float *rot) {
    __shared__ float tile[QR_THREADS * 9];
    const int base = blockIdx.x * QR_THREADS, i = base + threadIdx.x;
    if (i < N) {
        float w, x, y, z;
        quat_normalise(quats[i], w, x, y, z);
        float *R = tile + threadIdx.x * 9;   // stride 9 words: conflict-free (9 odd)
        R[0] = 1.0f - 2.0f * (y * y + z * z);
        R[1] = 2.0f * (x * y - w * z);
        R[2] = 2.0f * (x * z + w * y);
        R[3] = 2.0f * (x * y + w * z);
        R[4] = 1.0f - 2.0f * (x * x + z * z);
        R[5] = 2.0f * (y * z - w * x);
        R[6] = 2.0f * (x * z - w * y);
        R[7] = 2.0f * (y * z + w * x);
        R[8] = 1.0f - 2.0f * (x * x + y * y);
    }
    __syncthreads();
    const long lim = (long)min(QR_THREADS, N - base) * 9;
    float *dst = rot + (long)base * 9;
    for (int k = threadIdx.x; k < lim; k += QR_THREADS) dst[k] = tile[k];
}
__global__ void __launch_bounds__(QR_THREADS) quat_to_rotmat_bwd_kernel(int N, const float4 *quats,
                                                                       const float *v_rot,
                                                                       float4 *v_quats) {
    __shared__ float tile[QR_THREADS * 9];
    const int base = blockIdx.x * QR_THREADS, i = base + threadIdx.x;
    const long lim = (long)min(QR_THREADS, N - base) * 9;
    const float *src = v_rot + (long)base * 9;
    for (int k = threadIdx.x; k < lim; k += QR_THREADS) tile[k] = src[k];
    __syncthreads();
    if (i >= N) return;
    float w, x, y, z;
    const float d = quat_normalise(quats[i], w, x, y, z);
    const float *g = tile + threadIdx.x * 9;
    // gradient w.r.t. the normalised quaternion
    float vw = 2.0f * (x * (g[7] - g[5]) + y * (g[2] - g[6]) + z * (g[3] - g[1]));
    float vx = 2.0f * (y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] - g[5]) -
                       2.0f * x * (g[4] + g[8]));
    float vy = 2.0f * (x * (g[1] + g[3]) + z * (g[5] + g[7]) + w * (g[2] - g[6]) -
                       2.0f * y * (g[0] + g[8]));
    float vz = 2.0f * (x * (g[2] + g[6]) + y * (g[5] + g[7]) + w * (g[3] - g[1]) -
                       2.0f * z * (g[0] + g[4]));
    // through n = q / max(|q|, eps): (v - n <n, v>) / |q| above the floor, v / eps below it
    float4 out;
    if (d > GG_QUAT_NORM_EPS) {
        float dot = ((w * vw + x * vx) + y * vy) + z * vz;
        out = make_float4((vw - w * dot) / d, (vx - x * dot) / d, (vy - y * dot) / d,
                          (vz - z * dot) / d);
    } else {
        out = make_float4(vw / d, vx / d, vy / d, vz / d);
    }
    v_quats[i] = out;
}
extern "C" int gg_quat_to_rotmat_fwd(int N, const float *quats, float *rot, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    if (N == 0) return GG_OK;
    GG_REQUIRE(quats && rot, "null pointer");
    GG_REQUIRE(((uintptr_t)quats & 15) == 0, "quats must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_QUAT_FWD, s);
    hipLaunchKernelGGL(quat_to_rotmat_fwd_kernel, dim3((N + QR_THREADS - 1) / QR_THREADS),
                       dim3(QR_THREADS), 0, s, N, (const float4 *)quats, rot);
    gg_prof_end(GG_K_QUAT_FWD, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
extern "C" int gg_quat_to_rotmat_bwd(int N, const float *quats, const float *v_rot, float *v_quats,
                                     gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    if (N == 0) return GG_OK;
    GG_REQUIRE(quats && v_rot && v_quats, "null pointer");
    GG_REQUIRE((((uintptr_t)quats | (uintptr_t)v_quats) & 15) == 0,
               "quats / v_quats must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    gg_prof_begin(GG_K_QUAT_BWD, s);
    hipLaunchKernelGGL(quat_to_rotmat_bwd_kernel, dim3((N + QR_THREADS - 1) / QR_THREADS),
                       dim3(QR_THREADS), 0, s, N, (const float4 *)quats, v_rot, (float4 *)v_quats);
    gg_prof_end(GG_K_QUAT_BWD, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------
// Caller-side activations of one view in one kernel each way: what the reference's
// `get_outputs` does with ~12 torch launches forward and ~25 backward per view —
//   torch.exp(scales) :701, quats / quats.norm(dim=-1, keepdim=True) :703, torch.sigmoid(opacities) :742,
//   viewdirs = normalize(means.detach() - camera position) :727-728, and get_normals() :605-619 =
//   the column of quat_to_rotmat(quats) at argmin(exp(scales)).
// Used by the plugin's fused model (pipeline.activate_fused); the shim route keeps the caller's torch ops.
// One lane per Gaussian; O(N), ~70 B read and ~70 B written per Gaussian.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void activate_fwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float4 *__restrict__ quats, const float *__restrict__ opacities, const float *__restrict__ cam_pos,
    float *__restrict__ scales, float4 *__restrict__ quats_n, float *__restrict__ opac,
    float *__restrict__ viewdirs, float *__restrict__ normals, int32_t *__restrict__ axis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e[k] = expf(log_scales[3 * (size_t)i + k]);
        scales[3 * (size_t)i + k] = e[k];
    }
    int ax = 0;                                   // first minimum, as torch.min(dim) reports it
    if (e[1] < e[ax]) ax = 1;
    if (e[2] < e[ax]) ax = 2;
    axis[i] = ax;
    const float4 q = quats[i];
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    quats_n[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);           // :703 (no eps, as the reference)
    float w, x, y, z;
    quat_normalise(q, w, x, y, z);                                            // F.normalize inside quat_to_rotmat
    float col[3];
    if (ax == 0) { col[0] = 1.0f - 2.0f * (y * y + z * z); col[1] = 2.0f * (x * y + w * z); col[2] = 2.0f * (x * z - w * y); }
    else if (ax == 1) { col[0] = 2.0f * (x * y - w * z); col[1] = 1.0f - 2.0f * (x * x + z * z); col[2] = 2.0f * (y * z + w * x); }
    else { col[0] = 2.0f * (x * z + w * y); col[1] = 2.0f * (y * z - w * x); col[2] = 1.0f - 2.0f * (x * x + y * y); }
    opac[i] = 1.0f / (1.0f + expf(-opacities[i]));
    float d[3], dn = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = means[3 * (size_t)i + k] - cam_pos[k];
        dn += d[k] * d[k];
    }
    dn = sqrtf(dn);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        viewdirs[3 * (size_t)i + k] = d[k] / dn;
        normals[3 * (size_t)i + k] = col[k];
    }
}

// ------------------------------------------------------------------------------------------------
// activate_fwd_kernel + project_fwd_kernel in one pass over the Gaussians (round 4, ops.ViewGeometry): the activated
// scales and the normalised quaternion go from the activation to the projection in registers (they are still written:
// the backward reads them), cov3d is not written at all (nothing of the plugin route reads it), and every workgroup
// leaves three partial results behind — the sum of its num_tiles_hit (count_finish_kernel adds them up) and the smallest
// and largest depth bits of its visible Gaussians (what gg_bin_sort's depth buckets need: db_range_kernel's pass saved).
// The per-Gaussian arithmetic is the two kernels': same operation sequence, same bits.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void view_fwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ log_scales, const float4 *__restrict__ quats,
    const float *__restrict__ opacities, const float *__restrict__ cam_pos, const float *__restrict__ viewmat,
    const float *__restrict__ projmat, float fx, float fy, float cx, float cy, int img_h, int img_w, int tiles_x,
    int tiles_y, float clip_thresh, float *__restrict__ scales, float4 *__restrict__ quats_n, float *__restrict__ opac,
    float *__restrict__ viewdirs, float *__restrict__ normals, int32_t *__restrict__ axis, float *__restrict__ xys,
    float *__restrict__ depths, int32_t *__restrict__ radii, float *__restrict__ conics,
    int32_t *__restrict__ num_tiles_hit, unsigned *__restrict__ parts /* [3][workgroups]: sum, min bits, max bits */,
    GRec *__restrict__ records /* nullable: the blend kernels' packed records (blend_prep_kernel's pass saved) */) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    if (!live) i = N - 1;          // (a padding thread of the last workgroup: computes, stores and counts nothing)
    float V[12], P[16];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = viewmat[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = projmat[k];
    // ---- activations (activate_fwd_kernel)
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) e[k] = expf(log_scales[3 * (size_t)i + k]);
    int ax = 0;                                   // first minimum, as torch.min(dim) reports it
    if (e[1] < e[ax]) ax = 1;
    if (e[2] < e[ax]) ax = 2;
    const float4 q = quats[i];
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    const float4 qn = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);           // :703 (no eps, as the reference)
    float w, x, y, z;
    quat_normalise(q, w, x, y, z);                                                 // F.normalize inside quat_to_rotmat
    float col[3];
    if (ax == 0) { col[0] = 1.0f - 2.0f * (y * y + z * z); col[1] = 2.0f * (x * y + w * z); col[2] = 2.0f * (x * z - w * y); }
    else if (ax == 1) { col[0] = 2.0f * (x * y - w * z); col[1] = 1.0f - 2.0f * (x * x + z * z); col[2] = 2.0f * (y * z + w * x); }
    else { col[0] = 2.0f * (x * z + w * y); col[1] = 2.0f * (y * z - w * x); col[2] = 1.0f - 2.0f * (x * x + y * y); }
    const float op = 1.0f / (1.0f + expf(-opacities[i]));
    const float px = means[3 * (size_t)i], py = means[3 * (size_t)i + 1], pz = means[3 * (size_t)i + 2];
    float d[3] = {px - cam_pos[0], py - cam_pos[1], pz - cam_pos[2]};
    float dn = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) dn += d[k] * d[k];
    dn = sqrtf(dn);
    // ---- projection (project_fwd_kernel with glob_scale 1)
    float o_c3[6] = {0, 0, 0, 0, 0, 0}, o_con[3] = {0, 0, 0};
    float o_x = 0, o_y = 0, o_d = 0;
    int o_r = 0, o_n = 0;
    project_fwd_point(px, py, pz, qn, 1.0f * e[0], 1.0f * e[1], 1.0f * e[2], V, P, fx, fy, cx, cy, img_h, img_w, tiles_x,
                      tiles_y, clip_thresh, o_c3, o_con, o_x, o_y, o_d, o_r, o_n);
    if (live) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            scales[3 * (size_t)i + k] = e[k];
            viewdirs[3 * (size_t)i + k] = d[k] / dn;
            normals[3 * (size_t)i + k] = col[k];
            conics[3 * (size_t)i + k] = o_con[k];
        }
        axis[i] = ax;
        quats_n[i] = qn;
        opac[i] = op;
        xys[2 * (size_t)i] = o_x;
        xys[2 * (size_t)i + 1] = o_y;
        depths[i] = o_d;
        radii[i] = o_r;
        num_tiles_hit[i] = o_n;
        if (records) grec_pack(o_x, o_y, op, o_con[0], o_con[1], o_con[2], records + i);
    }
    // ---- the workgroup's partial results
    __shared__ unsigned s_sum[4], s_lo[4], s_hi[4];
    const bool vis = live && o_r > 0;
    unsigned mine = live ? (unsigned)o_n : 0u;
    unsigned lo = vis ? __builtin_bit_cast(unsigned, o_d) : 0xFFFFFFFFu, hi = vis ? __builtin_bit_cast(unsigned, o_d) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        mine += __shfl_down(mine, off, 64);
        lo = min(lo, (unsigned)__shfl_down((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_down((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = mine;
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        parts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        parts[gridDim.x + blockIdx.x] = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        parts[2 * gridDim.x + blockIdx.x] = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    }
}
extern "C" size_t gg_view_fwd_workspace(int N) { return sizeof(unsigned) * 3 * (size_t)(N > 0 ? (N + 255) / 256 : 1); }
extern "C" int gg_view_fwd(int N, const float *means, const float *log_scales, const float *quats, const float *opacities,
                           const float *cam_pos, const float *viewmat, const float *projmat, float fx, float fy, float cx,
                           float cy, int img_height, int img_width, int tiles_x, int tiles_y, float clip_thresh,
                           float *scales, float *quats_n, float *opac, float *viewdirs, float *normals, int32_t *axis,
                           float *xys, float *depths, int32_t *radii, float *conics, int32_t *num_tiles_hit,
                           int64_t *num_intersects_out, void *parts, size_t parts_bytes, void *records,
                           size_t records_bytes, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(img_height > 0 && img_width > 0, "empty image");
    GG_REQUIRE(tiles_x == (img_width + GG_BLOCK - 1) / GG_BLOCK && tiles_y == (img_height + GG_BLOCK - 1) / GG_BLOCK,
               "tile_bounds must be ceil(W/16), ceil(H/16)");
    GG_REQUIRE(num_intersects_out != nullptr, "null num_intersects_out");
    GG_REQUIRE(records == nullptr || (records_bytes >= sizeof(GRec) * (size_t)(N > 0 ? N : 1) && ((uintptr_t)records & 15) == 0),
               "records: gg_blend_workspace(num_points) bytes, 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (gg_fill_async(num_intersects_out, 0, sizeof(int64_t), s) != hipSuccess) {
            gg_set_error("gg_view_fwd: memset failed");
            return GG_ERR_LAUNCH;
        }
        return GG_OK;
    }
    GG_REQUIRE(means && log_scales && quats && opacities && cam_pos && viewmat && projmat && scales && quats_n && opac &&
                   viewdirs && normals && axis && xys && depths && radii && conics && num_tiles_hit, "null pointer");
    GG_REQUIRE((((uintptr_t)quats | (uintptr_t)quats_n) & 15) == 0, "quats / quats_n must be 16-byte aligned");
    if (parts == nullptr || parts_bytes < gg_view_fwd_workspace(N) || ((uintptr_t)parts & 3)) {
        gg_set_error("gg_view_fwd: partial-result array of gg_view_fwd_workspace() bytes expected");
        return GG_ERR_WORKSPACE;
    }
    const int blocks = (N + 255) / 256;
    gg_prof_begin(GG_K_VIEW_FWD, s);
    hipLaunchKernelGGL(view_fwd_kernel, dim3(blocks), dim3(256), 0, s, N, means, log_scales, (const float4 *)quats,
                       opacities, cam_pos, viewmat, projmat, fx, fy, cx, cy, img_height, img_width, tiles_x, tiles_y,
                       clip_thresh, scales, (float4 *)quats_n, opac, viewdirs, normals, axis, xys, depths, radii, conics,
                       num_tiles_hit, (unsigned *)parts, (GRec *)records);
    gg_prof_end(GG_K_VIEW_FWD, s);
    gg_prof_begin(GG_K_COUNT, s);
    hipLaunchKernelGGL(count_finish_kernel, dim3(1), dim3(1024), 0, s, blocks, (const unsigned *)parts,
                       (unsigned long long *)num_intersects_out);
    gg_prof_end(GG_K_COUNT, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// the backward of one Gaussian's activations: (v_scales, v_quats_n, v_opac, v_normals) -> gradients of the log
// scales, the raw quaternion and the opacity logit.  Shared by activate_bwd_kernel and view_bwd_kernel.
__device__ __forceinline__ void activate_bwd_point(const float4 q, const float (&sc)[3], const float s, const int ax,
                                                   const float (&vsc)[3], const float4 g, const float vop,
                                                   const float n0, const float n1, const float n2, float (&g_ls)[3],
                                                   float &g_op, float4 &out) {
#pragma unroll
    for (int k = 0; k < 3; ++k) g_ls[k] = vsc[k] * sc[k];
    g_op = vop * (s * (1.0f - s));
    // q / |q|
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    const float4 qh = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    const float dot = ((qh.x * g.x + qh.y * g.y) + qh.z * g.z) + qh.w * g.w;
    out = make_float4((g.x - qh.x * dot) / n, (g.y - qh.y * dot) / n, (g.z - qh.z * dot) / n, (g.w - qh.w * dot) / n);
    // the normal: one column of R(q / max(|q|, eps))
    float w, x, y, z;
    const float d = quat_normalise(q, w, x, y, z);
    float G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // v_R: only column `ax`
    G[ax] = n0;
    G[3 + ax] = n1;
    G[6 + ax] = n2;
    float vw = 2.0f * (x * (G[7] - G[5]) + y * (G[2] - G[6]) + z * (G[3] - G[1]));
    float vx = 2.0f * (y * (G[1] + G[3]) + z * (G[2] + G[6]) + w * (G[7] - G[5]) - 2.0f * x * (G[4] + G[8]));
    float vy = 2.0f * (x * (G[1] + G[3]) + z * (G[5] + G[7]) + w * (G[2] - G[6]) - 2.0f * y * (G[0] + G[8]));
    float vz = 2.0f * (x * (G[2] + G[6]) + y * (G[5] + G[7]) + w * (G[3] - G[1]) - 2.0f * z * (G[0] + G[4]));
    if (d > GG_QUAT_NORM_EPS) {
        const float dt = ((w * vw + x * vx) + y * vy) + z * vz;
        out.x += (vw - w * dt) / d;
        out.y += (vx - x * dt) / d;
        out.z += (vy - y * dt) / d;
        out.w += (vz - z * dt) / d;
    } else {
        out.x += vw / d;
        out.y += vx / d;
        out.z += vy / d;
        out.w += vz / d;
    }
}
__global__ __launch_bounds__(256) void activate_bwd_kernel(
    int N, const float4 *__restrict__ quats, const float *__restrict__ scales, const float *__restrict__ opac,
    const int32_t *__restrict__ axis, const float *__restrict__ v_scales, const float4 *__restrict__ v_quats_n,
    const float *__restrict__ v_opac, const float *__restrict__ v_normals, float *__restrict__ v_log_scales,
    float4 *__restrict__ v_quats, float *__restrict__ v_opacities, int opac_stride = 1, int acc = 0) {
    // opac_stride: floats between the entries of v_opac (read in place from the blend backward's record);
    // acc: the three outputs are added to (registered gradient buffers) instead of written
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float sc[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
    const float vsc[3] = {v_scales[3 * (size_t)i], v_scales[3 * (size_t)i + 1], v_scales[3 * (size_t)i + 2]};
    float g_ls[3], g_op;
    float4 out;
    activate_bwd_point(quats[i], sc, opac[i], axis[i], vsc, v_quats_n[i], v_opac[(size_t)i * opac_stride],
                       v_normals[3 * (size_t)i], v_normals[3 * (size_t)i + 1], v_normals[3 * (size_t)i + 2], g_ls, g_op, out);
#pragma unroll
    for (int k = 0; k < 3; ++k) v_log_scales[3 * (size_t)i + k] = acc ? v_log_scales[3 * (size_t)i + k] + g_ls[k] : g_ls[k];
    v_opacities[i] = acc ? v_opacities[i] + g_op : g_op;
    if (acc) {
        const float4 p = v_quats[i];
        out = make_float4(p.x + out.x, p.y + out.y, p.z + out.z, p.w + out.w);
    }
    v_quats[i] = out;
}

// ------------------------------------------------------------------------------------------------
// One backward pass over the Gaussians of a view (round 4; SURVEY rows a2-a4, the plugin's ops.ViewGeometry): what
// tail_split_kernel, project_bwd_kernel and activate_bwd_kernel did in three launches (26 + 36 + 40 us per view at 1 M
// Gaussians, each re-reading the blend backward's 64-byte record or the previous kernel's output).  Reads the record
// [v_xy 0..1 | v_conic 2..4 | v_opacity 5 | v_rgb 6..8 | v_depth 9 | v_normal 10..12] once, keeps the clamp-masked colour
// cotangent for the SH expansion (12 B), and adds the gradients of means, log scales, quaternions and opacity logits into
// the step's gradient buffers.  The per-Gaussian arithmetic is project_bwd_point and activate_bwd_point above: same
// operation sequence, same bits as the three kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void view_bwd_kernel(
    int N, const float *__restrict__ rec, int rec_stride, const uint8_t *__restrict__ clamp_mask,
    const float *__restrict__ means, const float *__restrict__ scales, float glob_scale,
    const float4 *__restrict__ quats_raw, const float4 *__restrict__ quats_n, const float *__restrict__ opac,
    const int32_t *__restrict__ axis,
    const float *__restrict__ viewmat, const float *__restrict__ projmat, float fx, float fy, int img_h, int img_w,
    const int32_t *__restrict__ radii, const float *__restrict__ conics, float *__restrict__ v_rgb,
    float *__restrict__ v_means, float *__restrict__ v_log_scales, float4 *__restrict__ v_quats,
    float *__restrict__ v_opacities) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *r = rec + (size_t)i * rec_stride;
    float t[13];
    if ((rec_stride & 3) == 0) {   // 16-byte aligned rows (the pair backward's 16-float records): four loads
        const float4 a = reinterpret_cast<const float4 *>(r)[0], b = reinterpret_cast<const float4 *>(r)[1],
                     c = reinterpret_cast<const float4 *>(r)[2];
        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
        t[8] = c.x; t[9] = c.y; t[10] = c.z; t[11] = c.w; t[12] = r[12];
    } else {
#pragma unroll
        for (int k = 0; k < 13; ++k) t[k] = r[k];
    }
    const unsigned m = clamp_mask[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) v_rgb[3 * (size_t)i + c] = ((m >> c) & 1u) ? t[6 + c] : 0.0f;
    float vm[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, vq4[4] = {0, 0, 0, 0};
    if (radii[i] > 0)
        project_bwd_point(i, means, scales, glob_scale, reinterpret_cast<const float *>(quats_n), viewmat, projmat, fx, fy, img_h,
                          img_w, conics, t[0], t[1], t[9], t[2], t[3], t[4], vm, vs, vq4);
#pragma unroll
    for (int k = 0; k < 3; ++k) v_means[3 * (size_t)i + k] += vm[k];
    const float sc[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
    float g_ls[3], g_op;
    float4 out;
    activate_bwd_point(quats_raw[i], sc, opac[i], axis[i], vs, make_float4(vq4[0], vq4[1], vq4[2], vq4[3]), t[5], t[10],
                       t[11], t[12], g_ls, g_op, out);
#pragma unroll
    for (int k = 0; k < 3; ++k) v_log_scales[3 * (size_t)i + k] += g_ls[k];
    v_opacities[i] += g_op;
    const float4 p = v_quats[i];
    v_quats[i] = make_float4(p.x + out.x, p.y + out.y, p.z + out.z, p.w + out.w);
}
extern "C" int gg_view_bwd(int N, const float *rec, int rec_stride, const uint8_t *clamp_mask, const float *means,
                           const float *scales, float glob_scale, const float *quats_raw, const float *quats_n, const float *opac,
                           const int32_t *axis, const float *viewmat, const float *projmat, float fx, float fy,
                           int img_height, int img_width, const int32_t *radii, const float *conics, float *v_rgb,
                           float *v_means, float *v_log_scales, float *v_quats, float *v_opacities, gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(rec_stride >= 13, "a record holds 13 values");
    if (N == 0) return GG_OK;
    GG_REQUIRE(rec && clamp_mask && means && scales && quats_raw && quats_n && opac && axis && viewmat && projmat && radii &&
                   conics && v_rgb && v_means && v_log_scales && v_quats && v_opacities, "null pointer");
    GG_REQUIRE((((uintptr_t)quats_raw | (uintptr_t)quats_n | (uintptr_t)v_quats) & 15) == 0,
               "quaternion arrays must be 16-byte aligned");
    GG_REQUIRE((rec_stride & 3) != 0 || ((uintptr_t)rec & 15) == 0, "records of a multiple of 4 floats must be 16-byte aligned");
    gg_prof_begin(GG_K_VIEW_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(view_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, rec, rec_stride,
                       clamp_mask, means, scales, glob_scale, (const float4 *)quats_raw, (const float4 *)quats_n, opac, axis, viewmat,
                       projmat, fx, fy, img_height, img_width, radii, conics, v_rgb, v_means, v_log_scales,
                       (float4 *)v_quats, v_opacities);
    gg_prof_end(GG_K_VIEW_BWD, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_activate_fwd(int N, const float *means, const float *log_scales, const float *quats,
                               const float *opacities, const float *cam_pos, float *scales, float *quats_n,
                               float *opac, float *viewdirs, float *normals, int32_t *axis,
                               gg_stream_t stream) {
    GG_REQUIRE(N >= 0, "num_points < 0");
    if (N == 0) return GG_OK;
    GG_REQUIRE(means && log_scales && quats && opacities && cam_pos && scales && quats_n && opac && viewdirs &&
                   normals && axis, "null pointer");
    GG_REQUIRE((((uintptr_t)quats | (uintptr_t)quats_n) & 15) == 0, "quats / quats_n must be 16-byte aligned");
    gg_prof_begin(GG_K_ACTIVATE_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(activate_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, means,
                       log_scales, (const float4 *)quats, opacities, cam_pos, scales, (float4 *)quats_n, opac,
                       viewdirs, normals, axis);
    gg_prof_end(GG_K_ACTIVATE_FWD, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
extern "C" int gg_activate_bwd_ex(int N, const float *quats, const float *scales, const float *opac,
                                  const int32_t *axis, const float *v_scales, const float *v_quats_n,
                                  const float *v_opac, int v_opac_stride, const float *v_normals, float *v_log_scales,
                                  float *v_quats, float *v_opacities, int accumulate, gg_stream_t stream);
extern "C" int gg_activate_bwd(int N, const float *quats, const float *scales, const float *opac,
                               const int32_t *axis, const float *v_scales, const float *v_quats_n,
                               const float *v_opac, const float *v_normals, float *v_log_scales,
                               float *v_quats, float *v_opacities, gg_stream_t stream) {
    return gg_activate_bwd_ex(N, quats, scales, opac, axis, v_scales, v_quats_n, v_opac, 1, v_normals, v_log_scales,
                              v_quats, v_opacities, 0, stream);
}
extern "C" int gg_activate_bwd_ex(int N, const float *quats, const float *scales, const float *opac,
                                  const int32_t *axis, const float *v_scales, const float *v_quats_n,
                                  const float *v_opac, int v_opac_stride, const float *v_normals, float *v_log_scales,
                                  float *v_quats, float *v_opacities, int accumulate, gg_stream_t stream) {
    GG_REQUIRE(v_opac_stride >= 1, "v_opac_stride < 1");
    GG_REQUIRE(N >= 0, "num_points < 0");
    if (N == 0) return GG_OK;
    GG_REQUIRE(quats && scales && opac && axis && v_scales && v_quats_n && v_opac && v_normals && v_log_scales &&
                   v_quats && v_opacities, "null pointer");
    GG_REQUIRE((((uintptr_t)quats | (uintptr_t)v_quats_n | (uintptr_t)v_quats) & 15) == 0,
               "quaternion arrays must be 16-byte aligned");
    gg_prof_begin(GG_K_ACTIVATE_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N,
                       (const float4 *)quats, scales, opac, axis, v_scales, (const float4 *)v_quats_n, v_opac,
                       v_normals, v_log_scales, (float4 *)v_quats, v_opacities, v_opac_stride, accumulate);
    gg_prof_end(GG_K_ACTIVATE_BWD, (hipStream_t)stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__global__ void expf_kernel(int n, const float *x, float *y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gg_expf(x[i]);
}
extern "C" int gg_expf_array(int n, const float *x, float *y, gg_stream_t stream) {
    GG_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GG_OK;
    GG_REQUIRE(x && y, "null pointer");
    hipLaunchKernelGGL(expf_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, x,
                       y);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
