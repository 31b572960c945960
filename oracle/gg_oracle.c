/*
 * gg_oracle.c — CPU ORACLE for the Gaussian-splatting hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (gaussiangrasper_amd/, shim/gsplat) never imports, links or
 * executes anything under oracle/.
 *
 * PARITY UNPINNED.  The arithmetic restated here lives in the reference's un-vendored
 * third-party dependency gsplat==0.1.0 (reference requirements.txt:78); its source is not
 * under /root/reference and is not installable offline, and the reference holds no golden
 * vector, known-answer test or fixture for this path.  What IS exact is
 * the call-site contract: nerfstudio/models/gaussian_splatting.py:699-713 (ProjectGaussians),
 * :730 (SphericalHarmonics), :735-784 (Rasterize / NDRasterize), :87-105 (projection_matrix).
 * Each function below names the gsplat-0.1.0 routine it restates (recollection of the public
 * source) and the reference call site that reaches it.  Every
 * uncertain constant comes from include/gg_constants.h.
 *
 * Numerics: one source, two builds.
 *   libgg_oracle_f32.so  REAL=float   -ffp-contract=off : the parity oracle.  Every fp32
 *       operation is written out in a fixed order (explicit fmaf where a fused op is meant),
 *       so a GPU kernel that follows the same order reproduces radii, tile counts, depth
 *       bits, sort keys and the forward image bit for bit.
 *   libgg_oracle_f64.so  REAL=double : same formulas in fp64, used only for
 *       finite-difference checks of the analytic backward passes.
 *
 * Layouts are torch's: row-major, array-of-structs; (N,3) means, (N,4) quats wxyz,
 * (N,K,3) SH coefficients, (N,C) colours, (H,W,C) images, (T,2) tile bins.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gg_constants.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef GGO_F64
typedef double REAL;
#define R_(x) ((double)(x))
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(x) sqrt(x)
#define CEIL(x) ceil(x)
#define FMIN(a, b) fmin((a), (b))
#define FMAX(a, b) fmax((a), (b))
#define NAME(x) ggo64_##x
#else
typedef float REAL;
#define R_(x) (x)
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define CEIL(x) ceilf(x)
#define FMIN(a, b) fminf((a), (b))
#define FMAX(a, b) fmaxf((a), (b))
#define NAME(x) ggo_##x
#endif

/* ------------------------------------------------------------------------------------------
 * gg_expf — stands in for CUDA's __expf(-sigma) in gsplat's rasterize_forward/backward.
 * Operation sequence fixed in include/gg_constants.h (the HIP kernels follow the same one).
 * ---------------------------------------------------------------------------------------- */
static inline REAL gg_exp(REAL x) {
#ifdef GGO_F64
    return exp(x);
#else
    if (x < GG_EXP_LO) return 0.0f;
    float t = x * GG_EXP_LOG2E;
    float n = rintf(t);
    float r = fmaf(n, -GG_EXP_LN2_HI, x);
    r = fmaf(n, -GG_EXP_LN2_LO, r);
    float p = GG_EXP_P0;
    p = fmaf(p, r, GG_EXP_P1);
    p = fmaf(p, r, GG_EXP_P2);
    p = fmaf(p, r, GG_EXP_P3);
    p = fmaf(p, r, GG_EXP_P4);
    p = fmaf(p, r, GG_EXP_P5);
    float z = r * r;
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    union {
        uint32_t u;
        float f;
    } s;
    s.u = (uint32_t)((int32_t)n + 127) << 23;
    return y * s.f;
#endif
}

/* exported for the tests that pin gg_expf against libm and against the GPU */
void NAME(expf_array)(int n, const REAL *x, REAL *y) {
    for (int i = 0; i < n; ++i) y[i] = gg_exp(x[i]);
}

/* ------------------------------------------------------------------------------------------
 * Tile bounding box — gsplat helpers.cuh get_tile_bbox/get_bbox (†):
 *   tile_min = clamp((int)(c/16 - r/16), 0, bound), tile_max = clamp((int)(c/16 + r/16 + 1), 0, bound).
 * The clamp is applied in the float domain before the conversion: identical for every finite
 * input ((int) truncates toward zero and everything below 0 clamps to 0) and defined for huge
 * or NaN centres, where the C cast is not.
 * ---------------------------------------------------------------------------------------- */
static inline int clampi_f(REAL v, int bound) {
    REAL c = FMIN(FMAX(v, R_(0.0f)), (REAL)bound);
    return (int)c;
}
static inline void tile_bbox(REAL cx, REAL cy, REAL radius, int tiles_x, int tiles_y, int *x0,
                             int *y0, int *x1, int *y1) {
    REAL tcx = cx / (REAL)GG_BLOCK, tcy = cy / (REAL)GG_BLOCK;
    REAL tr = radius / (REAL)GG_BLOCK;
    *x0 = clampi_f(tcx - tr, tiles_x);
    *x1 = clampi_f((tcx + tr) + R_(1.0f), tiles_x);
    *y0 = clampi_f(tcy - tr, tiles_y);
    *y1 = clampi_f((tcy + tr) + R_(1.0f), tiles_y);
}

/* rotation matrix of the NORMALISED quaternion (w,x,y,z), row-major — gsplat helpers.cuh
 * quat_to_rotmat (†; it normalises with rsqrtf, here 1/sqrt, both IEEE-exact on our side) */
static inline void quat_to_R(const REAL *q, REAL *R, REAL *qn, REAL *inv_norm) {
    REAL nn = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    REAL inv = R_(1.0f) / SQRT(nn);
    REAL w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    R[0] = R_(1.0f) - R_(2.0f) * (y * y + z * z);
    R[1] = R_(2.0f) * (x * y - w * z);
    R[2] = R_(2.0f) * (x * z + w * y);
    R[3] = R_(2.0f) * (x * y + w * z);
    R[4] = R_(1.0f) - R_(2.0f) * (x * x + z * z);
    R[5] = R_(2.0f) * (y * z - w * x);
    R[6] = R_(2.0f) * (x * z - w * y);
    R[7] = R_(2.0f) * (y * z + w * x);
    R[8] = R_(1.0f) - R_(2.0f) * (x * x + y * y);
    if (qn) {
        qn[0] = w;
        qn[1] = x;
        qn[2] = y;
        qn[3] = z;
    }
    if (inv_norm) *inv_norm = inv;
}

/* ------------------------------------------------------------------------------------------
 * quat_to_rotmat_fwd / _bwd — restate gsplat `_torch_impl.quat_to_rotmat` (†: F.normalize(q) then
 * the standard matrix; differentiable torch code) as called at
 * nerfstudio/models/gaussian_splatting.py:516,614 and scripts/update.py:204,229.  The backward
 * is the analytic VJP through the matrix entries and through n = q / max(|q|, 1e-12).
 * ---------------------------------------------------------------------------------------- */
static inline REAL quat_normalise(const REAL *q, REAL *n) {
    REAL nn = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    REAL d = SQRT(nn);
    if (!(d >= R_(GG_QUAT_NORM_EPS))) d = R_(GG_QUAT_NORM_EPS);
    for (int k = 0; k < 4; ++k) n[k] = q[k] / d;
    return d;
}
void NAME(quat_to_rotmat_fwd)(int N, const REAL *quats, REAL *rot) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL n[4];
        quat_normalise(quats + 4 * (size_t)i, n);
        const REAL w = n[0], x = n[1], y = n[2], z = n[3];
        REAL *R = rot + 9 * (size_t)i;
        R[0] = R_(1.0f) - R_(2.0f) * (y * y + z * z);
        R[1] = R_(2.0f) * (x * y - w * z);
        R[2] = R_(2.0f) * (x * z + w * y);
        R[3] = R_(2.0f) * (x * y + w * z);
        R[4] = R_(1.0f) - R_(2.0f) * (x * x + z * z);
        R[5] = R_(2.0f) * (y * z - w * x);
        R[6] = R_(2.0f) * (x * z - w * y);
        R[7] = R_(2.0f) * (y * z + w * x);
        R[8] = R_(1.0f) - R_(2.0f) * (x * x + y * y);
    }
}
void NAME(quat_to_rotmat_bwd)(int N, const REAL *quats, const REAL *v_rot, REAL *v_quats) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL n[4], v[4];
        const REAL d = quat_normalise(quats + 4 * (size_t)i, n);
        const REAL w = n[0], x = n[1], y = n[2], z = n[3];
        const REAL *g = v_rot + 9 * (size_t)i;
        v[0] = R_(2.0f) * (x * (g[7] - g[5]) + y * (g[2] - g[6]) + z * (g[3] - g[1]));
        v[1] = R_(2.0f) * (y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] - g[5]) -
                           R_(2.0f) * x * (g[4] + g[8]));
        v[2] = R_(2.0f) * (x * (g[1] + g[3]) + z * (g[5] + g[7]) + w * (g[2] - g[6]) -
                           R_(2.0f) * y * (g[0] + g[8]));
        v[3] = R_(2.0f) * (x * (g[2] + g[6]) + y * (g[5] + g[7]) + w * (g[3] - g[1]) -
                           R_(2.0f) * z * (g[0] + g[4]));
        REAL *o = v_quats + 4 * (size_t)i;
        if (d > R_(GG_QUAT_NORM_EPS)) {
            REAL dot = ((w * v[0] + x * v[1]) + y * v[2]) + z * v[3];
            for (int k = 0; k < 4; ++k) o[k] = (v[k] - n[k] * dot) / d;
        } else {
            for (int k = 0; k < 4; ++k) o[k] = v[k] / d;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * project_fwd — restates gsplat forward.cu project_gaussians_forward_kernel (†) as reached by
 * ProjectGaussians.apply at nerfstudio/models/gaussian_splatting.py:699-713.
 * viewmat: row-major, first 12 floats used (the caller passes viewmat[:3,:]); projmat: 4x4
 * (projmat @ viewmat).  Outputs are zero for culled Gaussians except cov3d/conics, which the
 * kernel writes before the later culls (cov3d after the near-plane test, conics after det!=0).
 * ---------------------------------------------------------------------------------------- */
void NAME(project_fwd)(int N, const REAL *means, const REAL *scales, REAL glob_scale,
                       const REAL *quats, const REAL *viewmat, const REAL *projmat, REAL fx,
                       REAL fy, REAL cx, REAL cy, int img_h, int img_w, int tiles_x, int tiles_y,
                       REAL clip_thresh, REAL *cov3d, REAL *xys, REAL *depths, int32_t *radii,
                       REAL *conics, int32_t *num_tiles_hit) {
    const REAL *V = viewmat, *P = projmat;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        radii[i] = 0;
        num_tiles_hit[i] = 0;
        depths[i] = 0;
        xys[2 * i] = xys[2 * i + 1] = 0;
        conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0;
        for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = 0;

        REAL px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
        /* clip_near_plane / transform_4x3 */
        REAL tx = ((V[0] * px + V[1] * py) + V[2] * pz) + V[3];
        REAL ty = ((V[4] * px + V[5] * py) + V[6] * pz) + V[7];
        REAL tz = ((V[8] * px + V[9] * py) + V[10] * pz) + V[11];
        if (tz <= clip_thresh) continue;

        /* scale_rot_to_cov3d: M = R * diag(glob*s), cov3d = M M^T (upper triangle) */
        REAL R[9];
        quat_to_R(quats + 4 * i, R, 0, 0);
        REAL s0 = glob_scale * scales[3 * i], s1 = glob_scale * scales[3 * i + 1],
             s2 = glob_scale * scales[3 * i + 2];
        REAL M[9] = {R[0] * s0, R[1] * s1, R[2] * s2, R[3] * s0, R[4] * s1,
                     R[5] * s2, R[6] * s0, R[7] * s1, R[8] * s2};
        REAL c3[6];
        c3[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
        c3[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
        c3[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
        c3[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
        c3[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
        c3[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
        for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = c3[k];

        /* project_cov3d_ewa */
        REAL tan_fovx = (R_(0.5f) * (REAL)img_w) / fx;
        REAL tan_fovy = (R_(0.5f) * (REAL)img_h) / fy;
        REAL lim_x = R_(GG_FOV_LIM) * tan_fovx, lim_y = R_(GG_FOV_LIM) * tan_fovy;
        REAL txc = tz * FMIN(lim_x, FMAX(-lim_x, tx / tz));
        REAL tyc = tz * FMIN(lim_y, FMAX(-lim_y, ty / tz));
        REAL rz = R_(1.0f) / tz;
        REAL rz2 = rz * rz;
        REAL J00 = fx * rz, J02 = (-(fx * txc)) * rz2;
        REAL J11 = fy * rz, J12 = (-(fy * tyc)) * rz2;
        /* T = J * W, W = viewmat[:3,:3] */
        REAL T00 = J00 * V[0] + J02 * V[8], T01 = J00 * V[1] + J02 * V[9],
             T02 = J00 * V[2] + J02 * V[10];
        REAL T10 = J11 * V[4] + J12 * V[8], T11 = J11 * V[5] + J12 * V[9],
             T12 = J11 * V[6] + J12 * V[10];
        /* TV = T * cov3d (2x3) */
        REAL A00 = (T00 * c3[0] + T01 * c3[1]) + T02 * c3[2];
        REAL A01 = (T00 * c3[1] + T01 * c3[3]) + T02 * c3[4];
        REAL A02 = (T00 * c3[2] + T01 * c3[4]) + T02 * c3[5];
        REAL A10 = (T10 * c3[0] + T11 * c3[1]) + T12 * c3[2];
        REAL A11 = (T10 * c3[1] + T11 * c3[3]) + T12 * c3[4];
        REAL A12 = (T10 * c3[2] + T11 * c3[4]) + T12 * c3[5];
        REAL a = ((A00 * T00 + A01 * T01) + A02 * T02) + R_(GG_BLUR);
        REAL b = (A00 * T10 + A01 * T11) + A02 * T12;
        REAL c = ((A10 * T10 + A11 * T11) + A12 * T12) + R_(GG_BLUR);

        /* compute_cov2d_bounds */
        REAL det = a * c - b * b;
        if (det == R_(0.0f)) continue;
        REAL inv_det = R_(1.0f) / det;
        conics[3 * i] = c * inv_det;
        conics[3 * i + 1] = (-b) * inv_det;
        conics[3 * i + 2] = a * inv_det;
        REAL bm = R_(0.5f) * (a + c);
        REAL sq = SQRT(FMAX(R_(GG_EIG_FLOOR), bm * bm - det));
        REAL v1 = bm + sq, v2 = bm - sq;
        REAL radius = CEIL(R_(GG_RADIUS_SIGMA) * SQRT(FMAX(v1, v2)));

        /* project_pix / ndc2pix */
        REAL hx = ((P[0] * px + P[1] * py) + P[2] * pz) + P[3];
        REAL hy = ((P[4] * px + P[5] * py) + P[6] * pz) + P[7];
        REAL hw = ((P[12] * px + P[13] * py) + P[14] * pz) + P[15];
        REAL rw = R_(1.0f) / (hw + R_(GG_W_EPS));
        REAL ndx = hx * rw, ndy = hy * rw;
        REAL ux = ((R_(0.5f) * (REAL)img_w) * ndx + cx) - R_(GG_PIX_OFFSET);
        REAL uy = ((R_(0.5f) * (REAL)img_h) * ndy + cy) - R_(GG_PIX_OFFSET);

        int x0, y0, x1, y1;
        tile_bbox(ux, uy, radius, tiles_x, tiles_y, &x0, &y0, &x1, &y1);
        int area = (x1 - x0) * (y1 - y0);
        if (area <= 0) continue;
        num_tiles_hit[i] = area;
        depths[i] = tz;
        radii[i] = (int32_t)radius;
        xys[2 * i] = ux;
        xys[2 * i + 1] = uy;
    }
}

/* ------------------------------------------------------------------------------------------
 * project_bwd — VJP of project_fwd w.r.t. means, scales, quats (autograd of
 * gaussian_splatting.py:699; gsplat backward.cu project_gaussians_backward_kernel †).
 * v_conic follows gsplat's convention: it is the gradient of the symmetric 2x2 conic MATRIX,
 * i.e. v_conic[1] is HALF of dL/d(conic.y) (rasterize_backward emits it that way, a11).
 * Gaussians with radii<=0 receive zero gradients.  This is the exact VJP of the forward as
 * written above (including the homogeneous-w path of the pixel projection and the FOV clamp
 * in the EWA Jacobian); PARITY.md lists where gsplat 0.1.0 is recalled to approximate.
 * ---------------------------------------------------------------------------------------- */
void NAME(project_bwd)(int N, const REAL *means, const REAL *scales, REAL glob_scale,
                       const REAL *quats, const REAL *viewmat, const REAL *projmat, REAL fx,
                       REAL fy, REAL cx, REAL cy, int img_h, int img_w, const int32_t *radii,
                       const REAL *conics, const REAL *v_xy, const REAL *v_depth,
                       const REAL *v_conic, REAL *v_mean3d, REAL *v_scale, REAL *v_quat) {
    (void)cx;
    (void)cy;
    const REAL *V = viewmat, *P = projmat;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) v_mean3d[3 * i + k] = 0, v_scale[3 * i + k] = 0;
        for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = 0;
        if (radii[i] <= 0) continue;
        REAL px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
        REAL vm[3] = {0, 0, 0};

        /* (1) pixel centre */
        REAL hx = ((P[0] * px + P[1] * py) + P[2] * pz) + P[3];
        REAL hy = ((P[4] * px + P[5] * py) + P[6] * pz) + P[7];
        REAL hw = ((P[12] * px + P[13] * py) + P[14] * pz) + P[15];
        REAL rw = R_(1.0f) / (hw + R_(GG_W_EPS));
        REAL vnx = (R_(0.5f) * (REAL)img_w) * v_xy[2 * i];
        REAL vny = (R_(0.5f) * (REAL)img_h) * v_xy[2 * i + 1];
        REAL vhx = vnx * rw, vhy = vny * rw;
#if GG_VJP_GSPLAT_COMPAT
        REAL vhw = R_(0.0f) * hx * hy; /* compat: the homogeneous-w path is dropped */
#else
        REAL vhw = -((vnx * hx + vny * hy) * (rw * rw));
#endif
        for (int j = 0; j < 3; ++j) vm[j] += (P[j] * vhx + P[4 + j] * vhy) + P[12 + j] * vhw;

        /* (2) depth = row 2 of viewmat */
        REAL vz = v_depth[i];
        for (int j = 0; j < 3; ++j) vm[j] += V[8 + j] * vz;

        /* (3) conic -> cov2d : v_Sigma = -X G X */
        REAL ca = conics[3 * i], cb = conics[3 * i + 1], cc = conics[3 * i + 2];
        REAL ga = v_conic[3 * i], gb = v_conic[3 * i + 1], gc = v_conic[3 * i + 2];
        /* XG */
        REAL xg00 = ca * ga + cb * gb, xg01 = ca * gb + cb * gc;
        REAL xg10 = cb * ga + cc * gb, xg11 = cb * gb + cc * gc;
        REAL s00 = -(xg00 * ca + xg01 * cb), s01 = -(xg00 * cb + xg01 * cc);
        REAL s10 = -(xg10 * ca + xg11 * cb), s11 = -(xg10 * cb + xg11 * cc);
        REAL v_a = s00, v_b = s01 + s10, v_c = s11;

        /* recompute forward intermediates */
        REAL tx = ((V[0] * px + V[1] * py) + V[2] * pz) + V[3];
        REAL ty = ((V[4] * px + V[5] * py) + V[6] * pz) + V[7];
        REAL tz = ((V[8] * px + V[9] * py) + V[10] * pz) + V[11];
        REAL R[9], qn[4], inv_norm;
        quat_to_R(quats + 4 * i, R, qn, &inv_norm);
        REAL s[3] = {glob_scale * scales[3 * i], glob_scale * scales[3 * i + 1],
                     glob_scale * scales[3 * i + 2]};
        REAL M[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) M[3 * r + k] = R[3 * r + k] * s[k];
        REAL C3[9]; /* full symmetric cov3d */
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k)
                C3[3 * r + k] = (M[3 * r] * M[3 * k] + M[3 * r + 1] * M[3 * k + 1]) +
                                M[3 * r + 2] * M[3 * k + 2];
        REAL tan_fovx = (R_(0.5f) * (REAL)img_w) / fx, tan_fovy = (R_(0.5f) * (REAL)img_h) / fy;
        REAL lim_x = R_(GG_FOV_LIM) * tan_fovx, lim_y = R_(GG_FOV_LIM) * tan_fovy;
        REAL rx = tx / tz, ry = ty / tz;
#if GG_VJP_GSPLAT_COMPAT
        int clx = 0, cly = 0; /* compat: Jacobian rebuilt at the unclamped point, no clamp derivative */
        REAL txc = tx + R_(0.0f) * rx * lim_x, tyc = ty + R_(0.0f) * ry * lim_y;
#else
        int clx = (rx > lim_x) ? 1 : ((rx < -lim_x) ? -1 : 0);
        int cly = (ry > lim_y) ? 1 : ((ry < -lim_y) ? -1 : 0);
        REAL txc = tz * FMIN(lim_x, FMAX(-lim_x, rx));
        REAL tyc = tz * FMIN(lim_y, FMAX(-lim_y, ry));
#endif
        REAL rz = R_(1.0f) / tz, rz2 = rz * rz, rz3 = rz2 * rz;
        REAL J00 = fx * rz, J02 = (-(fx * txc)) * rz2, J11 = fy * rz, J12 = (-(fy * tyc)) * rz2;
        REAL Tm[6] = {J00 * V[0] + J02 * V[8], J00 * V[1] + J02 * V[9], J00 * V[2] + J02 * V[10],
                      J11 * V[4] + J12 * V[8], J11 * V[5] + J12 * V[9], J11 * V[6] + J12 * V[10]};

        /* (4) cov2d = Tm C3 Tm^T ; Gc = [[v_a, v_b/2],[v_b/2, v_c]] */
        REAL g00 = v_a, g01 = R_(0.5f) * v_b, g11 = v_c;
        /* GT = Gc * Tm (2x3) */
        REAL GT[6];
        for (int k = 0; k < 3; ++k) {
            GT[k] = g00 * Tm[k] + g01 * Tm[3 + k];
            GT[3 + k] = g01 * Tm[k] + g11 * Tm[3 + k];
        }
        /* entrywise v_C3 = Tm^T Gc Tm (3x3, symmetric) */
        REAL G3[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) G3[3 * r + k] = Tm[r] * GT[k] + Tm[3 + r] * GT[3 + k];
        /* v_Tm = 2 * Gc Tm C3 */
        REAL vT[6];
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 3; ++k)
                vT[3 * r + k] = R_(2.0f) * ((GT[3 * r] * C3[k] + GT[3 * r + 1] * C3[3 + k]) +
                                            GT[3 * r + 2] * C3[6 + k]);
        /* (5) v_J = v_Tm W^T ; only J00,J02,J11,J12 are live */
        REAL vJ00 = (vT[0] * V[0] + vT[1] * V[1]) + vT[2] * V[2];
        REAL vJ02 = (vT[0] * V[8] + vT[1] * V[9]) + vT[2] * V[10];
        REAL vJ11 = (vT[3] * V[4] + vT[4] * V[5]) + vT[5] * V[6];
        REAL vJ12 = (vT[3] * V[8] + vT[4] * V[9]) + vT[5] * V[10];
        REAL v_txc = (-(fx * rz2)) * vJ02;
        REAL v_tyc = (-(fy * rz2)) * vJ12;
        REAL v_tz = ((-(fx * rz2)) * vJ00 - (fy * rz2) * vJ11) +
                    (R_(2.0f) * fx * txc * rz3) * vJ02 + (R_(2.0f) * fy * tyc * rz3) * vJ12;
        REAL v_tx, v_ty;
        if (clx) {
            v_tz += ((REAL)clx * lim_x) * v_txc;
            v_tx = 0;
        } else
            v_tx = v_txc;
        if (cly) {
            v_tz += ((REAL)cly * lim_y) * v_tyc;
            v_ty = 0;
        } else
            v_ty = v_tyc;
        for (int j = 0; j < 3; ++j) vm[j] += (V[j] * v_tx + V[4 + j] * v_ty) + V[8 + j] * v_tz;
        for (int j = 0; j < 3; ++j) v_mean3d[3 * i + j] = vm[j];

        /* (6) cov3d = M M^T -> v_M = 2 G3 M ; M = R diag(s) */
        REAL vM[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k)
                vM[3 * r + k] = R_(2.0f) * ((G3[3 * r] * M[k] + G3[3 * r + 1] * M[3 + k]) +
                                            G3[3 * r + 2] * M[6 + k]);
        for (int k = 0; k < 3; ++k)
            v_scale[3 * i + k] =
                glob_scale * ((R[k] * vM[k] + R[3 + k] * vM[3 + k]) + R[6 + k] * vM[6 + k]);
        REAL G[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) G[3 * r + k] = vM[3 * r + k] * s[k];
        REAL w = qn[0], x = qn[1], y = qn[2], z = qn[3];
        REAL vq[4];
        vq[0] = R_(2.0f) * ((x * (G[7] - G[5]) + y * (G[2] - G[6])) + z * (G[3] - G[1]));
        vq[1] = R_(2.0f) * (((R_(-2.0f) * x) * (G[4] + G[8]) + y * (G[1] + G[3])) +
                            (z * (G[2] + G[6]) + w * (G[7] - G[5])));
        vq[2] = R_(2.0f) * ((x * (G[1] + G[3]) + (R_(-2.0f) * y) * (G[0] + G[8])) +
                            (z * (G[5] + G[7]) + w * (G[2] - G[6])));
        vq[3] = R_(2.0f) * ((x * (G[2] + G[6]) + y * (G[5] + G[7])) +
                            ((R_(-2.0f) * z) * (G[0] + G[4]) + w * (G[3] - G[1])));
        /* through q/|q| */
        REAL dotp = ((qn[0] * vq[0] + qn[1] * vq[1]) + qn[2] * vq[2]) + qn[3] * vq[3];
#if GG_VJP_GSPLAT_COMPAT
        for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = vq[k] + R_(0.0f) * dotp * inv_norm;
#else
        for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = (vq[k] - qn[k] * dotp) * inv_norm;
#endif
    }
}

/* ------------------------------------------------------------------------------------------
 * Spherical harmonics — gsplat sh.cuh sh_coeffs_to_color / compute_sh_forward_kernel (†),
 * SphericalHarmonics.apply at gaussian_splatting.py:730.  K bases stored per Gaussian
 * ((N,K,3)); `degrees_to_use` = active degree n (:729).  The kernel re-normalises viewdirs.
 * ---------------------------------------------------------------------------------------- */
static inline void sh_basis(int deg, REAL dx, REAL dy, REAL dz, REAL *Y) {
    Y[0] = R_(GG_SH_C0);
    if (deg < 1) return;
    REAL norm = SQRT((dx * dx + dy * dy) + dz * dz);
    REAL x = dx / norm, y = dy / norm, z = dz / norm;
    Y[1] = R_(GG_SH_C1) * (-y);
    Y[2] = R_(GG_SH_C1) * z;
    Y[3] = R_(GG_SH_C1) * (-x);
    if (deg < 2) return;
    REAL xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
    Y[4] = R_(GG_SH_C2_0) * xy;
    Y[5] = R_(GG_SH_C2_1) * yz;
    Y[6] = R_(GG_SH_C2_2) * ((R_(2.0f) * zz - xx) - yy);
    Y[7] = R_(GG_SH_C2_3) * xz;
    Y[8] = R_(GG_SH_C2_4) * (xx - yy);
    if (deg < 3) return;
    Y[9] = (R_(GG_SH_C3_0) * y) * (R_(3.0f) * xx - yy);
    Y[10] = (R_(GG_SH_C3_1) * xy) * z;
    Y[11] = (R_(GG_SH_C3_2) * y) * ((R_(4.0f) * zz - xx) - yy);
    Y[12] = (R_(GG_SH_C3_3) * z) * ((R_(2.0f) * zz - R_(3.0f) * xx) - R_(3.0f) * yy);
    Y[13] = (R_(GG_SH_C3_4) * x) * ((R_(4.0f) * zz - xx) - yy);
    Y[14] = (R_(GG_SH_C3_5) * z) * (xx - yy);
    Y[15] = (R_(GG_SH_C3_6) * x) * (xx - R_(3.0f) * yy);
    if (deg < 4) return;
    Y[16] = (R_(GG_SH_C4_0) * xy) * (xx - yy);
    Y[17] = (R_(GG_SH_C4_1) * yz) * (R_(3.0f) * xx - yy);
    Y[18] = (R_(GG_SH_C4_2) * xy) * (R_(7.0f) * zz - R_(1.0f));
    Y[19] = (R_(GG_SH_C4_3) * yz) * (R_(7.0f) * zz - R_(3.0f));
    Y[20] = R_(GG_SH_C4_4) * (zz * (R_(35.0f) * zz - R_(30.0f)) + R_(3.0f));
    Y[21] = (R_(GG_SH_C4_5) * xz) * (R_(7.0f) * zz - R_(3.0f));
    Y[22] = (R_(GG_SH_C4_6) * (xx - yy)) * (R_(7.0f) * zz - R_(1.0f));
    Y[23] = (R_(GG_SH_C4_7) * xz) * (xx - R_(3.0f) * yy);
    Y[24] = R_(GG_SH_C4_8) * (xx * (xx - R_(3.0f) * yy) - yy * (R_(3.0f) * xx - yy));
}
static inline int sh_nbases(int deg) { return deg >= 4 ? 25 : (deg + 1) * (deg + 1); }

void NAME(sh_fwd)(int N, int K, int degrees_to_use, const REAL *viewdirs, const REAL *coeffs,
                  REAL *colors) {
    int nb = sh_nbases(degrees_to_use);
    if (nb > K) nb = K;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL Y[GG_SH_MAX_BASES];
        sh_basis(degrees_to_use, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        const REAL *cf = coeffs + (size_t)i * K * 3;
        for (int c = 0; c < 3; ++c) {
            REAL acc = Y[0] * cf[c];
            for (int k = 1; k < nb; ++k) acc = FMA(Y[k], cf[3 * k + c], acc);
            colors[3 * i + c] = acc;
        }
    }
}
/* compute_sh_backward_kernel (†): gradient to coefficients only (viewdirs are detached by the
 * caller, gaussian_splatting.py:727); inactive bases get zero. */
void NAME(sh_bwd)(int N, int K, int degrees_to_use, const REAL *viewdirs, const REAL *v_colors,
                  REAL *v_coeffs) {
    int nb = sh_nbases(degrees_to_use);
    if (nb > K) nb = K;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL Y[GG_SH_MAX_BASES];
        sh_basis(degrees_to_use, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        REAL *vc = v_coeffs + (size_t)i * K * 3;
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 3; ++c) vc[3 * k + c] = (k < nb) ? Y[k] * v_colors[3 * i + c] : 0;
    }
}

/* shade_tail: the plugin route's 7-channel colour array [ clamp(SH + 0.5, 0, 1) | depth | normal ] — what the
 * reference computes at gaussian_splatting.py:730-731 (`rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)`) and passes,
 * with the depth (:765) and normal (:779) arrays, to its rasterize calls.  mask bit c: the gradient of colour c
 * passes the clamp (torch.clamp backward: min <= x <= max).  Pinned in tests/test_shade_tail.py against torch
 * autograd of clamp / cat applied to this oracle's own sh_fwd. */
void NAME(shade_tail_fwd)(int N, int K, int degrees_to_use, const REAL *viewdirs, const REAL *coeffs,
                          const REAL *depths, const REAL *normals, REAL *tail, uint8_t *mask) {
    int nb = sh_nbases(degrees_to_use);
    if (nb > K) nb = K;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL Y[GG_SH_MAX_BASES];
        sh_basis(degrees_to_use, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        const REAL *cf = coeffs + (size_t)i * K * 3;
        unsigned bits = 0;
        for (int c = 0; c < 3; ++c) {
            REAL acc = Y[0] * cf[c];
            for (int k = 1; k < nb; ++k) acc = FMA(Y[k], cf[3 * k + c], acc);
            const REAL x = acc + R_(0.5f);
            if (x >= 0 && x <= 1) bits |= 1u << c;
            tail[7 * (size_t)i + c] = (x != x) ? x : (x < 0 ? 0 : (x > 1 ? 1 : x));
        }
        tail[7 * (size_t)i + 3] = depths[i];
        for (int c = 0; c < 3; ++c) tail[7 * (size_t)i + 4 + c] = normals[3 * (size_t)i + c];
        mask[i] = (uint8_t)bits;
    }
}
void NAME(shade_tail_bwd)(int N, int K, int degrees_to_use, const REAL *viewdirs, const REAL *v_tail,
                          int v_tail_stride, const uint8_t *mask, REAL *v_coeffs, int accumulate,
                          REAL *v_depths, REAL *v_normals) {
    int nb = sh_nbases(degrees_to_use);
    if (nb > K) nb = K;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        REAL Y[GG_SH_MAX_BASES];
        sh_basis(degrees_to_use, viewdirs[3 * i], viewdirs[3 * i + 1], viewdirs[3 * i + 2], Y);
        const REAL *vt = v_tail + (size_t)i * v_tail_stride;
        REAL vc3[3];
        for (int c = 0; c < 3; ++c) vc3[c] = ((mask[i] >> c) & 1u) ? vt[c] : 0;
        REAL *vc = v_coeffs + (size_t)i * K * 3;
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 3; ++c) {
                const REAL g = (k < nb) ? Y[k] * vc3[c] : 0;
                vc[3 * k + c] = accumulate ? vc[3 * k + c] + g : g;
            }
        v_depths[i] = vt[3];
        for (int c = 0; c < 3; ++c) v_normals[3 * (size_t)i + c] = vt[4 + c];
    }
}

/* ------------------------------------------------------------------------------------------
 * Image-space main loss (SURVEY 8f-4 tail) — get_loss_dict, gaussian_splatting.py:882-885, :931:
 *     Ll1 = torch.abs(gt_img[valid_mask, :] - outputs["rgb"][valid_mask, :]).mean()
 *     gt_img[~valid_mask, :] = 0.0;  outputs["rgb"][~valid_mask, :] = 0.0
 *     simloss = 1 - self.ssim(gt_img ..., outputs["rgb"] ...)
 *     main_loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * simloss
 * with self.ssim = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3) (:284).  pytorch-msssim
 * (requirements.txt:199, ==1.0.0) is NOT in the tree and not installed: its published algorithm is restated —
 * 1-D Gaussian window of 11 taps, sigma 1.5, normalised; `valid` separable filtering of X, Y, X^2, Y^2, XY;
 * K1 = 0.01, K2 = 0.03; ssim_map = (2 mu1 mu2 + C1)/(mu1^2 + mu2^2 + C1) * (2 s12 + C2)/(s1 + s2 + C2); mean over
 * channels and positions.  Pinned in tests/test_image_loss.py against torch autograd of that published
 * algorithm written with F.conv2d ("parity unpinned" against the absent package itself).
 * Images (H, W, 3) as the model returns them; valid (H, W) bytes, NULL = all valid.
 * ---------------------------------------------------------------------------------------- */
#define GG_SSIM_WIN 11
static void ssim_window(REAL *w) {
    REAL sum = 0;
    for (int k = 0; k < GG_SSIM_WIN; ++k) {
        const REAL c = (REAL)(k - GG_SSIM_WIN / 2);
        w[k] = (REAL)exp(-(double)(c * c) / (2.0 * 1.5 * 1.5));
        sum += w[k];
    }
    for (int k = 0; k < GG_SSIM_WIN; ++k) w[k] /= sum;
}
/* the five filtered moments at output (i, j) of channel c; X = gt, Y = rgb, both zeroed where invalid */
static void ssim_moments(int W, const REAL *rgb, const REAL *gt, const uint8_t *valid, const REAL *w,
                         int i, int j, int c, REAL *m) {
    REAL mx = 0, my = 0, mxx = 0, myy = 0, mxy = 0;
    for (int a = 0; a < GG_SSIM_WIN; ++a) {
        REAL rx = 0, ry = 0, rxx = 0, ryy = 0, rxy = 0;
        for (int b = 0; b < GG_SSIM_WIN; ++b) {
            const size_t p = (size_t)(i + a) * W + (j + b);
            const int ok = valid ? valid[p] != 0 : 1;
            const REAL x = ok ? gt[3 * p + c] : 0, y = ok ? rgb[3 * p + c] : 0;
            rx += w[b] * x; ry += w[b] * y; rxx += w[b] * (x * x); ryy += w[b] * (y * y); rxy += w[b] * (x * y);
        }
        mx += w[a] * rx; my += w[a] * ry; mxx += w[a] * rxx; myy += w[a] * ryy; mxy += w[a] * rxy;
    }
    m[0] = mx; m[1] = my; m[2] = mxx; m[3] = myy; m[4] = mxy;
}
/* ssim value and its partial derivatives with respect to mu2 (total, through s2 and s12 too), E[Y^2], E[XY] */
static REAL ssim_point(const REAL *m, REAL *g_b, REAL *g_yy, REAL *g_xy) {
    const REAL C1 = (REAL)(0.01 * 0.01), C2 = (REAL)(0.03 * 0.03);
    const REAL a = m[0], b = m[1];
    const REAL s1 = m[2] - a * a, s2 = m[3] - b * b, s12 = m[4] - a * b;
    const REAL ln = 2 * a * b + C1, ld = a * a + b * b + C1, cn = 2 * s12 + C2, cd = s1 + s2 + C2;
    const REAL L = ln / ld, CS = cn / cd;
    if (g_b) {
        const REAL dL_db = (2 * a * ld - ln * 2 * b) / (ld * ld);
        const REAL dCS_ds2 = -cn / (cd * cd), dCS_ds12 = 2 / cd;
        *g_b = CS * dL_db + L * (dCS_ds2 * (-2 * b) + dCS_ds12 * (-a));
        *g_yy = L * dCS_ds2;
        *g_xy = L * dCS_ds12;
    }
    return L * CS;
}
/* out3 = {main_loss, Ll1, ssim} */
void NAME(image_loss_fwd)(int H, int W, const REAL *rgb, const REAL *gt, const uint8_t *valid, REAL ssim_lambda,
                          REAL *out3) {
    REAL w[GG_SSIM_WIN];
    ssim_window(w);
    double l1 = 0, cnt = 0;
    for (size_t p = 0; p < (size_t)H * W; ++p)
        if (!valid || valid[p]) {
            for (int c = 0; c < 3; ++c) l1 += fabs((double)(gt[3 * p + c] - rgb[3 * p + c]));
            cnt += 3;
        }
    const int Ho = H - GG_SSIM_WIN + 1, Wo = W - GG_SSIM_WIN + 1;
    double ss = 0;
#pragma omp parallel for schedule(static) reduction(+ : ss)
    for (int i = 0; i < Ho; ++i)
        for (int j = 0; j < Wo; ++j)
            for (int c = 0; c < 3; ++c) {
                REAL m[5];
                ssim_moments(W, rgb, gt, valid, w, i, j, c, m);
                ss += (double)ssim_point(m, NULL, NULL, NULL);
            }
    const REAL Ll1 = (REAL)(l1 / cnt), ssim = (REAL)(ss / (3.0 * Ho * Wo));
    out3[0] = (1 - ssim_lambda) * Ll1 + ssim_lambda * (1 - ssim);
    out3[1] = Ll1;
    out3[2] = ssim;
}
/* v_rgb (H, W, 3) = v_main * d main_loss / d rgb (zero at invalid pixels) */
void NAME(image_loss_bwd)(int H, int W, const REAL *rgb, const REAL *gt, const uint8_t *valid, REAL ssim_lambda,
                          REAL v_main, REAL *v_rgb) {
    REAL w[GG_SSIM_WIN];
    ssim_window(w);
    const int Ho = H - GG_SSIM_WIN + 1, Wo = W - GG_SSIM_WIN + 1;
    double cnt = 0;
    for (size_t p = 0; p < (size_t)H * W; ++p)
        if (!valid || valid[p]) cnt += 3;
    REAL *g = (REAL *)malloc(sizeof(REAL) * 9 * (size_t)Ho * Wo);   /* [c][3 maps][Ho][Wo] */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Ho; ++i)
        for (int j = 0; j < Wo; ++j)
            for (int c = 0; c < 3; ++c) {
                REAL m[5], gb, gyy, gxy;
                ssim_moments(W, rgb, gt, valid, w, i, j, c, m);
                ssim_point(m, &gb, &gyy, &gxy);
                const size_t o = (size_t)i * Wo + j, plane = (size_t)Ho * Wo;
                g[(3 * c + 0) * plane + o] = gb;
                g[(3 * c + 1) * plane + o] = gyy;
                g[(3 * c + 2) * plane + o] = gxy;
            }
    const REAL ks = -ssim_lambda * v_main / (REAL)(3.0 * Ho * Wo);      /* d(lambda (1 - mean ssim)) */
    const REAL kl = (1 - ssim_lambda) * v_main / (REAL)cnt;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            const int ok = valid ? valid[p] != 0 : 1;
            for (int c = 0; c < 3; ++c) {
                if (!ok) { v_rgb[3 * p + c] = 0; continue; }
                REAL Gb = 0, Gyy = 0, Gxy = 0;
                for (int a = 0; a < GG_SSIM_WIN; ++a) {
                    const int i = y - a;
                    if (i < 0 || i >= Ho) continue;
                    REAL rb = 0, ryy = 0, rxy = 0;
                    for (int b = 0; b < GG_SSIM_WIN; ++b) {
                        const int j = x - b;
                        if (j < 0 || j >= Wo) continue;
                        const size_t o = (size_t)i * Wo + j, plane = (size_t)Ho * Wo;
                        rb += w[b] * g[(3 * c + 0) * plane + o];
                        ryy += w[b] * g[(3 * c + 1) * plane + o];
                        rxy += w[b] * g[(3 * c + 2) * plane + o];
                    }
                    Gb += w[a] * rb; Gyy += w[a] * ryy; Gxy += w[a] * rxy;
                }
                const REAL Y = rgb[3 * p + c], X = gt[3 * p + c];
                const REAL d = Y - X;
                const REAL sgn = d > 0 ? (REAL)1 : (d < 0 ? (REAL)-1 : (REAL)0);   /* torch.abs' gradient: sign */
                v_rgb[3 * p + c] = ks * (Gb + 2 * Y * Gyy + X * Gxy) + kl * sgn;
            }
        }
    free(g);
}

/* ------------------------------------------------------------------------------------------
 * Depth and normal losses of get_loss_dict (gaussian_splatting.py:879-880), P = H*W pixels, mask = depth_mask:
 *     normal_loss = 0.5 * F.mse_loss(normal[:, m], gt_normal[:, m]) + 0.5 * cosine_similarity_loss(normal[:, m], gt_normal[:, m])
 *     depth_loss  = F.l1_loss(depth[m], gt_depth[m])
 * cosine_similarity_loss (:113-118) normalises over the 3 channels with F.normalize's eps 1e-12.  Plain torch in
 * the reference: pinned in tests/test_image_loss.py against torch autograd of these lines.
 * normal / gt_normal are addressed as base[p * pstride + c * cstride] (the model's image is pixel-major, the
 * reference's ground truth channel-major); depth as base[p * stride].
 * ---------------------------------------------------------------------------------------- */
static inline REAL geom_max(REAL a, REAL b) { return a > b ? a : b; }
/* out3 = {depth_loss, normal_loss, number of masked pixels} */
void NAME(geom_loss_fwd)(int64_t P, const REAL *depth, int d_stride, const REAL *gt_depth, int gd_stride,
                         const REAL *normal, int n_pstride, int n_cstride, const REAL *gt_normal, int g_pstride,
                         int g_cstride, const uint8_t *mask, REAL *out3) {
    const REAL eps = (REAL)1e-12;
    double l1 = 0, mse = 0, cs = 0, cnt = 0;
    for (int64_t p = 0; p < P; ++p) {
        if (mask && !mask[p]) continue;
        l1 += fabs((double)(depth[p * d_stride] - gt_depth[p * gd_stride]));
        REAL dot = 0, sa = 0, sb = 0;
        for (int c = 0; c < 3; ++c) {
            const REAL u = normal[p * n_pstride + c * n_cstride], v = gt_normal[p * g_pstride + c * g_cstride];
            const REAL d = u - v;
            mse += (double)(d * d);
            dot = FMA(u, v, dot); sa = FMA(u, u, sa); sb = FMA(v, v, sb);
        }
        const REAL n1 = (REAL)sqrt((double)sa), n2 = (REAL)sqrt((double)sb);
        cs += (double)(dot / (geom_max(n1, eps) * geom_max(n2, eps)));
        cnt += 1;
    }
    out3[0] = (REAL)(l1 / cnt);
    out3[1] = (REAL)0.5 * (REAL)(mse / (3.0 * cnt)) + (REAL)0.5 * ((REAL)1 - (REAL)(cs / cnt));
    out3[2] = (REAL)cnt;
}
/* v_depth (P,), v_normal (P, 3) dense; zero outside the mask */
void NAME(geom_loss_bwd)(int64_t P, const REAL *depth, int d_stride, const REAL *gt_depth, int gd_stride,
                         const REAL *normal, int n_pstride, int n_cstride, const REAL *gt_normal, int g_pstride,
                         int g_cstride, const uint8_t *mask, REAL v_depth_loss, REAL v_normal_loss, REAL *v_depth,
                         REAL *v_normal) {
    const REAL eps = (REAL)1e-12;
    double cnt = 0;
    for (int64_t p = 0; p < P; ++p)
        if (!mask || mask[p]) cnt += 1;
    const REAL M = (REAL)cnt;
    const REAL kd = v_depth_loss / M;
    const REAL km = (REAL)0.5 * v_normal_loss * 2 / (3 * M);       /* d 0.5 mse */
    const REAL kc = -((REAL)0.5 * v_normal_loss) / M;              /* d 0.5 (1 - mean sim) */
    for (int64_t p = 0; p < P; ++p) {
        if (mask && !mask[p]) {
            v_depth[p] = 0;
            for (int c = 0; c < 3; ++c) v_normal[3 * p + c] = 0;
            continue;
        }
        const REAL d = depth[p * d_stride] - gt_depth[p * gd_stride];
        v_depth[p] = kd * (d > 0 ? (REAL)1 : (d < 0 ? (REAL)-1 : (REAL)0));
        REAL dot = 0, sa = 0, sb = 0, u3[3], v3[3];
        for (int c = 0; c < 3; ++c) {
            u3[c] = normal[p * n_pstride + c * n_cstride];
            v3[c] = gt_normal[p * g_pstride + c * g_cstride];
            dot = FMA(u3[c], v3[c], dot); sa = FMA(u3[c], u3[c], sa); sb = FMA(v3[c], v3[c], sb);
        }
        const REAL n1 = (REAL)sqrt((double)sa), n2 = (REAL)sqrt((double)sb);
        const REAL ca = geom_max(n1, eps), cb = geom_max(n2, eps);
        const REAL sim = dot / (ca * cb);
        for (int c = 0; c < 3; ++c) {
            const REAL ua = u3[c] / ca, vb = v3[c] / cb;
            const REAL dsim = (vb - (n1 > eps ? sim * ua : 0)) / ca;      /* as cosine_loss_bwd */
            v_normal[3 * p + c] = km * (u3[c] - v3[c]) + kc * dsim;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Binning — gsplat rasterize.py compute_cumulative_intersects + bin_and_sort_gaussians (†),
 * forward.cu map_gaussian_to_intersects / get_tile_bin_edges, torch.sort on the int64 keys;
 * run inside every Rasterize*.forward (gaussian_splatting.py:735,747,759,773).
 * ---------------------------------------------------------------------------------------- */
int64_t NAME(cumsum)(int N, const int32_t *num_tiles_hit, int32_t *cum) {
    int64_t acc = 0;
    for (int i = 0; i < N; ++i) {
        acc += num_tiles_hit[i];
        cum[i] = (int32_t)acc;
    }
    return acc;
}

void NAME(map_intersects)(int N, const REAL *xys, const REAL *depths, const int32_t *radii,
                          const int32_t *cum, int tiles_x, int tiles_y, int64_t *isect_ids,
                          int32_t *gaussian_ids) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (radii[i] <= 0) continue;
        int x0, y0, x1, y1;
        tile_bbox(xys[2 * i], xys[2 * i + 1], (REAL)radii[i], tiles_x, tiles_y, &x0, &y0, &x1,
                  &y1);
        int64_t cur = (i == 0) ? 0 : cum[i - 1];
        union {
            float f;
            int32_t i;
        } d;
        d.f = (float)depths[i];
        int64_t depth_id = (int64_t)(uint32_t)d.i; /* depth > 0: sign bit clear */
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                int64_t tile = (int64_t)ty * tiles_x + tx;
                isect_ids[cur] = (tile << 32) | depth_id;
                gaussian_ids[cur] = i;
                ++cur;
            }
    }
}

/* Ascending sort of the keys; ties keep emission order = ascending Gaussian id (the reference's
 * torch.sort does not promise a tie order; SURVEY a7 fixes this one).  Stable LSD radix. */
void NAME(sort_intersects)(int64_t I, const int64_t *keys_in, const int32_t *ids_in,
                           int64_t *keys_out, int32_t *ids_out) {
    if (I <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)I);
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)I);
    int32_t *va = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    int32_t *vb = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    memcpy(ka, keys_in, sizeof(uint64_t) * (size_t)I);
    memcpy(va, ids_in, sizeof(int32_t) * (size_t)I);
    for (int pass = 0; pass < 8; ++pass) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int64_t i = 0; i < I; ++i) cnt[((ka[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < I; ++i) {
            size_t dst = cnt[(ka[i] >> sh) & 255]++;
            kb[dst] = ka[i];
            vb[dst] = va[i];
        }
        uint64_t *tk = ka;
        ka = kb;
        kb = tk;
        int32_t *tv = va;
        va = vb;
        vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * (size_t)I);
    memcpy(ids_out, va, sizeof(int32_t) * (size_t)I);
    free(ka);
    free(kb);
    free(va);
    free(vb);
}

void NAME(tile_bins)(int64_t I, const int64_t *keys_sorted, int num_tiles, int32_t *tile_bins) {
    memset(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)num_tiles);
    for (int64_t i = 0; i < I; ++i) {
        int32_t cur = (int32_t)(keys_sorted[i] >> 32);
        if (i == 0) tile_bins[2 * cur] = 0;
        if (i == I - 1) tile_bins[2 * cur + 1] = (int32_t)I;
        if (i > 0) {
            int32_t prev = (int32_t)(keys_sorted[i - 1] >> 32);
            if (prev != cur) {
                tile_bins[2 * prev + 1] = (int32_t)i;
                tile_bins[2 * cur] = (int32_t)i;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * blend_fwd — gsplat forward.cu rasterize_forward (C=3, gaussian_splatting.py:735,759,773) and
 * nd_rasterize_forward (runtime C, :747) (†).  One image pixel at a time, walking its tile's
 * depth-sorted list front to back.  Pixel centre = integer (j, i); the -0.5 lives in xys.
 * final_idx[p] = one past the list position of the last Gaussian blended into p (range start
 * if none); the backward walks [range start, final_idx) in reverse, i.e. exactly the blended
 * Gaussians.  (gsplat stores an equivalent cursor; the value never leaves the operator.)
 * ---------------------------------------------------------------------------------------- */
void NAME(blend_fwd)(int C, int img_h, int img_w, int tiles_x, int tiles_y,
                     const int32_t *ids_sorted, const int32_t *tile_bins, const REAL *xys,
                     const REAL *conics, const REAL *colors, const REAL *opacity,
                     const REAL *background, REAL *out_img, REAL *final_T, int32_t *final_idx) {
    int num_tiles = tiles_x * tiles_y;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < num_tiles; ++tile) {
        int ty = tile / tiles_x, tx = tile % tiles_x;
        int r0 = tile_bins[2 * tile], r1 = tile_bins[2 * tile + 1];
        REAL *acc = (REAL *)malloc(sizeof(REAL) * (size_t)C);
        for (int ly = 0; ly < GG_BLOCK; ++ly)
            for (int lx = 0; lx < GG_BLOCK; ++lx) {
                int i = ty * GG_BLOCK + ly, j = tx * GG_BLOCK + lx;
                if (i >= img_h || j >= img_w) continue;
                REAL px = (REAL)j, py = (REAL)i;
                REAL T = R_(1.0f);
                int last = r0;
                for (int c = 0; c < C; ++c) acc[c] = 0;
                for (int idx = r0; idx < r1; ++idx) {
                    int g = ids_sorted[idx];
                    REAL dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
                    REAL ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    REAL sigma = FMA(R_(0.5f), FMA(ca * dx, dx, (cc * dy) * dy), (cb * dx) * dy);
                    if (sigma < R_(0.0f)) continue;
                    REAL alpha = FMIN(R_(GG_ALPHA_MAX_FWD), opacity[g] * gg_exp(-sigma));
                    if (alpha < R_(GG_ALPHA_MIN)) continue;
                    REAL next_T = T * (R_(1.0f) - alpha);
                    if (next_T <= R_(GG_T_EPS)) break;
                    REAL vis = alpha * T;
                    const REAL *col = colors + (size_t)g * C;
                    for (int c = 0; c < C; ++c) acc[c] = FMA(col[c], vis, acc[c]);
                    T = next_T;
                    last = idx + 1;
                }
                size_t p = (size_t)i * img_w + j;
                final_T[p] = T;
                final_idx[p] = last;
                for (int c = 0; c < C; ++c) out_img[p * C + c] = FMA(T, background[c], acc[c]);
            }
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------------
 * blend_bwd — gsplat backward.cu rasterize_backward_kernel / nd_rasterize_backward_kernel (†),
 * autograd of the four rasterize calls.  Per-pixel terms are computed in REAL exactly as a11
 * lists them; the sums over pixels (gsplat: atomicAdd) are accumulated in double per
 * (tile, list entry) in a fixed pixel order and then added per Gaussian in list order, so the
 * oracle is deterministic and carries no fp32 summation-order noise.
 * v_conic[.,1] carries gsplat's 0.5 factor (gradient of the symmetric matrix entry).
 * ---------------------------------------------------------------------------------------- */
void NAME(blend_bwd)(int C, int N, int img_h, int img_w, int tiles_x, int tiles_y,
                     const int32_t *ids_sorted, const int32_t *tile_bins, const REAL *xys,
                     const REAL *conics, const REAL *colors, const REAL *opacity,
                     const REAL *background, const REAL *final_T, const int32_t *final_idx,
                     const REAL *v_out, REAL *v_xy, REAL *v_conic, REAL *v_colors,
                     REAL *v_opacity) {
    int num_tiles = tiles_x * tiles_y;
    int64_t I = 0;
    for (int t = 0; t < num_tiles; ++t)
        if (tile_bins[2 * t + 1] > I) I = tile_bins[2 * t + 1];
    const int S = C + 6; /* per-entry partials: C colour, xy(2), conic(3), opacity(1) */
    double *slab = (double *)calloc((size_t)(I > 0 ? I : 1) * S, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < num_tiles; ++tile) {
        int ty = tile / tiles_x, tx = tile % tiles_x;
        int r0 = tile_bins[2 * tile];
        REAL *Sc = (REAL *)malloc(sizeof(REAL) * (size_t)C);
        for (int ly = 0; ly < GG_BLOCK; ++ly)
            for (int lx = 0; lx < GG_BLOCK; ++lx) {
                int i = ty * GG_BLOCK + ly, j = tx * GG_BLOCK + lx;
                if (i >= img_h || j >= img_w) continue;
                size_t p = (size_t)i * img_w + j;
                REAL px = (REAL)j, py = (REAL)i;
                REAL T_final = final_T[p];
                REAL T = T_final;
                const REAL *vo = v_out + p * C;
                for (int c = 0; c < C; ++c) Sc[c] = 0;
                for (int idx = final_idx[p] - 1; idx >= r0; --idx) {
                    int g = ids_sorted[idx];
                    REAL dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
                    REAL ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    REAL sigma = FMA(R_(0.5f), FMA(ca * dx, dx, (cc * dy) * dy), (cb * dx) * dy);
                    if (sigma < R_(0.0f)) continue;
                    REAL opac = opacity[g];
                    REAL vis = gg_exp(-sigma);
                    REAL alpha = FMIN(R_(GG_ALPHA_MAX_BWD), opac * vis);
                    if (alpha < R_(GG_ALPHA_MIN)) continue;
                    REAL ra = R_(1.0f) / (R_(1.0f) - alpha);
                    T = T * ra;
                    REAL fac = alpha * T;
                    REAL v_alpha = 0;
                    const REAL *col = colors + (size_t)g * C;
                    double *sl = slab + (size_t)idx * S;
                    for (int c = 0; c < C; ++c) {
                        sl[c] += (double)(fac * vo[c]);
                        v_alpha = FMA(col[c] * T - Sc[c] * ra, vo[c], v_alpha);
                        v_alpha = FMA(((-T_final) * ra) * background[c], vo[c], v_alpha);
                        Sc[c] = FMA(col[c], fac, Sc[c]);
                    }
                    REAL v_sigma = ((-opac) * vis) * v_alpha;
                    sl[C + 0] += (double)(v_sigma * (ca * dx + cb * dy));
                    sl[C + 1] += (double)(v_sigma * (cb * dx + cc * dy));
                    sl[C + 2] += (double)(((R_(0.5f) * v_sigma) * dx) * dx);
                    sl[C + 3] += (double)(((R_(0.5f) * v_sigma) * dx) * dy);
                    sl[C + 4] += (double)(((R_(0.5f) * v_sigma) * dy) * dy);
                    sl[C + 5] += (double)(vis * v_alpha);
                }
            }
        free(Sc);
    }
    /* per-Gaussian reduction, list order */
    double *acc = (double *)calloc((size_t)(N > 0 ? N : 1) * S, sizeof(double));
    /* every thread scans the whole list and adds the entries of "its" Gaussians (g mod nthreads):
     * each Gaussian's partials are still added in list order, whatever the thread count */
#pragma omp parallel
    {
        int nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads();
        tid = omp_get_thread_num();
#endif
        for (int64_t idx = 0; idx < I; ++idx) {
            int g = ids_sorted[idx];
            if (g % nt != tid) continue;
            for (int k = 0; k < S; ++k) acc[(size_t)g * S + k] += slab[(size_t)idx * S + k];
        }
    }
    for (int g = 0; g < N; ++g) {
        const double *a = acc + (size_t)g * S;
        for (int c = 0; c < C; ++c) v_colors[(size_t)g * C + c] = (REAL)a[c];
        v_xy[2 * g] = (REAL)a[C];
        v_xy[2 * g + 1] = (REAL)a[C + 1];
        v_conic[3 * g] = (REAL)a[C + 2];
        v_conic[3 * g + 1] = (REAL)a[C + 3];
        v_conic[3 * g + 2] = (REAL)a[C + 4];
        v_opacity[g] = (REAL)a[C + 5];
    }
    free(acc);
    free(slab);
}

/* ------------------------------------------------------------------------------------------
 * mlp_fwd — restates the reference's `MLP(in, out, hidden_list=[128])` forward
 * (nerfstudio/models/gaussian_splatting.py:198-213: Linear(in,128) -> ReLU -> Linear(128,out);
 * `self.fea_up` at :258, applied per pixel at nerfstudio/pipelines/base_pipeline.py:408 and to the
 * sampled points at :917).  x (P,in), w1 (128,in), b1 (128), w2 (out,128), b2 (out) -> y (P,out).
 * torch fixes no summation order for a Linear; this one is the HIP kernel's (csrc/mlp.hip): the
 * accumulator starts at the bias, layer 1 adds k = 0,1,2,...; layer 2 adds the hidden units in the
 * order the matrix pipe holds them: step (blk, r) = hidden 32 blk + (r&3) + 8 (r>>2), then that + 4.
 * ---------------------------------------------------------------------------------------- */
void NAME(mlp_fwd)(int64_t P, int in_dim, int out_dim, const REAL *x, const REAL *w1, const REAL *b1,
                   const REAL *w2, const REAL *b2, REAL *y) {
    enum { HID = 128 };
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        REAL h[HID];
        const REAL *xp = x + (size_t)p * in_dim;
        for (int j = 0; j < HID; ++j) {
            REAL acc = b1[j];
            for (int k = 0; k < in_dim; ++k) acc = FMA(w1[(size_t)j * in_dim + k], xp[k], acc);
            h[j] = FMAX(acc, R_(0.0f));
        }
        for (int o = 0; o < out_dim; ++o) {
            REAL acc = b2[o];
            const REAL *wr = w2 + (size_t)o * HID;
            for (int step = 0; step < HID / 2; ++step) {
                const int blk = step >> 4, r = step & 15;
                const int ha = 32 * blk + (r & 3) + 8 * (r >> 2), hb = ha + 4;
                acc = FMA(wr[ha], h[ha], acc);
                acc = FMA(wr[hb], h[hb], acc);
            }
            y[(size_t)p * out_dim + o] = acc;
        }
    }
}

/* Number of host threads the parallel loops above will use (bench.py cpu_baseline.cores). */
#ifdef _OPENMP
int NAME(num_threads)(void) { return omp_get_max_threads(); }
void NAME(set_num_threads)(int n) { omp_set_num_threads(n); }
#else
int NAME(num_threads)(void) { return 1; }
void NAME(set_num_threads)(int n) { (void)n; }
#endif

/* ==========================================================================================
 * densification statistics / masks, cull compaction, split / dup row append and the
 * Adam step.  Unlike the rasterizer these rows ARE pinnable: the reference does them with plain torch
 * (gaussian_splatting.py:333-393,402-546; engine/optimizers.py:158-171 -> torch.optim.Adam), which is
 * importable here; tests/test_densify_adam.py pins every function below against that torch code.
 * ======================================================================================== */

/* torch.optim.Adam (single-tensor path, amsgrad off), one parameter array:
 *   exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
 *   step_size = lr / (1-beta1^t); denom = sqrt(exp_avg_sq) / sqrt(1-beta2^t) + eps
 *   param.addcdiv_(exp_avg, denom, value=-step_size)
 * bias corrections in double (python floats), rounded to REAL where torch hands them to a kernel. */
void NAME(adam_step)(int64_t numel, REAL *p, const REAL *g_in, REAL *m, REAL *v, double lr, double beta1,
                     double beta2, double eps, double weight_decay, int64_t step) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const REAL step_size = (REAL)(lr / bc1), bc2s = (REAL)sqrt(bc2);
    const REAL w1 = (REAL)(1.0 - beta1), w2 = (REAL)(1.0 - beta2), b2 = (REAL)beta2, e = (REAL)eps,
               wd = (REAL)weight_decay;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < numel; ++i) {
        REAL g = g_in[i];
        if (wd != R_(0.0f)) g = FMA(wd, p[i], g);
        const REAL mi = FMA(w1, g - m[i], m[i]);
        const REAL vi = FMA(w2 * g, g, v[i] * b2);
        const REAL denom = SQRT(vi) / bc2s + e;
        p[i] = p[i] + (-step_size * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}

/* ranks[i] = number of selected rows before i; returns the number selected */
int64_t NAME(mask_scan)(int N, const uint8_t *mask, int invert, int32_t *ranks) {
    int64_t run = 0;
    for (int i = 0; i < N; ++i) {
        ranks[i] = (int32_t)run;
        run += ((mask[i] != 0) != (invert != 0)) ? 1 : 0;
    }
    return run;
}

/* t[~deleted] (cull_gaussians :497-502, remove_from_optim :341-342): kept rows move to the front */
int64_t NAME(compact_rows)(int N, const uint8_t *deleted, int w, const REAL *src, REAL *dst) {
    int64_t k = 0;
    for (int i = 0; i < N; ++i) {
        if (deleted[i]) continue;
        memcpy(dst + (size_t)k * w, src + (size_t)i * w, sizeof(REAL) * (size_t)w);
        ++k;
    }
    return k;
}

/* torch.cat([old, split samples (sample-major), dups]) of one array (refinement_after :434-439,
 * split_gaussians :504-531, dup_gaussians :533-546, dup_in_optim :352-371); kinds as gg_raster.h */
void NAME(densify_rows)(int N, const uint8_t *split_mask, const uint8_t *dup_mask, int nsamps,
                        const REAL *samples, REAL size_fac, const REAL *means, const REAL *scales,
                        const REAL *quats, int w, int kind, const REAL *src, REAL *dst) {
    int n_split = 0, n_dup = 0;
    for (int i = 0; i < N; ++i) {
        n_split += split_mask && split_mask[i] ? 1 : 0;
        n_dup += dup_mask && dup_mask[i] ? 1 : 0;
    }
    int sr = 0, dr = 0;
    for (int i = 0; i < N; ++i) {
        const int is_split = split_mask && split_mask[i], is_dup = dup_mask && dup_mask[i];
        for (int c = 0; c < w; ++c) {
            const REAL val = src[(size_t)i * w + c];
            REAL old_v = val, split_v = val;
            if (kind == 2 && is_split) {
#ifdef GGO_F64
                old_v = split_v = log(exp(val) / size_fac);
#else
                old_v = split_v = logf(expf(val) / size_fac);
#endif
            }
            if (kind == 3) split_v = R_(0.0f);
            dst[(size_t)i * w + c] = old_v;
            if (is_split) {
                for (int s = 0; s < nsamps; ++s) {
                    const size_t row = (size_t)N + (size_t)s * n_split + sr;
                    REAL out = split_v;
                    if (kind == 1) {
                        const REAL *q = quats + 4 * (size_t)i;
                        REAL n = SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                        REAL qw = q[0] / n, qx = q[1] / n, qy = q[2] / n, qz = q[3] / n;
                        const REAL n2 = FMAX(SQRT(qw * qw + qx * qx + qy * qy + qz * qz), R_(GG_QUAT_NORM_EPS));
                        qw /= n2; qx /= n2; qy /= n2; qz /= n2;
                        REAL Rm[9];
                        Rm[0] = R_(1.f) - R_(2.f) * (qy * qy + qz * qz); Rm[1] = R_(2.f) * (qx * qy - qw * qz);
                        Rm[2] = R_(2.f) * (qx * qz + qw * qy); Rm[3] = R_(2.f) * (qx * qy + qw * qz);
                        Rm[4] = R_(1.f) - R_(2.f) * (qx * qx + qz * qz); Rm[5] = R_(2.f) * (qy * qz - qw * qx);
                        Rm[6] = R_(2.f) * (qx * qz - qw * qy); Rm[7] = R_(2.f) * (qy * qz + qw * qx);
                        Rm[8] = R_(1.f) - R_(2.f) * (qx * qx + qy * qy);
                        const REAL *z = samples + 3 * ((size_t)s * n_split + sr);
                        REAL sc[3];
                        for (int k = 0; k < 3; ++k) {
#ifdef GGO_F64
                            sc[k] = exp(scales[3 * (size_t)i + k]) * z[k];
#else
                            sc[k] = expf(scales[3 * (size_t)i + k]) * z[k];
#endif
                        }
                        out = ((Rm[3 * c] * sc[0] + Rm[3 * c + 1] * sc[1]) + Rm[3 * c + 2] * sc[2]) +
                              means[3 * (size_t)i + c];
                    }
                    dst[row * w + c] = out;
                }
            }
            if (is_dup)   /* dup_gaussians reads self.scales after split_gaussians shrank the split rows (:524-526, :541) */
                dst[((size_t)N + (size_t)nsamps * n_split + dr) * w + c] = (kind == 3) ? R_(0.0f) : old_v;
        }
        sr += is_split;
        dr += is_dup;
    }
}

static REAL ggo_max_exp3(const REAL *s) {
#ifdef GGO_F64
    return FMAX(FMAX(exp(s[0]), exp(s[1])), exp(s[2]));
#else
    return FMAX(FMAX(expf(s[0]), expf(s[1])), expf(s[2]));
#endif
}

/* after_train (:373-393) */
void NAME(densify_stats)(int N, const REAL *xys_grad, const int32_t *radii, int max_dim, int first,
                         REAL *grad_norm, REAL *vis_counts, REAL *max_2dsize) {
    for (int i = 0; i < N; ++i) {
        const REAL gx = xys_grad[2 * (size_t)i], gy = xys_grad[2 * (size_t)i + 1];
        const REAL g = SQRT(gx * gx + gy * gy);
        const int vis = radii[i] > 0;
        if (first) {
            grad_norm[i] = g;
            vis_counts[i] = R_(1.0f);
            max_2dsize[i] = R_(0.0f);
        } else if (vis) {
            vis_counts[i] = vis_counts[i] + R_(1.0f);
            grad_norm[i] = g + grad_norm[i];
        }
        if (vis) max_2dsize[i] = FMAX(max_2dsize[i], (REAL)radii[i] / (REAL)max_dim);
    }
}

/* refinement_after (:412-421, :430-431).  The order of the reference's statements matters: split_gaussians
 * (:423-429) shrinks self.scales[split_mask] in place (:524-526) BEFORE `dups = scales.exp().max() <= thresh`
 * (:430), so the duplicate test of a split Gaussian runs on log(exp(s) / size_fac). */
void NAME(densify_masks)(int N, const REAL *grad_norm, const REAL *vis_counts, const REAL *max_2dsize,
                         const REAL *scales, int max_dim, REAL grad_thresh, REAL size_thresh,
                         REAL split_screen_size, int use_screen, REAL size_fac, uint8_t *split_mask,
                         uint8_t *dup_mask) {
    for (int i = 0; i < N; ++i) {
        const REAL avg = ((grad_norm[i] / vis_counts[i]) * R_(0.5f)) * (REAL)max_dim;
        const int high = avg > grad_thresh;
        const REAL *sc = scales + 3 * (size_t)i;
        REAL smax = ggo_max_exp3(sc);
        int split = smax > size_thresh;
        if (use_screen) split = split || (max_2dsize[i] > split_screen_size);
        split = split && high;
        split_mask[i] = (uint8_t)split;
        if (split) {
            REAL sh[3];
            for (int k = 0; k < 3; ++k) {
#ifdef GGO_F64
                sh[k] = log(exp(sc[k]) / size_fac);
#else
                sh[k] = logf(expf(sc[k]) / size_fac);
#endif
            }
            smax = ggo_max_exp3(sh);
        }
        dup_mask[i] = (uint8_t)((smax <= size_thresh) && high);
    }
}

/* cull_gaussians (:485-496) */
void NAME(cull_mask)(int N, const REAL *opacities, const REAL *scales, const REAL *max_2dsize,
                     REAL alpha_thresh, REAL scale_thresh, REAL screen_thresh, int use_scale, int use_screen,
                     uint8_t *mask) {
    for (int i = 0; i < N; ++i) {
#ifdef GGO_F64
        const REAL sig = 1.0 / (1.0 + exp(-opacities[i]));
#else
        const REAL sig = 1.0f / (1.0f + expf(-opacities[i]));
#endif
        int cull = sig < alpha_thresh;
        if (use_scale) {
            cull = cull || (ggo_max_exp3(scales + 3 * (size_t)i) > scale_thresh);
            if (use_screen) cull = cull || (max_2dsize[i] > screen_thresh);
        }
        mask[i] = (uint8_t)cull;
    }
}

/* ==========================================================================================
 * SURVEY 8f-2, second half: the backward of the fea_up MLP and the cosine-similarity loss.
 * Pinned against torch autograd of the published code (tests/test_mlp_losses.py).
 * ======================================================================================== */

/* VJP of y = W2 relu(W1 x + b1) + b2 (reference MLP, gaussian_splatting.py:198-213), g = dL/dy (P,out):
 *   v_x (P,in), v_w1 (128,in), v_b1 (128), v_w2 (out,128), v_b2 (out); parameter gradients are summed
 *   over the rows in row order (doubles in both builds: the GPU's atomics give another order anyway). */
void NAME(mlp_bwd)(int64_t P, int in_dim, int out_dim, const REAL *x, const REAL *w1, const REAL *b1,
                   const REAL *w2, const REAL *g, REAL *v_x, REAL *v_w1, REAL *v_b1, REAL *v_w2, REAL *v_b2) {
    enum { HID = 128 };
    double *aw1 = (double *)calloc((size_t)HID * in_dim, sizeof(double));
    double *aw2 = (double *)calloc((size_t)out_dim * HID, sizeof(double));
    double *ab1 = (double *)calloc(HID, sizeof(double)), *ab2 = (double *)calloc((size_t)out_dim, sizeof(double));
    for (int64_t p = 0; p < P; ++p) {
        const REAL *xp = x + (size_t)p * in_dim, *gp = g + (size_t)p * out_dim;
        REAL h[HID], gh[HID];
        for (int j = 0; j < HID; ++j) {
            REAL acc = b1[j];
            for (int k = 0; k < in_dim; ++k) acc = FMA(w1[(size_t)j * in_dim + k], xp[k], acc);
            h[j] = acc;
        }
        for (int j = 0; j < HID; ++j) {
            REAL acc = R_(0.0f);
            for (int o = 0; o < out_dim; ++o) acc = FMA(gp[o], w2[(size_t)o * HID + j], acc);
            gh[j] = h[j] > R_(0.0f) ? acc : R_(0.0f);
        }
        for (int k = 0; k < in_dim; ++k) {
            REAL acc = R_(0.0f);
            for (int j = 0; j < HID; ++j) acc = FMA(gh[j], w1[(size_t)j * in_dim + k], acc);
            v_x[(size_t)p * in_dim + k] = acc;
        }
        for (int j = 0; j < HID; ++j) {
            ab1[j] += (double)gh[j];
            for (int k = 0; k < in_dim; ++k) aw1[(size_t)j * in_dim + k] += (double)gh[j] * (double)xp[k];
        }
        for (int o = 0; o < out_dim; ++o) {
            ab2[o] += (double)gp[o];
            for (int j = 0; j < HID; ++j)
                aw2[(size_t)o * HID + j] += (double)gp[o] * (double)(h[j] > R_(0.0f) ? h[j] : R_(0.0f));
        }
    }
    for (int i = 0; i < HID * in_dim; ++i) v_w1[i] = (REAL)aw1[i];
    for (int i = 0; i < out_dim * HID; ++i) v_w2[i] = (REAL)aw2[i];
    for (int i = 0; i < HID; ++i) v_b1[i] = (REAL)ab1[i];
    for (int i = 0; i < out_dim; ++i) v_b2[i] = (REAL)ab2[i];
    free(aw1); free(aw2); free(ab1); free(ab2);
}

/* cosine_similarity_loss (gaussian_splatting.py:113-118) on M points of C channels, a and b stored
 * (M,C) — the reference passes the (C,M) transposes and normalises along dim 0, i.e. per point:
 *   sim_m = <a_m, b_m> / (max(|a_m|, 1e-12) max(|b_m|, 1e-12)),  loss = 1 - mean_m sim_m.
 * Saves sim, |a|, |b| for the backward. */
REAL NAME(cosine_loss_fwd)(int64_t M, int C, const REAL *a, const REAL *b, REAL *sim, REAL *na, REAL *nb) {
    double total = 0.0;
    for (int64_t m = 0; m < M; ++m) {
        REAL dot = R_(0.0f), sa = R_(0.0f), sb = R_(0.0f);
        for (int c = 0; c < C; ++c) {
            const REAL u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
            dot = FMA(u, v, dot);
            sa = FMA(u, u, sa);
            sb = FMA(v, v, sb);
        }
        na[m] = SQRT(sa);
        nb[m] = SQRT(sb);
        sim[m] = dot / (FMAX(na[m], R_(1e-12f)) * FMAX(nb[m], R_(1e-12f)));
        total += (double)sim[m];
    }
    return (REAL)(1.0 - total / (double)(M > 0 ? M : 1));
}

/* v_a, v_b for v_loss = dL/d loss */
void NAME(cosine_loss_bwd)(int64_t M, int C, const REAL *a, const REAL *b, const REAL *sim, const REAL *na,
                           const REAL *nb, REAL v_loss, REAL *v_a, REAL *v_b) {
    const REAL s = -v_loss / (REAL)(M > 0 ? M : 1);
    for (int64_t m = 0; m < M; ++m) {
        const REAL ca = FMAX(na[m], R_(1e-12f)), cb = FMAX(nb[m], R_(1e-12f));
        const int fa = na[m] > R_(1e-12f), fb = nb[m] > R_(1e-12f);   /* clamp active: norm is a constant */
        for (int c = 0; c < C; ++c) {
            const REAL u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
            const REAL ua = u / ca, vb = v / cb;
            v_a[(size_t)m * C + c] = s * (vb - (fa ? sim[m] * ua : R_(0.0f))) / ca;
            v_b[(size_t)m * C + c] = s * (ua - (fb ? sim[m] * vb : R_(0.0f))) / cb;
        }
    }
}
