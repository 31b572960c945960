/*
 * gg_constants.h — every †UNVERIFIED constant of the gsplat-0.1.0 rasterizer semantics
 * (SURVEY.md Appendix B), in ONE place.  Included by the HIP kernels
 * (gaussiangrasper_amd/csrc/) and by the CPU oracle (oracle/gg_oracle.c); mirrored
 * for Python in gaussiangrasper_amd/constants.py (tests check the two agree).
 *
 * The reference (leejaehot/GaussianGrasper) reaches this arithmetic only through the
 * un-vendored dependency gsplat==0.1.0 (reference requirements.txt:78, call sites
 * nerfstudio/models/gaussian_splatting.py:699-784).  The gsplat source is not
 * available offline, so each value below is a recollection of the public v0.1.0
 * source; a later session holding that source flips it here and nowhere else.
 */
#ifndef GG_CONSTANTS_H
#define GG_CONSTANTS_H

/* ---- projection ---- */
#define GG_CLIP_THRESH_DEFAULT 0.01f /* cull if view-space z <= clip_thresh            */
#define GG_BLUR 0.3f                 /* added to both diagonal terms of cov2d            */
#define GG_FOV_LIM 1.3f              /* clamp t.x/t.z to +-1.3*tan(fov/2) in the EWA J   */
#define GG_RADIUS_SIGMA 3.0f         /* radius = ceil(3*sqrt(lambda_max))                */
#define GG_EIG_FLOOR 0.1f            /* max(0.1, b^2-det) under the eigenvalue sqrt      */
#define GG_W_EPS 1e-6f               /* rw = 1/(w_clip + 1e-6)                           */
#define GG_PIX_OFFSET 0.5f           /* xy = 0.5*W*ndc + cx - 0.5 ; pixel centres integer */
#define GG_QUAT_NORM_EPS 1e-12f     /* quat_to_rotmat: q / max(|q|, eps) (F.normalize default) */
#define GG_BLOCK 16                  /* tile edge in pixels (BLOCK_X = BLOCK_Y)          */

/* ---- blending ---- */
#define GG_ALPHA_MAX_FWD 0.999f
#ifndef GG_ALPHA_MAX_BWD /* -DGG_ALPHA_MAX_BWD=0.99f: the "compat" variant build (PARITY.md) */
#define GG_ALPHA_MAX_BWD 0.999f /* †† least certain: early gsplat may have used 0.99 */
#endif
#define GG_ALPHA_MIN (1.0f / 255.0f)
#define GG_T_EPS 1e-4f /* stop when T*(1-alpha) <= 1e-4; that Gaussian is not blended */

/* ---- where gsplat 0.1.0 is RECALLED to deviate from exact calculus (PARITY.md) ----
 * 0 (default): exact VJPs of the forward.  1: the recalled gsplat-0.1.0 behaviour — project_pix_vjp
 * drops the homogeneous-w path, the EWA Jacobian VJP is taken at the UNCLAMPED view-space point, the
 * quaternion gradient is the one w.r.t. the normalised components.  The "compat" variant libraries
 * (libgg_raster_compat.so, libgg_oracle_compat_*.so) are built with 1 and GG_ALPHA_MAX_BWD=0.99f, and
 * tests/test_compat_variant.py holds them to each other, so that a later session holding the gsplat
 * source flips a constant here, not code. */
#ifndef GG_VJP_GSPLAT_COMPAT
#define GG_VJP_GSPLAT_COMPAT 0
#endif

/* ---- spherical harmonics (3DGS sign convention) ---- */
#define GG_SH_C0 0.28209479177387814f
#define GG_SH_C1 0.4886025119029199f
#define GG_SH_C2_0 1.0925484305920792f
#define GG_SH_C2_1 -1.0925484305920792f
#define GG_SH_C2_2 0.31539156525252005f
#define GG_SH_C2_3 -1.0925484305920792f
#define GG_SH_C2_4 0.5462742152960396f
#define GG_SH_C3_0 -0.5900435899266435f
#define GG_SH_C3_1 2.890611442640554f
#define GG_SH_C3_2 -0.4570457994644658f
#define GG_SH_C3_3 0.3731763325901154f
#define GG_SH_C3_4 -0.4570457994644658f
#define GG_SH_C3_5 1.445305721320277f
#define GG_SH_C3_6 -0.5900435899266435f
#define GG_SH_C4_0 2.5033429417967046f
#define GG_SH_C4_1 -1.7701307697799304f
#define GG_SH_C4_2 0.9461746957575601f
#define GG_SH_C4_3 -0.6690465435572892f
#define GG_SH_C4_4 0.10578554691520431f
#define GG_SH_C4_5 -0.6690465435572892f
#define GG_SH_C4_6 0.47308734787878004f
#define GG_SH_C4_7 -1.7701307697799304f
#define GG_SH_C4_8 0.6258357354491761f
#define GG_SH_MAX_BASES 25

/*
 * ---- gg_expf: the exponential of the blend, as an explicit fp32 operation sequence ----
 * gsplat calls the CUDA fast intrinsic __expf(-sigma); no two machines agree on its last
 * bits, and the alpha<1/255 and T<=1e-4 tests turn a 1-ulp difference into a 4e-3 jump.
 * Both the oracle and the HIP kernels therefore evaluate exp() by the SAME sequence of
 * correctly-rounded fp32 operations (mul, rint, fma, integer exponent insert), which
 * makes every skip/stop decision — and the whole forward image — bit-identical between
 * the CPU oracle and the GPU.  Cephes expf constants; |rel err| < 2 ulp on [-80, 0].
 *
 *   x < GG_EXP_LO            -> 0
 *   t = x * LOG2E ; n = rint(t)
 *   r = fma(n, -LN2_HI, x) ; r = fma(n, -LN2_LO, r)
 *   p = P0 ; p = fma(p,r,P1) ... fma(p,r,P5)
 *   y = fma(p, r*r, r) + 1
 *   return y * 2^n            (2^n built as the float with biased exponent n+127)
 */
#define GG_EXP_LO -80.0f
#define GG_EXP_LOG2E 1.44269504088896341f
#define GG_EXP_LN2_HI 0.693359375f
#define GG_EXP_LN2_LO -2.12194440e-4f
#define GG_EXP_P0 1.9875691500e-4f
#define GG_EXP_P1 1.3981999507e-3f
#define GG_EXP_P2 8.3334519073e-3f
#define GG_EXP_P3 4.1665795894e-2f
#define GG_EXP_P4 1.6666665459e-1f
#define GG_EXP_P5 5.0000001201e-1f

#endif /* GG_CONSTANTS_H */
