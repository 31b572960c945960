// blend.hip — alpha-blended splatting, forward and backward: the hot loop.
//
// Geometry.  One 256-thread workgroup per 16x16 tile; wave w owns the 8x8 pixel quadrant
// (w&1, w>>1) of the tile, one pixel per lane.  A tile's depth-sorted list is consumed in
// chunks of 64 entries: lane l gathers entry l's 32-byte record (xy, opacity, sigma cut-off,
// conic) and tests its alpha>=1/255 ellipse (axis-aligned extent) against the wave's quadrant;
// one ballot turns the survivors into a 64-bit mask that the wave walks in depth order with
// s_ff1 / s_flbit, broadcasting the record from the owning lane with v_readlane — the Gaussian
// lives in SGPRs, the pixels in VGPRs, no LDS and no barrier in the forward.  Colours are read
// through a wave-uniform pointer (scalar loads).
//
// Exactness.  The cull is conservative (margins below), and every surviving (pixel, Gaussian)
// pair runs the exact test sequence of the oracle — same sigma association, gg_expf, alpha
// clamp, 1/255 and 1e-4 tests, fmaf accumulation in list order — so out_img, final_Ts and
// final_idx are bit-identical to oracle/gg_oracle.c:blend_fwd.
//
// Backward.  Back-to-front over [range start, final_idx) with the transmittance recurrence
// T <- T/(1-alpha).  The colour suffix sums S_c of gsplat's formulation enter v_alpha only
// through <S, v_out>, so the kernel carries that scalar (W) instead of a C-vector:
//     D = <colour_g, v_out_p>,  v_alpha = T*D - ra*W,  W += D*alpha*T,  W0 = T_final*<bg, v_out_p>
// (algebraically the a11 formula; the oracle keeps the channel-wise form).  Per (wave, Gaussian)
// the C+6 partial gradients are reduced across the 64 lanes IN REGISTERS with a halving
// butterfly (v_permlane32_swap, v_permlane16_swap, then DPP row all-reduce), which leaves each
// value on its own lane; the four waves of a tile combine through an LDS slab (ds_add_f32) and
// the tile issues one global float-atomic wave-instruction per touched Gaussian per 64-entry
// batch.  Float atomics make the sums order-dependent in the last bits (tests use tolerances).
#include "gg_common.h"

struct __attribute__((aligned(16))) GRec {
    float x, y, opac, thr;  // thr: sigma above which alpha < 1/255 for certain (conservative)
    float ca, cb, cc, pad;
};

// ---------------------------------------------------------------------------------------------
// prep: pack xys / conics / opacity into 32-byte records (one gather per list entry later)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_prep_kernel(int N, const float *__restrict__ xys,
                                                         const float *__restrict__ conics,
                                                         const float *__restrict__ opacity,
                                                         GRec *__restrict__ rec) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    GRec r;
    r.x = xys[2 * (size_t)i];
    r.y = xys[2 * (size_t)i + 1];
    r.opac = opacity[i];
    r.ca = conics[3 * (size_t)i];
    r.cb = conics[3 * (size_t)i + 1];
    r.cc = conics[3 * (size_t)i + 2];
    r.pad = 0.0f;
    // alpha = opac*exp(-sigma) >= 1/255  <=>  sigma <= ln(255*opac).  Margins absorb the fp32
    // rounding of sigma (rel ~5e-7), of gg_expf (2 ulp) and of the fast log (1e-6).
    float t = __logf(255.0f * r.opac);
    t = t + 0.002f * fabsf(t) + 0.002f;
    if (!(r.opac > 0.0f)) t = -1.0f;                   // alpha <= 0 < 1/255 always
    if (r.opac != r.opac) t = __builtin_inff();         // NaN opacity: never cull (NaN propagates)
    r.thr = t;
    float4 *dst = reinterpret_cast<float4 *>(rec + i);
    dst[0] = make_float4(r.x, r.y, r.opac, r.thr);
    dst[1] = make_float4(r.ca, r.cb, r.cc, r.pad);
}

extern "C" size_t gg_blend_workspace(int num_points) {
    return gg_align_up(sizeof(GRec) * (size_t)(num_points > 0 ? num_points : 1), 256);
}

// tile index from block index: blocks are dealt round-robin to the 8 XCDs, so give each XCD a
// contiguous run of tiles (neighbouring tiles share Gaussians -> hits in that XCD's L2).
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
    int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// Does the alpha>=1/255 ellipse of a record reach the pixel rectangle [xlo,xhi]x[ylo,yhi]?
// Axis-aligned extent of {d : sigma(d) <= thr}: |dx| <= sqrt(2 thr cc/det), |dy| <= sqrt(2 thr ca/det).
__device__ __forceinline__ bool rec_hits_rect(const float4 a, const float4 b, float xlo, float xhi,
                                              float ylo, float yhi) {
    float thr = a.w;
    if (thr < 0.0f) return false;
    float det = b.x * b.z - b.y * b.y;
    float k = 2.0f * thr / det;
    float ex = sqrtf(k * b.z) * 1.001f + 0.01f;
    float ey = sqrtf(k * b.x) * 1.001f + 0.01f;
    // any NaN/inf (degenerate conic) -> comparisons below come out "hit" (conservative)
    bool miss = (a.x + ex < xlo) || (a.x - ex > xhi) || (a.y + ey < ylo) || (a.y - ey > yhi);
    return !miss;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int CH, bool FULL>
__global__ __launch_bounds__(256) void blend_fwd_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    float *__restrict__ out_img, float *__restrict__ final_T, int32_t *__restrict__ final_idx,
    int write_final) {
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const float xlo = (float)qx0, xhi = (float)(qx0 + 7), ylo = (float)qy0, yhi = (float)(qy0 + 7);
    const int2 range = bins[tile];

    float T = 1.0f;
    int last = range.x;
    bool done = !inside;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0f;

    for (int base = range.x; base < range.y; base += 64) {
        if (__ballot(!done) == 0ull) break;
        const int e = base + lane;
        const bool valid = e < range.y;
        const int g = valid ? ids[e] : 0;
        const float4 ra = reinterpret_cast<const float4 *>(rec + g)[0];
        const float4 rb = reinterpret_cast<const float4 *>(rec + g)[1];
        uint64_t m = __ballot(valid && rec_hits_rect(ra, rb, xlo, xhi, ylo, yhi));
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.x), src));
            const float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.y), src));
            const float gop = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.z), src));
            const float gthr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.w), src));
            const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.x), src));
            const float cb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.y), src));
            const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.z), src));
            const int gid = __builtin_amdgcn_readlane(g, src);
            const float dx = gx - px, dy = gy - py;
            const float sigma =
                __builtin_fmaf(0.5f, __builtin_fmaf(ca * dx, dx, (cc * dy) * dy), (cb * dx) * dy);
            // cheap conservative pre-test: nobody in the wave can reach alpha >= 1/255
            if (__ballot(!done && sigma >= 0.0f && sigma <= gthr) == 0ull) continue;
            const float alpha = fminf(GG_ALPHA_MAX_FWD, gop * gg_expf(-sigma));
            const bool pass = !done && sigma >= 0.0f && !(alpha < GG_ALPHA_MIN);
            const float next_T = T * (1.0f - alpha);
            const bool stop = pass && (next_T <= GG_T_EPS);
            const bool blend = pass && !stop;
            if (__ballot(blend) != 0ull) {
                const float *col = colors + (size_t)gid * C + ch_off;
                float cv[CH];  // wave-uniform -> one wide scalar load
#pragma unroll
                for (int c = 0; c < CH; ++c) cv[c] = (FULL || c < nch) ? col[c] : 0.0f;
                const float vis = alpha * T;
                if (blend) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[c] = __builtin_fmaf(cv[c], vis, acc[c]);
                    T = next_T;
                    last = base + src + 1;
                }
            }
            if (stop) done = true;
            if (__ballot(!done) == 0ull) break;
        }
    }
    if (inside) {
        const size_t p = (size_t)i * img_w + j;
        if (write_final) {
            final_T[p] = T;
            final_idx[p] = last;
        }
        float *o = out_img + p * C + ch_off;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (FULL || c < nch) o[c] = __builtin_fmaf(T, background[ch_off + c], acc[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_xadd_row(float v) {
    // all-reduce inside each 16-lane row: quad xor1, quad xor2, half mirror, row mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float swap_add32(float a, float b) {
    // lanes 0-31: a(own) + a(lane+32) ; lanes 32-63: b(lane-32) + b(own)
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                              __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
    // even rows: a(own) + a(lane+16) ; odd rows: b(lane-16) + b(own)
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                              __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Halving butterfly: K per-lane values -> H2 registers; afterwards the 16 lanes of row
// (b5,b4) all hold, in register j, the wave total of value red_var<K>(j,b4,b5).
template <int K>
struct Red {
    static constexpr int H1 = (K + 1) / 2;
    static constexpr int H2 = (H1 + 1) / 2;
    __device__ static __forceinline__ void run(const float (&v)[K], float (&out)[H2]) {
        float a[H1];
#pragma unroll
        for (int j = 0; j < H1; ++j) a[j] = swap_add32(v[j], (j + H1 < K) ? v[j + H1] : v[j]);
#pragma unroll
        for (int j = 0; j < H2; ++j)
            out[j] = dpp_xadd_row(swap_add16(a[j], (j + H2 < H1) ? a[j + H2] : a[j]));
    }
    // which value does (register j, row bits b4,b5) hold, and is this row its unique owner?
    __device__ static __forceinline__ int var(int j, int b4, int b5, bool &owner) {
        owner = true;
        int i = j;
        if (j + H2 < H1) i = j + H2 * b4; else owner = owner && (b4 == 0);
        int k = i;
        if (i + H1 < K) k = i + H1 * b5; else owner = owner && (b5 == 0);
        return k;
    }
};

#define BW_BATCH 64

template <int CH, bool FULL>
__global__ __launch_bounds__(256) void blend_bwd_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_out, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity) {
    constexpr int K = CH + 6;          // CH colour partials, xy(2), conic(3), opacity(1)
    constexpr int KP = (K + 3) & ~3;   // slab row stride
    using R = Red<K>;
    __shared__ float slab[BW_BATCH][KP];
    __shared__ int s_gid[BW_BATCH];
    __shared__ int s_flag[BW_BATCH];
    __shared__ int s_hi[4];

    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const float xlo = (float)qx0, xhi = (float)(qx0 + 7), ylo = (float)qy0, yhi = (float)(qy0 + 7);
    const int2 range = bins[tile];
    const size_t p = inside ? ((size_t)i * img_w + j) : 0;

    const float T_final = inside ? final_T[p] : 1.0f;
    const int fin = inside ? final_idx[p] : range.x;
    float T = T_final;
    float vo[CH];
    float W;
    {
        float B = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            vo[c] = (inside && (FULL || c < nch)) ? v_out[p * C + ch_off + c] : 0.0f;
            if (FULL || c < nch) B = __builtin_fmaf(background[ch_off + c], vo[c], B);
        }
        W = T_final * B;
    }
    // block-wide upper end of the walk
    int hi = fin;
    for (int off = 32; off > 0; off >>= 1) hi = max(hi, __shfl_xor(hi, off, 64));
    if (lane == 0) s_hi[wave] = hi;
    for (int t = threadIdx.x; t < BW_BATCH * KP; t += 256) (&slab[0][0])[t] = 0.0f;
    if (threadIdx.x < BW_BATCH) s_flag[threadIdx.x] = 0;
    __syncthreads();
    const int block_hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));

    // lane -> (register, value) map of the butterfly result
    const int b4 = (lane >> 4) & 1, b5 = lane >> 5, r16 = lane & 15;
    bool owner = false;
    int myvar = 0;
    if (r16 < R::H2) myvar = R::var(r16, b4, b5, owner);
    owner = owner && (r16 < R::H2);

    for (int top = block_hi; top > range.x; top -= BW_BATCH) {
        const int e = top - BW_BATCH + lane;
        const bool valid = e >= range.x;
        const int g = valid ? ids[e] : 0;
        if (wave == 0) s_gid[lane] = g;
        const float4 ra = reinterpret_cast<const float4 *>(rec + g)[0];
        const float4 rb = reinterpret_cast<const float4 *>(rec + g)[1];
        uint64_t m = __ballot(valid && (e < hi) && rec_hits_rect(ra, rb, xlo, xhi, ylo, yhi));
        while (m) {
            const int src = 63 - __builtin_clzll(m);
            m &= ~(1ull << src);
            const int idx = top - BW_BATCH + src;
            const float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.x), src));
            const float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.y), src));
            const float gop = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.z), src));
            const float gthr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ra.w), src));
            const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.x), src));
            const float cb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.y), src));
            const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb.z), src));
            const int gid = __builtin_amdgcn_readlane(g, src);
            const float dx = gx - px, dy = gy - py;
            const float sigma =
                __builtin_fmaf(0.5f, __builtin_fmaf(ca * dx, dx, (cc * dy) * dy), (cb * dx) * dy);
            const bool act = (idx < fin) && sigma >= 0.0f;
            if (__ballot(act && sigma <= gthr) == 0ull) continue;
            const float vis = gg_expf(-sigma);
            const float alpha = fminf(GG_ALPHA_MAX_BWD, gop * vis);
            const bool pass = act && !(alpha < GG_ALPHA_MIN);
            if (__ballot(pass) == 0ull) continue;

            float part[K];
            const float *col = colors + (size_t)gid * C + ch_off;
            const float ra_ = 1.0f / (1.0f - alpha);
            const float Tn = T * ra_;
            const float fac = pass ? alpha * Tn : 0.0f;
            float cv[CH];  // wave-uniform -> one wide scalar load
#pragma unroll
            for (int c = 0; c < CH; ++c) cv[c] = (FULL || c < nch) ? col[c] : 0.0f;
            float D = 0.0f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                D = __builtin_fmaf(cv[c], vo[c], D);
                part[c] = fac * vo[c];
            }
            const float v_alpha = pass ? (Tn * D - ra_ * W) : 0.0f;
            if (pass) {
                W = __builtin_fmaf(D, fac, W);
                T = Tn;
            }
            const float v_sigma = (-gop * vis) * v_alpha;
            part[CH + 0] = v_sigma * (ca * dx + cb * dy);
            part[CH + 1] = v_sigma * (cb * dx + cc * dy);
            const float hs = 0.5f * v_sigma;
            part[CH + 2] = (hs * dx) * dx;
            part[CH + 3] = (hs * dx) * dy;
            part[CH + 4] = (hs * dy) * dy;
            part[CH + 5] = vis * v_alpha;

            float red[R::H2];
            R::run(part, red);
            float mine = red[0];
#pragma unroll
            for (int q = 1; q < R::H2; ++q) mine = (r16 == q) ? red[q] : mine;
            if (owner) atomicAdd(&slab[src][myvar], mine);
            if (lane == 0) s_flag[src] = 1;
        }
        __syncthreads();
        // flush: one Gaussian row per wave-instruction (lanes 0..K-1), then clear
        for (int row = wave; row < BW_BATCH; row += 4) {
            if (s_flag[row]) {  // wave-uniform
                const int gid = s_gid[row];
                if (lane < K) {
                    const float val = slab[row][lane];
                    slab[row][lane] = 0.0f;
                    float *dst;
                    if (lane < CH) dst = v_colors + (size_t)gid * C + ch_off + lane;
                    else if (lane < CH + 2) dst = v_xy + 2 * (size_t)gid + (lane - CH);
                    else if (lane < CH + 5) dst = v_conic + 3 * (size_t)gid + (lane - CH - 2);
                    else dst = v_opacity + gid;
                    if (FULL || lane >= CH || lane < nch) atomicAdd(dst, val);
                }
                if (lane == 0) s_flag[row] = 0;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static int chunk_width(int remaining) {
    if (remaining >= 32) return 32;
    if (remaining > 16) return 32;
    if (remaining > 8) return 16;
    if (remaining > 4) return 8;
    if (remaining == 4) return 4;
    if (remaining == 3) return 3;
    return remaining <= 1 ? 1 : 4;
}

#define FWD_ARGS C, off, n, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, \
                 colors, background, out_img, final_Ts, final_idx, write_final
template <int CH>
static void launch_fwd(int C, int off, int n, int img_h, int img_w, int tiles_x, int ntiles,
                       const int32_t *ids, const int32_t *tile_bins, const GRec *rec,
                       const float *colors, const float *background, float *out_img,
                       float *final_Ts, int32_t *final_idx, int write_final, hipStream_t s) {
    gg_prof_begin(GG_K_BLEND_FWD + gg_width_index(CH), s);
    if (n == CH)
        hipLaunchKernelGGL((blend_fwd_kernel<CH, true>), dim3(ntiles), dim3(256), 0, s, FWD_ARGS);
    else
        hipLaunchKernelGGL((blend_fwd_kernel<CH, false>), dim3(ntiles), dim3(256), 0, s, FWD_ARGS);
    gg_prof_end(GG_K_BLEND_FWD + gg_width_index(CH), s);
}

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
    if (N > 0)
        hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics,
                           opacity, rec);
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    for (int off = 0; off < C;) {
        int w = chunk_width(C - off);
        int n = min(w, C - off);
        int write_final = (off == 0);
        switch (w) {
            case 1: launch_fwd<1>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
            case 3: launch_fwd<3>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
            case 4: launch_fwd<4>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
            case 8: launch_fwd<8>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
            case 16: launch_fwd<16>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
            default: launch_fwd<32>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, out_img, final_Ts, final_idx, write_final, s); break;
        }
        off += n;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int CH>
static void launch_bwd(int C, int off, int n, int img_h, int img_w, int tiles_x, int ntiles,
                       const int32_t *ids, const int32_t *tile_bins, const GRec *rec,
                       const float *colors, const float *background, const float *final_Ts,
                       const int32_t *final_idx, const float *v_out, float *v_xy, float *v_conic,
                       float *v_colors, float *v_opacity, hipStream_t s) {
    gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(CH), s);
    if (n == CH)
        hipLaunchKernelGGL((blend_bwd_kernel<CH, true>), dim3(ntiles), dim3(256), 0, s, C, off, n,
                           img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                           background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
    else
        hipLaunchKernelGGL((blend_bwd_kernel<CH, false>), dim3(ntiles), dim3(256), 0, s, C, off, n,
                           img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                           background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
    gg_prof_end(GG_K_BLEND_BWD + gg_width_index(CH), s);
}

extern "C" int gg_blend_bwd(int C, int N, int img_h, int img_w, const int32_t *ids,
                            const int32_t *tile_bins, const float *xys, const float *conics,
                            const float *colors, const float *opacity, const float *background,
                            const float *final_Ts, const int32_t *final_idx, const float *v_out,
                            float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
                            void *ws, size_t ws_bytes, gg_stream_t stream) {
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
    hipLaunchKernelGGL(blend_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, xys, conics,
                       opacity, rec);
    bool fail = hipMemsetAsync(v_xy, 0, sizeof(float) * 2 * (size_t)N, s) != hipSuccess;
    fail |= hipMemsetAsync(v_conic, 0, sizeof(float) * 3 * (size_t)N, s) != hipSuccess;
    fail |= hipMemsetAsync(v_colors, 0, sizeof(float) * (size_t)C * (size_t)N, s) != hipSuccess;
    fail |= hipMemsetAsync(v_opacity, 0, sizeof(float) * (size_t)N, s) != hipSuccess;
    if (fail) {
        gg_set_error("gg_blend_bwd: memset failed");
        return GG_ERR_LAUNCH;
    }
    const int tiles_x = (img_w + GG_BLOCK - 1) / GG_BLOCK, tiles_y = (img_h + GG_BLOCK - 1) / GG_BLOCK;
    const int ntiles = tiles_x * tiles_y;
    for (int off = 0; off < C;) {
        int w = chunk_width(C - off);
        int n = min(w, C - off);
        switch (w) {
            case 1: launch_bwd<1>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 3: launch_bwd<3>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 4: launch_bwd<4>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 8: launch_bwd<8>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 16: launch_bwd<16>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            default: launch_bwd<32>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
        }
        off += n;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}
