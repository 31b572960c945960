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
        // flush: RPI = 64/K Gaussian rows per global-atomic wave-instruction (K consecutive
        // floats of each touched gradient row), then clear
        {
            constexpr int RPI = 64 / K;
            const int rsub = lane / K, k = lane - rsub * K;
            for (int row0 = wave * RPI; row0 < BW_BATCH; row0 += 4 * RPI) {
                const int row = row0 + rsub;
                if (rsub < RPI && row < BW_BATCH && s_flag[row]) {
                    const int gid = s_gid[row];
                    const float val = slab[row][k];
                    slab[row][k] = 0.0f;
                    float *dst;
                    if (k < CH) dst = v_colors + (size_t)gid * C + ch_off + k;
                    else if (k < CH + 2) dst = v_xy + 2 * (size_t)gid + (k - CH);
                    else if (k < CH + 5) dst = v_conic + 3 * (size_t)gid + (k - CH - 2);
                    else dst = v_opacity + gid;
                    if (FULL || k >= CH || k < nch) atomicAdd(dst, val);
                    if (k == 0) s_flag[row] = 0;  // after every lane of this row has read it
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// backward, 32-channel chunks: colour-gradient reduction on the matrix pipe
//
// v_colors[g][c] = sum over pixels of fac[g][p] * v_out[p][c] is a genuine contraction over the
// pixel index.  The VALU kernel above pays ~2.6 cross-lane ops per value for it (butterfly); here a
// wave instead parks fac of up to 32 contributing Gaussians in LDS ([slot][pixel], one
// ds_write_b32 per Gaussian) and multiplies FAC[32 x 64] by V_OUT[64 x 32] with 32
// v_mfma_f32_32x32x2_f32 — exact fp32 fma chains (MI355X_MICROARCH: same rate as the VALU, but a
// separate pipe that is otherwise idle here), i.e. ~1 MFMA + 2 LDS ops per Gaussian instead of
// ~150 VALU ops.  The V_OUT B-operands (32 VGPRs) are loaded once per kernel.  The six geometry
// partials still go through the register butterfly, everything else is as in blend_bwd_kernel.
// ---------------------------------------------------------------------------------------------
#define BWW_SLOTS 32
#define BWW_FSTRIDE 65  // fac row stride in floats: odd -> conflict-free column reads

template <bool FULL>
__global__ __launch_bounds__(256) void blend_bwd_wide_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_out, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity) {
    constexpr int CH = 32;
    constexpr int K = CH + 6;
    constexpr int KP = 40;
    using R = Red<6>;
    __shared__ float slab[BW_BATCH][KP];
    __shared__ float s_fac[4][BWW_SLOTS * BWW_FSTRIDE];
    __shared__ int s_slotrow[4][BWW_SLOTS];
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
    float *fac_w = s_fac[wave];
    int *slotrow_w = s_slotrow[wave];

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
    // MFMA B operands: step s covers pixels 2s, 2s+1 of the quadrant; lane l holds
    // V_OUT[pixel 2s + (l>>5)][channel l&31]
    float vob[32];
    {
        const int cch = lane & 31, half = lane >> 5;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int pq = 2 * s + half;
            const int pj = qx0 + (pq & 7), pi = qy0 + (pq >> 3);
            const bool ok = (pi < img_h) && (pj < img_w) && (FULL || cch < nch);
            vob[s] = ok ? v_out[((size_t)pi * img_w + pj) * C + ch_off + cch] : 0.0f;
        }
    }
    int hi = fin;
    for (int off = 32; off > 0; off >>= 1) hi = max(hi, __shfl_xor(hi, off, 64));
    if (lane == 0) s_hi[wave] = hi;
    for (int t = threadIdx.x; t < BW_BATCH * KP; t += 256) (&slab[0][0])[t] = 0.0f;
    if (threadIdx.x < BW_BATCH) s_flag[threadIdx.x] = 0;
    __syncthreads();
    const int block_hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));

    const int b4 = (lane >> 4) & 1, b5 = lane >> 5, r16 = lane & 15;
    bool owner = false;
    int myvar = 0;
    if (r16 < R::H2) myvar = R::var(r16, b4, b5, owner);
    owner = owner && (r16 < R::H2);

    int nslots = 0;  // wave-uniform
    auto flush_slots = [&]() {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float *arow = fac_w + (lane & 31) * BWW_FSTRIDE + (lane >> 5);
#pragma unroll
        for (int s = 0; s < 32; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * s], vob[s], acc, 0, 0, 0);
        // D layout: column = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (slot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (slot < nslots) atomicAdd(&slab[slotrow_w[slot]][lane & 31], acc[r]);
        }
        nslots = 0;
    };

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

            const float *col = colors + (size_t)gid * C + ch_off;
            float cv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) cv[c] = (FULL || c < nch) ? col[c] : 0.0f;
            const float ra_ = 1.0f / (1.0f - alpha);
            const float Tn = T * ra_;
            const float fac = pass ? alpha * Tn : 0.0f;
            float D = 0.0f;
#pragma unroll
            for (int c = 0; c < CH; ++c) D = __builtin_fmaf(cv[c], vo[c], D);
            const float v_alpha = pass ? (Tn * D - ra_ * W) : 0.0f;
            if (pass) {
                W = __builtin_fmaf(D, fac, W);
                T = Tn;
            }
            const float v_sigma = pass ? (-gop * vis) * v_alpha : 0.0f;
            float part[6];
            part[0] = v_sigma * (ca * dx + cb * dy);
            part[1] = v_sigma * (cb * dx + cc * dy);
            const float hs = 0.5f * v_sigma;
            part[2] = (hs * dx) * dx;
            part[3] = (hs * dx) * dy;
            part[4] = (hs * dy) * dy;
            part[5] = pass ? vis * v_alpha : 0.0f;

            // colour part: park fac for the matrix pipe
            fac_w[nslots * BWW_FSTRIDE + lane] = fac;
            if (lane == 0) {
                slotrow_w[nslots] = src;
                s_flag[src] = 1;
            }
            ++nslots;

            float red[R::H2];
            R::run(part, red);
            float mine = red[0];
#pragma unroll
            for (int q = 1; q < R::H2; ++q) mine = (r16 == q) ? red[q] : mine;
            if (owner) atomicAdd(&slab[src][CH + myvar], mine);
            if (nslots == BWW_SLOTS) flush_slots();
        }
        if (nslots > 0) flush_slots();
        __syncthreads();
        for (int row = wave; row < BW_BATCH; row += 4) {
            if (s_flag[row]) {
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
// v2 kernels live in blend2.hip
void gg_launch_blend2_fwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, float *out_img,
                          float *final_Ts, int32_t *final_idx, int write_final, hipStream_t s);
void gg_launch_blend2_bwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, const float *final_Ts,
                          const int32_t *final_idx, const float *v_out, float *v_xy, float *v_conic,
                          float *v_colors, float *v_opacity, hipStream_t s);

// GG_BLEND_IMPL=1 selects the first-generation kernels of this file (kept for A/B measurement and
// as a cross-check in the tests); default 2 = blend2.hip.
static int blend_impl() {
    static int impl = -1;
    if (impl < 0) {
        const char *e = getenv("GG_BLEND_IMPL");
        impl = (e && e[0] == '1') ? 1 : 2;
    }
    return impl;
}
static int chunk_width2(int remaining) { return remaining <= 3 ? remaining : 32; }
static int g_wide_impl = 2;  // 2: blend2_bwd_wide_kernel, 1: blend_bwd_wide_kernel (A/B, gg_debug_set_ablation(100+x))
static int g_ablate = 0;  // measurement only (gg_debug_set_ablation)
extern "C" int gg_debug_set_ablation(int level) {
    int prev = g_ablate;
    if (level >= 100) {  // 101 / 102: choose the wide-backward implementation
        g_wide_impl = level - 100;
        return prev;
    }
    g_ablate = level;
    return prev;
}
void gg_launch_blend2_bwd_ablate(int abl, int C, int off, int img_h, int img_w, int tiles_x,
                                 int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                                 const float *colors, const float *background, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_opacity, hipStream_t s);

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
    for (int off = 0; off < C && blend_impl() == 2;) {
        int w = chunk_width2(C - off);
        int n = min(w, C - off);
        gg_prof_begin(GG_K_BLEND_FWD + gg_width_index(w), s);
        gg_launch_blend2_fwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins,
                             rec, colors, background, out_img, final_Ts, final_idx, off == 0, s);
        gg_prof_end(GG_K_BLEND_FWD + gg_width_index(w), s);
        off += n;
    }
    for (int off = 0; off < C && blend_impl() == 1;) {
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
    for (int off = 0; off < C && blend_impl() == 2;) {
        int w = chunk_width2(C - off);
        int n = min(w, C - off);
        gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(w), s);
        if (w == 3 && g_ablate > 0)
            gg_launch_blend2_bwd_ablate(g_ablate, C, off, img_h, img_w, tiles_x, ntiles, ids,
                                        (const int2 *)tile_bins, rec, colors, background, final_Ts,
                                        final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s);
        else if (w <= 3 || g_wide_impl == 2)
            gg_launch_blend2_bwd(w, C, off, n, img_h, img_w, tiles_x, ntiles, ids,
                                 (const int2 *)tile_bins, rec, colors, background, final_Ts,
                                 final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s);
        else if (n == 32)  // measured: the one-Gaussian-at-a-time MFMA kernel is the faster wide bwd
            hipLaunchKernelGGL((blend_bwd_wide_kernel<true>), dim3(ntiles), dim3(256), 0, s, C, off, n,
                               img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                               background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
        else
            hipLaunchKernelGGL((blend_bwd_wide_kernel<false>), dim3(ntiles), dim3(256), 0, s, C, off, n,
                               img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors,
                               background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
        gg_prof_end(GG_K_BLEND_BWD + gg_width_index(w), s);
        off += n;
    }
    for (int off = 0; off < C && blend_impl() == 1;) {
        int w = chunk_width(C - off);
        int n = min(w, C - off);
        switch (w) {
            case 1: launch_bwd<1>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 3: launch_bwd<3>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 4: launch_bwd<4>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 8: launch_bwd<8>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            case 16: launch_bwd<16>(C, off, n, img_h, img_w, tiles_x, ntiles, ids, tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, s); break;
            default:
                gg_prof_begin(GG_K_BLEND_BWD + gg_width_index(32), s);
                if (n == 32)
                    hipLaunchKernelGGL((blend_bwd_wide_kernel<true>), dim3(ntiles), dim3(256), 0, s, C, off, n, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
                else
                    hipLaunchKernelGGL((blend_bwd_wide_kernel<false>), dim3(ntiles), dim3(256), 0, s, C, off, n, img_h, img_w, tiles_x, ntiles, ids, (const int2 *)tile_bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity);
                gg_prof_end(GG_K_BLEND_BWD + gg_width_index(32), s);
                break;
        }
        off += n;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}
