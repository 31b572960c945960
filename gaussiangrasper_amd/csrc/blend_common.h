// blend_common.h — device helpers shared by the blend kernels (blend.hip: v1 kernels + C ABI,
// blend2.hip: v2 kernels).
#pragma once
#include "gg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define BW_BATCH 64

// Deterministic backward (gg_blend_bwd_deterministic): instead of float atomics the kernels store the total
// of every (list entry, quadrant) into a slab, slab[(entry * 4 + quadrant) * ks + column]; a second pass
// sums each Gaussian's entries in list order.  Columns: [0, C) colours, [C + 6 c, C + 6 c + 6) the geometry
// partials of channel chunk c.  p == nullptr: atomics.
struct DetSlab {
    float *p;
    int ks, goff, coff;
};

struct __attribute__((aligned(16))) GRec {
    float x, y, opac, thr;  // thr: sigma above which alpha < 1/255 for certain (conservative)
    float ca, cb, cc, pad;
};

// pack one Gaussian's record (shared by blend_prep_kernel and view_fwd_kernel: the same bits)
__device__ __forceinline__ void grec_pack(float x, float y, float opac, float ca, float cb, float cc, GRec *__restrict__ dst) {
    // alpha = opac*exp(-sigma) >= 1/255  <=>  sigma <= ln(255*opac).  Margins absorb the fp32
    // rounding of sigma (rel ~5e-7), of gg_expf (2 ulp) and of the fast log (1e-6).
    float t = __logf(255.0f * opac);
    t = t + 0.002f * fabsf(t) + 0.002f;
    if (!(opac > 0.0f)) t = -1.0f;                   // alpha <= 0 < 1/255 always
    if (opac != opac) t = __builtin_inff();           // NaN opacity: never cull (NaN propagates)
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    d4[0] = make_float4(x, y, opac, t);
    d4[1] = make_float4(ca, cb, cc, 0.0f);
}

// Tile of a workgroup.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD
// b & 7), each with its own L2.  XCD x takes the strips s = x, x+8, x+16, ... of GG_STRIP consecutive
// tiles of a tile row: the tiles of a strip share most of their Gaussians (L2 hits), and the strips
// of an XCD are spread over the whole image, so every XCD gets the same share of dense and empty
// regions.  The first mapping gave each XCD one contiguous band of the image: on the bench view the
// busiest XCD then had 24 % of the blend work and the top band 0 % (max/mean 1.93), and the blend
// kernels ran 1.1-1.4x slower (profiles/README.md "XCD balance").  Strips of 4, 2x2 / 4x4 / 8x8 tile
// blocks, whole tile rows and plain round-robin all measure the same within noise.
// Launch with gg_tile_grid(ntiles) workgroups; xcd_tile returns -1 for the padding workgroups.
#define GG_STRIP 4
__host__ __device__ __forceinline__ int gg_tile_grid(int ntiles) {
    return ((ntiles + 8 * GG_STRIP - 1) / (8 * GG_STRIP)) * (8 * GG_STRIP);
}
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
    const int x = bid & 7, k = bid >> 3;
    const int tile = ((k / GG_STRIP) * 8 + x) * GG_STRIP + (k % GG_STRIP);
    return tile < ntiles ? tile : -1;
}

// Waves per workgroup of the blend kernels (WPB).  The waves of a tile never synchronise (every wave walks its
// 8x8 quadrant on its own), so a workgroup need not hold all four.  With WPB == 1 every quadrant is its own
// 64-thread workgroup: a finished quadrant frees its registers and LDS at once instead of holding them until the
// slowest quadrant of its tile is done.  Workgroups b, b + 8, b + 16, b + 24 are then the four quadrants of one
// tile (same XCD: they share the tile's list and records through that XCD's L2).  Measured (r02, bench view):
// wide backward 0.997 -> 0.906 ms, pair backward 1.243 -> 1.188 with WPB 1; the forward and the narrow kernels
// (5-8 waves per SIMD, the four quadrants' list reads hitting in the CU's L1) are 3-5 % SLOWER with it.
#ifndef GG_WPB_WIDE_BWD
#define GG_WPB_WIDE_BWD 1
#endif
#ifndef GG_WPB_OTHER
#define GG_WPB_OTHER 4
#endif
__host__ __device__ __forceinline__ int gg_blend_grid(int ntiles, int wpb) { return gg_tile_grid(ntiles) * (4 / wpb); }
// tile (or -1) and quadrant of this wave
#ifdef GG_ABLATION
// measurement twin only (tools/kbench_order.py): an explicit workgroup -> tile order, e.g. longest list first
static __device__ const int *g_tile_order = nullptr;
#endif
template <int WPB>
__device__ __forceinline__ int blend_tile_wave(int bid, int tid, int ntiles, int &wave) {
    static_assert(WPB == 4 || WPB == 2 || WPB == 1, "four, two or one quadrant per workgroup");
    int k = bid;
    if (WPB == 4) {
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    } else if (WPB == 2) {   // workgroups b and b + 8: the upper and the lower pair of quadrants of one tile
        wave = 2 * ((bid >> 3) & 1) + __builtin_amdgcn_readfirstlane(tid >> 6);
        k = ((bid >> 4) << 3) | (bid & 7);
    } else {
        wave = (bid >> 3) & 3;
        k = ((bid >> 5) << 3) | (bid & 7);
    }
#ifdef GG_ABLATION
    if (g_tile_order) return k < ntiles ? __builtin_amdgcn_readfirstlane(g_tile_order[k]) : -1;   // (wave-uniform)
#endif
    return xcd_tile(k, ntiles);
}

// Does the alpha>=1/255 ellipse of a record reach the pixel rectangle [xlo,xhi]x[ylo,yhi]?
// Exact in real arithmetic: sigma is a convex quadratic, so its minimum over the rectangle is 0
// if the centre is inside and otherwise lies on one of the (at most two) edges facing the centre;
// on an edge the minimiser along the free coordinate is the clamped 1-D optimum.  The record's
// cut-off already carries the rounding margin, and the continuous minimum is <= the minimum over
// the pixel centres, so the test never rejects a pair the exact per-pixel test would accept.
__device__ __forceinline__ float sigma_at(const float4 b, float dx, float dy) {
    return 0.5f * (b.x * dx * dx + b.z * dy * dy) + b.y * dx * dy;
}
__device__ __forceinline__ bool rec_hits_rect(const float4 a, const float4 b, float xlo, float xhi,
                                              float ylo, float yhi) {
    const float thr = a.w;
    if (thr < 0.0f) return false;
    // the argument needs a positive-definite conic (always true for projected Gaussians: the 0.3
    // blur); anything else is never culled
    if (!(b.x > 0.0f) || !(b.x * b.z - b.y * b.y > 0.0f)) return true;
    // d = centre - pixel, with the pixel ranging over the rectangle
    const float dx_lo = a.x - xhi, dx_hi = a.x - xlo, dy_lo = a.y - yhi, dy_hi = a.y - ylo;
    const float dxn = fminf(fmaxf(0.0f, dx_lo), dx_hi);  // |d| nearest to 0 inside the range
    const float dyn = fminf(fmaxf(0.0f, dy_lo), dy_hi);
    // edge x = nearest x: dy free
    const float dy1 = fminf(fmaxf(-b.y * dxn * __builtin_amdgcn_rcpf(b.z), dy_lo), dy_hi);
    // edge y = nearest y: dx free
    const float dx2 = fminf(fmaxf(-b.y * dyn * __builtin_amdgcn_rcpf(b.x), dx_lo), dx_hi);
    const float smin = fminf(sigma_at(b, dxn, dy1), sigma_at(b, dx2, dyn));
    return !(smin > thr);  // NaN (degenerate conic) -> hit, conservative
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

// ---------------------------------------------------------------------------------------------
// Red6<K>: full halving butterfly, K <= 64 per-lane values -> ONE register per lane.
// Six levels, each pairs lanes that differ in one lane-index bit; at a level the lanes with that
// bit clear keep the first half of the values, the others the second half, and each receives its
// partner's copy of what it keeps — so the value count halves at every level and the expensive
// exchanges run last, on the fewest values.  Measured cost on gfx950 (tools/ubench_xlane.hip):
// plain VALU 1 ns, DPP add 2.5 ns, v_permlane16/32_swap 5.5 ns per SIMD.  Level order and exchange:
//   bit0 quad_perm[1,0,3,2]   bit1 quad_perm[2,3,0,1]   bit3 row_ror:8
//   bit2 row_shl:4 / row_shr:4 (two bank-masked DPP moves)   bit4 v_permlane16_swap   bit5 v_permlane32_swap
// After run(): lane l holds the wave total of value var(l) (owner(l) marks one lane per value).
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104 /*row_shl:4*/, 0xf, 0x5, true);
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114 /*row_shr:4*/, 0xf, 0xa, true);
    return __builtin_bit_cast(float, t);
}
// MODE: 0 bit0, 1 bit1, 2 bit3, 3 bit2, 4 bit4, 5 bit5
template <int MODE>
__device__ __forceinline__ float red6_partner(float send) {
    if (MODE == 0) return dpp_mov<0xB1>(send);
    if (MODE == 1) return dpp_mov<0x4E>(send);
    if (MODE == 2) return dpp_mov<0x128>(send);  // row_ror:8
    return dpp_xor4(send);
}
template <int NIN, int MODE>
__device__ __forceinline__ void red6_level(const float (&in)[NIN], float (&out)[(NIN + 1) / 2], bool bit) {
    constexpr int H = (NIN + 1) / 2;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        if (j + H < NIN) {
            const float a = in[j], b = in[j + H];
            if (MODE < 4) {
                const float send = bit ? a : b, keep = bit ? b : a;
                out[j] = keep + red6_partner<MODE>(send);
            } else if (MODE == 4) {
                out[j] = swap_add16(a, b);
            } else {
                out[j] = swap_add32(a, b);
            }
        } else {  // odd count: this value is fully added on both sides (stays replicated)
            if (MODE < 4) out[j] = in[j] + red6_partner<MODE>(in[j]);
            else if (MODE == 4) out[j] = swap_add16(in[j], in[j]);
            else out[j] = swap_add32(in[j], in[j]);
        }
    }
}
template <int K>
struct Red6 {
    static_assert(K >= 1 && K <= 64, "Red6 handles up to 64 values");
    static constexpr int N0 = K, N1 = (N0 + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2,
                         N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2, N6 = (N5 + 1) / 2;
    __device__ static __forceinline__ float run(const float (&v)[K], int lane) {
        float a1[N1], a2[N2], a3[N3], a4[N4], a5[N5], a6[N6];
        red6_level<N0, 0>(v, a1, (lane & 1) != 0);
        red6_level<N1, 1>(a1, a2, (lane & 2) != 0);
        red6_level<N2, 2>(a2, a3, (lane & 8) != 0);
        red6_level<N3, 3>(a3, a4, (lane & 4) != 0);
        red6_level<N4, 4>(a4, a5, (lane & 16) != 0);
        red6_level<N5, 5>(a5, a6, (lane & 32) != 0);
        return a6[0];
    }
    // which value does `lane` hold, and is it the unique owner?
    __device__ static __forceinline__ int var(int lane, bool &owner) {
        const int nin[6] = {N0, N1, N2, N3, N4, N5};
        const int bits[6] = {lane & 1, (lane >> 1) & 1, (lane >> 3) & 1, (lane >> 2) & 1,
                             (lane >> 4) & 1, (lane >> 5) & 1};
        int idx = 0;
        owner = true;
#pragma unroll
        for (int m = 5; m >= 0; --m) {
            const int h = (nin[m] + 1) / 2;
            if (idx + h < nin[m]) idx += h * bits[m];
            else owner = owner && (bits[m] == 0);
        }
        return idx;
    }
};
