// blend2.hip — the blend kernels (second generation; the first, profiles/r01_v1_*, is gone).
//
// What the first generation measured (profiles/r01_v1_*, PMC passes): only ~28 % of wave-cycles
// issue an instruction; SALU instruction count is ~70 % of VALU; the rest is s_waitcnt and
// dependency stalls of a branchy one-Gaussian-at-a-time loop (s_ff1 / 8x v_readlane / ballots /
// scalar colour loads waited on inside every iteration).  The engine here keeps the exact
// per-(pixel, Gaussian) arithmetic and restructures everything around it:
//
//   * chunk staging (64 list entries, one per lane) + exact ellipse-vs-quadrant cull as before,
//     but the survivors are COMPACTED INTO LDS in depth order (mbcnt prefix of the ballot), so the
//     walk is a plain counted loop and the per-Gaussian record reaches every lane through
//     wave-uniform ds_read_b128 broadcasts — no readlane, no mask juggling;
//   * the walk handles FOUR survivors per iteration: sigma / gg_expf / alpha of the four are
//     independent instruction streams (ILP hides the VALU latency a single chain exposes), only the
//     T recurrence is sequential, and it is branch-free (selects), so there is ONE scalar branch per
//     four Gaussians instead of ~6 per Gaussian;
//   * narrow calls (<= 3 channels: rgb / depth / normal) carry the colours inside the LDS record;
//   * 32-channel chunks accumulate on the matrix pipe: out[p][c] += vis[p][g]*colour[g][c] is
//     v_mfma_f32_32x32x2_f32 with the two Gaussians of a pair as the k index — an exact fp32 fma
//     chain in list order, i.e. bit-identical to the oracle's sequential fmaf, at no VALU cost.
//     The A operand (vis of 2 Gaussians x 64 pixels) is one v_permlane32_swap; the B operand is one
//     coalesced 2x128-byte colour load per pair.
//
// Bit-exactness: blended contributions are `acc = fma(colour, vis, acc)` with vis = 0 for pixels
// that skip the Gaussian; fma(c, 0, acc) == acc for finite c, so the unconditional form equals
// the oracle's conditional one (colours must be finite, as everywhere).
#include "blend_common.h"

#define GRP 4          // survivors per loop iteration
#define KEEP(x) asm volatile("" ::"v"(x))   // measurement builds: keep a value alive

#include "blend_measure.h"   // GG_ABLATION / GG_STAMPS / GG_WALK_STATS hooks: empty macros in the product build
#define LIST_CAP 72    // 4 leading pads + 64 + 4 trailing pads

// NC: colour float4s per record.  <= 3 channels: c = colours 0..2 and .w = list position; 8-channel
// narrow kernels (the 4..8-channel tail of a fused call: rgb | depth | normal behind 32 feature
// channels): c = colours 0..3, d = colours 4..7 and the position moves to a.w.
template <int NC = 1>
struct WaveListT {
    float4 a[LIST_CAP];  // x, y, opacity, (cut-off; NC == 2: list position)
    float4 b[LIST_CAP];  // conic a, b, c, Gaussian id (int bits)
    float4 c[LIST_CAP];  // narrow: colours 0..2(3) ; wide: .x = Gaussian id (int bits). NC == 1: .w = list position
    float4 d[NC >= 2 ? LIST_CAP : 1];
    float4 e[NC >= 3 ? LIST_CAP : 1];   // NC == 3: wide forward with a second, <= 8-channel colour array (d, e)
};

// second colour array blended in the same walk as a 32-channel chunk (gg_blend_fwd_pair)
struct Seg2 {
    const float *colors;      // (N, C2)
    const float *background;  // (C2,)
    float *out_img;           // (H, W, C2)
    int C2, nch2;             // row stride and channels used (<= 8)
};
typedef WaveListT<1> WaveList;

__device__ __forceinline__ int lane_prefix(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Stage one chunk: returns the survivor count; list[OFF + k] = k-th survivor in list order,
// GRP null records (opacity 0 -> never pass) on both sides.
template <int CH, bool WIDE, bool REL = false, typename LIST = WaveList>
__device__ __forceinline__ int stage_chunk(LIST &L, int lane, int e, bool valid,
                                           const int g /* ids[e], loaded by the caller one chunk ahead */,
                                           const GRec *__restrict__ rec,
                                           const float *__restrict__ colors, int C, int ch_off, int nch,
                                           float xlo, float xhi, float ylo, float yhi,
                                           const Seg2 *seg2 = nullptr) {
    const float4 ra = reinterpret_cast<const float4 *>(rec + g)[0];
    const float4 rb = reinterpret_cast<const float4 *>(rec + g)[1];
    const bool hit = valid && rec_hits_rect(ra, rb, xlo, xhi, ylo, yhi);
    const uint64_t m = __ballot(hit);
    const int cnt = __builtin_popcountll(m);
    const int pos = GRP + lane_prefix(m);
    if (hit) {
        // .w: forward = list position + 1 (final_idx); backward = position inside the chunk
        const float posf = __builtin_bit_cast(float, REL ? lane : e + 1);
        float4 cc = make_float4(0.f, 0.f, 0.f, posf);
        float4 ra2 = ra;
        if (WIDE) {
            cc.x = __builtin_bit_cast(float, g);
            if (seg2) {   // the second array's colours ride in the record (d, e)
                const float *c2 = seg2->colors + (size_t)g * seg2->C2;
                const int n2 = seg2->nch2;
                L.d[pos] = make_float4(c2[0], n2 > 1 ? c2[1] : 0.f, n2 > 2 ? c2[2] : 0.f, n2 > 3 ? c2[3] : 0.f);
                L.e[pos] = make_float4(n2 > 4 ? c2[4] : 0.f, n2 > 5 ? c2[5] : 0.f, n2 > 6 ? c2[6] : 0.f,
                                       n2 > 7 ? c2[7] : 0.f);
            }
        } else {
            const float *col = colors + (size_t)g * C + ch_off;
            cc.x = col[0];
            if (CH > 1 && nch > 1) cc.y = col[1];
            if (CH > 2 && nch > 2) cc.z = col[2];
            if (CH > 3) {   // 8-channel records: the position lives in a.w
                cc.w = nch > 3 ? col[3] : 0.f;
                float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nch > 4) dd.x = col[4];
                if (nch > 5) dd.y = col[5];
                if (nch > 6) dd.z = col[6];
                if (nch > 7) dd.w = col[7];
                L.d[pos] = dd;
                ra2.w = posf;
            }
        }
        L.a[pos] = ra2;
        L.b[pos] = make_float4(rb.x, rb.y, rb.z, __builtin_bit_cast(float, g));  // .w = Gaussian id
        L.c[pos] = cc;
    }
    if (lane < 2 * GRP) {  // null pads: opacity 0, conic 0 -> alpha = 0 < 1/255
        const int q = (lane < GRP) ? lane : (cnt + lane);
        L.a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        L.b[q] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));
        L.c[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CH > 3 && !WIDE) L.d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (WIDE && seg2) {
            L.d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            L.e[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

// =============================================================================================
// forward
// =============================================================================================
// EX: a second colour array of <= 8 channels (Seg2) is blended in the same walk as this 32-channel chunk
// (gg_blend_fwd_pair: the plugin's feature | rgb+depth+normal forward in one walk instead of two)
#ifndef GG_FWD_WAVES
#define GG_FWD_WAVES 5
#endif
// NCB (r03): 32-channel blocks per walk of a wide build — one cull, one sigma / exp / alpha / T per (pixel, Gaussian)
// for 32 NCB channels: NCB more colour dwords per lane and pair, 2 NCB MFMAs, 32 NCB accumulator registers.  What
// a 128-channel feature image (BASELINE config 5) is rendered with: per channel the same fma / MFMA sequence, so
// the images are bit-identical to the one-block walks'.
template <int CH, bool WIDE, bool FULL, bool EX = false, int FABL = 0, int NCB = 1>
__global__ __launch_bounds__(64 * GG_WPB_OTHER)
__attribute__((amdgpu_waves_per_eu((WIDE && EX && NCB == 1) ? GG_FWD_WAVES : (NCB == 2 ? (EX ? 4 : 3) : (NCB > 2 ? 2 : 1)))))
void blend2_fwd_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    float *__restrict__ out_img, float *__restrict__ final_T, int32_t *__restrict__ final_idx,
    int write_final, Seg2 seg2 = Seg2()) {
    constexpr bool N8 = !WIDE && CH > 3;          // 8-channel narrow record layout
    static_assert(!EX || WIDE, "the second array rides on the wide kernel");
    static_assert(NCB == 1 || (WIDE && FULL), "several channel blocks: the full wide builds");
    typedef WaveListT<EX ? 3 : (N8 ? 2 : 1)> LIST;
    __shared__ LIST lists[GG_WPB_OTHER];
    int wave;
    const int tile = blend_tile_wave<GG_WPB_OTHER>(blockIdx.x, threadIdx.x, ntiles, wave);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    const int wslot = wave & (GG_WPB_OTHER - 1);   // this wave's LDS
    LIST &L = lists[wslot];
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const float xlo = (float)qx0, xhi = (float)(qx0 + 7), ylo = (float)qy0, yhi = (float)(qy0 + 7);
    const int2 range = bins[tile];

    STAMP_DECL;
    float T = 1.0f;
    int last = range.x;
    bool done = !inside;
    float acc[WIDE ? 1 : CH];
    float acc2[EX ? 8 : 1];
#pragma unroll
    for (int c = 0; c < (EX ? 8 : 1); ++c) acc2[c] = 0.0f;
    f32x16 acc0[NCB], acc1[NCB];  // WIDE: pixels 0-31 / 32-63 of the quadrant x 32 channels (per channel block)
#pragma unroll
    for (int c = 0; c < (WIDE ? 1 : CH); ++c) acc[c] = 0.0f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[cb][r] = acc1[cb][r] = 0.0f;
    const int wch = lane & 31, half = lane >> 5;
    const bool wch_ok = FULL || wch < nch;

    STAMP(0);
    // (Requesting the staging's two dependent loads — list id, then the 32-byte record — two / one chunks ahead of
    //  the walk measured 0.490 -> 0.480 ms on the pair build in r02, but costs the 9 registers that the five-waves
    //  build (96) does not have; the second array's colour rows a chunk ahead as well: slower, 0.498.  Removed in r03.)
    for (int base = range.x; base < range.y; base += 64) {
        if (__ballot(!done) == 0ull) break;
        const int e = base + lane;
        const int g_cur = e < range.y ? ids[e] : 0;
        const int cnt = stage_chunk<CH, WIDE, false, LIST>(L, lane, e, e < range.y, g_cur, rec, colors, C,
                                                           ch_off, nch, xlo, xhi, ylo, yhi, EX ? &seg2 : nullptr);
        WALK_STAT(0, min(64, range.y - base));
        STAMP(1);
        STAMP_BATCH();
        constexpr int fabl = FABL;
        if (fabl >= 4) { KEEP(cnt); continue; }
        for (int k = 0; k < cnt; k += GRP) {
            if (k > 0 && __ballot(!done) == 0ull) break;
            WALK_STAT(1, min(GRP, cnt - k));
            // LEANF (pair build): the records of a group are not all requested up front
            // (12 ds_read_b128 = 48 registers in flight) but two Gaussians at a time, and the list position is read
            // again where it is needed: a fifth wave per SIMD needs <= 96 registers.
            constexpr bool LEANF = WIDE && (EX || NCB > 1);
            float4 A[GRP], B[GRP], Cc[GRP], Cd[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                if (LEANF) continue;
                A[q] = L.a[GRP + k + q];
                B[q] = L.b[GRP + k + q];
                Cc[q] = L.c[GRP + k + q];
                if (N8) Cd[q] = L.d[GRP + k + q];
            }
            float colB[GRP / 2][NCB];
            if (WIDE) {
#pragma unroll
                for (int pr = 0; pr < GRP / 2; ++pr) {
                    // lanes 0-31 fetch the colour row of the even Gaussian, 32-63 of the odd one
                    const int gid = __builtin_bit_cast(int, L.c[GRP + k + 2 * pr + half].x);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        colB[pr][cb] = (wch_ok && fabl < 2) ? colors[(size_t)gid * C + ch_off + 32 * cb + wch] : 0.0f;
                }
            }
            float alpha[GRP];
            bool pass[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                if (LEANF) {
                    A[q] = L.a[GRP + k + q];
                    B[q] = L.b[GRP + k + q];
                }
                const float dx = A[q].x - px, dy = A[q].y - py;
                const float sigma = __builtin_fmaf(
                    0.5f, __builtin_fmaf(B[q].x * dx, dx, (B[q].z * dy) * dy), (B[q].y * dx) * dy);
                alpha[q] = fminf(GG_ALPHA_MAX_FWD, A[q].z * gg_expf_walk(-sigma));
                pass[q] = sigma >= 0.0f && !(alpha[q] < GG_ALPHA_MIN);
                if (LEANF && q == 1) __builtin_amdgcn_sched_barrier(0);
            }
            float vis[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const float next_T = T * (1.0f - alpha[q]);
                const bool live = pass[q] && !done;
                const bool stop = live && (next_T <= GG_T_EPS);
                const bool blend = live && !stop;
                vis[q] = blend ? alpha[q] * T : 0.0f;
#ifdef GG_WALK_STATS
                {
                    const unsigned long long bm = __ballot(blend), pm = __ballot(pass[q]), lm = __ballot(!done);
                    WALK_STAT(2, bm != 0ull);
                    WALK_STAT(3, __builtin_popcountll(lm));
                    WALK_STAT(4, __builtin_popcountll(pm));
                    WALK_STAT(5, __builtin_popcountll(bm));
                }
#endif
                T = blend ? next_T : T;
                last = blend ? __builtin_bit_cast(int, N8 ? A[q].w : (LEANF ? L.c[GRP + k + q].w : Cc[q].w)) : last;
                done = done || stop;
                if (EX && fabl < 3) {   // the second array's colours: read from the record right where they are used
                    const float4 xd = L.d[GRP + k + q], xe = L.e[GRP + k + q];
                    acc2[0] = __builtin_fmaf(xd.x, vis[q], acc2[0]);
                    acc2[EX ? 1 : 0] = __builtin_fmaf(xd.y, vis[q], acc2[EX ? 1 : 0]);
                    acc2[EX ? 2 : 0] = __builtin_fmaf(xd.z, vis[q], acc2[EX ? 2 : 0]);
                    acc2[EX ? 3 : 0] = __builtin_fmaf(xd.w, vis[q], acc2[EX ? 3 : 0]);
                    acc2[EX ? 4 : 0] = __builtin_fmaf(xe.x, vis[q], acc2[EX ? 4 : 0]);
                    acc2[EX ? 5 : 0] = __builtin_fmaf(xe.y, vis[q], acc2[EX ? 5 : 0]);
                    acc2[EX ? 6 : 0] = __builtin_fmaf(xe.z, vis[q], acc2[EX ? 6 : 0]);
                    acc2[EX ? 7 : 0] = __builtin_fmaf(xe.w, vis[q], acc2[EX ? 7 : 0]);
                }
                if (!WIDE) {
                    acc[0] = __builtin_fmaf(Cc[q].x, vis[q], acc[0]);
                    if (CH > 1) acc[CH > 1 ? 1 : 0] = __builtin_fmaf(Cc[q].y, vis[q], acc[CH > 1 ? 1 : 0]);
                    if (CH > 2) acc[CH > 2 ? 2 : 0] = __builtin_fmaf(Cc[q].z, vis[q], acc[CH > 2 ? 2 : 0]);
                    if (N8) {
                        acc[CH > 3 ? 3 : 0] = __builtin_fmaf(Cc[q].w, vis[q], acc[CH > 3 ? 3 : 0]);
                        acc[CH > 4 ? 4 : 0] = __builtin_fmaf(Cd[q].x, vis[q], acc[CH > 4 ? 4 : 0]);
                        acc[CH > 5 ? 5 : 0] = __builtin_fmaf(Cd[q].y, vis[q], acc[CH > 5 ? 5 : 0]);
                        acc[CH > 6 ? 6 : 0] = __builtin_fmaf(Cd[q].z, vis[q], acc[CH > 6 ? 6 : 0]);
                        acc[CH > 7 ? 7 : 0] = __builtin_fmaf(Cd[q].w, vis[q], acc[CH > 7 ? 7 : 0]);
                    }
                }
            }
            if (WIDE && fabl >= 1) {
#pragma unroll
                for (int q = 0; q < GRP; ++q) KEEP(vis[q]);
#pragma unroll
                for (int pr = 0; pr < GRP / 2; ++pr) KEEP(colB[pr][0]);
            } else if (WIDE) {
                // (handing these four dependent MFMAs to the pipe at the top of the NEXT iteration, in front of that
                // group's arithmetic, measures the same for the pair kernel and +8 % for the plain one: 20 more VGPRs)
#pragma unroll
                for (int pr = 0; pr < GRP / 2; ++pr) {
                    auto r = __builtin_amdgcn_permlane32_swap(
                        __builtin_bit_cast(unsigned, vis[2 * pr]),
                        __builtin_bit_cast(unsigned, vis[2 * pr + 1]), false, false);
                    // r[0]: pixels 0-31 x {even, odd} Gaussian ; r[1]: pixels 32-63 x {even, odd}
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        acc0[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, (unsigned)r[0]),
                                                                        colB[pr][cb], acc0[cb], 0, 0, 0);
                        acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, (unsigned)r[1]),
                                                                        colB[pr][cb], acc1[cb], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // list is rewritten by the next chunk
        STAMP(4);
    }
    STAMP(8);
#ifndef GG_EPI_SKIP
#define GG_EPI_SKIP 0   // diagnostic builds only (with GG_STAMPS): 1 no final_T / final_idx stores, 2 no image stores.
                        // A backward after a GG_EPI_SKIP=1 forward reads an uninitialised final_idx image: it faulted
                        // once (r02).  Since r03 the backward kernels hold final_idx to the tile's list range, and
                        // tools/stamps.py asks the build (gg_debug_epi_skip) and runs no backward on such a forward.
#endif
    if (GG_EPI_SKIP == 1) { KEEP(T); KEEP(last); }
    if (GG_EPI_SKIP != 1 && inside && write_final) {
        const size_t p = (size_t)i * img_w + j;
        final_T[p] = T;
        final_idx[p] = last;
    }
    if (EX && inside) {
        float *o2 = seg2.out_img + ((size_t)i * img_w + j) * seg2.C2;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < seg2.nch2) o2[c] = __builtin_fmaf(T, seg2.background[c], acc2[EX ? c : 0]);
    }
    if (!WIDE) {
        if (inside) {
            float *o = out_img + ((size_t)i * img_w + j) * C + ch_off;
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < nch) o[c] = __builtin_fmaf(T, background[ch_off + c], acc[c]);
        }
    } else if (FULL && (C % 4 == 0) && (ch_off % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_img) & 15) == 0)) {
        // The accumulators hold one channel (lane & 31) of 32 pixels per lane: written as they stand that is 32 dword
        // store instructions per wave, and a store costs by the instruction (in-kernel stamps, tools/stamps.py: 29 %
        // of a forward wave's lifetime was this epilogue).  Here 16 pixels x 32 channels at a time go through the
        // (dead) list memory and leave as float4: 8 store instructions of 1 KB each (one image row of the quadrant).
        float *buf = reinterpret_cast<float *>(&L);   // 2 KB of the wave's list
        const int chunk = lane & 7;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
        const int ch_cb = ch_off + 32 * cb;
        const float4 bg4 = *reinterpret_cast<const float4 *>(background + ch_cb + 4 * chunk);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {   // pixels 16 q4 .. 16 q4 + 15: registers 8 (q4 & 1) .. + 7 of acc0 / acc1
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 8 * (q4 & 1) + rr;
                const int pl = (rr & 3) + 8 * (rr >> 2) + 4 * half;   // pixel inside the 16
                buf[pl * 32 + wch] = (q4 >> 1) ? acc1[cb][r] : acc0[cb][r];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int pl = (lane >> 3) + 8 * t;
                const float4 v = *reinterpret_cast<const float4 *>(buf + pl * 32 + 4 * chunk);
                const int pq = 16 * q4 + pl;
                const float Tp = __shfl(T, pq, 64);
                const int pj = qx0 + (pq & 7), pi = qy0 + (pq >> 3);
                if (GG_EPI_SKIP == 2) { KEEP(v.x + v.y + v.z + v.w + Tp); continue; }
                if (pi < img_h && pj < img_w) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v o = {__builtin_fmaf(Tp, bg4.x, v.x), __builtin_fmaf(Tp, bg4.y, v.y),
                                   __builtin_fmaf(Tp, bg4.z, v.z), __builtin_fmaf(Tp, bg4.w, v.w)};
                    f4v *dst = reinterpret_cast<f4v *>(out_img + ((size_t)pi * img_w + pj) * C + ch_cb + 4 * chunk);
                    *dst = o;   // (a non-temporal store measures the same: 0.483 against 0.490 ms)
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        }
    } else {
        const float bgc = wch_ok ? background[ch_off + wch] : 0.0f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pq = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * half;  // pixel of this value
                const float Tp = __shfl(T, pq, 64);
                const int pj = qx0 + (pq & 7), pi = qy0 + (pq >> 3);
                if (pi < img_h && pj < img_w && wch_ok)
                    out_img[((size_t)pi * img_w + pj) * C + ch_off + wch] =
                        __builtin_fmaf(Tp, bgc, blk ? acc1[0][r] : acc0[0][r]);
            }
    }
    STAMP(7);
    STAMP_END();
}

// =============================================================================================
// forward, pair, BATCHED (round 4; gg_blend_fwd_pair_fast): the backward's architecture for the forward walk
// =============================================================================================
// The exact-order pair forward above contracts over TWO Gaussians per v_mfma_f32_32x32x2_f32 (64 cycles each: a quarter
// of the kernel's issue budget, r03 counters), spends 5.5 issue slots per pair on the v_permlane32_swap that builds its A
// operand and 8 fma per (pixel, Gaussian) on the second array.  Here the survivors of the quadrant cull are QUEUED
// (ascending list order) until 32 are there; the walk — the same sigma / exp / alpha / T arithmetic, operation for
// operation, so final_T, final_idx and every pass / stop decision keep their bits — only writes vis = alpha T into a
// slab [pixel][slot]; the batch then ends with ONE product
//     OUT[64 pixels x 48 channels] += VIS[64 x 32 slots] * COL[32 x 48]        (32 feature channels | <= 8 of the second
// array | padding) as 4 x 3 tiles of v_mfma_f32_16x16x32_f16 on fp16 TWO-PIECE operands (gg_common.h: x s = hi + lo,
// four piece products per tile, fp32 accumulation): 48 MFMAs of 16 cycles per 32 Gaussians instead of 32 of 64 cycles
// plus 256 fma.  Scales (powers of two, exact): vis x 2^15 (alpha T lies in [3.9e-7, 1)); the colours of a batch one
// scale per CHANNEL (largest |colour| of the 32 slots into [2^14, 2^15)), taken out of the tile's result before it is
// added to the fp32 accumulators.  Images therefore equal the exact-order kernel's to fp32 rounding, not bit for bit:
// per channel the error is <= ~2^-22 of (largest |colour| of the batch) x (sum of vis) — elements more than 2^12 below
// their channel's largest colour of the batch keep an ABSOLUTE error of 2^-40 of that colour (fp16 denormals) —
// against the exact kernel's own ~n 2^-24 of the sequential fp32 sum.  tests/test_fast_forward.py holds the images to
// 1e-6 (1 + |value|) of the oracle's and final_T / final_idx to its bits.
// LDS per wave: queue 3.2 KB + slab 8 KB (16-byte chunks of a pixel's 32 slots XOR-swizzled by the pixel: the walk's
// ds_write_b128 and the product's ds_read_b128 both spread over the banks).  One quadrant per 64-thread workgroup
// (as the wide backward): 14 waves per CU by LDS.
#define FB_SLOTS 32
#define FB_QCAP 100     // <= 31 left over + 64 staged + 4 null records behind the last survivor
#ifndef GG_FB_WAVES
#define GG_FB_WAVES 3
#endif
struct __attribute__((aligned(16))) FwdQueue {
    float4 a[FB_QCAP];   // x, y, opacity, list position + 1 (int bits)
    float4 b[FB_QCAP];   // conic a, b, c, Gaussian id (int bits)
};
#define GG_VIS_SCALE 32768.0f
#define GG_VIS_UNSCALE (1.0f / 32768.0f)

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GG_FB_WAVES))) void blend2_fwd_batch_kernel(
    int C, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *__restrict__ ids,
    const int2 *__restrict__ bins, const GRec *__restrict__ rec, const float *__restrict__ colors,
    const float *__restrict__ background, float *__restrict__ out_img, float *__restrict__ final_T,
    int32_t *__restrict__ final_idx, Seg2 seg2, unsigned bytes1, unsigned bytes2) {
    __shared__ FwdQueue s_q;
    __shared__ __attribute__((aligned(16))) float s_vis[64 * FB_SLOTS];
    int wave;
    const int tile = blend_tile_wave<1>(blockIdx.x, threadIdx.x, ntiles, wave);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    FwdQueue &Q = s_q;
    float *vis_w = s_vis;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const int2 range = bins[tile];
    const int sl = lane & 15, q4 = lane >> 4;

    // The walk carries Ts = 2^15 T instead of T: a multiplication by a power of two commutes with every rounding of the
    // recurrence (no overflow, no denormal: T in [1e-4, 1]), so next_Ts <= 2^15 T_EPS decides exactly as next_T <= T_EPS,
    // final_T = 2^-15 Ts is the same bits, and alpha Ts IS the slab's 2^15 alpha T — the product's A operands need no
    // scaling multiply (32 per batch and lane).
    float T = GG_VIS_SCALE;
    int last = range.x;
    bool done = !inside;
    bool alive = true;   // wave-uniform: some pixel of the quadrant still takes contributions
    // acc[pixel block of 16][channel block of 16]: pixel 16 blk + 4 q4 + r, channel 16 nb + sl — in units of
    // 2^15 / inv(csc[nb]): csc[nb] is this lane's channel's current colour scale, a power of two that only ever shrinks
    // (the largest |colour| seen so far in [2^14, 2^15)); the MFMAs accumulate in place and the accumulators are
    // rescaled (exactly) when a batch brings a larger colour — instead of one fma per accumulator and batch
    float csc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) csc[nb] = pow2_scale(0.0f);
    f32x4 acc[4][3];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[blk][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool ch2_ok = sl < seg2.nch2;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(colors), 0, (int)bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(seg2.colors), 0, (int)bytes2, 0x00020000);
    const unsigned rowb1 = 4u * (unsigned)C, rowb2 = 4u * (unsigned)seg2.C2;
    const unsigned lane_off1 = 4u * (unsigned)sl, lane_off2 = ch2_ok ? 4u * (unsigned)sl : 0u;

    // one batch: queue entries [base, base + n), n <= 32, null records behind the last one up to a multiple of GRP
    auto run_batch = [&](const int base, const int n) {
        // B operands COL[slot 8 q4 + t][channel 16 nb + sl]: requested here, used after the walk
        // (buffer loads: row offset = id x row bytes as ONE 24-bit multiply-add per load instead of 64-bit address
        //  arithmetic — the launcher takes this kernel only for arrays below 4 GB and ids below 2^24)
        float cb[3][8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int s_ = min(8 * q4 + t, n - 1);          // slots past the batch repeat its last Gaussian (vis = 0 there)
            const unsigned gid = (unsigned)__builtin_bit_cast(int, Q.b[base + s_].w);
            const unsigned o1 = __umul24(gid, rowb1) + lane_off1;
            cb[0][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, (int)o1, 0, 0));
            cb[1][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, (int)o1 + 64, 0, 0));
            const unsigned o2 = __umul24(gid, rowb2) + lane_off2;
            const float c2v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, (int)o2, 0, 0));
            cb[2][t] = ch2_ok ? c2v : 0.0f;
        }
        // the walk: vis[pixel = lane][slot] = alpha T of the pairs that blend, 0 otherwise
        int nw = 0;
        for (int g = 0; g < n; g += GRP) {
            if (__ballot(!done) == 0ull) { alive = false; break; }
            float vis[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const float4 A = Q.a[base + g + q], B = Q.b[base + g + q];
                const float dx = A.x - px, dy = A.y - py;
                const float sigma = __builtin_fmaf(
                    0.5f, __builtin_fmaf(B.x * dx, dx, (B.z * dy) * dy), (B.y * dx) * dy);
                const float alpha = fminf(GG_ALPHA_MAX_FWD, A.z * gg_expf_walk(-sigma));
                const bool pass = sigma >= 0.0f && !(alpha < GG_ALPHA_MIN);
                const float next_T = T * (1.0f - alpha);
                const bool live = pass && !done;
                const bool stop = live && (next_T <= GG_T_EPS * GG_VIS_SCALE);
                const bool blend = live && !stop;
                vis[q] = blend ? alpha * T : 0.0f;
                T = blend ? next_T : T;
                last = blend ? __builtin_bit_cast(int, A.w) : last;
                done = done || stop;
                if (q == 1) __builtin_amdgcn_sched_barrier(0);   // two Gaussians' records in flight at a time
            }
            *reinterpret_cast<float4 *>(vis_w + lane * FB_SLOTS + ((((g >> 2) ^ lane) & 7) << 2)) =
                make_float4(vis[0], vis[1], vis[2], vis[3]);
            nw = g + GRP;
        }
        for (int g = nw; g < FB_SLOTS; g += GRP)
            *reinterpret_cast<float4 *>(vis_w + lane * FB_SLOTS + ((((g >> 2) ^ lane) & 7) << 2)) =
                make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __builtin_amdgcn_wave_barrier();
        // B: two fp16 pieces of colour x csc (the channel's running scale); a batch with a larger colour shrinks the
        // scale and rescales the channel's accumulators first (rare after a wave's first batches: wave-uniform branch)
        h16x8 Bh[3], Bl[3];
        float fac[3];
        bool shrink = false;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            float m = fmaxf(fmaxf(fabsf(cb[nb][0]), fabsf(cb[nb][1])), fabsf(cb[nb][2]));
            m = fmaxf(fmaxf(m, fabsf(cb[nb][3])), fabsf(cb[nb][4]));
            m = fmaxf(fmaxf(m, fabsf(cb[nb][5])), fmaxf(fabsf(cb[nb][6]), fabsf(cb[nb][7])));
            {
                auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
                m = fmaxf(__builtin_bit_cast(float, (unsigned)r16[0]), __builtin_bit_cast(float, (unsigned)r16[1]));
                auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
                m = fmaxf(__builtin_bit_cast(float, (unsigned)r32[0]), __builtin_bit_cast(float, (unsigned)r32[1]));
            }
            const float sb = fminf(csc[nb], pow2_scale(m));
            fac[nb] = sb * pow2_inv(csc[nb]);      // 1, or the power of two < 1 the accumulators shrink by
            shrink = shrink || sb != csc[nb];
            csc[nb] = sb;
            unsigned h_[4], l_[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) split2h(cb[nb][2 * t] * sb, cb[nb][2 * t + 1] * sb, h_[t], l_[t]);
            Bh[nb] = H8(h_[0], h_[1], h_[2], h_[3]);
            Bl[nb] = H8(l_[0], l_[1], l_[2], l_[3]);
        }
        if (__ballot(shrink) != 0ull) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[blk][nb][r] *= fac[nb];
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            // A: 2^15 VIS[pixel 16 blk + sl][slot 8 q4 + 0..7]
            const int P = 16 * blk + sl;
            const float4 v0 = *reinterpret_cast<const float4 *>(vis_w + P * FB_SLOTS + ((((2 * q4) ^ P) & 7) << 2));
            const float4 v1 = *reinterpret_cast<const float4 *>(vis_w + P * FB_SLOTS + ((((2 * q4 + 1) ^ P) & 7) << 2));
            unsigned ah[4], al[4];
            split2h(v0.x, v0.y, ah[0], al[0]);
            split2h(v0.z, v0.w, ah[1], al[1]);
            split2h(v1.x, v1.y, ah[2], al[2]);
            split2h(v1.z, v1.w, ah[3], al[3]);
            const h16x8 Ah = H8(ah[0], ah[1], ah[2], ah[3]), Al = H8(al[0], al[1], al[2], al[3]);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                acc[blk][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bl[nb], acc[blk][nb], 0, 0, 0);
                acc[blk][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh[nb], acc[blk][nb], 0, 0, 0);
                acc[blk][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl[nb], acc[blk][nb], 0, 0, 0);
                acc[blk][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh[nb], acc[blk][nb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next batch's walk
    };

    int qn = 0;
    // list ids one chunk ahead of the cull.  (Records a chunk ahead as well — 8 more registers — measured nothing:
    // 0.424 ms at three waves per SIMD, 0.407 at four with 3 spills, against 0.409 without; the kernel is VALU-bound,
    // VALUBusy 87 %.)
    int g_nxt = (range.x + lane < range.y) ? ids[range.x + lane] : 0;
    for (int cbase = range.x; cbase < range.y; cbase += 64) {
        if (__ballot(!done) == 0ull) { alive = false; break; }
        const int e = cbase + lane;
        const bool valid = e < range.y;
        const int g = g_nxt;
        g_nxt = (e + 64 < range.y) ? ids[e + 64] : 0;
        const float4 ra = reinterpret_cast<const float4 *>(rec + g)[0];
        const float4 rb = reinterpret_cast<const float4 *>(rec + g)[1];
        int qxl = qx0, qyl = qy0;
        asm volatile("" : "+s"(qxl), "+s"(qyl));
        const bool hit = valid && rec_hits_rect(ra, rb, (float)qxl, (float)(qxl + 7), (float)qyl, (float)(qyl + 7));
        const uint64_t m = __ballot(hit);
        if (hit) {   // ascending list order
            const int pos = qn + lane_prefix(m);
            Q.a[pos] = make_float4(ra.x, ra.y, ra.z, __builtin_bit_cast(float, e + 1));
            Q.b[pos] = make_float4(rb.x, rb.y, rb.z, __builtin_bit_cast(float, g));
        }
        qn += __builtin_popcountll(m);
        __builtin_amdgcn_wave_barrier();
        int done_n = 0;
        while (qn - done_n >= FB_SLOTS && alive) {
            run_batch(done_n, FB_SLOTS);
            done_n += FB_SLOTS;
        }
        if (!alive) break;
        if (done_n > 0) {   // bring the left-over (< 32) to the front; source and destination do not overlap
            const int left = qn - done_n;
            if (lane < left) { const float4 ta = Q.a[done_n + lane]; __builtin_amdgcn_wave_barrier(); Q.a[lane] = ta; }
            __builtin_amdgcn_wave_barrier();
            if (lane < left) { const float4 tb = Q.b[done_n + lane]; __builtin_amdgcn_wave_barrier(); Q.b[lane] = tb; }
            qn = left;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (alive && qn > 0) {   // what is left at the end of the list (< 32): null records behind it, one last batch
        if (lane < GRP) {
            Q.a[qn + lane] = make_float4(0.f, 0.f, 0.f, 0.f);                                 // opacity 0: never passes
            Q.b[qn + lane] = make_float4(0.f, 0.f, 0.f, Q.b[qn - 1].w);
        }
        __builtin_amdgcn_wave_barrier();
        run_batch(0, qn);
    }

    T *= GG_VIS_UNSCALE;   // (exact)
    if (inside) {
        const size_t p = (size_t)i * img_w + j;
        final_T[p] = T;
        final_idx[p] = last;
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const float un = pow2_inv(csc[nb]) * GG_VIS_UNSCALE;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[blk][nb][r] *= un;
    }
    // first array: the accumulators go through the slab ([pixel][32 channels]) and leave as float4, one image row of the
    // quadrant (1 KB) per store instruction
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) vis_w[(16 * blk + 4 * q4 + r) * 32 + 16 * nb + sl] = acc[blk][nb][r];
    __builtin_amdgcn_wave_barrier();
    {
        const int chunk = lane & 7;
        const float4 bg4 = *reinterpret_cast<const float4 *>(background + 4 * chunk);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int pq = (lane >> 3) + 8 * t;
            const float4 v = *reinterpret_cast<const float4 *>(vis_w + pq * 32 + 4 * chunk);
            const float Tp = __shfl(T, pq, 64);
            const int pj = qx0 + (pq & 7), pi = qy0 + (pq >> 3);
            if (pi < img_h && pj < img_w) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v o = {__builtin_fmaf(Tp, bg4.x, v.x), __builtin_fmaf(Tp, bg4.y, v.y),
                               __builtin_fmaf(Tp, bg4.z, v.z), __builtin_fmaf(Tp, bg4.w, v.w)};
                *reinterpret_cast<f4v *>(out_img + ((size_t)pi * img_w + pj) * C + 4 * chunk) = o;
            }
        }
    }
    // second array: [pixel][8] through the slab, every lane stores its own pixel's channels
    __builtin_amdgcn_wave_barrier();
    if (sl < 8) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) vis_w[(16 * blk + 4 * q4 + r) * 8 + sl] = acc[blk][2][r];
    }
    __builtin_amdgcn_wave_barrier();
    if (inside) {
        const float4 w0 = *reinterpret_cast<const float4 *>(vis_w + lane * 8);
        const float4 w1 = *reinterpret_cast<const float4 *>(vis_w + lane * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        float *o2 = seg2.out_img + ((size_t)i * img_w + j) * seg2.C2;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < seg2.nch2) o2[c] = __builtin_fmaf(T, seg2.background[c], wv[c]);
    }
}

// =============================================================================================
// backward, narrow (<= 3 channels): wave-autonomous, no workgroup barrier, no LDS slab
// =============================================================================================
// Every wave walks its quadrant's part of the tile list back to front on its own, four survivors
// per iteration.  All CH+6 partial gradients of the four Gaussians of a group go through ONE
// register butterfly (Red6<4*(CH+6)>), which leaves each of the 4*(CH+6) totals on its own lane;
// those lanes add them straight into the gradient rows with ONE global float-atomic
// wave-instruction per group (4 Gaussians x (CH+6) consecutive-ish floats).
// Measured alternatives (profiles/README.md): combining the four waves of a tile through an LDS
// slab with two __syncthreads() per chunk, and a wave-private slab flushed per chunk, were both
// slower — the per-chunk flush loop costs more instructions than the atomics it saves.
// ABL > 0: ablation builds for the measurement harness (tools/kbench.py).  They are instantiated only
// under -DGG_ABLATION, i.e. in libgg_raster_abl.so, never in the product library:
//   1 = no global atomics, 2 = also no butterfly, 3 = also no recurrence/partials (geometry only),
//   4 = staging + cull only (no group loop)
template <int CH, int ABL = 0, bool DET = false>
__global__ __launch_bounds__(64 * GG_WPB_OTHER) void blend2_bwd_narrow_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_out, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity, int gstride, int cstride,
    DetSlab det = DetSlab()) {
    constexpr int K = CH + 6;       // per-Gaussian values: CH colours, xy(2), conic(3), opacity(1)
    constexpr int KB = GRP * K;     // butterfly width
    using R = Red6<KB>;
    constexpr bool N8 = CH > 3;
    typedef WaveListT<N8 ? 2 : 1> LIST;
    __shared__ LIST lists[GG_WPB_OTHER];

    int wave;
    const int tile = blend_tile_wave<GG_WPB_OTHER>(blockIdx.x, threadIdx.x, ntiles, wave);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    const int wslot = wave & (GG_WPB_OTHER - 1);   // this wave's LDS
    LIST &L = lists[wslot];
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const float xlo = (float)qx0, xhi = (float)(qx0 + 7), ylo = (float)qy0, yhi = (float)(qy0 + 7);
    const int2 range = bins[tile];
    const size_t p = inside ? ((size_t)i * img_w + j) : 0;

    const float T_final = inside ? final_T[p] : 1.0f;
    // final_idx comes from the caller (public C ABI): a stale or garbage value must not index the list
    const int fin = inside ? min(max(final_idx[p], range.x), range.y) : range.x;
    float T = T_final;
    float vo[CH];
    float W;
    {
        float Bsum = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool chan = !N8 || c < nch;      // the 8-wide kernel also serves 4..7 channels
            vo[c] = (inside && chan) ? v_out[p * C + ch_off + c] : 0.0f;
            if (chan) Bsum = __builtin_fmaf(background[ch_off + c], vo[c], Bsum);
        }
        W = T_final * Bsum;
    }
    int hi = fin;  // this wave's upper end of the walk
    for (int off = 32; off > 0; off >>= 1) hi = max(hi, __shfl_xor(hi, off, 64));
    hi = __builtin_amdgcn_readfirstlane(hi);

    // which (group member q, value k) does this lane own after the butterfly, and where does it go?
    bool owner;
    const int myvar = R::var(lane, owner);
    const int my_q = myvar / K, my_k = myvar - my_q * K;
    float *my_base;     // gradient array of my value
    int my_stride;      // floats per Gaussian in that array
    // gstride / cstride: floats between consecutive Gaussians in the geometry / colour gradient
    // arrays (0 = dense 2 | 3 | 1 | C); interleaved records put all of a Gaussian's atomics on one line
    if (my_k < CH) {
        my_base = v_colors + ch_off + my_k;
        my_stride = cstride ? cstride : C;
        owner = owner && (!N8 || my_k < nch);      // channels past nch belong to the next row
    }
    else if (my_k < CH + 2) { my_base = v_xy + (my_k - CH); my_stride = gstride ? gstride : 2; }
    else if (my_k < CH + 5) { my_base = v_conic + (my_k - CH - 2); my_stride = gstride ? gstride : 3; }
    else { my_base = v_opacity; my_stride = gstride ? gstride : 1; }

    // 8-channel build: the first of the two dependent staging loads (id, then record) runs one chunk ahead
    // (-2.5 %; on the <= 3-channel builds the extra register costs a wave of occupancy and it measures +1 %)
    int g_nxt = (N8 && hi > range.x && hi - 64 + lane >= range.x) ? ids[hi - 64 + lane] : 0;
    for (int top = hi; top > range.x; top -= 64) {
        const int e = top - 64 + lane;
        const bool valid = e >= range.x;
        int g_cur;
        if (N8) {
            g_cur = g_nxt;
            g_nxt = (top - 64 > range.x && e - 64 >= range.x) ? ids[e - 64] : 0;
        } else {
            g_cur = valid ? ids[e] : 0;
        }
        const int cnt = stage_chunk<CH, false, true, LIST>(L, lane, e, valid, g_cur, rec, colors, C, ch_off,
                                                           N8 ? nch : CH, xlo, xhi, ylo, yhi);
        const int fin_rel = fin - (top - 64);  // entries at chunk position >= fin_rel are not mine
        if (ABL >= 4) { KEEP(cnt); continue; }
        for (int kk = GRP + cnt - 1; kk >= GRP; kk -= GRP) {
            float4 A[GRP], B[GRP], Cc[GRP], Cd[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                A[q] = L.a[kk - q];
                B[q] = L.b[kk - q];
                Cc[q] = L.c[kk - q];
                if (N8) Cd[q] = L.d[kk - q];
            }
            float vis[GRP], alpha[GRP], dxs[GRP], dys[GRP];
            bool pass[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const float dx = A[q].x - px, dy = A[q].y - py;
                dxs[q] = dx;
                dys[q] = dy;
                const float sigma = __builtin_fmaf(
                    0.5f, __builtin_fmaf(B[q].x * dx, dx, (B[q].z * dy) * dy), (B[q].y * dx) * dy);
                vis[q] = gg_expf_walk(-sigma);
                alpha[q] = fminf(GG_ALPHA_MAX_BWD, A[q].z * vis[q]);
                // .w = position inside the chunk; null pads have opacity 0 -> alpha 0 -> no pass
                pass[q] = (__builtin_bit_cast(int, N8 ? A[q].w : Cc[q].w) < fin_rel) && sigma >= 0.0f &&
                          !(alpha[q] < GG_ALPHA_MIN);
            }
            if (__ballot(pass[0] || pass[1] || pass[2] || pass[3]) == 0ull) continue;
            if (ABL >= 3) {
#pragma unroll
                for (int q = 0; q < GRP; ++q) { KEEP(vis[q]); KEEP(alpha[q]); KEEP((int)pass[q]); }
                continue;
            }

            float part[KB];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const float ra_ = __builtin_amdgcn_rcpf(1.0f - alpha[q]);
                const float Tn = T * ra_;
                const float fac = pass[q] ? alpha[q] * Tn : 0.0f;
                float D = Cc[q].x * vo[0];
                if (CH > 1) D = __builtin_fmaf(Cc[q].y, vo[CH > 1 ? 1 : 0], D);
                if (CH > 2) D = __builtin_fmaf(Cc[q].z, vo[CH > 2 ? 2 : 0], D);
                if (N8) {
                    D = __builtin_fmaf(Cc[q].w, vo[CH > 3 ? 3 : 0], D);
                    D = __builtin_fmaf(Cd[q].x, vo[CH > 4 ? 4 : 0], D);
                    D = __builtin_fmaf(Cd[q].y, vo[CH > 5 ? 5 : 0], D);
                    D = __builtin_fmaf(Cd[q].z, vo[CH > 6 ? 6 : 0], D);
                    D = __builtin_fmaf(Cd[q].w, vo[CH > 7 ? 7 : 0], D);
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) part[q * K + c] = fac * vo[c];
                const float v_alpha = pass[q] ? (Tn * D - ra_ * W) : 0.0f;
                W = pass[q] ? __builtin_fmaf(D, fac, W) : W;
                T = pass[q] ? Tn : T;
                const float v_sigma = pass[q] ? (-A[q].z * vis[q]) * v_alpha : 0.0f;  // vis may be inf
                const float dx = dxs[q], dy = dys[q];
                float *pg = part + q * K + CH;
                pg[0] = v_sigma * (B[q].x * dx + B[q].y * dy);
                pg[1] = v_sigma * (B[q].y * dx + B[q].z * dy);
                const float hs = 0.5f * v_sigma;
                pg[2] = (hs * dx) * dx;
                pg[3] = (hs * dx) * dy;
                pg[4] = (hs * dy) * dy;
                pg[5] = pass[q] ? vis[q] * v_alpha : 0.0f;
            }
            if (ABL >= 2) {
#pragma unroll
                for (int v = 0; v < KB; ++v) KEEP(part[v]);
                continue;
            }
            const float mine = R::run(part, lane);
            // Gaussian id of my group member: per-lane LDS read (4 distinct addresses per wave);
            // id bits live in b.w of the record
            const int my_gid = __builtin_bit_cast(int, L.b[kk - my_q].w);
            if (ABL >= 1) { KEEP(mine); KEEP(my_gid); continue; }
            if (DET) {   // deterministic mode: the total of (list entry, quadrant) goes to the slab
                const int pos = __builtin_bit_cast(int, N8 ? L.a[kk - my_q].w : L.c[kk - my_q].w);
                const size_t e = (size_t)(top - 64 + pos);
                if (owner && my_gid >= 0)
                    det.p[(e * 4 + wave) * det.ks + (my_k < CH ? det.coff + my_k : det.goff + (my_k - CH))] = mine;
                continue;
            }
            if (owner && my_gid >= 0 && mine != 0.0f) atomicAdd(my_base + (size_t)my_gid * my_stride, mine);
        }
        __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next chunk
    }
}

#define B2_SLOTS 32
// fac / D slab [slot][pixel]: 65 floats per slot — a column access (32 slots, one pixel: the MFMA operand
// layouts) and a row access both spread over the banks, and every address stays base + immediate (an XOR
// swizzle at stride 64 costs one live address register per access: 134 spilled VGPRs)
// (16-slot builds, measured r03: at stride 66 the flushes' A operands — lane = (slot = lane & 15, pixel 4 t +
//  (lane >> 4)) — sit in 32 distinct banks per half-wave instead of colliding pairwise (SQ_LDS_BANK_CONFLICT: 41 M
//  cycles per launch of the pair backward at 65); the kernel's time does not move: 0.970 / 0.980 against 0.966 / 0.976 ms)
#ifndef GG_S16_STRIDE
#define GG_S16_STRIDE 65
#endif
#define FIDX(slot, pix) ((slot) * FS + (pix))

// (split2h / pow2_scale / pow2_inv / H8: gg_common.h, "fp32 products on the fp16 matrix rate")
#define GG_FAC_SCALE 32768.0f          // fac = alpha T in [3.9e-7, 0.99]: x 2^15 before the split
#define GG_FAC_UNSCALE (1.0f / 32768.0f)
#ifndef GG_F16_SWAPMAX
#define GG_F16_SWAPMAX 1   // the slot's row maximum across its four lanes: v_permlane16/32_swap (0: ds_bpermute; measured +2 %)
#endif
#ifndef GG_F16_NLL
#define GG_F16_NLL 0       // 1: without the lo x lo piece products (2^-24 each; measured -1.5 % of the kernel: not taken)
#endif
// quadrants without any cotangent of the first array (feat_any, below): its flush is skipped (measured: training
// iteration -1.6 %, dense bench view +-0; skipping its colour rows and D k-steps as well measured +4 % on the dense
// view: profiles/r03_featany_d_experiment.patch)
#ifndef GG_FEATANY_FLUSH
#define GG_FEATANY_FLUSH 1
#endif
#ifndef GG_MG_MOMENTS
#define GG_MG_MOMENTS 1   // merged-flush builds: geometry sums as moments about the Gaussian's centre (0: the per-pixel form)
#endif
#ifndef GG_S16_F16
#define GG_S16_F16 1    // 0: the 16-slot backward's products on v_mfma_f32_16x16x4_f32 (the build before)
#endif
#ifndef GG_S16_WAVES
#define GG_S16_WAVES 4    // waves per SIMD the 16-slot backward is compiled for (128 registers)
#endif
// r04: the pair build's SECOND flush (FAC[16 x 64] V_OUT2[64 x 8]) on fp16 two-piece operands too: its A pieces are the
// first flush's (same FAC tile, same layout), its B pieces — the second array's cotangent tile, 64 pixels x 8 channels,
// carrying s_w — are split from the tile in LDS right where they are used: 8 v_mfma_f32_16x16x32_f16 (128 matrix cycles)
// instead of 16 v_mfma_f32_16x16x4_f32 (512 of the batch's 1 280) for 48 more vector instructions per batch.
// Measured (tools/pairbench.py, interleaved variants, bench view): 0.7825 -> 0.7604 ms.  The B pieces split ONCE per wave
// and held in 16 registers across the walk: 25 spilled registers at four waves per SIMD, 0.8898 ms — not kept.
#ifndef GG_F2_F16
#define GG_F2_F16 1   // 0: the fp32 second flush of round 3
#endif

// =============================================================================================
// backward, wide (32-channel chunk): wave-autonomous, matrix pipe for D = <colour, v_out> AND for the colour
// gradients; survivors queued to full batches of 32
// =============================================================================================
// The previous generation (round 1 .. mid round 2: colour row of every Gaussian in SGPRs through scalar loads,
// 32 fma for D on the VALU, fac parked per contributing Gaussian, one MFMA flush per 32 of them) spent 39 % of
// its wave-cycles in s_waitcnt — every Gaussian waited for its colour row — and issued 37 % of the VALU peak
// (profiles/r02_pmc.csv as of commit 22193a5).  Here the survivors of the quadrant cull are QUEUED in LDS
// (processing order = descending list order) until 28..32 are there; for a batch
//     D[64 pixels x 32 Gaussians] = V_OUT[64 x CH] * COLOUR^T[CH x 32]
// runs as 2 x CH/2 v_mfma_f32_32x32x2_f32 (B operand: every lane loads half a colour row of ITS Gaussian with
// vector loads, no scalar loads, nothing waited for inside the walk) and lands in the fac slab [slot][pixel];
// the walk reads D with one ds_read per Gaussian, writes fac over it, and the batch ends with the same
// FAC * V_OUT flush as before.  Per Gaussian the VALU loses the 32 fma and the colour-row bookkeeping; the
// matrix pipe (idle otherwise) takes 2 x 64 cycles per Gaussian.
//
// Queue capacity: 27 left over + 64 staged (last index 90); the null records sit behind <= 27 left over.  Every byte counts:
// with the second array's v_out tile the workgroup needs 53 248 B of LDS, and three workgroups per CU fit only
// up to there (the allocation granule; 54 272 B measured 2 per CU, whatever the occupancy API says)
#define BQ_CAP 92

template <int CAP>
struct __attribute__((aligned(16))) WaveQueueT {
    float4 a[CAP];   // x, y, opacity, list position (int bits)
    float4 b[CAP];   // conic a, b, c, Gaussian id (int bits; -1 = null record)
};
typedef WaveQueueT<BQ_CAP> WaveQueue;

// second colour array (<= 8 channels) whose backward rides on the walk of a 32-channel chunk (gg_blend_bwd_pair)
struct Seg2B {
    const float *colors;      // (N, C2)
    const float *background;  // (C2,)
    const float *v_out[3];    // the (H, W, C2) cotangent, as up to three images of vo_w[k] consecutive channels
    int vo_w[3];              // (the caller's rgb | depth | normal cotangents are read where they are)
    float *v_colors;          // gradient rows, cs2 floats apart
    int C2, nch2, cs2;
};

// EX: the backward of a second colour array of <= 8 channels (Seg2B) in the same walk: alpha, T and the geometry
// partials are those of ALL channels together (v_alpha is linear in <colour, v_out>), so D gets 4 more k-steps,
// the second array's v_out tile (64 pixels x 8) sits in LDS, and its colour gradients are a second, small flush
// (FAC[32 x 64] * V_OUT2[64 x 8] as v_mfma_f32_16x16x4_f32).  What the plugin route's feature | rgb+depth+normal
// operator uses: one backward walk per view instead of two.
// (A build for four workgroups per CU that re-read the cotangents per batch instead of holding them — 128 VGPRs,
// batches of 28 slots — spilled 81 registers in the walk and ran 1.50 ms against 1.00: removed in r03; the 16-slot
// build below took the fourth wave instead.)
// S16 (experiment for FOUR waves per SIMD, -DGG_BWD_S16=1): batches of 16 slots on v_mfma_f32_16x16x4_f32 — D
// accumulators 16 registers instead of 32, flush accumulators 8 instead of 16, slab 4 160 B instead of 8 320, the walk
// re-reading its records.  k-step s of lane group q = lane >> 4 is channel 8 q + s, so a lane's B operand is still two
// float4 loads of its Gaussian's colour row.  Needs 16-byte aligned colour rows and cotangent rows (the launcher
// checks) and the full 32-channel chunk.
// MG (round 3, the pair build with a 16-float gradient record per Gaussian — geometry 0..5 | second array 6..6 + C2 — on
// a 64-byte boundary): the geometry sums of a batch are parked in LDS by the walk and leave together with the second
// array's colour gradients, ONE atomic request per Gaussian and batch instead of three to four.  Float atomics execute
// at the L2 at ~20 G 64-byte requests/s chip-wide whatever the lanes of an instruction cover (tools/ubench_atomics.hip),
// and the pair backward ran at 17 G/s (TCC_EA0_ATOMIC 16.7 M per launch): per batch of 16 Gaussians 32 requests for the
// feature rows, ~22 for the 7 floats at offset 24 of 52-byte rows, ~21 for the 6 geometry floats.
template <bool FULL, int ABL = 0, int CHD = 32, bool DET = false, bool EX = false, bool S16 = false, bool MG = false>
__global__ __launch_bounds__(64 * GG_WPB_WIDE_BWD) __attribute__((amdgpu_waves_per_eu(S16 ? GG_S16_WAVES : 3))) void blend2_bwd_wide_kernel(
    int C, int ch_off, int nch, int img_h, int img_w, int tiles_x, int ntiles,
    const int32_t *__restrict__ ids, const int2 *__restrict__ bins, const GRec *__restrict__ rec,
    const float *__restrict__ colors, const float *__restrict__ background,
    const float *__restrict__ final_T, const int32_t *__restrict__ final_idx,
    const float *__restrict__ v_out, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity, int gstride, int cstride,
    DetSlab det = DetSlab(), Seg2B seg2 = Seg2B()) {
    static_assert(!MG || (S16 && EX), "merged record flush: the 16-slot pair build");
    static_assert(!EX || (FULL && CHD == 32 && !DET), "the second array rides on the full 32-channel build");
    static_assert(!S16 || (FULL && CHD == 32 && !DET && ABL == 0), "S16: the full 32-channel build");
    constexpr int NSLOT = S16 ? 16 : B2_SLOTS;
    constexpr int CH = CHD;
    constexpr int KS = CHD / 2;     // k-steps of the D product; lane half h supplies channels [KS h, KS h + KS)
    constexpr int KG = 6;
    constexpr int KB = GRP * KG;
    constexpr bool REREAD = S16;   // the 16-slot builds read a group's records again in the second pass (28 registers)
    // (Deferring a batch's float atomics to the start of the next batch's walk, so that no load waits behind them in
    //  the in-order memory counter, measured no gain in r02 — pair 1.064 against 1.051 ms: the waits are load latency,
    //  not the counter's ordering — and was removed in r03.)
    using R = Red6<KB>;
    typedef WaveQueueT<BQ_CAP> QUEUE;
    __shared__ QUEUE queues[GG_WPB_WIDE_BWD];
    constexpr int FS = S16 ? GG_S16_STRIDE : 65;   // slab row stride in floats (FIDX)
    __shared__ float s_fac[GG_WPB_WIDE_BWD][NSLOT * FS];
    __shared__ int s_slote[GG_WPB_WIDE_BWD][DET ? B2_SLOTS : 1];
    __shared__ __attribute__((aligned(16))) float s_vt[GG_WPB_WIDE_BWD][EX ? 64 * 8 + 8 : 4];   // EX: V_OUT2[pixel][8] (+ F2: 8 column scales)
    __shared__ float s_geo[GG_WPB_WIDE_BWD][MG ? 16 * 8 : 1];   // MG: the batch's geometry sums [slot][8]

    int wave;
    const int tile = blend_tile_wave<GG_WPB_WIDE_BWD>(blockIdx.x, threadIdx.x, ntiles, wave);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    const int wslot = wave & (GG_WPB_WIDE_BWD - 1);   // this wave's LDS
    float *vt = s_vt[wslot];
    QUEUE &Q = queues[wslot];
    float *fac_w = s_fac[wslot];
    float *geo_w = s_geo[wslot];
    int *slote = s_slote[wslot];
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int qx0 = tx * GG_BLOCK + (wave & 1) * 8, qy0 = ty * GG_BLOCK + (wave >> 1) * 8;
    const int j = qx0 + (lane & 7), i = qy0 + (lane >> 3);
    const bool inside = (i < img_h) && (j < img_w);
    const float px = (float)j, py = (float)i;
    const int2 range = bins[tile];
    const size_t p = inside ? ((size_t)i * img_w + j) : 0;

    STAMP_DECL;
    const float T_final = inside ? final_T[p] : 1.0f;
    // final_idx comes from the caller (public C ABI): a stale or garbage value must not index the list
    const int fin = inside ? min(max(final_idx[p], range.x), range.y) : range.x;
    float T = T_final;
    float W;
    const int wch = lane & 31, half = lane >> 5;
    const bool wch_ok = FULL || wch < nch;
    // The quadrant's cotangents V_OUT[64 pixels x CH] are needed three ways: <background, v_out> per pixel (lane =
    // pixel), the D product's A operands (pixel-major, half a row per lane) and the flush's B operands (lane =
    // channel).  Read straight from the image, each of the three is ~32 load instructions of 32-64 cache lines
    // each (lanes 128 bytes apart): in-kernel stamps (tools/stamps.py) showed 31-35 % of a wave's lifetime in this
    // prologue.  TILE: the 8 KB tile is loaded ONCE, 1 KB per instruction (one image row of the quadrant, float4
    // per lane), parked in the (still unused) fac slab — [pixel][32] with the 16-byte chunks of a row XOR-swizzled
    // by the pixel — and the three views are read from there.
    const bool tile_lds = FULL && CH == 32 && (C % 4 == 0) && (ch_off % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(v_out) & 15) == 0);   // wave-uniform
    float voa_keep[2][S16 ? 1 : KS], vob_keep[S16 ? 1 : 32];
    float va16[S16 ? 4 : 1][S16 ? 8 : 1], vb16[S16 ? 16 : 1][S16 ? 2 : 1];   // S16: A operands of D, B operands of the flush
    // F16: the same elements as two fp16 pieces each (packed pairs): D's A operands V_OUT[pixel 16 blk + (lane & 15)]
    // [channel 8 (lane >> 4) + 0..7] x s_w, the flush's B operands V_OUT[pixel 32 ks + 8 (lane >> 4) + 0..7][channel
    // 16 nb + (lane & 15)] x s_c.  s_w: one power of two for the quadrant's cotangents (all channels of both arrays),
    // s_c: one per channel of this lane; the raw values above are dead after the prologue.
    constexpr bool F16 = S16 && (GG_S16_F16 != 0);
    constexpr bool F2 = F16 && EX && (GG_F2_F16 != 0);
    unsigned vah[F16 ? 4 : 1][F16 ? 4 : 1], val[F16 ? 4 : 1][F16 ? 4 : 1];
    unsigned vbh[F16 ? 2 : 1][F16 ? 2 : 1][F16 ? 4 : 1], vbl[F16 ? 2 : 1][F16 ? 2 : 1][F16 ? 4 : 1];
    float sw = 1.0f, inv_sw = 1.0f, inv_scf[2] = {1.0f, 1.0f};
    float Bsum = 0.0f;
    // A wave starts with a chain of dependent loads: tile range -> final_idx -> (wave maximum) -> list ids -> records.
    // The ids of the first chunk are requested as soon as final_idx is there, i.e. BEFORE the cotangent tile is waited
    // for and re-laid, and the first records right after that: two of the chain's latencies run beside the tile's.
    auto wave_hi = [&]() {
        int h = fin;
        for (int off = 32; off > 0; off >>= 1) h = max(h, __shfl_xor(h, off, 64));
        return __builtin_amdgcn_readfirstlane(h);
    };
    auto load_id = [&](int top_) {
        const int e_ = top_ - 64 + lane;
        return (top_ > range.x && e_ >= range.x) ? ids[e_] : 0;
    };
    int hi, g_first;
    // records of the NEXT chunk, requested between a batch's walk and its flush: the vector memory counter is
    // in order, so loads issued behind the flush's 24-32 atomic instructions wait for every one of them
    float4 ra_p = make_float4(0.f, 0.f, 0.f, 0.f), rb_p = ra_p;
    if (S16) {   // the tile goes through the (16-slot) slab in two halves of 32 pixels
        hi = wave_hi();
        g_first = load_id(hi);
        const int sl = lane & 15, q4 = lane >> 4;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float4 rowv[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pi = qy0 + 4 * hh + rr, pj = qx0 + (lane >> 3);
                rowv[rr] = (pi < img_h && pj < img_w)
                               ? *reinterpret_cast<const float4 *>(v_out + ((size_t)pi * img_w + pj) * C + ch_off + 4 * (lane & 7))
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {   // local pixel 8 rr + (lane >> 3), chunk (lane & 7) ^ (pixel & 7)
                const int pl = 8 * rr + (lane >> 3);
                *reinterpret_cast<float4 *>(fac_w + pl * 32 + 4 * ((lane & 7) ^ (pl & 7))) = rowv[rr];
            }
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == hh) {   // <background, v_out> of this half's pixels
                const int pl = lane & 31;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 v = *reinterpret_cast<const float4 *>(fac_w + pl * 32 + 4 * (k ^ (pl & 7)));
                    Bsum = __builtin_fmaf(background[ch_off + 4 * k], v.x, Bsum);
                    Bsum = __builtin_fmaf(background[ch_off + 4 * k + 1], v.y, Bsum);
                    Bsum = __builtin_fmaf(background[ch_off + 4 * k + 2], v.z, Bsum);
                    Bsum = __builtin_fmaf(background[ch_off + 4 * k + 3], v.w, Bsum);
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {   // A operands of D: pixel 16 (2 hh + b) + sl, channels 8 q4 .. + 7
                const int pl = 16 * b + sl;
#pragma unroll
                for (int jq = 0; jq < 2; ++jq) {
                    const float4 v = *reinterpret_cast<const float4 *>(fac_w + pl * 32 + 4 * ((2 * q4 + jq) ^ (pl & 7)));
                    va16[S16 ? 2 * hh + b : 0][S16 ? 4 * jq : 0] = v.x;
                    va16[S16 ? 2 * hh + b : 0][S16 ? 4 * jq + 1 : 0] = v.y;
                    va16[S16 ? 2 * hh + b : 0][S16 ? 4 * jq + 2 : 0] = v.z;
                    va16[S16 ? 2 * hh + b : 0][S16 ? 4 * jq + 3 : 0] = v.w;
                }
            }
#pragma unroll
            for (int ss = 0; ss < 8; ++ss) {   // B operands of the flush: pixel 4 (8 hh + ss) + q4, channel 16 nb + sl
                const int pl = F16 ? 8 * q4 + ss : 4 * ss + q4;   // (F16: pixel 32 hh + 8 q4 + ss — the K = 32 layout)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int c = 16 * nb + sl;
                    vb16[S16 ? 8 * hh + ss : 0][S16 ? nb : 0] = fac_w[pl * 32 + 4 * ((c >> 2) ^ (pl & 7)) + (c & 3)];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the slab is free again
    } else if (tile_lds) {
        float4 rowv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int pi = qy0 + r, pj = qx0 + (lane >> 3);
            rowv[r] = (pi < img_h && pj < img_w)
                          ? *reinterpret_cast<const float4 *>(v_out + ((size_t)pi * img_w + pj) * C + ch_off + 4 * (lane & 7))
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        hi = wave_hi();
        g_first = load_id(hi);
#pragma unroll
        for (int r = 0; r < 8; ++r)   // pixel 8 r + (lane >> 3), chunk (lane & 7) ^ (pixel & 7)
            *reinterpret_cast<float4 *>(fac_w + (8 * r + (lane >> 3)) * 32 + 4 * ((lane & 7) ^ (lane >> 3))) = rowv[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // lane = pixel, channels in order (the same fma chain as the direct form)
            const float4 v = *reinterpret_cast<const float4 *>(fac_w + lane * 32 + 4 * (k ^ (lane & 7)));
            Bsum = __builtin_fmaf(background[ch_off + 4 * k], v.x, Bsum);
            Bsum = __builtin_fmaf(background[ch_off + 4 * k + 1], v.y, Bsum);
            Bsum = __builtin_fmaf(background[ch_off + 4 * k + 2], v.z, Bsum);
            Bsum = __builtin_fmaf(background[ch_off + 4 * k + 3], v.w, Bsum);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {   // A operands: pixel (lane & 31) + 32 c, channels KS half .. + KS
            const int pm = (lane & 31) + 32 * c;
#pragma unroll
            for (int jq = 0; jq < KS / 4; ++jq) {
                const float4 v = *reinterpret_cast<const float4 *>(fac_w + pm * 32 + 4 * (((KS / 4) * half + jq) ^ (pm & 7)));
                voa_keep[c][S16 ? 0 : 4 * jq] = v.x;
                voa_keep[c][S16 ? 0 : 4 * jq + 1] = v.y;
                voa_keep[c][S16 ? 0 : 4 * jq + 2] = v.z;
                voa_keep[c][S16 ? 0 : 4 * jq + 3] = v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {  // B operands: pixel 2 s + half, channel wch
            const int pq = 2 * s + half;
            vob_keep[S16 ? 0 : s] = fac_w[pq * 32 + 4 * ((wch >> 2) ^ (pq & 7)) + (wch & 3)];
        }
        __builtin_amdgcn_wave_barrier();   // the slab is free again
    } else {
        hi = wave_hi();
        g_first = load_id(hi);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float v = (inside && (FULL || c < nch)) ? v_out[p * C + ch_off + c] : 0.0f;
            if (FULL || c < nch) Bsum = __builtin_fmaf(background[ch_off + c], v, Bsum);
        }
    }
    // the first chunk's records: requested here (the ids have arrived beside the tile), consumed after the rest of
    // the prologue
    ra_p = reinterpret_cast<const float4 *>(rec + g_first)[0];
    rb_p = reinterpret_cast<const float4 *>(rec + g_first)[1];
    // F16: s_w from the largest |cotangent| of the quadrant — every element of the first array's tile is in exactly one
    // lane's va16, every element of the second array's in one lane's t8
    float mloc = 0.0f;
    // feat_any (wave-uniform): does the quadrant's cotangent of THIS array have a non-zero (or NaN) element at all?  The
    // reference's training loss reaches the feature image at <= 2 600 sampled pixels of 1.9 M (gaussian_splatting.py:
    // 909-918): nine quadrants in ten have none, and for them a batch's flush of this array is skipped (its colour
    // gradients are exactly zero; GG_FEATANY_* above)
    bool feat_any = true;
    if (F16) {
        bool nz = false;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float v = va16[F16 ? blk : 0][F16 ? t : 0];
                mloc = fmaxf(mloc, fabsf(v));
                nz = nz || (v != 0.0f);
            }
        feat_any = __ballot(nz) != 0ull;
    }
    auto tile_scale = [&]() {
        for (int off = 32; off > 0; off >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
        sw = pow2_scale(mloc);
        inv_sw = pow2_inv(sw);
    };
    if (F16 && !EX) tile_scale();
    if (EX) {
        float t8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            // channel c lives in part a at column c - first channel of that part
            const int w0 = seg2.vo_w[0], w01 = w0 + seg2.vo_w[1];
            const int a = c < w0 ? 0 : (c < w01 ? 1 : 2);
            const int col = c - (a == 0 ? 0 : (a == 1 ? w0 : w01));
            // (loaded unconditionally — pixel 0 / column 0 stand in where there is nothing to read — so that the eight
            //  loads are in flight together; as a conditional load each one was waited for in its own branch)
            const bool have = c < seg2.nch2;
            t8[c] = seg2.v_out[have ? a : 0][p * seg2.vo_w[have ? a : 0] + (have ? col : 0)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            t8[c] = (inside && c < seg2.nch2) ? t8[c] : 0.0f;
            if (c < seg2.nch2) Bsum = __builtin_fmaf(seg2.background[c], t8[c], Bsum);
        }
        if (F16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) mloc = fmaxf(mloc, fabsf(t8[c]));
            tile_scale();
#pragma unroll
            for (int c = 0; c < 8; ++c) t8[c] *= sw;   // the LDS tile carries s_w: D's tail k-steps use it as it is, the
        }                                               // second flush takes it out of its results
        reinterpret_cast<float4 *>(vt + lane * 8)[0] = make_float4(t8[0], t8[1], t8[2], t8[3]);
        reinterpret_cast<float4 *>(vt + lane * 8)[1] = make_float4(t8[4], t8[5], t8[6], t8[7]);
        if (F2) {
            // the second flush's B operands are split into fp16 pieces per CHANNEL: column c of the tile gets its own
            // power of two on top of s_w (channels of one array can be decades apart — rgb | depth | normal are),
            // kept behind the tile; lane (c = lane & 7, part = lane >> 3) takes the maximum of 8 pixels, three
            // exchanges join the parts
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(vt[(8 * (lane >> 3) + i) * 8 + (lane & 7)]));
            for (int off = 8; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if (lane < 8) vt[64 * 8 + lane] = pow2_scale(m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (F16) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                split2h(va16[F16 ? blk : 0][F16 ? 2 * t : 0] * sw, va16[F16 ? blk : 0][F16 ? 2 * t + 1 : 0] * sw,
                        vah[F16 ? blk : 0][F16 ? t : 0], val[F16 ? blk : 0][F16 ? t : 0]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float m = 0.0f;
#pragma unroll
            for (int t = 0; t < 16; ++t) m = fmaxf(m, fabsf(vb16[F16 ? t : 0][F16 ? nb : 0]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float sc = pow2_scale(m);
            inv_scf[nb] = pow2_inv(sc) * GG_FAC_UNSCALE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    split2h(vb16[F16 ? 8 * ks + 2 * t : 0][F16 ? nb : 0] * sc, vb16[F16 ? 8 * ks + 2 * t + 1 : 0][F16 ? nb : 0] * sc,
                            vbh[F16 ? ks : 0][F16 ? nb : 0][F16 ? t : 0], vbl[F16 ? ks : 0][F16 ? nb : 0][F16 ? t : 0]);
        }
    }
    W = T_final * Bsum;
    // D product, A operands: V_OUT[pixel (lane & 31) + 32 c][channel KS half + s];
    // flush, B operands:     V_OUT[pixel 2s + half][channel wch]
    auto load_voa = [&](const float *vo, float (&dst)[2][KS]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int pm = (lane & 31) + 32 * c;
            const int pj = qx0 + (pm & 7), pi = qy0 + (pm >> 3);
            const bool pin = (pi < img_h) && (pj < img_w);
            const float *row = vo + ((size_t)(pin ? pi : 0) * img_w + (pin ? pj : 0)) * C + ch_off + KS * half;
#pragma unroll
            for (int s = 0; s < KS; ++s) dst[c][s] = (pin && (FULL || KS * half + s < nch)) ? row[s] : 0.0f;
        }
    };
    auto load_vob = [&](const float *vo, float (&dst)[32]) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int pq = 2 * s + half;
            const int pj = qx0 + (pq & 7), pi = qy0 + (pq >> 3);
            const bool ok = (pi < img_h) && (pj < img_w) && wch_ok;
            dst[s] = ok ? vo[((size_t)pi * img_w + pj) * C + ch_off + wch] : 0.0f;
        }
    };
    if (!S16 && !tile_lds) {
        float ta[2][KS], tb[32];
        load_voa(v_out, ta);
        load_vob(v_out, tb);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < KS; ++s) voa_keep[c][S16 ? 0 : s] = ta[c][s];
#pragma unroll
        for (int s = 0; s < 32; ++s) vob_keep[S16 ? 0 : s] = tb[s];
    }
    bool owner;
    const int myvar = R::var(lane, owner);
    const int my_q = myvar / KG, my_k = myvar - my_q * KG;
    float *my_base;
    int my_stride;
    if (my_k < 2) { my_base = v_xy + my_k; my_stride = gstride ? gstride : 2; }
    else if (my_k < 5) { my_base = v_conic + (my_k - 2); my_stride = gstride ? gstride : 3; }
    else { my_base = v_opacity; my_stride = gstride ? gstride : 1; }
    const int cs = cstride ? cstride : C;
    // vector loads of the colour half-rows need 16-byte aligned rows
    const bool vec = FULL && (C % 4 == 0) && (ch_off % 4 == 0) && ((reinterpret_cast<uintptr_t>(colors) & 15) == 0);

    int g_nxt = 0;   // ids of the next chunk (loaded one chunk ahead)
    // one batch: queue entries [base, base + n), n = 28 or 32 (fewer only for the last batch of the walk, which
    // is followed by null records up to a multiple of GRP)
    auto run_batch = [&](const int base, const int n, const bool fetch_next) {
        // (no old value of the prefetch registers to carry through the walk)
        ra_p.x = __builtin_nondeterministic_value(ra_p.x); ra_p.y = __builtin_nondeterministic_value(ra_p.y);
        ra_p.z = __builtin_nondeterministic_value(ra_p.z); ra_p.w = __builtin_nondeterministic_value(ra_p.w);
        rb_p.x = __builtin_nondeterministic_value(rb_p.x); rb_p.y = __builtin_nondeterministic_value(rb_p.y);
        rb_p.z = __builtin_nondeterministic_value(rb_p.z); rb_p.w = __builtin_nondeterministic_value(rb_p.w);
        if (ABL >= 6) { KEEP(n); return; }   // staging + queue only
        STAMP(1);
        STAMP_BATCH();
        const int jl = S16 ? (lane & 15) : (lane & 31);
        const int cgid = (jl < n) ? __builtin_bit_cast(int, Q.b[base + jl].w) : -1;
        if (S16) {   // D[64 pixels x 16 slots] as 4 x 10 v_mfma_f32_16x16x4_f32
            const int q4 = lane >> 4;
            const float *row = colors + (size_t)(cgid < 0 ? 0 : cgid) * C + ch_off + 8 * q4;
            float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
            c0 = *reinterpret_cast<const float4 *>(row);
            c1 = *reinterpret_cast<const float4 *>(row + 4);
            const float colb[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float colb2[2] = {0.0f, 0.0f};
            float va2[4][2];
            if (EX) {
                const float *row2 = seg2.colors + (size_t)(cgid < 0 ? 0 : cgid) * seg2.C2 + 2 * q4;
#pragma unroll
                for (int t = 0; t < 2; ++t) colb2[t] = (2 * q4 + t < seg2.nch2) ? row2[t] : 0.0f;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const float2 t2 = *reinterpret_cast<const float2 *>(vt + (16 * blk + jl) * 8 + 2 * q4);
                    va2[blk][0] = t2.x;
                    va2[blk][1] = t2.y;
                }
            }
            f32x4 d[4];
            if (F16) {
                // s_g: the power of two for this slot's colour row (both arrays; the four lanes of a slot hold it all)
                float m = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; ++t) m = fmaxf(m, fabsf(colb[t]));
                m = fmaxf(m, fmaxf(fabsf(colb2[0]), fabsf(colb2[1])));
                m = cgid < 0 ? 0.0f : m;
#if GG_F16_SWAPMAX
                {
                    auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
                    m = fmaxf(__builtin_bit_cast(float, (unsigned)r16[0]), __builtin_bit_cast(float, (unsigned)r16[1]));
                    auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
                    m = fmaxf(__builtin_bit_cast(float, (unsigned)r32[0]), __builtin_bit_cast(float, (unsigned)r32[1]));
                }
#else
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
#endif
                const float sg = cgid < 0 ? 0.0f : pow2_scale(m);   // (null slot: every operand 0)
                const float unscale = pow2_inv(cgid < 0 ? 1.0f : sg) * inv_sw;
                unsigned ch_[4] = {0u, 0u, 0u, 0u}, cl_[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int t = 0; t < 4; ++t) split2h(colb[2 * t] * sg, colb[2 * t + 1] * sg, ch_[t], cl_[t]);
                const h16x8 Bh = H8(ch_[0], ch_[1], ch_[2], ch_[3]), Bl = H8(cl_[0], cl_[1], cl_[2], cl_[3]);
                const float b2[2] = {colb2[0] * sg, colb2[1] * sg};   // (zeros without a second array)
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const int bb = F16 ? blk : 0;
                    const h16x8 Ah = H8(vah[bb][0], vah[bb][F16 ? 1 : 0], vah[bb][F16 ? 2 : 0], vah[bb][F16 ? 3 : 0]);
                    const h16x8 Al = H8(val[bb][0], val[bb][F16 ? 1 : 0], val[bb][F16 ? 2 : 0], val[bb][F16 ? 3 : 0]);
                    d[blk] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#if !GG_F16_NLL
                    d[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bl, d[blk], 0, 0, 0);
#endif
                    d[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, d[blk], 0, 0, 0);
                    d[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, d[blk], 0, 0, 0);
                    d[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, d[blk], 0, 0, 0);
                    if (EX) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)   // the second array's channels: fp32 k-steps on operands carrying s_w, s_g
                            d[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(va2[blk][t], b2[t], d[blk], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fac_w[FIDX(jl, 16 * blk + 4 * q4 + r)] = d[blk][r] * unscale;
            } else {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                d[blk] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    d[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(va16[S16 ? blk : 0][S16 ? t : 0], cgid < 0 ? 0.0f : colb[t],
                                                                  d[blk], 0, 0, 0);
                if (EX) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        d[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(va2[blk][t], cgid < 0 ? 0.0f : colb2[t], d[blk], 0, 0, 0);
                }
            }
            // D[pixel 16 blk + 4 q4 + r][slot jl]
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) fac_w[FIDX(jl, 16 * blk + 4 * q4 + r)] = d[blk][r];
            }
        }
        f32x16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d0[r] = d1[r] = 0.0f;
        if (ABL < 4 && !S16) {
            float colb[KS];
            const float *row = colors + (size_t)(cgid < 0 ? 0 : cgid) * C + ch_off + KS * half;
            if (vec) {
#pragma unroll
                for (int s = 0; s < KS; s += 4) {
                    const float4 v4 = *reinterpret_cast<const float4 *>(row + s);
                    colb[s] = v4.x; colb[s + 1] = v4.y; colb[s + 2] = v4.z; colb[s + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) colb[s] = (FULL || KS * half + s < nch) ? row[s] : 0.0f;
            }
            float colb2[4];
            float4 va0, va1;
            if (EX) {   // second array: channels 4 half + s of ITS colour row, and of the v_out tile in LDS
                const float *row2 = seg2.colors + (size_t)(cgid < 0 ? 0 : cgid) * seg2.C2 + 4 * half;
#pragma unroll
                for (int s = 0; s < 4; ++s) colb2[s] = (4 * half + s < seg2.nch2) ? row2[s] : 0.0f;
                va0 = *reinterpret_cast<const float4 *>(vt + (lane & 31) * 8 + 4 * half);
                va1 = *reinterpret_cast<const float4 *>(vt + ((lane & 31) + 32) * 8 + 4 * half);
            }
            float voa[2][KS];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s = 0; s < KS; ++s) voa[c][s] = voa_keep[c][S16 ? 0 : s];
#ifdef GG_STAMPS
            {   // make the colour rows arrive inside phase 2
                float sink_ = 0.0f;
#pragma unroll
                for (int s = 0; s < KS; ++s) sink_ += colb[s];
                if (EX) sink_ += colb2[0] + colb2[1] + colb2[2] + colb2[3];
                KEEP(sink_);
            }
            STAMP(2);
#endif
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float bv = cgid < 0 ? 0.0f : colb[s];
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(voa[0][s], bv, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(voa[1][s], bv, d1, 0, 0, 0);
            }
            if (EX) {
                const float a0[4] = {va0.x, va0.y, va0.z, va0.w}, a1[4] = {va1.x, va1.y, va1.z, va1.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bv = cgid < 0 ? 0.0f : colb2[s];
                    d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bv, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bv, d1, 0, 0, 0);
                }
            }
        }
        if (DET && lane < 32) slote[jl] = __builtin_bit_cast(int, Q.a[base + min(jl, n - 1)].w);
        // D[pixel m][slot n]: lane holds n = lane & 31, m = (r & 3) + 8 (r >> 2) + 4 half (+ 32 for d1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (S16) continue;
            const int pix = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (NSLOT == 32 || jl < NSLOT) {
                fac_w[FIDX(jl, pix)] = d0[r];
                fac_w[FIDX(jl, 32 + pix)] = d1[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(3);
        unsigned slotmask = 0u;   // wave-uniform: slots with at least one blending pixel
        for (int g = 0; g < n; g += GRP) {
            // Two passes over the group.  The second one READS THE RECORDS AGAIN (wave-uniform LDS broadcasts) and
            // takes D from the slab only then: holding dx, dy, opacity, the conic and D of four Gaussians across
            // the passes is 28 registers at the kernel's register peak (the pair build spilled 16-26 of them, and a
            // scratch reload sits behind every float atomic in flight: the vector memory counter is in order).
            float vis[GRP], alpha[GRP];
            bool pass[GRP];
            float4 Ah[REREAD ? 1 : GRP], Bh[REREAD ? 1 : GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const float4 A = Q.a[base + g + q], B = Q.b[base + g + q];
                if (!REREAD) { Ah[REREAD ? 0 : q] = A; Bh[REREAD ? 0 : q] = B; }
                const int pos = __builtin_bit_cast(int, A.w);
                const float dx = A.x - px, dy = A.y - py;
                const float sigma = __builtin_fmaf(
                    0.5f, __builtin_fmaf(B.x * dx, dx, (B.z * dy) * dy), (B.y * dx) * dy);
                vis[q] = gg_expf_walk(-sigma);
                alpha[q] = fminf(GG_ALPHA_MAX_BWD, A.z * vis[q]);
                pass[q] = (pos < fin) && sigma >= 0.0f && !(alpha[q] < GG_ALPHA_MIN);
            }
            if (__ballot(pass[0] || pass[1] || pass[2] || pass[3]) == 0ull) continue;
            if (ABL >= 5) {
#pragma unroll
                for (int q = 0; q < GRP; ++q) { KEEP(vis[q]); KEEP(alpha[q]); KEEP((int)pass[q]); }
                continue;
            }
            int gi = base + g;
            if (REREAD) asm volatile("" : "+s"(gi));   // a second read of the records, not values carried over
            float part[KB];
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                float *pg = part + q * KG;
                if (__ballot(pass[q]) == 0ull) {  // wave-uniform: nobody blends this Gaussian
#pragma unroll
                    for (int v = 0; v < KG; ++v) pg[v] = 0.0f;
                    continue;
                }
                const float4 A = REREAD ? Q.a[gi + q] : Ah[REREAD ? 0 : q], B = REREAD ? Q.b[gi + q] : Bh[REREAD ? 0 : q];
                const float D = fac_w[FIDX(g + q, lane)];
                // MG: a pair that does not pass takes alpha = 0 and exp = 0 through the SAME arithmetic instead of five
                // selects: 1 / (1 - 0) = 1, so T and (D finite) W stay as they are and fac, v_sigma and the moments are zeros
                constexpr bool SEL2 = MG && GG_MG_MOMENTS;
                const float al = SEL2 ? (pass[q] ? alpha[q] : 0.0f) : alpha[q];
                const float ra_ = __builtin_amdgcn_rcpf(1.0f - al);
                const float Tn = T * ra_;
                const float fac = SEL2 ? al * Tn : (pass[q] ? alpha[q] * Tn : 0.0f);
                const float v_alpha = SEL2 ? __builtin_fmaf(Tn, D, -(ra_ * W)) : (pass[q] ? (Tn * D - ra_ * W) : 0.0f);
                W = SEL2 ? __builtin_fmaf(D, fac, W) : (pass[q] ? __builtin_fmaf(D, fac, W) : W);
                T = SEL2 ? Tn : (pass[q] ? Tn : T);
                const float dx = A.x - px, dy = A.y - py;
                if (MG && GG_MG_MOMENTS) {
                    // moments about the Gaussian's own centre: sum v_sigma {dx, dy, dx^2, dx dy, dy^2}; the conic enters
                    // AFTER the reduction, in the flush (v_x = a M0 + b M1, v_y = b M0 + c M1, conic gradients M2..4 / 2):
                    // 5 multiplies per pair instead of 14 operations.  (Not the experiment of 3.5c: no expansion about a
                    // distant centre, nothing cancels that did not cancel before.)
                    const float vis_s = pass[q] ? vis[q] : 0.0f;     // (vis may be inf where the pair does not pass)
                    const float v_sigma = (-A.z * vis_s) * v_alpha;
                    pg[0] = v_sigma * dx;
                    pg[1] = v_sigma * dy;
                    pg[2] = pg[0] * dx;
                    pg[3] = pg[0] * dy;
                    pg[4] = pg[1] * dy;
                    pg[5] = vis_s * v_alpha;
                } else {
                const float v_sigma = pass[q] ? (-A.z * vis[q]) * v_alpha : 0.0f;
                pg[0] = v_sigma * (B.x * dx + B.y * dy);
                pg[1] = v_sigma * (B.y * dx + B.z * dy);
                const float hs = 0.5f * v_sigma;
                pg[2] = (hs * dx) * dx;
                pg[3] = (hs * dx) * dy;
                pg[4] = (hs * dy) * dy;
                pg[5] = pass[q] ? vis[q] * v_alpha : 0.0f;
                }
                fac_w[FIDX(g + q, lane)] = fac;   // over D: the flush reads fac from here
                slotmask |= 1u << (g + q);
            }
            if (ABL >= 3) {
#pragma unroll
                for (int v = 0; v < KB; ++v) KEEP(part[v]);
                continue;
            }
            const float mine = R::run(part, lane);
            const int my_gid = __builtin_bit_cast(int, Q.b[base + g + my_q].w);
            if (DET) {
                const size_t e = (size_t)__builtin_bit_cast(int, Q.a[base + g + my_q].w);
                if (owner && my_gid >= 0) det.p[(e * 4 + wave) * det.ks + det.goff + my_k] = mine;
                continue;
            }
            if (MG) {   // parked: leaves with the second array's colour gradients in the flush
                if (owner) geo_w[(g + my_q) * 8 + my_k] = mine;
                continue;
            }
            if (owner && my_gid >= 0 && mine != 0.0f) atomicAdd(my_base + (size_t)my_gid * my_stride, mine);
        }
        STAMP(4);
        if (fetch_next) {
            ra_p = reinterpret_cast<const float4 *>(rec + g_nxt)[0];
            rb_p = reinterpret_cast<const float4 *>(rec + g_nxt)[1];
        }
        // flush: FAC[32 slots x 64 pixels] * V_OUT[64 x 32 channels]; rows outside slotmask still hold D and
        // are not written
        slotmask = __builtin_amdgcn_readfirstlane(slotmask);
        if (ABL >= 2 || slotmask == 0u) return;
        f32x4 a4f_keep = {0.0f, 0.0f, 0.0f, 0.0f};   // F2: the second array's flush, formed beside the first one's
        if (S16) {   // FAC[16 slots x 64 pixels] * V_OUT[64 x 32 channels] as 2 x 16 v_mfma_f32_16x16x4_f32
            int sl = lane & 15, q4 = lane >> 4;
            asm volatile("" : "+v"(sl), "+v"(q4));
            f32x4 acc2[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
            const bool first_on = !(F16 && !feat_any && GG_FEATANY_FLUSH);   // (no cotangent of the first array in the
            f32x4 &a4f = a4f_keep;                                            //  quadrant: its colour gradients are zeros)
            if (F16 && (first_on || F2)) {   // A: FAC[slot sl][pixel 32 ks + 8 q4 + 0..7] x 2^15 in two pieces
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    unsigned fh[4], fl[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        split2h(fac_w[FIDX(sl, 32 * ks + 8 * q4 + 2 * t)] * GG_FAC_SCALE,
                                fac_w[FIDX(sl, 32 * ks + 8 * q4 + 2 * t + 1)] * GG_FAC_SCALE, fh[t], fl[t]);
                    const h16x8 Fh = H8(fh[0], fh[1], fh[2], fh[3]), Fl = H8(fl[0], fl[1], fl[2], fl[3]);
                    if (first_on) {   // 2 x 2 x 4 v_mfma_f32_16x16x32_f16
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int kk = F16 ? ks : 0, nn = F16 ? nb : 0;
                        const h16x8 Vh = H8(vbh[kk][nn][0], vbh[kk][nn][F16 ? 1 : 0], vbh[kk][nn][F16 ? 2 : 0], vbh[kk][nn][F16 ? 3 : 0]);
                        const h16x8 Vl = H8(vbl[kk][nn][0], vbl[kk][nn][F16 ? 1 : 0], vbl[kk][nn][F16 ? 2 : 0], vbl[kk][nn][F16 ? 3 : 0]);
#if !GG_F16_NLL
                        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fl, Vl, acc2[nb], 0, 0, 0);
#endif
                        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fl, Vh, acc2[nb], 0, 0, 0);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fh, Vl, acc2[nb], 0, 0, 0);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fh, Vh, acc2[nb], 0, 0, 0);
                    }
                    }
                    if (F2) {         // the second array's flush rides on the same A pieces
                        // B: V_OUT2[pixel 32 ks + 8 q4 + 0..7][channel sl] (x s_w x the channel's own scale; lanes of
                        // channels 8..15 supply zeros)
                        unsigned wh[4], wl[4];
                        const float wm = sl < 8 ? vt[64 * 8 + (sl & 7)] : 0.0f;   // the channel's power of two
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            split2h(vt[(32 * ks + 8 * q4 + 2 * t) * 8 + (sl & 7)] * wm,
                                    vt[(32 * ks + 8 * q4 + 2 * t + 1) * 8 + (sl & 7)] * wm, wh[t], wl[t]);
                        const h16x8 Wh = H8(wh[0], wh[1], wh[2], wh[3]);
                        const h16x8 Wl = H8(wl[0], wl[1], wl[2], wl[3]);
#if !GG_F16_NLL
                        a4f = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fl, Wl, a4f, 0, 0, 0);
#endif
                        a4f = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fl, Wh, a4f, 0, 0, 0);
                        a4f = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fh, Wl, a4f, 0, 0, 0);
                        a4f = __builtin_amdgcn_mfma_f32_16x16x32_f16(Fh, Wh, a4f, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc2[nb][r] *= inv_scf[nb];
            } else if (F16) {
                // (nothing to add for the first array, and the second flush takes its operands itself)
            } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float a = fac_w[FIDX(sl, 4 * t + q4)];
                acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vb16[S16 ? t : 0][0], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vb16[S16 ? t : 0][S16 ? 1 : 0], acc2[1], 0, 0, 0);
            }
            }
            // lane holds channel 16 nb + sl of slots 4 q4 + r
            if (!F16 || feat_any || !GG_FEATANY_FLUSH)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = 4 * q4 + r;
                    const bool on = ((slotmask >> slot) & 1u) != 0u;
                    if (on && acc2[nb][r] != 0.0f) {
                        const int sg = __builtin_bit_cast(int, Q.b[base + slot].w);
                        atomicAdd(v_colors + (size_t)sg * cs + ch_off + 16 * nb + sl, acc2[nb][r]);
                    }
                }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        float vob[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) vob[s] = vob_keep[S16 ? 0 : s];
        const int arow = NSLOT == 32 ? (lane & 31) : min(lane & 31, NSLOT - 1);   // rows >= NSLOT: never written out
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (S16) continue;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fac_w[FIDX(arow, 2 * s + half)], vob[s], acc, 0, 0, 0);
        }
#ifdef GG_STAMPS
#pragma unroll
        for (int r = 0; r < 16; ++r) KEEP(acc[r]);
        STAMP(5);
#endif
        // (lane-derived slot numbers and mask bits are computed HERE, from laundered values: as loop invariants they
        //  took 6-16 registers for the whole kernel)
        int half_f = half, k4_f = lane >> 4;
        asm volatile("" : "+v"(half_f), "+v"(k4_f));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (S16) continue;
            const int slot = (r & 3) + 8 * (r >> 2) + 4 * half_f;
            if (ABL >= 1) { KEEP(acc[r]); continue; }
            const bool on = ((slotmask >> slot) & 1u) != 0u && wch_ok;
            if (DET) {
                if (on) det.p[((size_t)slote[slot] * 4 + wave) * det.ks + det.coff + wch] = acc[r];
                continue;
            }
            if (on && acc[r] != 0.0f) {
                const int sg = __builtin_bit_cast(int, Q.b[base + slot].w);   // the slot's Gaussian
                atomicAdd(v_colors + (size_t)sg * cs + ch_off + wch, acc[r]);
            }
        }
        STAMP(6);
        if (EX) {
            // second array: FAC[32 slots x 64 pixels] * V_OUT2[64 x 8] as 2 x 16 v_mfma_f32_16x16x4_f32
            // (A: slot 16 mb + (lane & 15), pixel 4 s + (lane >> 4); B: the same pixel, channel lane & 15;
            //  D: lane holds channel lane & 15 of slots 16 mb + 4 (lane >> 4) + r)
            const int n16 = lane & 15, k4 = k4_f;
            const float *vbp = vt + k4 * 8 + (n16 & 7);
            const float vmask = n16 < 8 ? 1.0f : 0.0f;   // lanes of channels 8..15 supply zeros
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (S16 && mb > 0) continue;
                if (((slotmask >> (16 * mb)) & 0xffffu) == 0u) continue;
                f32x4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
                const int m = 16 * mb + n16;
                if (F2) {   // formed beside the first flush, on fp16 pieces: the 2^15 of FAC comes out here
#pragma unroll
                    for (int r = 0; r < 4; ++r) a4[r] = a4f_keep[r] * GG_FAC_UNSCALE * pow2_inv(vt[64 * 8 + (n16 & 7)]);
                } else {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fac_w[FIDX(m, 4 * s + k4)], vbp[32 * s] * vmask, a4,
                                                              0, 0, 0);
                }
                if (MG) {
                    // record row of slot 4 k4 + r: lanes n16 < 8 carry channel n16 of the second array (column 6 + n16),
                    // lanes 8..13 the geometry sum n16 - 8 (column n16 - 8): 14 floats of one 64-byte row per request
                    const int col = n16 < 8 ? 6 + n16 : n16 - 8;
                    const bool lane_on = n16 < 8 ? n16 < seg2.nch2 : n16 < 14;
                    const int gcol = n16 & 7;   // geometry column of lanes 8..13: 0, 1 mix moments 0 and 1 with the conic
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int slot = 4 * k4 + r;
                        const float4 cq = Q.b[base + slot];     // conic a, b, c
                        const float ga = geo_w[slot * 8 + (gcol < 2 ? 0 : gcol)], gb = geo_w[slot * 8 + 1];
                        const float ca = gcol == 0 ? cq.x : (gcol == 1 ? cq.y : (gcol == 5 ? 1.0f : 0.5f));
                        const float cb = gcol == 0 ? cq.y : cq.z;
                        const float gv = !GG_MG_MOMENTS ? geo_w[slot * 8 + gcol] : (gcol < 2 ? ca * ga + cb * gb : ca * ga);
                        const float val = n16 < 8 ? (F16 ? a4[r] * inv_sw : a4[r]) : gv;
                        const bool on = ((slotmask >> slot) & 1u) != 0u && lane_on;
                        if (on && val != 0.0f) {
                            const int sg = __builtin_bit_cast(int, Q.b[base + slot].w);
                            atomicAdd(v_xy + (size_t)sg * 16 + col, val);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = 16 * mb + 4 * k4 + r;
                    if (ABL >= 1) { KEEP(a4[r]); continue; }
                    const bool on = ((slotmask >> slot) & 1u) != 0u && n16 < seg2.nch2;
                    if (on && a4[r] != 0.0f) {
                        const int sg = __builtin_bit_cast(int, Q.b[base + slot].w);
                        atomicAdd(seg2.v_colors + (size_t)sg * seg2.cs2 + n16, F16 ? a4[r] * inv_sw : a4[r]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(7);
    };

    int qn = 0;   // queued survivors (wave-uniform)
    // the first of the two dependent loads of the staging (id, then record) runs one chunk ahead of the walk
    g_nxt = g_first;
    STAMP(0);
    bool have_p = true;   // wave-uniform: ra_p / rb_p hold this chunk's records
    // ONE call site of run_batch (the tail batch runs through the same loop): inlined twice, the allocator kept two
    // sets of loop invariants and spilled one
    for (int top = hi;; top -= 64) {
      const bool more = top > range.x;   // wave-uniform: another chunk to stage
      if (more) {
        const int e = top - 64 + lane;
        const bool valid = e >= range.x;
        const int g = g_nxt;
        float4 ra, rb;
        if (have_p) {
            ra = ra_p;
            rb = rb_p;
        } else {
            ra = reinterpret_cast<const float4 *>(rec + g)[0];
            rb = reinterpret_cast<const float4 *>(rec + g)[1];
        }
        have_p = false;
        // the prefetch registers are dead until the next request (no old value to carry through the walk)
        ra_p.x = __builtin_nondeterministic_value(ra_p.x); ra_p.y = __builtin_nondeterministic_value(ra_p.y);
        ra_p.z = __builtin_nondeterministic_value(ra_p.z); ra_p.w = __builtin_nondeterministic_value(ra_p.w);
        rb_p.x = __builtin_nondeterministic_value(rb_p.x); rb_p.y = __builtin_nondeterministic_value(rb_p.y);
        rb_p.z = __builtin_nondeterministic_value(rb_p.z); rb_p.w = __builtin_nondeterministic_value(rb_p.w);
        g_nxt = load_id(top - 64);
        // the quadrant's rectangle is recomputed here (four conversions of wave-uniform integers): hoisted out of
        // the loop, the allocator spilled the four floats and reloaded them from scratch in every chunk — behind
        // every float atomic still in flight
        int qxl = qx0, qyl = qy0;
        asm volatile("" : "+s"(qxl), "+s"(qyl));
        const bool hit = valid && rec_hits_rect(ra, rb, (float)qxl, (float)(qxl + 7), (float)qyl, (float)(qyl + 7));
        const uint64_t m = __ballot(hit);
        const int cnt = __builtin_popcountll(m);
        if (hit) {   // processing order: descending list position
            const int pos = qn + __builtin_popcountll((m >> lane) >> 1);
            Q.a[pos] = make_float4(ra.x, ra.y, ra.z, __builtin_bit_cast(float, e));
            Q.b[pos] = make_float4(rb.x, rb.y, rb.z, __builtin_bit_cast(float, g));
        }
        qn += cnt;
        __builtin_amdgcn_wave_barrier();
      } else {
        if (qn == 0) break;
        if (lane < GRP) {   // null records behind the last survivor: opacity 0 -> alpha 0 -> never pass
            Q.a[qn + lane] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, 0x7fffffff));
            Q.b[qn + lane] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));
        }
        __builtin_amdgcn_wave_barrier();
      }
        int done = 0;
        // 28 or 32 (a multiple of GRP): at most 27 stay behind, BQ_CAP; after the last chunk: what is left (<= 27)
        while (more ? (qn - done >= (S16 ? 16 : 28)) : (done == 0)) {
            const int nb = !more ? qn : (S16 ? NSLOT : min(32, (qn - done) & ~3));
            // (not in the pair build: 8 more registers across its two flushes spill 30 more, 1.30 -> 1.33 ms)
            const bool last = more && !EX && (qn - done - nb < (S16 ? 16 : 28)) && (top - 64 > range.x);   // staging comes next
            run_batch(done, nb, last);
            have_p = last;
            done += nb;
        }
        if (!more) break;
        STAMP(1);
        if (done > 0) {   // bring the left-over (< 28) to the front; source and destination do not overlap
            const int left = qn - done;
            // (one array at a time: with both records in flight the allocator parked them in scratch)
            if (lane < left) { const float4 ta = Q.a[done + lane]; __builtin_amdgcn_wave_barrier(); Q.a[lane] = ta; }
            __builtin_amdgcn_wave_barrier();
            if (lane < left) { const float4 tb = Q.b[done + lane]; __builtin_amdgcn_wave_barrier(); Q.b[lane] = tb; }
            qn = left;
            __builtin_amdgcn_wave_barrier();
        }
        STAMP(8);
    }
    STAMP_END();
}

// =============================================================================================
// launchers used by the C ABI in blend.hip
// =============================================================================================
#ifndef GG_BWD_S16
#define GG_BWD_S16 1
#endif
#ifndef GG_BWD_MERGE
#define GG_BWD_MERGE 1   // 16-float records: geometry and second-array gradients in one atomic request per Gaussian (MG)
#endif
#define B2_FWD_ARGS C, off, n, img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, \
                    out_img, final_Ts, final_idx, write_final
void gg_launch_blend2_fwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, float *out_img,
                          float *final_Ts, int32_t *final_idx, int write_final, hipStream_t s) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_OTHER)), block(64 * GG_WPB_OTHER);
    if (width == 1)
        hipLaunchKernelGGL((blend2_fwd_kernel<1, false, true>), grid, block, 0, s, B2_FWD_ARGS);
    else if (width == 2)
        hipLaunchKernelGGL((blend2_fwd_kernel<2, false, true>), grid, block, 0, s, B2_FWD_ARGS);
    else if (width == 3)
        hipLaunchKernelGGL((blend2_fwd_kernel<3, false, true>), grid, block, 0, s, B2_FWD_ARGS);
    else if (width == 8)
        hipLaunchKernelGGL((blend2_fwd_kernel<8, false, true>), grid, block, 0, s, B2_FWD_ARGS);
    else if (n == 32)
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true>), grid, block, 0, s, B2_FWD_ARGS);
    else
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, false>), grid, block, 0, s, B2_FWD_ARGS);
}

// several 32-channel blocks in one walk (channels [off, off + 32 ncb)); 16-byte aligned rows (the launcher's caller
// checks), ncb = 2, 3 or 4
void gg_launch_blend2_fwd_blocks(int ncb, int C, int off, int img_h, int img_w, int tiles_x, int ntiles,
                                 const int32_t *ids, const int2 *bins, const GRec *rec, const float *colors,
                                 const float *background, float *out_img, float *final_Ts, int32_t *final_idx,
                                 int write_final, hipStream_t s) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_OTHER)), block(64 * GG_WPB_OTHER);
    const int n = 32;
    if (ncb == 2)
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, false, 0, 2>), grid, block, 0, s, B2_FWD_ARGS);
    else if (ncb == 3)
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, false, 0, 3>), grid, block, 0, s, B2_FWD_ARGS);
    else
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, false, 0, 4>), grid, block, 0, s, B2_FWD_ARGS);
}

// ncb: 32-channel blocks of the first array in this walk (1, 2 or 4; channels [0, 32 ncb)); fast: the batched fp16
// two-piece kernel for the first block + second array (ncb is then 1)
void gg_launch_blend2_fwd_pair(int C, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *ids,
                               const int2 *bins, const GRec *rec, const float *colors, const float *background,
                               float *out_img, float *final_Ts, int32_t *final_idx, const float *colors2, int C2,
                               const float *background2, float *out_img2, hipStream_t s, int ncb, bool fast,
                               unsigned bytes1, unsigned bytes2) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_OTHER)), block(64 * GG_WPB_OTHER);
    Seg2 seg2;
    seg2.colors = colors2;
    seg2.background = background2;
    seg2.out_img = out_img2;
    seg2.C2 = C2;
    seg2.nch2 = C2;
    if (fast) {
        hipLaunchKernelGGL(blend2_fwd_batch_kernel, dim3(gg_blend_grid(ntiles, 1)), dim3(64), 0, s, C, img_h, img_w, tiles_x,
                           ntiles, ids, bins, rec, colors, background, out_img, final_Ts, final_idx, seg2, bytes1, bytes2);
        return;
    }
#define B2_FPAIR(L) hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, true, L>), grid, block, 0, s, C, 0, 32, \
        img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, out_img, final_Ts, final_idx, 1, seg2)
#ifdef GG_ABLATION
    switch (g_fwd_abl) {   // measurement twin: wrong images on purpose
        case 1: B2_FPAIR(1); return;
        case 2: B2_FPAIR(2); return;
        case 3: B2_FPAIR(3); return;
        case 4: B2_FPAIR(4); return;
        default: break;
    }
#endif
    if (ncb == 2) {
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, true, 0, 2>), grid, block, 0, s, C, 0, 32, img_h,
                           img_w, tiles_x, ntiles, ids, bins, rec, colors, background, out_img, final_Ts, final_idx, 1, seg2);
        return;
    }
    if (ncb == 4) {
        hipLaunchKernelGGL((blend2_fwd_kernel<32, true, true, true, 0, 4>), grid, block, 0, s, C, 0, 32, img_h,
                           img_w, tiles_x, ntiles, ids, bins, rec, colors, background, out_img, final_Ts, final_idx, 1, seg2);
        return;
    }
    B2_FPAIR(0);
}

#define B2_BWDW_ARGS C, off, n, img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, \
                     final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride, det
#define B2_BWDN_ARGS C, off, n, img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, \
                     final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride, det
void gg_launch_blend2_bwd(int width, int C, int off, int n, int img_h, int img_w, int tiles_x,
                          int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                          const float *colors, const float *background, const float *final_Ts,
                          const int32_t *final_idx, const float *v_out, float *v_xy, float *v_conic,
                          float *v_colors, float *v_opacity, int gstride, int cstride, hipStream_t s,
                          DetSlab det) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_OTHER)), block(64 * GG_WPB_OTHER);
    dim3 gridw(gg_blend_grid(ntiles, GG_WPB_WIDE_BWD)), blockw(64 * GG_WPB_WIDE_BWD);   // the wide kernels
    if (det.p) {   // deterministic mode: same kernels with the atomics replaced by slab stores
        if (width == 1)
            hipLaunchKernelGGL((blend2_bwd_narrow_kernel<1, 0, true>), grid, block, 0, s, B2_BWDN_ARGS);
        else if (width == 2)
            hipLaunchKernelGGL((blend2_bwd_narrow_kernel<2, 0, true>), grid, block, 0, s, B2_BWDN_ARGS);
        else if (width == 3)
            hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 0, true>), grid, block, 0, s, B2_BWDN_ARGS);
        else if (width == 8)
            hipLaunchKernelGGL((blend2_bwd_narrow_kernel<8, 0, true>), grid, block, 0, s, B2_BWDN_ARGS);
        else if (n == 32)
            hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, 0, 32, true>), gridw, blockw, 0, s, B2_BWDW_ARGS);
        else   // partial 32-channel chunk: one masked variant is enough for this mode
            hipLaunchKernelGGL((blend2_bwd_wide_kernel<false, 0, 32, true>), gridw, blockw, 0, s, B2_BWDW_ARGS);
        return;
    }
    if (width == 1)
        hipLaunchKernelGGL((blend2_bwd_narrow_kernel<1>), grid, block, 0, s, B2_BWDN_ARGS);
    else if (width == 2)
        hipLaunchKernelGGL((blend2_bwd_narrow_kernel<2>), grid, block, 0, s, B2_BWDN_ARGS);
    else if (width == 3)
        hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3>), grid, block, 0, s, B2_BWDN_ARGS);
    else if (width == 8)
        hipLaunchKernelGGL((blend2_bwd_narrow_kernel<8>), grid, block, 0, s, B2_BWDN_ARGS);
    else if (n == 32 && GG_BWD_S16 && C % 4 == 0 && off % 4 == 0 && (reinterpret_cast<uintptr_t>(colors) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(v_out) & 15) == 0)   // the 16-slot build (four waves per SIMD)
        hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, 0, 32, false, false, GG_BWD_S16 != 0>), gridw, blockw, 0, s,
                           B2_BWDW_ARGS);
    else if (n == 32)
        hipLaunchKernelGGL((blend2_bwd_wide_kernel<true>), gridw, blockw, 0, s, B2_BWDW_ARGS);
    else if (n <= 8)
        hipLaunchKernelGGL((blend2_bwd_wide_kernel<false, 0, 8>), gridw, blockw, 0, s, B2_BWDW_ARGS);
    else if (n <= 16)
        hipLaunchKernelGGL((blend2_bwd_wide_kernel<false, 0, 16>), gridw, blockw, 0, s, B2_BWDW_ARGS);
    else
        hipLaunchKernelGGL((blend2_bwd_wide_kernel<false, 0, 32>), gridw, blockw, 0, s, B2_BWDW_ARGS);
}

#ifdef GG_ABLATION
static int g_pair_ablate = 0;
// explicit workgroup -> tile order of the blend2 kernels (device array of ntiles ints, or null: default mapping)
extern "C" int gg_debug_set_tile_order(const int *order) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_order), &order, sizeof(order)) == hipSuccess ? 0 : -1;
}
extern "C" int gg_debug_set_pair_ablation(int level) {
    const int prev = g_pair_ablate;
    g_pair_ablate = level;
    return prev;
}
#endif
// first 32 channels of `colors` + a second array of <= 8 channels in one walk (gg_blend_bwd_pair)
void gg_launch_blend2_bwd_pair(int C, int img_h, int img_w, int tiles_x, int ntiles, const int32_t *ids,
                               const int2 *bins, const GRec *rec, const float *colors, const float *background,
                               const float *final_Ts, const int32_t *final_idx, const float *v_out, float *v_xy,
                               float *v_conic, float *v_colors, float *v_opacity, int gstride, int cstride,
                               const float *colors2, int C2, const float *background2,
                               const float *const *v_out2_parts, const int *v_out2_channels, int num_parts,
                               float *v_colors2, int cstride2, hipStream_t s) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_WIDE_BWD)), block(64 * GG_WPB_WIDE_BWD);
    Seg2B seg2;
    seg2.colors = colors2;
    seg2.background = background2;
    for (int k = 0; k < 3; ++k) {
        seg2.v_out[k] = k < num_parts ? v_out2_parts[k] : v_out2_parts[0];
        seg2.vo_w[k] = k < num_parts ? v_out2_channels[k] : 0;   // (channels >= channels2 are never read)
    }
    seg2.v_colors = v_colors2;
    seg2.C2 = C2;
    seg2.nch2 = C2;
    seg2.cs2 = cstride2 ? cstride2 : C2;
#define B2_PAIR(L) hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, L, 32, false, true>), grid, block, 0, s, C, 0, 32, \
        img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, final_Ts, final_idx, v_out, v_xy, \
        v_conic, v_colors, v_opacity, gstride, cstride, DetSlab(), seg2)
#ifdef GG_ABLATION
    const bool ablated = g_pair_ablate != 0;   // the ablated builds are the 32-slot ones
#else
    const bool ablated = false;
#endif
    if (GG_BWD_S16 && !ablated && C % 4 == 0 && (reinterpret_cast<uintptr_t>(colors) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(v_out) & 15) == 0) {   // the 16-slot build (four waves per SIMD)
        // one 16-float record per Gaussian on a 64-byte boundary, geometry 0..5 | second array 6..: the merged flush
        const bool merged = GG_BWD_MERGE && gstride == 16 && seg2.cs2 == 16 && v_colors2 == v_xy + 6 && v_conic == v_xy + 2 &&
                            v_opacity == v_xy + 5 && (reinterpret_cast<uintptr_t>(v_xy) & 63) == 0;
        if (merged)
            hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, 0, 32, false, true, GG_BWD_S16 != 0, GG_BWD_S16 != 0>), grid,
                               block, 0, s, C, 0, 32, img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background,
                               final_Ts, final_idx, v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride, DetSlab(),
                               seg2);
        else
            hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, 0, 32, false, true, GG_BWD_S16 != 0>), grid, block, 0, s, C,
                               0, 32, img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, final_Ts, final_idx,
                               v_out, v_xy, v_conic, v_colors, v_opacity, gstride, cstride, DetSlab(), seg2);
        return;
    }
#ifdef GG_ABLATION
    switch (g_pair_ablate) {   // measurement twin (tools/kbench.py): wrong results on purpose
        case 1: B2_PAIR(1); return;
        case 2: B2_PAIR(2); return;
        case 3: B2_PAIR(3); return;
        case 4: B2_PAIR(4); return;
        case 5: B2_PAIR(5); return;
        case 6: B2_PAIR(6); return;
        default: break;
    }
#endif
    B2_PAIR(0);
}

#ifdef GG_ABLATION
// measurement-only entry (tools/kbench.py): ablated builds of the 3-channel backward
void gg_launch_blend2_bwd_ablate(int abl, int C, int off, int img_h, int img_w, int tiles_x,
                                 int ntiles, const int32_t *ids, const int2 *bins, const GRec *rec,
                                 const float *colors, const float *background, const float *final_Ts,
                                 const int32_t *final_idx, const float *v_out, float *v_xy,
                                 float *v_conic, float *v_colors, float *v_opacity, int gstride, int cstride, hipStream_t s) {
    dim3 grid(gg_blend_grid(ntiles, GG_WPB_OTHER)), block(64 * GG_WPB_OTHER);
    dim3 gridw(gg_blend_grid(ntiles, GG_WPB_WIDE_BWD)), blockw(64 * GG_WPB_WIDE_BWD);   // the wide kernels
    const int n = 3;   // B2_BWDN_ARGS: the ablated narrow builds are the 3-channel ones
    const DetSlab det = DetSlab();
    switch (abl) {
#define B2_WABL(L) hipLaunchKernelGGL((blend2_bwd_wide_kernel<true, L>), gridw, blockw, 0, s, C, off, 32, \
        img_h, img_w, tiles_x, ntiles, ids, bins, rec, colors, background, final_Ts, final_idx, v_out, \
        v_xy, v_conic, v_colors, v_opacity, gstride, cstride, det)
        case 11: B2_WABL(1); break;
        case 12: B2_WABL(2); break;
        case 13: B2_WABL(3); break;
        case 14: B2_WABL(4); break;
        case 15: B2_WABL(5); break;
        case 16: B2_WABL(6); break;
        case 1: hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 1>), grid, block, 0, s, B2_BWDN_ARGS); break;
        case 2: hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 2>), grid, block, 0, s, B2_BWDN_ARGS); break;
        case 3: hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 3>), grid, block, 0, s, B2_BWDN_ARGS); break;
        case 4: hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 4>), grid, block, 0, s, B2_BWDN_ARGS); break;
        default: hipLaunchKernelGGL((blend2_bwd_narrow_kernel<3, 0>), grid, block, 0, s, B2_BWDN_ARGS); break;
    }
}
// resident workgroups per CU of the backward kernels (tools/kbench.py prints them)
extern "C" int gg_debug_occupancy(int *out4) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend2_bwd_wide_kernel<true>, 64 * GG_WPB_WIDE_BWD, 0) != hipSuccess) return -1;
    out4[0] = n;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend2_bwd_wide_kernel<true, 0, 32, false, true>, 64 * GG_WPB_WIDE_BWD, 0) !=
        hipSuccess) return -1;
    out4[1] = n;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend2_bwd_narrow_kernel<8>, 64 * GG_WPB_OTHER, 0) != hipSuccess) return -1;
    out4[2] = n;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, blend2_fwd_kernel<32, true, true, true>, 64 * GG_WPB_OTHER, 0) != hipSuccess)
        return -1;
    out4[3] = n;
    return 0;
}
#endif  // GG_ABLATION
