// binning.hip — tile binning and depth ordering.
//
// The reference builds I int64 keys (tile_id << 32 | depth bits), sorts them globally with torch.sort (6+ radix
// passes over 12-byte pairs) and repeats that in each of its four rasterize calls.  The per-tile lists it obtains are
// ordered by (tile, depth bits, Gaussian id).  This file produces the SAME lists with far less HBM traffic by splitting
// the key:
//
//   1. the N Gaussians are ordered by (depth bits, id) — round 4: by buckets of the depth range, one wave ordering each
//      bucket (db_* below; 7 launches.  Rounds 1-3: a stable 4-pass LSD radix sort, 12 launches + 3 for the scan);
//      culled Gaussians go behind the visible ones;
//   2. the exclusive scan of num_tiles_hit in that order comes out of the same kernels (offset inside the bucket + the
//      bucket's offset);
//   3. every Gaussian, in depth order, emits (tile id, Gaussian id) for the tiles of its bbox;
//   4. stable LSD radix sort of the I pairs by tile id only (two passes: 7 + 6 bits at 1600x1200).
//
// Stability of the tile passes makes the result identical to the reference's global sort with ties broken by ascending
// Gaussian id (the order SURVEY a7 fixes); tests compare it bit for bit with the oracle's straightforward 64-bit sort.
//
// Radix pass = 3 launches: per-block digit histogram -> per-digit column scan -> stable scatter.  Within the scatter a
// wave ranks its keys with ballot-based match-any (one ballot per digit bit), so equal digits keep their input order
// without any LDS sorting network.
#include "scan.h"

// Device-side item count: the tile-sort kernels can take the number of intersections from device
// memory (n_dev != NULL), clamped to the capacity n the launch was sized for — the host then needs no
// read-back before it can enqueue them (gg_bin_sort_dev).
__device__ __forceinline__ int64_t dev_count(int64_t n, const int64_t *__restrict__ n_dev) {
    if (n_dev) {
        const int64_t d = *n_dev;
        if (d < n) n = d;
    }
    return n;
}

#define RS_THREADS 256
#define RS_WAVES (RS_THREADS / GG_WAVE)
// keys per thread: 16 for large inputs; 4 when that would leave fewer than ~4 workgroups per CU
// (the N = 1 M depth sort: 245 workgroups of 16 keys/thread ran latency-bound at 32 us per pass)
#ifndef GG_RS_BIG
#define GG_RS_BIG 16
#endif
#ifndef GG_RS_ATOMIC_RANK
#define GG_RS_ATOMIC_RANK 1
#endif
static inline int rs_items(int64_t n) { return n >= (int64_t)4096 * 1024 ? GG_RS_BIG : 4; }

// ---------------------------------------------------------------------------------------------
// sum(num_tiles_hit) -> device int64
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_kernel(int N, const int32_t *__restrict__ nth,
                                                    unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[4];
    unsigned long long acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        acc += (unsigned long long)(uint32_t)nth[i];
    // wave reduce through DPP on two 32-bit halves would need carries; a shuffle tree is fine here
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

extern "C" size_t gg_count_workspace(int num_points) {
    (void)num_points;
    return 0;
}
extern "C" int gg_count_intersects(int N, const int32_t *num_tiles_hit, int64_t *out, void *ws,
                                   size_t ws_bytes, gg_stream_t stream) {
    (void)ws;
    (void)ws_bytes;
    GG_REQUIRE(N >= 0, "num_points < 0");
    GG_REQUIRE(out != nullptr, "null output");
    hipStream_t s = (hipStream_t)stream;
    if (gg_fill_async(out, 0, sizeof(int64_t), s) != hipSuccess) {
        gg_set_error("gg_count_intersects: memset failed");
        return GG_ERR_LAUNCH;
    }
    if (N == 0) return GG_OK;
    GG_REQUIRE(num_tiles_hit != nullptr, "null pointer");
    int blocks = min((N + 255) / 256, 128);   // one same-address atomic per block: keep them few
    gg_prof_begin(GG_K_COUNT, s);
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, s, N, num_tiles_hit,
                       (unsigned long long *)out);
    gg_prof_end(GG_K_COUNT, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// radix pass, step 1: per-block digit histogram, stored digit-major: G[d * nblocks + b]
// ---------------------------------------------------------------------------------------------
// First pass of the depth sort: keys and values are not read but formed — key = depth bits of a visible Gaussian
// (radius > 0), 0xFFFFFFFF otherwise (culled ones sink to the end), value = its index (r03: this was a kernel of its
// own, 9 us and 8 MB written + read back per view)
struct DepthSrc {
    const float *depths;      // nullptr: keys / values come from memory
    const int32_t *radii;
};
__device__ __forceinline__ uint32_t depth_key(const DepthSrc d, int64_t idx) {
    return (d.radii[idx] > 0) ? __builtin_bit_cast(uint32_t, d.depths[idx]) : 0xFFFFFFFFu;
}

template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(
    int64_t n, const int64_t *__restrict__ n_dev, const uint32_t *__restrict__ keys, int shift,
    uint32_t mask, int nblocks, uint32_t *__restrict__ G, DepthSrc dsrc) {
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    __shared__ uint32_t hist[256];
    n = dev_count(n, n_dev);
    hist[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; ++it) {
        int64_t idx = base + (int64_t)it * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&hist[((dsrc.depths ? depth_key(dsrc, idx) : keys[idx]) >> shift) & mask], 1u);
    }
    __syncthreads();
    G[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// step 2: one block per digit: exclusive scan of its row of G in place, row total -> totals[d]
__global__ __launch_bounds__(256) void radix_colscan_kernel(int nblocks, uint32_t *__restrict__ G,
                                                            uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = G + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int start = 0; start < nblocks; start += 256) {
        int i = start + threadIdx.x;
        uint32_t v = (i < nblocks) ? row[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; ++w) wpre += wsum[w];
        uint32_t carry = carry_s;
        if (i < nblocks) row[i] = carry + wpre + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + wpre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// step 3: stable scatter
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    int64_t n, const int64_t *__restrict__ n_dev, const uint32_t *__restrict__ keys_in,
    const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int shift, uint32_t mask, int nblocks,
    const uint32_t *__restrict__ G, const uint32_t *__restrict__ totals, DepthSrc dsrc) {
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    n = dev_count(n, n_dev);
    const int nbits = 32 - __builtin_clz(mask | 1u);   // ballots per key: the digit's bits (a 6-bit pass ranks with 6, not 8)
    __shared__ uint32_t whist[RS_WAVES][256];
    __shared__ uint32_t digit_base[256];
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // exclusive scan of the 256 digit totals (same in every block; 1 KB, L2 resident)
    {
        uint32_t v = totals[tid];
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        for (int w = 0; w < RS_WAVES; ++w) whist[w][tid] = 0;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; ++w) wpre += wsum[w];
        digit_base[tid] = wpre + incl - v + G[(size_t)tid * nblocks + blockIdx.x];
    }
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (GG_WAVE * RS_ITEMS);
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#if !GG_RS_ATOMIC_RANK
    volatile uint32_t *wh = whist[wave];
#endif
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        int64_t idx = wbase + (int64_t)it * GG_WAVE + lane;
        bool valid = idx < n;
        key[it] = valid ? (dsrc.depths ? depth_key(dsrc, idx) : keys_in[idx]) : 0u;
        val[it] = valid ? (dsrc.depths ? (uint32_t)idx : vals_in[idx]) : 0u;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        int64_t idx = wbase + (int64_t)it * GG_WAVE + lane;
        bool valid = idx < n;
        uint32_t d = (key[it] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b < nbits) {                              // (uniform)
                bool bit = (d >> b) & 1u;
                uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
        }
        uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t before = (uint32_t)__popcll(peers & lt_mask);
#if GG_RS_ATOMIC_RANK
        // the group's first lane takes the digit's running count with ONE returning LDS add and hands it to the group
        // through the lane crossbar; the adds of the 16 rounds are independent instructions in program order (a wave's
        // LDS operations execute in order), so their latencies overlap — the read / write / barrier / read of the first
        // version was three dependent LDS round trips per key
        uint32_t old = 0u;
        if (valid && before == 0) old = atomicAdd(&whist[wave][d], cnt);
        old = __shfl(old, (int)__builtin_ctzll(peers | (1ull << 63)), 64);
        rank[it] = old + before;
#else
        if (valid && before == 0) wh[d] = wh[d] + cnt;  // group leader (lowest lane)
        __builtin_amdgcn_wave_barrier();
        uint32_t after = valid ? wh[d] : 0u;
        rank[it] = after - cnt + before;
        __builtin_amdgcn_wave_barrier();
#endif
    }
    __syncthreads();
    // Per digit: offsets of the waves inside the block's run of that digit, the block's count, and —
    // exclusive scan over the digits — where the run starts in the block's locally sorted order.
    __shared__ uint32_t lstart[256];
    __shared__ uint32_t skey[RS_TILE], sval[RS_TILE];
    {
        uint32_t run = 0;
        for (int w = 0; w < RS_WAVES; ++w) {
            uint32_t c = whist[w][tid];
            whist[w][tid] = run;
            run += c;
        }
        uint32_t incl = run;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; ++w) wpre += wsum[w];
        lstart[tid] = wpre + incl - run;
    }
    __syncthreads();
    // The keys go through LDS in locally sorted order, so that consecutive threads write consecutive
    // addresses of a digit's run: a store instruction touches a few cache lines instead of one per
    // lane (the direct scatter moved 85 MB in 63 us at I = 5.3 M).
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        int64_t idx = wbase + (int64_t)it * GG_WAVE + lane;
        if (idx < n) {
            uint32_t d = (key[it] >> shift) & mask;
            uint32_t lp = lstart[d] + whist[wave][d] + rank[it];
            skey[lp] = key[it];
            sval[lp] = val[it];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)blockIdx.x * RS_TILE;
    const int count = (int)(left < RS_TILE ? left : RS_TILE);
    for (int jj = tid; jj < count; jj += RS_THREADS) {
        const uint32_t k = skey[jj];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t dst = digit_base[d] + ((uint32_t)jj - lstart[d]);
        keys_out[dst] = k;
        vals_out[dst] = sval[jj];
    }
}


// ---------------------------------------------------------------------------------------------
// Depth order of the Gaussians by buckets (round 4).  The stable 4-pass radix sort above is 12 launches of which every
// one runs at launch latency (N = 1 M: 8-13 us each, 103 us of the 282 us binning of the bench view), and the offsets
// scan behind it is three more (22 us).  Instead:
//   db_range    min / max of the visible Gaussians' depth bits (256 partial pairs; every later kernel reduces them);
//   db_count    bucket = floor((bits - min) NB / (max - min + 1)) — monotone in the bits, so ordering the buckets and
//               then each bucket's members by (bits, id) IS the order by (bits, id); NB ~ N / 64; DB_BLOCKS workgroups
//               count their contiguous share of the Gaussians per bucket in LDS -> G[block][bucket]; culled Gaussians
//               (radius <= 0) go to one extra last bucket;
//   db_prefix   G[block][bucket] -> the block's first position inside the bucket; bucket totals;
//   db_starts   exclusive scan of the bucket totals (one workgroup) -> bucket starts; the visible count;
//   db_scatter  the same walk as db_count; every Gaussian's record {bits, id, tiles hit, tile box} — 16 bytes, one store
//               request — goes to its bucket's run (position from an LDS cursor: the order inside a run is whatever the
//               atomics gave);
//   db_sort     ONE WAVE per bucket: a run of <= 256 entries (mean 64) is ordered by counting, for every word, the words
//               below it (v_readlane broadcasts, no LDS, no dependent chain); up to 512 entries by stable LSD byte passes
//               in LDS (depth bytes first, passes whose digit is constant skipped, the id bytes only if equal depths came
//               out of order); then the wave writes the ids in order and, per position, {the exclusive
//               scan of the tile counts inside the bucket, the bucket, the tile box, the count}, and the bucket's sum; longer runs (bit-identical depths
//               en masse) go through a compare-exchange network in global memory first;
//   db_offsets  exclusive scan of the bucket sums (one workgroup).
// emit_kernel adds a Gaussian's in-bucket offset to its bucket's and reads nothing by Gaussian id.  7 launches instead of
// 15; same order bit for bit
// (tests/test_gpu_parity.py binning tests, incl. depth ties and runs beyond 512).
// ---------------------------------------------------------------------------------------------
typedef unsigned long long u64;
#define DB_BLOCKS 64
#define DB_THREADS 1024
#define DB_CAP 512             // entries one wave sorts in LDS (8 per lane)
#define DB_MAX_BUCKETS 32768   // LDS histogram of db_walk_kernel: 128 KB of the CU's 160
static inline int db_buckets(int N) {
    int nb = 256;
    while (nb < DB_MAX_BUCKETS && nb * 64 < N) nb <<= 1;
    return nb;
}
struct DbRange {
    uint32_t kmin;
    u64 mul;   // bucket = ((bits - kmin) * mul) >> 32
};
// partial minima / maxima of the visible depth bits: db_range_kernel's 256 pairs, or the per-workgroup ones the caller's
// projection left behind (gg_view_fwd -> gg_bin_sort_dev_ex); every workgroup reduces them itself (a few KB from L2)
struct DbParts {
    const uint32_t *lo, *hi;
    int n;
};
__device__ __forceinline__ DbRange db_range_of(const DbParts mm, int nb, uint32_t *s_mm /*[2]*/) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int i = tid; i < mm.n; i += 64) {
            lo = min(lo, mm.lo[i]);
            hi = max(hi, mm.hi[i]);
        }
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
        }
        if (tid == 0) { s_mm[0] = lo; s_mm[1] = hi; }
    }
    __syncthreads();
    DbRange r;
    const uint32_t lo = s_mm[0], hi = s_mm[1];
    r.kmin = lo;
    const u64 span = hi >= lo ? (u64)(hi - lo) + 1ull : 1ull;     // (nothing visible: every Gaussian is in the last bucket)
    r.mul = (((u64)nb) << 32) / span;                              // (key - kmin) * mul < nb * 2^32
    return r;
}
__device__ __forceinline__ uint32_t db_bucket(const DbRange r, uint32_t key, int nb) {
    if (key == 0xFFFFFFFFu) return (uint32_t)nb;                   // culled
    return min((uint32_t)(((u64)(key - r.kmin) * r.mul) >> 32), (uint32_t)(nb - 1));
}
__global__ __launch_bounds__(256) void db_range_kernel(int N, DepthSrc dsrc, uint32_t *__restrict__ mm) {
    __shared__ uint32_t s_lo[4], s_hi[4];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    const int stride = 256 * gridDim.x;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += 4 * stride) {      // four loads in flight per thread
        uint32_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = (i + u * stride < N) ? depth_key(dsrc, i + u * stride) : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k[u] != 0xFFFFFFFFu) { lo = min(lo, k[u]); hi = max(hi, k[u]); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[blockIdx.x] = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        mm[gridDim.x + blockIdx.x] = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    }
}
// What travels with a Gaussian through the bucket sort: {depth bits, id, tiles hit, tile box}.  The box — x0 | y0 << 10 |
// width << 20 of gg_tile_bbox — and the count are formed where the Gaussians are still walked in index order (coalesced
// reads of xys / radii), so that neither the sort's scan of the counts nor the emission gathers per-Gaussian data by id
// (1 M scattered reads each: 16 and 35 us of the first version of this path).  Tile grids up to 1023 x 1023.
#define DB_BOX_MAX 1023
struct DbGeom {
    const float *xys;
    int tiles_x, tiles_y;
};
__device__ __forceinline__ void db_box(const DbGeom g, const DepthSrc d, int i, uint32_t &cnt, uint32_t &box) {
    const int rad = d.radii[i];
    cnt = 0u;
    box = 1u << 20;
    if (rad > 0) {
        int x0, y0, x1, y1;
        gg_tile_bbox(g.xys[2 * (size_t)i], g.xys[2 * (size_t)i + 1], (float)rad, g.tiles_x, g.tiles_y, x0, y0, x1, y1);
        const int bw = x1 - x0;
        if (bw > 0 && y1 > y0) {
            cnt = (uint32_t)(bw * (y1 - y0));
            box = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)bw << 20);
        }
    }
}
// SCATTER false: G[block][bucket] = the block's count; true: G holds the block's first position in each bucket
template <bool SCATTER>
__global__ __launch_bounds__(DB_THREADS) void db_walk_kernel(int N, int nb, DepthSrc dsrc, DbGeom geom,
                                                             const DbParts mm, uint32_t *__restrict__ G,
                                                             const uint32_t *__restrict__ start, uint4 *__restrict__ recs,
                                                             int32_t *__restrict__ zero = nullptr, int nzero = 0) {
    // (the counting walk is the chain's first kernel with threads to spare: it zeroes the tile ranges for tile_bins_kernel)
    for (int i = blockIdx.x * DB_THREADS + threadIdx.x; i < nzero; i += DB_BLOCKS * DB_THREADS) zero[i] = 0;
    extern __shared__ uint32_t s_hist[];   // nb + 1 counters (SCATTER: cursors), then the range
    uint32_t *s_mm = s_hist + nb + 1;
    const DbRange r = db_range_of(mm, nb, s_mm);
    const int per = (N + DB_BLOCKS - 1) / DB_BLOCKS;
    const int lo = blockIdx.x * per, hi = min(N, lo + per);
    uint32_t *row = G + (size_t)blockIdx.x * (nb + 1);
    for (int b = threadIdx.x; b <= nb; b += DB_THREADS) s_hist[b] = SCATTER ? start[b] + row[b] : 0u;
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += 4 * DB_THREADS) {                 // four loads in flight per thread
        uint32_t k[4], cnt[4], box[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int iu = i + u * DB_THREADS;
            k[u] = iu < hi ? depth_key(dsrc, iu) : 0u;
            cnt[u] = box[u] = 0u;
            if (SCATTER && iu < hi) db_box(geom, dsrc, iu, cnt[u], box[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * DB_THREADS < hi) {
                const uint32_t pos = atomicAdd(&s_hist[db_bucket(r, k[u], nb)], 1u);
                if (SCATTER && k[u] != 0xFFFFFFFFu)                            // (nothing reads the culled run)
                    recs[pos] = make_uint4(k[u], (uint32_t)(i + u * DB_THREADS), cnt[u], box[u]);
            }
        }
    }
    if (!SCATTER) {
        __syncthreads();
        for (int b = threadIdx.x; b <= nb; b += DB_THREADS) row[b] = s_hist[b];
    }
}
// one thread per bucket: counts of the blocks -> exclusive over the blocks, in place; the bucket's total
__global__ __launch_bounds__(256) void db_prefix_kernel(int nb, uint32_t *__restrict__ G, uint32_t *__restrict__ total) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > nb) return;
    uint32_t run = 0;
    uint32_t c[DB_BLOCKS];
#pragma unroll
    for (int k = 0; k < DB_BLOCKS; ++k) c[k] = G[(size_t)k * (nb + 1) + b];     // (all in flight)
#pragma unroll
    for (int k = 0; k < DB_BLOCKS; ++k) {
        G[(size_t)k * (nb + 1) + b] = run;
        run += c[k];
    }
    total[b] = run;
}
// one workgroup: v[0 .. n) -> exclusive scan in place, v[n] = total.  A round covers 4 096 consecutive values (four per
// thread: coalesced 16-byte accesses); the values of four rounds are requested before the first is scanned, and a round
// costs one barrier (the wave sums alternate between two LDS rows, the carry is the same number in every thread)
__global__ __launch_bounds__(1024) void db_scan_kernel(int n, uint32_t *__restrict__ v) {
    __shared__ uint32_t s_w[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry = 0u;
    int round = 0;
    for (int base = 0; base < n; base += 4 * 4096) {
        uint32_t x[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + 4096 * q + 4 * tid + k;
                x[q][k] = i < n ? v[i] : 0u;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q, ++round) {
            if (base + 4096 * q >= n) break;
            const uint32_t mine = x[q][0] + x[q][1] + x[q][2] + x[q][3];
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            uint32_t *sw = s_w[round & 1];
            if (lane == 63) sw[wave] = incl;
            __syncthreads();
            uint32_t ex = carry + incl - mine, all = 0u;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const uint32_t t = sw[w];
                if (w < wave) ex += t;
                all += t;
            }
            carry += all;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + 4096 * q + 4 * tid + k;
                if (i < n) v[i] = ex;
                ex += x[q][k];
            }
        }
    }
    if (tid == 0) v[n] = carry;
}

// compare-exchange network over u64 keys in global memory by ONE WAVE (its own stores are visible to its later loads behind
// a workgroup-scope fence: one CU, one L1).  Every exchange puts the smaller key at the lower index (the "flip" form: the
// first step of a merge of size k compares i with k - 1 - i), so keys at indices >= n can be VIRTUAL +inf: nothing is ever
// stored there — no padding, any n.  Blocks of 8 consecutive keys are handled in registers.
__device__ __forceinline__ void db_ce(u64 &a, u64 &b) {
    const bool sw = b < a;
    const u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}
__device__ __forceinline__ void db_bitonic64(u64 *buf, const int n) {
    constexpr int NB = 8, LNB = 3, THREADS = 64;
    const int tid = threadIdx.x & 63;
    int P = NB;
    while (P < n) P <<= 1;
    const u64 INF = ~0ull;
    for (int blk = tid; blk * NB < n; blk += THREADS) {   // every aligned block of NB keys: sorted
        u64 v[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) v[r] = (NB * blk + r < n) ? buf[NB * blk + r] : INF;
#pragma unroll
        for (int k = 2; k <= NB; k <<= 1) {
#pragma unroll
            for (int i = 0; i < NB / 2; ++i) {
                const int b2 = i / (k / 2), off = i % (k / 2);
                db_ce(v[b2 * k + off], v[b2 * k + k - 1 - off]);
            }
#pragma unroll
            for (int j = k >> 2; j >= 1; j >>= 1)
#pragma unroll
                for (int i = 0; i < NB / 2; ++i) {
                    const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    db_ce(v[lo], v[lo + j]);
                }
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if (NB * blk + r < n) buf[NB * blk + r] = v[r];
    }
    { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    for (int k = 2 * NB, lk = LNB + 1; k <= P; k <<= 1, ++lk) {
        const int hk = k >> 1;
        for (int i = tid; i < (P >> 1); i += THREADS) {   // flip: i-th key of a block's lower half with its mirror
            const int blk = i >> (lk - 1), off = i & (hk - 1);
            const int lo = blk * k + off, hi = blk * k + k - 1 - off;
            if (hi < n) {
                const u64 a = buf[lo], b = buf[hi];
                if (b < a) { buf[lo] = b; buf[hi] = a; }
            }
        }
        { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        for (int j = k >> 2; j >= NB; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += THREADS) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo + j;
                if (hi < n) {
                    const u64 a = buf[lo], b = buf[hi];
                    if (b < a) { buf[lo] = b; buf[hi] = a; }
                }
            }
            { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        }
        for (int blk = tid; blk * NB < n; blk += THREADS) {   // strides NB / 2 .. 1
            u64 v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) v[r] = (NB * blk + r < n) ? buf[NB * blk + r] : INF;
#pragma unroll
            for (int j = NB >> 1; j >= 1; j >>= 1)
#pragma unroll
                for (int i = 0; i < NB / 2; ++i) {
                    const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    db_ce(v[lo], v[lo + j]);
                }
#pragma unroll
            for (int r = 0; r < NB; ++r)
                if (NB * blk + r < n) buf[NB * blk + r] = v[r];
        }
        { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    }
}

// one WAVE per bucket, four buckets per workgroup; the waves never synchronise with each other
#define DB_ITEMS (DB_CAP / 64)
#define DB_RANK_CAP 256        // runs up to this length are ordered by counting, for every key, the keys below it
#define DB_WPB 4
__device__ __forceinline__ void db_wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// rank[t] += the number of words of buf[0 .. 4 ceil(n / 4)) below key[t], t < NS
template <int NS>
__device__ __forceinline__ void db_rank(const u64 *buf, const int n, const u64 (&key)[DB_RANK_CAP / 64],
                                        uint32_t (&rank)[DB_RANK_CAP / 64]) {
    for (int j = 0; j < n; j += 4) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 a = *reinterpret_cast<const u64x2 *>(buf + j), b = *reinterpret_cast<const u64x2 *>(buf + j + 2);
#pragma unroll
        for (int t = 0; t < NS; ++t)
            rank[t] += (uint32_t)(a[0] < key[t]) + (uint32_t)(a[1] < key[t]) + (uint32_t)(b[0] < key[t]) +
                       (uint32_t)(b[1] < key[t]);
    }
}
__global__ __launch_bounds__(64 * DB_WPB) void db_sort_kernel(int N, int nb, const uint32_t *__restrict__ start,
                                                              const uint4 *__restrict__ recs, u64 *__restrict__ pairs,
                                                              DepthSrc dsrc, DbGeom geom, uint32_t *__restrict__ order,
                                                              uint4 *__restrict__ einfo, uint32_t *__restrict__ bucket_sum) {
    __shared__ __attribute__((aligned(16))) u64 s_buf[DB_WPB][DB_CAP];
    __shared__ uint32_t s_hist[DB_WPB][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * DB_WPB + wave;
    if (b >= nb) return;          // (bucket nb holds the culled Gaussians: they emit nothing and nothing reads their order)
    u64 *buf = s_buf[wave];
    uint32_t *hist = s_hist[wave];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int s0 = (int)start[b], n = (int)start[b + 1] - s0;
    if (n <= 0) {
        if (lane == 0) bucket_sum[b] = 0u;
        return;
    }
    // ids, tile counts and tile boxes of positions 64 r + lane of the sorted run (runs <= DB_CAP)
    uint32_t gid[DB_ITEMS], cntv[DB_ITEMS], boxv[DB_ITEMS];
    bool carried = false;         // counts / boxes came with the records (else: looked up by id below)
    if (n <= DB_RANK_CAP) {
        carried = true;
        // every key's place = the number of keys below it (the words are distinct: they end in the id).  Key j comes to all
        // lanes by v_readlane, no LDS, no dependent chain: ~n^2 / 64 compare-and-add per lane (mean run: 128 entries)
        constexpr int RS = DB_RANK_CAP / 64;
        u64 key[RS];
        uint32_t rank[RS], pc[RS], pb[RS];
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            const bool in = 64 * t + lane < n;
            const uint4 rc = in ? recs[s0 + 64 * t + lane] : make_uint4(~0u, ~0u, 0u, 0u);
            key[t] = ((u64)rc.x << 32) | rc.y;
            pc[t] = rc.z;
            pb[t] = rc.w;
            rank[t] = 0u;
        }
        const int nslots = (n + 63) >> 6;
        // the words go to LDS once (positions >= n hold the largest word: below nothing) and come back as wave-uniform
        // broadcast reads, four per step; a lane compares them with the 1..4 words it owns — no branch inside the loop
#pragma unroll
        for (int t = 0; t < RS; ++t)
            if (t < nslots) buf[64 * t + lane] = key[t];
        db_wsync();
        {
            if (nslots == 1) db_rank<1>(buf, n, key, rank);
            else if (nslots == 2) db_rank<2>(buf, n, key, rank);
            else if (nslots == 3) db_rank<3>(buf, n, key, rank);
            else db_rank<4>(buf, n, key, rank);
        }
        db_wsync();
        // id, count and box go to their places through LDS ([256] words each) and come back in order
        uint32_t *xw = reinterpret_cast<uint32_t *>(buf);
#pragma unroll
        for (int t = 0; t < RS; ++t)
            if (64 * t + lane < n) {
                xw[rank[t]] = (uint32_t)key[t];
                xw[DB_RANK_CAP + rank[t]] = pc[t];
                xw[2 * DB_RANK_CAP + rank[t]] = pb[t];
            }
        db_wsync();
#pragma unroll
        for (int r = 0; r < DB_ITEMS; ++r) {
            const bool in = r < RS && 64 * r + lane < n;
            gid[r] = in ? xw[64 * r + lane] : 0u;
            cntv[r] = in ? xw[DB_RANK_CAP + 64 * r + lane] : 0u;
            boxv[r] = in ? xw[2 * DB_RANK_CAP + 64 * r + lane] : 0u;
        }
    } else if (n <= DB_CAP) {
        // stable LSD byte passes in LDS (round 4's per-tile sort, profiles/r04_counting_sort_binning_experiment.patch)
        // (validity of a slot is recomputed from a laundered n - lane wherever it is needed: as loop invariants the lane
        //  masks of `64 r + lane < n` took 2 SGPRs each for the whole kernel)
#define DB_LIM(nv) int nv = n - lane; asm volatile("" : "+v"(nv))
        u64 key[DB_ITEMS];
        {
            DB_LIM(nv);
#pragma unroll
            for (int r = 0; r < DB_ITEMS; ++r) {
                const uint4 rc = (64 * r < nv) ? recs[s0 + 64 * r + lane] : make_uint4(~0u, ~0u, 0u, 0u);
                key[r] = ((u64)rc.x << 32) | rc.y;
            }
        }
        // one stable pass on bits [shift, shift + 8); nothing moves if the digit is the same for every key
        auto pass = [&](const int shift) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hist[lane + 64 * k] = 0u;
            db_wsync();
            {
                DB_LIM(nv);
#pragma unroll
                for (int r = 0; r < DB_ITEMS; ++r)
                    if (64 * r < nv) atomicAdd(&hist[(uint32_t)(key[r] >> shift) & 255u], 1u);
            }
            db_wsync();
            // digits [4 lane, 4 lane + 4): totals -> exclusive starts
            uint32_t tot[4], mine = 0;
            bool constant = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tot[k] = hist[4 * lane + k];
                constant = constant || tot[k] == (uint32_t)n;
                mine += tot[k];
            }
            if (__ballot(constant) != 0ull) {
                db_wsync();
                return;
            }
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(incl, off, 64);
                if (lane >= off) incl += u;
            }
            uint32_t ex = incl - mine;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hist[4 * lane + k] = ex;
                ex += tot[k];
            }
            db_wsync();
            volatile uint32_t *wh = hist;
            DB_LIM(nv_s);
#pragma unroll
            for (int r = 0; r < DB_ITEMS; ++r) {
                if (64 * r < n) {                                          // (wave-uniform)
                    const bool valid = 64 * r < nv_s;
                    const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
                    uint64_t peers = __ballot(valid);
#pragma unroll
                    for (int bit = 0; bit < 8; ++bit) {
                        const bool on = (d >> bit) & 1u;
                        const uint64_t m = __ballot(on);
                        peers &= on ? m : ~m;
                    }
                    const uint32_t cnt = (uint32_t)__popcll(peers);
                    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
                    if (valid && before == 0) wh[d] = wh[d] + cnt;       // the group's first lane moves the digit's cursor
                    db_wsync();
                    if (valid) buf[wh[d] - cnt + before] = key[r];
                    db_wsync();
                    __builtin_amdgcn_sched_barrier(0);   // (or the ballots of all rounds are formed up front: 270 SGPRs)
                }
            }
            {
                DB_LIM(nv);
#pragma unroll
                for (int r = 0; r < DB_ITEMS; ++r) key[r] = (64 * r < nv) ? buf[64 * r + lane] : ~0ull;
            }
            db_wsync();
        };
        // byte p of the word: 0..3 the id, 4..7 the depth bits.  The depth bytes first; a second round with the id bytes in
        // front only if that left equal depths out of order.  (ONE copy of the pass body: runtime loops.)
#pragma nounroll
        for (int round = 0; round < 2; ++round) {
#pragma nounroll
            for (int pb = round == 0 ? 4 : 0; pb < 8; ++pb) pass(8 * pb);
            if (round == 1) break;
            {
                DB_LIM(nv);
#pragma unroll
                for (int r = 0; r < DB_ITEMS; ++r)
                    if (64 * r < nv) buf[64 * r + lane] = key[r];
            }
            db_wsync();
            bool bad = false;
            {
                DB_LIM(nv);
#pragma unroll
                for (int r = 0; r < DB_ITEMS; ++r)
                    if (64 * r + 1 < nv) {
                        const u64 nx = buf[64 * r + lane + 1];
                        bad = bad || ((uint32_t)(nx >> 32) == (uint32_t)(key[r] >> 32) && (uint32_t)nx < (uint32_t)key[r]);
                    }
            }
            db_wsync();
            if (__ballot(bad) == 0ull) break;
        }
#undef DB_LIM
#pragma unroll
        for (int r = 0; r < DB_ITEMS; ++r) gid[r] = (uint32_t)key[r];
    } else {                             // (rare: more than 512 Gaussians in 1 / nb of the depth range)
        for (int i = lane; i < n; i += 64) {
            const uint4 rc = recs[s0 + i];
            pairs[s0 + i] = ((u64)rc.x << 32) | rc.y;
        }
        db_wsync();
        db_bitonic64(pairs + s0, n);
    }
    // output: ids in order; per position {offset inside the bucket (exclusive scan of the counts), bucket, box, count}
    uint32_t carry = 0;
    for (int c0 = 0; c0 < n; c0 += DB_CAP) {
        if (n > DB_CAP) {
#pragma unroll
            for (int r = 0; r < DB_ITEMS; ++r) gid[r] = (c0 + 64 * r + lane < n) ? (uint32_t)pairs[s0 + c0 + 64 * r + lane] : 0u;
        }
#pragma unroll
        for (int r = 0; r < DB_ITEMS; ++r) {
            const int p = c0 + 64 * r + lane;
            if (c0 + 64 * r < n) {                                        // (wave-uniform)
                const bool valid = p < n;
                const uint32_t g = min(gid[r], (uint32_t)(N - 1));
                uint32_t c = cntv[r], bx = boxv[r];
                if (!carried) {                                           // (wave-uniform)
                    c = 0u;
                    bx = 1u << 20;
                    if (valid) db_box(geom, dsrc, (int)g, c, bx);
                }
                if (!valid) c = 0u;
                uint32_t incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t u = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += u;
                }
                if (valid) {
                    order[s0 + p] = g;
                    einfo[s0 + p] = make_uint4(carry + incl - c, (uint32_t)b, bx, c);
                }
                carry += __shfl(incl, 63, 64);
            }
        }
    }
    if (lane == 0) bucket_sum[b] = carry;
}

// ---------------------------------------------------------------------------------------------
// emit (tile id, Gaussian id) in depth order; row-major inside the bbox like the reference's
// map_gaussian_to_intersects
// ---------------------------------------------------------------------------------------------
// Wave-cooperative expansion: a wave takes 64 consecutive Gaussians of the depth order, keeps their
// tile boxes and start offsets in LDS, and then the LANES walk the wave's contiguous output range —
// lane j of an iteration finds its source Gaussian by binary search over the 64 start offsets — so
// every store instruction writes 64 consecutive entries.  (The first version looped per Gaussian over
// its own tiles: 64 short runs per instruction and as many iterations as the largest box of the wave.)
// einfo[r] = {offset inside the bucket, bucket, tile box, tile count} of the r-th Gaussian of the depth order
// (db_sort_kernel), bucket_base: the buckets' offsets (db_scan_kernel); only the first *visible positions carry them
__global__ __launch_bounds__(256) void emit_kernel(int N, const uint32_t *__restrict__ order,
                                                   const uint4 *__restrict__ einfo,
                                                   const uint32_t *__restrict__ bucket_base,
                                                   const uint32_t *__restrict__ visible, int tiles_x,
                                                   int64_t I, const int64_t *__restrict__ I_dev,
                                                   uint32_t *__restrict__ tkeys,
                                                   uint32_t *__restrict__ tvals) {
    __shared__ uint32_t s_rel[4][64];
    I = dev_count(I, I_dev);   // start of each Gaussian's run, relative to the wave's first
    __shared__ int4 s_box[4][64];       // x0, y0, box width, Gaussian id
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t g = 0, cnt = 0, off = 0;
    int x0 = 0, y0 = 0, bw = 1;
    if (r < N && (uint32_t)r < *visible) {
        g = min(order[r], (uint32_t)(N - 1));
        const uint4 e = einfo[r];
        off = e.x + bucket_base[e.y];
        x0 = (int)(e.z & 1023u);
        y0 = (int)((e.z >> 10) & 1023u);
        bw = max(1, (int)(e.z >> 20));
        cnt = e.w;
    }
    // offsets[] is the exclusive scan of the counts in this order, so the wave's range starts at lane
    // 0's offset and the relative starts are an exclusive scan of cnt over the lanes
    uint32_t incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    const uint32_t base = __shfl(off, 0, 64);
    s_rel[wave][lane] = incl - cnt;
    s_box[wave][lane] = make_int4(x0, y0, bw, (int)g);
    __builtin_amdgcn_wave_barrier();
    const uint32_t *rel = s_rel[wave];
    for (uint32_t j = lane; j < total; j += 64) {
        int lo = 0;                       // largest s with rel[s] <= j (empty runs share a start: the
#pragma unroll                             // last of them is the one that owns j, the others own nothing)
        for (int step = 32; step > 0; step >>= 1)
            if (rel[lo + step] <= j) lo += step;
        const int4 b = s_box[wave][lo];
        const uint32_t k = j - rel[lo];
        const uint32_t row = k / (uint32_t)b.z;
        const int64_t cur = (int64_t)base + j;
        if (cur < I) {  // guards a caller-supplied I smaller than the true total
            tkeys[cur] = (uint32_t)((b.y + (int)row) * tiles_x + b.x + (int)(k - row * (uint32_t)b.z));
            tvals[cur] = (uint32_t)b.w;
        }
    }
}

__global__ __launch_bounds__(256) void tile_bins_kernel(int64_t I, const int64_t *__restrict__ I_dev,
                                                        const uint32_t *__restrict__ tkeys_sorted,
                                                        uint32_t num_tiles,
                                                        int32_t *__restrict__ tile_bins,
                                                        int32_t *__restrict__ tile_out) {
    I = dev_count(I, I_dev);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    uint32_t cur = tkeys_sorted[i];
    if (tile_out) tile_out[i] = (int32_t)cur;
    // tile ids >= num_tiles can only appear if the caller's I exceeds sum(num_tiles_hit) (entries
    // never emitted); they are ignored instead of being used as an index
    const bool cur_ok = cur < num_tiles;
    if (i == 0 && cur_ok) tile_bins[2 * cur] = 0;
    if (i == I - 1 && cur_ok) tile_bins[2 * cur + 1] = (int32_t)I;
    if (i > 0) {
        uint32_t prev = tkeys_sorted[i - 1];
        if (prev != cur) {
            if (prev < num_tiles) tile_bins[2 * prev + 1] = (int32_t)i;
            if (cur_ok) tile_bins[2 * cur] = (int32_t)i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int radix_nblocks(int64_t n) {
    const int64_t tile = (int64_t)RS_THREADS * rs_items(n);
    return (int)((n + tile - 1) / tile);
}

struct BinWs {
    uint32_t *order;                          // N: Gaussian ids in depth order (the visible ones first)
    uint4 *einfo;                             // N: {offset inside the bucket, bucket, tile box, tile count} in that order
    uint32_t *G;                              // 256 * max nblocks
    uint32_t *totals;                         // 256
    uint32_t *tkeyA, *tkeyB, *tvalTmp;        // I each
    // depth order by buckets (db_*)
    uint4 *recs;                              // N: {depth bits, id, tile count, tile box}, grouped by bucket
    u64 *pairs;                               // N: scratch of the runs too long for LDS (depth bits << 32 | id)
    uint32_t *dbG;                            // DB_BLOCKS x (buckets + 1)
    uint32_t *dbStart;                        // buckets + 2: bucket totals -> starts; [buckets] = the visible count
    uint32_t *dbBase;                         // buckets + 1: bucket sums of num_tiles_hit -> offsets
    uint32_t *dbmm;                           // 256 + 256: partial minima | maxima of the depth bits
    size_t bytes;
};
static BinWs bin_ws_layout(void *ws, int N, int64_t I) {
    BinWs w;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        char *p = ws ? (char *)ws + off : nullptr;
        off += gg_align_up(nbytes, 256);
        return (uint32_t *)p;
    };
    size_t n = (size_t)(N > 0 ? N : 1), i = (size_t)(I > 0 ? I : 1);
    w.order = take(4 * n);
    w.einfo = (uint4 *)take(16 * n);
    int nb = max(radix_nblocks(N), radix_nblocks(I));
    w.G = take(4 * 256 * (size_t)(nb + 1));
    w.totals = take(4 * 256);
    w.tkeyA = take(4 * i);
    w.tkeyB = take(4 * i);
    w.tvalTmp = take(4 * i);
    const size_t nbk = (size_t)db_buckets(N);
    w.recs = (uint4 *)take(16 * n);
    w.pairs = (u64 *)take(8 * n);
    w.dbG = take(4 * DB_BLOCKS * (nbk + 1));
    w.dbStart = take(4 * (nbk + 2));
    w.dbBase = take(4 * (nbk + 1));
    w.dbmm = take(4 * 2 * 256);
    w.bytes = off;
    return w;
}
extern "C" size_t gg_bin_sort_workspace(int num_points, int64_t num_intersects) {
    return bin_ws_layout(nullptr, num_points, num_intersects).bytes;
}

static void radix_pass(int64_t n, const int64_t *n_dev, const uint32_t *kin, const uint32_t *vin,
                       uint32_t *kout, uint32_t *vout, int shift, uint32_t mask, BinWs &w, hipStream_t s,
                       DepthSrc dsrc = DepthSrc{nullptr, nullptr}) {
    int nb = radix_nblocks(n);
    if (rs_items(n) != 4)
        hipLaunchKernelGGL(radix_hist_kernel<GG_RS_BIG>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, shift,
                           mask, nb, w.G, dsrc);
    else
        hipLaunchKernelGGL(radix_hist_kernel<4>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, shift,
                           mask, nb, w.G, dsrc);
    hipLaunchKernelGGL(radix_colscan_kernel, dim3(256), dim3(256), 0, s, nb, w.G, w.totals);
    if (rs_items(n) != 4)
        hipLaunchKernelGGL(radix_scatter_kernel<GG_RS_BIG>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, vin,
                           kout, vout, shift, mask, nb, w.G, w.totals, dsrc);
    else
        hipLaunchKernelGGL(radix_scatter_kernel<4>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, vin,
                           kout, vout, shift, mask, nb, w.G, w.totals, dsrc);
}

static int bin_sort_impl(int N, int64_t I, const int64_t *I_dev, const float *xys, const float *depths,
                         const int32_t *radii, const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                         int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *isect_tile_sorted,
                         void *ws, size_t ws_bytes, gg_stream_t stream, const uint32_t *range_lo = nullptr,
                         const uint32_t *range_hi = nullptr, int range_parts = 0) {
    GG_REQUIRE(N >= 0 && I >= 0, "negative size");
    GG_REQUIRE(tiles_x > 0 && tiles_y > 0, "empty tile grid");
    GG_REQUIRE(tiles_x <= DB_BOX_MAX && tiles_y <= DB_BOX_MAX, "tile grid beyond 1023 x 1023 (images beyond 16 368 pixels a side)");
    GG_REQUIRE(I < (int64_t)1 << 31, "num_intersects must fit int32 (tile_bins are int32)");
    GG_REQUIRE(tile_bins != nullptr, "null tile_bins");
    hipStream_t s = (hipStream_t)stream;
    const int T = tiles_x * tiles_y;
    if (N == 0 || I == 0) {       // (otherwise the first kernel below zeroes the ranges: no launch of its own)
        if (gg_fill_async(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)T, s) != hipSuccess) {
            gg_set_error("gg_bin_sort: memset failed");
            return GG_ERR_LAUNCH;
        }
        return GG_OK;
    }
    GG_REQUIRE(xys && depths && radii && num_tiles_hit && gaussian_ids_sorted, "null pointer");
    BinWs w = bin_ws_layout(ws, N, I);
    if (ws == nullptr || ws_bytes < w.bytes) {
        gg_set_error("gg_bin_sort: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return GG_ERR_WORKSPACE;
    }
    gg_prof_begin(GG_K_BIN_SORT, s);
    // 1. depth order of the Gaussians, 2. their offsets in that order: by buckets (db_* above)
    const int nbk = db_buckets(N);
    const DepthSrc dsrc{depths, radii};
    const size_t walk_lds = sizeof(uint32_t) * ((size_t)nbk + 1 + 2);
    if (walk_lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)db_walk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
        (void)hipFuncSetAttribute((const void *)db_walk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
    }
    uint32_t *order = w.order;
    DbParts mm{w.dbmm, w.dbmm + 256, 256};
    if (range_lo && range_hi && range_parts > 0)      // the caller's projection left the partial minima / maxima behind
        mm = DbParts{range_lo, range_hi, range_parts};
    else
        hipLaunchKernelGGL(db_range_kernel, dim3(256), dim3(256), 0, s, N, dsrc, w.dbmm);
    const DbGeom geom{xys, tiles_x, tiles_y};
    hipLaunchKernelGGL(db_walk_kernel<false>, dim3(DB_BLOCKS), dim3(DB_THREADS), walk_lds, s, N, nbk, dsrc, geom, mm,
                       w.dbG, (const uint32_t *)nullptr, (uint4 *)nullptr, tile_bins, 2 * T);
    hipLaunchKernelGGL(db_prefix_kernel, dim3((nbk + 1 + 255) / 256), dim3(256), 0, s, nbk, w.dbG, w.dbStart);
    hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, s, nbk, w.dbStart);   // (-> [nbk] = the visible count)
    hipLaunchKernelGGL(db_walk_kernel<true>, dim3(DB_BLOCKS), dim3(DB_THREADS), walk_lds, s, N, nbk, dsrc, geom, mm,
                       w.dbG, (const uint32_t *)w.dbStart, w.recs, (int32_t *)nullptr, 0);
    hipLaunchKernelGGL(db_sort_kernel, dim3((nbk + DB_WPB - 1) / DB_WPB), dim3(64 * DB_WPB), 0, s, N, nbk,
                       (const uint32_t *)w.dbStart, (const uint4 *)w.recs, w.pairs, dsrc, geom, order, w.einfo, w.dbBase);
    hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, s, nbk, w.dbBase);
    // 3./4. emit + sort by tile id; ping-pong so the last pass lands in gaussian_ids_sorted
    int tile_bits = 1;
    while ((1 << tile_bits) < T) ++tile_bits;
    int passes = (tile_bits + 7) / 8;
    uint32_t *out_vals = (uint32_t *)gaussian_ids_sorted;
    uint32_t *kcur = w.tkeyA, *kalt = w.tkeyB;
    uint32_t *vcur = (passes % 2 == 0) ? out_vals : w.tvalTmp;
    uint32_t *valt = (passes % 2 == 0) ? w.tvalTmp : out_vals;
    // entries the emission does not reach (caller's I larger than the true total) get an
    // out-of-range tile id and id 0, so they sort to the end and are never dereferenced
    // (with a device-side count every processed entry is emitted: nothing to pre-fill)
    if (!I_dev) {
        (void)gg_fill_async(kcur, 0xFF, sizeof(uint32_t) * (size_t)I, s);
        (void)gg_fill_async(vcur, 0, sizeof(uint32_t) * (size_t)I, s);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, (const uint32_t *)order,
                       (const uint4 *)w.einfo, (const uint32_t *)w.dbBase, (const uint32_t *)(w.dbStart + nbk), tiles_x, I,
                       I_dev, kcur, vcur);
    // the tile id's bits are split evenly over the passes (13 bits: 7 + 6, not 8 + 5): a pass ranks its keys with one
    // ballot per digit bit, and the scatter kernels are bound by that ranking
    const int per_pass = (tile_bits + passes - 1) / passes;
    for (int pass = 0; pass < passes; ++pass) {
        int bits = min(per_pass, tile_bits - per_pass * pass);
        radix_pass(I, I_dev, kcur, vcur, kalt, valt, per_pass * pass, (1u << bits) - 1u, w, s);
        uint32_t *t = kcur; kcur = kalt; kalt = t;
        t = vcur; vcur = valt; valt = t;
    }
    // 5. tile ranges
    hipLaunchKernelGGL(tile_bins_kernel, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, s, I, I_dev, kcur,
                       (uint32_t)T, tile_bins, isect_tile_sorted);
    gg_prof_end(GG_K_BIN_SORT, s);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// Kept for ABI stability: up to round 3 the offsets scan waited on other workgroups inside the launch (decoupled
// look-back) and a wait that gave up was reported here.  Since round 4 no binning kernel waits on another workgroup, so
// there is no such failure to report: the call synchronises the stream and returns GG_OK.
extern "C" int gg_bin_sort_status(int N, int64_t I, const void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(N >= 0 && I >= 0, "negative size");
    (void)ws;
    (void)ws_bytes;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        gg_set_error("gg_bin_sort_status: stream synchronisation failed");
        return GG_ERR_LAUNCH;
    }
    return GG_OK;
}

extern "C" int gg_bin_sort(int N, int64_t I, const float *xys, const float *depths,
                           const int32_t *radii, const int32_t *num_tiles_hit, int tiles_x,
                           int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                           int32_t *isect_tile_sorted, void *ws, size_t ws_bytes,
                           gg_stream_t stream) {
    return bin_sort_impl(N, I, nullptr, xys, depths, radii, num_tiles_hit, tiles_x, tiles_y,
                         gaussian_ids_sorted, tile_bins, isect_tile_sorted, ws, ws_bytes, stream);
}

extern "C" int gg_bin_sort_dev(int N, int64_t capacity, const int64_t *num_intersects_dev,
                               const float *xys, const float *depths, const int32_t *radii,
                               const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                               int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                               int32_t *isect_tile_sorted, void *ws, size_t ws_bytes,
                               gg_stream_t stream) {
    GG_REQUIRE(num_intersects_dev != nullptr, "null num_intersects_dev");
    GG_REQUIRE(capacity >= 1, "capacity < 1");
    return bin_sort_impl(N, capacity, num_intersects_dev, xys, depths, radii, num_tiles_hit, tiles_x,
                         tiles_y, gaussian_ids_sorted, tile_bins, isect_tile_sorted, ws, ws_bytes, stream);
}

// gg_bin_sort_dev with the partial minima / maxima of the visible depth bits handed over (gg_view_fwd's `parts`: one
// pair per 256 Gaussians) — the depth buckets' range pass over depths and radii is not run
extern "C" int gg_bin_sort_dev_ex(int N, int64_t capacity, const int64_t *num_intersects_dev,
                                  const float *xys, const float *depths, const int32_t *radii,
                                  const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                                  int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                                  int32_t *isect_tile_sorted, void *ws, size_t ws_bytes,
                                  const uint32_t *depth_bits_min, const uint32_t *depth_bits_max, int range_parts,
                                  gg_stream_t stream) {
    GG_REQUIRE(num_intersects_dev != nullptr, "null num_intersects_dev");
    GG_REQUIRE(capacity >= 1, "capacity < 1");
    GG_REQUIRE(range_parts == 0 || (depth_bits_min && depth_bits_max && range_parts > 0), "range parts without arrays");
    return bin_sort_impl(N, capacity, num_intersects_dev, xys, depths, radii, num_tiles_hit, tiles_x,
                         tiles_y, gaussian_ids_sorted, tile_bins, isect_tile_sorted, ws, ws_bytes, stream,
                         depth_bits_min, depth_bits_max, range_parts);
}

// ---------------------------------------------------------------------------------------------
// stable sort of (key, value) u32 pairs by the low `bits` bits of the key — the radix passes above,
// exported for the deterministic backward (blend.hip: list entries grouped by Gaussian id, in list order)
// ---------------------------------------------------------------------------------------------
size_t gg_sort_pairs_workspace(int64_t n) {
    const size_t i = (size_t)(n > 0 ? n : 1);
    return gg_align_up(4 * i, 256) * 2 + gg_align_up(4 * 256 * (size_t)(radix_nblocks(n) + 1), 256) +
           gg_align_up(4 * 256, 256);
}
// keys / vals are sorted in place (ping-pong through the workspace); returns 0 on success
int gg_sort_pairs(int64_t n, uint32_t *keys, uint32_t *vals, int bits, void *ws, size_t ws_bytes, hipStream_t s) {
    if (n <= 0) return GG_OK;
    if (ws == nullptr || ws_bytes < gg_sort_pairs_workspace(n)) return GG_ERR_WORKSPACE;
    char *p = (char *)ws;
    const size_t i = (size_t)n;
    uint32_t *kalt = (uint32_t *)p;
    p += gg_align_up(4 * i, 256);
    uint32_t *valt = (uint32_t *)p;
    p += gg_align_up(4 * i, 256);
    BinWs w;
    w.G = (uint32_t *)p;
    p += gg_align_up(4 * 256 * (size_t)(radix_nblocks(n) + 1), 256);
    w.totals = (uint32_t *)p;
    uint32_t *kc = keys, *vc = vals, *ka = kalt, *va = valt;
    const int passes = (bits + 7) / 8;
    for (int pass = 0; pass < passes; ++pass) {
        const int b = min(8, bits - 8 * pass);
        radix_pass(n, nullptr, kc, vc, ka, va, 8 * pass, (1u << b) - 1u, w, s);
        uint32_t *t = kc; kc = ka; ka = t;
        t = vc; vc = va; va = t;
    }
    if (kc != keys) {   // odd number of passes: bring the result home
        if (hipMemcpyAsync(keys, kc, 4 * i, hipMemcpyDeviceToDevice, s) != hipSuccess) return GG_ERR_LAUNCH;
        if (hipMemcpyAsync(vals, vc, 4 * i, hipMemcpyDeviceToDevice, s) != hipSuccess) return GG_ERR_LAUNCH;
    }
    return GG_OK;
}
