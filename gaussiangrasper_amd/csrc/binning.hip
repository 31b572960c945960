// binning.hip — tile binning and depth ordering.
//
// The reference builds I int64 keys (tile_id << 32 | depth bits), sorts them globally with
// torch.sort (6+ radix passes over 12-byte pairs) and repeats that in each of its four rasterize
// calls.  The per-tile lists it obtains are ordered by (tile, depth bits, Gaussian id).  This file
// produces the SAME lists with far less HBM traffic by splitting the key:
//
//   1. stable LSD radix sort of the N Gaussians by their 32 depth bits (4 byte-wide passes over
//      N items; culled Gaussians get key 0xFFFFFFFF and sink to the end; the first pass forms the keys itself);
//   2. exclusive scan of num_tiles_hit taken in that depth order (one launch, decoupled look-back);
//   3. every Gaussian, in depth order, emits (tile id, Gaussian id) for the tiles of its bbox;
//   4. stable LSD radix sort of the I pairs by tile id only (ceil(log2 T)/8 = 2 passes).
//
// Stability of both sorts makes the result identical to the reference's global sort with ties
// broken by ascending Gaussian id (the order SURVEY a7 fixes); tests compare it bit for bit with
// the oracle's straightforward 64-bit sort.
//
// Radix pass = 3 launches: per-block digit histogram -> per-digit column scan -> stable scatter.
// Within the scatter a wave ranks its keys with ballot-based match-any (8 ballots per key), so
// equal digits keep their input order without any LDS sorting network.
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
static inline int rs_items(int64_t n) { return n >= (int64_t)4096 * 1024 ? 16 : 4; }

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
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, s, N, num_tiles_hit,
                       (unsigned long long *)out);
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
    volatile uint32_t *wh = whist[wave];
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
            bool bit = (d >> b) & 1u;
            uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        if (valid && before == 0) wh[d] = wh[d] + cnt;  // group leader (lowest lane)
        __builtin_amdgcn_wave_barrier();
        uint32_t after = valid ? wh[d] : 0u;
        rank[it] = after - cnt + before;
        __builtin_amdgcn_wave_barrier();
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
// exclusive scan of num_tiles_hit[order[r]]: block sums -> their scan -> offsets.  (r02-r03: ONE launch with a decoupled
// look-back across workgroups; a look-back that gave up left every tile range empty and only gg_bin_sort_status, which
// nothing called, said so — ADVICE r03.  Three plain launches have no wait that could give up; measured the same.)
// ---------------------------------------------------------------------------------------------
#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_TILE (SC_THREADS * SC_ITEMS)

__device__ __forceinline__ uint32_t scan_item(int N, const int32_t *__restrict__ nth, const uint32_t *__restrict__ order, int r) {
    return (r < N) ? (uint32_t)nth[min(order[r], (uint32_t)(N - 1))] : 0u;   // (order: a permutation of 0..N-1)
}
__global__ __launch_bounds__(SC_THREADS) void scan_blocksum_kernel(int N, const int32_t *__restrict__ nth,
                                                                  const uint32_t *__restrict__ order,
                                                                  uint32_t *__restrict__ bsum) {
    __shared__ unsigned int wsum[4];
    const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) acc += scan_item(N, nth, order, base + k);
    unsigned int total;
    (void)scan_block256(acc, wsum, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
// one workgroup: exclusive scan of the block sums in place
__global__ __launch_bounds__(256) void scan_bsum_kernel(int nblocks, uint32_t *__restrict__ bsum) {
    __shared__ unsigned int wsum[4];
    uint32_t carry = 0;
    for (int start = 0; start < nblocks; start += 256) {
        const int i = start + threadIdx.x;
        const uint32_t v = i < nblocks ? bsum[i] : 0u;
        unsigned int total;
        const uint32_t ex = scan_block256(v, wsum, total);
        if (i < nblocks) bsum[i] = carry + ex;
        carry += total;
    }
}
__global__ __launch_bounds__(SC_THREADS) void scan_offsets_kernel(
    int N, const int32_t *__restrict__ nth, const uint32_t *__restrict__ order,
    uint32_t *__restrict__ offsets, const uint32_t *__restrict__ bsum) {
    __shared__ unsigned int wsum[4];
    const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        v[k] = scan_item(N, nth, order, base + k);
        acc += v[k];
    }
    unsigned int total;
    uint32_t ex = scan_block256(acc, wsum, total) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int r = base + k;
        if (r < N) offsets[r] = ex;
        ex += v[k];
    }
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
__global__ __launch_bounds__(256) void emit_kernel(int N, const uint32_t *__restrict__ order,
                                                   const uint32_t *__restrict__ offsets,
                                                   const float *__restrict__ xys,
                                                   const int32_t *__restrict__ radii, int tiles_x,
                                                   int tiles_y, int64_t I,
                                                   const int64_t *__restrict__ I_dev,
                                                   uint32_t *__restrict__ tkeys,
                                                   uint32_t *__restrict__ tvals) {
    __shared__ uint32_t s_rel[4][64];
    I = dev_count(I, I_dev);   // start of each Gaussian's run, relative to the wave's first
    __shared__ int4 s_box[4][64];       // x0, y0, box width, Gaussian id
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t g = 0, cnt = 0, off = 0;
    int x0 = 0, y0 = 0, bw = 1;
    if (r < N) {
        g = min(order[r], (uint32_t)(N - 1));
        off = offsets[r];
        const int rad = radii[g];
        if (rad > 0) {
            int x1, y1;
            gg_tile_bbox(xys[2 * (size_t)g], xys[2 * (size_t)g + 1], (float)rad, tiles_x, tiles_y, x0, y0,
                         x1, y1);
            bw = x1 - x0;
            cnt = (uint32_t)(bw * (y1 - y0));
            if (bw <= 0) { bw = 1; cnt = 0; }
        }
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
    uint32_t *dkeyA, *dkeyB, *dvalA, *dvalB;  // N each
    uint32_t *offsets;                        // N
    uint32_t *block_sums;                     // scan
    uint32_t *G;                              // 256 * max nblocks
    uint32_t *totals;                         // 256
    uint32_t *tkeyA, *tkeyB, *tvalTmp;        // I each
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
    w.dkeyA = take(4 * n);
    w.dkeyB = take(4 * n);
    w.dvalA = take(4 * n);
    w.dvalB = take(4 * n);
    w.offsets = take(4 * n);
    w.block_sums = take(4 * ((n + SC_TILE - 1) / SC_TILE));   // block sums of the offsets scan
    int nb = max(radix_nblocks(N), radix_nblocks(I));
    w.G = take(4 * 256 * (size_t)(nb + 1));
    w.totals = take(4 * 256);
    w.tkeyA = take(4 * i);
    w.tkeyB = take(4 * i);
    w.tvalTmp = take(4 * i);
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
    if (rs_items(n) == 16)
        hipLaunchKernelGGL(radix_hist_kernel<16>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, shift,
                           mask, nb, w.G, dsrc);
    else
        hipLaunchKernelGGL(radix_hist_kernel<4>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, shift,
                           mask, nb, w.G, dsrc);
    hipLaunchKernelGGL(radix_colscan_kernel, dim3(256), dim3(256), 0, s, nb, w.G, w.totals);
    if (rs_items(n) == 16)
        hipLaunchKernelGGL(radix_scatter_kernel<16>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, vin,
                           kout, vout, shift, mask, nb, w.G, w.totals, dsrc);
    else
        hipLaunchKernelGGL(radix_scatter_kernel<4>, dim3(nb), dim3(RS_THREADS), 0, s, n, n_dev, kin, vin,
                           kout, vout, shift, mask, nb, w.G, w.totals, dsrc);
}

static int bin_sort_impl(int N, int64_t I, const int64_t *I_dev, const float *xys, const float *depths,
                         const int32_t *radii, const int32_t *num_tiles_hit, int tiles_x, int tiles_y,
                         int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *isect_tile_sorted,
                         void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(N >= 0 && I >= 0, "negative size");
    GG_REQUIRE(tiles_x > 0 && tiles_y > 0, "empty tile grid");
    GG_REQUIRE(I < (int64_t)1 << 31, "num_intersects must fit int32 (tile_bins are int32)");
    GG_REQUIRE(tile_bins != nullptr, "null tile_bins");
    hipStream_t s = (hipStream_t)stream;
    const int T = tiles_x * tiles_y;
    if (gg_fill_async(tile_bins, 0, sizeof(int32_t) * 2 * (size_t)T, s) != hipSuccess) {
        gg_set_error("gg_bin_sort: memset failed");
        return GG_ERR_LAUNCH;
    }
    if (N == 0 || I == 0) return GG_OK;
    GG_REQUIRE(xys && depths && radii && num_tiles_hit && gaussian_ids_sorted, "null pointer");
    BinWs w = bin_ws_layout(ws, N, I);
    if (ws == nullptr || ws_bytes < w.bytes) {
        gg_set_error("gg_bin_sort: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return GG_ERR_WORKSPACE;
    }
    gg_prof_begin(GG_K_BIN_SORT, s);
    // 1. depth order of the Gaussians
    uint32_t *ka = w.dkeyA, *kb = w.dkeyB, *va = w.dvalA, *vb = w.dvalB;
    for (int pass = 0; pass < 4; ++pass) {
        radix_pass(N, nullptr, ka, va, kb, vb, 8 * pass, 0xFFu, w, s,
                   pass == 0 ? DepthSrc{depths, radii} : DepthSrc{nullptr, nullptr});
        uint32_t *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    const uint32_t *order = va;
    // 2. offsets in depth order
    int nsb = (N + SC_TILE - 1) / SC_TILE;
    hipLaunchKernelGGL(scan_blocksum_kernel, dim3(nsb), dim3(SC_THREADS), 0, s, N, num_tiles_hit, order, w.block_sums);
    hipLaunchKernelGGL(scan_bsum_kernel, dim3(1), dim3(256), 0, s, nsb, w.block_sums);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(nsb), dim3(SC_THREADS), 0, s, N, num_tiles_hit, order,
                       w.offsets, w.block_sums);
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
    hipLaunchKernelGGL(emit_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, order, w.offsets,
                       xys, radii, tiles_x, tiles_y, I, I_dev, kcur, vcur);
    for (int pass = 0; pass < passes; ++pass) {
        int bits = min(8, tile_bits - 8 * pass);
        radix_pass(I, I_dev, kcur, vcur, kalt, valt, 8 * pass, (1u << bits) - 1u, w, s);
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
