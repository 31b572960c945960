// scan.h — single-pass exclusive prefix sum across workgroups (decoupled look-back).
//
// One launch instead of the three of a reduce / scan-of-sums / apply chain: a workgroup publishes
// its aggregate as soon as it has it and then looks back over its predecessors' published words until
// it meets one that already carries an inclusive prefix.
//
// Inter-workgroup protocol on gfx950 (8 XCDs, private non-coherent L2s, per-CU L1 never refreshed by
// other CUs' stores; MI355X_MICROARCH.md "inter-workgroup visibility", form R2 "the data IS the flag"):
// a workgroup's state is ONE naturally aligned 8-byte word {value:32 | flag:32} written by ONE relaxed
// agent-scope atomic store (an `sc1` write-through store) and read by relaxed agent-scope atomic loads
// (`sc1` loads, served by L2 / the fabric, never by the CU's L1) — no separate flag, so no fence.
// Forward progress: a workgroup takes its logical index from an atomic ticket, so every predecessor it
// waits on has already started running; spins are bounded and a stuck wait sets a status word
// instead of hanging the GPU.  That word is not left for nobody to read: the last workgroup publishes an IMPOSSIBLE
// grand total (all ones) when it is set — the callers that read the total back (gg_mask_scan, gg_compact_rows)
// turn it into an error code on the host — and consumers on the device ask scan_failed() and write nothing
// (binning.hip: no list entry is emitted and no tile range is set, so every tile renders its background;
// gg_bin_sort_status reports it).  The state block must be zeroed (hipMemsetAsync) before every launch.
#pragma once
#include "gg_common.h"

#define GG_SCAN_AGG 1u
#define GG_SCAN_PREFIX 2u
#define GG_SCAN_SPIN_LIMIT (1u << 24)

struct ScanState {
    unsigned int ticket;       // next logical workgroup index
    unsigned int error;        // != 0: a look-back gave up (never expected)
    unsigned long long total;  // grand total, written by the last workgroup
    unsigned long long pad;
    // followed by one 8-byte word per workgroup
};
static inline size_t gg_scan_state_bytes(int nblocks) {
    return gg_align_up(sizeof(ScanState) + 8 * (size_t)(nblocks > 0 ? nblocks : 1), 256);
}
__device__ __forceinline__ unsigned long long *scan_words(ScanState *st) {
    return reinterpret_cast<unsigned long long *>(st + 1);
}

// Logical index of the calling workgroup (call once, by all threads, before anything else).
__device__ __forceinline__ int scan_ticket(ScanState *st, unsigned int *s_slot) {
    if (threadIdx.x == 0) *s_slot = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    const int bid = (int)*s_slot;
    __syncthreads();
    return bid;
}

// block_total: this workgroup's aggregate (same value in every thread, or at least in thread 0..63).
// Returns the sum of the aggregates of all logical predecessors; publishes this workgroup's inclusive
// prefix; the last workgroup writes the grand total.  256-thread workgroups; contains __syncthreads().
__device__ __forceinline__ unsigned int scan_lookback(ScanState *st, int bid, int nblocks,
                                                      unsigned int block_total, unsigned int *s_excl) {
    unsigned long long *words = scan_words(st);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {   // wave 0 does the look-back
        if (bid == 0) {
            if (lane == 0) {
                __hip_atomic_store(&words[0], ((unsigned long long)block_total << 32) | GG_SCAN_PREFIX,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_excl = 0u;
            }
        } else {
            if (lane == 0)
                __hip_atomic_store(&words[bid], ((unsigned long long)block_total << 32) | GG_SCAN_AGG,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned int excl = 0u;
            int top = bid - 1;          // highest predecessor not yet accounted for
            bool failed = false;
            while (top >= 0) {
                const int idx = top - lane;
                unsigned long long w = 0ull;
                unsigned int spins = 0;
                bool have = idx < 0;    // lanes past the start have nothing to wait for
                // every lane re-reads its own word until it is published
                while (true) {
                    if (!have) {
                        w = __hip_atomic_load(&words[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        have = ((unsigned int)w) != 0u;
                    }
                    if (__all(have)) break;
                    if (++spins > GG_SCAN_SPIN_LIMIT) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (failed) break;
                const bool is_prefix = idx >= 0 && ((unsigned int)w) == GG_SCAN_PREFIX;
                const unsigned long long pm = __ballot(is_prefix);
                const int stop = pm ? __builtin_ctzll(pm) : 64;           // nearest predecessor with a prefix
                unsigned int v = (idx >= 0 && lane <= stop) ? (unsigned int)(w >> 32) : 0u;
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl += v;
                if (pm) break;
                top -= 64;
            }
            if (lane == 0) {
                if (failed) atomicExch(&st->error, 1u);
                __hip_atomic_store(&words[bid],
                                   ((unsigned long long)(excl + block_total) << 32) | GG_SCAN_PREFIX,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_excl = excl;
            }
        }
    }
    __syncthreads();
    const unsigned int excl = *s_excl;
    if (bid == nblocks - 1 && threadIdx.x == 0) {
        // (the failing workgroup raised the word before it published its prefix, and both went to L2 in that order)
        const unsigned int err = __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st->total = err ? ~0ull : (unsigned long long)excl + block_total;
    }
    return excl;
}

#define GG_SCAN_FAILED_TOTAL (~0ull)
// for kernels launched behind the scan on the same stream
__device__ __forceinline__ bool scan_failed(const ScanState *st) {
    return __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// exclusive scan of one value per thread inside a 256-thread workgroup; total -> every thread
__device__ __forceinline__ unsigned int scan_block256(unsigned int v, unsigned int *wsum /*[4]*/,
                                                      unsigned int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned int wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned int s = wsum[w];
        if (w < wave) wpre += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return wpre + incl - v;
}
