// mlp.hip — feature up-projection MLP on the matrix pipe.
//
// Replaces the reference's `MLP(32, 512, hidden_list=[128])` forward (nerfstudio/models/
// gaussian_splatting.py:198-213; `self.fea_up`, applied to EVERY pixel of the rendered 32-channel
// feature image by render.sh, nerfstudio/pipelines/base_pipeline.py:408):
//     y = W2 · relu(W1 · x + b1) + b2        x (P, IN)  ->  y (P, OUT),  hidden = 128, fp32.
// At 1600x1200 that is 1.92 M x (32·128 + 128·512) x 2 = 267 GFLOP and 3.9 GB of output per view:
// MFMA-bound in fp32 (157 TFLOP/s dense -> 1.70 ms), the only dense contraction near the path.
//
// One fused kernel, no intermediate in HBM, no LDS transpose between the layers:
//   * everything is computed TRANSPOSED — H^T = W1·X^T, Y^T = W2·H^T — with
//     v_mfma_f32_32x32x2_f32 (A[32x2]·B[2x32], exact fp32 fma chain).  The accumulator layout of
//     layer 1 (lane = pixel column, registers = hidden rows (r&3) + 8(r>>2) + 4·(lane>>5)) IS the
//     B-operand layout layer 2 wants (lane%32 = pixel, lane>>5 = which of the two k of the step) if
//     the contraction over the hidden units runs in the order the registers hold them: step
//     (blk, r) contracts hidden 32·blk + (r&3) + 8(r>>2) [lanes 0-31] and that + 4 [lanes 32-63].
//     The A operand (W2) is laid out in LDS in that order once per workgroup.  The oracle
//     (oracle/gg_oracle.c: mlp_fwd) sums in the same order, so results are bit-identical;
//   * biases initialise the accumulators (C operand), ReLU is one v_max per hidden value;
//   * persistent workgroups (one per CU, 4 waves = one per SIMD): the W2 slice of 256 output rows
//     (128 KB, the LDS of a CU holds one) is staged once per slice, then every wave streams 64-pixel
//     blocks: 64 MFMAs per 32-pixel tile for layer 1 (W1 lives in 64 VGPRs), 8 x 64 for layer 2 with
//     one conflict-free ds_read_b32 per MFMA, results stored as 16-byte pieces (32 contiguous bytes
//     per pixel and instruction).  Layer 1 is recomputed per slice (+6 % flops) instead of keeping H.
// Measured alternatives (tools/mlp_bench.py): 4 waves x 2 tiles (one wave per SIMD) 2.52 ms; H as the
// A operand of layer 2 (output columns on the lanes, full 128-byte rows per store) 2.57 ms; stores
// of the previous block pair interleaved with the MFMAs of the next (ping-pong accumulators) 3.96 ms
// (register cap at two waves per SIMD); x of the next block prefetched during layer 2: no change; this
// form 2.46 ms = 109 TFLOP/s.  Ablations: no stores
// -15 %, no x loads -6 %, no W2 staging gathers -6 %; a bare MFMA loop on this box sustains 144-153.
#include "gg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MLP_HID 128
#define MLP_SLICE_NB 8                       // 32-row output blocks per W2 slice (8 x 64 x 64 x 4 B = 128 KB)
#define MLP_STEPS (MLP_HID / 2)
#ifndef MLP_ABL
#define MLP_ABL 0   // measurement builds: 1 no stores, 2 no x loads, 3 no W2 gathers
#endif
#ifndef MLP_TPW
#define MLP_TPW 1                            // 32-pixel tiles per wave (see the kernel)
#endif              // k-steps of layer 2 (two hidden units per MFMA)

// hidden unit supplied by lane-half `half` at k-step `step` of layer 2 (= accumulator row of layer 1)
__host__ __device__ __forceinline__ int mlp_hidden_of(int step, int half) {
    const int blk = step >> 4, r = step & 15;
    return 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * half;
}

// TPW: 32-pixel tiles per wave.  A workgroup always covers 256 pixels per iteration:
// TPW = 2 -> 4 waves (one per SIMD, ~390 registers each); TPW = 1 -> 8 waves (two per SIMD, <= 256).
template <int IN, int TPW>
__global__ __launch_bounds__(256 * 2 / TPW) void mlp_fwd_kernel(long P, int out_dim,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ w1,
                                                               const float *__restrict__ b1,
                                                               const float *__restrict__ w2,
                                                               const float *__restrict__ b2,
                                                               float *__restrict__ y) {
    static_assert(IN % 4 == 0 && IN >= 4 && IN <= 64, "in_dim: multiple of 4, <= 64");
    constexpr int THREADS = 256 * 2 / TPW;
    extern __shared__ float lds[];
    float *w2s = lds;                                         // [nb][step][lane]
    float *b1s = lds + MLP_SLICE_NB * MLP_STEPS * 64;         // [128]
    float *b2s = b1s + MLP_HID;                               // [slice rows <= 256]
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // W1 as A operands of layer 1: lane holds W1[32 blk + col][2 s + half]
    float w1r[4][IN / 2];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int s = 0; s < IN / 2; ++s) w1r[blk][s] = w1[(size_t)(32 * blk + col) * IN + 2 * s + half];
    for (int i = threadIdx.x; i < MLP_HID; i += THREADS) b1s[i] = b1[i];

    const long nblocks = (P + 255) / 256;   // 256 pixels per workgroup iteration
    const int nb_total = out_dim / 32;
    for (int nb0 = 0; nb0 < nb_total; nb0 += MLP_SLICE_NB) {
        const int nbs = min(MLP_SLICE_NB, nb_total - nb0);
        __syncthreads();   // previous slice fully consumed
        // stage the slice: w2s[nb][step][lane] = W2[32 (nb0+nb) + lane%32][hidden_of(step, lane>>5)].
        // A thread takes four consecutive hidden units k = 4 q .. 4 q + 3 of one W2 row (one 16-byte
        // load): they are the rows (r & 3) = 0..3 of one (blk, r >> 2, half), i.e. four consecutive
        // k-steps for the same lane — four conflict-free LDS writes.
        for (int i = threadIdx.x; i < nbs * 32 * (MLP_HID / 4); i += THREADS) {
            const int row = i % (nbs * 32), q = i / (nbs * 32);       // lanes of a wave: consecutive rows
            const float4 v = (MLP_ABL == 3) ? make_float4(1.f, 2.f, 3.f, 4.f)
                                            : *reinterpret_cast<const float4 *>(
                                                  w2 + (size_t)(32 * nb0 + row) * MLP_HID + 4 * q);
            const int blk = q >> 3, w8 = q & 7;          // k = 32 blk + 4 w8 + (0..3)
            const int hf = w8 & 1, rhi = w8 >> 1;        // k % 32 = (r & 3) + 4 half + 8 (r >> 2)
            const int step = 16 * blk + 4 * rhi;         // + (r & 3)
            float *dst = w2s + ((size_t)(row >> 5) * MLP_STEPS + step) * 64 + (row & 31) + 32 * hf;
            dst[0] = v.x;
            dst[64] = v.y;
            dst[128] = v.z;
            dst[192] = v.w;
        }
        for (int i = threadIdx.x; i < nbs * 32; i += THREADS) b2s[i] = b2[32 * nb0 + i];
        __syncthreads();

        for (long blkid = blockIdx.x; blkid < nblocks; blkid += gridDim.x) {
            const long pix0 = blkid * 256 + wave * (32 * TPW);   // this wave's tiles
            if (pix0 >= P) continue;
            // ---------------- layer 1: H^T[hid][pixel] = W1 · X^T + b1, ReLU --------------------
            f32x16 h[TPW][4];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const long pix = pix0 + 32 * t + col;
                const bool ok = pix < P;
                float xr[IN / 2];   // X^T as B operand: x[pixel][2 s + half]
                const float4 *xp = reinterpret_cast<const float4 *>(x + (size_t)(ok ? pix : 0) * IN);
#pragma unroll
                for (int j = 0; j < IN / 4; ++j) {
                    const float4 v = (MLP_ABL == 2) ? make_float4(1.f, 2.f, (float)pix, 4.f)
                                     : ok ? xp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    xr[2 * j] = half ? v.y : v.x;
                    xr[2 * j + 1] = half ? v.w : v.z;
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    f32x16 acc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {   // rows 8 j + 4 half + 0..3 of the block
                        const float4 bv = *reinterpret_cast<const float4 *>(b1s + 32 * blk + 8 * j + 4 * half);
                        acc[4 * j] = bv.x;
                        acc[4 * j + 1] = bv.y;
                        acc[4 * j + 2] = bv.z;
                        acc[4 * j + 3] = bv.w;
                    }
#pragma unroll
                    for (int s = 0; s < IN / 2; ++s)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[blk][s], xr[s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
                    h[t][blk] = acc;
                }
            }
            // ---------------- layer 2: Y^T[out][pixel] = W2 · H^T + b2 ---------------------------
            // 2 / TPW output blocks x TPW pixel tiles = two independent accumulator chains per wave
            constexpr int NBW = 2 / TPW;
            for (int nb = 0; nb < nbs; nb += NBW) {
                f32x16 acc[NBW][TPW];
                const float *arow[NBW];
#pragma unroll
                for (int u = 0; u < NBW; ++u) {
                    const int nbu = min(nb + u, nbs - 1);   // odd block count: the last one twice
#pragma unroll
                    for (int j = 0; j < 4; ++j) {            // rows 8 j + 4 half + 0..3 of the block
                        const float4 bv = *reinterpret_cast<const float4 *>(b2s + 32 * nbu + 8 * j + 4 * half);
#pragma unroll
                        for (int t = 0; t < TPW; ++t) {
                            acc[u][t][4 * j] = bv.x;
                            acc[u][t][4 * j + 1] = bv.y;
                            acc[u][t][4 * j + 2] = bv.z;
                            acc[u][t][4 * j + 3] = bv.w;
                        }
                    }
                    arow[u] = w2s + (size_t)nbu * MLP_STEPS * 64 + lane;
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
#pragma unroll
                        for (int u = 0; u < NBW; ++u) {
                            const float a = arow[u][(16 * blk + r) * 64];
#pragma unroll
                            for (int t = 0; t < TPW; ++t)
                                acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[t][blk][r], acc[u][t], 0, 0, 0);
                        }
                    }
                // lane holds Y[pixel col][32 (nb0+nb+u) + 8 j + 4 half + 0..3]: one 16-byte store per j
#pragma unroll
                for (int u = 0; u < NBW; ++u) {
                    if (nb + u >= nbs) break;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const long pix = pix0 + 32 * t + col;
                        if (pix < P && MLP_ABL != 1) {
                            float *yp = y + (size_t)pix * out_dim + 32 * (nb0 + nb + u) + 4 * half;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *reinterpret_cast<float4 *>(yp + 8 * j) =
                                    make_float4(acc[u][t][4 * j], acc[u][t][4 * j + 1], acc[u][t][4 * j + 2],
                                                acc[u][t][4 * j + 3]);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// in_dim = 128 (BASELINE config 5: a 128-dim feature image through fea_up = MLP(128, 512, [128]),
// gaussian_splatting.py:258 with feature_dim = 128).  W1 (128 x 128) no longer fits the registers (256 VGPRs per
// lane) and W1 + a 128 KB slice of W2 no longer fit the LDS, so the loop nest is turned inside out: W1 stays in
// LDS as the A operand of layer 1 (64 KB, the layout layer 2 uses for W2: one conflict-free ds_read_b32 per MFMA),
// x is the register-resident B operand (64 VGPRs), the hidden activations of a wave's 32-pixel tile stay in
// registers (64) while the FOUR 64 KB slices of W2 pass through the other half of the LDS — re-staged from L2 per
// 256-pixel block (256 KB per block: 0.5 GB per 1080p view against 34 TB/s of L2, but two workgroup barriers per
// slice) instead of recomputing layer 1 per slice (which would cost +60 % MFMAs here).  Same contraction orders as
// the narrow kernel — inputs 0, 1, 2, ... for layer 1, mlp_hidden_of() for layer 2 — so the oracle's mlp_fwd is
// matched bit for bit as well.
// ---------------------------------------------------------------------------------------------
#define MLPW_IN 128
#define MLPW_NB 4                               // 32-row output blocks per W2 slice (4 x 64 x 64 x 4 B = 64 KB)
__global__ __launch_bounds__(512) void mlp_fwd_wide_kernel(long P, int out_dim, const float *__restrict__ x,
                                                           const float *__restrict__ w1, const float *__restrict__ b1,
                                                           const float *__restrict__ w2, const float *__restrict__ b2,
                                                           float *__restrict__ y) {
    constexpr int IN = MLPW_IN, THREADS = 512, STEPS1 = IN / 2;
    extern __shared__ float lds[];
    float *w1s = lds;                                         // [blk 4][step 64][lane 64]
    float *w2s = w1s + 4 * STEPS1 * 64;                       // [nb 4][step 64][lane 64]
    float *b1s = w2s + MLPW_NB * MLP_STEPS * 64;              // [128]
    float *b2s = b1s + MLP_HID;                               // [out_dim]
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // w1s[blk][s][lane] = W1[32 blk + lane % 32][2 s + lane / 32]: a thread takes four consecutive inputs of one W1
    // row (one 16-byte load) = k-steps s, s + 1 for both lane halves
    for (int i = threadIdx.x; i < MLP_HID * (IN / 4); i += THREADS) {
        const int row = i % MLP_HID, q = i / MLP_HID;         // lanes of a wave: consecutive rows
        const float4 v = *reinterpret_cast<const float4 *>(w1 + (size_t)row * IN + 4 * q);
        float *dst = w1s + ((size_t)(row >> 5) * STEPS1 + 2 * q) * 64 + (row & 31);
        dst[0] = v.x;         // input 4 q     : step 2 q,     half 0
        dst[32] = v.y;        // input 4 q + 1 : step 2 q,     half 1
        dst[64] = v.z;        // input 4 q + 2 : step 2 q + 1, half 0
        dst[96] = v.w;        // input 4 q + 3 : step 2 q + 1, half 1
    }
    for (int i = threadIdx.x; i < MLP_HID; i += THREADS) b1s[i] = b1[i];
    for (int i = threadIdx.x; i < out_dim; i += THREADS) b2s[i] = b2[i];
    __syncthreads();

    const long nblocks = (P + 255) / 256;
    const int nb_total = out_dim / 32;
    for (long blkid = blockIdx.x; blkid < nblocks; blkid += gridDim.x) {   // (every wave runs every barrier)
        const long pix = blkid * 256 + wave * 32 + col;
        const bool ok = pix < P;
        // ---------------- layer 1: H^T[hid][pixel] = W1 · X^T + b1, ReLU ------------------------
        f32x16 h[4];
        {
            float xr[STEPS1];   // X^T as B operand: x[pixel][2 s + half]
            const float4 *xp = reinterpret_cast<const float4 *>(x + (size_t)(ok ? pix : 0) * IN);
#pragma unroll
            for (int j = 0; j < IN / 4; ++j) {
                const float4 v = ok ? xp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                xr[2 * j] = half ? v.y : v.x;
                xr[2 * j + 1] = half ? v.w : v.z;
            }
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                f32x16 acc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 bv = *reinterpret_cast<const float4 *>(b1s + 32 * blk + 8 * j + 4 * half);
                    acc[4 * j] = bv.x;
                    acc[4 * j + 1] = bv.y;
                    acc[4 * j + 2] = bv.z;
                    acc[4 * j + 3] = bv.w;
                }
                const float *arow = w1s + (size_t)blk * STEPS1 * 64 + lane;
#pragma unroll
                for (int s = 0; s < STEPS1; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[s * 64], xr[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
                h[blk] = acc;
            }
        }
        // ---------------- layer 2, slice by slice ------------------------------------------------
        for (int nb0 = 0; nb0 < nb_total; nb0 += MLPW_NB) {
            const int nbs = min(MLPW_NB, nb_total - nb0);
            __syncthreads();   // the previous slice is fully consumed
            for (int i = threadIdx.x; i < nbs * 32 * (MLP_HID / 4); i += THREADS) {
                const int row = i % (nbs * 32), q = i / (nbs * 32);
                const float4 v = *reinterpret_cast<const float4 *>(w2 + (size_t)(32 * nb0 + row) * MLP_HID + 4 * q);
                const int blk = q >> 3, w8 = q & 7;
                const int hf = w8 & 1, rhi = w8 >> 1;
                const int step = 16 * blk + 4 * rhi;
                float *dst = w2s + ((size_t)(row >> 5) * MLP_STEPS + step) * 64 + (row & 31) + 32 * hf;
                dst[0] = v.x;
                dst[64] = v.y;
                dst[128] = v.z;
                dst[192] = v.w;
            }
            __syncthreads();
            for (int nb = 0; nb < nbs; nb += 2) {   // two output blocks at a time: two independent accumulator chains
                f32x16 acc[2];
                const float *arow[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int nbu = min(nb + u, nbs - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 bv = *reinterpret_cast<const float4 *>(b2s + 32 * (nb0 + nbu) + 8 * j + 4 * half);
                        acc[u][4 * j] = bv.x;
                        acc[u][4 * j + 1] = bv.y;
                        acc[u][4 * j + 2] = bv.z;
                        acc[u][4 * j + 3] = bv.w;
                    }
                    arow[u] = w2s + (size_t)nbu * MLP_STEPS * 64 + lane;
                }
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[u][(16 * blk + r) * 64], h[blk][r], acc[u], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (nb + u >= nbs) break;
                    if (ok) {
                        float *yp = y + (size_t)pix * out_dim + 32 * (nb0 + nb + u) + 4 * half;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<float4 *>(yp + 8 * j) =
                                make_float4(acc[u][4 * j], acc[u][4 * j + 1], acc[u][4 * j + 2], acc[u][4 * j + 3]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast forward (round 3): both layers as fp32-grade products of fp16 two-piece operands on
// v_mfma_f32_16x16x32_f16 (gg_common.h: four 16-cycle MFMAs per 16 x 16 x 32 block where the fp32 instruction takes
// eight of 32 cycles) — a quarter of the matrix cycles of the kernels above, at the same or better accuracy against
// a double-precision sum (tools/check_f16split.hip), but NOT the oracle's summation order: results agree with
// oracle/gg_oracle.c:mlp_fwd to ~1e-6 of the largest output instead of bit for bit (gg_mlp_fwd stays the exact-order
// call).  The reference's side is cuBLAS behind nn.Linear — no summation order to match there.
//
//   * gg_mlp_pack (once per weight set): every row of W1 and W2 gets a power-of-two scale (row maximum into
//     [2^14, 2^15)) and is written as (hi, lo) fp16 pieces in the A-operand order of the MFMA — per (16-row tile,
//     k-step of 32, piece) 64 lanes x 16 bytes, lane = (row % 16) + 16 (k-block) — so that staging a 128-row
//     slice into LDS is a 64 KB copy and every A operand one conflict-free ds_read_b128.  W2's k order is the order
//     layer 1's accumulators hold the hidden units in (below); 1 / scale per row goes beside them.
//   * transposed chain as above: H^T = W1 X^T, Y^T = W2 H^T, N = 16 pixels per MFMA.  Lane (p = lane % 16,
//     q = lane / 16) loads x[pixel p][32 ks + 8 q .. + 7] (two float4), scales by the pixel's power of two (row
//     maximum over the lane's values and the three other q lanes: two permlane swaps) and splits: the B operand of
//     layer 1.  Layer 1's accumulators — lane (p, q) holds hidden units 16 t + 4 q + r of pixel p — become layer 2's
//     B operand WITHOUT moving: k-step ks of layer 2 contracts the hidden units of tiles 2 ks and 2 ks + 1, slot j of
//     k-block q is hidden unit 16 (2 ks + j / 4) + 4 q + j % 4 (mlpf_hidden_of); W2 is packed in that order.
//     Bias, un-scaling and ReLU are one multiply, one fma and one max per hidden value; the pixel's second scale
//     comes from the ReLU outputs the same way as the first.
//   * persistent workgroups of 8 waves (two per SIMD), 256 pixels per iteration, 32 per wave as two 16-pixel blocks
//     (each A operand read feeds 8 MFMAs); two tiles x two blocks = four independent accumulator chains.  The five
//     weight slices of an iteration (W1, four of W2 at out = 512) go through a two-deep ring of 64 KB LDS buffers:
//     the next slice is requested from L2 into registers before a slice's MFMAs and written to the other buffer
//     after them — one barrier per slice.
// ---------------------------------------------------------------------------------------------
#define MLPF_THREADS 512
#ifndef MLPF_NPB
#define MLPF_NPB 4   // 16-pixel blocks per wave and iteration (2: measured below)
#endif
#ifndef MLPF_ABL
#define MLPF_ABL 0   // measurement builds: 1 no output stores, 2 no layer-2 MFMAs, 3 no stores and layer 2's A operands read once
#endif
#define GG_MLP_FAST_MAX_OUT 3968                 // (160 KB - 2 x 64 KB slices) / 4 B = 8192 floats = 2 x 128 + 2 x out_dim
#define MLPF_SLICE_Q 4096                        // uint4 per 64 KB slice: 8 tiles x 4 k-steps x 2 pieces x 64 lanes
__host__ __device__ __forceinline__ int mlpf_hidden_of(int ks, int q, int j) { return 16 * (2 * ks + (j >> 2)) + 4 * q + (j & 3); }

// one wave per weight row: rows [0, 128) are W1's, [128, 128 + out) W2's
template <int IN>
__global__ __launch_bounds__(64) void mlpf_pack_kernel(int out_dim, const float *__restrict__ w1,
                                                       const float *__restrict__ w2, uint4 *__restrict__ packed,
                                                       float *__restrict__ inv_s) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const bool first = row < MLP_HID;
    const int K = first ? IN : MLP_HID, KS = K / 32;
    const float *src = first ? w1 + (size_t)row * IN : w2 + (size_t)(row - MLP_HID) * MLP_HID;
    float m = 0.0f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[k]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float sc = pow2_scale(m);
    if (lane == 0) inv_s[row] = pow2_inv(sc);
    // lane = (ks, q): 8 values of this row
    const int ks = lane >> 2, q = lane & 3;
    if (ks >= KS) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[first ? 32 * ks + 8 * q + j : mlpf_hidden_of(ks, q, j)] * sc;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) split2h(v[2 * t], v[2 * t + 1], hi[t], lo[t]);
    const int r2 = first ? row : row - MLP_HID;
    const int slice = r2 >> 7, tile = (r2 & 127) >> 4, l16 = r2 & 15;
    // W1's slice has KS1 k-steps per tile, W2's four
    uint4 *base = packed + (first ? (size_t)0 : (size_t)8 * (IN / 32) * 2 * 64 + (size_t)slice * MLPF_SLICE_Q);
    uint4 *dst = base + ((size_t)(tile * KS + ks) * 2) * 64 + 16 * q + l16;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__device__ __forceinline__ float mlpf_max_over_q(float m) {   // maximum over the four lanes p, p + 16, p + 32, p + 48
    auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    m = fmaxf(__builtin_bit_cast(float, (unsigned)r16[0]), __builtin_bit_cast(float, (unsigned)r16[1]));
    auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r32[0]), __builtin_bit_cast(float, (unsigned)r32[1]));
}

template <int IN>
__global__ __launch_bounds__(MLPF_THREADS) void mlp_fwd_f16_kernel(long P, int out_dim, const float *__restrict__ x,
                                                                   const uint4 *__restrict__ packed,
                                                                   const float *__restrict__ inv_s,
                                                                   const float *__restrict__ b1,
                                                                   const float *__restrict__ b2, float *__restrict__ y) {
    constexpr int KS1 = IN / 32;
    constexpr int W1_Q = 8 * KS1 * 2 * 64;                      // uint4 of the W1 slice
    constexpr int PF = MLPF_SLICE_Q / MLPF_THREADS;             // uint4 a thread carries of a slice in flight (8)
    extern __shared__ uint4 ldsq[];
    uint4 *buf0 = ldsq, *buf1 = ldsq + MLPF_SLICE_Q;
    float *tab = reinterpret_cast<float *>(ldsq + 2 * MLPF_SLICE_Q);
    float *b1s = tab, *i1s = tab + MLP_HID, *b2s = tab + 2 * MLP_HID, *i2s = b2s + out_dim;
    const int lane = threadIdx.x & 63, l16 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsl = (out_dim + 127) >> 7;                       // slices of W2
    const uint4 *w2p = packed + W1_Q;
    for (int i = threadIdx.x; i < MLP_HID; i += MLPF_THREADS) { b1s[i] = b1[i]; i1s[i] = inv_s[i]; }
    for (int i = threadIdx.x; i < out_dim; i += MLPF_THREADS) { b2s[i] = b2[i]; i2s[i] = inv_s[MLP_HID + i]; }
    for (int i = threadIdx.x; i < W1_Q; i += MLPF_THREADS) buf0[i] = packed[i];
    __syncthreads();
    int g = 0;   // stages done: the current stage's weights are in buffer g & 1
    // request a slice (W2's slice sl, or W1 for sl < 0): global_load_lds_dwordx4 copies it straight into the buffer the
    // NEXT stage reads — 1 KB per wave and instruction (lane l: 16 bytes at base + 16 l), no registers in between (with
    // the slice held in 32 registers per thread across a stage's MFMAs the four-block build spilled 26-57 of them).
    // The other buffer is free: every wave passed the barrier that ended the stage which read it.
    auto request = [&](int sl) {
        const uint4 *src = sl < 0 ? packed : w2p + (size_t)sl * MLPF_SLICE_Q;
        const int n = sl < 0 ? W1_Q : min(MLPF_SLICE_Q, (out_dim - 128 * sl) / 16 * 4 * 2 * 64);
        uint4 *dst = (g & 1) ? buf0 : buf1;
        const int wbase = wave * 64;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i0 = wbase + u * MLPF_THREADS;          // first uint4 of this wave's 1 KB (wave-uniform)
            if (i0 < n)
                __builtin_amdgcn_global_load_lds(src + i0 + lane, (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
        }
    };
    auto deliver = [&]() {
        __builtin_amdgcn_s_waitcnt(0);     // (vmcnt, lgkmcnt, expcnt = 0: the copies have landed)
        if (MLPF_ABL != 4) __syncthreads();
        ++g;
    };
    // NPB pixel blocks of 16 per wave and iteration: every A operand read (1 KB per piece and wave) feeds 2 NPB MFMAs.
    // With two blocks the LDS ran at its 128 B/clk: 0.91 ms of compute against 0.60 with the A operands read once
    // (output stores off, 32 -> 128 -> 512 at 1600x1200); four blocks halve that traffic.  Layer 1 runs on two blocks
    // at a time (its B operands: 64 registers at 128 inputs).
    constexpr int NPB = MLPF_NPB;
    constexpr int PIX_PER_WG = (MLPF_THREADS / 64) * 16 * NPB;      // 512
    const long nblocks = (P + PIX_PER_WG - 1) / PIX_PER_WG;
    for (long blkid = blockIdx.x; blkid < nblocks; blkid += gridDim.x) {   // (every wave runs every barrier)
        long pix[NPB];
        bool ok[NPB];
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
            pix[b] = blkid * PIX_PER_WG + wave * (16 * NPB) + 16 * b + l16;
            ok[b] = pix[b] < P;
        }
        unsigned hh[NPB][4][4], hl[NPB][4][4];
        float inv_sh[NPB];
#pragma unroll
        for (int half = 0; half < NPB / 2; ++half) {
            if (half == NPB / 2 - 1) request(0);       // (in flight during the last pair's layer 1)
            // ---------------- x: scale per pixel, two fp16 pieces ---------------------------------------
            unsigned xh[2][KS1][4], xl[2][KS1][4];
            float inv_sx[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pb = 2 * half + b;
                float xv[KS1][8];
                const float *xp = x + (size_t)(ok[pb] ? pix[pb] : 0) * IN + 8 * q4;
                float m = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp + 32 * ks), v1 = *reinterpret_cast<const float4 *>(xp + 32 * ks + 4);
                    const float t[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xv[ks][j] = ok[pb] ? t[j] : 0.0f;
                        m = fmaxf(m, fabsf(xv[ks][j]));
                    }
                }
                const float sx = pow2_scale(mlpf_max_over_q(m));
                inv_sx[b] = pow2_inv(sx);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t) split2h(xv[ks][2 * t] * sx, xv[ks][2 * t + 1] * sx, xh[b][ks][t], xl[b][ks][t]);
            }
            // ---------------- layer 1: H^T = relu(W1 X^T + b1) -------------------------------------------
            float h[2][32];
            {
                const uint4 *wb = (g & 1) ? buf1 : buf0;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    f32x4 acc[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[u][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        h16x8 Ah[2], Al[2], Bh[2], Bl[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const uint4 ah = wb[(((2 * tp + u) * KS1 + ks) * 2) * 64 + lane], al = wb[(((2 * tp + u) * KS1 + ks) * 2 + 1) * 64 + lane];
                            Ah[u] = H8(ah.x, ah.y, ah.z, ah.w);
                            Al[u] = H8(al.x, al.y, al.z, al.w);
                            Bh[u] = H8(xh[u][ks][0], xh[u][ks][1], xh[u][ks][2], xh[u][ks][3]);      // (u doubles as the pixel block)
                            Bl[u] = H8(xl[u][ks][0], xl[u][ks][1], xl[u][ks][2], xl[u][ks][3]);
                        }
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
#pragma unroll
                                for (int b = 0; b < 2; ++b)
                                    acc[u][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16((pr >> 1) ? Ah[u] : Al[u], (pr & 1) ? Bh[b] : Bl[b],
                                                                                       acc[u][b], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int m0 = 16 * (2 * tp + u) + 4 * q4;       // hidden units m0 .. m0 + 3
                        const float4 bi = *reinterpret_cast<const float4 *>(b1s + m0), iv = *reinterpret_cast<const float4 *>(i1s + m0);
                        const float bb[4] = {bi.x, bi.y, bi.z, bi.w}, ii[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                h[b][4 * (2 * tp + u) + r] = fmaxf(__builtin_fmaf(acc[u][b][r], ii[r] * inv_sx[b], bb[r]), 0.0f);
                    }
                }
            }
            // ---------------- hidden: second scale per pixel, two fp16 pieces ----------------------------
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int pb = 2 * half + b;
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; ++i) m = fmaxf(m, h[b][i]);
                const float sh = pow2_scale(mlpf_max_over_q(m));
                inv_sh[pb] = pow2_inv(sh);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        split2h(h[b][8 * ks + 2 * t] * sh, h[b][8 * ks + 2 * t + 1] * sh, hh[pb][ks][t], hl[pb][ks][t]);
            }
        }
        deliver();
        // ---------------- layer 2, slice by slice ------------------------------------------------------
        for (int sl = 0; sl < nsl; ++sl) {
            request(sl + 1 < nsl ? sl + 1 : -1);       // (after the last slice: W1 for the next iteration)
            const uint4 *wb = (g & 1) ? buf1 : buf0;
            const int nt = min(8, (out_dim - 128 * sl) >> 4);
            for (int tp = 0; 2 * tp < nt; ++tp) {
                f32x4 acc[2][NPB];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int b = 0; b < NPB; ++b) acc[u][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    h16x8 Ah[2], Al[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int tile = min(2 * tp + u, nt - 1);
                        const int aidx = MLPF_ABL == 3 ? 0 : ((tile * 4 + ks) * 2) * 64;     // (3: the A operands read once)
                        const uint4 ah = wb[aidx + lane], al = wb[aidx + 64 + lane];
                        Ah[u] = H8(ah.x, ah.y, ah.z, ah.w);
                        Al[u] = H8(al.x, al.y, al.z, al.w);
                    }
                    if (MLPF_ABL == 2) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int b = 0; b < NPB; ++b) acc[u][b][0] += (float)Ah[u][0] + (float)Al[u][1] + __builtin_bit_cast(float, hh[b][ks][0]);
                        continue;
                    }
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int b = 0; b < NPB; ++b) {
                                const h16x8 Bp = (pr & 1) ? H8(hh[b][ks][0], hh[b][ks][1], hh[b][ks][2], hh[b][ks][3])
                                                          : H8(hl[b][ks][0], hl[b][ks][1], hl[b][ks][2], hl[b][ks][3]);
                                acc[u][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16((pr >> 1) ? Ah[u] : Al[u], Bp, acc[u][b], 0, 0, 0);
                            }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (2 * tp + u >= nt) break;
                    const int m0 = 128 * sl + 16 * (2 * tp + u) + 4 * q4;   // outputs m0 .. m0 + 3
                    const float4 bi = *reinterpret_cast<const float4 *>(b2s + m0), iv = *reinterpret_cast<const float4 *>(i2s + m0);
#pragma unroll
                    for (int b = 0; b < NPB; ++b) {
                        if (!ok[b]) continue;
                        float4 o;
                        o.x = __builtin_fmaf(acc[u][b][0], iv.x * inv_sh[b], bi.x);
                        o.y = __builtin_fmaf(acc[u][b][1], iv.y * inv_sh[b], bi.y);
                        o.z = __builtin_fmaf(acc[u][b][2], iv.z * inv_sh[b], bi.z);
                        o.w = __builtin_fmaf(acc[u][b][3], iv.w * inv_sh[b], bi.w);
                        if (MLPF_ABL == 1 || MLPF_ABL == 3 || MLPF_ABL == 4) { asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w)); continue; }
                        *reinterpret_cast<float4 *>(y + (size_t)pix[b] * out_dim + m0) = o;
                    }
                }
            }
            deliver();
        }
    }
}

extern "C" size_t gg_mlp_fwd_fast_workspace(int in_dim, int hidden_dim, int out_dim) {
    if (hidden_dim != MLP_HID || in_dim <= 0 || in_dim % 32 || out_dim <= 0 || out_dim % 16 || out_dim > GG_MLP_FAST_MAX_OUT) return 0;
    const size_t nsl = ((size_t)out_dim + 127) / 128;
    return sizeof(uint4) * ((size_t)8 * (in_dim / 32) * 2 * 64 + nsl * MLPF_SLICE_Q) + sizeof(float) * (size_t)(MLP_HID + out_dim);
}

extern "C" int gg_mlp_fwd_fast(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
                               const float *w1, const float *b1, const float *w2, const float *b2, float *y,
                               void *ws, size_t ws_bytes, gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0, "num_rows < 0");
    GG_REQUIRE(hidden_dim == MLP_HID, "hidden_dim must be 128 (the reference's fea_up)");
    GG_REQUIRE(in_dim == 32 || in_dim == 64 || in_dim == 128, "in_dim must be 32, 64 or 128 (gg_mlp_fwd takes 8 and 16)");
    // two 64 KB slice buffers + [2 x 128 + 2 x out_dim] floats of biases must fit the CU's 160 KB of LDS
    GG_REQUIRE(out_dim > 0 && out_dim % 16 == 0 && out_dim <= GG_MLP_FAST_MAX_OUT,
               "out_dim must be a multiple of 16, at most 3968 (two weight slices and the biases share 160 KB of LDS)");
    if (num_rows == 0) return GG_OK;
    GG_REQUIRE(x && w1 && b1 && w2 && b2 && y, "null pointer");
    GG_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "x and y must be 16-byte aligned");
    if (ws == nullptr || ws_bytes < gg_mlp_fwd_fast_workspace(in_dim, hidden_dim, out_dim) || ((uintptr_t)ws & 15)) {
        gg_set_error("gg_mlp_fwd_fast: workspace of gg_mlp_fwd_fast_workspace() bytes, 16-byte aligned, expected");
        return GG_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t nsl = ((size_t)out_dim + 127) / 128;
    uint4 *packed = reinterpret_cast<uint4 *>(ws);
    float *inv_s = reinterpret_cast<float *>(packed + (size_t)8 * (in_dim / 32) * 2 * 64 + nsl * MLPF_SLICE_Q);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long nblocks = (num_rows + 128 * MLPF_NPB - 1) / (128 * MLPF_NPB);      // pixels per workgroup iteration (mlp_fwd_f16_kernel: NPB)
    const int grid = (int)(nblocks < cus ? nblocks : cus);
    const size_t lds_bytes = sizeof(uint4) * 2 * MLPF_SLICE_Q + sizeof(float) * (size_t)(2 * MLP_HID + 2 * out_dim);
    hipError_t e = hipSuccess;
#define MLPF_LAUNCH(IN_)                                                                                              \
    do {                                                                                                             \
        hipLaunchKernelGGL((mlpf_pack_kernel<IN_>), dim3(MLP_HID + out_dim), dim3(64), 0, s, out_dim, w1, w2, packed, \
                           inv_s);                                                                                   \
        e = hipFuncSetAttribute((const void *)mlp_fwd_f16_kernel<IN_>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)lds_bytes);                                                                     \
        if (e == hipSuccess)                                                                                         \
            hipLaunchKernelGGL((mlp_fwd_f16_kernel<IN_>), dim3(grid), dim3(MLPF_THREADS), lds_bytes, s,              \
                               (long)num_rows, out_dim, x, packed, inv_s, b1, b2, y);                                \
    } while (0)
    gg_prof_begin(GG_K_MLP_FWD, s);
    if (in_dim == 32) MLPF_LAUNCH(32);
    else if (in_dim == 64) MLPF_LAUNCH(64);
    else MLPF_LAUNCH(128);
    gg_prof_end(GG_K_MLP_FWD, s);
    if (e != hipSuccess) {
        gg_set_error("gg_mlp_fwd_fast: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return GG_ERR_LAUNCH;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_mlp_fwd(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
                          const float *w1, const float *b1, const float *w2, const float *b2, float *y,
                          gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0, "num_rows < 0");
    GG_REQUIRE(hidden_dim == MLP_HID, "hidden_dim must be 128 (the reference's fea_up)");
    GG_REQUIRE(in_dim == 8 || in_dim == 16 || in_dim == 32 || in_dim == 64 || in_dim == 128,
               "in_dim must be 8, 16, 32, 64 or 128");
    GG_REQUIRE(in_dim != 128 || out_dim <= 4096, "in_dim 128: out_dim <= 4096 (the output biases live in LDS)");
    GG_REQUIRE(out_dim > 0 && out_dim % 32 == 0, "out_dim must be a positive multiple of 32");
    if (num_rows == 0) return GG_OK;
    GG_REQUIRE(x && w1 && b1 && w2 && b2 && y, "null pointer");
    GG_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "x and y must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_bytes = sizeof(float) * ((size_t)MLP_SLICE_NB * MLP_STEPS * 64 + MLP_HID + 32 * MLP_SLICE_NB);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long nblocks = (num_rows + 255) / 256;
    const int grid = (int)(nblocks < cus ? nblocks : cus);
    hipError_t e = hipSuccess;
    if (in_dim == MLPW_IN) {
        const size_t wide_lds = sizeof(float) * ((size_t)4 * (MLPW_IN / 2) * 64 + (size_t)MLPW_NB * MLP_STEPS * 64 +
                                                 MLP_HID + (size_t)out_dim);
        gg_prof_begin(GG_K_MLP_FWD, s);
        e = hipFuncSetAttribute((const void *)mlp_fwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)wide_lds);
        if (e == hipSuccess)
            hipLaunchKernelGGL(mlp_fwd_wide_kernel, dim3(grid), dim3(512), wide_lds, s, (long)num_rows, out_dim, x, w1,
                               b1, w2, b2, y);
        gg_prof_end(GG_K_MLP_FWD, s);
        if (e != hipSuccess) {
            gg_set_error("gg_mlp_fwd: cannot reserve %zu bytes of LDS: %s", wide_lds, hipGetErrorString(e));
            return GG_ERR_LAUNCH;
        }
        GG_CHECK_LAUNCH();
        return GG_OK;
    }
#define MLP_LAUNCH(IN_)                                                                                   \
    do {                                                                                                  \
        e = hipFuncSetAttribute((const void *)mlp_fwd_kernel<IN_, MLP_TPW>,                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);              \
        if (e == hipSuccess)                                                                              \
            hipLaunchKernelGGL((mlp_fwd_kernel<IN_, MLP_TPW>), dim3(grid), dim3(256 * 2 / MLP_TPW),       \
                               lds_bytes, s, (long)num_rows, out_dim, x, w1, b1, w2, b2, y);              \
    } while (0)
    gg_prof_begin(GG_K_MLP_FWD, s);
    if (in_dim == 8) MLP_LAUNCH(8);
    else if (in_dim == 16) MLP_LAUNCH(16);
    else if (in_dim == 32) MLP_LAUNCH(32);
    else MLP_LAUNCH(64);
    gg_prof_end(GG_K_MLP_FWD, s);
    if (e != hipSuccess) {
        gg_set_error("gg_mlp_fwd: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return GG_ERR_LAUNCH;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}
