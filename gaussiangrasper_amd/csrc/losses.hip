// losses.hip — SURVEY.md §8f-2, second half: the training-time users of the feature field.
//
//   gg_mlp_bwd              backward of `fea_up = MLP(32, 512, hidden_list=[128])`
//                           (nerfstudio/models/gaussian_splatting.py:198-213), which the reference
//                           evaluates on 1000 sampled pixels per step (:905-918, `up_loss`)
//   gg_cosine_loss_fwd/bwd  `cosine_similarity_loss` (:113-118), used for the contrastive feature loss
//                           over 800 pixel pairs per mask (:909-914) and for `up_loss` (:917-918)
//
// Design point: 10^3-10^5 rows per call — a training step, not the 1.9 M-pixel render.sh pass (whose
// forward is csrc/mlp.hip).  One launch each instead of torch's ~10 GEMM / elementwise launches with
// their intermediates in HBM.  Row tiles of 8 x hidden quarters through LDS, fp32 FMA on the VALU (the matrix pipe
// would not be filled by such tiles and the whole problem is 0.3 GFLOP), weight and input gradients summed
// across tiles with float atomics (the entry point clears them first).
#include "gg_common.h"

#define MB_HID 128
#ifndef MB_ROWS
#define MB_ROWS 16     // rows per tile: 1000 rows x 4 hidden quarters = 252 workgroups, one round on 256 CUs
#endif
#define MB_JQ 32       // hidden units per workgroup: blockIdx.y = which quarter of the hidden layer
#define MB_RT (MB_ROWS / 8)   // rows per thread in the hidden-layer phase (32 units x 8 row groups)
#define MB_THREADS 256
#ifndef MB_ABL
#define MB_ABL 0   // measurement builds: 1 no v_w2, 2 also no dL/dh product, 3 also no v_x / v_w1
#endif

// r03: grid (row tiles, 4 hidden quarters).  The first form — one workgroup per 16-row tile doing everything — put 63
// workgroups of ~190 us of serial work on 256 CUs for the reference's 1000 sampled points (the call took 0.19 ms, 7 % of
// a training iteration); most of that work does not depend on the tile's row count (512 dependent w2 loads per thread,
// 256 weight-gradient elements with an atomic each).  A workgroup now owns 32 hidden units: dL/dh of ITS units over all
// outputs, v_w1 / v_w2 rows of its units, and its share of v_x (added with float atomics: four partial sums per element).
template <int IN>
__global__ __launch_bounds__(MB_THREADS) void mlp_bwd_kernel(long P, int out_dim, const float *__restrict__ x,
                                                             const float *__restrict__ w1,
                                                             const float *__restrict__ b1,
                                                             const float *__restrict__ w2,
                                                             const float *__restrict__ g,
                                                             float *__restrict__ v_x, float *__restrict__ v_w1,
                                                             float *__restrict__ v_b1, float *__restrict__ v_w2,
                                                             float *__restrict__ v_b2) {
    extern __shared__ float lds[];
    float *xs = lds;                          // [rows][IN]
    float *hs = xs + MB_ROWS * IN;            // [rows][32]   relu(h) of this quarter
    float *ghs = hs + MB_ROWS * MB_JQ;        // [rows][32]   dL/dh_pre
    float *part = ghs + MB_ROWS * MB_JQ;      // [8 output eighths][rows][32]   partial dL/dh
    float *gs = part + 8 * MB_ROWS * MB_JQ;   // [out][rows]  (transposed: 16-byte aligned rows of four)
    const int t = threadIdx.x;
    const int j0 = MB_JQ * blockIdx.y;
    const long ntiles = (P + MB_ROWS - 1) / MB_ROWS;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * MB_ROWS;
        const int rows = (int)min((long)MB_ROWS, P - row0);
        __syncthreads();
        for (int e = t; e < MB_ROWS * IN; e += MB_THREADS)
            xs[e] = (e / IN < rows) ? x[(size_t)row0 * IN + e] : 0.0f;
        for (int e = t; e < MB_ROWS * out_dim; e += MB_THREADS) {   // g tile TRANSPOSED: gs[o][r], one float4 = four rows
            const int r = e / out_dim, o = e - r * out_dim;
            gs[o * MB_ROWS + r] = (r < rows) ? g[(size_t)row0 * out_dim + e] : 0.0f;
        }
        __syncthreads();
        // v_b2[o] += sum_r g[r][o]
        if (blockIdx.y == 0) {
            for (int o = t; o < out_dim; o += MB_THREADS) {
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < MB_ROWS; ++r) s += gs[o * MB_ROWS + r];
                if (s != 0.0f) atomicAdd(v_b2 + o, s);
            }
        }
        // hidden layer: thread -> hidden unit j0 + (t & 31), rows rb .. rb + MB_RT - 1
        const int jl = t & (MB_JQ - 1), j = j0 + jl, rb = (t >> 5) * MB_RT;
        float hp[MB_RT];
        {
            const float bias = b1[j];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) hp[r] = bias;
#pragma unroll
            for (int k = 0; k < IN; ++k) {
                const float w = w1[(size_t)j * IN + k];
#pragma unroll
                for (int r = 0; r < MB_RT; ++r) hp[r] = __builtin_fmaf(w, xs[(rb + r) * IN + k], hp[r]);
            }
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) hs[(rb + r) * MB_JQ + jl] = fmaxf(hp[r], 0.0f);
        }
        // dL/dh of this quarter: thread -> (unit, one eighth of the outputs), ALL rows of the tile — every w2 value is
        // loaded once per workgroup, at most 64 + 64 loads per thread in two batches (the first form had every thread
        // walk all 512 outputs with eight loads in flight: a chain of L2 latencies, most of the call's time)
        {
            const int og = t >> 5, ob = (out_dim + 7) / 8, o_lo = og * ob, o_hi = min(out_dim, o_lo + ob);
            float gp[MB_ROWS];
#pragma unroll
            for (int r = 0; r < MB_ROWS; ++r) gp[r] = 0.0f;
            for (int o0 = o_lo; o0 < (MB_ABL >= 2 ? o_lo : o_hi); o0 += 32) {
                float wv[32];
#pragma unroll
                for (int u = 0; u < 32; ++u)                            // 32 consecutive j: one 128-byte segment
                    wv[u] = (o0 + u < o_hi) ? w2[(size_t)(o0 + u) * MB_HID + j] : 0.0f;
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int o = min(o0 + u, out_dim - 1);
#pragma unroll
                    for (int r4 = 0; r4 < MB_ROWS; r4 += 4) {
                        const float4 g4 = *reinterpret_cast<const float4 *>(gs + o * MB_ROWS + r4);
                        gp[r4] = __builtin_fmaf(g4.x, wv[u], gp[r4]);
                        gp[r4 + 1] = __builtin_fmaf(g4.y, wv[u], gp[r4 + 1]);
                        gp[r4 + 2] = __builtin_fmaf(g4.z, wv[u], gp[r4 + 2]);
                        gp[r4 + 3] = __builtin_fmaf(g4.w, wv[u], gp[r4 + 3]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < MB_ROWS; ++r) part[(og * MB_ROWS + r) * MB_JQ + jl] = gp[r];
        }
        __syncthreads();
        {
            float sb = 0.0f;
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) {
                float gh = 0.0f;
#pragma unroll
                for (int og = 0; og < 8; ++og) gh += part[(og * MB_ROWS + rb + r) * MB_JQ + jl];
                gh = hp[r] > 0.0f ? gh : 0.0f;
                ghs[(rb + r) * MB_JQ + jl] = gh;
                sb += gh;
            }
            if (sb != 0.0f) atomicAdd(v_b1 + j, sb);
        }
        __syncthreads();
        // v_x[r][k] += sum_{j in quarter} gh[r][j] w1[j][k]
        for (int e = t; e < (MB_ABL >= 3 ? 0 : rows * IN); e += MB_THREADS) {
            const int r = e / IN, k = e - r * IN;
            float s = 0.0f;
#pragma unroll
            for (int jj = 0; jj < MB_JQ; ++jj) s = __builtin_fmaf(ghs[r * MB_JQ + jj], w1[(size_t)(j0 + jj) * IN + k], s);
            if (s != 0.0f) atomicAdd(v_x + (size_t)row0 * IN + e, s);
        }
        // v_w1[j][k] += sum_r gh[r][j] x[r][k]
        for (int e = t; e < (MB_ABL >= 3 ? 0 : MB_JQ * IN); e += MB_THREADS) {
            const int jj = e / IN, k = e - jj * IN;
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < MB_ROWS; ++r) s = __builtin_fmaf(ghs[r * MB_JQ + jj], xs[r * IN + k], s);
            if (s != 0.0f) atomicAdd(v_w1 + (size_t)j0 * IN + e, s);
        }
        // v_w2[o][j] += sum_r g[r][o] relu(h)[r][j]   (consecutive threads: consecutive j -> 128-byte segments)
        float hcol[MB_ROWS];      // relu(h)[:, jj] of this thread's unit: the same for every output it takes
#pragma unroll
        for (int r = 0; r < MB_ROWS; ++r) hcol[r] = hs[r * MB_JQ + (t & (MB_JQ - 1))];
        for (int e = t; e < (MB_ABL >= 1 ? 0 : out_dim * MB_JQ); e += MB_THREADS) {
            const int o = e >> 5, jj = e & (MB_JQ - 1);
            float s = 0.0f;
#pragma unroll
            for (int r4 = 0; r4 < MB_ROWS; r4 += 4) {
                const float4 g4 = *reinterpret_cast<const float4 *>(gs + o * MB_ROWS + r4);
                s = __builtin_fmaf(g4.x, hcol[r4], s);
                s = __builtin_fmaf(g4.y, hcol[r4 + 1], s);
                s = __builtin_fmaf(g4.z, hcol[r4 + 2], s);
                s = __builtin_fmaf(g4.w, hcol[r4 + 3], s);
            }
            if (s != 0.0f) atomicAdd(v_w2 + (size_t)o * MB_HID + j0 + jj, s);
        }
    }
}

extern "C" int gg_mlp_bwd(int64_t num_rows, int in_dim, int hidden_dim, int out_dim, const float *x,
                          const float *w1, const float *b1, const float *w2, const float *g, float *v_x,
                          float *v_w1, float *v_b1, float *v_w2, float *v_b2, gg_stream_t stream) {
    GG_REQUIRE(num_rows >= 0, "num_rows < 0");
    GG_REQUIRE(hidden_dim == MB_HID, "hidden_dim must be 128 (the reference's fea_up)");
    GG_REQUIRE(in_dim == 8 || in_dim == 16 || in_dim == 32 || in_dim == 64 || in_dim == 128,
               "in_dim must be 8, 16, 32, 64 or 128");
    GG_REQUIRE(out_dim > 0 && out_dim <= 1024, "out_dim must be in 1..1024");
    GG_REQUIRE(v_w1 && v_b1 && v_w2 && v_b2, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    bool fail = hipMemsetAsync(v_w1, 0, sizeof(float) * MB_HID * in_dim, s) != hipSuccess;
    fail |= hipMemsetAsync(v_b1, 0, sizeof(float) * MB_HID, s) != hipSuccess;
    fail |= hipMemsetAsync(v_w2, 0, sizeof(float) * (size_t)out_dim * MB_HID, s) != hipSuccess;
    fail |= hipMemsetAsync(v_b2, 0, sizeof(float) * out_dim, s) != hipSuccess;
    if (num_rows > 0 && v_x) fail |= gg_fill_async(v_x, 0, sizeof(float) * (size_t)num_rows * in_dim, s) != hipSuccess;
    if (fail) {
        gg_set_error("gg_mlp_bwd: memset failed");
        return GG_ERR_LAUNCH;
    }
    if (num_rows == 0) return GG_OK;
    GG_REQUIRE(x && w1 && b1 && w2 && g && v_x, "null pointer");
    const size_t lds_bytes = sizeof(float) * (size_t)MB_ROWS * (in_dim + 10 * MB_JQ + out_dim);
    const long ntiles = (num_rows + MB_ROWS - 1) / MB_ROWS;
    const dim3 grid((unsigned)(ntiles < 2048 ? ntiles : 2048), MB_HID / MB_JQ);
    hipError_t e = hipSuccess;
#define MB_LAUNCH(IN_)                                                                                     \
    do {                                                                                                   \
        e = hipFuncSetAttribute((const void *)mlp_bwd_kernel<IN_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds_bytes);                                                           \
        if (e == hipSuccess)                                                                               \
            hipLaunchKernelGGL((mlp_bwd_kernel<IN_>), grid, dim3(MB_THREADS), lds_bytes, s,           \
                               (long)num_rows, out_dim, x, w1, b1, w2, g, v_x, v_w1, v_b1, v_w2, v_b2);     \
    } while (0)
    gg_prof_begin(GG_K_MLP_BWD, s);
    switch (in_dim) {
        case 8: MB_LAUNCH(8); break;
        case 16: MB_LAUNCH(16); break;
        case 32: MB_LAUNCH(32); break;
        case 64: MB_LAUNCH(64); break;
        default: MB_LAUNCH(128); break;
    }
    gg_prof_end(GG_K_MLP_BWD, s);
    if (e != hipSuccess) {
        gg_set_error("gg_mlp_bwd: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
        return GG_ERR_LAUNCH;
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------
// cosine-similarity loss: one wave per point, lanes stride over the channels
// ---------------------------------------------------------------------------------------------
#define CL_EPS 1e-12f    // F.normalize default eps

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// waves stride over the points and add ONE total per wave to sim_sum (a point-per-atomic form serialises on that
// address: 140 ms for the 1.7 M masked pixels of the reference's normal loss, :879)
__global__ __launch_bounds__(256) void cosine_fwd_kernel(long M, int C, const float *__restrict__ a,
                                                         const float *__restrict__ b, float *__restrict__ sim,
                                                         float *__restrict__ na, float *__restrict__ nb,
                                                         float *__restrict__ sim_sum) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    float total = 0.f;   // lane 0's
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += nwaves) {
        float dot = 0.f, sa = 0.f, sb = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
            dot = __builtin_fmaf(u, v, dot);
            sa = __builtin_fmaf(u, u, sa);
            sb = __builtin_fmaf(v, v, sb);
        }
        dot = wave_sum(dot);
        sa = wave_sum(sa);
        sb = wave_sum(sb);
        if (lane == 0) {
            const float n1 = sqrtf(sa), n2 = sqrtf(sb);
            const float s = dot / (fmaxf(n1, CL_EPS) * fmaxf(n2, CL_EPS));
            na[m] = n1;
            nb[m] = n2;
            sim[m] = s;
            total += s;
        }
    }
    if (lane == 0) atomicAdd(sim_sum, total);
}

// <= 16 channels (the reference's normal loss: 3 channels x every masked pixel): one LANE per point
template <int CMAX>
__global__ __launch_bounds__(256) void cosine_fwd_small_kernel(long M, int C, const float *__restrict__ a,
                                                               const float *__restrict__ b,
                                                               float *__restrict__ sim, float *__restrict__ na,
                                                               float *__restrict__ nb, float *__restrict__ sim_sum) {
    const long stride = (long)gridDim.x * blockDim.x;
    float total = 0.f;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
        float dot = 0.f, sa = 0.f, sb = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                const float u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
                dot = __builtin_fmaf(u, v, dot);
                sa = __builtin_fmaf(u, u, sa);
                sb = __builtin_fmaf(v, v, sb);
            }
        const float n1 = sqrtf(sa), n2 = sqrtf(sb);
        const float s = dot / (fmaxf(n1, CL_EPS) * fmaxf(n2, CL_EPS));
        na[m] = n1;
        nb[m] = n2;
        sim[m] = s;
        total += s;
    }
    total = wave_sum(total);
    if ((threadIdx.x & 63) == 0) atomicAdd(sim_sum, total);
}
template <int CMAX>
__global__ __launch_bounds__(256) void cosine_bwd_small_kernel(long M, int C, const float *__restrict__ a,
                                                               const float *__restrict__ b,
                                                               const float *__restrict__ sim,
                                                               const float *__restrict__ na,
                                                               const float *__restrict__ nb,
                                                               const float *__restrict__ v_loss,
                                                               float *__restrict__ v_a, float *__restrict__ v_b) {
    const long stride = (long)gridDim.x * blockDim.x;
    const float s = -v_loss[0] / (float)M;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
        const float n1 = na[m], n2 = nb[m], sm = sim[m];
        const float ca = fmaxf(n1, CL_EPS), cb = fmaxf(n2, CL_EPS);
        const bool fa = n1 > CL_EPS, fb = n2 > CL_EPS;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) {
                const float u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
                const float ua = u / ca, vb = v / cb;
                v_a[(size_t)m * C + c] = s * (vb - (fa ? sm * ua : 0.0f)) / ca;
                v_b[(size_t)m * C + c] = s * (ua - (fb ? sm * vb : 0.0f)) / cb;
            }
    }
}

__global__ __launch_bounds__(256) void cosine_bwd_kernel(long M, int C, const float *__restrict__ a,
                                                         const float *__restrict__ b,
                                                         const float *__restrict__ sim,
                                                         const float *__restrict__ na,
                                                         const float *__restrict__ nb,
                                                         const float *__restrict__ v_loss,
                                                         float *__restrict__ v_a, float *__restrict__ v_b) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float s = -v_loss[0] / (float)M;
    const float n1 = na[m], n2 = nb[m], sm = sim[m];
    const float ca = fmaxf(n1, CL_EPS), cb = fmaxf(n2, CL_EPS);
    const bool fa = n1 > CL_EPS, fb = n2 > CL_EPS;
    for (int c = lane; c < C; c += 64) {
        const float u = a[(size_t)m * C + c], v = b[(size_t)m * C + c];
        const float ua = u / ca, vb = v / cb;
        v_a[(size_t)m * C + c] = s * (vb - (fa ? sm * ua : 0.0f)) / ca;
        v_b[(size_t)m * C + c] = s * (ua - (fb ? sm * vb : 0.0f)) / cb;
    }
}

extern "C" int gg_cosine_loss_fwd(int64_t num_points, int channels, const float *a, const float *b,
                                  float *sim, float *norm_a, float *norm_b, float *sim_sum,
                                  gg_stream_t stream) {
    GG_REQUIRE(num_points >= 0 && channels >= 1, "bad size");
    GG_REQUIRE(sim_sum != nullptr, "null sim_sum");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sim_sum, 0, sizeof(float), s) != hipSuccess) return GG_ERR_LAUNCH;
    if (num_points == 0) return GG_OK;
    GG_REQUIRE(a && b && sim && norm_a && norm_b, "null pointer");
    if (channels <= 16) {
        const unsigned nb = (unsigned)((num_points + 255) / 256 < 4096 ? (num_points + 255) / 256 : 4096);
        hipLaunchKernelGGL(cosine_fwd_small_kernel<16>, dim3(nb), dim3(256), 0, s, (long)num_points, channels, a, b,
                           sim, norm_a, norm_b, sim_sum);
    } else {
        const unsigned nb = (unsigned)((num_points + 3) / 4 < 8192 ? (num_points + 3) / 4 : 8192);
        hipLaunchKernelGGL(cosine_fwd_kernel, dim3(nb), dim3(256), 0, s, (long)num_points, channels, a, b, sim,
                           norm_a, norm_b, sim_sum);
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int gg_cosine_loss_bwd(int64_t num_points, int channels, const float *a, const float *b,
                                  const float *sim, const float *norm_a, const float *norm_b,
                                  const float *v_loss, float *v_a, float *v_b, gg_stream_t stream) {
    GG_REQUIRE(num_points >= 0 && channels >= 1, "bad size");
    if (num_points == 0) return GG_OK;
    GG_REQUIRE(a && b && sim && norm_a && norm_b && v_loss && v_a && v_b, "null pointer");
    if (channels <= 16) {
        const unsigned nb = (unsigned)((num_points + 255) / 256 < 4096 ? (num_points + 255) / 256 : 4096);
        hipLaunchKernelGGL(cosine_bwd_small_kernel<16>, dim3(nb), dim3(256), 0, (hipStream_t)stream,
                           (long)num_points, channels, a, b, sim, norm_a, norm_b, v_loss, v_a, v_b);
    } else {
        hipLaunchKernelGGL(cosine_bwd_kernel, dim3((unsigned)((num_points + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, (long)num_points, channels, a, b, sim, norm_a, norm_b, v_loss, v_a,
                           v_b);
    }
    GG_CHECK_LAUNCH();
    return GG_OK;
}
