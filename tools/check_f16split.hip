// Check (1) the operand layout assumed for v_mfma_f32_16x16x32_f16 on gfx950 — A: lane l holds row l % 16, k = 8 (l / 16) + j;
// B: lane l holds column l % 16, k = 8 (l / 16) + j; D: lane l holds column l % 16, rows 4 (l / 16) + r — and (2) the accuracy of
// an fp32 product formed from two-piece fp16 operands (hi = RNE(x s), lo = RNE(x s - hi), rows scaled by a power of two so
// that the row maximum sits in [2^14, 2^15)) with the four piece products accumulated in fp32 by the matrix pipe, against
// the fp32 MFMA (v_mfma_f32_16x16x4_f32) and a double-precision sum.  Used by csrc/blend2.hip's 16-slot backward.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/check_f16split.hip -o tools/bin/check_f16split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &hi, unsigned &lo) {
    const h2 h = __builtin_convertvector((f32x2){x0, x1}, h2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const h2 l = __builtin_convertvector((f32x2){x0 - back[0], x1 - back[1]}, h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// power of two that brings |m| into [2^14, 2^15) (1 for m = 0 / inf / nan)
__device__ __forceinline__ float pow2_scale(float m) {
    const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
    return (e == 0 || e == 255) ? 1.0f : __builtin_bit_cast(float, (unsigned)(127 + 14 + 127 - e) << 23);
}
// A[16][32], B[32][16] row-major; out[0] = split product, out[1] = fp32 MFMA product (16 x 16 each)
__global__ void k(const float *A, const float *B, float *out_split, float *out_f32) {
    const int lane = threadIdx.x, l16 = lane & 15, q4 = lane >> 4;
    float a[8], b[8];
    float am = 0.f, bm = 0.f;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[l16 * 32 + 8 * q4 + j];
        b[j] = B[(8 * q4 + j) * 16 + l16];
        am = fmaxf(am, fabsf(a[j]));
        bm = fmaxf(bm, fabsf(b[j]));
    }
    am = fmaxf(am, __shfl_xor(am, 16, 64)); am = fmaxf(am, __shfl_xor(am, 32, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64)); bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float sa = pow2_scale(am), sb = pow2_scale(bm);
    unsigned ah[4], al[4], bh[4], bl[4];
    for (int t = 0; t < 4; ++t) {
        split2h(a[2 * t] * sa, a[2 * t + 1] * sa, ah[t], al[t]);
        split2h(b[2 * t] * sb, b[2 * t + 1] * sb, bh[t], bl[t]);
    }
    const h8 Ah = __builtin_bit_cast(h8, (u32x4){ah[0], ah[1], ah[2], ah[3]}), Al = __builtin_bit_cast(h8, (u32x4){al[0], al[1], al[2], al[3]});
    const h8 Bh = __builtin_bit_cast(h8, (u32x4){bh[0], bh[1], bh[2], bh[3]}), Bl = __builtin_bit_cast(h8, (u32x4){bl[0], bl[1], bl[2], bl[3]});
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, c, 0, 0, 0);
    // the row scale of D's rows 4 q4 + r belongs to other lanes (lane 4 q4 + r holds row 4 q4 + r as ITS A row)
    const float inv_sb = 1.0f / sb;
    for (int r = 0; r < 4; ++r) {
        const float inv_sa = 1.0f / __shfl(sa, 4 * q4 + r, 64);
        out_split[(4 * q4 + r) * 16 + l16] = (c[r] * inv_sa) * inv_sb;
    }
    f32x4 d = {0, 0, 0, 0};
    for (int t = 0; t < 8; ++t)   // k-slice t of the fp32 form: channel 8 q4 + t
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out_f32[(4 * q4 + r) * 16 + l16] = d[r];
}
static float frand() { return (float)rand() / RAND_MAX * 2.0f - 1.0f; }
int main() {
    float hA[512], hB[512], o1[256], o2[256];
    float *dA, *dB, *d1, *d2;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024);
    int bad = 0;
    for (int trial = 0; trial < 6; ++trial) {
        srand(1 + trial);
        // trial 0..1: N(0,1)-like; 2: rows of very different magnitude; 3: 2^12 spread inside a row; 4: tiny values; 5: fac-like [4e-7, 1)
        for (int i = 0; i < 512; ++i) {
            float va = frand(), vb = frand();
            if (trial == 2) { va *= ldexpf(1.0f, (i / 32) * 5 - 40); vb *= ldexpf(1.0f, (i % 16) * 4 - 30); }
            if (trial == 3) { va *= ldexpf(1.0f, -(i % 13)); vb *= ldexpf(1.0f, -(i % 11)); }
            if (trial == 4) { va *= 1e-30f; vb *= 1e-6f; }
            if (trial == 5) { va = fabsf(va) < 0.3f ? 0.0f : expf(-14.7f * fabsf(frand())); }
            hA[i] = va; hB[i] = vb;
        }
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2);
        hipMemcpy(o1, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(o2, d2, 1024, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, mag = 0;
                for (int kk = 0; kk < 32; ++kk) { ref += (double)hA[m * 32 + kk] * hB[kk * 16 + n]; mag += fabs((double)hA[m * 32 + kk] * hB[kk * 16 + n]); }
                if (mag == 0) continue;
                e1 = fmax(e1, fabs(o1[m * 16 + n] - ref) / mag);
                e2 = fmax(e2, fabs(o2[m * 16 + n] - ref) / mag);
            }
        printf("trial %d: max |err| / sum|terms|: fp16 two-piece %.3e   fp32 MFMA %.3e\n", trial, e1, e2);
        if (!(e1 < 4e-7)) bad = 1;
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad;
}
