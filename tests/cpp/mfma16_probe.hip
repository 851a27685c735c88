// mfma16_probe.hip — diagnostic: which summation order does v_mfma_f32_16x16x4_f32 use over k = 0..3?
// A[i][k] at lane i + 16k, B[k][j] at lane j + 16k, D[4*(l/16)+v][l%16] in register v of lane l.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__ ((ext_vector_type (4)));
__global__ void k (const float *A, const float *B, float *out)
{
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32 (a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[(4 * (l >> 4) + v) * 16 + (l & 15)] = acc[v];
}
int main ()
{
    std::vector<float> A (64), B (64), D (256);
    srand (7);
    int nseq = 0, nrev = 0, npair = 0, nexact = 0, nother = 0, total = 0;
    float *dA, *dB, *dO; hipMalloc (&dA, 256); hipMalloc (&dB, 256); hipMalloc (&dO, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto &x : A) x = (float) ((rand () % 20001) - 10000) / 3217.f * (1.f + (rand () % 7) * 100.f);
        for (auto &x : B) x = (float) ((rand () % 20001) - 10000) / 1931.f;
        hipMemcpy (dA, A.data (), 256, hipMemcpyHostToDevice); hipMemcpy (dB, B.data (), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, dA, dB, dO);
        hipMemcpy (D.data (), dO, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float a[4], b[4]; for (int kk = 0; kk < 4; ++kk) { a[kk] = A[i * 4 + kk]; b[kk] = B[kk * 16 + j]; }
            float seq = 0.f; for (int kk = 0; kk < 4; ++kk) seq = fmaf (a[kk], b[kk], seq);
            float rev = 0.f; for (int kk = 3; kk >= 0; --kk) rev = fmaf (a[kk], b[kk], rev);
            float pr = fmaf (a[1], b[1], a[0] * b[0]) + fmaf (a[3], b[3], a[2] * b[2]);
            double ex = (double) a[0] * b[0] + (double) a[1] * b[1] + (double) a[2] * b[2] + (double) a[3] * b[3];
            float exf = (float) ex;
            float got = D[i * 16 + j];
            ++total;
            if (got == seq) ++nseq; if (got == rev) ++nrev; if (got == pr) ++npair; if (got == exf) ++nexact;
            if (got != seq && got != rev && got != pr && got != exf) ++nother;
        }
    }
    printf ("total %d: == sequential fma chain %d, == reversed chain %d, == pairwise %d, == exact-then-round %d, none %d\n",
            total, nseq, nrev, npair, nexact, nother);
    return 0;
}
