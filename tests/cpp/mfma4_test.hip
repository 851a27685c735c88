// mfma4_test.hip — layout and numerics of v_mfma_f32_4x4x1_16b_f32 as used for the symmetric 4x4 squaring:
// lane (b, r) holds row r of B in 4 registers; D = sum_k B[:,k] (x) B[k,:] as an fmaf chain over k.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__ ((ext_vector_type (4)));
__global__ void k (const float *B, float *out)
{
    int l = threadIdx.x, r = l & 3;
    float row[4] = { B[r * 4 + 0], B[r * 4 + 1], B[r * 4 + 2], B[r * 4 + 3] };
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (row[0], row[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (row[1], row[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (row[2], row[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (row[3], row[3], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}
int main ()
{
    float B[16] = { 1.3f, 0.2f, -0.7f, 0.11f, 0.2f, -2.1f, 0.5f, 0.9f, -0.7f, 0.5f, 0.33f, -1.7f, 0.11f, 0.9f, -1.7f, 0.77f };
    float *dB, *dO; hipMalloc (&dB, 64); hipMalloc (&dO, 64 * 16);
    hipMemcpy (dB, B, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, dB, dO);
    std::vector<float> o (256); hipMemcpy (o.data (), dO, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            int j = l & 3;                       // expect: lane j register i = C[i][j]
            float acc = 0.f;
            for (int kk = 0; kk < 4; ++kk) acc = fmaf (B[i * 4 + kk], B[kk * 4 + j], acc);
            if (acc != o[l * 4 + i]) { if (bad < 8) printf ("lane %d reg %d: got %.9g want %.9g\n", l, i, o[l * 4 + i], acc); ++bad; }
        }
    printf ("mismatches: %d\n", bad);
    return bad != 0;
}
