// prim_test.hip — checks the cross-lane primitives the canonical trees rely on (run on a gfx950 box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
template <int CTRL> __device__ __forceinline__ float dpp (float v)
{ return __builtin_bit_cast (float, __builtin_amdgcn_update_dpp (0, __builtin_bit_cast (int, v), CTRL, 0xF, 0xF, true)); }

__global__ void k_shl (const float *a, float *o)
{
    int l = threadIdx.x; float v = a[l];
    o[l] = dpp<0x108> (v);           // row_shl:8
    o[64 + l] = dpp<0x101> (v);      // row_shl:1
    o[128 + l] = dpp<0x118> (v);     // row_shr:8
    auto r = __builtin_amdgcn_permlane32_swap (__builtin_bit_cast (u32, v), __builtin_bit_cast (u32, v), false, false);
    o[192 + l] = __builtin_bit_cast (float, r[0]);
    o[256 + l] = __builtin_bit_cast (float, r[1]);
    auto s = __builtin_amdgcn_permlane16_swap (__builtin_bit_cast (u32, v), __builtin_bit_cast (u32, v), false, false);
    o[320 + l] = __builtin_bit_cast (float, s[0]);
    o[384 + l] = __builtin_bit_cast (float, s[1]);
    o[448 + l] = dpp<0x55> (v);      // quad_perm [1,1,1,1]
}
int main ()
{
    std::vector<float> h (64); for (int i = 0; i < 64; ++i) h[i] = (float) i;
    float *a, *o; hipMalloc (&a, 256); hipMalloc (&o, 512 * 4);
    hipMemcpy (a, h.data (), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL (k_shl, dim3 (1), dim3 (64), 0, 0, a, o);
    std::vector<float> r (512); hipMemcpy (r.data (), o, 512 * 4, hipMemcpyDeviceToHost);
    const char *names[] = { "row_shl:8", "row_shl:1", "row_shr:8", "pl32.0", "pl32.1", "pl16.0", "pl16.1", "quad1111" };
    for (int k = 0; k < 8; ++k) { printf ("%-10s", names[k]); for (int i = 0; i < 64; ++i) printf (" %g", r[k * 64 + i]); printf ("\n"); }
    return 0;
}
