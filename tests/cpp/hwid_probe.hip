// hwid_probe.hip — diagnostic (not a test, not part of the engine): where the waves of co-resident workgroups land.
// 1024 workgroups of 512 threads with 37 KB of LDS each (the shape of the dense search at |F| = 65536: four per CU); every wave
// records HW_REG_HW_ID and HW_REG_XCC_ID.  Prints, per CU, the workgroups it holds, their TG_ID and the SIMD of each of their waves.
//   hipcc --offload-arch=gfx950 -O2 -o tests/cpp/hwid_probe tests/cpp/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__ (512) void k (unsigned *out, float *sink)
{
    __shared__ float pad[9400];
    pad[threadIdx.x] = (float) threadIdx.x;
    __syncthreads ();
    unsigned hw, xcc;
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroups resident together for a while
    float a = pad[(threadIdx.x * 7) % 9400];
    for (int i = 0; i < 20000; ++i) a = a * 1.0000001f + 0.5f;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (a == 12345.f) sink[0] = a;
}
int main ()
{
    const int nb = 1024;
    unsigned *d; float *s;
    hipMalloc (&d, nb * 8 * 2 * 4); hipMalloc (&s, 4);
    hipLaunchKernelGGL (k, nb, 512, 0, 0, d, s);
    if (hipDeviceSynchronize () != hipSuccess) { printf ("launch failed\n"); return 1; }
    std::vector<unsigned> h (nb * 16);
    hipMemcpy (h.data (), d, nb * 16 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;      // key (xcc, se, sh, cu) -> blocks
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xF;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF);
        cu[key].push_back (b);
    }
    printf ("%zu distinct (xcc, se, sh, cu) keys for %d workgroups\n", cu.size (), nb);
    int shown = 0, clash = 0, tgclash = 0;
    for (auto &kv : cu) {
        std::vector<unsigned> tg;
        int simd0[4] = { 0, 0, 0, 0 };
        for (int b : kv.second) { tg.push_back ((h[b * 16] >> 16) & 0xF); simd0[(h[b * 16] >> 4) & 3]++; }
        std::sort (tg.begin (), tg.end ());
        for (size_t i = 1; i < tg.size (); ++i) if ((tg[i] & 3) == (tg[i - 1] & 3)) { ++tgclash; break; }
        for (int q = 0; q < 4; ++q) if (simd0[q] > 1) { ++clash; break; }
        if (shown < 6) {
            printf ("key %05x:", kv.first);
            for (int b : kv.second) {
                printf ("  wg %4d tg %2u simd of waves", b, (h[b * 16] >> 16) & 0xF);
                for (int w = 0; w < 8; ++w) printf (" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
            }
            printf ("\n"); ++shown;
        }
    }
    printf ("CUs where two workgroups have wave 0 on the same SIMD: %d of %zu; CUs where TG_ID mod 4 repeats: %d\n", clash, cu.size (), tgclash);
    return 0;
}
