// launch_probe.hip — what a dependent chain of kernel launches costs per launch as PLAIN launches and as nodes of ONE hipGraph, for a kernel
// with 16 bytes of arguments and for one with a 480-byte struct (the size of icp_params): is the 0.4 us a host-driven checked launch pays over a
// graph node a matter of the argument block, or of plain launches as such?   hipcc --offload-arch=gfx950 -O2 -o launch_probe launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct big { unsigned v[120]; };
__global__ void k_small (unsigned *p, unsigned x) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += x; }
__global__ void k_big (unsigned *p, big b) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.v[7] + b.v[119]; }
// the same with a grid that fills the chip and touches memory first (the shape of the chained search's prologue: every block loads, then leaves)
// (and stays for ~6 us, so that the host — 2.5 us per plain launch call — is ahead of the device as it is in a checked run: what is measured then
// is the device-side cost of a launch, not the host's enqueue rate)
static __device__ void stay (unsigned long long cycles) { const unsigned long long t0 = wall_clock64 (); while (wall_clock64 () - t0 < cycles) __builtin_amdgcn_s_sleep (2); }
__global__ void k_small_wide (unsigned *p, unsigned x) { unsigned v = p[(blockIdx.x * 64 + (threadIdx.x & 63)) & 1023]; stay (600); if (v == 0xFFFFFFFFu) p[1] = x; }
__global__ void k_big_wide (unsigned *p, big b) { unsigned v = p[(blockIdx.x * 64 + (threadIdx.x & 63)) & 1023]; stay (600); if (v == 0xFFFFFFFFu) p[1] = b.v[119]; }
// the same again with kernels that NEED their arguments for the address of their first load (as the chained search does): by value (a plain
// launch's argument block sits at a fresh address every time) against a pointer to a block that stays where it is in device memory
__global__ void k_val_wide (unsigned *p, big b) { const unsigned i0 = b.v[119] + b.v[60] + b.v[20]; unsigned v = p[(i0 + blockIdx.x * 64 + (threadIdx.x & 63)) & 1023]; stay (600); if (v == 0xFFFFFFFFu) p[1] = i0; }
__global__ void k_ptr_wide (unsigned *p, const big *bp) { const unsigned i0 = bp->v[119] + bp->v[60] + bp->v[20]; unsigned v = p[(i0 + blockIdx.x * 64 + (threadIdx.x & 63)) & 1023]; stay (600); if (v == 0xFFFFFFFFu) p[1] = i0; }
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e_)); return 1; } } while (0)
template <typename F> static double timed (hipStream_t s, int reps, F &&f)
{
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        (void) hipStreamSynchronize (s);
        auto t0 = std::chrono::steady_clock::now ();
        f ();
        (void) hipStreamSynchronize (s);
        best = std::min (best, std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t0).count ());
    }
    return best;
}
int main ()
{
    const int N = 400;
    unsigned *d; CHK (hipMalloc (&d, 4096)); CHK (hipMemset (d, 0, 4096));
    hipStream_t s; CHK (hipStreamCreateWithFlags (&s, hipStreamNonBlocking));
    big b {}; b.v[7] = 1;
    for (int wide = 0; wide < 2; ++wide) {
        const dim3 g (wide ? 256 : 1), t (wide ? 1024 : 64);
        auto small = [&] { for (int i = 0; i < N; ++i) { if (wide) hipLaunchKernelGGL (k_small_wide, g, t, 0, s, d, 1u); else hipLaunchKernelGGL (k_small, g, t, 0, s, d, 1u); } };
        auto bigl = [&] { for (int i = 0; i < N; ++i) { if (wide) hipLaunchKernelGGL (k_big_wide, g, t, 0, s, d, b); else hipLaunchKernelGGL (k_big, g, t, 0, s, d, b); } };
        hipGraph_t gs, gb; hipGraphExec_t es, eb;
        CHK (hipStreamBeginCapture (s, hipStreamCaptureModeThreadLocal)); small (); CHK (hipStreamEndCapture (s, &gs)); CHK (hipGraphInstantiate (&es, gs, nullptr, nullptr, 0));
        CHK (hipStreamBeginCapture (s, hipStreamCaptureModeThreadLocal)); bigl (); CHK (hipStreamEndCapture (s, &gb)); CHK (hipGraphInstantiate (&eb, gb, nullptr, nullptr, 0));
        small (); bigl (); (void) hipGraphLaunch (es, s); (void) hipGraphLaunch (eb, s); CHK (hipStreamSynchronize (s));
        const double ps = timed (s, 7, small), pb = timed (s, 7, bigl);
        const double gsu = timed (s, 7, [&] { (void) hipGraphLaunch (es, s); }), gbu = timed (s, 7, [&] { (void) hipGraphLaunch (eb, s); });
        printf ("%s grid, chain of %d dependent launches, us per launch: plain 16 B args %.3f, plain 480 B args %.3f, graph 16 B %.3f, graph 480 B %.3f\n",
                wide ? "256 x 1024" : "1 x 64", N, ps / N, pb / N, gsu / N, gbu / N);
    }
    {
        big z {}; big *dz; CHK (hipMalloc (&dz, sizeof (big))); CHK (hipMemcpy (dz, &z, sizeof (big), hipMemcpyHostToDevice));
        const dim3 g (256), t (1024);
        auto byval = [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL (k_val_wide, g, t, 0, s, d, z); };
        auto byptr = [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL (k_ptr_wide, g, t, 0, s, d, (const big *) dz); };
        hipGraph_t gv, gp; hipGraphExec_t ev, ep;
        CHK (hipStreamBeginCapture (s, hipStreamCaptureModeThreadLocal)); byval (); CHK (hipStreamEndCapture (s, &gv)); CHK (hipGraphInstantiate (&ev, gv, nullptr, nullptr, 0));
        CHK (hipStreamBeginCapture (s, hipStreamCaptureModeThreadLocal)); byptr (); CHK (hipStreamEndCapture (s, &gp)); CHK (hipGraphInstantiate (&ep, gp, nullptr, nullptr, 0));
        byval (); byptr (); (void) hipGraphLaunch (ev, s); (void) hipGraphLaunch (ep, s); CHK (hipStreamSynchronize (s));
        for (int r = 0; r < 3; ++r) {
            const double pv = timed (s, 7, byval), pp = timed (s, 7, byptr);
            const double gvu = timed (s, 7, [&] { (void) hipGraphLaunch (ev, s); }), gpu = timed (s, 7, [&] { (void) hipGraphLaunch (ep, s); });
            printf ("256 x 1024 grid, arguments needed for the first load, us per launch: plain by value (480 B) %.3f, plain by pointer to a resident block %.3f, graph by value %.3f, graph by pointer %.3f\n",
                    pv / N, pp / N, gvu / N, gpu / N);
        }
    }
    return 0;
}
