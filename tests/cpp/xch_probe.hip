// xch_probe.hip — diagnostic (not a test, not part of the engine): what an in-launch all-to-all of the per-block moments costs on
// this machine, measured with s_memtime inside one launch of 256 co-resident 1024-thread blocks (the shape of round 2's
// persistent ICP run: one block per CU, 18 double moments per block and iteration).
//
//   flat      every block publishes its 18 records with agent-scope (sc1, write-through) stores and polls all 256 x 18 records
//             with agent-scope loads (36 rows of 16 lanes, 8 records of 16 bytes per lane): round 2's exchange.
//   two-hop   hop 1 inside an XCD: workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b mod 8; checked against
//             HW_REG_XCC_ID), the 32 blocks of an XCD share one L2, so their records need only bypass the L1 (sc0 loads; plain
//             stores are written through to the L2): 32 x 18 records per block — polled with sc0 loads (mode 1), sc1 loads (mode 2) or
//             returning L2 atomics (mode 3).  Hop 2 across XCDs: the first block of every
//             XCD publishes the XCD's 18 sums (sc1), every block polls 8 x 18 records (sc1).
//             (With the identity block -> tile mapping the blocks of an XCD are the tiles = x (mod 8): closed under the levels
//              64 .. 8 of the canonical 128-position tree, so the two hops can reproduce the canonical sum bit for bit.)
//
// Every spin is bounded; a block that gives up raises an abort word the others watch.  Prints cycles (shader clock) per phase:
// mean over blocks and rounds.      hipcc --offload-arch=gfx950 -O2 -o tests/cpp/xch_probe tests/cpp/xch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define NB 256u
#define NM 18u
#define SPIN_LIMIT 400000u
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef unsigned long long __attribute__ ((address_space (1))) gu64;

static __device__ __forceinline__ unsigned long long now ()
{
    unsigned long long t;
    asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
static __device__ __forceinline__ void st_sc1 (void *p, u32x4 v) { asm volatile ("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
static __device__ __forceinline__ void st_plain (void *p, u32x4 v) { asm volatile ("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
static __device__ __forceinline__ u32x4 ld_sc1 (const void *p) { u32x4 v; asm volatile ("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
static __device__ __forceinline__ u32x4 ld_sc0 (const void *p) { u32x4 v; asm volatile ("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

// recF: [2][NM][NB] flat records; recL: [2][8][NM][32] XCD-local; recG: [2][NM][8] XCD sums.  16 bytes each: {lo, epoch, hi, epoch}.
__global__ __launch_bounds__ (1024) void k_probe (u32x4 *recF, u32x4 *recL, u32x4 *recG, uint32_t *abortw, uint32_t *xcc, unsigned long long *out,
                                                  uint32_t rounds, int mode)
{
    __shared__ uint32_t s_fail;
    __shared__ double s_sum[NM];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, l = tid & 15u, row = tid >> 4;
    const uint32_t x = b & 7u, j = b >> 3;
    if (tid == 0) { uint32_t id; asm volatile ("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(id)); xcc[b] = id; s_fail = 0u; }
    __syncthreads ();
    unsigned long long acc[3] = { 0ull, 0ull, 0ull };
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t epoch = r + 1u, par = r & 1u;
        const unsigned long long t0 = now ();
        const double mine = (double) (b + 1u) * 0.5 + (double) r;
        const unsigned long long u = __builtin_bit_cast (unsigned long long, mine);
        const u32x4 rec = { (uint32_t) u, epoch, (uint32_t) (u >> 32), epoch };
        bool gave_up = false;
        if (mode == 0) {
            if (tid < NM) st_sc1 (recF + ((size_t) par * NM + tid) * NB + b, rec);
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            if (row < 2u * NM) {
                const uint32_t k = row >> 1, g = row & 1u;
                uint32_t spins = 0;
                for (;;) {
                    bool ok = true;
                    for (uint32_t q = 0; q < 8u; ++q) {
                        const u32x4 v = ld_sc1 (recF + ((size_t) par * NM + k) * NB + g * 128u + l + 16u * q);
                        ok = ok && v.y == epoch && v.w == epoch;
                    }
                    if (__builtin_amdgcn_ballot_w64 (!ok) == 0ull) break;
                    if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load (abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep (1);
                }
            }
        } else {
            // hop 1: inside the XCD
            if (tid < NM) {
                if (mode == 1 || mode == 3) st_plain (recL + (((size_t) par * 8u + x) * NM + tid) * 32u + j, rec);
                else st_sc1 (recL + (((size_t) par * 8u + x) * NM + tid) * 32u + j, rec);          // mode 2: the same two hops, agent scope throughout
            }
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            double part = 0.0;
            if (row < 2u * NM) {
                const uint32_t k = row >> 1, h = row & 1u;
                uint32_t spins = 0;
                for (;;) {
                    const u32x4 *p = recL + (((size_t) par * 8u + x) * NM + k) * 32u + h * 16u + l;
                    u32x4 v;
                    if (mode == 3) {                 // returning L2 atomics (add 0) on the two self-validating 8-byte granules: RMW atomics never hit the L1
                        unsigned long long *q = (unsigned long long *) p;
                        const unsigned long long a0 = __hip_atomic_fetch_add (q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const unsigned long long a1 = __hip_atomic_fetch_add (q + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        v = (u32x4) { (uint32_t) a0, (uint32_t) (a0 >> 32), (uint32_t) a1, (uint32_t) (a1 >> 32) };
                    } else v = mode == 1 ? ld_sc0 (p) : ld_sc1 (p);
                    const bool ok = v.y == epoch && v.w == epoch;
                    part = __builtin_bit_cast (double, (unsigned long long) v.x | ((unsigned long long) v.z << 32));
                    if (__builtin_amdgcn_ballot_w64 (!ok) == 0ull) break;
                    if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load (abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep (1);
                }
                for (int d = 8; d > 0; d >>= 1) part += __shfl_down (part, d, 16);
                const double other = __shfl_down (part, 16);
                if (l == 0 && h == 0) s_sum[k] = part + other;
            }
        }
        if (gave_up && lane == 0) { s_fail = 1u; __hip_atomic_store (abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __syncthreads ();
        if (s_fail) { if (tid == 0) out[(size_t) b * 4 + 3] = 1ull; return; }
        const unsigned long long t1 = now ();
        if (mode != 0) {
            // hop 2: the XCD's first block publishes the 18 sums, every block polls the 8 x 18 records
            if (j == 0 && tid < NM) {
                const unsigned long long us = __builtin_bit_cast (unsigned long long, s_sum[tid]);
                st_sc1 (recG + ((size_t) par * NM + tid) * 8u + x, (u32x4) { (uint32_t) us, epoch, (uint32_t) (us >> 32), epoch });
            }
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
            if (row < 9u) {                          // 144 records, 16 per row
                uint32_t spins = 0;
                for (;;) {
                    const u32x4 v = ld_sc1 (recG + (size_t) par * NM * 8u + row * 16u + l);
                    const bool ok = v.y == epoch && v.w == epoch;
                    if (__builtin_amdgcn_ballot_w64 (!ok) == 0ull) break;
                    if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load (abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep (1);
                }
            }
            if (gave_up && lane == 0) { s_fail = 1u; __hip_atomic_store (abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            __syncthreads ();
            if (s_fail) { if (tid == 0) out[(size_t) b * 4 + 3] = 1ull; return; }
        }
        const unsigned long long t2 = now ();
        if (r >= 16u) { acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += 1ull; }
    }
    if (tid == 0) { out[(size_t) b * 4] = acc[0]; out[(size_t) b * 4 + 1] = acc[1]; out[(size_t) b * 4 + 2] = acc[2]; }
}

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf (stderr, "%s: %s\n", #e, hipGetErrorString (e_)); return 1; } } while (0)

int main ()
{
    u32x4 *recF, *recL, *recG; uint32_t *abortw, *xcc; unsigned long long *out;
    const size_t nF = 2ull * NM * NB, nL = 2ull * 8 * NM * 32, nG = 2ull * NM * 8;
    CHK (hipMalloc ((void **) &recF, nF * 16)); CHK (hipMalloc ((void **) &recL, nL * 16)); CHK (hipMalloc ((void **) &recG, nG * 16));
    CHK (hipMalloc ((void **) &abortw, 4)); CHK (hipMalloc ((void **) &xcc, NB * 4)); CHK (hipMalloc ((void **) &out, NB * 4 * 8));
    const uint32_t rounds = 216;
    const char *names[4] = { "flat (256 x 18 records, sc1)", "two-hop (XCD-local sc0 loads, then 8 x 18 sc1)", "two-hop, agent scope (sc1) in both hops",
                             "two-hop (XCD-local L2 atomics, then sc1)" };
    for (int mode = 0; mode < 4; ++mode) {
        CHK (hipMemset (recF, 0, nF * 16)); CHK (hipMemset (recL, 0, nL * 16)); CHK (hipMemset (recG, 0, nG * 16));
        CHK (hipMemset (abortw, 0, 4)); CHK (hipMemset (out, 0, NB * 4 * 8));
        hipLaunchKernelGGL (k_probe, dim3 (NB), dim3 (1024), 0, 0, recF, recL, recG, abortw, xcc, out, rounds, mode);
        CHK (hipDeviceSynchronize ());
        std::vector<unsigned long long> h (NB * 4); std::vector<uint32_t> hx (NB);
        CHK (hipMemcpy (h.data (), out, NB * 4 * 8, hipMemcpyDeviceToHost)); CHK (hipMemcpy (hx.data (), xcc, NB * 4, hipMemcpyDeviceToHost));
        unsigned failed = 0, mismatch = 0; double a0 = 0, a1 = 0, n = 0;
        for (unsigned b = 0; b < NB; ++b) { failed += h[b * 4 + 3] ? 1u : 0u; mismatch += (hx[b] != (b & 7u)) ? 1u : 0u; a0 += (double) h[b * 4]; a1 += (double) h[b * 4 + 1]; n += (double) h[b * 4 + 2]; }
        if (failed) { printf ("%-44s GAVE UP in %u blocks (XCC_ID != block mod 8 for %u blocks)\n", names[mode], failed, mismatch); continue; }
        printf ("%-44s publish + %s %8.0f cycles%s   total %8.0f cycles = %.2f us at 2.3 GHz   (XCC_ID != block mod 8 for %u of %u blocks)\n", names[mode],
                mode ? "hop 1" : "gather", a0 / n, mode ? (std::string (", hop 2 ") + std::to_string ((long) (a1 / n)) + " cycles").c_str () : "", (a0 + a1) / n, (a0 + a1) / n / 2300.0, mismatch, NB);
    }
    return 0;
}
