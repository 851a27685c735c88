// valu_issue_probe.hip — measurement (not a test, not part of the engine): cycles a SIMD spends per wave64 vector instruction, by
// instruction class, at 1 / 2 / 4 / 8 waves per SIMD.  bench.py prices the dominant kernel's executed VALU instructions (SQ_INSTS_VALU)
// with these constants (profiles/r06_valu_issue.txt -> profiles/valu_issue.json); round 5 assumed 4 cycles for every class.
//
// Method: every CU holds exactly w waves per SIMD (blocks of 256 w threads for w <= 4, two blocks of 1024 for w = 8; a block's dynamic
// LDS is sized so that no CU can take more than its share, and HW_REG_HW_ID says where every wave ran).  A wave runs N trips of a loop
// whose body is 32 independent instructions of ONE class over 16 accumulators (the same register is written every 16 instructions: no
// dependency stall), between two s_memtime stamps (shader cycles).  Per SIMD: (last end - first start) / instructions issued on it.
//   hipcc --offload-arch=gfx950 -O2 -o tests/cpp/valu_issue_probe tests/cpp/valu_issue_probe.hip && tests/cpp/valu_issue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

typedef float f2 __attribute__ ((ext_vector_type (2)));

enum { OP_FMA, OP_ADD, OP_MUL, OP_MAX, OP_FMAC, OP_PK_FMA, OP_PK_ADD, OP_PK_MUL, OP_MIN_DPP, OP_MOV_DPP, OP_CNDMASK, OP_CMP, OP_CMP_CND, OP_FMA64, OP_ADD64, OP_ADD_U32,
       OP_AND, OP_LSHL, OP_MOV, OP_MAD_U24, OP_SUB_CO, OP_RCP, OP_SQRT, OP_READLANE, OP_CND_SGPR, OP_MIN, OP_MIN3, OP_SUB, OP_MIN_U32, OP_CMP_SGPR, OP_BFE, OP_AND_OR, OP_MUL_LO, OP_MIX_SEARCH, OP_COUNT };
static const char *NAMES[OP_COUNT] = { "v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_f32", "v_fmac_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32",
                                       "v_min_f32_dpp", "v_mov_b32_dpp", "v_cndmask_b32", "v_cmp_lt_f32", "v_cmp_lt_f32+v_cndmask_b32 (pair)", "v_fma_f64", "v_add_f64", "v_add_u32",
                                       "v_and_b32", "v_lshlrev_b32", "v_mov_b32", "v_mad_u32_u24", "v_sub_co_u32", "v_rcp_f32", "v_sqrt_f32", "v_readlane_b32",
                                       "v_cndmask_b32 (e64, SGPR mask)", "v_min_f32", "v_min3_f32", "v_sub_f32", "v_min_u32", "v_cmp_lt_f32 (e64, SGPR pair)", "v_bfe_u32", "v_and_or_b32", "v_mul_lo_u32",
                                       "search mix (3 v_pk_add + 1 v_pk_mul + 2 v_pk_fma + v_fma + v_cmp + 2 v_cndmask per candidate)" };

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__ (1024) void k_probe (unsigned long long *out, float *sink, int trips, float seed)
{
    extern __shared__ float s_pad[];
    if (threadIdx.x == 0) s_pad[0] = seed;
    float a[16]; f2 p[16]; double d[16]; unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + (float) (threadIdx.x + i); p[i] = f2 { a[i], a[i] * 0.5f }; d[i] = (double) a[i]; u[i] = (unsigned) threadIdx.x * 2654435761u + (unsigned) i; }
    float c1 = 1.0000001f, c2 = 1e-9f; f2 pc1 = { c1, c1 }, pc2 = { c2, c2 }; double dc1 = 1.0000001, dc2 = 1e-9;
    asm volatile ("" : "+v"(c1), "+v"(c2), "+v"(pc1), "+v"(pc2), "+v"(dc1), "+v"(dc2));      // (loop-invariant operands in VGPRs: nothing but the class under test inside the loop)
    int sacc = 0; unsigned long long macc = 0ull, smask = 0x5555555555555555ull;
    asm volatile ("s_mov_b64 vcc, %0" :: "s"(smask) : "vcc");      // (the selects of the v_cndmask stream read a defined mask)
    asm volatile ("" : "+s"(smask));
    __syncthreads ();
    unsigned long long t0, t1;
    asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < trips; ++t) {
#define TWICE(X) X X
#define FMA_(i) asm volatile ("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define ADD_(i) asm volatile ("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define MUL_(i) asm volatile ("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
#define MAX_(i) asm volatile ("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define FMAC_(i) asm volatile ("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define PKFMA_(i) asm volatile ("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc1), "v"(pc2));
#define PKADD_(i) asm volatile ("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc2));
#define PKMUL_(i) asm volatile ("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
#define MINDPP_(i) asm volatile ("v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define MOVDPP_(i) asm volatile ("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define CND_(i) asm volatile ("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c2) : );
#define CMP_(i) asm volatile ("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c2) : "vcc");
#define CMPCND_(i) asm volatile ("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c2) : "vcc");
#define FMA64_(i) asm volatile ("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc1), "v"(dc2));
#define ADD64_(i) asm volatile ("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc2));
#define ADDU_(i) asm volatile ("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define AND_(i) asm volatile ("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define LSHL_(i) asm volatile ("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define MOV_(i) asm volatile ("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define MAD24_(i) asm volatile ("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define SUBCO_(i) asm volatile ("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "vcc");
#define RCP_(i) asm volatile ("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define SQRT_(i) asm volatile ("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define CNDS_(i) asm volatile ("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c2), "s"(smask));
#define MIN_(i) asm volatile ("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define MIN3_(i) asm volatile ("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c2), "v"(c1));
#define SUB_(i) asm volatile ("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define MINU_(i) asm volatile ("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define CMPS_(i) { unsigned long long m_; asm volatile ("v_cmp_lt_f32 %0, %1, %2" : "=s"(m_) : "v"(a[i]), "v"(c2)); macc ^= m_; }
#define BFE_(i) asm volatile ("v_bfe_u32 %0, %0, 3, 7" : "+v"(u[i]));
#define ANDOR_(i) asm volatile ("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define MULLO_(i) asm volatile ("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define RDL_(i) { int s_; asm volatile ("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(u[i])); sacc ^= s_; }
        // one list candidate of the search's stage 2 (KS_CAND): three packed subtractions, a packed multiply, two packed fmas, one fma, the
        // compare and the two selects; 10 instructions, 16 candidates' worth over the accumulators = 160 per trip
#define MIX_(i) asm volatile ("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %0, %0, %5\n\tv_pk_mul_f32 %0, %0, %0\n\t"       \
                              "v_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %0, %5, %0\n\tv_fma_f32 %2, %2, %6, %7\n\tv_cmp_lt_f32 vcc, %2, %7\n\t" \
                              "v_cndmask_b32 %2, %2, %6, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc"                                                        \
                              : "+v"(p[i]), "+v"(p[(i + 8) & 15]), "+v"(a[i]), "+v"(u[i]) : "v"(pc2), "v"(pc1), "v"(c1), "v"(c2), "v"(u[(i + 1) & 15]) : "vcc");
        if (OP == OP_FMA)      { TWICE (R16 (FMA_)) }
        if (OP == OP_ADD)      { TWICE (R16 (ADD_)) }
        if (OP == OP_MUL)      { TWICE (R16 (MUL_)) }
        if (OP == OP_MAX)      { TWICE (R16 (MAX_)) }
        if (OP == OP_FMAC)     { TWICE (R16 (FMAC_)) }
        if (OP == OP_PK_FMA)   { TWICE (R16 (PKFMA_)) }
        if (OP == OP_PK_ADD)   { TWICE (R16 (PKADD_)) }
        if (OP == OP_PK_MUL)   { TWICE (R16 (PKMUL_)) }
        if (OP == OP_MIN_DPP)  { TWICE (R16 (MINDPP_)) }
        if (OP == OP_MOV_DPP)  { TWICE (R16 (MOVDPP_)) }
        if (OP == OP_CNDMASK)  { TWICE (R16 (CND_)) }
        if (OP == OP_CMP)      { TWICE (R16 (CMP_)) }
        if (OP == OP_CMP_CND)  { R16 (CMPCND_) }
        if (OP == OP_FMA64)    { TWICE (R16 (FMA64_)) }
        if (OP == OP_ADD64)    { TWICE (R16 (ADD64_)) }
        if (OP == OP_ADD_U32)  { TWICE (R16 (ADDU_)) }
        if (OP == OP_AND)      { TWICE (R16 (AND_)) }
        if (OP == OP_LSHL)     { TWICE (R16 (LSHL_)) }
        if (OP == OP_MOV)      { TWICE (R16 (MOV_)) }
        if (OP == OP_MAD_U24)  { TWICE (R16 (MAD24_)) }
        if (OP == OP_SUB_CO)   { TWICE (R16 (SUBCO_)) }
        if (OP == OP_RCP)      { TWICE (R16 (RCP_)) }
        if (OP == OP_SQRT)     { TWICE (R16 (SQRT_)) }
        if (OP == OP_READLANE) { TWICE (R16 (RDL_)) }
        if (OP == OP_CND_SGPR) { TWICE (R16 (CNDS_)) }
        if (OP == OP_MIN)      { TWICE (R16 (MIN_)) }
        if (OP == OP_MIN3)     { TWICE (R16 (MIN3_)) }
        if (OP == OP_SUB)      { TWICE (R16 (SUB_)) }
        if (OP == OP_MIN_U32)  { TWICE (R16 (MINU_)) }
        if (OP == OP_CMP_SGPR) { TWICE (R16 (CMPS_)) }
        if (OP == OP_BFE)      { TWICE (R16 (BFE_)) }
        if (OP == OP_AND_OR)   { TWICE (R16 (ANDOR_)) }
        if (OP == OP_MUL_LO)   { TWICE (R16 (MULLO_)) }
        if (OP == OP_MIX_SEARCH) { R16 (MIX_) }
    }
    asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    unsigned hw, xcc;
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y + (float) d[i] + (float) u[i];
    if (acc == 12345.678f || sacc == 0x7fffffff || macc == 0x123456789ull) sink[0] = acc + s_pad[0];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[3 * w] = t0; out[3 * w + 1] = t1; out[3 * w + 2] = ((unsigned long long) (xcc & 0xF) << 32) | hw;
    }
}

struct result { double cyc_per_inst, lo, hi; int simds; double waves_per_simd; };

template <int OP>
static result run (int w, int ncu, unsigned long long *dout, float *dsink, int trips)
{
    const int threads = w <= 4 ? 256 * w : 1024, blocks_per_cu = w <= 4 ? 1 : w / 4;
    const int nblocks = ncu * blocks_per_cu, waves = nblocks * (threads / 64);
    // LDS per block: more than half of what is left once this CU's share is resident, so one more block never fits
    const size_t lds = (size_t) (160 * 1024 / blocks_per_cu) - 1024;
    hipFuncSetAttribute ((const void *) k_probe<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    const int per_trip = (OP == OP_CMP_CND) ? 32 : (OP == OP_MIX_SEARCH) ? 160 : 32;
    for (int rep = 0; rep < 2; ++rep)                                  // (first pass: clocks)
        hipLaunchKernelGGL (k_probe<OP>, dim3 (nblocks), dim3 (threads), lds, 0, dout, dsink, trips, 1.0f);
    if (hipDeviceSynchronize () != hipSuccess) { printf ("launch failed: %s\n", NAMES[OP]); exit (1); }
    std::vector<unsigned long long> h ((size_t) waves * 3);
    hipMemcpy (h.data (), dout, h.size () * 8, hipMemcpyDeviceToHost);
    struct acc { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; };
    std::map<unsigned long long, acc> simd;
    for (int i = 0; i < waves; ++i) {
        const unsigned long long id = h[3 * i + 2], hw = id & 0xFFFFFFFFull, xcc = id >> 32;
        const unsigned long long key = (xcc << 24) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3);
        acc &a = simd[key];
        a.t0 = std::min (a.t0, h[3 * i]); a.t1 = std::max (a.t1, h[3 * i + 1]); ++a.n;
    }
    std::vector<double> c;
    double wsum = 0;
    for (auto &kv : simd) { c.push_back ((double) (kv.second.t1 - kv.second.t0) / ((double) kv.second.n * trips * per_trip)); wsum += kv.second.n; }
    std::sort (c.begin (), c.end ());
    return { c[c.size () / 2], c[c.size () / 20], c[c.size () - 1 - c.size () / 20], (int) c.size (), wsum / c.size () };
}

template <int OP>
static void all (int ncu, unsigned long long *dout, float *dsink, FILE *js, bool last)
{
    printf ("%-44s", NAMES[OP]);
    fprintf (js, "  \"%s\": {", NAMES[OP]);
    const int ws[4] = { 1, 2, 4, 8 };
    for (int i = 0; i < 4; ++i) {
        const result r = run<OP> (ws[i], ncu, dout, dsink, OP == OP_MIX_SEARCH ? 400 : 2000);
        printf ("  %5.2f (%4.2f-%5.2f; %.1f w/SIMD on %d)", r.cyc_per_inst, r.lo, r.hi, r.waves_per_simd, r.simds);
        fprintf (js, "\"%d\": %.3f%s", ws[i], r.cyc_per_inst, i < 3 ? ", " : "");
    }
    printf ("\n");
    fprintf (js, "}%s\n", last ? "" : ",");
    fflush (stdout);
}

int main (int argc, char **argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties (&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned long long *dout; float *dsink;
    hipMalloc (&dout, (size_t) ncu * 2 * 16 * 3 * 8); hipMalloc (&dsink, 4);
    FILE *js = fopen (argc > 1 ? argv[1] : "valu_issue.json", "w");
    printf ("%s, %d CUs: SIMD cycles per wave64 instruction (median over the SIMDs; 5th - 95th percentile; waves per SIMD as HW_ID reports them)\n", prop.name, ncu);
    printf ("%-44s  %-38s  %-38s  %-38s  %-38s\n", "instruction", "1 wave / SIMD", "2 waves / SIMD", "4 waves / SIMD", "8 waves / SIMD");
    fprintf (js, "{\n  \"_what\": \"SIMD cycles per wave64 instruction at 1, 2, 4, 8 waves per SIMD (tests/cpp/valu_issue_probe.hip)\",\n  \"_device\": \"%s\",\n", prop.name);
    all<OP_FMA> (ncu, dout, dsink, js, false); all<OP_ADD> (ncu, dout, dsink, js, false); all<OP_MUL> (ncu, dout, dsink, js, false); all<OP_MAX> (ncu, dout, dsink, js, false);
    all<OP_FMAC> (ncu, dout, dsink, js, false); all<OP_PK_FMA> (ncu, dout, dsink, js, false); all<OP_PK_ADD> (ncu, dout, dsink, js, false); all<OP_PK_MUL> (ncu, dout, dsink, js, false);
    all<OP_MIN_DPP> (ncu, dout, dsink, js, false); all<OP_MOV_DPP> (ncu, dout, dsink, js, false); all<OP_CNDMASK> (ncu, dout, dsink, js, false); all<OP_CMP> (ncu, dout, dsink, js, false);
    all<OP_CMP_CND> (ncu, dout, dsink, js, false); all<OP_FMA64> (ncu, dout, dsink, js, false); all<OP_ADD64> (ncu, dout, dsink, js, false); all<OP_ADD_U32> (ncu, dout, dsink, js, false);
    all<OP_AND> (ncu, dout, dsink, js, false); all<OP_LSHL> (ncu, dout, dsink, js, false); all<OP_MOV> (ncu, dout, dsink, js, false); all<OP_MAD_U24> (ncu, dout, dsink, js, false);
    all<OP_SUB_CO> (ncu, dout, dsink, js, false); all<OP_RCP> (ncu, dout, dsink, js, false); all<OP_SQRT> (ncu, dout, dsink, js, false); all<OP_READLANE> (ncu, dout, dsink, js, false);
    all<OP_CND_SGPR> (ncu, dout, dsink, js, false); all<OP_MIN> (ncu, dout, dsink, js, false); all<OP_MIN3> (ncu, dout, dsink, js, false); all<OP_SUB> (ncu, dout, dsink, js, false);
    all<OP_MIN_U32> (ncu, dout, dsink, js, false); all<OP_CMP_SGPR> (ncu, dout, dsink, js, false); all<OP_BFE> (ncu, dout, dsink, js, false); all<OP_AND_OR> (ncu, dout, dsink, js, false); all<OP_MUL_LO> (ncu, dout, dsink, js, false);
    all<OP_MIX_SEARCH> (ncu, dout, dsink, js, true);
    fprintf (js, "}\n");
    fclose (js);
    return 0;
}
