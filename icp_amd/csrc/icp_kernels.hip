// icp_kernels.hip — hand-written HIP kernels (gfx950) of the photogeometric ICP iteration.
//
// Replaces kernels/icp_kernels.cl, kernels/reduce_kernels.cl, kernels/scan_kernels.cl and the
// un-vendored kernels/RBC/*.cl of the reference.  Per-iteration launch set (reference: >= 18 launches
// plus a blocking 32-byte read and a 32-byte write, src/ICP/algorithms.cpp:4670-4698):
//
//   reference-order reductions (4 launches)
//   k_search    transform (a3) + nearest representative + list scan (a4) + weights and their
//               128-element tree partials (a5, first level)
//   k_means     sum of weights (a5, second level, in the prologue) + weighted block means (a6)
//   k_sij       global means (a6 icpGMean, prologue) + deviations (a7) + S products and their
//               512-column tree partials (a8)
//   k_finalize  S final tree (a8) + power method / SVD (a9, a12) + composition (a10) + check (a11)
//
//   fused reductions (2 launches, or 1 chained launch where the size is latency-bound: DESIGN.md §5)
//   k_search<FUSED>   the same search + the 18 double moments of every 64-pair block
//   k_finalize_fused  moment trees, means / S, rotation, composition, check — or, chained, the prologue of the
//                     next k_search<FUSED, CHAIN>
//
// RBC construction (once per fixed frame; icp_build.hip): k_search<.., OWNER> (here: step 1, the owner search) + k_place_lists at the
// latency-bound sizes; k_reps_and_boxes, k_search<.., OWNER>, k_chunk_hist, k_count_offsets (or k_count, k_offsets beyond 1024
// representatives), k_place otherwise.
//
// Every reduction follows the canonical tree of DESIGN.md §3, so results are bit-identical to
// oracle/icp_oracle.c.  blockIdx.y is the registration index of a batch.
#include "icp_kernels.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ float sum4 (float4 v) { return ((v.x + v.y) + v.z) + v.w; }

// Sum of weights.  k_search leaves, per 128-query group g, the two half-trees hp[2g] (even positions)
// and hp[2g+1] (odd positions); their float sum is the work-group partial of
// icpComputeReduceWeights_WG (kernels/icp_kernels.cl:244-253, last tree level).  The partials then go
// through reduce_sum_fd (:295-329): position p = ((w[4p] + w[4p+1]) + w[4p+2]) + w[4p+3] in double, tree
// over 128 positions; chunks of 512 partials are summed in index order (oracle orc_weights).
// Executed by every 16-lane row (all lanes active); the result is valid in lane 0 of each row.
static __device__ double sum_w_row (const float *hp, uint32_t nwp, uint32_t l)
{
    if (nwp == 1) return (double) (hp[0] + hp[1]);
    double total = 0.0;
    for (uint32_t c0 = 0; c0 < nwp; c0 += 512) {
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t w0 = c0 + 4u * (l + 16u * k);
            a[k] = 0.0;
            if (w0 < nwp) {
                float4 h0 = *reinterpret_cast<const float4 *> (hp + 2 * (size_t) w0);
                float4 h1 = *reinterpret_cast<const float4 *> (hp + 2 * (size_t) w0 + 4);
                float p0 = h0.x + h0.y, p1 = h0.z + h0.w, p2 = h1.x + h1.y, p3 = h1.z + h1.w;
                a[k] = (((double) p0 + (double) p1) + (double) p2) + (double) p3;
            }
        }
        double cs = row_tree8_d (a);
        total = (c0 == 0) ? cs : total + cs;
    }
    return total;
}

// icpGMean over <= 128 block means of one set (kernels/icp_kernels.cl:530-566) by one 16-lane row.
static __device__ float4 gmean_row (const float4 *blk, uint32_t nblk, uint32_t l)
{
    float x[8], y[8], z[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t i = l + 16u * k;
        float4 v = make_float4 (0.f, 0.f, 0.f, 0.f);
        if (i < nblk) v = blk[i];
        x[k] = v.x; y[k] = v.y; z[k] = v.z;
    }
    float4 r;
    r.x = row_tree8 (x); r.y = row_tree8 (y); r.z = row_tree8 (z); r.w = 0.f;
    return r;
}

// Final S value of one row of the 11 x G product matrix from the first-level partials of k_sij.
// k_sij leaves, per 512-column work-group w, the 8 sub-trees over positions = r (mod 8); the remaining
// levels d = 4, 2, 1 of reduce_sum_f's tree (kernels/reduce_kernels.cl:254-259) combine them, then the
// second reduce_sum_f pass (src/ICP/algorithms.cpp:140-173) runs over the work-group partials.
// One 16-lane row, all lanes active; valid in lane 0 of the row.  nwgp = padded work-group count.
static __device__ float s_reduce_row (const float *sp, uint32_t nwgp, uint32_t l)
{
    if (nwgp == 1) {
        float4 a = *reinterpret_cast<const float4 *> (sp), b = *reinterpret_cast<const float4 *> (sp + 4);
        return ((a.x + b.x) + (a.z + b.z)) + ((a.y + b.y) + (a.w + b.w));
    }
    float pos[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t w0 = 4u * (l + 16u * k);
        pos[k] = 0.f;
        if (w0 < nwgp) {
            float R[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *s8 = sp + (size_t) (w0 + q) * 8;
                float4 a = *reinterpret_cast<const float4 *> (s8), b = *reinterpret_cast<const float4 *> (s8 + 4);
                R[q] = ((a.x + b.x) + (a.z + b.z)) + ((a.y + b.y) + (a.w + b.w));
            }
            pos[k] = ((R[0] + R[1]) + R[2]) + R[3];
        }
    }
    return row_tree8 (pos);
}

// ------------------------------------------------------------------------------------------
// K1  search: transform (a3) + RBC one-shot search (a4) + weights and their first tree levels (a5)
//
//   block  = 64 queries x LPQ lanes (LPQ waves); reference order: the 64 even (or odd) positions of one 128-query
//            group, so that the block owns a closed sub-tree of the weight reduction; fused: an 8 x 8 tile of the grid;
//   wave   = 64 / LPQ queries end to end: the LPQ lanes of a query split the representatives (stage 1) and the list
//            positions (stage 2); representatives are staged through LDS in tiles (broadcast ds_read_b128).
// ------------------------------------------------------------------------------------------
// KS_SPLIT = lanes per query = waves per block (8 or 16); inside k_search it names the template parameter LPQ
#define KS_SPLIT LPQ
#define KS_QPW (64 / KS_SPLIT)  // queries per wave
#define KS_TILE 1024u            // representatives per LDS tile
#ifndef ICP_S2_DEPTH16
#define ICP_S2_DEPTH16 8u            // stage 2, 16 lanes per query: candidates in flight per lane (8 x 16 = 128 covers every list at |R| = m/64)
#endif
#ifndef ICP_S1_SEED
#define ICP_S1_SEED 1                // stage 1: prune with the distance to the previous search's nearest representative
#endif
#ifndef ICP_S1_REJECT_MIN_NR
#define ICP_S1_REJECT_MIN_NR 1024u   // stage 1: exact early rejection from this many representatives on
#endif

typedef float float2v __attribute__ ((ext_vector_type (2)));
// MASKED search, home tile (see k_search): 1 = staged into LDS in the prologue (seed from LDS); 2 = loads issued in the prologue, LDS
// write after the tile masks (seed from global: the tile's round trip overlaps the seed's); 3 = no home tile (only its list offsets).
// Same box, alternating, us per iteration of fresh 40-iteration runs at |F| = 65536 / 10-iteration runs at 2^20 (profiles/
// r03_home_tile_ab.txt): without the home tile 18.62 / 263.8, mode 1 17.90 / 258.8, mode 2 18.18 / 258.7, mode 3 18.79 / 261.6.
#ifndef ICP_HOME_MODE
#define ICP_HOME_MODE 1
#endif

#ifdef ICP_DBG_STAMPS
#define KS_STAMP(k)                                                                                       \
    {                                                                                                     \
        unsigned long long t_;                                                                            \
        asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (tid == 0 && p.dbg) p.dbg[(size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;    \
    }
#elif defined (ICP_DBG_EXIT_AFTER)
// diagnostic builds (tests/diag_phase_insts.sh): the search kernel ends behind phase k — the instruction counters of a PMC run
// then hold the phases up to k, and differences between builds are the phases themselves (every thread of a block gets here)
#define KS_STAMP(k) { if ((k) == ICP_DBG_EXIT_AFTER) return; }
#define KS_KEEP(a, b) asm volatile ("" :: "v"(a), "v"(b));      // (the phase's results count as used: nothing of it is optimised away)
#else
#define KS_STAMP(k)
#endif
#ifndef KS_KEEP
#define KS_KEEP(a, b)
#endif

// minimum over each group of LPQ (8 or 16) consecutive lanes, returned in all of them (min is exact: any pairing
// gives the same bits)
// One v_min_f32 whose first operand comes through DPP per step: fminf (v, dpp (v)) compiles to a DPP move, two canonicalising
// v_max and the v_min — the values here are results of arithmetic (never signalling NaNs) and v_min_f32 returns the other
// operand for a quiet NaN exactly as fminf does, so the one instruction gives the same bits (tests: every distance bit for bit).
// (s_nop 1: a DPP operand must not be read for two wait states after a VALU wrote it; the assembler text is opaque to the
// compiler's hazard pass.)
#define KS_MIN_DPP(v, CTRL) asm ("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))
template <int LPQ> static __device__ __forceinline__ float ks_grp_min_f (float v)
{
    KS_MIN_DPP (v, "quad_perm:[1,0,3,2]");
    KS_MIN_DPP (v, "quad_perm:[2,3,0,1]");
    KS_MIN_DPP (v, "row_half_mirror");               // lane i <-> 7 - i
    if (LPQ == 16) KS_MIN_DPP (v, "row_mirror");     // lane i <-> 15 - i
    return v;
}
template <int LPQ> static __device__ __forceinline__ uint32_t ks_grp_min_u (uint32_t v)
{
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0xB1, 0xF, 0xF, true));
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x4E, 0xF, 0xF, true));
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x141, 0xF, 0xF, true));
    if (LPQ == 16) v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x140, 0xF, 0xF, true));
    return v;
}

// candidate j of a list: XQ = [x r y g | z b id 0].  The geometric and the photometric sum of the metric are
// evaluated side by side, one packed instruction per step: (dx, dr), (dy, dg), (dz, db) -> (geo, pho) with exactly the
// operations of icp_metric8 (mul, fma, fma per half), then d = fma (a, pho, geo).  Keeps the best (distance, position).
#define KS_CAND(G, C, J)                                                                              \
    {                                                                                                 \
        const float2v d1_ = vq_xr - float2v { (G).x, (G).y }, d2_ = vq_yg - float2v { (G).z, (G).w }, \
                      d3_ = vq_zb - float2v { (C).x, (C).y };                                         \
        const float2v gp_ = __builtin_elementwise_fma (d3_, d3_, __builtin_elementwise_fma (d2_, d2_, d1_ * d1_)); \
        const float d_ = __builtin_fmaf (alpha, gp_.y, gp_.x);                                        \
        if (d_ < best2) { best2 = d_; bj = (J); }                                                     \
    }

#define ICP_NMOM 18

#ifdef ICP_DBG_STAMPS
#define FF_STAMP(k)                                                                                       \
    {                                                                                                     \
        unsigned long long t_;                                                                            \
        asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && p.dbg) p.dbg[(size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;    \
    }
#else
#define FF_STAMP(k)
#endif

// XCD-aware block -> tile mapping (fused mode, one-block-per-CU variants).  Workgroups are dealt round-robin over the 8 XCDs
// (blocks b and b + 8 share one; every XCD has its own L2).  Block b works on tile (b mod 8) * (nb / 8) + b / 8: the blocks of
// one XCD cover a contiguous band of tile rows, and — what the measurement says matters — a contiguous range of moment
// slots: every 128-byte line of the per-block moments is then written inside ONE L2 instead of collecting eight partial
// write-backs.  Measured at A (same box, alternating): identity 9.61, bands 9.42, a 2 x 4 arrangement of compact rectangles
// (better list locality, lines shared by four XCDs again) 9.66 us per iteration.  Only which block computes which tile
// changes — the tile index is what the query index, the moment slot and the canonical trees use, so the bits do not.
// (Speed only: nothing depends on the placement actually being round-robin.)
static __device__ __forceinline__ uint32_t ks_tile_of_block (uint32_t bx, uint32_t nbx)
{
#ifdef ICP_NO_XCD_MAP
    return bx;
#else
    return (nbx & 7u) == 0u ? (bx & 7u) * (nbx >> 3) + (bx >> 3) : bx;
#endif
}

// fused mode: query index of local element e of block b (CPU twin: orc_fused_query).  8 x 8 tiles of the
// landmark grid when its side is a multiple of 8, else 64 consecutive queries.
// tpr_magic = floor (2^32 / tpr) + 1 (host: icp_tpr_magic): b / tpr == umulhi (b, tpr_magic) for b * tpr < 2^32.
static __device__ __forceinline__ uint32_t fused_query_index (uint32_t m, uint32_t side, uint32_t tpr_magic, uint32_t b, uint32_t e)
{
    const uint32_t tpr = side >> 3, ty = __umulhi (b, tpr_magic), tx = b - ty * tpr;
    const uint32_t tiled = (8u * ty + (e >> 3)) * side + 8u * tx + (e & 7u);
    return (side && (side & 7u) == 0u && side * side == m) ? tiled : b * 64u + e;
}


// Representative sampled from the grid cell of point i (getReps' grid: cells of (side / nrx) x (side / nry) points), without a
// division: the magics are floor (2^32 / d) + 1 for d = side, side / nrx, side / nry (host: icp_div_magic; exact for n d < 2^32).
static __device__ __forceinline__ uint32_t cell_rep_of (const icp_params &p, uint32_t i)
{
    const uint32_t y = __umulhi (i, p.side_magic), x = i - y * p.side;
    return (p.cellh_magic ? __umulhi (y, p.cellh_magic) : y) * p.nrx + (p.cellw_magic ? __umulhi (x, p.cellw_magic) : x);      // (magic 0: cells of one point)
}

// Progress word of a host-driven checked run (ICP_MIRROR_WORD) -> fine-grained host memory: one 8-byte system-scope store nobody waits for
static __device__ __forceinline__ void icp_mirror_store (unsigned long long *dst, unsigned long long v)
{
    __hip_atomic_store (dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Result of turning one iteration's moments into the next transform: the registration state itself, staged
// in LDS (one per block) so that one wave publishes it with a single store
typedef icp_reg_state icp_fin_result;

// Block-cooperative: 128-position double trees over the block moments (rows 0..17 of the calling block,
// needs >= 288 threads), then wave 0: means / S from the moments (oracle orc_moments_finish), rotation,
// composition with the previous (T, R), convergence.  Every thread of the block must call it; `res` is valid
// for all threads after the call.  NG = capacity of the second tree level (groups of 128 blocks).
// `sv` = the previous state, lane-distributed: lane j of every wave holds dword j of the icp_reg_state (one
// coalesced vector load that is in flight together with the moment loads; scalar loads of the state would be
// waited for before the moment addresses exist).
static __device__ __forceinline__ uint32_t state_load_lanes (const icp_reg_state *st)
{
    const uint32_t lane = threadIdx.x & 63u;
    return reinterpret_cast<const uint32_t *> (st)[min (lane, (uint32_t) sizeof (icp_reg_state) / 4u - 1u)];
}
static __device__ __forceinline__ float state_lane_f (uint32_t sv, uint32_t dword)
{
    return __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) dword));
}
#define ICP_ST_DW(field) (offsetof (icp_reg_state, field) / 4)

// First tree level of the moments, pass ps: one 16-lane row per (moment k, group g) task.  The loads are a separate
// step so that a caller can issue them together with its other prologue loads (before anything waits).
template <int NT>
static __device__ __forceinline__ void fused_moment_task (uint32_t nb, uint32_t ps, uint32_t &k, uint32_t &g, bool &live)
{
    constexpr uint32_t nrow = NT / 16;
    const uint32_t row = threadIdx.x >> 4, ng = (nb + 127u) / 128u, ntask = ICP_NMOM * ng;
    const uint32_t task = min (ps * nrow + row, ntask - 1u);
    if (ng == 1) { k = task; g = 0u; } else if (ng == 2) { k = task >> 1; g = task & 1u; } else { k = task / ng; g = task - k * ng; }
    live = ps * nrow + row < ntask;                  // rows past the last task (whole waves when ntask % 4 == 0) load nothing
}
template <int NT>
static __device__ __forceinline__ void fused_moment_loads (const double *mom, uint32_t nb, uint32_t ps, double *a)
{
    uint32_t k, g; bool live;
    fused_moment_task<NT> (nb, ps, k, g, live);
    const double *src = mom + (size_t) k * nb;
    const uint32_t l = threadIdx.x & 15u;
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.0;
    if (live) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {                // clamped address + select: eight loads back to back
            const uint32_t i = g * 128u + l + 16u * q;
            const double t = src[min (i, nb - 1u)];
            a[q] = (i < nb) ? t : 0.0;
        }
    }
}

// Returns false (for every thread, before any barrier) when the registration had already converged (checked mode).
// a0 = the values of pass 0 (fused_moment_loads (mom, nb, 0, a0), issued by the caller with its other loads).
// gl1 != nullptr: the first tree level was evaluated by k_moment_level1 (many blocks: large sets); gl1[k * ng + g].
// LEAN (the chained search's prologue: every block runs this, all of them wait for T): the block is handed T and `done` only —
// ten LDS dwords instead of 62 —, and the one block that publishes the state (direct != nullptr) stores it straight from the
// composing lane's registers to global memory (16 vector stores nobody waits for).
// after (LEAN): called by every lane of the finishing wave with the new T, before the barrier that releases the block — the
// chained search transforms and hands over its queries there, so that one barrier covers T's consumers.
struct ff_no_hook { __device__ void operator() (const float *) const {} };
template <int NG, int NT, int ROT, bool LEAN = false, typename AFTER = ff_no_hook>
static __device__ bool fused_finalize_block (const icp_params &p, const double *mom, uint32_t nb, uint32_t check, uint32_t sv,
                                             const double *a0, icp_fin_result *res, double (*s_l1)[NG], double *s_t,
                                             const double *gl1 = nullptr, icp_reg_state *direct = nullptr, AFTER after = AFTER (),
                                             uint32_t pending_unless_done = 1u, unsigned long long *mirror = nullptr, bool progress = false,
                                             icp_reg_state *final_dst = nullptr, icp_reg_state *host_dst = nullptr, icp_reg_state *prev_dst = nullptr)
{
    // Host-driven checked runs (run_ctl in icp_capi.hip; mirror != nullptr on the lane that publishes): `progress` — every new (k, done) goes
    // to the registration's word in host memory (chained form: nothing waits for the store); a CONVERGED registration's final state is stored
    // to final_dst (chained: the user-visible state, which the launches behind this one no longer touch) and host_dst (host memory) in front
    // of the word's DONE | FINAL bits: the run needs no end kernel and the host reads the result the moment the flag shows.
    // prev_dst: the transform the search of this iteration used (T before the composition) — what a later search needs to reproduce the
    // iteration's per-query outputs, which checked runs do not store on the way (icp_launch_search on p.st_prev).
    // NT = threads of the calling block (compile-time: reading blockDim costs a dependent cold load at kernel start)
    constexpr uint32_t nrow = NT / 16;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, l = tid & 15u, row = tid >> 4;
    const uint32_t ng = (nb + 127u) / 128u;
    const uint32_t ntask = ICP_NMOM * ng, npass = (ntask + nrow - 1u) / nrow;
    auto pass = [&] (uint32_t ps, const double *a) {
        uint32_t k, g; bool live;
        fused_moment_task<NT> (nb, ps, k, g, live);
        FF_STAMP (13)
        double v = row_tree8_d (a);
        if (nb == 1) v = a[0];
        if (ng == 2) {
            // the two groups of a moment sit in adjacent rows of one wave, and the second level (a 128-position
            // tree over [g0, g1, 0, ..]) is (g0 + 0) + (g1 + 0): no LDS round trip, no second barrier
            const double o = __shfl_down (v, 16);
            if (l == 0 && !(row & 1u) && live) s_t[k] = (v + 0.0) + (o + 0.0);
        } else if (l == 0 && live) { if (ng == 1) s_t[k] = v; else s_l1[k][g] = v; }
    };
    if (gl1 == nullptr) {
        pass (0u, a0);
        for (uint32_t ps = 1; ps < npass; ++ps) {    // small blocks / many groups only
            double a[8];
            fused_moment_loads<NT> (mom, nb, ps, a);
            pass (ps, a);
        }
    }
    if (check && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done))) return false;    // block-uniform
    FF_STAMP (9)
    if (ng > 2) {                                    // second level: rows 0..17, one moment each
        __syncthreads ();
        if (row < 20) {
            const uint32_t k = min (row, (uint32_t) ICP_NMOM - 1u);
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t i = l + 16u * q;
                a[q] = (i < ng) ? (gl1 ? gl1[(size_t) k * ng + i] : s_l1[k][i]) : 0.0;
            }
            double r = row_tree8_d (a);
            if (l == 0 && row < ICP_NMOM) s_t[row] = r;
        }
    }
    __syncthreads ();
    FF_STAMP (10)
    if (tid < 64) {
        double t[ICP_NMOM];
#pragma unroll
        for (int k = 0; k < ICP_NMOM; ++k) t[k] = s_t[k];
        const double sw = t[0];
        // oracle orc_moments_finish: ONE division, the means by multiplication, fused multiply-adds (this wave is what the
        // block — in the chained form the whole grid — waits for: six double divisions and the unfused products measured
        // 0.5 us of the iteration at A; a division per lane with the quotients handed round as scalars measured slower than
        // six overlapping ones, 9.19 -> 9.24 us)
        const double rs = 1.0 / sw;
        double mf[3], mq[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mf[a] = t[1 + a] * rs; mq[a] = t[4 + a] * rs; }
        const double c2 = (double) p.c * (double) p.c;
        float S[11], means[8];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) S[3 * a + bb] = (float) (c2 * __builtin_fma (-t[4 + a], mf[bb], t[7 + 3 * a + bb]));
        S[9]  = (float) (c2 * (t[16] - __builtin_fma (t[3], mf[2], __builtin_fma (t[2], mf[1], t[1] * mf[0]))));
        S[10] = (float) (c2 * (t[17] - __builtin_fma (t[6], mq[2], __builtin_fma (t[5], mq[1], t[4] * mq[0]))));
        means[0] = (float) mf[0]; means[1] = (float) mf[1]; means[2] = (float) mf[2]; means[3] = 0.f;
        means[4] = (float) mq[0]; means[5] = (float) mq[1]; means[6] = (float) mq[2]; means[7] = 0.f;
        float Tk[8], Rk[9], Rkin[9];
        int iters = 0;
        FF_STAMP (11)
        // ROT = rotation solver, compile-time: a power-method kernel carries no SVD code (registers, instruction cache)
        if constexpr (ROT == 1) iters = icp_power_method_quad (S, means, Tk, p.power_mode, lane);
        else icp_svd_rotation (S, means, Rkin, Tk);
        FF_STAMP (12)
#ifdef ICP_DBG_STAMPS
        if (lane < 8 && p.dbg && ROT == 1) p.dbg[16 + lane] = icp_pm_stamps[lane];
#endif
        float Tprev[8], Rprev[9];
#pragma unroll
        for (int k = 0; k < 8; ++k) Tprev[k] = state_lane_f (sv, ICP_ST_DW (T) + k);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rprev[k] = state_lane_f (sv, ICP_ST_DW (R) + k);
        const uint32_t kprev = (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k));
        if constexpr (LEAN) {
            // the composition on every lane (the same instructions as on one): T is then in registers where the hook wants it
            float Tn[8], Rn[9];
            icp_compose_pure (Tprev, Rprev, Tk, Rkin, ROT != 1, Tn, Rn, Rk);
            const uint32_t done = (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) ? 1u : 0u;
            after (Tn);
            if (lane == 0) {
                res->done = done;
                if (direct) {
                    // the state image, dword for dword what fin_result_to_state would publish (pending = !done)
                    typedef float f4u __attribute__ ((ext_vector_type (4), aligned (4)));
                    typedef float f2u __attribute__ ((ext_vector_type (2), aligned (4)));
                    static_assert (sizeof (icp_reg_state) == 62 * 4 && ICP_ST_DW (means) == 46 && ICP_ST_DW (sum_w) == 54 && ICP_ST_DW (k) == 56, "state layout");
                    const unsigned long long swb = __builtin_bit_cast (unsigned long long, sw);
                    const float img[64] = {
                        Tn[0], Tn[1], Tn[2], Tn[3], Tn[4], Tn[5], Tn[6], Tn[7], Tk[0], Tk[1], Tk[2], Tk[3], Tk[4], Tk[5], Tk[6], Tk[7],
                        Rn[0], Rn[1], Rn[2], Rn[3], Rn[4], Rn[5], Rn[6], Rn[7], Rn[8], Rk[0], Rk[1], Rk[2], Rk[3], Rk[4], Rk[5], Rk[6], Rk[7], Rk[8],
                        S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8], S[9], S[10], 0.f,
                        means[0], means[1], means[2], means[3], means[4], means[5], means[6], means[7],
                        __uint_as_float ((uint32_t) swb), __uint_as_float ((uint32_t) (swb >> 32)),
                        __uint_as_float (kprev + 1u), __uint_as_float (done), __uint_as_float ((uint32_t) iters), __uint_as_float (done ? 0u : pending_unless_done),
                        0.f, 0.f, 0.f, 0.f };
                    auto store_image = [&] (icp_reg_state *to) {
                        float *dst = reinterpret_cast<float *> (to);
#pragma unroll
                        for (int k = 0; k < 15; ++k) *reinterpret_cast<f4u *> (dst + 4 * k) = f4u { img[4 * k], img[4 * k + 1], img[4 * k + 2], img[4 * k + 3] };
                        *reinterpret_cast<f2u *> (dst + 60) = f2u { img[60], img[61] };
                    };
                    store_image (direct);
                    if (prev_dst) {
                        float *dst = reinterpret_cast<float *> (prev_dst);
                        *reinterpret_cast<f4u *> (dst) = f4u { Tprev[0], Tprev[1], Tprev[2], Tprev[3] };
                        *reinterpret_cast<f4u *> (dst + 4) = f4u { Tprev[4], Tprev[5], Tprev[6], Tprev[7] };
                    }
                    if (mirror) {
                        if (done) {
                            if (final_dst) store_image (final_dst);
                            if (host_dst) { store_image (host_dst); __threadfence_system (); }
                            icp_mirror_store (mirror, ICP_MIRROR_WORD (p.epoch, kprev + 1u, 1u) | (host_dst ? ICP_MIRROR_FINAL : 0ull));
                            if (progress && p.run_flag) {
                                // the run is over: its later launches see the flag; a tracked sequence's next frame (held by k_gate on
                                // the other stream) may start — behind the user-visible state above (release at agent scope)
                                p.run_flag[blockIdx.y] = p.epoch;
                                if (p.track_seq) __hip_atomic_store (p.track_seq, p.seq_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        } else if (progress) icp_mirror_store (mirror, ICP_MIRROR_WORD (p.epoch, kprev + 1u, 0u));
                    }
                }
            }
        } else if (lane == 0) {
            float Tn[8], Rn[9];
            icp_compose_pure (Tprev, Rprev, Tk, Rkin, ROT != 1, Tn, Rn, Rk);
#pragma unroll
            for (int k = 0; k < 8; ++k) { res->T[k] = Tn[k]; res->Tk[k] = Tk[k]; res->means[k] = means[k]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { res->R[k] = Rn[k]; res->Rk[k] = Rk[k]; }
#pragma unroll
            for (int k = 0; k < 11; ++k) res->S[k] = S[k];
            res->sum_w = sw; res->pm_iters = (uint32_t) iters; res->k = kprev + 1u; res->pad0 = 0.f; res->pending = 0u;
            res->reserved0 = 0u; res->reserved1 = 0u;
            res->done = (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) ? 1u : 0u;
            if (prev_dst) {
#pragma unroll
                for (int k = 0; k < 8; ++k) prev_dst->T[k] = Tprev[k];
            }
        }
    }
    __syncthreads ();
    return true;
}

// cooperative publish: thread t of the block copies dword t (call with the whole first wave)
static __device__ __forceinline__ void fin_result_to_state (const icp_fin_result *res, icp_reg_state *st, uint32_t pending)
{
    static_assert (sizeof (icp_reg_state) / 4 <= 64, "the state is published by one wave");
    const uint32_t t = threadIdx.x;
    if (t < sizeof (icp_reg_state) / 4) {
        uint32_t v = reinterpret_cast<const uint32_t *> (res)[t];
        if (t == offsetof (icp_reg_state, pending) / 4) v = pending;
        reinterpret_cast<uint32_t *> (st)[t] = v;
    }
}

// CHAIN (fused mode only): launch j reads state slot j&1 and the moments buffer j&1, turns the previous
// iteration's moments into T in its prologue (every block redundantly; block 0 publishes the result in the
// other slot), searches, and leaves its own moments in the other buffer: ONE launch per ICP iteration.
// MINW = waves per SIMD the register allocation must leave room for: 2 (one block per CU: a single registration,
// nothing to hide latency behind, no spills) or 4 (two blocks per CU: batched registrations, +70 % throughput).
// LPQ = lanes per query = waves per block: 16 when the grid is at most one block per CU (more waves per SIMD to
// overlap the L2-cold loads), 8 when occupancy comes from the number of blocks.
// OWNER: the kernel is RBC construct step 1 instead (owner(x) = nearest representative of the FIXED point x: gM = F,
// no transform, stage 1 only, result to p.owner) — the same stage-1 code, pruning included, seeded with the
// representative of the point's own grid cell.
// TILE = representatives per LDS tile: 1024, or 256 for the dense variant at |R| <= 256 (batches of config 4): 22 KB instead of
// 47 KB of LDS per block and a register budget for 8 waves per SIMD — four blocks per CU instead of three.
// SINGLE: the launcher guarantees |R| <= TILE (the tile loop and everything multi-tile fold away).
// HOSTRUN (CHAIN only): the launch belongs to a host-driven checked run (run_ctl in icp_capi.hip) — progress words, the final state of a
// converged registration to the user-visible state and to host memory, the transform each search used (p.st_prev).  Fixed-length graphs
// (the metric's path) instantiate the kernel without any of it: what the publishing lane of block 0 carries in its prologue is on the
// path the whole grid waits for (measured with everything decided at run time: 8.58 -> 8.73 us per dispatch).
template <bool FUSED, bool CHAIN, int MINW, int LPQ, bool OWNER = false, int ROT = 1, int TILE = 1024, bool SINGLE = false, bool S2W = false, bool HOSTRUN = false>
__global__ __launch_bounds__ (64 * LPQ, TILE == 256 ? 8 : MINW) void k_search (const float *gM, const float *gR, icp_reg_state *gst, const double *gmom,
                                                              uint32_t m, uint32_t nr, uint32_t side, uint32_t tpr_magic,
                                                              uint32_t nb, uint32_t check_flags, icp_params p)
{
    // The first 14 dwords of the kernel arguments (everything the prologue's addresses need) are plain scalars so
    // that they arrive preloaded in SGPRs / in one scalar load; the rest of icp_params is fetched while the first
    // global loads are in flight.  gst = the state this launch reads (CHAIN: slot p.slot of every pair of slots),
    // gmom = the moments it turns into T first (CHAIN only: buffer p.slot).
    // check_flags: bit 0 = convergence checks on; bit 1 (CHAIN) = first launch of a chain: gst is the user-visible
    // state array (stride 1) instead of a pair of slots; bit 3 = store the matched / transformed points too (fused
    // mode needs them only after the last iteration of a graph; the reference-order kernels read them every time).
    constexpr uint32_t KT = (uint32_t) TILE;
    // MASKED (dense variant, several small tiles): the set of tiles a block needs is decided ONCE, before anything is staged —
    // every query tests the boxes of all tiles against its seed bound, the block ORs the answers — and only those tiles
    // are staged and scanned, in ascending order, without a vote per tile.
    constexpr bool MASKED = (TILE == 256) && !SINGLE && (MINW == 4);
    if constexpr (SINGLE) __builtin_assume (nr <= KT);
    if constexpr (MASKED) __builtin_assume (nr > KT && nr <= 32u * KT);
    const uint32_t b = blockIdx.y, check = check_flags & 1u;
    icp_reg_state *st = (CHAIN && !(check_flags & 2u)) ? gst + (size_t) b * 2 : gst + b;
#ifdef ICP_DBG_STAMPS
    { const uint32_t tid = threadIdx.x; unsigned long long t_; asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
      if (tid == 0 && p.dbg) p.dbg[(size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8] = t_; }
#endif

    // representatives of the current tile, pair-interleaved for packed fp32 math:
    //   pair P = reps (2P, 2P+1) -> 3 float4: [x0 x1 y0 y1] [z0 z1 r0 r1] [g0 g1 b0 b1]
    // MASKED keeps TWO tile buffers: buffer 1 holds the block's HOME tile — the tile of the representative sampled from the grid cell
    // of the block's first query —, staged in the prologue with everything else that does not depend on T; buffer 0 is for the other
    // tiles the block turns out to need.  A block's 64 neighbouring queries mostly need the home tile alone: then nothing is fetched
    // between the tile masks and the scan, the seed representative and the winner's (offset, size) come from LDS too — three dependent
    // memory round trips fewer on a path that is a chain of them (|F| = 65536: one wave of 1024 blocks, latency-bound; stamps in
    // profiles/r03_stamps_dense.txt).
    constexpr uint32_t NBUF = MASKED ? 2u : 1u, PB = 3u * KT / 2u, BB = 2u * (KT / 16u);
    __shared__ float4 s_pair[NBUF * PB];
    __shared__ uint2 s_on[(MASKED && OWNER) ? 1 : KT];            // (offset, size) of the lists of the tile's representatives (MASKED: of the home tile's)
    __shared__ uint32_t s_tmask;                     // MASKED: tiles some query of the block needs
    __shared__ float4 s_box[NBUF * BB];              // (lo, hi) of the tile's groups of 2 * LPQ representatives
    __shared__ float4 s_tbox[(MINW == 4 && !SINGLE) ? 2 * 32 : 2];      // (lo, hi) of every tile (multi-tile sets: |R| <= 32768)
    __shared__ float s_w[64];
    __shared__ float4 s_qc[64];                      // query hand-in: (r, g, b, pruning seed); s_qa carries (q', index)
    __shared__ float4 s_qa[64];                      // per-query hand-off to the finishing wave: (q', distance)
    __shared__ uint4 s_qb[64];                       //   (winner position or representative, representative, flags, query index)
    __shared__ double s_mom[FUSED ? ICP_NMOM : 1][64];
    __shared__ icp_fin_result s_fin;
    __shared__ double s_l1[CHAIN ? ICP_NMOM : 1][CHAIN ? 32 : 1];
    __shared__ double s_t[ICP_NMOM];

    // A wave serves KS_QPW queries end to end, KS_SPLIT (= LPQ) lanes per query: lane ss of a query takes the
    // representative pairs = ss mod LPQ in stage 1 and the list positions = ss mod LPQ in stage 2, and the
    // query's winner is an LPQ-lane DPP reduction — no cross-wave exchange inside the two stages (block barriers only around the LDS hand-overs).
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t slice = __builtin_amdgcn_readfirstlane (tid >> 6);
    const uint32_t qe = slice * KS_QPW + lane / KS_SPLIT, ss = lane & (KS_SPLIT - 1u);
    // reference-order mode: the 64 even (or odd) positions of one 128-query group (a closed sub-tree of
    // the weight reduction); fused mode: an 8 x 8 tile of the landmark grid (spatially coherent lists)
    // One wave (the last; in the chained variant wave 0, which holds the new T in registers the moment it exists and hands the
    // transformed queries over before the barrier that ends the finalize) prepares the block's 64 queries —
    // lane e = query e: index, load, transform, pruning seed — and hands them to the lanes of each query through LDS;
    // the other 15 (7) waves neither compute the index nor load / transform the same point LPQ times over.
    const bool qwave = CHAIN ? slice == 0u : slice == KS_SPLIT - 1u;
    // (the one-block-per-CU variants only: measured 9.63 -> 9.43 us per iteration at A; the dense variant runs several blocks
    // per CU over grids of thousands and measured 0 ... 4 % slower with it)
    // (dense variant: bit 6 of check_flags, set for a single large registration — there the bands halve the fabric-side traffic,
    // C 171 -> 86 MB per launch against 75.6 MB algorithmic, at the same speed; batched grids measured 2 % slower with them)
    // OWNER_LISTS (the owner search of the latency-bound sizes, buildRBC in two launches: see k_place_lists): blocks of 64 CONSECUTIVE
    // fixed points — the stable placement ranks a point among the earlier points of its owner, and a block's (owner, count) list
    // is a piece of exactly that count —, the representatives gathered straight from F (getReps' sampling rule: the launch does
    // not wait for a kernel that writes R; block 0 writes R and rep_src on the side).
    constexpr bool OWNER_LISTS = OWNER && MINW == 2;
    const uint32_t tile_id = OWNER_LISTS ? blockIdx.x : (FUSED && (MINW == 2 || (check_flags & 64u))) ? ks_tile_of_block (blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t iq = OWNER_LISTS ? blockIdx.x * 64u + lane :
                        FUSED ? fused_query_index (m, side, tpr_magic, tile_id, lane)
                              : (blockIdx.x >> 1) * 128u + 2u * lane + (blockIdx.x & 1u);

    const float4 *M4 = reinterpret_cast<const float4 *> (gM + (size_t) b * m * 8);
    const float4 *R4 = reinterpret_cast<const float4 *> (gR + (size_t) b * nr * 8);

    // every independent global load of the prologue is issued before anything waits: the state (one vector load,
    // lane j = dword j: scalar loads of T would queue behind the waits of the vector loads), the first tile of
    // representatives (+ list offsets / sizes), the query point (clamped address, selected afterwards)
    uint32_t sv = OWNER ? 0u : state_load_lanes (st);
    // HOSTRUN: the run this launch belongs to may have converged already (its flag holds the run's epoch): such a launch leaves below
    // without a single store — the next tracked frame may be running on the other stream, in the same state slots and moment buffers
    uint32_t run_over = 0u;
    if constexpr (CHAIN && HOSTRUN) { if (p.run_flag) run_over = (p.run_flag[b] == p.epoch) ? 1u : 0u; }
    if constexpr (CHAIN) {
        // bit 4 of check_flags (first launch of a chain): the run starts from the identity transform — what k_reset_state
        // would have left in the state (T = Tk = (0,0,0,1 | 0,0,0,1), R = Rk = I, S = means = sum_w = 0, k = done = 0),
        // without a launch of its own
        static_assert (ICP_ST_DW (T) == 0 && ICP_ST_DW (Tk) == 8 && ICP_ST_DW (R) == 16 && ICP_ST_DW (Rk) == 25 && ICP_ST_DW (reserved0) == 60, "state layout");
        if (check_flags & 16u) {
            constexpr unsigned long long ones = (1ull << 3) | (1ull << 7) | (1ull << 11) | (1ull << 15) | (1ull << 16) | (1ull << 20) | (1ull << 24) |
                                                (1ull << 25) | (1ull << 29) | (1ull << 33);
            if (lane < ICP_ST_DW (reserved0)) sv = ((ones >> lane) & 1ull) ? 0x3F800000u : 0u;
        }
    }
    // chained variant: the previous iteration's block moments (first tree level of this block's finalize) travel with
    // the other prologue loads
    double ma0[8];
    if constexpr (CHAIN) fused_moment_loads<64 * LPQ> (gmom + (size_t) b * 2 * ICP_NMOM * nb, nb, 0u, ma0);
    float *s_pairf = reinterpret_cast<float *> (s_pair);
    const uint32_t tn0 = min (KT, nr);
    float4 rg[2], rc[2]; uint2 ron[2];
    uint32_t ht = 0u, tnH = 0u;                      // MASKED: home tile of the block and its size (block-uniform)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        if (MASKED && u == 0) continue;
        rg[u] = make_float4 (0.f, 0.f, 0.f, 0.f); rc[u] = rg[u];
        if constexpr (OWNER_LISTS) {
            if (k < tn0) {
                const uint32_t src = rep_src_index (p, k);
                rg[u] = M4[2 * (size_t) src]; rc[u] = M4[2 * (size_t) src + 1];
                if (blockIdx.x == 0) {               // getReps (a1): R and rep_src, written once, read by the searches that follow
                    float4 *Rw = reinterpret_cast<float4 *> (p.R + (size_t) b * nr * 8);
                    Rw[2 * (size_t) k] = rg[u]; Rw[2 * (size_t) k + 1] = rc[u];
                    p.rep_src[(size_t) b * nr + k] = src;
                }
            }
        } else if (!MASKED && k < tn0) { rg[u] = R4[2 * (size_t) k]; rc[u] = R4[2 * (size_t) k + 1]; }
    }
    const uint32_t ic = min (iq, m - 1u);
    float4 mg = make_float4 (0.f, 0.f, 0.f, 1.f), mc = mg;
    if (qwave) { mg = M4[2 * (size_t) ic]; mc = M4[2 * (size_t) ic + 1]; }
    if constexpr (MASKED) {                          // the home tile: of the cell of the block's first query (block-uniform; no division)
        const uint32_t i0 = (uint32_t) __builtin_amdgcn_readfirstlane ((int) min (FUSED ? fused_query_index (m, side, tpr_magic, tile_id, 0u) : (blockIdx.x >> 1) * 128u + (blockIdx.x & 1u), m - 1u));
        const uint32_t cell = p.side_magic ? cell_rep_of (p, i0) : 0u;
        ht = min (cell, nr - 1u) / KT; tnH = min (KT, nr - ht * KT);
        if (ICP_HOME_MODE != 3 && tid < tnH) { rg[0] = R4[2 * (size_t) (ht * KT + tid)]; rc[0] = R4[2 * (size_t) (ht * KT + tid) + 1]; }
    }
    // (list offsets / sizes: their base pointers come with the second batch of kernel arguments)
    const uint32_t *gO = p.O + (size_t) b * nr, *gN = p.N + (size_t) b * nr;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        ron[u] = make_uint2 (0u, 0u);
        if (!OWNER && !MASKED && k < tn0) ron[u] = make_uint2 (gO[k], gN[k]);
        if (!OWNER && MASKED && u == 0 && tid < tnH) ron[0] = make_uint2 (gO[ht * KT + tid], gN[ht * KT + tid]);
    }
    // seed of the stage-1 pruning bound: this query's nearest representative of the previous search (any index < nr
    // is a valid seed; the buffer starts zeroed)
    // Pruning pays where stage 1 is throughput-bound: the dense variant (MINW == 4: several blocks per CU, or a
    // large representative set — see icp_launch_search); a single small registration is latency-bound and keeps
    // the branch-free loop (compile-time: the pruning code costs 0.25 us there even when it is switched off).
    constexpr bool PRUNE = ICP_S1_SEED && MINW == 4;
    const bool prune = PRUNE && p.a > 0.f;
    const uint32_t gt_lg1 = p.gtile;                 // 0: strip groups; 1 + log2 (nrx / 4): 4 x 4 tile groups (k_rep_boxes)
    // A registration's FIRST search (k == 0: ICP::buildRBC / a reset came before it) has no previous search of its own: whatever
    // p.rid holds then belongs to another registration (legal, but a converged neighbour's answer would flatter a benchmark
    // that re-registers one pair, and a stale one prunes nothing).  It is seeded like the owner search: with the representative
    // sampled from the query's own grid cell — a moving frame starts near the fixed one (frame-to-frame registration).
    // check_flags bit 5 (ICP_AMD_WARM_SEED=1, diagnostics): always the previous search's answer.
    uint32_t seed = 0u, seed_cell = 0xFFFFFFFFu;
    if (qwave && prune && p.side_magic) seed_cell = cell_rep_of (p, ic);     // the representative sampled from the point's own cell
    if constexpr (OWNER) seed = seed_cell == 0xFFFFFFFFu ? 0u : seed_cell;
    else if (qwave && prune) {
        seed = p.rid[(size_t) b * m + ic];           // (selected against seed_cell below, once the state has arrived)
        if ((check_flags & 32u) || seed_cell == 0xFFFFFFFFu) seed_cell = 0xFFFFFFFFu;
    }
    // (lo, hi) boxes of the groups of 2 * LPQ representatives: the 16-boxes (LPQ = 8) or the 32-boxes behind them
    const float4 *GBt = p.GB + (size_t) b * 2 * (p.n16 + p.n1k) + (KS_SPLIT == 8 ? 0u : 2u * p.n16);
    const uint32_t nbox0 = 2u * ((tn0 + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
    float4 boxv = make_float4 (0.f, 0.f, 0.f, 0.f);
    if (!MASKED && prune && tid < nbox0) boxv = GBt[tid];
    const uint32_t nboxH = 2u * ((tnH + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
    if (MASKED && ICP_HOME_MODE != 3 && prune && tid < nboxH) boxv = GBt[2u * (ht * KT / (2u * KS_SPLIT)) + tid];
    float T[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = state_lane_f (sv, ICP_ST_DW (T) + k);
    if constexpr (PRUNE && !OWNER) {
        if (seed_cell != 0xFFFFFFFFu && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k)) == 0) seed = seed_cell;    // first search of a registration
    }
    icp_reg_state *sout = CHAIN ? p.cst + (size_t) b * 2 + (p.slot ^ 1u) : st;
    if constexpr (!OWNER && !CHAIN) {
        // host-driven checked runs (icp_run; see run_ctl in icp_capi.hip), separate launches: the search of iteration j tells the host that j
        // iterations are through and whether the last one converged — one 8-byte store into host memory that nothing here waits for
        if (p.hmirror && blockIdx.x == 0 && tid == 0)
            icp_mirror_store (p.hmirror + b, ICP_MIRROR_WORD (p.epoch, (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k)),
                                                              __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done))));
    }
    if constexpr (CHAIN && HOSTRUN) { if (run_over) return; }
    if (!OWNER && check && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done))) {    // converged earlier
        if constexpr (CHAIN) {                       // carry the state forward
            if (blockIdx.x == 0 && tid < sizeof (icp_reg_state) / 4) reinterpret_cast<uint32_t *> (sout)[tid] = sv;
        }
        return;
    }
    if (iq >= m) { mg = make_float4 (0.f, 0.f, 0.f, 1.f); mc = mg; }
    const float4 *XQ4 = reinterpret_cast<const float4 *> (p.XQ + (size_t) b * m * 8);
    const char *XQb = reinterpret_cast<const char *> (XQ4);
    // the first tile of representatives goes to LDS now: nothing in it depends on T, and in the chained variant the
    // writes and their barrier disappear behind the power method of wave 0
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        if (!MASKED && k < tn0) {
            float *dst = s_pairf + (k >> 1) * 12u + (k & 1u);
            dst[0] = rg[u].x; dst[2] = rg[u].y; dst[4] = rg[u].z; dst[6] = rc[u].x; dst[8] = rc[u].y; dst[10] = rc[u].z;
            if constexpr (!OWNER && !MASKED) s_on[k] = ron[u];
        }
    }
    if (!MASKED && prune && tid < nbox0) s_box[tid] = boxv;
    auto home_to_lds = [&] () {                      // the home tile -> buffer 1
        if (tid < tnH) {
            float *dst = s_pairf + PB * 4u + (tid >> 1) * 12u + (tid & 1u);
            dst[0] = rg[0].x; dst[2] = rg[0].y; dst[4] = rg[0].z; dst[6] = rc[0].x; dst[8] = rc[0].y; dst[10] = rc[0].z;
        }
        if (prune && tid < nboxH) s_box[BB + tid] = boxv;
    };
    if constexpr (MASKED) {
        if (ICP_HOME_MODE == 1) home_to_lds ();
        if constexpr (!OWNER) { if (tid < tnH) s_on[tid] = ron[0]; }
    }
    if (MASKED && tid == 0) s_tmask = 0u;
    if constexpr (MINW == 4 && !SINGLE) {            // the boxes of all tiles: a tile is tested before it is staged (stage 1 below)
        if (prune && nr > KT && tid < 2u * p.n1k) s_tbox[tid] = p.GB[(size_t) b * 2 * (p.n16 + p.n1k) + 2u * p.n16 + tid];
    }
    bool handed = false;                             // the queries are in LDS already (chained variant, see below)
    if constexpr (CHAIN) {
        const bool pending = __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (pending)) != 0;
        if (pending) {
            // wave 0 (= the query wave): T -> the block's 64 transformed queries -> LDS, inside the finalize, before its barrier
            auto hand_over = [&] (const float *Tn) {
                float tx, ty, tz;
                icp_transform_point (Tn, mg.x, mg.y, mg.z, tx, ty, tz);
                s_qa[lane] = make_float4 (tx, ty, tz, __uint_as_float (iq));
                s_qc[lane] = make_float4 (mc.x, mc.y, mc.z, __uint_as_float (seed));
            };
            fused_finalize_block<32, 64 * LPQ, ROT, true> (p, gmom + (size_t) b * 2 * ICP_NMOM * nb, nb, 0u, sv, ma0, &s_fin, s_l1, s_t, nullptr,
                                                           blockIdx.x == 0 ? sout : nullptr, hand_over, 1u,
                                                           (HOSTRUN && blockIdx.x == 0 && p.hmirror) ? p.hmirror + b : nullptr, HOSTRUN, HOSTRUN ? p.st + b : nullptr,
                                                           (HOSTRUN && p.hstate) ? p.hstate + b : nullptr, (HOSTRUN && p.st_prev) ? p.st_prev + b : nullptr);
            handed = true;
            KS_STAMP (9)
            if (s_fin.done) return;
        } else if (blockIdx.x == 0 && tid < sizeof (icp_reg_state) / 4) {
            reinterpret_cast<uint32_t *> (sout)[tid] = (tid == offsetof (icp_reg_state, pending) / 4) ? 1u : sv;
        }
    }
    if (qwave && !handed) {
        float tx = mg.x, ty = mg.y, tz = mg.z;
        if constexpr (!OWNER) icp_transform_point (T, mg.x, mg.y, mg.z, tx, ty, tz);
        s_qa[lane] = make_float4 (tx, ty, tz, __uint_as_float (iq));
        s_qc[lane] = make_float4 (mc.x, mc.y, mc.z, __uint_as_float (seed));
    }
    float qx = 0.f, qy = 0.f, qz = 0.f, qr = 0.f, qg = 0.f, qb = 0.f;
    uint32_t i = 0u; bool valid = false;
    const float alpha = p.a;
    KS_STAMP (0)

    // ---- stage 1: nearest representative, two representatives per packed instruction ----
    float best = __builtin_inff (), s1_lim = __builtin_inff (); uint32_t bid = 0xFFFFFFFFu;
    // coarse pass of the pruning over the groups of the tile in LDS (s_box): bit t of the result = this lane's t-th
    // group (ss, ss + LPQ, ..) may hold a representative nearer than `lim`.  The lower bound applies the metric's own
    // operations to the per-axis distances to the group's bounding box; every operation is monotone under
    // round-to-nearest, so bound <= geo <= d for every member, and a group whose bound is not below `lim` cannot hold the winner.
    auto coarse_pass = [&] (uint32_t tn_, float lim_, uint32_t hb_ = 0u) -> uint32_t {
        const uint32_t ngt = (((tn_ + 1u) >> 1) + KS_SPLIT - 1u) / KS_SPLIT;
        uint32_t cm = 0u;
        auto test = [&] (uint32_t t, uint32_t g) {
            const float4 lo = s_box[hb_ * BB + 2 * g], hi = s_box[hb_ * BB + 2 * g + 1];
            const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
            const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
            const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
            const float bound = __builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex));
            if (bound < lim_) cm |= 1u << t;
        };
        if constexpr (TILE == 256) {                 // at most two trips per lane: not unrolled (the 64-register budget of this variant)
#pragma unroll 1
            for (uint32_t t = 0, g = ss; g < ngt; ++t, g += KS_SPLIT) test (t, g);
        } else
            for (uint32_t t = 0, g = ss; g < ngt; ++t, g += KS_SPLIT) test (t, g);
        return cm;
    };
    // fine pass of the pruning over the tile in LDS (first representative t0_, npair_ pairs): the groups whose bit is set in
    // cmask_ for some query of the wave, full evaluation.
    auto fine_pass = [&] (uint32_t t0_, uint32_t npair_, uint32_t cmask_, uint32_t hb_ = 0u) {
        const uint32_t ngt_ = (npair_ + KS_SPLIT - 1u) / KS_SPLIT;
        const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
        const float2v va = { alpha, alpha };
        // fine pass: the groups some query of the wave still needs, in ascending order (a lane's pairs must ascend
        // for the tie rule), full evaluation
        // (scalar control flow: the lane ballot of trip t is folded over the wave's queries into one bit per group
        // and only the set bits are visited — a taken branch costs more than the arithmetic it guards)
        for (uint32_t t = 0; t * KS_SPLIT < ngt_; ++t) {
            unsigned long long bal = __ballot ((cmask_ >> t) & 1u);
            if (bal == 0ull) continue;
            bal |= bal >> 32; bal |= bal >> 16;
            if (KS_SPLIT == 8) bal |= bal >> 8;
            uint32_t need = (uint32_t) bal & ((1u << KS_SPLIT) - 1u);
            while (need) {
                const uint32_t sg = (uint32_t) __builtin_ctz (need);
                need &= need - 1u;
                // pair of this lane in group gl of the LDS tile.  Strips: the group's 16 consecutive representatives.
                // Tiles (LPQ == 8): lane ss holds row ss >> 1, columns 2 (ss & 1) and + 1 of the 4 x 4 tile; groups are
                // visited in ascending (tile row, tile column) order, so every lane's pairs still ascend in index —
                // what the tie rule (strict '<' keeps a lane's lowest index) relies on.
                const uint32_t gl = sg + KS_SPLIT * t;
                const uint32_t P = (KS_SPLIT == 8 && gt_lg1) ? (((4u * (gl >> (gt_lg1 - 1u)) + (ss >> 1)) << gt_lg1) + 2u * (gl & ((1u << (gt_lg1 - 1u)) - 1u)) + (ss & 1u))
                                                              : gl * KS_SPLIT + ss;
                if (P < npair_) {
                    const uint32_t P3 = hb_ * PB + __umul24 (P, 3u);       // (24-bit multiply: full rate; a 32-bit v_mul_lo costs four issue slots)
                    float4 A = s_pair[P3], B = s_pair[P3 + 1], C = s_pair[P3 + 2];
                    float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                    float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                    float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                    float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                    float2v d = __builtin_elementwise_fma (va, pho, geo);
                    const uint32_t r0 = t0_ + 2u * P;
                    if (d.x < best) { best = d.x; bid = r0; }
                    if (d.y < best) { best = d.y; bid = r0 + 1u; }
                }
            }
        }
    };
    if constexpr (MASKED) {
        __syncthreads ();                            // the queries (s_qa / s_qc), the tile boxes, s_tmask = 0
        {
            const float4 a4 = s_qa[qe], c4 = s_qc[qe];
            qx = a4.x; qy = a4.y; qz = a4.z; i = __float_as_uint (a4.w); valid = i < m;
            qr = c4.x; qg = c4.y; qb = c4.z; seed = __float_as_uint (c4.w);
        }
        const uint32_t ntile = (nr + KT - 1u) / KT;  // <= 32
        uint32_t qmask = 0xFFFFFFFFu >> (32u - ntile);
        if (prune) {
            // the seed bound: the seed representative from the home tile in LDS, from global memory where it lies outside
            seed = min (seed, nr - 1u);
            float sx, sy, sz, sr, sg, sb;
            if (ICP_HOME_MODE == 1 && seed / KT == ht) {
                const float *sp = s_pairf + PB * 4u + ((seed - ht * KT) >> 1) * 12u + (seed & 1u);
                sx = sp[0]; sy = sp[2]; sz = sp[4]; sr = sp[6]; sg = sp[8]; sb = sp[10];
                // (keeps the compiler from merging this with the global path below into flat loads)
                asm volatile ("" : "+v"(sx), "+v"(sy), "+v"(sz), "+v"(sr), "+v"(sg), "+v"(sb));
            } else {
                const float4 g = R4[2 * (size_t) seed], c = R4[2 * (size_t) seed + 1];
                sx = g.x; sy = g.y; sz = g.z; sr = c.x; sg = c.y; sb = c.z;
            }
            const float b0 = icp_metric8 (qx, qy, qz, qr, qg, qb, sx, sy, sz, sr, sg, sb, alpha);
            if (b0 >= 0.f && b0 < __builtin_inff ()) s1_lim = __uint_as_float (__float_as_uint (b0) + 1u);     // next float up
            // tiles this query can find a nearer representative in: lane ss tests the tiles ss, ss + LPQ, ..; OR over the lanes
            uint32_t tm = 0u;
            for (uint32_t t = ss; t < ntile; t += KS_SPLIT) {
                const float4 lo = s_tbox[2u * t], hi = s_tbox[2u * t + 1u];
                const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
                const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
                const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
                if (__builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex)) < s1_lim) tm |= 1u << t;
            }
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0x141, 0xF, 0xF, true);     // row_half_mirror
            qmask = tm;
        }
        {   // the block's union: OR over the wave (8 queries: one per half row), one LDS atomic per wave
            uint32_t wm = qmask | (uint32_t) __builtin_amdgcn_update_dpp (0, (int) qmask, 0x140, 0xF, 0xF, true);      // row_mirror: both half rows
            wm = (uint32_t) __builtin_amdgcn_readlane ((int) wm, 0) | (uint32_t) __builtin_amdgcn_readlane ((int) wm, 16) |
                 (uint32_t) __builtin_amdgcn_readlane ((int) wm, 32) | (uint32_t) __builtin_amdgcn_readlane ((int) wm, 48);
            if (lane == 0) atomicOr (&s_tmask, wm);
        }
        if (ICP_HOME_MODE == 2) home_to_lds ();
        __syncthreads ();
        KS_STAMP (10)
        uint32_t bm = s_tmask;                       // block-uniform
        bool used0 = false;                          // buffer 0 holds a tile some wave may still be scanning
        while (bm) {                                 // ascending: the tie rule needs every lane's representatives to ascend
            const uint32_t tl = (uint32_t) __builtin_ctz (bm);
            bm &= bm - 1u;
            const uint32_t t0 = tl * KT, tn = min (KT, nr - t0), npair = (tn + 1u) >> 1;
            const uint32_t hb = (ICP_HOME_MODE != 3 && tl == ht) ? 1u : 0u;    // the home tile is there already: no fetch, no barrier
            if (!hb) {
                if (used0) __syncthreads ();         // every wave is done with the previous tile of buffer 0
                used0 = true;
                if (prune) {
                    const uint32_t nbx = 2u * ((tn + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
                    if (tid < nbx) s_box[tid] = GBt[2u * (t0 / (2u * KS_SPLIT)) + tid];
                }
                if (tid < tn) {
                    const float4 g = R4[2 * (size_t) (t0 + tid)], c = R4[2 * (size_t) (t0 + tid) + 1];
                    float *dst = s_pairf + (tid >> 1) * 12u + (tid & 1u);
                    dst[0] = g.x; dst[2] = g.y; dst[4] = g.z; dst[6] = c.x; dst[8] = c.y; dst[10] = c.z;
                }
                __syncthreads ();
            }
            if (prune) {
                const bool mine = ((qmask >> tl) & 1u) != 0u;
                uint32_t cmask = 0u;
                if (__ballot (mine)) cmask = coarse_pass (tn, mine ? s1_lim : -__builtin_inff (), hb);
                fine_pass (t0, npair, cmask, hb);
                s1_lim = fminf (s1_lim, ks_grp_min_f<KS_SPLIT> (best));
            } else {
                const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
                const float2v va = { alpha, alpha };
                for (uint32_t P = ss; P < npair; P += KS_SPLIT) {
                    float4 A = s_pair[hb * PB + 3 * P], B = s_pair[hb * PB + 3 * P + 1], C = s_pair[hb * PB + 3 * P + 2];
                    float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                    float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                    float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                    float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                    float2v d = __builtin_elementwise_fma (va, pho, geo);
                    const uint32_t r0 = t0 + 2u * P;
                    if (d.x < best) { best = d.x; bid = r0; }
                    if (d.y < best) { best = d.y; bid = r0 + 1u; }
                }
            }
        }
    } else
    for (uint32_t t0 = 0; t0 < nr; t0 += KT) {
        const uint32_t tn = min (KT, nr - t0);
        uint32_t cmask = 0u;
        // several tiles: the box of the whole tile first (same bound as for a group) — a query far from the tile skips its
        // 64 group tests, and a tile no query of the block is near is neither tested further nor staged (at |R| = 4096 a
        // block's 64 neighbouring queries need one, seldom two, of the four tiles)
        auto tile_near = [&] (float lim_) -> bool {
            if (SINGLE || nr <= KT) return true;
            const float4 lo = s_tbox[2u * (t0 / ICP_TBOX)], hi = s_tbox[2u * (t0 / ICP_TBOX) + 1u];     // (staged in the prologue)
            const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
            const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
            const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
            return __builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex)) < lim_;
        };
        if (t0) {                                    // further tiles (nr > KT)
            __syncthreads ();
            if (prune) {
                const bool near = tile_near (s1_lim);
                if (!__syncthreads_or (near)) continue;
                const uint32_t nbx = 2u * ((tn + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
                for (uint32_t k = tid; k < nbx; k += 64 * KS_SPLIT) s_box[k] = GBt[2u * (t0 / (2u * KS_SPLIT)) + k];
                __syncthreads ();
                if (__ballot (near)) cmask = coarse_pass (tn, s1_lim);
                if (!__syncthreads_or (cmask != 0u)) continue;
            }
            for (uint32_t k = tid; k < tn; k += 64 * KS_SPLIT) {
                float4 g = R4[2 * (size_t) (t0 + k)], c = R4[2 * (size_t) (t0 + k) + 1];
                float *dst = s_pairf + (k >> 1) * 12u + (k & 1u);
                dst[0] = g.x; dst[2] = g.y; dst[4] = g.z; dst[6] = c.x; dst[8] = c.y; dst[10] = c.z;
                if constexpr (!OWNER) s_on[k] = make_uint2 (gO[t0 + k], gN[t0 + k]);
            }
        }
        if ((tn & 1u) && tid == 0) {                 // odd tile (nr == 1): the pad slot never wins (NaN distance)
            float *dst = s_pairf + (tn >> 1) * 12u + 1u;
            const float qnan = __builtin_nanf ("");
            dst[0] = qnan; dst[2] = qnan; dst[4] = qnan; dst[6] = qnan; dst[8] = qnan; dst[10] = qnan;
        }
        // (chained variant, queries handed over inside the finalize: its closing barrier already stands behind every LDS write of
        // the prologue — representatives, list headers, boxes, queries —; a second one here would only be waited for)
        if (!(CHAIN && handed && t0 == 0u && !(tn & 1u))) __syncthreads ();
        if (t0 == 0) {                               // the query prepared by the query wave
            const float4 a4 = s_qa[qe], c4 = s_qc[qe];
            qx = a4.x; qy = a4.y; qz = a4.z; i = __float_as_uint (a4.w); valid = i < m;
            qr = c4.x; qg = c4.y; qb = c4.z; seed = __float_as_uint (c4.w);
        }
        KS_STAMP (1)
        const uint32_t npair = (tn + 1u) >> 1;
        const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
        const float2v va = { alpha, alpha };
        // a lane's pairs ascend (P = ss, ss+8, ..) and an update needs a strict '<', so each lane keeps its lowest
        // index among equal distances; the group reduction below then takes the lowest index overall.
        //
        // Exact pruning.  d = fma (a, pho, geo) >= geo for a > 0, so a pair whose two geo terms are not below `lim`
        // cannot hold the nearest representative when lim <= max (own best, a known upper bound of the query's
        // minimum): the photometric half, the third LDS read and the compare / select chain are skipped when no lane
        // of the wave needs them (wave-uniform branch; the wave's queries are neighbours).  The upper bound is the
        // distance to the seed (the previous search's nearest representative), bumped by one ulp so that a plain '<'
        // keeps every representative that could tie with it.
        float lim = __builtin_inff ();
        if (t0 == 0 && prune) {
            seed = min (seed, nr - 1u);
            float sx, sy, sz, sr, sg, sb;
            if (MINW == 2 || nr <= KT) {        // one tile: the seed is in LDS (MINW == 2: always, see icp_launch_search)
                const float *sp = s_pairf + (seed >> 1) * 12u + (seed & 1u);
                sx = sp[0]; sy = sp[2]; sz = sp[4]; sr = sp[6]; sg = sp[8]; sb = sp[10];
                // (keeps the compiler from merging this with the global path below into flat loads)
                asm volatile ("" : "+v"(sx), "+v"(sy), "+v"(sz), "+v"(sr), "+v"(sg), "+v"(sb));
            } else {
                const float4 g = R4[2 * (size_t) seed], c = R4[2 * (size_t) seed + 1];
                sx = g.x; sy = g.y; sz = g.z; sr = c.x; sg = c.y; sb = c.z;
            }
            const float b0 = icp_metric8 (qx, qy, qz, qr, qg, qb, sx, sy, sz, sr, sg, sb, alpha);
            if (b0 >= 0.f && b0 < __builtin_inff ()) lim = __uint_as_float (__float_as_uint (b0) + 1u);     // next float up
            s1_lim = lim;
        } else if (prune) lim = s1_lim;
        if (prune) {
            // coarse pass: a group = the 2 * LPQ representatives of one trip of the query's lanes; lane ss tests the
            // groups ss, ss + LPQ, ..  (further tiles: done above, before the tile was staged)
            if (t0 == 0 && __ballot (tile_near (lim))) cmask = coarse_pass (tn, lim);
            fine_pass (t0, npair, cmask);
            s1_lim = fminf (lim, ks_grp_min_f<KS_SPLIT> (best));
        } else {
#pragma unroll 8
            for (uint32_t P = ss; P < npair; P += KS_SPLIT) {
                float4 A = s_pair[3 * P], B = s_pair[3 * P + 1], C = s_pair[3 * P + 2];
                float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                float2v d = __builtin_elementwise_fma (va, pho, geo);
                const uint32_t r0 = t0 + 2u * P;
                if (d.x < best) { best = d.x; bid = r0; }
                if (d.y < best) { best = d.y; bid = r0 + 1u; }
            }
        }
    }
    KS_KEEP (best, bid)
    KS_STAMP (2)
    const float dr = ks_grp_min_f<KS_SPLIT> (best);           // the query's nearest representative: smallest distance,
    uint32_t rstar = ks_grp_min_u<KS_SPLIT> (best == dr ? bid : 0xFFFFFFFFu);     // ties -> lowest index
    if (rstar == 0xFFFFFFFFu) rstar = 0u;            // every distance inf / NaN: representative 0, as the serial scan would
    if constexpr (OWNER_LISTS) {
        // the block's 64 owners meet in LDS; wave 0 (lane e = point 64 blockIdx.x + e) stores them in one coalesced row, ranks every
        // point among the earlier points of the block with the same owner (one ballot per distinct owner: neighbours share a
        // handful) and leaves the block's (owner, count) list: k_place_lists needs nothing else to place the points
        if (ss == 0u) s_qb[qe] = make_uint4 (rstar, valid ? 1u : 0u, 0u, 0u);
        __syncthreads ();
        if (slice != 0u) return;
        const uint4 e4 = s_qb[lane];
        const bool v = e4.y != 0u;
        const uint32_t own = v ? e4.x : 0xFFFFFFFFu, ip = blockIdx.x * 64u + lane;
        uint32_t rank = 0u, kk = 0u;
        uint2 *bl = p.blist + ((size_t) b * nb + blockIdx.x) * 64u;
        for (unsigned long long rem = __ballot (v); rem; ++kk) {
            const uint32_t o = (uint32_t) __builtin_amdgcn_readlane ((int) own, (int) __builtin_ctzll (rem));
            const unsigned long long same = __ballot (own == o);     // (an owner is < nr: never the marker of an invalid lane)
            if (own == o) rank = (uint32_t) __builtin_popcountll (same & ((1ull << lane) - 1ull));
            if (lane == 0) bl[kk] = make_uint2 (o, (uint32_t) __builtin_popcountll (same));
            rem &= ~same;
        }
        if (lane == 0) p.bn[(size_t) b * nb + blockIdx.x] = kk;
        if (v) { p.owner[(size_t) b * m + ip] = own; p.brank[(size_t) b * m + ip] = (uint8_t) rank; }
        return;
    } else if constexpr (OWNER) {
        if (ss == 0u && valid) p.owner[(size_t) b * m + i] = rstar;
        return;
    }
    KS_KEEP (dr, rstar)
    KS_STAMP (3)
    // list offset / size of the winner.  One tile (always the case for the MINW == 2 variants, see icp_launch_search):
    // from LDS; the compile-time split keeps the compiler from merging the two sources into flat loads.
    uint32_t o, n;
    if constexpr (MINW == 2) { const uint2 on = s_on[rstar]; o = on.x; n = on.y; }
    else if constexpr (MASKED) {
        // the home tile's (offset, size) pairs are in LDS (buffer 1 is never overwritten): a winner there — the usual case — costs
        // no dependent global load
        if (rstar / KT == ht) { const uint2 on = s_on[rstar - ht * KT]; o = on.x; n = on.y; asm volatile ("" : "+v"(o), "+v"(n)); }
        else { o = gO[rstar]; n = gN[rstar]; }
    }
    else if (nr <= KT) { const uint2 on = s_on[rstar]; o = on.x; n = on.y; asm volatile ("" : "+v"(o), "+v"(n)); }
    else { o = gO[rstar]; n = gN[rstar]; }

    // ---- stage 2: exhaustive scan of that representative's list: the LPQ lanes of a query read LPQ consecutive
    // candidates (32 contiguous bytes each) per load.  (Staging the block's lists through LDS first was measured and is
    // slower: enumerating the distinct lists and the extra barrier cost more than the direct gathers.)
    float dmin; uint32_t jmin;
    if constexpr (S2W) {
        static_assert (MINW == 4 && LPQ == 8 && !OWNER, "lanes = candidates: the dense search variants");
        // ---- stage 2, long lists (dense variant, icp_s2_wave): lanes = candidates.  The scan above is bound by the vector-memory
        // path (every query's lanes load their list for themselves: 24 bytes per candidate and query through the L1); the
        // wave's 8 queries are neighbours and mostly share ONE list, so here the wave loads a list once — lane l takes the
        // positions l, l + 64, .. — and every lane evaluates its candidate against each query of the wave that has this list,
        // the query's six coordinates in SGPRs.  Distinct lists of the wave are served one after the other.  Per lane and
        // query: best (distance, trip); at the end one butterfly over the 64 lanes that halves the number of queries a lane
        // holds while it doubles the lanes reduced ((distance bits, position) as one 64-bit key: distances are >= +0, so
        // the unsigned order of the bits is the order of the values; smallest distance, ties -> lowest position).
        const uint32_t je = valid ? o + n : o;
        unsigned long long todo = __ballot (je != o);                // lanes of the queries with a list to scan
        float sx[8], sy[8], sz[8], sr[8], sg[8], sb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sx[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qx), 8 * q));
            sy[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qy), 8 * q));
            sz[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qz), 8 * q));
            sr[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qr), 8 * q));
            sg[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qg), 8 * q));
            sb[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qb), 8 * q));
        }
        float bd[8]; uint32_t btr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { bd[q] = __builtin_inff (); btr[q] = 0xFFFFFFFFu; }
        // (m8: one bit per query of the wave that has the list; made opaque per trip so that the eight tests stay scalar bit
        // tests inside the loop instead of eight hoisted lane masks)
#define KS_WCAND(G, C, TRIP)                                                                                  \
        asm volatile ("" : "+s"(m8));                                                                         \
        _Pragma ("unroll") for (int q = 0; q < 8; ++q)                                                        \
            if (m8 & (1u << q)) {                                                                             \
                const float2v d1_ = float2v { sx[q], sr[q] } - float2v { (G).x, (G).y },                      \
                              d2_ = float2v { sy[q], sg[q] } - float2v { (G).z, (G).w },                      \
                              d3_ = float2v { sz[q], sb[q] } - float2v { (C).x, (C).y };                      \
                const float2v gp_ = __builtin_elementwise_fma (d3_, d3_, __builtin_elementwise_fma (d2_, d2_, d1_ * d1_)); \
                const float d_ = __builtin_fmaf (alpha, gp_.y, gp_.x);                                        \
                if (d_ < bd[q]) { bd[q] = d_; btr[q] = (TRIP); }                                              \
            }
        while (todo) {
            const int l0 = (int) __builtin_ctzll (todo);
            const uint32_t rL = (uint32_t) __builtin_amdgcn_readlane ((int) rstar, l0);
            const uint32_t oL = (uint32_t) __builtin_amdgcn_readlane ((int) o, l0), nL = (uint32_t) __builtin_amdgcn_readlane ((int) n, l0);
            const unsigned long long match = __ballot (rstar == rL) & todo;      // same representative = same list
            todo &= ~match;
            uint32_t m8 = 0u;
#pragma unroll
            for (int q = 0; q < 8; ++q) m8 |= (uint32_t) ((match >> (8 * q)) & 1ull) << q;
            m8 = (uint32_t) __builtin_amdgcn_readfirstlane ((int) m8);
            const uint32_t vlastL = (oL + nL - 1u) << 5, ntr = (nL + 63u) >> 6;
            uint32_t voff = (oL + lane) << 5;
            for (uint32_t t = 0; t < ntr; t += 2u, voff += 2u * 64u * 32u) {
                const char *rec0 = XQb + min (voff, vlastL), *rec1 = XQb + min (voff + 64u * 32u, vlastL);
                const float4 g0 = *reinterpret_cast<const float4 *> (rec0); const float2 c0 = *reinterpret_cast<const float2 *> (rec0 + 16);
                const float4 g1 = *reinterpret_cast<const float4 *> (rec1); const float2 c1 = *reinterpret_cast<const float2 *> (rec1 + 16);
                KS_WCAND (g0, c0, t)
                if (t + 1u < ntr) { KS_WCAND (g1, c1, t + 1u) }
            }
        }
#undef KS_WCAND
        // the wave's winner per query: (distance bits, position) as one 64-bit key (distances are >= +0: the unsigned order of
        // the bits is the order of the values), and a butterfly over the 64 lanes that halves the queries a lane holds while it
        // doubles the lanes reduced — after three steps lane l holds query l & 7 over its group of 8 lanes, after six over the wave.
        // (Measured against it and slower, 283 -> 292 us at C: the distances alone through the butterfly and the winner's
        // position looked up with ballots / readlanes in scalars.)
        unsigned long long key[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t oq = (uint32_t) __builtin_amdgcn_readlane ((int) o, 8 * q), jeq = (uint32_t) __builtin_amdgcn_readlane ((int) je, 8 * q);
            const uint32_t pos = (btr[q] == 0xFFFFFFFFu) ? 0xFFFFFFFFu : min (oq + lane + 64u * btr[q], jeq - 1u);
            key[q] = ((unsigned long long) __float_as_uint (bd[q]) << 32) | pos;
        }
        auto xchg_dpp = [] (unsigned long long v, auto ctrl) -> unsigned long long {
            const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp (0, (int) (uint32_t) v, decltype (ctrl)::value, 0xF, 0xF, true);
            const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp (0, (int) (uint32_t) (v >> 32), decltype (ctrl)::value, 0xF, 0xF, true);
            return ((unsigned long long) hi << 32) | lo;
        };
        auto xchg_lane = [] (unsigned long long v, uint32_t src) -> unsigned long long {      // v of lane src
            const uint32_t lo = (uint32_t) __builtin_amdgcn_ds_bpermute ((int) (src << 2), (int) (uint32_t) v);
            const uint32_t hi = (uint32_t) __builtin_amdgcn_ds_bpermute ((int) (src << 2), (int) (uint32_t) (v >> 32));
            return ((unsigned long long) hi << 32) | lo;
        };
        auto min64 = [] (unsigned long long a_, unsigned long long b_) { return a_ < b_ ? a_ : b_; };
        unsigned long long k4[4], k2[2], k1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                // lane bit 0: keeps the queries 2j + (lane & 1)
            const bool odd = (lane & 1u) != 0u;
            const unsigned long long keep = odd ? key[2 * j + 1] : key[2 * j], send = odd ? key[2 * j] : key[2 * j + 1];
            k4[j] = min64 (keep, xchg_dpp (send, std::integral_constant<int, 0xB1> {}));      // quad_perm [1,0,3,2]
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                // lane bit 1
            const bool odd = (lane & 2u) != 0u;
            const unsigned long long keep = odd ? k4[2 * j + 1] : k4[2 * j], send = odd ? k4[2 * j] : k4[2 * j + 1];
            k2[j] = min64 (keep, xchg_dpp (send, std::integral_constant<int, 0x4E> {}));      // quad_perm [2,3,0,1]
        }
        {                                            // lane bit 2
            const bool odd = (lane & 4u) != 0u;
            const unsigned long long keep = odd ? k2[1] : k2[0], send = odd ? k2[0] : k2[1];
            k1 = min64 (keep, xchg_lane (send, lane ^ 4u));
        }
        k1 = min64 (k1, xchg_lane (k1, lane ^ 8u));  // over the wave's 8 groups of 8 lanes
        k1 = min64 (k1, xchg_lane (k1, lane ^ 16u));
        k1 = min64 (k1, xchg_lane (k1, lane ^ 32u));
        k1 = xchg_lane (k1, (lane & 56u) | (lane >> 3));            // to the lanes of query lane >> 3
        dmin = __uint_as_float ((uint32_t) (k1 >> 32)); jmin = (uint32_t) k1;
    } else {
    float best2 = __builtin_inff (); uint32_t bj = 0xFFFFFFFFu;
    const float2v vq_xr = { qx, qr }, vq_yg = { qy, qg }, vq_zb = { qz, qb };
    {
        // a batch = KS_DEPTH candidates per lane, all loads issued before the first distance (clamped addresses, the
        // tail is masked): one memory round trip per batch, and one batch covers a list of KS_DEPTH * LPQ candidates.
        // The number of trips the wave needs (its longest list) is a scalar: a trip no lane needs is neither loaded
        // nor evaluated, at the cost of scalar compares only (the scan is bound by the vector-memory issue rate).
        constexpr uint32_t KS_DEPTH = (KS_SPLIT == 16) ? ICP_S2_DEPTH16 : 4u;
        const uint32_t je = valid ? o + n : o;       // (invalid queries: an empty range)
        // the wave's trip count = its longest list: the lanes of a query hold the same count, so one mirror inside the 16-lane rows
        // (two queries per row at 8 lanes per query) and the two row broadcasts of a wave reduction leave the maximum in lane 63
        uint32_t nl = (je - o + KS_SPLIT - 1u) / KS_SPLIT;
        if (KS_SPLIT == 8) nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) nl, 0x140, 0xF, 0xF, true));        // row_mirror
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x142, 0xA, 0xF, false));                   // row_bcast:15 -> rows 1, 3
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x143, 0xC, 0xF, false));                   // row_bcast:31 -> rows 2, 3
        const uint32_t ntrips = (uint32_t) __builtin_amdgcn_readlane ((int) nl, 63);
        // byte offsets from the uniform base (m <= 2^20: < 2^25 bytes); a position past the list's end is clamped to its last
        // element, whose (distance, position) some lane holds anyway — a duplicate changes neither the minimum nor the lowest
        // position among equals, so there is no tail test.  The lane keeps the TRIP of its best candidate (a scalar + constant per
        // candidate instead of a recomputed position); trips ascend, so a strict '<' keeps the lane's lowest position.
        const uint32_t alast = (max (je, 1u) - 1u) << 5;
        for (uint32_t tb = 0; tb < ntrips; tb += KS_DEPTH) {
            const uint32_t a0 = (o + ss + tb * KS_SPLIT) << 5, nt = min (KS_DEPTH, ntrips - tb);
            float4 g[KS_DEPTH], c[KS_DEPTH];
#pragma unroll
            for (uint32_t t = 0; t < KS_DEPTH; ++t) {
                if (t >= nt) break;
                const char *rec = XQb + min (a0 + t * (KS_SPLIT * 32u), alast);
                g[t] = *reinterpret_cast<const float4 *> (rec); c[t] = *reinterpret_cast<const float4 *> (rec + 16);
            }
#pragma unroll
            for (uint32_t t = 0; t < KS_DEPTH; ++t) {
                if (t >= nt) break;
                KS_CAND (g[t], c[t], tb + t);
            }
        }
        if (bj != 0xFFFFFFFFu) bj = min (o + ss + bj * KS_SPLIT, max (je, 1u) - 1u);     // trip -> list position
        if (je == o) { best2 = __builtin_inff (); bj = 0xFFFFFFFFu; }      // empty list / invalid query: nothing above was a candidate
    }
    KS_STAMP (4)
    // the query's winner among its lanes: smallest distance, ties -> lowest list position; that lane finishes
    // the query (lane ss == 0 when the list is empty or no candidate has a finite distance)
    dmin = ks_grp_min_f<KS_SPLIT> (best2);
    jmin = ks_grp_min_u<KS_SPLIT> (best2 == dmin ? bj : 0xFFFFFFFFu);
    }
    KS_KEEP (dmin, jmin)
    KS_STAMP (5)
    // Hand-off: lane 0 of every query leaves (q, distance, winner position, representative, flags) in LDS, and ONE wave
    // finishes all 64 queries of the block with every lane active (lane e = query e): the winner's record, the weight,
    // the per-query outputs and the 18 moment products are then issued once per block instead of once per wave for a
    // handful of active lanes (an instruction costs the same whatever its lane count).
    if (ss == 0u) {
        const bool empty = (n == 0u);
        s_qa[qe] = make_float4 (qx, qy, qz, empty ? dr : dmin);
        s_qb[qe] = make_uint4 (empty ? rstar : ((jmin == 0xFFFFFFFFu) ? o : jmin), rstar, (valid ? 1u : 0u) | (empty ? 2u : 0u), i);
    }
    __syncthreads ();
    if (slice == 0u) {
        const float4 qa = s_qa[lane]; const uint4 qb = s_qb[lane];
        const bool v = (qb.z & 1u) != 0u, empty = (qb.z & 2u) != 0u;
        // the search ran on geo + a pho (a positive common factor changes neither the argmin nor the ties, and the pruning
        // bound d >= geo stays as it is); the distance reported and fed to the weights carries the metric's absolute scale
        const float ex = qa.x, ey = qa.y, ez = qa.z, d = p.dist_scale * qa.w;
        const uint32_t ei = qb.w;
        float w = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
        if (v) {
            uint32_t id;
            if (empty) {             // empty list: fall back to the representative itself
                const float4 nn = R4[2 * (size_t) qb.x];
                id = p.rep_src[(size_t) b * nr + qb.x]; f0 = nn.x; f1 = nn.y; f2 = nn.z;
            } else {                 // the winner's point, or, when every distance is inf / NaN, the first list element as
                                     // the serial scan would: one reload instead of tracking it per candidate
                const char *rec = XQb + (qb.x << 5);
                const float4 wg = *reinterpret_cast<const float4 *> (rec), wc = *reinterpret_cast<const float4 *> (rec + 16);
                f0 = wg.x; f1 = wg.z; f2 = wc.x; id = __float_as_uint (wc.z);
            }
            w = p.weighted ? 100.f / (100.f + d) : 1.f;                // icp_kernels.cl:232
            // per-query outputs: uniform bases + 32-bit byte offsets (i < 2^20)
            icp_dist_id di; di.dist = d; di.id = id;
            char *o_nn = reinterpret_cast<char *> (p.nn_id + (size_t) b * m), *o_pf = reinterpret_cast<char *> (p.PF + (size_t) b * m);
            char *o_pm = reinterpret_cast<char *> (p.PM + (size_t) b * m), *o_rid = reinterpret_cast<char *> (p.rid + (size_t) b * m);
            // (fused mode consumes none of these itself: inside a graph of a fixed length only the last iteration
            // stores them — except the nearest representative where the next search seeds its pruning with it)
            const bool emit = !FUSED || (check_flags & 8u);
            if (emit) {
                *reinterpret_cast<icp_dist_id *> (o_nn + (ei << 3)) = di;
                *reinterpret_cast<float4 *> (o_pf + (ei << 4)) = make_float4 (f0, f1, f2, w);
                *reinterpret_cast<float4 *> (o_pm + (ei << 4)) = make_float4 (ex, ey, ez, d);
            }
            if (emit || PRUNE) *reinterpret_cast<uint32_t *> (o_rid + (ei << 2)) = qb.y;
        }
        if constexpr (FUSED) {
            // the 18 moments of this pair in double (oracle orc_moments_fused); invalid queries contribute 0
            double W = (double) w;
            double g0 = v ? (double) f0 : 0.0, g1 = v ? (double) f1 : 0.0, g2 = v ? (double) f2 : 0.0;
            double q0 = (double) ex, q1 = (double) ey, q2 = (double) ez;
            if (!v) { W = 0.0; q0 = q1 = q2 = 0.0; }
            double wq0 = W * q0, wq1 = W * q1, wq2 = W * q2;
            s_mom[0][lane] = W;
            s_mom[1][lane] = W * g0; s_mom[2][lane] = W * g1; s_mom[3][lane] = W * g2;
            s_mom[4][lane] = wq0; s_mom[5][lane] = wq1; s_mom[6][lane] = wq2;
            s_mom[7][lane] = wq0 * g0; s_mom[8][lane] = wq0 * g1; s_mom[9][lane] = wq0 * g2;
            s_mom[10][lane] = wq1 * g0; s_mom[11][lane] = wq1 * g1; s_mom[12][lane] = wq1 * g2;
            s_mom[13][lane] = wq2 * g0; s_mom[14][lane] = wq2 * g1; s_mom[15][lane] = wq2 * g2;
            s_mom[16][lane] = W * ((g0 * g0 + g1 * g1) + g2 * g2);
            s_mom[17][lane] = W * ((q0 * q0 + q1 * q1) + q2 * q2);
        } else
            s_w[lane] = w;
    }
    KS_STAMP (6)
    __syncthreads ();
    if constexpr (FUSED) {
        // halving tree over the block's 64 pairs, one 16-lane row per moment (rows 0..17 of the 32 rows)
        if (slice * 4u >= (uint32_t) ICP_NMOM) return;                     // (waves without a row: done)
        const uint32_t l = lane & 15u, mrow = slice * 4u + (lane >> 4);     // first 18 of the block's 4*KS_SPLIT rows
        const uint32_t k = min (mrow, (uint32_t) ICP_NMOM - 1u);
        double c0 = s_mom[k][l] + s_mom[k][l + 32], c1 = s_mom[k][l + 16] + s_mom[k][l + 48];
        double v = row_tree_tail_d (c0 + c1);
        const uint32_t obuf = CHAIN ? (p.slot ^ 1u) : 0u;
        if (l == 0 && mrow < ICP_NMOM) p.mom[(((size_t) b * 2 + obuf) * ICP_NMOM + mrow) * p.nb + tile_id] = v;
    } else if (slice == 0 && p.weighted) {
        // tree levels d = 64 .. 2 restricted to this block's parity class (icp_kernels.cl:244-249):
        // element e of the class is position 2e + parity; levels pair e with e+32, e+16, .., e+1.
        const uint32_t l = lane & 15u;
        float a[4] = { s_w[l], s_w[l + 16], s_w[l + 32], s_w[l + 48] };
        float v = row_tree4 (a);
        if (lane == 0) p.wpart[(size_t) b * 2 * p.nwp + blockIdx.x] = v;
    }
    KS_STAMP (7)
}

// ------------------------------------------------------------------------------------------
// second-level kernels for sizes beyond one work-group of partials (m > 65536 / m > 16384)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64) void k_sum_w (icp_params p)
{
    uint32_t b = blockIdx.y, lane = threadIdx.x;
    icp_reg_state *st = p.st + b;
    if (p.check && st->done) return;
    double sw = sum_w_row (p.wpart + (size_t) b * 2 * p.nwp, p.nwp, lane & 15u);
    if (lane == 0) st->sum_w = sw;
}

// multi-level icpGMean: groups of 128 block means per pass until one vector per set remains
// (nwg <= 8192 here, i.e. at most two passes; one 16-lane row per group)
__global__ __launch_bounds__ (1024) void k_gmean (icp_params p)
{
    uint32_t b = blockIdx.y, l = threadIdx.x & 15u, row = threadIdx.x >> 4;      // 64 rows
    icp_reg_state *st = p.st + b;
    if (p.check && st->done) return;
    const uint32_t ng = (p.nwg + 127u) / 128u;                                    // <= 64
    for (uint32_t set = 0; set < 2; ++set) {
        const float4 *cur = p.mpart + ((size_t) b * 2 + set) * p.nwg;
        float4 *scr = p.mscr + ((size_t) b * 2 + set) * ng;
        float4 r = gmean_row (cur + (size_t) min (row, ng - 1) * 128, row < ng ? min (128u, p.nwg - row * 128u) : 0u, l);
        if (row < ng && l == 0) scr[row] = r;
        __syncthreads ();
        float4 f = gmean_row (scr, ng, l);
        if (threadIdx.x == 0) {
            st->means[set * 4 + 0] = f.x; st->means[set * 4 + 1] = f.y; st->means[set * 4 + 2] = f.z; st->means[set * 4 + 3] = 0.f;
        }
        __syncthreads ();
    }
}

// ------------------------------------------------------------------------------------------
// K2  means: icpMean(_Weighted) — kernels/icp_kernels.cl:371-411, 455-495.
//     One 16-lane row per 128-pair work-group (8 pairs per lane), 4 work-groups per wave.
// ------------------------------------------------------------------------------------------
// (leading scalars: preloaded with the dispatch, see k_search)
__global__ __launch_bounds__ (64) void k_means (const float4 *gPF, const float4 *gPM, const float *gwpart, icp_reg_state *gst,
                                                uint32_t m, uint32_t nwp, uint32_t weighted, uint32_t check, icp_params p)
{
    const uint32_t b = blockIdx.y, lane = threadIdx.x, l = lane & 15u;
    const uint32_t g = blockIdx.x * 4u + (lane >> 4);
    icp_reg_state *st = gst + b;
    if (check && st->done) return;

    const float4 *PF = gPF + (size_t) b * m, *PM = gPM + (size_t) b * m;
    float4 pf[8], pm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t e = g * 128u + l + 16u * k;
        bool ok = (e & ~1u) < m;                   // the pair's flag guards both points (icp_kernels.cl:390-392)
        pf[k] = make_float4 (0.f, 0.f, 0.f, 0.f); pm[k] = pf[k];
        if (ok) { pf[k] = PF[e]; pm[k] = PM[e]; }
    }
    double sum_w = 1.0;
    if (weighted) {
        if (nwp <= 512) {
            double sw = sum_w_row (gwpart + (size_t) b * 2 * nwp, nwp, l);
            unsigned long long u = __builtin_bit_cast (unsigned long long, sw);
            uint32_t lo = __builtin_amdgcn_readfirstlane ((uint32_t) u), hi = __builtin_amdgcn_readfirstlane ((uint32_t) (u >> 32));
            sum_w = __builtin_bit_cast (double, ((unsigned long long) hi << 32) | lo);
            if (blockIdx.x == 0 && lane == 0) st->sum_w = sum_w;
        } else sum_w = st->sum_w;                    // written by k_sum_w
    }
    float fx[8], fy[8], fz[8], qx[8], qy[8], qz[8];
    const float nf = (float) m;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t e = g * 128u + l + 16u * k;
        bool ok = (e & ~1u) < m;
        if (weighted) {
            float kk = (float) ((double) pf[k].w / sum_w);               // icp_kernels.cl:475
            fx[k] = ok ? kk * pf[k].x : 0.f; fy[k] = ok ? kk * pf[k].y : 0.f; fz[k] = ok ? kk * pf[k].z : 0.f;
            qx[k] = ok ? kk * pm[k].x : 0.f; qy[k] = ok ? kk * pm[k].y : 0.f; qz[k] = ok ? kk * pm[k].z : 0.f;
        } else {                                                         // icp_kernels.cl:391
            fx[k] = ok ? pf[k].x / nf : 0.f; fy[k] = ok ? pf[k].y / nf : 0.f; fz[k] = ok ? pf[k].z / nf : 0.f;
            qx[k] = ok ? pm[k].x / nf : 0.f; qy[k] = ok ? pm[k].y / nf : 0.f; qz[k] = ok ? pm[k].z / nf : 0.f;
        }
    }
    float4 mf, mm;
    mf.x = row_tree8 (fx); mf.y = row_tree8 (fy); mf.z = row_tree8 (fz); mf.w = 0.f;
    mm.x = row_tree8 (qx); mm.y = row_tree8 (qy); mm.z = row_tree8 (qz); mm.w = 0.f;
    if (l == 0 && g < p.nwg) {
        p.mpart[((size_t) b * 2 + 0) * p.nwg + g] = mf;
        p.mpart[((size_t) b * 2 + 1) * p.nwg + g] = mm;
    }
}

// ------------------------------------------------------------------------------------------
// K3  S matrix: icpSubtractMean + icpSijProducts(_Weighted) + reduce_sum_f first level
//     kernels/icp_kernels.cl:588-602, 633-671, 703-743; kernels/reduce_kernels.cl:230-264.
//     reduce_sum_f work-group w covers 512 columns = 128 positions of 4 columns.  One wave takes the
//     16 positions = r (mod 8) of a work-group (64 columns, one per lane, 4 strided points each) and
//     runs the tree levels d = 64, 32, 16, 8, which stay inside that class.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64) void k_sij (const float4 *gPF, const float4 *gPM, const float4 *gmpart, icp_reg_state *gst,
                                              uint32_t m, uint32_t G, uint32_t nwg, uint32_t check, icp_params p)
{
    const uint32_t b = blockIdx.y, lane = threadIdx.x, l = lane & 15u, row = lane >> 4;
    icp_reg_state *st = gst + b;
    if (check && st->done) return;

    __shared__ __attribute__ ((aligned (16))) float s_col[11][64];

    const uint32_t wg = blockIdx.x >> 3, res = blockIdx.x & 7u;
    const uint32_t col = wg * 512u + 4u * (res + 8u * (lane >> 2)) + (lane & 3u);
    const float4 *PF = gPF + (size_t) b * m, *PM = gPM + (size_t) b * m;
    float4 pf[4], pm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                        // icp_kernels.cl:718: pi = gX + k gXdim
        uint32_t pi = col + (uint32_t) k * G;
        pf[k] = make_float4 (0.f, 0.f, 0.f, 0.f); pm[k] = pf[k];
        if (col < G && pi < m) { pf[k] = PF[pi]; pm[k] = PM[pi]; }
    }
    float4 mf, mm;
    if (nwg <= 128) {                              // icpGMean in the prologue
        float4 r0 = gmean_row (gmpart + ((size_t) b * 2 + 0) * nwg, nwg, l);
        float4 r1 = gmean_row (gmpart + ((size_t) b * 2 + 1) * nwg, nwg, l);
        mf.x = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r0.x)));
        mf.y = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r0.y)));
        mf.z = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r0.z)));
        mm.x = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r1.x)));
        mm.y = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r1.y)));
        mm.z = __builtin_bit_cast (float, __builtin_amdgcn_readfirstlane (__builtin_bit_cast (uint32_t, r1.z)));
        if (blockIdx.x == 0 && lane == 0) {
            st->means[0] = mf.x; st->means[1] = mf.y; st->means[2] = mf.z; st->means[3] = 0.f;
            st->means[4] = mm.x; st->means[5] = mm.y; st->means[6] = mm.z; st->means[7] = 0.f;
        }
    } else {                                         // written by k_gmean
        mf = make_float4 (st->means[0], st->means[1], st->means[2], 0.f);
        mm = make_float4 (st->means[4], st->means[5], st->means[6], 0.f);
    }

    const float c = p.c;
    float A[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) A[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t pi = col + (uint32_t) k * G;
        if (col < G && pi < m) {
            float Mp[3] = { c * (pm[k].x - mm.x), c * (pm[k].y - mm.y), c * (pm[k].z - mm.z) };
            float Fp[3] = { c * (pf[k].x - mf.x), c * (pf[k].y - mf.y), c * (pf[k].z - mf.z) };
            float ff = (Fp[0] * Fp[0] + Fp[1] * Fp[1]) + Fp[2] * Fp[2];
            float m2 = (Mp[0] * Mp[0] + Mp[1] * Mp[1]) + Mp[2] * Mp[2];
            if (p.weighted) {
                float w = pf[k].w;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) A[a * 3 + bb] = A[a * 3 + bb] + w * (Mp[a] * Fp[bb]);
                A[9] = A[9] + w * ff; A[10] = A[10] + w * m2;
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) A[a * 3 + bb] = A[a * 3 + bb] + Mp[a] * Fp[bb];
                A[9] = A[9] + ff; A[10] = A[10] + m2;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) s_col[k][lane] = A[k];
    __syncthreads ();
    // position j (= quad j) = ((c0 + c1) + c2) + c3 — reduce_kernels.cl:245-251; row `row` takes S rows row, row+4, row+8
    const uint32_t nsp8 = p.nsp * 8u;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        uint32_t a = row + 4u * it;
        float4 v = *reinterpret_cast<const float4 *> (&s_col[min (a, 10u)][4 * l]);
        float t = row_tree_tail (sum4 (v));
        if (l == 0 && a < 11) p.spart[((size_t) b * 11 + a) * nsp8 + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// K4  finalize: S final tree (a8), rotation (a9 / a12), composition (a10), convergence (a11).
//     One block of 3 waves per registration: 11 rows reduce the 11 rows of S, then wave 0 runs the
//     lane-parallel power method and lane 0 composes.
// ------------------------------------------------------------------------------------------
template <int ROT>
__global__ __launch_bounds__ (192) void k_finalize (const float *gspart, icp_reg_state *gst, uint32_t nsp, uint32_t check, icp_params p)
{
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, l = tid & 15u, row = tid >> 4;
    icp_reg_state *st = gst + b;
    if (check && st->done) return;

    __shared__ float s_S[12];
    {
        const uint32_t a = min (row, 10u);
        float v = s_reduce_row (gspart + ((size_t) b * 11 + a) * nsp * 8u, nsp, l);
        if (l == 0 && row < 11) s_S[row] = v;
    }
    __syncthreads ();
    if (tid >= 64) return;

    float S[11], means[8];
#pragma unroll
    for (int k = 0; k < 11; ++k) S[k] = s_S[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) means[k] = st->means[k];

    float Tk[8], Rk[9];
    int iters = 0;
    if constexpr (ROT == 1) iters = icp_power_method_quad (S, means, Tk, p.power_mode, lane);
    else icp_svd_rotation (S, means, Rk, Tk);

    if (lane == 0) {
        if (p.st_prev) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p.st_prev[b].T[k] = st->T[k];
        }
        icp_compose (st, Tk, Rk, ROT != 1);
#pragma unroll
        for (int k = 0; k < 11; ++k) st->S[k] = S[k];
        st->pm_iters = (uint32_t) iters;
        st->k = st->k + 1;
        if (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) st->done = 1;
    }
}

// ------------------------------------------------------------------------------------------
// fused mode finalize: 128-position double trees over the block moments, means / S from the moments
// (oracle orc_moments_fused / orc_moments_finish), then rotation, composition, convergence.
// One block of 5 waves per registration: row k (of 20) reduces moment k.
// ------------------------------------------------------------------------------------------
template <int ROT>
__global__ __launch_bounds__ (1024) void k_finalize_fused (const double *gmom, icp_reg_state *gst, uint32_t nb, uint32_t check, icp_params p)
{
    // (the leading scalars: see k_search)
#ifdef ICP_DBG_EXIT_AFTER
    return;                                          // (phase-count builds: the transform stays what it is, every iteration sees the same queries)
#endif
    const uint32_t b = blockIdx.x;
    icp_reg_state *st = gst + b;
    FF_STAMP (14)
    __shared__ icp_fin_result s_fin;
    __shared__ double s_l1[ICP_NMOM][128];
    __shared__ double s_t[ICP_NMOM];
    // the previous state travels with the moment loads (one vector load); a registration that has converged
    // (done, checked mode) leaves as soon as it has arrived
    const uint32_t sv = state_load_lanes (st);
    const double *mom = gmom + (size_t) b * 2 * ICP_NMOM * nb;
    const uint32_t ng = (nb + 127u) / 128u;
    const double *gl1 = (ng > ICP_L1_MIN_GROUPS && p.ml1) ? p.ml1 + (size_t) b * ICP_NMOM * ng : nullptr;   // (block-uniform)
    // (ng <= ICP_L1_MIN_GROUPS: the first tree level in this block, one pass; beyond it k_moment_level1 has run.  Loading the
    // passes of a larger in-block first level back to back measured slower than one after the other — B 6.64 against 6.29 us —, and
    // both lose to the level-1 kernel: 5.71 us for the two launches, profiles/r03_finalize_sweep_B.txt)
    double a0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a0[q] = 0.0;
    if (!gl1) fused_moment_loads<1024> (mom, nb, 0u, a0);
    // (the state goes to memory straight from the composing lane's registers: no LDS image, no second pass)
    // (host-driven checked runs: the new (k, done) is published by the NEXT search's prologue, not here — a store into host memory at the
    // end of this kernel is waited for by the kernel's end, and the next search by that: |F| = 65536, 18.7 -> 19.8 us per iteration)
    // A registration that converges here leaves its final state in host memory too, in front of the word's DONE | FINAL bits.
    fused_finalize_block<128, 1024, ROT, true> (p, mom, nb, check, sv, a0, &s_fin, s_l1, s_t, gl1, st, ff_no_hook (), 0u,
                                                p.hmirror ? p.hmirror + b : nullptr, false, nullptr, p.hstate ? p.hstate + b : nullptr,
                                                p.st_prev ? p.st_prev + b : nullptr);
}

// First tree level of the moments for large sets (|F| / 64 blocks > 128 * ICP_L1_MIN_GROUPS): one 16-lane row per
// (moment k, group g) task, 16 tasks per block, spread over the chip — k_finalize_fused alone would walk the
// 18 x ceil (nb / 128) tasks 64 at a time (config C: 36 dependent passes, 39 us).  Same tree, same bits.
__global__ __launch_bounds__ (256) void k_moment_level1 (const double *gmom, const icp_reg_state *gst, uint32_t nb, uint32_t check, uint32_t ng_magic, icp_params p)
{
    const uint32_t b = blockIdx.y, l = threadIdx.x & 15u, row = threadIdx.x >> 4;
    if (check && gst[b].done) return;
    const uint32_t ng = (nb + 127u) / 128u, ntask = ICP_NMOM * ng;
    const uint32_t task = min (blockIdx.x * 16u + row, ntask - 1u), k = ng_magic ? __umulhi (task, ng_magic) : task, g = task - k * ng;   // (ng_magic = floor (2^32 / ng) + 1, a preloaded scalar: no runtime division in front of the loads)
    const double *src = gmom + (size_t) b * 2 * ICP_NMOM * nb + (size_t) k * nb;
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t i = g * 128u + l + 16u * q;
        const double t = src[min (i, nb - 1u)];
        a[q] = (i < nb) ? t : 0.0;
    }
    const double v = row_tree8_d (a);
    if (l == 0 && blockIdx.x * 16u + row < ntask) p.ml1[(size_t) b * ntask + task] = v;
}

// chain end: finalize the last iteration's moments (slot given by p.slot) into the user-visible state.  (There is no
// begin kernel: the first launch of a chain reads the user-visible state itself.)
// The final state of a host-driven checked run -> the host (p.hstate: fine-grained pinned memory), then the word's FINAL bit: one wave,
// so that the wave's own wait (the fence) covers every lane's stores before the word goes out.  v = dword t of the state.
static __device__ __forceinline__ void state_to_host (const icp_params &p, uint32_t b, uint32_t t, uint32_t v, uint32_t k, uint32_t done)
{
    if (!p.hstate) return;                              // (kernel-uniform)
    if (t < sizeof (icp_reg_state) / 4) reinterpret_cast<uint32_t *> (p.hstate + b)[t] = v;
    __threadfence_system ();
    if (t == 0 && p.hmirror) icp_mirror_store (p.hmirror + b, ICP_MIRROR_WORD (p.epoch, k, done) | ICP_MIRROR_FINAL);
}

// (tracked sequences: the end kernel of a registration that did not converge releases the next frame's gate — behind the state it has written)
static __device__ __forceinline__ void seq_release (const icp_params &p, uint32_t t)
{
    if (!p.track_seq) return;
    __threadfence ();
    if (t == 0) __hip_atomic_store (p.track_seq, p.seq_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ROT>
__global__ __launch_bounds__ (320) void k_chain_end (icp_params p)
{
    const uint32_t b = blockIdx.x;
    const icp_reg_state *sin = p.cst + (size_t) b * 2 + p.slot;
    icp_reg_state *st = p.st + b;
    __shared__ icp_fin_result s_fin;
    __shared__ double s_l1[ICP_NMOM][32];
    __shared__ double s_t[ICP_NMOM];
    if ((p.check && sin->done) || !sin->pending) {
        if (threadIdx.x < 64) {
            const uint32_t t = threadIdx.x;
            uint32_t v = reinterpret_cast<const uint32_t *> (sin)[min (t, (uint32_t) sizeof (icp_reg_state) / 4u - 1u)];
            if (t == offsetof (icp_reg_state, pending) / 4) v = 0u;
            if (t < sizeof (icp_reg_state) / 4) reinterpret_cast<uint32_t *> (st)[t] = v;
            state_to_host (p, b, t, v, sin->k, sin->done);
            seq_release (p, t);
        }
        return;
    }
    const double *mom = p.mom + ((size_t) b * 2 + p.slot) * ICP_NMOM * p.nb;
    double a0[8];
    fused_moment_loads<320> (mom, p.nb, 0u, a0);
    fused_finalize_block<32, 320, ROT> (p, mom, p.nb, 0u, state_load_lanes (sin), a0, &s_fin, s_l1, s_t, nullptr, nullptr, ff_no_hook (), 1u,
                                        nullptr, false, nullptr, nullptr, p.st_prev ? p.st_prev + b : nullptr);
    fin_result_to_state (&s_fin, st, 0u);
    if (threadIdx.x < 64) {
        const uint32_t t = threadIdx.x;
        uint32_t v = reinterpret_cast<const uint32_t *> (&s_fin)[min (t, (uint32_t) sizeof (icp_reg_state) / 4u - 1u)];
        if (t == offsetof (icp_reg_state, pending) / 4) v = 0u;
        state_to_host (p, b, t, v, s_fin.k, s_fin.done);
        seq_release (p, t);
    }
}

// separate launches (dense sizes, reference order): the end of a host-driven checked run — p.st -> p.hstate + the FINAL bit
__global__ __launch_bounds__ (64) void k_publish_state (icp_params p)
{
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    const icp_reg_state *st = p.st + b;
    const uint32_t v = reinterpret_cast<const uint32_t *> (st)[min (t, (uint32_t) sizeof (icp_reg_state) / 4u - 1u)];
    state_to_host (p, b, t, v, st->k, st->done);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline uint32_t icp_tpr_magic (uint32_t side)
{   // floor (2^32 / tpr) + 1 for tpr = side / 8 tiles per row (tpr == 1: one block, b == 0, any value works)
    const uint32_t tpr = side >> 3;
    return tpr ? (uint32_t) ((1ull << 32) / tpr + 1ull) : 0u;
}
bool icp_dense (const icp_params &p)
{   // more blocks than one per CU: trade registers for occupancy; many representatives: stage 1 is throughput-bound
    return (size_t) p.batch * p.nb > 512u || p.nr >= ICP_S1_REJECT_MIN_NR;
}

// LDS tile of the dense search variant: 256 representatives (21 KB of LDS, 8 waves per SIMD) where a tile holds whole rows of
// 4 x 4 pruning groups (representative grid at most 64 wide: |R| <= 4096), else 1024.
// (Several 256-tiles with a block vote per tile measured slower than the 1024-tile — B 16.6 -> 17.5 us, C 417 -> 520 us —: the
// MASKED form of k_search decides a block's tile set in one pre-pass instead.)
uint32_t icp_dense_tile (const icp_params &p) { return (p.nr <= 256u || p.nrx <= 64u) ? 256u : 1024u; }
// Stage 2 with lanes = candidates pays where the lists are long (its 64-lane reduction per query is a fixed cost; lists of 64 are
// one trip either way): from ICP_S2_WAVE_MIN candidates per list on average.  ICP_AMD_S2WAVE=0/1 forces it (diagnostics).
#ifndef ICP_S2_WAVE_MIN
#define ICP_S2_WAVE_MIN 128u
#endif
uint32_t icp_s2_wave_of (const icp_params &p)
{
    if (const char *e = std::getenv ("ICP_AMD_S2WAVE")) return (e[0] == '1' && icp_dense (p)) ? 1u : 0u;
    return (icp_dense (p) && p.nr && p.m / p.nr >= ICP_S2_WAVE_MIN) ? 1u : 0u;
}
uint32_t icp_tbox_of (const icp_params &p) { return (icp_dense (p) && p.nr > 256u && icp_dense_tile (p) == 256u) ? 256u : 1024u; }

void icp_search_layout_of (const icp_params &p, int *dense, int *tile, int *stage2)
{
    const bool d = icp_dense (p);
    if (dense) *dense = d ? 1 : 0;
    if (tile) *tile = d ? (int) icp_dense_tile (p) : 1024;
    if (stage2) *stage2 = (d && p.s2wave) ? 1 : 0;
}

// buildRBC in two launches where the search is the latency variant (one registration of up to 32768 points, or a few small ones)
bool icp_build_lists (const icp_params &p) { return !icp_dense (p) && p.nr < 1024u && p.blist != nullptr; }

// RBC construct, step 1: owner(x) = nearest representative — the search kernel's stage 1 over the fixed points
// (dense variant: LDS tiles of 256 representatives up to |R| = 4096 — four blocks per CU —, of 1024 beyond, where a 4 x 4 tile
// group no longer fits a 256-tile: icp_dense_tile; latency variant: also leaves the per-block owner lists for k_place_lists)
void icp_launch_owner_search (const icp_params &p, hipStream_t s)
{
    if (icp_dense (p) && p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, true, 1, 256, false>), dim3 (p.nb, p.batch), dim3 (512), 0, s, p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p);
    else if (icp_dense (p) && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, true, 1, 256, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p);
    else if (icp_dense (p)) hipLaunchKernelGGL ((k_search<true, false, 4, 8, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p);
    else hipLaunchKernelGGL ((k_search<true, false, 2, 16, true>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p);
}

void icp_launch_search (const icp_params &p, hipStream_t s)
{
    const bool dense = icp_dense (p);
#define KS_FLAGS(p) ((uint32_t) ((p).check ? 1u : 0u) | ((p).emit ? 8u : 0u) | ((p).warm_seed ? 32u : 0u) | ((p).xcdmap ? 64u : 0u))
#define KS_ARGS p.M, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, KS_FLAGS (p), p
#define KS_CHAIN_ARGS p.M, p.R, p.cst + p.slot, (const double *) p.mom + (size_t) p.slot * ICP_NMOM * p.nb, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, KS_FLAGS (p), p
    if (p.fused) {
        if (dense && p.s2wave && p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, false, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.s2wave && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, true, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.s2wave) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 1024, false, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, false>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense) hipLaunchKernelGGL ((k_search<true, false, 4, 8>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else hipLaunchKernelGGL ((k_search<true, false, 2, 16>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, KS_ARGS);
    } else {
        // (the same tile choice as the fused variants: the tile boxes of a registration are built for one tile size, p.tbox)
        if (dense && p.s2wave && p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, false, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.s2wave && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, true, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.s2wave) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 1024, false, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, false>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (dense) hipLaunchKernelGGL ((k_search<false, false, 4, 8>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else hipLaunchKernelGGL ((k_search<false, false, 2, 16>), dim3 (2 * p.nwg, p.batch), dim3 (1024), 0, s, KS_ARGS);
    }
}

void icp_launch_means (const icp_params &p, hipStream_t s)
{
    if (p.weighted && p.nwp > 512) hipLaunchKernelGGL (k_sum_w, dim3 (1, p.batch), dim3 (64), 0, s, p);
    hipLaunchKernelGGL (k_means, dim3 ((p.nwg + 3) / 4, p.batch), dim3 (64), 0, s, (const float4 *) p.PF, (const float4 *) p.PM, (const float *) p.wpart, p.st,
                        p.m, p.nwp, (uint32_t) p.weighted, (uint32_t) p.check, p);
}

void icp_launch_sij (const icp_params &p, hipStream_t s)
{
    if (p.nwg > 128) hipLaunchKernelGGL (k_gmean, dim3 (1, p.batch), dim3 (1024), 0, s, p);
    hipLaunchKernelGGL (k_sij, dim3 (((p.G + 511) / 512) * 8, p.batch), dim3 (64), 0, s, (const float4 *) p.PF, (const float4 *) p.PM, (const float4 *) p.mpart, p.st,
                        p.m, p.G, p.nwg, (uint32_t) p.check, p);
}

void icp_launch_finalize (const icp_params &p, hipStream_t s)
{
    // the rotation solver is a template parameter (p.rot: 1 power method, else SVD)
    if (p.fused) {
        const uint32_t ng = (p.nb + 127u) / 128u;
        if (ng > ICP_L1_MIN_GROUPS && p.ml1)
            hipLaunchKernelGGL (k_moment_level1, dim3 ((ICP_NMOM * ng + 15u) / 16u, p.batch), dim3 (256), 0, s, (const double *) p.mom, (const icp_reg_state *) p.st, p.nb, (uint32_t) p.check, p.ng_magic, p);
        if (p.rot == 1) hipLaunchKernelGGL (k_finalize_fused<1>, dim3 (p.batch), dim3 (1024), 0, s, (const double *) p.mom, p.st, p.nb, (uint32_t) p.check, p);
        else hipLaunchKernelGGL (k_finalize_fused<0>, dim3 (p.batch), dim3 (1024), 0, s, (const double *) p.mom, p.st, p.nb, (uint32_t) p.check, p);
    } else {
        if (p.rot == 1) hipLaunchKernelGGL (k_finalize<1>, dim3 (p.batch), dim3 (192), 0, s, (const float *) p.spart, p.st, p.nsp, (uint32_t) p.check, p);
        else hipLaunchKernelGGL (k_finalize<0>, dim3 (p.batch), dim3 (192), 0, s, (const float *) p.spart, p.st, p.nsp, (uint32_t) p.check, p);
    }
}

__global__ void k_nop (icp_params p) { if (p.m == 0xFFFFFFFFu) p.st->k = 0; }

// diagnostic: any subset of the iteration's kernels (bit 0 search, 1 means, 2 sij, 3 finalize, 4 empty kernel)
void icp_launch_masked (const icp_params &p, hipStream_t s, unsigned mask)
{
    if (mask & 1u) icp_launch_search (p, s);
    if ((mask & 2u) && !p.fused) icp_launch_means (p, s);
    if ((mask & 4u) && !p.fused) icp_launch_sij (p, s);
    if (mask & 8u) icp_launch_finalize (p, s);
    if (mask & 16u) hipLaunchKernelGGL (k_nop, dim3 (256, p.batch), dim3 (64), 0, s, p);
}

// chained fused run: begin, one launch per iteration, end (icp_chain_supported: second tree level fits 32 groups)
// Measured at |F|=|M|=16384: the replicated prologue (every block fetching the 36 KB of fresh moment partials)
// costs more than the launch boundary it removes (15.7 vs 14.9 us per iteration), so the chain is opt-in.
// One launch per iteration (fused mode): p.chain = 0 never, 1 automatic (latency-bound sizes: the launch boundary it
// removes outweighs every block re-deriving T), 2 always (sizes the second tree level of the prologue can hold).
bool icp_chain_supported (const icp_params &p)
{
    return p.fused && p.nb <= 4096u && p.nr <= 1024u && (p.chain == 2 || (p.chain == 1 && !icp_dense (p)));
}

// launch j of a chain: reads state slot / moments buffer j & 1 and leaves the other (j = 0: reads the user-visible state, nothing to finalize yet)
void icp_launch_chain_one (const icp_params &p0, hipStream_t s, uint32_t j, bool fresh, bool emit)
{
    icp_params p = p0;
    p.slot = j & 1u;
    p.emit = emit ? 1 : 0;
    const uint32_t first_flags = 2u | (fresh ? 16u : 0u);             // (fresh: the run starts from the identity, see k_search)
    // (host-driven checked runs — p.hmirror set — take the HOSTRUN instantiation, fixed-length graphs the plain one)
#define KS_CHAIN_LAUNCH(ROT_, HR_)                                                                                                              \
    do {                                                                                                                                        \
        if (j == 0) hipLaunchKernelGGL ((k_search<true, true, 2, 16, false, ROT_, 1024, false, false, HR_>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, p.M, p.R, p.st,   \
                                        (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, KS_FLAGS (p) | first_flags, p);     \
        else hipLaunchKernelGGL ((k_search<true, true, 2, 16, false, ROT_, 1024, false, false, HR_>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, KS_CHAIN_ARGS);          \
    } while (0)
    const bool hostrun = p.hmirror != nullptr;
    if (p.rot == 1) { if (hostrun) KS_CHAIN_LAUNCH (1, true); else KS_CHAIN_LAUNCH (1, false); }
    else            { if (hostrun) KS_CHAIN_LAUNCH (0, true); else KS_CHAIN_LAUNCH (0, false); }
#undef KS_CHAIN_LAUNCH
}

void icp_launch_chain_end (const icp_params &p0, hipStream_t s, uint32_t launches)
{
    icp_params p = p0;
    p.slot = launches & 1u;
    if (p.rot == 1) hipLaunchKernelGGL (k_chain_end<1>, dim3 (p.batch), dim3 (320), 0, s, p);
    else hipLaunchKernelGGL (k_chain_end<0>, dim3 (p.batch), dim3 (320), 0, s, p);
}

void icp_launch_publish_state (const icp_params &p, hipStream_t s)
{
    hipLaunchKernelGGL (k_publish_state, dim3 (p.batch), dim3 (64), 0, s, p);
}

void icp_launch_chain (const icp_params &p, hipStream_t s, uint32_t iterations, bool fresh)
{
    if (iterations == 0) return;
    for (uint32_t j = 0; j < iterations; ++j)                       // (with checks on, any iteration may be the last executed: every launch emits)
        icp_launch_chain_one (p, s, j, fresh, p.check || j + 1 == iterations);
    icp_launch_chain_end (p, s, iterations);
}

void icp_launch_iteration (const icp_params &p, hipStream_t s)
{
    icp_launch_search (p, s);
    if (!p.fused) {
        icp_launch_means (p, s);
        icp_launch_sij (p, s);
    }
    icp_launch_finalize (p, s);
}
