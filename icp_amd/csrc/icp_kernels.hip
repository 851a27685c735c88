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
#include "icp_search.h"

// ------------------------------------------------------------------------------------------
// second-level trees of the reference-order reductions (k_means, k_sij, k_finalize)
// ------------------------------------------------------------------------------------------
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
__global__ __launch_bounds__ (256) void k_moment_level1 (const double *gmom, const icp_reg_state *gst, double *gl1, uint32_t nb, uint32_t check, uint32_t ng_magic)
{
    const uint32_t b = blockIdx.y, l = threadIdx.x & 15u, row = threadIdx.x >> 4;
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
    // (a converged registration: asked behind the loads — in front of them the flag's round trip would come first, every iteration of a checked run)
    if (check && gst[b].done) return;
    const double v = row_tree8_d (a);
    // (nine dwords of arguments, all preloaded: the whole parameter block would put a scalar load of the output pointer in front of the kernel's only store)
    if (l == 0 && blockIdx.x * 16u + row < ntask) gl1[(size_t) b * ntask + task] = v;
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
    // a run whose flag is up has converged: the launch that found out has stored the final state (user-visible and host) and released the
    // sequence word itself, and the launches behind it carried nothing forward — the state slot this kernel would read is stale.  (The host
    // enqueues this kernel without knowing when all max_iterations launches went out at once.)
    // (everything the kernel may need is asked for at once — the block moments, the state slot as one vector load (lane j = dword j), the
    // run's flag — and the decisions are taken on what arrives: flag, `done` and `pending` one after the other, each a scalar round trip, stood
    // in front of the moment loads before: 4.5 us per dispatch, once per fixed-length graph and per run that ends at max_iterations)
    const double *mom = p.mom + ((size_t) b * 2 + p.slot) * ICP_NMOM * p.nb;
    double a0[8];
    fused_moment_loads<320> (mom, p.nb, 0u, a0);
    const uint32_t sv = state_load_lanes (sin);
    if (p.run_flag && p.run_flag[b] == p.epoch) return;
    const uint32_t sin_done = (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done));
    const uint32_t sin_pending = (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (pending));
    if ((p.check && sin_done) || !sin_pending) {
        if (threadIdx.x < 64) {
            const uint32_t t = threadIdx.x;
            uint32_t v = sv;
            if (t == offsetof (icp_reg_state, pending) / 4) v = 0u;
            if (t < sizeof (icp_reg_state) / 4) reinterpret_cast<uint32_t *> (st)[t] = v;
            state_to_host (p, b, t, v, (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k)), sin_done);
            seq_release (p, t);
        }
        return;
    }
    fused_finalize_block<32, 320, ROT> (p, mom, p.nb, 0u, sv, a0, &s_fin, s_l1, s_t, nullptr, nullptr, ff_no_hook (), 1u,
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
    if (icp_dense (p)) { icp_launch_owner_search_dense (p, s); return; }                     // (icp_search_dense.hip)
    hipLaunchKernelGGL ((k_search<true, false, 2, 16, true>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p);
}

void icp_launch_search (const icp_params &p, hipStream_t s)
{
    if (icp_dense (p)) { icp_launch_search_dense (p, s); return; }                           // (icp_search_dense.hip)
    if (p.fused) hipLaunchKernelGGL ((k_search<true, false, 2, 16>), dim3 (p.nb, p.batch), dim3 (1024), 0, s, KS_ARGS);
    else hipLaunchKernelGGL ((k_search<false, false, 2, 16>), dim3 (2 * p.nwg, p.batch), dim3 (1024), 0, s, KS_ARGS);
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
            hipLaunchKernelGGL (k_moment_level1, dim3 ((ICP_NMOM * ng + 15u) / 16u, p.batch), dim3 (256), 0, s, (const double *) p.mom, (const icp_reg_state *) p.st, p.ml1, p.nb, (uint32_t) p.check, p.ng_magic);
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
