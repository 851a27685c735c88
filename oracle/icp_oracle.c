/* icp_oracle.c — CPU ORACLE (test infrastructure, see icp_oracle.h for status and rules).
 *
 * Plain-C restatement of the reference's ICP iteration.  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fno-fast-math; -fopenmp only parallelises the per-query
 * search loops, whose results do not depend on the thread count).
 *
 * Canonical arithmetic: fp32 RN, no implicit FMA (explicit fmaf in the metric and the squared start; explicit
 * double fma in the finish of the fused moments, orc_moments_finish), W = 64.
 *   LDS tree of every reference reduction (kernels/icp_kernels.cl:170-175, 244-249, 319-324,
 *   396-405, 480-489, 551-560; kernels/reduce_kernels.cl:254-259):
 *       data[0..2W) filled, then for d = W, W/2, .., 1:  data[i] += data[i+d]  (i < d)
 */
#include "icp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WF  64u            /* CL_KERNEL_PREFERRED_WORK_GROUP_SIZE_MULTIPLE on GCN/CDNA */
#define WF2 (2u * WF)

/* ======================================================================================= */
/* canonical trees                                                                          */
/* ======================================================================================= */

static float tree_f (float *data)            /* data[2W], destroyed */
{
    for (uint32_t d = WF; d > 0; d >>= 1)
        for (uint32_t i = 0; i < d; ++i) data[i] = data[i] + data[i + d];
    return data[0];
}

static double tree_d (double *data)
{
    for (uint32_t d = WF; d > 0; d >>= 1)
        for (uint32_t i = 0; i < d; ++i) data[i] = data[i] + data[i + d];
    return data[0];
}

/* dot (v, (1,1,1,1)) canonicalised as ((x+y)+z)+w   */
static float  sum4_f (const float *v)  { return ((v[0] + v[1]) + v[2]) + v[3]; }

static void cross3 (const float *a, const float *b, float *c)
{   /* include/ICP/tests/helper_funcs.hpp:442-447 */
    c[0] = (a[1] * b[2]) - (a[2] * b[1]);
    c[1] = (a[2] * b[0]) - (a[0] * b[2]);
    c[2] = (a[0] * b[1]) - (a[1] * b[0]);
}

/* ======================================================================================= */
/* a14  getLMs  — kernels/icp_kernels.cl:63-76, twin helper_funcs.hpp:220-233               */
/* ======================================================================================= */
void orc_get_lms (const float *cloud, float *lms)
{
    for (uint32_t gY = 0; gY < 128; ++gY) {
        uint32_t yi = gY * 3 + 1;
        for (uint32_t gX = 0; gX < 128; ++gX) {
            /* float4 index (48+yi)*1280 + 128 + ((gX*2)>>1<<1<<2) + 2 (+1 for the colour half)
             * == pixel row 48+yi, column 64 + 4*gX + 1 */
            uint32_t row = 48 + yi, col = 64 + 4 * gX + 1;
            memcpy (lms + (size_t) (gY * 128 + gX) * 8, cloud + ((size_t) row * 640 + col) * 8,
                    8 * sizeof (float));
        }
    }
}

/* ======================================================================================= */
/* a1  getReps — kernels/icp_kernels.cl:97-114, host src/ICP/algorithms.cpp:852-857         */
/*     generalised: the 128 of the kernel becomes side = sqrt(m).       */
/* ======================================================================================= */
int orc_reps_grid (uint32_t m, uint32_t nr, uint32_t *nrx, uint32_t *nry, uint32_t *side)
{
    if (m == 0 || nr == 0 || nr > m) return -1;
    if (nr & (nr - 1)) return -1;                     /* power of two (log2 truncation otherwise) */
    uint32_t g = (uint32_t) floor (sqrt ((double) m) + 0.5);
    if ((uint64_t) g * g != m) return -1;             /* landmark grid side */
    uint32_t p = 0; while ((1u << (p + 1)) <= nr) ++p;
    uint32_t x = 1u << (p - p / 2), y = 1u << (p / 2);   /* algorithms.cpp:852-854 */
    if (g % x || g % y) return -1;
    *nrx = x; *nry = y; *side = g;
    return 0;
}

int orc_get_reps (const float *F, uint32_t m, uint32_t nr, float *R, uint32_t *rep_src)
{
    uint32_t nrx, nry, g;
    if (orc_reps_grid (m, nr, &nrx, &nry, &g)) return -1;
    uint32_t stepX = g / nrx, stepY = g / nry;
    for (uint32_t gY = 0; gY < nry; ++gY)
        for (uint32_t gX = 0; gX < nrx; ++gX) {
            uint32_t xi = gX * stepX + (stepX >> 1) - 1;      /* icp_kernels.cl:110-111 */
            uint32_t yi = gY * stepY + (stepY >> 1) - 1;
            if (stepX == 1) xi = gX;                           /* (step>>1)-1 underflows at step 1 */
            if (stepY == 1) yi = gY;
            uint32_t src = yi * g + xi;
            memcpy (R + (size_t) (gY * nrx + gX) * 8, F + (size_t) src * 8, 8 * sizeof (float));
            if (rep_src) rep_src[gY * nrx + gX] = src;
        }
    return 0;
}

/* ======================================================================================= */
/* a3  icpTransform_Quaternion — kernels/icp_kernels.cl:772-802, twin helper_funcs.hpp:451   */
/* ======================================================================================= */
static int g_threads = 1;      /* OpenMP team size of the current call (orc_icp_set_threads) */

static void transform_point (const float *T, const float *p, float *tp)
{
    const float *q = T;
    float q2[3] = { 2 * q[0], 2 * q[1], 2 * q[2] };
    float u[3]; cross3 (q, p, u);
    u[0] = u[0] + q[3] * p[0];
    u[1] = u[1] + q[3] * p[1];
    u[2] = u[2] + q[3] * p[2];
    float v[3]; cross3 (q2, u, v);
    tp[0] = T[7] * (p[0] + v[0]) + T[4];
    tp[1] = T[7] * (p[1] + v[1]) + T[5];
    tp[2] = T[7] * (p[2] + v[2]) + T[6];
}

void orc_transform_q (const float *M, float *tM, const float *T, uint32_t m)
{
    #pragma omp parallel for schedule(static) num_threads(g_threads) if (m >= 4096)
    for (uint32_t i = 0; i < m; ++i) {
        float tp[3]; transform_point (T, M + (size_t) i * 8, tp);
        tM[i * 8 + 0] = tp[0]; tM[i * 8 + 1] = tp[1]; tM[i * 8 + 2] = tp[2];
        for (int k = 3; k < 8; ++k) tM[i * 8 + k] = M[i * 8 + k];
    }
}

/* icpTransform_Quaternion_2 — kernels/icp_kernels.cl:842-879, twin helper_funcs.hpp:488 */
void orc_transform_q2 (const float *M, float *tM, const float *T, uint32_t m)
{
    const float *q = T;
    float Q[4][4] = { {  q[3], -q[2],  q[1], q[0] }, {  q[2],  q[3], -q[0], q[1] },
                      { -q[1],  q[0],  q[3], q[2] }, { -q[0], -q[1], -q[2], q[3] } };
    float Q_[3][4] = { {  q[3], -q[2],  q[1], -q[0] }, {  q[2],  q[3], -q[0], -q[1] },
                       { -q[1],  q[0],  q[3], -q[2] } };
    for (uint32_t i = 0; i < m; ++i) {
        float p[4] = { M[i * 8], M[i * 8 + 1], M[i * 8 + 2], 0.f }, p_[4];
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = s + Q[r][k] * p[k];
            p_[r] = s;
        }
        for (int r = 0; r < 3; ++r) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = s + Q_[r][k] * p_[k];
            tM[i * 8 + r] = T[7] * s + T[4 + r];
        }
        for (int k = 3; k < 8; ++k) tM[i * 8 + k] = M[i * 8 + k];
    }
}

/* icpTransform_Matrix — kernels/icp_kernels.cl:904-933, twin helper_funcs.hpp:531 */
void orc_transform_m (const float *M, float *tM, const float *T16, uint32_t m)
{
    for (uint32_t i = 0; i < m; ++i) {
        for (int r = 0; r < 3; ++r) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = s + T16[r * 4 + k] * M[i * 8 + k];
            tM[i * 8 + r] = s;
        }
        for (int k = 3; k < 8; ++k) tM[i * 8 + k] = M[i * 8 + k];
    }
}

/* ======================================================================================= */
/* a2/a4  Random Ball Cover — ** PARITY UNPINNED ** (source: nlamprian/RandomBallCover,      */
/*        un-vendored, no pinned version; external/RandomBallCover/CMakeLists.txt:5-12).    */
/*        Published algorithm: L. Cayton, "Accelerating nearest neighbor search on manycore */
/*        systems", IPDPS 2012 — one-shot RBC.  Call sites: src/ICP/algorithms.cpp:4503-4536*/
/*        (wiring), :4659 (construct), :4674 (search).                                      */
/* ======================================================================================= */

/* ASSUMPTION-METRIC (single swap point; the GPU twin is icp_metric8 in icp_amd/csrc/icp_device.h).
 * Reference text: "|x-x'|^2 = f_g(a)|x_g-x'_g|^2 + f_p(a)|x_p-x'_p|^2, see euclideanSquaredMetric8"
 * (src/ICP/algorithms.cpp:4393-4398).  Lanes 3 and 7 (the homogeneous 1s) are ignored. */
float orc_metric8 (const float *x, const float *y, float a)
{   /* explicit fused multiply-adds (one rounding each), the form a GPU compiler gives OpenCL's dot();
     * everything else in this file is built with -ffp-contract=off */
    float dx = x[0] - y[0], dy = x[1] - y[1], dz = x[2] - y[2];
    float dr = x[4] - y[4], dg = x[5] - y[5], db = x[6] - y[6];
    float g = fmaf (dz, dz, fmaf (dy, dy, dx * dx));
    float p = fmaf (db, db, fmaf (dg, dg, dr * dr));
    return fmaf (a, p, g);
}


/* owner(x) = argmin_r d(x, R[r]), ties -> lowest r */
static uint32_t nearest_rep (const float *x, const float *R, uint32_t nr, float a, float *dist)
{
    /* NaN / +inf distances never win a '<' (DESIGN.md §3 item 4), the first representative's included: the scan starts from +inf, not from
     * the first distance — with finite data the same bits (a first distance of +inf left best = +inf, bid = 0 before as well) */
    float best = INFINITY; uint32_t bid = 0;
    for (uint32_t r = 0; r < nr; ++r) {
        float d = orc_metric8 (x, R + (size_t) r * 8, a);
        if (d < best) { best = d; bid = r; }
    }
    if (dist) *dist = best;
    return bid;
}

void orc_exscan_u32 (const uint32_t *in, uint32_t n, uint32_t *out)
{   /* exclusiveScan_i semantics (kernels/scan_kernels.cl:188; twin helper_funcs.hpp:200-209) */
    uint32_t run = 0;
    for (uint32_t i = 0; i < n; ++i) { uint32_t v = in[i]; out[i] = run; run += v; }
}

/* RBC construct: owner per fixed point, list sizes N, offsets O = exscan(N), database stably
 * permuted by owner (original index order inside each list), perm[pos] = original index. */
void orc_rbc_construct (const float *F, uint32_t m, const float *R, uint32_t nr, float a,
                        uint32_t *owner, uint32_t *N, uint32_t *O, uint32_t *perm, float *XP)
{
    memset (N, 0, nr * sizeof (uint32_t));
    #pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t i = 0; i < (int64_t) m; ++i)
        owner[i] = nearest_rep (F + (size_t) i * 8, R, nr, a, NULL);
    for (uint32_t i = 0; i < m; ++i) N[owner[i]]++;
    orc_exscan_u32 (N, nr, O);
    uint32_t *cur = (uint32_t *) malloc (nr * sizeof (uint32_t));
    memcpy (cur, O, nr * sizeof (uint32_t));
    for (uint32_t i = 0; i < m; ++i) {
        uint32_t pos = cur[owner[i]]++;
        perm[pos] = i;
        if (XP) memcpy (XP + (size_t) pos * 8, F + (size_t) i * 8, 8 * sizeof (float));
    }
    free (cur);
}

/* RBC one-shot search: nearest representative, then exhaustive scan of its list.
 * Outputs in QUERY order: nn_id[i] = { d(q_i, nn), original fixed index }, NN[i] = the NN point.
 * Ties -> lowest list position (= lowest original index).  Empty list -> the representative. */
void orc_rbc_search (const float *Q, uint32_t nq, const float *R, uint32_t nr,
                     const float *XP, const uint32_t *perm, const uint32_t *O,
                     const uint32_t *N, const uint32_t *rep_src, float a,
                     orc_dist_id *nn_id, float *NN, uint32_t *rid)
{
    #pragma omp parallel for schedule(dynamic, 64) num_threads(g_threads)
    for (int64_t i = 0; i < (int64_t) nq; ++i) {
        const float *q = Q + (size_t) i * 8;
        float dr; uint32_t r = nearest_rep (q, R, nr, a, &dr);
        if (rid) rid[i] = r;
        uint32_t o = O[r], n = N[r];
        if (n == 0) {
            nn_id[i].dist = dr; nn_id[i].id = rep_src ? rep_src[r] : 0xFFFFFFFFu;
            if (NN) memcpy (NN + (size_t) i * 8, R + (size_t) r * 8, 8 * sizeof (float));
            continue;
        }
        float best = INFINITY; uint32_t bj = o;                     /* (as nearest_rep: no candidate that compares -> +inf and the list's first member) */
        for (uint32_t j = o; j < o + n; ++j) {
            float d = orc_metric8 (q, XP + (size_t) j * 8, a);
            if (d < best) { best = d; bj = j; }
        }
        nn_id[i].dist = best; nn_id[i].id = perm[bj];
        if (NN) memcpy (NN + (size_t) i * 8, XP + (size_t) bj * 8, 8 * sizeof (float));
    }
}

/* exact NN (no RBC) — used by tests to measure how approximate the one-shot search is */
void orc_nn_brute (const float *Q, uint32_t nq, const float *F, uint32_t m, float a,
                   orc_dist_id *nn_id)
{
    #pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t i = 0; i < (int64_t) nq; ++i) {
        float d; uint32_t id = nearest_rep (Q + (size_t) i * 8, F, m, a, &d);
        nn_id[i].dist = d; nn_id[i].id = id;
    }
}

/* ======================================================================================= */
/* a5  icpComputeReduceWeights(_WG) + reduce_sum_fd — kernels/icp_kernels.cl:139-180,       */
/*     213-254, 295-329; host src/ICP/algorithms.cpp:1038-1075, 1151-1154                   */
/* ======================================================================================= */
void orc_weights (const orc_dist_id *D, uint32_t n, float *W_, double *sum_w)
{
    uint32_t wg = (n + WF2 - 1) / WF2;                         /* algorithms.cpp:1038 */
    uint32_t wgp = wg;
    if (wgp != 1 && (wgp % 4)) wgp += 4 - wgp % 4;           /* :1040 */
    float *part = (float *) calloc (wgp, sizeof (float));
    float data[WF2];
    for (uint32_t g = 0; g < wgp; ++g) {
        for (uint32_t p = 0; p < WF2; p += 2) {
            uint32_t idx = g * WF2 + p; float a = 0.f, b = 0.f;
            if (idx < n) {                                   /* one flag guards the pair (n even) */
                a = 100.f / (100.f + D[idx].dist);     W_[idx] = a;
                b = 100.f / (100.f + D[idx + 1].dist); W_[idx + 1] = b;
            }
            data[p] = a; data[p + 1] = b;
        }
        part[g] = tree_f (data);
    }
    if (wgp == 1) { *sum_w = (double) part[0]; free (part); return; }   /* icp_kernels.cl:179 */
    /* reduce_sum_fd: one work-group covers 2W float4 = 512 partials.  Beyond that (m > 65536,
     * which the reference rejects at algorithms.cpp:1054) chunks of 512 partials are summed in
     * index order in double (build's generalisation). */
    double total = 0.0; int first = 1;
    for (uint32_t c0 = 0; c0 < wgp; c0 += 4 * WF2) {
        double dd[WF2];
        for (uint32_t p = 0; p < WF2; ++p) {
            uint32_t i4 = c0 + 4 * p;
            if (i4 < wgp) {
                double x = (double) part[i4], y = (double) part[i4 + 1];
                double z = (double) part[i4 + 2], w = (double) part[i4 + 3];
                dd[p] = ((x + y) + z) + w;
            } else dd[p] = 0.0;
        }
        double cs = tree_d (dd);
        if (first) { total = cs; first = 0; } else total = total + cs;
    }
    *sum_w = total;
    free (part);
}

/* ======================================================================================= */
/* a6  icpMean / icpMean_Weighted + icpGMean — kernels/icp_kernels.cl:371-411, 455-495,     */
/*     530-566; host src/ICP/algorithms.cpp:1563-1594, 1695-1698                            */
/* ======================================================================================= */
static void gmean3 (float *blk, uint32_t nblk, float *out3)
{   /* icpGMean applied until one vector remains (reference: exactly one pass, nblk <= 128) */
    float data[3][WF2];
    while (nblk > 1) {
        uint32_t ng = (nblk + WF2 - 1) / WF2;
        for (uint32_t g = 0; g < ng; ++g) {
            for (uint32_t p = 0; p < WF2; ++p) {
                uint32_t i = g * WF2 + p;
                for (int k = 0; k < 3; ++k) data[k][p] = (i < nblk) ? blk[i * 3 + k] : 0.f;
            }
            for (int k = 0; k < 3; ++k) blk[g * 3 + k] = tree_f (data[k]);
        }
        nblk = ng;
    }
    out3[0] = blk[0]; out3[1] = blk[1]; out3[2] = blk[2];
}

static void mean_impl (const float *F, const float *M, const float *Wt, double sum_w,
                       uint32_t n, float *mean8)
{
    uint32_t wg = (n + WF2 - 1) / WF2;
    float *blk = (float *) malloc ((size_t) wg * 3 * sizeof (float));
    float data[3][WF2];
    const float *SET[2] = { F, M };
    for (int s = 0; s < 2; ++s) {
        const float *in = SET[s];
        for (uint32_t g = 0; g < wg; ++g) {
            for (uint32_t p = 0; p < WF2; ++p) {
                uint32_t idx = g * WF2 + p;
                uint32_t pair0 = idx & ~1u;                       /* flag of the pair's first element */
                for (int k = 0; k < 3; ++k) {
                    float v = 0.f;
                    if (pair0 < n) {
                        if (Wt) {
                            float kf = (float) ((double) Wt[idx] / sum_w);   /* icp_kernels.cl:475 */
                            v = kf * in[(size_t) idx * 8 + k];
                        } else
                            v = in[(size_t) idx * 8 + k] / (float) n;        /* icp_kernels.cl:391 */
                    }
                    data[k][p] = v;
                }
            }
            for (int k = 0; k < 3; ++k) blk[g * 3 + k] = tree_f (data[k]);
        }
        gmean3 (blk, wg, mean8 + 4 * s);
        mean8[4 * s + 3] = 0.f;
    }
    free (blk);
}

void orc_mean (const float *F, const float *M, uint32_t n, float *mean8)
{ mean_impl (F, M, NULL, 0.0, n, mean8); }

void orc_mean_weighted (const float *F, const float *M, const float *Wt, double sum_w,
                        uint32_t n, float *mean8)
{ mean_impl (F, M, Wt, sum_w, n, mean8); }

/* ======================================================================================= */
/* a7  icpSubtractMean — kernels/icp_kernels.cl:588-602                                     */
/* ======================================================================================= */
void orc_devs (const float *F, const float *M, const float *mean8, uint32_t n,
               float *DF, float *DM)
{
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) {
            DF[i * 4 + k] = F[(size_t) i * 8 + k] - mean8[k];
            DM[i * 4 + k] = M[(size_t) i * 8 + k] - mean8[4 + k];
        }
}

/* ======================================================================================= */
/* reduce_sum_f — kernels/reduce_kernels.cl:230-264; host src/ICP/algorithms.cpp:131-173    */
/* rows x cols (row-major, cols padded with zeros to a multiple of 4) -> rows floats.       */
/* ======================================================================================= */
void orc_reduce_sum_f (const float *in, uint32_t cols, uint32_t rows, float *out)
{
    uint32_t c4 = (cols + 3) & ~3u;
    float *cur = (float *) calloc ((size_t) rows * c4, sizeof (float));
    for (uint32_t r = 0; r < rows; ++r) memcpy (cur + (size_t) r * c4, in + (size_t) r * cols, cols * sizeof (float));
    uint32_t ccols = c4;
    float data[WF2];
    for (;;) {
        uint32_t wg = (ccols + 8 * WF - 1) / (8 * WF);                 /* algorithms.cpp:140 */
        uint32_t wgp = wg; if (wgp != 1 && (wgp % 4)) wgp += 4 - wgp % 4;   /* :142 */
        float *nxt = (float *) calloc ((size_t) rows * wgp, sizeof (float));
        for (uint32_t r = 0; r < rows; ++r)
            for (uint32_t g = 0; g < wgp; ++g) {
                for (uint32_t p = 0; p < WF2; ++p) {
                    uint32_t c = g * 8 * WF + 4 * p;
                    data[p] = (c < ccols) ? sum4_f (cur + (size_t) r * ccols + c) : 0.f;
                }
                nxt[(size_t) r * wgp + g] = tree_f (data);
            }
        free (cur); cur = nxt; ccols = wgp;
        if (wgp == 1) break;
    }
    for (uint32_t r = 0; r < rows; ++r) out[r] = cur[r];
    free (cur);
}

/* ======================================================================================= */
/* a8  icpSijProducts(_Weighted) — kernels/icp_kernels.cl:633-671, 703-743;                 */
/*     host src/ICP/algorithms.cpp:2344-2362 (G = ceil4(m)/4 work-items)                     */
/*     NOTE kernel order: S[9] = sum w|Fp|^2, S[10] = sum w|Mp|^2 (icp_kernels.cl:669-670);  */
/*     the twin cpuICPS swaps them (helper_funcs.hpp:406-407).                              */
/* ======================================================================================= */
void orc_sij (const float *DM, const float *DF, const float *Wt, uint32_t m, float c, float *S11)
{
    uint32_t n = m; if (n % 4) n += 4 - n % 4; n /= 4;
    float *Sij = (float *) calloc ((size_t) 11 * n, sizeof (float));
    for (uint32_t g = 0; g < n; ++g) {
        float A[11]; for (int k = 0; k < 11; ++k) A[k] = 0.f;
        for (uint32_t pi = g; pi < m; pi += n) {
            float Mp[3] = { c * DM[pi * 4], c * DM[pi * 4 + 1], c * DM[pi * 4 + 2] };
            float Fp[3] = { c * DF[pi * 4], c * DF[pi * 4 + 1], c * DF[pi * 4 + 2] };
            float ff = (Fp[0] * Fp[0] + Fp[1] * Fp[1]) + Fp[2] * Fp[2];
            float mm = (Mp[0] * Mp[0] + Mp[1] * Mp[1]) + Mp[2] * Mp[2];
            if (Wt) {
                float w = Wt[pi];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) A[a * 3 + b] = A[a * 3 + b] + w * (Mp[a] * Fp[b]);
                A[9] = A[9] + w * ff; A[10] = A[10] + w * mm;
            } else {
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) A[a * 3 + b] = A[a * 3 + b] + Mp[a] * Fp[b];
                A[9] = A[9] + ff; A[10] = A[10] + mm;
            }
        }
        for (int k = 0; k < 11; ++k) Sij[(size_t) k * n + g] = A[k];
    }
    orc_reduce_sum_f (Sij, n, 11, S11);
    free (Sij);
}

/* ======================================================================================= */
/* a9  icpPowerMethod — kernels/icp_kernels.cl:977-1054; twin helper_funcs.hpp:682-764.     */
/*     fast_normalize / fast_distance / normalize are implementation-defined in OpenCL; the */
/*     canonical forms are the twin's (sqrt of the sequential sum of squares, 4 divides).   */
/*     `error_new` is read uninitialised in both kernel (:1008,1018) and twin; canonical     */
/*     initial value +inf (never equal to a finite distance).                               */
/* ======================================================================================= */
static float dot4 (const float *a, const float *b)
{   /* std::inner_product (a, a+4, b, 0.f) */
    float s = 0.f;
    s = s + a[0] * b[0]; s = s + a[1] * b[1]; s = s + a[2] * b[2]; s = s + a[3] * b[3];
    return s;
}

static void prod4 (const float *N, const float *x, float *y)
{ for (int r = 0; r < 4; ++r) y[r] = dot4 (N + 4 * r, x); }

static void normalize4 (float *x)
{
    float s = 0.f;
    s += x[0] * x[0]; s += x[1] * x[1]; s += x[2] * x[2]; s += x[3] * x[3];
    float n = sqrtf (s);
    x[0] /= n; x[1] /= n; x[2] /= n; x[3] /= n;
}

static float distance4 (const float *a, const float *b)
{
    float s = 0.f, d;
    d = a[0] - b[0]; s += d * d; d = a[1] - b[1]; s += d * d;
    d = a[2] - b[2]; s += d * d; d = a[3] - b[3]; s += d * d;
    return sqrtf (s);
}

static float distance4_sq (const float *a, const float *b)
{   /* the same sum without the root (squared-start stop rule) */
    float s = 0.f, d;
    d = a[0] - b[0]; s += d * d; d = a[1] - b[1]; s += d * d;
    d = a[2] - b[2]; s += d * d; d = a[3] - b[3]; s += d * d;
    return s;
}

static void build_N (const float *S, float *N)
{   /* icp_kernels.cl:993-999 */
    float Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5],
          Szx = S[6], Szy = S[7], Szz = S[8];
    float n[16] = {
        Sxx - Syy - Szz,       Sxy + Syx,         Szx + Sxz,       Syz - Szy,
              Sxy + Syx, - Sxx + Syy - Szz,       Syz + Szy,       Szx - Sxz,
              Szx + Sxz,       Syz + Szy, - Sxx - Syy + Szz,       Sxy - Syx,
              Syz - Szy,       Szx - Sxz,         Sxy - Syx, Sxx + Syy + Szz };
    memcpy (N, n, sizeof n);
}

static void finish_Tk (const float *S, const float *means, const float *qk, float *Tk)
{   /* icp_kernels.cl:989, 1045-1053; twin helper_funcs.hpp:749-763 */
    float sk = sqrtf (S[9] / S[10]);
    const float *mf = means, *mm = means + 4;
    float q2[3] = { 2 * qk[0], 2 * qk[1], 2 * qk[2] };
    float cp1[3]; cross3 (qk, mm, cp1);
    float t1[3] = { cp1[0] + qk[3] * mm[0], cp1[1] + qk[3] * mm[1], cp1[2] + qk[3] * mm[2] };
    float cp2[3]; cross3 (q2, t1, cp2);
    Tk[0] = qk[0]; Tk[1] = qk[1]; Tk[2] = qk[2]; Tk[3] = qk[3];
    Tk[4] = mf[0] - sk * (mm[0] + cp2[0]);
    Tk[5] = mf[1] - sk * (mm[1] + cp2[1]);
    Tk[6] = mf[2] - sk * (mm[2] + cp2[2]);
    Tk[7] = sk;
}

/* Exact power-of-two rescale of a 4x4 so that max|entry| lies in [1,2) — no rounding. */
static void rescale16 (float *B)
{
    float mx = 0.f;
    for (int i = 0; i < 16; ++i) { float a = fabsf (B[i]); if (a > mx) mx = a; }
    uint32_t bits; memcpy (&bits, &mx, 4);
    uint32_t e = (bits >> 23) & 0xFFu;
    if (e == 0 || e == 0xFFu) return;                /* zero / subnormal / inf / nan: leave */
    uint32_t sb = (254u - e) << 23;                  /* 2^(127-e) */
    if (254u - e == 0 || 254u - e >= 255u) return;
    float sc; memcpy (&sc, &sb, 4);
    for (int i = 0; i < 16; ++i) B[i] = B[i] * sc;
}

#define PM_SQUARINGS 10

/* The reference's power method, literally (icp_kernels.cl:977-1054). */
static int power_literal (const float *S, const float *means, float *Tk)
{
    float N[16]; build_N (S, N);
    float x[4] = { 1.f, 1.f, 1.f, 1.f }, xn[4];
    int iters = 0;
    for (;;) {
        float error, error_new = INFINITY;
        for (uint32_t it = 0; it < 1000; ++it) {            /* icp_kernels.cl:1012-1022 */
            prod4 (N, x, xn);
            normalize4 (xn);
            ++iters;
            error = error_new;
            error_new = distance4 (x, xn);
            if (error_new == error) break;                  /* stop when the step length repeats (:1019) */
            memcpy (x, xn, sizeof x);
        }
        float lambda = dot4 (N, xn) / xn[0];                 /* :1024 */
        if (lambda < 0) {
            N[0] -= lambda; N[5] -= lambda; N[10] -= lambda; N[15] -= lambda;
            x[0] = x[1] = x[2] = x[3] = 1.f;
        } else break;
    }
    memcpy (x, xn, sizeof x);                                /* :1039-1041 */
    prod4 (N, x, xn);
    normalize4 (xn);
    finish_Tk (S, means, xn, Tk);
    return iters;
}

/* The build's accelerated power method (the GPU twin is icp_power_method_quad, squared start).
 *   B = N^(2^PM_SQUARINGS) by repeated squaring (k-ordered fmaf chains, exact power-of-two rescaling every fifth
 *   squaring); u = B 1 (not normalised), v = N u; a fast exit decided on (u, v) (see there); otherwise
 *   x = normalize (u) and xn = normalize (v) (independent of each other);
 *   then the reference's loop on SQUARED step lengths: it goes on only while the step is above 2^-22 (one ulp of a
 *   unit vector) and still decreasing.  The sign test of :1024 divides only when it has to shift; the vector the
 *   loop ends with is the result (the reference's extra pass after its loop, :1039-1041, is what the loop's last
 *   trip already is here). */
static int power_fast (const float *S, const float *means, float *Tk)
{
    float N[16]; build_N (S, N);
    float x[4], xn[4], u[4];
    const float ones[4] = { 1.f, 1.f, 1.f, 1.f };
    int iters = 0, shifted = 0;
    for (;;) {
        float B[16], C[16]; memcpy (B, N, sizeof B); rescale16 (B);
        for (int s = 0; s < PM_SQUARINGS; ++s) {
            /* C = B B as a k-ordered fmaf chain (what v_mfma_f32_4x4x1 evaluates: one rounding per step) */
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.f;
                    for (int k = 0; k < 4; ++k) acc = fmaf (B[i * 4 + k], B[k * 4 + j], acc);
                    C[i * 4 + j] = acc;
                }
            memcpy (B, C, sizeof B);
            if (s % 5 == 4) rescale16 (B);          /* max|entry| < 2 after a rescale, < 2^94 five squarings later */
        }
        prod4 (B, ones, u);
        float v[4]; prod4 (N, u, v);
        ++iters;
        /* Fast exit, decided on the unnormalised pair (u, v = N u) — nothing on the way to the result waits for it:
         * |u x v|^2 = sum over the six index pairs of (u_i v_j - u_j v_i)^2 (Lagrange's identity: no cancellation between
         * large terms) against 2^-44 (u.u)(v.v), i.e. sin^2 of the angle between two successive iterates against the
         * square of one ulp of a unit vector; the sign of the eigenvalue from the Rayleigh quotient u.v / u.u.  False on
         * NaN like the loop's own test below.  Converged and positive: the result is normalize (N u), what the general
         * path below ends with in this case. */
        {
            float c2 = 0.f, t;
            t = u[0] * v[1] - u[1] * v[0]; c2 += t * t;  t = u[0] * v[2] - u[2] * v[0]; c2 += t * t;
            t = u[0] * v[3] - u[3] * v[0]; c2 += t * t;  t = u[1] * v[2] - u[2] * v[1]; c2 += t * t;
            t = u[1] * v[3] - u[3] * v[1]; c2 += t * t;  t = u[2] * v[3] - u[3] * v[2]; c2 += t * t;
            float uu = 0.f, vv = 0.f, uv = 0.f;
            for (int k = 0; k < 4; ++k) { uu += u[k] * u[k]; vv += v[k] * v[k]; uv += u[k] * v[k]; }
            if (!(c2 > 0x1p-44f * (uu * vv))) {
                if (uv < 0.f) {                              /* negative dominant eigenvalue: shift by it (:1024-1037), start over */
                    const float lambda = uv / uu;
                    N[0] -= lambda; N[5] -= lambda; N[10] -= lambda; N[15] -= lambda;
                    continue;
                }
                memcpy (xn, v, sizeof xn); normalize4 (xn);
                break;
            }
            /* Not converged after 1024 steps: two eigenvalues of (nearly) the same magnitude.  The case that occurs is a PLANAR scene (the
             * reference's kg_pc8d_wall, data/README.md:11-16): S has rank 2, and N's eigenvalues come in pairs +-lambda — the power method,
             * squared or not, cannot separate +lambda_max from -lambda_max (an even power least of all), and the literal loop spends its
             * 1000 trips on a mixture of the two eigenvectors.  N is symmetric with trace 0: N + sigma I with sigma = the largest absolute
             * row sum (>= every |eigenvalue|) has the same eigenvectors, every eigenvalue >= 0 and lambda_max + sigma on top alone.  Once per
             * solve; the squaring starts over on the shifted matrix (rows summed left to right, the maximum is exact in any order). */
            if (!shifted) {
                float sigma = 0.f;
                for (int i = 0; i < 4; ++i) {
                    const float rs = ((fabsf (N[4 * i]) + fabsf (N[4 * i + 1])) + fabsf (N[4 * i + 2])) + fabsf (N[4 * i + 3]);
                    if (rs > sigma) sigma = rs;
                }
                shifted = 1;
                if (sigma > 0.f && sigma < INFINITY) {
                    N[0] += sigma; N[5] += sigma; N[10] += sigma; N[15] += sigma;
                    continue;
                }
            }
        }
        memcpy (x, u, sizeof x); normalize4 (x);
        memcpy (xn, v, sizeof xn); normalize4 (xn);
        float e2_prev = INFINITY, e2 = distance4_sq (x, xn);
        while (e2 > 0x1p-44f && e2 < e2_prev && iters < 1000) {       /* false on NaN: the loop ends */
            memcpy (x, xn, sizeof x);
            prod4 (N, x, xn); normalize4 (xn);
            ++iters;
            e2_prev = e2; e2 = distance4_sq (x, xn);
        }
        const float lam_num = dot4 (N, xn), den = xn[0];              /* :1024, lambda = lam_num / den */
        if ((lam_num < 0.f && den > 0.f) || (lam_num > 0.f && den < 0.f)) {
            const float lambda = lam_num / den;
            N[0] -= lambda; N[5] -= lambda; N[10] -= lambda; N[15] -= lambda;
        } else break;
    }
    finish_Tk (S, means, xn, Tk);
    return iters;
}

static int power_impl (const float *S, const float *means, float *Tk, int fast)
{ return fast ? power_fast (S, means, Tk) : power_literal (S, means, Tk); }

int orc_power_method (const float *S11, const float *mean8, float *Tk8)
{ return power_impl (S11, mean8, Tk8, 0); }

int orc_power_method_fast (const float *S11, const float *mean8, float *Tk8)
{ return power_impl (S11, mean8, Tk8, 1); }

/* ======================================================================================= */
/* a10  Eigen pieces restated by hand (Eigen 3.2.4 is un-vendored: external/Eigen/           */
/*      CMakeLists.txt:7-8; call sites src/ICP/algorithms.cpp:4683-4694).                   */
/*      Quaternion::toRotationMatrix and the matrix->quaternion constructor follow Eigen's  */
/*      published formulas; 3-term sums are canonicalised as (a0*b0 + a1*b1) + a2*b2.       */
/* ======================================================================================= */
void orc_quat_to_rot (const float *q, float *R)
{
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    float twx = tx * w, twy = ty * w, twz = tz * w;
    float txx = tx * x, txy = ty * x, txz = tz * x;
    float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void orc_rot_to_quat (const float *m, float *q)
{
    float t = (m[0] + m[4]) + m[8];
    if (t > 0.f) {
        t = sqrtf (t + 1.f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf (((m[i * 4] - m[j * 4]) - m[k * 4]) + 1.f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

static float dot3c (const float *a, int sa, const float *b, int sb)
{ return (a[0] * b[0] + a[sa] * b[sb]) + a[2 * sa] * b[2 * sb]; }

/* ======================================================================================= */
/* a12  EIGEN branch — src/ICP/algorithms.cpp:3867-3909: JacobiSVD of S, Rk = V U^T with    */
/*      det fix.  Restated as a one-sided (Hestenes) Jacobi SVD in fp32 (Eigen un-vendored). */
/* ======================================================================================= */
void orc_svd_rotation (const float *S11, const float *means, float *Rk, float *Tk)
{
    /* Eigen maps S row-major: S(a,b) = S11[3a+b], a = moving component, b = fixed component */
    float A[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy (A, S11, sizeof A);
    for (int sweep = 0; sweep < 30; ++sweep) {
        float off = 0.f;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                float alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i * 3 + p] * A[i * 3 + p];
                    beta  += A[i * 3 + q] * A[i * 3 + q];
                    gamma += A[i * 3 + p] * A[i * 3 + q];
                }
                if (gamma == 0.f) continue;
                off = fmaxf (off, fabsf (gamma) / sqrtf (alpha * beta));
                float zeta = (beta - alpha) / (2.f * gamma);
                float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf (zeta) + sqrtf (1.f + zeta * zeta));
                float cs = 1.f / sqrtf (1.f + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    float ap = A[i * 3 + p], aq = A[i * 3 + q];
                    A[i * 3 + p] = cs * ap - sn * aq; A[i * 3 + q] = sn * ap + cs * aq;
                    float vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = cs * vp - sn * vq; V[i * 3 + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-7f) break;
    }
    /* A = U * diag(sigma): columns of A normalised give U */
    float U[9], sig[3];
    for (int j = 0; j < 3; ++j) {
        sig[j] = sqrtf ((A[j] * A[j] + A[3 + j] * A[3 + j]) + A[6 + j] * A[6 + j]);
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = sig[j] > 0.f ? A[i * 3 + j] / sig[j] : 0.f;
    }
    /* smallest singular value last matters only for the det fix: find its column */
    int smin = 0; for (int j = 1; j < 3; ++j) if (sig[j] < sig[smin]) smin = j;
    /* Rk = V * U^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rk[i * 3 + j] = (V[i * 3] * U[j * 3] + V[i * 3 + 1] * U[j * 3 + 1]) + V[i * 3 + 2] * U[j * 3 + 2];
    float det = Rk[0] * (Rk[4] * Rk[8] - Rk[5] * Rk[7]) - Rk[1] * (Rk[3] * Rk[8] - Rk[5] * Rk[6])
              + Rk[2] * (Rk[3] * Rk[7] - Rk[4] * Rk[6]);
    if (det < 0.f) {      /* algorithms.cpp:3889-3894: B = diag(1,1,det) on the smallest sigma */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float acc = 0.f;
                for (int k = 0; k < 3; ++k)
                    acc += V[i * 3 + k] * (k == smin ? det : 1.f) * U[j * 3 + k];
                Rk[i * 3 + j] = acc;
            }
    }
    float qk[4]; orc_rot_to_quat (Rk, qk);
    float sk = sqrtf (S11[9] / S11[10]);
    const float *mf = means, *mm = means + 4;
    Tk[0] = qk[0]; Tk[1] = qk[1]; Tk[2] = qk[2]; Tk[3] = qk[3];
    for (int i = 0; i < 3; ++i)                     /* tk = mf - sk * Rk * mm  (:3897) */
        Tk[4 + i] = mf[i] - ((sk * Rk[i * 3]) * mm[0] + (sk * Rk[i * 3 + 1]) * mm[1] + (sk * Rk[i * 3 + 2]) * mm[2]);
    Tk[7] = sk;
}

/* ======================================================================================= */
/* FUSED reduction mode (build's single-pass formulation).                  */
/*   The reference chains three global reductions (sum of weights -> means -> S,            */
/*   kernels/icp_kernels.cl:213-329, 455-566, 588-743).  Algebraically                       */
/*       mean_f = sum(w f)/sum(w),   S_ab = c^2 (sum(w q_a f_b) - sum(w q_a) mean_f_b), ...   */
/*   so one pass over the pairs that accumulates 18 moments in DOUBLE gives the same means,  */
/*   S and scale terms (to ~1e-15 relative before the final rounding to float), with one     */
/*   global reduction instead of three.  Canonical order: blocks of 64 pairs (8x8 tiles of   */
/*   the landmark grid when its side is a multiple of 8, else 64 consecutive pairs), halving */
/*   tree inside a block, then the 128-position tree over block partials until one remains.  */
/* ======================================================================================= */
#define NMOM 18

static double tree64_d (double *data)
{
    for (uint32_t d = 32; d > 0; d >>= 1)
        for (uint32_t i = 0; i < d; ++i) data[i] = data[i] + data[i + d];
    return data[0];
}

/* query index of local element e of block b (the GPU twin: fused_query_index in icp_kernels.hip) */
uint32_t orc_fused_query (uint32_t m, uint32_t side, uint32_t b, uint32_t e)
{
    if (side && (side % 8u) == 0 && (uint64_t) side * side == m) {
        uint32_t tpr = side / 8u, ty = b / tpr, tx = b % tpr;
        return (8u * ty + (e >> 3)) * side + 8u * tx + (e & 7u);
    }
    return b * 64u + e;
}

/* NN/tM are m x float8 (xyz used), W may be NULL (REGULAR: w = 1).  Outputs: sum_w, means[8], S[11]. */
void orc_moments_fused (const float *NN, const float *tM, const float *Wt, uint32_t m, uint32_t side,
                        float c, double *sum_w, float *mean8, float *S11)
{
    uint32_t nb = (m + 63u) / 64u;
    double *part = (double *) calloc ((size_t) nb * NMOM, sizeof (double));
    /* the block partials are independent of each other (a fixed tree per block): any thread may compute any block — deterministic
     * per-thread partials, the same bits as the serial loop (bench.py's cpu_baseline runs this on all host cores) */
    #pragma omp parallel for schedule(static) num_threads(g_threads) if (nb >= 64)
    for (uint32_t b = 0; b < nb; ++b) {
        double data[NMOM][64];
        for (uint32_t e = 0; e < 64; ++e) {
            uint32_t i = orc_fused_query (m, side, b, e);
            double t[NMOM];
            for (int k = 0; k < NMOM; ++k) t[k] = 0.0;
            if (i < m) {
                double w = Wt ? (double) Wt[i] : 1.0;
                double f[3] = { NN[(size_t) i * 8], NN[(size_t) i * 8 + 1], NN[(size_t) i * 8 + 2] };
                double q[3] = { tM[(size_t) i * 8], tM[(size_t) i * 8 + 1], tM[(size_t) i * 8 + 2] };
                t[0] = w;
                for (int a = 0; a < 3; ++a) { t[1 + a] = w * f[a]; t[4 + a] = w * q[a]; }
                for (int a = 0; a < 3; ++a)
                    for (int bb = 0; bb < 3; ++bb) t[7 + 3 * a + bb] = (w * q[a]) * f[bb];
                t[16] = w * ((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
                t[17] = w * ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
            }
            for (int k = 0; k < NMOM; ++k) data[k][e] = t[k];
        }
        for (int k = 0; k < NMOM; ++k) part[(size_t) b * NMOM + k] = tree64_d (data[k]);
    }
    /* 128-position tree over the block partials, repeated until one value per moment remains */
    uint32_t n = nb;
    double d128[WF2];
    while (n > 1) {
        uint32_t ng = (n + WF2 - 1) / WF2;
        for (uint32_t g = 0; g < ng; ++g)
            for (int k = 0; k < NMOM; ++k) {
                for (uint32_t p = 0; p < WF2; ++p) {
                    uint32_t i = g * WF2 + p;
                    d128[p] = (i < n) ? part[(size_t) i * NMOM + k] : 0.0;
                }
                double r = tree_d (d128);
                part[(size_t) g * NMOM + k] = r;      /* g <= i for every i of the group: in-place is safe per moment */
            }
        n = ng;
    }
    double t[NMOM];
    for (int k = 0; k < NMOM; ++k) t[k] = part[k];
    free (part);
    orc_moments_finish (t, c, sum_w, mean8, S11);
}

/* moments -> sum of weights, means (float), S (float) */
void orc_moments_finish (const double *t, float c, double *sum_w, float *mean8, float *S11)
{
    /* canonical form (the build's own: fused mode has no counterpart in the reference): ONE division, the means by
       multiplication, every product-and-add a fused multiply-add */
    double sw = t[0], rs = 1.0 / sw, mf[3], mq[3];
    for (int a = 0; a < 3; ++a) { mf[a] = t[1 + a] * rs; mq[a] = t[4 + a] * rs; }
    double c2 = (double) c * (double) c;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) S11[3 * a + b] = (float) (c2 * fma (-t[4 + a], mf[b], t[7 + 3 * a + b]));
    S11[9]  = (float) (c2 * (t[16] - fma (t[3], mf[2], fma (t[2], mf[1], t[1] * mf[0]))));
    S11[10] = (float) (c2 * (t[17] - fma (t[6], mq[2], fma (t[5], mq[1], t[4] * mq[0]))));
    for (int a = 0; a < 3; ++a) { mean8[a] = (float) mf[a]; mean8[4 + a] = (float) mq[a]; }
    mean8[3] = 0.f; mean8[7] = 0.f;
    *sum_w = sw;
}

/* ======================================================================================= */
/* pipeline object — ICPStep<CR,CW> / ICP<CR,CW>                                            */
/*   init   src/ICP/algorithms.cpp:4403-4582     buildRBC :4655-4660                        */
/*   run    :4670-4698 (POWER_METHOD), :3867-3909 (EIGEN)     ICP::run/check :4806-4834     */
/* ======================================================================================= */
struct orc_icp {
    int rot, weighted, fast, threads, fused;
    float dist_scale;            /* f_g of the metric text (src/ICP/algorithms.cpp:4393-4398): reported dist = f_g (geo + a pho) */
    uint32_t side;
    uint32_t m, nr, max_it, k;
    float a, c;
    double angle_thr, trans_thr, tan_half_thr;
    float *F, *M, *R, *XP, *tM, *NN, *W, *DF, *DM;
    uint32_t *rep_src, *owner, *N, *O, *perm, *rid;
    orc_dist_id *nn_id;
    double sum_w;
    float T[8], Tk[8], Rm[9], Rk[9], q[4], t[3], s, S[11], means[8];
    int pm_iters, converged;
};

orc_icp *orc_icp_create (int rot, int weighted)
{
    orc_icp *h = (orc_icp *) calloc (1, sizeof *h);
    h->rot = rot; h->weighted = weighted; h->threads = 1; h->dist_scale = 1.f;
    return h;
}

static void free_bufs (orc_icp *h)
{
    free (h->F); free (h->M); free (h->R); free (h->XP); free (h->tM); free (h->NN); free (h->W);
    free (h->DF); free (h->DM); free (h->rep_src); free (h->owner); free (h->N); free (h->O);
    free (h->perm); free (h->rid); free (h->nn_id);
}

void orc_icp_destroy (orc_icp *h) { if (h) { free_bufs (h); free (h); } }

static void reset_T (orc_icp *h)
{
    static const float T0[8] = { 0, 0, 0, 1, 0, 0, 0, 1 };     /* algorithms.cpp:4486 */
    static const float I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy (h->T, T0, sizeof T0); memcpy (h->Tk, T0, sizeof T0);
    memcpy (h->Rm, I3, sizeof I3); memcpy (h->Rk, I3, sizeof I3);
    h->q[0] = h->q[1] = h->q[2] = 0; h->q[3] = 1; h->t[0] = h->t[1] = h->t[2] = 0; h->s = 1.f;
}

int orc_icp_init (orc_icp *h, uint32_t m, uint32_t nr, float a, float c, uint32_t max_it,
                  double angle_thr, double trans_thr)
{
    uint32_t nrx, nry, g;
    if (m == 0 || nr == 0 || a == 0.f) return -1;             /* algorithms.cpp:4413-4420 */
    if (m % 2) return -1;                                      /* :1573 (means need even n) */
    if (orc_reps_grid (m, nr, &nrx, &nry, &g)) return -1;
    free_bufs (h);
    h->m = m; h->nr = nr; h->a = a; h->c = c; h->max_it = max_it; h->side = g;
    h->angle_thr = angle_thr; h->trans_thr = trans_thr;
    h->tan_half_thr = tan (angle_thr * M_PI / 360.0);
    size_t fm = (size_t) m * 8;
    h->F = calloc (fm, 4); h->M = calloc (fm, 4); h->XP = calloc (fm, 4); h->tM = calloc (fm, 4);
    h->NN = calloc (fm, 4); h->R = calloc ((size_t) nr * 8, 4); h->W = calloc (m, 4);
    h->DF = calloc ((size_t) m * 4, 4); h->DM = calloc ((size_t) m * 4, 4);
    h->rep_src = calloc (nr, 4); h->owner = calloc (m, 4); h->N = calloc (nr, 4);
    h->O = calloc (nr, 4); h->perm = calloc (m, 4); h->rid = calloc (m, 4);
    h->nn_id = calloc (m, sizeof (orc_dist_id));
    h->k = 0; h->converged = 0;
    reset_T (h);
    return 0;
}

void orc_icp_set_power_fast (orc_icp *h, int fast) { h->fast = fast; }
void orc_icp_set_fused (orc_icp *h, int fused) { h->fused = fused; }
void orc_icp_set_threads (orc_icp *h, int threads) { h->threads = threads > 0 ? threads : 1; }
void orc_icp_write_f (orc_icp *h, const float *F) { memcpy (h->F, F, (size_t) h->m * 32); }
void orc_icp_write_m (orc_icp *h, const float *M) { memcpy (h->M, M, (size_t) h->m * 32); }

void orc_icp_write_t (orc_icp *h, const float *T8)
{   /* write(D_IO_T): src/ICP/algorithms.cpp:4613-4617.  The host state (R,q,t,s) is derived
     * from the written T so that compose() continues from it (build's extension). */
    memcpy (h->T, T8, sizeof h->T);
    memcpy (h->q, T8, 16); memcpy (h->t, T8 + 4, 12); h->s = T8[7];
    orc_quat_to_rot (h->q, h->Rm);
}

void orc_icp_build_rbc (orc_icp *h)
{
    g_threads = h->threads;
    orc_get_reps (h->F, h->m, h->nr, h->R, h->rep_src);                         /* fReps.run */
    orc_rbc_construct (h->F, h->m, h->R, h->nr, h->a, h->owner, h->N, h->O, h->perm, h->XP);
    h->k = 0; h->converged = 0;                                                  /* ICP::buildRBC :4796 */
}

static void compose (orc_icp *h)
{   /* src/ICP/algorithms.cpp:4683-4695 */
    const float *Tk = h->Tk;
    float sk = Tk[7];
    if (h->rot == ORC_ROT_POWER) orc_quat_to_rot (Tk, h->Rk);   /* SVD branch filled Rk already */
    float Rn[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = dot3c (h->Rk + i * 3, 1, h->Rm + j, 3);
    memcpy (h->Rm, Rn, sizeof Rn);
    orc_rot_to_quat (h->Rm, h->q);
    float tn[3];
    for (int i = 0; i < 3; ++i) {
        float r0 = sk * h->Rk[i * 3], r1 = sk * h->Rk[i * 3 + 1], r2 = sk * h->Rk[i * 3 + 2];
        tn[i] = ((r0 * h->t[0] + r1 * h->t[1]) + r2 * h->t[2]) + Tk[4 + i];
    }
    memcpy (h->t, tn, sizeof tn);
    h->s = sk * h->s;
    memcpy (h->T, h->q, 16); memcpy (h->T + 4, h->t, 12); h->T[7] = h->s;
}

static int check_converged (const orc_icp *h)
{   /* ICP::check, src/ICP/algorithms.cpp:4824-4834.  2*atan2(|v|, w)*180/pi < thr is evaluated
     * as  w > 0 && |v| < w * tan(thr*pi/360)  (same predicate for thr < 180 deg; no libm atan2 in
     * the decision so that CPU and GPU agree bit for bit). */
    const float *qk = h->Tk, *tk = h->Tk + 4;
    float vn = sqrtf ((qk[0] * qk[0] + qk[1] * qk[1]) + qk[2] * qk[2]);
    float tn = sqrtf ((tk[0] * tk[0] + tk[1] * tk[1]) + tk[2] * tk[2]);
    int ang = (qk[3] > 0.f) && ((double) vn < (double) qk[3] * h->tan_half_thr);
    int tra = (double) tn < h->trans_thr;
    return ang && tra;
}

void orc_icp_step (orc_icp *h)
{
    g_threads = h->threads;
    orc_transform_q (h->M, h->tM, h->T, h->m);                                     /* transform.run */
    orc_rbc_search (h->tM, h->m, h->R, h->nr, h->XP, h->perm, h->O, h->N, h->rep_src, h->a,
                    h->nn_id, h->NN, h->rid);                                      /* rbcS.run      */
    /* absolute scale of the metric (ASSUMPTION-METRIC, second half): the search runs on geo + a pho (argmin and ties do not
     * depend on a positive common factor f_g); the distance it reports, which feeds the weights, is f_g times that */
    #pragma omp parallel for schedule(static) num_threads(g_threads) if (h->m >= 4096)
    for (uint32_t i = 0; i < h->m; ++i) h->nn_id[i].dist = h->dist_scale * h->nn_id[i].dist;
    if (h->fused) {
        if (h->weighted) {
            #pragma omp parallel for schedule(static) num_threads(g_threads) if (h->m >= 4096)
            for (uint32_t i = 0; i < h->m; ++i) h->W[i] = 100.f / (100.f + h->nn_id[i].dist);
        }
        orc_moments_fused (h->NN, h->tM, h->weighted ? h->W : NULL, h->m, h->side, h->c, &h->sum_w, h->means, h->S);
    } else {
    if (h->weighted) {
        orc_weights (h->nn_id, h->m, h->W, &h->sum_w);                             /* weights.run   */
        orc_mean_weighted (h->NN, h->tM, h->W, h->sum_w, h->m, h->means);          /* means.run     */
    } else
        orc_mean (h->NN, h->tM, h->m, h->means);
    orc_devs (h->NN, h->tM, h->means, h->m, h->DF, h->DM);                         /* devs.run      */
    orc_sij (h->DM, h->DF, h->weighted ? h->W : NULL, h->m, h->c, h->S);           /* matrixS.run   */
    }
    if (h->rot == ORC_ROT_POWER)
        h->pm_iters = h->fast ? orc_power_method_fast (h->S, h->means, h->Tk)      /* powMethod.run */
                              : orc_power_method (h->S, h->means, h->Tk);
    else
        orc_svd_rotation (h->S, h->means, h->Rk, h->Tk);
    compose (h);
    h->k++;
    h->converged = check_converged (h);
}

uint32_t orc_icp_run (orc_icp *h)
{   /* ICP::run: step; while (check()) step.  check(): k++ ; stop at k == max or converged. */
    h->k = 0;
    for (;;) {
        orc_icp_step (h);
        if (h->k >= h->max_it) break;
        if (h->converged) break;
    }
    return h->k;
}

void               orc_icp_set_dist_scale (orc_icp *h, float f_g) { h->dist_scale = f_g; }
void               orc_icp_set_alpha (orc_icp *h, float a) { h->a = a; }     /* setAlpha, src/ICP/algorithms.cpp:4712-4717 (the lists are rebuilt by the caller) */
int                orc_icp_converged (const orc_icp *h) { return h->converged; }
const float       *orc_icp_T (const orc_icp *h) { return h->T; }
const float       *orc_icp_Tk (const orc_icp *h) { return h->Tk; }
const float       *orc_icp_R (const orc_icp *h) { return h->Rm; }
const float       *orc_icp_Rk (const orc_icp *h) { return h->Rk; }
const float       *orc_icp_S (const orc_icp *h) { return h->S; }
const float       *orc_icp_means (const orc_icp *h) { return h->means; }
const float       *orc_icp_W (const orc_icp *h) { return h->W; }
double             orc_icp_sum_w (const orc_icp *h) { return h->sum_w; }
const orc_dist_id *orc_icp_nn_id (const orc_icp *h) { return h->nn_id; }
const uint32_t    *orc_icp_rid (const orc_icp *h) { return h->rid; }
const float       *orc_icp_reps (const orc_icp *h) { return h->R; }
const uint32_t    *orc_icp_rbc_N (const orc_icp *h) { return h->N; }
const uint32_t    *orc_icp_rbc_O (const orc_icp *h) { return h->O; }
const uint32_t    *orc_icp_rbc_perm (const orc_icp *h) { return h->perm; }
const uint32_t    *orc_icp_rbc_owner (const orc_icp *h) { return h->owner; }
uint32_t           orc_icp_k (const orc_icp *h) { return h->k; }
int                orc_icp_last_power_iters (const orc_icp *h) { return h->pm_iters; }
