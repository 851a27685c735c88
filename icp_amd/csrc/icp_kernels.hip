// icp_kernels.hip — hand-written HIP kernels (gfx950) of the photogeometric ICP iteration.
//
// Replaces kernels/icp_kernels.cl, kernels/reduce_kernels.cl, kernels/scan_kernels.cl and the
// un-vendored kernels/RBC/*.cl of the reference.  Per-iteration launch set (reference: >= 18 launches
// plus a blocking 32-byte read and a 32-byte write, src/ICP/algorithms.cpp:4670-4698):
//
//   k_search    transform (a3) + nearest representative + list scan (a4) + weights and their
//               128-element tree partials (a5, first level)                         grid (m/128, B)
//   k_means     sum of weights (a5, second level, in the prologue) + weighted block means (a6)
//   k_sij       global means (a6 icpGMean, prologue) + deviations (a7) + S products and their
//               512-column tree partials (a8)
//   k_finalize  S final tree (a8) + power method / SVD (a9, a12) + composition (a10) + check (a11)
//
// Every reduction follows the canonical tree of DESIGN.md §3, so results are bit-identical to
// oracle/icp_oracle.c.  blockIdx.y is the registration index of a batch.
#include "icp_kernels.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ float sum4 (float4 v) { return ((v.x + v.y) + v.z) + v.w; }

// Sum of weights from the 128-element partials: reduce_sum_fd (kernels/icp_kernels.cl:295-329),
// chunks of 512 partials summed in index order (oracle orc_weights).  One wave; every lane returns it.
static __device__ double finalize_sum_w (const float *wpart, uint32_t nwp, uint32_t lane)
{
    if (nwp == 1) return (double) wpart[0];
    double total = 0.0;
    for (uint32_t c0 = 0; c0 < nwp; c0 += 512) {
        uint32_t i0 = c0 + 4 * lane, i1 = c0 + 4 * (lane + 64);
        double d0 = 0.0, d1 = 0.0;
        if (i0 < nwp) {
            float4 v = *reinterpret_cast<const float4 *> (wpart + i0);
            d0 = (((double) v.x + (double) v.y) + (double) v.z) + (double) v.w;
        }
        if (i1 < nwp) {
            float4 v = *reinterpret_cast<const float4 *> (wpart + i1);
            d1 = (((double) v.x + (double) v.y) + (double) v.z) + (double) v.w;
        }
        double cs = wave_tree_d (d0 + d1);
        total = (c0 == 0) ? cs : total + cs;
    }
    return total;
}

// icpGMean over <= 128 block means of one set (kernels/icp_kernels.cl:530-566). One wave.
static __device__ float4 gmean_128 (const float4 *blk, uint32_t nblk, uint32_t lane)
{
    float4 a = make_float4 (0.f, 0.f, 0.f, 0.f), b = a;
    if (lane < nblk) a = blk[lane];
    if (lane + 64 < nblk) b = blk[lane + 64];
    float4 r;
    r.x = wave_tree_f (a.x + b.x);
    r.y = wave_tree_f (a.y + b.y);
    r.z = wave_tree_f (a.z + b.z);
    r.w = 0.f;
    return r;
}

// ------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------
__global__ void k_reset_state (icp_params p, int reset_T)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.batch) return;
    icp_reg_state *st = p.st + b;
    if (reset_T) {                                   // identity T0 — src/ICP/algorithms.cpp:4486-4493
        const float T0[8] = { 0, 0, 0, 1, 0, 0, 0, 1 };
        const float I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        for (int i = 0; i < 8; ++i) { st->T[i] = T0[i]; st->Tk[i] = T0[i]; }
        for (int i = 0; i < 9; ++i) { st->R[i] = I3[i]; st->Rk[i] = I3[i]; }
        for (int i = 0; i < 11; ++i) st->S[i] = 0.f;
        for (int i = 0; i < 8; ++i) st->means[i] = 0.f;
        st->sum_w = 0.0;
    }
    st->k = 0; st->done = 0; st->pm_iters = 0;       // ICP::buildRBC — :4796
}

// write (D_IO_T): T is replaced and the cumulative rotation re-derived from it
__global__ void k_set_T (icp_reg_state *st, const float *T8)
{
    if (threadIdx.x != 0) return;
    float T[8]; for (int i = 0; i < 8; ++i) T[i] = T8[i];
    for (int i = 0; i < 8; ++i) st->T[i] = T[i];
    float R[9]; icp_quat_to_rot (T, R);
    for (int i = 0; i < 9; ++i) st->R[i] = R[i];
}

// ------------------------------------------------------------------------------------------
// a14 getLMs — kernels/icp_kernels.cl:63-76: landmark (gX, gY) = pixel (col 65+4gX, row 49+3gY)
// ------------------------------------------------------------------------------------------
__global__ void k_get_lms (const float4 *cloud, float4 *lms)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;      // float4 index into the landmarks
    if (t >= 16384u * 2u) return;
    uint32_t lm = t >> 1, half = t & 1u;
    uint32_t gX = lm & 127u, gY = lm >> 7;
    uint32_t row = 48u + gY * 3u + 1u, col = 64u + 4u * gX + 1u;
    lms[t] = cloud[((size_t) row * 640u + col) * 2u + half];
}

// icpTransform_Quaternion on a whole cloud — kernels/icp_kernels.cl:772-802
__global__ void k_transform_cloud (const float4 *in, float4 *out, const icp_reg_state *st, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float T[8];
    for (int k = 0; k < 8; ++k) T[k] = st->T[k];
    float4 g = in[2 * (size_t) i], c = in[2 * (size_t) i + 1];
    float x, y, z;
    icp_transform_point (T, g.x, g.y, g.z, x, y, z);
    out[2 * (size_t) i] = make_float4 (x, y, z, g.w);
    out[2 * (size_t) i + 1] = c;
}

// ------------------------------------------------------------------------------------------
// buildRBC
// ------------------------------------------------------------------------------------------

// a1 getReps — kernels/icp_kernels.cl:97-114 with the 128 replaced by the landmark grid side
__global__ void k_get_reps (icp_params p)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (r >= p.nr) return;
    uint32_t gX = r % p.nrx, gY = r / p.nrx;
    uint32_t stepX = p.side / p.nrx, stepY = p.side / p.nry;
    uint32_t xi = (stepX == 1) ? gX : gX * stepX + (stepX >> 1) - 1;
    uint32_t yi = (stepY == 1) ? gY : gY * stepY + (stepY >> 1) - 1;
    uint32_t src = yi * p.side + xi;
    const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
    float4 *R4 = reinterpret_cast<float4 *> (p.R + (size_t) b * p.nr * 8);
    R4[2 * r] = F4[2 * (size_t) src];
    R4[2 * r + 1] = F4[2 * (size_t) src + 1];
    p.rep_src[(size_t) b * p.nr + r] = src;
}

// nearest representative over [r0, r1): strict '<' in ascending r keeps the lowest index on ties
static __device__ __forceinline__ void nearest_rep_range (const float4 *__restrict__ R4, uint32_t r0, uint32_t r1,
                                                          float qx, float qy, float qz, float qr, float qg, float qb,
                                                          float a, float &best, uint32_t &bid)
{
    for (uint32_t r = r0; r < r1; ++r) {
        float4 g = R4[2 * r], c = R4[2 * r + 1];
        float d = icp_metric8 (qx, qy, qz, qr, qg, qb, g.x, g.y, g.z, c.x, c.y, c.z, a);
        if (d < best) { best = d; bid = r; }
    }
}

// RBC construct, step 1: owner(x) = argmin_r d(x, R[r])
__global__ __launch_bounds__ (256) void k_owner (icp_params p)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= p.m) return;
    const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
    const float4 *R4 = reinterpret_cast<const float4 *> (p.R + (size_t) b * p.nr * 8);
    float4 g = F4[2 * (size_t) i], c = F4[2 * (size_t) i + 1];
    float best = __builtin_inff (); uint32_t bid = 0;
    nearest_rep_range (R4, 0, p.nr, g.x, g.y, g.z, c.x, c.y, c.z, p.a, best, bid);
    p.owner[(size_t) b * p.m + i] = bid;
}

// step 2: per-chunk histograms of the owners (integer LDS atomics: deterministic)
__global__ __launch_bounds__ (256) void k_chunk_hist (icp_params p)
{
    extern __shared__ __attribute__ ((aligned (16))) uint32_t s_hist[];
    uint32_t chunk = blockIdx.x, b = blockIdx.y;
    for (uint32_t r = threadIdx.x; r < p.nr; r += blockDim.x) s_hist[r] = 0;
    __syncthreads ();
    const uint32_t *owner = p.owner + (size_t) b * p.m;
    for (uint32_t k = threadIdx.x; k < ICP_CHUNK; k += blockDim.x) {
        uint32_t i = chunk * ICP_CHUNK + k;
        if (i < p.m) atomicAdd (&s_hist[owner[i]], 1u);
    }
    __syncthreads ();
    uint32_t *H = p.chunk_hist + ((size_t) b * p.nchunk + chunk) * p.nr;
    for (uint32_t r = threadIdx.x; r < p.nr; r += blockDim.x) H[r] = s_hist[r];
}

// step 3: N[r] = sum over chunks; chunk_hist[chunk][r] becomes the rank base of that chunk in list r
__global__ void k_count (icp_params p)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (r >= p.nr) return;
    uint32_t run = 0;
    for (uint32_t ch = 0; ch < p.nchunk; ++ch) {
        uint32_t *h = p.chunk_hist + ((size_t) b * p.nchunk + ch) * p.nr + r;
        uint32_t v = *h; *h = run; run += v;
    }
    p.N[(size_t) b * p.nr + r] = run;
}

// step 4: O = exclusive scan of N (exclusiveScan_i semantics, kernels/scan_kernels.cl:188). One block.
__global__ __launch_bounds__ (1024) void k_offsets (icp_params p)
{
    __shared__ uint32_t s_sum[1024];
    uint32_t b = blockIdx.y, t = threadIdx.x;
    const uint32_t *N = p.N + (size_t) b * p.nr;
    uint32_t *O = p.O + (size_t) b * p.nr;
    uint32_t per = (p.nr + 1023u) / 1024u, lo = t * per, hi = min (lo + per, p.nr);
    uint32_t s = 0;
    for (uint32_t r = lo; r < hi; ++r) s += N[r];
    s_sum[t] = s;
    __syncthreads ();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 1024; ++k) { uint32_t v = s_sum[k]; s_sum[k] = run; run += v; }
    }
    __syncthreads ();
    uint32_t run = s_sum[t];
    for (uint32_t r = lo; r < hi; ++r) { O[r] = run; run += N[r]; }
}

// step 5: stable placement: position = O[owner] + #{j < i : owner[j] == owner[i]}; perm and X_P.
// One block per chunk; its 16 waves take turns in index order so that ranks follow the index.
__global__ __launch_bounds__ (1024) void k_place (icp_params p)
{
    extern __shared__ __attribute__ ((aligned (16))) uint32_t s_base[];
    uint32_t chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t *O = p.O + (size_t) b * p.nr;
    const uint32_t *H = p.chunk_hist + ((size_t) b * p.nchunk + chunk) * p.nr;
    for (uint32_t r = t; r < p.nr; r += blockDim.x) s_base[r] = O[r] + H[r];
    uint32_t i = chunk * ICP_CHUNK + t;
    bool valid = i < p.m;
    uint32_t own = valid ? p.owner[(size_t) b * p.m + i] : 0xFFFFFFFFu;
    uint32_t pos = 0;
    for (uint32_t w = 0; w < ICP_CHUNK / 64u; ++w) {
        __syncthreads ();
        if (wave == w) {
            bool active = valid;
            unsigned long long todo = __ballot (active);
            while (todo) {
                int leader = __ffsll ((long long) todo) - 1;
                uint32_t oo = __shfl (own, leader);
                unsigned long long same = __ballot (active && own == oo);
                if (active && own == oo) {
                    unsigned long long below = same & ((1ull << lane) - 1ull);
                    pos = s_base[oo] + (uint32_t) __popcll (below);
                    active = false;
                }
                __builtin_amdgcn_wave_barrier ();
                if ((int) lane == leader) s_base[oo] += (uint32_t) __popcll (same);
                todo &= ~same;
            }
        }
    }
    if (valid) {
        const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
        float4 *X4 = reinterpret_cast<float4 *> (p.XP + (size_t) b * p.m * 8);
        p.perm[(size_t) b * p.m + pos] = i;
        X4[2 * (size_t) pos] = F4[2 * (size_t) i];
        X4[2 * (size_t) pos + 1] = F4[2 * (size_t) i + 1];
    }
}

// ------------------------------------------------------------------------------------------
// K1  search: transform + RBC one-shot search + weights (first-level tree)
//     block = 128 queries x SPLIT slices; wave (2*slice + grp) serves queries grp*64 + lane.
// ------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__ (128 * SPLIT) void k_search (icp_params p)
{
    const uint32_t b = blockIdx.y;
    icp_reg_state *st = p.st + b;
    if (st->done) return;

    __shared__ float s_best[SPLIT][128];
    __shared__ uint32_t s_idx[SPLIT][128];
    __shared__ float s_w[128];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane (tid >> 6);
    const uint32_t grp = wave & 1u, slice = wave >> 1;
    const uint32_t ql = grp * 64u + lane;
    const uint32_t i = blockIdx.x * 128u + ql;
    const bool valid = i < p.m;

    float T[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = st->T[k];

    const float4 *M4 = reinterpret_cast<const float4 *> (p.M + (size_t) b * p.m * 8);
    const float4 *R4 = reinterpret_cast<const float4 *> (p.R + (size_t) b * p.nr * 8);
    const float4 *X4 = reinterpret_cast<const float4 *> (p.XP + (size_t) b * p.m * 8);

    float4 mg = make_float4 (0.f, 0.f, 0.f, 1.f), mc = mg;
    if (valid) { mg = M4[2 * (size_t) i]; mc = M4[2 * (size_t) i + 1]; }
    float qx, qy, qz;
    icp_transform_point (T, mg.x, mg.y, mg.z, qx, qy, qz);
    const float qr = mc.x, qg = mc.y, qb = mc.z;

    // ---- stage 1: nearest representative (this wave's slice of the representatives) ----
    const uint32_t per = (p.nr + SPLIT - 1) / SPLIT;
    const uint32_t r0 = min (slice * per, p.nr), r1 = min (r0 + per, p.nr);
    float best = __builtin_inff (); uint32_t bid = r0;
    nearest_rep_range (R4, r0, r1, qx, qy, qz, qr, qg, qb, p.a, best, bid);
    s_best[slice][ql] = best; s_idx[slice][ql] = bid;
    __syncthreads ();
    float dr = s_best[0][ql]; uint32_t rstar = s_idx[0][ql];
#pragma unroll
    for (int s = 1; s < SPLIT; ++s) {
        float d = s_best[s][ql]; uint32_t id = s_idx[s][ql];
        if (d < dr) { dr = d; rstar = id; }
    }
    __syncthreads ();

    // ---- stage 2: exhaustive scan of that representative's list, positions interleaved over slices ----
    const uint32_t o = p.O[(size_t) b * p.nr + rstar], n = p.N[(size_t) b * p.nr + rstar];
    float best2 = __builtin_inff (); uint32_t bj = 0xFFFFFFFFu;
    if (valid)
        for (uint32_t j = o + slice; j < o + n; j += SPLIT) {
            float4 g = X4[2 * (size_t) j], c = X4[2 * (size_t) j + 1];
            float d = icp_metric8 (qx, qy, qz, qr, qg, qb, g.x, g.y, g.z, c.x, c.y, c.z, p.a);
            if (d < best2) { best2 = d; bj = j; }
        }
    s_best[slice][ql] = best2; s_idx[slice][ql] = bj;
    __syncthreads ();

    if (slice == 0) {
        float d = s_best[0][ql]; uint32_t j = s_idx[0][ql];
#pragma unroll
        for (int s = 1; s < SPLIT; ++s) {
            float ds = s_best[s][ql]; uint32_t js = s_idx[s][ql];
            if (ds < d || (ds == d && js < j)) { d = ds; j = js; }     // ties -> lowest list position
        }
        float w = 0.f;
        if (valid) {
            uint32_t id; float4 nn;
            if (n == 0) {            // empty list: fall back to the representative itself
                d = dr; id = p.rep_src[(size_t) b * p.nr + rstar]; nn = R4[2 * rstar];
            } else {
                id = p.perm[(size_t) b * p.m + j]; nn = X4[2 * (size_t) j];
            }
            w = p.weighted ? 100.f / (100.f + d) : 1.f;                // icp_kernels.cl:232
            icp_dist_id di; di.dist = d; di.id = id;
            p.nn_id[(size_t) b * p.m + i] = di;
            p.PF[(size_t) b * p.m + i] = make_float4 (nn.x, nn.y, nn.z, w);
            p.PM[(size_t) b * p.m + i] = make_float4 (qx, qy, qz, d);
            p.rid[(size_t) b * p.m + i] = rstar;
        }
        s_w[ql] = w;
    }
    __syncthreads ();
    if (wave == 0 && p.weighted) {   // icpComputeReduceWeights_WG tree — icp_kernels.cl:244-253
        float v = wave_tree_f (s_w[lane] + s_w[lane + 64]);
        if (lane == 0) p.wpart[(size_t) b * p.nwp + blockIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// second-level kernels for sizes beyond one work-group of partials (m > 65536 / m > 16384)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64) void k_sum_w (icp_params p)
{
    uint32_t b = blockIdx.y, lane = threadIdx.x;
    icp_reg_state *st = p.st + b;
    if (st->done) return;
    double sw = finalize_sum_w (p.wpart + (size_t) b * p.nwp, p.nwp, lane);
    if (lane == 0) st->sum_w = sw;
}

// multi-level icpGMean: groups of 128 block means per pass until one vector per set remains
__global__ __launch_bounds__ (1024) void k_gmean (icp_params p)
{
    uint32_t b = blockIdx.y, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    icp_reg_state *st = p.st + b;
    if (st->done) return;
    uint32_t nscr = (p.nwg + 127u) / 128u;
    for (uint32_t set = 0; set < 2; ++set) {
        const float4 *cur = p.mpart + ((size_t) b * 2 + set) * p.nwg;
        float4 *scr = p.mscr + ((size_t) b * 2 + set) * nscr;
        uint32_t n = p.nwg;
        while (n > 1) {
            uint32_t ng = (n + 127u) / 128u;
            __syncthreads ();
            float4 keep[8]; uint32_t cnt = 0;                 // results first (scr may alias cur)
            for (uint32_t g = wave; g < ng && cnt < 8; g += 16) keep[cnt++] = gmean_128 (cur + (size_t) g * 128, min (128u, n - g * 128u), lane);
            __syncthreads ();
            cnt = 0;
            for (uint32_t g = wave; g < ng && cnt < 8; g += 16) { if (lane == 0) scr[g] = keep[cnt]; ++cnt; }
            __threadfence_block ();
            __syncthreads ();
            cur = scr; n = ng;
        }
        if (threadIdx.x == 0) {
            float4 r = cur[0];
            st->means[set * 4 + 0] = r.x; st->means[set * 4 + 1] = r.y; st->means[set * 4 + 2] = r.z; st->means[set * 4 + 3] = 0.f;
        }
        __syncthreads ();
    }
}

// ------------------------------------------------------------------------------------------
// K3  means: icpMean(_Weighted) — kernels/icp_kernels.cl:371-411, 455-495.  One wave per 128 pairs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64) void k_means (icp_params p)
{
    const uint32_t b = blockIdx.y, lane = threadIdx.x, g = blockIdx.x;
    icp_reg_state *st = p.st + b;
    if (st->done) return;

    double sum_w = 1.0;
    if (p.weighted) {
        if (p.nwp <= 512) {
            sum_w = finalize_sum_w (p.wpart + (size_t) b * p.nwp, p.nwp, lane);
            if (g == 0 && lane == 0) st->sum_w = sum_w;
        } else sum_w = st->sum_w;                    // written by k_sum_w
    }
    const float4 *PF = p.PF + (size_t) b * p.m, *PM = p.PM + (size_t) b * p.m;
    const uint32_t e0 = g * 128u + lane, e1 = e0 + 64u;
    float f[2][3], q[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t e = h ? e1 : e0;
        bool ok = (e & ~1u) < p.m;                   // the pair's flag guards both points (icp_kernels.cl:390-392)
        float4 pf = make_float4 (0.f, 0.f, 0.f, 0.f), pm = pf;
        if (ok) { pf = PF[e]; pm = PM[e]; }
        if (p.weighted) {
            float k = (float) ((double) pf.w / sum_w);                   // icp_kernels.cl:475
            f[h][0] = ok ? k * pf.x : 0.f; f[h][1] = ok ? k * pf.y : 0.f; f[h][2] = ok ? k * pf.z : 0.f;
            q[h][0] = ok ? k * pm.x : 0.f; q[h][1] = ok ? k * pm.y : 0.f; q[h][2] = ok ? k * pm.z : 0.f;
        } else {
            float nf = (float) p.m;                                      // icp_kernels.cl:391
            f[h][0] = ok ? pf.x / nf : 0.f; f[h][1] = ok ? pf.y / nf : 0.f; f[h][2] = ok ? pf.z / nf : 0.f;
            q[h][0] = ok ? pm.x / nf : 0.f; q[h][1] = ok ? pm.y / nf : 0.f; q[h][2] = ok ? pm.z / nf : 0.f;
        }
    }
    float4 mf, mm;
    mf.x = wave_tree_f (f[0][0] + f[1][0]); mf.y = wave_tree_f (f[0][1] + f[1][1]); mf.z = wave_tree_f (f[0][2] + f[1][2]); mf.w = 0.f;
    mm.x = wave_tree_f (q[0][0] + q[1][0]); mm.y = wave_tree_f (q[0][1] + q[1][1]); mm.z = wave_tree_f (q[0][2] + q[1][2]); mm.w = 0.f;
    if (lane == 0) {
        p.mpart[((size_t) b * 2 + 0) * p.nwg + g] = mf;
        p.mpart[((size_t) b * 2 + 1) * p.nwg + g] = mm;
    }
}

// ------------------------------------------------------------------------------------------
// K4  S matrix: icpSubtractMean + icpSijProducts(_Weighted) + reduce_sum_f first level
//     kernels/icp_kernels.cl:588-602, 633-671, 703-743; kernels/reduce_kernels.cl:230-264.
//     Block = 512 columns = one reduce_sum_f work-group; thread = one column (4 strided points).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (512) void k_sij (icp_params p)
{
    const uint32_t b = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane (tid >> 6);
    icp_reg_state *st = p.st + b;
    if (st->done) return;

    __shared__ float4 s_mean[2];
    __shared__ float s_pos[11][128];

    if (p.nwg <= 128) {                              // icpGMean in the prologue (one wave per set)
        if (wave < 2) {
            float4 r = gmean_128 (p.mpart + ((size_t) b * 2 + wave) * p.nwg, p.nwg, lane);
            if (lane == 0) {
                s_mean[wave] = r;
                if (blockIdx.x == 0) { st->means[wave * 4] = r.x; st->means[wave * 4 + 1] = r.y; st->means[wave * 4 + 2] = r.z; st->means[wave * 4 + 3] = 0.f; }
            }
        }
    } else if (tid < 2) {                            // written by k_gmean
        s_mean[tid] = make_float4 (st->means[tid * 4], st->means[tid * 4 + 1], st->means[tid * 4 + 2], 0.f);
    }
    __syncthreads ();
    const float4 mf = s_mean[0], mm = s_mean[1];

    const float4 *PF = p.PF + (size_t) b * p.m, *PM = p.PM + (size_t) b * p.m;
    const uint32_t col = blockIdx.x * 512u + tid;
    const float c = p.c;
    float A[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) A[k] = 0.f;
    if (col < p.G)
        for (uint32_t pi = col; pi < p.m; pi += p.G) {                   // icp_kernels.cl:718
            float4 pf = PF[pi], pm = PM[pi];
            float Mp[3] = { c * (pm.x - mm.x), c * (pm.y - mm.y), c * (pm.z - mm.z) };
            float Fp[3] = { c * (pf.x - mf.x), c * (pf.y - mf.y), c * (pf.z - mf.z) };
            float ff = (Fp[0] * Fp[0] + Fp[1] * Fp[1]) + Fp[2] * Fp[2];
            float m2 = (Mp[0] * Mp[0] + Mp[1] * Mp[1]) + Mp[2] * Mp[2];
            if (p.weighted) {
                float w = pf.w;
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
    // position p = ((c[4p] + c[4p+1]) + c[4p+2]) + c[4p+3]  — reduce_kernels.cl:245-251
    const uint32_t qb = lane & ~3u;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float v = A[k];
        float s = __shfl (v, qb) + __shfl (v, qb + 1);
        s = s + __shfl (v, qb + 2);
        s = s + __shfl (v, qb + 3);
        if ((tid & 3u) == 0) s_pos[k][tid >> 2] = s;
    }
    __syncthreads ();
    for (uint32_t row = wave; row < 11; row += 8) {
        float v = wave_tree_f (s_pos[row][lane] + s_pos[row][lane + 64]);
        if (lane == 0) p.spart[((size_t) b * 11 + row) * p.nsp + blockIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K5  finalize: reduce_sum_f second level, rotation, composition, convergence. One wave per registration.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64) void k_finalize (icp_params p)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    icp_reg_state *st = p.st + b;
    if (st->done) return;

    float S[11];
    for (int row = 0; row < 11; ++row) {
        const float *cur = p.spart + ((size_t) b * 11 + row) * p.nsp;
        uint32_t n = p.nsp;
        float *scr = p.sscr + ((size_t) b * 11 + row) * (((p.nsp + 511u) / 512u + 3u) & ~3u);
        while (n > 1) {                              // same shape as the first level: 512 columns per group
            uint32_t wg = (n + 511u) / 512u, wgp = wg;
            if (wgp != 1 && (wgp & 3u)) wgp += 4u - (wgp & 3u);
            float res = 0.f;
            for (uint32_t g = 0; g < wgp; ++g) {
                uint32_t c0 = g * 512u + 4u * lane, c1 = c0 + 256u;
                float a0 = (c0 < n) ? sum4 (*reinterpret_cast<const float4 *> (cur + c0)) : 0.f;
                float a1 = (c1 < n) ? sum4 (*reinterpret_cast<const float4 *> (cur + c1)) : 0.f;
                float v = wave_tree_f (a0 + a1);
                if (wgp == 1) res = v; else if (lane == 0) scr[g] = v;
            }
            if (wgp == 1) { S[row] = res; n = 1; break; }
            __threadfence_block ();
            cur = scr; n = wgp;
        }
        if (p.nsp == 1) S[row] = cur[0];
    }
    float means[8];
    for (int k = 0; k < 8; ++k) means[k] = st->means[k];

    float Tk[8], Rk[9];
    int iters = 0;
    if (p.rot == 1) iters = icp_power_method (S, means, Tk, p.power_mode);
    else icp_svd_rotation (S, means, Rk, Tk);

    if (lane == 0) {
        icp_compose (st, Tk, Rk, p.rot != 1);
        for (int k = 0; k < 11; ++k) st->S[k] = S[k];
        st->pm_iters = (uint32_t) iters;
        st->k = st->k + 1;
        if (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) st->done = 1;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void icp_launch_reset_state (const icp_params &p, hipStream_t s, int reset_T)
{
    hipLaunchKernelGGL (k_reset_state, dim3 ((p.batch + 63) / 64), dim3 (64), 0, s, p, reset_T);
}

void icp_launch_set_T (const icp_params &p, uint32_t b, const float *dT8, hipStream_t s)
{
    hipLaunchKernelGGL (k_set_T, dim3 (1), dim3 (64), 0, s, p.st + b, dT8);
}

void icp_launch_get_lms (const float *cloud, float *lms, hipStream_t s)
{
    hipLaunchKernelGGL (k_get_lms, dim3 (16384 * 2 / 256), dim3 (256), 0, s,
                        reinterpret_cast<const float4 *> (cloud), reinterpret_cast<float4 *> (lms));
}

void icp_launch_transform_cloud (const float *in, float *out, const icp_reg_state *st, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL (k_transform_cloud, dim3 ((n + 255) / 256), dim3 (256), 0, s,
                        reinterpret_cast<const float4 *> (in), reinterpret_cast<float4 *> (out), st, n);
}

void icp_launch_build_rbc (const icp_params &p, hipStream_t s)
{
    hipLaunchKernelGGL (k_get_reps, dim3 ((p.nr + 63) / 64, p.batch), dim3 (64), 0, s, p);
    hipLaunchKernelGGL (k_owner, dim3 ((p.m + 255) / 256, p.batch), dim3 (256), 0, s, p);
    hipLaunchKernelGGL (k_chunk_hist, dim3 (p.nchunk, p.batch), dim3 (256), p.nr * sizeof (uint32_t), s, p);
    hipLaunchKernelGGL (k_count, dim3 ((p.nr + 63) / 64, p.batch), dim3 (64), 0, s, p);
    hipLaunchKernelGGL (k_offsets, dim3 (1, p.batch), dim3 (1024), 0, s, p);
    hipLaunchKernelGGL (k_place, dim3 (p.nchunk, p.batch), dim3 (1024), p.nr * sizeof (uint32_t), s, p);
}

void icp_launch_search (const icp_params &p, hipStream_t s)
{
    dim3 grid ((p.m + 127) / 128, p.batch);
    hipLaunchKernelGGL (k_search<4>, grid, dim3 (512), 0, s, p);
}

void icp_launch_means (const icp_params &p, hipStream_t s)
{
    if (p.weighted && p.nwp > 512) hipLaunchKernelGGL (k_sum_w, dim3 (1, p.batch), dim3 (64), 0, s, p);
    hipLaunchKernelGGL (k_means, dim3 (p.nwg, p.batch), dim3 (64), 0, s, p);
}

void icp_launch_sij (const icp_params &p, hipStream_t s)
{
    if (p.nwg > 128) hipLaunchKernelGGL (k_gmean, dim3 (1, p.batch), dim3 (1024), 0, s, p);
    hipLaunchKernelGGL (k_sij, dim3 ((p.G + 511) / 512, p.batch), dim3 (512), 0, s, p);
}

void icp_launch_finalize (const icp_params &p, hipStream_t s)
{
    hipLaunchKernelGGL (k_finalize, dim3 (p.batch), dim3 (64), 0, s, p);
}

void icp_launch_iteration (const icp_params &p, hipStream_t s)
{
    icp_launch_search (p, s);
    icp_launch_means (p, s);
    icp_launch_sij (p, s);
    icp_launch_finalize (p, s);
}
