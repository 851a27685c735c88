// icp_build.hip — hand-written HIP kernels (gfx950) around the iteration: the registration state, landmark extraction (getLMs),
// the cloud transforms, the RBC construction behind ICPStep::buildRBC (src/ICP/algorithms.cpp:4655-4660: getReps + the
// un-vendored RBCConstruct) and ICPPowerMethod as a kernel of its own.  The owner search of the construction (step 1) is the
// search kernel's stage 1 and lives with it in icp_kernels.hip (icp_launch_owner_search).  blockIdx.y is the registration index of
// a batch.  Integer work throughout (histograms, scans, ranks): bit-identical to oracle/icp_oracle.c by construction.
#include "icp_kernels.h"

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
__global__ void k_set_T (icp_reg_state *st, const float *T8, int reset_k)
{
    if (threadIdx.x != 0) return;
    if (reset_k) { st->k = 0; st->done = 0; st->pm_iters = 0; }      // (tracking: what buildRBC would have reset, see icp_params::no_state_reset)
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

// getLMs from the BAND of a frame: the part of a 640 x 480 cloud the kernel above reads, packed — row j of the band = cloud row
// 49 + 3 j, pixels 65 .. 573 (ICP_BAND_* in icp_kernels.h: 128 rows x 509 pixels = 2.08 MB of the frame's 9.83 MB).  Tracking
// uploads only that (icp_track_submit); landmark (gX, gY) = band pixel (4 gX, gY): the same points as k_get_lms.
__global__ void k_get_lms_band (const float4 *band, float4 *lms)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 16384u * 2u) return;
    uint32_t lm = t >> 1, half = t & 1u;
    uint32_t gX = lm & 127u, gY = lm >> 7;
    lms[t] = band[((size_t) gY * ICP_BAND_COLS + 4u * gX) * 2u + half];
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

// ICPTransform<QUATERNION | MATRIX> with an explicit transformation (the cloud kernels of the reference that take
// their parameters from a buffer of their own): KIND 0 icpTransform_Quaternion (kernels/icp_kernels.cl:772-802),
// 1 icpTransform_Quaternion_2 (:842-879: the two quaternion products as 4x4 matrix-vector products), 2
// icpTransform_Matrix (:904-933: rows 0..2 of a row-major 4x4 applied to the homogeneous point as stored).
// dot (a, b) = (((0 + a0 b0) + a1 b1) + a2 b2) + a3 b3, the CPU twins' order (oracle orc_transform_q2 / orc_transform_m).
struct icp_T16 { float v[16]; };
static __device__ __forceinline__ float dot4_seq (float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3)
{
    float s = 0.f;
    s = s + a0 * b0; s = s + a1 * b1; s = s + a2 * b2; s = s + a3 * b3;
    return s;
}
template <int KIND>
__global__ __launch_bounds__ (256) void k_transform_cloud_ex (const float4 *in, float4 *out, icp_T16 T, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 g = in[2 * (size_t) i], c = in[2 * (size_t) i + 1];
    float x, y, z;
    if constexpr (KIND == 0) icp_transform_point (T.v, g.x, g.y, g.z, x, y, z);
    else if constexpr (KIND == 1) {
        const float qx = T.v[0], qy = T.v[1], qz = T.v[2], qw = T.v[3];
        const float p0 = dot4_seq ( qw, -qz,  qy, qx, g.x, g.y, g.z, 0.f);
        const float p1 = dot4_seq ( qz,  qw, -qx, qy, g.x, g.y, g.z, 0.f);
        const float p2 = dot4_seq (-qy,  qx,  qw, qz, g.x, g.y, g.z, 0.f);
        const float p3 = dot4_seq (-qx, -qy, -qz, qw, g.x, g.y, g.z, 0.f);
        x = T.v[7] * dot4_seq ( qw, -qz,  qy, -qx, p0, p1, p2, p3) + T.v[4];
        y = T.v[7] * dot4_seq ( qz,  qw, -qx, -qy, p0, p1, p2, p3) + T.v[5];
        z = T.v[7] * dot4_seq (-qy,  qx,  qw, -qz, p0, p1, p2, p3) + T.v[6];
    } else {
        x = dot4_seq (T.v[0], T.v[1], T.v[2],  T.v[3],  g.x, g.y, g.z, g.w);
        y = dot4_seq (T.v[4], T.v[5], T.v[6],  T.v[7],  g.x, g.y, g.z, g.w);
        z = dot4_seq (T.v[8], T.v[9], T.v[10], T.v[11], g.x, g.y, g.z, g.w);
    }
    out[2 * (size_t) i] = make_float4 (x, y, z, g.w);
    out[2 * (size_t) i + 1] = c;
}

// ------------------------------------------------------------------------------------------
// buildRBC
// ------------------------------------------------------------------------------------------

#define ICP_ORIGIN_SOLO 16u       // k_reps_and_boxes: up to this many blocks of 64 representatives, one wave lists those at the origin by itself

// the end of the list of the representatives at the origin (k_reps_and_boxes): colour boxes of its chunks, its length where the search reads it
#define ICP_OL_SORT_MAX 2048u      // k_reps_and_boxes: lists of the representatives at the origin up to this length are ordered by colour (LDS: 24 bytes per entry)

static __device__ __forceinline__ void origin_list_close (const icp_params &p, uint32_t b, uint32_t lane, float4 *OL, uint32_t run, uint32_t nw, float4 *s_ent, uint32_t *s_key, uint32_t sort_cap)
{
    const float inf = __builtin_inff ();
    // colour boxes of the chunks of 8 consecutive entries (ks_origin_list tests a chunk before it scans it — the owner search of THIS
    // construction already does, so the boxes cannot wait for a later launch): the wave reads its own entries back behind a fence
    __threadfence ();
    // Lists the search prunes by chunk boxes (more than 128 entries): ordered by a Morton key of the colour instead of by index, so that the
    // 8 entries of a chunk are neighbours in colour whatever the invalid points' pattern in the frame — in index order 10 of 49 / 157
    // chunks passed a query's test at |F| = 2^20 with 10 % scattered / 30 % contiguous invalid points, ordered by colour 4
    // (tools/diag/origin_list_sim.py).  The scan's tie rule is explicit (ks_origin_list), so the list's order is free.  One wave: entries and
    // keys (colour key << 16 | position) in LDS, a bitonic sort of the keys, the entries written back in their order.
    if (run > ICP_OL_BOXED_MIN && run <= sort_cap) {
        float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
        for (uint32_t e = lane; e < run; e += 64u) {
            const float4 v = OL[1u + e];
            s_ent[e] = v;
            lo[0] = fminf (lo[0], v.x); lo[1] = fminf (lo[1], v.y); lo[2] = fminf (lo[2], v.z);
            hi[0] = fmaxf (hi[0], v.x); hi[1] = fmaxf (hi[1], v.y); hi[2] = fmaxf (hi[2], v.z);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf (lo[k], __shfl_xor (lo[k], d)); hi[k] = fmaxf (hi[k], __shfl_xor (hi[k], d)); }
        float sc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) sc[k] = hi[k] > lo[k] ? 32.f / (hi[k] - lo[k]) : 0.f;
        uint32_t n2 = 256u;
        while (n2 < run) n2 <<= 1;
        __syncthreads ();                                             // (the block is this one wave: the entries are in LDS)
        auto spread5 = [] (uint32_t x) -> uint32_t { return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4) | ((x & 8u) << 6) | ((x & 16u) << 8); };
        for (uint32_t e = lane; e < n2; e += 64u) {
            uint32_t key = 0xFFFFFFFFu;
            if (e < run) {
                const float4 v = s_ent[e];
                // (a NaN colour: fmaxf / fminf leave 0 — any key will do, the order only has to be the same every time)
                const uint32_t q0 = (uint32_t) fminf (fmaxf ((v.x - lo[0]) * sc[0], 0.f), 31.f), q1 = (uint32_t) fminf (fmaxf ((v.y - lo[1]) * sc[1], 0.f), 31.f),
                               q2 = (uint32_t) fminf (fmaxf ((v.z - lo[2]) * sc[2], 0.f), 31.f);
                key = ((spread5 (q0) | (spread5 (q1) << 1) | (spread5 (q2) << 2)) << 16) | e;
            }
            s_key[e] = key;
        }
        for (uint32_t k = 2u; k <= n2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                __syncthreads ();
                // (two or four compare-exchanges per lane in flight — n2 >= 256 —: one at a time every step was a chain of dependent LDS round trips)
                if (n2 >= 512u) {
                    for (uint32_t t0 = lane; t0 < (n2 >> 1); t0 += 256u) {
                        uint32_t i0[4], x[4], y[4];
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) { const uint32_t t = t0 + 64u * u; i0[u] = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)); x[u] = s_key[i0[u]]; y[u] = s_key[i0[u] + j]; }
#pragma unroll
                        for (uint32_t u = 0; u < 4u; ++u) { const bool sw = (x[u] > y[u]) == ((i0[u] & k) == 0u); s_key[i0[u]] = sw ? y[u] : x[u]; s_key[i0[u] + j] = sw ? x[u] : y[u]; }
                    }
                } else {
                    for (uint32_t t0 = lane; t0 < (n2 >> 1); t0 += 128u) {
                        const uint32_t ta = t0, tb = t0 + 64u;
                        const uint32_t a0 = ((ta & ~(j - 1u)) << 1) | (ta & (j - 1u)), a1 = a0 + j, b0 = ((tb & ~(j - 1u)) << 1) | (tb & (j - 1u)), b1 = b0 + j;
                        const uint32_t xa = s_key[a0], ya = s_key[a1], xb = s_key[b0], yb = s_key[b1];
                        const bool sa = (xa > ya) == ((a0 & k) == 0u), sb = (xb > yb) == ((b0 & k) == 0u);
                        s_key[a0] = sa ? ya : xa; s_key[a1] = sa ? xa : ya;
                        s_key[b0] = sb ? yb : xb; s_key[b1] = sb ? xb : yb;
                    }
                }
            }
        __syncthreads ();
        for (uint32_t e = lane; e < run; e += 64u) OL[1u + e] = s_ent[s_key[e] & 0xFFFFu];
        __threadfence ();                                             // (the boxes below read the entries back)
    }
    // Round 6 — representatives at the origin that REPEAT their predecessor's colour: identical points (a frame's invalid points whose
    // colour was zeroed too are ONE point; |F| = 65536 with 30 % of them: 300 of the 1024 representatives), of which only the one with the
    // lowest index can ever be a query's nearest representative (equal coordinates, equal distance bits, ties -> lowest index) — yet every
    // invalid query scanned them all: their chunk boxes are one point, and a bound that EQUALS the best so far does not prune (the list is
    // visited out of index order).  An entry whose colour equals that of the entry in front of it is dropped: in index order and in colour
    // order alike (equal colours = equal sort keys, ordered by position in the index-ordered list) the entry in front has the lower index.
    // The ballots keep every representative at the origin (a dropped one is still a legitimate seed of its kind).  One wave, in place:
    // a round's reads come before its writes, which land at or below the round's own positions.
    {
        uint32_t w = 0u;
        for (uint32_t e0 = 0u; e0 < run; e0 += 64u) {
            const uint32_t e = e0 + lane;
            const float4 v = OL[1u + min (e, run - 1u)];
            const float4 pv = OL[1u + min (max (e, 1u) - 1u, run - 1u)];
            const bool keep = e < run && (e == 0u || !(v.x == pv.x && v.y == pv.y && v.z == pv.z));
            const unsigned long long bal = __ballot (keep);
            if (keep) OL[1u + w + (uint32_t) __builtin_popcountll (bal & ((1ull << lane) - 1ull))] = v;
            w += (uint32_t) __builtin_popcountll (bal);
        }
        if (w != run) { run = w; __threadfence (); }
    }
    {
        float4 *BX = OL + 1u + p.nr;
        const uint32_t n_oc = (run + 7u) >> 3;
        for (uint32_t c = lane; c < n_oc; c += 64u) {
            float4 lo = make_float4 (inf, inf, inf, 0.f), hi = make_float4 (-inf, -inf, -inf, 0.f);
            float4 v[8];                                                  // (the chunk's eight entries in flight; past the list's end: its last entry once more)
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) v[e] = OL[1u + min (8u * c + e, run - 1u)];
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) {
                lo.x = fminf (lo.x, v[e].x); lo.y = fminf (lo.y, v[e].y); lo.z = fminf (lo.z, v[e].z);
                hi.x = fmaxf (hi.x, v[e].x); hi.y = fmaxf (hi.y, v[e].y); hi.z = fmaxf (hi.z, v[e].z);
            }
            BX[2u * c] = lo; BX[2u * c + 1u] = hi;
        }
    }
    if (lane == 0) {
        OL[0] = make_float4 (__uint_as_float (run), 0.f, 0.f, 0.f);      // (.y: the arrival counter, back at zero for the next construction)
        // the number the search reads: a spare lane of the box array (hi.w of the first tile box: nobody else writes that word)
        reinterpret_cast<float *> (p.GB + (size_t) b * 2 * (p.n16 + p.n1k) + 2u * p.n16 + 1u)[3] = __uint_as_float (run);      // tile box 0, hi.w
        reinterpret_cast<float *> (p.GB + (size_t) b * 2 * (p.n16 + p.n1k) + 1u)[3] = __uint_as_float (run);                    // group box 0, hi.w
    }
}

// a1 + the boxes of the stage-1 pruning, ONE launch of 64-thread blocks with three kinds of duty (buildRBC is a chain of
// small dependent launches: every launch saved is ~4 us of its ~40 at |F| = 16384):
//   blocks [0, nbr)            the representatives: R[r] = F[src (r)], rep_src[r] = src (r)                     (getReps)
//   blocks [nbr, nbr + nbg)    geometry bounding boxes of the pruning groups of 16 representatives: a 4 x 4 tile of the
//                              representative grid where the grid allows (p.gtile: the representatives are a regular sample
//                              of the landmark grid, so a tile is compact in space: 1.8 - 2.9 groups per wave survive the
//                              bound instead of 3.3 - 8.8 with 16 x 1 strips), else 16 consecutive representatives
//   blocks [nbr + nbg, ..)     the box of every LDS tile of the dense k_search for multi-tile sets (p.tbox consecutive
//                              representatives: 256, or 1024 for the largest sets), one wave per box
//   (the block of the first kind that finishes last) the representatives at the origin (a frame's invalid points), colour + index,
//                              ascending: they stay out of every box above (one such member would stretch a box from the scene to the
//                              origin) and are scanned as a list of their own by the queries that are near the origin (k_search:
//                              ks_origin_list)
// The boxes read the representatives' points from F at src (r): they do not wait for R.  fminf / fmaxf skip NaN
// coordinates: a representative with a NaN coordinate never wins a '<' anyway; min / max are exact in any order.
__global__ __launch_bounds__ (64) void k_reps_and_boxes (icp_params p, uint32_t nbr, uint32_t nbg, uint32_t sort_cap)
{
    // (origin_list_close: the one block that orders the list of the representatives at the origin — sort_cap entries of 16 + 4 bytes, dynamic:
    // a launch of many small registrations goes without, 40 KB per 64-thread block would leave four blocks to a CU)
    extern __shared__ __attribute__ ((aligned (16))) float4 s_ent[];
    uint32_t *s_key = reinterpret_cast<uint32_t *> (s_ent + sort_cap);
    const uint32_t b = blockIdx.y, lane = threadIdx.x;
    const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
    const float inf = __builtin_inff ();
    const rep_src_map src_of (p);
    if (blockIdx.x < nbr) {
        const uint32_t r = blockIdx.x * 64u + lane;
        float4 *R4 = reinterpret_cast<float4 *> (p.R + (size_t) b * p.nr * 8);
        bool at0 = false;
        if (r < p.nr) {
            const uint32_t src = src_of (r);
            const float4 g = F4[2 * (size_t) src];
            R4[2 * r] = g;
            R4[2 * r + 1] = F4[2 * (size_t) src + 1];
            p.rep_src[(size_t) b * p.nr + r] = src;
            at0 = g.x == 0.f && g.y == 0.f && g.z == 0.f;
        }
        // The representatives at the origin (a frame's invalid points), listed in ascending order.  Large sets (more than ICP_ORIGIN_SOLO
        // blocks of representatives): every block leaves the ballot of its 64, and the block that arrives last (a counter; nobody waits
        // for anybody) compacts the list from the ballots and from R — one wave that walked all of F's 4096 sampled points by itself took
        // 21 of this kernel's 27 us at |F| = 2^20.  Small sets: one more block walks them in one round of loads (below) — an arrival costs
        // every block a release fence, 256 of them at 64 batched registrations of 256 representatives tripled this kernel's time.
        if (nbr <= ICP_ORIGIN_SOLO) return;
        float4 *OL = p.OL + (size_t) b * ICP_OL_STRIDE (p.nr);
        unsigned long long *MK = reinterpret_cast<unsigned long long *> (OL + ICP_OL_MASKS (p.nr));
        const unsigned long long bal = __ballot (at0);
        uint32_t last = 0u;
        if (lane == 0) MK[blockIdx.x] = bal;
        __threadfence ();                                             // this block's R rows and ballot, before its arrival counts
        if (lane == 0) last = __hip_atomic_fetch_add (reinterpret_cast<uint32_t *> (OL) + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nbr - 1u ? 1u : 0u;
        if (!__builtin_amdgcn_readfirstlane (last)) return;
        __threadfence ();                                             // the others' R rows and ballots
        uint32_t run = 0u;
        for (uint32_t i0 = 0; i0 < nbr; i0 += 64u) {                  // 64 ballots at a time: lane l holds ballot i0 + l and the number of entries in front of it
            const unsigned long long mk = i0 + lane < nbr ? __hip_atomic_load (MK + i0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            const uint32_t cnt = (uint32_t) __builtin_popcountll (mk);
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up (incl, d); if ((int) lane >= d) incl += t; }
            const uint32_t before = run + incl - cnt, mlo = (uint32_t) mk, mhi = (uint32_t) (mk >> 32);
            if (__ballot (mk != 0ull)) {
                for (uint32_t u0 = 0; u0 < 64u; u0 += 8u) {           // eight ballots' colour loads in flight
                    float4 c[8]; uint32_t pos[8];
#pragma unroll
                    for (uint32_t k = 0; k < 8u; ++k) {
                        const uint32_t lo_ = (uint32_t) __builtin_amdgcn_readlane ((int) mlo, (int) (u0 + k)), hi_ = (uint32_t) __builtin_amdgcn_readlane ((int) mhi, (int) (u0 + k));
                        const unsigned long long m = ((unsigned long long) hi_ << 32) | lo_;
                        const uint32_t base = (uint32_t) __builtin_amdgcn_readlane ((int) before, (int) (u0 + k));
                        pos[k] = 0xffffffffu;
                        if ((m >> lane) & 1ull) {
                            pos[k] = base + (uint32_t) __builtin_popcountll (m & ((1ull << lane) - 1ull));
                            c[k] = R4[2 * ((i0 + u0 + k) * 64u + lane) + 1];
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8u; ++k)
                        if (pos[k] != 0xffffffffu) OL[1u + pos[k]] = make_float4 (c[k].x, c[k].y, c[k].z, __uint_as_float ((i0 + u0 + k) * 64u + lane));
                }
            }
            run += (uint32_t) __builtin_amdgcn_readlane ((int) incl, 63);
        }
        origin_list_close (p, b, lane, OL, run, nbr, s_ent, s_key, sort_cap);
    } else if (blockIdx.x < nbr + nbg) {
        const uint32_t g = (blockIdx.x - nbr) * 64u + lane;
        if (g >= p.n16) return;
        float4 lo = make_float4 (inf, inf, inf, 0.f), hi = make_float4 (-inf, -inf, -inf, 0.f);
        const bool tiled = p.gtile != 0u;
        const uint32_t lg = p.gtile - 1u, ty = tiled ? g >> lg : 0u, tx = tiled ? g & ((1u << lg) - 1u) : 0u;
        float4 v[16];                                                 // (all sixteen loads in flight, then the selects: a loop with an early
#pragma unroll                                                        // `continue` is sixteen dependent round trips)
        for (uint32_t e = 0; e < 16u; ++e) {
            const uint32_t r = tiled ? (4u * ty + (e >> 2)) * p.nrx + 4u * tx + (e & 3u) : g * 16u + e;
            v[e] = make_float4 (0.f, 0.f, 0.f, 0.f);
            if (r < p.nr) v[e] = F4[2 * (size_t) src_of (r)];
        }
#pragma unroll
        for (uint32_t e = 0; e < 16u; ++e) {
            // an invalid point (at the origin; so reads a slot beyond the set): kept out of the box, listed in p.OL (k_search: ks_origin_list)
            const bool in = !(v[e].x == 0.f && v[e].y == 0.f && v[e].z == 0.f);
            lo.x = fminf (lo.x, in ? v[e].x : inf); lo.y = fminf (lo.y, in ? v[e].y : inf); lo.z = fminf (lo.z, in ? v[e].z : inf);
            hi.x = fmaxf (hi.x, in ? v[e].x : -inf); hi.y = fmaxf (hi.y, in ? v[e].y : -inf); hi.z = fmaxf (hi.z, in ? v[e].z : -inf);
        }
        float4 *GB = p.GB + (size_t) b * 2 * (p.n16 + p.n1k);
        GB[2 * g] = lo;
        float *hi3 = reinterpret_cast<float *> (GB + 2 * g + 1);          // (hi.w of group box 0 belongs to the block that lists the representatives at the origin)
        hi3[0] = hi.x; hi3[1] = hi.y; hi3[2] = hi.z;
    } else if (nbr <= ICP_ORIGIN_SOLO && blockIdx.x == gridDim.x - 1u) {
        // (small sets) the representatives at the origin: one wave, a ballot and a running offset per 64 representatives, all geometry
        // loads in flight at once; the colour is fetched by the lanes that list an entry
        float4 *OL = p.OL + (size_t) b * ICP_OL_STRIDE (p.nr);
        float4 v[ICP_ORIGIN_SOLO];
#pragma unroll
        for (uint32_t u = 0; u < ICP_ORIGIN_SOLO; ++u) {
            const uint32_t r = 64u * u + lane;
            v[u] = make_float4 (1.f, 1.f, 1.f, 0.f);
            if (r < p.nr) v[u] = F4[2 * (size_t) src_of (r)];
        }
        uint32_t run = 0u;
#pragma unroll
        for (uint32_t u = 0; u < ICP_ORIGIN_SOLO; ++u) {
            const uint32_t r = 64u * u + lane;
            const bool at0 = r < p.nr && v[u].x == 0.f && v[u].y == 0.f && v[u].z == 0.f;
            const unsigned long long bal = __ballot (at0);
            if (at0) {
                const float4 c = F4[2 * (size_t) src_of (r) + 1];
                OL[1u + run + (uint32_t) __builtin_popcountll (bal & ((1ull << lane) - 1ull))] = make_float4 (c.x, c.y, c.z, __uint_as_float (r));
            }
            if (lane == 0 && u < nbr) reinterpret_cast<unsigned long long *> (OL + ICP_OL_MASKS (p.nr))[u] = bal;      // (the ballots: a search looks a seed of the right kind up in them — icp_other_kind_near)
            run += (uint32_t) __builtin_popcountll (bal);
        }
        origin_list_close (p, b, lane, OL, run, nbr, s_ent, s_key, sort_cap);
    } else {
        const uint32_t tile = blockIdx.x - nbr - nbg;
        float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
        const uint32_t r_end = min (p.nr, (tile + 1u) * p.tbox);
        for (uint32_t r0 = tile * p.tbox; r0 < r_end; r0 += 16u * 64u) {      // (sixteen passes' loads in flight: p.tbox = 256 or 1024)
            float4 v[16];
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const uint32_t r = r0 + 64u * u + lane;
                v[u] = make_float4 (0.f, 0.f, 0.f, 0.f);
                if (r < r_end) v[u] = F4[2 * (size_t) src_of (r)];
            }
#pragma unroll
            for (uint32_t u = 0; u < 16u; ++u) {
                const bool in = !(v[u].x == 0.f && v[u].y == 0.f && v[u].z == 0.f);
                lo[0] = fminf (lo[0], in ? v[u].x : inf); lo[1] = fminf (lo[1], in ? v[u].y : inf); lo[2] = fminf (lo[2], in ? v[u].z : inf);
                hi[0] = fmaxf (hi[0], in ? v[u].x : -inf); hi[1] = fmaxf (hi[1], in ? v[u].y : -inf); hi[2] = fmaxf (hi[2], in ? v[u].z : -inf);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = fminf (lo[k], __shfl_xor (lo[k], d)); hi[k] = fmaxf (hi[k], __shfl_xor (hi[k], d)); }
        if (lane == 0) {
            float4 *GB = p.GB + (size_t) b * 2 * (p.n16 + p.n1k) + 2u * p.n16;
            GB[2 * tile] = make_float4 (lo[0], lo[1], lo[2], 0.f);
            float *hi4 = reinterpret_cast<float *> (GB + 2 * tile + 1);       // (hi.w of tile box 0 belongs to the block that lists the representatives at the origin)
            hi4[0] = hi[0]; hi4[1] = hi[1]; hi4[2] = hi[2];
        }
    }
}

// RBC construct, step 1 (owner(x) = argmin_r d(x, R[r]), ties -> lowest r) is k_search<.., OWNER = true> below.

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

// step 3: N[r] = sum over chunks; chunk_hist[chunk][r] becomes the rank base of that chunk in list r.
// Block = 64 representatives x 16 groups of consecutive chunks (thread (rr, cg): coalesced over rr): the sum of every group,
// an exchange through LDS, then the group's chunks again with the running base — two parallel passes instead of one thread
// walking all chunks of its representative (|F| = 2^20: 1024 chunks, 79 -> 17 us).
__global__ __launch_bounds__ (1024) void k_count (icp_params p)
{
    __shared__ uint32_t s_sum[16][64];
    const uint32_t b = blockIdx.y, rr = threadIdx.x & 63u, cg = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * 64u + rr;
    const bool live = r < p.nr;
    const uint32_t cpg = (p.nchunk + 15u) / 16u, c_lo = min (cg * cpg, p.nchunk), c_hi = min (c_lo + cpg, p.nchunk);
    uint32_t *h0 = p.chunk_hist + (size_t) b * p.nchunk * p.nr + (live ? r : 0u);
    uint32_t sum = 0;
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += 8u) {                   // eight independent loads in flight
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) v[k] = (live && c0 + k < c_hi) ? h0[(size_t) (c0 + k) * p.nr] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) sum += v[k];
    }
    s_sum[cg][rr] = sum;
    __syncthreads ();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (uint32_t g = 0; g < 16u; ++g) { const uint32_t v = s_sum[g][rr]; run += (g < cg) ? v : 0u; total += v; }
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += 8u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) v[k] = (live && c0 + k < c_hi) ? h0[(size_t) (c0 + k) * p.nr] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            if (live && c0 + k < c_hi) { h0[(size_t) (c0 + k) * p.nr] = run; run += v[k]; }
    }
    if (live && cg == 0u) { p.N[(size_t) b * p.nr + r] = total; ICP_N_FULL (p, b)[r] = total; }
}

// step 4: O = exclusive scan of N (exclusiveScan_i semantics, kernels/scan_kernels.cl:188). One block of 1024 threads:
// per-thread runs, wave scans (shuffles), a scan of the 16 wave totals.
__global__ __launch_bounds__ (1024) void k_offsets (icp_params p)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t b = blockIdx.y, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t *N = p.N + (size_t) b * p.nr;
    uint32_t *O = p.O + (size_t) b * p.nr;
    const uint32_t per = (p.nr + 1023u) / 1024u, lo = t * per, hi = min (lo + per, p.nr);
    uint32_t part = 0;
    for (uint32_t r = lo; r < hi; ++r) part += N[r];
    uint32_t inc = part;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up (inc, d); if (lane >= d) inc += v; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads ();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    uint32_t run = base + inc - part;
    for (uint32_t r = lo; r < hi; ++r) { O[r] = run; run += N[r]; }
}

// steps 3 + 4 in one launch for |R| <= 1024 (one block, thread r = representative r): the serial walk over the chunks, then
// the block scan of the counts — the same integers as k_count + k_offsets.
__global__ __launch_bounds__ (1024) void k_count_offsets (icp_params p)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t b = blockIdx.y, r = threadIdx.x, lane = r & 63u, wave = r >> 6;
    uint32_t run = 0;
    if (r < p.nr) {
        uint32_t *h0 = p.chunk_hist + (size_t) b * p.nchunk * p.nr + r;
        for (uint32_t c0 = 0; c0 < p.nchunk; c0 += 8u) {              // eight independent loads in flight, then the serial scan
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) v[k] = (c0 + k < p.nchunk) ? h0[(size_t) (c0 + k) * p.nr] : 0u;
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                if (c0 + k < p.nchunk) { h0[(size_t) (c0 + k) * p.nr] = run; run += v[k]; }
        }
        p.N[(size_t) b * p.nr + r] = run; ICP_N_FULL (p, b)[r] = run;
    }
    uint32_t inc = run;                              // exclusive scan over r (exclusiveScan_i, kernels/scan_kernels.cl:188)
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up (inc, d); if (lane >= d) inc += v; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads ();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    if (r < p.nr) p.O[(size_t) b * p.nr + r] = base + inc - run;
}

// step 5: stable placement: position = O[owner] + #{j < i : owner[j] == owner[i]}; perm, X_P and the search copy.
// One block per chunk of 1024 points; the rank base of the chunk comes from step 3.  The rank inside the chunk: the points of
// a wave are 64 neighbours in index order and share a handful of owners, so every wave lists its distinct owners with
// their counts (one ballot per distinct owner: the lanes' rank inside the wave is a popcount of the lanes below), and a
// point adds the counts of its owner in the lists of the earlier waves (broadcast LDS reads of short lists) — instead of
// comparing itself with every earlier point of the chunk (round 2: 15 -> 5 us at |F| = 16384, where the kernel is a third
// of buildRBC).  Integer arithmetic: the same positions whatever the path.
__global__ __launch_bounds__ (1024) void k_place (icp_params p)
{
    __shared__ uint2 s_list[ICP_CHUNK / 64][64];     // per wave: (owner, count) of its distinct owners
    __shared__ uint32_t s_n[ICP_CHUNK / 64];
    const uint32_t chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane (t >> 6);
    const uint32_t i = chunk * ICP_CHUNK + t;
    const bool valid = i < p.m;
    const uint32_t own = valid ? p.owner[(size_t) b * p.m + i] : 0xFFFFFFFFu;
    uint32_t base = 0;
    if (valid) base = p.O[(size_t) b * p.nr + own] + p.chunk_hist[((size_t) b * p.nchunk + chunk) * p.nr + own];
    uint32_t rank = 0, k = 0;
    for (unsigned long long rem = __ballot (valid); rem; ++k) {
        const uint32_t o = (uint32_t) __builtin_amdgcn_readlane ((int) own, (int) __builtin_ctzll (rem));
        const unsigned long long same = __ballot (own == o);         // (an owner is < nr: never the marker of an invalid lane)
        if (own == o) rank = (uint32_t) __builtin_popcountll (same & ((1ull << lane) - 1ull));
        if (lane == 0) s_list[wave][k] = make_uint2 (o, (uint32_t) __builtin_popcountll (same));
        rem &= ~same;
    }
    if (lane == 0) s_n[wave] = k;
    __syncthreads ();
    for (uint32_t w = 0; w < wave; ++w) {            // earlier waves: every one of their points precedes t
        const uint32_t n = s_n[w];
        for (uint32_t e = 0; e < n; ++e) {
            const uint2 v = s_list[w][e];
            rank += (v.x == own) ? v.y : 0u;
        }
    }
    if (chunk == 0u && t == 0u && !p.no_state_reset) { icp_reg_state *st = p.st + b; st->k = 0; st->done = 0; st->pm_iters = 0; }     // ICP::buildRBC (:4796)
    if (valid) {
        const uint32_t pos = base + rank;
        const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
        float4 *X4 = reinterpret_cast<float4 *> (p.XP + (size_t) b * p.m * 8);
        p.perm[(size_t) b * p.m + pos] = i;
        float4 g = F4[2 * (size_t) i], c = F4[2 * (size_t) i + 1];
        X4[2 * (size_t) pos] = g;
        X4[2 * (size_t) pos + 1] = c;
        // search copy, laid out for packed fp32 math: [x r y g | z b id 0] — (geometry, colour) pairs side by side, and
        // the unused homogeneous lane carries the original index (saves the perm[] round trip)
        float4 *Q4 = reinterpret_cast<float4 *> (p.XQ + (size_t) b * p.m * 8);
        Q4[2 * (size_t) pos] = make_float4 (g.x, c.x, g.y, c.y);
        Q4[2 * (size_t) pos + 1] = make_float4 (g.z, c.z, __uint_as_float (i), 0.f);
    }
}

// the boxes of a list (n positions from offset o) of registration b: 16-lane row `row` of `nrows` takes the chunks c_first + row, + nrows, ..
static __device__ __forceinline__ void list_boxes_of (const icp_params &p, uint32_t b, uint32_t n, uint32_t o, uint32_t c_first, uint32_t l, uint32_t row, uint32_t nrows)
{
    const uint32_t nch = (n + 15u) >> 4;
    const float4 *Q4 = reinterpret_cast<const float4 *> (p.XQ + (size_t) b * p.m * 8);
    float4 *LB = p.LB + (size_t) b * 3 * p.nlb;
    const float inf = __builtin_inff ();
    for (uint32_t c = c_first + row; c < nch; c += nrows) {
        const uint32_t j = 16u * c + l;
        float lo[6] = { inf, inf, inf, inf, inf, inf }, hi[6] = { -inf, -inf, -inf, -inf, -inf, -inf };
        if (j < n) {
            const float4 g = Q4[2 * (size_t) (o + j)], cc = Q4[2 * (size_t) (o + j) + 1];     // [x r y g | z b id 0]
            lo[0] = hi[0] = g.x; lo[1] = hi[1] = g.z; lo[2] = hi[2] = cc.x; lo[3] = hi[3] = g.y; lo[4] = hi[4] = g.w; lo[5] = hi[5] = cc.y;
#pragma unroll
            for (int k = 0; k < 6; ++k) if (lo[k] != lo[k]) { lo[k] = inf; hi[k] = -inf; }    // NaN: out of the box
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1)
#pragma unroll
            for (int k = 0; k < 6; ++k) { lo[k] = fminf (lo[k], __shfl_xor (lo[k], d, 16)); hi[k] = fmaxf (hi[k], __shfl_xor (hi[k], d, 16)); }
        if (l == 0) {
            float4 *dst = LB + 3 * (size_t) ((o >> 4) + c);
            dst[0] = make_float4 (lo[0], lo[1], lo[2], lo[3]);
            dst[1] = make_float4 (lo[4], lo[5], hi[0], hi[1]);
            dst[2] = make_float4 (hi[2], hi[3], hi[4], hi[5]);
        }
    }
}

// RBC construct, steps 2 - 5 in ONE launch for the latency-bound sizes (at most 512 blocks of 64 points over the batch, |R| < 1024:
// the sizes whose owner search is k_search<.., OWNER, MINW = 2>, which leaves owner[], the rank of every point inside its block of
// 64 and the block's (owner, count) list).  A block places 256 consecutive points = 4 owner blocks.  Nothing here waits for
// another block: every block re-derives what it needs from the lists of ALL owner blocks (a few KB, L2-resident) —
//   total[r]  = points owned by r                       (N; its exclusive scan is O: exclusiveScan_i, kernels/scan_kernels.cl:188)
//   before[r] = points owned by r in earlier chunks
// with integer LDS atomics (deterministic), then position = O[owner] + before[owner] + counts of the owner in the chunk's earlier
// owner blocks + rank inside the own block: the stable order by original index, the same integers as k_chunk_hist +
// k_count_offsets + k_place.  Block 0 also writes N and O and resets k / done (ICP::buildRBC, src/ICP/algorithms.cpp:4796).
// buildRBC at |F| = 16384: 6 launches, 33.6 us -> 2 launches.
__global__ __launch_bounds__ (256) void k_place_lists (icp_params p)
{
    __shared__ uint32_t s_total[1024], s_before[1024];
    __shared__ uint2 s_list[4][64];
    __shared__ uint32_t s_n[4], s_wave[4];
    const uint32_t c = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane (t >> 6);
    const uint32_t nb = p.nb, ob0 = c * 4u, i = c * 256u + t;
    const bool valid = i < p.m;
    // every load that depends on nothing is issued first: the point, its owner and rank, the lists' lengths
    const float4 *F4 = reinterpret_cast<const float4 *> (p.F + (size_t) b * p.m * 8);
    float4 g = make_float4 (0.f, 0.f, 0.f, 0.f), cc = g;
    uint32_t own = 0xFFFFFFFFu, rk = 0u;
    if (valid) { own = p.owner[(size_t) b * p.m + i]; rk = p.brank[(size_t) b * p.m + i]; g = F4[2 * (size_t) i]; cc = F4[2 * (size_t) i + 1]; }
    const uint2 *BL = p.blist + (size_t) b * nb * 64u;
    const uint32_t *BN = p.bn + (size_t) b * nb;
    // the list of owner block t (thread t; further ones in the loop below): its length and, without waiting for it, its first 16
    // entries — one memory round trip for everything above and this.  (64 neighbouring points share a dozen owners: 12 at the median,
    // 16 at the 90th percentile, 20 at most on the benchmark pair and on a tracked frame; with 8 entries up front nearly every thread
    // walked four more, one dependent load each: buildRBC at |F| = 16384 13.85 -> 13.5 us.)
    constexpr int KP_UP = 8;                          // uint4 = pairs of entries fetched up front
    uint32_t n0 = 0u; uint4 e0[KP_UP];
#pragma unroll
    for (int k = 0; k < KP_UP; ++k) e0[k] = make_uint4 (0u, 0u, 0u, 0u);
    if (t < nb) {
        n0 = BN[t];
        const uint4 *q = reinterpret_cast<const uint4 *> (BL + (size_t) t * 64u);
#pragma unroll
        for (int k = 0; k < KP_UP; ++k) e0[k] = q[k];
    }
    uint32_t nown = 0u; uint2 lown = make_uint2 (0u, 0u);
    if (ob0 + wave < nb) {                           // this chunk's own lists
        nown = BN[ob0 + wave];
        lown = BL[(size_t) (ob0 + wave) * 64u + lane];      // (entries past the list's end: whatever the buffer holds, never read back)
    }
    for (uint32_t r = t; r < p.nr; r += 256u) { s_total[r] = 0u; s_before[r] = 0u; }
    s_list[wave][lane] = lown;
    if (lane == 0) s_n[wave] = nown;
    __syncthreads ();
    {
        const bool earlier = t < ob0;
#pragma unroll
        for (int e = 0; e < 2 * KP_UP; ++e) {
            const uint32_t o_ = (e & 1) ? e0[e >> 1].z : e0[e >> 1].x, n_ = (e & 1) ? e0[e >> 1].w : e0[e >> 1].y;
            if ((uint32_t) e < n0) { atomicAdd (&s_total[o_], n_); if (earlier) atomicAdd (&s_before[o_], n_); }
        }
        // (the rest of a long list sixteen entries per round trip, not one: a frame's invalid points — each with the representative at the
        // origin whose colour is nearest — give a block of 64 points 30 - 40 owners: buildRBC at |F| = 16384 with 10 % of them 39 -> 2x us)
        for (uint32_t e0 = 2u * KP_UP; e0 < n0; e0 += 2u * KP_UP) {
            const uint4 *q = reinterpret_cast<const uint4 *> (BL + (size_t) t * 64u + e0);
            uint4 e1[KP_UP];
#pragma unroll
            for (int k = 0; k < KP_UP; ++k) e1[k] = q[k];
#pragma unroll
            for (int e = 0; e < 2 * KP_UP; ++e) {
                const uint32_t o_ = (e & 1) ? e1[e >> 1].z : e1[e >> 1].x, n_ = (e & 1) ? e1[e >> 1].w : e1[e >> 1].y;
                if (e0 + (uint32_t) e < n0) { atomicAdd (&s_total[o_], n_); if (earlier) atomicAdd (&s_before[o_], n_); }
            }
        }
    }
    for (uint32_t ob = t + 256u; ob < nb; ob += 256u) {
        const uint32_t n = BN[ob];
        const bool earlier = ob < ob0;
        for (uint32_t e = 0; e < n; ++e) {
            const uint2 v = BL[(size_t) ob * 64u + e];
            atomicAdd (&s_total[v.x], v.y);
            if (earlier) atomicAdd (&s_before[v.x], v.y);
        }
    }
    __syncthreads ();
    // O = exclusive scan of the totals; thread t takes `per` consecutive representatives (|R| < 1024: per <= 4)
    const uint32_t per = (p.nr + 255u) / 256u, lo = min (t * per, p.nr), hi = min (lo + per, p.nr);
    uint32_t part = 0u;
    for (uint32_t r = lo; r < hi; ++r) part += s_total[r];
    uint32_t inc = part;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up (inc, d); if (lane >= d) inc += v; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads ();
    uint32_t run = inc - part;
    for (uint32_t w = 0; w < wave; ++w) run += s_wave[w];
    for (uint32_t r = lo; r < hi; ++r) {
        const uint32_t n = s_total[r];
        if (c == 0u) { p.N[(size_t) b * p.nr + r] = n; ICP_N_FULL (p, b)[r] = n; p.O[(size_t) b * p.nr + r] = run; }
        s_before[r] += run;                          // position of the chunk's first point of list r
        run += n;
    }
    if (c == 0u && t == 0u && !p.no_state_reset) { icp_reg_state *st = p.st + b; st->k = 0; st->done = 0; st->pm_iters = 0; }
    __syncthreads ();
    if (valid) {
        uint32_t pos = s_before[own] + rk;
        for (uint32_t w = 0; w < wave; ++w) {        // the chunk's earlier owner blocks: every one of their points precedes this one
            const uint32_t n = s_n[w];
            for (uint32_t e = 0; e < n; ++e) { const uint2 v = s_list[w][e]; pos += (v.x == own) ? v.y : 0u; }
        }
        float4 *X4 = reinterpret_cast<float4 *> (p.XP + (size_t) b * p.m * 8);
        float4 *Q4 = reinterpret_cast<float4 *> (p.XQ + (size_t) b * p.m * 8);
        p.perm[(size_t) b * p.m + pos] = i;
        X4[2 * (size_t) pos] = g;
        X4[2 * (size_t) pos + 1] = cc;
        Q4[2 * (size_t) pos] = make_float4 (g.x, cc.x, g.y, cc.y);                      // search copy: see k_place
        Q4[2 * (size_t) pos + 1] = make_float4 (g.z, cc.z, __uint_as_float (i), 0.f);
    }
}

// 6-D bounding boxes of the list chunks (stage-2 pruning of long lists, k_search): chunk c of list r = its positions 16 c .. 16 c + 15.
// A one-shot search scans the whole list of the query's representative; a list that is much longer than the others — every invalid
// point of a frame whose colours were zeroed too is ONE point, and one representative owns them all; a cluttered corner of a scene —
// is scanned 16 (8) positions at a time by every query that lands in it.  With a box per chunk a query tests a chunk before it loads
// it.  One block per representative, one 16-lane row per chunk; box of chunk c >= 1 at index (O[r] >> 4) + c: unique over all lists
// (a list's last chunk can share floor (position / 16) only with the NEXT list's chunk 0, which has no box), < m / 16 + 1.
// Chunk 0 is always scanned.  min / max are exact in any order; fminf / fmaxf skip NaN coordinates (such a point never wins a '<').
// Round 6 — members of a long list's tail that REPEAT an earlier member.  A frame's invalid points whose colour was zeroed too are ONE
// point: one representative owns them all (|F| = 16384 with 30 % of them: a list of 4973; |F| = 65536: 19673), and every query that is
// such a point walked that list's chunk boxes (311 / 1230 box tests per query for nothing: 15.8 / 127 us per iteration).  A member that
// equals an earlier member of its list in all six coordinates can win neither a strict '<' nor the tie for the lowest position, whatever
// the query (equal coordinates: the same differences, the same distance bits; +0 and -0 give the same differences too): the scan may
// skip it.  The block of a long list therefore compacts the list's tail IN PLACE in the search copy XQ (the head — the positions every scan
// takes unconditionally — stays as it is): a tail member is dropped when it equals its predecessor in the list or the list's first
// member (sufficient, not necessary: runs of one point and repeats of the first point — both are what invalid points make; any other
// repeat is merely scanned), the order of the rest is kept, and the search's view of the list's length (p.N) shrinks to head + kept;
// N, O, perm, XP — the construction's outputs — are untouched, and a record carries its original index, so every output of a search is
// the one of the exhaustive scan.  The chunk boxes are then those of the compacted tail.
// One pass = 256 threads x 4 consecutive positions; reads of a pass complete (barrier) before its writes, which land at or below the
// pass's own positions: later passes read untouched records; the predecessor of a pass's first position travels through LDS.
__global__ __launch_bounds__ (256) void k_list_boxes (icp_params p)
{
    const uint32_t r = blockIdx.x, b = blockIdx.y;
    const uint32_t n = ICP_N_FULL (p, b)[r];
    // (only what the scans ask for: a list's first ICP_S2_UNCOND = 128 positions are always scanned as they come — ks_stage2_lanes; a wave
    // whose longest list has more than 256 tests every list of its queries from chunk 8 on: ks_list_tail —; on a clean frame of the
    // reference's size every block of this kernel leaves here)
    // (lanes = candidates — p.s2wave: a list's first ICP_S2W_UNCOND = 1024 positions are scanned wave-cooperatively, the tail from chunk 64 on)
    const uint32_t n_min = p.s2wave ? 1024u : 128u, c_first = p.s2wave ? 64u : 8u;
    if (n <= n_min) return;
    const uint32_t o = p.O[(size_t) b * p.nr + r], t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    __shared__ float s_prev[6];
    __shared__ uint32_t s_cnt[4];
    float4 *Q4 = reinterpret_cast<float4 *> (p.XQ + (size_t) b * p.m * 8);
    const float4 fg = Q4[2 * (size_t) o], fc = Q4[2 * (size_t) o + 1];                   // the list's first member [x r y g | z b id 0]
    if (t == 0) {
        const float4 g = Q4[2 * (size_t) (o + n_min - 1u)], c = Q4[2 * (size_t) (o + n_min - 1u) + 1];
        s_prev[0] = g.x; s_prev[1] = g.y; s_prev[2] = g.z; s_prev[3] = g.w; s_prev[4] = c.x; s_prev[5] = c.y;
    }
    __syncthreads ();
    constexpr uint32_t E = 4u;
    uint32_t kept = 0u;                              // tail members kept so far (the same number in every thread)
    for (uint32_t s = n_min; s < n; s += 256u * E) {
        const uint32_t j0 = s + t * E;
        float4 g[E + 1], c[E + 1];                   // [0]: the predecessor of the thread's first position
        if (t == 0) { g[0] = make_float4 (s_prev[0], s_prev[1], s_prev[2], s_prev[3]); c[0] = make_float4 (s_prev[4], s_prev[5], 0.f, 0.f); }
        else if (j0 <= n) { g[0] = Q4[2 * (size_t) (o + j0 - 1u)]; c[0] = Q4[2 * (size_t) (o + j0 - 1u) + 1]; }
#pragma unroll
        for (uint32_t e = 0; e < E; ++e)
            if (j0 + e < n) { g[e + 1] = Q4[2 * (size_t) (o + j0 + e)]; c[e + 1] = Q4[2 * (size_t) (o + j0 + e) + 1]; }
        uint32_t keep = 0u;
#pragma unroll
        for (uint32_t e = 0; e < E; ++e) {
            if (j0 + e >= n) break;
            const float4 a = g[e + 1], a2 = c[e + 1], q = g[e], q2 = c[e];
            const bool as_prev = a.x == q.x && a.y == q.y && a.z == q.z && a.w == q.w && a2.x == q2.x && a2.y == q2.y;
            const bool as_first = a.x == fg.x && a.y == fg.y && a.z == fg.z && a.w == fg.w && a2.x == fc.x && a2.y == fc.y;
            if (!as_prev && !as_first) keep |= 1u << e;
        }
        const uint32_t cnt = (uint32_t) __builtin_popcount (keep);
        uint32_t inc = cnt;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up (inc, d); if (lane >= d) inc += v; }
        __syncthreads ();                            // (every read of this pass is done; s_prev / s_cnt of the previous pass have been used)
        if (lane == 63u) s_cnt[wave] = inc;
        const uint32_t last = min (s + 256u * E, n) - 1u;       // the pass's last position: the next pass's predecessor
#pragma unroll
        for (uint32_t e = 0; e < E; ++e)
            if (j0 + e == last) { s_prev[0] = g[e + 1].x; s_prev[1] = g[e + 1].y; s_prev[2] = g[e + 1].z; s_prev[3] = g[e + 1].w; s_prev[4] = c[e + 1].x; s_prev[5] = c[e + 1].y; }
        __syncthreads ();
        uint32_t base = kept + inc - cnt, total = 0u;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) { const uint32_t v = s_cnt[w]; base += (w < wave) ? v : 0u; total += v; }
        if (kept + total != min (s + 256u * E, n) - n_min) {          // (else: nothing dropped so far, every record is where it belongs)
            uint32_t k = 0u;
#pragma unroll
            for (uint32_t e = 0; e < E; ++e)
                if (keep & (1u << e)) {
                    const size_t dst = (size_t) o + n_min + base + k;
                    Q4[2 * dst] = g[e + 1]; Q4[2 * dst + 1] = c[e + 1];
                    ++k;
                }
        }
        kept += total;
    }
    const uint32_t ns = n_min + kept;
    if (ns != n) {
        if (t == 0) p.N[(size_t) b * p.nr + r] = ns;
        // (the boxes below read records other threads of the block have just written: write-back + invalidate of this CU's vector cache on both sides of the barrier)
        __threadfence ();
        __syncthreads ();
        __threadfence ();
    }
    if (ns <= n_min) return;
    list_boxes_of (p, b, ns, o, c_first, t & 15u, t >> 4, 16u);
}

// ICPPowerMethod as a kernel of its own (reference include/ICP/algorithms.hpp:1451-1537, kernels/icp_kernels.cl:977-1054: an
// enqueueTask of one work-item; here one wave): S[11], means[8] -> Tk[8] with the rotation solvers the iteration uses
// (icp_power_method_quad literal / squared start, icp_svd_rotation) — the entry the reference's known-answer test drives
// (tests/testsICP.cpp:988-1052).  out[0..8) = Tk, out[8..17) = Rk (EIGEN branch; else the rotation of qk), out[17] = loop trips.
template <int ROT>
__global__ __launch_bounds__ (64) void k_rotation_solver (const float *gin, float *gout, int power_mode)
{
    const uint32_t lane = threadIdx.x;
    float S[11], means[8], Tk[8], Rk[9];
#pragma unroll
    for (int k = 0; k < 11; ++k) S[k] = gin[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) means[k] = gin[11 + k];
    int iters = 0;
    if constexpr (ROT == 1) { iters = icp_power_method_quad (S, means, Tk, power_mode, lane); icp_quat_to_rot (Tk, Rk); }
    else icp_svd_rotation (S, means, Rk, Tk);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) gout[k] = Tk[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) gout[8 + k] = Rk[k];
        gout[17] = __uint_as_float ((uint32_t) iters);
    }
}
void icp_launch_rotation_solver (int rot, int power_mode, const float *din19, float *dout18, hipStream_t s)
{
    if (rot == 1) hipLaunchKernelGGL (k_rotation_solver<1>, dim3 (1), dim3 (64), 0, s, din19, dout18, power_mode);
    else hipLaunchKernelGGL (k_rotation_solver<0>, dim3 (1), dim3 (64), 0, s, din19, dout18, power_mode);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void icp_launch_reset_state (const icp_params &p, hipStream_t s, int reset_T)
{
    hipLaunchKernelGGL (k_reset_state, dim3 ((p.batch + 63) / 64), dim3 (64), 0, s, p, reset_T);
}

void icp_launch_set_T (const icp_params &p, uint32_t b, const float *dT8, hipStream_t s, int reset_k)
{
    hipLaunchKernelGGL (k_set_T, dim3 (1), dim3 (64), 0, s, p.st + b, dT8, reset_k);
}

// Gate of a tracked frame: one wave that holds its stream until registration `want` of the sequence has finished (*seq >= want, written
// by the launch that found the previous frame converged, or by its end kernel) — the previous frame runs on the OTHER stream, and the
// launches behind this kernel were enqueued while it was still running.  Every wave reaches the exit: the wait is bounded (max_spins
// s_sleep rounds: ~0.5 s by default, far beyond any registration); a timeout raises *flag (host memory) and lets the stream go on.
// warm != nullptr: a warm-started frame — once the gate is open (the acquire stands behind the previous registration's final state) the same
// lane does what k_set_T would do in a launch of its own (T stays, the cumulative rotation is re-derived from its quaternion, k = done = 0):
// one kernel and one launch boundary fewer between two frames.
// A gate that gives up (the predecessor is launch-complete or converged before the frame behind it is handed to the device — see track_submit —,
// so this is a device that has stopped making progress, not a slow caller) turns the frame behind it into no-ops: it raises the frame's own
// run flag (run_flag = epoch: every launch of the run leaves at its first load and stores nothing, the end kernel too), so nothing runs into
// the predecessor's state slots and moment buffers; the host finds *flag set and reports the stall.
__global__ __launch_bounds__ (64) void k_gate (const uint32_t *seq, uint32_t want, uint32_t *flag, uint32_t max_spins, icp_reg_state *warm, uint32_t *run_flag, uint32_t epoch)
{
    if (threadIdx.x != 0) return;
    bool open = false;
    for (uint32_t spin = 0; spin < max_spins && !open; ++spin) {
        const uint32_t v = __hip_atomic_load (seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        open = (int32_t) (v - want) >= 0;
        if (!open) __builtin_amdgcn_s_sleep (8);
    }
    if (!open) {
        if (run_flag) *run_flag = epoch;
        if (flag) __hip_atomic_store (flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (warm) {
        warm->k = 0; warm->done = 0; warm->pm_iters = 0;
        float T[4]; for (int i = 0; i < 4; ++i) T[i] = warm->T[i];
        float R[9]; icp_quat_to_rot (T, R);
        for (int i = 0; i < 9; ++i) warm->R[i] = R[i];
    }
}

void icp_launch_gate (const uint32_t *seq, uint32_t want, uint32_t *host_timeout_flag, hipStream_t s, uint32_t max_spins, icp_reg_state *warm, uint32_t *run_flag, uint32_t epoch)
{
    hipLaunchKernelGGL (k_gate, dim3 (1), dim3 (64), 0, s, seq, want, host_timeout_flag, max_spins, warm, run_flag, epoch);
}

__global__ void k_seq_set (uint32_t *seq, uint32_t v) { if (threadIdx.x == 0) __hip_atomic_store (seq, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
void icp_launch_seq_set (uint32_t *seq, uint32_t v, hipStream_t s) { hipLaunchKernelGGL (k_seq_set, dim3 (1), dim3 (64), 0, s, seq, v); }

void icp_launch_get_lms (const float *cloud, float *lms, hipStream_t s)
{
    hipLaunchKernelGGL (k_get_lms, dim3 (16384 * 2 / 256), dim3 (256), 0, s,
                        reinterpret_cast<const float4 *> (cloud), reinterpret_cast<float4 *> (lms));
}

void icp_launch_get_lms_band (const float *band, float *lms, hipStream_t s)
{
    hipLaunchKernelGGL (k_get_lms_band, dim3 (16384 * 2 / 256), dim3 (256), 0, s,
                        reinterpret_cast<const float4 *> (band), reinterpret_cast<float4 *> (lms));
}

void icp_launch_transform_cloud (const float *in, float *out, const icp_reg_state *st, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL (k_transform_cloud, dim3 ((n + 255) / 256), dim3 (256), 0, s,
                        reinterpret_cast<const float4 *> (in), reinterpret_cast<float4 *> (out), st, n);
}

void icp_launch_transform_cloud_ex (int kind, const float *in, float *out, const float *T, uint32_t n, hipStream_t s)
{
    icp_T16 t {};
    for (int i = 0; i < (kind == 2 ? 16 : 8); ++i) t.v[i] = T[i];
    const float4 *i4 = reinterpret_cast<const float4 *> (in); float4 *o4 = reinterpret_cast<float4 *> (out);
    const dim3 grid ((n + 255) / 256), block (256);
    if (kind == 0) hipLaunchKernelGGL (k_transform_cloud_ex<0>, grid, block, 0, s, i4, o4, t, n);
    else if (kind == 1) hipLaunchKernelGGL (k_transform_cloud_ex<1>, grid, block, 0, s, i4, o4, t, n);
    else hipLaunchKernelGGL (k_transform_cloud_ex<2>, grid, block, 0, s, i4, o4, t, n);
}

void icp_launch_build_rbc (const icp_params &p, hipStream_t s)
{
    if (icp_build_lists (p)) {                       // two launches: the owner search (gathers the representatives itself, leaves the lists), the placement
        icp_launch_owner_search (p, s);
        hipLaunchKernelGGL (k_place_lists, dim3 ((p.nb + 3u) / 4u, p.batch), dim3 (256), 0, s, p);
        // (the chunk boxes of long lists: folded into k_place_lists — its blocks count their arrivals where a long list exists, the last one
        // builds the boxes — the construction of a clean frame took 13.5 instead of 15.2 us, but that of a frame with invalid points, which
        // always has a few lists beyond 128 positions, 35 instead of 19: every block's release fence writes the XCD's L2 back, megabytes
        // of fresh placements.  The launch boundary does that once.)
        hipLaunchKernelGGL (k_list_boxes, dim3 (p.nr, p.batch), dim3 (256), 0, s, p);
        return;
    }
    {   // the representatives, the boxes of their pruning groups and (several tiles only) of the LDS tiles: one launch
        const uint32_t nbr = (p.nr + 63u) / 64u, nbg = (p.n16 + 63u) / 64u, nbt = p.nr > p.tbox ? p.n1k : 0u;
        // (the ordered origin list: sets large enough to hold a list the search prunes by boxes, launches small enough to afford the LDS)
        const uint32_t nblk = (nbr + nbg + nbt + (nbr <= ICP_ORIGIN_SOLO ? 1u : 0u)) * p.batch;
        uint32_t sort_cap = 0u;
        if (p.nr >= 256u && nblk <= 2048u) { sort_cap = 256u; while (sort_cap < p.nr && sort_cap < ICP_OL_SORT_MAX) sort_cap <<= 1; }
        hipLaunchKernelGGL (k_reps_and_boxes, dim3 (nbr + nbg + nbt + (nbr <= ICP_ORIGIN_SOLO ? 1u : 0u), p.batch), dim3 (64), sort_cap * 20u, s, p, nbr, nbg, sort_cap);    // (+ 1: small sets' origin list)
    }
    icp_launch_owner_search (p, s);                  // step 1, owner(x) = nearest representative (icp_kernels.hip)
    hipLaunchKernelGGL (k_chunk_hist, dim3 (p.nchunk, p.batch), dim3 (256), p.nr * sizeof (uint32_t), s, p);
    if (p.nr <= 1024u) hipLaunchKernelGGL (k_count_offsets, dim3 (1, p.batch), dim3 (1024), 0, s, p);
    else {
        hipLaunchKernelGGL (k_count, dim3 ((p.nr + 63) / 64, p.batch), dim3 (1024), 0, s, p);
        hipLaunchKernelGGL (k_offsets, dim3 (1, p.batch), dim3 (1024), 0, s, p);
    }
    hipLaunchKernelGGL (k_place, dim3 (p.nchunk, p.batch), dim3 (1024), 0, s, p);
    hipLaunchKernelGGL (k_list_boxes, dim3 (p.nr, p.batch), dim3 (256), 0, s, p);
}
