// icp_device.h — canonical arithmetic of the ICP iteration on gfx950 (device functions).
//
// Everything here is the GPU twin of DESIGN.md §3 / oracle/icp_oracle.c: fp32 round-to-nearest,
// no FMA contraction (the translation unit is built with -ffp-contract=off), IEEE divide/sqrt,
// reduction trees of the reference's shape for a 64-wide wavefront.  One wavefront == one
// reference work-group, so the reference's LDS tree
//     data[0..128) ; for d = 64,32,..,1 : data[i] += data[i+d]            (kernels/icp_kernels.cl:170-175)
// is evaluated by 16-lane DPP rows that hold 8 positions per lane (see "row trees" below): same
// additions, same order, so lane 0 of the row ends with the bit pattern the reference leaves in data[0].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ICP_WF 64

struct icp_dist_id { float dist; uint32_t id; };   // kernels/icp_kernels.cl:34-38

// Device-resident state of one registration (what the reference keeps on the host in
// ICPStep::{R,q,t,s,Rk,qk,tk,sk} + ICP::k, src/ICP/algorithms.cpp:4683-4695, 4826).
struct icp_reg_state {
    float T[8];          // [q | t, s] cumulative — D_IO_T
    float Tk[8];         // [qk | tk, sk]
    float R[9];          // cumulative rotation, row-major
    float Rk[9];
    float S[11];
    float pad0;
    float means[8];      // [mean_f,0 | mean_m,0]
    double sum_w;
    uint32_t k;          // iterations executed
    uint32_t done;       // 1: ICP::check() said stop
    uint32_t pm_iters;
    uint32_t pending;    // chained fused mode: moments of an iteration are waiting to be turned into T
    uint32_t reserved0, reserved1;   // (the state is 62 dwords: one lane-per-dword vector load, 15 x 16-byte + one 8-byte store)
};

// ------------------------------------------------------------------------------------------
// row trees
//
// The canonical tree over 128 positions P[0..128) (result in P[0]):
//     for d = 64, 32, 16, 8, 4, 2, 1 :  P[i] += P[i+d]   (i < d)
// is evaluated by one 16-lane DPP row: lane l holds P[l + 16k], k = 0..7.  The levels d = 64, 32, 16
// pair positions held by the same lane (register adds); d = 8, 4, 2, 1 pair lane l with lane l+d of
// the row: v_add_f32_dpp row_shl:d.  Only lane 0 of the row ends with P[0]; the other lanes hold
// values the reference tree also computes and discards.  No LDS, no ds_bpermute, no cross-row traffic.
// All 64 lanes must be active when these are called (DPP reads of disabled lanes return 0).
// ------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float icp_dpp (float v)
{
    return __builtin_bit_cast (float, __builtin_amdgcn_update_dpp (0, __builtin_bit_cast (int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double icp_dpp_d (double v)
{
    unsigned long long u = __builtin_bit_cast (unsigned long long, v);
    int lo = (int) (u & 0xFFFFFFFFull), hi = (int) (u >> 32);
    lo = __builtin_amdgcn_update_dpp (0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp (0, hi, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast (double, ((unsigned long long) (unsigned int) hi << 32) | (unsigned int) lo);
}
#define ICP_ROW_SHL(n) (0x100 + (n))
#define ICP_QUAD_BCAST(k) ((k) * 0x55)

// levels d = 8, 4, 2, 1 across the 16 lanes of a row
__device__ __forceinline__ float row_tree_tail (float v)
{
    v = v + icp_dpp<ICP_ROW_SHL (8)> (v);
    v = v + icp_dpp<ICP_ROW_SHL (4)> (v);
    v = v + icp_dpp<ICP_ROW_SHL (2)> (v);
    v = v + icp_dpp<ICP_ROW_SHL (1)> (v);
    return v;
}
__device__ __forceinline__ double row_tree_tail_d (double v)
{
    v = v + icp_dpp_d<ICP_ROW_SHL (8)> (v);
    v = v + icp_dpp_d<ICP_ROW_SHL (4)> (v);
    v = v + icp_dpp_d<ICP_ROW_SHL (2)> (v);
    v = v + icp_dpp_d<ICP_ROW_SHL (1)> (v);
    return v;
}
// full 128-position tree: a[k] = P[l + 16k]
__device__ __forceinline__ float row_tree8 (const float *a)
{
    float b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];     // d = 64
    float c0 = b0 + b2, c1 = b1 + b3;                                                 // d = 32
    return row_tree_tail (c0 + c1);                                                   // d = 16, then 8..1
}
__device__ __forceinline__ double row_tree8_d (const double *a)
{
    double b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];
    double c0 = b0 + b2, c1 = b1 + b3;
    return row_tree_tail_d (c0 + c1);
}
// 64-position tree (levels d = 32 .. 1 of a tree whose d = 64 level was applied, or is absent): a[k] = P[l + 16k], k < 4
__device__ __forceinline__ float row_tree4 (const float *a)
{
    float c0 = a[0] + a[2], c1 = a[1] + a[3];
    return row_tree_tail (c0 + c1);
}

// ------------------------------------------------------------------------------------------
// element-wise pieces
// ------------------------------------------------------------------------------------------

// ASSUMPTION-METRIC — the single swap point on the GPU side (CPU twin: orc_metric8).
// d = fma (a, pho, geo), geo = fma (dz, dz, fma (dy, dy, dx*dx)), pho likewise on r g b; lanes 3 and 7 ignored
// (metric text: src/ICP/algorithms.cpp:4393-4398; RandomBallCover source is un-vendored).  The fmas are explicit:
// the translation unit is built with -ffp-contract=off, nothing else is contracted.
__device__ __forceinline__ float icp_metric8 (float qx, float qy, float qz, float qr, float qg, float qb,
                                              float x, float y, float z, float r, float g, float b, float a)
{
    float dx = qx - x, dy = qy - y, dz = qz - z;
    float dr = qr - r, dg = qg - g, db = qb - b;
    float geo = __builtin_fmaf (dz, dz, __builtin_fmaf (dy, dy, dx * dx));
    float pho = __builtin_fmaf (db, db, __builtin_fmaf (dg, dg, dr * dr));
    return __builtin_fmaf (a, pho, geo);
}

// icpTransform_Quaternion — kernels/icp_kernels.cl:789-801:
//   p' = t.w * (p + cross (2 q.xyz, cross (q.xyz, p) + q.w p)) + t.xyz
__device__ __forceinline__ void icp_transform_point (const float *T, float px, float py, float pz,
                                                     float &ox, float &oy, float &oz)
{
    float qx = T[0], qy = T[1], qz = T[2], qw = T[3];
    float ux = (qy * pz - qz * py) + qw * px;
    float uy = (qz * px - qx * pz) + qw * py;
    float uz = (qx * py - qy * px) + qw * pz;
    float ax = 2 * qx, ay = 2 * qy, az = 2 * qz;
    float vx = ay * uz - az * uy;
    float vy = az * ux - ax * uz;
    float vz = ax * uy - ay * ux;
    ox = T[7] * (px + vx) + T[4];
    oy = T[7] * (py + vy) + T[5];
    oz = T[7] * (pz + vz) + T[6];
}

// ------------------------------------------------------------------------------------------
// a9  icpPowerMethod — kernels/icp_kernels.cl:977-1054 (canonical forms: oracle power_impl)
//
// Lane-parallel: every quad of the wave holds the same data, lane (l & 3) = i owns row i of N (and of B)
// and component i of x.  The operations and their order are exactly those of the oracle; only the
// placement changes:
//   N x        : y_i = (((0 + N[i][0] x_0) + N[i][1] x_1) + N[i][2] x_2) + N[i][3] x_3, x_k read from lane k
//   sums of 4  : sequential (((0 + v_0) + v_1) + v_2) + v_3 with v_k read from lane k
//   B B        : (squared start) four v_mfma_f32_4x4x1 steps: with one (symmetric) row per lane the A and B
//                operands of step k are the same register, and the result comes back one row per lane; the
//                instruction evaluates the k-ordered fmaf chain the oracle writes (tests/cpp/mfma4_test.hip)
// All 64 lanes must be active.  Returns the loop-trip count; Tk is valid in every lane.
// ------------------------------------------------------------------------------------------
#define ICP_PM_SQUARINGS 10
typedef float icp_f4 __attribute__ ((ext_vector_type (4)));

// value of lane k of the quad, in every lane of the quad: one DPP quad broadcast (no SGPR round trip, no hazards)
template <int K> __device__ __forceinline__ float pmq_q (float v) { return icp_dpp<ICP_QUAD_BCAST (K)> (v); }
// the same as a wave-uniform scalar (control flow only)
__device__ __forceinline__ float pmq_lane (float v, int k)
{
    return __builtin_bit_cast (float, __builtin_amdgcn_readlane (__builtin_bit_cast (int, v), k));
}
__device__ __forceinline__ float pmq_seq4 (float v)
{
    float s = 0.f;
    s = s + pmq_q<0> (v); s = s + pmq_q<1> (v); s = s + pmq_q<2> (v); s = s + pmq_q<3> (v);
    return s;
}
__device__ __forceinline__ float pmq_matvec (const float *Nrow, float x)
{
    float s = 0.f;
    s = s + Nrow[0] * pmq_q<0> (x); s = s + Nrow[1] * pmq_q<1> (x);
    s = s + Nrow[2] * pmq_q<2> (x); s = s + Nrow[3] * pmq_q<3> (x);
    return s;
}
__device__ __forceinline__ float pmq_normalize (float y)
{
    float n = sqrtf (pmq_seq4 (y * y));
    return y / n;
}
__device__ __forceinline__ void pmq_rescale (float *Brow)
{   // exact power-of-two rescale so that max|entry| lies in [1,2)
    // (oracle rescale16; max is exact in any order: the quad's rows are combined with two DPP steps)
    // (four instructions, written out: fmaxf / fabsf compile to a canonicalising v_max per operand and a DPP move per step —
    // twelve instructions on the one wave the whole grid waits for; v_max3 / v_max return the non-NaN operand as fmaxf does, the
    // entries are results of arithmetic (no signalling NaNs).  s_nop 1: DPP operand hazard, opaque to the compiler's hazard pass.)
    float mx;
    asm ("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32_e64 %0, %0, |%4|\n\ts_nop 1\n\t"
         "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
         "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
         : "=&v"(mx) : "v"(Brow[0]), "v"(Brow[1]), "v"(Brow[2]), "v"(Brow[3]));
    const uint32_t e = (__float_as_uint (mx) >> 23) & 0xFFu;
    const float sc = (e == 0u || e >= 254u) ? 1.f : __uint_as_float ((254u - e) << 23);    // zero / subnormal / inf / nan: leave
    Brow[0] = Brow[0] * sc; Brow[1] = Brow[1] * sc; Brow[2] = Brow[2] * sc; Brow[3] = Brow[3] * sc;
}

#ifdef ICP_DBG_STAMPS
__device__ unsigned long long icp_pm_stamps[8];
#define PM_STAMP(k) { unsigned long long t_; asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0) icp_pm_stamps[k] = t_; }
#else
#define PM_STAMP(k)
#endif
__device__ inline int icp_power_method_quad (const float *S, const float *means, float *Tk, int squared_start, uint32_t lane)
{
    const uint32_t i = lane & 3u;
    const float sk = sqrtf (S[9] / S[10]);                             // :989 (independent of the eigenvector: issued first)
    float Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5],
          Szx = S[6], Szy = S[7], Szz = S[8];
    // rows of N — icp_kernels.cl:993-999
    float r0[4] = { Sxx - Syy - Szz,       Sxy + Syx,         Szx + Sxz,       Syz - Szy };
    float r1[4] = {       Sxy + Syx, - Sxx + Syy - Szz,       Syz + Szy,       Szx - Sxz };
    float r2[4] = {       Szx + Sxz,       Syz + Szy, - Sxx - Syy + Szz,       Sxy - Syx };
    float r3[4] = {       Syz - Szy,       Szx - Sxz,         Sxy - Syx, Sxx + Syy + Szz };
    float Nrow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) Nrow[k] = (i == 0) ? r0[k] : (i == 1) ? r1[k] : (i == 2) ? r2[k] : r3[k];

    float x = 1.f, xn = 0.f;
    int iters = 0;
    // Tk from the unit quaternion in xn (icp_kernels.cl:1050)
    const float *mf = means, *mm = means + 4;
    auto tk_of = [&] (float xq) {
        float qx = pmq_q<0> (xq), qy = pmq_q<1> (xq), qz = pmq_q<2> (xq), qw = pmq_q<3> (xq);
        float c1x = (qy * mm[2] - qz * mm[1]) + qw * mm[0];
        float c1y = (qz * mm[0] - qx * mm[2]) + qw * mm[1];
        float c1z = (qx * mm[1] - qy * mm[0]) + qw * mm[2];
        float ax = 2 * qx, ay = 2 * qy, az = 2 * qz;
        float c2x = ay * c1z - az * c1y;
        float c2y = az * c1x - ax * c1z;
        float c2z = ax * c1y - ay * c1x;
        Tk[0] = qx; Tk[1] = qy; Tk[2] = qz; Tk[3] = qw;
        Tk[4] = mf[0] - sk * (mm[0] + c2x);
        Tk[5] = mf[1] - sk * (mm[1] + c2y);
        Tk[6] = mf[2] - sk * (mm[2] + c2z);
        Tk[7] = sk;
    };
    PM_STAMP (0)
    bool shifted = false;
    if (squared_start) {
        // oracle power_fast: B = N^1024, u = B 1, x = normalize (u), xn = normalize (N u) (two independent chains),
        // loop on squared step lengths, division-free sign test, no extra pass after the loop
        for (;;) {
            float Brow[4] = { Nrow[0], Nrow[1], Nrow[2], Nrow[3] };
            pmq_rescale (Brow);
            for (int s = 0; s < ICP_PM_SQUARINGS; ++s) {
                icp_f4 acc = { 0.f, 0.f, 0.f, 0.f };
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (Brow[0], Brow[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (Brow[1], Brow[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (Brow[2], Brow[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32 (Brow[3], Brow[3], acc, 0, 0, 0);
                Brow[0] = acc[0]; Brow[1] = acc[1]; Brow[2] = acc[2]; Brow[3] = acc[3];
                if (s % 5 == 4) pmq_rescale (Brow);              // max|entry| < 2 after a rescale, < 2^94 five squarings later
            }
            PM_STAMP (1)
            const float u = pmq_matvec (Brow, 1.f);
            const float v = pmq_matvec (Nrow, u);
            xn = pmq_normalize (v);
            ++iters;
            // fast exit on the unnormalised pair (oracle power_fast): |u x v|^2 over the six index pairs against
            // 2^-44 (u.u)(v.v), sign of the eigenvalue from u.v — decided while the normalisation above is still under way.
            // Lane i holds the pairs (i, i+1) and (i, i+2) (indices mod 4; (3,0) and (2,0), (3,1) are the negatives of
            // (0,3), (0,2), (1,3): the same squares); summed in the oracle's order 01, 02, 03, 12, 13, 23.
            {
                const float u1 = icp_dpp<0x39> (u), v1 = icp_dpp<0x39> (v);      // quad_perm [1,2,3,0]: component i + 1
                const float u2 = icp_dpp<0x4E> (u), v2 = icp_dpp<0x4E> (v);      // quad_perm [2,3,0,1]: component i + 2
                const float ta = u * v1 - u1 * v, tb = u * v2 - u2 * v;
                const float a1 = ta * ta, a2 = tb * tb;
                float c2 = 0.f;
                c2 = c2 + pmq_q<0> (a1); c2 = c2 + pmq_q<0> (a2); c2 = c2 + pmq_q<3> (a1);
                c2 = c2 + pmq_q<1> (a1); c2 = c2 + pmq_q<1> (a2); c2 = c2 + pmq_q<2> (a1);
                const float uu = pmq_seq4 (u * u), vv = pmq_seq4 (v * v), uv = pmq_seq4 (u * v);
                // (scalar branches: every quad holds the same values; as per-lane compares the compiler builds divergent control flow around the
                // one path every iteration of every ordinary scene takes)
                if (__builtin_expect (!__ballot (c2 > 0x1p-44f * (uu * vv)), 1)) {
                    if (__ballot (uv < 0.f)) {
                        const float lambda = uv / uu;
#pragma unroll
                        for (int k = 0; k < 4; ++k) Nrow[k] = (i == (uint32_t) k) ? Nrow[k] - lambda : Nrow[k];
                        continue;
                    }
                    tk_of (xn);
                    break;
                }
                // not converged after 1024 steps: +-lambda pairs of a planar scene (oracle power_fast: the norm shift, once per solve)
                if (!shifted) {
                    shifted = true;
                    float sigma = ((__builtin_fabsf (Nrow[0]) + __builtin_fabsf (Nrow[1])) + __builtin_fabsf (Nrow[2])) + __builtin_fabsf (Nrow[3]);
                    sigma = fmaxf (sigma, icp_dpp<0xB1> (sigma));           // quad_perm [1,0,3,2]
                    sigma = fmaxf (sigma, icp_dpp<0x4E> (sigma));           // quad_perm [2,3,0,1]: the largest absolute row sum (max is exact)
                    if (sigma > 0.f && sigma < __builtin_inff ()) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) Nrow[k] = (i == (uint32_t) k) ? Nrow[k] + sigma : Nrow[k];
                        continue;
                    }
                }
            }
            x = pmq_normalize (u);
            PM_STAMP (2)
            float e2_prev = __builtin_inff (), d = x - xn, e2 = pmq_seq4 (d * d);
            while (e2 > 0x1p-44f && e2 < e2_prev && iters < 1000) {
                x = xn;
                xn = pmq_normalize (pmq_matvec (Nrow, x));
                ++iters;
                e2_prev = e2; d = x - xn; e2 = pmq_seq4 (d * d);
            }
            PM_STAMP (3)
            tk_of (xn);                              // (before the sign test, not after it: the two chains are independent, and a restart is rare)
            const float lam_num = pmq_lane (pmq_matvec (Nrow, xn), 0), den = pmq_lane (xn, 0);
            if ((lam_num < 0.f && den > 0.f) || (lam_num > 0.f && den < 0.f)) {
                const float lambda = lam_num / den;
#pragma unroll
                for (int k = 0; k < 4; ++k) Nrow[k] = (i == (uint32_t) k) ? Nrow[k] - lambda : Nrow[k];
            } else break;
        }
        PM_STAMP (4)
        PM_STAMP (5)
    } else {
        // The reference's loop (icp_kernels.cl:1012-1022), the same operations on the same values, arranged for ONE wave that the whole grid
        // waits for: a trip is a chain of ~55 dependent instructions (N x, the sum of squares, an IEEE square root, an IEEE division; then
        // the step length: a difference, a sum of squares, another IEEE square root, a compare), and only the first half is carried from
        // trip to trip.  So the NEXT trip's vector is computed while this trip's step length is still under way (it depends on x_new only;
        // if the loop stops here it is exactly the pass the reference makes behind its loop, :1039-1041, so it is never wasted), and the
        // stop test is a scalar branch (every quad holds the same values; left as a per-lane compare the compiler builds a divergent
        // loop: ~20 mask instructions per trip).  Round 4: 0.57 us per trip (k_finalize<1> 32.6 us per dispatch).
        // What it does NOT buy is the factor the loop would need to matter less: a trip is ~75 vector instructions that no re-arrangement
        // removes (N x: 8, two sums of squares: 12, two IEEE square roots: 34, one IEEE division: 12, the rest moves and compares) plus ~20
        // hazard no-ops around the quad broadcasts, and ONE wave issues a vector instruction every 4 cycles at best, a dependent one every
        // ~6.6, a no-op every 4 (MI355X_MICROARCH.md, cycle constants): ~520 cycles = 0.22 us per trip, 118 trips per iteration on the
        // benchmark pair.  Measured (tools/diag/power_time.py, fused reductions + literal loop): 34.9 -> 33.2 us per iteration with the
        // look-ahead and the scalar branch; replacing the step length's square root by a comparison of the sums (equal sums have equal
        // roots, sums more than 8 ulps apart have different ones, both roots only in between) was bit-identical and no faster (34.2 us: the
        // band test costs what the root's 17 instructions cost less the hazards) and is not kept.  The squared start (the default) is the
        // way out of this loop, not a faster trip.
        float xnn = 0.f;
        for (;;) {
            float error = __builtin_inff ();
            xn = pmq_normalize (pmq_matvec (Nrow, x));                // trip 1
            ++iters;
            for (uint32_t it = 1;; ++it) {
                xnn = pmq_normalize (pmq_matvec (Nrow, xn));          // trip it + 1, ahead of the test (or the pass behind the loop)
                const float d = x - xn;
                const float error_new = sqrtf (pmq_seq4 (d * d));
                if (__ballot (error_new == error) || it >= 1000u) break;      // :1019 (stop when the step length repeats) / :1012 (1000 trips)
                error = error_new; x = xn; xn = xnn;
                ++iters;
            }
            float lam_num = pmq_lane (pmq_matvec (Nrow, xn), 0);
            float lambda = lam_num / pmq_lane (xn, 0);                // :1024
            if (lambda < 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Nrow[k] = (i == (uint32_t) k) ? Nrow[k] - lambda : Nrow[k];
                x = 1.f;
            } else break;
        }
        xn = xnn;                                                     // :1039-1041: x = x_new; x_new = normalize (N x)
        tk_of (xn);
    }
    PM_STAMP (6)
    return iters;
}

// ------------------------------------------------------------------------------------------
// a10  host composition of the reference moved on-device (src/ICP/algorithms.cpp:4683-4695);
//      Eigen formulas restated (oracle: orc_quat_to_rot / orc_rot_to_quat / compose).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void icp_quat_to_rot (const float *q, float *R)
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

// Eigen's matrix -> quaternion (oracle orc_rot_to_quat).  The largest-diagonal branch is written out per axis
// (compile-time indices): a runtime index into m[] would put the matrix in scratch memory.
template <int I> __device__ __forceinline__ void icp_rot_to_quat_axis (const float *m, float *q)
{
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    float t = sqrtf (((m[I * 4] - m[J * 4]) - m[K * 4]) + 1.f);
    q[I] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (m[K * 3 + J] - m[J * 3 + K]) * t;
    q[J] = (m[J * 3 + I] + m[I * 3 + J]) * t;
    q[K] = (m[K * 3 + I] + m[I * 3 + K]) * t;
}
__device__ inline void icp_rot_to_quat (const float *m, float *q)
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
        const bool i1 = m[4] > m[0];
        const bool i2 = m[8] > (i1 ? m[4] : m[0]);
        if (i2) icp_rot_to_quat_axis<2> (m, q);
        else if (i1) icp_rot_to_quat_axis<1> (m, q);
        else icp_rot_to_quat_axis<0> (m, q);
    }
}

// a12  EIGEN branch (src/ICP/algorithms.cpp:3867-3909) — one-sided Jacobi SVD, oracle orc_svd_rotation.
// One rotation of the sweep on the column pair (P, Q), compile-time indices (no scratch memory).
template <int P, int Q> __device__ __forceinline__ void icp_svd_rotate_pair (float *A, float *V, float &off)
{
    float alpha = 0, beta = 0, gamma = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        alpha = alpha + A[i * 3 + P] * A[i * 3 + P];
        beta  = beta  + A[i * 3 + Q] * A[i * 3 + Q];
        gamma = gamma + A[i * 3 + P] * A[i * 3 + Q];
    }
    if (gamma == 0.f) return;
    off = fmaxf (off, fabsf (gamma) / sqrtf (alpha * beta));
    float zeta = (beta - alpha) / (2.f * gamma);
    float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf (zeta) + sqrtf (1.f + zeta * zeta));
    float cs = 1.f / sqrtf (1.f + t * t), sn = cs * t;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float ap = A[i * 3 + P], aq = A[i * 3 + Q];
        A[i * 3 + P] = cs * ap - sn * aq; A[i * 3 + Q] = sn * ap + cs * aq;
        float vp = V[i * 3 + P], vq = V[i * 3 + Q];
        V[i * 3 + P] = cs * vp - sn * vq; V[i * 3 + Q] = sn * vp + cs * vq;
    }
}
__device__ inline void icp_svd_rotation (const float *S11, const float *means, float *Rk, float *Tk)
{
    float A[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = S11[i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        float off = 0.f;
        icp_svd_rotate_pair<0, 1> (A, V, off);
        icp_svd_rotate_pair<0, 2> (A, V, off);
        icp_svd_rotate_pair<1, 2> (A, V, off);
        if (off < 1e-7f) break;
    }
    float U[9], sig[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        sig[j] = sqrtf ((A[j] * A[j] + A[3 + j] * A[3 + j]) + A[6 + j] * A[6 + j]);
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = sig[j] > 0.f ? A[i * 3 + j] / sig[j] : 0.f;
    }
    int smin = 0; float sminv = sig[0];                                // first smallest singular value
    if (sig[1] < sminv) { smin = 1; sminv = sig[1]; }
    if (sig[2] < sminv) { smin = 2; sminv = sig[2]; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Rk[i * 3 + j] = (V[i * 3] * U[j * 3] + V[i * 3 + 1] * U[j * 3 + 1]) + V[i * 3 + 2] * U[j * 3 + 2];
    float det = Rk[0] * (Rk[4] * Rk[8] - Rk[5] * Rk[7]) - Rk[1] * (Rk[3] * Rk[8] - Rk[5] * Rk[6])
              + Rk[2] * (Rk[3] * Rk[7] - Rk[4] * Rk[6]);
    if (det < 0.f) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc = acc + V[i * 3 + k] * (k == smin ? det : 1.f) * U[j * 3 + k];
                Rk[i * 3 + j] = acc;
            }
    }
    float qk[4]; icp_rot_to_quat (Rk, qk);
    float sk = sqrtf (S11[9] / S11[10]);
    const float *mf = means, *mm = means + 4;
    Tk[0] = qk[0]; Tk[1] = qk[1]; Tk[2] = qk[2]; Tk[3] = qk[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        Tk[4 + i] = mf[i] - ((sk * Rk[i * 3]) * mm[0] + (sk * Rk[i * 3 + 1]) * mm[1] + (sk * Rk[i * 3 + 2]) * mm[2]);
    Tk[7] = sk;
}

// compose: R = Rk R ; q = quat(R) ; t = sk Rk t + tk ; s = sk s   (algorithms.cpp:4688-4691)
__device__ inline void icp_compose (icp_reg_state *st, const float *Tk, const float *Rk_in, int have_rk)
{
    float Rk[9];
    if (have_rk) { for (int i = 0; i < 9; ++i) Rk[i] = Rk_in[i]; }
    else icp_quat_to_rot (Tk, Rk);
    float sk = Tk[7];
    float Rn[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Rn[i * 3 + j] = (Rk[i * 3] * st->R[j] + Rk[i * 3 + 1] * st->R[3 + j]) + Rk[i * 3 + 2] * st->R[6 + j];
    float q[4]; icp_rot_to_quat (Rn, q);
    float t0 = st->T[4], t1 = st->T[5], t2 = st->T[6];
    float tn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float r0 = sk * Rk[i * 3], r1 = sk * Rk[i * 3 + 1], r2 = sk * Rk[i * 3 + 2];
        tn[i] = ((r0 * t0 + r1 * t1) + r2 * t2) + Tk[4 + i];
    }
    float s = sk * st->T[7];
    for (int i = 0; i < 9; ++i) { st->R[i] = Rn[i]; st->Rk[i] = Rk[i]; }
    for (int i = 0; i < 8; ++i) st->Tk[i] = Tk[i];
    st->T[0] = q[0]; st->T[1] = q[1]; st->T[2] = q[2]; st->T[3] = q[3];
    st->T[4] = tn[0]; st->T[5] = tn[1]; st->T[6] = tn[2]; st->T[7] = s;
}

// same composition as icp_compose, from explicit inputs to explicit outputs (chained fused mode)
__device__ inline void icp_compose_pure (const float *Tprev, const float *Rprev, const float *Tk, const float *Rk_in, int have_rk,
                                         float *Tn, float *Rn, float *Rk)
{
    if (have_rk) { for (int i = 0; i < 9; ++i) Rk[i] = Rk_in[i]; }
    else icp_quat_to_rot (Tk, Rk);
    float sk = Tk[7];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Rn[i * 3 + j] = (Rk[i * 3] * Rprev[j] + Rk[i * 3 + 1] * Rprev[3 + j]) + Rk[i * 3 + 2] * Rprev[6 + j];
    float q[4]; icp_rot_to_quat (Rn, q);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float r0 = sk * Rk[i * 3], r1 = sk * Rk[i * 3 + 1], r2 = sk * Rk[i * 3 + 2];
        Tn[4 + i] = ((r0 * Tprev[4] + r1 * Tprev[5]) + r2 * Tprev[6]) + Tk[4 + i];
    }
    Tn[0] = q[0]; Tn[1] = q[1]; Tn[2] = q[2]; Tn[3] = q[3];
    Tn[7] = sk * Tprev[7];
}

// ICP::check — src/ICP/algorithms.cpp:4824-4834 (predicate form: oracle check_converged)
__device__ __forceinline__ int icp_check_converged (const float *Tk, double tan_half_thr, double trans_thr)
{
    float vn = sqrtf ((Tk[0] * Tk[0] + Tk[1] * Tk[1]) + Tk[2] * Tk[2]);
    float tn = sqrtf ((Tk[4] * Tk[4] + Tk[5] * Tk[5]) + Tk[6] * Tk[6]);
    int ang = (Tk[3] > 0.f) && ((double) vn < (double) Tk[3] * tan_half_thr);
    int tra = (double) tn < trans_thr;
    return ang && tra;
}
