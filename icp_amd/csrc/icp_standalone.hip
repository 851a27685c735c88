// icp_standalone.hip — the reference's per-kernel wrapper classes as stand-alone operations (host in, host out, one call each):
// ICPLMs, ICPReps, ICPWeights, ICPMean<REGULAR | WEIGHTED>, ICPDevs, ICPS<REGULAR | WEIGHTED> (include/ICP/algorithms.hpp:397-1183,
// src/ICP/algorithms.cpp:621-2547, kernels/icp_kernels.cl:63-743, kernels/reduce_kernels.cl:230-264).  The iteration itself runs
// none of this — it fuses these steps into k_search / k_means / k_sij / k_finalize (icp_kernels.hip) —; the classes exist because a
// user of the reference can call them one by one, and its tests do (tests/testsICP.cpp:66-790).  Simple kernels, the canonical
// reduction trees of DESIGN.md §3 (one 16-lane DPP row = one 128-position work-group tree): bit-identical to the oracle's twins
// (orc_get_lms, orc_get_reps, orc_weights, orc_mean, orc_mean_weighted, orc_devs, orc_sij).
#include "../../include/icp_amd.h"
#include "icp_kernels.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_sa_error;
int sa_fail (int code, const std::string &m) { g_sa_error = m; return code; }

// out[row][g] = canonical 128-position tree over in[row][g * 128 .. g * 128 + 128) (zeros past n): one 16-lane row per group
__global__ __launch_bounds__ (64) void k_sa_tree128 (const float *in, uint32_t n, uint32_t stride_in, uint32_t ngroups, uint32_t stride_out, float *out)
{
    const uint32_t lane = threadIdx.x, l = lane & 15u, g = blockIdx.x * 4u + (lane >> 4), row = blockIdx.y;
    const float *src = in + (size_t) row * stride_in;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint32_t i = g * 128u + l + 16u * k; a[k] = (g < ngroups && i < n) ? src[i] : 0.f; }
    const float v = row_tree8 (a);
    if (l == 0 && g < ngroups) out[(size_t) row * stride_out + g] = v;
}

// reduce_sum_f pass (kernels/reduce_kernels.cl:230-264): position p of work-group g = ((c0 + c1) + c2) + c3 of 4 consecutive columns
__global__ __launch_bounds__ (64) void k_sa_sum_level (const float *in, uint32_t cols, uint32_t wgp, float *out)
{
    const uint32_t lane = threadIdx.x, l = lane & 15u, g = blockIdx.x * 4u + (lane >> 4), row = blockIdx.y;
    const float *src = in + (size_t) row * cols;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t c = g * 512u + 4u * (l + 16u * k);
        float4 v = make_float4 (0.f, 0.f, 0.f, 0.f);
        if (g < wgp && c < cols) v = *reinterpret_cast<const float4 *> (src + c);
        a[k] = ((v.x + v.y) + v.z) + v.w;
    }
    const float r = row_tree8 (a);
    if (l == 0 && g < wgp) out[(size_t) row * wgp + g] = r;
}

// ICPWeights, first kernel (icpComputeReduceWeights_WG, kernels/icp_kernels.cl:213-254): w = 100 / (100 + dist); the plane is
// zero past n (one flag guards a pair: n is even)
__global__ void k_sa_weights (const icp_dist_id *D, uint32_t n, uint32_t npad, float *W, float *plane)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    float w = 0.f;
    if ((i & ~1u) < n) { w = 100.f / (100.f + D[i].dist); W[i] = w; }
    plane[i] = w;
}

// reduce_sum_fd (kernels/icp_kernels.cl:295-329): float4 partials -> double, ((x + y) + z) + w, double tree over 128 positions;
// chunks of 512 partials in index order (oracle orc_weights).  One wave; lane 0 of row 0 holds the result.
__global__ __launch_bounds__ (64) void k_sa_sum_fd (const float *part, uint32_t wgp, double *sum_w)
{
    const uint32_t l = threadIdx.x & 15u;
    if (wgp == 1) { if (threadIdx.x == 0) *sum_w = (double) part[0]; return; }
    double total = 0.0;
    for (uint32_t c0 = 0; c0 < wgp; c0 += 512u) {
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t i4 = c0 + 4u * (l + 16u * k);
            a[k] = 0.0;
            if (i4 < wgp) a[k] = (((double) part[i4] + (double) part[i4 + 1]) + (double) part[i4 + 2]) + (double) part[i4 + 3];
        }
        const double cs = row_tree8_d (a);
        total = (c0 == 0) ? cs : total + cs;
    }
    if (threadIdx.x == 0) *sum_w = total;
}

// ICPMean, first kernel (icpMean :371-411 / icpMean_Weighted :455-495): planes[set][k][i] = (float) (W[i] / sum_w) * x_k, or x_k / n
__global__ void k_sa_mean_scale (const float *F, const float *M, const float *W, const double *sum_w, uint32_t n, uint32_t npad, int weighted, float *planes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    const bool ok = (i & ~1u) < n;
    float kf = 0.f;
    if (ok && weighted) kf = (float) ((double) W[i] / *sum_w);
    const float nf = (float) n;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float *in = s ? M : F;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = 0.f;
            if (ok) v = weighted ? kf * in[(size_t) i * 8 + k] : in[(size_t) i * 8 + k] / nf;
            planes[(size_t) (s * 3 + k) * npad + i] = v;
        }
    }
}

// ICPDevs (icpSubtractMean :588-602): float4 subtract, xyz - mean, .w = 1 - 0
__global__ void k_sa_devs (const float4 *F, const float4 *M, const float *mean8, uint32_t n, float4 *DF, float4 *DM)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = F[2 * (size_t) i], m = M[2 * (size_t) i];
    DF[i] = make_float4 (f.x - mean8[0], f.y - mean8[1], f.z - mean8[2], f.w - mean8[3]);
    DM[i] = make_float4 (m.x - mean8[4], m.y - mean8[5], m.z - mean8[6], m.w - mean8[7]);
}

// ICPS, first kernel (icpSijProducts :633-671 / _Weighted :703-743): column g accumulates the points g, g + G, g + 2G, g + 3G
__global__ void k_sa_sij (const float4 *DM, const float4 *DF, const float *W, uint32_t m, uint32_t G, uint32_t Gp, float c, int weighted, float *Sij)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float A[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) A[k] = 0.f;
    for (uint32_t pi = g; pi < m; pi += G) {
        const float4 dm = DM[pi], df = DF[pi];
        const float Mp[3] = { c * dm.x, c * dm.y, c * dm.z }, Fp[3] = { c * df.x, c * df.y, c * df.z };
        const float ff = (Fp[0] * Fp[0] + Fp[1] * Fp[1]) + Fp[2] * Fp[2];
        const float mm = (Mp[0] * Mp[0] + Mp[1] * Mp[1]) + Mp[2] * Mp[2];
        if (weighted) {
            const float w = W[pi];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) A[a * 3 + b] = A[a * 3 + b] + w * (Mp[a] * Fp[b]);
            A[9] = A[9] + w * ff; A[10] = A[10] + w * mm;
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) A[a * 3 + b] = A[a * 3 + b] + Mp[a] * Fp[b];
            A[9] = A[9] + ff; A[10] = A[10] + mm;
        }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) Sij[(size_t) k * Gp + g] = A[k];        // rows of Gp = G padded with zeros to a multiple of 4 (reduce_sum_f reads float4)
}

// ICPLMs / ICPReps
__global__ void k_sa_get_lms (const float4 *cloud, float4 *lms)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 16384u * 2u) return;
    const uint32_t lm = t >> 1, half = t & 1u, gX = lm & 127u, gY = lm >> 7;
    lms[t] = cloud[((size_t) (48u + gY * 3u + 1u) * 640u + 64u + 4u * gX + 1u) * 2u + half];      // kernels/icp_kernels.cl:63-76
}
__global__ void k_sa_get_reps (const float4 *F, float4 *R, icp_params p)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.nr) return;
    const uint32_t src = rep_src_index (p, r);
    R[2 * r] = F[2 * (size_t) src]; R[2 * r + 1] = F[2 * (size_t) src + 1];
}

struct dev_scope {                                   // device selection + a bag of allocations freed on the way out
    std::vector<void *> allocs;
    bool ok = true; std::string err;
    ~dev_scope () { for (void *q : allocs) (void) hipFree (q); }
    template <typename T> T *alloc (size_t count)
    {
        void *q = nullptr;
        if (!ok) return nullptr;
        hipError_t e = hipMalloc (&q, (count ? count : 1) * sizeof (T));
        if (e != hipSuccess) { ok = false; err = std::string ("hipMalloc: ") + hipGetErrorString (e); return nullptr; }
        allocs.push_back (q);
        return static_cast<T *> (q);
    }
    void chk (hipError_t e, const char *what) { if (ok && e != hipSuccess) { ok = false; err = std::string (what) + ": " + hipGetErrorString (e); } }
};

int sa_device (int device)
{
    int count = 0;
    if (hipGetDeviceCount (&count) != hipSuccess || count <= 0) return sa_fail (ICP_ENODEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= count) return sa_fail (ICP_EINVAL, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties (&prop, device) != hipSuccess || std::strncmp (prop.gcnArchName, "gfx950", 6) != 0)
        return sa_fail (ICP_ENODEVICE, "kernels are built for gfx950 only");
    if (hipSetDevice (device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    return ICP_OK;
}

uint32_t pad4 (uint32_t x) { return (x != 1u && (x % 4u)) ? x + 4u - x % 4u : x; }

// row-wise reduce_sum_f of a rows x cols device array (cols % 4 == 0) until one value per row; returns the device pointer of the result
const float *sa_reduce_rows (dev_scope &d, const float *in, uint32_t cols, uint32_t rows)
{
    const float *cur = in; uint32_t ccols = cols;
    for (;;) {
        const uint32_t wgp = pad4 ((ccols + 511u) / 512u);
        float *nxt = d.alloc<float> ((size_t) rows * wgp);
        if (!d.ok) return nullptr;
        d.chk (hipMemset (nxt, 0, (size_t) rows * wgp * sizeof (float)), "hipMemset");
        hipLaunchKernelGGL (k_sa_sum_level, dim3 ((wgp + 3u) / 4u, rows), dim3 (64), 0, 0, cur, ccols, wgp, nxt);
        cur = nxt; ccols = wgp;
        if (wgp == 1u) return cur;
    }
}

}  // namespace

extern "C" {

const char *icp_kernel_last_error (void) { return g_sa_error.c_str (); }

int icp_kernel_lms (int device, const void *cloud, void *lms)
{
    if (!cloud || !lms) return sa_fail (ICP_EINVAL, "null pointer");
    int rc = sa_device (device); if (rc) return rc;
    dev_scope d;
    float4 *dc = d.alloc<float4> ((size_t) 640 * 480 * 2), *dl = d.alloc<float4> (16384 * 2);
    if (d.ok) d.chk (hipMemcpy (dc, cloud, (size_t) 640 * 480 * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) { hipLaunchKernelGGL (k_sa_get_lms, dim3 (128), dim3 (256), 0, 0, dc, dl); d.chk (hipGetLastError (), "k_sa_get_lms"); }
    if (d.ok) d.chk (hipMemcpy (lms, dl, (size_t) 16384 * 32, hipMemcpyDeviceToHost), "hipMemcpy");
    return d.ok ? ICP_OK : sa_fail (ICP_EHIP, d.err);
}

int icp_kernel_reps (int device, const void *F, uint32_t m, uint32_t nr, void *R)
{
    if (!F || !R) return sa_fail (ICP_EINVAL, "null pointer");
    // the representative grid — src/ICP/algorithms.cpp:842-854 generalised to a sqrt (m) x sqrt (m) landmark grid
    if (m == 0 || nr == 0 || nr > m || (nr & (nr - 1))) return sa_fail (ICP_EINVAL, "nr must be a power of two, at most m");
    const uint32_t side = (uint32_t) std::floor (std::sqrt ((double) m) + 0.5);
    if ((uint64_t) side * side != m) return sa_fail (ICP_EINVAL, "m must be a square number (the landmark grid)");
    uint32_t pw = 0; while ((1u << (pw + 1)) <= nr) ++pw;
    icp_params p {};
    p.m = m; p.nr = nr; p.side = side; p.nrx = 1u << (pw - pw / 2); p.nry = 1u << (pw / 2);
    if (side % p.nrx || side % p.nry) return sa_fail (ICP_EINVAL, "the representative grid must tile the landmark grid");
    int rc = sa_device (device); if (rc) return rc;
    dev_scope d;
    float4 *dF = d.alloc<float4> ((size_t) m * 2), *dR = d.alloc<float4> ((size_t) nr * 2);
    if (d.ok) d.chk (hipMemcpy (dF, F, (size_t) m * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) { hipLaunchKernelGGL (k_sa_get_reps, dim3 ((nr + 255u) / 256u), dim3 (256), 0, 0, dF, dR, p); d.chk (hipGetLastError (), "k_sa_get_reps"); }
    if (d.ok) d.chk (hipMemcpy (R, dR, (size_t) nr * 32, hipMemcpyDeviceToHost), "hipMemcpy");
    return d.ok ? ICP_OK : sa_fail (ICP_EHIP, d.err);
}

int icp_kernel_weights (int device, const void *nn_id, uint32_t n, float *W, double *sum_w)
{
    if (!nn_id || !W || !sum_w) return sa_fail (ICP_EINVAL, "null pointer");
    if (n == 0 || (n % 2)) return sa_fail (ICP_EINVAL, "The number of points in the array must be a (positive) multiple of 2");      // src/ICP/algorithms.cpp:1050
    int rc = sa_device (device); if (rc) return rc;
    const uint32_t wg = (n + 127u) / 128u, wgp = pad4 (wg), npad = wgp * 128u;
    dev_scope d;
    icp_dist_id *dD = d.alloc<icp_dist_id> (n); float *dW = d.alloc<float> (n), *plane = d.alloc<float> (npad), *part = d.alloc<float> (wgp);
    double *dsw = d.alloc<double> (1);
    if (d.ok) d.chk (hipMemcpy (dD, nn_id, (size_t) n * 8, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) {
        hipLaunchKernelGGL (k_sa_weights, dim3 ((npad + 255u) / 256u), dim3 (256), 0, 0, dD, n, npad, dW, plane);
        hipLaunchKernelGGL (k_sa_tree128, dim3 ((wgp + 3u) / 4u, 1), dim3 (64), 0, 0, plane, npad, npad, wgp, wgp, part);
        hipLaunchKernelGGL (k_sa_sum_fd, dim3 (1), dim3 (64), 0, 0, part, wgp, dsw);
        d.chk (hipGetLastError (), "weights kernels");
    }
    if (d.ok) d.chk (hipMemcpy (W, dW, (size_t) n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (sum_w, dsw, 8, hipMemcpyDeviceToHost), "hipMemcpy");
    return d.ok ? ICP_OK : sa_fail (ICP_EHIP, d.err);
}

int icp_kernel_mean (int device, int weighted, const void *F, const void *M, const float *W, double sum_w, uint32_t n, float *mean8)
{
    if (!F || !M || !mean8 || (weighted && !W)) return sa_fail (ICP_EINVAL, "null pointer");
    if (n == 0 || (n % 2)) return sa_fail (ICP_EINVAL, "The number of points in the array must be a (positive) multiple of 2");      // :1306, :1573
    int rc = sa_device (device); if (rc) return rc;
    const uint32_t wg = (n + 127u) / 128u, npad = wg * 128u;
    dev_scope d;
    float *dF = d.alloc<float> ((size_t) n * 8), *dM = d.alloc<float> ((size_t) n * 8), *dW = d.alloc<float> (n), *planes = d.alloc<float> ((size_t) 6 * npad);
    double *dsw = d.alloc<double> (1);
    if (d.ok) d.chk (hipMemcpy (dF, F, (size_t) n * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (dM, M, (size_t) n * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok && weighted) d.chk (hipMemcpy (dW, W, (size_t) n * 4, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (dsw, &sum_w, 8, hipMemcpyHostToDevice), "hipMemcpy");
    const float *cur = planes; uint32_t cnt = npad, stride = npad;
    if (d.ok) {
        hipLaunchKernelGGL (k_sa_mean_scale, dim3 ((npad + 255u) / 256u), dim3 (256), 0, 0, dF, dM, dW, dsw, n, npad, weighted, planes);
        // block means (one per 128 pairs), then icpGMean (:530-566) until one vector per set remains
        for (;;) {
            const uint32_t ng = (cnt + 127u) / 128u;
            float *nxt = d.alloc<float> ((size_t) 6 * ng);
            if (!d.ok) break;
            hipLaunchKernelGGL (k_sa_tree128, dim3 ((ng + 3u) / 4u, 6), dim3 (64), 0, 0, cur, cnt, stride, ng, ng, nxt);
            cur = nxt; cnt = ng; stride = ng;
            if (ng == 1u) break;
        }
        d.chk (hipGetLastError (), "mean kernels");
    }
    float six[6];
    if (d.ok) d.chk (hipMemcpy (six, cur, sizeof six, hipMemcpyDeviceToHost), "hipMemcpy");
    if (!d.ok) return sa_fail (ICP_EHIP, d.err);
    for (int s = 0; s < 2; ++s) { mean8[4 * s] = six[3 * s]; mean8[4 * s + 1] = six[3 * s + 1]; mean8[4 * s + 2] = six[3 * s + 2]; mean8[4 * s + 3] = 0.f; }
    return ICP_OK;
}

int icp_kernel_devs (int device, const void *F, const void *M, const float *mean8, uint32_t n, float *DF, float *DM)
{
    if (!F || !M || !mean8 || !DF || !DM) return sa_fail (ICP_EINVAL, "null pointer");
    if (n == 0) return sa_fail (ICP_EINVAL, "The array cannot have zero points");
    int rc = sa_device (device); if (rc) return rc;
    dev_scope d;
    float4 *dF = d.alloc<float4> ((size_t) n * 2), *dM = d.alloc<float4> ((size_t) n * 2), *dDF = d.alloc<float4> (n), *dDM = d.alloc<float4> (n);
    float *dmean = d.alloc<float> (8);
    if (d.ok) d.chk (hipMemcpy (dF, F, (size_t) n * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (dM, M, (size_t) n * 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (dmean, mean8, 32, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) { hipLaunchKernelGGL (k_sa_devs, dim3 ((n + 255u) / 256u), dim3 (256), 0, 0, dF, dM, dmean, n, dDF, dDM); d.chk (hipGetLastError (), "k_sa_devs"); }
    if (d.ok) d.chk (hipMemcpy (DF, dDF, (size_t) n * 16, hipMemcpyDeviceToHost), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (DM, dDM, (size_t) n * 16, hipMemcpyDeviceToHost), "hipMemcpy");
    return d.ok ? ICP_OK : sa_fail (ICP_EHIP, d.err);
}

int icp_kernel_s (int device, int weighted, const float *DM, const float *DF, const float *W, uint32_t m, float c, float *S11)
{
    if (!DM || !DF || !S11 || (weighted && !W)) return sa_fail (ICP_EINVAL, "null pointer");
    if (m == 0) return sa_fail (ICP_EINVAL, "The array cannot have zero points");
    int rc = sa_device (device); if (rc) return rc;
    uint32_t n4 = m; if (n4 % 4u) n4 += 4u - n4 % 4u;
    const uint32_t G = n4 / 4u, Gp = (G + 3u) & ~3u;                        // src/ICP/algorithms.cpp:2344-2346; columns padded to float4
    dev_scope d;
    float4 *dDM = d.alloc<float4> (m), *dDF = d.alloc<float4> (m); float *dW = d.alloc<float> (m), *Sij = d.alloc<float> ((size_t) 11 * Gp);
    if (d.ok) d.chk (hipMemcpy (dDM, DM, (size_t) m * 16, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemcpy (dDF, DF, (size_t) m * 16, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok && weighted) d.chk (hipMemcpy (dW, W, (size_t) m * 4, hipMemcpyHostToDevice), "hipMemcpy");
    if (d.ok) d.chk (hipMemset (Sij, 0, (size_t) 11 * Gp * sizeof (float)), "hipMemset");
    const float *res = nullptr;
    if (d.ok) {
        hipLaunchKernelGGL (k_sa_sij, dim3 ((G + 255u) / 256u), dim3 (256), 0, 0, dDM, dDF, dW, m, G, Gp, c, weighted, Sij);
        d.chk (hipGetLastError (), "k_sa_sij");
    }
    if (d.ok) res = sa_reduce_rows (d, Sij, Gp, 11);
    if (d.ok) d.chk (hipGetLastError (), "reduce_sum_f");
    if (d.ok) d.chk (hipMemcpy (S11, res, 44, hipMemcpyDeviceToHost), "hipMemcpy");
    return d.ok ? ICP_OK : sa_fail (ICP_EHIP, d.err);
}

}  // extern "C"
