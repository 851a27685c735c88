// icp_standalone.hip — the reference's per-kernel wrapper classes as RESIDENT objects (icp_ko_*: device buffers live with the object,
// get (Memory) = a device pointer another object can adopt before its own buffers exist, run () = kernels only — the way ICPStep::init wires
// the reference's objects by shared cl::Buffers, src/ICP/algorithms.cpp:4499-4581) and, on top of them, as one-call operations (host in, host out):
// ICPLMs, ICPReps, ICPWeights, ICPMean<REGULAR | WEIGHTED>, ICPDevs, ICPS<REGULAR | WEIGHTED> (include/ICP/algorithms.hpp:397-1183,
// src/ICP/algorithms.cpp:621-2547, kernels/icp_kernels.cl:63-743, kernels/reduce_kernels.cl:230-264).  The iteration itself runs
// none of this — it fuses these steps into k_search / k_means / k_sij / k_finalize (icp_kernels.hip) —; the classes exist because a
// user of the reference can call them one by one, and its tests do (tests/testsICP.cpp:66-790).  Simple kernels, the canonical
// reduction trees of DESIGN.md §3 (one 16-lane DPP row = one 128-position work-group tree): bit-identical to the oracle's twins
// (orc_get_lms, orc_get_reps, orc_weights, orc_mean, orc_mean_weighted, orc_devs, orc_sij).
#include "../../include/icp_amd.h"
#include "icp_cguard.h"
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

// the last level of the mean trees leaves [set][k] (six floats): -> [mean_F, 0 | mean_M, 0] (the layout ICPDevs and the power method read)
__global__ void k_sa_pack_mean8 (const float *six, float *mean8)
{
    const uint32_t t = threadIdx.x;
    if (t < 8u) mean8[t] = (t & 3u) == 3u ? 0.f : six[(t >> 2) * 3u + (t & 3u)];
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

}  // namespace

// A resident kernel object: `slots` are its Memory objects on the device (inputs and outputs; a slot another object's buffer was adopted
// for is not owned), `scratch` the intermediate levels of its trees, sized at creation; run () enqueues kernels on the device's null
// stream — one in-order queue for all kernel objects of a device, like the reference's command queue — and nothing else.
struct icp_ko {
    int device = 0, kind = 0;
    uint32_t n = 0, aux = 0; float c = 1e-6f;
    icp_params p {};                                 // REPS: the grids
    static constexpr int MAXS = 5;
    void *slot[MAXS] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t bytes[MAXS] = { 0, 0, 0, 0, 0 };
    bool owned[MAXS] = { false, false, false, false, false };
    int nslots = 0;
    std::vector<void *> scratch;                     // owned
    std::vector<uint32_t> level_cols;                // reduce levels (S) / group counts (mean)
};

namespace {

int ko_alloc (icp_ko *k, void **out, size_t bytes, bool zero)
{
    void *q = nullptr;
    hipError_t e = hipMalloc (&q, bytes ? bytes : 1);
    if (e != hipSuccess) return sa_fail (ICP_ENOMEM, std::string ("hipMalloc: ") + hipGetErrorString (e));
    if (zero && (e = hipMemset (q, 0, bytes ? bytes : 1)) != hipSuccess) { (void) hipFree (q); return sa_fail (ICP_EHIP, std::string ("hipMemset: ") + hipGetErrorString (e)); }
    *out = q;
    (void) k;
    return ICP_OK;
}

// slots of every kind: (bytes, is it an input)
int ko_layout (icp_ko *k)
{
    const size_t n = k->n;
    switch (k->kind) {
        case ICP_KO_LMS:     k->nslots = 2; k->bytes[0] = (size_t) 640 * 480 * 32; k->bytes[1] = (size_t) 16384 * 32; break;
        case ICP_KO_REPS:    k->nslots = 2; k->bytes[0] = n * 32; k->bytes[1] = (size_t) k->aux * 32; break;
        case ICP_KO_WEIGHTS: k->nslots = 3; k->bytes[0] = n * 8; k->bytes[1] = n * 4; k->bytes[2] = 8; break;
        case ICP_KO_MEAN: case ICP_KO_MEAN_WEIGHTED:
                             k->nslots = 5; k->bytes[0] = n * 32; k->bytes[1] = n * 32; k->bytes[2] = n * 4; k->bytes[3] = 8; k->bytes[4] = 32; break;
        case ICP_KO_DEVS:    k->nslots = 5; k->bytes[0] = n * 32; k->bytes[1] = n * 32; k->bytes[2] = 32; k->bytes[3] = n * 16; k->bytes[4] = n * 16; break;
        case ICP_KO_S: case ICP_KO_S_WEIGHTED:
                             k->nslots = 4; k->bytes[0] = n * 16; k->bytes[1] = n * 16; k->bytes[2] = n * 4; k->bytes[3] = 44; break;
        default: return sa_fail (ICP_EINVAL, "unknown kernel object kind");
    }
    return ICP_OK;
}

}  // namespace

extern "C" {

const char *icp_kernel_last_error (void) { return g_sa_error.c_str (); }

int icp_ko_create (icp_ko_handle *out, int device, int kind, uint32_t n, uint32_t aux, float c) try
{
    if (!out) return sa_fail (ICP_EINVAL, "null pointer");
    *out = nullptr;
    int rc = sa_device (device); if (rc) return rc;
    icp_ko *k = new icp_ko ();
    k->device = device; k->kind = kind; k->n = n; k->aux = aux; k->c = c;
    auto bail = [&] (int code) { (void) icp_ko_destroy (k); return code; };
    switch (kind) {
        case ICP_KO_LMS: k->n = 640u * 480u; break;
        case ICP_KO_REPS: {
            // the representative grid — src/ICP/algorithms.cpp:842-854 generalised to a sqrt (m) x sqrt (m) landmark grid
            const uint32_t m = n, nr = aux;
            if (m == 0 || nr == 0 || nr > m || (nr & (nr - 1))) return bail (sa_fail (ICP_EINVAL, "nr must be a power of two, at most m"));
            const uint32_t side = (uint32_t) std::floor (std::sqrt ((double) m) + 0.5);
            if ((uint64_t) side * side != m) return bail (sa_fail (ICP_EINVAL, "m must be a square number (the landmark grid)"));
            uint32_t pw = 0; while ((1u << (pw + 1)) <= nr) ++pw;
            k->p.m = m; k->p.nr = nr; k->p.side = side; k->p.nrx = 1u << (pw - pw / 2); k->p.nry = 1u << (pw / 2);
            if (side % k->p.nrx || side % k->p.nry) return bail (sa_fail (ICP_EINVAL, "the representative grid must tile the landmark grid"));
            break;
        }
        case ICP_KO_WEIGHTS: case ICP_KO_MEAN: case ICP_KO_MEAN_WEIGHTED:
            if (n == 0 || (n % 2)) return bail (sa_fail (ICP_EINVAL, "The number of points in the array must be a (positive) multiple of 2"));      // src/ICP/algorithms.cpp:1050, :1306, :1573
            break;
        case ICP_KO_DEVS: case ICP_KO_S: case ICP_KO_S_WEIGHTED:
            if (n == 0) return bail (sa_fail (ICP_EINVAL, "The array cannot have zero points"));
            break;
        default: return bail (sa_fail (ICP_EINVAL, "unknown kernel object kind"));
    }
    if ((rc = ko_layout (k))) return bail (rc);
    // scratch: the levels of the trees
    auto scratch = [&] (size_t bytes) -> int { void *q = nullptr; int r = ko_alloc (k, &q, bytes, true); if (!r) k->scratch.push_back (q); return r; };
    if (kind == ICP_KO_WEIGHTS) {
        const uint32_t wgp = pad4 ((n + 127u) / 128u), npad = wgp * 128u;
        if ((rc = scratch ((size_t) npad * 4)) || (rc = scratch ((size_t) wgp * 4))) return bail (rc);
    } else if (kind == ICP_KO_MEAN || kind == ICP_KO_MEAN_WEIGHTED) {
        const uint32_t npad = ((n + 127u) / 128u) * 128u;
        if ((rc = scratch ((size_t) 6 * npad * 4))) return bail (rc);
        for (uint32_t cnt = npad;;) {               // block means (one per 128 pairs), then icpGMean (:530-566) until one vector per set remains
            const uint32_t ng = (cnt + 127u) / 128u;
            if ((rc = scratch ((size_t) 6 * ng * 4))) return bail (rc);
            k->level_cols.push_back (ng);
            cnt = ng;
            if (ng == 1u) break;
        }
    } else if (kind == ICP_KO_S || kind == ICP_KO_S_WEIGHTED) {
        uint32_t n4 = n; if (n4 % 4u) n4 += 4u - n4 % 4u;
        const uint32_t G = n4 / 4u, Gp = (G + 3u) & ~3u;                    // src/ICP/algorithms.cpp:2344-2346; columns padded to float4
        if ((rc = scratch ((size_t) 11 * Gp * 4))) return bail (rc);
        for (uint32_t cols = Gp;;) {                // reduce_sum_f until one value per row
            const uint32_t wgp = pad4 ((cols + 511u) / 512u);
            if ((rc = scratch ((size_t) 11 * wgp * 4))) return bail (rc);
            k->level_cols.push_back (wgp);
            cols = wgp;
            if (wgp == 1u) break;
        }
    }
    *out = k;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_ko_destroy (icp_ko_handle k) try
{
    if (!k) return ICP_EINVAL;
    (void) hipSetDevice (k->device);
    (void) hipStreamSynchronize (nullptr);           // (the null stream the kernel objects use — not the device: another handle's tracking streams may hold a waiting gate)
    for (int s = 0; s < icp_ko::MAXS; ++s) if (k->owned[s] && k->slot[s]) (void) hipFree (k->slot[s]);
    for (void *q : k->scratch) (void) hipFree (q);
    delete k;
    return ICP_OK;
}
ICP_CATCH_ALL

// A slot's buffer comes into being at its first use (write, device_ptr, run): until then another object's device pointer can take its
// place (icp_ko_adopt — the reference: `get (Memory)` assigned before `init`, include/ICP/algorithms.hpp:2214-2220).
static int ko_ensure (icp_ko *k, int s)
{
    if (k->slot[s]) return ICP_OK;
    int rc = ko_alloc (k, &k->slot[s], k->bytes[s], true);
    if (!rc) k->owned[s] = true;
    return rc;
}

int icp_ko_adopt (icp_ko_handle k, int s, void *dptr) try
{
    if (!k || !dptr || s < 0 || s >= k->nslots) return sa_fail (ICP_EINVAL, "icp_ko_adopt: bad arguments");
    if (hipSetDevice (k->device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    if (k->owned[s] && k->slot[s]) { (void) hipStreamSynchronize (nullptr); (void) hipFree (k->slot[s]); }
    k->slot[s] = dptr; k->owned[s] = false;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_ko_device_ptr (icp_ko_handle k, int s, void **dptr) try
{
    if (!k || !dptr || s < 0 || s >= k->nslots) return sa_fail (ICP_EINVAL, "icp_ko_device_ptr: bad arguments");
    if (hipSetDevice (k->device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    int rc = ko_ensure (k, s); if (rc) return rc;
    *dptr = k->slot[s];
    return ICP_OK;
}
ICP_CATCH_ALL

size_t icp_ko_slot_bytes (icp_ko_handle k, int s) { return (k && s >= 0 && s < k->nslots) ? k->bytes[s] : 0; }

int icp_ko_write (icp_ko_handle k, int s, const void *host) try
{
    if (!k || !host || s < 0 || s >= k->nslots) return sa_fail (ICP_EINVAL, "icp_ko_write: bad arguments");
    if (hipSetDevice (k->device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    int rc = ko_ensure (k, s); if (rc) return rc;
    hipError_t e = hipMemcpy (k->slot[s], host, k->bytes[s], hipMemcpyHostToDevice);
    return e == hipSuccess ? ICP_OK : sa_fail (ICP_EHIP, std::string ("hipMemcpy: ") + hipGetErrorString (e));
}
ICP_CATCH_ALL

int icp_ko_read (icp_ko_handle k, int s, void *host) try
{
    if (!k || !host || s < 0 || s >= k->nslots) return sa_fail (ICP_EINVAL, "icp_ko_read: bad arguments");
    if (hipSetDevice (k->device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    int rc = ko_ensure (k, s); if (rc) return rc;
    hipError_t e = hipMemcpy (host, k->slot[s], k->bytes[s], hipMemcpyDeviceToHost);       // (blocking: behind the kernels on the null stream)
    return e == hipSuccess ? ICP_OK : sa_fail (ICP_EHIP, std::string ("hipMemcpy: ") + hipGetErrorString (e));
}
ICP_CATCH_ALL

int icp_ko_set_scaling (icp_ko_handle k, float c) try { if (!k) return ICP_EINVAL; k->c = c; return ICP_OK; } ICP_CATCH_ALL

int icp_ko_run (icp_ko_handle k) try
{
    if (!k) return sa_fail (ICP_EINVAL, "null handle");
    if (hipSetDevice (k->device) != hipSuccess) return sa_fail (ICP_EHIP, "hipSetDevice failed");
    for (int s = 0; s < k->nslots; ++s) { int rc = ko_ensure (k, s); if (rc) return rc; }
    const uint32_t n = k->n;
    switch (k->kind) {
        case ICP_KO_LMS:
            hipLaunchKernelGGL (k_sa_get_lms, dim3 (128), dim3 (256), 0, 0, (const float4 *) k->slot[0], (float4 *) k->slot[1]);
            break;
        case ICP_KO_REPS:
            hipLaunchKernelGGL (k_sa_get_reps, dim3 ((k->aux + 255u) / 256u), dim3 (256), 0, 0, (const float4 *) k->slot[0], (float4 *) k->slot[1], k->p);
            break;
        case ICP_KO_WEIGHTS: {
            const uint32_t wgp = pad4 ((n + 127u) / 128u), npad = wgp * 128u;
            float *plane = (float *) k->scratch[0], *part = (float *) k->scratch[1];
            hipLaunchKernelGGL (k_sa_weights, dim3 ((npad + 255u) / 256u), dim3 (256), 0, 0, (const icp_dist_id *) k->slot[0], n, npad, (float *) k->slot[1], plane);
            hipLaunchKernelGGL (k_sa_tree128, dim3 ((wgp + 3u) / 4u, 1), dim3 (64), 0, 0, plane, npad, npad, wgp, wgp, part);
            hipLaunchKernelGGL (k_sa_sum_fd, dim3 (1), dim3 (64), 0, 0, part, wgp, (double *) k->slot[2]);
            break;
        }
        case ICP_KO_MEAN: case ICP_KO_MEAN_WEIGHTED: {
            const uint32_t npad = ((n + 127u) / 128u) * 128u;
            float *planes = (float *) k->scratch[0];
            hipLaunchKernelGGL (k_sa_mean_scale, dim3 ((npad + 255u) / 256u), dim3 (256), 0, 0, (const float *) k->slot[0], (const float *) k->slot[1],
                                (const float *) k->slot[2], (const double *) k->slot[3], n, npad, k->kind == ICP_KO_MEAN_WEIGHTED ? 1 : 0, planes);
            const float *cur = planes; uint32_t cnt = npad, stride = npad;
            for (size_t l = 0; l < k->level_cols.size (); ++l) {
                const uint32_t ng = k->level_cols[l];
                float *nxt = (float *) k->scratch[1 + l];
                hipLaunchKernelGGL (k_sa_tree128, dim3 ((ng + 3u) / 4u, 6), dim3 (64), 0, 0, cur, cnt, stride, ng, ng, nxt);
                cur = nxt; cnt = ng; stride = ng;
            }
            hipLaunchKernelGGL (k_sa_pack_mean8, dim3 (1), dim3 (64), 0, 0, cur, (float *) k->slot[4]);
            break;
        }
        case ICP_KO_DEVS:
            hipLaunchKernelGGL (k_sa_devs, dim3 ((n + 255u) / 256u), dim3 (256), 0, 0, (const float4 *) k->slot[0], (const float4 *) k->slot[1],
                                (const float *) k->slot[2], n, (float4 *) k->slot[3], (float4 *) k->slot[4]);
            break;
        case ICP_KO_S: case ICP_KO_S_WEIGHTED: {
            uint32_t n4 = n; if (n4 % 4u) n4 += 4u - n4 % 4u;
            const uint32_t G = n4 / 4u, Gp = (G + 3u) & ~3u;
            float *Sij = (float *) k->scratch[0];
            hipLaunchKernelGGL (k_sa_sij, dim3 ((G + 255u) / 256u), dim3 (256), 0, 0, (const float4 *) k->slot[0], (const float4 *) k->slot[1],
                                (const float *) k->slot[2], n, G, Gp, k->c, k->kind == ICP_KO_S_WEIGHTED ? 1 : 0, Sij);
            const float *cur = Sij; uint32_t cols = Gp;
            for (size_t l = 0; l < k->level_cols.size (); ++l) {
                const uint32_t wgp = k->level_cols[l];
                float *nxt = (float *) k->scratch[1 + l];
                hipLaunchKernelGGL (k_sa_sum_level, dim3 ((wgp + 3u) / 4u, 11), dim3 (64), 0, 0, cur, cols, wgp, nxt);
                cur = nxt; cols = wgp;
            }
            (void) hipMemcpyAsync (k->slot[3], cur, 44, hipMemcpyDeviceToDevice, 0);
            break;
        }
        default: return sa_fail (ICP_EINVAL, "unknown kernel object kind");
    }
    hipError_t e = hipGetLastError ();
    return e == hipSuccess ? ICP_OK : sa_fail (ICP_EHIP, std::string ("icp_ko_run: ") + hipGetErrorString (e));
}
ICP_CATCH_ALL

// ---- the one-call forms: create, upload, run, download, destroy ----------------------------------------------------------------------

namespace {
struct ko_guard { icp_ko_handle k = nullptr; ~ko_guard () { if (k) (void) icp_ko_destroy (k); } };
}

int icp_kernel_lms (int device, const void *cloud, void *lms) try
{
    if (!cloud || !lms) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, ICP_KO_LMS, 0, 0, 0.f)) || (rc = icp_ko_write (g.k, 0, cloud)) || (rc = icp_ko_run (g.k))) return rc;
    return icp_ko_read (g.k, 1, lms);
}
ICP_CATCH_ALL

int icp_kernel_reps (int device, const void *F, uint32_t m, uint32_t nr, void *R) try
{
    if (!F || !R) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, ICP_KO_REPS, m, nr, 0.f)) || (rc = icp_ko_write (g.k, 0, F)) || (rc = icp_ko_run (g.k))) return rc;
    return icp_ko_read (g.k, 1, R);
}
ICP_CATCH_ALL

int icp_kernel_weights (int device, const void *nn_id, uint32_t n, float *W, double *sum_w) try
{
    if (!nn_id || !W || !sum_w) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, ICP_KO_WEIGHTS, n, 0, 0.f)) || (rc = icp_ko_write (g.k, 0, nn_id)) || (rc = icp_ko_run (g.k))) return rc;
    if ((rc = icp_ko_read (g.k, 1, W))) return rc;
    return icp_ko_read (g.k, 2, sum_w);
}
ICP_CATCH_ALL

int icp_kernel_mean (int device, int weighted, const void *F, const void *M, const float *W, double sum_w, uint32_t n, float *mean8) try
{
    if (!F || !M || !mean8 || (weighted && !W)) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, weighted ? ICP_KO_MEAN_WEIGHTED : ICP_KO_MEAN, n, 0, 0.f))) return rc;
    if ((rc = icp_ko_write (g.k, 0, F)) || (rc = icp_ko_write (g.k, 1, M)) || (weighted && (rc = icp_ko_write (g.k, 2, W))) || (rc = icp_ko_write (g.k, 3, &sum_w))) return rc;
    if ((rc = icp_ko_run (g.k))) return rc;
    return icp_ko_read (g.k, 4, mean8);
}
ICP_CATCH_ALL

int icp_kernel_devs (int device, const void *F, const void *M, const float *mean8, uint32_t n, float *DF, float *DM) try
{
    if (!F || !M || !mean8 || !DF || !DM) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, ICP_KO_DEVS, n, 0, 0.f))) return rc;
    if ((rc = icp_ko_write (g.k, 0, F)) || (rc = icp_ko_write (g.k, 1, M)) || (rc = icp_ko_write (g.k, 2, mean8)) || (rc = icp_ko_run (g.k))) return rc;
    if ((rc = icp_ko_read (g.k, 3, DF))) return rc;
    return icp_ko_read (g.k, 4, DM);
}
ICP_CATCH_ALL

int icp_kernel_s (int device, int weighted, const float *DM, const float *DF, const float *W, uint32_t m, float c, float *S11) try
{
    if (!DM || !DF || !S11 || (weighted && !W)) return sa_fail (ICP_EINVAL, "null pointer");
    ko_guard g; int rc;
    if ((rc = icp_ko_create (&g.k, device, weighted ? ICP_KO_S_WEIGHTED : ICP_KO_S, m, 0, c))) return rc;
    if ((rc = icp_ko_write (g.k, 0, DM)) || (rc = icp_ko_write (g.k, 1, DF)) || (weighted && (rc = icp_ko_write (g.k, 2, W))) || (rc = icp_ko_run (g.k))) return rc;
    return icp_ko_read (g.k, 3, S11);
}
ICP_CATCH_ALL

}  // extern "C"
