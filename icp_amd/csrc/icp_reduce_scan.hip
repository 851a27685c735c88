// icp_reduce_scan.hip — the reference's standalone `Reduce` and `Scan` classes as resident objects
// (icp_rs_*: device buffers and a stream per object, run = kernels only) and as one-shot calls (icp_reduce / icp_scan): Reduce<MIN,float> / Reduce<MAX,uint> / Reduce<SUM,float> (kernels/reduce_kernels.cl:68, 149, 230;
// host src/ICP/algorithms.cpp:131-322) and Scan<INCLUSIVE|EXCLUSIVE,int> (kernels/scan_kernels.cl:67, 188, 296;
// host :403-600).  Row-wise over a rows x cols array (cols a multiple of 4, as the reference requires).
// SUM follows reduce_sum_f's tree exactly (oracle orc_reduce_sum_f): bit-identical sums; MIN/MAX and the
// integer scans are exact in any order.
#include "../../include/icp_amd.h"
#include "icp_cguard.h"
#include "icp_device.h"

#include <string>

namespace {

thread_local std::string g_rs_error;

static __device__ __forceinline__ float rs_sum4 (float4 v) { return ((v.x + v.y) + v.z) + v.w; }

// one reduce_sum_f pass: work-group g covers 512 columns = 128 positions of 4; one 16-lane row per work-group
__global__ __launch_bounds__ (64) void k_rs_sum_level (const float *in, uint32_t cols, uint32_t wgp, float *out)
{
    const uint32_t lane = threadIdx.x, l = lane & 15u, g = blockIdx.x * 4u + (lane >> 4), row = blockIdx.y;
    const float *src = in + (size_t) row * cols;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t c = g * 512u + 4u * (l + 16u * k);
        a[k] = (g < wgp && c < cols) ? rs_sum4 (*reinterpret_cast<const float4 *> (src + c)) : 0.f;
    }
    const float v = row_tree8 (a);
    if (l == 0 && g < wgp) out[(size_t) row * wgp + g] = v;
}

template <typename T, bool MAXOP>
__global__ __launch_bounds__ (256) void k_rs_minmax (const T *in, uint32_t cols, T *out)
{
    __shared__ T s[4];
    const uint32_t row = blockIdx.x, t = threadIdx.x;
    const T *src = in + (size_t) row * cols;
    T v = src[min (t, cols - 1u)];
    for (uint32_t c = t; c < cols; c += 256u) { T x = src[c]; v = MAXOP ? (x > v ? x : v) : (x < v ? x : v); }
    for (int d = 32; d > 0; d >>= 1) { T x = __shfl_xor (v, d); v = MAXOP ? (x > v ? x : v) : (x < v ? x : v); }
    if ((t & 63u) == 0) s[t >> 6] = v;
    __syncthreads ();
    if (t == 0) {
        for (int k = 1; k < 4; ++k) { T x = s[k]; v = MAXOP ? (x > v ? x : v) : (x < v ? x : v); }
        out[row] = v;
    }
}

// row-wise int scan: 256 threads, contiguous chunk per thread, block scan of the chunk sums
__global__ __launch_bounds__ (256) void k_rs_scan (const int *in, uint32_t cols, int inclusive, int *out)
{
    __shared__ int s[256];
    const uint32_t row = blockIdx.x, t = threadIdx.x;
    const int *src = in + (size_t) row * cols;
    int *dst = out + (size_t) row * cols;
    const uint32_t per = (cols + 255u) / 256u, lo = min (t * per, cols), hi = min (lo + per, cols);
    int sum = 0;
    for (uint32_t c = lo; c < hi; ++c) sum += src[c];
    s[t] = sum;
    __syncthreads ();
    for (uint32_t d = 1; d < 256u; d <<= 1) {        // Hillis-Steele inclusive scan of the 256 chunk sums
        int x = (t >= d) ? s[t - d] : 0;
        __syncthreads ();
        s[t] += x;
        __syncthreads ();
    }
    int run = (t == 0) ? 0 : s[t - 1];
    for (uint32_t c = lo; c < hi; ++c) {
        int x = src[c];
        if (inclusive) { run += x; dst[c] = run; } else { dst[c] = run; run += x; }
    }
}

int rs_fail (int code, const std::string &m) { g_rs_error = m; return code; }

#define RSCHK(expr)                                                                            \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ok = false; err = std::string (#expr) + ": " + hipGetErrorString (e_); } } while (0)

}  // namespace

struct icp_rs_context {
    int device = 0, kind = 0;
    uint32_t cols = 0, rows = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *din = nullptr, *dout = nullptr, *dtmp = nullptr;
    const void *result = nullptr;                // where the last run left the rows results (SUM: ping-pong buffers)
    size_t out_bytes = 0;
};

extern "C" {

const char *icp_reduce_scan_last_error (void) { return g_rs_error.c_str (); }

int icp_rs_destroy (icp_rs_handle r) try
{
    if (!r) return ICP_EINVAL;
    (void) hipSetDevice (r->device);
    if (r->stream) (void) hipStreamSynchronize (r->stream);
    if (r->din) (void) hipFree (r->din);
    if (r->dout) (void) hipFree (r->dout);
    if (r->dtmp) (void) hipFree (r->dtmp);
    if (r->ev0) (void) hipEventDestroy (r->ev0);
    if (r->ev1) (void) hipEventDestroy (r->ev1);
    if (r->stream) (void) hipStreamDestroy (r->stream);
    delete r;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_rs_create (icp_rs_handle *out, int device, int kind, uint32_t cols, uint32_t rows) try
{
    if (!out) return ICP_EINVAL;
    *out = nullptr;
    if (cols == 0 || rows == 0) return rs_fail (ICP_EINVAL, "The array cannot have zero columns");
    if (cols % 4) return rs_fail (ICP_EINVAL, "The number of columns in the array must be a multiple of 4");   // algorithms.cpp:151, :421
    if (kind < 0 || kind > ICP_RS_SCAN_EXCLUSIVE) return rs_fail (ICP_EINVAL, "kind must be an icp_rs_kind");
    int count = 0;
    if (hipGetDeviceCount (&count) != hipSuccess || device < 0 || device >= count) return rs_fail (ICP_ENODEVICE, "no HIP device");
    bool ok = true; std::string err;
    icp_rs_context *r = new icp_rs_context ();
    r->device = device; r->kind = kind; r->cols = cols; r->rows = rows;
    RSCHK (hipSetDevice (device));
    if (ok) RSCHK (hipStreamCreateWithFlags (&r->stream, hipStreamNonBlocking));
    if (ok) RSCHK (hipEventCreate (&r->ev0));
    if (ok) RSCHK (hipEventCreate (&r->ev1));
    const size_t n = (size_t) cols * rows;
    if (ok) RSCHK (hipMalloc (&r->din, n * 4));
    if (kind >= ICP_RS_SCAN_INCLUSIVE) {
        r->out_bytes = n * 4;
        if (ok) RSCHK (hipMalloc (&r->dout, n * 4));
    } else {
        const uint32_t wg0 = (cols + 511u) / 512u, wgp0 = (wg0 != 1 && (wg0 % 4)) ? wg0 + 4 - wg0 % 4 : wg0;
        r->out_bytes = (size_t) rows * 4;
        if (ok) RSCHK (hipMalloc (&r->dout, (size_t) rows * wgp0 * 4));
        if (ok) RSCHK (hipMalloc (&r->dtmp, (size_t) rows * wgp0 * 4));
    }
    r->result = r->dout;
    if (!ok) { icp_rs_destroy (r); return rs_fail (ICP_EHIP, err); }
    *out = r;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_rs_write (icp_rs_handle r, const void *host_in) try
{
    if (!r || !host_in) return ICP_EINVAL;
    bool ok = true; std::string err;
    RSCHK (hipSetDevice (r->device));
    if (ok) RSCHK (hipMemcpyAsync (r->din, host_in, (size_t) r->cols * r->rows * 4, hipMemcpyHostToDevice, r->stream));
    if (ok) RSCHK (hipStreamSynchronize (r->stream));            // (pageable source)
    return ok ? ICP_OK : rs_fail (ICP_EHIP, err);
}
ICP_CATCH_ALL

// enqueue only: the kernels of one run on the object's stream, device buffers resident
static int rs_enqueue (icp_rs_context *r)
{
    const uint32_t cols = r->cols, rows = r->rows;
    if (r->kind == ICP_RS_SUM_F) {
        const float *cur = static_cast<const float *> (r->din);
        float *a = static_cast<float *> (r->dout), *b = static_cast<float *> (r->dtmp);
        uint32_t c = cols;
        for (;;) {
            uint32_t wg = (c + 511u) / 512u, wgp = (wg != 1 && (wg % 4)) ? wg + 4 - wg % 4 : wg;   // algorithms.cpp:140-142
            hipLaunchKernelGGL (k_rs_sum_level, dim3 ((wgp + 3) / 4, rows), dim3 (64), 0, r->stream, cur, c, wgp, a);
            cur = a; c = wgp;
            float *t = a; a = b; b = t;
            if (wgp == 1) break;
        }
        r->result = cur;
    } else if (r->kind == ICP_RS_MIN_F)
        hipLaunchKernelGGL ((k_rs_minmax<float, false>), dim3 (rows), dim3 (256), 0, r->stream, static_cast<const float *> (r->din), cols, static_cast<float *> (r->dout));
    else if (r->kind == ICP_RS_MAX_UI)
        hipLaunchKernelGGL ((k_rs_minmax<uint32_t, true>), dim3 (rows), dim3 (256), 0, r->stream, static_cast<const uint32_t *> (r->din), cols, static_cast<uint32_t *> (r->dout));
    else
        hipLaunchKernelGGL (k_rs_scan, dim3 (rows), dim3 (256), 0, r->stream, static_cast<const int *> (r->din), cols, r->kind == ICP_RS_SCAN_INCLUSIVE ? 1 : 0, static_cast<int *> (r->dout));
    hipError_t e = hipGetLastError ();
    return e == hipSuccess ? ICP_OK : rs_fail (ICP_EHIP, std::string ("kernel launch: ") + hipGetErrorString (e));
}

int icp_rs_run (icp_rs_handle r) try
{
    if (!r) return ICP_EINVAL;
    if (hipSetDevice (r->device) != hipSuccess) return rs_fail (ICP_EHIP, "hipSetDevice");
    return rs_enqueue (r);
}
ICP_CATCH_ALL

int icp_rs_read (icp_rs_handle r, void *host_out) try
{
    if (!r || !host_out) return ICP_EINVAL;
    bool ok = true; std::string err;
    RSCHK (hipSetDevice (r->device));
    if (ok) RSCHK (hipMemcpyAsync (host_out, r->result, r->out_bytes, hipMemcpyDeviceToHost, r->stream));
    if (ok) RSCHK (hipStreamSynchronize (r->stream));
    return ok ? ICP_OK : rs_fail (ICP_EHIP, err);
}
ICP_CATCH_ALL

int icp_rs_device_ptr (icp_rs_handle r, int output, void **dptr) try
{
    if (!r || !dptr) return ICP_EINVAL;
    *dptr = output ? const_cast<void *> (r->result) : r->din;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_rs_time (icp_rs_handle r, uint32_t reps, float *us_per_run) try
{
    if (!r || !us_per_run || reps == 0) return ICP_EINVAL;
    bool ok = true; std::string err;
    RSCHK (hipSetDevice (r->device));
    int rc = rs_enqueue (r);                                          // warm-up
    if (rc) return rc;
    if (ok) RSCHK (hipEventRecord (r->ev0, r->stream));
    for (uint32_t i = 0; i < reps && rc == ICP_OK; ++i) rc = rs_enqueue (r);
    if (rc) return rc;
    if (ok) RSCHK (hipEventRecord (r->ev1, r->stream));
    if (ok) RSCHK (hipEventSynchronize (r->ev1));
    float ms = 0.f;
    if (ok) RSCHK (hipEventElapsedTime (&ms, r->ev0, r->ev1));
    *us_per_run = ms * 1e3f / (float) reps;
    return ok ? ICP_OK : rs_fail (ICP_EHIP, err);
}
ICP_CATCH_ALL

// one-shot forms: create, write, run, read, destroy
static int rs_oneshot (int device, int kind, const void *host_in, uint32_t cols, uint32_t rows, void *host_out)
{
    if (!host_in || !host_out) return rs_fail (ICP_EINVAL, "null pointer");
    icp_rs_handle r = nullptr;
    int rc = icp_rs_create (&r, device, kind, cols, rows);
    if (rc) return rc;
    rc = icp_rs_write (r, host_in);
    if (rc == ICP_OK) rc = icp_rs_run (r);
    if (rc == ICP_OK) rc = icp_rs_read (r, host_out);
    icp_rs_destroy (r);
    return rc;
}

int icp_reduce (int device, int op, const void *host_in, uint32_t cols, uint32_t rows, void *host_out) try
{
    if (op < 0 || op > 2) return rs_fail (ICP_EINVAL, "op must be ICP_REDUCE_MIN_F, ICP_REDUCE_MAX_UI or ICP_REDUCE_SUM_F");
    return rs_oneshot (device, op, host_in, cols, rows, host_out);
}
ICP_CATCH_ALL

int icp_scan (int device, int inclusive, const int32_t *host_in, uint32_t cols, uint32_t rows, int32_t *host_out) try
{
    return rs_oneshot (device, inclusive ? ICP_RS_SCAN_INCLUSIVE : ICP_RS_SCAN_EXCLUSIVE, host_in, cols, rows, host_out);
}
ICP_CATCH_ALL

}  // extern "C"
