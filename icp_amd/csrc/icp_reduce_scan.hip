// icp_reduce_scan.hip — the reference's standalone `Reduce` and `Scan` classes as stateless
// C-ABI calls: Reduce<MIN,float> / Reduce<MAX,uint> / Reduce<SUM,float> (kernels/reduce_kernels.cl:68, 149, 230;
// host src/ICP/algorithms.cpp:131-322) and Scan<INCLUSIVE|EXCLUSIVE,int> (kernels/scan_kernels.cl:67, 188, 296;
// host :403-600).  Row-wise over a rows x cols array (cols a multiple of 4, as the reference requires).
// SUM follows reduce_sum_f's tree exactly (oracle orc_reduce_sum_f): bit-identical sums; MIN/MAX and the
// integer scans are exact in any order.
#include "../../include/icp_amd.h"
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

extern "C" {

const char *icp_reduce_scan_last_error (void) { return g_rs_error.c_str (); }

int icp_reduce (int device, int op, const void *host_in, uint32_t cols, uint32_t rows, void *host_out)
{
    if (!host_in || !host_out || cols == 0 || rows == 0) return rs_fail (ICP_EINVAL, "The array cannot have zero columns");
    if (cols % 4) return rs_fail (ICP_EINVAL, "The number of columns in the array must be a multiple of 4");   // algorithms.cpp:151
    if (op < 0 || op > 2) return rs_fail (ICP_EINVAL, "op must be ICP_REDUCE_MIN_F, ICP_REDUCE_MAX_UI or ICP_REDUCE_SUM_F");
    int count = 0;
    if (hipGetDeviceCount (&count) != hipSuccess || device < 0 || device >= count) return rs_fail (ICP_ENODEVICE, "no HIP device");
    bool ok = true; std::string err;
    RSCHK (hipSetDevice (device));
    const size_t n = (size_t) cols * rows;
    void *din = nullptr, *dout = nullptr, *dtmp = nullptr;
    RSCHK (hipMalloc (&din, n * 4));
    const uint32_t wg0 = (cols + 511u) / 512u, wgp0 = (wg0 != 1 && (wg0 % 4)) ? wg0 + 4 - wg0 % 4 : wg0;
    RSCHK (hipMalloc (&dout, (size_t) rows * wgp0 * 4));
    RSCHK (hipMalloc (&dtmp, (size_t) rows * wgp0 * 4));
    if (ok) RSCHK (hipMemcpy (din, host_in, n * 4, hipMemcpyHostToDevice));
    if (ok) {
        if (op == ICP_REDUCE_SUM_F) {
            const float *cur = static_cast<const float *> (din);
            float *a = static_cast<float *> (dout), *b = static_cast<float *> (dtmp);
            uint32_t c = cols;
            for (;;) {
                uint32_t wg = (c + 511u) / 512u, wgp = (wg != 1 && (wg % 4)) ? wg + 4 - wg % 4 : wg;   // algorithms.cpp:140-142
                hipLaunchKernelGGL (k_rs_sum_level, dim3 ((wgp + 3) / 4, rows), dim3 (64), 0, 0, cur, c, wgp, a);
                cur = a; c = wgp;
                float *t = a; a = b; b = t;
                if (wgp == 1) break;
            }
            RSCHK (hipGetLastError ());
            RSCHK (hipMemcpy (host_out, cur, (size_t) rows * 4, hipMemcpyDeviceToHost));
        } else {
            if (op == ICP_REDUCE_MIN_F) hipLaunchKernelGGL ((k_rs_minmax<float, false>), dim3 (rows), dim3 (256), 0, 0, static_cast<const float *> (din), cols, static_cast<float *> (dout));
            else hipLaunchKernelGGL ((k_rs_minmax<uint32_t, true>), dim3 (rows), dim3 (256), 0, 0, static_cast<const uint32_t *> (din), cols, static_cast<uint32_t *> (dout));
            RSCHK (hipGetLastError ());
            RSCHK (hipMemcpy (host_out, dout, (size_t) rows * 4, hipMemcpyDeviceToHost));
        }
    }
    if (din) (void) hipFree (din);
    if (dout) (void) hipFree (dout);
    if (dtmp) (void) hipFree (dtmp);
    return ok ? ICP_OK : rs_fail (ICP_EHIP, err);
}

int icp_scan (int device, int inclusive, const int32_t *host_in, uint32_t cols, uint32_t rows, int32_t *host_out)
{
    if (!host_in || !host_out || cols == 0 || rows == 0) return rs_fail (ICP_EINVAL, "The array cannot have zero columns");
    if (cols % 4) return rs_fail (ICP_EINVAL, "The number of columns in the array must be a multiple of 4");   // algorithms.cpp:421
    int count = 0;
    if (hipGetDeviceCount (&count) != hipSuccess || device < 0 || device >= count) return rs_fail (ICP_ENODEVICE, "no HIP device");
    bool ok = true; std::string err;
    RSCHK (hipSetDevice (device));
    const size_t n = (size_t) cols * rows;
    int *din = nullptr, *dout = nullptr;
    RSCHK (hipMalloc ((void **) &din, n * 4));
    RSCHK (hipMalloc ((void **) &dout, n * 4));
    if (ok) RSCHK (hipMemcpy (din, host_in, n * 4, hipMemcpyHostToDevice));
    if (ok) {
        hipLaunchKernelGGL (k_rs_scan, dim3 (rows), dim3 (256), 0, 0, din, cols, inclusive ? 1 : 0, dout);
        RSCHK (hipGetLastError ());
        RSCHK (hipMemcpy (host_out, dout, n * 4, hipMemcpyDeviceToHost));
    }
    if (din) (void) hipFree (din);
    if (dout) (void) hipFree (dout);
    return ok ? ICP_OK : rs_fail (ICP_EHIP, err);
}

}  // extern "C"
