// icp_capi.hip — C-ABI (include/icp_amd.h) over the HIP kernels: life cycle of a handle, its buffers, reads and writes, single steps and
// runs, setters, diagnostics.
//
// Host-side counterpart of ICPStep<CR,CW> / ICP<CR,CW> (include/ICP/algorithms.hpp:2234-2496,
// src/ICP/algorithms.cpp:4348-4903).  The reference wires ten kernel-wrapper objects by sharing
// cl::Buffer handles (:4499-4581) and syncs with the host every iteration (:4681-4697); here one
// handle owns its streams, all device buffers of a batch of registrations and the device-resident
// registration state; a fixed-length run is one hipGraph launch, a checked run is driven launch by launch
// from the host (icp_run.hip), frame-to-frame tracking lives in icp_track.hip.
//
// There is NO CPU fallback: without a gfx950 device icp_create fails with ICP_ENODEVICE.
#include "icp_host.h"

using namespace icp_host;

namespace {

void free_all (icp_context *h)
{
    drop_graphs (h);
    for (void *q : h->dev_allocs) (void) hipFree (q);
    h->dev_allocs.clear ();
    if (h->hF) (void) hipHostFree (h->hF);
    if (h->hM) (void) hipHostFree (h->hM);
    if (h->hT) (void) hipHostFree (h->hT);
    if (h->hState) (void) hipHostFree (h->hState);
    if (h->hMirror) (void) hipHostFree (h->hMirror);
    if (h->hTrackMirror) (void) hipHostFree (h->hTrackMirror);
    h->hState = nullptr; h->hMirror = h->hTrackMirror = nullptr; h->hstate_fresh = false;
    h->run = run_ctl {}; h->track_k_hist[0] = h->track_k_hist[1] = 0; h->track_hist_frame = 0;
    if (h->dCloud) (void) hipFree (h->dCloud);
    if (h->dCloudOut) (void) hipFree (h->dCloudOut);
    h->hF = h->hM = h->hT = nullptr; h->dCloud = h->dCloudOut = nullptr; h->cloud_cap = 0;
    h->dF = h->dM = nullptr; h->ownF = h->ownM = true;
    for (int k = 0; k < 2; ++k) {
        if (h->hBand[k]) (void) hipHostFree (h->hBand[k]);
        if (h->hFrame[k]) (void) hipHostFree (h->hFrame[k]);
        if (h->dBand[k]) (void) hipFree (h->dBand[k]);
        h->hBand[k] = h->hFrame[k] = h->dBand[k] = nullptr;
    }
    for (const auto &r : h->sources) (void) hipHostUnregister (const_cast<char *> (r.base));
    h->sources.clear ();
    if (h->lm[2]) (void) hipFree (h->lm[2]);
    h->lm[0] = h->lm[1] = h->lm[2] = nullptr;
    if (h->hTrack) (void) hipHostFree (h->hTrack);
    h->hTrack = nullptr;
    {   // (whatever track_prepare got to: a failed stream probe leaves the buffers allocated and rbc2_ready false)
        icp_context::rbc_set &q = h->rbc[1];
        void *ptrs[] = { q.R, q.GB, q.OL, q.LB, q.XP, q.XQ, q.rep_src, q.owner, q.N, q.O, q.perm, q.chunk_hist, q.blist, q.bn, q.brank };
        for (void *x : ptrs) if (x) (void) hipFree (x);
    }
    h->rbc[0] = h->rbc[1] = icp_context::rbc_set {}; h->rbc2_ready = false;
    if (h->dSeq) (void) hipFree (h->dSeq);
    if (h->dRunFlag) (void) hipFree (h->dRunFlag);
    if (h->hGateFlag) (void) hipHostFree (h->hGateFlag);
    h->dSeq = h->dRunFlag = h->hGateFlag = nullptr; h->run2 = run_ctl {}; h->stream2_dirty = false; h->track_last_gated = false;
    h->inited = h->built = false; h->parity = 0; h->track_submitted = h->track_collected = 0;
}

// A setter is about to change what a re-run search would produce (alpha, the metric's scale, the reduction mode): per-query outputs a
// checked run left to be reproduced on demand (lazy outputs: materialize_outputs) are reproduced NOW, with the parameters the run used —
// icp_read then returns the bits of the last executed iteration, as include/icp_amd.h promises, whatever was set in between.
int outputs_before_change (icp_context *h)
{
    if (!h->inited || !h->outputs_stale) return ICP_OK;
    int rc = set_device (h); if (rc) return rc;
    if ((rc = run_close_all (h))) return rc;
    return materialize_outputs (h, ICP_MEM_NN_ID);
}

// landmark-grid / representative-grid validation — src/ICP/algorithms.cpp:842-854 generalised (oracle: orc_reps_grid)
bool reps_grid (uint32_t m, uint32_t nr, uint32_t *nrx, uint32_t *nry, uint32_t *side)
{
    if (m == 0 || nr == 0 || nr > m) return false;
    if (nr & (nr - 1)) return false;
    uint32_t g = (uint32_t) std::floor (std::sqrt ((double) m) + 0.5);
    if ((uint64_t) g * g != m) return false;
    uint32_t pw = 0; while ((1u << (pw + 1)) <= nr) ++pw;
    uint32_t x = 1u << (pw - pw / 2), y = 1u << (pw / 2);
    if (g % x || g % y) return false;
    *nrx = x; *nry = y; *side = g;
    return true;
}

}  // namespace

extern "C" {

const char *icp_version (void) { return "icp_amd 0.1 (gfx950)"; }

const char *icp_last_error (icp_handle h) { return h ? h->err.c_str () : g_create_error.c_str (); }

// PCI bus id of a device ("0000:c1:00.0"): icp_batch_create looks the device's NUMA node up with it.
int icp_device_pci_bus_id (int device, char *out, size_t cap) try
{
    if (!out || cap < 16) return ICP_EINVAL;
    int n = 0;
    if (hipGetDeviceCount (&n) != hipSuccess || device < 0 || device >= n) { (void) hipGetLastError (); return ICP_ENODEVICE; }
    if (hipDeviceGetPCIBusId (out, (int) cap, device) != hipSuccess) { (void) hipGetLastError (); out[0] = 0; return ICP_EHIP; }
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_device_count (int *n) try
{
    if (!n) return fail (nullptr, ICP_EINVAL, "icp_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount (&c);
    if (e != hipSuccess) { *n = 0; return fail (nullptr, ICP_ENODEVICE, std::string ("hipGetDeviceCount: ") + hipGetErrorString (e)); }
    *n = c;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_create (icp_handle *out, int device, int rot, int weighted) try
{
    if (!out) return ICP_EINVAL;
    *out = nullptr;
    if ((rot != ICP_ROT_EIGEN && rot != ICP_ROT_POWER_METHOD) || (weighted != 0 && weighted != 1))
        return fail (nullptr, ICP_EINVAL, "icp_create: rot must be 0|1 and weighted 0|1");
    int count = 0;
    hipError_t e = hipGetDeviceCount (&count);
    if (e != hipSuccess || count <= 0)
        return fail (nullptr, ICP_ENODEVICE, "icp_create: no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= count) return fail (nullptr, ICP_EINVAL, "icp_create: device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties (&prop, device);
    if (e != hipSuccess) return fail (nullptr, ICP_EHIP, std::string ("hipGetDeviceProperties: ") + hipGetErrorString (e));
    if (std::strncmp (prop.gcnArchName, "gfx950", 6) != 0)
        return fail (nullptr, ICP_ENODEVICE, std::string ("icp_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    icp_context *h = new icp_context ();
    h->device = device;
    // Default modes = the benchmarked path: single-pass double moments + squared power start.
    // ICP_AMD_MODE=reference (read here) starts the handle in the reference-order / literal modes instead, whose
    // intermediates restate the reference's arithmetic order; icp_set_reduce_mode / icp_set_power_mode switch later.
    h->p.rot = rot; h->p.weighted = weighted; h->p.power_mode = ICP_POWER_SQUARED; h->p.fused = ICP_REDUCE_FUSED;
    h->p.dist_scale = 1.f;
    { const char *e = std::getenv ("ICP_AMD_MODE"); if (e && (e[0] == 'r' || e[0] == 'R')) { h->p.power_mode = ICP_POWER_LITERAL; h->p.fused = ICP_REDUCE_REFERENCE_ORDER; } }
    { const char *e = std::getenv ("ICP_AMD_CHAIN"); h->p.chain = !e ? 1 : (e[0] == '1') ? 2 : (e[0] == '0') ? 0 : 1; }   // see icp_chain_supported
    { const char *e = std::getenv ("ICP_AMD_RUN_ADAPTIVE"); if (e && e[0] == '0') h->run_adaptive = 0; }                  // see run_ctl
    { const char *e = std::getenv ("ICP_AMD_TRACK_GATE"); if (e && e[0] == '0') h->track_gate = 0; }
    { const char *e = std::getenv ("ICP_AMD_OUTPUTS"); if (e && (e[0] == 'e' || e[0] == 'E')) h->outputs_lazy = 0; }
    { const char *e = std::getenv ("ICP_AMD_RUN_DEPTH"); if (e) { const int d = std::atoi (e); if (d >= 1 && d <= 64) h->run_depth = (uint32_t) d; } }
    e = hipSetDevice (device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags (&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags (&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate (&h->ev0);
    if (e == hipSuccess) e = hipEventCreate (&h->ev1);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evUp[k], hipEventDisableTiming);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evFrame[k], hipEventDisableTiming);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evDone[k], hipEventDisableTiming);
    for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evStage[k], hipEventDisableTiming);
    if (e != hipSuccess) { std::string m = hipGetErrorString (e); h->inited = false; icp_destroy (h); return fail (nullptr, ICP_EHIP, "icp_create: " + m); }
    *out = h;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_destroy (icp_handle h) try
{
    if (!h) return ICP_EINVAL;
    keeper_stop (h);                                 // (the tracking keeper, if this handle ever started one: joined before anything is freed)
    (void) hipSetDevice (h->device);
    if (h->run.active || h->run2.active) (void) run_close_all (h);
    if (h->stream2) (void) hipStreamSynchronize (h->stream2);
    if (h->copy_stream) (void) hipStreamSynchronize (h->copy_stream);
    if (h->stream) (void) hipStreamSynchronize (h->stream);
    free_all (h);
    if (h->dTin) (void) hipFree (h->dTin);
    if (h->ev0) (void) hipEventDestroy (h->ev0);
    if (h->ev1) (void) hipEventDestroy (h->ev1);
    for (int k = 0; k < 2; ++k) if (h->evUp[k]) (void) hipEventDestroy (h->evUp[k]);
    for (int k = 0; k < 2; ++k) if (h->evFrame[k]) (void) hipEventDestroy (h->evFrame[k]);
    for (int k = 0; k < 4; ++k) if (h->evDone[k]) (void) hipEventDestroy (h->evDone[k]);
    for (int k = 0; k < 3; ++k) if (h->evStage[k]) (void) hipEventDestroy (h->evStage[k]);
    if (h->copy_stream) (void) hipStreamDestroy (h->copy_stream);
    if (h->stream2) (void) hipStreamDestroy (h->stream2);
    if (h->stream) (void) hipStreamDestroy (h->stream);
    delete h;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_init_batched (icp_handle h, uint32_t batch, uint32_t m, uint32_t nr, float a, float c,
                      uint32_t max_iterations, double angle_threshold, double translation_threshold) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    // argument checks of the reference: src/ICP/algorithms.cpp:4413-4420, :1573, :842-854
    if (m == 0) return fail (h, ICP_EINVAL, "The sets of landmarks cannot have zero points");
    if (nr == 0) return fail (h, ICP_EINVAL, "The sets of representatives cannot have zero points");
    if (a == 0.f) return fail (h, ICP_EINVAL, "The alpha parameter cannot be equal to zero");
    if (m % 2) return fail (h, ICP_EINVAL, "The number of points in the array must be a multiple of 2");
    if (batch == 0 || batch > 65535u) return fail (h, ICP_EINVAL, "batch must be in [1, 65535]");
    if (max_iterations == 0) return fail (h, ICP_EINVAL, "max_iterations must be positive");
    uint32_t nrx, nry, side;
    if (!reps_grid (m, nr, &nrx, &nry, &side))
        return fail (h, ICP_EINVAL, "nr must be a power of two whose grid tiles the sqrt(m) x sqrt(m) landmark grid");
    if (nr > 32768u) return fail (h, ICP_EINVAL, "nr must be <= 32768");
    if (m > (1u << 20)) return fail (h, ICP_EINVAL, "m must be <= 2^20");
    int rc = set_device (h); if (rc) return rc;
    if ((rc = run_close_all (h))) return rc;
    if (h->copy_stream) HIPCHK (h, hipStreamSynchronize (h->copy_stream));
    if (h->stream2) HIPCHK (h, hipStreamSynchronize (h->stream2));
    if (h->stream) HIPCHK (h, hipStreamSynchronize (h->stream));
    int rot = h->p.rot, weighted = h->p.weighted, pmode = h->p.power_mode, fused = h->p.fused, chain = h->p.chain;
    const float dist_scale = h->p.dist_scale;
    free_all (h);
    icp_params &p = h->p;
    p = icp_params {};
    p.rot = rot; p.weighted = weighted; p.power_mode = pmode; p.check = 0; p.fused = fused; p.chain = chain; p.emit = 1;
    p.dist_scale = dist_scale;
    p.m = m; p.nr = nr; p.batch = batch; p.side = side; p.nrx = nrx; p.nry = nry;
    p.a = a; p.c = c;
    {   // division-free cell lookups in the kernels (reps_grid guarantees a square grid that the representative grid tiles)
        auto magic = [] (uint32_t d) { return d > 1u ? (uint32_t) ((1ull << 32) / d + 1ull) : 0u; };
        p.side_magic = magic (side); p.cellw_magic = magic (side / nrx); p.cellh_magic = magic (side / nry);
    }
    h->max_iterations = max_iterations; h->angle_threshold = angle_threshold; h->translation_threshold = translation_threshold;
    p.tan_half_thr = std::tan (angle_threshold * M_PI / 360.0);
    p.trans_thr = translation_threshold;
    p.nwg = (m + 127u) / 128u;                                       // src/ICP/algorithms.cpp:1038
    p.nwp = p.nwg; if (p.nwp != 1 && (p.nwp % 4)) p.nwp += 4 - p.nwp % 4;          // :1040
    uint32_t n4 = m; if (n4 % 4) n4 += 4 - n4 % 4;
    p.G = n4 / 4;                                                    // :2344-2346
    p.nsp = (p.G + 511u) / 512u; if (p.nsp != 1 && (p.nsp % 4)) p.nsp += 4 - p.nsp % 4;   // :140-142
    p.nchunk = (m + ICP_CHUNK - 1) / ICP_CHUNK;
    p.nb = (m + 63u) / 64u;
    { const uint32_t ng = (p.nb + 127u) / 128u; p.ng_magic = ng > 1u ? (uint32_t) ((1ull << 32) / ng + 1ull) : 0u; }    // (tasks: 18 ng < 2^16)

    const size_t B = batch;
    float *F = nullptr, *M = nullptr;
    if ((rc = dalloc (h, &F, B * m * 8))) return rc;
    if ((rc = dalloc (h, &M, B * m * 8))) return rc;
    h->dF = F; h->dM = M; p.F = F; p.M = M; h->lm[0] = F; h->lm[1] = M;
    if ((rc = dalloc (h, &p.R, B * nr * 8))) return rc;
    p.n16 = (nr + 15u) / 16u;
    p.nb = (m + 63u) / 64u;
    p.tbox = icp_tbox_of (p); p.n1k = (nr + p.tbox - 1u) / p.tbox;
    p.s2wave = icp_s2_wave_of (p);
    { const char *e = std::getenv ("ICP_AMD_XCDMAP"); p.xcdmap = e ? (e[0] == '1') : (B == 1u); }
    { const char *e = std::getenv ("ICP_AMD_WARM_SEED"); p.warm_seed = (e && e[0] == '1') ? 1u : 0u; }
    p.gtile = 0u;                                                    // 4 x 4 tile groups where the representative grid allows
    if (nrx % 4u == 0u && nry % 4u == 0u && !std::getenv ("ICP_AMD_STRIP_GROUPS")) { uint32_t lg = 0; while ((4u << lg) < nrx) ++lg; p.gtile = lg + 1u; }
    if ((rc = dalloc (h, &p.GB, B * 2 * (p.n16 + p.n1k)))) return rc;
    if ((rc = dalloc (h, &p.XP, B * m * 8))) return rc;
    if ((rc = dalloc (h, &p.XQ, B * m * 8))) return rc;
    if ((rc = dalloc (h, &p.OL, B * ICP_OL_STRIDE (nr)))) return rc;
    p.nlb = m / 16u + 2u;
    if ((rc = dalloc (h, &p.LB, B * 3 * p.nlb))) return rc;
    if ((rc = dalloc (h, &p.rep_src, B * nr))) return rc;
    if ((rc = dalloc (h, &p.owner, B * m))) return rc;
    if ((rc = dalloc (h, &p.N, 2 * B * nr))) return rc;         // (the search's view of the lengths, then the lengths: ICP_N_FULL)
    if ((rc = dalloc (h, &p.O, B * nr))) return rc;
    if ((rc = dalloc (h, &p.perm, B * m))) return rc;
    if ((rc = dalloc (h, &p.chunk_hist, B * p.nchunk * nr))) return rc;
    if ((rc = dalloc (h, &p.blist, B * p.nb * 64))) return rc;
    if ((rc = dalloc (h, &p.bn, B * p.nb))) return rc;
    if ((rc = dalloc (h, &p.brank, B * m))) return rc;
    if ((rc = dalloc (h, &p.rid, B * m))) return rc;
    if ((rc = dalloc (h, &p.nn_id, B * m))) return rc;
    if ((rc = dalloc (h, &p.PF, B * m))) return rc;
    if ((rc = dalloc (h, &p.PM, B * m))) return rc;
    if ((rc = dalloc (h, &p.wpart, B * 2 * p.nwp))) return rc;      // two half-trees per group; padding stays 0.f
    if ((rc = dalloc (h, &p.mpart, B * 2 * p.nwg))) return rc;
    if ((rc = dalloc (h, &p.mscr, B * 2 * ((p.nwg + 127u) / 128u)))) return rc;
    if ((rc = dalloc (h, &p.spart, B * 11 * p.nsp * 8))) return rc;    // 8 sub-trees per work-group; padding stays 0.f
    if ((rc = dalloc (h, &p.mom, B * 2 * 18 * p.nb))) return rc;
    if ((rc = dalloc (h, &p.ml1, B * 18 * ((p.nb + 127u) / 128u)))) return rc;
    if ((rc = dalloc (h, &p.cst, B * 2))) return rc;
    if ((rc = dalloc (h, &p.st, B))) return rc;
    if ((rc = dalloc (h, &p.st_prev, B))) return rc;
    if (!h->dTin) HIPCHK (h, hipMalloc ((void **) &h->dTin, 8 * sizeof (float)));
    HIPCHK (h, hipHostMalloc ((void **) &h->hF, B * m * 8 * sizeof (float), hipHostMallocDefault));
    HIPCHK (h, hipHostMalloc ((void **) &h->hM, B * m * 8 * sizeof (float), hipHostMallocDefault));
    HIPCHK (h, hipHostMalloc ((void **) &h->hT, 64 * sizeof (float), hipHostMallocDefault));
    // (fine-grained: the device stores into these while the host polls them — run_ctl)
    HIPCHK (h, hipHostMalloc ((void **) &h->hState, B * sizeof (icp_reg_state), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK (h, hipHostMalloc ((void **) &h->hMirror, B * sizeof (unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset (h->hMirror, 0, B * sizeof (unsigned long long));
    icp_launch_reset_state (p, h->stream, 1);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));
    h->inited = true; h->built = false;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_init (icp_handle h, uint32_t m, uint32_t nr, float a, float c, uint32_t max_iterations,
              double angle_threshold, double translation_threshold) try
{
    api_guard guard_ (h);
    return icp_init_batched (h, 1, m, nr, a, c, max_iterations, angle_threshold, translation_threshold);
}
ICP_CATCH_ALL

int icp_write_b (icp_handle h, uint32_t b, int mem, const void *host_ptr, int block) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (b >= h->p.batch) return fail (h, ICP_EINVAL, "batch index out of range");
    if ((rc = set_device (h))) return rc;
    const size_t fm = (size_t) h->p.m * 8 * sizeof (float);
    switch (mem) {
        case ICP_MEM_F:
        case ICP_MEM_M: {
            float *stage = (mem == ICP_MEM_F ? h->hF : h->hM) + (size_t) b * h->p.m * 8;
            float *dst = (mem == ICP_MEM_F ? h->dF : h->dM) + (size_t) b * h->p.m * 8;
            // the staging buffer may still feed an earlier asynchronous copy of the same kind: wait for THAT copy (an event of a
            // never-recorded event returns at once), not for whatever else the stream holds (a run in flight keeps going)
            hipEvent_t ev = h->evStage[mem == ICP_MEM_F ? 0 : 1];
            note_inputs_change (h);
            HIPCHK (h, hipEventSynchronize (ev));
            if (host_ptr) std::memcpy (stage, host_ptr, fm);           // algorithms.cpp:4604-4606
            HIPCHK (h, hipMemcpyAsync (dst, stage, fm, hipMemcpyHostToDevice, h->stream));
            HIPCHK (h, hipEventRecord (ev, h->stream));
            break;
        }
        case ICP_MEM_T: {
            HIPCHK (h, hipEventSynchronize (h->evStage[2]));
            if (host_ptr) std::memcpy (h->hT, host_ptr, 8 * sizeof (float));   // :4613-4617
            HIPCHK (h, hipMemcpyAsync (h->dTin, h->hT, 8 * sizeof (float), hipMemcpyHostToDevice, h->stream));
            HIPCHK (h, hipEventRecord (h->evStage[2], h->stream));
            { const long long kb = h->k_base; note_enqueue (h); h->k_base = kb; }       // (T changes, the iteration count does not)
            icp_launch_set_T (h->p, b, h->dTin, h->stream);
            HIPCHK (h, hipGetLastError ());
            break;
        }
        default:
            return fail (h, ICP_EINVAL, "icp_write: mem must be ICP_MEM_F, ICP_MEM_M or ICP_MEM_T");
    }
    if (block) HIPCHK (h, hipStreamSynchronize (h->stream));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_write (icp_handle h, int mem, const void *host_ptr, int block) try { api_guard guard_ (h); return icp_write_b (h, 0, mem, host_ptr, block); } ICP_CATCH_ALL

size_t icp_mem_size (icp_handle h, int mem)
{
    if (!h || !h->inited) return 0;
    const icp_params &p = h->p;
    switch (mem) {
        case ICP_MEM_F: case ICP_MEM_M: case ICP_MEM_RBC_XP: return (size_t) p.m * 32;
        case ICP_MEM_T: case ICP_MEM_TK: case ICP_MEM_MEANS: return 32;
        case ICP_MEM_S: return 44;
        case ICP_MEM_NN_ID: return (size_t) p.m * 8;
        case ICP_MEM_W: case ICP_MEM_RBC_PERM: case ICP_MEM_RBC_OWNER: case ICP_MEM_RID: return (size_t) p.m * 4;
        case ICP_MEM_SUM_W: return 8;
        case ICP_MEM_REPS: return (size_t) p.nr * 32;
        case ICP_MEM_RBC_N: case ICP_MEM_RBC_O: return (size_t) p.nr * 4;
        case ICP_MEM_R: case ICP_MEM_RK: return 36;
        case ICP_MEM_NN: case ICP_MEM_QT: return (size_t) p.m * 16;
        default: return 0;
    }
}

static int mem_ptr (icp_context *h, uint32_t b, int mem, const void **src)
{
    const icp_params &p = h->p;
    const char *st = reinterpret_cast<const char *> (p.st + b);
    switch (mem) {
        case ICP_MEM_F: *src = h->dF + (size_t) b * p.m * 8; break;
        case ICP_MEM_M: *src = h->dM + (size_t) b * p.m * 8; break;
        case ICP_MEM_RBC_XP: *src = p.XP + (size_t) b * p.m * 8; break;
        case ICP_MEM_T: *src = st + offsetof (icp_reg_state, T); break;
        case ICP_MEM_TK: *src = st + offsetof (icp_reg_state, Tk); break;
        case ICP_MEM_MEANS: *src = st + offsetof (icp_reg_state, means); break;
        case ICP_MEM_S: *src = st + offsetof (icp_reg_state, S); break;
        case ICP_MEM_SUM_W: *src = st + offsetof (icp_reg_state, sum_w); break;
        case ICP_MEM_R: *src = st + offsetof (icp_reg_state, R); break;
        case ICP_MEM_RK: *src = st + offsetof (icp_reg_state, Rk); break;
        case ICP_MEM_NN_ID: *src = p.nn_id + (size_t) b * p.m; break;
        case ICP_MEM_RBC_PERM: *src = p.perm + (size_t) b * p.m; break;
        case ICP_MEM_RBC_OWNER: *src = p.owner + (size_t) b * p.m; break;
        case ICP_MEM_RID: *src = p.rid + (size_t) b * p.m; break;
        case ICP_MEM_REPS: *src = p.R + (size_t) b * p.nr * 8; break;
        case ICP_MEM_RBC_N: *src = ICP_N_FULL (p, b); break;
        case ICP_MEM_RBC_O: *src = p.O + (size_t) b * p.nr; break;
        case ICP_MEM_NN: *src = p.PF + (size_t) b * p.m; break;
        case ICP_MEM_QT: *src = p.PM + (size_t) b * p.m; break;
        case ICP_MEM_W: *src = reinterpret_cast<const float *> (p.PF + (size_t) b * p.m) + 3; break;
        default: return fail (h, ICP_EINVAL, "unknown icp_mem value");
    }
    return ICP_OK;
}

int icp_read_b (icp_handle h, uint32_t b, int mem, void *host_dst, size_t bytes) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!host_dst) return fail (h, ICP_EINVAL, "icp_read: null destination");
    if (b >= h->p.batch) return fail (h, ICP_EINVAL, "batch index out of range");
    size_t full = icp_mem_size (h, mem);
    if (full == 0) return fail (h, ICP_EINVAL, "unknown icp_mem value");
    if (bytes > full) return fail (h, ICP_EINVAL, "icp_read: more bytes requested than the object holds");
    if ((rc = set_device (h))) return rc;
    const void *src = nullptr;
    if ((rc = mem_ptr (h, b, mem, &src))) return rc;
    if ((rc = materialize_outputs (h, mem))) return rc;
    if (mem == ICP_MEM_W) {                        // weights live in the .w lane of the matched points
        size_t rows = bytes / 4;
        HIPCHK (h, hipMemcpy2DAsync (host_dst, 4, src, 16, 4, rows, hipMemcpyDeviceToHost, h->stream));
    } else
        HIPCHK (h, hipMemcpyAsync (host_dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_read (icp_handle h, int mem, void *host_dst, size_t bytes) try { api_guard guard_ (h); return icp_read_b (h, 0, mem, host_dst, bytes); } ICP_CATCH_ALL

int icp_device_ptr (icp_handle h, int mem, void **dptr) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!dptr) return fail (h, ICP_EINVAL, "null pointer");
    const void *src = nullptr;
    if ((rc = mem_ptr (h, 0, mem, &src))) return rc;
    if ((rc = set_device (h))) return rc;
    if ((rc = materialize_outputs (h, mem))) return rc;
    *dptr = const_cast<void *> (src);
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_adopt_device_buffer (icp_handle h, int mem, void *dptr) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!dptr) return fail (h, ICP_EINVAL, "null pointer");
    if (mem == ICP_MEM_F) { h->dF = static_cast<float *> (dptr); h->p.F = h->dF; h->ownF = false; h->built = false; }
    else if (mem == ICP_MEM_M) { h->dM = static_cast<float *> (dptr); h->p.M = h->dM; h->ownM = false; }
    else return fail (h, ICP_EINVAL, "only ICP_MEM_F and ICP_MEM_M can be adopted");
    note_inputs_change (h);
    drop_graphs (h);
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_build_rbc (icp_handle h) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    note_inputs_change (h);
    // The two (latency-bound sizes) to six launches of the construction are enqueued as they are: a graph of so few nodes costs more
    // at its head and tail than it saves between them — same box, back to back, graph against plain launches: A 22.3 -> 13.6 us,
    // B 46.9 -> 39.8, C 193 -> 185, A x 64 115 -> 106 us (ICP_AMD_BUILD_GRAPH=1 brings the cached graph back for the comparison).
    {
        static const char *e = std::getenv ("ICP_AMD_BUILD_GRAPH");
        const bool direct = !(e && e[0] == '1');
        if (direct) {
            note_enqueue (h);
            icp_launch_build_rbc (h->p, h->stream);
            HIPCHK (h, hipGetLastError ());
            h->built = true; h->k_base = 0;                                 // (ICP::buildRBC resets k, :4796)
            return ICP_OK;
        }
    }
    // the five or six launches of the construction as one cached graph (key: all ones; dropped with the others when a
    // parameter or a buffer changes)
    const uint64_t key = ~0ull - h->parity;
    auto it = h->graphs.find (key);
    if (it == h->graphs.end ()) {
        graph_entry ge;
        if ((rc = capture_graph (h, [&] { icp_launch_build_rbc (h->p, h->stream); }, &ge))) return rc;     // (the placement kernel also sets k = 0: ICP::buildRBC, :4796)
        it = h->graphs.emplace (key, ge).first;
    }
    note_enqueue (h);
    HIPCHK (h, hipGraphLaunch (it->second.exec, h->stream));
    h->built = true; h->k_base = 0;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_step (icp_handle h, int config) try
{
    api_guard guard_ (h);
    (void) config;   // the reference sizes the list-scan launch from a host read when config is set; nothing to configure here
    int rc = need (h, true); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    icp_params p = h->p; p.check = 0; p.hmirror = nullptr; p.hstate = nullptr;
    { const long long kb = h->k_base; note_enqueue (h); if (kb >= 0) h->k_base = kb + 1; }
    note_outputs_stored (h);
    icp_launch_iteration (p, h->stream);
    HIPCHK (h, hipGetLastError ());
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_run_fixed (icp_handle h, uint32_t iterations) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (iterations == 0) return ICP_OK;
    if ((rc = set_device (h))) return rc;
    return launch_run (h, iterations, 0);
}
ICP_CATCH_ALL

int icp_run_fixed_fresh (icp_handle h, uint32_t iterations) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (iterations == 0) return icp_reset_transform (h);
    if ((rc = set_device (h))) return rc;
    return launch_run (h, iterations, 0, true);
}
ICP_CATCH_ALL

int icp_run (icp_handle h, uint32_t *k) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    if (!h->run_adaptive) {                                              // rounds 1 - 3: one graph of max_iterations launches
        if ((rc = launch_run (h, h->max_iterations, 1))) return rc;
        if ((rc = settle (h))) return rc;                                // queue.finish () — :4813
        h->stat_launches = h->max_iterations; h->stat_k = h->hState[0].k; h->stat_dead = 0;
    } else {
        // the host loop of the reference (:4806-4814: run one step, check (), stop), with the check on the device and the host `run_depth`
        // launches ahead of it: the calling thread polls the registration's progress word and tops the queue up
        if ((rc = run_begin (h, h->run, h->stream, h->p, false, false, h->run_depth + 1u, h->hMirror, h->hState, -1))) return rc;
        if ((rc = run_finish (h))) return rc;
        if ((rc = run_wait_final (h, h->hMirror, h->p.batch, h->run.p.epoch))) return rc;      // (the end kernel is the last thing on the stream: queue.finish ())
        h->stat_t[5] = now_s ();
        h->hstate_fresh = true; h->hstate_here = true;
        if (h->p.batch == 1u) h->k_base = h->hState[0].k;
        {   // the statistics of the run, now that its outcome is known (a run whose launches all went out at once was "decided" before it ran)
            uint32_t kmax = 0u, all_done = 1u;
            for (uint32_t b = 0; b < h->p.batch; ++b) { kmax = std::max (kmax, h->hState[b].k); all_done &= h->hState[b].done ? 1u : 0u; }
            h->stat_k = kmax;
            const uint32_t ran = kmax > h->run.k0 ? kmax - h->run.k0 : 0u;
            h->stat_dead = all_done ? h->run.enq - std::min (h->run.enq, ran + 1u) : 0u;
        }
    }
    if (k) {
        if (h->hstate_fresh) *k = h->hState[0].k;                        // (the run left the states in the pinned mirror)
        else {
            icp_reg_state st;
            HIPCHK (h, hipMemcpy (&st, h->p.st, sizeof st, hipMemcpyDeviceToHost));
            *k = st.k;
        }
    }
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_run_stats (icp_handle h, uint32_t *launches, uint32_t *k, uint32_t *dead) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (launches) *launches = h->stat_launches;
    if (k) *k = h->stat_k;
    if (dead) *dead = h->stat_dead;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_launch_stats (icp_handle h, double *max_us, uint64_t *slower_than_10us, uint64_t *total, int reset) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (max_us) *max_us = h->stat_launch_max_us;
    if (slower_than_10us) *slower_than_10us = h->stat_launch_slow;
    if (total) *total = h->stat_launch_total;
    if (reset) { h->stat_launch_max_us = 0.0; h->stat_launch_slow = h->stat_launch_total = 0; }
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_set_output_mode (icp_handle h, int mode) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (mode != ICP_OUTPUTS_LAZY && mode != ICP_OUTPUTS_EVERY_ITERATION) return fail (h, ICP_EINVAL, "unknown output mode");
    h->outputs_lazy = mode == ICP_OUTPUTS_LAZY;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_run_timeline (icp_handle h, double *us6) try
{
    api_guard guard_ (h);
    if (!h || !us6) return ICP_EINVAL;
    for (int i = 0; i < 6; ++i) us6[i] = (h->stat_t[i] - h->stat_t[0]) * 1e6;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_set_run_depth (icp_handle h, uint32_t depth, int adaptive) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (depth == 0 || depth > 64u) return fail (h, ICP_EINVAL, "icp_set_run_depth: depth must be in [1, 64]");
    { int rc = set_device (h); if (rc) return rc; if ((rc = run_close_all (h))) return rc; }
    h->run_depth = depth; h->run_adaptive = adaptive ? 1 : 0;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_sync (icp_handle h) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    int rc = set_device (h); if (rc) return rc;
    return settle (h);
}
ICP_CATCH_ALL

int icp_get_alpha (icp_handle h, float *a) try { api_guard guard_ (h); if (!h || !a) return ICP_EINVAL; *a = h->p.a; return ICP_OK; } ICP_CATCH_ALL
// The setters change a number in the handle's parameters and nothing else: checked runs are plain launches that read the parameters as they
// are, and a cached fixed-length graph of an older parameter generation is updated in place when it is next used (get_graph).
int icp_set_alpha (icp_handle h, float a) try
{   // setAlpha updates construct and search (src/ICP/algorithms.cpp:4712-4717); lists must be rebuilt by the caller
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (a == 0.f) return fail (h, ICP_EINVAL, "The alpha parameter cannot be equal to zero");
    { int rc = outputs_before_change (h); if (rc) return rc; }
    h->p.a = a; ++h->param_gen; return ICP_OK;
}
ICP_CATCH_ALL
int icp_set_metric_scale (icp_handle h, float f_g) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (!(f_g > 0.f) || !std::isfinite (f_g)) return fail (h, ICP_EINVAL, "the metric scale must be positive and finite");
    { int rc = outputs_before_change (h); if (rc) return rc; }
    h->p.dist_scale = f_g; ++h->param_gen; return ICP_OK;
}
ICP_CATCH_ALL
int icp_get_metric_scale (icp_handle h, float *f_g) try { api_guard guard_ (h); if (!h || !f_g) return ICP_EINVAL; *f_g = h->p.dist_scale; return ICP_OK; } ICP_CATCH_ALL
int icp_get_scaling (icp_handle h, float *c) try { api_guard guard_ (h); if (!h || !c) return ICP_EINVAL; *c = h->p.c; return ICP_OK; } ICP_CATCH_ALL
int icp_set_scaling (icp_handle h, float c) try { api_guard guard_ (h); if (!h) return ICP_EINVAL; h->p.c = c; ++h->param_gen; return ICP_OK; } ICP_CATCH_ALL
int icp_get_max_iterations (icp_handle h, uint32_t *n) try { api_guard guard_ (h); if (!h || !n) return ICP_EINVAL; *n = h->max_iterations; return ICP_OK; } ICP_CATCH_ALL
int icp_set_max_iterations (icp_handle h, uint32_t n) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (n == 0) return fail (h, ICP_EINVAL, "max_iterations must be positive");
    h->max_iterations = n; return ICP_OK;
}
ICP_CATCH_ALL
int icp_get_angle_threshold (icp_handle h, double *d) try { api_guard guard_ (h); if (!h || !d) return ICP_EINVAL; *d = h->angle_threshold; return ICP_OK; } ICP_CATCH_ALL
int icp_set_angle_threshold (icp_handle h, double d) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    h->angle_threshold = d; h->p.tan_half_thr = std::tan (d * M_PI / 360.0); ++h->param_gen; return ICP_OK;
}
ICP_CATCH_ALL
int icp_get_translation_threshold (icp_handle h, double *d) try { api_guard guard_ (h); if (!h || !d) return ICP_EINVAL; *d = h->translation_threshold; return ICP_OK; } ICP_CATCH_ALL
int icp_set_translation_threshold (icp_handle h, double d) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    h->translation_threshold = d; h->p.trans_thr = d; ++h->param_gen; return ICP_OK;
}
ICP_CATCH_ALL
int icp_set_power_mode (icp_handle h, int mode) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (mode != ICP_POWER_LITERAL && mode != ICP_POWER_SQUARED) return fail (h, ICP_EINVAL, "unknown power mode");
    h->p.power_mode = mode; ++h->param_gen; return ICP_OK;
}
ICP_CATCH_ALL

int icp_set_reduce_mode (icp_handle h, int mode) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (mode != ICP_REDUCE_REFERENCE_ORDER && mode != ICP_REDUCE_FUSED) return fail (h, ICP_EINVAL, "unknown reduce mode");
    { int rc = outputs_before_change (h); if (rc) return rc; }
    h->p.fused = mode; drop_graphs (h); return ICP_OK;
}
ICP_CATCH_ALL

int icp_state_b (icp_handle h, uint32_t b, icp_state_t *out) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!out) return fail (h, ICP_EINVAL, "null pointer");
    if (b >= h->p.batch) return fail (h, ICP_EINVAL, "batch index out of range");
    if ((rc = set_device (h))) return rc;
    icp_reg_state st;
    if (h->hstate_fresh) {                       // a checked run was the last thing that changed the states: its end left them in the mirror
        if (!h->hstate_here) HIPCHK (h, hipStreamSynchronize (h->stream));
        st = h->hState[b];
    } else {
        HIPCHK (h, hipMemcpyAsync (&st, h->p.st + b, sizeof st, hipMemcpyDeviceToHost, h->stream));
        HIPCHK (h, hipStreamSynchronize (h->stream));
    }
    std::memcpy (out->R, st.R, sizeof st.R); std::memcpy (out->Rk, st.Rk, sizeof st.Rk);
    std::memcpy (out->q, st.T, 16); std::memcpy (out->t, st.T + 4, 12); out->s = st.T[7];
    std::memcpy (out->qk, st.Tk, 16); std::memcpy (out->tk, st.Tk + 4, 12); out->sk = st.Tk[7];
    out->k = st.k; out->converged = st.done; out->power_iterations = st.pm_iters; out->reserved = 0;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_state (icp_handle h, icp_state_t *out) try { api_guard guard_ (h); return icp_state_b (h, 0, out); } ICP_CATCH_ALL

int icp_write_cloud (icp_handle h, int which, const void *cloud, int block) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (h->p.m != 16384u) return fail (h, ICP_EINVAL, "getLMs produces 128 x 128 landmarks: m must be 16384");
    if (which != ICP_MEM_F && which != ICP_MEM_M) return fail (h, ICP_EINVAL, "which must be ICP_MEM_F or ICP_MEM_M");
    if (!cloud) return fail (h, ICP_EINVAL, "null pointer");
    if ((rc = set_device (h))) return rc;
    const uint32_t n = 640u * 480u;
    if (h->cloud_cap < n) {
        if (h->dCloud) (void) hipFree (h->dCloud);
        if (h->dCloudOut) (void) hipFree (h->dCloudOut);
        h->dCloud = h->dCloudOut = nullptr; h->cloud_cap = 0;
        HIPCHK (h, hipMalloc ((void **) &h->dCloud, (size_t) n * 32));
        HIPCHK (h, hipMalloc ((void **) &h->dCloudOut, (size_t) n * 32));
        h->cloud_cap = n;
    }
    note_inputs_change (h);
    HIPCHK (h, hipMemcpyAsync (h->dCloud, cloud, (size_t) n * 32, hipMemcpyHostToDevice, h->stream));
    icp_launch_get_lms (h->dCloud, which == ICP_MEM_F ? h->dF : h->dM, h->stream);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipStreamSynchronize (h->stream));   // the source is pageable host memory
    (void) block;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_transform_cloud (icp_handle h, const void *host_in, void *host_out, uint32_t n) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!host_in || !host_out || n == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    if (h->cloud_cap < n) {
        if (h->dCloud) (void) hipFree (h->dCloud);
        if (h->dCloudOut) (void) hipFree (h->dCloudOut);
        h->dCloud = h->dCloudOut = nullptr; h->cloud_cap = 0;
        HIPCHK (h, hipMalloc ((void **) &h->dCloud, (size_t) n * 32));
        HIPCHK (h, hipMalloc ((void **) &h->dCloudOut, (size_t) n * 32));
        h->cloud_cap = n;
    }
    HIPCHK (h, hipMemcpyAsync (h->dCloud, host_in, (size_t) n * 32, hipMemcpyHostToDevice, h->stream));
    icp_launch_transform_cloud (h->dCloud, h->dCloudOut, h->p.st, n, h->stream);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipMemcpyAsync (host_out, h->dCloudOut, (size_t) n * 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return ICP_OK;
}
ICP_CATCH_ALL


int icp_transform_cloud_ex (icp_handle h, int kind, const float *T, const void *host_in, void *host_out, uint32_t n) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (kind != ICP_TRANSFORM_QUATERNION && kind != ICP_TRANSFORM_QUATERNION_2 && kind != ICP_TRANSFORM_MATRIX)
        return fail (h, ICP_EINVAL, "icp_transform_cloud_ex: unknown transformation kind");
    if (!T || !host_in || !host_out || n == 0) return fail (h, ICP_EINVAL, "bad arguments");
    int rc = set_device (h); if (rc) return rc;
    if (h->cloud_cap < n) {
        if (h->dCloud) (void) hipFree (h->dCloud);
        if (h->dCloudOut) (void) hipFree (h->dCloudOut);
        h->dCloud = h->dCloudOut = nullptr; h->cloud_cap = 0;
        HIPCHK (h, hipMalloc ((void **) &h->dCloud, (size_t) n * 32));
        HIPCHK (h, hipMalloc ((void **) &h->dCloudOut, (size_t) n * 32));
        h->cloud_cap = n;
    }
    HIPCHK (h, hipMemcpyAsync (h->dCloud, host_in, (size_t) n * 32, hipMemcpyHostToDevice, h->stream));
    icp_launch_transform_cloud_ex (kind, h->dCloud, h->dCloudOut, T, n, h->stream);
    HIPCHK (h, hipGetLastError ());
    HIPCHK (h, hipMemcpyAsync (host_out, h->dCloudOut, (size_t) n * 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_power_method (int device, int rot, int power_mode, const float *S11, const float *means8, float *Tk8, float *Rk9, uint32_t *iters) try
{
    if (!S11 || !means8 || !Tk8) return fail (nullptr, ICP_EINVAL, "icp_power_method: null pointer");
    if ((rot != ICP_ROT_EIGEN && rot != ICP_ROT_POWER_METHOD) || (power_mode != ICP_POWER_LITERAL && power_mode != ICP_POWER_SQUARED))
        return fail (nullptr, ICP_EINVAL, "icp_power_method: rot must be 0|1 and power_mode 0|1");
    int count = 0;
    if (hipGetDeviceCount (&count) != hipSuccess || count <= 0)
        return fail (nullptr, ICP_ENODEVICE, "icp_power_method: no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= count) return fail (nullptr, ICP_EINVAL, "icp_power_method: device ordinal out of range");
    hipDeviceProp_t prop;
    HIPCHK (nullptr, hipGetDeviceProperties (&prop, device));
    if (std::strncmp (prop.gcnArchName, "gfx950", 6) != 0)
        return fail (nullptr, ICP_ENODEVICE, std::string ("icp_power_method: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    HIPCHK (nullptr, hipSetDevice (device));
    float *d = nullptr;
    HIPCHK (nullptr, hipMalloc ((void **) &d, (19 + 18) * sizeof (float)));
    float in[19], out[18];
    std::memcpy (in, S11, 11 * sizeof (float)); std::memcpy (in + 11, means8, 8 * sizeof (float));
    hipError_t e = hipMemcpy (d, in, sizeof in, hipMemcpyHostToDevice);
    if (e == hipSuccess) { icp_launch_rotation_solver (rot, power_mode, d, d + 19, nullptr); e = hipGetLastError (); }
    if (e == hipSuccess) e = hipMemcpy (out, d + 19, sizeof out, hipMemcpyDeviceToHost);      // (blocking: waits for the kernel on the null stream)
    (void) hipFree (d);
    if (e != hipSuccess) return fail (nullptr, ICP_EHIP, std::string ("icp_power_method: ") + hipGetErrorString (e));
    std::memcpy (Tk8, out, 8 * sizeof (float));
    if (Rk9) std::memcpy (Rk9, out + 8, 9 * sizeof (float));
    if (iters) std::memcpy (iters, out + 17, sizeof (uint32_t));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_reset_transform (icp_handle h) try
{   // T <- identity, k <- 0 (what ICPStep::init uploads, src/ICP/algorithms.cpp:4486-4493); enqueue only
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    note_enqueue (h);
    icp_launch_reset_state (h->p, h->stream, 1);
    HIPCHK (h, hipGetLastError ());
    h->k_base = 0;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_time_run_fixed (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_total) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (!ms_total || iterations == 0 || reps == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    hipGraphExec_t exec;
    if ((rc = get_graph (h, iterations, 0, &exec, from_identity != 0))) return rc;     // from_identity: every pass is a fresh registration
    HIPCHK (h, hipEventRecord (h->ev0, h->stream));
    for (uint32_t r = 0; r < reps; ++r) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    h->hstate_fresh = false; h->k_base = -1; note_outputs_stored (h);
    HIPCHK (h, hipEventRecord (h->ev1, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));                       // (not hipEventSynchronize: its wake-up now and then takes 0.5 ms, tools/diag/overhead.py)
    HIPCHK (h, hipEventElapsedTime (ms_total, h->ev0, h->ev1));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_time_run_fixed_tail (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_timed, uint32_t *reps_timed) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (!ms_timed || !reps_timed || iterations == 0 || reps == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    hipGraphExec_t exec;
    if ((rc = get_graph (h, iterations, 0, &exec, from_identity != 0))) return rc;
    // the first event goes in BEHIND the first pass: a marker recorded on an idle stream delays the graph launched right after it by
    // 0.1 - 0.25 ms (7.22 - 7.39 ms for 20 passes of 0.357 ms against 7.17 ms wall-clock for the same launches without events;
    // with the GPU busy when the marker arrives: 7.14 ms, tools/diag/overhead.py); the events then bracket the passes 2 .. reps
    const uint32_t lead = reps >= 2u ? 1u : 0u;
    if (lead) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    HIPCHK (h, hipEventRecord (h->ev0, h->stream));
    for (uint32_t r = lead; r < reps; ++r) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    h->hstate_fresh = false; h->k_base = -1; note_outputs_stored (h);
    HIPCHK (h, hipEventRecord (h->ev1, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));                       // (not hipEventSynchronize: its wake-up now and then takes 0.5 ms, tools/diag/overhead.py)
    HIPCHK (h, hipEventElapsedTime (ms_timed, h->ev0, h->ev1));
    *reps_timed = reps - lead;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_run_form (icp_handle h, int *form) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!form) return fail (h, ICP_EINVAL, "null output");
    if (icp_chain_supported (h->p)) *form = ICP_FORM_CHAINED;
    else *form = ICP_FORM_SEPARATE;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_search_layout (icp_handle h, int *dense, int *tile, int *stage2) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    icp_search_layout_of (h->p, dense, tile, stage2);
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_launches_per_iteration (icp_handle h, uint32_t *n) try
{
    api_guard guard_ (h);
    int rc = need (h, false); if (rc) return rc;
    if (!n) return fail (h, ICP_EINVAL, "null output");
    int form = ICP_FORM_SEPARATE;
    if ((rc = icp_run_form (h, &form))) return rc;
    // (fused, large sets: the first level of the moment tree is a launch of its own — icp_launch_finalize)
    *n = form != ICP_FORM_SEPARATE ? 1u : h->p.fused ? ((h->p.nb + 127u) / 128u > ICP_L1_MIN_GROUPS ? 3u : 2u) : 4u;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_time_masked (icp_handle h, uint32_t mask, uint32_t iterations, uint32_t reps, float *ms_total) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (!ms_total || iterations == 0 || reps == 0 || mask == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    icp_params p = h->p; p.check = 0; p.hmirror = nullptr; p.hstate = nullptr;
    note_enqueue (h); note_outputs_stored (h);                          // (the masked graphs change the device state: the pinned mirror is stale)
    graph_entry ge;
    // (the per-query outputs follow the policy of the fixed-length graphs: stored by the last iteration only in fused mode)
    if ((rc = capture_graph (h, [&] {
             for (uint32_t k = 0; k < iterations; ++k) { p.emit = (k + 1 == iterations) ? 1 : 0; icp_launch_masked (p, h->stream, mask); }
         }, &ge))) return rc;
    hipError_t e = hipGraphLaunch (ge.exec, h->stream);                 // warm-up
    if (e == hipSuccess) e = hipEventRecord (h->ev0, h->stream);
    for (uint32_t r = 0; r < reps && e == hipSuccess; ++r) e = hipGraphLaunch (ge.exec, h->stream);
    if (e == hipSuccess) e = hipEventRecord (h->ev1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize (h->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime (ms_total, h->ev0, h->ev1);
    (void) hipGraphExecDestroy (ge.exec); (void) hipGraphDestroy (ge.graph);
    if (e != hipSuccess) return fail (h, ICP_EHIP, std::string ("icp_time_masked: ") + hipGetErrorString (e));
    return ICP_OK;
}
ICP_CATCH_ALL

#ifdef ICP_DBG_STAMPS
__attribute__ ((visibility ("default")))          // (diagnostic builds only: not part of the ABI)
#endif
int icp_debug_stamps (icp_handle h, unsigned long long *out, uint32_t nblocks) try
{   // diagnostic builds (ICP_DBG_STAMPS): one k_search launch, per-block s_memtime stamps
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (!out || nblocks == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    unsigned long long *d = nullptr;
    HIPCHK (h, hipMalloc ((void **) &d, (size_t) nblocks * 16 * 8));
    hipError_t e = hipMemset (d, 0, (size_t) nblocks * 16 * 8);
    icp_params p = h->p; p.check = 0; p.dbg = d; p.hmirror = nullptr; p.hstate = nullptr;
    note_enqueue (h); note_outputs_stored (h);
    if (e == hipSuccess) {
        if (icp_chain_supported (p)) icp_launch_chain (p, h->stream, 2);
        else { icp_launch_search (p, h->stream); if (p.fused) icp_launch_finalize (p, h->stream); }
        e = hipGetLastError ();
    }
    if (e == hipSuccess) e = hipStreamSynchronize (h->stream);
    if (e == hipSuccess) e = hipMemcpy (out, d, (size_t) nblocks * 16 * 8, hipMemcpyDeviceToHost);
    (void) hipFree (d);
    if (e != hipSuccess) return fail (h, ICP_EHIP, std::string ("icp_debug_stamps: ") + hipGetErrorString (e));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_profile_run (icp_handle h, uint32_t iterations, float *out_ms, float *total_ms) try
{
    api_guard guard_ (h);
    int rc = need (h, true); if (rc) return rc;
    if (!out_ms || iterations == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    icp_params p = h->p; p.check = 0; p.emit = 1; p.hmirror = nullptr; p.hstate = nullptr;
    note_enqueue (h); note_outputs_stored (h);
    std::vector<hipEvent_t> ev ((size_t) iterations * 5, nullptr);
    hipError_t e = hipSuccess;
    for (auto &x : ev) if (e == hipSuccess) e = hipEventCreate (&x);
    // the stages as separate launches (the chained form has no stage boundaries to time), events around each
    for (uint32_t r = 0; r < iterations && e == hipSuccess; ++r) {
        hipEvent_t *x = &ev[(size_t) r * 5];
        e = hipEventRecord (x[0], h->stream); icp_launch_search (p, h->stream);
        if (e == hipSuccess) e = hipEventRecord (x[1], h->stream);
        if (!p.fused) icp_launch_means (p, h->stream);
        if (e == hipSuccess) e = hipEventRecord (x[2], h->stream);
        if (!p.fused) icp_launch_sij (p, h->stream);
        if (e == hipSuccess) e = hipEventRecord (x[3], h->stream);
        icp_launch_finalize (p, h->stream);
        if (e == hipSuccess) e = hipEventRecord (x[4], h->stream);
    }
    if (e == hipSuccess) e = hipGetLastError ();
    if (e == hipSuccess) e = hipStreamSynchronize (h->stream);
    for (uint32_t r = 0; r < iterations && e == hipSuccess; ++r)
        for (int k = 0; k < 4 && e == hipSuccess; ++k)
            e = hipEventElapsedTime (&out_ms[(size_t) r * 4 + k], ev[(size_t) r * 5 + k], ev[(size_t) r * 5 + k + 1]);
    if (e == hipSuccess && total_ms) e = hipEventElapsedTime (total_ms, ev[0], ev[(size_t) iterations * 5 - 1]);
    for (auto &x : ev) if (x) (void) hipEventDestroy (x);
    if (e != hipSuccess) return fail (h, ICP_EHIP, std::string ("icp_profile_run: ") + hipGetErrorString (e));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_time_kernels (icp_handle h, uint32_t reps, float *out_ms4) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    if (!out_ms4 || reps == 0) return fail (h, ICP_EINVAL, "bad arguments");
    std::vector<float> t ((size_t) reps * 4);
    int rc = icp_profile_run (h, reps, t.data (), nullptr);
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) {
        double acc = 0.0;
        for (uint32_t r = 0; r < reps; ++r) acc += t[(size_t) r * 4 + k];
        out_ms4[k] = (float) (acc / reps);
    }
    return ICP_OK;
}
ICP_CATCH_ALL

}  // extern "C"
