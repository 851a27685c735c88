// icp_track.hip — frame-to-frame tracking behind the C-ABI (icp_track_*): what the reference's registration example does per camera
// frame (src/ocl_icp_reg.cpp:128-172: getLMs of the new frame, buildRBC on the previous one, ICP::run), kept on the device across
// frames, with the next frame's upload, landmark extraction and RBC construction enqueued while the current registration runs.
#include "icp_host.h"

using namespace icp_host;

extern "C" {

// ---- frame-to-frame tracking ---------------------------------------------------------------------------------------------------
// Frame f's landmarks live in lm[f mod 3]; registration f (frame f onto frame f - 1) reads lm[f mod 3] as the moving and
// lm[(f - 1) mod 3] as the fixed set, so frame f + 1 can be uploaded and its landmarks extracted (copy stream) while registration f
// runs (main stream): the buffer it goes to was last read by registration f - 1.  Two staging slots (f mod 2) hold the band of
// a frame (the 2.08 MB of its 9.83 MB that getLMs reads) in pinned memory.  A frame's registration is buildRBC + a host-driven checked
// run (icp_run.hip) whose final state the device stores into the frame's slot of a pinned ring (hTrack, words in hTrackMirror);
// consecutive registrations alternate between two streams, each behind a one-wave gate kernel that waits for its predecessor's release
// of the sequence word (track_submit).  ICP_AMD_RUN_ADAPTIVE=0 keeps rounds 1 - 3's form: one graph per frame on one stream.
#define ICP_TRACK_RING 4u

static int track_prepare (icp_context *h)
{
    if (h->p.m != 16384u || h->p.batch != 1u) return fail (h, ICP_EINVAL, "tracking needs m == 16384 (getLMs) and a single registration");
    if (!h->ownF || !h->ownM) return fail (h, ICP_ESTATE, "tracking rotates the handle's own landmark buffers: not available with adopted F / M buffers");
    if (!h->lm[2]) HIPCHK (h, hipMalloc ((void **) &h->lm[2], (size_t) h->p.m * 8 * sizeof (float)));
    for (int k = 0; k < 2; ++k) {
        if (!h->hBand[k]) HIPCHK (h, hipHostMalloc ((void **) &h->hBand[k], ICP_BAND_BYTES, hipHostMallocDefault));
        if (!h->dBand[k]) HIPCHK (h, hipMalloc ((void **) &h->dBand[k], ICP_BAND_BYTES));
    }
    if (!h->hTrack) HIPCHK (h, hipHostMalloc ((void **) &h->hTrack, ICP_TRACK_RING * sizeof (icp_reg_state), hipHostMallocMapped | hipHostMallocCoherent));
    if (!h->hTrackMirror) {
        HIPCHK (h, hipHostMalloc ((void **) &h->hTrackMirror, ICP_TRACK_RING * sizeof (unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset (h->hTrackMirror, 0, ICP_TRACK_RING * sizeof (unsigned long long));
    }
    if (h->run_adaptive && h->track_gate && !h->rbc2_ready) {
        // frames gated on the device: a second stream, the sequence word and the run flags, a second set of RBC buffers (frame f builds its
        // RBC while frame f - 1 is still searching its own)
        const icp_params &p = h->p;
        if (!h->stream2) HIPCHK (h, hipStreamCreateWithFlags (&h->stream2, hipStreamNonBlocking));
        if (!h->dSeq) { HIPCHK (h, hipMalloc ((void **) &h->dSeq, sizeof (uint32_t))); HIPCHK (h, hipMemset (h->dSeq, 0, sizeof (uint32_t))); }
        // (one flag per stream: a run's flag must stay what it is until the last of that run's launches has gone through — the NEXT frame, on
        // the other stream, may converge while launches of this one are still queued; the frame after that is behind them on this stream)
        if (!h->dRunFlag) { HIPCHK (h, hipMalloc ((void **) &h->dRunFlag, 2 * sizeof (uint32_t))); HIPCHK (h, hipMemset (h->dRunFlag, 0, 2 * sizeof (uint32_t))); }
        if (!h->hGateFlag) { HIPCHK (h, hipHostMalloc ((void **) &h->hGateFlag, sizeof (uint32_t), hipHostMallocMapped | hipHostMallocCoherent)); *h->hGateFlag = 0u; }
        icp_context::rbc_set &a = h->rbc[0], &b = h->rbc[1];
        a.R = p.R; a.GB = p.GB; a.OL = p.OL; a.LB = p.LB; a.XP = p.XP; a.XQ = p.XQ; a.rep_src = p.rep_src; a.owner = p.owner; a.N = p.N; a.O = p.O; a.perm = p.perm;
        a.chunk_hist = p.chunk_hist; a.blist = p.blist; a.bn = p.bn; a.brank = p.brank;
        auto al = [&] (void **q, size_t bytes) -> int {
            hipError_t e = hipMalloc (q, bytes ? bytes : 1);
            if (e == hipSuccess) e = hipMemset (*q, 0, bytes ? bytes : 1);
            return e == hipSuccess ? ICP_OK : fail (h, ICP_ENOMEM, std::string ("tracking (second RBC set): ") + hipGetErrorString (e));
        };
        int rc;
        if (!b.R && ((rc = al ((void **) &b.R, (size_t) p.nr * 32)) || (rc = al ((void **) &b.GB, (size_t) 2 * (p.n16 + p.n1k) * 16)) || (rc = al ((void **) &b.LB, (size_t) 3 * p.nlb * 16)) || (rc = al ((void **) &b.OL, (size_t) ICP_OL_STRIDE (p.nr) * 16)) || (rc = al ((void **) &b.XP, (size_t) p.m * 32)) ||
            (rc = al ((void **) &b.XQ, (size_t) p.m * 32)) || (rc = al ((void **) &b.rep_src, (size_t) p.nr * 4)) || (rc = al ((void **) &b.owner, (size_t) p.m * 4)) ||
            (rc = al ((void **) &b.N, (size_t) 2 * p.batch * p.nr * 4)) || (rc = al ((void **) &b.O, (size_t) p.nr * 4)) || (rc = al ((void **) &b.perm, (size_t) p.m * 4)) ||
            (rc = al ((void **) &b.chunk_hist, (size_t) p.nchunk * p.nr * 4)) || (rc = al ((void **) &b.blist, (size_t) p.nb * 64 * 8)) ||
            (rc = al ((void **) &b.bn, (size_t) p.nb * 4)) || (rc = al ((void **) &b.brank, (size_t) p.m)))) {
            void *ptrs[] = { b.R, b.GB, b.OL, b.LB, b.XP, b.XQ, b.rep_src, b.owner, b.N, b.O, b.perm, b.chunk_hist, b.blist, b.bn, b.brank };
            for (void *x : ptrs) if (x) (void) hipFree (x);
            b = icp_context::rbc_set {};
            return rc;
        }
        // Do the two streams really run side by side?  HIP spreads streams over a few hardware queues; two streams that share one are served in
        // order, and a gate would then hold back the very launches it is waiting for.  One probe at set-up: a short-lived gate on stream2
        // waits for a word that a kernel on the handle's own stream sets.  If the gate gives up (2 ms), gating stays off for this handle.
        *h->hGateFlag = 0u;
        icp_launch_gate (h->dSeq, 1u, h->hGateFlag, h->stream2, 1u << 13);
        icp_launch_seq_set (h->dSeq, 1u, h->stream);
        HIPCHK (h, hipGetLastError ());
        HIPCHK (h, hipStreamSynchronize (h->stream2));
        HIPCHK (h, hipStreamSynchronize (h->stream));
        if (*h->hGateFlag) { h->track_gate = 0; *h->hGateFlag = 0u; }
        HIPCHK (h, hipMemset (h->dSeq, 0, sizeof (uint32_t)));
        h->rbc2_ready = true;
    }
    return ICP_OK;
}

static void rbc_into (icp_params &p, const icp_context::rbc_set &q)
{
    p.R = q.R; p.GB = q.GB; p.OL = q.OL; p.LB = q.LB; p.XP = q.XP; p.XQ = q.XQ; p.rep_src = q.rep_src; p.owner = q.owner; p.N = q.N; p.O = q.O; p.perm = q.perm;
    p.chunk_hist = q.chunk_hist; p.blist = q.blist; p.bn = q.bn; p.brank = q.brank;
}

// the iteration counts of the last two registrations the host knows the outcome of (a run that was decided because all max_iterations
// launches were out tells nothing yet: its real k comes with its final state, at icp_track_collect)
static void track_note_k (icp_context *h, uint64_t frame, uint32_t k)
{
    if (frame + 1u <= h->track_hist_frame) return;                       // (this frame, or a later one, is in the history already)
    h->track_hist_frame = frame + 1u;
    h->track_k_hist[1] = h->track_k_hist[0]; h->track_k_hist[0] = k;
}
static void track_note_k (icp_context *h, const run_ctl &r) { if (r.done_seen) track_note_k (h, r.p.seq_value, r.k_final); }

// ---- the keeper (icp_host.h: track_keeper) ----------------------------------------------------------------------------------------------
static bool keeper_enabled ()
{
    const char *e = std::getenv ("ICP_AMD_TRACK_KEEPER");
    return !(e && e[0] == '0');
}

// what the keeper does when asked: both open tracked runs, the older first, to their decision and their end kernel
static int keeper_work (icp_context *h)
{
    run_ctl *a = &h->run, *b = &h->run2;
    if (a->active && b->active && b->p.seq_value < a->p.seq_value) std::swap (a, b);
    for (int i = 0; i < 2; ++i) {
        run_ctl *r = i ? b : a, *o = i ? a : b;
        if (!r->active || r->track_slot < 0) continue;
        const int rc = run_finish (h, *r, (o->active && o->track_slot >= 0) ? o : nullptr);
        if (rc) return rc;
        track_note_k (h, *r);
    }
    return ICP_OK;
}

static void keeper_main (icp_context *h)
{
    track_keeper *K = h->keeper;
    K->tid = std::this_thread::get_id ();
    (void) hipSetDevice (h->device);
    for (;;) {
        // idle: a short spin (a tracking loop comes back within a frame's time), then asleep
        int spins = 0;
        while (K->state.load (std::memory_order_acquire) != 1 && !K->quit.load (std::memory_order_acquire)) {
            if (++spins < 20000) { _mm_pause (); continue; }
            std::unique_lock<std::mutex> lk (K->mx);
            K->cv.wait (lk, [&] { return K->state.load (std::memory_order_acquire) == 1 || K->quit.load (std::memory_order_acquire); });
        }
        if (K->quit.load (std::memory_order_acquire)) return;
        int one = 1;
        if (!K->state.compare_exchange_strong (one, 2, std::memory_order_acq_rel)) continue;      // (taken back before it began)
        int rc = ICP_OK;
        if (!K->pause.load (std::memory_order_acquire)) {
            try { rc = keeper_work (h); } catch (...) { rc = on_exception (); }
        }
        if (rc != ICP_OK && rc != ICP_KEEPER_ABORTED && K->rc == ICP_OK) {
            K->rc = rc;
            try { K->err = h->err; } catch (...) { }
        }
        K->state.store (0, std::memory_order_release);
    }
}

}  // extern "C" (the keeper's entry points have C++ linkage)
namespace icp_host {

void keeper_quiesce (icp_context *h)
{
    track_keeper *K = h->keeper;
    if (!K || !K->started || on_keeper_thread (h)) return;
    int s = K->state.load (std::memory_order_acquire);
    if (s == 0) return;
    K->pause.store (true, std::memory_order_release);
    int one = 1;
    if (s == 1 && K->state.compare_exchange_strong (one, 0, std::memory_order_acq_rel)) { K->pause.store (false, std::memory_order_release); return; }
    while (K->state.load (std::memory_order_acquire) != 0) _mm_pause ();      // (at most one launch call of the keeper: microseconds)
    K->pause.store (false, std::memory_order_release);
}

void keeper_kick (icp_context *h)
{
    track_keeper *K = h->keeper;
    if (!K || !K->started || on_keeper_thread (h)) return;
    const bool open = (h->run.active && h->run.track_slot >= 0) || (h->run2.active && h->run2.track_slot >= 0);
    if (!open || !h->track_last_gated || K->rc != ICP_OK) return;
    int zero = 0;
    if (K->state.compare_exchange_strong (zero, 1, std::memory_order_acq_rel)) {
        // (the keeper spins for a while after its last job: the notify is for the one that went to sleep)
        std::lock_guard<std::mutex> lk (K->mx);
        K->cv.notify_one ();
    }
}

void keeper_stop (icp_context *h)
{
    track_keeper *K = h->keeper;
    if (!K) return;
    if (K->started) {
        keeper_quiesce (h);
        { std::lock_guard<std::mutex> lk (K->mx); K->quit.store (true, std::memory_order_release); }
        K->cv.notify_all ();
        if (K->th.joinable ()) K->th.join ();
    }
    delete K;
    h->keeper = nullptr;
}

}  // namespace icp_host
extern "C" {

// started with the first gated frame of a handle (ICP_AMD_TRACK_KEEPER=0: never — the calling thread looks after the runs as in round 5)
static void keeper_start (icp_context *h)
{
    if (h->keeper || !keeper_enabled ()) return;
    track_keeper *K = new (std::nothrow) track_keeper ();
    if (!K) return;
    h->keeper = K;
    try { K->started = true; K->th = std::thread (keeper_main, h); }
    catch (...) { K->started = false; h->keeper = nullptr; delete K; }
}

// an error the keeper ran into while the caller was away: reported by the next tracking call
static int keeper_error (icp_context *h)
{
    track_keeper *K = h->keeper;
    if (!K || K->rc == ICP_OK) return ICP_OK;
    const int rc = K->rc;
    K->rc = ICP_OK;
    return fail (h, rc, K->err.empty () ? std::string ("tracking: the engine's thread failed") : K->err);
}

// Diagnostic (ICP_AMD_TRACK_PROF=1): where the calling thread's time goes inside icp_track_submit — accumulated per step, printed to stderr
// by icp_track_reset.  Steps: 0 bookkeeping + the slot's old run, 1 wait for frame f - 2, 2 band copy / DMA enqueue, 3 getLMs + event,
// 4 run_begin (RBC construction, gate, blind launches), 5 the predecessor brought to its decision, 6 everything.
static struct track_prof_t { bool on; double t[8]; uint64_t n; } g_tp = { std::getenv ("ICP_AMD_TRACK_PROF") != nullptr, { 0, 0, 0, 0, 0, 0, 0, 0 }, 0 };
#define TP(k) do { if (g_tp.on) { const double now_ = now_s (); g_tp.t[k] += now_ - tp_last; tp_last = now_; } } while (0)

// How long a frame's gate waits for its predecessor before it gives up (rounds of ~0.25 us): ~0.5 s, and longer where the predecessor may
// legitimately run longer — it is launch-complete by then, with up to max_iterations iterations still to execute; 2^14 rounds (~4 ms) are
// allowed per iteration (a registration of the largest configuration on a GPU shared with other work), at most 2^28 rounds (~1 min).
// ICP_AMD_GATE_SPINS (diagnostics, tests: read at every call) overrides it.
static uint32_t gate_spins (const icp_context *h)
{
    if (const char *e = std::getenv ("ICP_AMD_GATE_SPINS")) { const long v = std::atol (e); if (v > 0) return (uint32_t) std::min<long> (v, 1l << 28); }
    const uint64_t per_run = (uint64_t) std::max<uint32_t> (h->max_iterations, 1u) << 14;
    return (uint32_t) std::min<uint64_t> (std::max<uint64_t> (per_run, 1ull << 21), 1ull << 28);
}

int icp_track_reset (icp_handle h) try
{
    api_guard guard_ (h);
    if (!h) return ICP_EINVAL;
    int rc = set_device (h); if (rc) return rc;
    if (g_tp.on && g_tp.n) {
        std::fprintf (stderr, "icp_track_submit, %llu calls, us per call: bookkeeping %.1f | wait f-2 %.1f | band %.1f | getLMs + event %.1f | run_begin %.1f | predecessor decided %.1f | all %.1f\n",
                      (unsigned long long) g_tp.n, g_tp.t[0] * 1e6 / g_tp.n, g_tp.t[1] * 1e6 / g_tp.n, g_tp.t[2] * 1e6 / g_tp.n, g_tp.t[3] * 1e6 / g_tp.n, g_tp.t[4] * 1e6 / g_tp.n,
                      g_tp.t[5] * 1e6 / g_tp.n, g_tp.t[6] * 1e6 / g_tp.n);
        for (double &v : g_tp.t) v = 0.0;
        g_tp.n = 0;
    }
    if (h->keeper) { h->keeper->rc = ICP_OK; h->keeper->err.clear (); }      // (whatever the keeper ran into belongs to the sequence that ends here)
    if (h->hGateFlag && *h->hGateFlag) {
        // A gate gave up: the frames behind it were turned into no-ops (their run flag = their epoch: every launch leaves at its first
        // load, the end kernel too), so their runs can never publish a decision and run_finish on them can only fail.  Recovery = what
        // the error text promises: the open runs are dropped, everything queued drains (no-ops and whatever the stalled predecessor
        // still does), and the sequence words start over.
        h->run.active = h->run2.active = false;
        h->stream2_dirty = false;
    } else if ((rc = run_close_all (h))) return rc;
    if (h->copy_stream) HIPCHK (h, hipStreamSynchronize (h->copy_stream));
    if (h->stream2) HIPCHK (h, hipStreamSynchronize (h->stream2));
    if (h->stream) HIPCHK (h, hipStreamSynchronize (h->stream));
    if (h->dSeq) HIPCHK (h, hipMemset (h->dSeq, 0, sizeof (uint32_t)));
    if (h->dRunFlag) HIPCHK (h, hipMemset (h->dRunFlag, 0, 2 * sizeof (uint32_t)));
    if (h->hGateFlag) *h->hGateFlag = 0u;
    h->track_submitted = h->track_collected = 0;
    h->track_k_hist[0] = h->track_k_hist[1] = 0; h->track_hist_frame = 0;
    h->track_last_gated = false;
    return ICP_OK;
}
ICP_CATCH_ALL

// The caller's own frame buffers as DMA sources: a capture loop that fills the same few buffers over and over
// registers them once (hipHostRegister: the pages are locked, the runtime maps them for the device), and a frame submitted from inside a
// registered range goes the way of the engine's pinned frame buffers — the band getLMs reads by one 2-D DMA, no host copy: the
// calling thread's 60 us per frame (the 128 row segments into pinned staging) were the largest fixed part of a warm-started frame's host time.
int icp_track_register_source (icp_handle h, void *frames, size_t bytes) try
{
    api_guard guard_ (h);
    int rc = need (h, false, true); if (rc) return rc;
    if (!frames || bytes < (size_t) 640 * 480 * 32) return fail (h, ICP_EINVAL, "icp_track_register_source: a range of at least one 640 x 480 float8 frame");
    if ((rc = set_device (h))) return rc;
    for (const auto &r : h->sources)
        if (static_cast<const char *> (frames) < r.base + r.bytes && r.base < static_cast<const char *> (frames) + bytes)
            return fail (h, ICP_ESTATE, "icp_track_register_source: the range overlaps one that is registered already");
    const hipError_t e = hipHostRegister (frames, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void) hipGetLastError (); return fail (h, ICP_EHIP, std::string ("hipHostRegister: ") + hipGetErrorString (e)); }
    h->sources.push_back ({ static_cast<const char *> (frames), bytes });
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_track_unregister_source (icp_handle h, void *frames) try
{
    api_guard guard_ (h);
    int rc = need (h, false, true); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    for (size_t i = 0; i < h->sources.size (); ++i)
        if (h->sources[i].base == static_cast<const char *> (frames)) {
            if (h->copy_stream) HIPCHK (h, hipStreamSynchronize (h->copy_stream));      // (no upload out of the range is still queued)
            HIPCHK (h, hipHostUnregister (frames));
            h->sources.erase (h->sources.begin () + (long) i);
            return ICP_OK;
        }
    return fail (h, ICP_EINVAL, "icp_track_unregister_source: not the start of a registered range");
}
ICP_CATCH_ALL

int icp_track_staging (icp_handle h, uint32_t slot, void **host_ptr) try
{
    api_guard guard_ (h);
    int rc = need (h, false, true); if (rc) return rc;
    if (slot > 1u || !host_ptr) return fail (h, ICP_EINVAL, "icp_track_staging: slot must be 0 or 1");
    if ((rc = set_device (h))) return rc;
    if (!h->hFrame[slot]) HIPCHK (h, hipHostMalloc ((void **) &h->hFrame[slot], (size_t) 640 * 480 * 32, hipHostMallocDefault));
    // the buffer is handed out once the band of the frame it last held has left it (its upload may still be queued on the copy stream
    // when more than two frames are in flight; an event that was never recorded returns at once).  The event belongs to the BUFFER: the
    // frames' own upload events go by frame parity, which is the buffer's number only for a caller who alternates the two from frame 0 on
    HIPCHK (h, hipEventSynchronize (h->evFrame[slot]));
    *host_ptr = h->hFrame[slot];
    return ICP_OK;
}
ICP_CATCH_ALL

// Tracking: blind launches of a frame's registration — what is enqueued before icp_track_submit returns (the caller is away until its
// next call: copying the next frame, typically).  The smaller of the last two registrations' k + the launch that finds out; a first
// registration of a sequence gets depth + 1 and is topped up by the next call.
static uint32_t track_blind (const icp_context *h)
{
    { const char *e = std::getenv ("ICP_AMD_TRACK_BLIND"); if (e) return (uint32_t) std::max (1, std::atoi (e)); }     // diagnostics / tests: a fixed number
    const uint32_t a = h->track_k_hist[0], b = h->track_k_hist[1];
    const uint32_t k = a && b ? std::min (a, b) : (a ? a : b);
    return k ? k + 1u : h->run_depth + 1u;
}

static int track_submit (icp_context *h, const void *cloud, int warm_start, bool blocking)
{
    double tp_last = g_tp.on ? now_s () : 0.0; const double tp_first = tp_last;
    int rc = need (h, false, true); if (rc) return rc;
    if ((rc = keeper_error (h))) return rc;
    if (!cloud) return fail (h, ICP_EINVAL, "null pointer");
    if ((rc = set_device (h))) return rc;
    if ((rc = track_prepare (h))) return rc;                            // (everything that can fail for lack of memory comes first)
    if (h->track_submitted - h->track_collected >= ICP_TRACK_RING)
        return fail (h, ICP_ESTATE, "icp_track_submit: four frames are in flight: collect a result first (icp_track_collect)");
    const uint64_t f = h->track_submitted;
    const uint32_t s = (uint32_t) (f & 1u), buf = (uint32_t) (f % 3u), ring = (uint32_t) (f % ICP_TRACK_RING);
    const char *src = static_cast<const char *> (cloud) + ((size_t) ICP_BAND_ROW0 * 640u + ICP_BAND_COL0) * 32u;
    const size_t spitch = (size_t) ICP_BAND_ROW_STEP * 640u * 32u;
    // gated: registration f lives in run slot f & 1 on stream f & 1; its predecessor (f - 1) in the other slot, possibly still open
    // (the release of the sequence word lives in the chained kernel: other forms — reference-order reductions, |R| > 1024 — stay host-ordered)
    // (and the blocking icp_track_next has nothing to overlap: it stays on one stream and spares itself the gate)
    const bool gated = !blocking && h->run_adaptive && h->track_gate && h->rbc2_ready && icp_chain_supported (h->p);
    if (gated != h->track_last_gated) {                                 // the form changes in mid-sequence (a mode was switched): start from a drained device
        if ((rc = run_close_all (h))) return rc;
        if (h->stream2) HIPCHK (h, hipStreamSynchronize (h->stream2));
        HIPCHK (h, hipStreamSynchronize (h->stream));
        if (gated && f > 0u) { icp_launch_seq_set (h->dSeq, (uint32_t) (f - 1u), h->stream); HIPCHK (h, hipGetLastError ()); HIPCHK (h, hipStreamSynchronize (h->stream)); }
        h->track_last_gated = gated;
    }
    run_ctl &R = (gated && (f & 1u)) ? h->run2 : h->run;
    run_ctl *P = gated ? ((f & 1u) ? &h->run : &h->run2) : &h->run;     // the run to look after meanwhile (ungated: the one and only)
    hipStream_t st = (gated && (f & 1u)) ? h->stream2 : h->stream;
    // the previous frame's registration may still need launches while this call does its own work: it is looked after between the steps
    // (a word read; a launch if its queue has run down)
    auto tend = [&] () { if (P->active) (void) run_pump (h, *P); };
    if (gated && R.active) {                                            // the slot still holds registration f - 2: decided long ago, or nearly
        if ((rc = run_finish (h, R, P->active ? P : nullptr))) return rc;
        track_note_k (h, R);
    }
    TP (0);
    // the copy stream: not before registration f - 2 (the last reader of lm[buf], as its fixed set) is done.  Host-driven runs: the host
    // knows — the FINAL bit of that frame's word —, and neither stream carries an event for it (a record + a cross-stream wait cost the
    // main stream ~10 us per frame between the RBC construction and the first iteration, profiles/r04_track_trace.txt)
    if (f >= 2u) {
        const uint32_t r2 = (uint32_t) ((f - 2u) % ICP_TRACK_RING);
        if (h->track_epoch[r2]) {
            // (its end kernel, if it needs one, is enqueued: run_finish above / at the previous submit)
            if (!gated && h->run.active && h->run.track_slot == (int) r2) { if ((rc = run_finish (h))) return rc; track_note_k (h, h->run); }
            if ((rc = run_wait_final (h, h->hTrackMirror + r2, 1u, h->track_epoch[r2], true))) return rc;
        } else HIPCHK (h, hipStreamWaitEvent (h->copy_stream, h->evDone[r2], 0));
    }
    tend ();
    TP (1);
    const bool staged = cloud == h->hFrame[0] || cloud == h->hFrame[1];
    bool pinned = staged;
    for (const auto &r : h->sources)                                    // a frame in one of the caller's registered (page-locked) buffers
        if (static_cast<const char *> (cloud) >= r.base && static_cast<const char *> (cloud) + (size_t) 640 * 480 * 32 <= r.base + r.bytes) pinned = true;
    if (pinned) {
        // the caller filled one of the engine's pinned frame buffers (icp_track_staging): the band goes by DMA straight from there
        HIPCHK (h, hipMemcpy2DAsync (h->dBand[s], ICP_BAND_ROW_BYTES, src, spitch, ICP_BAND_ROW_BYTES, ICP_BAND_ROWS, hipMemcpyHostToDevice, h->copy_stream));
        if (staged) HIPCHK (h, hipEventRecord (h->evFrame[cloud == h->hFrame[0] ? 0 : 1], h->copy_stream));
        tend ();
    } else {
        // pageable source: the band's 128 row segments into the slot's pinned staging (free once the upload of frame f - 2 is through)
        if (f >= 2u) HIPCHK (h, hipEventSynchronize (h->evUp[s]));
        // in two pieces, each uploaded as soon as it is staged: the DMA of the first runs under the host copy of the second (a blocking
        // icp_track_next waits for 60 us of copy + 45 us of upload otherwise; every copy command costs ~12 us by itself, so more pieces
        // give the gain back: 1 / 2 / 4 pieces = 393 / 371 / 393 us per blocking cold frame; ICP_AMD_BAND_PIECES for the comparison)
        static const uint32_t npieces = [] { const char *e = std::getenv ("ICP_AMD_BAND_PIECES"); const int v = e ? std::atoi (e) : 2; return (v == 1 || v == 2 || v == 4 || v == 8) ? (uint32_t) v : 2u; } ();
        const uint32_t piece = ICP_BAND_ROWS / npieces;
        for (uint32_t j = 0; j < ICP_BAND_ROWS; ++j) {
            std::memcpy (reinterpret_cast<char *> (h->hBand[s]) + (size_t) j * ICP_BAND_ROW_BYTES, src + (size_t) j * spitch, ICP_BAND_ROW_BYTES);
            // (the copy takes ~60 us: the previous frame's open registration is looked after on the way — a word read, a launch if it needs one)
            if ((j & 7u) == 7u && P->active) (void) run_pump (h, *P);
            if ((j + 1u) % piece == 0u) {
                const size_t off = (size_t) (j + 1u - piece) * ICP_BAND_ROW_BYTES;
                HIPCHK (h, hipMemcpyAsync (reinterpret_cast<char *> (h->dBand[s]) + off, reinterpret_cast<char *> (h->hBand[s]) + off, (size_t) piece * ICP_BAND_ROW_BYTES,
                                           hipMemcpyHostToDevice, h->copy_stream));
                tend ();
            }
        }
    }
    TP (2);
    icp_launch_get_lms_band (h->dBand[s], h->lm[buf], h->copy_stream);
    HIPCHK (h, hipGetLastError ());
    tend ();
    HIPCHK (h, hipEventRecord (h->evUp[s], h->copy_stream));
    tend ();
    TP (3);
    // ungated: one stream, in order — the previous frame's registration is brought to its end before this frame's work goes behind it
    int prev_slot = -1;
    if (!gated && h->run.active) {
        prev_slot = h->run.track_slot;
        if ((rc = run_finish (h))) return rc;
        track_note_k (h, h->run);
    }
    // (rounds 1 - 3's form only: host-driven runs order the streams from the host, see above)
    auto record_prev = [&] () -> int { if (prev_slot >= 0 && !h->track_epoch[prev_slot]) { HIPCHK (h, hipEventRecord (h->evDone[prev_slot], h->stream)); } prev_slot = -1; return ICP_OK; };
    // an upload is waited for on the stream only if it is not through yet (it is, whenever a registration takes longer than an upload)
    auto wait_upload = [&] (uint32_t slot) -> int { if (hipEventQuery (h->evUp[slot]) != hipSuccess) { (void) hipGetLastError (); HIPCHK (h, hipStreamWaitEvent (st, h->evUp[slot], 0)); } return ICP_OK; };
    note_inputs_change (h);
    float *newM = h->lm[buf], *newF = h->lm[(f + 2u) % 3u];             // (f - 1) mod 3: the previous frame's landmarks (first frame: a buffer that is not M)
    icp_params p = h->p; p.M = newM; p.F = newF; p.seq_value = (uint32_t) f;
    if (gated) rbc_into (p, h->rbc[f & 1u]);
    if (f > 0u) {
        note_enqueue (h);
        // warm start: from the previous hop's transform, as by write (D_IO_T) — the first registration of a sequence has no previous hop
        // and starts from the identity whatever the state holds (an earlier sequence's last transform, an icp_run before the reset)
        const bool warm = warm_start && f > 1u;
        if (h->run_adaptive) {
            const uint32_t blind = blocking ? h->run_depth + 1u : track_blind (h);
            if (gated) {
                // buildRBC runs AHEAD of the previous frame's end (its own RBC set, the fixed landmarks resident since that frame's upload)
                // and must not touch the registration state; behind it the gate: registration f - 1 has released the sequence word
                p.no_state_reset = 1u; p.run_flag = h->dRunFlag + (f & 1u); p.track_seq = h->dSeq; p.seq_value = (uint32_t) f;
                if ((rc = wait_upload ((uint32_t) ((f - 1u) & 1u)))) return rc;          // (the fixed set: frame f - 1's landmarks)
            }
            // buildRBC reads the fixed set only — the previous frame's landmarks —: this frame's upload is waited for behind it
            auto between = [&] () -> int {
                int rc2 = record_prev (); if (rc2) return rc2;
                if ((rc2 = wait_upload (s))) return rc2;
                // (a warm-started frame behind a gate: the gate kernel writes the state as k_set_T would, once it is open)
                const bool in_gate = gated && f >= 2u && warm;
                if (gated && f >= 2u) { icp_launch_gate (h->dSeq, (uint32_t) (f - 1u), h->hGateFlag, st, gate_spins (h), in_gate ? p.st : nullptr, R.p.run_flag, R.p.epoch); HIPCHK (h, hipGetLastError ()); }
                if (warm && !in_gate) { icp_launch_set_T (p, 0, p.st->T, st, gated ? 1 : 0); HIPCHK (h, hipGetLastError ()); }
                return ICP_OK;
            };
            if ((rc = run_begin (h, R, st, p, !warm, true, blind, h->hTrackMirror + ring, h->hTrack + ring, (int) ring, between, (gated && P->active) ? P : nullptr))) return rc;
            h->track_epoch[ring] = R.p.epoch;
            // (gated: a launch past the convergence of a frame runs beside the next frame and costs ~0.6 - 0.9 us, a queue that runs dry while
            // the host is busy with the next frame costs what the host is late by: the queue is kept twice as deep)
            if (gated) R.depth = std::max (R.depth, 2u * h->run_depth);
            if (gated && (f & 1u)) h->stream2_dirty = true;
        } else {
            // rounds 1 - 3: buildRBC + a checked run of max_iterations launches as one cached graph (the graphs hold the buffer pointers)
            if ((rc = record_prev ())) return rc;
            HIPCHK (h, hipStreamWaitEvent (h->stream, h->evUp[s], 0));
            if (warm) { icp_launch_set_T (p, 0, p.st->T, h->stream); HIPCHK (h, hipGetLastError ()); }
            float *oF = h->dF, *oM = h->dM; const float *opF = h->p.F, *opM = h->p.M; const uint32_t opar = h->parity;
            h->dM = newM; h->p.M = newM; h->dF = newF; h->p.F = newF; h->parity = 1u + buf;
            rc = launch_run (h, h->max_iterations, 1, !warm, true);
            if (rc) { h->dF = oF; h->dM = oM; h->p.F = opF; h->p.M = opM; h->parity = opar; return rc; }
            HIPCHK (h, hipMemcpyAsync (&h->hTrack[ring], h->p.st, sizeof (icp_reg_state), hipMemcpyDeviceToHost, h->stream));
            HIPCHK (h, hipEventRecord (h->evDone[ring], h->stream));
            h->track_epoch[ring] = 0u;
        }
    } else {
        if ((rc = record_prev ())) return rc;
        HIPCHK (h, hipStreamWaitEvent (h->stream, h->evUp[s], 0));      // this frame's landmarks (nothing to register against yet)
        HIPCHK (h, hipEventRecord (h->evDone[ring], h->stream));
        h->track_epoch[ring] = 0u;
    }
    // Gated form: frame f waits on the DEVICE for its predecessor's release of the sequence word, and an undecided predecessor gets its
    // launches from this thread — which is about to leave the library for as long as the application likes (a camera's next frame, a
    // debugger, a garbage collection).  So the predecessor is brought to its decision HERE: converged (its final state stored, the word
    // released by the launch that found out), or all max_iterations launches and its end kernel enqueued.  From then on the device goes
    // from frame f - 1 to frame f by itself; what is left open when this returns is frame f, whose queue may run dry while the caller is
    // away — nothing waits behind it on the device, it is topped up by the next call: a slow caller costs time, never a frame.
    // (The caller's cadence is unchanged where it matters: icp_track_collect of frame f - 1, the next call of a pipelined loop, would
    // have waited for the same decision.  Frame f's launches went out above, while the predecessor was still running.)
    TP (4);
    if (gated && !h->keeper) keeper_start (h);
    if (gated && P->active && !h->keeper) {              // (no keeper — ICP_AMD_TRACK_KEEPER=0, or no thread to be had: round 5's rule, the caller brings the predecessor to its decision)
        if ((rc = run_finish (h, *P, R.active ? &R : nullptr))) {
            // (frame f's launches are queued, its bookkeeping is not committed: a retry would reuse its ring slot and landmark buffer under
            // them.  Drop the run and let the queues drain before the error leaves; icp_track_reset starts over.)
            R.active = false;
            if (h->stream2) (void) hipStreamSynchronize (h->stream2);
            if (h->stream) (void) hipStreamSynchronize (h->stream);
            return rc;
        }
        track_note_k (h, *P);
    }
    TP (5);
    if (g_tp.on) { g_tp.t[6] += now_s () - tp_first; ++g_tp.n; }
    // everything that can fail is behind us: the handle now points at this frame's buffers
    h->dM = newM; h->p.M = newM; h->dF = newF; h->p.F = newF;
    if (gated && f > 0u) rbc_into (h->p, h->rbc[f & 1u]);
    h->parity = 1u + buf;                                               // graphs hold the pointers: one cached set per rotation step (0: the buffers of icp_init)
    h->built = f > 0u;
    h->track_submitted = f + 1u;
    return ICP_OK;
}

int icp_track_submit (icp_handle h, const void *cloud, int warm_start) try { api_guard guard_ (h); return track_submit (h, cloud, warm_start, false); } ICP_CATCH_ALL

int icp_track_form (icp_handle h, int *gated) try
{
    api_guard guard_ (h);
    int rc = need (h, false, true); if (rc) return rc;
    if (!gated) return fail (h, ICP_EINVAL, "null output");
    if ((rc = set_device (h))) return rc;
    if (h->run_adaptive && h->track_gate && h->p.m == 16384u && h->p.batch == 1u && h->ownF && h->ownM && (rc = track_prepare (h))) return rc;   // (runs the probe)
    *gated = (h->run_adaptive && h->track_gate && h->rbc2_ready && icp_chain_supported (h->p)) ? 1 : 0;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_track_collect (icp_handle h, uint32_t *k, float *T8, int *registered) try
{
    api_guard guard_ (h);
    int rc = need (h, false, true); if (rc) return rc;
    if ((rc = keeper_error (h))) return rc;
    if (h->track_collected >= h->track_submitted) return fail (h, ICP_ESTATE, "icp_track_collect: no frame in flight");
    if ((rc = set_device (h))) return rc;
    const uint64_t f = h->track_collected;
    const uint32_t ring = (uint32_t) (f % ICP_TRACK_RING);
    for (int i = 0; i < 2; ++i) {                                       // the frame's registration is still open: top it up to its end
        run_ctl &R = i ? h->run2 : h->run;
        run_ctl &O = i ? h->run : h->run2;
        if (R.active && R.track_slot == (int) ring && R.p.epoch == h->track_epoch[ring]) {
            if ((rc = run_finish (h, R, O.active ? &O : nullptr))) return rc;
            track_note_k (h, R);
        }
    }
    if (f > 0u && h->track_epoch[ring]) { if ((rc = run_wait_final (h, h->hTrackMirror + ring, 1u, h->track_epoch[ring], true))) return rc; }
    else HIPCHK (h, hipEventSynchronize (h->evDone[ring]));
    if (h->hGateFlag && *h->hGateFlag)
        return fail (h, ICP_EHIP, "tracking: the device made no progress for ~0.5 s (a frame's gate gave up waiting for its launch-complete predecessor; the frames behind it "
                                  "were skipped, nothing was overwritten): icp_track_reset starts a new sequence");
    h->track_collected = f + 1u;
    if (f + 1u == h->track_submitted && f > 0u && h->track_epoch[ring]) {
        // nothing behind this frame: the handle's state is this registration's final state, and the host holds it
        h->hState[0] = h->hTrack[ring]; h->hstate_fresh = true; h->hstate_here = true;
    }
    if (k) *k = 0;
    if (registered) *registered = f > 0u ? 1 : 0;
    if (f > 0u) {
        const icp_reg_state &st = h->hTrack[ring];
        track_note_k (h, f, st.k);
        if (k) *k = st.k;
        if (T8) std::memcpy (T8, st.T, 8 * sizeof (float));
    } else if (T8) { const float T0[8] = { 0, 0, 0, 1, 0, 0, 0, 1 }; std::memcpy (T8, T0, sizeof T0); }
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_track_next (icp_handle h, const void *cloud, int warm_start, uint32_t *k, int *registered) try
{
    api_guard guard_ (h);
    if (k) *k = 0;
    if (registered) *registered = 0;
    int rc = need (h, false, true); if (rc) return rc;
    while (h->track_collected < h->track_submitted)                     // (results of an earlier pipelined use nobody collected)
        if ((rc = icp_track_collect (h, nullptr, nullptr, nullptr))) return rc;
    if ((rc = track_submit (h, cloud, warm_start, true))) return rc;
    if ((rc = icp_track_collect (h, k, nullptr, registered))) return rc;
    if (!h->run_adaptive) return settle (h);
    return ICP_OK;
}
ICP_CATCH_ALL

}  // extern "C"
