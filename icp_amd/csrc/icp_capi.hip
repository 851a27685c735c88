// icp_capi.hip — C-ABI (include/icp_amd.h) over the HIP kernels: handle, buffers, stream, hipGraphs.
//
// Host-side counterpart of ICPStep<CR,CW> / ICP<CR,CW> (include/ICP/algorithms.hpp:2234-2496,
// src/ICP/algorithms.cpp:4348-4903).  The reference wires ten kernel-wrapper objects by sharing
// cl::Buffer handles (:4499-4581) and syncs with the host every iteration (:4681-4697); here one
// handle owns one stream, all device buffers of a batch of registrations and the device-resident
// registration state, and an ICP run is a single hipGraph launch.
//
// There is NO CPU fallback: without a gfx950 device icp_create fails with ICP_ENODEVICE.
#include "../../include/icp_amd.h"
#include "icp_kernels.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <immintrin.h>

namespace {

thread_local std::string g_create_error;

struct graph_entry { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t used = 0, gen = 0; };

// A checked run (ICP::run — src/ICP/algorithms.cpp:4806-4834: iterate until check () says stop) that the HOST drives, launch by launch.
// The device publishes every new transform's (k, done) as one 8-byte store into fine-grained host memory (icp_params::hmirror); the
// host keeps `depth` launches queued behind the one in flight and stops enqueueing the moment `done` shows: a run costs k launches
// (+ at most `depth` that leave at their first load), not max_iterations.  Plain launches, not graphs: back to back they run at the
// graph's rate (8.77 against 8.73 us per iteration at |F| = 16384) and every graph boundary costs 4 - 8 us (profiles/r04_segments.txt).
// The end kernel leaves the final state in host memory too (icp_params::hstate) and sets the word's FINAL bit: the caller polls that
// instead of synchronising the stream.  At most one run per handle is open; tracking keeps it open across calls (icp_track_submit
// returns with a frame's predicted launches enqueued, the next call tops it up).
struct run_ctl {
    bool active = false, decided = false, chained = false, fresh = false;
    icp_params p {};
    uint32_t enq = 0, maxit = 0, depth = 0, k_seen = 0, k_final = 0, k0 = 0;     // k0: the device's k when the run began (k_seen is relative to it)
    int done_seen = 0;
    bool final_seen = false;                            // every registration's final state has arrived with its converged flag: no end kernel
    volatile unsigned long long *mirror = nullptr;      // host view of p.hmirror
    int track_slot = -1;                                // tracking: the ring slot of the frame this run registers
    hipStream_t stream = nullptr;                       // the stream the run's launches go to (tracking alternates between two)
    // host timeline of the run (icp_run_timeline), seconds on the steady clock: begin, blind launches enqueued, first progress word seen,
    // decided, end kernel enqueued
    double t[5] = { 0, 0, 0, 0, 0 };
    double launch_max_us = 0.0; uint32_t launch_slow = 0;   // the host's own launch calls: the longest, and how many took more than 10 us
};

inline double now_s () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }

}  // namespace

struct icp_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool inited = false, built = false;
    icp_params p {};
    uint32_t max_iterations = 40;
    double angle_threshold = 0.001, translation_threshold = 0.01;
    std::string err;
    // owned allocations
    std::vector<void *> dev_allocs;
    float *dF = nullptr, *dM = nullptr;          // may be adopted
    bool ownF = true, ownM = true;
    float *hF = nullptr, *hM = nullptr, *hT = nullptr;   // pinned staging (H_IN_F / H_IN_M / H_IO_T)
    icp_reg_state *hState = nullptr;             // pinned (fine-grained) mirror of the registration states: the end kernel of a checked run stores into it
    bool hstate_fresh = false;                   // the mirror is what the device holds (a checked run was the last state-changing thing on the stream)
    bool hstate_here = false;                    // ... and it has arrived (host-driven run: its FINAL bit was seen); else: once the stream has drained
    unsigned long long *hMirror = nullptr;       // pinned (fine-grained): progress words of the checked run in flight, [batch] (run_ctl)
    uint32_t epoch = 0;                          // tag of the last checked run
    // what the host knows about the device's iteration counter k (all registrations alike): a checked run's progress words carry k itself,
    // and a run that does not start at 0 (a second icp_run without buildRBC) paces itself against k - k_base.  -1: unknown (paced as from 0:
    // a few launches more queued than `depth`, nothing else)
    long long k_base = 0;
    uint32_t run_depth = 3;                      // launches kept queued behind the one in flight (ICP_AMD_RUN_DEPTH)
    int run_adaptive = 1;                        // 0 (ICP_AMD_RUN_ADAPTIVE=0): checked runs as one graph of max_iterations launches (rounds 1 - 3)
    run_ctl run;                                 // the open checked run (tracking: of the frames on the handle's own stream)
    run_ctl run2;                                // tracking with device-side gates: the open run of the frames on stream2
    // Per-query outputs (NN_ID, W, NN, QT, RID) of checked runs: fused kernels consume none of them, and a checked run cannot know which
    // iteration is its last — storing them every iteration costs 0.4 us of every 9 at |F| = 16384.  lazy: the run stores none; every finalize
    // leaves the transform its search used in p.st_prev, and the first read of such an output re-runs that one search (same T, same
    // lists: same bits).  Inputs changed in between (F / M written, RBC rebuilt, tracking moved on): the outputs are gone, reads say so.
    int outputs_lazy = 1;                        // ICP_AMD_OUTPUTS=eager / icp_set_output_mode
    bool outputs_stale = false, outputs_lost = false;
    uint32_t stat_launches = 0, stat_k = 0, stat_dead = 0;   // last finished checked run: iteration launches enqueued, final k, launches past the last live one
    double stat_t[6] = { 0, 0, 0, 0, 0, 0 };     // its host timeline (run_ctl::t) + the moment its FINAL bit was seen
    double stat_launch_max_us = 0.0; uint64_t stat_launch_slow = 0, stat_launch_total = 0;   // launch calls of all checked runs since icp_init
    uint64_t graph_clock = 0, param_gen = 0;     // LRU stamp of the graph cache; generation of the parameters the cached graphs were captured with
    float *dTin = nullptr;                       // device scratch for write(T)
    float *dCloud = nullptr, *dCloudOut = nullptr; uint32_t cloud_cap = 0;
    std::map<uint64_t, graph_entry> graphs;      // key: iterations << 3 | check << 2 | parity (+ fresh, + kind: see get_graph)
    uint32_t parity = 0;                         // tracking: which landmark buffers are the fixed / moving set (graphs hold pointers): frame f -> f mod 3
    // frame-to-frame tracking (icp_track_*): three landmark buffers in rotation, band staging, a copy stream
    float *lm[3] = { nullptr, nullptr, nullptr };            // landmarks of frame f live in lm[f mod 3] (lm[0] / lm[1] = the handle's F / M buffers)
    float *hBand[2] = { nullptr, nullptr }, *dBand[2] = { nullptr, nullptr };     // the part of a frame getLMs reads (ICP_BAND_*), pinned / device
    float *hFrame[2] = { nullptr, nullptr };                 // whole-frame pinned staging handed to the caller (icp_track_staging)
    icp_reg_state *hTrack = nullptr;                         // pinned: final state of the frames in flight (ICP_TRACK_RING slots)
    unsigned long long *hTrackMirror = nullptr;              // pinned: their progress words
    uint32_t track_epoch[4] = { 0, 0, 0, 0 };                // epoch of the run in each ring slot
    uint32_t track_k_hist[2] = { 0, 0 };                     // k of the last two registrations of the sequence (0: none yet): the next frame's blind launches
    uint64_t track_hist_frame = 0;                           // 1 + the latest frame whose k is in that history
    hipStream_t copy_stream = nullptr;
    // Tracking with frames gated on the device (track_gate; ICP_AMD_TRACK_GATE=0 switches it off): registration f runs on stream f & 1 (the
    // handle's own stream / stream2) behind k_gate, which waits for registration f - 1's release of *dSeq — so frame f's RBC construction
    // and all its predicted launches are enqueued while frame f - 1 is still running, and the host is nowhere on the path between two
    // frames.  The RBC of two consecutive frames lives in two sets of buffers (rbc2 = the second set; swapped into h->p by frame parity).
    hipStream_t stream2 = nullptr;
    int track_gate = 1;
    uint32_t *dSeq = nullptr, *dRunFlag = nullptr, *hGateFlag = nullptr;
    bool stream2_dirty = false;                  // stream2 holds work the handle's own stream must not overtake
    struct rbc_set { float *R = nullptr; float4 *GB = nullptr; float *XP = nullptr, *XQ = nullptr; uint32_t *rep_src = nullptr, *owner = nullptr, *N = nullptr, *O = nullptr,
                     *perm = nullptr, *chunk_hist = nullptr; uint2 *blist = nullptr; uint32_t *bn = nullptr; uint8_t *brank = nullptr; } rbc[2];
    bool rbc2_ready = false;
    bool track_last_gated = false;                // the form of the last submitted frame
    hipEvent_t evUp[2] = { nullptr, nullptr }, evDone[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t evStage[3] = { nullptr, nullptr, nullptr };   // the last asynchronous copy out of the pinned staging of F / M / T (icp_write)
    uint64_t track_submitted = 0, track_collected = 0;       // frames fed / frames whose result has been handed out since init / icp_track_reset
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

int fail (icp_context *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail ((h), ICP_EHIP, std::string (#expr) + ": " + hipGetErrorString (e_));   \
    } while (0)

void drop_graphs (icp_context *h)
{
    for (auto &kv : h->graphs) {
        if (kv.second.exec) (void) hipGraphExecDestroy (kv.second.exec);
        if (kv.second.graph) (void) hipGraphDestroy (kv.second.graph);
    }
    h->graphs.clear ();
}

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
    if (h->lm[2]) (void) hipFree (h->lm[2]);
    h->lm[0] = h->lm[1] = h->lm[2] = nullptr;
    if (h->hTrack) (void) hipHostFree (h->hTrack);
    h->hTrack = nullptr;
    if (h->rbc2_ready) {
        icp_context::rbc_set &q = h->rbc[1];
        void *ptrs[] = { q.R, q.GB, q.XP, q.XQ, q.rep_src, q.owner, q.N, q.O, q.perm, q.chunk_hist, q.blist, q.bn, q.brank };
        for (void *x : ptrs) if (x) (void) hipFree (x);
    }
    h->rbc[0] = h->rbc[1] = icp_context::rbc_set {}; h->rbc2_ready = false;
    if (h->dSeq) (void) hipFree (h->dSeq);
    if (h->dRunFlag) (void) hipFree (h->dRunFlag);
    if (h->hGateFlag) (void) hipHostFree (h->hGateFlag);
    h->dSeq = h->dRunFlag = h->hGateFlag = nullptr; h->run2 = run_ctl {}; h->stream2_dirty = false; h->track_last_gated = false;
    h->inited = h->built = false; h->parity = 0; h->track_submitted = h->track_collected = 0;
}

template <typename T>
int dalloc (icp_context *h, T **ptr, size_t count, bool zero = true)
{
    void *q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof (T);
    hipError_t e = hipMalloc (&q, bytes);
    if (e != hipSuccess) return fail (h, ICP_ENOMEM, std::string ("hipMalloc: ") + hipGetErrorString (e));
    h->dev_allocs.push_back (q);
    if (zero) {
        e = hipMemsetAsync (q, 0, bytes, h->stream);
        if (e != hipSuccess) return fail (h, ICP_EHIP, std::string ("hipMemsetAsync: ") + hipGetErrorString (e));
    }
    *ptr = static_cast<T *> (q);
    return ICP_OK;
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

int run_finish (icp_context *h, bool defer_event = false);
int run_close_all (icp_context *h);
void note_outputs_stored (icp_context *h);

// keep_run: the caller is one of the tracking entries, which carry an open checked run (run_ctl) across calls themselves; everything
// else that touches the handle's stream first brings an open run to its end (its remaining launches must not interleave with others)
int need (icp_context *h, bool built, bool keep_run = false)
{
    if (!h) return ICP_EINVAL;
    if (!h->inited) return fail (h, ICP_ESTATE, "icp_init has not been called");
    if (built && !h->built) return fail (h, ICP_ESTATE, "icp_build_rbc has not been called");
    if (!keep_run && (h->run.active || h->run2.active || h->stream2_dirty)) {
        if (hipSetDevice (h->device) != hipSuccess) return fail (h, ICP_EHIP, "hipSetDevice");
        int rc = run_close_all (h); if (rc) return rc;
    }
    return ICP_OK;
}

int set_device (icp_context *h)
{
    HIPCHK (h, hipSetDevice (h->device));
    return ICP_OK;
}

// Captures the launches `launches ()` enqueues on the handle's stream into a graph (instantiate: also into an executable one).
// Whatever fails, the stream has left capture mode and nothing is leaked when this returns.
template <typename Fn>
int capture_graph (icp_context *h, Fn &&launches, graph_entry *out, bool instantiate = true)
{
    graph_entry ge;
    HIPCHK (h, hipStreamBeginCapture (h->stream, hipStreamCaptureModeThreadLocal));
    launches ();
    const hipError_t le = hipGetLastError ();                        // launch-configuration errors of the captured kernels
    hipError_t e = hipStreamEndCapture (h->stream, &ge.graph);      // always: ends the capture also on the error path
    if (e == hipSuccess && le != hipSuccess) e = le;
    if (e != hipSuccess) {
        if (ge.graph) (void) hipGraphDestroy (ge.graph);
        return fail (h, ICP_EHIP, std::string ("graph capture: ") + hipGetErrorString (e));
    }
    if (instantiate) {
        e = hipGraphInstantiate (&ge.exec, ge.graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void) hipGraphDestroy (ge.graph);
            return fail (h, ICP_EHIP, std::string ("hipGraphInstantiate: ") + hipGetErrorString (e));
        }
    }
    *out = ge;
    return ICP_OK;
}

#define ICP_GRAPH_CACHE 8u       // cached run graphs per handle (least recently used goes first)

// Graph of `iterations` iterations, cached (fixed-length runs: icp_run_fixed*, the timing entries; checked runs only with
// ICP_AMD_RUN_ADAPTIVE=0).  A parameter change (setAlpha, setScaling, thresholds, modes) does not throw the executable graphs away: an
// entry of an older parameter generation is re-captured and its executable graph UPDATED in place (hipGraphExecUpdate: the kernel
// nodes' arguments; instantiating anew costs milliseconds) — same topology by construction, re-instantiated only if the update is refused.
// fresh: the graph starts the registration from the identity transform (icp_reset_transform + the run as one graph; the
// chained form folds the reset into its first launch).  with_build: buildRBC in front of the run.
int get_graph (icp_context *h, uint32_t iterations, int check, hipGraphExec_t *out, bool fresh = false, bool with_build = false)
{
    uint64_t key = ((uint64_t) iterations << 3) | (uint64_t) (check ? 4 : 0) | (uint64_t) h->parity | ((uint64_t) (fresh ? 1 : 0) << 62) | ((uint64_t) (with_build ? 1 : 0) << 61);
    auto it = h->graphs.find (key);
    if (it != h->graphs.end () && it->second.gen == h->param_gen) { it->second.used = ++h->graph_clock; *out = it->second.exec; return ICP_OK; }
    icp_params p = h->p;
    p.check = check; p.hmirror = nullptr; p.hstate = nullptr;
    auto launches = [&] {
        if (with_build) icp_launch_build_rbc (p, h->stream);
        if (fresh && !icp_chain_supported (p)) icp_launch_reset_state (p, h->stream, 1);
        if (icp_chain_supported (p)) icp_launch_chain (p, h->stream, iterations, fresh);   // one launch per iteration
        else for (uint32_t k = 0; k < iterations; ++k) {
            p.emit = (check || k + 1 == iterations) ? 1 : 0;            // (with checks on, any iteration may be the last executed)
            icp_launch_iteration (p, h->stream);
        }
        // checked graphs: the states travel to the pinned mirror as the last node of the graph
        if (check) (void) hipMemcpyAsync (h->hState, p.st, sizeof (icp_reg_state) * p.batch, hipMemcpyDeviceToHost, h->stream);
    };
    if (it != h->graphs.end ()) {                                       // stale parameters: update the executable graph in place
        graph_entry ng;
        int rc = capture_graph (h, launches, &ng, false);
        if (rc) return rc;
        hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
        hipError_t e = hipGraphExecUpdate (it->second.exec, ng.graph, &bad, &res);
        if (e != hipSuccess || res != hipGraphExecUpdateSuccess) {
            (void) hipGetLastError ();
            (void) hipGraphExecDestroy (it->second.exec); it->second.exec = nullptr;
            e = hipGraphInstantiate (&it->second.exec, ng.graph, nullptr, nullptr, 0);
            if (e != hipSuccess) {
                (void) hipGraphDestroy (ng.graph); (void) hipGraphDestroy (it->second.graph);
                h->graphs.erase (it);
                return fail (h, ICP_EHIP, std::string ("hipGraphInstantiate: ") + hipGetErrorString (e));
            }
        }
        (void) hipGraphDestroy (it->second.graph);
        it->second.graph = ng.graph; it->second.gen = h->param_gen; it->second.used = ++h->graph_clock;
        *out = it->second.exec;
        return ICP_OK;
    }
    graph_entry ge;
    int rc = capture_graph (h, launches, &ge);
    if (rc) return rc;
    ge.gen = h->param_gen; ge.used = ++h->graph_clock;
    if (h->graphs.size () >= ICP_GRAPH_CACHE) {                          // bounded: the least recently used entry goes
        auto lru = h->graphs.begin ();
        for (auto jt = h->graphs.begin (); jt != h->graphs.end (); ++jt) if (jt->second.used < lru->second.used) lru = jt;
        // (an executable graph may still be queued on the stream: the runtime keeps what a launched graph needs until it has run)
        if (lru->second.exec) (void) hipGraphExecDestroy (lru->second.exec);
        if (lru->second.graph) (void) hipGraphDestroy (lru->second.graph);
        h->graphs.erase (lru);
    }
    h->graphs[key] = ge;
    *out = ge.exec;
    return ICP_OK;
}

// ---- host-driven checked runs (run_ctl) ------------------------------------------------------------------------------------------

void run_launch_one (icp_context *h, run_ctl &r)
{
    (void) h;
    const double t0 = now_s ();
    if (r.chained) icp_launch_chain_one (r.p, r.stream, r.enq, r.fresh, r.p.emit != 0);
    else icp_launch_iteration (r.p, r.stream);
    const double us = (now_s () - t0) * 1e6;
    if (us > r.launch_max_us) r.launch_max_us = us;
    if (us > 10.0) ++r.launch_slow;
    ++r.enq;
}

// Opens a checked run on the handle's stream with `blind` iterations enqueued at once (at least one).  p: the parameters of THIS run
// (tracking passes the frame's own landmark buffers); mirror / hstate: the host memory its words and final state go to.
// between (): enqueued after the RBC construction and in front of the first iteration (tracking: the waits and records that need not
// hold the construction back).
struct run_no_hook { int operator() () const { return ICP_OK; } };
bool run_pump (icp_context *h, run_ctl &r);
// r: the slot the run lives in (h->run; tracking with gates: h->run / h->run2 by frame parity), stream: where its launches go;
// other: another open run that is looked after while this one's launches are being enqueued (a tracked frame's predecessor).
template <typename BETWEEN = run_no_hook>
int run_begin (icp_context *h, run_ctl &r, hipStream_t stream, const icp_params &p, bool fresh, bool with_build, uint32_t blind,
               unsigned long long *mirror, icp_reg_state *hstate, int track_slot, BETWEEN between = BETWEEN (), run_ctl *other = nullptr)
{
    r = run_ctl {};
    r.stream = stream;
    r.t[0] = now_s ();
    r.p = p; r.p.check = 1;
    r.p.emit = (h->outputs_lazy && p.fused) ? 0 : 1;                    // (reference-order kernels read the outputs themselves: always stored)
    h->outputs_stale = r.p.emit == 0; h->outputs_lost = false;
    if (++h->epoch == 0u) h->epoch = 1u;
    r.p.epoch = h->epoch; r.p.hmirror = mirror; r.p.hstate = hstate;    // (fine-grained host allocations: the host pointer is the device pointer)
    r.mirror = mirror; r.track_slot = track_slot;
    for (uint32_t b = 0; b < p.batch; ++b) mirror[b] = 0ull;
    std::atomic_thread_fence (std::memory_order_seq_cst);
    r.chained = icp_chain_supported (r.p); r.fresh = fresh;
    r.k0 = (fresh || with_build || h->k_base < 0) ? 0u : (uint32_t) h->k_base;      // (a fresh run and a rebuilt RBC start the count at 0)
    r.maxit = h->max_iterations; r.depth = h->run_depth ? h->run_depth : 1u;
    if (other && other->active) (void) run_pump (h, *other);
    if (with_build) icp_launch_build_rbc (r.p, r.stream);
    if (other && other->active) (void) run_pump (h, *other);
    { int rc = between (); if (rc) return rc; }
    if (fresh && !r.chained) icp_launch_reset_state (r.p, r.stream, 1);
    r.active = true;
    const uint32_t n = std::min (std::max (blind, 1u), r.maxit);
    while (r.enq < n) {
        run_launch_one (h, r);
        if (other && other->active && (r.enq & 1u) == 0u) (void) run_pump (h, *other);
    }
    if (r.enq >= r.maxit) { r.decided = true; r.k_final = r.maxit + r.k0; }
    HIPCHK (h, hipGetLastError ());
    h->hstate_fresh = false; h->hstate_here = false; h->k_base = -1;
    r.t[1] = now_s ();
    return ICP_OK;
}

// One look at the run's words, then the queue topped up to `depth` launches behind the one in flight.  Returns true once the run is
// decided: every registration has converged, or max_iterations launches are enqueued (nothing more will be).
bool run_pump (icp_context *h, run_ctl &r)
{
    if (r.decided) return true;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u; bool all_done = true, all_final = true;
    for (uint32_t b = 0; b < r.p.batch; ++b) {
        const unsigned long long w = r.mirror[b];
        const bool mine = (uint32_t) (w >> 32) == r.p.epoch;
        const uint32_t kabs = mine ? (uint32_t) (w & 0xFFFFFFull) : 0u, k = kabs > r.k0 ? kabs - r.k0 : 0u;
        const bool done = mine && (w & ICP_MIRROR_DONE);
        kmax = std::max (kmax, k);
        if (!done) { all_done = false; kmin = std::min (kmin, k); }
        if (!(mine && (w & ICP_MIRROR_FINAL))) all_final = false;
    }
    if (all_done) { r.decided = true; r.done_seen = 1; r.final_seen = all_final; r.k_seen = kmax; r.k_final = kmax + r.k0; return true; }
    if (kmin && !r.k_seen) r.t[2] = now_s ();
    r.k_seen = kmin;
    // launch (chained) / search (separate launches) j publishes k = j in its prologue: k_seen = the iteration in flight, `depth`
    // iterations are kept queued behind it
    while (r.enq < r.maxit && r.enq < r.k_seen + 1u + r.depth) run_launch_one (h, r);
    if (r.enq >= r.maxit) { r.decided = true; r.k_final = r.maxit + r.k0; }
    return r.decided;
}

// Drives the open run to its decision (the calling thread polls; bounded wait on a device that has stopped answering), then
// enqueues its end kernel — final state -> p.st and -> host memory, FINAL bit — and closes it.
// other: a run this one may be waiting for on the device (a gated frame's predecessor): it is topped up in the same loop.
int run_finish (icp_context *h, run_ctl &r, run_ctl *other)
{
    if (!r.active) return ICP_OK;
    uint32_t spins = 0, k_last = 0xFFFFFFFFu;
    auto t_last = std::chrono::steady_clock::now ();
    while (!run_pump (h, r)) {
        if (other && other->active) { const uint32_t ko = other->k_seen; (void) run_pump (h, *other); if (other->k_seen != ko || other->decided) t_last = std::chrono::steady_clock::now (); }
        _mm_pause ();
        if ((++spins & 0x3FFu) == 0u) {
            const auto now = std::chrono::steady_clock::now ();
            if (r.k_seen != k_last) { k_last = r.k_seen; t_last = now; }
            else if (r.k0 && std::chrono::duration<double> (now - t_last).count () > 0.05) { r.k0 = 0u; t_last = now; }      // (a stale idea of where the count began: pace on k itself)
            else if (std::chrono::duration<double> (now - t_last).count () > 20.0) {
                r.active = false;
                return fail (h, ICP_EHIP, "checked run: the device has published no progress for 20 s");
            }
        }
    }
    r.t[3] = now_s ();
    // (converged in the fused forms: the finalize that set the flag has left the final state in p.st and in host memory already)
    if (!r.final_seen) {
        if (r.chained) icp_launch_chain_end (r.p, r.stream, r.enq);
        else icp_launch_publish_state (r.p, r.stream);
    }
    r.t[4] = now_s ();
    for (int i = 0; i < 5; ++i) h->stat_t[i] = r.t[i];
    h->stat_launch_max_us = std::max (h->stat_launch_max_us, r.launch_max_us); h->stat_launch_slow += r.launch_slow; h->stat_launch_total += r.enq;
    r.active = false;
    h->stat_launches = r.enq; h->stat_k = r.k_final;
    // iterations enqueued past the one that found out (converged at k: iterations 0 .. k - 1 ran, launch k saw the flag — in the chained form it
    // is the one that sets it —, the rest leave at their first load)
    h->stat_dead = r.done_seen ? r.enq - std::min (r.enq, r.k_final - r.k0 + 1u) : 0u;
    HIPCHK (h, hipGetLastError ());
    return ICP_OK;
}
int run_finish (icp_context *h, bool) { return run_finish (h, h->run, h->run2.active ? &h->run2 : nullptr); }

// Waits for the FINAL bit of `n` words of epoch `epoch` (the end kernel's last store: the final states are in host memory).
int run_wait_final (icp_context *h, volatile unsigned long long *mirror, uint32_t n, uint32_t epoch)
{
    uint32_t spins = 0;
    const auto t0 = std::chrono::steady_clock::now ();
    for (uint32_t b = 0; b < n; ++b) {
        for (;;) {
            const unsigned long long w = mirror[b];
            if ((uint32_t) (w >> 32) == epoch && (w & ICP_MIRROR_FINAL)) break;
            _mm_pause ();
            if ((++spins & 0x3FFFu) == 0u && std::chrono::duration<double> (std::chrono::steady_clock::now () - t0).count () > 60.0) {
                // (is the stream in error?  hipStreamQuery reports a faulted queue)
                const hipError_t e = hipStreamQuery (h->stream);
                if (e != hipSuccess && e != hipErrorNotReady) return fail (h, ICP_EHIP, std::string ("checked run: ") + hipGetErrorString (e));
                return fail (h, ICP_EHIP, "checked run: the final state has not arrived after 60 s");
            }
        }
    }
    std::atomic_thread_fence (std::memory_order_acquire);
    return ICP_OK;
}

// Brings every open run to its end (older frame first) and, after gated tracking, drains stream2: whatever is enqueued on the handle's own
// stream next must not overtake it.
int run_close_all (icp_context *h)
{
    run_ctl *a = &h->run, *b = &h->run2;
    if (a->active && b->active && b->p.seq_value < a->p.seq_value) std::swap (a, b);      // a = the older frame
    int rc;
    if (a->active && (rc = run_finish (h, *a, b->active ? b : nullptr))) return rc;
    if (b->active && (rc = run_finish (h, *b, nullptr))) return rc;
    if (h->stream2_dirty && h->stream2) { HIPCHK (h, hipStreamSynchronize (h->stream2)); h->stream2_dirty = false; }
    return ICP_OK;
}

// Launches the graph of a run.
int launch_run (icp_context *h, uint32_t iterations, int check, bool fresh = false, bool with_build = false)
{
    {   // diagnostic (ICP_AMD_RUN_GRAPH=0): the same launches enqueued one by one instead of as a cached graph
        static const char *e = std::getenv ("ICP_AMD_RUN_GRAPH");
        if (e && e[0] == '0') {
            icp_params p = h->p; p.check = check; p.hmirror = nullptr; p.hstate = nullptr;
            if (with_build) icp_launch_build_rbc (p, h->stream);
            if (fresh && !icp_chain_supported (p)) icp_launch_reset_state (p, h->stream, 1);
            if (icp_chain_supported (p)) icp_launch_chain (p, h->stream, iterations, fresh);
            else for (uint32_t k = 0; k < iterations; ++k) { p.emit = (check || k + 1 == iterations) ? 1 : 0; icp_launch_iteration (p, h->stream); }
            if (check) HIPCHK (h, hipMemcpyAsync (h->hState, p.st, sizeof (icp_reg_state) * p.batch, hipMemcpyDeviceToHost, h->stream));
            HIPCHK (h, hipGetLastError ());
            h->hstate_fresh = check != 0; h->hstate_here = false;
            h->k_base = check ? -1 : (fresh || with_build) ? (long long) iterations : (h->k_base >= 0 ? h->k_base + iterations : -1);
            note_outputs_stored (h);
            return ICP_OK;
        }
    }
    hipGraphExec_t exec;
    int rc = get_graph (h, iterations, check, &exec, fresh, with_build);
    if (rc) return rc;
    HIPCHK (h, hipGraphLaunch (exec, h->stream));
    h->hstate_fresh = check != 0; h->hstate_here = false;
    h->k_base = check ? -1 : (fresh || with_build) ? (long long) iterations : (h->k_base >= 0 ? h->k_base + iterations : -1);
    note_outputs_stored (h);
    return ICP_OK;
}

// Waits for everything enqueued on the handle's stream.
int settle (icp_context *h)
{
    HIPCHK (h, hipStreamSynchronize (h->stream));
    return ICP_OK;
}

// every state-changing enqueue that is not a checked run graph: the pinned mirror of the states is stale from here on
void note_enqueue (icp_context *h) { h->hstate_fresh = false; h->hstate_here = false; h->k_base = -1; }

// the inputs of the last checked run are about to change (F / M / the RBC): per-query outputs it did not store can no longer be reproduced
void note_inputs_change (icp_context *h) { if (h->outputs_stale) { h->outputs_stale = false; h->outputs_lost = true; } }
// an enqueue that stores the per-query outputs itself (single steps, fixed-length runs: their last iteration)
void note_outputs_stored (icp_context *h) { h->outputs_stale = false; h->outputs_lost = false; }

bool is_query_output (int mem) { return mem == ICP_MEM_NN_ID || mem == ICP_MEM_W || mem == ICP_MEM_NN || mem == ICP_MEM_QT || mem == ICP_MEM_RID; }

// Per-query outputs of a checked run that stored none: the search of its last executed iteration again — p.st_prev holds the transform it
// used — with the stores on.  The moments it leaves are nobody's (the run is over); the state is not touched.
int materialize_outputs (icp_context *h, int mem)
{
    if (!is_query_output (mem)) return ICP_OK;
    if (h->outputs_lost)
        return fail (h, ICP_ESTATE, "the per-query outputs of the last checked run were not stored (lazy outputs) and its inputs have changed since: "
                                    "read them before F / M / the RBC change, or switch to icp_set_output_mode (h, ICP_OUTPUTS_EVERY_ITERATION)");
    if (!h->outputs_stale) return ICP_OK;
    icp_params q = h->p;
    q.st = q.st_prev; q.check = 0; q.emit = 1; q.hmirror = nullptr; q.hstate = nullptr;
    icp_launch_search (q, h->stream);
    HIPCHK (h, hipGetLastError ());
    h->outputs_stale = false;
    return ICP_OK;
}

}  // namespace

extern "C" {

const char *icp_version (void) { return "icp_amd 0.1 (gfx950)"; }

const char *icp_last_error (icp_handle h) { return h ? h->err.c_str () : g_create_error.c_str (); }

int icp_device_count (int *n)
{
    if (!n) return fail (nullptr, ICP_EINVAL, "icp_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount (&c);
    if (e != hipSuccess) { *n = 0; return fail (nullptr, ICP_ENODEVICE, std::string ("hipGetDeviceCount: ") + hipGetErrorString (e)); }
    *n = c;
    return ICP_OK;
}

int icp_create (icp_handle *out, int device, int rot, int weighted)
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
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evDone[k], hipEventDisableTiming);
    for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreateWithFlags (&h->evStage[k], hipEventDisableTiming);
    if (e != hipSuccess) { std::string m = hipGetErrorString (e); h->inited = false; icp_destroy (h); return fail (nullptr, ICP_EHIP, "icp_create: " + m); }
    *out = h;
    return ICP_OK;
}

int icp_destroy (icp_handle h)
{
    if (!h) return ICP_EINVAL;
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
    for (int k = 0; k < 4; ++k) if (h->evDone[k]) (void) hipEventDestroy (h->evDone[k]);
    for (int k = 0; k < 3; ++k) if (h->evStage[k]) (void) hipEventDestroy (h->evStage[k]);
    if (h->copy_stream) (void) hipStreamDestroy (h->copy_stream);
    if (h->stream2) (void) hipStreamDestroy (h->stream2);
    if (h->stream) (void) hipStreamDestroy (h->stream);
    delete h;
    return ICP_OK;
}

int icp_init_batched (icp_handle h, uint32_t batch, uint32_t m, uint32_t nr, float a, float c,
                      uint32_t max_iterations, double angle_threshold, double translation_threshold)
{
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
    if ((rc = dalloc (h, &p.rep_src, B * nr))) return rc;
    if ((rc = dalloc (h, &p.owner, B * m))) return rc;
    if ((rc = dalloc (h, &p.N, B * nr))) return rc;
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

int icp_init (icp_handle h, uint32_t m, uint32_t nr, float a, float c, uint32_t max_iterations,
              double angle_threshold, double translation_threshold)
{
    return icp_init_batched (h, 1, m, nr, a, c, max_iterations, angle_threshold, translation_threshold);
}

int icp_write_b (icp_handle h, uint32_t b, int mem, const void *host_ptr, int block)
{
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

int icp_write (icp_handle h, int mem, const void *host_ptr, int block) { return icp_write_b (h, 0, mem, host_ptr, block); }

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
        case ICP_MEM_RBC_N: *src = p.N + (size_t) b * p.nr; break;
        case ICP_MEM_RBC_O: *src = p.O + (size_t) b * p.nr; break;
        case ICP_MEM_NN: *src = p.PF + (size_t) b * p.m; break;
        case ICP_MEM_QT: *src = p.PM + (size_t) b * p.m; break;
        case ICP_MEM_W: *src = reinterpret_cast<const float *> (p.PF + (size_t) b * p.m) + 3; break;
        default: return fail (h, ICP_EINVAL, "unknown icp_mem value");
    }
    return ICP_OK;
}

int icp_read_b (icp_handle h, uint32_t b, int mem, void *host_dst, size_t bytes)
{
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

int icp_read (icp_handle h, int mem, void *host_dst, size_t bytes) { return icp_read_b (h, 0, mem, host_dst, bytes); }

int icp_device_ptr (icp_handle h, int mem, void **dptr)
{
    int rc = need (h, false); if (rc) return rc;
    if (!dptr) return fail (h, ICP_EINVAL, "null pointer");
    const void *src = nullptr;
    if ((rc = mem_ptr (h, 0, mem, &src))) return rc;
    if ((rc = set_device (h))) return rc;
    if ((rc = materialize_outputs (h, mem))) return rc;
    *dptr = const_cast<void *> (src);
    return ICP_OK;
}

int icp_adopt_device_buffer (icp_handle h, int mem, void *dptr)
{
    int rc = need (h, false); if (rc) return rc;
    if (!dptr) return fail (h, ICP_EINVAL, "null pointer");
    if (mem == ICP_MEM_F) { h->dF = static_cast<float *> (dptr); h->p.F = h->dF; h->ownF = false; h->built = false; }
    else if (mem == ICP_MEM_M) { h->dM = static_cast<float *> (dptr); h->p.M = h->dM; h->ownM = false; }
    else return fail (h, ICP_EINVAL, "only ICP_MEM_F and ICP_MEM_M can be adopted");
    note_inputs_change (h);
    drop_graphs (h);
    return ICP_OK;
}

int icp_build_rbc (icp_handle h)
{
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

int icp_step (icp_handle h, int config)
{
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

int icp_run_fixed (icp_handle h, uint32_t iterations)
{
    int rc = need (h, true); if (rc) return rc;
    if (iterations == 0) return ICP_OK;
    if ((rc = set_device (h))) return rc;
    return launch_run (h, iterations, 0);
}

int icp_run_fixed_fresh (icp_handle h, uint32_t iterations)
{
    int rc = need (h, true); if (rc) return rc;
    if (iterations == 0) return icp_reset_transform (h);
    if ((rc = set_device (h))) return rc;
    return launch_run (h, iterations, 0, true);
}

int icp_run (icp_handle h, uint32_t *k)
{
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

int icp_run_stats (icp_handle h, uint32_t *launches, uint32_t *k, uint32_t *dead)
{
    if (!h) return ICP_EINVAL;
    if (launches) *launches = h->stat_launches;
    if (k) *k = h->stat_k;
    if (dead) *dead = h->stat_dead;
    return ICP_OK;
}

int icp_launch_stats (icp_handle h, double *max_us, uint64_t *slower_than_10us, uint64_t *total, int reset)
{
    if (!h) return ICP_EINVAL;
    if (max_us) *max_us = h->stat_launch_max_us;
    if (slower_than_10us) *slower_than_10us = h->stat_launch_slow;
    if (total) *total = h->stat_launch_total;
    if (reset) { h->stat_launch_max_us = 0.0; h->stat_launch_slow = h->stat_launch_total = 0; }
    return ICP_OK;
}

int icp_set_output_mode (icp_handle h, int mode)
{
    if (!h) return ICP_EINVAL;
    if (mode != ICP_OUTPUTS_LAZY && mode != ICP_OUTPUTS_EVERY_ITERATION) return fail (h, ICP_EINVAL, "unknown output mode");
    h->outputs_lazy = mode == ICP_OUTPUTS_LAZY;
    return ICP_OK;
}

int icp_run_timeline (icp_handle h, double *us6)
{
    if (!h || !us6) return ICP_EINVAL;
    for (int i = 0; i < 6; ++i) us6[i] = (h->stat_t[i] - h->stat_t[0]) * 1e6;
    return ICP_OK;
}

int icp_set_run_depth (icp_handle h, uint32_t depth, int adaptive)
{
    if (!h) return ICP_EINVAL;
    if (depth == 0 || depth > 64u) return fail (h, ICP_EINVAL, "icp_set_run_depth: depth must be in [1, 64]");
    { int rc = set_device (h); if (rc) return rc; if ((rc = run_close_all (h))) return rc; }
    h->run_depth = depth; h->run_adaptive = adaptive ? 1 : 0;
    return ICP_OK;
}

int icp_sync (icp_handle h)
{
    if (!h) return ICP_EINVAL;
    int rc = set_device (h); if (rc) return rc;
    return settle (h);
}

int icp_get_alpha (icp_handle h, float *a) { if (!h || !a) return ICP_EINVAL; *a = h->p.a; return ICP_OK; }
// The setters change a number in the handle's parameters and nothing else: checked runs are plain launches that read the parameters as they
// are, and a cached fixed-length graph of an older parameter generation is updated in place when it is next used (get_graph).
int icp_set_alpha (icp_handle h, float a)
{   // setAlpha updates construct and search (src/ICP/algorithms.cpp:4712-4717); lists must be rebuilt by the caller
    if (!h) return ICP_EINVAL;
    if (a == 0.f) return fail (h, ICP_EINVAL, "The alpha parameter cannot be equal to zero");
    h->p.a = a; ++h->param_gen; return ICP_OK;
}
int icp_set_metric_scale (icp_handle h, float f_g)
{
    if (!h) return ICP_EINVAL;
    if (!(f_g > 0.f) || !std::isfinite (f_g)) return fail (h, ICP_EINVAL, "the metric scale must be positive and finite");
    h->p.dist_scale = f_g; ++h->param_gen; return ICP_OK;
}
int icp_get_metric_scale (icp_handle h, float *f_g) { if (!h || !f_g) return ICP_EINVAL; *f_g = h->p.dist_scale; return ICP_OK; }
int icp_get_scaling (icp_handle h, float *c) { if (!h || !c) return ICP_EINVAL; *c = h->p.c; return ICP_OK; }
int icp_set_scaling (icp_handle h, float c) { if (!h) return ICP_EINVAL; h->p.c = c; ++h->param_gen; return ICP_OK; }
int icp_get_max_iterations (icp_handle h, uint32_t *n) { if (!h || !n) return ICP_EINVAL; *n = h->max_iterations; return ICP_OK; }
int icp_set_max_iterations (icp_handle h, uint32_t n)
{
    if (!h) return ICP_EINVAL;
    if (n == 0) return fail (h, ICP_EINVAL, "max_iterations must be positive");
    h->max_iterations = n; return ICP_OK;
}
int icp_get_angle_threshold (icp_handle h, double *d) { if (!h || !d) return ICP_EINVAL; *d = h->angle_threshold; return ICP_OK; }
int icp_set_angle_threshold (icp_handle h, double d)
{
    if (!h) return ICP_EINVAL;
    h->angle_threshold = d; h->p.tan_half_thr = std::tan (d * M_PI / 360.0); ++h->param_gen; return ICP_OK;
}
int icp_get_translation_threshold (icp_handle h, double *d) { if (!h || !d) return ICP_EINVAL; *d = h->translation_threshold; return ICP_OK; }
int icp_set_translation_threshold (icp_handle h, double d)
{
    if (!h) return ICP_EINVAL;
    h->translation_threshold = d; h->p.trans_thr = d; ++h->param_gen; return ICP_OK;
}
int icp_set_power_mode (icp_handle h, int mode)
{
    if (!h) return ICP_EINVAL;
    if (mode != ICP_POWER_LITERAL && mode != ICP_POWER_SQUARED) return fail (h, ICP_EINVAL, "unknown power mode");
    h->p.power_mode = mode; ++h->param_gen; return ICP_OK;
}

int icp_set_reduce_mode (icp_handle h, int mode)
{
    if (!h) return ICP_EINVAL;
    if (mode != ICP_REDUCE_REFERENCE_ORDER && mode != ICP_REDUCE_FUSED) return fail (h, ICP_EINVAL, "unknown reduce mode");
    h->p.fused = mode; drop_graphs (h); return ICP_OK;
}

int icp_state_b (icp_handle h, uint32_t b, icp_state_t *out)
{
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

int icp_state (icp_handle h, icp_state_t *out) { return icp_state_b (h, 0, out); }

int icp_write_cloud (icp_handle h, int which, const void *cloud, int block)
{
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

int icp_transform_cloud (icp_handle h, const void *host_in, void *host_out, uint32_t n)
{
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

// ---- frame-to-frame tracking ---------------------------------------------------------------------------------------------------
// Frame f's landmarks live in lm[f mod 3]; registration f (frame f onto frame f - 1) reads lm[f mod 3] as the moving and
// lm[(f - 1) mod 3] as the fixed set, so frame f + 1 can be uploaded and its landmarks extracted (copy stream) while registration f
// runs (main stream): the buffer it goes to was last read by registration f - 1.  Two staging slots (f mod 2) hold the band of
// a frame (the 2.08 MB of its 9.83 MB that getLMs reads) in pinned memory; per frame the main stream gets ONE graph — buildRBC +
// the checked run — and a 248-byte copy of the final state into the frame's slot of a pinned ring.
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
        a.R = p.R; a.GB = p.GB; a.XP = p.XP; a.XQ = p.XQ; a.rep_src = p.rep_src; a.owner = p.owner; a.N = p.N; a.O = p.O; a.perm = p.perm;
        a.chunk_hist = p.chunk_hist; a.blist = p.blist; a.bn = p.bn; a.brank = p.brank;
        auto al = [&] (void **q, size_t bytes) -> int {
            hipError_t e = hipMalloc (q, bytes ? bytes : 1);
            if (e == hipSuccess) e = hipMemset (*q, 0, bytes ? bytes : 1);
            return e == hipSuccess ? ICP_OK : fail (h, ICP_ENOMEM, std::string ("tracking (second RBC set): ") + hipGetErrorString (e));
        };
        int rc;
        if ((rc = al ((void **) &b.R, (size_t) p.nr * 32)) || (rc = al ((void **) &b.GB, (size_t) 2 * (p.n16 + p.n1k) * 16)) || (rc = al ((void **) &b.XP, (size_t) p.m * 32)) ||
            (rc = al ((void **) &b.XQ, (size_t) p.m * 32)) || (rc = al ((void **) &b.rep_src, (size_t) p.nr * 4)) || (rc = al ((void **) &b.owner, (size_t) p.m * 4)) ||
            (rc = al ((void **) &b.N, (size_t) p.nr * 4)) || (rc = al ((void **) &b.O, (size_t) p.nr * 4)) || (rc = al ((void **) &b.perm, (size_t) p.m * 4)) ||
            (rc = al ((void **) &b.chunk_hist, (size_t) p.nchunk * p.nr * 4)) || (rc = al ((void **) &b.blist, (size_t) p.nb * 64 * 8)) ||
            (rc = al ((void **) &b.bn, (size_t) p.nb * 4)) || (rc = al ((void **) &b.brank, (size_t) p.m))) {
            void *ptrs[] = { b.R, b.GB, b.XP, b.XQ, b.rep_src, b.owner, b.N, b.O, b.perm, b.chunk_hist, b.blist, b.bn, b.brank };
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
    p.R = q.R; p.GB = q.GB; p.XP = q.XP; p.XQ = q.XQ; p.rep_src = q.rep_src; p.owner = q.owner; p.N = q.N; p.O = q.O; p.perm = q.perm;
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

int icp_track_reset (icp_handle h)
{
    if (!h) return ICP_EINVAL;
    int rc = set_device (h); if (rc) return rc;
    if ((rc = run_close_all (h))) return rc;
    if (h->copy_stream) HIPCHK (h, hipStreamSynchronize (h->copy_stream));
    if (h->stream2) HIPCHK (h, hipStreamSynchronize (h->stream2));
    if (h->stream) HIPCHK (h, hipStreamSynchronize (h->stream));
    if (h->dSeq) HIPCHK (h, hipMemset (h->dSeq, 0, sizeof (uint32_t)));
    if (h->hGateFlag) *h->hGateFlag = 0u;
    h->track_submitted = h->track_collected = 0;
    h->track_k_hist[0] = h->track_k_hist[1] = 0; h->track_hist_frame = 0;
    h->track_last_gated = false;
    return ICP_OK;
}

int icp_track_staging (icp_handle h, uint32_t slot, void **host_ptr)
{
    int rc = need (h, false, true); if (rc) return rc;
    if (slot > 1u || !host_ptr) return fail (h, ICP_EINVAL, "icp_track_staging: slot must be 0 or 1");
    if ((rc = set_device (h))) return rc;
    if (!h->hFrame[slot]) HIPCHK (h, hipHostMalloc ((void **) &h->hFrame[slot], (size_t) 640 * 480 * 32, hipHostMallocDefault));
    // the buffer is handed out once the band of the frame it last held has left it (its upload may still be queued on the copy stream
    // when more than two frames are in flight; an event that was never recorded returns at once)
    HIPCHK (h, hipEventSynchronize (h->evUp[slot]));
    *host_ptr = h->hFrame[slot];
    return ICP_OK;
}

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
    int rc = need (h, false, true); if (rc) return rc;
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
    // the copy stream: not before registration f - 2 (the last reader of lm[buf], as its fixed set) is done.  Host-driven runs: the host
    // knows — the FINAL bit of that frame's word —, and neither stream carries an event for it (a record + a cross-stream wait cost the
    // main stream ~10 us per frame between the RBC construction and the first iteration, profiles/r04_track_trace.txt)
    if (f >= 2u) {
        const uint32_t r2 = (uint32_t) ((f - 2u) % ICP_TRACK_RING);
        if (h->track_epoch[r2]) {
            // (its end kernel, if it needs one, is enqueued: run_finish above / at the previous submit)
            if (!gated && h->run.active && h->run.track_slot == (int) r2) { if ((rc = run_finish (h))) return rc; track_note_k (h, h->run); }
            if ((rc = run_wait_final (h, h->hTrackMirror + r2, 1u, h->track_epoch[r2]))) return rc;
        } else HIPCHK (h, hipStreamWaitEvent (h->copy_stream, h->evDone[r2], 0));
    }
    tend ();
    const bool pinned = cloud == h->hFrame[0] || cloud == h->hFrame[1];
    if (pinned) {
        // the caller filled one of the engine's pinned frame buffers (icp_track_staging): the band goes by DMA straight from there
        HIPCHK (h, hipMemcpy2DAsync (h->dBand[s], ICP_BAND_ROW_BYTES, src, spitch, ICP_BAND_ROW_BYTES, ICP_BAND_ROWS, hipMemcpyHostToDevice, h->copy_stream));
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
    icp_launch_get_lms_band (h->dBand[s], h->lm[buf], h->copy_stream);
    HIPCHK (h, hipGetLastError ());
    tend ();
    HIPCHK (h, hipEventRecord (h->evUp[s], h->copy_stream));
    tend ();
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
                if (gated && f >= 2u) { icp_launch_gate (h->dSeq, (uint32_t) (f - 1u), h->hGateFlag, st); HIPCHK (h, hipGetLastError ()); }
                if (warm) { icp_launch_set_T (p, 0, p.st->T, st, gated ? 1 : 0); HIPCHK (h, hipGetLastError ()); }
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
    // everything that can fail is behind us: the handle now points at this frame's buffers
    h->dM = newM; h->p.M = newM; h->dF = newF; h->p.F = newF;
    if (gated && f > 0u) rbc_into (h->p, h->rbc[f & 1u]);
    h->parity = 1u + buf;                                               // graphs hold the pointers: one cached set per rotation step (0: the buffers of icp_init)
    h->built = f > 0u;
    h->track_submitted = f + 1u;
    return ICP_OK;
}

int icp_track_submit (icp_handle h, const void *cloud, int warm_start) { return track_submit (h, cloud, warm_start, false); }

int icp_track_form (icp_handle h, int *gated)
{
    int rc = need (h, false, true); if (rc) return rc;
    if (!gated) return fail (h, ICP_EINVAL, "null output");
    if ((rc = set_device (h))) return rc;
    if (h->run_adaptive && h->track_gate && h->p.m == 16384u && h->p.batch == 1u && h->ownF && h->ownM && (rc = track_prepare (h))) return rc;   // (runs the probe)
    *gated = (h->run_adaptive && h->track_gate && h->rbc2_ready && icp_chain_supported (h->p)) ? 1 : 0;
    return ICP_OK;
}

int icp_track_collect (icp_handle h, uint32_t *k, float *T8, int *registered)
{
    int rc = need (h, false, true); if (rc) return rc;
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
    if (f > 0u && h->track_epoch[ring]) { if ((rc = run_wait_final (h, h->hTrackMirror + ring, 1u, h->track_epoch[ring]))) return rc; }
    else HIPCHK (h, hipEventSynchronize (h->evDone[ring]));
    if (h->hGateFlag && *h->hGateFlag) return fail (h, ICP_EHIP, "tracking: a frame's gate gave up waiting for its predecessor (device-side wait of ~0.5 s exceeded)");
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

int icp_track_next (icp_handle h, const void *cloud, int warm_start, uint32_t *k, int *registered)
{
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

int icp_transform_cloud_ex (icp_handle h, int kind, const float *T, const void *host_in, void *host_out, uint32_t n)
{
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

int icp_power_method (int device, int rot, int power_mode, const float *S11, const float *means8, float *Tk8, float *Rk9, uint32_t *iters)
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

int icp_reset_transform (icp_handle h)
{   // T <- identity, k <- 0 (what ICPStep::init uploads, src/ICP/algorithms.cpp:4486-4493); enqueue only
    int rc = need (h, false); if (rc) return rc;
    if ((rc = set_device (h))) return rc;
    note_enqueue (h);
    icp_launch_reset_state (h->p, h->stream, 1);
    HIPCHK (h, hipGetLastError ());
    h->k_base = 0;
    return ICP_OK;
}

int icp_time_run_fixed (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_total)
{
    int rc = need (h, true); if (rc) return rc;
    if (!ms_total || iterations == 0 || reps == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    hipGraphExec_t exec;
    if ((rc = get_graph (h, iterations, 0, &exec, from_identity != 0))) return rc;     // from_identity: every pass is a fresh registration
    HIPCHK (h, hipEventRecord (h->ev0, h->stream));
    for (uint32_t r = 0; r < reps; ++r) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    h->hstate_fresh = false; h->k_base = -1; note_outputs_stored (h);
    HIPCHK (h, hipEventRecord (h->ev1, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));                       // (not hipEventSynchronize: its wake-up now and then takes 0.5 ms, tests/diag_overhead.py)
    HIPCHK (h, hipEventElapsedTime (ms_total, h->ev0, h->ev1));
    return ICP_OK;
}

int icp_time_run_fixed_tail (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_timed, uint32_t *reps_timed)
{
    int rc = need (h, true); if (rc) return rc;
    if (!ms_timed || !reps_timed || iterations == 0 || reps == 0) return fail (h, ICP_EINVAL, "bad arguments");
    if ((rc = set_device (h))) return rc;
    hipGraphExec_t exec;
    if ((rc = get_graph (h, iterations, 0, &exec, from_identity != 0))) return rc;
    // the first event goes in BEHIND the first pass: a marker recorded on an idle stream delays the graph launched right after it by
    // 0.1 - 0.25 ms (7.22 - 7.39 ms for 20 passes of 0.357 ms against 7.17 ms wall-clock for the same launches without events;
    // with the GPU busy when the marker arrives: 7.14 ms, tests/diag_overhead.py); the events then bracket the passes 2 .. reps
    const uint32_t lead = reps >= 2u ? 1u : 0u;
    if (lead) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    HIPCHK (h, hipEventRecord (h->ev0, h->stream));
    for (uint32_t r = lead; r < reps; ++r) HIPCHK (h, hipGraphLaunch (exec, h->stream));
    h->hstate_fresh = false; h->k_base = -1; note_outputs_stored (h);
    HIPCHK (h, hipEventRecord (h->ev1, h->stream));
    HIPCHK (h, hipStreamSynchronize (h->stream));                       // (not hipEventSynchronize: its wake-up now and then takes 0.5 ms, tests/diag_overhead.py)
    HIPCHK (h, hipEventElapsedTime (ms_timed, h->ev0, h->ev1));
    *reps_timed = reps - lead;
    return ICP_OK;
}

int icp_run_form (icp_handle h, int *form)
{
    int rc = need (h, false); if (rc) return rc;
    if (!form) return fail (h, ICP_EINVAL, "null output");
    if (icp_chain_supported (h->p)) *form = ICP_FORM_CHAINED;
    else *form = ICP_FORM_SEPARATE;
    return ICP_OK;
}

int icp_search_layout (icp_handle h, int *dense, int *tile, int *stage2)
{
    int rc = need (h, false); if (rc) return rc;
    icp_search_layout_of (h->p, dense, tile, stage2);
    return ICP_OK;
}

int icp_launches_per_iteration (icp_handle h, uint32_t *n)
{
    int rc = need (h, false); if (rc) return rc;
    if (!n) return fail (h, ICP_EINVAL, "null output");
    int form = ICP_FORM_SEPARATE;
    if ((rc = icp_run_form (h, &form))) return rc;
    // (fused, large sets: the first level of the moment tree is a launch of its own — icp_launch_finalize)
    *n = form != ICP_FORM_SEPARATE ? 1u : h->p.fused ? ((h->p.nb + 127u) / 128u > ICP_L1_MIN_GROUPS ? 3u : 2u) : 4u;
    return ICP_OK;
}

int icp_time_masked (icp_handle h, uint32_t mask, uint32_t iterations, uint32_t reps, float *ms_total)
{
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

int icp_debug_stamps (icp_handle h, unsigned long long *out, uint32_t nblocks)
{   // diagnostic builds (ICP_DBG_STAMPS): one k_search launch, per-block s_memtime stamps
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

int icp_profile_run (icp_handle h, uint32_t iterations, float *out_ms, float *total_ms)
{
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

int icp_time_kernels (icp_handle h, uint32_t reps, float *out_ms4)
{
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

}  // extern "C"
