// icp_host.h — what the host-side translation units of the engine share: the handle (icp_context), the bookkeeping of a host-driven
// checked run (run_ctl), the graph cache's entry, and the internal functions icp_run.hip provides to icp_capi.hip (life cycle, buffers,
// setters, diagnostics) and icp_track.hip (frame-to-frame tracking).  Nothing in here is part of the C-ABI (include/icp_amd.h): the
// functions live in a namespace of hidden visibility.
#pragma once

#include "../../include/icp_amd.h"
#include "icp_kernels.h"
#include "icp_cguard.h"

#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <immintrin.h>

namespace icp_host __attribute__ ((visibility ("hidden"))) {

inline thread_local std::string g_create_error;     // icp_last_error (NULL): why the last icp_create of this thread failed

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
    bool end_enqueued = false;                          // the end kernel is on the stream already (a run whose queue reached max_iterations: nothing waits for the host)
    volatile unsigned long long *mirror = nullptr;      // host view of p.hmirror
    int track_slot = -1;                                // tracking: the ring slot of the frame this run registers
    hipStream_t stream = nullptr;                       // the stream the run's launches go to (tracking alternates between two)
    // host timeline of the run (icp_run_timeline), seconds on the steady clock: begin, blind launches enqueued, first progress word seen,
    // decided, end kernel enqueued
    double t[5] = { 0, 0, 0, 0, 0 };
    double launch_max_us = 0.0; uint32_t launch_slow = 0;   // the host's own launch calls: the longest, and how many took more than 10 us
};

// Tracking, gated form: the thread that looks after the open runs while the application is outside the library.
// A host-driven checked run gets its launches from whoever polls its progress word; icp_track_submit used to stay in the library until
// the PREVIOUS frame was decided (the liveness rule of round 5: 87 of the 123 us the call held the caller) and the newest frame's queue
// ran dry whenever the caller stayed away.  Now the call returns once its own launches are out and the keeper pumps both open runs to
// their decision and enqueues their end kernels.  Mutual exclusion by construction: the keeper works only while no API call is inside the
// library on this handle — every entry point pauses it first (api_guard) and hands the runs back when it leaves.
struct track_keeper {
    std::thread th;
    std::mutex mx;
    std::condition_variable cv;
    std::atomic<int> state { 0 };                // 0 idle, 1 asked to look after the open runs, 2 doing it
    std::atomic<bool> pause { false }, quit { false };
    bool started = false;
    std::thread::id tid;                         // the keeper's own thread (run_finish asks: am I the keeper?)
    int rc = 0; std::string err;                 // what it ran into (a device that stopped answering): reported by the next tracking call
};
constexpr int ICP_KEEPER_ABORTED = -77;          // internal: run_finish on the keeper's thread was asked to stop (never leaves the library)

inline double now_s () { return std::chrono::duration<double> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }

}  // namespace icp_host

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
    icp_host::run_ctl run;                                 // the open checked run (tracking: of the frames on the handle's own stream)
    icp_host::run_ctl run2;                                // tracking with device-side gates: the open run of the frames on stream2
    int api_depth = 0;                                     // entry points on the stack (api_guard)
    icp_host::track_keeper *keeper = nullptr;             // tracking, gated form: the thread that pumps the open runs between API calls (icp_track.hip)
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
    std::map<uint64_t, icp_host::graph_entry> graphs;      // key: iterations << 3 | check << 2 | parity (+ fresh, + kind: see get_graph)
    uint32_t parity = 0;                         // tracking: which landmark buffers are the fixed / moving set (graphs hold pointers): frame f -> f mod 3
    // frame-to-frame tracking (icp_track_*): three landmark buffers in rotation, band staging, a copy stream
    float *lm[3] = { nullptr, nullptr, nullptr };            // landmarks of frame f live in lm[f mod 3] (lm[0] / lm[1] = the handle's F / M buffers)
    float *hBand[2] = { nullptr, nullptr }, *dBand[2] = { nullptr, nullptr };     // the part of a frame getLMs reads (ICP_BAND_*), pinned / device
    float *hFrame[2] = { nullptr, nullptr };                 // whole-frame pinned staging handed to the caller (icp_track_staging)
    struct host_range { const char *base; size_t bytes; };
    std::vector<host_range> sources;                         // the caller's own frame buffers, page-locked for DMA (icp_track_register_source)
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
    struct rbc_set { float *R = nullptr; float4 *GB = nullptr, *OL = nullptr, *LB = nullptr; float *XP = nullptr, *XQ = nullptr; uint32_t *rep_src = nullptr, *owner = nullptr, *N = nullptr, *O = nullptr,
                     *perm = nullptr, *chunk_hist = nullptr; uint2 *blist = nullptr; uint32_t *bn = nullptr; uint8_t *brank = nullptr; } rbc[2];
    bool rbc2_ready = false;
    bool track_last_gated = false;                // the form of the last submitted frame
    hipEvent_t evUp[2] = { nullptr, nullptr }, evDone[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t evFrame[2] = { nullptr, nullptr };           // the last upload out of the pinned frame buffer hFrame[k] (whatever frame parity it was submitted under)
    hipEvent_t evStage[3] = { nullptr, nullptr, nullptr };   // the last asynchronous copy out of the pinned staging of F / M / T (icp_write)
    uint64_t track_submitted = 0, track_collected = 0;       // frames fed / frames whose result has been handed out since init / icp_track_reset
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace icp_host __attribute__ ((visibility ("hidden"))) {

inline int fail (icp_context *h, int code, const std::string &msg)
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

// ---- icp_run.hip ---------------------------------------------------------------------------------------------------------------
void drop_graphs (icp_context *h);
int need (icp_context *h, bool built, bool keep_run = false);
int set_device (icp_context *h);
int get_graph (icp_context *h, uint32_t iterations, int check, hipGraphExec_t *out, bool fresh = false, bool with_build = false);
void run_launch_one (icp_context *h, run_ctl &r);
void run_enqueue_end (icp_context *h, run_ctl &r);
bool run_pump (icp_context *h, run_ctl &r);
int run_finish (icp_context *h, run_ctl &r, run_ctl *other);
inline int run_finish (icp_context *h) { return run_finish (h, h->run, h->run2.active ? &h->run2 : nullptr); }   // the run on the handle's own stream
int run_wait_final (icp_context *h, volatile unsigned long long *mirror, uint32_t n, uint32_t epoch, bool tracked = false);
int run_close_all (icp_context *h);
inline bool on_keeper_thread (const icp_context *h) { return h->keeper && h->keeper->started && std::this_thread::get_id () == h->keeper->tid; }
// the keeper (icp_track.hip): pause it and take the runs back (returns at once when there is none) / let it look after the open tracked runs
void keeper_quiesce (icp_context *h);
void keeper_kick (icp_context *h);
void keeper_stop (icp_context *h);
// first statement of every entry point that takes a handle (not icp_destroy, which ends the thread itself)
struct api_guard {
    icp_context *h;
    // (entry points call one another — icp_track_next -> icp_track_collect —: only the outermost call hands the runs back)
    explicit api_guard (icp_context *h_) : h (h_) { if (h && h->api_depth++ == 0 && h->keeper) keeper_quiesce (h); }
    ~api_guard () { if (h && --h->api_depth == 0 && h->keeper) keeper_kick (h); }
    api_guard (const api_guard &) = delete; api_guard &operator= (const api_guard &) = delete;
};
int launch_run (icp_context *h, uint32_t iterations, int check, bool fresh = false, bool with_build = false);
int settle (icp_context *h);
void note_enqueue (icp_context *h);
void note_inputs_change (icp_context *h);
void note_outputs_stored (icp_context *h);
int materialize_outputs (icp_context *h, int mem);

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

// Opens a checked run on the handle's stream with `blind` iterations enqueued at once (at least one).  p: the parameters of THIS run
// (tracking passes the frame's own landmark buffers); mirror / hstate: the host memory its words and final state go to.
// between (): enqueued after the RBC construction and in front of the first iteration (tracking: the waits and records that need not
// hold the construction back).
struct run_no_hook { int operator() () const { return ICP_OK; } };
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
    if (r.enq >= r.maxit) { r.decided = true; r.k_final = r.maxit + r.k0; run_enqueue_end (h, r); }
    HIPCHK (h, hipGetLastError ());
    h->hstate_fresh = false; h->hstate_here = false; h->k_base = -1;
    r.t[1] = now_s ();
    return ICP_OK;
}

}  // namespace icp_host
