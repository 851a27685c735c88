// icp_run.hip — how a run reaches the device: the cache of fixed-length run graphs (one hipGraph of n chained launches, updated in place
// when a parameter changes) and the host-driven checked run (ICP::run — src/ICP/algorithms.cpp:4806-4834: iterate until check () says
// stop): plain launches, the device's (epoch, k, done) word polled in fine-grained host memory, `depth` launches kept queued, the final
// state delivered to host memory by the launch that finds the run converged.  Shared with icp_capi.hip and icp_track.hip through icp_host.h.
#include "icp_host.h"

namespace icp_host {

void drop_graphs (icp_context *h)
{
    for (auto &kv : h->graphs) {
        if (kv.second.exec) (void) hipGraphExecDestroy (kv.second.exec);
        if (kv.second.graph) (void) hipGraphDestroy (kv.second.graph);
    }
    h->graphs.clear ();
}

// keep_run: the caller is one of the tracking entries, which carry an open checked run (run_ctl) across calls themselves; everything
// else that touches the handle's stream first brings an open run to its end (its remaining launches must not interleave with others)
int need (icp_context *h, bool built, bool keep_run)
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

#define ICP_GRAPH_CACHE 8u       // cached run graphs per handle (least recently used goes first)

// Graph of `iterations` iterations, cached (fixed-length runs: icp_run_fixed*, the timing entries; checked runs only with
// ICP_AMD_RUN_ADAPTIVE=0).  A parameter change (setAlpha, setScaling, thresholds, modes) does not throw the executable graphs away: an
// entry of an older parameter generation is re-captured and its executable graph UPDATED in place (hipGraphExecUpdate: the kernel
// nodes' arguments; instantiating anew costs milliseconds) — same topology by construction, re-instantiated only if the update is refused.
// fresh: the graph starts the registration from the identity transform (icp_reset_transform + the run as one graph; the
// chained form folds the reset into its first launch).  with_build: buildRBC in front of the run.
int get_graph (icp_context *h, uint32_t iterations, int check, hipGraphExec_t *out, bool fresh, bool with_build)
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

// The end kernel of a run that did not (or may not) end by itself: final state -> p.st and -> host memory, FINAL bit, the release of the
// tracking sequence word.  It goes behind the last iteration launch the moment the queue has reached max_iterations — nothing more will be
// enqueued, and a tracked frame's successor (gated on the device) must not wait for the host's next visit —; a run that converged earlier
// has left its final state already and the kernel returns at its first load (k_chain_end: run_flag).
void run_enqueue_end (icp_context *h, run_ctl &r)
{
    (void) h;
    if (r.end_enqueued || r.final_seen) return;
    if (r.chained) icp_launch_chain_end (r.p, r.stream, r.enq);
    else icp_launch_publish_state (r.p, r.stream);
    r.end_enqueued = true;
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
    if (r.enq >= r.maxit) { r.decided = true; r.k_final = r.maxit + r.k0; run_enqueue_end (h, r); }
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
        if (on_keeper_thread (h) && h->keeper->pause.load (std::memory_order_relaxed)) return ICP_KEEPER_ABORTED;      // (an API call wants the runs back: they stay open)
        if (other && other->active) { const uint32_t ko = other->k_seen; (void) run_pump (h, *other); if (other->k_seen != ko || other->decided) t_last = std::chrono::steady_clock::now (); }
        _mm_pause ();
        if ((++spins & 0x3FFu) == 0u) {
            const auto now = std::chrono::steady_clock::now ();
            if (r.k_seen != k_last) { k_last = r.k_seen; t_last = now; }
            else if (r.k0 && std::chrono::duration<double> (now - t_last).count () > 0.05) { r.k0 = 0u; t_last = now; }      // (a stale idea of where the count began: pace on k itself)
            else if (r.track_slot >= 0 && h->hGateFlag && *h->hGateFlag) {      // (a tracked frame only: a plain run on the handle does not depend on any gate)
                r.active = false;
                return fail (h, ICP_EHIP, "tracking: the device made no progress for ~0.5 s (a frame's gate gave up; the frames behind it were skipped): icp_track_reset starts a new sequence");
            }
            else if (std::chrono::duration<double> (now - t_last).count () > 20.0) {
                r.active = false;
                return fail (h, ICP_EHIP, "checked run: the device has published no progress for 20 s");
            }
        }
    }
    r.t[3] = now_s ();
    // (converged in the fused forms: the finalize that set the flag has left the final state in p.st and in host memory already)
    run_enqueue_end (h, r);
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

// Waits for the FINAL bit of `n` words of epoch `epoch` (the end kernel's last store: the final states are in host memory).
int run_wait_final (icp_context *h, volatile unsigned long long *mirror, uint32_t n, uint32_t epoch, bool tracked)
{
    uint32_t spins = 0;
    const auto t0 = std::chrono::steady_clock::now ();
    for (uint32_t b = 0; b < n; ++b) {
        for (;;) {
            const unsigned long long w = mirror[b];
            if ((uint32_t) (w >> 32) == epoch && (w & ICP_MIRROR_FINAL)) break;
            _mm_pause ();
            if ((++spins & 0x3FFFu) == 0u && tracked && h->hGateFlag && *h->hGateFlag)
                return fail (h, ICP_EHIP, "tracking: the device made no progress for ~0.5 s (a frame's gate gave up; the frames behind it were skipped): icp_track_reset starts a new sequence");
            if ((spins & 0x3FFFu) == 0u && std::chrono::duration<double> (std::chrono::steady_clock::now () - t0).count () > 60.0) {
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
int launch_run (icp_context *h, uint32_t iterations, int check, bool fresh, bool with_build)
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

static bool is_query_output (int mem) { return mem == ICP_MEM_NN_ID || mem == ICP_MEM_W || mem == ICP_MEM_NN || mem == ICP_MEM_QT || mem == ICP_MEM_RID; }

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

}  // namespace icp_host
