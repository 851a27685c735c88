// icp_batch.cpp — icp_batch_*: B independent registrations spread over a list of devices, inside the library.
//
// SURVEY.md §8b ("batched twins icp_batch_* (B registrations, device list)") / §8e: a frame pair does not shard, so
// multi-GPU = replicas only.  Registration i lives on device slot i mod n as batch entry i / n of that slot's engine
// handle (icp_init_batched: one launch set per slot serves all its registrations); one host thread and one HIP stream
// per slot, pinned staging inside each handle, no collective, no peer access, no RCCL.  The reference has no
// counterpart (single context, single in-order queue: src/ICP/algorithms.cpp:4351-4352).
// Host code only: everything goes through the C-ABI of include/icp_amd.h.
#include "../../include/icp_amd.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

// One host thread per device slot, for the life of the batch object (round 5; rounds 1 - 4 spawned and joined a thread per slot in
// every icp_batch_* call — eight creations and joins per call are noise beside 64 x 40 iterations and a visible fraction of a converged
// warm batch, and not what a service loop over eight GPUs wants).  A call publishes a job (a generation number + the function), every
// worker runs it for its slot, the caller waits for the count of pending slots to reach zero.  Both sides spin briefly before they sleep
// on a condition variable: back-to-back calls hand over in a microsecond or two, an idle batch object costs no CPU.
// ICP_AMD_SLOT_CPUS: a comma-separated list of CPU numbers; the worker of slot s is pinned to entry s mod (length) — on a multi-socket
// host the thread that drives a GPU belongs on that GPU's NUMA node.
struct icp_batch_context {
    std::vector<int> devices;                    // device ordinal of every slot (ordinals may repeat)
    std::vector<icp_handle> slots;               // one engine handle per slot
    std::vector<uint32_t> count;                 // registrations of every slot
    uint32_t registrations = 0;
    int rot = 1, weighted = 1;
    bool inited = false;
    std::string err;
    // the slot workers
    std::vector<std::thread> workers;
    std::mutex mx;
    std::condition_variable cv_job, cv_done;
    std::atomic<uint64_t> generation { 0 };
    std::atomic<int> pending { 0 };
    std::atomic<bool> stop { false };
    std::function<int (size_t)> job;
    std::vector<int> rc;
};

namespace {

thread_local std::string g_batch_create_error;

int bfail (icp_batch_context *b, int code, const std::string &msg)
{
    if (b) b->err = msg; else g_batch_create_error = msg;
    return code;
}

constexpr int SPIN = 4000;                       // ~20 - 40 us of _mm_pause before a side goes to sleep

void worker_main (icp_batch_context *b, size_t s)
{
    uint64_t seen = 0;
    for (;;) {
        int spins = 0;
        while (b->generation.load (std::memory_order_acquire) == seen && !b->stop.load (std::memory_order_acquire)) {
            if (++spins < SPIN) { _mm_pause (); continue; }
            std::unique_lock<std::mutex> lk (b->mx);
            b->cv_job.wait (lk, [&] { return b->generation.load (std::memory_order_acquire) != seen || b->stop.load (std::memory_order_acquire); });
        }
        if (b->stop.load (std::memory_order_acquire)) return;
        seen = b->generation.load (std::memory_order_acquire);
        b->rc[s] = b->count[s] ? b->job (s) : (int) ICP_OK;
        if (b->pending.fetch_sub (1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk (b->mx);       // (the caller may be on its way to sleep: the lock orders this notify behind its wait)
            b->cv_done.notify_one ();
        }
    }
}

int start_workers (icp_batch_context *b)
{
    const size_t n = b->slots.size ();
    b->rc.assign (n, ICP_OK);
    std::vector<int> cpus;
    if (const char *e = std::getenv ("ICP_AMD_SLOT_CPUS")) {
        const char *q = e;
        while (*q) { char *end = nullptr; const long v = std::strtol (q, &end, 10); if (end == q) break; if (v >= 0) cpus.push_back ((int) v); q = (*end == ',') ? end + 1 : end; if (*end && *end != ',') break; }
    }
    try {
        for (size_t s = 0; s < n; ++s) {
            b->workers.emplace_back (worker_main, b, s);
            if (!cpus.empty ()) {
                cpu_set_t set; CPU_ZERO (&set); CPU_SET (cpus[s % cpus.size ()], &set);
                (void) pthread_setaffinity_np (b->workers.back ().native_handle (), sizeof (set), &set);      // (a CPU that is not there: the thread stays where it is)
            }
        }
    } catch (const std::system_error &) { return ICP_ENOMEM; }
    return ICP_OK;
}

void stop_workers (icp_batch_context *b)
{
    { std::lock_guard<std::mutex> lk (b->mx); b->stop.store (true, std::memory_order_release); }
    b->cv_job.notify_all ();
    for (auto &t : b->workers) if (t.joinable ()) t.join ();
    b->workers.clear ();
}

// runs fn (slot) on the worker of every slot that holds registrations; returns the first failing status
template <typename Fn>
int for_each_slot (icp_batch_context *b, Fn &&fn)
{
    const size_t n = b->slots.size ();
    b->job = std::forward<Fn> (fn);
    b->pending.store ((int) n, std::memory_order_release);
    { std::lock_guard<std::mutex> lk (b->mx); b->generation.fetch_add (1, std::memory_order_acq_rel); }
    b->cv_job.notify_all ();
    int spins = 0;
    while (b->pending.load (std::memory_order_acquire) != 0) {
        if (++spins < SPIN) { _mm_pause (); continue; }
        std::unique_lock<std::mutex> lk (b->mx);
        b->cv_done.wait (lk, [&] { return b->pending.load (std::memory_order_acquire) == 0; });
    }
    b->job = nullptr;
    for (size_t s = 0; s < n; ++s)
        if (b->rc[s] != ICP_OK) return bfail (b, b->rc[s], "slot " + std::to_string (s) + " (device " + std::to_string (b->devices[s]) + "): " + icp_last_error (b->slots[s]));
    return ICP_OK;
}

}  // namespace

extern "C" {

int icp_batch_partition (uint32_t registrations, uint32_t n_slots, uint32_t i, uint32_t *slot, uint32_t *index, uint32_t *slot_count)
{
    if (n_slots == 0 || i >= registrations) return ICP_EINVAL;
    const uint32_t s = i % n_slots;
    if (slot) *slot = s;
    if (index) *index = i / n_slots;
    if (slot_count) *slot_count = (registrations - s + n_slots - 1u) / n_slots;    // registrations s, s + n, s + 2n, ..
    return ICP_OK;
}

const char *icp_batch_last_error (icp_batch_handle b) { return b ? b->err.c_str () : g_batch_create_error.c_str (); }

int icp_batch_create (icp_batch_handle *out, const int *devices, int n_devices, int rot, int weighted)
{
    if (!out) return ICP_EINVAL;
    *out = nullptr;
    if (!devices || n_devices <= 0) return bfail (nullptr, ICP_EINVAL, "icp_batch_create: empty device list");
    icp_batch_context *b = new icp_batch_context ();
    b->rot = rot; b->weighted = weighted;
    for (int d = 0; d < n_devices; ++d) {
        icp_handle h = nullptr;
        int rc = icp_create (&h, devices[d], rot, weighted);
        if (rc != ICP_OK) {
            std::string m = icp_last_error (nullptr);
            for (icp_handle q : b->slots) icp_destroy (q);
            delete b;
            return bfail (nullptr, rc, "icp_batch_create: device " + std::to_string (devices[d]) + ": " + m);
        }
        b->devices.push_back (devices[d]); b->slots.push_back (h);
    }
    b->count.assign (b->slots.size (), 0u);
    if (start_workers (b) != ICP_OK) {
        stop_workers (b);
        for (icp_handle q : b->slots) icp_destroy (q);
        delete b;
        return bfail (nullptr, ICP_ENOMEM, "icp_batch_create: a host thread per device slot could not be started");
    }
    *out = b;
    return ICP_OK;
}

int icp_batch_destroy (icp_batch_handle b)
{
    if (!b) return ICP_EINVAL;
    stop_workers (b);
    for (icp_handle h : b->slots) icp_destroy (h);
    delete b;
    return ICP_OK;
}

int icp_batch_size (icp_batch_handle b, uint32_t *registrations, uint32_t *n_slots)
{
    if (!b) return ICP_EINVAL;
    if (registrations) *registrations = b->registrations;
    if (n_slots) *n_slots = (uint32_t) b->slots.size ();
    return ICP_OK;
}

int icp_batch_init (icp_batch_handle b, uint32_t registrations, uint32_t m, uint32_t nr, float a, float c,
                    uint32_t max_iterations, double angle_threshold, double translation_threshold)
{
    if (!b) return ICP_EINVAL;
    if (registrations == 0) return bfail (b, ICP_EINVAL, "icp_batch_init: no registrations");
    const uint32_t n = (uint32_t) b->slots.size ();
    b->inited = false;
    for (uint32_t s = 0; s < n; ++s) b->count[s] = s < registrations ? (registrations - s + n - 1u) / n : 0u;
    b->registrations = registrations;
    int rc = for_each_slot (b, [&] (size_t s) {
        return icp_init_batched (b->slots[s], b->count[s], m, nr, a, c, max_iterations, angle_threshold, translation_threshold);
    });
    if (rc == ICP_OK) b->inited = true;
    return rc;
}

#define BATCH_SLOT(b, i)                                                                              \
    if (!(b)) return ICP_EINVAL;                                                                      \
    if (!(b)->inited) return bfail ((b), ICP_ESTATE, "icp_batch_init has not been called");           \
    if ((i) >= (b)->registrations) return bfail ((b), ICP_EINVAL, "registration index out of range"); \
    const uint32_t slot_ = (i) % (uint32_t) (b)->slots.size (), idx_ = (i) / (uint32_t) (b)->slots.size (); \
    icp_handle h_ = (b)->slots[slot_];

#define BATCH_CALL(b, expr)                                                                           \
    do { int rc_ = (expr); if (rc_ != ICP_OK) return bfail ((b), rc_, icp_last_error (h_)); } while (0)

int icp_batch_write (icp_batch_handle b, uint32_t i, int mem, const void *host_ptr)
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_write_b (h_, idx_, mem, host_ptr, 0));
    return ICP_OK;
}

int icp_batch_read (icp_batch_handle b, uint32_t i, int mem, void *host_dst, size_t bytes)
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_read_b (h_, idx_, mem, host_dst, bytes));
    return ICP_OK;
}

int icp_batch_state (icp_batch_handle b, uint32_t i, icp_state_t *out)
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_state_b (h_, idx_, out));
    return ICP_OK;
}

int icp_batch_set_modes (icp_batch_handle b, int reduce_mode, int power_mode)
{
    if (!b) return ICP_EINVAL;
    for (icp_handle h : b->slots) {
        int rc = icp_set_reduce_mode (h, reduce_mode); if (rc == ICP_OK) rc = icp_set_power_mode (h, power_mode);
        if (rc != ICP_OK) return bfail (b, rc, icp_last_error (h));
    }
    return ICP_OK;
}

int icp_batch_build_rbc (icp_batch_handle b)
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) { int rc = icp_build_rbc (b->slots[s]); return rc ? rc : icp_sync (b->slots[s]); });
}

int icp_batch_run (icp_batch_handle b)
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) { return icp_run (b->slots[s], nullptr); });
}

int icp_batch_run_fixed (icp_batch_handle b, uint32_t iterations, int from_identity)
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) {
        int rc = from_identity ? icp_reset_transform (b->slots[s]) : ICP_OK;
        if (rc == ICP_OK) rc = icp_run_fixed (b->slots[s], iterations);
        return rc ? rc : icp_sync (b->slots[s]);
    });
}

int icp_batch_time_run_fixed_slots (icp_batch_handle b, uint32_t iterations, uint32_t reps, uint32_t warmup, double *seconds, float *slot_ms)
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    if (!seconds || iterations == 0 || reps == 0) return bfail (b, ICP_EINVAL, "bad arguments");
    // Every slot's host thread: `warmup` untimed passes (the first captures and instantiates the graph), its stream drained, then
    // it waits at the gate; the gate opens when all slots are there (the barrier in front of the timed region), the clock starts,
    // every slot runs its `reps` passes (HIP events on its own stream around them: slot_ms) and drains its stream; the clock stops
    // when the last slot is done: wall time = max over devices.
    const size_t n = b->slots.size ();
    size_t active = 0;
    for (size_t s = 0; s < n; ++s) active += b->count[s] ? 1u : 0u;
    // the barrier in front of the timed region lives in the slot threads themselves: the last one to arrive takes t0 and opens it
    std::atomic<size_t> arrived { 0 };
    std::atomic<int> go { 0 };
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now ();
    std::vector<float> ms (n, 0.f);
    int rc = for_each_slot (b, [&] (size_t s) {
        int r = ICP_OK;
        for (uint32_t w = 0; w < warmup && r == ICP_OK; ++w) r = icp_run_fixed_fresh (b->slots[s], iterations);
        if (r == ICP_OK) r = icp_sync (b->slots[s]);
        if (arrived.fetch_add (1, std::memory_order_acq_rel) + 1 == active) {      // (also on failure: the barrier must open for the others)
            t0 = std::chrono::steady_clock::now ();
            go.store (1, std::memory_order_release);
        }
        while (!go.load (std::memory_order_acquire)) std::this_thread::yield ();
        if (r != ICP_OK) return r;
        uint32_t timed = 0;                                           // (events behind the first pass: see icp_time_run_fixed_tail)
        r = icp_time_run_fixed_tail (b->slots[s], iterations, reps, 1, &ms[s], &timed);
        if (r == ICP_OK && timed) ms[s] = ms[s] * (float) reps / (float) timed;      // per-slot figure over all `reps` passes at the timed passes' rate
        return r ? r : icp_sync (b->slots[s]);
    });
    *seconds = std::chrono::duration<double> (std::chrono::steady_clock::now () - t0).count ();
    if (slot_ms) for (size_t s = 0; s < n; ++s) slot_ms[s] = ms[s];
    return rc;
}

int icp_batch_time_run_fixed (icp_batch_handle b, uint32_t iterations, uint32_t reps, double *seconds)
{
    return icp_batch_time_run_fixed_slots (b, iterations, reps, 1u, seconds, nullptr);
}

}  // extern "C"
