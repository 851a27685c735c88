// icp_batch.cpp — icp_batch_*: B independent registrations spread over a list of devices, inside the library.
//
// SURVEY.md §8b ("batched twins icp_batch_* (B registrations, device list)") / §8e: a frame pair does not shard, so
// multi-GPU = replicas only.  Registration i lives on device slot i mod n as batch entry i / n of that slot's engine
// handle (icp_init_batched: one launch set per slot serves all its registrations); one host thread and one HIP stream
// per slot, pinned staging inside each handle, no collective, no peer access, no RCCL.  The reference has no
// counterpart (single context, single in-order queue: src/ICP/algorithms.cpp:4351-4352).
// Host code only: everything goes through the C-ABI of include/icp_amd.h.
#include "../../include/icp_amd.h"
#include "icp_cguard.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    std::mutex call_mx;                          // one caller at a time (job / pending / generation are single slots): a second thread waits
    std::vector<std::string> slot_cpus;          // what every worker was pinned to ("" = nowhere), icp_batch_slot_cpus
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
        try { b->rc[s] = b->count[s] ? b->job (s) : (int) ICP_OK; }
        catch (...) { b->rc[s] = icp_host::on_exception (); }          // (a throwing job must not take the process down: std::terminate on a worker)
        if (b->pending.fetch_sub (1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk (b->mx);       // (the caller may be on its way to sleep: the lock orders this notify behind its wait)
            b->cv_done.notify_one ();
        }
    }
}

// "0-3,8,10-11" -> CPU numbers (a malformed tail ends the list)
void parse_cpulist (const char *q, std::vector<int> &out)
{
    while (q && *q) {
        char *end = nullptr;
        const long a = std::strtol (q, &end, 10);
        if (end == q || a < 0) break;
        long z = a;
        if (*end == '-') { const char *r = end + 1; z = std::strtol (r, &end, 10); if (end == r || z < a) break; }
        for (long v = a; v <= z && out.size () < 4096; ++v) out.push_back ((int) v);
        if (*end != ',') break;
        q = end + 1;
    }
}

bool read_line (const std::string &path, std::string &line)
{
    FILE *f = std::fopen (path.c_str (), "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = std::fgets (buf, sizeof (buf), f) != nullptr;
    std::fclose (f);
    if (!ok) return false;
    line = buf;
    while (!line.empty () && (line.back () == '\n' || line.back () == '\r' || line.back () == ' ')) line.pop_back ();
    return true;
}

// The CPUs next to a PCI device: <root>/bus/pci/devices/<id>/local_cpulist, else the cpulist of the node <id>/numa_node names
// (<root>/devices/system/node/node<N>/cpulist); "" when the tree says nothing (a container without it, numa_node = -1).
std::string numa_cpulist_of (const std::string &root, const std::string &pci)
{
    std::string id = pci, line;
    for (char &ch : id) if (ch >= 'A' && ch <= 'F') ch = (char) (ch - 'A' + 'a');      // (sysfs spells bus ids in lower case)
    const std::string dev = root + "/bus/pci/devices/" + id;
    if (read_line (dev + "/local_cpulist", line) && !line.empty ()) return line;
    if (read_line (dev + "/numa_node", line)) {
        const long node = std::strtol (line.c_str (), nullptr, 10);
        if (node >= 0 && read_line (root + "/devices/system/node/node" + std::to_string (node) + "/cpulist", line)) return line;
    }
    return std::string ();
}

// One worker per slot.  Where it runs: ICP_AMD_SLOT_CPUS (a comma-separated list of CPU numbers or ranges; slot s takes entry s mod
// length) if set; else the CPUs of the NUMA node its GPU hangs on (sysfs through the device's PCI bus id: on a two-socket 8-GPU node the
// thread that drives a GPU belongs on that GPU's socket — its launches and the polled host words then stay off the inter-socket link),
// intersected with what the process may use; else, silently, wherever the scheduler puts it.  ICP_AMD_SLOT_NUMA=0 switches the default off.
int start_workers (icp_batch_context *b)
{
    const size_t n = b->slots.size ();
    b->rc.assign (n, ICP_OK);
    b->slot_cpus.assign (n, std::string ());
    std::vector<std::vector<int>> want (n);
    if (const char *e = std::getenv ("ICP_AMD_SLOT_CPUS")) {
        std::vector<std::string> items;
        std::string cur;
        for (const char *q = e;; ++q) { if (*q == ',' || !*q) { if (!cur.empty ()) items.push_back (cur); cur.clear (); if (!*q) break; } else cur.push_back (*q); }
        for (size_t s = 0; s < n && !items.empty (); ++s) parse_cpulist (items[s % items.size ()].c_str (), want[s]);
    } else {
        const char *off = std::getenv ("ICP_AMD_SLOT_NUMA");
        const char *root = std::getenv ("ICP_AMD_SYSFS_ROOT");
        if (!(off && off[0] == '0'))
            for (size_t s = 0; s < n; ++s) {
                char id[64] = { 0 };
                if (icp_device_pci_bus_id (b->devices[s], id, sizeof (id)) != ICP_OK) continue;
                parse_cpulist (numa_cpulist_of (root ? root : "/sys", id).c_str (), want[s]);
            }
    }
    cpu_set_t allowed; CPU_ZERO (&allowed);
    const bool have_allowed = sched_getaffinity (0, sizeof (allowed), &allowed) == 0;
    try {
        for (size_t s = 0; s < n; ++s) {
            b->workers.emplace_back (worker_main, b, s);
            cpu_set_t set; CPU_ZERO (&set);
            std::string text;
            for (int c : want[s])
                if (c < CPU_SETSIZE && (!have_allowed || CPU_ISSET (c, &allowed))) { CPU_SET (c, &set); text += (text.empty () ? "" : ",") + std::to_string (c); }
            if (!text.empty () && pthread_setaffinity_np (b->workers.back ().native_handle (), sizeof (set), &set) == 0) b->slot_cpus[s] = text;
            // (no usable CPU in the list, or the call refused: the thread stays where the scheduler puts it, and slot_cpus says so)
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
    std::lock_guard<std::mutex> call (b->call_mx);
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

int icp_batch_partition (uint32_t registrations, uint32_t n_slots, uint32_t i, uint32_t *slot, uint32_t *index, uint32_t *slot_count) try
{
    if (n_slots == 0 || i >= registrations) return ICP_EINVAL;
    const uint32_t s = i % n_slots;
    if (slot) *slot = s;
    if (index) *index = i / n_slots;
    if (slot_count) *slot_count = (registrations - s + n_slots - 1u) / n_slots;    // registrations s, s + n, s + 2n, ..
    return ICP_OK;
}
ICP_CATCH_ALL

const char *icp_batch_last_error (icp_batch_handle b) { return b ? b->err.c_str () : g_batch_create_error.c_str (); }

int icp_batch_create (icp_batch_handle *out, const int *devices, int n_devices, int rot, int weighted) try
{
    if (!out) return ICP_EINVAL;
    *out = nullptr;
    if (!devices || n_devices <= 0) return bfail (nullptr, ICP_EINVAL, "icp_batch_create: empty device list");
    struct guard {                                   // whatever leaves this function early — a status or an exception — takes the half-built object with it
        icp_batch_context *b;
        ~guard () { if (b) { stop_workers (b); for (icp_handle q : b->slots) icp_destroy (q); delete b; } }
    } g { new icp_batch_context () };
    icp_batch_context *b = g.b;
    b->rot = rot; b->weighted = weighted;
    b->devices.reserve ((size_t) n_devices); b->slots.reserve ((size_t) n_devices);
    for (int d = 0; d < n_devices; ++d) {
        icp_handle h = nullptr;
        int rc = icp_create (&h, devices[d], rot, weighted);
        if (rc != ICP_OK) return bfail (nullptr, rc, "icp_batch_create: device " + std::to_string (devices[d]) + ": " + icp_last_error (nullptr));
        b->devices.push_back (devices[d]); b->slots.push_back (h);          // (reserved above: cannot throw between icp_create and here)
    }
    b->count.assign (b->slots.size (), 0u);
    if (start_workers (b) != ICP_OK) return bfail (nullptr, ICP_ENOMEM, "icp_batch_create: a host thread per device slot could not be started");
    g.b = nullptr;
    *out = b;
    return ICP_OK;
}
ICP_CATCH_ALL

// The CPUs the worker of a slot was pinned to, as a comma-separated list ("" = not pinned: no ICP_AMD_SLOT_CPUS entry, no NUMA
// information for its device, or none of the CPUs is available to the process).
int icp_batch_slot_cpus (icp_batch_handle b, uint32_t slot, char *out, size_t cap) try
{
    if (!b || !out || cap == 0 || slot >= b->slots.size ()) return ICP_EINVAL;
    const std::string &t = b->slot_cpus[slot];
    if (t.size () + 1 > cap) return bfail (b, ICP_EINVAL, "icp_batch_slot_cpus: the buffer is too small");
    std::memcpy (out, t.c_str (), t.size () + 1);
    return ICP_OK;
}
ICP_CATCH_ALL

// The cpulist of the NUMA node a PCI device hangs on, read from a sysfs tree (sysfs_root NULL = "/sys"): what icp_batch_create pins a
// slot's worker to by default.  out = "" when the tree has no answer.
int icp_numa_cpulist (const char *sysfs_root, const char *pci_bus_id, char *out, size_t cap) try
{
    if (!pci_bus_id || !out || cap == 0) return ICP_EINVAL;
    const std::string t = numa_cpulist_of (sysfs_root ? sysfs_root : "/sys", pci_bus_id);
    if (t.size () + 1 > cap) return ICP_EINVAL;
    std::memcpy (out, t.c_str (), t.size () + 1);
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_destroy (icp_batch_handle b) try
{
    if (!b) return ICP_EINVAL;
    stop_workers (b);
    for (icp_handle h : b->slots) icp_destroy (h);
    delete b;
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_size (icp_batch_handle b, uint32_t *registrations, uint32_t *n_slots) try
{
    if (!b) return ICP_EINVAL;
    if (registrations) *registrations = b->registrations;
    if (n_slots) *n_slots = (uint32_t) b->slots.size ();
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_init (icp_batch_handle b, uint32_t registrations, uint32_t m, uint32_t nr, float a, float c,
                    uint32_t max_iterations, double angle_threshold, double translation_threshold) try
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
ICP_CATCH_ALL

#define BATCH_SLOT(b, i)                                                                              \
    if (!(b)) return ICP_EINVAL;                                                                      \
    if (!(b)->inited) return bfail ((b), ICP_ESTATE, "icp_batch_init has not been called");           \
    if ((i) >= (b)->registrations) return bfail ((b), ICP_EINVAL, "registration index out of range"); \
    const uint32_t slot_ = (i) % (uint32_t) (b)->slots.size (), idx_ = (i) / (uint32_t) (b)->slots.size (); \
    icp_handle h_ = (b)->slots[slot_];

#define BATCH_CALL(b, expr)                                                                           \
    do { int rc_ = (expr); if (rc_ != ICP_OK) return bfail ((b), rc_, icp_last_error (h_)); } while (0)

int icp_batch_write (icp_batch_handle b, uint32_t i, int mem, const void *host_ptr) try
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_write_b (h_, idx_, mem, host_ptr, 0));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_read (icp_batch_handle b, uint32_t i, int mem, void *host_dst, size_t bytes) try
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_read_b (h_, idx_, mem, host_dst, bytes));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_state (icp_batch_handle b, uint32_t i, icp_state_t *out) try
{
    BATCH_SLOT (b, i)
    BATCH_CALL (b, icp_state_b (h_, idx_, out));
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_set_modes (icp_batch_handle b, int reduce_mode, int power_mode) try
{
    if (!b) return ICP_EINVAL;
    for (icp_handle h : b->slots) {
        int rc = icp_set_reduce_mode (h, reduce_mode); if (rc == ICP_OK) rc = icp_set_power_mode (h, power_mode);
        if (rc != ICP_OK) return bfail (b, rc, icp_last_error (h));
    }
    return ICP_OK;
}
ICP_CATCH_ALL

int icp_batch_build_rbc (icp_batch_handle b) try
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) { int rc = icp_build_rbc (b->slots[s]); return rc ? rc : icp_sync (b->slots[s]); });
}
ICP_CATCH_ALL

int icp_batch_run (icp_batch_handle b) try
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) { return icp_run (b->slots[s], nullptr); });
}
ICP_CATCH_ALL

int icp_batch_run_fixed (icp_batch_handle b, uint32_t iterations, int from_identity) try
{
    if (!b || !b->inited) return b ? bfail (b, ICP_ESTATE, "icp_batch_init has not been called") : ICP_EINVAL;
    return for_each_slot (b, [&] (size_t s) {
        int rc = from_identity ? icp_reset_transform (b->slots[s]) : ICP_OK;
        if (rc == ICP_OK) rc = icp_run_fixed (b->slots[s], iterations);
        return rc ? rc : icp_sync (b->slots[s]);
    });
}
ICP_CATCH_ALL

int icp_batch_time_run_fixed_slots (icp_batch_handle b, uint32_t iterations, uint32_t reps, uint32_t warmup, double *seconds, float *slot_ms) try
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
ICP_CATCH_ALL

int icp_batch_time_run_fixed (icp_batch_handle b, uint32_t iterations, uint32_t reps, double *seconds) try
{
    return icp_batch_time_run_fixed_slots (b, iterations, reps, 1u, seconds, nullptr);
}
ICP_CATCH_ALL

}  // extern "C"
