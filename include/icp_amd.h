/* icp_amd.h — C-ABI of the MI355X-native photogeometric ICP iteration engine.
 *
 * Drop-in boundary for the hot path of nlamprian/ICP: every entry point below names the
 * reference interface it replaces (paths relative to the reference checkout).  Plain C: opaque
 * handle, plain pointers and sizes, int status codes; no OpenCL, Eigen, CLUtils or torch types.
 * The C++ facade include/ICP/algorithms.hpp re-creates the reference's class templates
 * (cl_algo::ICP::ICPStep<CR,CW>, ICP<CR,CW>) on top of this file.
 *
 * Data layouts (unchanged from the reference):
 *   landmark   float[8]  = [x y z 1 r g b 1], xyz in mm, rgb in [0,1]   (src/kinect_frame_grabber.cpp:252-261)
 *   transform  float[8]  = [qx qy qz qw | tx ty tz s]                   (include/ICP/algorithms.hpp:2245-2254)
 *   dist/id    {float dist; uint32 id}                                  (kernels/icp_kernels.cl:34-38)
 *
 * Threading: a handle owns one device, one HIP stream and all its buffers; it is not
 * thread-safe, distinct handles are independent (one per GPU / host thread for batched work).
 * Errors: no exit(), no exceptions: every call returns an icp_status; icp_last_error() gives
 * the text (reference: exit(EXIT_FAILURE) from library code, src/ICP/algorithms.cpp:4411-4427).
 */
#ifndef ICP_AMD_H
#define ICP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with hidden visibility; what this header declares is what it exports. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct icp_context *icp_handle;

typedef enum {
    ICP_OK = 0,
    ICP_EINVAL = 1,        /* bad argument (reference: throw const char* -> exit)            */
    ICP_EHIP = 2,          /* HIP runtime failure (reference: cl::Error exception)           */
    ICP_ENOMEM = 3,
    ICP_ESTATE = 4,        /* call order violated (e.g. run before init / buildRBC)          */
    ICP_ENODEVICE = 5      /* no usable gfx950 device: the engine has NO CPU fallback        */
} icp_status;

/* ICPStepConfigT / ICPStepConfigW — include/ICP/algorithms.hpp:1544-1564 */
typedef enum { ICP_ROT_EIGEN = 0, ICP_ROT_POWER_METHOD = 1 } icp_rot;
typedef enum { ICP_W_REGULAR = 0, ICP_W_WEIGHTED = 1 } icp_weighting;

/* How the power method starts.  LITERAL = the reference loop from x=(1,1,1,1)
 * (kernels/icp_kernels.cl:1003-1022); SQUARED = same loop started from normalize(N^1024 * 1).
 * Default of icp_create: SQUARED (the benchmarked path). */
typedef enum { ICP_POWER_LITERAL = 0, ICP_POWER_SQUARED = 1 } icp_power_mode;

/* How the three reductions of an iteration are evaluated.  Default of icp_create: FUSED (the
 * benchmarked path: 1 launch per iteration at latency-bound sizes, 2 otherwise); REFERENCE_ORDER (4 launches per
 * iteration) is the opt-in for intermediates that restate the reference's arithmetic order.  The environment variable
 * ICP_AMD_MODE=reference, read by icp_create, starts handles in REFERENCE_ORDER + LITERAL without a code change.
 * Contract between the two: final [q | t, s] within 1e-5 relative (|q| = 1, scene scale for t, s itself).
 * REFERENCE_ORDER: sum of weights -> means -> S as three global trees in the reference's order
 * (kernels/icp_kernels.cl:213-329, 455-566, 588-743); every intermediate matches the literal oracle.
 * FUSED: one pass accumulating 18 moments in double, means and S derived from them; one global tree;
 * correspondences identical for identical T, means / S / T within 1 ulp of the coordinates. */
typedef enum { ICP_REDUCE_REFERENCE_ORDER = 0, ICP_REDUCE_FUSED = 1 } icp_reduce_mode;

/* Memory objects.  F/M/T are the reference's ICPStep::Memory D_IN_F / D_IN_M / D_IO_T
 * (include/ICP/algorithms.hpp:2241-2267); the rest are the intermediates the reference exposes
 * through the get() of its sub-objects (src/ICP/algorithms.cpp:4499-4581). */
typedef enum {
    ICP_MEM_F = 0,         /* in   m x float8   fixed landmarks                         */
    ICP_MEM_M = 1,         /* in   m x float8   moving landmarks                        */
    ICP_MEM_T = 2,         /* io   float8       cumulative [q | t, s]   (D_IO_T)        */
    ICP_MEM_TK = 3,        /* out  float8       incremental [qk | tk, sk]               */
    ICP_MEM_MEANS = 4,     /* out  2 x float4   [mean_fixed | mean_moving]              */
    ICP_MEM_S = 5,         /* out  float[11]    S (row-major, a=moving,b=fixed), Sf2, Sm2 */
    ICP_MEM_NN_ID = 6,     /* out  m x {dist,id}  query order, id = index into F        */
    ICP_MEM_W = 7,         /* out  m x float    weights 100/(100+dist)                  */
    ICP_MEM_SUM_W = 8,     /* out  double       sum of weights                          */
    ICP_MEM_REPS = 9,      /* out  nr x float8  representatives                         */
    ICP_MEM_RBC_N = 10,    /* out  nr x uint32  list sizes        (RBCConstruct D_OUT_N)   */
    ICP_MEM_RBC_O = 11,    /* out  nr x uint32  list offsets      (RBCConstruct D_OUT_O)   */
    ICP_MEM_RBC_PERM = 12, /* out  m x uint32   list position -> index into F           */
    ICP_MEM_RBC_OWNER = 13,/* out  m x uint32   owner representative of each fixed point */
    ICP_MEM_RBC_XP = 14,   /* out  m x float8   permuted database (RBCConstruct D_OUT_X_P) */
    ICP_MEM_RID = 15,      /* out  m x uint32   nearest representative of each query    */
    ICP_MEM_R = 16,        /* out  float[9]     cumulative rotation, row-major          */
    ICP_MEM_RK = 17,       /* out  float[9]     incremental rotation, row-major         */
    ICP_MEM_NN = 18,       /* out  m x float4   matched fixed xyz (+ weight in .w)      */
    ICP_MEM_QT = 19,       /* out  m x float4   transformed moving xyz (+ dist in .w)   */
    ICP_MEM_COUNT_
} icp_mem;

/* Host-visible state of one registration: the public members Rk qk tk sk R q t s k of
 * ICPStep / ICP (include/ICP/algorithms.hpp:2302-2320, 2462). */
typedef struct {
    float R[9], q[4], t[3], s;         /* cumulative, up to iteration k   */
    float Rk[9], qk[4], tk[3], sk;     /* incremental, iteration k        */
    uint32_t k;                        /* iterations executed             */
    uint32_t converged;                /* ICP::check() said stop before max_iterations */
    uint32_t power_iterations;         /* power-method loop trips of the last step */
    uint32_t reserved;
} icp_state_t;

/* ---- life cycle ------------------------------------------------------------------------- */

/* ICPStep<CR,CW>::ICPStep (env, infoRBC, infoICP) — include/ICP/algorithms.hpp:2269,
 * src/ICP/algorithms.cpp:4348-4358.  `device` replaces the CLEnv/CLEnvInfo pair.  The handle starts in the modes
 * ICP_REDUCE_FUSED + ICP_POWER_SQUARED (see above). */
int icp_create (icp_handle *h, int device, int rot, int weighted);
int icp_destroy (icp_handle h);

/* ICP<CR,CW>::init (m, nr, a, c, max_iterations, angle_threshold, translation_threshold, staging)
 * — include/ICP/algorithms.hpp:2437-2440, src/ICP/algorithms.cpp:4777-4786 and ICPStep::init
 * :4403-4582.  Rejects m == 0, nr == 0, a == 0 (reference :4413-4420), odd m (:1573),
 * nr not a power of two or not tiling the sqrt(m) landmark grid (:842-854).
 * `batch` >= 1 independent registrations share the launch set (SURVEY §8e "replicas only"). */
int icp_init (icp_handle h, uint32_t m, uint32_t nr, float a, float c,
              uint32_t max_iterations, double angle_threshold, double translation_threshold);
int icp_init_batched (icp_handle h, uint32_t batch, uint32_t m, uint32_t nr, float a, float c,
                      uint32_t max_iterations, double angle_threshold,
                      double translation_threshold);

/* ---- data movement ------------------------------------------------------------------------ */

/* ICPStep::write (mem, ptr, block, events, event) — include/ICP/algorithms.hpp:2273,
 * src/ICP/algorithms.cpp:4596-4622.  mem in {ICP_MEM_F, ICP_MEM_M, ICP_MEM_T}.  Host -> pinned
 * staging -> device on the handle's stream; block != 0 waits for completion. */
int icp_write (icp_handle h, int mem, const void *host_ptr, int block);
int icp_write_b (icp_handle h, uint32_t batch_index, int mem, const void *host_ptr, int block);

/* ICPStep::read (mem, block, events, event) — include/ICP/algorithms.hpp:2275,
 * src/ICP/algorithms.cpp:4634-4649; also the read() of the sub-objects (ICPMean :1757, ICPS :2496,
 * ICPPowerMethod :3123, ICPWeights :1198).  Copies `bytes` (<= object size) to host_dst; always blocking. */
int icp_read (icp_handle h, int mem, void *host_dst, size_t bytes);
int icp_read_b (icp_handle h, uint32_t batch_index, int mem, void *host_dst, size_t bytes);
size_t icp_mem_size (icp_handle h, int mem);

/* cl::Memory& ICPStep::get (Memory) — include/ICP/algorithms.hpp:2270,
 * src/ICP/algorithms.cpp:4366-4383: the device buffer itself, for zero-copy chaining.
 * The pointers icp_device_ptr returns are valid until the next icp_init* / icp_destroy on the handle (init frees and
 * re-creates every buffer the handle owns).
 * adopt: the caller's device buffer replaces the handle's (F/M only; call after init, again after every re-init; the
 * buffer stays the caller's — never freed by the handle — and must outlive its use by the handle). */
int icp_device_ptr (icp_handle h, int mem, void **dptr);
int icp_adopt_device_buffer (icp_handle h, int mem, void *dptr);

/* ---- the hot path ------------------------------------------------------------------------- */

/* ICPStep::buildRBC (events, event) — include/ICP/algorithms.hpp:2277,
 * src/ICP/algorithms.cpp:4655-4660 (getReps + RBCConstruct); ICP::buildRBC also resets k (:4796). */
int icp_build_rbc (icp_handle h);

/* ICPStep::run (events, event, config) — include/ICP/algorithms.hpp:2278,
 * src/ICP/algorithms.cpp:4670-4698: one iteration; on return T, Tk, R.. are updated on the
 * device (the reference blocks on a 32-byte read here; this call only enqueues). */
int icp_step (icp_handle h, int config);

/* ICP::run () — include/ICP/algorithms.hpp:2446, src/ICP/algorithms.cpp:4806-4834: iterate until
 * check() stops; blocking.  *k receives the iteration count (ICP::k) of registration 0 (a batch: icp_state_b gives every
 * registration's own k and converged flag; a registration that has converged is skipped by the launches the others still need).
 * The loop is the reference's host loop (:4806-4814) with the check on the device: every new transform's (k, converged) reaches
 * the host as one 8-byte store into pinned memory, the calling thread keeps `depth` launches queued behind the one in flight and
 * stops enqueueing when the flag shows — a run costs k launches plus at most `depth` that leave at their first load, not
 * max_iterations; the final state arrives in pinned memory with the end kernel (no stream synchronisation, no copy). */
int icp_run (icp_handle h, uint32_t *k);

/* The last finished checked run (icp_run, a tracked frame): iteration launches enqueued, its final k, and how many of the launches
 * ran past the registration's last live iteration.  Any pointer may be NULL. */
int icp_run_stats (icp_handle h, uint32_t *launches, uint32_t *k, uint32_t *dead_launches);

/* Diagnostic: the host's own kernel-launch calls inside checked runs since icp_init (or the last reset): the longest one, how many took
 * more than 10 us, how many there were.  A host-driven run is as good as the host is punctual. */
int icp_launch_stats (icp_handle h, double *max_us, uint64_t *slower_than_10us, uint64_t *total, int reset);

/* Per-query outputs of checked runs (icp_run, tracked frames: ICP_MEM_NN_ID, _W, _NN, _QT, _RID — the reference's D_OUT_NN_ID etc. of
 * the last executed iteration).  The fused kernels read none of them, and a checked run cannot know which iteration is its last:
 *   ICP_OUTPUTS_LAZY (default)     the run stores none; the first icp_read / icp_device_ptr of one re-runs the search of the last executed
 *                                  iteration with the transform it used (kept on the device): same bits, one extra launch, only when asked.
 *                                  If F, M or the RBC have changed since the run (icp_write, icp_build_rbc, the next tracked frame), the
 *                                  read fails with ICP_ESTATE instead;
 *   ICP_OUTPUTS_EVERY_ITERATION    every iteration stores them (0.4 us of every 9 at |F| = 16384); ICP_AMD_OUTPUTS=eager at icp_create.
 * Single steps, fixed-length runs and the reference-order mode always store them. */
typedef enum { ICP_OUTPUTS_LAZY = 0, ICP_OUTPUTS_EVERY_ITERATION = 1 } icp_output_mode;
int icp_set_output_mode (icp_handle h, int mode);

/* Diagnostic: host timeline of the last icp_run, microseconds after its begin — [0] 0, [1] the first launches enqueued, [2] the first
 * progress word seen, [3] decided (converged flag seen or max_iterations enqueued), [4] end kernel enqueued, [5] FINAL bit seen. */
int icp_run_timeline (icp_handle h, double *us6);

/* depth: launches kept queued behind the one in flight by checked runs (default 3; ICP_AMD_RUN_DEPTH at icp_create).
 * adaptive = 0 brings back rounds 1 - 3's form — one cached graph of max_iterations launches per checked run, converged iterations
 * leaving early — for comparisons (ICP_AMD_RUN_ADAPTIVE=0 at icp_create). */
int icp_set_run_depth (icp_handle h, uint32_t depth, int adaptive);

/* ICP::run (timer) — include/ICP/algorithms.hpp:2482-2494: exactly `iterations` steps, no
 * convergence test (the reference's profiling run; 40 there).  Enqueue only. */
int icp_run_fixed (icp_handle h, uint32_t iterations);

/* T <- identity, k <- 0: the state ICPStep::init uploads (src/ICP/algorithms.cpp:4486-4493). Enqueue only. */
int icp_reset_transform (icp_handle h);

/* icp_reset_transform + icp_run_fixed as ONE graph: a fresh registration of exactly `iterations` steps (the reference's
 * profiling run right after init, include/ICP/algorithms.hpp:2482-2494).  In the chained form the reset costs no launch
 * (the first search of the chain starts from the identity itself).  Enqueue only. */
int icp_run_fixed_fresh (icp_handle h, uint32_t iterations);

/* Blocks until everything enqueued on the handle's stream is done (queue.finish ()). */
int icp_sync (icp_handle h);

/* ---- parameters ---------------------------------------------------------------------------- */
/* getAlpha/setAlpha/getScaling/setScaling — include/ICP/algorithms.hpp:2279-2295;
 * get/setMaxIterations, AngleThreshold, TranslationThreshold — :2447-2460. */
int icp_get_alpha (icp_handle h, float *a);
int icp_set_alpha (icp_handle h, float a);
int icp_get_scaling (icp_handle h, float *c);
/* Absolute scale of the photogeometric metric.  The reference only states d = f_g(a) |x_g - x'_g|^2 + f_p(a) |x_p - x'_p|^2
 * (src/ICP/algorithms.cpp:4393-4398); euclideanSquaredMetric8 itself lives in the un-vendored RandomBallCover.  The engine
 * searches on geo + a pho (i.e. f_p / f_g = a: the correspondences depend only on that ratio) and reports
 * dist = f_g (geo + a pho), f_g = 1 by default.  f_g matters in WEIGHTED mode only, through w = 100 / (100 + dist)
 * (kernels/icp_kernels.cl:232): a normalised metric, e.g. f_g = 1 / (1 + a), f_p = a / (1 + a), is selected with
 * icp_set_alpha (h, a) + icp_set_metric_scale (h, 1 / (1 + a)).  Positive and finite. */
int icp_set_metric_scale (icp_handle h, float f_g);
int icp_get_metric_scale (icp_handle h, float *f_g);
int icp_set_scaling (icp_handle h, float c);
int icp_get_max_iterations (icp_handle h, uint32_t *n);
int icp_set_max_iterations (icp_handle h, uint32_t n);
int icp_get_angle_threshold (icp_handle h, double *deg);
int icp_set_angle_threshold (icp_handle h, double deg);
int icp_get_translation_threshold (icp_handle h, double *mm);
int icp_set_translation_threshold (icp_handle h, double mm);
int icp_set_power_mode (icp_handle h, int mode);      /* icp_power_mode */
int icp_set_reduce_mode (icp_handle h, int mode);     /* icp_reduce_mode */

/* Public state members of ICPStep/ICP (Rk qk tk sk R q t s k) — blocking. */
int icp_state (icp_handle h, icp_state_t *out);
int icp_state_b (icp_handle h, uint32_t batch_index, icp_state_t *out);

/* ---- adjacent steps -------------------------------------------------------------- */

/* ICPLMs: getLMs — kernels/icp_kernels.cl:63-76, src/ICP/algorithms.cpp:621-785.
 * cloud: 640x480 float8 on the host; which = ICP_MEM_F or ICP_MEM_M (m must be 16384). */
int icp_write_cloud (icp_handle h, int which, const void *host_cloud_640x480x8, int block);

/* ICPTransform<QUATERNION> on an arbitrary cloud with the handle's current T —
 * src/ocl_icp_reg.cpp:175 (full-cloud transform after run()).  Host in, host out; n points. */
int icp_transform_cloud (icp_handle h, const void *host_in, void *host_out, uint32_t n);

/* Frame-to-frame tracking — README.md:4 ("real-time frame-to-frame registration"); per pair the demo's flow
 * src/ocl_icp_reg.cpp:128-172 (init: getLMs of both clouds; registerPC: buildRBC + run).  Frames of a sequence are fed one by
 * one (640x480 float8 each); frame f is registered against frame f - 1, whose landmarks are already resident: they become the
 * fixed set by a rotation of three landmark buffers (no copy, no host trip).  Only the band of a frame that getLMs reads
 * (128 rows x 509 pixels = 2.08 MB of the 9.83 MB) is uploaded; the landmarks are extracted on the device (kernels/icp_kernels.cl:63-76).
 *
 * icp_track_submit   upload + getLMs on a copy stream, then buildRBC + ICP::run; up to four frames may be in flight, so frame f + 1 is
 *                    uploaded while frame f registers.  The registration is a host-driven checked run (icp_run) that gets as many
 *                    iterations up front as the last two registrations suggest it needs (the smaller k, + 1); later icp_track_* calls
 *                    top it up.  No launch is spent on iterations past the convergence of a frame beyond that prediction / the run depth.
 *                    Consecutive registrations alternate between two streams (icp_track_form: gated): this frame's RBC construction
 *                    (into its own set of RBC buffers) and its launches are enqueued at once, behind a one-wave gate kernel that holds
 *                    its stream until the previous registration has released the sequence word — the device goes from one frame to
 *                    the next without the host.  (Host-ordered form: the call first brings the previous frame's run to its end.)
 *                    No deadline for the caller, and (round 6) no waiting for the device either: in the gated form the call returns
 *                    once this frame's own launches are out, and a thread of the engine (the keeper, one per tracking handle,
 *                    started with the first gated frame) looks after the open runs while the application is outside the library:
 *                    it polls their progress words, keeps their queues topped up, enqueues their end kernels.  Every entry point
 *                    pauses the keeper on its way in and hands the runs back on its way out — the two never touch the handle at the
 *                    same time.  So what a frame's gate waits for never depends on a later call, and neither does the frame submitted
 *                    last: a caller that stays away for a second or an hour finds its results waiting.  An error the keeper runs into
 *                    (below) is reported by the next icp_track_* call.  ICP_AMD_TRACK_KEEPER=0: no thread — the call itself brings
 *                    the previous frame's registration to its decision before it returns (round 5's rule), later calls top up the rest.
 *                    (The gate's own bound — ~0.5 s, more for large max_iterations — is a guard against a device that has stopped:
 *                    the frames behind it are then skipped without a store, the next call returns ICP_EHIP, icp_track_reset recovers.)
 *                    warm_start != 0: the registration starts from the previous hop's transform (written back as by
 *                    icp_write (ICP_MEM_T): the rotation state is re-derived from it) instead of the identity; the first
 *                    registration of a sequence (after icp_init / icp_track_reset) has no previous hop and starts from the
 *                    identity.
 *                    `cloud` may be pageable host memory (the band is copied into pinned staging by the calling thread) or one
 *                    of the engine's two pinned frame buffers (icp_track_staging: the band goes by DMA straight from there —
 *                    the reference's mapped staging buffers hPtrInF / hPtrInM, src/ICP/algorithms.cpp:4438-4475;
 *                    icp_track_staging returns a buffer only after the band of the frame it last held has left it — whichever
 *                    of the two buffers a frame came from, in any order, mixed with pageable frames).
 * icp_track_collect  blocks until the oldest frame in flight is done: *registered = 0 for the first frame after icp_init /
 *                    icp_track_reset (nothing to register against; *k = 0, T8 = identity), else 1, *k = iterations executed and
 *                    T8 = [q | t, s] mapping that frame onto the previous one.  Any output pointer may be NULL.
 * icp_track_next     submit + collect (blocking): afterwards T (icp_read, icp_state) maps the new frame onto the previous one.
 * m must be 16384 (getLMs), batch 1, and the handle's own F / M buffers (not adopted ones). */
int icp_track_next (icp_handle h, const void *host_cloud_640x480x8, int warm_start, uint32_t *k, int *registered);
int icp_track_submit (icp_handle h, const void *host_cloud_640x480x8, int warm_start);
int icp_track_collect (icp_handle h, uint32_t *k, float *T8, int *registered);
int icp_track_staging (icp_handle h, uint32_t slot /* 0 | 1 */, void **pinned_host_frame);
/* The caller's own frame buffers as DMA sources: page-locks `bytes` (>= one frame) at `frames` (hipHostRegister, once: a capture loop
 * reuses its buffers); a frame submitted from inside a registered range is uploaded like one from the engine's pinned frame buffers —
 * the band by one 2-D DMA, no copy by the calling thread (60 us of a frame's host time).  A frame must stay as it is until it has been
 * collected (icp_track_collect) — the engine cannot tell when the caller refills its own memory.  icp_track_unregister_source (the
 * range's start) before the memory is freed; icp_init / icp_destroy unregister what is left. */
int icp_track_register_source (icp_handle h, void *frames, size_t bytes);
int icp_track_unregister_source (icp_handle h, void *frames);
int icp_track_reset (icp_handle h);
/* How tracked frames follow each other on the device: *gated = 1 — consecutive registrations alternate between two streams, each held by a
 * device-side gate (a one-wave kernel, bounded wait) until its predecessor has released the sequence word: a frame's RBC construction and
 * its predicted launches are enqueued while the previous frame is still running, and the host is not on the path between two frames;
 * 0 — one stream, a frame's work enqueued when the host has seen the previous one decided (ICP_AMD_TRACK_GATE=0, checked runs as graphs, or
 * a runtime that serves the two streams from one hardware queue: probed once per handle).  Same results either way. */
int icp_track_form (icp_handle h, int *gated);

/* ICPTransform<QUATERNION> / ICPTransform<MATRIX> with an explicit transformation — include/ICP/algorithms.hpp:1189-1211,
 * 1240, 1348; src/ICP/algorithms.cpp:2554-2753 (quaternion), :2760-2960 (matrix); kernels/icp_kernels.cl:772-802
 * (icpTransform_Quaternion), :842-879 (icpTransform_Quaternion_2: the same mapping through two 4x4 products),
 * :904-933 (icpTransform_Matrix).  T: 8 floats [q | t, s] for the quaternion kinds, 16 floats (row-major 4x4, the
 * scaling already in the rotation block) for MATRIX.  Host in, host out; n points of 8 floats; needs no icp_init. */
typedef enum { ICP_TRANSFORM_QUATERNION = 0, ICP_TRANSFORM_QUATERNION_2 = 1, ICP_TRANSFORM_MATRIX = 2 } icp_transform_kind;
int icp_transform_cloud_ex (icp_handle h, int kind, const float *T, const void *host_in, void *host_out, uint32_t n);

/* ICPPowerMethod — include/ICP/algorithms.hpp:1451-1537 (init / write (D_IN_S, D_IN_MEAN) / run / read (H_OUT_T_K)),
 * src/ICP/algorithms.cpp:2966-3150, kernel icpPowerMethod kernels/icp_kernels.cl:977-1054 — and, with rot = ICP_ROT_EIGEN, the
 * host JacobiSVD of ICPStep<EIGEN, *>::run (src/ICP/algorithms.cpp:3877-3902) as the engine evaluates it on the device.
 * S: the 11 floats of ICPS (S row-major, then the numerator and denominator of the scale); means: [mean_fixed, 0 | mean_moving, 0];
 * Tk: [qk | tk, sk].  power_mode: icp_power_mode (ignored for ICP_ROT_EIGEN).  Rk9 (row-major rotation) and iters (power-method
 * loop trips) may be NULL.  One wave of the same device code the iteration's finalize runs (no other math); host pointers in and
 * out, blocking, needs no icp_init.  The reference's known-answer test (tests/testsICP.cpp:988-1052) drives exactly this entry. */
int icp_power_method (int device, int rot, int power_mode, const float *S11, const float *means8, float *Tk8, float *Rk9, uint32_t *iters);

/* ---- the reference's per-kernel wrapper classes as stand-alone operations (host in, host out, blocking, no icp_init) ---------------
 * A user of the reference can call its kernel classes one by one (and its tests do, tests/testsICP.cpp:66-790); the iteration here
 * fuses these steps (icp_step), so the classes are served by small kernels of their own with the canonical reduction trees —
 * bit-identical to what the fused path computes for the same inputs in reference-order mode.
 *   icp_kernel_lms      ICPLMs      include/ICP/algorithms.hpp:312-383   getLMs: 640 x 480 float8 -> 128 x 128 float8
 *   icp_kernel_reps     ICPReps     :397-468                             getReps (grid side sqrt (m)): m float8 -> nr float8
 *   icp_kernel_weights  ICPWeights  :485-568                             {dist, id}[n] -> W[n], sum of weights (double)
 *   icp_kernel_mean     ICPMean<REGULAR | WEIGHTED>  :625-843            F[n] float8, M[n] float8 (, W[n], sum_w) -> [mean_F, 0 | mean_M, 0]
 *   icp_kernel_devs     ICPDevs     :867-940                             F, M, means -> DF[n] float4, DM[n] float4
 *   icp_kernel_s        ICPS<REGULAR | WEIGHTED>     :976-1183           DM, DF (, W), c -> S[11] (S row-major, sum w |f|^2, sum w |m|^2) */
int icp_kernel_lms (int device, const void *cloud_640x480x8, void *lms_16384x8);
int icp_kernel_reps (int device, const void *F, uint32_t m, uint32_t nr, void *R);
int icp_kernel_weights (int device, const void *nn_id, uint32_t n, float *W, double *sum_w);
int icp_kernel_mean (int device, int weighted, const void *F, const void *M, const float *W, double sum_w, uint32_t n, float *mean8);
int icp_kernel_devs (int device, const void *F, const void *M, const float *mean8, uint32_t n, float *DF, float *DM);
int icp_kernel_s (int device, int weighted, const float *DM, const float *DF, const float *W, uint32_t m, float c, float *S11);
const char *icp_kernel_last_error (void);

/* The same classes as RESIDENT objects — the reference's L2 classes own cl::Buffers, hand them out through get (Memory) and are wired
 * by sharing them (include/ICP/algorithms.hpp:312-1537; wiring src/ICP/algorithms.cpp:4499-4581): device buffers live with the object,
 * icp_ko_device_ptr = get (Memory), icp_ko_adopt = a buffer assigned through get () before init (:2214-2220: the object then does not
 * own it), icp_ko_run = kernels only on the device's null stream (one in-order queue for all kernel objects), icp_ko_write / _read = the
 * staged upload / blocking download of one Memory object.  A slot's buffer is created at its first use, so adoption costs no allocation.
 * Slots (Memory objects) per kind, inputs first:
 *   ICP_KO_LMS      0 cloud 640 x 480 float8          | 1 landmarks 16384 float8
 *   ICP_KO_REPS     0 F n float8 (aux = nr)           | 1 R nr float8
 *   ICP_KO_WEIGHTS  0 {dist, id}[n]                   | 1 W[n], 2 sum of weights (double)
 *   ICP_KO_MEAN(_WEIGHTED)  0 F, 1 M, 2 W[n], 3 sum of weights (double)  | 4 [mean_F, 0 | mean_M, 0]
 *   ICP_KO_DEVS     0 F, 1 M, 2 means (8 floats)      | 3 DF n float4, 4 DM n float4
 *   ICP_KO_S(_WEIGHTED)     0 DM, 1 DF, 2 W[n]        | 3 S[11]     (c: the scaling, icp_ko_set_scaling) */
typedef struct icp_ko *icp_ko_handle;
typedef enum { ICP_KO_LMS = 0, ICP_KO_REPS = 1, ICP_KO_WEIGHTS = 2, ICP_KO_MEAN = 3, ICP_KO_MEAN_WEIGHTED = 4, ICP_KO_DEVS = 5, ICP_KO_S = 6,
               ICP_KO_S_WEIGHTED = 7 } icp_ko_kind;
int icp_ko_create (icp_ko_handle *out, int device, int kind, uint32_t n, uint32_t aux, float c);
int icp_ko_destroy (icp_ko_handle k);
int icp_ko_adopt (icp_ko_handle k, int slot, void *device_ptr);
int icp_ko_device_ptr (icp_ko_handle k, int slot, void **device_ptr);
size_t icp_ko_slot_bytes (icp_ko_handle k, int slot);
int icp_ko_write (icp_ko_handle k, int slot, const void *host);
int icp_ko_read (icp_ko_handle k, int slot, void *host);
int icp_ko_run (icp_ko_handle k);
int icp_ko_set_scaling (icp_ko_handle k, float c);

/* ---- standalone Reduce / Scan classes of the reference ------------------------------ */

/* Reduce<MIN,float> / Reduce<MAX,uint> / Reduce<SUM,float> — include/ICP/algorithms.hpp:52-166,
 * kernels/reduce_kernels.cl:68, 149, 230, src/ICP/algorithms.cpp:131-322.  Row-wise over rows x cols (cols % 4 == 0);
 * host buffers in and out (rows results).  SUM reproduces reduce_sum_f's tree bit for bit. */
typedef enum { ICP_REDUCE_MIN_F = 0, ICP_REDUCE_MAX_UI = 1, ICP_REDUCE_SUM_F = 2 } icp_reduce_op;
int icp_reduce (int device, int op, const void *host_in, uint32_t cols, uint32_t rows, void *host_out);

/* Scan<INCLUSIVE|EXCLUSIVE,int> — include/ICP/algorithms.hpp:169-290, kernels/scan_kernels.cl:67, 188, 296,
 * src/ICP/algorithms.cpp:403-600.  Row-wise over rows x cols ints (cols % 4 == 0). */
int icp_scan (int device, int inclusive, const int32_t *host_in, uint32_t cols, uint32_t rows, int32_t *host_out);
const char *icp_reduce_scan_last_error (void);

/* The same as resident objects, the shape of the reference's classes (ctor / init / write / run / read / get,
 * include/ICP/algorithms.hpp:83-166, 200-290): the device buffers live as long as the object, run enqueues kernels only
 * (no allocation, no copy), device_ptr is `get (Memory::D_IN / D_OUT)` for zero-copy chaining, time brackets `reps` runs
 * with HIP events on the object's stream (the reference's run (timer): 44 us for a 1024 x 1024 sum, 151 us for a scan on its
 * R9 270X, tests/testsReduce.cpp:252, tests/testsScan.cpp:175). */
typedef enum { ICP_RS_MIN_F = 0, ICP_RS_MAX_UI = 1, ICP_RS_SUM_F = 2, ICP_RS_SCAN_INCLUSIVE = 3, ICP_RS_SCAN_EXCLUSIVE = 4 } icp_rs_kind;
typedef struct icp_rs_context *icp_rs_handle;
int icp_rs_create (icp_rs_handle *r, int device, int kind, uint32_t cols, uint32_t rows);
int icp_rs_write (icp_rs_handle r, const void *host_in);            /* cols x rows elements (4 bytes each) */
int icp_rs_run (icp_rs_handle r);                                   /* enqueue only */
int icp_rs_read (icp_rs_handle r, void *host_out);                  /* rows results (reduce) / cols x rows (scan); blocking */
int icp_rs_device_ptr (icp_rs_handle r, int output, void **dptr);   /* 0: input buffer, 1: result of the last run */
int icp_rs_time (icp_rs_handle r, uint32_t reps, float *us_per_run);
int icp_rs_destroy (icp_rs_handle r);

/* ---- batches across devices (SURVEY.md §8b "icp_batch_*", §8e "replicas only") ---------------------------------------------
 * B independent registrations over a device list, inside the library: registration i lives on slot i mod n (slot s =
 * devices[s]; an ordinal may appear more than once) as batch entry i / n of that slot's engine handle, so every slot
 * serves its registrations with one launch set (icp_init_batched).  One host thread + one HIP stream per slot, pinned
 * staging per handle, no collective / peer access / RCCL.  The reference has no counterpart: one context, one in-order
 * queue (src/ICP/algorithms.cpp:4351-4352); per registration the calls mean what ICP<CR,CW>::init / write / buildRBC / run
 * mean (include/ICP/algorithms.hpp:2437-2462).  All calls block until every slot is done. */
typedef struct icp_batch_context *icp_batch_handle;
int icp_batch_create (icp_batch_handle *b, const int *devices, int n_devices, int rot, int weighted);
int icp_batch_destroy (icp_batch_handle b);
int icp_batch_init (icp_batch_handle b, uint32_t registrations, uint32_t m, uint32_t nr, float a, float c,
                    uint32_t max_iterations, double angle_threshold, double translation_threshold);
int icp_batch_set_modes (icp_batch_handle b, int reduce_mode, int power_mode);
int icp_batch_write (icp_batch_handle b, uint32_t i, int mem, const void *host_ptr);       /* mem: F, M or T of registration i */
int icp_batch_build_rbc (icp_batch_handle b);
int icp_batch_run (icp_batch_handle b);                                                      /* ICP::run of every registration */
int icp_batch_run_fixed (icp_batch_handle b, uint32_t iterations, int from_identity);
int icp_batch_state (icp_batch_handle b, uint32_t i, icp_state_t *out);
int icp_batch_read (icp_batch_handle b, uint32_t i, int mem, void *host_dst, size_t bytes);
int icp_batch_size (icp_batch_handle b, uint32_t *registrations, uint32_t *n_slots);
/* wall-clock seconds of `reps` fixed-length passes (from the identity) on all slots at once = max over devices */
int icp_batch_time_run_fixed (icp_batch_handle b, uint32_t iterations, uint32_t reps, double *seconds);
/* the same with `warmup` untimed passes per slot first, a gate in front of the timed region (every slot has drained its stream
 * before the clock starts), and the HIP-event time of every slot's own passes in slot_ms[n_slots] (may be NULL; 0 for a slot
 * without registrations): what bench.py --gpus N prints per GPU */
int icp_batch_time_run_fixed_slots (icp_batch_handle b, uint32_t iterations, uint32_t reps, uint32_t warmup, double *seconds, float *slot_ms);
/* the partition rule as a pure function (no device needed): slot, index inside the slot, registrations of that slot */
int icp_batch_partition (uint32_t registrations, uint32_t n_slots, uint32_t i, uint32_t *slot, uint32_t *index, uint32_t *slot_count);
/* The CPUs the host thread of `slot` was pinned to, comma-separated ("" = not pinned).             */
int icp_batch_slot_cpus (icp_batch_handle b, uint32_t slot, char *out, size_t cap);
const char *icp_batch_last_error (icp_batch_handle b);   /* b may be NULL: error of the last failed create */

/* ---- measurement (bench.py, HIP events on the handle's stream) --------------------------------- */

/* Times `reps` back-to-back icp_run_fixed(iterations) passes with hipEvents recorded on the
 * handle's own stream; *ms_total = elapsed ms over all reps.  from_identity != 0: every pass starts from
 * the identity transform (a fresh registration, like the reference's 40-step profiling run). */
int icp_time_run_fixed (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_total);
/* The same `reps` passes, the events around all but the first: *ms_timed = elapsed ms over *reps_timed = reps - 1 passes (1 of 1).  A marker
 * recorded on an idle stream holds back the graph launched behind it by 0.1 - 0.25 ms; behind a pass in flight it costs nothing.
 * What bench.py brackets its K wall-clock-timed steps with (the wall clock covers all K, the events K - 1 of them). */
int icp_time_run_fixed_tail (icp_handle h, uint32_t iterations, uint32_t reps, int from_identity, float *ms_timed, uint32_t *reps_timed);
/* ICP::run (timer) — include/ICP/algorithms.hpp:2482-2494: the reference's profiling run, exactly `iterations` steps
 * (40 there) from the current state, no convergence test, with a per-step, per-stage table (the reference fills a
 * ProfilingInfo<40> per kernel class through the run (timer) overloads, e.g. :2359-2399).  The stages run as separate
 * launches with HIP events around each: out_ms[it * 4 + s], s = icp_stage; fused reductions have no means / Sij
 * stage (those entries read ~0: two events back to back).  *total_ms (may be NULL) = first event to last.  Blocking.
 * (The graphs behind icp_run / icp_run_fixed fuse stages further — icp_launches_per_iteration — and are timed whole by
 * icp_time_run_fixed.) */
typedef enum { ICP_STAGE_SEARCH = 0, ICP_STAGE_MEANS = 1, ICP_STAGE_SIJ = 2, ICP_STAGE_FINALIZE = 3, ICP_STAGE_COUNT_ = 4 } icp_stage;
int icp_profile_run (icp_handle h, uint32_t iterations, float *out_ms, float *total_ms);
/* Means of the same over `reps` iterations: out_ms[0..3] = mean ms of {search, means, sij, finalize}. */
int icp_time_kernels (icp_handle h, uint32_t reps, float *out_ms4);

/* Kernel launches per iteration of the graphs behind icp_run / icp_run_fixed with the current modes and sizes:
 * 4 (reference-order reductions), 2 (fused: search + finalize; 3 beyond |F| = 16384, where the first level of the
 * moment tree is a launch of its own) or 1 (fused, latency-bound sizes: chained, see icp_run_form). */
int icp_launches_per_iteration (icp_handle h, uint32_t *n);

/* How icp_run / icp_run_fixed execute with the current modes and sizes:
 *   SEPARATE    one launch per stage (2 fused, 4 reference order);
 *   CHAINED     fused, one launch per iteration: the finalize of iteration k runs in the prologue of the search of
 *               iteration k+1 (latency-bound sizes; ICP_AMD_CHAIN=0 / 1 at icp_create forces it off / on);
 *   (a third form — one persistent launch per run, the per-iteration moment exchange between the blocks in-launch — was built and
 *   measured in round 2: 10.7 against 9.8 us per iteration at |F| = 16384, the all-to-all seam costs more than the launch boundary
 *   it replaces; retired in round 3)
 * Same bits in both forms. */
typedef enum { ICP_FORM_SEPARATE = 0, ICP_FORM_CHAINED = 1 } icp_run_form_t;
int icp_run_form (icp_handle h, int *form);
/* Diagnostic: how the search kernel behind the current modes and sizes is laid out.
 *   *dense   0: the latency variant (one 1024-thread block per CU, 16 lanes per query: a single small registration);
 *            1: the dense variant (512-thread blocks, several per CU, 8 lanes per query, exact stage-1 pruning);
 *   *tile    representatives per LDS tile (256 or 1024);
 *   *stage2  0: the lanes of a query scan its representative's list; 1: lanes = candidates — a wave loads a list once for
 *            all of its queries that share it (dense variant, lists of >= 128 candidates on average; ICP_AMD_S2WAVE=0 / 1 at
 *            icp_init forces it off / on).
 * Same bits in every layout.  Any output pointer may be NULL. */
int icp_search_layout (icp_handle h, int *dense, int *tile, int *stage2);

/* Diagnostic: a graph of `iterations` x (the kernels selected by mask: bit 0 search, 1 means, 2 sij,
 * 3 finalize, 4 an empty 256-block kernel), launched `reps` times; *ms_total = elapsed ms. */
int icp_time_masked (icp_handle h, uint32_t mask, uint32_t iterations, uint32_t reps, float *ms_total);

/* ---- utilities ---------------------------------------------------------------------------------- */

const char *icp_last_error (icp_handle h);      /* h may be NULL: error of the last failed create */
const char *icp_version (void);
int icp_device_count (int *n);
/* PCI bus id of device `device` ("0000:c1:00.0"; cap >= 16).                                        */
int icp_device_pci_bus_id (int device, char *out, size_t cap);
/* The cpulist ("0-47,96-143") of the NUMA node a PCI device hangs on, from a sysfs tree (sysfs_root NULL = "/sys"):
 * <root>/bus/pci/devices/<id>/local_cpulist, else node<numa_node>/cpulist; out = "" when the tree has no answer.
 * icp_batch_create pins the host thread of every device slot there unless ICP_AMD_SLOT_CPUS says otherwise
 * (ICP_AMD_SLOT_NUMA=0: no default placement).  No reference counterpart: one device, one queue
 * (src/ICP/algorithms.cpp:4351-4352).                                                                */
int icp_numa_cpulist (const char *sysfs_root, const char *pci_bus_id, char *out, size_t cap);

/* Synthetic RGB-D landmark pair: side x side grid, fixed and moving frame.
 * Host only; deterministic in (seed, side). */
int icp_synth_pair (uint64_t seed, uint32_t side, float rot_deg, const float *axis3,
                    const float *t3, float noise_mm, float noise_rgb, float zero_fraction,
                    float *F, float *M);
/* The same with a choice of scene.  scene 0: the curved scene of icp_synth_pair.  scene 1: a WALL — the reference's second example pair,
 * data/kg_pc8d_wall ("non-salient surface geometry ... highlights the benefit of utilizing the photometric information",
 * data/README.md:11-16): a tilted plane with a millimetre of surface roughness and the procedural texture, moved IN its own plane (a
 * rotation by rot_deg about the plane's normal through its centre — axis3 is ignored — and the in-plane part of t3): geometry alone
 * cannot see that motion.  T_true8 (may be NULL): the ground truth [q | t, 1] mapping the moving frame onto the fixed one. */
int icp_synth_pair_scene (uint64_t seed, uint32_t side, int scene, float rot_deg, const float *axis3, const float *t3,
                          float noise_mm, float noise_rgb, float *F, float *M, float *T_true8);
/* Synthetic 640x480 float8 cloud for the getLMs path; `moved` = frame number of a sequence (0: the scene, f: moved
 * rigidly by f steps of 3 degrees / (25, -10, 15) mm, with noise). */
int icp_synth_cloud_vga (uint64_t seed, int moved, float *cloud);
/* Invalid pixels as a Kinect frame holds them — depth 0: x = y = z = 0, the colour written regardless (reference
 * src/kinect_frame_grabber.cpp:246-262; getLMs picks such points on purpose, kernels/icp_kernels.cl:49-50) — punched in place into a
 * width x height grid of float8 points (a landmark set: side x side; a cloud: 640 x 480).  pattern 0: scattered, every point with
 * probability `fraction`; 1: contiguous — a band along the left edge and random ellipses until `fraction` of the points is covered.
 * keep_rgb 0 zeroes the colour too: all invalid points identical, one representative's list holds them all (the degenerate case). */
int icp_synth_punch_holes (uint64_t seed, uint32_t width, uint32_t height, int pattern, float fraction, int keep_rgb, float *cloud);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ICP_AMD_H */
