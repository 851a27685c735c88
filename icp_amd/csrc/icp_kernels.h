// icp_kernels.h — launch parameters shared by the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "icp_device.h"

// fused finalize: beyond this many 128-block groups (|F| > 16384) the first level of the moment tree gets a kernel of its own
// (k_moment_level1: the block moments come from all over the chip; one CU fetching 147 KB of them at |F| = 65536 costs more than
// a launch that spreads the fetch — finalize 6.29 -> 5.71 us for the two launches, iteration 19.4 -> 18.55 us; same tree, same bits)
#ifndef ICP_L1_MIN_GROUPS
#define ICP_L1_MIN_GROUPS 2u
#endif
// The band of a 640 x 480 frame getLMs reads (kernels/icp_kernels.cl:63-76: landmark (i, j) = pixel (col 65 + 4 i, row 49 + 3 j)):
// rows 49, 52, .., 430 and, of each, the pixels 65 .. 573.
#define ICP_BAND_ROW0 49u
#define ICP_BAND_ROW_STEP 3u
#define ICP_BAND_ROWS 128u
#define ICP_BAND_COL0 65u
#define ICP_BAND_COLS 509u
#define ICP_BAND_ROW_BYTES (ICP_BAND_COLS * 32u)
#define ICP_BAND_BYTES ((size_t) ICP_BAND_ROWS * ICP_BAND_ROW_BYTES)
#define ICP_TBOX 1024u            // representatives per LDS tile box of the 1024-tile dense search (k_reps_and_boxes, k_search)
#ifndef ICP_OL_BOXED_MIN
#define ICP_OL_BOXED_MIN 128u        // the list of the representatives at the origin: beyond this length ordered by colour, its chunks of 8 tested by their boxes before they are scanned
#endif
#define ICP_OL_MASKS(nr) ((nr) + 1u + 2u * (((nr) + 7u) / 8u))                            // float4 offset of the ballots inside a registration's OL
#define ICP_OL_STRIDE(nr) (ICP_OL_MASKS (nr) + ((nr) + 127u) / 128u)                          // float4 per registration of icp_params::OL
#define ICP_CHUNK 1024u          // fixed points per block in the stable RBC placement

struct icp_params {
    // problem
    uint32_t m, nr, batch;
    uint32_t side, nrx, nry;     // landmark grid side and representative grid (getReps)
    uint32_t ng_magic;                               // floor (2^32 / ng) + 1 for ng = groups of 128 block moments: task / ng = umulhi (task, magic)
    uint32_t side_magic, cellw_magic, cellh_magic;   // floor (2^32 / d) + 1 for d = side, side / nrx, side / nry (0: the set is no square grid): n / d = umulhi (n, magic)
    float a, c;
    float dist_scale;            // f_g of the metric text: reported distance (NN_ID.dist, the weights' input) = dist_scale * (geo + a pho); default 1
    int weighted, rot, power_mode, check;
    int chain;                   // fused mode, one launch per iteration (finalize in the next search's prologue): 0 never, 1 automatic, 2 always
    int fused;                   // 0: reference-order reductions (3 global trees), 1: single-pass double moments
    int emit;                    // fused mode: this search also stores the per-query outputs (nn_id, PF, PM; rid unless pruning needs it anyway); graphs of a
                                 // fixed length switch it off for all but their last iteration (nothing in between can read them)
    double tan_half_thr, trans_thr;
    // derived sizes
    uint32_t nwg;                // 128-element groups = ceil(m/128)         (weights / means partials)
    uint32_t nwp;                // weight partials padded (multiple of 4 unless 1)
    uint32_t G;                  // S columns = ceil4(m)/4                    (icpSijProducts work-items)
    uint32_t nsp;                // reduce_sum_f work-groups per S row, padded (multiple of 4 unless 1)
    uint32_t nchunk;             // ceil(m / ICP_CHUNK)
    uint32_t nb;                 // fused mode: blocks of 64 pairs = ceil(m/64)
    // inputs / RBC
    const float *F, *M;          // [batch][m][8]
    float *R;                    // [batch][nr][8]
    float4 *GB;                  // [batch][2*(n16+n32)] geometry bounding boxes (lo, hi) of the groups of 16 representatives, then of the tiles of 1024
    uint32_t n16, n1k;           // ceil(nr/16), ceil(nr/tbox)
    uint32_t tbox;               // representatives per tile box (k_tile_boxes): the LDS tile of the dense k_search for multi-tile sets, 256 or 1024
    uint32_t gtile;              // stage-1 pruning groups of 16: 0 = 16 consecutive representatives, 1 + log2 (nrx / 4) = 4 x 4 tiles of the representative grid
    uint32_t xcdmap;             // dense search: XCD-aware block -> tile bands (a single large registration: halves the fabric-side traffic)
    uint32_t warm_seed;          // diagnostics (ICP_AMD_WARM_SEED=1): a registration's first search is seeded with p.rid as it stands (the previous
                                 // registration's answer) instead of the query's own grid cell
    uint32_t s2wave;             // stage 2 of the dense search with lanes = candidates (lists of >= ICP_S2_WAVE_MIN candidates on average: see k_search)
    float *XP;                   // [batch][m][8]  permuted database (RBCConstruct D_OUT_X_P)
    float *XQ;                   // [batch][m][8]  same, lane 3 = original index bits (search copy)
    float4 *OL;                  // [batch][ICP_OL_STRIDE (nr)]  the representatives at the origin (invalid points), ascending: [0].x = their number (bits),
                                 // [1 .. nr]: (r, g, b, index bits) each — kept out of the pruning boxes, scanned by the queries near the origin (dense
                                 // search) —, behind them the colour boxes (lo rgb, hi rgb) of the chunks of 8 consecutive entries, then one 64-bit ballot per 64
                                 // representatives ([0].y: the arrival counter of k_reps_and_boxes' blocks, zero between constructions; a search looks the seed of a query
                                 // whose own seed is of the wrong kind up in the ballots: icp_other_kind_near)
    float4 *LB;                  // [batch][3 * nlb]  6-D bounding boxes of the list chunks (16 consecutive positions of one list, chunk c >= 1 of list r at
                                 // index (O[r] >> 4) + c: k_list_boxes) as [lo.x lo.y lo.z lo.r | lo.g lo.b hi.x hi.y | hi.z hi.r hi.g hi.b]
    uint32_t nlb;                // m / 16 + 2 boxes per registration
    uint32_t *rep_src, *owner, *N, *O, *perm, *chunk_hist;   // [batch][...]; N: [2][batch][nr] — [0] the length of a list AS THE SEARCH SCANS IT (k_search reads
                                 // p.N: a long list without the tail members that repeat an earlier member bit for bit, k_list_boxes), [1] = ICP_N_FULL: the list's
                                 // length N of the construction (RBCConstruct's output, ICP_MEM_RBC_N); the two differ for long lists with duplicates only
    uint2 *blist; uint32_t *bn; uint8_t *brank;   // buildRBC of the latency-bound sizes (k_place_lists): per block of 64 fixed points its (owner, count) list
                                                  // [batch][nb][64] and the list's length [batch][nb]; rank of a point inside its block [batch][m]
    // per-iteration
    uint32_t *rid;               // [batch][m]
    icp_dist_id *nn_id;          // [batch][m]
    float4 *PF, *PM;             // [batch][m]  (nn.xyz, w) / (q'.xyz, dist)
    float *wpart;                // [batch][2*nwp]   even / odd half-trees of every 128-query group
    float4 *mpart;               // [batch][2][nwg]
    float4 *mscr;                // [batch][2][ceil(nwg/128)]  scratch of the multi-level icpGMean
    float *spart;                // [batch][11][nsp*8] 8 residue sub-trees per work-group
    double *mom;                 // [batch][2][18][nb]  fused mode: per-block moment partials (double-buffered for the chain)
    double *ml1;                 // [batch][18][ceil(nb/128)]  fused mode, large sets: first tree level of the moments (k_moment_level1)
    icp_reg_state *cst;          // [batch][2]  chained fused mode: state slots, launch j reads slot j&1 and writes the other
    uint32_t slot;               // chained fused mode: slot this launch reads
    icp_reg_state *st;           // [batch]
    icp_reg_state *st_prev;      // [batch]  .T = the transform the last executed search used (stored by every finalize; everything else stays zero):
                                 // icp_launch_search on it reproduces that iteration's per-query outputs (checked runs do not store them on the way)
    unsigned long long *dbg;     // diagnostic builds only (ICP_DBG_STAMPS): [blocks][16] s_memtime stamps
    // checked runs driven from the host (icp_run, icp_track_*: see run_ctl in icp_capi.hip): progress words and the final states in
    // host memory the device writes straight into (fine-grained pinned allocations; nullptr: nobody is watching)
    unsigned long long *hmirror; // [batch]  ICP_MIRROR_WORD (epoch, done, k) stored by the lane that publishes a registration's state
    icp_reg_state *hstate;       // [batch]  the final state of a run, stored by its end kernel in front of the word's FINAL bit
    uint32_t epoch;              // tag of the run the words belong to (a word of another epoch is stale)
    // tracking with frames gated on the device (icp_capi.hip: track_submit): consecutive frames alternate between two streams, so the
    // launches of a frame that has converged and the next frame's run concurrently
    uint32_t *run_flag;          // [batch]  epoch of the run that has converged: its remaining launches leave at once and WRITE NOTHING (one flag per
                                 // stream: it has to outlive the run's last queued launch, and the next frame on the other stream may converge before that)
    uint32_t *track_seq;         // number of the last registration of a tracked sequence that has finished (k_gate waits on it)
    uint32_t seq_value;          // this registration's number
    uint32_t no_state_reset;     // buildRBC leaves k / done alone (it runs ahead of the previous frame's end; the run's first launch resets them)
};

#define ICP_N_FULL(p, b) ((p).N + ((size_t) (p).batch + (b)) * (p).nr)

// progress word of a checked run: bits 0..23 k (iterations whose transform has been published), bit 30 FINAL (the end kernel has
// left the state in p.hstate), bit 31 done (ICP::check said stop), bits 32..63 the run's epoch
#define ICP_MIRROR_FINAL (1ull << 30)
#define ICP_MIRROR_DONE (1ull << 31)
#define ICP_MIRROR_WORD(epoch, k, done) (((unsigned long long) (epoch) << 32) | ((done) ? ICP_MIRROR_DONE : 0ull) | (unsigned long long) ((k) & 0xFFFFFFu))

// Index of the fixed point representative r is sampled from (generalised getReps: kernels/icp_kernels.cl:107-113 with the
// grid side of the set instead of 128).
static __device__ __forceinline__ uint32_t rep_src_index (const icp_params &p, uint32_t r)
{
    uint32_t gX = r % p.nrx, gY = r / p.nrx;
    uint32_t stepX = p.side / p.nrx, stepY = p.side / p.nry;
    uint32_t xi = (stepX == 1) ? gX : gX * stepX + (stepX >> 1) - 1;
    uint32_t yi = (stepY == 1) ? gY : gY * stepY + (stepY >> 1) - 1;
    return yi * p.side + xi;
}

// The same mapping for a thread that walks many representatives (the boxes of k_reps_and_boxes): the two step divisions once, and shifts
// for r / nrx, r % nrx — the grid's width is a power of two (icp_init accepts no other: icp_capi.hip, "nr must be a power of two").
struct rep_src_map {
    uint32_t nrx, side, stepX, stepY, offX, offY, lgx;
    __device__ __forceinline__ explicit rep_src_map (const icp_params &p)
        : nrx (p.nrx), side (p.side), stepX (p.side / p.nrx), stepY (p.side / p.nry), lgx (31u - (uint32_t) __builtin_clz (p.nrx | 1u))
    {
        offX = (stepX == 1u) ? 0u : (stepX >> 1) - 1u; offY = (stepY == 1u) ? 0u : (stepY >> 1) - 1u;
    }
    __device__ __forceinline__ uint32_t operator() (uint32_t r) const
    {
        return ((r >> lgx) * stepY + offY) * side + (r & (nrx - 1u)) * stepX + offX;
    }
};


// The representative nearest by index to r that is of the OTHER kind — not at the origin when r is, at the origin when r is not —, looked up in
// the ballots k_reps_and_boxes leaves (bit l of word w: representative 64 w + l is an invalid point of its frame: at the origin); `fallback`
// where r's word and four words to either side hold none.  (ks_seed_against_invalid: a handful of lanes, a registration's first search.)
static __device__ __forceinline__ uint32_t icp_other_kind_near (const unsigned long long *MK, uint32_t nr, uint32_t r, bool r_at_origin, uint32_t fallback)
{
    const uint32_t nw = (nr + 63u) / 64u, w = r >> 6, l = r & 63u;
    auto other = [&] (uint32_t ww) -> unsigned long long {
        const uint32_t left = nr - 64u * ww;
        const unsigned long long m = MK[ww];
        return (r_at_origin ? ~m : m) & (left < 64u ? (1ull << left) - 1ull : ~0ull);
    };
    const unsigned long long o = other (w), below = o & ((1ull << l) - 1ull), above = l < 63u ? o & ~((2ull << l) - 1ull) : 0ull;
    uint32_t best_d = 0xFFFFFFFFu, best = fallback;
    if (below) { const uint32_t q = 63u - (uint32_t) __builtin_clzll (below); best_d = l - q; best = 64u * w + q; }
    if (above) { const uint32_t q = (uint32_t) __builtin_ctzll (above); if (q - l < best_d) { best_d = q - l; best = 64u * w + q; } }
    if (best_d != 0xFFFFFFFFu) return best;
    for (uint32_t d = 1; d <= 4u; ++d) {
        if (w >= d) { const unsigned long long ol = other (w - d); if (ol) return 64u * (w - d) + 63u - (uint32_t) __builtin_clzll (ol); }
        if (w + d < nw) { const unsigned long long orr = other (w + d); if (orr) return 64u * (w + d) + (uint32_t) __builtin_ctzll (orr); }
    }
    return fallback;
}

// launchers (icp_kernels.hip, icp_build.hip)
void icp_launch_build_rbc (const icp_params &p, hipStream_t s);
void icp_launch_search (const icp_params &p, hipStream_t s);
void icp_launch_means (const icp_params &p, hipStream_t s);
void icp_launch_sij (const icp_params &p, hipStream_t s);
void icp_launch_finalize (const icp_params &p, hipStream_t s);
void icp_launch_iteration (const icp_params &p, hipStream_t s);
void icp_launch_masked (const icp_params &p, hipStream_t s, unsigned mask);
void icp_launch_chain (const icp_params &p, hipStream_t s, uint32_t iterations, bool fresh = false);
void icp_launch_chain_one (const icp_params &p, hipStream_t s, uint32_t j, bool fresh, bool emit);   // launch j of a chain (j = 0: reads the user-visible state)
void icp_launch_chain_end (const icp_params &p, hipStream_t s, uint32_t launches);                  // after `launches` chained launches: the last moments -> p.st (and p.hstate)
void icp_launch_publish_state (const icp_params &p, hipStream_t s);                                  // separate launches: p.st -> p.hstate + the FINAL bit
bool icp_chain_supported (const icp_params &p);
bool icp_build_lists (const icp_params &p);      // buildRBC = owner search + k_place_lists (2 launches)
bool icp_dense (const icp_params &p);            // the dense search variant (several blocks per CU, stage-1 pruning)
uint32_t icp_dense_tile (const icp_params &p);   // its LDS tile: 256 or 1024 representatives
void icp_launch_owner_search (const icp_params &p, hipStream_t s);   // RBC construct, step 1 (k_search<.., OWNER>)
void icp_launch_search_dense (const icp_params &p, hipStream_t s);          // icp_search_dense.hip: the dense variants (icp_dense (p))
void icp_launch_owner_search_dense (const icp_params &p, hipStream_t s);
uint32_t icp_tbox_of (const icp_params &p);
uint32_t icp_s2_wave_of (const icp_params &p);
void icp_search_layout_of (const icp_params &p, int *dense, int *tile, int *stage2);   // what icp_launch_search selects          // 1: the dense search scans the lists with lanes = candidates (long lists)
void icp_launch_reset_state (const icp_params &p, hipStream_t s, int reset_T);
void icp_launch_set_T (const icp_params &p, uint32_t b, const float *dT8, hipStream_t s, int reset_k = 0);
// holds the stream until *seq >= want; bounded: max_spins rounds of ~0.25 us, then *host_timeout_flag = 1 and the stream goes on
// (on a timeout also *run_flag = epoch: the launches of the run behind the gate leave without a store)
void icp_launch_gate (const uint32_t *seq, uint32_t want, uint32_t *host_timeout_flag, hipStream_t s, uint32_t max_spins = 1u << 21, icp_reg_state *warm = nullptr,
                      uint32_t *run_flag = nullptr, uint32_t epoch = 0u);
void icp_launch_seq_set (uint32_t *seq, uint32_t v, hipStream_t s);
void icp_launch_rotation_solver (int rot, int power_mode, const float *din19, float *dout18, hipStream_t s);   // [S 11 | means 8] -> [Tk 8 | Rk 9 | trips]
void icp_launch_get_lms (const float *cloud, float *lms, hipStream_t s);
void icp_launch_get_lms_band (const float *band, float *lms, hipStream_t s);
void icp_launch_transform_cloud (const float *in, float *out, const icp_reg_state *st, uint32_t n, hipStream_t s);
// kind 0 / 1: T = [q | t, s] (8 floats), 2: T = row-major 4x4 (16 floats); host pointer, passed by value
void icp_launch_transform_cloud_ex (int kind, const float *in, float *out, const float *T, uint32_t n, hipStream_t s);
