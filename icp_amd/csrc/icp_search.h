// icp_search.h — the search kernel template and the device helpers it shares with the finalize kernels (trees over the block partials,
// the finalize of a block: moments -> T).  Included by the two translation units that instantiate k_search:
//   icp_kernels.hip        the latency variants (one 1024-thread block per CU: a single small registration; chained form; the owner
//                          search with per-block lists) + every other per-iteration kernel
//   icp_search_dense.hip   the dense variants (512-thread blocks, several per CU, exact stage-1 pruning, 256- / 1024-representative tiles,
//                          both list-scan forms) + the dense owner search
// so that the two families compile side by side and a change to one family's launch code does not rebuild the other.
//
// Map of this file (round 5: k_search split into its phases; what a phase needs from the others is in its parameter list):
//   helpers                      sums, group minima (DPP), KS_CAND / KS_CAND_IF (one list candidate against the lane's query), ks_tile_of_block,
//                                fused_query_index, cell_rep_of
//   ks_seed_against_invalid, ks_origin_list, ks_origin_section    a frame's invalid points (representatives at the origin): the seed of a
//                                query whose cell's representative is of the other kind; their list, scanned behind the tiles
//   fused_moment_* / fused_finalize_block / fin_result_to_state      the finalize of a block: 18 double moments -> T (shared with icp_kernels.hip)
//   ks_stage2_wave               stage 2, lanes = candidates (dense variant, long lists)
//   ks_stage2_lanes<LPQ>         stage 2, a query's lanes scan its list; exact chunk-box pruning beyond the first 128 positions of long lists
//   ks_epilogue<...>             hand-off to the finishing wave, winner record, weights, per-query outputs, block moments / weight tree
//   ks_owner_lists_tail          RBC construct of the latency-bound sizes: what the owner search leaves for k_place_lists
//   ks_coarse_pass / ks_fine_pass  stage 1 over one LDS tile of representatives: group-box tests, then the groups some query of the wave needs
//                                (free functions with plain scalar parameters: with the query passed as a struct the 64-register variants spilled)
//   k_search<...>                the kernel: PROLOGUE (every independent load issued before the first wait; chained form: the previous
//                                iteration's finalize), the STAGE 1 drivers (MASKED: tile set decided once; else a tile loop; the origin list),
//                                then the calls above
//   host side                    icp_tpr_magic, KS_FLAGS, KS_ARGS
#pragma once
#include "icp_kernels.h"

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ float sum4 (float4 v) { return ((v.x + v.y) + v.z) + v.w; }

// ------------------------------------------------------------------------------------------
// K1  search: transform (a3) + RBC one-shot search (a4) + weights and their first tree levels (a5)
//
//   block  = 64 queries x LPQ lanes (LPQ waves); reference order: the 64 even (or odd) positions of one 128-query
//            group, so that the block owns a closed sub-tree of the weight reduction; fused: an 8 x 8 tile of the grid;
//   wave   = 64 / LPQ queries end to end: the LPQ lanes of a query split the representatives (stage 1) and the list
//            positions (stage 2); representatives are staged through LDS in tiles (broadcast ds_read_b128).
// ------------------------------------------------------------------------------------------
// KS_SPLIT = lanes per query = waves per block (8 or 16); inside k_search it names the template parameter LPQ
#define KS_SPLIT LPQ
#define KS_QPW (64 / KS_SPLIT)  // queries per wave
#define KS_TILE 1024u            // representatives per LDS tile
#ifndef ICP_S2_DEPTH16
#define ICP_S2_DEPTH16 8u            // stage 2, 16 lanes per query: candidates in flight per lane (8 x 16 = 128 covers every list at |R| = m/64)
#endif
#ifndef ICP_S2_UNCOND
#define ICP_S2_UNCOND 128u           // stage 2 (a query's lanes scan its list): positions of a list scanned unconditionally; beyond them chunk boxes first
#endif
#ifndef ICP_S1_ORIGIN_LIST
#define ICP_S1_ORIGIN_LIST 1         // dense variants: the representatives at the origin are scanned as a list of their own (0: A/B builds without it — wrong results on frames with invalid points)
#endif
#ifndef ICP_OL_STAGED_MIN
#define ICP_OL_STAGED_MIN 32u         // the list of the representatives at the origin: up to this length (one trip of a query's lanes) read per wave, not staged per block
#endif
#ifndef ICP_S1_SEED
#define ICP_S1_SEED 1                // stage 1: prune with the distance to the previous search's nearest representative
#endif
#ifndef ICP_S1_REJECT_MIN_NR
#define ICP_S1_REJECT_MIN_NR 1024u   // stage 1: exact early rejection from this many representatives on
#endif

typedef float float2v __attribute__ ((ext_vector_type (2)));

// The kernel's explicit arguments as the kernarg segment lays them out (k_search's parameter list, in order): phases that hardly any launch
// takes fetch their arguments WHERE THEY ARE USED through an opaque copy of the segment pointer (held from the top of the kernel they make
// the compiler spill scalars in front of the prologue's loads).  One place knows where icp_params sits in the segment: a
// change of k_search's signature has to change this mirror with it (the launcher's KS_ARGS / KS_CHAIN_ARGS pass exactly these, in order).
struct k_search_args {
    const float *gM, *gR; icp_reg_state *gst; const double *gmom;
    uint32_t m, nr, side, tpr_magic, nb, check_flags;
    icp_params p;
};
#define KS_PARAMS_OFFSET offsetof (k_search_args, p)
static_assert (KS_PARAMS_OFFSET == 4 * 8 + 6 * 4, "k_search: four pointers and six dwords in front of icp_params");
static_assert (KS_PARAMS_OFFSET % alignof (icp_params) == 0, "kernel-argument layout: icp_params follows the scalars without padding");
static __device__ __forceinline__ const icp_params *ks_params_from_kernarg ()
{
    unsigned long long a_ = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr () + KS_PARAMS_OFFSET;
    asm volatile ("" : "+s"(a_));                // (opaque: the scalar loads through it stay where the caller is, they do not join the ones at the kernel's top)
    return (const icp_params *) (const icp_params __attribute__ ((address_space (4))) *) a_;
}
// MASKED search, home tile (see k_search): 1 = staged into LDS in the prologue (seed from LDS); 2 = loads issued in the prologue, LDS
// write after the tile masks (seed from global: the tile's round trip overlaps the seed's); 3 = no home tile (only its list offsets).
// Same box, alternating, us per iteration of fresh 40-iteration runs at |F| = 65536 / 10-iteration runs at 2^20 (profiles/
// r03_home_tile_ab.txt): without the home tile 18.62 / 263.8, mode 1 17.90 / 258.8, mode 2 18.18 / 258.7, mode 3 18.79 / 261.6.
#ifndef ICP_HOME_MODE
#define ICP_HOME_MODE 1
#endif

#ifdef ICP_DBG_STAMPS
#define KS_STAMP(k)                                                                                       \
    {                                                                                                     \
        unsigned long long t_;                                                                            \
        asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        KS_STAMP_STORE (k)                                                                                \
    }
#ifdef ICP_DBG_STAMPS_WAVES     /* every wave's own timeline (row = block x waves per block + wave): which wave a block waits for, and where it was */
#define KS_STAMP_STORE(k) if ((threadIdx.x & 63u) == 0u && p.dbg) p.dbg[((size_t) (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (k)] = t_;
#else
#define KS_STAMP_STORE(k) if (tid == 0 && p.dbg) p.dbg[(size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;
#endif
#elif defined (ICP_DBG_EXIT_AFTER)
// diagnostic builds (tools/diag/phase_insts.sh): the search kernel ends behind phase k — the instruction counters of a PMC run
// then hold the phases up to k, and differences between builds are the phases themselves (every thread of a block gets here)
#define KS_STAMP(k) { if ((k) == ICP_DBG_EXIT_AFTER) return; }
#define KS_KEEP(a, b) asm volatile ("" :: "v"(a), "v"(b));      // (the phase's results count as used: nothing of it is optimised away)
#else
#define KS_STAMP(k)
#endif
#ifndef KS_KEEP
#define KS_KEEP(a, b)
#endif

// minimum over each group of LPQ (8 or 16) consecutive lanes, returned in all of them (min is exact: any pairing
// gives the same bits)
// One v_min_f32 whose first operand comes through DPP per step: fminf (v, dpp (v)) compiles to a DPP move, two canonicalising
// v_max and the v_min — the values here are results of arithmetic (never signalling NaNs) and v_min_f32 returns the other
// operand for a quiet NaN exactly as fminf does, so the one instruction gives the same bits (tests: every distance bit for bit).
// (s_nop 1: a DPP operand must not be read for two wait states after a VALU wrote it; the assembler text is opaque to the
// compiler's hazard pass.)
#define KS_MIN_DPP(v, CTRL) asm ("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "v"(v))
template <int LPQ> static __device__ __forceinline__ float ks_grp_min_f (float v)
{
    KS_MIN_DPP (v, "quad_perm:[1,0,3,2]");
    KS_MIN_DPP (v, "quad_perm:[2,3,0,1]");
    KS_MIN_DPP (v, "row_half_mirror");               // lane i <-> 7 - i
    if (LPQ == 16) KS_MIN_DPP (v, "row_mirror");     // lane i <-> 15 - i
    return v;
}
template <int LPQ> static __device__ __forceinline__ uint32_t ks_grp_min_u (uint32_t v)
{
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0xB1, 0xF, 0xF, true));
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x4E, 0xF, 0xF, true));
    v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x141, 0xF, 0xF, true));
    if (LPQ == 16) v = min (v, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x140, 0xF, 0xF, true));
    return v;
}

// Invalid points (a Kinect frame's pixels without depth: x = y = z = 0, the colour kept — reference src/kinect_frame_grabber.cpp:246-262,
// kernels/icp_kernels.cl:49-50) among the members of a pruning box would stretch it from the scene to the origin: a box every query is
// near, i.e. no pruning at all in a frame with holes.  The boxes are therefore built over the representatives that are NOT at the
// origin (k_reps_and_boxes), and those at the origin come as a compact list of their own (p.OL: colour + index, ascending): their
// geometric term is the same for all of them — exactly qq = fma (qz, qz, fma (qy, qy, qx qx)), the metric's operations on q - 0 —, so
// a query either needs none of them (qq above its bound: every valid point of a scene) or scans the list (a query that is itself an
// invalid point, moved by T: its nearest representative is the invalid one nearest in colour).  Same bits as the exhaustive scan:
// the list is visited behind the tiles, out of index order, so its updates carry the tie rule explicitly (equal distance: lower index).
// The seed of the stage-1 bound against a frame's invalid points (dense variants).  A registration's first search — and the owner search of
// buildRBC — seeds a query with the representative sampled from its own grid cell: where that is an invalid point (at the origin) and the
// query is not, or the other way round, the distance to it bounds nothing, no tile and no group is pruned, and the block stages and scans
// every tile of the set for that one query (|F| = 2^20 with 10 % invalid points: first search 308 -> 697 us, owner search 127 -> 605).
// Such a query takes another seed: the representative nearest by index to its own seed that is of the query's kind — at the origin for an
// invalid query (flagged by the query wave: bit 31 of the seed it hands over), not at the origin for a valid one — looked up in the
// ballots k_reps_and_boxes leaves (icp_other_kind_near): a neighbour on the grid, near the query in space or, neighbouring pixels, in colour.  Any
// representative is a legitimate seed — the bound stays exact.  Clean frames: one compare and a scalar branch per search behind the first.
static __device__ __forceinline__ void ks_seed_against_invalid (uint32_t sfl, const float4 *s_count, uint32_t nr, uint32_t b, const float4 *R4, uint32_t &seed,
                                                                float &sx, float &sy, float &sz, float &sr, float &sg, float &sb)
{
    // (sfl: the two flag bits the query wave put on the seed — 2: the query is an invalid point, 1: the seed is its grid cell's representative.
    // A valid query seeded with its previous winner has neither: a later search of a clean frame leaves at this one test)
    if (__builtin_expect (__ballot (sfl != 0u) == 0ull, 1)) return;
    const bool hq = (sfl & 2u) != 0u;
    const bool s0 = sx == 0.f && sy == 0.f && sz == 0.f;
    if (__ballot (hq != s0) == 0ull) return;
    if (__builtin_amdgcn_readfirstlane ((int) __float_as_uint (s_count->w)) == 0) return;      // (no representative at the origin: nothing to choose from — s_count: hi of box 0 in LDS)
    const icp_params *po = ks_params_from_kernarg ();
    typedef float4 __attribute__ ((address_space (1))) *gf4;
    const float4 *OLb = (const float4 *) (gf4) po->OL + (size_t) b * ICP_OL_STRIDE (nr);
    if (hq != s0) {
        uint32_t s2 = icp_other_kind_near (reinterpret_cast<const unsigned long long *> (OLb + ICP_OL_MASKS (nr)), nr, seed, s0, hq ? __float_as_uint (OLb[1].w) : seed);
        s2 = min (s2, nr - 1u);
        const float4 g = R4[2 * (size_t) s2], c = R4[2 * (size_t) s2 + 1];
        seed = s2; sx = g.x; sy = g.y; sz = g.z; sr = c.x; sg = c.y; sb = c.z;
    }
}

template <int LPQ>
static __device__ __forceinline__ void ks_origin_list (const float4 *ent, const float4 *box, uint32_t n_e, bool boxed, float qq, bool need, float qr, float qg, float qb, float alpha,
                                                       float lim, uint32_t lane, uint32_t ss, float &best, uint32_t &bid)
{
    // One staged segment of the list (LDS: k_search stages it for the whole block, entries `ent[0 .. n_e)`, the colour boxes of their chunks
    // of 8 behind them) — a wave that fetched the list from memory by itself paid 2 - 4 dependent round trips for a few dozen entries, a
    // quarter to a half of a dense search's time for every wave with one invalid query in it (|F| = 65536, 10 % invalid points scattered:
    // 13.0 -> 19.6 us).  qq = |q|^2: every member's geometric term.  need: d >= qq for every member, so none can win, or tie at a lower
    // index, when qq is above the query's bound.
    // The members differ in colour only: chunks of 8 consecutive entries — neighbours on the representative grid where the invalid points
    // form regions: similar colours — have a colour box each (k_reps_and_boxes), and a chunk whose bound fma (a, pho (box), qq) is ABOVE the
    // query's bound holds no winner and no tie (<=, not <: the list is visited behind the tiles, a member may tie with the best so far at a
    // lower index).  Lane ss tests chunk cb + ss; the chunks that pass are evaluated one entry per lane.
    if (!boxed) {
        // (a short list as it comes, four entries per lane and trip: up to 128 entries the tests below cost more than the entries they save)
        for (uint32_t e0 = 0; e0 < n_e; e0 += 4u * (uint32_t) LPQ) {
            float4 v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) v[k] = ent[min (e0 + ss + k * (uint32_t) LPQ, n_e - 1u)];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const float dr_ = qr - v[k].x, dg_ = qg - v[k].y, db_ = qb - v[k].z;
                const float d = __builtin_fmaf (alpha, __builtin_fmaf (db_, db_, __builtin_fmaf (dg_, dg_, dr_ * dr_)), qq);
                const uint32_t idx = __float_as_uint (v[k].w);
                if (need && (d < best || (d == best && idx < bid))) { best = d; bid = idx; }
            }
        }
        return;
    }
    const uint32_t n_oc = (n_e + 7u) >> 3;
    for (uint32_t cb = 0; cb < n_oc; cb += (uint32_t) LPQ) {
        const uint32_t c = min (cb + ss, n_oc - 1u);
        const float4 lo = box[2u * c], hi = box[2u * c + 1u];
        const float er = fmaxf (fmaxf (lo.x - qr, qr - hi.x), 0.f), eg = fmaxf (fmaxf (lo.y - qg, qg - hi.y), 0.f), eb = fmaxf (fmaxf (lo.z - qb, qb - hi.z), 0.f);
        const float bound = __builtin_fmaf (alpha, __builtin_fmaf (eb, eb, __builtin_fmaf (eg, eg, er * er)), qq);
        const float lim2 = fminf (lim, ks_grp_min_f<LPQ> (best));
        const bool pass = need && cb + ss < n_oc && bound <= lim2;
        const unsigned long long bal = __ballot (pass);
        uint32_t mask = (uint32_t) (bal >> (lane & (64u - (uint32_t) LPQ))) & ((1u << LPQ) - 1u);      // bit k: chunk cb + k, for this query
        while (__ballot (mask != 0u)) {
            const bool live = mask != 0u;
            const uint32_t e = 8u * (cb + (live ? (uint32_t) __builtin_ctz (mask) : 0u)) + (ss & 7u);
            mask &= mask - 1u;
            const float4 v = ent[min (e, n_e - 1u)];                  // (clamped: a duplicate of the last entry changes nothing under the explicit tie rule)
            const float dr_ = qr - v.x, dg_ = qg - v.y, db_ = qb - v.z;
            const float d = __builtin_fmaf (alpha, __builtin_fmaf (db_, db_, __builtin_fmaf (dg_, dg_, dr_ * dr_)), qq);
            const uint32_t idx = __float_as_uint (v.w);
            if (live && (d < best || (d == best && idx < bid))) { best = d; bid = idx; }
        }
    }
}

// STAGE 1, the representatives at the origin (a frame's invalid points; kept out of the pruning boxes): scanned behind the tiles by the queries
// that are near the origin.  Block-uniform entry (n_origin comes from LDS); s_pair = the tile buffer (PB_ float4), s_ovote = a block-wide flag
// that is zero on entry.
template <int LPQ, uint32_t PB_>
static __device__ __forceinline__ void ks_origin_section (float4 *s_pair, uint32_t *s_ovote, uint32_t n_origin, uint32_t nr, uint32_t b, uint32_t slice, uint32_t tid,
                                                          float qx, float qy, float qz, float qr, float qg, float qb, float alpha, float s1_lim,
                                                          uint32_t lane, uint32_t ss, float &best, uint32_t &bid)
{
    const float qq = __builtin_fmaf (qz, qz, __builtin_fmaf (qy, qy, qx * qx));
    const bool need = qq <= s1_lim;
    // the list's pointer: fetched here, through an opaque copy of the kernarg pointer (held from the top of the kernel it made the compiler
    // spill scalars in front of the prologue's loads: search at |F| = 65536 11.88 -> 12.22 us)
    auto list_base = [&] () -> const float4 * {
        const icp_params *po = ks_params_from_kernarg ();
        typedef float4 __attribute__ ((address_space (1))) *gf4;
        return (const float4 *) (gf4) po->OL + (size_t) b * ICP_OL_STRIDE (nr);
    };
    // (a list of up to 32 entries — one trip of a query's lanes — is read per wave: with a block's invalid queries in its last wave(s) only
    // those pay for it, and nobody stands at the two barriers of the staging: 64 x 16384 with 10 % invalid points 2.30 -> 2.24 us.)
    // (a longer list the tile buffer holds: staged once for the block's queries — the buffer is free behind the barrier.
    // Measured and not kept: the first segment fetched straight into a buffer of its own from the prologue on (global_load_lds, by
    // the builtin and written out), by the block or by every wave for itself (no barrier at all): |F| = 65536 with 10 - 30 %
    // invalid points 20.6 - 23.8 against 20.6 - 23.4 us, 64 x 16384: 2.28 against 2.35; colour boxes for lists of 16 / 48 entries
    // and more: slower.  What is left of a wave's 1 - 1.5 us here is the scan itself on a busy SIMD: profiles/r05_stamps_holes_dense.txt.
    // Again with the invalid queries in a block's last waves: every wave that needs the list fetching it for itself (global_load_lds into a
    // buffer of the block, no vote, no barrier) — |F| = 65536 21.2 -> 20.9 (10 % scattered), 21.5 -> 21.8 (30 % contiguous): not kept)
    constexpr uint32_t OE = ((PB_ * 4u / 5u) / 8u) * 8u;             // entries the tile buffer holds with the boxes of their chunks of 8 behind them: OE + OE / 4 <= PB_
    const bool boxed = n_origin > ICP_OL_BOXED_MIN;
    if (n_origin > ICP_OL_STAGED_MIN && n_origin <= OE) {
        if (need) *s_ovote = 1u;
        __syncthreads ();
        if (*s_ovote) {
            const float4 *OLb = list_base ();
            if constexpr (OE <= 64u * LPQ) {                          // (one load per thread)
                // (the thread's number from the wave's and the lane's, not from the register the kernel received it in: held
                // until here it costs the 64-register variants a spill in the prologue)
                const uint32_t tl = slice * 64u + __builtin_amdgcn_mbcnt_hi (~0u, __builtin_amdgcn_mbcnt_lo (~0u, 0u));
                if (tl < n_origin) s_pair[tl] = OLb[1u + tl];
                if (boxed && tl < 2u * ((n_origin + 7u) >> 3)) s_pair[OE + tl] = OLb[1u + nr + tl];
            } else {
                for (uint32_t k = tid; k < n_origin; k += 64u * LPQ) s_pair[k] = OLb[1u + k];
                if (boxed) for (uint32_t k = tid; k < 2u * ((n_origin + 7u) >> 3); k += 64u * LPQ) s_pair[OE + k] = OLb[1u + nr + k];
            }
            __syncthreads ();
            if (__ballot (need)) ks_origin_list<LPQ> (s_pair, s_pair + OE, n_origin, boxed, qq, need, qr, qg, qb, alpha, s1_lim, lane, ss, best, bid);
        }
    } else if (__ballot (need)) {
        // a short list, or one the tile buffer does not hold at once: every wave that needs it reads it from memory by itself, a long one's
        // chunk boxes before the chunks (staged a segment at a time it costs two block-wide barriers per segment: |F| = 2^20 with 10 - 30 %
        // invalid points, 400 - 1200 entries, 287 - 393 -> 293 - 403 us; through both tile buffers of the small-tile variant,
        // 608 entries at once: no gain either)
        const float4 *OLb = list_base ();
        ks_origin_list<LPQ> (OLb + 1u, OLb + 1u + nr, n_origin, boxed, qq, need, qr, qg, qb, alpha, s1_lim, lane, ss, best, bid);
    }
}

// candidate j of a list: XQ = [x r y g | z b id 0].  The geometric and the photometric sum of the metric are
// evaluated side by side, one packed instruction per step: (dx, dr), (dy, dg), (dz, db) -> (geo, pho) with exactly the
// operations of icp_metric8 (mul, fma, fma per half), then d = fma (a, pho, geo).  Keeps the best (distance, position).
#define KS_CAND(G, C, J)                                                                              \
    {                                                                                                 \
        const float2v d1_ = vq_xr - float2v { (G).x, (G).y }, d2_ = vq_yg - float2v { (G).z, (G).w }, \
                      d3_ = vq_zb - float2v { (C).x, (C).y };                                         \
        const float2v gp_ = __builtin_elementwise_fma (d3_, d3_, __builtin_elementwise_fma (d2_, d2_, d1_ * d1_)); \
        const float d_ = __builtin_fmaf (alpha, gp_.y, gp_.x);                                        \
        if (d_ < best2) { best2 = d_; bj = (J); }                                                     \
    }

#define KS_CAND_IF(G, C, J, LIVE)                                                                     \
    {                                                                                                 \
        const float2v d1_ = vq_xr - float2v { (G).x, (G).y }, d2_ = vq_yg - float2v { (G).z, (G).w }, \
                      d3_ = vq_zb - float2v { (C).x, (C).y };                                         \
        const float2v gp_ = __builtin_elementwise_fma (d3_, d3_, __builtin_elementwise_fma (d2_, d2_, d1_ * d1_)); \
        const float d_ = __builtin_fmaf (alpha, gp_.y, gp_.x);                                        \
        if (d_ < best2 && (LIVE)) { best2 = d_; bj = (J); }                                           \
    }

#define ICP_NMOM 18

#ifdef ICP_DBG_STAMPS
#define FF_STAMP(k)                                                                                       \
    {                                                                                                     \
        unsigned long long t_;                                                                            \
        asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && p.dbg) p.dbg[(size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;    \
    }
#else
#define FF_STAMP(k)
#endif

// XCD-aware block -> tile mapping (fused mode, one-block-per-CU variants).  Workgroups are dealt round-robin over the 8 XCDs
// (blocks b and b + 8 share one; every XCD has its own L2).  Block b works on tile (b mod 8) * (nb / 8) + b / 8: the blocks of
// one XCD cover a contiguous band of tile rows, and — what the measurement says matters — a contiguous range of moment
// slots: every 128-byte line of the per-block moments is then written inside ONE L2 instead of collecting eight partial
// write-backs.  Measured at A (same box, alternating): identity 9.61, bands 9.42, a 2 x 4 arrangement of compact rectangles
// (better list locality, lines shared by four XCDs again) 9.66 us per iteration.  Only which block computes which tile
// changes — the tile index is what the query index, the moment slot and the canonical trees use, so the bits do not.
// (Speed only: nothing depends on the placement actually being round-robin.)
static __device__ __forceinline__ uint32_t ks_tile_of_block (uint32_t bx, uint32_t nbx)
{
#ifdef ICP_NO_XCD_MAP
    return bx;
#else
    return (nbx & 7u) == 0u ? (bx & 7u) * (nbx >> 3) + (bx >> 3) : bx;
#endif
}

// fused mode: query index of local element e of block b (CPU twin: orc_fused_query).  8 x 8 tiles of the
// landmark grid when its side is a multiple of 8, else 64 consecutive queries.
// tpr_magic = floor (2^32 / tpr) + 1 (host: icp_tpr_magic): b / tpr == umulhi (b, tpr_magic) for b * tpr < 2^32.
static __device__ __forceinline__ uint32_t fused_query_index (uint32_t m, uint32_t side, uint32_t tpr_magic, uint32_t b, uint32_t e)
{
    const uint32_t tpr = side >> 3, ty = __umulhi (b, tpr_magic), tx = b - ty * tpr;
    const uint32_t tiled = (8u * ty + (e >> 3)) * side + 8u * tx + (e & 7u);
    return (side && (side & 7u) == 0u && side * side == m) ? tiled : b * 64u + e;
}


// Representative sampled from the grid cell of point i (getReps' grid: cells of (side / nrx) x (side / nry) points), without a
// division: the magics are floor (2^32 / d) + 1 for d = side, side / nrx, side / nry (host: icp_div_magic; exact for n d < 2^32).
static __device__ __forceinline__ uint32_t cell_rep_of (const icp_params &p, uint32_t i)
{
    const uint32_t y = __umulhi (i, p.side_magic), x = i - y * p.side;
    return (p.cellh_magic ? __umulhi (y, p.cellh_magic) : y) * p.nrx + (p.cellw_magic ? __umulhi (x, p.cellw_magic) : x);      // (magic 0: cells of one point)
}

// Progress word of a host-driven checked run (ICP_MIRROR_WORD) -> fine-grained host memory: one 8-byte system-scope store nobody waits for
static __device__ __forceinline__ void icp_mirror_store (unsigned long long *dst, unsigned long long v)
{
    __hip_atomic_store (dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Result of turning one iteration's moments into the next transform: the registration state itself, staged
// in LDS (one per block) so that one wave publishes it with a single store
typedef icp_reg_state icp_fin_result;

// Block-cooperative: 128-position double trees over the block moments (rows 0..17 of the calling block,
// needs >= 288 threads), then wave 0: means / S from the moments (oracle orc_moments_finish), rotation,
// composition with the previous (T, R), convergence.  Every thread of the block must call it; `res` is valid
// for all threads after the call.  NG = capacity of the second tree level (groups of 128 blocks).
// `sv` = the previous state, lane-distributed: lane j of every wave holds dword j of the icp_reg_state (one
// coalesced vector load that is in flight together with the moment loads; scalar loads of the state would be
// waited for before the moment addresses exist).
static __device__ __forceinline__ uint32_t state_load_lanes (const icp_reg_state *st)
{
    const uint32_t lane = threadIdx.x & 63u;
    return reinterpret_cast<const uint32_t *> (st)[min (lane, (uint32_t) sizeof (icp_reg_state) / 4u - 1u)];
}
static __device__ __forceinline__ float state_lane_f (uint32_t sv, uint32_t dword)
{
    return __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) dword));
}
#define ICP_ST_DW(field) (offsetof (icp_reg_state, field) / 4)

// First tree level of the moments, pass ps: one 16-lane row per (moment k, group g) task.  The loads are a separate
// step so that a caller can issue them together with its other prologue loads (before anything waits).
template <int NT>
static __device__ __forceinline__ void fused_moment_task (uint32_t nb, uint32_t ps, uint32_t &k, uint32_t &g, bool &live)
{
    constexpr uint32_t nrow = NT / 16;
    const uint32_t row = threadIdx.x >> 4, ng = (nb + 127u) / 128u, ntask = ICP_NMOM * ng;
    const uint32_t task = min (ps * nrow + row, ntask - 1u);
    if (ng == 1) { k = task; g = 0u; } else if (ng == 2) { k = task >> 1; g = task & 1u; } else { k = task / ng; g = task - k * ng; }
    live = ps * nrow + row < ntask;                  // rows past the last task (whole waves when ntask % 4 == 0) load nothing
}
template <int NT>
static __device__ __forceinline__ void fused_moment_loads (const double *mom, uint32_t nb, uint32_t ps, double *a)
{
    uint32_t k, g; bool live;
    fused_moment_task<NT> (nb, ps, k, g, live);
    const double *src = mom + (size_t) k * nb;
    const uint32_t l = threadIdx.x & 15u;
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.0;
    if (live) {
        if (nb & 127u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {            // clamped address + select: eight loads back to back
                const uint32_t i = g * 128u + l + 16u * q;
                const double t = src[min (i, nb - 1u)];
                a[q] = (i < nb) ? t : 0.0;
            }
        } else {
            // whole groups of 128 block moments (|F| a multiple of 8192, e.g. the 16384 of the reference): no position is past the end —
            // the selects (two v_cndmask + a compare per load, executed as the loads arrive) sit on the path from the moments to T
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = src[g * 128u + l + 16u * q];
        }
    }
}

// Returns false (for every thread, before any barrier) when the registration had already converged (checked mode).
// a0 = the values of pass 0 (fused_moment_loads (mom, nb, 0, a0), issued by the caller with its other loads).
// gl1 != nullptr: the first tree level was evaluated by k_moment_level1 (many blocks: large sets); gl1[k * ng + g].
// LEAN (the chained search's prologue: every block runs this, all of them wait for T): the block is handed T and `done` only —
// ten LDS dwords instead of 62 —, and the one block that publishes the state (direct != nullptr) stores it straight from the
// composing lane's registers to global memory (16 vector stores nobody waits for).
// after (LEAN): called by every lane of the finishing wave with the new T, before the barrier that releases the block — the
// chained search transforms and hands over its queries there, so that one barrier covers T's consumers.
struct ff_no_hook { __device__ void operator() (const float *) const {} };
template <int NG, int NT, int ROT, bool LEAN = false, typename AFTER = ff_no_hook>
static __device__ bool fused_finalize_block (const icp_params &p, const double *mom, uint32_t nb, uint32_t check, uint32_t sv,
                                             const double *a0, icp_fin_result *res, double (*s_l1)[NG], double *s_t,
                                             const double *gl1 = nullptr, icp_reg_state *direct = nullptr, AFTER after = AFTER (),
                                             uint32_t pending_unless_done = 1u, unsigned long long *mirror = nullptr, bool progress = false,
                                             icp_reg_state *final_dst = nullptr, icp_reg_state *host_dst = nullptr, icp_reg_state *prev_dst = nullptr)
{
    // Host-driven checked runs (run_ctl in icp_capi.hip; mirror != nullptr on the lane that publishes): `progress` — every new (k, done) goes
    // to the registration's word in host memory (chained form: nothing waits for the store); a CONVERGED registration's final state is stored
    // to final_dst (chained: the user-visible state, which the launches behind this one no longer touch) and host_dst (host memory) in front
    // of the word's DONE | FINAL bits: the run needs no end kernel and the host reads the result the moment the flag shows.
    // prev_dst: the transform the search of this iteration used (T before the composition) — what a later search needs to reproduce the
    // iteration's per-query outputs, which checked runs do not store on the way (icp_launch_search on p.st_prev).
    // NT = threads of the calling block (compile-time: reading blockDim costs a dependent cold load at kernel start)
    constexpr uint32_t nrow = NT / 16;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, l = tid & 15u, row = tid >> 4;
    const uint32_t ng = (nb + 127u) / 128u;
    const uint32_t ntask = ICP_NMOM * ng, npass = (ntask + nrow - 1u) / nrow;
    auto pass = [&] (uint32_t ps, const double *a) {
        uint32_t k, g; bool live;
        fused_moment_task<NT> (nb, ps, k, g, live);
        FF_STAMP (13)
        double v = row_tree8_d (a);
        if (nb == 1) v = a[0];
        if (ng == 2) {
            // the two groups of a moment sit in adjacent rows of one wave, and the second level (a 128-position
            // tree over [g0, g1, 0, ..]) is (g0 + 0) + (g1 + 0): no LDS round trip, no second barrier
            const double o = __shfl_down (v, 16);
            if (l == 0 && !(row & 1u) && live) s_t[k] = (v + 0.0) + (o + 0.0);
        } else if (l == 0 && live) { if (ng == 1) s_t[k] = v; else s_l1[k][g] = v; }
    };
    if (gl1 == nullptr) {
        pass (0u, a0);
        for (uint32_t ps = 1; ps < npass; ++ps) {    // small blocks / many groups only
            double a[8];
            fused_moment_loads<NT> (mom, nb, ps, a);
            pass (ps, a);
        }
    }
    if (check && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done))) return false;    // block-uniform
    FF_STAMP (9)
    if (ng > 2) {                                    // second level: rows 0..17, one moment each
        __syncthreads ();
        if (row < 20) {
            const uint32_t k = min (row, (uint32_t) ICP_NMOM - 1u);
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t i = l + 16u * q;
                a[q] = (i < ng) ? (gl1 ? gl1[(size_t) k * ng + i] : s_l1[k][i]) : 0.0;
            }
            double r = row_tree8_d (a);
            if (l == 0 && row < ICP_NMOM) s_t[row] = r;
        }
    }
    __syncthreads ();
    FF_STAMP (10)
    if (tid < 64) {
        double t[ICP_NMOM];
#pragma unroll
        for (int k = 0; k < ICP_NMOM; ++k) t[k] = s_t[k];
        const double sw = t[0];
        // oracle orc_moments_finish: ONE division, the means by multiplication, fused multiply-adds (this wave is what the
        // block — in the chained form the whole grid — waits for: six double divisions and the unfused products measured
        // 0.5 us of the iteration at A; a division per lane with the quotients handed round as scalars measured slower than
        // six overlapping ones, 9.19 -> 9.24 us)
        const double rs = 1.0 / sw;
        double mf[3], mq[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mf[a] = t[1 + a] * rs; mq[a] = t[4 + a] * rs; }
        const double c2 = (double) p.c * (double) p.c;
        float S[11], means[8];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) S[3 * a + bb] = (float) (c2 * __builtin_fma (-t[4 + a], mf[bb], t[7 + 3 * a + bb]));
        S[9]  = (float) (c2 * (t[16] - __builtin_fma (t[3], mf[2], __builtin_fma (t[2], mf[1], t[1] * mf[0]))));
        S[10] = (float) (c2 * (t[17] - __builtin_fma (t[6], mq[2], __builtin_fma (t[5], mq[1], t[4] * mq[0]))));
        means[0] = (float) mf[0]; means[1] = (float) mf[1]; means[2] = (float) mf[2]; means[3] = 0.f;
        means[4] = (float) mq[0]; means[5] = (float) mq[1]; means[6] = (float) mq[2]; means[7] = 0.f;
        float Tk[8], Rk[9], Rkin[9];
        int iters = 0;
        FF_STAMP (11)
        // ROT = rotation solver, compile-time: a power-method kernel carries no SVD code (registers, instruction cache)
        if constexpr (ROT == 1) iters = icp_power_method_quad (S, means, Tk, p.power_mode, lane);
        else icp_svd_rotation (S, means, Rkin, Tk);
        FF_STAMP (12)
#ifdef ICP_DBG_STAMPS
        if (lane < 8 && p.dbg && ROT == 1) p.dbg[16 + lane] = icp_pm_stamps[lane];
#endif
        float Tprev[8], Rprev[9];
#pragma unroll
        for (int k = 0; k < 8; ++k) Tprev[k] = state_lane_f (sv, ICP_ST_DW (T) + k);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rprev[k] = state_lane_f (sv, ICP_ST_DW (R) + k);
        const uint32_t kprev = (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k));
        if constexpr (LEAN) {
            // the composition on every lane (the same instructions as on one): T is then in registers where the hook wants it
            float Tn[8], Rn[9];
            icp_compose_pure (Tprev, Rprev, Tk, Rkin, ROT != 1, Tn, Rn, Rk);
            const uint32_t done = (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) ? 1u : 0u;
            after (Tn);
            if (lane == 0) {
                res->done = done;
                if (direct) {
                    // the state image, dword for dword what fin_result_to_state would publish (pending = !done)
                    typedef float f4u __attribute__ ((ext_vector_type (4), aligned (4)));
                    typedef float f2u __attribute__ ((ext_vector_type (2), aligned (4)));
                    static_assert (sizeof (icp_reg_state) == 62 * 4 && ICP_ST_DW (means) == 46 && ICP_ST_DW (sum_w) == 54 && ICP_ST_DW (k) == 56, "state layout");
                    const unsigned long long swb = __builtin_bit_cast (unsigned long long, sw);
                    const float img[64] = {
                        Tn[0], Tn[1], Tn[2], Tn[3], Tn[4], Tn[5], Tn[6], Tn[7], Tk[0], Tk[1], Tk[2], Tk[3], Tk[4], Tk[5], Tk[6], Tk[7],
                        Rn[0], Rn[1], Rn[2], Rn[3], Rn[4], Rn[5], Rn[6], Rn[7], Rn[8], Rk[0], Rk[1], Rk[2], Rk[3], Rk[4], Rk[5], Rk[6], Rk[7], Rk[8],
                        S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8], S[9], S[10], 0.f,
                        means[0], means[1], means[2], means[3], means[4], means[5], means[6], means[7],
                        __uint_as_float ((uint32_t) swb), __uint_as_float ((uint32_t) (swb >> 32)),
                        __uint_as_float (kprev + 1u), __uint_as_float (done), __uint_as_float ((uint32_t) iters), __uint_as_float (done ? 0u : pending_unless_done),
                        0.f, 0.f, 0.f, 0.f };
                    auto store_image = [&] (icp_reg_state *to) {
                        float *dst = reinterpret_cast<float *> (to);
#pragma unroll
                        for (int k = 0; k < 15; ++k) *reinterpret_cast<f4u *> (dst + 4 * k) = f4u { img[4 * k], img[4 * k + 1], img[4 * k + 2], img[4 * k + 3] };
                        *reinterpret_cast<f2u *> (dst + 60) = f2u { img[60], img[61] };
                    };
                    store_image (direct);
                    if (prev_dst) {
                        float *dst = reinterpret_cast<float *> (prev_dst);
                        *reinterpret_cast<f4u *> (dst) = f4u { Tprev[0], Tprev[1], Tprev[2], Tprev[3] };
                        *reinterpret_cast<f4u *> (dst + 4) = f4u { Tprev[4], Tprev[5], Tprev[6], Tprev[7] };
                    }
                    if (mirror) {
                        if (done) {
                            if (final_dst) store_image (final_dst);
                            if (host_dst) { store_image (host_dst); __threadfence_system (); }
                            icp_mirror_store (mirror, ICP_MIRROR_WORD (p.epoch, kprev + 1u, 1u) | (host_dst ? ICP_MIRROR_FINAL : 0ull));
                            if (progress && p.run_flag) {
                                // the run is over: its later launches see the flag; a tracked sequence's next frame (held by k_gate on
                                // the other stream) may start — behind the user-visible state above (release at agent scope)
                                p.run_flag[blockIdx.y] = p.epoch;
                                if (p.track_seq) __hip_atomic_store (p.track_seq, p.seq_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        } else if (progress) icp_mirror_store (mirror, ICP_MIRROR_WORD (p.epoch, kprev + 1u, 0u));
                    }
                }
            }
        } else if (lane == 0) {
            float Tn[8], Rn[9];
            icp_compose_pure (Tprev, Rprev, Tk, Rkin, ROT != 1, Tn, Rn, Rk);
#pragma unroll
            for (int k = 0; k < 8; ++k) { res->T[k] = Tn[k]; res->Tk[k] = Tk[k]; res->means[k] = means[k]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { res->R[k] = Rn[k]; res->Rk[k] = Rk[k]; }
#pragma unroll
            for (int k = 0; k < 11; ++k) res->S[k] = S[k];
            res->sum_w = sw; res->pm_iters = (uint32_t) iters; res->k = kprev + 1u; res->pad0 = 0.f; res->pending = 0u;
            res->reserved0 = 0u; res->reserved1 = 0u;
            res->done = (p.check && icp_check_converged (Tk, p.tan_half_thr, p.trans_thr)) ? 1u : 0u;
            if (prev_dst) {
#pragma unroll
                for (int k = 0; k < 8; ++k) prev_dst->T[k] = Tprev[k];
            }
        }
    }
    __syncthreads ();
    return true;
}

// cooperative publish: thread t of the block copies dword t (call with the whole first wave)
static __device__ __forceinline__ void fin_result_to_state (const icp_fin_result *res, icp_reg_state *st, uint32_t pending)
{
    static_assert (sizeof (icp_reg_state) / 4 <= 64, "the state is published by one wave");
    const uint32_t t = threadIdx.x;
    if (t < sizeof (icp_reg_state) / 4) {
        uint32_t v = reinterpret_cast<const uint32_t *> (res)[t];
        if (t == offsetof (icp_reg_state, pending) / 4) v = pending;
        reinterpret_cast<uint32_t *> (st)[t] = v;
    }
}

// ------------------------------------------------------------------------------------------
// ks_list_tail — stage 2, what lies beyond the unconditionally scanned head of a LONG list (both forms of stage 2 end here; no clean
// scene has such a list: every branch into this function is a scalar one that is never taken then).
// Exact pruning over chunks of 16 consecutive positions (boxes: k_list_boxes).  A query tests a chunk with the
// metric's own operations applied to the per-axis distances to the chunk's 6-D box — every operation is monotone under
// round-to-nearest, so bound <= d of every member (a > 0) — against `lim`: the distance to the representative itself, bumped one
// ulp (the representative is a member of its own list: a nearer-or-equal identical point with a lower index would have been
// the nearest representative instead; so the list's minimum is <= dr, and a chunk whose bound is above dr holds neither the
// minimum nor a tie with it), and the query's best so far (a chunk whose bound is not BELOW it cannot replace a candidate at a
// lower position: trips ascend, updates need a strict '<' — the tie rule of the serial scan).  Lane ss tests the chunks
// cb + ss, cb + ss + LPQ of a round; the answers of a query's lanes come back through a ballot; the chunks that pass are
// scanned in ascending order by all lanes of the query, two chunks per memory round trip.  The bits are those of the
// exhaustive scan; a list of identical points (invalid pixels with their colour zeroed) costs a box test per chunk behind
// its first 128 candidates instead of the candidates themselves.
//   cb0: first chunk to test; ntrips: trips (of LPQ positions) of the wave's longest list; best2 / bj: the lane's best so far (distance, TRIP
//   index: position o + ss + trip * LPQ), updated.
// ------------------------------------------------------------------------------------------
template <int LPQ>
static __device__ __forceinline__ void ks_list_tail (const char *XQb, uint32_t o, uint32_t je, uint32_t alast, uint32_t cb0, uint32_t ntrips, float qx, float qy, float qz,
                                                     float qr, float qg, float qb, float alpha, float dr, uint32_t b, uint32_t lane, uint32_t ss, float &best2, uint32_t &bj)
{
    const float2v vq_xr = { qx, qr }, vq_yg = { qy, qg }, vq_zb = { qz, qb };
    constexpr uint32_t CPT = 16u / KS_SPLIT, BD = KS_SPLIT == 16 ? 2u : 1u;     // trips per chunk; boxes per lane and round (LPQ = 8: the 64-register variants have room for one)
    const uint32_t nch = (je - o + 15u) >> 4, nchw = (ntrips + CPT - 1u) / CPT;     // chunks of this query's list / of the wave's longest
    // (the boxes' base and stride are fetched from the kernel arguments HERE, through an opaque copy of the argument pointer — see the
    // per-query output pointers of the epilogue: left to the compiler, their scalar loads join the ones at the top of the kernel, in
    // front of the prologue's vector loads, for a path hardly any launch takes: 8.66 -> 8.71 us per iteration at |F| = 16384)
    const icp_params *pl = ks_params_from_kernarg ();
    typedef float4 __attribute__ ((address_space (1))) *gf4;
    const float4 *LBq = (const float4 *) (gf4) pl->LB + (size_t) b * 3u * pl->nlb + 3u * (o >> 4);
    const float inf_ = __builtin_inff ();
    float lim = (alpha > 0.f && dr >= 0.f && dr < inf_) ? __uint_as_float (__float_as_uint (dr) + 1u) : inf_;
    const bool bounds = alpha > 0.f;                            // (a <= 0: d >= bound does not hold; everything is scanned)
    for (uint32_t cb = cb0; cb < nchw; cb += BD * KS_SPLIT) {
        lim = fminf (lim, ks_grp_min_f<KS_SPLIT> (best2));
        float4 bx[BD][3];
#pragma unroll
        for (uint32_t j = 0; j < BD; ++j) {
            const uint32_t cc = min (cb + ss + KS_SPLIT * j, max (nch, 1u) - 1u);       // (clamped: inside the buffer; masked below)
#pragma unroll
            for (int k = 0; k < 3; ++k) bx[j][k] = LBq[3u * cc + (uint32_t) k];
        }
        uint32_t cmask = 0u;
#pragma unroll
        for (uint32_t j = 0; j < BD; ++j) {
            const float4 b0 = bx[j][0], b1 = bx[j][1], b2 = bx[j][2];                   // [lo.x lo.y lo.z lo.r | lo.g lo.b hi.x hi.y | hi.z hi.r hi.g hi.b]
            const float ex = fmaxf (fmaxf (b0.x - qx, qx - b1.z), 0.f), ey = fmaxf (fmaxf (b0.y - qy, qy - b1.w), 0.f);
            const float ez = fmaxf (fmaxf (b0.z - qz, qz - b2.x), 0.f), er = fmaxf (fmaxf (b0.w - qr, qr - b2.y), 0.f);
            const float eg = fmaxf (fmaxf (b1.x - qg, qg - b2.z), 0.f), eb = fmaxf (fmaxf (b1.y - qb, qb - b2.w), 0.f);
            const float geo_ = __builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex)), pho_ = __builtin_fmaf (eb, eb, __builtin_fmaf (eg, eg, er * er));
            const bool pass = cb + ss + KS_SPLIT * j < nch && (!bounds || __builtin_fmaf (alpha, pho_, geo_) < lim);
            const unsigned long long bal = __ballot (pass);
            cmask |= ((uint32_t) (bal >> (lane & (64u - KS_SPLIT))) & ((1u << KS_SPLIT) - 1u)) << (KS_SPLIT * j);      // bit k: chunk cb + k of this query's list
        }
        // the chunks that pass, ascending: two per memory round trip where the registers are there (LPQ = 16: the latency variant), one in the
        // 64-register variants (LPQ = 8: a chunk is two trips)
        while (__ballot (cmask != 0u)) {
            constexpr uint32_t NS = KS_SPLIT == 16 ? 2u : 1u;
            bool live[NS]; uint32_t tr[NS];
#pragma unroll
            for (uint32_t u = 0; u < NS; ++u) {
                live[u] = cmask != 0u;
                tr[u] = (cb + (live[u] ? (uint32_t) __builtin_ctz (cmask) : 0u)) * CPT;
                cmask &= cmask - 1u;
            }
            float4 g[NS * CPT], c[NS * CPT];
#pragma unroll
            for (uint32_t u = 0; u < NS; ++u)
#pragma unroll
                for (uint32_t h = 0; h < CPT; ++h) {
                    const char *r0 = XQb + min ((o + ss + (tr[u] + h) * KS_SPLIT) << 5, alast);
                    g[u * CPT + h] = *reinterpret_cast<const float4 *> (r0); c[u * CPT + h] = *reinterpret_cast<const float4 *> (r0 + 16);
                }
#pragma unroll
            for (uint32_t u = 0; u < NS; ++u)
#pragma unroll
                for (uint32_t h = 0; h < CPT; ++h) KS_CAND_IF (g[u * CPT + h], c[u * CPT + h], tr[u] + h, live[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// ks_stage2_wave — stage 2, lanes = candidates (dense variant, long lists: icp_s2_wave_of): called by every lane of the wave.
//   o, n, valid, rstar, q*: the calling lane's query (LPQ = 8: query lane >> 3); dmin / jmin: that query's winner (distance, list
//   position; 0xFFFFFFFF: no candidate) in all of its lanes.
// ------------------------------------------------------------------------------------------
#ifndef ICP_S2W_UNCOND
#define ICP_S2W_UNCOND 1024u         // lanes = candidates: positions of a list scanned unconditionally (16 trips of 64); beyond them chunk boxes first
#endif
static __device__ __forceinline__ void ks_stage2_wave (const char *XQb, uint32_t o, uint32_t n, bool valid, uint32_t rstar, float qx, float qy, float qz,
                                                       float qr, float qg, float qb, float alpha, float dr, uint32_t b, uint32_t lane, const float4 *s_qa, const float4 *s_qc,
                                                       uint32_t qe, float &dmin, uint32_t &jmin)
{
    // ---- stage 2, long lists (dense variant, icp_s2_wave): lanes = candidates.  The scan above is bound by the vector-memory
    // path (every query's lanes load their list for themselves: 24 bytes per candidate and query through the L1); the
    // wave's 8 queries are neighbours and mostly share ONE list, so here the wave loads a list once — lane l takes the
    // positions l, l + 64, .. — and every lane evaluates its candidate against each query of the wave that has this list,
    // the query's six coordinates in SGPRs.  Distinct lists of the wave are served one after the other.  Per lane and
    // query: best (distance, trip); at the end one butterfly over the 64 lanes that halves the number of queries a lane
    // holds while it doubles the lanes reduced ((distance bits, position) as one 64-bit key: distances are >= +0, so
    // the unsigned order of the bits is the order of the values; smallest distance, ties -> lowest position).
    const uint32_t je = valid ? o + n : o;
    unsigned long long todo = __ballot (je != o);                // lanes of the queries with a list to scan
    float sx[8], sy[8], sz[8], sr[8], sg[8], sb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sx[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qx), 8 * q));
        sy[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qy), 8 * q));
        sz[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qz), 8 * q));
        sr[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qr), 8 * q));
        sg[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qg), 8 * q));
        sb[q] = __uint_as_float ((uint32_t) __builtin_amdgcn_readlane ((int) __float_as_uint (qb), 8 * q));
    }
    float bd[8]; uint32_t btr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bd[q] = __builtin_inff (); btr[q] = 0xFFFFFFFFu; }
    // (m8: one bit per query of the wave that has the list; made opaque per trip so that the eight tests stay scalar bit
    // tests inside the loop instead of eight hoisted lane masks)
#define KS_WCAND(G, C, TRIP)                                                                                  \
    asm volatile ("" : "+s"(m8));                                                                         \
    _Pragma ("unroll") for (int q = 0; q < 8; ++q)                                                        \
        if (m8 & (1u << q)) {                                                                             \
            const float2v d1_ = float2v { sx[q], sr[q] } - float2v { (G).x, (G).y },                      \
                          d2_ = float2v { sy[q], sg[q] } - float2v { (G).z, (G).w },                      \
                          d3_ = float2v { sz[q], sb[q] } - float2v { (C).x, (C).y };                      \
            const float2v gp_ = __builtin_elementwise_fma (d3_, d3_, __builtin_elementwise_fma (d2_, d2_, d1_ * d1_)); \
            const float d_ = __builtin_fmaf (alpha, gp_.y, gp_.x);                                        \
            if (d_ < bd[q]) { bd[q] = d_; btr[q] = (TRIP); }                                              \
        }
    while (todo) {
        const int l0 = (int) __builtin_ctzll (todo);
        const uint32_t rL = (uint32_t) __builtin_amdgcn_readlane ((int) rstar, l0);
        const uint32_t oL = (uint32_t) __builtin_amdgcn_readlane ((int) o, l0), nL = (uint32_t) __builtin_amdgcn_readlane ((int) n, l0);
        const unsigned long long match = __ballot (rstar == rL) & todo;      // same representative = same list
        todo &= ~match;
        uint32_t m8 = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) m8 |= (uint32_t) ((match >> (8 * q)) & 1ull) << q;
        m8 = (uint32_t) __builtin_amdgcn_readfirstlane ((int) m8);
        const uint32_t vlastL = (oL + nL - 1u) << 5, ntr = (nL + 63u) >> 6;
        // (a list's first ICP_S2W_UNCOND positions here; what a LONG list holds beyond them is scanned behind chunk-box tests by the
        // queries' own lanes afterwards: see the end of this function)
        constexpr uint32_t TU = ICP_S2W_UNCOND / 64u;
        const uint32_t ntr0 = min (ntr, TU);
        uint32_t voff = (oL + lane) << 5;
        for (uint32_t t = 0; t < ntr0; t += 2u, voff += 2u * 64u * 32u) {
            const char *rec0 = XQb + min (voff, vlastL), *rec1 = XQb + min (voff + 64u * 32u, vlastL);
            const float4 g0 = *reinterpret_cast<const float4 *> (rec0); const float2 c0 = *reinterpret_cast<const float2 *> (rec0 + 16);
            const float4 g1 = *reinterpret_cast<const float4 *> (rec1); const float2 c1 = *reinterpret_cast<const float2 *> (rec1 + 16);
            KS_WCAND (g0, c0, t)
            if (t + 1u < ntr0) { KS_WCAND (g1, c1, t + 1u) }
        }
    }
#undef KS_WCAND
    // the wave's winner per query: (distance bits, position) as one 64-bit key (distances are >= +0: the unsigned order of
    // the bits is the order of the values), and a butterfly over the 64 lanes that halves the queries a lane holds while it
    // doubles the lanes reduced — after three steps lane l holds query l & 7 over its group of 8 lanes, after six over the wave.
    // (Measured against it and slower, 283 -> 292 us at C: the distances alone through the butterfly and the winner's
    // position looked up with ballots / readlanes in scalars.)
    unsigned long long key[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t oq = (uint32_t) __builtin_amdgcn_readlane ((int) o, 8 * q), jeq = (uint32_t) __builtin_amdgcn_readlane ((int) je, 8 * q);
        const uint32_t pos = (btr[q] == 0xFFFFFFFFu) ? 0xFFFFFFFFu : min (oq + lane + 64u * btr[q], jeq - 1u);
        key[q] = ((unsigned long long) __float_as_uint (bd[q]) << 32) | pos;
    }
    auto xchg_dpp = [] (unsigned long long v, auto ctrl) -> unsigned long long {
        const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp (0, (int) (uint32_t) v, decltype (ctrl)::value, 0xF, 0xF, true);
        const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp (0, (int) (uint32_t) (v >> 32), decltype (ctrl)::value, 0xF, 0xF, true);
        return ((unsigned long long) hi << 32) | lo;
    };
    auto xchg_lane = [] (unsigned long long v, uint32_t src) -> unsigned long long {      // v of lane src
        const uint32_t lo = (uint32_t) __builtin_amdgcn_ds_bpermute ((int) (src << 2), (int) (uint32_t) v);
        const uint32_t hi = (uint32_t) __builtin_amdgcn_ds_bpermute ((int) (src << 2), (int) (uint32_t) (v >> 32));
        return ((unsigned long long) hi << 32) | lo;
    };
    auto min64 = [] (unsigned long long a_, unsigned long long b_) { return a_ < b_ ? a_ : b_; };
    unsigned long long k4[4], k2[2], k1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                // lane bit 0: keeps the queries 2j + (lane & 1)
        const bool odd = (lane & 1u) != 0u;
        const unsigned long long keep = odd ? key[2 * j + 1] : key[2 * j], send = odd ? key[2 * j] : key[2 * j + 1];
        k4[j] = min64 (keep, xchg_dpp (send, std::integral_constant<int, 0xB1> {}));      // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                // lane bit 1
        const bool odd = (lane & 2u) != 0u;
        const unsigned long long keep = odd ? k4[2 * j + 1] : k4[2 * j], send = odd ? k4[2 * j] : k4[2 * j + 1];
        k2[j] = min64 (keep, xchg_dpp (send, std::integral_constant<int, 0x4E> {}));      // quad_perm [2,3,0,1]
    }
    {                                            // lane bit 2
        const bool odd = (lane & 4u) != 0u;
        const unsigned long long keep = odd ? k2[1] : k2[0], send = odd ? k2[0] : k2[1];
        k1 = min64 (keep, xchg_lane (send, lane ^ 4u));
    }
    k1 = min64 (k1, xchg_lane (k1, lane ^ 8u));  // over the wave's 8 groups of 8 lanes
    k1 = min64 (k1, xchg_lane (k1, lane ^ 16u));
    k1 = min64 (k1, xchg_lane (k1, lane ^ 32u));
    k1 = xchg_lane (k1, (lane & 56u) | (lane >> 3));            // to the lanes of query lane >> 3
    dmin = __uint_as_float ((uint32_t) (k1 >> 32)); jmin = (uint32_t) k1;
    // ---- long lists (beyond ICP_S2W_UNCOND positions: no clean scene has one; a frame's invalid points with their colours zeroed are ONE
    // point, and the list that holds them all is scanned by every query that is such a point — |F| = 65536, |R| = 256, 30 % of them: 377 us
    // per iteration instead of 23): the rest of such a list is scanned by the query's own 8 lanes behind chunk-box tests (ks_list_tail),
    // starting from the winner of the head: positions ascend, updates need a strict '<' — the serial scan's tie rule.
    {
        uint32_t nl = (je - o + 7u) >> 3;                                // trips of 8 positions; the wave's maximum as in ks_stage2_lanes
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) nl, 0x140, 0xF, 0xF, true));
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x142, 0xA, 0xF, false));
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x143, 0xC, 0xF, false));
        const uint32_t ntrips = (uint32_t) __builtin_amdgcn_readlane ((int) nl, 63);
        if (__builtin_expect (ntrips > ICP_S2W_UNCOND / 8u, 0)) {
            const uint32_t ss = lane & 7u;
            float best2 = dmin; uint32_t bj = 0xFFFFFFFFu;               // (bj: a TRIP of this scan, or none: the head's winner stands)
            // (the query comes back from LDS, where the query wave left it: held in registers across the scan above and the butterfly, its
            // colour alone made the 64-register variants spill)
            const float4 a4 = s_qa[qe], c4 = s_qc[qe];
            ks_list_tail<8> (XQb, o, je, (max (je, 1u) - 1u) << 5, ICP_S2W_UNCOND / 16u, ntrips, a4.x, a4.y, a4.z, c4.x, c4.y, c4.z, alpha, dr, b, lane, ss, best2, bj);
            const uint32_t pos = bj != 0xFFFFFFFFu ? min (o + ss + bj * 8u, max (je, 1u) - 1u) : jmin;
            const float d2 = ks_grp_min_f<8> (best2);
            jmin = ks_grp_min_u<8> (best2 == d2 ? pos : 0xFFFFFFFFu);
            dmin = d2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// ks_stage2_lanes — stage 2, the lanes of a query scan its representative's list (latency variant: LPQ = 16; dense variant, short
// lists: LPQ = 8): exhaustive over the first positions, behind chunk boxes beyond them (long lists: see inside).  dr = the query's
// distance to the representative (the bound of the chunk tests); dmin / jmin as in ks_stage2_wave.
// ------------------------------------------------------------------------------------------
template <int LPQ>
static __device__ __forceinline__ void ks_stage2_lanes (const char *XQb, uint32_t o, uint32_t n, bool valid, float qx, float qy, float qz, float qr, float qg, float qb,
                                                        float alpha, float dr, uint32_t b, uint32_t lane, uint32_t ss, float &dmin, uint32_t &jmin)
{
    float best2 = __builtin_inff (); uint32_t bj = 0xFFFFFFFFu;
    const float2v vq_xr = { qx, qr }, vq_yg = { qy, qg }, vq_zb = { qz, qb };
    {
        // a batch = KS_DEPTH candidates per lane, all loads issued before the first distance (clamped addresses, the
        // tail is masked): one memory round trip per batch, and one batch covers a list of KS_DEPTH * LPQ candidates.
        // The number of trips the wave needs (its longest list) is a scalar: a trip no lane needs is neither loaded
        // nor evaluated, at the cost of scalar compares only (the scan is bound by the vector-memory issue rate).
        constexpr uint32_t KS_DEPTH = (KS_SPLIT == 16) ? ICP_S2_DEPTH16 : 4u;
        const uint32_t je = valid ? o + n : o;       // (invalid queries: an empty range)
        // the wave's trip count = its longest list: the lanes of a query hold the same count, so one mirror inside the 16-lane rows
        // (two queries per row at 8 lanes per query) and the two row broadcasts of a wave reduction leave the maximum in lane 63
        uint32_t nl = (je - o + KS_SPLIT - 1u) / KS_SPLIT;
        if (KS_SPLIT == 8) nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp (0, (int) nl, 0x140, 0xF, 0xF, true));        // row_mirror
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x142, 0xA, 0xF, false));                   // row_bcast:15 -> rows 1, 3
        nl = max (nl, (uint32_t) __builtin_amdgcn_update_dpp ((int) nl, (int) nl, 0x143, 0xC, 0xF, false));                   // row_bcast:31 -> rows 2, 3
        const uint32_t ntrips = (uint32_t) __builtin_amdgcn_readlane ((int) nl, 63);
        // byte offsets from the uniform base (m <= 2^20: < 2^25 bytes); a position past the list's end is clamped to its last
        // element, whose (distance, position) some lane holds anyway — a duplicate changes neither the minimum nor the lowest
        // position among equals, so there is no tail test.  The lane keeps the TRIP of its best candidate (a scalar + constant per
        // candidate instead of a recomputed position); trips ascend, so a strict '<' keeps the lane's lowest position.
        const uint32_t alast = (max (je, 1u) - 1u) << 5;
        // Lists of up to 2 x ICP_S2_UNCOND positions are scanned as they come (a second batch costs less than the test in front of it:
        // |F| = 16384 with a list of 135, 9.54 against 8.95 us per iteration); of longer lists the first ICP_S2_UNCOND positions, and what lies
        // beyond chunk by chunk behind a box test (below).  (The wave's longest list decides: scalar control flow.)
        constexpr uint32_t KS_UNC = ICP_S2_UNCOND / KS_SPLIT;      // (ICP_S2_UNCOND = 0, A/B builds: no chunk tests, every list scanned as it comes)
        const uint32_t ntr0 = (ICP_S2_UNCOND == 0u || ntrips <= 2u * KS_UNC) ? ntrips : KS_UNC;
        for (uint32_t tb = 0; tb < ntr0; tb += KS_DEPTH) {
            const uint32_t a0 = (o + ss + tb * KS_SPLIT) << 5, nt = min (KS_DEPTH, ntr0 - tb);
            float4 g[KS_DEPTH], c[KS_DEPTH];
#pragma unroll
            for (uint32_t t = 0; t < KS_DEPTH; ++t) {
                if (t >= nt) break;
                const char *rec = XQb + min (a0 + t * (KS_SPLIT * 32u), alast);
                g[t] = *reinterpret_cast<const float4 *> (rec); c[t] = *reinterpret_cast<const float4 *> (rec + 16);
            }
#pragma unroll
            for (uint32_t t = 0; t < KS_DEPTH; ++t) {
                if (t >= nt) break;
                KS_CAND (g[t], c[t], tb + t);
            }
        }
        if (__builtin_expect (ntrips > ntr0, 0))     // ---- long lists: the rest chunk by chunk behind box tests (ks_list_tail)
            ks_list_tail<LPQ> (XQb, o, je, alast, ICP_S2_UNCOND / 16u, ntrips, qx, qy, qz, qr, qg, qb, alpha, dr, b, lane, ss, best2, bj);
        if (bj != 0xFFFFFFFFu) bj = min (o + ss + bj * KS_SPLIT, max (je, 1u) - 1u);     // trip -> list position
        if (je == o) { best2 = __builtin_inff (); bj = 0xFFFFFFFFu; }      // empty list / invalid query: nothing above was a candidate
    }
    // the query's winner among its lanes: smallest distance, ties -> lowest list position; that lane finishes
    // the query (lane ss == 0 when the list is empty or no candidate has a finite distance)
    dmin = ks_grp_min_f<KS_SPLIT> (best2);
    jmin = ks_grp_min_u<KS_SPLIT> (best2 == dmin ? bj : 0xFFFFFFFFu);
}

// ------------------------------------------------------------------------------------------
// ks_epilogue — the per-query hand-off to the finishing wave, the winner's record, weights (a5), per-query outputs, the block's
// moments (fused) or the first levels of the weight tree (reference order).  The LAST thing k_search does: every thread of the
// block calls it (two block barriers inside), and its returns end the kernel.
// ------------------------------------------------------------------------------------------
template <bool FUSED, bool CHAIN, int MINW, int LPQ, bool OWNER, bool PRUNE>
static __device__ __forceinline__ void ks_epilogue (const icp_params &p, float4 *s_qa, uint4 *s_qb, const uint32_t *s_slot, double (*s_mom)[64], float *s_w, const float4 *R4, const char *XQb,
                                                    uint32_t b, uint32_t m, uint32_t nr, uint32_t check_flags, uint32_t tid, uint32_t lane, uint32_t slice, uint32_t tile_id,
                                                    uint32_t qe, uint32_t ss, uint32_t i, bool valid, uint32_t o, uint32_t n, uint32_t rstar, float dr, float dmin, uint32_t jmin,
                                                    float qx, float qy, float qz)
{
    // Hand-off: lane 0 of every query leaves (q, distance, winner position, representative, flags) in LDS, and ONE wave
    // finishes all 64 queries of the block with every lane active (lane e = query e): the winner's record, the weight,
    // the per-query outputs and the 18 moment products are then issued once per block instead of once per wave for a
    // handful of active lanes (an instruction costs the same whatever its lane count).
    if (ss == 0u) {
        const bool empty = (n == 0u);
        s_qa[qe] = make_float4 (qx, qy, qz, empty ? dr : dmin);
        s_qb[qe] = make_uint4 (empty ? rstar : ((jmin == 0xFFFFFFFFu) ? o : jmin), rstar, (valid ? 1u : 0u) | (empty ? 2u : 0u), i);
    }
    __syncthreads ();
    if (slice == 0u) {
        const float4 qa = s_qa[lane]; const uint4 qb = s_qb[lane];
        // (dense variants: the queries were handed over with a frame's invalid ones last — s_slot: the place in the tile of the query at this position)
        const uint32_t eo = (PRUNE && ICP_S1_ORIGIN_LIST) ? s_slot[lane] : lane;
        const bool v = (qb.z & 1u) != 0u, empty = (qb.z & 2u) != 0u;
        // the search ran on geo + a pho (a positive common factor changes neither the argmin nor the ties, and the pruning
        // bound d >= geo stays as it is); the distance reported and fed to the weights carries the metric's absolute scale
        const float ex = qa.x, ey = qa.y, ez = qa.z, d = p.dist_scale * qa.w;
        const uint32_t ei = qb.w;
        float w = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
        if (v) {
            uint32_t id;
            if (empty) {             // empty list: fall back to the representative itself
                const float4 nn = R4[2 * (size_t) qb.x];
                // (dense variants: rep_src — needed for this rare case only — and the moments' base below are fetched where they are used: held
                // from the top of the kernel they were the scalar registers the compiler spilled in front of the prologue's loads)
                const uint32_t *rsrc = p.rep_src;
                if constexpr (MINW == 4) {
                    typedef uint32_t __attribute__ ((address_space (1))) *gu32;
                    rsrc = (const uint32_t *) (gu32) ks_params_from_kernarg ()->rep_src;
                }
                id = rsrc[(size_t) b * nr + qb.x]; f0 = nn.x; f1 = nn.y; f2 = nn.z;
            } else {                 // the winner's point, or, when every distance is inf / NaN, the first list element as
                                     // the serial scan would: one reload instead of tracking it per candidate
                const char *rec = XQb + (qb.x << 5);
                const float4 wg = *reinterpret_cast<const float4 *> (rec), wc = *reinterpret_cast<const float4 *> (rec + 16);
                f0 = wg.x; f1 = wg.z; f2 = wc.x; id = __float_as_uint (wc.z);
            }
            w = p.weighted ? 100.f / (100.f + d) : 1.f;                // icp_kernels.cl:232
            // per-query outputs: uniform bases + 32-bit byte offsets (i < 2^20)
            icp_dist_id di; di.dist = d; di.id = id;
            // One-block-per-CU variants: the four output pointers are fetched from the kernel arguments HERE — an opaque copy of the
            // argument pointer keeps the compiler from hoisting their scalar loads to the top of the kernel with all the others, where
            // eight more live SGPRs make it spill freshly loaded arguments to VGPR lanes, i.e. wait for the argument block in front of
            // the prologue's first vector loads (the chained kernel of a host-driven run: 9.29 -> 8.97 us per iteration); a run that
            // stores no per-query outputs on the way never loads them at all.
            const icp_params *pe = &p;
            if constexpr (MINW == 2 && !OWNER) {
                // (the kernel's explicit arguments: four pointers, six dwords, then icp_params — no padding in between; a change of the
                // signature has to move this offset with it: every test that reads per-query outputs at a latency-bound size would show it)
                            pe = ks_params_from_kernarg ();
            }
            typedef char __attribute__ ((address_space (1))) *gchar;           // (pointers read through `pe` are generic to the compiler: say that they are global)
            char *o_nn = (char *) (gchar) reinterpret_cast<char *> (pe->nn_id + (size_t) b * m), *o_pf = (char *) (gchar) reinterpret_cast<char *> (pe->PF + (size_t) b * m);
            char *o_pm = (char *) (gchar) reinterpret_cast<char *> (pe->PM + (size_t) b * m), *o_rid = (char *) (gchar) reinterpret_cast<char *> (pe->rid + (size_t) b * m);
            // (fused mode consumes none of these itself: inside a graph of a fixed length only the last iteration
            // stores them — except the nearest representative where the next search seeds its pruning with it)
            const bool emit = !FUSED || (check_flags & 8u);
            if (emit) {
                *reinterpret_cast<icp_dist_id *> (o_nn + (ei << 3)) = di;
                *reinterpret_cast<float4 *> (o_pf + (ei << 4)) = make_float4 (f0, f1, f2, w);
                *reinterpret_cast<float4 *> (o_pm + (ei << 4)) = make_float4 (ex, ey, ez, d);
            }
            if (emit || PRUNE) *reinterpret_cast<uint32_t *> (o_rid + (ei << 2)) = qb.y;
        }
        if constexpr (FUSED) {
            // the 18 moments of this pair in double (oracle orc_moments_fused); invalid queries contribute 0
            double W = (double) w;
            double g0 = v ? (double) f0 : 0.0, g1 = v ? (double) f1 : 0.0, g2 = v ? (double) f2 : 0.0;
            double q0 = (double) ex, q1 = (double) ey, q2 = (double) ez;
            if (!v) { W = 0.0; q0 = q1 = q2 = 0.0; }
            double wq0 = W * q0, wq1 = W * q1, wq2 = W * q2;
            s_mom[0][eo] = W;
            s_mom[1][eo] = W * g0; s_mom[2][eo] = W * g1; s_mom[3][eo] = W * g2;
            s_mom[4][eo] = wq0; s_mom[5][eo] = wq1; s_mom[6][eo] = wq2;
            s_mom[7][eo] = wq0 * g0; s_mom[8][eo] = wq0 * g1; s_mom[9][eo] = wq0 * g2;
            s_mom[10][eo] = wq1 * g0; s_mom[11][eo] = wq1 * g1; s_mom[12][eo] = wq1 * g2;
            s_mom[13][eo] = wq2 * g0; s_mom[14][eo] = wq2 * g1; s_mom[15][eo] = wq2 * g2;
            s_mom[16][eo] = W * ((g0 * g0 + g1 * g1) + g2 * g2);
            s_mom[17][eo] = W * ((q0 * q0 + q1 * q1) + q2 * q2);
        } else
            s_w[eo] = w;
    }
    KS_STAMP (6)
    __syncthreads ();
    if constexpr (FUSED) {
        // halving tree over the block's 64 pairs, one 16-lane row per moment (rows 0..17 of the 32 rows)
        if (slice * 4u >= (uint32_t) ICP_NMOM) return;                     // (waves without a row: done)
        const uint32_t l = lane & 15u, mrow = slice * 4u + (lane >> 4);     // first 18 of the block's 4*KS_SPLIT rows
        const uint32_t k = min (mrow, (uint32_t) ICP_NMOM - 1u);
        double c0 = s_mom[k][l] + s_mom[k][l + 32], c1 = s_mom[k][l + 16] + s_mom[k][l + 48];
        double v = row_tree_tail_d (c0 + c1);
        const uint32_t obuf = CHAIN ? (p.slot ^ 1u) : 0u;
        double *momw = p.mom;
        if constexpr (MINW == 4) {
            typedef double __attribute__ ((address_space (1))) *gf64;
            momw = (double *) (gf64) ks_params_from_kernarg ()->mom;
        }
        if (l == 0 && mrow < ICP_NMOM) momw[(((size_t) b * 2 + obuf) * ICP_NMOM + mrow) * p.nb + tile_id] = v;
    } else if (slice == 0 && p.weighted) {
        // tree levels d = 64 .. 2 restricted to this block's parity class (icp_kernels.cl:244-249):
        // element e of the class is position 2e + parity; levels pair e with e+32, e+16, .., e+1.
        const uint32_t l = lane & 15u;
        float a[4] = { s_w[l], s_w[l + 16], s_w[l + 32], s_w[l + 48] };
        float v = row_tree4 (a);
        if (lane == 0) p.wpart[(size_t) b * 2 * p.nwp + blockIdx.x] = v;
    }
    KS_STAMP (7)
}

// ------------------------------------------------------------------------------------------
// ks_owner_lists_tail — RBC construct of the latency-bound sizes (the owner search with per-block lists, see k_place_lists): what
// the block leaves behind stage 1.  Every thread of the block calls it; it ends the kernel.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ void ks_owner_lists_tail (const icp_params &p, uint4 *s_qb, uint32_t b, uint32_t m, uint32_t nb, uint32_t lane, uint32_t slice, uint32_t qe,
                                                            uint32_t ss, bool valid, uint32_t rstar)
{
    // the block's 64 owners meet in LDS; wave 0 (lane e = point 64 blockIdx.x + e) stores them in one coalesced row, ranks every
    // point among the earlier points of the block with the same owner (one ballot per distinct owner: neighbours share a
    // handful) and leaves the block's (owner, count) list: k_place_lists needs nothing else to place the points
    if (ss == 0u) s_qb[qe] = make_uint4 (rstar, valid ? 1u : 0u, 0u, 0u);
    __syncthreads ();
    if (slice != 0u) return;
    const uint4 e4 = s_qb[lane];
    const bool v = e4.y != 0u;
    const uint32_t own = v ? e4.x : 0xFFFFFFFFu, ip = blockIdx.x * 64u + lane;
    uint32_t rank = 0u, kk = 0u;
    uint2 *bl = p.blist + ((size_t) b * nb + blockIdx.x) * 64u;
    for (unsigned long long rem = __ballot (v); rem; ++kk) {
        const uint32_t o = (uint32_t) __builtin_amdgcn_readlane ((int) own, (int) __builtin_ctzll (rem));
        const unsigned long long same = __ballot (own == o);     // (an owner is < nr: never the marker of an invalid lane)
        if (own == o) rank = (uint32_t) __builtin_popcountll (same & ((1ull << lane) - 1ull));
        if (lane == 0) bl[kk] = make_uint2 (o, (uint32_t) __builtin_popcountll (same));
        rem &= ~same;
    }
    if (lane == 0) p.bn[(size_t) b * nb + blockIdx.x] = kk;
    if (v) { p.owner[(size_t) b * m + ip] = own; p.brank[(size_t) b * m + ip] = (uint8_t) rank; }
    return;
}

// ------------------------------------------------------------------------------------------
// stage 1 (nearest representative) — the two passes over one tile of representatives in LDS; their drivers (which tiles, in which order,
// behind which barriers: MASKED / the tile loop) are in k_search.
// ------------------------------------------------------------------------------------------
// coarse pass of the pruning over the groups of the tile in LDS (box[box0 ..]: the tile's (lo, hi) pairs): bit t of the result = this lane's
// t-th group (ss, ss + LPQ, ..) may hold a representative nearer than `lim`.  The lower bound applies the metric's own operations to the
// per-axis distances to the group's bounding box; every operation is monotone under round-to-nearest, so bound <= geo <= d for every
// member, and a group whose bound is not below `lim` cannot hold the winner.
template <int LPQ, int TILE>
static __device__ __forceinline__ uint32_t ks_coarse_pass (const float4 *s_box, uint32_t box0, float qx, float qy, float qz, uint32_t ss, uint32_t tn_, float lim_)
{
    const uint32_t ngt = (((tn_ + 1u) >> 1) + KS_SPLIT - 1u) / KS_SPLIT;
    uint32_t cm = 0u;
    auto test = [&] (uint32_t t, uint32_t g) {
        const float4 lo = s_box[box0 + 2 * g], hi = s_box[box0 + 2 * g + 1];
        const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
        const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
        const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
        const float bound = __builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex));
        if (bound < lim_) cm |= 1u << t;
    };
    if constexpr (TILE == 256) {                     // at most two trips per lane: not unrolled (the 64-register budget of this variant)
#pragma unroll 1
        for (uint32_t t = 0, g = ss; g < ngt; ++t, g += KS_SPLIT) test (t, g);
    } else
        for (uint32_t t = 0, g = ss; g < ngt; ++t, g += KS_SPLIT) test (t, g);
    return cm;
}

// fine pass of the pruning over the tile in LDS (s_pair[pair0 ..]; first representative t0_, npair_ pairs): the groups whose bit is set in
// cmask_ for some query of the wave, in ascending order (a lane's pairs must ascend for the tie rule), full evaluation.
// (scalar control flow: the lane ballot of trip t is folded over the wave's queries into one bit per group and only the set bits are
// visited — a taken branch costs more than the arithmetic it guards)
template <int LPQ>
static __device__ __forceinline__ void ks_fine_pass (const float4 *s_pair, uint32_t pair0, float qx, float qy, float qz, float qr, float qg, float qb, float alpha,
                                                     uint32_t gt_lg1, uint32_t ss, uint32_t t0_, uint32_t npair_, uint32_t cmask_, float &best, uint32_t &bid)
{
    const uint32_t ngt_ = (npair_ + KS_SPLIT - 1u) / KS_SPLIT;
    const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
    const float2v va = { alpha, alpha };
    for (uint32_t t = 0; t * KS_SPLIT < ngt_; ++t) {
        unsigned long long bal = __ballot ((cmask_ >> t) & 1u);
        if (bal == 0ull) continue;
        bal |= bal >> 32; bal |= bal >> 16;
        if (KS_SPLIT == 8) bal |= bal >> 8;
        uint32_t need = (uint32_t) bal & ((1u << KS_SPLIT) - 1u);
        while (need) {
            const uint32_t sg = (uint32_t) __builtin_ctz (need);
            need &= need - 1u;
            // pair of this lane in group gl of the LDS tile.  Strips: the group's 16 consecutive representatives.
            // Tiles (LPQ == 8): lane ss holds row ss >> 1, columns 2 (ss & 1) and + 1 of the 4 x 4 tile; groups are
            // visited in ascending (tile row, tile column) order, so every lane's pairs still ascend in index —
            // what the tie rule (strict '<' keeps a lane's lowest index) relies on.
            const uint32_t gl = sg + KS_SPLIT * t;
            const uint32_t P = (KS_SPLIT == 8 && gt_lg1) ? (((4u * (gl >> (gt_lg1 - 1u)) + (ss >> 1)) << gt_lg1) + 2u * (gl & ((1u << (gt_lg1 - 1u)) - 1u)) + (ss & 1u))
                                                          : gl * KS_SPLIT + ss;
            if (P < npair_) {
                const uint32_t P3 = pair0 + __umul24 (P, 3u);        // (24-bit multiply: full rate; a 32-bit v_mul_lo costs four issue slots)
                float4 A = s_pair[P3], B = s_pair[P3 + 1], C = s_pair[P3 + 2];
                float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                float2v d = __builtin_elementwise_fma (va, pho, geo);
                const uint32_t r0 = t0_ + 2u * P;
                if (d.x < best) { best = d.x; bid = r0; }
                if (d.y < best) { best = d.y; bid = r0 + 1u; }
            }
        }
    }
}

// CHAIN (fused mode only): launch j reads state slot j&1 and the moments buffer j&1, turns the previous
// iteration's moments into T in its prologue (every block redundantly; block 0 publishes the result in the
// other slot), searches, and leaves its own moments in the other buffer: ONE launch per ICP iteration.
// MINW = waves per SIMD the register allocation must leave room for: 2 (one block per CU: a single registration,
// nothing to hide latency behind, no spills) or 4 (two blocks per CU: batched registrations, +70 % throughput).
// LPQ = lanes per query = waves per block: 16 when the grid is at most one block per CU (more waves per SIMD to
// overlap the L2-cold loads), 8 when occupancy comes from the number of blocks.
// OWNER: the kernel is RBC construct step 1 instead (owner(x) = nearest representative of the FIXED point x: gM = F,
// no transform, stage 1 only, result to p.owner) — the same stage-1 code, pruning included, seeded with the
// representative of the point's own grid cell.
// TILE = representatives per LDS tile: 1024, or 256 for the dense variant at |R| <= 256 (batches of config 4): 22 KB instead of
// 47 KB of LDS per block and a register budget for 8 waves per SIMD — four blocks per CU instead of three.
// SINGLE: the launcher guarantees |R| <= TILE (the tile loop and everything multi-tile fold away).
// HOSTRUN (CHAIN only): the launch belongs to a host-driven checked run (run_ctl in icp_capi.hip) — progress words, the final state of a
// converged registration to the user-visible state and to host memory, the transform each search used (p.st_prev).  Fixed-length graphs
// (the metric's path) instantiate the kernel without any of it: what the publishing lane of block 0 carries in its prologue is on the
// path the whole grid waits for (measured with everything decided at run time: 8.58 -> 8.73 us per dispatch).
template <bool FUSED, bool CHAIN, int MINW, int LPQ, bool OWNER = false, int ROT = 1, int TILE = 1024, bool SINGLE = false, bool S2W = false, bool HOSTRUN = false>
__global__ __launch_bounds__ (64 * LPQ, TILE == 256 ? 8 : MINW) void k_search (const float *gM, const float *gR, icp_reg_state *gst, const double *gmom,
                                                              uint32_t m, uint32_t nr, uint32_t side, uint32_t tpr_magic,
                                                              uint32_t nb, uint32_t check_flags, icp_params p)
{
    // The first 14 dwords of the kernel arguments (everything the prologue's addresses need) are plain scalars so
    // that they arrive preloaded in SGPRs / in one scalar load; the rest of icp_params is fetched while the first
    // global loads are in flight.  gst = the state this launch reads (CHAIN: slot p.slot of every pair of slots),
    // gmom = the moments it turns into T first (CHAIN only: buffer p.slot).
    // check_flags: bit 0 = convergence checks on; bit 1 (CHAIN) = first launch of a chain: gst is the user-visible
    // state array (stride 1) instead of a pair of slots; bit 3 = store the matched / transformed points too (fused
    // mode needs them only after the last iteration of a graph; the reference-order kernels read them every time).
    // ==================================================== PROLOGUE ====================================================
    constexpr uint32_t KT = (uint32_t) TILE;
    // MASKED (dense variant, several small tiles): the set of tiles a block needs is decided ONCE, before anything is staged —
    // every query tests the boxes of all tiles against its seed bound, the block ORs the answers — and only those tiles
    // are staged and scanned, in ascending order, without a vote per tile.
    constexpr bool MASKED = (TILE == 256) && !SINGLE && (MINW == 4);
    if constexpr (SINGLE) __builtin_assume (nr <= KT);
    if constexpr (MASKED) __builtin_assume (nr > KT && nr <= 32u * KT);
    const uint32_t b = blockIdx.y, check = check_flags & 1u;
    icp_reg_state *st = (CHAIN && !(check_flags & 2u)) ? gst + (size_t) b * 2 : gst + b;
#ifdef ICP_DBG_STAMPS
    { const uint32_t tid = threadIdx.x; unsigned long long t_; asm volatile ("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
      KS_STAMP_STORE (8) }
#endif

    // representatives of the current tile, pair-interleaved for packed fp32 math:
    //   pair P = reps (2P, 2P+1) -> 3 float4: [x0 x1 y0 y1] [z0 z1 r0 r1] [g0 g1 b0 b1]
    // MASKED keeps TWO tile buffers: buffer 1 holds the block's HOME tile — the tile of the representative sampled from the grid cell
    // of the block's first query —, staged in the prologue with everything else that does not depend on T; buffer 0 is for the other
    // tiles the block turns out to need.  A block's 64 neighbouring queries mostly need the home tile alone: then nothing is fetched
    // between the tile masks and the scan, the seed representative and the winner's (offset, size) come from LDS too — three dependent
    // memory round trips fewer on a path that is a chain of them (|F| = 65536: one wave of 1024 blocks, latency-bound; stamps in
    // profiles/r03_stamps_dense.txt).
    constexpr uint32_t NBUF = MASKED ? 2u : 1u, PB = 3u * KT / 2u, BB = 2u * (KT / 16u);
    __shared__ float4 s_pair[NBUF * PB];
    __shared__ uint2 s_on[(MASKED && OWNER) ? 1 : KT];            // (offset, size) of the lists of the tile's representatives (MASKED: of the home tile's)
    __shared__ uint32_t s_tmask;                     // MASKED: tiles some query of the block needs
    __shared__ uint32_t s_ovote;                     // some query of the block is near the origin (the list of the representatives there is staged for it)
    __shared__ float4 s_box[NBUF * BB];              // (lo, hi) of the tile's groups of 2 * LPQ representatives
    __shared__ float4 s_tbox[(MINW == 4 && !SINGLE) ? 2 * 32 : 2];      // (lo, hi) of every tile (multi-tile sets: |R| <= 32768)
    __shared__ float s_w[64];
    __shared__ float4 s_qc[64];                      // query hand-in: (r, g, b, pruning seed); s_qa carries (q', index)
    __shared__ float4 s_qa[64];                      // per-query hand-off to the finishing wave: (q', distance)
    __shared__ uint4 s_qb[64];                       //   (winner position or representative, representative, flags, query index)
    // dense variants: the query wave hands a frame's invalid queries over BEHIND the valid ones (see there); the finishing wave puts a query's
    // moment products back at the slot of its place in the block's tile: the sums keep their order
    __shared__ uint32_t s_slot[(ICP_S1_SEED && ICP_S1_ORIGIN_LIST && MINW == 4) ? 64 : 1];      // the tile slot of the query handed over at position p
    __shared__ double s_mom[FUSED ? ICP_NMOM : 1][64];
    __shared__ icp_fin_result s_fin;
    __shared__ double s_l1[CHAIN ? ICP_NMOM : 1][CHAIN ? 32 : 1];
    __shared__ double s_t[ICP_NMOM];

    // A wave serves KS_QPW queries end to end, KS_SPLIT (= LPQ) lanes per query: lane ss of a query takes the
    // representative pairs = ss mod LPQ in stage 1 and the list positions = ss mod LPQ in stage 2, and the
    // query's winner is an LPQ-lane DPP reduction — no cross-wave exchange inside the two stages (block barriers only around the LDS hand-overs).
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t slice = __builtin_amdgcn_readfirstlane (tid >> 6);
    const uint32_t qe = slice * KS_QPW + lane / KS_SPLIT, ss = lane & (KS_SPLIT - 1u);
    // reference-order mode: the 64 even (or odd) positions of one 128-query group (a closed sub-tree of
    // the weight reduction); fused mode: an 8 x 8 tile of the landmark grid (spatially coherent lists)
    // One wave (the last; in the chained variant wave 0, which holds the new T in registers the moment it exists and hands the
    // transformed queries over before the barrier that ends the finalize) prepares the block's 64 queries —
    // lane e = query e: index, load, transform, pruning seed — and hands them to the lanes of each query through LDS;
    // the other 15 (7) waves neither compute the index nor load / transform the same point LPQ times over.
    const bool qwave = CHAIN ? slice == 0u : slice == KS_SPLIT - 1u;
    // (the one-block-per-CU variants only: measured 9.63 -> 9.43 us per iteration at A; the dense variant runs several blocks
    // per CU over grids of thousands and measured 0 ... 4 % slower with it)
    // (dense variant: bit 6 of check_flags, set for a single large registration — there the bands halve the fabric-side traffic,
    // C 171 -> 86 MB per launch against 75.6 MB algorithmic, at the same speed; batched grids measured 2 % slower with them)
    // OWNER_LISTS (the owner search of the latency-bound sizes, buildRBC in two launches: see k_place_lists): blocks of 64 CONSECUTIVE
    // fixed points — the stable placement ranks a point among the earlier points of its owner, and a block's (owner, count) list
    // is a piece of exactly that count —, the representatives gathered straight from F (getReps' sampling rule: the launch does
    // not wait for a kernel that writes R; block 0 writes R and rep_src on the side).
    constexpr bool OWNER_LISTS = OWNER && MINW == 2;
    const uint32_t tile_id = OWNER_LISTS ? blockIdx.x : (FUSED && (MINW == 2 || (check_flags & 64u))) ? ks_tile_of_block (blockIdx.x, nb) : blockIdx.x;     // (fused grids are (nb, batch); gridDim is a hidden kernel argument: a scalar load + wait in front of the first vector load)
    const uint32_t iq = OWNER_LISTS ? blockIdx.x * 64u + lane :
                        FUSED ? fused_query_index (m, side, tpr_magic, tile_id, lane)
                              : (blockIdx.x >> 1) * 128u + 2u * lane + (blockIdx.x & 1u);

    const float4 *M4 = reinterpret_cast<const float4 *> (gM + (size_t) b * m * 8);
    const float4 *R4 = reinterpret_cast<const float4 *> (gR + (size_t) b * nr * 8);

    // every independent global load of the prologue is issued before anything waits: the state (one vector load,
    // lane j = dword j: scalar loads of T would queue behind the waits of the vector loads), the first tile of
    // representatives (+ list offsets / sizes), the query point (clamped address, selected afterwards)
    uint32_t sv = OWNER ? 0u : state_load_lanes (st);
    // HOSTRUN: the run this launch belongs to may have converged already (its flag holds the run's epoch): such a launch leaves below
    // without a single store — the next tracked frame may be running on the other stream, in the same state slots and moment buffers
    uint32_t run_over = 0u;
    if constexpr (CHAIN) {
        // bit 4 of check_flags (first launch of a chain): the run starts from the identity transform — what k_reset_state
        // would have left in the state (T = Tk = (0,0,0,1 | 0,0,0,1), R = Rk = I, S = means = sum_w = 0, k = done = 0),
        // without a launch of its own
        static_assert (ICP_ST_DW (T) == 0 && ICP_ST_DW (Tk) == 8 && ICP_ST_DW (R) == 16 && ICP_ST_DW (Rk) == 25 && ICP_ST_DW (reserved0) == 60, "state layout");
        if (check_flags & 16u) {
            constexpr unsigned long long ones = (1ull << 3) | (1ull << 7) | (1ull << 11) | (1ull << 15) | (1ull << 16) | (1ull << 20) | (1ull << 24) |
                                                (1ull << 25) | (1ull << 29) | (1ull << 33);
            if (lane < ICP_ST_DW (reserved0)) sv = ((ones >> lane) & 1ull) ? 0x3F800000u : 0u;
        }
    }
    // chained variant: the previous iteration's block moments (first tree level of this block's finalize) travel with
    // the other prologue loads
    double ma0[8];
    if constexpr (CHAIN) fused_moment_loads<64 * LPQ> (gmom + (size_t) b * 2 * ICP_NMOM * nb, nb, 0u, ma0);
    float *s_pairf = reinterpret_cast<float *> (s_pair);
    const uint32_t tn0 = min (KT, nr);
    float4 rg[2], rc[2]; uint2 ron[2];
    uint32_t ht = 0u, tnH = 0u;                      // MASKED: home tile of the block and its size (block-uniform)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        if (MASKED && u == 0) continue;
        rg[u] = make_float4 (0.f, 0.f, 0.f, 0.f); rc[u] = rg[u];
        if constexpr (OWNER_LISTS) {
            if (k < tn0) {
                const uint32_t src = rep_src_index (p, k);
                rg[u] = M4[2 * (size_t) src]; rc[u] = M4[2 * (size_t) src + 1];
                if (blockIdx.x == 0) {               // getReps (a1): R and rep_src, written once, read by the searches that follow
                    float4 *Rw = reinterpret_cast<float4 *> (p.R + (size_t) b * nr * 8);
                    Rw[2 * (size_t) k] = rg[u]; Rw[2 * (size_t) k + 1] = rc[u];
                    p.rep_src[(size_t) b * nr + k] = src;
                }
            }
        } else if (!MASKED && k < tn0) { rg[u] = R4[2 * (size_t) k]; rc[u] = R4[2 * (size_t) k + 1]; }
    }
    const uint32_t ic = min (iq, m - 1u);
    typedef float ks_f4 __attribute__ ((ext_vector_type (4)));
    ks_f4 mgv = { 0.f, 0.f, 0.f, 1.f }, mcv = mgv;    // (whole 128-bit values until the opaque use below: see there)
    if (qwave) { mgv = *reinterpret_cast<const ks_f4 *> (M4 + 2 * (size_t) ic); mcv = *reinterpret_cast<const ks_f4 *> (M4 + 2 * (size_t) ic + 1); }
    if constexpr (MASKED) {                          // the home tile: of the cell of the block's first query (block-uniform; no division)
        const uint32_t i0 = (uint32_t) __builtin_amdgcn_readfirstlane ((int) min (FUSED ? fused_query_index (m, side, tpr_magic, tile_id, 0u) : (blockIdx.x >> 1) * 128u + (blockIdx.x & 1u), m - 1u));
        const uint32_t cell = p.side_magic ? cell_rep_of (p, i0) : 0u;
        ht = min (cell, nr - 1u) / KT; tnH = min (KT, nr - ht * KT);
        if (ICP_HOME_MODE != 3 && tid < tnH) { rg[0] = R4[2 * (size_t) (ht * KT + tid)]; rc[0] = R4[2 * (size_t) (ht * KT + tid) + 1]; }
    }
    // (list offsets / sizes: their base pointers come with the second batch of kernel arguments)
    const uint32_t *gO = p.O + (size_t) b * nr, *gN = p.N + (size_t) b * nr;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        ron[u] = make_uint2 (0u, 0u);
        if (!OWNER && !MASKED && k < tn0) ron[u] = make_uint2 (gO[k], gN[k]);
        if (!OWNER && MASKED && u == 0 && tid < tnH) ron[0] = make_uint2 (gO[ht * KT + tid], gN[ht * KT + tid]);
    }
    // (HOSTRUN: the flag is read HERE, behind the prologue's vector loads — a scalar load of the flag's address, a second one of the flag and a
    // wait for both: in front of them it would hold every load of the prologue back by two scalar round trips)
    if constexpr (CHAIN && HOSTRUN) { if (p.run_flag) run_over = (p.run_flag[b] == p.epoch) ? 1u : 0u; }
    // seed of the stage-1 pruning bound: this query's nearest representative of the previous search (any index < nr
    // is a valid seed; the buffer starts zeroed)
    // Pruning pays where stage 1 is throughput-bound: the dense variant (MINW == 4: several blocks per CU, or a
    // large representative set — see icp_launch_search); a single small registration is latency-bound and keeps
    // the branch-free loop (compile-time: the pruning code costs 0.25 us there even when it is switched off).
    constexpr bool PRUNE = ICP_S1_SEED && MINW == 4;
    const bool prune = PRUNE && p.a > 0.f;
    const uint32_t gt_lg1 = p.gtile;                 // 0: strip groups; 1 + log2 (nrx / 4): 4 x 4 tile groups (k_rep_boxes)
    // A registration's FIRST search (k == 0: ICP::buildRBC / a reset came before it) has no previous search of its own: whatever
    // p.rid holds then belongs to another registration (legal, but a converged neighbour's answer would flatter a benchmark
    // that re-registers one pair, and a stale one prunes nothing).  It is seeded like the owner search: with the representative
    // sampled from the query's own grid cell — a moving frame starts near the fixed one (frame-to-frame registration).
    // check_flags bit 5 (ICP_AMD_WARM_SEED=1, diagnostics): always the previous search's answer.
    uint32_t seed = 0u, seed_cell = 0xFFFFFFFFu;
    if (qwave && prune && p.side_magic) seed_cell = cell_rep_of (p, ic);     // the representative sampled from the point's own cell
    if constexpr (OWNER) seed = (seed_cell == 0xFFFFFFFFu ? 0u : seed_cell) | ((PRUNE && ICP_S1_ORIGIN_LIST) ? 0x40000000u : 0u);      // (bit 30: a seed from the grid cell — ks_seed_against_invalid)
    else if (qwave && prune) {
        seed = p.rid[(size_t) b * m + ic];           // (selected against seed_cell below, once the state has arrived)
        if ((check_flags & 32u) || seed_cell == 0xFFFFFFFFu) seed_cell = 0xFFFFFFFFu;
    }
    // (lo, hi) boxes of the groups of 2 * LPQ representatives: the 16-boxes (LPQ = 8) or the 32-boxes behind them
    const float4 *GBt = p.GB + (size_t) b * 2 * (p.n16 + p.n1k) + (KS_SPLIT == 8 ? 0u : 2u * p.n16);
    const uint32_t nbox0 = 2u * ((tn0 + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
    float4 boxv = make_float4 (0.f, 0.f, 0.f, 0.f);
    if (!MASKED && prune && tid < nbox0) boxv = GBt[tid];
    const uint32_t nboxH = 2u * ((tnH + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
    if (MASKED && ICP_HOME_MODE != 3 && prune && tid < nboxH) boxv = GBt[2u * (ht * KT / (2u * KS_SPLIT)) + tid];
    float T[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = state_lane_f (sv, ICP_ST_DW (T) + k);
    if constexpr (PRUNE && !OWNER) {
        if (seed_cell != 0xFFFFFFFFu && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k)) == 0) seed = seed_cell | (ICP_S1_ORIGIN_LIST ? 0x40000000u : 0u);    // first search of a registration (bit 30: a seed from the grid cell)
    }
    icp_reg_state *sout = CHAIN ? p.cst + (size_t) b * 2 + (p.slot ^ 1u) : st;
    if constexpr (!OWNER && !CHAIN) {
        // host-driven checked runs (icp_run; see run_ctl in icp_capi.hip), separate launches: the search of iteration j tells the host that j
        // iterations are through and whether the last one converged — one 8-byte store into host memory that nothing here waits for
        // (fused mode: the finalize that found a registration converged has published DONE | FINAL with the final state itself — a search
        // behind it must not overwrite that word; reference order: the search is the only one that tells)
        const uint32_t done_ = (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done));
        if (p.hmirror && blockIdx.x == 0 && tid == 0 && !(FUSED && done_))
            icp_mirror_store (p.hmirror + b, ICP_MIRROR_WORD (p.epoch, (uint32_t) __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (k)), done_));
    }
    if constexpr (CHAIN && HOSTRUN) { if (run_over) return; }
    if (!OWNER && check && __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (done))) {    // converged earlier
        if constexpr (CHAIN) {                       // carry the state forward
            if (blockIdx.x == 0 && tid < sizeof (icp_reg_state) / 4) reinterpret_cast<uint32_t *> (sout)[tid] = sv;
        }
        return;
    }
    // The query point is used as (x, y), (y, z) pairs (packed math): left alone, the compiler loads those pairs with overlapping narrow
    // loads and assembles them with moves INSIDE the conditional block of the load — i.e. the query wave waits there for every load it has
    // issued and sends its share of the list headers a memory round trip late, in front of a barrier the whole block stands at.  An opaque
    // use HERE (everything is issued and waited for by now) keeps the loads whole and the block free of anything that touches their result.
    if constexpr (!S2W) asm volatile ("" : "+v"(mgv), "+v"(mcv));     // (lanes = candidates, long lists: throughput-bound, measured 0.3 % slower with it)
    float4 mg = make_float4 (mgv.x, mgv.y, mgv.z, mgv.w), mc = make_float4 (mcv.x, mcv.y, mcv.z, mcv.w);
    if (iq >= m) { mg = make_float4 (0.f, 0.f, 0.f, 1.f); mc = mg; }
    const float4 *XQ4 = reinterpret_cast<const float4 *> (p.XQ + (size_t) b * m * 8);
    const char *XQb = reinterpret_cast<const char *> (XQ4);
    // the first tile of representatives goes to LDS now: nothing in it depends on T, and in the chained variant the
    // writes and their barrier disappear behind the power method of wave 0
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t k = tid + (uint32_t) u * 64u * KS_SPLIT;
        if (!MASKED && k < tn0) {
            float *dst = s_pairf + (k >> 1) * 12u + (k & 1u);
            dst[0] = rg[u].x; dst[2] = rg[u].y; dst[4] = rg[u].z; dst[6] = rc[u].x; dst[8] = rc[u].y; dst[10] = rc[u].z;
            if constexpr (!OWNER && !MASKED) s_on[k] = ron[u];
        }
    }
    if (!MASKED && prune && tid < nbox0) s_box[tid] = boxv;
    auto home_to_lds = [&] () {                      // the home tile -> buffer 1
        if (tid < tnH) {
            float *dst = s_pairf + PB * 4u + (tid >> 1) * 12u + (tid & 1u);
            dst[0] = rg[0].x; dst[2] = rg[0].y; dst[4] = rg[0].z; dst[6] = rc[0].x; dst[8] = rc[0].y; dst[10] = rc[0].z;
        }
        if (prune && tid < nboxH) s_box[BB + tid] = boxv;
    };
    if constexpr (MASKED) {
        if (ICP_HOME_MODE == 1) home_to_lds ();
        if constexpr (!OWNER) { if (tid < tnH) s_on[tid] = ron[0]; }
    }
    if (MASKED && tid == 0) s_tmask = 0u;
    if (PRUNE && ICP_S1_ORIGIN_LIST && tid == 0) s_ovote = 0u;
    if constexpr (MINW == 4 && !SINGLE) {            // the boxes of all tiles: a tile is tested before it is staged (stage 1 below)
        if (prune && nr > KT && tid < 2u * p.n1k) s_tbox[tid] = p.GB[(size_t) b * 2 * (p.n16 + p.n1k) + 2u * p.n16 + tid];
    }
    bool handed = false;                             // the queries are in LDS already (chained variant, see below)
    if constexpr (CHAIN) {
        const bool pending = __builtin_amdgcn_readlane ((int) sv, (int) ICP_ST_DW (pending)) != 0;
        if (pending) {
            // wave 0 (= the query wave): T -> the block's 64 transformed queries -> LDS, inside the finalize, before its barrier
            auto hand_over = [&] (const float *Tn) {
                float tx, ty, tz;
                icp_transform_point (Tn, mg.x, mg.y, mg.z, tx, ty, tz);
                s_qa[lane] = make_float4 (tx, ty, tz, __uint_as_float (iq));
                s_qc[lane] = make_float4 (mc.x, mc.y, mc.z, __uint_as_float (seed));
            };
            fused_finalize_block<32, 64 * LPQ, ROT, true> (p, gmom + (size_t) b * 2 * ICP_NMOM * nb, nb, 0u, sv, ma0, &s_fin, s_l1, s_t, nullptr,
                                                           blockIdx.x == 0 ? sout : nullptr, hand_over, 1u,
                                                           (HOSTRUN && blockIdx.x == 0 && p.hmirror) ? p.hmirror + b : nullptr, HOSTRUN, HOSTRUN ? p.st + b : nullptr,
                                                           (HOSTRUN && p.hstate) ? p.hstate + b : nullptr, (HOSTRUN && p.st_prev) ? p.st_prev + b : nullptr);
            handed = true;
            KS_STAMP (9)
            if (s_fin.done) return;
        } else if (blockIdx.x == 0 && tid < sizeof (icp_reg_state) / 4) {
            reinterpret_cast<uint32_t *> (sout)[tid] = (tid == offsetof (icp_reg_state, pending) / 4) ? 1u : sv;
        }
    }
    if (qwave && !handed) {
        float tx = mg.x, ty = mg.y, tz = mg.z;
        if constexpr (!OWNER) icp_transform_point (T, mg.x, mg.y, mg.z, tx, ty, tz);
        if constexpr (PRUNE && ICP_S1_ORIGIN_LIST) {
            // Dense variants.  A query that is an invalid point of its frame (at the origin before the transformation) is flagged — bit 31 of
            // the seed: ks_seed_against_invalid — and handed over BEHIND the block's valid queries (a stable partition of the 64: the valid
            // ones stay neighbours).  Scattered over the block's waves such queries made nearly every wave walk the list of the
            // representatives at the origin behind its tiles, and the block waited for each of them at the hand-over to the finishing wave;
            // together they cost one wave that walk.  s_slot keeps the slot of a query's place in the tile: the finishing wave puts the
            // query's moment products there, so the block's sums run in the order of the tile as before.
            const bool hole = mg.x == 0.f && mg.y == 0.f && mg.z == 0.f;
            const float4 ha = make_float4 (tx, ty, tz, __uint_as_float (iq)), hc = make_float4 (mc.x, mc.y, mc.z, __uint_as_float (hole ? (seed | 0x80000000u) : seed));
            // (first in tile order, addresses that do not wait for the point; a block with invalid queries writes all 64 once more, permuted —
            // one wave's LDS writes land in order.  With the position itself in the address the whole block waited ~0.15 us longer for this wave)
            s_qa[lane] = ha; s_qc[lane] = hc; s_slot[lane] = lane;
            const unsigned long long hb = __ballot (hole);
            if (__builtin_expect (hb != 0ull, 0)) {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const uint32_t pos = hole ? 64u - (uint32_t) __builtin_popcountll (hb) + (uint32_t) __builtin_popcountll (hb & lt) : (uint32_t) __builtin_popcountll (~hb & lt);
                s_qa[pos] = ha; s_qc[pos] = hc; s_slot[pos] = lane;
            }
        } else {
            s_qa[lane] = make_float4 (tx, ty, tz, __uint_as_float (iq));
            s_qc[lane] = make_float4 (mc.x, mc.y, mc.z, __uint_as_float (seed));
        }
    }
    float qx = 0.f, qy = 0.f, qz = 0.f, qr = 0.f, qg = 0.f, qb = 0.f;
    uint32_t i = 0u; bool valid = false;
    const float alpha = p.a;
    KS_STAMP (0)

    // ==================================================== STAGE 1 =====================================================
    // ---- stage 1: nearest representative, two representatives per packed instruction ----
    float best = __builtin_inff (), s1_lim = __builtin_inff (); uint32_t bid = 0xFFFFFFFFu;
    // (the number of representatives at the origin — a spare lane of box 0 in LDS: k_reps_and_boxes — is read where the query is read, so that
    // the wait is the query's; read behind stage 1 its LDS round trip stood alone at the end of every wave's stage 1)
    uint32_t n_origin = 0u;
    if constexpr (MASKED) {
        __syncthreads ();                            // the queries (s_qa / s_qc), the tile boxes, s_tmask = 0
        {
            const float4 a4 = s_qa[qe], c4 = s_qc[qe];
            if constexpr (PRUNE && ICP_S1_ORIGIN_LIST) { if (prune) n_origin = (uint32_t) __builtin_amdgcn_readfirstlane ((int) __float_as_uint (s_tbox[1].w)); }
            qx = a4.x; qy = a4.y; qz = a4.z; i = __float_as_uint (a4.w); valid = i < m;
            qr = c4.x; qg = c4.y; qb = c4.z; seed = __float_as_uint (c4.w);
        }
        const uint32_t ntile = (nr + KT - 1u) / KT;  // <= 32
        uint32_t qmask = 0xFFFFFFFFu >> (32u - ntile);
        if (prune) {
            // the seed bound: the seed representative from the home tile in LDS, from global memory where it lies outside
            const uint32_t sfl = seed >> 30;             // (bit 1: an invalid point of its frame, bit 0: a seed from the grid cell — ks_seed_against_invalid)
            seed = min (seed & 0x3FFFFFFFu, nr - 1u);
            float sx, sy, sz, sr, sg, sb;
            if (ICP_HOME_MODE == 1 && seed / KT == ht) {
                const float *sp = s_pairf + PB * 4u + ((seed - ht * KT) >> 1) * 12u + (seed & 1u);
                sx = sp[0]; sy = sp[2]; sz = sp[4]; sr = sp[6]; sg = sp[8]; sb = sp[10];
                // (keeps the compiler from merging this with the global path below into flat loads)
                asm volatile ("" : "+v"(sx), "+v"(sy), "+v"(sz), "+v"(sr), "+v"(sg), "+v"(sb));
            } else {
                const float4 g = R4[2 * (size_t) seed], c = R4[2 * (size_t) seed + 1];
                sx = g.x; sy = g.y; sz = g.z; sr = c.x; sg = c.y; sb = c.z;
            }
            if constexpr (ICP_S1_ORIGIN_LIST) ks_seed_against_invalid (sfl, s_tbox + 1, nr, b, R4, seed, sx, sy, sz, sr, sg, sb);
            const float b0 = icp_metric8 (qx, qy, qz, qr, qg, qb, sx, sy, sz, sr, sg, sb, alpha);
            if (b0 >= 0.f && b0 < __builtin_inff ()) s1_lim = __uint_as_float (__float_as_uint (b0) + 1u);     // next float up
            // tiles this query can find a nearer representative in: lane ss tests the tiles ss, ss + LPQ, ..; OR over the lanes
            uint32_t tm = 0u;
            // (not unrolled: at most 32 tiles over the query's 8 lanes; unrolled twice the loop held the query's colour in registers the
            // 64-register variants then spilled — or read again from LDS behind it: |F| = 65536 18.19 -> 18.09 us, 2^20 274.7 -> 273.2)
#pragma unroll 1
            for (uint32_t t = ss; t < ntile; t += KS_SPLIT) {
                const float4 lo = s_tbox[2u * t], hi = s_tbox[2u * t + 1u];
                const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
                const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
                const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
                if (__builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex)) < s1_lim) tm |= 1u << t;
            }
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
            tm |= (uint32_t) __builtin_amdgcn_update_dpp (0, (int) tm, 0x141, 0xF, 0xF, true);     // row_half_mirror
            qmask = tm;
        }
        {   // the block's union: OR over the wave (8 queries: one per half row), one LDS atomic per wave
            uint32_t wm = qmask | (uint32_t) __builtin_amdgcn_update_dpp (0, (int) qmask, 0x140, 0xF, 0xF, true);      // row_mirror: both half rows
            wm = (uint32_t) __builtin_amdgcn_readlane ((int) wm, 0) | (uint32_t) __builtin_amdgcn_readlane ((int) wm, 16) |
                 (uint32_t) __builtin_amdgcn_readlane ((int) wm, 32) | (uint32_t) __builtin_amdgcn_readlane ((int) wm, 48);
            if (lane == 0) atomicOr (&s_tmask, wm);
        }
        if (ICP_HOME_MODE == 2) home_to_lds ();
        __syncthreads ();
        KS_STAMP (10)
        uint32_t bm = s_tmask;                       // block-uniform
        bool used0 = false;                          // buffer 0 holds a tile some wave may still be scanning
        while (bm) {                                 // ascending: the tie rule needs every lane's representatives to ascend
            const uint32_t tl = (uint32_t) __builtin_ctz (bm);
            bm &= bm - 1u;
            const uint32_t t0 = tl * KT, tn = min (KT, nr - t0), npair = (tn + 1u) >> 1;
            const uint32_t hb = (ICP_HOME_MODE != 3 && tl == ht) ? 1u : 0u;    // the home tile is there already: no fetch, no barrier
            if (!hb) {
                if (used0) __syncthreads ();         // every wave is done with the previous tile of buffer 0
                used0 = true;
                if (prune) {
                    const uint32_t nbx = 2u * ((tn + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
                    if (tid < nbx) s_box[tid] = GBt[2u * (t0 / (2u * KS_SPLIT)) + tid];
                }
                if (tid < tn) {
                    const float4 g = R4[2 * (size_t) (t0 + tid)], c = R4[2 * (size_t) (t0 + tid) + 1];
                    float *dst = s_pairf + (tid >> 1) * 12u + (tid & 1u);
                    dst[0] = g.x; dst[2] = g.y; dst[4] = g.z; dst[6] = c.x; dst[8] = c.y; dst[10] = c.z;
                }
                __syncthreads ();
            }
            if (prune) {
                const bool mine = ((qmask >> tl) & 1u) != 0u;
                uint32_t cmask = 0u;
                if (__ballot (mine)) cmask = ks_coarse_pass<LPQ, TILE> (s_box, hb * BB, qx, qy, qz, ss, tn, mine ? s1_lim : -__builtin_inff ());
                ks_fine_pass<LPQ> (s_pair, hb * PB, qx, qy, qz, qr, qg, qb, alpha, gt_lg1, ss, t0, npair, cmask, best, bid);
                s1_lim = fminf (s1_lim, ks_grp_min_f<KS_SPLIT> (best));
            } else {
                const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
                const float2v va = { alpha, alpha };
                for (uint32_t P = ss; P < npair; P += KS_SPLIT) {
                    float4 A = s_pair[hb * PB + 3 * P], B = s_pair[hb * PB + 3 * P + 1], C = s_pair[hb * PB + 3 * P + 2];
                    float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                    float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                    float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                    float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                    float2v d = __builtin_elementwise_fma (va, pho, geo);
                    const uint32_t r0 = t0 + 2u * P;
                    if (d.x < best) { best = d.x; bid = r0; }
                    if (d.y < best) { best = d.y; bid = r0 + 1u; }
                }
            }
        }
    } else
    for (uint32_t t0 = 0; t0 < nr; t0 += KT) {
        const uint32_t tn = min (KT, nr - t0);
        uint32_t cmask = 0u;
        // several tiles: the box of the whole tile first (same bound as for a group) — a query far from the tile skips its
        // 64 group tests, and a tile no query of the block is near is neither tested further nor staged (at |R| = 4096 a
        // block's 64 neighbouring queries need one, seldom two, of the four tiles)
        auto tile_near = [&] (float lim_) -> bool {
            if (SINGLE || nr <= KT) return true;
            const float4 lo = s_tbox[2u * (t0 / ICP_TBOX)], hi = s_tbox[2u * (t0 / ICP_TBOX) + 1u];     // (staged in the prologue)
            const float ex = fmaxf (fmaxf (lo.x - qx, qx - hi.x), 0.f);
            const float ey = fmaxf (fmaxf (lo.y - qy, qy - hi.y), 0.f);
            const float ez = fmaxf (fmaxf (lo.z - qz, qz - hi.z), 0.f);
            return __builtin_fmaf (ez, ez, __builtin_fmaf (ey, ey, ex * ex)) < lim_;
        };
        if (t0) {                                    // further tiles (nr > KT)
            __syncthreads ();
            if (prune) {
                const bool near = tile_near (s1_lim);
                if (!__syncthreads_or (near)) continue;
                const uint32_t nbx = 2u * ((tn + 2u * KS_SPLIT - 1u) / (2u * KS_SPLIT));
                for (uint32_t k = tid; k < nbx; k += 64 * KS_SPLIT) s_box[k] = GBt[2u * (t0 / (2u * KS_SPLIT)) + k];
                __syncthreads ();
                if (__ballot (near)) cmask = ks_coarse_pass<LPQ, TILE> (s_box, 0u, qx, qy, qz, ss, tn, s1_lim);
                if (!__syncthreads_or (cmask != 0u)) continue;
            }
            for (uint32_t k = tid; k < tn; k += 64 * KS_SPLIT) {
                float4 g = R4[2 * (size_t) (t0 + k)], c = R4[2 * (size_t) (t0 + k) + 1];
                float *dst = s_pairf + (k >> 1) * 12u + (k & 1u);
                dst[0] = g.x; dst[2] = g.y; dst[4] = g.z; dst[6] = c.x; dst[8] = c.y; dst[10] = c.z;
                if constexpr (!OWNER) s_on[k] = make_uint2 (gO[t0 + k], gN[t0 + k]);
            }
        }
        if ((tn & 1u) && tid == 0) {                 // odd tile (nr == 1): the pad slot never wins (NaN distance)
            float *dst = s_pairf + (tn >> 1) * 12u + 1u;
            const float qnan = __builtin_nanf ("");
            dst[0] = qnan; dst[2] = qnan; dst[4] = qnan; dst[6] = qnan; dst[8] = qnan; dst[10] = qnan;
        }
        // (chained variant, queries handed over inside the finalize: its closing barrier already stands behind every LDS write of
        // the prologue — representatives, list headers, boxes, queries —; a second one here would only be waited for)
        if (!(CHAIN && handed && t0 == 0u && !(tn & 1u))) __syncthreads ();
        if (t0 == 0) {                               // the query prepared by the query wave
            const float4 a4 = s_qa[qe], c4 = s_qc[qe];
            if constexpr (PRUNE && ICP_S1_ORIGIN_LIST) { if (prune) n_origin = (uint32_t) __builtin_amdgcn_readfirstlane ((int) __float_as_uint ((!SINGLE && nr > KT) ? s_tbox[1].w : s_box[1].w)); }
            qx = a4.x; qy = a4.y; qz = a4.z; i = __float_as_uint (a4.w); valid = i < m;
            qr = c4.x; qg = c4.y; qb = c4.z; seed = __float_as_uint (c4.w);
        }
        KS_STAMP (1)
        const uint32_t npair = (tn + 1u) >> 1;
        const float2v vqx = { qx, qx }, vqy = { qy, qy }, vqz = { qz, qz }, vqr = { qr, qr }, vqg = { qg, qg }, vqb = { qb, qb };
        const float2v va = { alpha, alpha };
        // a lane's pairs ascend (P = ss, ss+8, ..) and an update needs a strict '<', so each lane keeps its lowest
        // index among equal distances; the group reduction below then takes the lowest index overall.
        //
        // Exact pruning.  d = fma (a, pho, geo) >= geo for a > 0, so a pair whose two geo terms are not below `lim`
        // cannot hold the nearest representative when lim <= max (own best, a known upper bound of the query's
        // minimum): the photometric half, the third LDS read and the compare / select chain are skipped when no lane
        // of the wave needs them (wave-uniform branch; the wave's queries are neighbours).  The upper bound is the
        // distance to the seed (the previous search's nearest representative), bumped by one ulp so that a plain '<'
        // keeps every representative that could tie with it.
        float lim = __builtin_inff ();
        if (t0 == 0 && prune) {
            const uint32_t sfl = seed >> 30;             // (bit 1: an invalid point of its frame, bit 0: a seed from the grid cell — ks_seed_against_invalid)
            seed = min (seed & 0x3FFFFFFFu, nr - 1u);
            float sx, sy, sz, sr, sg, sb;
            if (MINW == 2 || nr <= KT) {        // one tile: the seed is in LDS (MINW == 2: always, see icp_launch_search)
                const float *sp = s_pairf + (seed >> 1) * 12u + (seed & 1u);
                sx = sp[0]; sy = sp[2]; sz = sp[4]; sr = sp[6]; sg = sp[8]; sb = sp[10];
                // (keeps the compiler from merging this with the global path below into flat loads)
                asm volatile ("" : "+v"(sx), "+v"(sy), "+v"(sz), "+v"(sr), "+v"(sg), "+v"(sb));
            } else {
                const float4 g = R4[2 * (size_t) seed], c = R4[2 * (size_t) seed + 1];
                sx = g.x; sy = g.y; sz = g.z; sr = c.x; sg = c.y; sb = c.z;
            }
            if constexpr (ICP_S1_ORIGIN_LIST && MINW == 4)
                ks_seed_against_invalid (sfl, (!SINGLE && nr > KT) ? s_tbox + 1 : s_box + 1, nr, b, R4, seed, sx, sy, sz, sr, sg, sb);
            const float b0 = icp_metric8 (qx, qy, qz, qr, qg, qb, sx, sy, sz, sr, sg, sb, alpha);
            if (b0 >= 0.f && b0 < __builtin_inff ()) lim = __uint_as_float (__float_as_uint (b0) + 1u);     // next float up
            s1_lim = lim;
        } else if (prune) lim = s1_lim;
        if (prune) {
            // coarse pass: a group = the 2 * LPQ representatives of one trip of the query's lanes; lane ss tests the
            // groups ss, ss + LPQ, ..  (further tiles: done above, before the tile was staged)
            if (t0 == 0 && __ballot (tile_near (lim))) cmask = ks_coarse_pass<LPQ, TILE> (s_box, 0u, qx, qy, qz, ss, tn, lim);
            ks_fine_pass<LPQ> (s_pair, 0u, qx, qy, qz, qr, qg, qb, alpha, gt_lg1, ss, t0, npair, cmask, best, bid);
            s1_lim = fminf (lim, ks_grp_min_f<KS_SPLIT> (best));
        } else {
#pragma unroll 8
            for (uint32_t P = ss; P < npair; P += KS_SPLIT) {
                float4 A = s_pair[3 * P], B = s_pair[3 * P + 1], C = s_pair[3 * P + 2];
                float2v x = { A.x, A.y }, y = { A.z, A.w }, z = { B.x, B.y }, r = { B.z, B.w }, g = { C.x, C.y }, bb = { C.z, C.w };
                float2v dx = vqx - x, dy = vqy - y, dz = vqz - z, dr = vqr - r, dg = vqg - g, db = vqb - bb;
                float2v geo = __builtin_elementwise_fma (dz, dz, __builtin_elementwise_fma (dy, dy, dx * dx));
                float2v pho = __builtin_elementwise_fma (db, db, __builtin_elementwise_fma (dg, dg, dr * dr));
                float2v d = __builtin_elementwise_fma (va, pho, geo);
                const uint32_t r0 = t0 + 2u * P;
                if (d.x < best) { best = d.x; bid = r0; }
                if (d.y < best) { best = d.y; bid = r0 + 1u; }
            }
        }
    }
    KS_STAMP (14)
    if constexpr (PRUNE && ICP_S1_ORIGIN_LIST) {
        // the representatives at the origin (invalid points): kept out of the boxes above, scanned here by the queries that are near the origin
        // Their number rides in a spare lane of the box array the block has staged in LDS anyway (hi.w of group box 0 / of tile box 0:
        // k_reps_and_boxes): one LDS read here.  (Loaded from global memory in the prologue — by a vector load, or by a scalar one with its
        // address arithmetic — it cost 0.8 - 1.3 % at |F| = 16384 x 64; the list's own pointer, held from the top of the kernel, made the
        // compiler spill scalars in front of the prologue's loads: search at |F| = 65536 11.88 -> 12.22 us.  Both are fetched here.)
        if (prune && __builtin_expect (n_origin != 0u, 0))
            ks_origin_section<KS_SPLIT, PB> (s_pair, &s_ovote, n_origin, nr, b, slice, tid, qx, qy, qz, qr, qg, qb, alpha, s1_lim, lane, ss, best, bid);
    }
    KS_KEEP (best, bid)
    KS_STAMP (2)
    const float dr = ks_grp_min_f<KS_SPLIT> (best);           // the query's nearest representative: smallest distance,
    // (the pruning variants start from a seed's distance and the origin list carries an explicit tie rule: with every distance +inf — a
    // query with an infinite coordinate — their (best, bid) is the seed or a tie among infinities, where the serial scan, which starts from
    // +inf and updates on a strict '<' only, has found nothing: a nearest distance that is not below +inf names no representative)
    const bool named = PRUNE ? (best == dr && dr < __builtin_inff ()) : (best == dr);
    uint32_t rstar = ks_grp_min_u<KS_SPLIT> (named ? bid : 0xFFFFFFFFu);     // ties -> lowest index
    if (rstar == 0xFFFFFFFFu) rstar = 0u;            // every distance inf / NaN: representative 0, as the serial scan would
    if constexpr (OWNER_LISTS) {
        ks_owner_lists_tail (p, s_qb, b, m, nb, lane, slice, qe, ss, valid, rstar);
        return;
    } else if constexpr (OWNER) {
        if (ss == 0u && valid) p.owner[(size_t) b * m + i] = rstar;
        return;
    }
    KS_KEEP (dr, rstar)
    KS_STAMP (3)
    // list offset / size of the winner.  One tile (always the case for the MINW == 2 variants, see icp_launch_search):
    // from LDS; the compile-time split keeps the compiler from merging the two sources into flat loads.
    uint32_t o, n;
    if constexpr (MINW == 2) { const uint2 on = s_on[rstar]; o = on.x; n = on.y; }
    else if constexpr (MASKED) {
        // the home tile's (offset, size) pairs are in LDS (buffer 1 is never overwritten): a winner there — the usual case — costs
        // no dependent global load
        if (rstar / KT == ht) { const uint2 on = s_on[rstar - ht * KT]; o = on.x; n = on.y; asm volatile ("" : "+v"(o), "+v"(n)); }
        else { o = gO[rstar]; n = gN[rstar]; }
    }
    else if (nr <= KT) { const uint2 on = s_on[rstar]; o = on.x; n = on.y; asm volatile ("" : "+v"(o), "+v"(n)); }
    else { o = gO[rstar]; n = gN[rstar]; }

    // ==================================================== STAGE 2 =====================================================
    // ---- stage 2: the list of that representative (ks_stage2_lanes / ks_stage2_wave)
    float dmin; uint32_t jmin;
    if constexpr (S2W) {
        static_assert (MINW == 4 && LPQ == 8 && !OWNER, "lanes = candidates: the dense search variants");
        ks_stage2_wave (XQb, o, n, valid, rstar, qx, qy, qz, qr, qg, qb, alpha, dr, b, lane, s_qa, s_qc, qe, dmin, jmin);
    } else ks_stage2_lanes<LPQ> (XQb, o, n, valid, qx, qy, qz, qr, qg, qb, alpha, dr, b, lane, ss, dmin, jmin);
    KS_KEEP (dmin, jmin)
    KS_STAMP (5)
    // ==================================================== EPILOGUE ====================================================
    // ---- epilogue (ks_epilogue): hand-off, finishing wave, outputs, block moments
    ks_epilogue<FUSED, CHAIN, MINW, LPQ, OWNER, PRUNE> (p, s_qa, s_qb, s_slot, s_mom, s_w, R4, XQb, b, m, nr, check_flags, tid, lane, slice, tile_id, qe, ss, i, valid, o, n, rstar, dr, dmin, jmin,
                                                        qx, qy, qz);
}


// ------------------------------------------------------------------------------------------
// host side, shared by the two translation units' launchers
// ------------------------------------------------------------------------------------------
static inline uint32_t icp_tpr_magic (uint32_t side)
{   // floor (2^32 / tpr) + 1 for tpr = side / 8 tiles per row (tpr == 1: one block, b == 0, any value works)
    const uint32_t tpr = side >> 3;
    return tpr ? (uint32_t) ((1ull << 32) / tpr + 1ull) : 0u;
}
#define KS_FLAGS(p) ((uint32_t) ((p).check ? 1u : 0u) | ((p).emit ? 8u : 0u) | ((p).warm_seed ? 32u : 0u) | ((p).xcdmap ? 64u : 0u))
#define KS_ARGS p.M, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, KS_FLAGS (p), p
#define KS_CHAIN_ARGS p.M, p.R, p.cst + p.slot, (const double *) p.mom + (size_t) p.slot * ICP_NMOM * p.nb, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, KS_FLAGS (p), p
