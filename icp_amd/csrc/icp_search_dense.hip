// icp_search_dense.hip — the DENSE variants of the search kernel (icp_search.h: k_search<.., MINW = 4, LPQ = 8, ..>): 512-thread blocks,
// several per CU, eight lanes per query, exact stage-1 pruning; LDS tiles of 256 representatives (SINGLE: |R| <= 256, batches; MASKED:
// 256 < |R| <= 4096, the tile set of a block decided in one pre-pass) or 1024 (beyond); stage 2 by a query's lanes or with lanes = candidates
// (S2W: lists of >= 128 candidates).  Which one runs: icp_dense / icp_dense_tile / icp_s2_wave_of (icp_kernels.hip), reported by
// icp_search_layout.  The latency variants, the chained form and every other kernel of the iteration live in icp_kernels.hip.
#include "icp_search.h"

// RBC construct, step 1: owner(x) = nearest representative — the search kernel's stage 1 over the fixed points
// (LDS tiles of 256 representatives up to |R| = 4096 — four blocks per CU —, of 1024 beyond, where a 4 x 4 tile group no longer fits a
// 256-tile: icp_dense_tile)
void icp_launch_owner_search_dense (const icp_params &p, hipStream_t s)
{
#define KS_OWNER_ARGS p.F, p.R, p.st, (const double *) p.mom, p.m, p.nr, p.side, icp_tpr_magic (p.side), p.nb, 0u, p
    if (p.nr > 256u && icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, true, 1, 256, false>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_OWNER_ARGS);
    else if (icp_dense_tile (p) == 256u) hipLaunchKernelGGL ((k_search<true, false, 4, 8, true, 1, 256, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_OWNER_ARGS);
    else hipLaunchKernelGGL ((k_search<true, false, 4, 8, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_OWNER_ARGS);
#undef KS_OWNER_ARGS
}

void icp_launch_search_dense (const icp_params &p, hipStream_t s)
{
    const bool t256 = icp_dense_tile (p) == 256u, multi = p.nr > 256u;
    if (p.fused) {
        if (p.s2wave && multi && t256) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, false, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (p.s2wave && t256) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, true, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (p.s2wave) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 1024, false, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (multi && t256) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, false>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (t256) hipLaunchKernelGGL ((k_search<true, false, 4, 8, false, 1, 256, true>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
        else hipLaunchKernelGGL ((k_search<true, false, 4, 8>), dim3 (p.nb, p.batch), dim3 (512), 0, s, KS_ARGS);
    } else {
        // (the same tile choice as the fused variants: the tile boxes of a registration are built for one tile size, p.tbox)
        if (p.s2wave && multi && t256) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, false, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (p.s2wave && t256) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, true, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (p.s2wave) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 1024, false, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (multi && t256) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, false>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else if (t256) hipLaunchKernelGGL ((k_search<false, false, 4, 8, false, 1, 256, true>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
        else hipLaunchKernelGGL ((k_search<false, false, 4, 8>), dim3 (2 * p.nwg, p.batch), dim3 (512), 0, s, KS_ARGS);
    }
}
