/* capi_example.c — plain C99 user of the C-ABI (no C++ anywhere on the caller's side): registers the synthetic
 * pair with every ICPStep specialisation and prints k and T.  Built by tests/test_gpu_facade.py with gcc. */
#include <stdio.h>
#include <stdlib.h>
#include "icp_amd.h"

int main (int argc, char **argv)
{
    /* argument "reference": the reference-order / literal modes instead of the defaults (fused / squared) */
    const int reference = argc > 1 && argv[1][0] == 'r';
    const uint32_t side = 64, m = side * side, nr = 64;
    float *F = (float *) malloc ((size_t) m * 32), *M = (float *) malloc ((size_t) m * 32);
    const float axis[3] = { 0.3f, 0.9f, 0.1f }, t[3] = { 25.f, -10.f, 15.f };
    if (icp_synth_pair (0x1C9D5EEDull, side, 3.f, axis, t, 1.f, 0.01f, 0.f, F, M)) return 2;
    for (int rot = 0; rot < 2; ++rot)
        for (int w = 0; w < 2; ++w) {
            icp_handle h = NULL;
            if (icp_create (&h, 0, rot, w) != ICP_OK) { fprintf (stderr, "%s\n", icp_last_error (NULL)); return 1; }
            uint32_t k = 0; float T[8];
            if (reference && (icp_set_reduce_mode (h, ICP_REDUCE_REFERENCE_ORDER) || icp_set_power_mode (h, ICP_POWER_LITERAL))) return 1;
            if (icp_init (h, m, nr, 2e2f, 1e-6f, 40, 0.001, 0.01) || icp_write (h, ICP_MEM_F, F, 0) || icp_write (h, ICP_MEM_M, M, 0) ||
                icp_build_rbc (h) || icp_run (h, &k) || icp_read (h, ICP_MEM_T, T, sizeof T)) {
                fprintf (stderr, "%s\n", icp_last_error (h)); return 1;
            }
            printf ("rot %d w %d k %u T %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", rot, w, k, T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7]);
            icp_destroy (h);
        }
    free (F); free (M);
    return 0;
}
