// facade_test.cpp — drives the engine through the C++ facade the way a user of the reference drives
// cl_algo::ICP::ICP<POWER_METHOD, WEIGHTED> (compare src/ocl_icp_reg.cpp:103-120, 165-210).
// Prints k and the final [q | t, s] so that tests/test_gpu_facade.py can compare them with the oracle.
#include <algorithm>
#include <cstdio>
#include <vector>
#include <ICP/algorithms.hpp>

using namespace cl_algo::ICP;

int main (int argc, char **argv)
{
    const unsigned int side = argc > 1 ? (unsigned) atoi (argv[1]) : 64, r = argc > 2 ? (unsigned) atoi (argv[2]) : 64;
    const unsigned int m = side * side;
    const float a = 2e2f, c = 1e-6f;                                  // src/ocl_icp_reg.cpp:88
    try
    {
        std::vector<float> F ((size_t) m * 8), M ((size_t) m * 8);
        const float axis[3] = { 0.3f, 0.9f, 0.1f }, t[3] = { 25.f, -10.f, 15.f };
        icp_synth_pair (0x1C9D5EEDull, side, 3.f, axis, t, 1.f, 0.01f, 0.f, F.data (), M.data ());

        icp::Env env (0);
        ICP<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> reg (env);
        reg.init (m, r, a, c, 40, 0.001, 0.01, Staging::IO);
        reg.write (decltype (reg)::Memory::D_IN_F, F.data ());
        reg.write (decltype (reg)::Memory::D_IN_M, M.data ());
        reg.buildRBC ();
        reg.run ();
        float *T = (float *) reg.read ();
        printf ("k %u\n", reg.k);
        printf ("T %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7]);
        printf ("q %.9g %.9g %.9g %.9g s %.9g\n", reg.q.x (), reg.q.y (), reg.q.z (), reg.q.w (), reg.s);

        // step-by-step object (src/ocl_icp_sbs.cpp:167-181): two iterations
        ICPStep<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> sbs (env);
        sbs.init (m, r, a, c);
        sbs.write (decltype (sbs)::Memory::D_IN_F, F.data ());
        sbs.write (decltype (sbs)::Memory::D_IN_M, M.data ());
        sbs.buildRBC ();
        sbs.run (true); sbs.run ();
        printf ("S %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", sbs.q.x (), sbs.q.y (), sbs.q.z (), sbs.q.w (), sbs.t (0), sbs.t (1), sbs.t (2), sbs.s);

        // standalone Reduce / Scan classes (tests/testsReduce.cpp, tests/testsScan.cpp of the reference)
        {
            const unsigned cols = 1024, rows = 3;
            std::vector<float> v ((size_t) cols * rows);
            for (size_t i = 0; i < v.size (); ++i) v[i] = (float) ((i * 2654435761u) % 1000u) * 0.25f - 100.f;
            Reduce<ReduceConfig::MIN, float> rmin (env); rmin.init (cols, rows); rmin.write (decltype (rmin)::Memory::D_IN, v.data ()); rmin.run ();
            Reduce<ReduceConfig::SUM, float> rsum (env); rsum.init (cols, rows); rsum.write (decltype (rsum)::Memory::D_IN, v.data ()); rsum.run ();
            const float *mn = (const float *) rmin.read (), *sm = (const float *) rsum.read ();
            std::vector<int32_t> w ((size_t) cols * rows);
            for (size_t i = 0; i < w.size (); ++i) w[i] = (int32_t) (i % 7u) - 3;
            Scan<ScanConfig::EXCLUSIVE> sc (env); sc.init (cols, rows); sc.write (decltype (sc)::Memory::D_IN, w.data ()); sc.run ();
            const int32_t *so = (const int32_t *) sc.read ();
            int bad = 0;
            for (unsigned r = 0; r < rows; ++r) {
                float m0 = v[(size_t) r * cols]; long long run = 0;
                for (unsigned c = 0; c < cols; ++c) {
                    m0 = std::min (m0, v[(size_t) r * cols + c]);
                    if (so[(size_t) r * cols + c] != run) ++bad;
                    run += w[(size_t) r * cols + c];
                }
                if (mn[r] != m0) ++bad;
            }
            printf ("RS %d %.9g %.9g %.9g\n", bad, sm[0], sm[1], sm[2]);
        }

        // argument errors surface as exceptions, not exit()
        try { ICP<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> bad (env); bad.init (m, r, 0.f); printf ("ERR missing\n"); }
        catch (const std::runtime_error &e) { printf ("ERR %s\n", e.what ()); }
    }
    catch (const std::exception &e)
    {
        fprintf (stderr, "%s\n", e.what ());
        return 1;
    }
    return 0;
}
