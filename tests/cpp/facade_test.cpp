// facade_test.cpp — drives the engine through the C++ facade the way a user of the reference drives
// cl_algo::ICP::ICP<POWER_METHOD, WEIGHTED> (compare src/ocl_icp_reg.cpp:103-120, 165-210).
// Prints k and the final [q | t, s] so that tests/test_gpu_facade.py can compare them with the oracle.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include <ICP/algorithms.hpp>

using namespace cl_algo::ICP;

int main (int argc, char **argv)
{
    const unsigned int side = argc > 1 ? (unsigned) atoi (argv[1]) : 64, r = argc > 2 ? (unsigned) atoi (argv[2]) : 64;
    const unsigned int m = side * side;
    const float a = 2e2f, c = 1e-6f;                                  // src/ocl_icp_reg.cpp:88
    // third argument "reference": the reference-order / literal modes (default: the benchmarked fused / squared ones)
    const icp::Mode mode = (argc > 3 && argv[3][0] == 'r') ? icp::Mode::REFERENCE_ORDER : icp::Mode::FAST;
    try
    {
        std::vector<float> F ((size_t) m * 8), M ((size_t) m * 8);
        const float axis[3] = { 0.3f, 0.9f, 0.1f }, t[3] = { 25.f, -10.f, 15.f };
        icp_synth_pair (0x1C9D5EEDull, side, 3.f, axis, t, 1.f, 0.01f, 0.f, F.data (), M.data ());

        icp::Env env (0);
        ICP<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> reg (env, mode);
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
        ICPStep<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> sbs (env, mode);
        sbs.init (m, r, a, c);
        sbs.write (decltype (sbs)::Memory::D_IN_F, F.data ());
        sbs.write (decltype (sbs)::Memory::D_IN_M, M.data ());
        sbs.buildRBC ();
        sbs.run (true); sbs.run ();
        printf ("S %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", sbs.q.x (), sbs.q.y (), sbs.q.z (), sbs.q.w (), sbs.t (0), sbs.t (1), sbs.t (2), sbs.s);

        // diagnostic (fourth argument "time"): what a user of the facade gets per iteration from the blocking ICP::run () —
        // graph launch, convergence checks on the device, host synchronisation and the pull of the public members included
        if (argc > 4 && argv[4][0] == 't')
        {
            const int reps = 50; unsigned long ktot = 0;
            reg.buildRBC (); reg.run ();                               // warm-up (graph capture)
            double us = 0.0, us_run = 0.0;
            for (int i = 0; i < reps; ++i)
            {
                const float T0[8] = { 0, 0, 0, 1, 0, 0, 0, 1 };
                const auto t0 = std::chrono::steady_clock::now ();
                reg.write (decltype (reg)::Memory::D_IO_T, (void *) T0, true);
                reg.buildRBC ();
                icp_sync (reg.handle ());
                const auto t1 = std::chrono::steady_clock::now ();
                reg.run ();
                const auto t2 = std::chrono::steady_clock::now ();
                us += std::chrono::duration<double, std::micro> (t2 - t0).count ();
                us_run += std::chrono::duration<double, std::micro> (t2 - t1).count ();
                ktot += reg.k;
            }
            printf ("TIME mode %s: %d registrations, k = %.1f: ICP::run () alone %.1f us = %.2f us per iteration (launch, device-side checks, "
                    "host synchronisation and the pull of the public members included); with write (T) + buildRBC %.1f us per registration\n",
                    mode == icp::Mode::FAST ? "fast" : "reference", reps, (double) ktot / reps, us_run / reps, us_run / (double) ktot, us / reps);
        }

        // re-init of the same object at another size, then at the first size again (the buffers fetched through get ()
        // after the first init belong to the engine and are re-created, never adopted): same result as a fresh object
        {
            const unsigned int side2 = side / 2, m2 = side2 * side2, r2 = r / 4 ? r / 4 : 1;
            std::vector<float> F2 ((size_t) m2 * 8), M2 ((size_t) m2 * 8);
            icp_synth_pair (0x1C9D5EEDull, side2, 3.f, axis, t, 1.f, 0.01f, 0.f, F2.data (), M2.data ());
            reg.init (m2, r2, a, c, 40, 0.001, 0.01, Staging::IO);
            reg.write (decltype (reg)::Memory::D_IN_F, F2.data ());
            reg.write (decltype (reg)::Memory::D_IN_M, M2.data ());
            reg.buildRBC ();
            reg.run ();
            reg.init (m, r, a, c, 40, 0.001, 0.01, Staging::IO);
            reg.write (decltype (reg)::Memory::D_IN_F, F.data ());
            reg.write (decltype (reg)::Memory::D_IN_M, M.data ());
            reg.buildRBC ();
            reg.run ();
            float *T2 = (float *) reg.read ();
            printf ("k2 %u\n", reg.k);
            printf ("T2 %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", T2[0], T2[1], T2[2], T2[3], T2[4], T2[5], T2[6], T2[7]);
        }

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

        // ICPTransform<QUATERNION> and <MATRIX> on the reference test's rotation (tests/testsICP.cpp:917-922: 36.21 degrees
        // about (1,1,1)/sqrt(3)): the two must agree within that test's tolerance, 42000 eps
        {
            const unsigned int n = 4096;
            ICPTransform<ICPTransformConfig::QUATERNION> tq (env); tq.init (n);
            ICPTransform<ICPTransformConfig::MATRIX> tm (env); tm.init (n);
            for (unsigned int i = 0; i < n * 8; ++i) tq.hPtrInM[i] = tm.hPtrInM[i] = (i % 8 == 3) ? 1.f : (float) ((i * 2654435761u) % 25500u) * 0.01f;   // homogeneous points
            const double half = 36.21 * M_PI / 360.0, ax = 1.0 / std::sqrt (3.0);
            const float Tq[8] = { (float) (ax * std::sin (half)), (float) (ax * std::sin (half)), (float) (ax * std::sin (half)), (float) std::cos (half), 7.f, -3.f, 11.f, 1.f };
            const float Tm[16] = { 0.871238f, -0.276687f, 0.405449f, 7.f, 0.405449f, 0.871238f, -0.276687f, -3.f, -0.276687f, 0.405449f, 0.871238f, 11.f, 0.f, 0.f, 0.f, 1.f };
            std::memcpy (tq.hPtrInT, Tq, sizeof Tq); std::memcpy (tm.hPtrInT, Tm, sizeof Tm);
            tq.write (); tm.write (); tq.run (); tm.run ();
            const float *a = (const float *) tq.read (), *b = (const float *) tm.read ();
            float worst = 0.f; int copied = 1;
            for (unsigned int i = 0; i < n; ++i) {
                for (int k = 0; k < 3; ++k) worst = std::max (worst, std::fabs (a[i * 8 + k] - b[i * 8 + k]));
                for (int k = 3; k < 8; ++k) copied &= (a[i * 8 + k] == tq.hPtrInM[i * 8 + k]) && (b[i * 8 + k] == tm.hPtrInM[i * 8 + k]);
            }
            printf ("TR %.9g %d\n", worst, copied);
        }

        // ICPPowerMethod on the reference's known-answer vector, written the way its test does (tests/testsICP.cpp:1000-1052):
        // init, write (D_IN_S), write (D_IN_MEAN), run, read; worst |Tk - svdTk| of the literal loop, the squared start and EIGEN
        {
            float S[11] = { 0.00168053f, 0.000131408f, -0.000775179f, 0.000156595f, 0.00102674f, -0.000563479f,
                            -0.000722137f, -0.000559463f, 0.00246661f, 0.00521271f, 0.00515292f };
            float means[8] = { -33.9694f, -17.6421f, 1494.22f, 0.f, -44.8322f, -19.3835f, 1485.93f, 0.f };
            const float svdTk[8] = { 0.00111412f, 0.00730956f, -0.00647493f, 0.999952f, -10.4598f, 4.74009f, -0.762817f, 1.00578f };
            ICPPowerMethod pm (env);
            pm.init ();
            pm.write (ICPPowerMethod::Memory::D_IN_S, S);
            pm.write (ICPPowerMethod::Memory::D_IN_MEAN, means);
            float worst[3]; unsigned int trips = 0;
            for (int v = 0; v < 3; ++v) {
                pm.setMode (v == 1 ? icp::Mode::FAST : icp::Mode::REFERENCE_ORDER);
                pm.setRotation (v == 2 ? ICPStepConfigT::EIGEN : ICPStepConfigT::POWER_METHOD);
                pm.run ();
                const float *res = (const float *) pm.read ();
                worst[v] = 0.f;
                for (int k = 0; k < 8; ++k) worst[v] = std::max (worst[v], std::fabs (res[k] - svdTk[k]));
                if (v == 0) trips = pm.iterations;
            }
            printf ("PM %.9g %.9g %.9g %u\n", worst[0], worst[1], worst[2], trips);
        }

        // the per-kernel classes chained the way ICPStep::init wires them (src/ICP/algorithms.cpp:4538-4576) — by SHARED DEVICE BUFFERS: an
        // input's get (Memory) is assigned the producer's get (Memory) before init, so nothing travels through the host between the stages
        // (weights -> weighted means -> deviations -> S; four run () calls enqueue kernels only), on the synthetic pair with distances made
        // up from the index; the last S goes to the test
        {
            ICPLMs lmsK (env); lmsK.init ();                             // (instantiation only: a VGA cloud is exercised by the Python test)
            ICPReps reps (env); reps.init (r, m); reps.write (ICPReps::Memory::D_IN, F.data ()); reps.run ();
            ICPWeights wts (env); wts.init (m);
            for (unsigned int i = 0; i < m; ++i) { wts.hPtrIn[i].dist = (float) ((i * 2654435761u) % 1000u) * 0.001f; wts.hPtrIn[i].id = i; }
            wts.write ();                                                // staging -> device (reference: write (D_IN, nullptr))
            typedef ICPMean<ICPMeanConfig::WEIGHTED> MeanW; typedef ICPS<ICPSConfig::WEIGHTED> SW;
            MeanW mean (env);
            mean.get (MeanW::Memory::D_IN_W) = wts.get (ICPWeights::Memory::D_OUT_W);
            mean.get (MeanW::Memory::D_IN_SUM_W) = wts.get (ICPWeights::Memory::D_OUT_SUM_W);
            mean.init (m);
            mean.write (MeanW::Memory::D_IN_F, F.data ()); mean.write (MeanW::Memory::D_IN_M, M.data ());
            ICPDevs devs (env);
            devs.get (ICPDevs::Memory::D_IN_F) = mean.get (MeanW::Memory::D_IN_F);
            devs.get (ICPDevs::Memory::D_IN_M) = mean.get (MeanW::Memory::D_IN_M);
            devs.get (ICPDevs::Memory::D_IN_MEAN) = mean.get (MeanW::Memory::D_OUT);
            devs.init (m);
            SW S (env);
            S.get (SW::Memory::D_IN_DEV_M) = devs.get (ICPDevs::Memory::D_OUT_DEV_M);
            S.get (SW::Memory::D_IN_DEV_F) = devs.get (ICPDevs::Memory::D_OUT_DEV_F);
            S.get (SW::Memory::D_IN_W) = wts.get (ICPWeights::Memory::D_OUT_W);
            S.init (m, c);
            if (S.get (SW::Memory::D_IN_W) != wts.get (ICPWeights::Memory::D_OUT_W) || devs.get (ICPDevs::Memory::D_IN_MEAN) != mean.get (MeanW::Memory::D_OUT))
                throw std::runtime_error ("the wired objects do not share their buffers");
            wts.run (); mean.run (); devs.run (); S.run ();              // kernels only: no host copy between the stages
            const float *s11 = (const float *) S.read (), *r0 = (const float *) reps.read ();
            printf ("KC %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.17g %.9g\n", s11[0], s11[1], s11[2], s11[3], s11[4], s11[5], s11[6], s11[7], s11[8], s11[9], s11[10],
                    *(const double *) wts.read (ICPWeights::Memory::H_OUT_SUM_W), r0[8 * (r - 1)]);
        }

        // the reference's profiling run: 40 steps, per-stage table (include/ICP/algorithms.hpp:2482-2494)
        {
            ICP<ICPStepConfigT::POWER_METHOD, ICPStepConfigW::WEIGHTED> prof (env, mode);
            prof.init (m, r, a, c, 40, 0.001, 0.01, Staging::IO);
            prof.write (decltype (prof)::Memory::D_IN_F, F.data ());
            prof.write (decltype (prof)::Memory::D_IN_M, M.data ());
            prof.buildRBC ();
            icp::ProfilingInfo info;
            const double ms = prof.run (info, 40, false);
            printf ("PROF %u %u %.6f %.6f %.6f\n", info.steps (), prof.k, ms, info.total (ICP_STAGE_SEARCH), info.total (ICP_STAGE_FINALIZE));
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
