// icpreg_test.cpp — the reference's demo flow (examples/registration.cpp:285-337 + src/ocl_icp_reg.cpp) through ICPReg:
// two synthetic VGA clouds, init, registerPC.  Prints k, [q | t, s] and a checksum of the transformed cloud so that
// tests/test_gpu_facade.py can compare them with the Python / oracle flow.
#include <cstdio>
#include <cstring>
#include <ocl_icp_reg.hpp>
#include <ocl_icp_sbs.hpp>

int main (int argc, char **argv)
{
    const icp::Mode mode = (argc > 1 && argv[1][0] == 'r') ? icp::Mode::REFERENCE_ORDER : icp::Mode::FAST;
    try
    {
        std::vector<icp_float8> pc1 (640 * 480), pc2 (640 * 480);
        if (icp_synth_cloud_vga (0x1C9D5EEDull, 0, pc1[0].data ()) || icp_synth_cloud_vga (0x1C9D5EEDull, 1, pc2[0].data ())) return 2;
        ICPReg<cl_algo::ICP::ICPStepConfigT::POWER_METHOD, cl_algo::ICP::ICPStepConfigW::WEIGHTED> app (0, mode);
        app.init (pc1, pc2);
        app.registerPC ();
        auto &reg = app.registration ();
        printf ("k %u\n", reg.k);             // (Staging::NONE as in the reference: results through the public members)
        printf ("T %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", reg.q.x (), reg.q.y (), reg.q.z (), reg.q.w (), reg.t (0), reg.t (1), reg.t (2), reg.s);
        double cs[3] = { 0, 0, 0 };
        for (const auto &p : app.transformed ()) { cs[0] += p[0]; cs[1] += p[1]; cs[2] += p[2]; }
        printf ("C %.17g %.17g %.17g\n", cs[0], cs[1], cs[2]);

        // the step-by-step application (src/ocl_icp_sbs.cpp): three steps
        ICPSBS<cl_algo::ICP::ICPStepConfigT::POWER_METHOD, cl_algo::ICP::ICPStepConfigW::WEIGHTED> sbs (0, mode);
        sbs.init (pc1, pc2);
        sbs.step (); sbs.step (); sbs.step ();
        auto &st = sbs.stepper ();
        printf ("S %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", st.q.x (), st.q.y (), st.q.z (), st.q.w (), st.t (0), st.t (1), st.t (2), st.s);

        // frame-to-frame tracking (ICPTrack): four frames of the synthetic sequence, two in flight, the first two through the
        // engine's pinned frame buffers; one line per hop
        {
            cl_algo::ICP::ICPTrack<cl_algo::ICP::ICPStepConfigT::POWER_METHOD, cl_algo::ICP::ICPStepConfigW::WEIGHTED> trk (icp::Env (0), mode);
            trk.init ();
            std::vector<icp_float8> f (640 * 480);
            int pending = 0;
            for (int i = 0; i < 4; ++i) {
                if (icp_synth_cloud_vga (0x1C9D5EEDull, i, f[0].data ())) return 2;
                const void *src = f[0].data ();
                if (i < 2) { float *st = trk.staging ((unsigned) i); std::memcpy (st, src, (size_t) 640 * 480 * 32); src = st; }
                if (pending == 2) { if (trk.collect ()) printf ("H %u %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", trk.k, trk.q.x (), trk.q.y (), trk.q.z (), trk.q.w (), trk.t (0), trk.t (1), trk.t (2), trk.s); --pending; }
                trk.submit (src);                    // (pageable sources are copied out before submit returns: f can be refilled)
                ++pending;
            }
            while (pending--) if (trk.collect ()) printf ("H %u %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", trk.k, trk.q.x (), trk.q.y (), trk.q.z (), trk.q.w (), trk.t (0), trk.t (1), trk.t (2), trk.s);
        }
    }
    catch (const std::exception &e)
    {
        fprintf (stderr, "%s\n", e.what ());
        return 1;
    }
    return 0;
}
