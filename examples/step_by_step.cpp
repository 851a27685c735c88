// step_by_step.cpp — the reference's second example (examples/step_by_step.cpp: "T" = one ICP iteration, report, show the
// result) as a command-line program over the MI355X engine: N iterations, one report each.
//
//   step_by_step [N] [A B]     N iterations (default 10) on a synthetic pair, or on the cloud files A and B
//                              (640 x 480 points of 8 floats [x y z 1 r g b 1]; names without a path: ../data/NAME.bin)
//   ... [--device N] [--reference-order]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <ocl_icp_sbs.hpp>

namespace {

const size_t kPoints = 640u * 480u;

void read_cloud (const std::string &name, std::vector<icp_float8> &pc)
{
    std::string path = name;
    { std::ifstream probe (path, std::ios::binary); if (!probe.good ()) path = "../data/" + name + ".bin"; }
    std::ifstream f (path, std::ios::binary);
    if (!f) throw std::runtime_error ("cannot open " + path);
    pc.resize (kPoints);
    f.read (reinterpret_cast<char *> (pc.data ()), (std::streamsize) (kPoints * sizeof (icp_float8)));
    if ((size_t) f.gcount () != kPoints * sizeof (icp_float8)) throw std::runtime_error (path + ": expected 640 x 480 x 8 floats");
}

}  // namespace

int main (int argc, char **argv)
{
    std::vector<std::string> names;
    int device = 0, steps = 10;
    icp::Mode mode = icp::Mode::FAST;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = std::atoi (argv[++i]);
        else if (a == "--reference-order") mode = icp::Mode::REFERENCE_ORDER;
        else if (a.rfind ("--", 0) == 0) { std::fprintf (stderr, "unknown option %s\n", a.c_str ()); return 2; }
        else if (names.empty () && a.find_first_not_of ("0123456789") == std::string::npos) steps = std::atoi (a.c_str ());
        else names.push_back (a);
    }
    try
    {
        std::vector<icp_float8> pc1, pc2;
        if (names.size () >= 2) { read_cloud (names[0], pc1); read_cloud (names[1], pc2); }
        else {
            pc1.resize (kPoints); pc2.resize (kPoints);
            if (icp_synth_cloud_vga (0x1C9D5EEDull, 0, pc1[0].data ()) || icp_synth_cloud_vga (0x1C9D5EEDull, 1, pc2[0].data ())) return 2;
            std::printf ("(no files given: a synthetic pair)\n");
        }
        ICPSBS<cl_algo::ICP::ICPStepConfigT::POWER_METHOD, cl_algo::ICP::ICPStepConfigW::WEIGHTED> app (device, mode);
        app.init (pc1, pc2);
        for (int k = 0; k < steps; ++k) app.step ();          // (first call: buildRBC) one iteration, transform, the reference's report
        auto &st = app.stepper ();
        std::printf ("\n    q = (%.9g, %.9g, %.9g, %.9g)   t = (%.9g, %.9g, %.9g)   s = %.9g   after %d iterations\n",
                     st.q.x (), st.q.y (), st.q.z (), st.q.w (), st.t (0), st.t (1), st.t (2), st.s, steps);
        return 0;
    }
    catch (const std::exception &e)
    {
        std::fprintf (stderr, "%s\n", e.what ());
        return 1;
    }
}
