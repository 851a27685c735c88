// registration.cpp — the reference's first example (examples/registration.cpp: two 640 x 480 8-D clouds, "T" = register, report,
// show the result) as a command-line program over the MI355X engine: no window, no GL buffers — the transformed cloud goes to a
// file instead of a vertex buffer.
//
//   registration                          a synthetic pair (the data files of the reference are not distributed: .MISSING_LARGE_BLOBS)
//   registration NAME                     data/NAME_1.bin, data/NAME_2.bin      (the reference's argument convention, :299-329)
//   registration A B                      data/A.bin, data/B.bin — or A and B themselves when they name existing files
//   ... [--out FILE] [--device N] [--reference-order] [--svd]
//
// A cloud file is 640 x 480 points of 8 floats [x y z 1 r g b 1], little endian, row-major (src/kinect_frame_grabber.cpp:252-272).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <ocl_icp_reg.hpp>

namespace {

const size_t kPoints = 640u * 480u;

bool exists (const std::string &p) { std::ifstream f (p, std::ios::binary); return f.good (); }

void read_cloud (const std::string &path, std::vector<icp_float8> &pc)
{
    std::ifstream f (path, std::ios::binary);
    if (!f) throw std::runtime_error ("cannot open " + path);
    pc.resize (kPoints);
    f.read (reinterpret_cast<char *> (pc.data ()), (std::streamsize) (kPoints * sizeof (icp_float8)));
    if ((size_t) f.gcount () != kPoints * sizeof (icp_float8)) throw std::runtime_error (path + ": expected 640 x 480 x 8 floats");
}

std::string data_path (const std::string &name) { return exists (name) ? name : "../data/" + name + ".bin"; }

template <cl_algo::ICP::ICPStepConfigT RC>
int run (int device, icp::Mode mode, const std::vector<icp_float8> &pc1, const std::vector<icp_float8> &pc2, const std::string &out)
{
    ICPReg<RC, cl_algo::ICP::ICPStepConfigW::WEIGHTED> app (device, mode);
    app.init (pc1, pc2);
    app.registerPC ();                                        // buildRBC + run + transform + the reference's report
    auto &reg = app.registration ();
    std::printf ("\n    q = (%.9g, %.9g, %.9g, %.9g)   t = (%.9g, %.9g, %.9g)   s = %.9g   k = %u\n",
                 reg.q.x (), reg.q.y (), reg.q.z (), reg.q.w (), reg.t (0), reg.t (1), reg.t (2), reg.s, reg.k);
    if (!out.empty ()) {
        std::ofstream f (out, std::ios::binary);
        f.write (reinterpret_cast<const char *> (app.transformed ().data ()), (std::streamsize) (kPoints * sizeof (icp_float8)));
        if (!f) throw std::runtime_error ("cannot write " + out);
        std::printf ("    transformed cloud     :    %s\n", out.c_str ());
    }
    return 0;
}

}  // namespace

int main (int argc, char **argv)
{
    std::vector<std::string> names;
    std::string out;
    int device = 0; bool svd = false;
    icp::Mode mode = icp::Mode::FAST;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi (argv[++i]);
        else if (a == "--reference-order") mode = icp::Mode::REFERENCE_ORDER;
        else if (a == "--svd") svd = true;
        else if (a.rfind ("--", 0) == 0) { std::fprintf (stderr, "unknown option %s\n", a.c_str ()); return 2; }
        else names.push_back (a);
    }
    try
    {
        std::vector<icp_float8> pc1, pc2;
        if (names.empty ()) {
            pc1.resize (kPoints); pc2.resize (kPoints);
            if (icp_synth_cloud_vga (0x1C9D5EEDull, 0, pc1[0].data ()) || icp_synth_cloud_vga (0x1C9D5EEDull, 1, pc2[0].data ())) return 2;
            std::printf ("(no files given: a synthetic pair)\n");
        } else if (names.size () == 1) {
            read_cloud ("../data/" + names[0] + "_1.bin", pc1); read_cloud ("../data/" + names[0] + "_2.bin", pc2);
        } else {
            read_cloud (data_path (names[0]), pc1); read_cloud (data_path (names[1]), pc2);
        }
        return svd ? run<cl_algo::ICP::ICPStepConfigT::EIGEN> (device, mode, pc1, pc2, out)
                   : run<cl_algo::ICP::ICPStepConfigT::POWER_METHOD> (device, mode, pc1, pc2, out);
    }
    catch (const std::exception &e)
    {
        std::fprintf (stderr, "%s\n", e.what ());
        return 1;
    }
}
