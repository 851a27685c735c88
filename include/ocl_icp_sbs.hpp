/*! \file ocl_icp_sbs.hpp
 *  \brief `ICPSBS<RC, WC>` — the step-by-step class of the reference's second demo application
 *         (include/ocl_icp_sbs.hpp:54-88, src/ocl_icp_sbs.cpp:81-215) on top of the MI355X engine:
 *         `init (pc8d1, pc8d2)` and `step ()`, one ICP iteration per call with the reference's per-step report.
 *         As with `ICPReg` (ocl_icp_reg.hpp) the OpenGL / OpenCL plumbing is gone: the transformed moving cloud is
 *         returned by `transformed ()`.
 */
#ifndef OCL_ICP_SBS_HPP
#define OCL_ICP_SBS_HPP

#include <array>
#include <chrono>
#include <cmath>
#include <iostream>
#include <vector>
#include <ICP/algorithms.hpp>

#ifndef OCL_ICP_REG_HPP
typedef std::array<float, 8> icp_float8;   /*!< the layout of the reference's `cl_float8` points */
#endif

template <cl_algo::ICP::ICPStepConfigT RC, cl_algo::ICP::ICPStepConfigW WC>
class ICPSBS
{
public:
    /*! \brief reference: `ICPSBS (GLuint*, GLuint*)`, src/ocl_icp_sbs.cpp:81-118 (sizes and parameters :82, :88). */
    explicit ICPSBS (int device = 0, icp::Mode mode = icp::Mode::FAST) :
        width (640), height (480), n (640 * 480), m (16384), r (256), a (2e2f), c (1e-6f),
        env (device), icpStep (env, mode), config (true), k (0)
    {
        icpStep.init (m, r, a, c, cl_algo::ICP::Staging::NONE);
    }

    /*! \brief reference `init`, src/ocl_icp_sbs.cpp:126-158: the two clouds and their landmarks. */
    void init (const std::vector<icp_float8> &pc8d1, const std::vector<icp_float8> &pc8d2)
    {
        if (pc8d1.size () != n || pc8d2.size () != n) throw std::runtime_error ("ICPSBS::init: the clouds must hold 640 x 480 points");
        moving = pc8d2;
        check (icp_write_cloud (icpStep.handle (), ICP_MEM_F, pc8d1.data (), 1));
        check (icp_write_cloud (icpStep.handle (), ICP_MEM_M, pc8d2.data (), 1));
        config = true; k = 0;
    }

    /*! \brief reference `step`, src/ocl_icp_sbs.cpp:167-215: (first call: buildRBC) one iteration, transform, report. */
    void step ()
    {
        if (config) icpStep.buildRBC ();
        const auto t0 = std::chrono::steady_clock::now ();
        icpStep.run (config);                                 // Take one ICP step (refine transformation); blocking read-back
        const double latency = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count ();
        moved.resize (n);
        check (icp_transform_cloud (icpStep.handle (), moving.data (), moved.data (), n));   // Transform the moving point cloud
        config = false;

        const double sinth_2 = icpStep.q.vec ().norm ();
        const double angle = 180.0 / M_PI * 2 * std::atan2 (sinth_2, (double) icpStep.q.w ());
        icp::Vector3f axis;
        if (sinth_2 != 0.0) for (int i = 0; i < 3; ++i) axis (i) = (float) (icpStep.q.vec () (i) / sinth_2);
        std::cout << std::endl << "================" << std::endl << std::endl;
        std::cout << "Iteration k = " << k++ << ":  " << std::endl << std::endl;
        std::cout << "    Latency               :    " << latency << " ms" << std::endl;
        std::cout << "    Rotation angle        :    " << angle << " degrees" << std::endl;
        std::cout << "    Rotation axis         :    " << axis (0) << " " << axis (1) << " " << axis (2) << std::endl;
        std::cout << "    Translation vector    :    " << icpStep.t (0) << " " << icpStep.t (1) << " " << icpStep.t (2) << std::endl;
        std::cout << "    Scale                 :    " << icpStep.s << std::endl;
        std::cout << "    Change in translation :    " << icpStep.tk.norm () << " mm" << std::endl;
    }

    const std::vector<icp_float8>& transformed () const { return moved; }
    cl_algo::ICP::ICPStep<RC, WC>& stepper () { return icpStep; }

private:
    void check (int rc) { if (rc != ICP_OK) throw std::runtime_error (std::string ("ICPSBS: ") + icp_last_error (icpStep.handle ())); }

    unsigned int width, height, n, m, r;
    float a, c;
    icp::Env env;
    cl_algo::ICP::ICPStep<RC, WC> icpStep;
    std::vector<icp_float8> moving, moved;
    bool config;
    int k;
};

#endif  // OCL_ICP_SBS_HPP
