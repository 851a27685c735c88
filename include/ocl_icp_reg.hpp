/*! \file ocl_icp_reg.hpp
 *  \brief `ICPReg<RC, WC>` — the registration class of the reference's demo application
 *         (include/ocl_icp_reg.hpp:51-98, src/ocl_icp_reg.cpp:81-210) on top of the MI355X engine.
 *
 *  Same public surface: a constructor, `init (pc8d1, pc8d2)` with the two 640 x 480 point clouds
 *  (8-D points [x y z 1 r g b 1]) and `registerPC ()`, which builds the RBC structure, runs the registration,
 *  transforms the whole moving cloud and prints the same report.  What is gone is the OpenGL / OpenCL plumbing of
 *  the demo (the GL vertex-buffer ids of the constructor and the interop copies): the transformed cloud is
 *  available through `transformed ()` instead of a GL buffer, and the device ordinal replaces the CL environment.
 */
#ifndef OCL_ICP_REG_HPP
#define OCL_ICP_REG_HPP

#include <array>
#include <chrono>
#include <cmath>
#include <iostream>
#include <vector>
#include <ICP/algorithms.hpp>

/*! \brief Plain 8-float point, the layout of the reference's `cl_float8` point clouds. */
typedef std::array<float, 8> icp_float8;

template <cl_algo::ICP::ICPStepConfigT RC, cl_algo::ICP::ICPStepConfigW WC>
class ICPReg
{
public:
    /*! \brief reference: `ICPReg (GLuint *glPC4DBuffer, GLuint *glRGBABuffer)`, src/ocl_icp_reg.cpp:81-120 —
     *         sizes and parameters as there (:82, :88): 640 x 480 clouds, 16384 landmarks, 256 representatives,
     *         a = 2e2, c = 1e-6, 40 iterations, 0.001 degrees, 0.01 mm. */
    explicit ICPReg (int device = 0, icp::Mode mode = icp::Mode::FAST) :
        width (640), height (480), n (640 * 480), m (16384), r (256),
        a (2e2f), c (1e-6f), max_iterations (40), angle_threshold (0.001), translation_threshold (0.01),
        env (device), reg (env, mode), latency_ms (0.0)
    {
        reg.init (m, r, a, c, max_iterations, angle_threshold, translation_threshold, cl_algo::ICP::Staging::NONE);
    }

    /*! \brief reference `init (pc8d1, pc8d2)`, src/ocl_icp_reg.cpp:128-153: uploads the two clouds and extracts
     *         the 128 x 128 landmarks of each (`ICPLMs`, kernels/icp_kernels.cl:63-76) as fixed / moving set. */
    void init (const std::vector<icp_float8> &pc8d1, const std::vector<icp_float8> &pc8d2)
    {
        if (pc8d1.size () != n || pc8d2.size () != n) throw std::runtime_error ("ICPReg::init: the clouds must hold 640 x 480 points");
        moving = pc8d2;
        check (icp_write_cloud (reg.handle (), ICP_MEM_F, pc8d1.data (), 1));
        check (icp_write_cloud (reg.handle (), ICP_MEM_M, pc8d2.data (), 1));
    }

    /*! \brief reference `registerPC ()`, src/ocl_icp_reg.cpp:165-202. */
    void registerPC ()
    {
        reg.buildRBC ();                                      // Build the RBC data structure
        const auto t0 = std::chrono::steady_clock::now ();
        reg.run ();                                           // Perform the ICP registration
        latency_ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count ();
        moved.resize (n);
        check (icp_transform_cloud (reg.handle (), moving.data (), moved.data (), n));   // Transform the moving point cloud

        const double sinth_2 = reg.q.vec ().norm ();
        const double angle = 180.0 / M_PI * 2 * std::atan2 (sinth_2, (double) reg.q.w ());
        icp::Vector3f axis;
        if (sinth_2 != 0.0) for (int i = 0; i < 3; ++i) axis (i) = (float) (reg.q.vec () (i) / sinth_2);
        std::cout << std::endl << "================" << std::endl << std::endl;
        std::cout << "    Iterations            :    " << reg.k << std::endl;
        std::cout << "    Latency               :    " << latency_ms << " ms" << std::endl;
        std::cout << "    Rotation angle        :    " << angle << " degrees" << std::endl;
        std::cout << "    Rotation axis         :    " << axis (0) << " " << axis (1) << " " << axis (2) << std::endl;
        std::cout << "    Translation vector    :    " << reg.t (0) << " " << reg.t (1) << " " << reg.t (2) << std::endl;
        std::cout << "    Scale                 :    " << reg.s << std::endl;
    }

    /*! \brief The moving cloud after `registerPC ()` (the reference writes it into the GL vertex buffer). */
    const std::vector<icp_float8>& transformed () const { return moved; }
    /*! \brief The registration object itself: `k`, `q`, `t`, `s`, `R` as in the reference's `reg` member. */
    cl_algo::ICP::ICP<RC, WC>& registration () { return reg; }
    double latency () const { return latency_ms; }

private:
    void check (int rc) { if (rc != ICP_OK) throw std::runtime_error (std::string ("ICPReg: ") + icp_last_error (reg.handle ())); }

    unsigned int width, height, n, m, r;
    float a, c;
    unsigned int max_iterations;
    double angle_threshold;
    double translation_threshold;
    icp::Env env;
    cl_algo::ICP::ICP<RC, WC> reg;
    std::vector<icp_float8> moving, moved;
    double latency_ms;
};

#endif  // OCL_ICP_REG_HPP
