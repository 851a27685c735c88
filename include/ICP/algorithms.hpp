/*! \file ICP/algorithms.hpp
 *  \brief C++ facade with the names of the reference's `cl_algo::ICP` pipeline classes, forwarding
 *         to the MI355X engine's C-ABI (include/icp_amd.h).
 *
 *  A user of nlamprian/ICP's `ICPStep<CR,CW>` / `ICP<CR,CW>` (reference: include/ICP/algorithms.hpp:
 *  1582-2496, src/ICP/algorithms.cpp:3158-4903) switches by replacing the OpenCL plumbing types:
 *
 *      reference                                     here
 *      clutils::CLEnv &env, CLEnvInfo<1> infoRBC,    icp::Env env (device ordinal)
 *        CLEnvInfo<1> infoICP
 *      cl::Memory& get (Memory)                      void*& get (Memory)   (device pointer)
 *      const std::vector<cl::Event>*, cl::Event*     dropped (one in-order HIP stream per object)
 *      Eigen::Matrix3f / Quaternionf / Vector3f      icp::Matrix3f / Quaternionf / Vector3f (PODs)
 *
 *  Everything else keeps its name, argument order, defaults and meaning: init, write, read, buildRBC,
 *  run, getAlpha/setAlpha, getScaling/setScaling, the ICP thresholds, and the public state members
 *  Rk qk tk sk R q t s k hPtrInF hPtrInM hPtrIOT.  Argument errors throw std::runtime_error
 *  (the reference prints and calls exit(), src/ICP/algorithms.cpp:4411-4427).
 */
#ifndef ICP_ALGORITHMS_HPP
#define ICP_ALGORITHMS_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../icp_amd.h"

namespace icp
{
    /*! \brief Replaces the clutils::CLEnv / CLEnvInfo pair: which GPU the object lives on. */
    struct Env { int device; explicit Env (int d = 0) : device (d) {} };

    /*! \brief How an iteration is evaluated (include/icp_amd.h: icp_reduce_mode / icp_power_mode).
     *  FAST (default): single-pass double moments + squared power start — the benchmarked path, one launch per
     *  iteration at the reference's size.  REFERENCE_ORDER: the reference's three reductions in its own order and its
     *  literal power loop — every intermediate restates the reference's arithmetic, four launches per iteration.
     *  Final [q | t, s] of the two agree within 1e-5 relative (|q| = 1, scene scale for t, s itself). */
    enum class Mode : uint8_t { FAST, REFERENCE_ORDER };

    struct Vector3f
    {
        float v[3] = { 0.f, 0.f, 0.f };
        float& operator() (int i) { return v[i]; }
        float operator() (int i) const { return v[i]; }
        float x () const { return v[0]; } float y () const { return v[1]; } float z () const { return v[2]; }
        float norm () const { return std::sqrt ((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
    };

    /*! \brief Per-step, per-stage times of a profiling run (stands in for clutils::ProfilingInfo<N>, Appendix C of the
     *         survey: `operator[]`, `print`, `total`).  ms; stage = icp_stage (search, means, sij, finalize). */
    struct ProfilingInfo
    {
        std::vector<float> ms;      /*!< [step * 4 + stage] */
        float total_ms = 0.f;
        unsigned int steps () const { return (unsigned int) (ms.size () / 4); }
        float operator() (unsigned int step, int stage) const { return ms[(size_t) step * 4 + stage]; }
        double total (int stage) const { double s = 0; for (unsigned int i = 0; i < steps (); ++i) s += (*this) (i, stage); return s; }
        double total () const { return total_ms; }
        void print (const char *title = "ICP") const
        {
            static const char *names[4] = { "search", "means", "sij", "finalize" };
            std::printf (" %s: %u steps, %.3f ms\n %-10s %10s %10s %10s %10s\n", title, steps (), total_ms, "stage", "mean [us]", "min [us]", "max [us]", "total [ms]");
            for (int s = 0; s < 4; ++s) {
                float mn = 1e30f, mx = 0.f;
                for (unsigned int i = 0; i < steps (); ++i) { mn = std::fmin (mn, (*this) (i, s)); mx = std::fmax (mx, (*this) (i, s)); }
                std::printf (" %-10s %10.2f %10.2f %10.2f %10.3f\n", names[s], total (s) / steps () * 1e3, mn * 1e3, mx * 1e3, total (s));
            }
        }
    };

    /*! \brief Unit quaternion stored like Eigen::Quaternionf::coeffs (): x, y, z, w. */
    struct Quaternionf
    {
        float c[4] = { 0.f, 0.f, 0.f, 1.f };
        float x () const { return c[0]; } float y () const { return c[1]; }
        float z () const { return c[2]; } float w () const { return c[3]; }
        Vector3f vec () const { Vector3f r; r.v[0] = c[0]; r.v[1] = c[1]; r.v[2] = c[2]; return r; }
        const float* coeffs () const { return c; }
    };

    /*! \brief Row-major 3x3. */
    struct Matrix3f
    {
        float m[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
        float& operator() (int r, int c) { return m[r * 3 + c]; }
        float operator() (int r, int c) const { return m[r * 3 + c]; }
    };
}

namespace cl_algo
{
namespace ICP
{
    /*! \brief Rotation computation of an ICP step — reference include/ICP/algorithms.hpp:1544. */
    enum class ICPStepConfigT : uint8_t { EIGEN, POWER_METHOD, JACOBI };
    /*! \brief Residual weighting of an ICP step — reference include/ICP/algorithms.hpp:1560. */
    enum class ICPStepConfigW : uint8_t { REGULAR, WEIGHTED };
    /*! \brief Staging buffers to instantiate — reference include/ICP/common.hpp. */
    enum class Staging : uint8_t { NONE, I, O, IO };

    /*! \brief reference include/ICP/algorithms.hpp:52-58 */
    enum class ReduceConfig : uint8_t { MIN, MAX, SUM };
    /*! \brief reference include/ICP/algorithms.hpp:169-174 */
    enum class ScanConfig : uint8_t { INCLUSIVE, EXCLUSIVE };

    /*! \brief Row-wise reduction — mirrors `Reduce<C, T>` (reference include/ICP/algorithms.hpp:83-166,
     *         src/ICP/algorithms.cpp:131-322): MIN over floats, MAX over unsigned ints, SUM over floats (the SUM
     *         reproduces `reduce_sum_f`'s tree bit for bit).  Host staging buffers as with `Staging::IO`:
     *         `write` copies `cols x rows` elements in, `run` reduces every row, `read` returns `rows` results.
     */
    template <ReduceConfig C, typename T = float>
    class Reduce
    {
    public:
        enum class Memory : uint8_t { H_IN, H_OUT, D_IN, D_RED, D_OUT };
        static_assert ((C == ReduceConfig::MAX && sizeof (T) == 4) || C != ReduceConfig::MAX, "MAX reduces 32-bit unsigned integers");

        explicit Reduce (icp::Env _env) : env (_env), h (nullptr), cols (0), rows (0) {}
        Reduce (const Reduce&) = delete;
        Reduce& operator= (const Reduce&) = delete;
        ~Reduce () { if (h) icp_rs_destroy (h); }
        /*! \brief Creates the device buffers (they live until the next init / destruction). */
        void init (unsigned int _cols, unsigned int _rows, Staging = Staging::IO)
        {
            if (_cols == 0 || _rows == 0 || _cols % 4) throw std::runtime_error ("Reduce::init: cols must be a positive multiple of 4");
            if (h) { icp_rs_destroy (h); h = nullptr; }
            const int kind = C == ReduceConfig::MIN ? ICP_RS_MIN_F : C == ReduceConfig::MAX ? ICP_RS_MAX_UI : ICP_RS_SUM_F;
            chk (icp_rs_create (&h, env.device, kind, _cols, _rows));
            cols = _cols; rows = _rows; in.assign ((size_t) cols * rows, T ()); out.assign (rows, T ());
        }
        /*! \brief Device buffer (reference: cl::Memory& get (Memory)): D_IN, or D_OUT = the result of the last run. */
        void* get (Memory mem) { void *p = nullptr; chk (icp_rs_device_ptr (h, mem == Memory::D_OUT || mem == Memory::D_RED ? 1 : 0, &p)); return p; }
        /*! \brief Host -> device (ptr == nullptr: the staging buffer hPtrIn () as it stands). */
        void write (Memory = Memory::D_IN, void *ptr = nullptr, bool = false)
        { if (ptr) std::memcpy (in.data (), ptr, in.size () * sizeof (T)); chk (icp_rs_write (h, in.data ())); }
        /*! \brief Device -> host staging; blocking. */
        void* read (Memory = Memory::H_OUT, bool = true) { chk (icp_rs_read (h, out.data ())); return out.data (); }
        /*! \brief Enqueues the kernels; nothing is allocated or copied. */
        void run () { chk (icp_rs_run (h)); }
        /*! \brief The reference's run (timer): mean microseconds of `reps` runs (HIP events on the object's stream). */
        double run (unsigned int reps, bool) { float us = 0.f; chk (icp_rs_time (h, reps, &us)); return us; }
        T *hPtrIn () { return in.data (); }
        T *hPtrOut () { return out.data (); }

    private:
        void chk (int rc) { if (rc != ICP_OK) throw std::runtime_error (std::string ("Reduce: ") + icp_reduce_scan_last_error ()); }
        icp::Env env;
        icp_rs_handle h;
        unsigned int cols, rows;
        std::vector<T> in, out;
    };

    /*! \brief Row-wise prefix sum of ints — mirrors `Scan<C>` (reference include/ICP/algorithms.hpp:200-290,
     *         src/ICP/algorithms.cpp:403-600). */
    template <ScanConfig C>
    class Scan
    {
    public:
        enum class Memory : uint8_t { H_IN, H_OUT, D_IN, D_SUMS, D_OUT };

        explicit Scan (icp::Env _env) : env (_env), h (nullptr), cols (0), rows (0) {}
        Scan (const Scan&) = delete;
        Scan& operator= (const Scan&) = delete;
        ~Scan () { if (h) icp_rs_destroy (h); }
        void init (unsigned int _cols, unsigned int _rows, Staging = Staging::IO)
        {
            if (_cols == 0 || _rows == 0 || _cols % 4) throw std::runtime_error ("Scan::init: cols must be a positive multiple of 4");
            if (h) { icp_rs_destroy (h); h = nullptr; }
            chk (icp_rs_create (&h, env.device, C == ScanConfig::INCLUSIVE ? ICP_RS_SCAN_INCLUSIVE : ICP_RS_SCAN_EXCLUSIVE, _cols, _rows));
            cols = _cols; rows = _rows; in.assign ((size_t) cols * rows, 0); out.assign ((size_t) cols * rows, 0);
        }
        void* get (Memory mem) { void *p = nullptr; chk (icp_rs_device_ptr (h, mem == Memory::D_OUT ? 1 : 0, &p)); return p; }
        void write (Memory = Memory::D_IN, void *ptr = nullptr, bool = false)
        { if (ptr) std::memcpy (in.data (), ptr, in.size () * sizeof (int32_t)); chk (icp_rs_write (h, in.data ())); }
        void* read (Memory = Memory::H_OUT, bool = true) { chk (icp_rs_read (h, out.data ())); return out.data (); }
        void run () { chk (icp_rs_run (h)); }
        double run (unsigned int reps, bool) { float us = 0.f; chk (icp_rs_time (h, reps, &us)); return us; }

    private:
        void chk (int rc) { if (rc != ICP_OK) throw std::runtime_error (std::string ("Scan: ") + icp_reduce_scan_last_error ()); }
        icp::Env env;
        icp_rs_handle h;
        unsigned int cols, rows;
        std::vector<int32_t> in, out;
    };

    /*! \brief reference include/ICP/algorithms.hpp:1189-1207 */
    enum class ICPTransformConfig : uint8_t { QUATERNION, MATRIX };

    /*! \brief Transforms a set of points — mirrors `ICPTransform<QUATERNION>` (`icpTransform_Quaternion`, reference
     *         include/ICP/algorithms.hpp:1240-1330, src/ICP/algorithms.cpp:2554-2753) and `ICPTransform<MATRIX>`
     *         (`icpTransform_Matrix`, :1348-1430, :2760-2960).  Same staging members (`hPtrInM`, `hPtrInT`, `hPtrOut`);
     *         the transformation is 8 floats `[q | t, s]` (QUATERNION) or a row-major 4x4 (MATRIX). */
    template <ICPTransformConfig C>
    class ICPTransform
    {
    public:
        enum class Memory : uint8_t { H_IN_M, H_IN_T, H_OUT, D_IN_M, D_IN_T, D_OUT };

        explicit ICPTransform (icp::Env _env) : hPtrInM (nullptr), hPtrInT (nullptr), hPtrOut (nullptr), env (_env), h (nullptr), m (0)
        {
            if (icp_create (&h, env.device, ICP_ROT_POWER_METHOD, ICP_W_WEIGHTED) != ICP_OK)
                throw std::runtime_error (std::string ("ICPTransform: ") + icp_last_error (nullptr));
        }
        ICPTransform (const ICPTransform&) = delete;
        ICPTransform& operator= (const ICPTransform&) = delete;
        ~ICPTransform () { if (h) icp_destroy (h); }

        void init (unsigned int _m, Staging = Staging::IO)
        {
            if (_m == 0) throw std::runtime_error ("ICPTransform::init: the set cannot have zero points");
            m = _m; in.assign ((size_t) m * 8, 0.f); out.assign ((size_t) m * 8, 0.f); T.assign (C == ICPTransformConfig::MATRIX ? 16 : 8, 0.f);
            hPtrInM = in.data (); hPtrInT = T.data (); hPtrOut = out.data ();
        }
        void write (Memory mem = Memory::D_IN_M, void *ptr = nullptr, bool = false)
        {
            if (!ptr) return;                       // (data written through the staging pointers is used as it stands)
            if (mem == Memory::D_IN_M) std::memcpy (in.data (), ptr, in.size () * sizeof (float));
            else if (mem == Memory::D_IN_T) std::memcpy (T.data (), ptr, T.size () * sizeof (float));
        }
        void* read (Memory = Memory::H_OUT, bool = true) { return out.data (); }
        void run ()
        {
            const int kind = C == ICPTransformConfig::MATRIX ? ICP_TRANSFORM_MATRIX : ICP_TRANSFORM_QUATERNION;
            if (icp_transform_cloud_ex (h, kind, T.data (), in.data (), out.data (), m) != ICP_OK)
                throw std::runtime_error (std::string ("ICPTransform: ") + icp_last_error (h));
        }

        float *hPtrInM;  /*!< Staging buffer of the set of points. */
        float *hPtrInT;  /*!< Staging buffer of the transformation. */
        float *hPtrOut;  /*!< Staging buffer of the transformed set. */

    private:
        icp::Env env;
        icp_handle h;
        unsigned int m;
        std::vector<float> in, out, T;
    };

    /*! \brief reference include/ICP/algorithms.hpp:576-582 / :943-947 */
    enum class ICPMeanConfig : uint8_t { REGULAR, WEIGHTED };
    enum class ICPSConfig : uint8_t { REGULAR, WEIGHTED };

    /*! \brief Base of the per-kernel classes below — each one a RESIDENT object (include/icp_amd.h: icp_ko_*), wired like the reference's:
     *  `get (Memory)` is a reference to a device pointer; assigned BEFORE `init` (e.g. `mean.get (D_IN_W) = weights.get (D_OUT_W)`) the
     *  object uses that buffer instead of creating its own (reference note include/ICP/algorithms.hpp:2214-2220; wiring
     *  src/ICP/algorithms.cpp:4499-4581), after `init` it is the object's own buffer.  `write` uploads a staging buffer, `run` enqueues
     *  kernels only, `read` downloads.  Errors throw (the reference prints and calls `exit ()`). */
    class KernelClass
    {
    public:
        KernelClass (const KernelClass&) = delete;
        KernelClass& operator= (const KernelClass&) = delete;
        ~KernelClass () { if (ko) icp_ko_destroy (ko); }
    protected:
        explicit KernelClass (icp::Env _env) : env (_env), ko (nullptr) { for (auto &q : dptr) q = nullptr; }
        void chk (int rc, const char *who) { if (rc != ICP_OK) throw std::runtime_error (std::string (who) + ": " + icp_kernel_last_error ()); }
        /*! creates the device object; slots whose pointer was assigned through get () before are adopted, the others become the object's own */
        void create (int kind, uint32_t n, uint32_t aux, float c, int nslots, const char *who)
        {
            void *preset[5];
            for (int s = 0; s < 5; ++s) preset[s] = ko ? nullptr : dptr[s];
            if (ko) { icp_ko_destroy (ko); ko = nullptr; }
            chk (icp_ko_create (&ko, env.device, kind, n, aux, c), who);
            for (int s = 0; s < nslots; ++s) {
                if (preset[s]) chk (icp_ko_adopt (ko, s, preset[s]), who);
                chk (icp_ko_device_ptr (ko, s, &dptr[s]), who);
            }
        }
        void upload (int slot, const void *host, const char *who) { chk (icp_ko_write (ko, slot, host), who); }
        void download (int slot, void *host, const char *who) { chk (icp_ko_read (ko, slot, host), who); }
        void launch (const char *who) { if (!ko) throw std::runtime_error (std::string (who) + ": init has not been called"); chk (icp_ko_run (ko), who); }
        icp::Env env;
        icp_ko_handle ko;
        void *dptr[5];
    };

    /*! \brief Landmark extraction — mirrors `ICPLMs` (reference include/ICP/algorithms.hpp:312-383, kernel `getLMs`
     *         kernels/icp_kernels.cl:63-76): 640 x 480 points in, 128 x 128 landmarks out; `get / init / write / run / read`, `hPtrIn`, `hPtrOut`. */
    class ICPLMs : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN, H_OUT, D_IN, D_OUT };
        explicit ICPLMs (icp::Env _env) : KernelClass (_env), hPtrIn (nullptr), hPtrOut (nullptr) {}
        void*& get (Memory mem) { return dptr[mem == Memory::D_OUT || mem == Memory::H_OUT ? 1 : 0]; }
        void init (Staging = Staging::IO)
        { in.assign ((size_t) 640 * 480 * 8, 0.f); out.assign ((size_t) 16384 * 8, 0.f); hPtrIn = in.data (); hPtrOut = out.data (); create (ICP_KO_LMS, 0, 0, 0.f, 2, "ICPLMs"); }
        /*! uploads the staging buffer (after copying `ptr` into it, if given) — reference `write (D_IN, ptr, block)` */
        void write (Memory = Memory::D_IN, void *ptr = nullptr, bool = false) { if (ptr) std::memcpy (in.data (), ptr, in.size () * sizeof (float)); upload (0, in.data (), "ICPLMs"); }
        void* read (Memory = Memory::H_OUT, bool = true) { download (1, out.data (), "ICPLMs"); return out.data (); }
        void run () { launch ("ICPLMs"); }
        float *hPtrIn, *hPtrOut;
    private:
        std::vector<float> in, out;
    };

    /*! \brief Representatives — mirrors `ICPReps` (reference :397-468, kernel `getReps` kernels/icp_kernels.cl:97-114, grid rule
     *         src/ICP/algorithms.cpp:842-854): `init (nr)` for the reference's 128 x 128 landmarks, `init (nr, m)` for any square set. */
    class ICPReps : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN, H_OUT, D_IN, D_OUT };
        explicit ICPReps (icp::Env _env) : KernelClass (_env), hPtrIn (nullptr), hPtrOut (nullptr), m (16384), nr (0) {}
        void*& get (Memory mem) { return dptr[mem == Memory::D_OUT || mem == Memory::H_OUT ? 1 : 0]; }
        void init (unsigned int _nr, Staging = Staging::IO) { init (_nr, 16384u); }
        void init (unsigned int _nr, unsigned int _m)
        { m = _m; nr = _nr; in.assign ((size_t) m * 8, 0.f); out.assign ((size_t) nr * 8, 0.f); hPtrIn = in.data (); hPtrOut = out.data (); create (ICP_KO_REPS, m, nr, 0.f, 2, "ICPReps"); }
        void write (Memory = Memory::D_IN, void *ptr = nullptr, bool = false) { if (ptr) std::memcpy (in.data (), ptr, in.size () * sizeof (float)); upload (0, in.data (), "ICPReps"); }
        void* read (Memory = Memory::H_OUT, bool = true) { download (1, out.data (), "ICPReps"); return out.data (); }
        void run () { launch ("ICPReps"); }
        float *hPtrIn, *hPtrOut;
    private:
        unsigned int m, nr;
        std::vector<float> in, out;
    };

    /*! \brief Weights and their sum — mirrors `ICPWeights` (reference :485-568, kernels `icpComputeReduceWeights(_WG)`,
     *         `reduce_sum_fd` kernels/icp_kernels.cl:139-329): `w = 100 / (100 + dist)`, the sum as a double. */
    class ICPWeights : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN, H_OUT_W, H_OUT_SUM_W, D_IN, D_OUT_W, D_GW, D_OUT_SUM_W };
        struct dist_id { float dist; uint32_t id; };                  /*!< `rbc_dist_id` (kernels/icp_kernels.cl:34-38) */
        explicit ICPWeights (icp::Env _env) : KernelClass (_env), hPtrIn (nullptr), hPtrOutW (nullptr), hPtrOutSW (&sw), n (0), sw (0.0) {}
        void*& get (Memory mem) { return dptr[(mem == Memory::D_OUT_W || mem == Memory::H_OUT_W) ? 1 : (mem == Memory::D_OUT_SUM_W || mem == Memory::H_OUT_SUM_W) ? 2 : 0]; }
        void init (unsigned int _n, Staging = Staging::IO)
        { n = _n; in.assign (n, dist_id { 0.f, 0u }); W.assign (n, 0.f); hPtrIn = in.data (); hPtrOutW = W.data (); create (ICP_KO_WEIGHTS, n, 0, 0.f, 3, "ICPWeights"); }
        void write (Memory = Memory::D_IN, void *ptr = nullptr, bool = false) { if (ptr) std::memcpy (in.data (), ptr, in.size () * sizeof (dist_id)); upload (0, in.data (), "ICPWeights"); }
        void* read (Memory mem = Memory::H_OUT_SUM_W, bool = true)
        { if (mem == Memory::H_OUT_W) { download (1, W.data (), "ICPWeights"); return W.data (); } download (2, &sw, "ICPWeights"); return &sw; }
        void run () { launch ("ICPWeights"); }
        dist_id *hPtrIn; float *hPtrOutW; double *hPtrOutSW;
    private:
        unsigned int n; double sw;
        std::vector<dist_id> in; std::vector<float> W;
    };

    /*! \brief Set means — mirrors `ICPMean<REGULAR>` / `ICPMean<WEIGHTED>` (reference :625-843, kernels `icpMean`,
     *         `icpMean_Weighted`, `icpGMean` kernels/icp_kernels.cl:371-566): output `[mean_F, 0 | mean_M, 0]`. */
    template <ICPMeanConfig C>
    class ICPMean : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN_F, H_IN_M, H_IN_W, H_IN_SUM_W, H_OUT, D_IN_F, D_IN_M, D_IN_W, D_IN_SUM_W, D_GM, D_OUT };
        explicit ICPMean (icp::Env _env) : KernelClass (_env), hPtrInF (nullptr), hPtrInM (nullptr), hPtrInW (nullptr), hPtrInSW (&sw), hPtrOut (mean), n (0), sw (1.0) { std::memset (mean, 0, sizeof mean); }
        static int slot_of (Memory mem)
        {
            switch (mem) {
                case Memory::H_IN_M: case Memory::D_IN_M: return 1;
                case Memory::H_IN_W: case Memory::D_IN_W: return 2;
                case Memory::H_IN_SUM_W: case Memory::D_IN_SUM_W: return 3;
                case Memory::H_OUT: case Memory::D_OUT: return 4;
                default: return 0;
            }
        }
        void*& get (Memory mem) { return dptr[slot_of (mem)]; }
        void init (unsigned int _n, Staging = Staging::IO)
        {
            n = _n; F.assign ((size_t) n * 8, 0.f); M.assign ((size_t) n * 8, 0.f); W.assign (n, 0.f); hPtrInF = F.data (); hPtrInM = M.data (); hPtrInW = W.data ();
            create (C == ICPMeanConfig::WEIGHTED ? ICP_KO_MEAN_WEIGHTED : ICP_KO_MEAN, n, 0, 0.f, 5, "ICPMean");
        }
        void write (Memory mem = Memory::D_IN_F, void *ptr = nullptr, bool = false)
        {
            const int s = slot_of (mem);
            void *stage = s == 0 ? (void *) F.data () : s == 1 ? (void *) M.data () : s == 2 ? (void *) W.data () : (void *) &sw;
            const size_t bytes = s <= 1 ? F.size () * sizeof (float) : s == 2 ? W.size () * sizeof (float) : sizeof sw;
            if (s > 3) return;
            if (ptr) std::memcpy (stage, ptr, bytes);
            upload (s, stage, "ICPMean");
        }
        void* read (Memory = Memory::H_OUT, bool = true) { download (4, mean, "ICPMean"); return mean; }
        void run () { launch ("ICPMean"); }
        float *hPtrInF, *hPtrInM, *hPtrInW; double *hPtrInSW; float *hPtrOut;
    private:
        unsigned int n; double sw; float mean[8];
        std::vector<float> F, M, W;
    };

    /*! \brief Deviations from the means — mirrors `ICPDevs` (reference :867-940, kernel `icpSubtractMean` kernels/icp_kernels.cl:588-602). */
    class ICPDevs : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN_F, H_IN_M, H_IN_MEAN, H_OUT_DEV_F, H_OUT_DEV_M, D_IN_F, D_IN_M, D_IN_MEAN, D_OUT_DEV_F, D_OUT_DEV_M };
        explicit ICPDevs (icp::Env _env) : KernelClass (_env), hPtrInF (nullptr), hPtrInM (nullptr), hPtrInMean (mean), hPtrOutDevF (nullptr), hPtrOutDevM (nullptr), n (0) { std::memset (mean, 0, sizeof mean); }
        static int slot_of (Memory mem)
        {
            switch (mem) {
                case Memory::H_IN_M: case Memory::D_IN_M: return 1;
                case Memory::H_IN_MEAN: case Memory::D_IN_MEAN: return 2;
                case Memory::H_OUT_DEV_F: case Memory::D_OUT_DEV_F: return 3;
                case Memory::H_OUT_DEV_M: case Memory::D_OUT_DEV_M: return 4;
                default: return 0;
            }
        }
        void*& get (Memory mem) { return dptr[slot_of (mem)]; }
        void init (unsigned int _n, Staging = Staging::IO)
        { n = _n; F.assign ((size_t) n * 8, 0.f); M.assign ((size_t) n * 8, 0.f); DF.assign ((size_t) n * 4, 0.f); DM.assign ((size_t) n * 4, 0.f);
          hPtrInF = F.data (); hPtrInM = M.data (); hPtrOutDevF = DF.data (); hPtrOutDevM = DM.data (); create (ICP_KO_DEVS, n, 0, 0.f, 5, "ICPDevs"); }
        void write (Memory mem = Memory::D_IN_F, void *ptr = nullptr, bool = false)
        {
            const int s = slot_of (mem);
            if (s > 2) return;
            void *stage = s == 0 ? (void *) F.data () : s == 1 ? (void *) M.data () : (void *) mean;
            if (ptr) std::memcpy (stage, ptr, s == 2 ? sizeof mean : F.size () * sizeof (float));
            upload (s, stage, "ICPDevs");
        }
        void* read (Memory mem = Memory::H_OUT_DEV_F, bool = true)
        { if (mem == Memory::H_OUT_DEV_M) { download (4, DM.data (), "ICPDevs"); return DM.data (); } download (3, DF.data (), "ICPDevs"); return DF.data (); }
        void run () { launch ("ICPDevs"); }
        float *hPtrInF, *hPtrInM, *hPtrInMean, *hPtrOutDevF, *hPtrOutDevM;
    private:
        unsigned int n; float mean[8];
        std::vector<float> F, M, DF, DM;
    };

    /*! \brief Sums of products — mirrors `ICPS<REGULAR>` / `ICPS<WEIGHTED>` (reference :976-1183, kernels `icpSijProducts(_Weighted)`
     *         kernels/icp_kernels.cl:633-743 + `reduce_sum_f` twice): `S` row-major (a = moving, b = fixed), then the sums of
     *         squares of the fixed and of the moving deviations (kernel order, kernels/icp_kernels.cl:669-670). */
    template <ICPSConfig C>
    class ICPS : public KernelClass
    {
    public:
        enum class Memory : uint8_t { H_IN_DEV_M, H_IN_DEV_F, H_IN_W, H_OUT, D_IN_DEV_M, D_IN_DEV_F, D_IN_W, D_SIJ, D_OUT };
        explicit ICPS (icp::Env _env) : KernelClass (_env), hPtrInDevM (nullptr), hPtrInDevF (nullptr), hPtrInW (nullptr), hPtrOut (S), m (0), c (1e-6f) { std::memset (S, 0, sizeof S); }
        static int slot_of (Memory mem)
        {
            switch (mem) {
                case Memory::H_IN_DEV_F: case Memory::D_IN_DEV_F: return 1;
                case Memory::H_IN_W: case Memory::D_IN_W: return 2;
                case Memory::H_OUT: case Memory::D_OUT: return 3;
                default: return 0;
            }
        }
        void*& get (Memory mem) { return dptr[slot_of (mem)]; }
        void init (unsigned int _m, float _c, Staging = Staging::IO)
        { m = _m; c = _c; DM.assign ((size_t) m * 4, 0.f); DF.assign ((size_t) m * 4, 0.f); W.assign (m, 0.f); hPtrInDevM = DM.data (); hPtrInDevF = DF.data (); hPtrInW = W.data ();
          create (C == ICPSConfig::WEIGHTED ? ICP_KO_S_WEIGHTED : ICP_KO_S, m, 0, c, 4, "ICPS"); }
        void write (Memory mem = Memory::D_IN_DEV_M, void *ptr = nullptr, bool = false)
        {
            const int s = slot_of (mem);
            if (s > 2) return;
            void *stage = s == 0 ? (void *) DM.data () : s == 1 ? (void *) DF.data () : (void *) W.data ();
            if (ptr) std::memcpy (stage, ptr, (s == 2 ? W.size () : DM.size ()) * sizeof (float));
            upload (s, stage, "ICPS");
        }
        void* read (Memory = Memory::H_OUT, bool = true) { download (3, S, "ICPS"); return S; }
        void run () { launch ("ICPS"); }
        float getScaling () { return c; }
        void setScaling (float _c) { c = _c; if (ko) chk (icp_ko_set_scaling (ko, c), "ICPS"); }
        float *hPtrInDevM, *hPtrInDevF, *hPtrInW, *hPtrOut;
    private:
        unsigned int m; float c; float S[11];
        std::vector<float> DM, DF, W;
    };

    /*! \brief Incremental transformation from S and the set means — mirrors `ICPPowerMethod` (reference
     *         include/ICP/algorithms.hpp:1451-1537, src/ICP/algorithms.cpp:2966-3150, kernel `icpPowerMethod`
     *         kernels/icp_kernels.cl:977-1054): `init`, `write (D_IN_S | D_IN_MEAN)`, `run`, `read (H_OUT_T_K)` and the staging
     *         members `hPtrInS` (11 floats), `hPtrInMean` (2 x float4), `hPtrOutTk` (2 x float4: `[qk | tk, sk]`).  The
     *         reference's known-answer test drives this class (tests/testsICP.cpp:988-1052).  `mode` selects the loop the
     *         engine runs: the reference's literal one (default here, as in the test) or the squared start of the benchmarked
     *         path; `setRotation (ICPStepConfigT::EIGEN)` evaluates the SVD branch instead. */
    class ICPPowerMethod
    {
    public:
        enum class Memory : uint8_t { H_IN_S, H_IN_MEAN, H_OUT_T_K, D_IN_S, D_IN_MEAN, D_OUT_T_K };

        explicit ICPPowerMethod (icp::Env _env, icp::Mode _mode = icp::Mode::REFERENCE_ORDER)
            : hPtrInS (S), hPtrInMean (mean), hPtrOutTk (Tk), iterations (0), env (_env), mode (_mode), rot (ICP_ROT_POWER_METHOD)
        { std::memset (S, 0, sizeof S); std::memset (mean, 0, sizeof mean); std::memset (Tk, 0, sizeof Tk); std::memset (Rk, 0, sizeof Rk); }
        void init (Staging = Staging::IO) {}
        void setMode (icp::Mode _mode) { mode = _mode; }
        void setRotation (ICPStepConfigT r) { rot = r == ICPStepConfigT::EIGEN ? ICP_ROT_EIGEN : ICP_ROT_POWER_METHOD; }
        /*! \brief Host -> staging (ptr == nullptr: the staging buffer as it stands); the upload happens with `run`. */
        void write (Memory mem = Memory::D_IN_S, void *ptr = nullptr, bool = false)
        {
            if (!ptr) return;
            if (mem == Memory::D_IN_S) std::memcpy (S, ptr, sizeof S);
            else if (mem == Memory::D_IN_MEAN) std::memcpy (mean, ptr, sizeof mean);
        }
        void* read (Memory = Memory::H_OUT_T_K, bool = true) { return Tk; }
        /*! \brief One wave of the engine's rotation solver on the device; blocking. */
        void run ()
        {
            if (icp_power_method (env.device, rot, mode == icp::Mode::FAST ? ICP_POWER_SQUARED : ICP_POWER_LITERAL, S, mean, Tk, Rk, &iterations) != ICP_OK)
                throw std::runtime_error (std::string ("ICPPowerMethod: ") + icp_last_error (nullptr));
        }

        float *hPtrInS;     /*!< Staging buffer of the sums of products. */
        float *hPtrInMean;  /*!< Staging buffer of the fixed and moving set means. */
        float *hPtrOutTk;   /*!< Staging buffer of the incremental parameters. */
        uint32_t iterations;  /*!< power-method loop trips of the last run (the reference's comment: 56 on its test vector) */
        const float *rotation () const { return Rk; }   /*!< row-major Rk of the last run */

    private:
        icp::Env env;
        icp::Mode mode;
        int rot;
        float S[11], mean[8], Tk[8], Rk[9];
    };

    /*! \brief One ICP iteration — mirrors the four specialisations of the reference's
     *         `ICPStep<CR, CW>` (include/ICP/algorithms.hpp:1613, 1825, 2038, 2234).
     */
    template <ICPStepConfigT CR, ICPStepConfigW CW>
    class ICPStep
    {
    public:
        /*! \brief reference include/ICP/algorithms.hpp:2241-2267 */
        enum class Memory : uint8_t { H_IN_F, H_IN_M, H_IO_T, D_IN_F, D_IN_M, D_IO_T };

        ICPStep (icp::Env _env, icp::Mode _mode = icp::Mode::FAST) : env (_env), h (nullptr), a (1e2f), c (1e-6f), m (0), nr (0)
        {
            static_assert (CR != ICPStepConfigT::JACOBI, "JACOBI is a \\todo in the reference as well");
            int rc = icp_create (&h, env.device, CR == ICPStepConfigT::POWER_METHOD ? ICP_ROT_POWER_METHOD : ICP_ROT_EIGEN,
                                 CW == ICPStepConfigW::WEIGHTED ? ICP_W_WEIGHTED : ICP_W_REGULAR);
            if (rc != ICP_OK) throw std::runtime_error (std::string ("ICPStep: ") + icp_last_error (nullptr));
            hPtrInF = hPtrInM = hPtrIOT = nullptr; sk = 1.f; s = 1.f;
            setMode (_mode);
        }
        ICPStep (const ICPStep&) = delete;
        ICPStep& operator= (const ICPStep&) = delete;
        virtual ~ICPStep () { if (h) icp_destroy (h); }

        /*! \brief Device pointer of a buffer (reference: cl::Memory& get (Memory), algorithms.cpp:4366-4383).
         *  \note Assigning a pointer before `init` makes the object adopt that buffer, as the reference does
         *        (src/ocl_icp_reg.cpp:111-113); the buffer stays the caller's.  After `init` the references hold the
         *        buffers in use (the engine's own ones unless adopted), valid until the next `init`. */
        void*& get (Memory mem)
        {
            switch (mem)
            {
                case Memory::D_IN_F: return dPtr[0];
                case Memory::D_IN_M: return dPtr[1];
                case Memory::D_IO_T: return dPtr[2];
                default: throw std::runtime_error ("ICPStep::get: host staging buffers are reached through hPtrInF/hPtrInM/hPtrIOT");
            }
        }

        /*! \brief reference include/ICP/algorithms.hpp:2271, src/ICP/algorithms.cpp:4403-4582 */
        void init (unsigned int _m, unsigned int _nr, float _a = 1e2f, float _c = 1e-6f, Staging _staging = Staging::IO)
        { init_ (_m, _nr, _a, _c, 40, 0.001, 0.01, _staging); }

        /*! \brief reference include/ICP/algorithms.hpp:2273, src/ICP/algorithms.cpp:4596-4622 */
        void write (Memory mem = Memory::D_IN_F, void *ptr = nullptr, bool block = false)
        {
            if (!(staging == Staging::I || staging == Staging::IO)) return;
            int which; float *stage; size_t n = (size_t) m * 8;
            switch (mem)
            {
                case Memory::D_IN_F: which = ICP_MEM_F; stage = hPtrInF; break;
                case Memory::D_IN_M: which = ICP_MEM_M; stage = hPtrInM; break;
                case Memory::D_IO_T: which = ICP_MEM_T; stage = hPtrIOT; n = 8; break;
                default: return;
            }
            if (ptr != nullptr) std::memcpy (stage, ptr, n * sizeof (float));
            check (icp_write (h, which, stage, block ? 1 : 0));
        }

        /*! \brief reference include/ICP/algorithms.hpp:2275, src/ICP/algorithms.cpp:4634-4649 */
        void* read (Memory mem = Memory::H_IO_T, bool block = true)
        {
            (void) block;
            if (!(staging == Staging::O || staging == Staging::IO)) return nullptr;
            if (mem != Memory::H_IO_T) return nullptr;
            check (icp_read (h, ICP_MEM_T, hPtrIOT, 8 * sizeof (float)));
            return hPtrIOT;
        }

        /*! \brief reference include/ICP/algorithms.hpp:2277, src/ICP/algorithms.cpp:4655-4660 */
        void buildRBC () { check (icp_build_rbc (h)); }

        /*! \brief One iteration; updates Rk qk tk sk R q t s like the reference (src/ICP/algorithms.cpp:4670-4698). */
        void run (bool config = false) { check (icp_step (h, config ? 1 : 0)); pull (); }

        /*! \brief Switches between the benchmarked evaluation and the reference-order one (see icp::Mode). */
        void setMode (icp::Mode _mode)
        {
            mode = _mode;
            check (icp_set_reduce_mode (h, mode == icp::Mode::FAST ? ICP_REDUCE_FUSED : ICP_REDUCE_REFERENCE_ORDER));
            check (icp_set_power_mode (h, mode == icp::Mode::FAST ? ICP_POWER_SQUARED : ICP_POWER_LITERAL));
        }
        icp::Mode getMode () const { return mode; }

        float getAlpha () { return a; }
        void setAlpha (float _a) { check (icp_set_alpha (h, _a)); a = _a; }
        float getScaling () { return c; }
        void setScaling (float _c) { check (icp_set_scaling (h, _c)); c = _c; }

        float *hPtrInF;  /*!< Staging buffer of the fixed set (reference: mapped H_IN_F). */
        float *hPtrInM;  /*!< Staging buffer of the moving set. */
        float *hPtrIOT;  /*!< Staging buffer of [q | t, s]. */

        icp::Matrix3f Rk; icp::Quaternionf qk; icp::Vector3f tk; float sk;   /*!< iteration k */
        icp::Matrix3f R;  icp::Quaternionf q;  icp::Vector3f t;  float s;    /*!< up to iteration k */

        icp_handle handle () { return h; }

    protected:
        void check (int rc) { if (rc != ICP_OK) throw std::runtime_error (std::string ("ICP: ") + icp_last_error (h)); }

        void init_ (unsigned int _m, unsigned int _nr, float _a, float _c, unsigned int max_it, double ang, double tra, Staging _staging)
        {
            m = _m; nr = _nr; a = _a; c = _c; staging = _staging;
            // A pointer found in dPtr[] is the caller's only if it is not one this object fetched from the engine after
            // an earlier init: those belong to the handle, and icp_init is about to free them (re-init).
            for (int i = 0; i < 2; ++i) { user[i] = dPtr[i] != nullptr && dPtr[i] != engine[i]; if (!user[i]) dPtr[i] = nullptr; }
            check (icp_init (h, m, nr, a, c, max_it, ang, tra));
            if (user[0]) check (icp_adopt_device_buffer (h, ICP_MEM_F, dPtr[0]));
            if (user[1]) check (icp_adopt_device_buffer (h, ICP_MEM_M, dPtr[1]));
            check (icp_device_ptr (h, ICP_MEM_F, &dPtr[0]));
            check (icp_device_ptr (h, ICP_MEM_M, &dPtr[1]));
            check (icp_device_ptr (h, ICP_MEM_T, &dPtr[2]));
            for (int i = 0; i < 2; ++i) engine[i] = user[i] ? nullptr : dPtr[i];
            stageF.assign (staging == Staging::I || staging == Staging::IO ? (size_t) m * 8 : 0, 0.f);
            stageM.assign (stageF.size (), 0.f);
            hPtrInF = stageF.empty () ? nullptr : stageF.data ();
            hPtrInM = stageM.empty () ? nullptr : stageM.data ();
            const float T0[8] = { 0, 0, 0, 1, 0, 0, 0, 1 };
            std::memcpy (stageT, T0, sizeof T0); hPtrIOT = stageT;
            R = icp::Matrix3f (); q = icp::Quaternionf (); t = icp::Vector3f (); s = 1.f;
        }

        void pull ()
        {
            icp_state_t st; check (icp_state (h, &st));
            std::memcpy (R.m, st.R, sizeof st.R); std::memcpy (q.c, st.q, sizeof st.q); std::memcpy (t.v, st.t, sizeof st.t); s = st.s;
            std::memcpy (Rk.m, st.Rk, sizeof st.Rk); std::memcpy (qk.c, st.qk, sizeof st.qk); std::memcpy (tk.v, st.tk, sizeof st.tk); sk = st.sk;
            std::memcpy (hPtrIOT, st.q, 16); std::memcpy (hPtrIOT + 4, st.t, 12); hPtrIOT[7] = st.s;
            k_ = st.k;
        }

        icp::Env env;
        icp_handle h;
        Staging staging = Staging::IO;
        float a, c;
        unsigned int m, nr;
        unsigned int k_ = 0;
        void *dPtr[3] = { nullptr, nullptr, nullptr };
        void *engine[2] = { nullptr, nullptr };   // F / M buffers owned by the handle (as fetched after the last init)
        bool user[2] = { false, false };          // F / M adopted from the caller at the last init
        icp::Mode mode = icp::Mode::FAST;
        std::vector<float> stageF, stageM;
        float stageT[8];
    };

    /*! \brief The iterative registration — mirrors `ICP<CR, CW>` (include/ICP/algorithms.hpp:2433-2496,
     *         src/ICP/algorithms.cpp:4750-4903).
     */
    template <ICPStepConfigT CR, ICPStepConfigW CW>
    class ICP : public ICPStep<CR, CW>
    {
    public:
        ICP (icp::Env _env, icp::Mode _mode = icp::Mode::FAST) : ICPStep<CR, CW> (_env, _mode), k (0), max_iterations (40), angle_threshold (0.001), translation_threshold (0.01) {}

        void init (unsigned int _m, unsigned int _nr, float _a = 1e2f, float _c = 1e-6f, unsigned int _max_iterations = 40,
                   double _angle_threshold = 0.001, double _translation_threshold = 0.01, Staging _staging = Staging::IO)
        {
            max_iterations = _max_iterations; angle_threshold = _angle_threshold; translation_threshold = _translation_threshold;
            this->init_ (_m, _nr, _a, _c, _max_iterations, _angle_threshold, _translation_threshold, _staging);
        }

        void buildRBC () { ICPStep<CR, CW>::buildRBC (); k = 0; }

        /*! \brief Blocking; iterates until check () stops (src/ICP/algorithms.cpp:4806-4834). */
        void run ()
        {
            this->check (icp_run (this->h, nullptr));
            this->pull ();                            // (the state arrived in pinned host memory with the run's graph: no further copy)
            k = this->k_;
        }

        /*! \brief The reference's profiling run, `double run (clutils::GPUTimer<period>&)` (include/ICP/algorithms.hpp:
         *         2482-2494): exactly 40 steps (or `steps`), no convergence test, per-step / per-stage table; prints it
         *         like `steps.print ("ICP")` there and returns the total time in ms. */
        double run (icp::ProfilingInfo &info, unsigned int steps = 40, bool print = true)
        {
            info.ms.assign ((size_t) steps * 4, 0.f);
            this->check (icp_profile_run (this->h, steps, info.ms.data (), &info.total_ms));
            this->pull ();
            k = this->k_;
            if (print) info.print ("ICP");
            return info.total ();
        }

        unsigned int getMaxIterations () { return max_iterations; }
        void setMaxIterations (unsigned int n) { this->check (icp_set_max_iterations (this->h, n)); max_iterations = n; }
        double getAngleThreshold () { return angle_threshold; }
        void setAngleThreshold (double d) { this->check (icp_set_angle_threshold (this->h, d)); angle_threshold = d; }
        double getTranslationThreshold () { return translation_threshold; }
        void setTranslationThreshold (double d) { this->check (icp_set_translation_threshold (this->h, d)); translation_threshold = d; }

        unsigned int k;  /*!< iterations executed (reference: ICP::k, include/ICP/algorithms.hpp:2462) */

    protected:
        unsigned int max_iterations;
        double angle_threshold;
        double translation_threshold;
    };

    /*! \brief Frame-to-frame registration of a sequence of 640 x 480 clouds ("real-time frame-to-frame registration",
     *         reference README.md:4; per pair the demo's flow src/ocl_icp_reg.cpp:128-172): every frame is registered against the
     *         previous one, whose landmarks stay on the device.  `submit` only enqueues (the band of the frame that `getLMs`
     *         reads is uploaded and the landmarks extracted on a copy stream; buildRBC + a host-driven checked run, consecutive frames on
     *         two streams gated on the device: see icp_track_submit in icp_amd.h), `collect` blocks for
     *         the oldest frame in flight and updates `k q t s`; up to four frames may be in flight, so the next frame's upload
     *         overlaps the current registration.  `staging (slot)` hands out the engine's two pinned frame buffers (the
     *         reference's mapped `hPtrInF` / `hPtrInM`): a capture loop that writes there and calls `submit (staging (slot))`
     *         uploads by DMA without a host copy.  The reference has no such class; it is its `ICPReg` loop made resident.
     */
    template <ICPStepConfigT CR, ICPStepConfigW CW>
    class ICPTrack : public ICP<CR, CW>
    {
    public:
        ICPTrack (icp::Env _env, icp::Mode _mode = icp::Mode::FAST, bool _warm_start = false) : ICP<CR, CW> (_env, _mode), warm_start (_warm_start), registered (false) {}
        /*! \brief Sizes and parameters of the demo: 16384 landmarks, 256 representatives (src/ocl_icp_reg.cpp:82-88). */
        void init (float _a = 2e2f, float _c = 1e-6f, unsigned int _max_iterations = 40, double _angle_threshold = 0.001, double _translation_threshold = 0.01)
        { ICP<CR, CW>::init (16384, 256, _a, _c, _max_iterations, _angle_threshold, _translation_threshold, Staging::NONE); }
        float* staging (unsigned int slot) { void *p = nullptr; this->check (icp_track_staging (this->h, slot, &p)); return static_cast<float *> (p); }
        void submit (const void *cloud) { this->check (icp_track_submit (this->h, cloud, warm_start ? 1 : 0)); }
        /*! \brief Blocks for the oldest frame in flight; false for the first frame of a sequence (nothing to register against). */
        bool collect ()
        {
            uint32_t kk = 0; int reg = 0; float T[8];
            this->check (icp_track_collect (this->h, &kk, T, &reg));
            registered = reg != 0;
            if (registered) { this->k = kk; std::memcpy (this->q.c, T, 16); std::memcpy (this->t.v, T + 4, 12); this->s = T[7]; }
            return registered;
        }
        bool next (const void *cloud) { submit (cloud); return collect (); }
        void reset () { this->check (icp_track_reset (this->h)); }
        bool warm_start;   /*!< start every registration from the previous hop's transform instead of the identity */
        bool registered;   /*!< the last collected frame had a predecessor */
    };
}
}

#endif  // ICP_ALGORITHMS_HPP
