// icp_synth.cpp — deterministic synthetic RGB-D landmark pairs (host only).
//
// The reference's sample clouds data/kg_pc8d_{1,2}.bin are absent from the checkout
// (.MISSING_LARGE_BLOBS); this generator produces the "kg-like" pair of SURVEY.md §8d:
//   * pinhole model of the Kinect grabber (f = 595, cx = 319.5, cy = 239.5;
//     src/kinect_frame_grabber.cpp:252-255), point layout [x y z 1 r g b 1], mm / [0,1];
//   * a smooth depth field with ridges and a procedural texture on a side x side grid;
//   * the moving frame samples the same scene at a half-cell offset and is moved rigidly
//     (rotation about `axis`, translation `t`), plus Gaussian geometric / colour noise.
// RNG: splitmix64 -> xoshiro256**.  Deterministic in (seed, side, parameters) on one libm.
#include "../../include/icp_amd.h"
#include "icp_cguard.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {

struct rng {
    uint64_t s[4];
    static uint64_t splitmix (uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit rng (uint64_t seed) { for (auto &v : s) v = splitmix (seed); }
    static uint64_t rotl (uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next ()
    {
        uint64_t r = rotl (s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl (s[3], 45);
        return r;
    }
    double uniform () { return (double) (next () >> 11) * (1.0 / 9007199254740992.0); }
    double normal ()
    {
        double u1 = uniform (), u2 = uniform ();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt (-2.0 * std::log (u1)) * std::cos (2.0 * M_PI * u2);
    }
};

void scene (double u, double v, double G, double *xyz, double *rgb)
{
    const double tau = 2.0 * M_PI;
    double ridge = std::fabs (std::sin (tau * 3.0 * (u + 0.5 * v) / G));
    double z = 1500.0 + 400.0 * std::sin (tau * u / G * 1.5) * std::cos (tau * v / G) + 150.0 * ridge;
    xyz[0] = (u * 640.0 / G - 319.5) * z / 595.0;
    xyz[1] = (v * 480.0 / G - 239.5) * z / 595.0;
    xyz[2] = z;
    double chk = (double) (((int) std::floor (8.0 * u / G) + (int) std::floor (8.0 * v / G)) & 1);
    double r = 0.5 + 0.35 * std::sin (tau * 2.0 * u / G + 1.0) * std::cos (tau * v / G) + 0.1 * chk;
    double g = 0.5 + 0.35 * std::cos (tau * 3.0 * v / G + 0.5) - 0.1 * chk;
    double b = 0.5 + 0.30 * std::sin (tau * (u + 2.0 * v) / G);
    rgb[0] = std::fmin (1.0, std::fmax (0.0, r));
    rgb[1] = std::fmin (1.0, std::fmax (0.0, g));
    rgb[2] = std::fmin (1.0, std::fmax (0.0, b));
}

// The reference's second example scene, data/kg_pc8d_wall (data/README.md:11-16): "non-salient surface geometry" — a wall.  Here: a plane
// through (0, 0, 600) mm (a wall at arm's length: landmarks 5 mm apart, so that a patch edge outweighs a few landmark spacings at a = 2e2) with the normal (0.25, 0.10, -1) / |.|, seen by the same pinhole camera, a millimetre of roughness that is part of
// the SURFACE (both frames see the same bumps), and the procedural texture.  Geometry alone cannot tell where on the wall a point is.
const double WALL_N[3] = { 0.25 / 1.0356157588603989, 0.10 / 1.0356157588603989, -1.0 / 1.0356157588603989 };
const double WALL_Z0 = 600.0;

void scene_wall (double u, double v, double G, double *xyz, double *rgb)
{
    // the texture: what hangs on a wall — patches of 1/16 of the view with a colour of their own (a hash of the patch), sharp edges, a
    // little shading across each: an edge between two patches is worth about one landmark spacing in the metric at a = 2e2
    {
        const long cu = (long) std::floor (16.0 * u / G), cv = (long) std::floor (16.0 * v / G);
        uint64_t hsh = (uint64_t) (cu + 1000) * 0x9E3779B97F4A7C15ull ^ (uint64_t) (cv + 1000) * 0xC2B2AE3D27D4EB4Full;
        for (int k = 0; k < 3; ++k) {
            hsh = rng::splitmix (hsh);
            const double base = 0.1 + 0.8 * (double) (hsh >> 11) * (1.0 / 9007199254740992.0);
            const double fu = 16.0 * u / G - (double) cu, fv = 16.0 * v / G - (double) cv;
            rgb[k] = std::fmin (1.0, std::fmax (0.0, base + 0.05 * (fu - 0.5) + 0.03 * (fv - 0.5)));
        }
    }
    const double rx = (u * 640.0 / G - 319.5) / 595.0, ry = (v * 480.0 / G - 239.5) / 595.0;
    const double tau = 2.0 * M_PI;
    const double bump = 1.0 * std::sin (tau * 11.0 * u / G) * std::sin (tau * 9.0 * v / G);      // +- 1 mm along the viewing ray
    const double z = WALL_N[2] * WALL_Z0 / (WALL_N[0] * rx + WALL_N[1] * ry + WALL_N[2]) + bump;
    xyz[0] = rx * z; xyz[1] = ry * z; xyz[2] = z;
}

void rotation (double deg, const float *axis, double R[9])
{
    double n = std::sqrt ((double) axis[0] * axis[0] + (double) axis[1] * axis[1] + (double) axis[2] * axis[2]);
    double x = axis[0] / n, y = axis[1] / n, z = axis[2] / n;
    double th = deg * M_PI / 180.0, c = std::cos (th), s = std::sin (th), C = 1.0 - c;
    R[0] = c + x * x * C;     R[1] = x * y * C - z * s; R[2] = x * z * C + y * s;
    R[3] = y * x * C + z * s; R[4] = c + y * y * C;     R[5] = y * z * C - x * s;
    R[6] = z * x * C - y * s; R[7] = z * y * C + x * s; R[8] = c + z * z * C;
}

}  // namespace

extern "C" int icp_synth_pair (uint64_t seed, uint32_t side, float rot_deg, const float *axis3, const float *t3,
                               float noise_mm, float noise_rgb, float zero_fraction, float *F, float *M) try
{
    if (!F || !M || side == 0 || !axis3 || !t3) return ICP_EINVAL;
    rng g (seed);
    double R[9]; rotation (rot_deg, axis3, R);
    const double G = (double) side;
    for (uint32_t v = 0; v < side; ++v)
        for (uint32_t u = 0; u < side; ++u) {
            size_t i = (size_t) v * side + u;
            double p[3], c[3];
            scene ((double) u, (double) v, G, p, c);
            float *f = F + i * 8;
            f[0] = (float) p[0]; f[1] = (float) p[1]; f[2] = (float) p[2]; f[3] = 1.f;
            f[4] = (float) c[0]; f[5] = (float) c[1]; f[6] = (float) c[2]; f[7] = 1.f;
            scene ((double) u + 0.5, (double) v + 0.5, G, p, c);
            double q[3];
            for (int k = 0; k < 3; ++k)
                q[k] = R[k * 3] * p[0] + R[k * 3 + 1] * p[1] + R[k * 3 + 2] * p[2] + (double) t3[k] + noise_mm * g.normal ();
            float *mo = M + i * 8;
            mo[0] = (float) q[0]; mo[1] = (float) q[1]; mo[2] = (float) q[2]; mo[3] = 1.f;
            for (int k = 0; k < 3; ++k) {
                double cc = c[k] + noise_rgb * g.normal ();
                mo[4 + k] = (float) std::fmin (1.0, std::fmax (0.0, cc));
            }
            mo[7] = 1.f;
            // invalid points: zero coordinates, as the Kinect pipeline leaves them (kernels/icp_kernels.cl:50-51)
            double zf = g.uniform (), zm = g.uniform ();
            if (zf < zero_fraction) { f[0] = f[1] = f[2] = 0.f; f[4] = f[5] = f[6] = 0.f; }
            if (zm < zero_fraction) { mo[0] = mo[1] = mo[2] = 0.f; mo[4] = mo[5] = mo[6] = 0.f; }
        }
    return ICP_OK;
}
ICP_CATCH_ALL

// Invalid pixels as the Kinect grabber leaves them (src/kinect_frame_grabber.cpp:246-262: depth 0 -> x = y = z = 0, the colour is
// written regardless) and as getLMs picks them (kernels/icp_kernels.cl:49-50), punched into a width x height float8 grid in place.
//   pattern 0  scattered: every point on its own with probability `fraction`;
//   pattern 1  contiguous: a band along the left edge (a quarter of the fraction: the shadow of the projector's baseline) and
//              random ellipses (depth shadows, absorbing surfaces) until `fraction` of the points is covered;
//   keep_rgb   != 0: the colour stays (a real frame); 0: zeroed too — all invalid points identical, the degenerate case in which
//              one representative's list holds every one of them.
// scene 0: the curved scene of icp_synth_pair (rotation about axis3 through the origin, translation t3);
// scene 1: the wall — the motion is IN the wall's plane: a rotation by rot_deg about the plane's normal through its centre (axis3 is
//          ignored) and the in-plane part of t3; *T_true (may be NULL) receives the ground truth [q | t, 1] that maps the moving frame
//          onto the fixed one, in the engine's convention.
extern "C" int icp_synth_pair_scene (uint64_t seed, uint32_t side, int scene_kind, float rot_deg, const float *axis3, const float *t3,
                                     float noise_mm, float noise_rgb, float *F, float *M, float *T_true) try
{
    if (!F || !M || side == 0 || !t3 || (scene_kind == 0 && !axis3) || (scene_kind != 0 && scene_kind != 1)) return ICP_EINVAL;
    rng g (seed);
    const float wall_axis[3] = { (float) WALL_N[0], (float) WALL_N[1], (float) WALL_N[2] };
    const float *axis = scene_kind == 1 ? wall_axis : axis3;
    double R[9]; rotation (rot_deg, axis, R);
    double t[3] = { t3[0], t3[1], t3[2] };
    if (scene_kind == 1) {
        // in-plane translation, rotation about the normal through the plane's centre c0: q = R (p - c0) + c0 + t_in
        const double tn = t[0] * WALL_N[0] + t[1] * WALL_N[1] + t[2] * WALL_N[2];
        const double c0[3] = { 0.0, 0.0, WALL_Z0 };
        for (int k = 0; k < 3; ++k) t[k] = t[k] - tn * WALL_N[k] + c0[k] - (R[k * 3] * c0[0] + R[k * 3 + 1] * c0[1] + R[k * 3 + 2] * c0[2]);
    }
    const double G = (double) side;
    for (uint32_t v = 0; v < side; ++v)
        for (uint32_t u = 0; u < side; ++u) {
            size_t i = (size_t) v * side + u;
            double p[3], c[3];
            (scene_kind == 1 ? scene_wall : scene) ((double) u, (double) v, G, p, c);
            float *f = F + i * 8;
            f[0] = (float) p[0]; f[1] = (float) p[1]; f[2] = (float) p[2]; f[3] = 1.f;
            f[4] = (float) c[0]; f[5] = (float) c[1]; f[6] = (float) c[2]; f[7] = 1.f;
            (scene_kind == 1 ? scene_wall : scene) ((double) u + 0.5, (double) v + 0.5, G, p, c);
            float *mo = M + i * 8;
            for (int k = 0; k < 3; ++k)
                mo[k] = (float) (R[k * 3] * p[0] + R[k * 3 + 1] * p[1] + R[k * 3 + 2] * p[2] + t[k] + noise_mm * g.normal ());
            mo[3] = 1.f;
            for (int k = 0; k < 3; ++k) mo[4 + k] = (float) std::fmin (1.0, std::fmax (0.0, c[k] + noise_rgb * g.normal ()));
            mo[7] = 1.f;
        }
    if (T_true) {
        // M = R P + t  =>  P = R^T (M - t): q = quaternion of R^T (axis, -angle), translation -R^T t
        const double n = std::sqrt ((double) axis[0] * axis[0] + (double) axis[1] * axis[1] + (double) axis[2] * axis[2]);
        const double th = -rot_deg * M_PI / 180.0, s = std::sin (th / 2.0);
        T_true[0] = (float) (axis[0] / n * s); T_true[1] = (float) (axis[1] / n * s); T_true[2] = (float) (axis[2] / n * s); T_true[3] = (float) std::cos (th / 2.0);
        for (int k = 0; k < 3; ++k) T_true[4 + k] = (float) -(R[k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]);
        T_true[7] = 1.f;
    }
    return ICP_OK;
}
ICP_CATCH_ALL

extern "C" int icp_synth_punch_holes (uint64_t seed, uint32_t width, uint32_t height, int pattern, float fraction, int keep_rgb, float *cloud) try
{
    if (!cloud || width == 0 || height == 0 || !(fraction >= 0.f) || fraction > 1.f || (pattern != 0 && pattern != 1)) return ICP_EINVAL;
    rng g (seed ^ 0x401E5ull);
    const size_t n = (size_t) width * height;
    auto punch = [&] (size_t i) {
        float *o = cloud + i * 8;
        o[0] = o[1] = o[2] = 0.f;
        if (!keep_rgb) o[4] = o[5] = o[6] = 0.f;
    };
    if (pattern == 0) {
        for (size_t i = 0; i < n; ++i)
            if (g.uniform () < (double) fraction) punch (i);
        return ICP_OK;
    }
    // contiguous: a bitmap first (ellipses overlap), then the punch
    uint8_t *mask = new (std::nothrow) uint8_t[n];
    if (!mask) return ICP_ENOMEM;
    std::memset (mask, 0, n);
    size_t covered = 0;
    const size_t want = (size_t) ((double) fraction * (double) n);
    const uint32_t band = (uint32_t) std::lround (0.25 * fraction * width);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < band && x < width; ++x) { mask[(size_t) y * width + x] = 1; ++covered; }
    for (int tries = 0; covered < want && tries < 100000; ++tries) {
        const double cx = g.uniform () * width, cy = g.uniform () * height;
        const double a = (0.03 + 0.09 * g.uniform ()) * width, b = (0.03 + 0.09 * g.uniform ()) * height;
        const long x0 = std::max (0l, (long) std::floor (cx - a)), x1 = std::min ((long) width - 1, (long) std::ceil (cx + a));
        const long y0 = std::max (0l, (long) std::floor (cy - b)), y1 = std::min ((long) height - 1, (long) std::ceil (cy + b));
        for (long y = y0; y <= y1; ++y)
            for (long x = x0; x <= x1; ++x) {
                const double ex = ((double) x - cx) / a, ey = ((double) y - cy) / b;
                uint8_t &mk = mask[(size_t) y * width + (size_t) x];
                if (ex * ex + ey * ey < 1.0 && !mk) { mk = 1; ++covered; }
            }
    }
    for (size_t i = 0; i < n; ++i)
        if (mask[i]) punch (i);
    delete[] mask;
    return ICP_OK;
}
ICP_CATCH_ALL

extern "C" int icp_synth_cloud_vga (uint64_t seed, int moved, float *cloud) try
{
    // `moved` = frame number of a synthetic sequence: frame 0 is the scene itself, frame f > 0 the scene moved rigidly by
    // f times the step (3 degrees about (0.3, 0.9, 0.1), t = (25, -10, 15) mm), sampled at a half-pixel offset, with
    // 1 mm noise (noise stream of frame f: seed for f = 1, seed + f beyond)
    if (!cloud || moved < 0) return ICP_EINVAL;
    rng g (moved > 1 ? seed + (uint64_t) moved : seed);
    const float axis[3] = { 0.3f, 0.9f, 0.1f };
    const double f = (double) moved;
    double R[9]; rotation (3.0 * f, axis, R);
    const double t[3] = { 25.0 * f, -10.0 * f, 15.0 * f };
    for (uint32_t v = 0; v < 480; ++v)
        for (uint32_t u = 0; u < 640; ++u) {
            // scene() is parametrised on a square grid: map the VGA pixel onto a 640-wide one
            double p[3], c[3];
            scene ((double) u + (moved ? 0.5 : 0.0), ((double) v + (moved ? 0.5 : 0.0)) * 640.0 / 480.0, 640.0, p, c);
            float *o = cloud + ((size_t) v * 640 + u) * 8;
            for (int k = 0; k < 3; ++k)
                o[k] = (float) (R[k * 3] * p[0] + R[k * 3 + 1] * p[1] + R[k * 3 + 2] * p[2] + t[k] + (moved ? g.normal () : 0.0));
            o[3] = 1.f; o[4] = (float) c[0]; o[5] = (float) c[1]; o[6] = (float) c[2]; o[7] = 1.f;
        }
    return ICP_OK;
}
ICP_CATCH_ALL
