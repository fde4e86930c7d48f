// ray_harness.cpp — TEST-ONLY host build of the device ray functions in slam_amd/csrc/ray.h.
// Lets the CPU test-suite check the exact-jump / clearance logic of ray_fast_L against the
// literal walk without a GPU.  Not shipped, not reachable from libslam_hip.so.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../slam_amd/csrc/clearance.h"
#include "../../slam_amd/csrc/ray.h"

#define HH_API extern "C" __attribute__((visibility("default")))

HH_API void hh_build_clearance(const uint8_t* map, int rows, int cols, uint8_t* E)
{
    slam_build_clearance(map, rows, cols, E);
}

// the 8 padded cone-clearance planes the kernels read (clearance.h), plane stride returned through *stride
HH_API void hh_build_cone_planes(const uint8_t* map, int rows, int cols, uint8_t* planes)
{
    slam_build_cone_clearance_padded(map, rows, cols, planes);
}
HH_API unsigned long hh_cone_plane_stride(int rows, int cols) { return (unsigned long)slam_cone_plane_stride(rows, cols); }

static std::vector<uint8_t> cone_planes(const uint8_t* map, int rows, int cols)
{
    std::vector<uint8_t> P(8 * slam_cone_plane_stride(rows, cols));
    slam_build_cone_clearance_padded(map, rows, cols, P.data());
    return P;
}

// returns number of rays whose (hx, hy, hcell) differ bitwise between ray_fast_L and ray_walk.
HH_API long hh_rays_compare(const uint8_t* map, const uint8_t* E, int rows, int cols, long n, const double* sx,
                            const double* sy, const double* theta, double zmax, long* first_bad,
                            unsigned long long* reads_fast, unsigned long long* reads_walk, double* out_hits)
{
    long bad = 0;
    *first_bad = -1;
    *reads_fast = 0;
    *reads_walk = 0;
    const double zmax2 = zmax * zmax;
    int k_safe = (int)__builtin_floor(zmax) - 1;
    if (k_safe < 0) k_safe = 0;
    const int max_iter = (int)__builtin_ceil(zmax) + 8;
    const std::vector<uint8_t> Ep = cone_planes(map, rows, cols);
    const size_t stride = slam_cone_plane_stride(rows, cols);
    // E (the unpadded Chebyshev clearance) in the padded layout: the one-plane form the library falls back to for
    // maps too large for 32-bit plane offsets (plane stride 0); checked on every ray as well
    std::vector<uint8_t> Ec((size_t)(rows + 4) * (cols + 4), 0);
    for (int i = 0; i < rows; ++i) std::memcpy(Ec.data() + (size_t)(i + 2) * (cols + 4) + 2, E + (size_t)i * cols, cols);
    for (long r = 0; r < n; ++r)
    {
        double s, c;
        dm_sincos(theta[r], &s, &c);
        slam_ray_result a, b, b1;
        ray_fast_L(Ec.data(), 0u, cols + 4, rows, cols, sx[r], sy[r], c, s, zmax2, k_safe, &b1, nullptr);
        slam_ray_stats st = {0, 0, 0};
        unsigned reads = 0;
        ray_walk<false, true>(const_cast<uint8_t*>(map), rows, cols, sx[r], sy[r], c, s, zmax2, zmax2, nullptr, nullptr,
                              max_iter, &a, &st);
        ray_fast_L(Ep.data(), (unsigned)stride, cols + 4, rows, cols, sx[r], sy[r], c, s, zmax2, k_safe, &b, &reads);
        *reads_fast += reads;
        *reads_walk += st.cells_read;
        if (out_hits)
        {
            out_hits[2 * r] = a.hx;
            out_hits[2 * r + 1] = a.hy;
        }
        if (std::memcmp(&a.hx, &b.hx, 8) || std::memcmp(&a.hy, &b.hy, 8) || a.hcell != b.hcell ||
            std::memcmp(&a.hx, &b1.hx, 8) || std::memcmp(&a.hy, &b1.hy, 8) || a.hcell != b1.hcell)
        {
            if (!bad) *first_bad = r;
            ++bad;
        }
    }
    return bad;
}

// Checks slam_advance against plain repeated addition: walks `total` steps in jumps taken from
// jumps[] (cyclic) and compares every landing position bitwise.  Returns mismatches.
HH_API long hh_advance_check(double x0, double y0, double dx, double dy, int total, const int* jumps, int njumps,
                             long* jumped_steps)
{
    slam_axis X = {x0, 0.0, 0}, Y = {y0, 0.0, 0};
    slam_axis_setup(&X, dx);
    slam_axis_setup(&Y, dy);
    double x = x0, y = y0;
    long bad = 0;
    int k = 0, q = 0;
    *jumped_steps = 0;
    while (k < total)
    {
        int m = jumps[q++ % njumps];
        if (m > total - k) m = total - k;
        const int rem_before = X.rem < Y.rem ? X.rem : Y.rem;
        if (rem_before > 1) *jumped_steps += (m < rem_before ? m : rem_before);
        slam_advance(&X, &Y, dx, dy, m);
        for (int t = 0; t < m; ++t)
        {
            x += dx;
            y += dy;
        }
        k += m;
        if (std::memcmp(&x, &X.v, 8) || std::memcmp(&y, &Y.v, 8)) ++bad;
    }
    return bad;
}

// per-ray number of clearance-grid reads of ray_fast_L (diagnostics for kernel tuning)
HH_API void hh_fast_reads(const uint8_t* map, int rows, int cols, long n, const double* sx, const double* sy,
                          const double* theta, double zmax, int* reads_out)
{
    const double zmax2 = zmax * zmax;
    int k_safe = (int)__builtin_floor(zmax) - 1;
    if (k_safe < 0) k_safe = 0;
    const std::vector<uint8_t> Ep = cone_planes(map, rows, cols);
    const size_t stride = slam_cone_plane_stride(rows, cols);
    for (long r = 0; r < n; ++r)
    {
        double s, c;
        dm_sincos(theta[r], &s, &c);
        slam_ray_result b;
        unsigned reads = 0;
        ray_fast_L(Ep.data(), (unsigned)stride, cols + 4, rows, cols, sx[r], sy[r], c, s, zmax2, k_safe, &b, &reads);
        reads_out[r] = (int)reads;
    }
}

// per-ray number of clearance lookups of a whole weighting pass as the kernels run it (search limited to z + 4 sigma):
// n particles x B beams, reads_out[p * B + b] (design studies of the lane assignment, tools/lane_sim.py)
HH_API void hh_pass_reads(const uint8_t* planes, int rows, int cols, long n, const double* px, const double* py, const double* pth,
                          int B, const double* angles, const double* z, double zmax, double stddev, double sd, double stheta,
                          double srot, int* reads_out)
{
    const double zmax2 = zmax * zmax;
    int k_safe = (int)__builtin_floor(zmax) - 1;
    if (k_safe < 0) k_safe = 0;
    const size_t stride = slam_cone_plane_stride(rows, cols);
    for (long p = 0; p < n; ++p)
    {
        double s0, c0;
        dm_sincos(pth[p] + stheta, &s0, &c0);
        const double sx = px[p] + c0 * sd, sy = py[p] + s0 * sd;
        for (int b = 0; b < B; ++b)
        {
            double s, c;
            dm_sincos(pth[p] + angles[b] + srot, &s, &c);
            slam_ray_result r;
            unsigned reads = 0;
            ray_fast_L(planes, (unsigned)stride, cols + 4, rows, cols, sx, sy, c, s, zmax2, k_safe, &r, &reads, slam_sample_limit(z[b], stddev));
            reads_out[p * B + b] = (int)reads;
        }
    }
}

// slam_beam_logw against its two-step form (slam_beam_d2 / slam_beam_clamped / slam_beam_logw_d2, what k_weights_L<0>
// runs): the value must be the same bits for every ray, and a ray declared clamped must really evaluate to log_eps.
// hx < 0 selects the markers (-1: miss, -2: limit).  Returns the differing rays; *deferred counts the undecided ones.
HH_API long hh_logw_two_step(long n, const double* hx, const double* hy, const double* sx, const double* sy, const double* z,
                             double zmax, double stddev, int fast_div, long* deferred)
{
    const double clamp = 4.0 * stddev, norm = 1.0 / (stddev * __builtin_sqrt(2 * M_PI)), log_eps = dm_log(0.0 + 0.1);
    const double inv = fast_div ? 1.0 / stddev : 0.0;
    long bad = 0;
    *deferred = 0;
    for (long r = 0; r < n; ++r)
    {
        slam_ray_result res = {hx[r], hy[r], 0};
        const double want = slam_beam_logw(&res, sx[r], sy[r], z[r], zmax, stddev, clamp, norm, log_eps, inv);
        const double d2 = slam_beam_d2(&res, sx[r], sy[r]);
        double got;
        if (slam_beam_clamped(d2, z[r], zmax, clamp))
            got = log_eps;
        else
        {
            ++*deferred;
            got = slam_beam_logw_d2(d2, z[r], zmax, stddev, clamp, norm, log_eps, inv);
        }
        if (std::memcmp(&want, &got, 8)) ++bad;
    }
    return bad;
}

// The sample limit of ray_fast_L0 (search only as far as z + 4 sigma): the beam log-likelihood must equal, bit for bit,
// the one computed from the unlimited search.  Returns the number of differing rays; *stopped counts the rays the
// limit ended, *reads_lim / *reads_full the clearance lookups of the two searches.
HH_API long hh_limit_compare(const uint8_t* map, int rows, int cols, long n, const double* sx, const double* sy,
                             const double* theta, const double* z, double zmax, double stddev, long* first_bad, long* stopped,
                             unsigned long long* reads_lim, unsigned long long* reads_full)
{
    long bad = 0;
    *first_bad = -1;
    *stopped = 0;
    *reads_lim = *reads_full = 0;
    const double zmax2 = zmax * zmax;
    int k_safe = (int)__builtin_floor(zmax) - 1;
    if (k_safe < 0) k_safe = 0;
    const std::vector<uint8_t> Ep = cone_planes(map, rows, cols);
    const size_t stride = slam_cone_plane_stride(rows, cols);
    const double norm = 1.0 / (stddev * __builtin_sqrt(2 * 3.14159265358979323846));
    const double log_eps = dm_log(0.0 + 0.1);
    for (long r = 0; r < n; ++r)
    {
        double s, c;
        dm_sincos(theta[r], &s, &c);
        slam_ray_result full, lim;
        unsigned rf = 0, rl = 0;
        ray_fast_L(Ep.data(), (unsigned)stride, cols + 4, rows, cols, sx[r], sy[r], c, s, zmax2, k_safe, &full, &rf);
        ray_fast_L(Ep.data(), (unsigned)stride, cols + 4, rows, cols, sx[r], sy[r], c, s, zmax2, k_safe, &lim, &rl,
                   slam_sample_limit(z[r], stddev));
        *reads_full += rf;
        *reads_lim += rl;
        if (lim.hx == -2.0) ++*stopped;
        const double zz = __builtin_sqrt(z[r] * z[r]);
        const double a = slam_beam_logw(&full, sx[r], sy[r], zz, zmax, stddev, 4.0 * stddev, norm, log_eps);
        const double b = slam_beam_logw(&lim, sx[r], sy[r], zz, zmax, stddev, 4.0 * stddev, norm, log_eps);
        const bool same_hit = lim.hx == -2.0 || (!std::memcmp(&full.hx, &lim.hx, 8) && !std::memcmp(&full.hy, &lim.hy, 8) && full.hcell == lim.hcell);
        if (std::memcmp(&a, &b, 8) || !same_hit)
        {
            if (!bad) *first_bad = r;
            ++bad;
        }
    }
    return bad;
}

// The quotient xarg / stddev of slam_beam_logw formed from the reciprocal (two correction steps) against the division:
// returns the number of pairs whose results differ in any bit.
HH_API long hh_fast_div_check(long n, const double* a, const double* b)
{
    long bad = 0;
    for (long i = 0; i < n; ++i)
    {
        const double y = 1.0 / b[i];
        const double q0 = a[i] * y;
        const double q1 = dm_fma(dm_fma(-q0, b[i], a[i]), y, q0);
        const double q = dm_fma(dm_fma(-q1, b[i], a[i]), y, q1);
        const double want = a[i] / b[i];
        if (std::memcmp(&q, &want, 8)) ++bad;
    }
    return bad;
}
