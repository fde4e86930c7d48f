// oracle.cpp — CPU restatement of the reference particle-filter step.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product (slam_amd/, libslam_hip.so) never links, imports or calls it.
//
// PARITY STATUS: "parity unpinned" against a *binary* of the reference.  The reference
// (/root/reference, ferasboulala/slam) has no tests, golden vectors or fixtures for this path
// (SURVEY.md §4), and every one of its source files includes <opencv2/opencv.hpp>, which this image
// does not have, so it is unbuildable here without writing stand-in headers (not done).  What pins
// this file instead: (1) it is a statement-by-statement restatement, each function citing the
// reference file:line it follows; (2) the third-party arithmetic the reference relies on is used
// directly, not restated: libstdc++ <random> (minstd_rand0, normal_distribution,
// uniform_real_distribution, std::sort) and glibc libm; (3) the known answers SURVEY.md records
// from the survey session's run of the unmodified reference (update-table values, all-zero-weight
// selection index (N-1)/2, offset truncation, first-update tie 0.1^B) are asserted in
// tests/test_oracle.py.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off; the reference's own -Ofast is NOT
// parity-safe).
//
// "math" argument of the entry points: 0 = glibc libm (what the reference calls), 1 = the
// deterministic functions of slam_amd/csrc/detmath.h (what the HIP kernels call; used so that GPU
// and oracle can be compared bit for bit).  tests/test_oracle.py compares the two modes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <tuple>
#include <vector>

#include "../slam_amd/csrc/detmath.h"
#include "../slam_amd/csrc/noise.h"

namespace
{
struct LibmMath
{
    static double cos(double x) { return std::cos(x); }
    static double sin(double x) { return std::sin(x); }
    static double exp(double x) { return std::exp(x); }
    static double log(double x) { return std::log(x); }
    static double fmod2pi(double x) { return std::fmod(x, 2 * M_PI); }
};
struct DetMath
{
    static double cos(double x) { return dm_cos(x); }
    static double sin(double x) { return dm_sin(x); }
    static double exp(double x) { return dm_exp(x); }
    static double log(double x) { return dm_log(x); }
    static double fmod2pi(double x) { return dm_fmod_pos(x, 2 * M_PI); }
};

constexpr long HANG_GUARD = 1L << 24;  // the reference loops forever when |coordinate| >= 2^31

// slam/util.h:30-33  (map.rows - pose.y - 1, pose.x) converted to int by truncation.
inline void pose_to_image_coordinates(int rows, double x, double y, int& i, int& j)
{
    i = static_cast<int>(rows - y - 1);
    j = static_cast<int>(x);
}

// slam/util.h:45-49
inline bool within_boundaries(int rows, int cols, int i, int j)
{
    return static_cast<unsigned>(i) < static_cast<unsigned>(rows) &&
           static_cast<unsigned>(j) < static_cast<unsigned>(cols);
}

// slam/util.h:118-125
inline double euclidean_distance_squared(double Ax, double Ay, double Bx, double By)
{
    const double dx = Ax - Bx;
    const double dy = Ay - By;
    return dx * dx + dy * dy;
}

// slam/raycast.cpp:196-201 (free) and :207-211 (occupied): the per-cell update is a pure function
// of the old byte.  Evaluated exactly as written there (double arithmetic, truncating casts).
inline unsigned char update_free(unsigned char c)
{
    constexpr double L0 = 0.5;
    constexpr double Lfree = 0.60;
    double p = static_cast<double>(c) / 255;
    p *= Lfree / L0;
    p = std::min(1.0, p);
    const unsigned char p_ = std::max(static_cast<unsigned char>(1), static_cast<unsigned char>(p * 255));
    return static_cast<unsigned char>(static_cast<double>(p_));
}
inline unsigned char update_occ(unsigned char c)
{
    constexpr double L0 = 0.5;
    constexpr double Locc = 0.40;
    double p = static_cast<double>(c) / 255;
    p *= Locc / L0;
    const unsigned char p_ = std::max(static_cast<unsigned char>(1), static_cast<unsigned char>(p * 255));
    return static_cast<unsigned char>(static_cast<double>(p_));
}

struct RayStats
{
    long cells_read = 0;     // iterations that reach the map read (in-bounds new cells)
    long cells_written = 0;  // map stores
    long steps = 0;          // loop iterations (x += dx)
};

// slam/raycast.cpp:143-223 raycast_mapping.  `write` = false suppresses the two map stores and
// nothing else: this is the localization-mode composition of SURVEY.md §8a row aL (within one ray
// no cell is visited twice and the hit test precedes the store, so a ray's own stores never change
// its result).
template <class M>
int raycast_mapping(unsigned char* map, int rows, int cols, double px, double py, double theta, double z_squared,
                    double z_max_squared, double step_size, bool write, double& hit_x, double& hit_y, int& hit_cell,
                    RayStats* st)
{
    const double dx = step_size * M::cos(theta);
    const double dy = step_size * M::sin(theta);

    double x = px;
    double y = py;
    int i, j;
    pose_to_image_coordinates(rows, x, y, i, j);
    int prev_i = i;
    int prev_j = j;

    hit_x = -1;
    hit_y = -1;
    hit_cell = -1;
    bool drew = false;
    bool foundHit = false;
    long guard = 0;

    while (!drew || !foundHit)
    {
        if (++guard > HANG_GUARD) return -1;
        x += dx;
        y += dy;
        if (st) st->steps++;
        pose_to_image_coordinates(rows, x, y, i, j);
        if (prev_i == i && prev_j == j) continue;
        if (!within_boundaries(rows, cols, i, j)) break;
        unsigned char& cell = map[static_cast<size_t>(i) * cols + j];
        if (st) st->cells_read++;

        // must run before the map update below (slam/raycast.cpp:182-189)
        if (!foundHit && cell < 128)
        {
            hit_x = x;
            hit_y = y;
            hit_cell = i * cols + j;
            foundHit = true;
        }

        const double distance_squared = euclidean_distance_squared(x, y, px, py);
        if (distance_squared < z_squared)
        {
            if (write)
            {
                cell = update_free(cell);
                if (st) st->cells_written++;
            }
        }
        else
        {
            if (!drew && z_squared != z_max_squared)
            {
                if (write)
                {
                    cell = update_occ(cell);
                    if (st) st->cells_written++;
                }
            }
            drew = true;
        }
        if (distance_squared >= z_max_squared) break;
        prev_i = i;
        prev_j = j;
    }
    return 0;
}

// slam/util.cpp:9-19
template <class M>
double pdf_normal_distribution(double stddev, double x)
{
    const double q = x / stddev;
    return (1.0 / (stddev * std::sqrt(2 * M_PI))) * M::exp(-0.5 * (q * q));  // std::pow(q, 2) == q*q
}
template <class M>
double pdf_normal_distribution_clamp(double stddev, double x, double multiple_stddev = 4)
{
    if (std::fabs(x) > multiple_stddev * stddev) return 0;
    return pdf_normal_distribution<M>(stddev, x);
}

// slam/raycast.cpp:225-242 measurement_model_beam (step_size default 1, slam/raycast.h:28).
template <class M>
int measurement_model_beam(unsigned char* map, int rows, int cols, double px, double py, double theta,
                           double distance_squared, double stddev, double max_distance_squared, bool write, double& w,
                           double& hx, double& hy, int& hcell, RayStats* st)
{
    if (raycast_mapping<M>(map, rows, cols, px, py, theta, distance_squared, max_distance_squared, 1.0, write, hx, hy,
                           hcell, st))
        return -1;
    constexpr double EPSILON = 0.1;
    if (hx == -1)
    {
        w = pdf_normal_distribution_clamp<M>(stddev, std::sqrt(distance_squared) - std::sqrt(max_distance_squared)) +
            EPSILON;
        return 0;
    }
    const double distance = std::sqrt(euclidean_distance_squared(hx, hy, px, py));
    w = pdf_normal_distribution_clamp<M>(stddev, distance - std::sqrt(distance_squared)) + EPSILON;
    return 0;
}

// slam/mcl.cpp:80-86 with slam/util.h:138-146: the squared distance passes through an `int`.
inline void scanner_offset_to_displacement(const double off[3], double& d, double& theta)
{
    const double dx = 0 - off[0];
    const double dy = 0 - off[1];
    const int sq = static_cast<int>(dx * dx + dy * dy);
    d = std::sqrt(static_cast<double>(sq));
    theta = std::atan2(off[1], off[0]);
}

// slam/mcl.cpp:49-78 update_inner for particles [start, start+n).
template <class M>
int update_inner(int start, int n, const double* x, const double* y, const double* th, unsigned char* maps,
                 size_t map_stride, int rows, int cols, int B, const double* angles, const double* dists, double stddev,
                 double max_dist, double scanner_d, double scanner_theta, double scanner_rot, bool write, double* w_out,
                 double* hit_xy, int* hit_cell, RayStats* st, bool log_domain = false)
{
    for (int p = start; p < start + n; ++p)
    {
        unsigned char* map = maps + map_stride * p;
        const double dx = M::cos(th[p] + scanner_theta) * scanner_d;
        const double dy = M::sin(th[p] + scanner_theta) * scanner_d;
        const double sx = x[p] + dx;
        const double sy = y[p] + dy;
        double weight = 0;
        const double max_dist_squared = max_dist * max_dist;
        for (int b = 0; b < B; ++b)
        {
            const double theta = th[p] + angles[b] + scanner_rot;
            double w, hx, hy;
            int hc;
            if (measurement_model_beam<M>(map, rows, cols, sx, sy, theta, dists[b] * dists[b], stddev, max_dist_squared,
                                          write, w, hx, hy, hc, st))
                return -1;
            weight += M::log(w);
            if (hit_xy)
            {
                hit_xy[(static_cast<size_t>(p) * B + b) * 2 + 0] = hx;
                hit_xy[(static_cast<size_t>(p) * B + b) * 2 + 1] = hy;
            }
            if (hit_cell) hit_cell[static_cast<size_t>(p) * B + b] = hc;
        }
        // log_domain: opt-in extension, the sum of slam/mcl.cpp:73 without the exp of :75
        w_out[p] = log_domain ? weight : M::exp(weight);
    }
    return 0;
}

// slam/motion.cpp:9-32 with the noise seam: z[0..2] are the unit normals that libstdc++'s
// normal_distribution(0, sigma) would have scaled by sigma (slam/util.cpp:27-34).
template <class M>
void sample_motion_model_odometry(const double odom[3], const double alphas[4], const double z[3], double& x, double& y,
                                  double& th)
{
    const double r1 = odom[0], t = odom[1], r2 = odom[2];
    const double rotation_1 = r1 - z[0] * std::sqrt(alphas[0] * (r1 * r1) + alphas[1] * (t * t));
    const double translation = t - z[1] * std::sqrt(alphas[2] * (t * t) + alphas[3] * ((r1 * r1) + (r2 * r2)));
    const double rotation_2 = r2 - z[2] * std::sqrt(alphas[0] * (r2 * r2) + alphas[1] * (t * t));
    const double nx = x + translation * M::cos(th + rotation_1);
    const double ny = y + translation * M::sin(th + rotation_1);
    // slam/util.cpp:45-51 normalize_angle
    double angle = M::fmod2pi(th + rotation_1 + rotation_2);
    if (angle > M_PI) angle -= 2 * M_PI;
    x = nx;
    y = ny;
    th = angle;
}

// slam/motion.cpp:34-56 sample_motion_model_velocity with the same noise seam: z[0..2] are the unit
// normals behind the three draws (v, w, gamma), in the order the reference draws them.
template <class M>
void sample_motion_model_velocity(const double vel[2], double dt, const double alphas[6], const double z[3], double& x,
                                  double& y, double& th)
{
    constexpr double EPSILON = 1e-6;
    const double v0 = vel[0], w0 = vel[1];
    const double v = v0 + z[0] * std::sqrt(alphas[0] * (v0 * v0) + alphas[1] * (w0 * w0));
    const double w = w0 + z[1] * std::sqrt(alphas[2] * (w0 * w0) + alphas[3] * (v0 * v0));
    const double gamma = z[2] * std::sqrt(alphas[4] * (v0 * v0) + alphas[5] * (w0 * w0));
    const double nx = x - v / (w == 0 ? 1 : w) * (M::sin(th) + M::sin(th + (w + EPSILON) * dt));
    const double ny = y + v / (w == 0 ? 1 : w) * (M::cos(th) - M::cos(th + (w + EPSILON) * dt));
    double angle = M::fmod2pi(th + w * dt + gamma * dt);
    if (angle > M_PI) angle -= 2 * M_PI;
    x = nx;
    y = ny;
    th = angle;
}

// Canonical blocked prefix order (radix-64 hierarchy of Hillis-Steele scans); the HIP kernels
// implement exactly this order, independent of the number of GPUs.
void hs_scan_chunks(const double* a, long n, double* s)
{
    for (long c0 = 0; c0 < n; c0 += 64)
    {
        const int m = static_cast<int>(std::min<long>(64, n - c0));
        double v[64];
        for (int i = 0; i < m; ++i) v[i] = a[c0 + i];
        for (int d = 1; d < 64; d <<= 1)
        {
            double t[64];
            for (int i = 0; i < m; ++i) t[i] = (i >= d) ? v[i - d] + v[i] : v[i];
            for (int i = 0; i < m; ++i) v[i] = t[i];
        }
        for (int i = 0; i < m; ++i) s[c0 + i] = v[i];
    }
}

void blocked_prefix(const double* w, long n, double* P)
{
    std::vector<double> s(n);
    hs_scan_chunks(w, n, s.data());
    if (n <= 64)
    {
        std::memcpy(P, s.data(), sizeof(double) * n);
        return;
    }
    const long n1 = (n + 63) / 64;
    std::vector<double> tot(n1), P1(n1);
    for (long c = 0; c < n1; ++c) tot[c] = s[std::min(n - 1, c * 64 + 63)];
    blocked_prefix(tot.data(), n1, P1.data());
    for (long j = 0; j < n; ++j)
    {
        const long c = j / 64;
        P[j] = c ? P1[c - 1] + s[j] : s[j];
    }
}

// slam/mcl.cpp:182-200: bisection where equality returns `mid` at once.
inline int select_index(const double* cum, long n, double cutoff)
{
    int i = 0;
    int j = static_cast<int>(n) - 1;
    while (i < j)
    {
        const int mid = i + (j - i) / 2;
        if (cum[mid] == cutoff)
        {
            i = mid;
            j = mid;
        }
        else if (cum[mid] < cutoff)
            i = mid + 1;
        else
            j = mid;
    }
    return i;
}

// slam/raycast.cpp:7-141 raycast<T>: order of tests differs between T = int (bounds, distance,
// hit) and T = double / unsigned char (distance, bounds, hit).
template <class M, class T, bool BOUNDS_FIRST, class Hit>
int raycast_ro(const T* map, int rows, int cols, double px, double py, double theta, double max_distance_squared,
               double step_size, Hit is_hit, double& hx, double& hy)
{
    const double dx = step_size * M::cos(theta);
    const double dy = step_size * M::sin(theta);
    double x = px, y = py;
    int i, j;
    pose_to_image_coordinates(rows, x, y, i, j);
    int prev_i = i, prev_j = j;
    long guard = 0;
    hx = -1;
    hy = -1;
    while (true)
    {
        if (++guard > HANG_GUARD) return -1;
        x += dx;
        y += dy;
        pose_to_image_coordinates(rows, x, y, i, j);
        if (prev_i == i && prev_j == j) continue;
        if (BOUNDS_FIRST)
        {
            if (!within_boundaries(rows, cols, i, j)) return 0;
            if (euclidean_distance_squared(x, y, px, py) >= max_distance_squared) return 0;
        }
        else
        {
            if (euclidean_distance_squared(x, y, px, py) >= max_distance_squared) return 0;
            if (!within_boundaries(rows, cols, i, j)) return 0;
        }
        if (is_hit(map[static_cast<size_t>(i) * cols + j]))
        {
            hx = x;
            hy = y;
            return 0;
        }
        prev_i = i;
        prev_j = j;
    }
}

template <class M>
int fake_lidar_scan(const int* map, int rows, int cols, const double pose[3], double start, double stop,
                    double max_dist, int n_rays, double* hits_xy)
{
    // slam/fake_lidar.cpp:15-33
    const double range = stop - start;
    const double step = range / n_rays;
    double theta = pose[2];
    theta -= range / 2;
    const double max_dist_squared = max_dist * max_dist;
    for (int i = 0; i < n_rays; ++i)
    {
        double hx, hy;
        if (raycast_ro<M, int, true>(map, rows, cols, pose[0], pose[1], theta, max_dist_squared, 0.5,
                                     [](int v) { return !v; }, hx, hy))
            return -1;
        hits_xy[2 * i] = hx;
        hits_xy[2 * i + 1] = hy;
        theta += step;
    }
    return 0;
}

}  // namespace

#define ORC_API extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------------
ORC_API void orc_update_tables(uint8_t* free_lut, uint8_t* occ_lut)
{
    for (int c = 0; c < 256; ++c)
    {
        free_lut[c] = update_free(static_cast<unsigned char>(c));
        occ_lut[c] = update_occ(static_cast<unsigned char>(c));
    }
}

// One ray of slam/raycast.cpp:143-223 on `map` (modified in place when write != 0).
ORC_API int orc_raycast_mapping(uint8_t* map, int rows, int cols, double px, double py, double theta, double z2,
                                double zmax2, double step_size, int write, int math, double* hit_xy, int* hit_cell,
                                long* stats3)
{
    RayStats st;
    double hx, hy;
    int hc;
    const int rc = math ? raycast_mapping<DetMath>(map, rows, cols, px, py, theta, z2, zmax2, step_size, write != 0,
                                                   hx, hy, hc, &st)
                        : raycast_mapping<LibmMath>(map, rows, cols, px, py, theta, z2, zmax2, step_size, write != 0,
                                                    hx, hy, hc, &st);
    hit_xy[0] = hx;
    hit_xy[1] = hy;
    *hit_cell = hc;
    if (stats3)
    {
        stats3[0] = st.cells_read;
        stats3[1] = st.cells_written;
        stats3[2] = st.steps;
    }
    return rc;
}

// slam/raycast.cpp:225-242 for one beam.
ORC_API int orc_measurement_model_beam(uint8_t* map, int rows, int cols, double px, double py, double theta,
                                       double distance_squared, double stddev, double max_distance_squared, int write,
                                       int math, double* w_out)
{
    double hx, hy;
    int hc;
    return math ? measurement_model_beam<DetMath>(map, rows, cols, px, py, theta, distance_squared, stddev,
                                                  max_distance_squared, write != 0, *w_out, hx, hy, hc, nullptr)
                : measurement_model_beam<LibmMath>(map, rows, cols, px, py, theta, distance_squared, stddev,
                                                   max_distance_squared, write != 0, *w_out, hx, hy, hc, nullptr);
}

// slam/mcl.cpp:102-142 (the weighting half of MCL::update).  mode_slam = 1: maps holds N maps of
// rows*cols bytes, each particle walks and updates its own (slam::MCL as written).  mode_slam = 0:
// maps is ONE shared map, read only (SURVEY.md §8a row aL).  n_threads follows the reference's
// slicing (contiguous N/T, remainder on the last thread); n_threads <= 1 runs inline.
// math: bit 0 = deterministic elementary functions, bit 1 = log-domain weights (extension).
ORC_API int orc_update_weights(int mode_slam, long N, const double* x, const double* y, const double* th,
                               uint8_t* maps, int rows, int cols, int B, const double* angles, const double* dists,
                               double stddev, double max_dist, const double* offset3, int math, int n_threads,
                               double* w_out, double* hit_xy, int* hit_cell, long* stats3)
{
    double sd, stheta;
    scanner_offset_to_displacement(offset3, sd, stheta);
    const size_t stride = mode_slam ? static_cast<size_t>(rows) * cols : 0;
    const bool write = mode_slam != 0;
    const int T = std::max(1, n_threads);
    std::vector<RayStats> st(T);
    std::vector<int> rc(T, 0);
    const bool det = (math & 1) != 0, logd = (math & 2) != 0;
    auto run = [&](int t, long start, long n) {
        rc[t] = det ? update_inner<DetMath>(static_cast<int>(start), static_cast<int>(n), x, y, th, maps, stride, rows,
                                            cols, B, angles, dists, stddev, max_dist, sd, stheta, offset3[2], write,
                                            w_out, hit_xy, hit_cell, &st[t], logd)
                    : update_inner<LibmMath>(static_cast<int>(start), static_cast<int>(n), x, y, th, maps, stride, rows,
                                             cols, B, angles, dists, stddev, max_dist, sd, stheta, offset3[2], write,
                                             w_out, hit_xy, hit_cell, &st[t], logd);
    };
    if (T == 1)
        run(0, 0, N);
    else
    {
        std::vector<std::thread> threads(T);
        const long per = N / T;
        for (int t = 0; t < T; ++t)
        {
            const long start = t * per;
            long n = per;
            if (t == T - 1) n += N % T;
            threads[t] = std::thread(run, t, start, n);
        }
        for (auto& t : threads) t.join();
    }
    if (stats3)
    {
        stats3[0] = stats3[1] = stats3[2] = 0;
        for (auto& s : st)
        {
            stats3[0] += s.cells_read;
            stats3[1] += s.cells_written;
            stats3[2] += s.steps;
        }
    }
    for (int r : rc)
        if (r) return r;
    return 0;
}

// Log-domain extension: the weights the selection of slam/mcl.cpp:157-203 runs on,
// w'[p] = exp(lw[p] - max_q lw[q]) (1 for the best particle, no underflow of the whole set).
ORC_API void orc_log_domain_weights(long N, const double* lw, int math, double* out)
{
    double m = lw[0];
    for (long p = 1; p < N; ++p) m = lw[p] > m ? lw[p] : m;
    for (long p = 0; p < N; ++p) out[p] = (math & 1) ? DetMath::exp(lw[p] - m) : LibmMath::exp(lw[p] - m);
}

// slam/mcl.cpp:41-47 MCL::predict with injected unit normals z[3*N].
ORC_API void orc_predict(long N, double* x, double* y, double* th, const double* odom3, const double* alphas4,
                         const double* z, int math)
{
    for (long p = 0; p < N; ++p)
    {
        if (math)
            sample_motion_model_odometry<DetMath>(odom3, alphas4, z + 3 * p, x[p], y[p], th[p]);
        else
            sample_motion_model_odometry<LibmMath>(odom3, alphas4, z + 3 * p, x[p], y[p], th[p]);
    }
}

// slam/motion.cpp:34-56 for N poses with injected unit normals z[3*N].
ORC_API void orc_predict_velocity(long N, double* x, double* y, double* th, const double* vel2, double dt,
                                  const double* alphas6, const double* z, int math)
{
    for (long p = 0; p < N; ++p)
    {
        if (math)
            sample_motion_model_velocity<DetMath>(vel2, dt, alphas6, z + 3 * p, x[p], y[p], th[p]);
        else
            sample_motion_model_velocity<LibmMath>(vel2, dt, alphas6, z + 3 * p, x[p], y[p], th[p]);
    }
}

// slam/util.cpp:36-43 sample_triangular_distribution: one clock re-seed, then TWO draws of
// uniform_real_distribution(-stddev, stddev) from the same engine; the first is scaled by sqrt(6)/2.
ORC_API void orc_reference_triangular(long n, long tick0, long tick_stride, double stddev, double* out)
{
    for (long k = 0; k < n; ++k)
    {
        std::default_random_engine generator;
        generator.seed(static_cast<unsigned long>(tick0 + k * tick_stride));
        std::uniform_real_distribution<double> distribution(-stddev, stddev);
        out[k] = std::sqrt(6) / 2 * (distribution(generator)) + distribution(generator);
    }
}

// slam/util.cpp:20-24 pdf_triangular_distribution.
ORC_API void orc_pdf_triangular(long n, double stddev, const double* x, double* out)
{
    for (long k = 0; k < n; ++k)
    {
        const double variance = stddev * stddev;
        out[k] = std::max(0.0, 1 / std::sqrt(6 * variance) - std::fabs(x[k]) / 6 / variance);
    }
}

// The reference's own noise source, for completeness of the restatement: slam/util.cpp:27-34
// re-seeds std::default_random_engine (= minstd_rand0) from the clock for EVERY draw and takes one
// normal_distribution(0, stddev) sample.  `tick` stands for system_clock::now() ticks; draw k uses
// tick0 + k*tick_stride.  stddev = 1 yields the unit normals the seam above consumes.
ORC_API void orc_reference_normals(long n, long tick0, long tick_stride, double stddev, double* out)
{
    for (long k = 0; k < n; ++k)
    {
        std::default_random_engine generator;
        generator.seed(static_cast<unsigned long>(tick0 + k * tick_stride));
        std::normal_distribution<double> distribution(0, stddev);
        out[k] = distribution(generator);
    }
}

// slam/mcl.cpp:166-168,181: one re-seed, then N draws of uniform_real_distribution(0, sum).
// Returns the canonical uniforms (draw / sum is not exact, so the generator is run with sum = 1:
// libstdc++ computes canonical * (b - a) + a, and canonical * 1 + 0 is the canonical value).
ORC_API void orc_reference_uniforms(long n, long tick, double* out)
{
    std::default_random_engine generator;
    generator.seed(static_cast<unsigned long>(tick));
    std::uniform_real_distribution<double> distribution(0, 1.0);
    for (long k = 0; k < n; ++k) out[k] = distribution(generator);
}

// Philox streams of slam_amd/csrc/noise.h evaluated on the host (the device generates the same
// values in its kernels).
ORC_API void orc_philox_normals(uint64_t seed, uint64_t step, long first, long n, double* z3)
{
    for (long p = 0; p < n; ++p) slam_motion_normals(seed, step, static_cast<uint64_t>(first + p), z3 + 3 * p);
}
ORC_API void orc_philox_uniforms(uint64_t seed, uint64_t step, long first, long n, double* u)
{
    for (long k = 0; k < n; ++k) u[k] = slam_resample_uniform(seed, step, static_cast<uint64_t>(first + k));
}

// Host replay of the device seeding helpers of slam_amd/csrc/noise.h (no reference counterpart).
ORC_API void orc_seed_uniform_free(uint64_t seed, long first, long n, const uint8_t* map, int rows, int cols, double* x,
                                   double* y, double* th)
{
    for (long p = 0; p < n; ++p)
        slam_seed_uniform_free(seed, static_cast<uint64_t>(first + p), map, rows, cols, &x[p], &y[p], &th[p]);
}
ORC_API void orc_seed_gaussian(uint64_t seed, long first, long n, const double* mean3, const double* sigma3, double* x,
                               double* y, double* th)
{
    for (long p = 0; p < n; ++p)
    {
        double a, b, c, d;
        slam_normal_pair(seed, SLAM_STREAM_SEED, 0, static_cast<uint64_t>(first + p), 0, &a, &b);
        slam_normal_pair(seed, SLAM_STREAM_SEED, 0, static_cast<uint64_t>(first + p), 1, &c, &d);
        x[p] = mean3[0] + a * sigma3[0];
        y[p] = mean3[1] + b * sigma3[1];
        th[p] = mean3[2] + c * sigma3[2];
    }
}

// slam/mcl.cpp:157-203 probabilistic_fitness_selection (selection only; map cloning is a memcpy).
// order = 0: the reference's sequential running sum (:170-176).  order = 1: canonical blocked order.
// canon[k] are canonical uniforms; cutoff = canon * sum (uniform_real_distribution(0, sum)).
// resampler = 0 multinomial (reference); 1 systematic (extension): one
// uniform canon[0], cutoff_k = (k + canon[0]) * (sum / N), same search rule.
ORC_API void orc_resample(long N, const double* w, const double* canon, int order, int resampler, double* prefix_out,
                          int* idx_out, double* sum_out)
{
    std::vector<double> cum(N);
    if (order == 0)
    {
        cum[0] = w[0];
        for (long i = 1; i < N; ++i) cum[i] = cum[i - 1] + w[i];
    }
    else
        blocked_prefix(w, N, cum.data());
    double sum;
    if (order == 0)
    {
        sum = 0;  // slam/mcl.cpp:160-164
        for (long i = 0; i < N; ++i) sum += w[i];
    }
    else
        sum = cum[N - 1];
    // The draws are independent of each other (slam/mcl.cpp:179-203 reads only `cumulative_sums`), so above a
    // million of them the checker runs the same loop body on contiguous slices in threads: same results,
    // and a 20 M-particle comparison takes seconds instead of half a minute of cache-missing bisections.
    auto draws = [&](long k0, long k1) {
        for (long k = k0; k < k1; ++k)
        {
            double cutoff;
            if (resampler == 0)
                cutoff = canon[k] * (sum - 0) + 0;
            else
                cutoff = (static_cast<double>(k) + canon[0]) * (sum / static_cast<double>(N));
            idx_out[k] = select_index(cum.data(), N, cutoff);
        }
    };
    const long T = N >= 1000000 ? std::min<long>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (T <= 1)
        draws(0, N);
    else
    {
        std::vector<std::thread> pool;
        for (long t = 0; t < T; ++t) pool.emplace_back(draws, N * t / T, N * (t + 1) / T);
        for (auto& th : pool) th.join();
    }
    if (prefix_out) std::memcpy(prefix_out, cum.data(), sizeof(double) * N);
    if (sum_out) *sum_out = sum;
}

// slam/mcl.cpp:232-239: sort by weight descending.  mode = 0: canonical order (weight desc, slot
// asc) — what the HIP path produces.  mode = 1: the reference's literal std::sort (unstable;
// tie order is whatever this libstdc++ does).
ORC_API void orc_sort_order(long N, const double* w_sel, int mode, int* perm_out)
{
    struct Rec
    {
        double weight;
        int slot;
    };
    std::vector<Rec> r(N);
    for (long k = 0; k < N; ++k) r[k] = {w_sel[k], static_cast<int>(k)};
    if (mode == 0)
        std::stable_sort(r.begin(), r.end(), [](const Rec& a, const Rec& b) { return a.weight > b.weight; });
    else
        std::sort(r.begin(), r.end(), [](const Rec& lhs, const Rec& rhs) { return lhs.weight > rhs.weight; });
    for (long k = 0; k < N; ++k) perm_out[k] = r[k].slot;
}

// slam/mcl.cpp:88-100 MCL::sensor_position.
ORC_API void orc_sensor_position(const double* frame3, const double* offset3, int math, double* out3)
{
    double d, theta;
    scanner_offset_to_displacement(offset3, d, theta);
    const double a = frame3[2] + theta;
    const double dx = (math ? dm_cos(a) : std::cos(a)) * d;
    const double dy = (math ? dm_sin(a) : std::sin(a)) * d;
    out3[0] = frame3[0] + dx;
    out3[1] = frame3[1] + dy;
    out3[2] = frame3[2] + offset3[2];
}

ORC_API void orc_scanner_offset(const double* offset3, double* d_theta2)
{
    scanner_offset_to_displacement(offset3, d_theta2[0], d_theta2[1]);
}

// slam/util.cpp:66-85 average_pose.
ORC_API void orc_average_pose(long N, const double* x, const double* y, const double* th, int math, double* out3)
{
    double avg_x = 0, avg_y = 0, theta_x = 0, theta_y = 0;
    for (long p = 0; p < N; ++p)
    {
        avg_x += x[p];
        avg_y += y[p];
        theta_x += math ? dm_cos(th[p]) : std::cos(th[p]);
        theta_y += math ? dm_sin(th[p]) : std::sin(th[p]);
    }
    avg_x /= N;
    avg_y /= N;
    out3[0] = avg_x;
    out3[1] = avg_y;
    out3[2] = std::atan2(theta_y, theta_x);
}

// slam/fake_lidar.cpp:15-33 on a CV_32S map (0 = wall).  hits_xy[2*i..] = hit or (-1,-1).
ORC_API int orc_fake_lidar_scan(const int* map, int rows, int cols, const double* pose3, double start, double stop,
                                double max_dist, int n_rays, int math, double* hits_xy)
{
    return math ? fake_lidar_scan<DetMath>(map, rows, cols, pose3, start, stop, max_dist, n_rays, hits_xy)
                : fake_lidar_scan<LibmMath>(map, rows, cols, pose3, start, stop, max_dist, n_rays, hits_xy);
}

// The callers' conversion of hits to (angle, dist): benchmark/mcl.cpp:37-56, apps/grid_slam.cpp:147-166.
ORC_API void orc_hits_to_scan(int n_rays, const double* hits_xy, const double* sensor3, double start, double stop,
                              double max_dist, double* angles, double* dists)
{
    const double range = stop - start;
    const double step = range / n_rays;
    for (int i = 0; i < n_rays; ++i)
    {
        double dist;
        if (hits_xy[2 * i] != -1)
        {
            const double ey = hits_xy[2 * i + 1] - sensor3[1];
            const double ex = hits_xy[2 * i] - sensor3[0];
            dist = std::sqrt(ey * ey + ex * ex);
        }
        else
            dist = max_dist;
        angles[i] = i * step - range / 2;
        dists[i] = dist;
    }
}

// slam/raycast.cpp:7-141 read-only raycasts.  kind: 0 = double map (<0.5), 1 = uchar (<128), 2 = int (==0).
ORC_API int orc_raycast(int kind, const void* map, int rows, int cols, double px, double py, double theta,
                        double max_distance_squared, double step_size, int math, double* hit_xy)
{
#define ORC_RC(MM)                                                                                                  \
    (kind == 0   ? raycast_ro<MM, double, false>(static_cast<const double*>(map), rows, cols, px, py, theta,         \
                                                 max_distance_squared, step_size, [](double v) { return v < 0.5; }, \
                                                 hit_xy[0], hit_xy[1])                                               \
     : kind == 1 ? raycast_ro<MM, unsigned char, false>(static_cast<const unsigned char*>(map), rows, cols, px, py,  \
                                                        theta, max_distance_squared, step_size,                     \
                                                        [](unsigned char v) { return v < 128; }, hit_xy[0],          \
                                                        hit_xy[1])                                                   \
                 : raycast_ro<MM, int, true>(static_cast<const int*>(map), rows, cols, px, py, theta,                \
                                             max_distance_squared, step_size, [](int v) { return !v; }, hit_xy[0],   \
                                             hit_xy[1]))
    return math ? ORC_RC(DetMath) : ORC_RC(LibmMath);
#undef ORC_RC
}

// detmath evaluated elementwise (tests/test_detmath.py measures accuracy against long double).
// fn: 0 sin, 1 cos, 2 exp, 3 log, 4 fmod(x, 2pi)
ORC_API void orc_detmath(int fn, long n, const double* in, double* out)
{
    for (long k = 0; k < n; ++k)
    {
        switch (fn)
        {
            case 0: out[k] = dm_sin(in[k]); break;
            case 1: out[k] = dm_cos(in[k]); break;
            case 2: out[k] = dm_exp(in[k]); break;
            case 3: out[k] = dm_log(in[k]); break;
            default: out[k] = dm_fmod_pos(in[k], 2 * M_PI); break;
        }
    }
}
ORC_API void orc_libm(int fn, long n, const double* in, double* out)
{
    for (long k = 0; k < n; ++k)
    {
        switch (fn)
        {
            case 0: out[k] = std::sin(in[k]); break;
            case 1: out[k] = std::cos(in[k]); break;
            case 2: out[k] = std::exp(in[k]); break;
            case 3: out[k] = std::log(in[k]); break;
            default: out[k] = std::fmod(in[k], 2 * M_PI); break;
        }
    }
}
