// slam_host.cpp — C++ shim: the reference's public classes over the C ABI of libslam_hip.so.
// Errors from the library are fatal here (the reference's interface returns void everywhere).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/slam_hip.h"
#include "../include/slam/fake_lidar.h"
#include "../include/slam/mcl.h"
#include "../include/slam/motion.h"
#include "../include/slam/raycast.h"
#include "../include/slam/util.h"

namespace
{
void check(int rc, const char* what)
{
    if (rc != SLAM_OK) throw std::runtime_error(std::string(what) + ": " + slam_last_error());
}
}  // namespace

namespace slam
{
MCL::MCL(int n_particles, const cv::Size& canvas_size) : m_canvas_size(canvas_size), m_n(n_particles)
{
    slam_mcl_params p{};
    p.struct_size = sizeof(p);
    p.mode = SLAM_MODE_SLAM;
    p.n_particles = n_particles;
    p.rows = canvas_size.height;
    p.cols = canvas_size.width;
    p.device = std::getenv("SLAM_DEVICE") ? std::atoi(std::getenv("SLAM_DEVICE")) : 0;
    p.noise = SLAM_NOISE_PHILOX;
    // the reference seeds from the clock (slam/util.cpp:30); SLAM_SEED makes a run repeatable
    p.seed = std::getenv("SLAM_SEED") ? std::strtoull(std::getenv("SLAM_SEED"), nullptr, 0)
                                      : static_cast<uint64_t>(std::chrono::system_clock::now().time_since_epoch().count());
    p.sort = SLAM_SORT_FULL;
    // same results either way: copy-on-write tiles by default, SLAM_MAPS=dense|tiled forces one
    const char* maps = std::getenv("SLAM_MAPS");
    p.map_storage = !maps ? SLAM_MAPS_AUTO : (std::string(maps) == "tiled" ? SLAM_MAPS_TILED : SLAM_MAPS_DENSE);
    check(slam_mcl_create(&p, &m_handle), "slam_mcl_create");
}

MCL::~MCL() { slam_mcl_destroy(m_handle); }

Pose MCL::starting_pose() const { return {m_canvas_size.width / 2.0, m_canvas_size.height / 2.0, M_PI / 2}; }

void MCL::predict(const Odometry& odom, const std::array<double, 4>& alphas)
{
    const double o[3] = {odom.rotation_1, odom.translation, odom.rotation_2};
    check(slam_mcl_predict(m_handle, o, alphas.data()), "slam_mcl_predict");
    m_dirty = true;
}

void MCL::update(const std::vector<std::tuple<double, double>>& scan, double stddev, double max_dist,
                 const Pose& scanner_offset, int /*n_threads*/)
{
    std::vector<double> ang(scan.size()), dist(scan.size());
    for (size_t k = 0; k < scan.size(); ++k) std::tie(ang[k], dist[k]) = scan[k];
    const double off[3] = {scanner_offset.x, scanner_offset.y, scanner_offset.theta};
    check(slam_mcl_update(m_handle, ang.data(), dist.data(), static_cast<int32_t>(scan.size()), stddev, max_dist, off),
          "slam_mcl_update");
    m_dirty = true;
}

cv::Mat MCL::particle_map(int index) const
{
    cv::Mat m(m_canvas_size, CV_8UC1, cv::Scalar(0));
    check(slam_mcl_get_map_u8(m_handle, index, &m.at<unsigned char>(0, 0)), "slam_mcl_get_map_u8");
    return m;
}

const std::vector<Particle>& MCL::get_particles() const
{
    if (m_dirty)
    {
        std::vector<double> x(m_n), y(m_n), th(m_n), w(m_n);
        check(slam_mcl_get_poses(m_handle, x.data(), y.data(), th.data(), 0, m_n), "slam_mcl_get_poses");
        check(slam_mcl_get_weights(m_handle, w.data(), 0, m_n), "slam_mcl_get_weights");
        m_particles.assign(m_n, Particle{});
        for (int k = 0; k < m_n; ++k)
        {
            m_particles[k].pose = {x[k], y[k], th[k]};
            m_particles[k].weight = w[k];
        }
        m_particles.front().map = particle_map(0);
        m_dirty = false;
    }
    return m_particles;
}

Pose MCL::sensor_position(const Pose& frame_position, const Pose& scanner_offset)
{
    // slam/mcl.cpp:80-100; the squared distance passes through an int (slam/util.h:138-146)
    const double ddx = 0 - scanner_offset.x, ddy = 0 - scanner_offset.y;
    const double d = std::sqrt(static_cast<double>(static_cast<int>(ddx * ddx + ddy * ddy)));
    const double theta = std::atan2(scanner_offset.y, scanner_offset.x);
    Pose ret = frame_position;
    ret.x += std::cos(frame_position.theta + theta) * d;
    ret.y += std::sin(frame_position.theta + theta) * d;
    ret.theta += scanner_offset.theta;
    return ret;
}

#ifndef SLAM_SHIM_REFERENCE_UTIL  // inside the reference's tree its own util.cpp defines these
// ---- caller-side scalar helpers of slam/util.h:13-27 (declared there; the values are those of slam/util.cpp:9-51) ----
namespace
{
constexpr double kTurn = 2 * M_PI;

// The reference draws every sample from an engine it has just re-seeded from the wall clock (slam/util.cpp:29-30,38-39):
// one helper does the seeding, the two samplers differ only in what they draw from it.
std::default_random_engine& engine_seeded_from_the_clock()
{
    static thread_local std::default_random_engine engine;
    engine.seed(std::chrono::system_clock::now().time_since_epoch().count());
    return engine;
}
}  // namespace

double normalize_angle(double angle)
{
    const double wrapped = std::fmod(angle, kTurn);  // (-2 pi, 2 pi); only the upper half turn is folded down
    return wrapped > M_PI ? wrapped - kTurn : wrapped;
}

double pdf_normal_distribution(double stddev, double x)
{
    const double peak = 1.0 / (stddev * std::sqrt(kTurn)), in_sigmas = x / stddev;
    return peak * std::exp(-0.5 * std::pow(in_sigmas, 2));
}

double pdf_normal_distribution_clamp(double stddev, double x, double multiple_stddev)
{
    const bool beyond = std::fabs(x) > multiple_stddev * stddev;
    return beyond ? 0 : pdf_normal_distribution(stddev, x);
}

double pdf_triangular_distribution(double stddev, double x)
{
    const double var = std::pow(stddev, 2);
    const double tent = 1 / std::sqrt(6 * var) - std::fabs(x) / 6 / var;
    return std::max(0.0, tent);
}

double sample_normal_distribution(double stddev)
{
    std::normal_distribution<double> gauss(0, stddev);
    return gauss(engine_seeded_from_the_clock());
}

double sample_triangular_distribution(double stddev)
{
    std::default_random_engine& engine = engine_seeded_from_the_clock();  // ONE seeding, two draws
    std::uniform_real_distribution<double> flat(-stddev, stddev);
    const double a = flat(engine);
    const double b = flat(engine);
    return std::sqrt(6) / 2 * a + b;
}

#endif

Pose sample_motion_model_odometry(const Odometry& odom, const Pose& pose, std::array<double, 4> alphas)
{
    // slam/motion.cpp:9-32 for ONE pose (the ground-truth robot of the callers), noise as in slam/util.cpp:27-34
    auto sample = sample_normal_distribution;
    const double r1 = odom.rotation_1 - sample(std::sqrt(alphas[0] * std::pow(odom.rotation_1, 2) + alphas[1] * std::pow(odom.translation, 2)));
    const double t = odom.translation - sample(std::sqrt(alphas[2] * std::pow(odom.translation, 2) +
                                                         alphas[3] * (std::pow(odom.rotation_1, 2) + std::pow(odom.rotation_2, 2))));
    const double r2 = odom.rotation_2 - sample(std::sqrt(alphas[0] * std::pow(odom.rotation_2, 2) + alphas[1] * std::pow(odom.translation, 2)));
    return {pose.x + t * std::cos(pose.theta + r1), pose.y + t * std::sin(pose.theta + r1), normalize_angle(pose.theta + r1 + r2)};
}

Pose sample_motion_model_velocity(const Velocity& vel, const Pose& pose, double dt, std::array<double, 6> alphas)
{
    // slam/motion.cpp:34-56 for ONE pose; the per-particle form is slam_mcl_predict_velocity
    constexpr double EPSILON = 1e-6;
    auto sample = sample_normal_distribution;
    const double v = vel.v + sample(std::sqrt(alphas[0] * std::pow(vel.v, 2) + alphas[1] * std::pow(vel.w, 2)));
    const double w = vel.w + sample(std::sqrt(alphas[2] * std::pow(vel.w, 2) + alphas[3] * std::pow(vel.v, 2)));
    const double gamma = sample(std::sqrt(alphas[4] * std::pow(vel.v, 2) + alphas[5] * std::pow(vel.w, 2)));
    const double q = v / (w == 0 ? 1 : w);
    return {pose.x - q * (std::sin(pose.theta) + std::sin(pose.theta + (w + EPSILON) * dt)),
            pose.y + q * (std::cos(pose.theta) - std::cos(pose.theta + (w + EPSILON) * dt)),
            normalize_angle(pose.theta + w * dt + gamma * dt)};
}

#ifndef SLAM_SHIM_REFERENCE_UTIL
Pose average_pose(const std::vector<Particle>& particles)
{
    // slam/util.cpp:66-85: arithmetic mean of the positions, direction of the summed unit headings; sums in particle order
    double sum[4] = {0, 0, 0, 0};  // x, y, cos, sin
    for (const Particle& each : particles)
    {
        const double term[4] = {each.pose.x, each.pose.y, std::cos(each.pose.theta), std::sin(each.pose.theta)};
        for (int q = 0; q < 4; ++q) sum[q] += term[q];
    }
    const double count = particles.size();
    return {sum[0] / count, sum[1] / count, std::atan2(sum[3], sum[2])};
}

#endif

// ---- scan synthesis and raycasts on the GPU -------------------------------------------------------------
namespace
{
int shim_device() { return std::getenv("SLAM_DEVICE") ? std::atoi(std::getenv("SLAM_DEVICE")) : 0; }

// One resident grid per source cv::Mat: the converted bytes are uploaded when the handle is made and
// again only when the caller's matrix has changed (exact comparison against a host copy: a 4 MB memcmp
// is far cheaper than the conversion, the allocation and the upload it saves).
struct MapCache
{
    const void* key = nullptr;
    int rows = 0, cols = 0, elem = 0, kind = 0;
    std::vector<unsigned char> source;  // the caller's bytes at upload time
    slam_map* map = nullptr;
    ~MapCache() { slam_map_destroy(map); }
};

template <class T, class F>
slam_map* resident(MapCache& c, const cv::Mat& m, int kind, F to_u8)
{
    const size_t n = static_cast<size_t>(m.rows) * m.cols, row_bytes = static_cast<size_t>(m.cols) * sizeof(T);
    if (m.rows <= 0 || m.cols <= 0) throw std::invalid_argument("slam shim: empty matrix");
    // row by row through ptr(i): a cv::Mat may be a view (ROI) whose rows are not adjacent in memory
    const bool same_shape = c.map && c.rows == m.rows && c.cols == m.cols && c.elem == (int)sizeof(T) && c.kind == kind;
    bool same = same_shape && c.source.size() == n * sizeof(T);
    for (int i = 0; same && i < m.rows; ++i) same = std::memcmp(c.source.data() + i * row_bytes, m.ptr<T>(i), row_bytes) == 0;
    if (same) return c.map;
    std::vector<uint8_t> u8(n);
    for (int i = 0; i < m.rows; ++i)
    {
        const T* row = m.ptr<T>(i);
        for (int j = 0; j < m.cols; ++j) u8[static_cast<size_t>(i) * m.cols + j] = to_u8(row[j]);
    }
    if (same_shape)
        check(slam_map_upload(c.map, u8.data()), "slam_map_upload");
    else
    {
        slam_map_destroy(c.map);
        c.map = nullptr;
        check(slam_map_create(shim_device(), u8.data(), m.rows, m.cols, &c.map), "slam_map_create");
    }
    c.rows = m.rows;
    c.cols = m.cols;
    c.elem = sizeof(T);
    c.kind = kind;
    c.source.resize(n * sizeof(T));
    for (int i = 0; i < m.rows; ++i) std::memcpy(c.source.data() + i * row_bytes, m.ptr<T>(i), row_bytes);
    return c.map;
}

thread_local MapCache g_lidar_map, g_raycast_map[3], g_canvas_map;
}  // namespace

std::vector<Pose> FakeLidar::scan(const cv::Mat& map, const Pose& pose) const
{
    slam_map* m = resident<int>(g_lidar_map, map, 2, [](int v) { return static_cast<uint8_t>(v ? 1 : 0); });
    const double p3[3] = {pose.x, pose.y, pose.theta};
    std::vector<double> hits(2 * static_cast<size_t>(n_rays));
    check(slam_map_fake_lidar_scan(m, p3, 1, start, stop, max_dist, n_rays, hits.data()), "slam_map_fake_lidar_scan");
    // slam/fake_lidar.cpp:23-30: every returned pose carries the ray's own angle
    const double range = stop - start, step = range / n_rays;
    std::vector<Pose> out(n_rays);
    double theta = pose.theta;
    theta -= range / 2;
    for (int r = 0; r < n_rays; ++r)
    {
        out[r] = {hits[2 * r], hits[2 * r + 1], theta};
        theta += step;
    }
    return out;
}

namespace
{
template <class T, class F>
Pose raycast_one(MapCache& c, const cv::Mat& map, const Pose& pose, double max2, double step, int kind, F to_u8)
{
    slam_map* m = resident<T>(c, map, kind, to_u8);
    const double p3[3] = {pose.x, pose.y, pose.theta};
    double hit[2];
    check(slam_map_raycast_batch(m, kind, p3, 1, max2, step, hit), "slam_map_raycast_batch");
    return {hit[0], hit[1], pose.theta};
}
}  // namespace

template <>
Pose raycast<unsigned char>(const cv::Mat& map, const Pose& pose, double max2, double step)
{
    return raycast_one<unsigned char>(g_raycast_map[0], map, pose, max2, step, 1, [](unsigned char v) { return v; });
}
template <>
Pose raycast<double>(const cv::Mat& map, const Pose& pose, double max2, double step)
{
    return raycast_one<double>(g_raycast_map[1], map, pose, max2, step, 1,
                               [](double v) { return static_cast<uint8_t>(v < 0.5 ? 0 : 255); });
}
template <>
Pose raycast<int>(const cv::Mat& map, const Pose& pose, double max2, double step)
{
    return raycast_one<int>(g_raycast_map[2], map, pose, max2, step, 2, [](int v) { return static_cast<uint8_t>(v ? 1 : 0); });
}

// slam/raycast.cpp:143-223 on the caller's canvas: walked and updated on the GPU, read back.  The canvas stays resident
// between calls (one handle per thread): consecutive rays on one particle's map, which is how slam/mcl.cpp:69-74 calls
// this, upload nothing (the host copy kept at the last download equals the caller's bytes) and each call costs one
// launch and one download.  (A caller with many rays for one canvas should hand them to slam_map_raycast_mapping in one
// call; the filter's own path is MCL::update.)
Pose raycast_mapping(Particle& particle, double z_squared, double z_max_squared, double step_size)
{
    cv::Mat& canvas = particle.map;
    if (canvas.type() != CV_8UC1) throw std::invalid_argument("slam::raycast_mapping: the canvas must be CV_8UC1 (slam/mcl.cpp:30)");
    slam_map* m = resident<unsigned char>(g_canvas_map, canvas, 3, [](unsigned char v) { return v; });
    const double p3[3] = {particle.pose.x, particle.pose.y, particle.pose.theta};
    double hit[2] = {-1, -1};
    check(slam_map_raycast_mapping(m, p3, &z_squared, 1, z_max_squared, step_size, hit), "slam_map_raycast_mapping");
    // the updated canvas back into the caller's matrix, row by row, and into the cache's picture of it
    std::vector<unsigned char>& now = g_canvas_map.source;
    check(slam_map_download(m, now.data()), "slam_map_download");
    for (int i = 0; i < canvas.rows; ++i) std::memcpy(canvas.ptr<unsigned char>(i), now.data() + static_cast<size_t>(i) * canvas.cols, canvas.cols);
    return {hit[0], hit[1], particle.pose.theta};
}

double measurement_model_beam(double distance_squared, double stddev, Particle& particle, double max_distance_squared,
                              double step_size)
{
    const Pose hit = raycast_mapping(particle, distance_squared, max_distance_squared, step_size);
    constexpr double EPSILON = 0.1;
    if (hit.x == -1)  // no hit
        return pdf_normal_distribution_clamp(stddev, std::sqrt(distance_squared) - std::sqrt(max_distance_squared)) + EPSILON;
    const double ddx = hit.x - particle.pose.x, ddy = hit.y - particle.pose.y;
    const double distance = std::sqrt(ddx * ddx + ddy * ddy);
    return pdf_normal_distribution_clamp(stddev, distance - std::sqrt(distance_squared)) + EPSILON;
}
}  // namespace slam
