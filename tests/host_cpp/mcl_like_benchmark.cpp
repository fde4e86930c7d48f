// mcl_like_benchmark.cpp — the call sequence of the reference's benchmark/mcl.cpp:14-64 written against
// the drop-in C++ headers (slam_amd/host/include/slam/*.h).  Differences from the reference driver:
// the map comes from a raw int32 file instead of cv::imread (no OpenCV / PNG decoder in this image),
// the iteration count is an argument, and the final particle set is printed for the test to check.
//   usage: mcl_like_benchmark map_i32.raw rows cols iterations
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

#include "slam/fake_lidar.h"
#include "slam/raycast.h"
#include "slam/mcl.h"
#include "slam/motion.h"
#include "slam/pose.h"
#include "slam/util.h"

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), iters = std::atoi(argv[4]);
    cv::Mat map(cv::Size(cols, rows), CV_32S, cv::Scalar(0));
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(&map.at<int>(0, 0), sizeof(int), static_cast<size_t>(rows) * cols, f) != static_cast<size_t>(rows) * cols) return 3;
    std::fclose(f);

    constexpr double FAKE_LIDAR_STDDEV = 5;
    constexpr double FAKE_LIDAR_MAX_DIST = 500;
    slam::FakeLidar fake_lidar(0, M_PI, FAKE_LIDAR_MAX_DIST, FAKE_LIDAR_STDDEV, 90);
    constexpr slam::Pose SCANNER_OFFSET = {0, 30, 0};
    constexpr unsigned N_PARTICLES = 25;
    slam::MCL mcl(N_PARTICLES, {1600, 900});

    slam::Pose real_position{400, 400, M_PI};
    const slam::Pose sensor_position = slam::MCL::sensor_position(real_position, SCANNER_OFFSET);
    const std::vector<slam::Pose> hits = fake_lidar.scan(map, sensor_position);
    std::vector<std::tuple<double, double>> scans;
    const double range = fake_lidar.stop - fake_lidar.start;
    const double step = range / fake_lidar.n_rays;
    int i = 0;
    for (const slam::Pose& hit : hits)
    {
        double dist;
        if (hit.x != -1)
            dist = std::sqrt(std::pow(hit.y - sensor_position.y, 2) + std::pow(hit.x - sensor_position.x, 2));
        else
            dist = fake_lidar.max_dist;
        const double angle = i++ * step - range / 2;
        scans.push_back({angle, dist});
    }
    for (int it = 0; it < iters; ++it)
    {
        const slam::Odometry odom = {2.5, 0.02, 0.02};
        mcl.predict(odom, {0.0005, 0.0005, 0.01, 0.01});
        real_position = slam::sample_motion_model_odometry(odom, real_position);
        mcl.update(scans, FAKE_LIDAR_STDDEV, fake_lidar.max_dist, SCANNER_OFFSET);
    }
    const std::vector<slam::Particle>& ps = mcl.get_particles();
    for (const slam::Particle& p : ps)
    {
        unsigned long long b[4];
        std::memcpy(&b[0], &p.pose.x, 8), std::memcpy(&b[1], &p.pose.y, 8), std::memcpy(&b[2], &p.pose.theta, 8), std::memcpy(&b[3], &p.weight, 8);
        std::printf("P %016llx %016llx %016llx %016llx\n", b[0], b[1], b[2], b[3]);
    }
    unsigned long long sum = 0;
    const cv::Mat best = ps.front().map;
    for (int r = 0; r < best.rows; ++r)
        for (int c = 0; c < best.cols; ++c) sum = sum * 1099511628211ull + best.at<unsigned char>(r, c);
    const slam::Pose avg = slam::average_pose(ps);
    std::printf("MAP %d %d %016llx\nAVG %.17g %.17g %.17g\nREAL %.17g %.17g %.17g\n", best.rows, best.cols, sum, avg.x, avg.y, avg.theta,
                real_position.x, real_position.y, real_position.theta);
    // the other free functions of slam/motion.h and slam/util.h (deterministic with zero alphas / fixed arguments)
    const slam::Pose v = slam::sample_motion_model_velocity({2.0, 0.5}, {10.0, 20.0, 0.3}, 0.25);
    const double tri = slam::sample_triangular_distribution(1.5);
    std::printf("VEL %.17g %.17g %.17g\nPDF %.17g %.17g %.17g\nTRI %d\n", v.x, v.y, v.theta, slam::pdf_normal_distribution(2.0, 0.7),
                slam::pdf_normal_distribution_clamp(2.0, 9.0), slam::pdf_triangular_distribution(1.5, 0.5),
                (std::fabs(tri) <= 1.5 * (std::sqrt(6.0) / 2 + 1)) ? 1 : 0);
    // slam/raycast.h:19-28 on a caller-owned canvas: a fan of rays drawn twice (the second fan meets the first one's walls)
    {
        slam::Particle part{{60.5, 40.25, 0.0}, 1.0, cv::Mat(cv::Size(160, 120), CV_8UC1, cv::Scalar(128))};
        std::printf("RCM");
        for (int round = 0; round < 2; ++round)
            for (int k = 0; k < 12; ++k)
            {
                part.pose.theta = 0.3 + 0.5 * k;
                const double z = 20.0 + 3.0 * k;
                if (k % 3 == 2)
                {
                    const double w = slam::measurement_model_beam(z * z, 2.0, part, 70.0 * 70.0);
                    std::printf(" %.17g", w);
                }
                else
                {
                    const slam::Pose hit = slam::raycast_mapping(part, k % 4 == 1 ? 70.0 * 70.0 : z * z, 70.0 * 70.0);
                    std::printf(" %.17g %.17g", hit.x, hit.y);
                }
            }
        unsigned long long hsum = 0;
        for (int r = 0; r < part.map.rows; ++r)
            for (int c = 0; c < part.map.cols; ++c) hsum = hsum * 1099511628211ull + part.map.at<unsigned char>(r, c);
        std::printf("\nRCMMAP %016llx\n", hsum);
    }
    return 0;
}
