// mcl.h — slam::MCL with the reference's public interface (slam/mcl.h:12-46), computed by
// libslam_hip.so on an MI355X.  Source-compatible with apps/grid_slam.cpp:100,112-135,168 and
// benchmark/mcl.cpp:29-30,58-64.
#pragma once
#include <array>
#include <tuple>
#include <vector>

#include "pose.h"

struct slam_mcl;  // include/slam_hip.h

namespace slam
{
class MCL
{
public:
    MCL(int n_particles, const cv::Size& canvas_size = {1000, 1000});
    ~MCL();
    MCL(const MCL&) = delete;
    MCL& operator=(const MCL&) = delete;

    Pose starting_pose() const;
    void predict(const Odometry& odom, const std::array<double, 4>& alphas);
    // n_threads is accepted for source compatibility and ignored (slam/mcl.cpp:108-116 is host threading)
    void update(const std::vector<std::tuple<double, double>>& scan, double stddev, double max_dist,
                const Pose& scanner_offset = {0, 0, 0}, int n_threads = -1);

    // Poses and weights of all particles, sorted by weight descending after update(); front().map is
    // the best particle's canvas (what apps/grid_slam.cpp:112 reads).  Valid until the next call.
    const std::vector<Particle>& get_particles() const;
    cv::Mat particle_map(int index) const;  // any other particle's canvas, on request

    static Pose sensor_position(const Pose& frame_position, const Pose& scanner_offset);

    slam_mcl* handle() const { return m_handle; }

private:
    slam_mcl* m_handle = nullptr;
    cv::Size m_canvas_size;
    int m_n = 0;
    mutable std::vector<Particle> m_particles;
    mutable bool m_dirty = true;
};
}  // namespace slam
