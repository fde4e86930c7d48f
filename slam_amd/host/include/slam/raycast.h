// raycast.h — slam/raycast.h:13-28, evaluated on the GPU.  The read-only raycasts go through a map
// handle that keeps the grid resident (slam_map_* of slam_hip.h; one small upload + one launch per
// call); raycast_mapping / measurement_model_beam work on the caller's Particle::map, one ray per
// call as in the reference (the filter itself never calls them: MCL::update runs all rays of all
// particles in its own kernels).
#pragma once
#include "pose.h"

namespace slam
{
template <typename T>
Pose raycast(const cv::Mat& map, const Pose& pose, double max_distance_squared, double step_size = 0.5);
template <>
Pose raycast<unsigned char>(const cv::Mat&, const Pose&, double, double);
template <>
Pose raycast<int>(const cv::Mat&, const Pose&, double, double);
template <>
Pose raycast<double>(const cv::Mat&, const Pose&, double, double);

// slam/raycast.cpp:143-223: walks particle.map (CV_8UC1) from particle.pose, updates it, returns the hit or {-1, -1, theta}
Pose raycast_mapping(Particle& particle, double z_squared, double z_max_squared, double step_size = 1);
// slam/raycast.cpp:225-242
double measurement_model_beam(double distance_squared, double stddev, Particle& particle, double max_distance_squared,
                              double step_size = 1);
}  // namespace slam
