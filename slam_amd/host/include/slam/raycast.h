// raycast.h — slam/raycast.h:13-17: read-only raycasts, evaluated on the GPU (one ray per call here;
// batches go through slam_raycast_u8 / slam_fake_lidar_scan of slam_hip.h).
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
}  // namespace slam
