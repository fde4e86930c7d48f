// util.h — caller-side helpers the apps use next to the filter (reference: slam/util.h:19,30-33).
#pragma once
#include <cmath>
#include <tuple>
#include <vector>

#include "pose.h"

namespace slam
{
Pose average_pose(const std::vector<Particle>& particles);  // slam/util.cpp:66-85 (host reduction over the given vector)
double normalize_angle(double angle);                       // slam/util.cpp:45-51

inline std::tuple<int, int> pose_to_image_coordinates(const cv::Mat& map, const Pose& pose)
{
    return std::tuple<int, int>(static_cast<int>(map.rows - pose.y - 1), static_cast<int>(pose.x));
}
}  // namespace slam
