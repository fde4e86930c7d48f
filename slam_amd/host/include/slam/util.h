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
// the samplers and densities of slam/util.cpp:9-43 (host scalars; clock-seeded like the reference)
double pdf_normal_distribution(double stddev, double x);
double pdf_normal_distribution_clamp(double stddev, double x, double multiple_stddev = 4);
double pdf_triangular_distribution(double stddev, double x);
double sample_normal_distribution(double stddev);
double sample_triangular_distribution(double stddev);

inline std::tuple<int, int> pose_to_image_coordinates(const cv::Mat& map, const Pose& pose)
{
    return std::tuple<int, int>(static_cast<int>(map.rows - pose.y - 1), static_cast<int>(pose.x));
}
}  // namespace slam
