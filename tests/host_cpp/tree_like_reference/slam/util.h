// Stand-in for "the other slam/util.h further down the include path" (see pose.h next to this file).
#pragma once
#include <tuple>
#include <vector>
#include "pose.h"
#define TREE_UTIL_H_SEEN 1
namespace slam
{
double pdf_normal_distribution(double stddev, double x);
double pdf_normal_distribution_clamp(double stddev, double x, double multiple_stddev = 4);
double sample_normal_distribution(double stddev);
double normalize_angle(double angle);
Pose average_pose(const std::vector<Particle>& particles);
struct Coordinate { int i, j; };
inline int planner_only_helper(const Coordinate& c) { return c.i + c.j; }
inline std::tuple<int, int> pose_to_image_coordinates(const cv::Mat& map, const Pose& pose)
{
    return std::tuple<int, int>(map.rows - pose.y - 1, pose.x);
}
}  // namespace slam
