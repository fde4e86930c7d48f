// util.h — caller-side helpers the apps use next to the filter (slam/util.h:13-49).
//
// Inside the reference's tree the build defines SLAM_REFERENCE_UTIL_H as the path of
// the reference's own slam/util.h and that file is used — the planners and trees need all of it
// (Coordinate, random_point, adjacency_*, euclidean_distance*, ...) and its util.cpp stays in the
// build; on its own this file declares the part the filter's callers use and slam_host.cpp defines it.
#pragma once
#ifdef SLAM_REFERENCE_UTIL_H
#define SLAM_SHIM_REFERENCE_UTIL 1
#include SLAM_REFERENCE_UTIL_H
#endif
#ifndef SLAM_SHIM_REFERENCE_UTIL
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

struct Coordinate  // slam/util.h:21-28
{
    int i;
    int j;
    inline bool operator==(const Coordinate& other) const { return i == other.i && j == other.j; }
    inline bool operator!=(const Coordinate& other) const { return !(*this == other); }
};

inline std::tuple<int, int> pose_to_image_coordinates(const cv::Mat& map, const Pose& pose)
{
    return std::tuple<int, int>(static_cast<int>(map.rows - pose.y - 1), static_cast<int>(pose.x));
}

inline bool within_boundaries(const cv::Mat& map, const int i, const int j)  // slam/util.h:45-49
{
    return static_cast<unsigned>(i) < static_cast<unsigned>(map.rows) && static_cast<unsigned>(j) < static_cast<unsigned>(map.cols);
}
}  // namespace slam
#endif
