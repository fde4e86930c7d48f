// pose.h — the PODs of the filter's public interface (slam/pose.h:7-45).
//
// Inside the reference's tree the build defines SLAM_REFERENCE_POSE_H as the path of
// the reference's own slam/pose.h and that file is used, so that the planners, the trees and the filter
// share ONE definition of Pose, Box, ...; on its own this file defines the same members itself.
#pragma once
#ifdef SLAM_REFERENCE_POSE_H
#define SLAM_SHIM_REFERENCE_POSE 1
#include SLAM_REFERENCE_POSE_H
#endif
#ifndef SLAM_SHIM_REFERENCE_POSE
#include "cv_compat.h"

namespace slam
{
struct Pose
{
    double x;
    double y;
    double theta;
};
inline bool operator==(const Pose& a, const Pose& b) { return a.x == b.x && a.y == b.y && a.theta == b.theta; }

struct Odometry
{
    double rotation_1;
    double translation;
    double rotation_2;
};

struct Velocity
{
    double v;
    double w;
};

struct Particle
{
    Pose pose;
    double weight;
    cv::Mat map;  // filled for the best particle (front()) after update(); others on request
};

struct Box  // slam/pose.h:39-45 (planner-side helper; here for source compatibility)
{
    int start_i;
    int start_j;
    int stop_i;
    int stop_j;
};
}  // namespace slam
#endif
