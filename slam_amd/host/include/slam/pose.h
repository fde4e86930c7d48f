// pose.h — the PODs of the filter's public interface (same members as the reference's slam/pose.h:7-37).
#pragma once
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
}  // namespace slam
