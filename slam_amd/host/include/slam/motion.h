// motion.h — slam/motion.h:9-11.  Single-pose sampler used by callers for the ground-truth robot
// (apps/grid_slam.cpp:134-135, benchmark/mcl.cpp:62); the per-particle sampler runs on the GPU inside
// MCL::predict.
#pragma once
#include <array>

#include "pose.h"

namespace slam
{
Pose sample_motion_model_odometry(const Odometry& odom, const Pose& pose, std::array<double, 4> alphas = {0, 0, 0, 0});
Pose sample_motion_model_velocity(const Velocity& vel, const Pose& pose, double dt,
                                  std::array<double, 6> alphas = {0, 0, 0, 0, 0, 0});  // slam/motion.cpp:34-56
}  // namespace slam
