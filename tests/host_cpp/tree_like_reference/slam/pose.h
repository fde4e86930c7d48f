// Stand-in for "the other slam/pose.h further down the include path" in the include-order test
// (tests/test_host_cpp.py::test_shim_headers_defer_to_headers_behind_them).  Written for the test:
// same type names as the filter's interface, plus a marker the test checks for.
#pragma once
#include "../../../../slam_amd/host/include/slam/cv_compat.h"
#define TREE_POSE_H_SEEN 1
namespace slam
{
struct Pose { double x, y, theta; };
struct Odometry { double rotation_1, translation, rotation_2; };
struct Velocity { double v, w; };
struct Particle { Pose pose; double weight; cv::Mat map; };
struct Box { int start_i, start_j, stop_i, stop_j; };
}  // namespace slam
