// fake_lidar.h — slam/fake_lidar.h:10-23; scan() runs raycast<int> for all rays on the GPU.
#pragma once
#include <vector>

#include "pose.h"

namespace slam
{
class FakeLidar
{
public:
    FakeLidar(double start, double stop, double max_dist, double stddev, int n_rays)
        : start(start), stop(stop), max_dist(max_dist), stddev(stddev), n_rays(n_rays) {}
    std::vector<Pose> scan(const cv::Mat& map /* CV_32S, 0 = wall */, const Pose& pose) const;

    double start;
    double stop;
    double max_dist;
    double stddev;
    int n_rays;
};
}  // namespace slam
