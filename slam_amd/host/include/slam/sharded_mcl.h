// sharded_mcl.h — one shard of a multi-GPU localization filter, driven from C++ with RCCL called
// directly (one process per GPU).  No reference counterpart: the reference is a single
// process.  Same step as slam::MCL::predict / update (slam/mcl.cpp:41-47,102-145) on this rank's
// contiguous slice of the particles against the replicated map; the global resampling needs ONE
// exchange, an all-gather of the packed per-shard state [x | y | theta | w] (32 B per particle) over
// xGMI, after which every rank draws its own output slots (slam_mcl_shard_resample).  Results are
// bit-identical for any number of ranks.  `split_exchange`: the poses (24 B, final after predict) are
// gathered on a second stream under the weighting pass, only the weights (8 B) on the critical path.
#pragma once
#include <array>
#include <cstdint>
#include <tuple>
#include <vector>

#include "pose.h"

struct slam_mcl;                      // include/slam_hip.h
typedef struct ncclComm* ncclComm_t;  // rccl/rccl.h

namespace slam
{
class ShardedMCL
{
public:
    // comm: an RCCL communicator of `world` ranks in which this process is `rank`; device: this rank's GPU
    // (the one the communicator was made on); map: rows x cols uint8, hit test < 128 (slam/raycast.cpp:183-184)
    ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const unsigned char* map, int device,
               uint64_t seed, bool split_exchange = false);
    ~ShardedMCL();
    ShardedMCL(const ShardedMCL&) = delete;
    ShardedMCL& operator=(const ShardedMCL&) = delete;

    void seed_gaussian(const Pose& mean, const Pose& sigma);
    void predict(const Odometry& odom, const std::array<double, 4>& alphas);
    void update(const std::vector<std::tuple<double, double>>& scan, double stddev, double max_dist, const Pose& scanner_offset = {0, 0, 0});
    void synchronize();
    // this rank's particles (global indices rank * n_local ...), in slot order
    void get_local(std::vector<double>& x, std::vector<double>& y, std::vector<double>& theta, std::vector<double>& weight);
    int64_t n_local() const { return m_n; }
    slam_mcl* handle() const { return m_handle; }

private:
    void gather_poses();
    slam_mcl* m_handle = nullptr;
    ncclComm_t m_comm;
    int m_rank, m_world, m_device;
    int64_t m_n;
    bool m_split;
    void* m_stream = nullptr;       // hipStream_t of the filter
    void* m_comm_stream = nullptr;  // hipStream_t of the early pose gather (split exchange)
    void *m_ev_predict = nullptr, *m_ev_poses = nullptr;  // hipEvent_t
    double* m_gathered = nullptr;   // [world][4][n_local], or [world][3][n_local] + [world][n_local] when split
    double* m_gathered_w = nullptr;
    bool m_poses_in_flight = false;
};
}  // namespace slam
