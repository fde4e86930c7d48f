// sharded_mcl.h — one shard of a multi-GPU filter, driven from C++ with RCCL called directly (one process per
// GPU).  No reference counterpart: the reference is a single process.  Same step as
// slam::MCL::predict / update (slam/mcl.cpp:41-47,102-145) on this rank's contiguous slice of the particles.
//
// Localization mode (replicated read-only map): the global resampling needs ONE exchange, an all-gather of the
// packed per-shard state [x | y | theta | w] (32 B per particle) over xGMI, after which every rank draws its own
// output slots (slam_mcl_shard_resample).  `split_exchange`: the poses (24 B, final after predict) are gathered on
// a second stream under the weighting pass, only the weights (8 B) on the critical path.
//
// SLAM mode (one canvas per particle, exactly slam::MCL; BASELINE configs[3]): canvases are copy-on-write 64x64
// tiles; after the same all-gather and slam_mcl_shard_select every rank asks the owners of its selected sources
// for their canvases, which travel as tables + every distinct tile once (grouped ncclSend / ncclRecv: xGMI is
// point to point, every peer has its own link), and slam_mcl_shard_import_tiles completes the deep copy of
// slam/mcl.cpp:147-155,205-227 across ranks.
//
// Read-out across ranks (what apps/grid_slam.cpp:112-119 reads every frame): average_pose() and best().
// Results are bit-identical for any number of ranks (the mean pose to rounding: its sums run in another order).
// Verified on hardware at world = 1 (one-rank communicator against a plain handle) and as shards emulated on one
// device; the 8-rank command in tests/host_cpp/shard_rccl.cpp is the check for a multi-GPU node.
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
    // Localization mode.  comm: an RCCL communicator of `world` ranks in which this process is `rank`; device: this
    // rank's GPU (the one the communicator was made on); map: rows x cols uint8, hit test < 128 (slam/raycast.cpp:183-184)
    ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const unsigned char* map, int device,
               uint64_t seed, bool split_exchange = false);
    // SLAM mode (map == nullptr above is NOT this: use this constructor): every particle owns a rows x cols canvas, all
    // particles start at the canvas centre (slam/mcl.cpp:16-39); tile_pool_tiles = 0: a pool that cannot run out
    struct SlamMode
    {
        int64_t tile_pool_tiles = 0;
    };
    ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const SlamMode& slam, int device,
               uint64_t seed);
    ~ShardedMCL();
    ShardedMCL(const ShardedMCL&) = delete;
    ShardedMCL& operator=(const ShardedMCL&) = delete;

    void seed_gaussian(const Pose& mean, const Pose& sigma);
    void predict(const Odometry& odom, const std::array<double, 4>& alphas);
    void update(const std::vector<std::tuple<double, double>>& scan, double stddev, double max_dist, const Pose& scanner_offset = {0, 0, 0});
    void synchronize();
    // this rank's particles (global indices rank * n_local ...), in slot order
    void get_local(std::vector<double>& x, std::vector<double>& y, std::vector<double>& theta, std::vector<double>& weight);
    void get_local_map(int64_t local_slot, std::vector<unsigned char>& canvas);  // SLAM mode

    // ---- collective read-out: EVERY rank calls these, every rank gets the same bits -------------------------------
    // slam::average_pose (slam/util.cpp:66-85) over the particles of all ranks
    Pose average_pose();
    // get_particles().front() of the whole filter (apps/grid_slam.cpp:112): the heaviest particle, the lowest global
    // slot among equals (where slam/mcl.cpp:236-238's sort puts it on one handle); canvas != nullptr (SLAM mode):
    // its rows x cols canvas, broadcast by the owner
    struct Best
    {
        Pose pose;
        double weight;
        int64_t global_slot;
        int rank;
        int64_t local_slot;
    };
    Best best(std::vector<unsigned char>* canvas = nullptr);

    int64_t n_local() const { return m_n; }
    slam_mcl* handle() const { return m_handle; }

private:
    void init(int rows, int cols, const unsigned char* map, uint64_t seed, const SlamMode* slam);
    void release() noexcept;
    void gather_poses();
    void exchange_canvases();
    void summarize(double avg[3], Best& b);
    void* scratch(int which, size_t bytes);  // grow-only device buffers of the canvas exchange
    slam_mcl* m_handle = nullptr;
    ncclComm_t m_comm;
    int m_rank, m_world, m_device;
    int64_t m_n;
    bool m_split, m_slam = false;
    int m_rows = 0, m_cols = 0, m_T = 0;
    void* m_stream = nullptr;       // hipStream_t of the filter
    void* m_comm_stream = nullptr;  // hipStream_t of the early pose gather (split exchange)
    void *m_ev_predict = nullptr, *m_ev_poses = nullptr;  // hipEvent_t
    double* m_gathered = nullptr;   // [world][4][n_local], or [world][3][n_local] + [world][n_local] when split
    double* m_gathered_w = nullptr;
    double* m_small = nullptr;      // device: summaries / counts of the small collectives
    bool m_poses_in_flight = false;
    enum { N_SCRATCH = 8 };
    void* m_scr[N_SCRATCH] = {};
    size_t m_scr_bytes[N_SCRATCH] = {};
};
}  // namespace slam
