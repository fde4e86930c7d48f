// slam_shard_rccl.cpp — slam::ShardedMCL: the sharded filter step in C++ with RCCL called directly
// (see slam/sharded_mcl.h).  Plain host C++: HIP runtime API for the streams and the exchange buffer,
// RCCL for the collective, everything else through the C ABI of libslam_hip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>

#include "../../../include/slam_hip.h"
#include "../include/slam/sharded_mcl.h"

namespace
{
void check(int rc, const char* what)
{
    if (rc != SLAM_OK) throw std::runtime_error(std::string(what) + ": " + slam_last_error());
}
void hip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

namespace slam
{
ShardedMCL::ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const unsigned char* map,
                       int device, uint64_t seed, bool split_exchange)
    : m_comm(comm), m_rank(rank), m_world(world), m_device(device), m_n(n_local), m_split(split_exchange)
{
    hip(hipSetDevice(device), "hipSetDevice");
    hipStream_t s = nullptr, c = nullptr;
    hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    m_stream = s;
    slam_mcl_params p{};
    p.struct_size = sizeof(p);
    p.mode = SLAM_MODE_LOCALIZATION;
    p.n_particles = n_local;
    p.rows = rows;
    p.cols = cols;
    p.device = device;
    p.noise = SLAM_NOISE_PHILOX;
    p.seed = seed;
    p.prefix_order = SLAM_PREFIX_BLOCKED;  // the canonical order: independent of the number of ranks
    p.sort = SLAM_SORT_NONE;               // no caller needs a global sort (apps/grid_slam.cpp:112-119)
    p.global_first = (int64_t)rank * n_local;
    p.global_n = (int64_t)world * n_local;
    p.stream = m_stream;
    p.async_launch = 1;  // the step is enqueued; synchronize() / get_local() wait
    check(slam_mcl_create(&p, &m_handle), "slam_mcl_create");
    check(slam_mcl_set_map_u8(m_handle, map, rows, cols), "slam_mcl_set_map_u8");
    const size_t n = (size_t)n_local, w = (size_t)world;
    if (m_split)
    {
        hip(hipStreamCreateWithFlags(&c, hipStreamNonBlocking), "hipStreamCreate");
        m_comm_stream = c;
        hipEvent_t a = nullptr, b = nullptr;
        hip(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
        hip(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
        m_ev_predict = a;
        m_ev_poses = b;
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered), sizeof(double) * 3 * n * w), "hipMalloc");
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered_w), sizeof(double) * n * w), "hipMalloc");
    }
    else
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered), sizeof(double) * 4 * n * w), "hipMalloc");
}

ShardedMCL::~ShardedMCL()
{
    hipSetDevice(m_device);
    if (m_stream) hipStreamSynchronize(static_cast<hipStream_t>(m_stream));
    if (m_comm_stream) hipStreamSynchronize(static_cast<hipStream_t>(m_comm_stream));
    slam_mcl_destroy(m_handle);
    if (m_gathered) hipFree(m_gathered);
    if (m_gathered_w) hipFree(m_gathered_w);
    if (m_ev_predict) hipEventDestroy(static_cast<hipEvent_t>(m_ev_predict));
    if (m_ev_poses) hipEventDestroy(static_cast<hipEvent_t>(m_ev_poses));
    if (m_comm_stream) hipStreamDestroy(static_cast<hipStream_t>(m_comm_stream));
    if (m_stream) hipStreamDestroy(static_cast<hipStream_t>(m_stream));
}

void ShardedMCL::seed_gaussian(const Pose& mean, const Pose& sigma)
{
    const double m[3] = {mean.x, mean.y, mean.theta}, s[3] = {sigma.x, sigma.y, sigma.theta};
    check(slam_mcl_seed_gaussian(m_handle, m, s), "slam_mcl_seed_gaussian");
}

void ShardedMCL::gather_poses()
{
    // [x | y | theta] of every shard, on the second stream: it waits for predict (and through it for the
    // previous resample, the last reader of the buffer); the resample of this step waits for m_ev_poses
    slam_mcl_device_state st{};
    check(slam_mcl_device_state_get(m_handle, &st), "slam_mcl_device_state_get");
    hipStream_t s = static_cast<hipStream_t>(m_stream), c = static_cast<hipStream_t>(m_comm_stream);
    hip(hipEventRecord(static_cast<hipEvent_t>(m_ev_predict), s), "hipEventRecord");
    hip(hipStreamWaitEvent(c, static_cast<hipEvent_t>(m_ev_predict), 0), "hipStreamWaitEvent");
    nccl(ncclAllGather(st.state_dev, m_gathered, 3 * (size_t)m_n, ncclDouble, m_comm, c), "ncclAllGather(poses)");
    hip(hipEventRecord(static_cast<hipEvent_t>(m_ev_poses), c), "hipEventRecord");
    m_poses_in_flight = true;
}

void ShardedMCL::predict(const Odometry& odom, const std::array<double, 4>& alphas)
{
    const double o[3] = {odom.rotation_1, odom.translation, odom.rotation_2};
    check(slam_mcl_predict(m_handle, o, alphas.data()), "slam_mcl_predict");
    if (m_split) gather_poses();
}

void ShardedMCL::update(const std::vector<std::tuple<double, double>>& scan, double stddev, double max_dist, const Pose& off)
{
    std::vector<double> ang(scan.size()), dist(scan.size());
    for (size_t k = 0; k < scan.size(); ++k) std::tie(ang[k], dist[k]) = scan[k];
    const double o[3] = {off.x, off.y, off.theta};
    if (m_split && !m_poses_in_flight) gather_poses();  // update without a predict before it
    check(slam_mcl_update_weights(m_handle, ang.data(), dist.data(), (int32_t)scan.size(), stddev, max_dist, o),
          "slam_mcl_update_weights");
    slam_mcl_device_state st{};
    check(slam_mcl_device_state_get(m_handle, &st), "slam_mcl_device_state_get");
    hipStream_t s = static_cast<hipStream_t>(m_stream);
    if (m_split)
    {
        // the weights (the last n doubles of the packed state) on the critical path, the poses have travelled already
        nccl(ncclAllGather(st.state_dev + 3 * m_n, m_gathered_w, (size_t)m_n, ncclDouble, m_comm, s), "ncclAllGather(weights)");
        hip(hipStreamWaitEvent(s, static_cast<hipEvent_t>(m_ev_poses), 0), "hipStreamWaitEvent");
        check(slam_mcl_shard_resample_split(m_handle, m_gathered, m_gathered_w, m_world), "slam_mcl_shard_resample_split");
        m_poses_in_flight = false;
        return;
    }
    // the ONE exchange of the step, straight out of the resident state buffer
    nccl(ncclAllGather(st.state_dev, m_gathered, 4 * (size_t)m_n, ncclDouble, m_comm, s), "ncclAllGather(state)");
    check(slam_mcl_shard_resample(m_handle, m_gathered, m_world), "slam_mcl_shard_resample");
}

void ShardedMCL::synchronize() { check(slam_mcl_synchronize(m_handle), "slam_mcl_synchronize"); }

void ShardedMCL::get_local(std::vector<double>& x, std::vector<double>& y, std::vector<double>& theta, std::vector<double>& weight)
{
    x.resize(m_n), y.resize(m_n), theta.resize(m_n), weight.resize(m_n);
    check(slam_mcl_get_poses(m_handle, x.data(), y.data(), theta.data(), 0, m_n), "slam_mcl_get_poses");
    check(slam_mcl_get_weights(m_handle, weight.data(), 0, m_n), "slam_mcl_get_weights");
}
}  // namespace slam
