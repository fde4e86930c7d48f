// slam_shard_rccl.cpp — slam::ShardedMCL: the sharded filter step in C++ with RCCL called directly
// (see slam/sharded_mcl.h).  Plain host C++: HIP runtime API for the streams and the exchange buffers,
// RCCL for the collectives, everything else through the C ABI of libslam_hip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
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
hipStream_t S(void* p) { return static_cast<hipStream_t>(p); }
}  // namespace

namespace slam
{
ShardedMCL::ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const unsigned char* map,
                       int device, uint64_t seed, bool split_exchange)
    : m_comm(comm), m_rank(rank), m_world(world), m_device(device), m_n(n_local), m_split(split_exchange)
{
    if (!map) throw std::invalid_argument("ShardedMCL: localization mode needs a map (SLAM mode: the SlamMode constructor)");
    try
    {
        init(rows, cols, map, seed, nullptr);
    }
    catch (...)
    {
        release();  // the destructor does not run for a constructor that throws
        throw;
    }
}

ShardedMCL::ShardedMCL(ncclComm_t comm, int rank, int world, int64_t n_local, int rows, int cols, const SlamMode& slam, int device,
                       uint64_t seed)
    : m_comm(comm), m_rank(rank), m_world(world), m_device(device), m_n(n_local), m_split(false), m_slam(true)
{
    try
    {
        init(rows, cols, nullptr, seed, &slam);
    }
    catch (...)
    {
        release();
        throw;
    }
}

void ShardedMCL::init(int rows, int cols, const unsigned char* map, uint64_t seed, const SlamMode* slam)
{
    if (m_world < 1 || m_rank < 0 || m_rank >= m_world || m_n < 1) throw std::invalid_argument("ShardedMCL: bad rank / world / n_local");
    m_rows = rows;
    m_cols = cols;
    m_T = SLAM_TILES_PER_CANVAS(rows, cols);
    hip(hipSetDevice(m_device), "hipSetDevice");
    hipStream_t s = nullptr, c = nullptr;
    hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    m_stream = s;
    slam_mcl_params p{};
    p.struct_size = sizeof(p);
    p.mode = slam ? SLAM_MODE_SLAM : SLAM_MODE_LOCALIZATION;
    p.n_particles = m_n;
    p.rows = rows;
    p.cols = cols;
    p.device = m_device;
    p.noise = SLAM_NOISE_PHILOX;
    p.seed = seed;
    p.prefix_order = SLAM_PREFIX_BLOCKED;  // the canonical order: independent of the number of ranks
    p.sort = SLAM_SORT_NONE;               // no caller needs a global sort (apps/grid_slam.cpp:112-119): best() / average_pose()
    p.global_first = (int64_t)m_rank * m_n;
    p.global_n = (int64_t)m_world * m_n;
    p.stream = m_stream;
    p.async_launch = 1;  // the step is enqueued; synchronize() / get_local() wait
    if (slam)
    {
        p.map_storage = SLAM_MAPS_TILED;  // a shard holds far more canvases than would fit dense (125 000 maze-sized ones in configs[3])
        p.tile_pool_tiles = slam->tile_pool_tiles;
    }
    check(slam_mcl_create(&p, &m_handle), "slam_mcl_create");
    if (!slam) check(slam_mcl_set_map_u8(m_handle, map, rows, cols), "slam_mcl_set_map_u8");
    const size_t n = (size_t)m_n, w = (size_t)m_world;
    if (m_split)
    {
        hip(hipStreamCreateWithFlags(&c, hipStreamNonBlocking), "hipStreamCreate");
        m_comm_stream = c;
        hipEvent_t a = nullptr, b = nullptr;
        hip(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
        m_ev_predict = a;
        hip(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
        m_ev_poses = b;
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered), sizeof(double) * 3 * n * w), "hipMalloc");
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered_w), sizeof(double) * n * w), "hipMalloc");
    }
    else
        hip(hipMalloc(reinterpret_cast<void**>(&m_gathered), sizeof(double) * 4 * n * w), "hipMalloc");
    // the small collectives: shard summaries (8 doubles per rank), request / tile counts (world int64 per rank)
    hip(hipMalloc(reinterpret_cast<void**>(&m_small), sizeof(double) * (w * w + w + SLAM_SHARD_SUMMARY_DOUBLES * (w + 1) + 8)), "hipMalloc");
}

void ShardedMCL::release() noexcept
{
    (void)hipSetDevice(m_device);
    if (m_stream) (void)hipStreamSynchronize(S(m_stream));
    if (m_comm_stream) (void)hipStreamSynchronize(S(m_comm_stream));
    if (m_handle) slam_mcl_destroy(m_handle);
    m_handle = nullptr;
    if (m_gathered) (void)hipFree(m_gathered);
    if (m_gathered_w) (void)hipFree(m_gathered_w);
    if (m_small) (void)hipFree(m_small);
    m_gathered = m_gathered_w = m_small = nullptr;
    for (int k = 0; k < N_SCRATCH; ++k)
    {
        if (m_scr[k]) (void)hipFree(m_scr[k]);
        m_scr[k] = nullptr;
        m_scr_bytes[k] = 0;
    }
    if (m_ev_predict) (void)hipEventDestroy(static_cast<hipEvent_t>(m_ev_predict));
    if (m_ev_poses) (void)hipEventDestroy(static_cast<hipEvent_t>(m_ev_poses));
    if (m_comm_stream) (void)hipStreamDestroy(S(m_comm_stream));
    if (m_stream) (void)hipStreamDestroy(S(m_stream));
    m_ev_predict = m_ev_poses = m_comm_stream = m_stream = nullptr;
}

ShardedMCL::~ShardedMCL() { release(); }

// grow-only device buffer `which`; the old contents are NOT kept
void* ShardedMCL::scratch(int which, size_t bytes)
{
    if (bytes > m_scr_bytes[which])
    {
        if (m_scr[which])
        {
            hip(hipStreamSynchronize(S(m_stream)), "hipStreamSynchronize");  // queued work may still read the old buffer
            hip(hipFree(m_scr[which]), "hipFree");
            m_scr[which] = nullptr;
            m_scr_bytes[which] = 0;
        }
        const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
        hip(hipMalloc(&m_scr[which], want), "hipMalloc");
        m_scr_bytes[which] = want;
    }
    return m_scr[which];
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
    hipStream_t s = S(m_stream), c = S(m_comm_stream);
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
    hipStream_t s = S(m_stream);
    if (m_split)
    {
        // the weights (the last n doubles of the packed state) on the critical path, the poses have travelled already
        nccl(ncclAllGather(st.state_dev + 3 * m_n, m_gathered_w, (size_t)m_n, ncclDouble, m_comm, s), "ncclAllGather(weights)");
        hip(hipStreamWaitEvent(s, static_cast<hipEvent_t>(m_ev_poses), 0), "hipStreamWaitEvent");
        check(slam_mcl_shard_resample_split(m_handle, m_gathered, m_gathered_w, m_world), "slam_mcl_shard_resample_split");
        m_poses_in_flight = false;
        return;
    }
    // the ONE exchange of the step (localization), straight out of the resident state buffer
    nccl(ncclAllGather(st.state_dev, m_gathered, 4 * (size_t)m_n, ncclDouble, m_comm, s), "ncclAllGather(state)");
    if (!m_slam)
    {
        check(slam_mcl_shard_resample(m_handle, m_gathered, m_world), "slam_mcl_shard_resample");
        return;
    }
    // SLAM mode: poses and weights now, the canvases of the selected particles from their owners
    check(slam_mcl_shard_select(m_handle, m_gathered, m_world), "slam_mcl_shard_select");
    exchange_canvases();
}

// slam/mcl.cpp:147-155,205-227 across ranks: slot k of this rank needs the canvas of global particle idx[k], which lives
// with rank idx[k] / n_local.  Every distinct (source, destination rank) pair travels once, as the source's table and
// the distinct 4 KiB tiles the requested tables name (copy-on-write sharing between them survives the trip).
void ShardedMCL::exchange_canvases()
{
    hipStream_t s = S(m_stream);
    const int W = m_world;
    const int64_t n = m_n;
    const size_t T = (size_t)m_T;
    slam_mcl_device_maps dm{};
    check(slam_mcl_device_maps_get(m_handle, &dm), "slam_mcl_device_maps_get");
    // 1. the global sources of my slots; the distinct ones, sorted, are grouped by owner
    std::vector<int32_t> idx((size_t)n);
    hip(hipMemcpyAsync(idx.data(), dm.idx_dev, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(idx)");
    hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    std::vector<int32_t> want(idx);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    std::vector<int64_t> n_want((size_t)W, 0), off_want((size_t)W + 1, 0);
    std::vector<int32_t> local_ids(want.size()), pos((size_t)n);
    for (size_t j = 0; j < want.size(); ++j)
    {
        const int64_t owner = want[j] / n;
        if (want[j] < 0 || owner >= W) throw std::runtime_error("ShardedMCL: selected source out of range");
        n_want[(size_t)owner]++;
        local_ids[j] = (int32_t)(want[j] - owner * n);
    }
    for (int o = 0; o < W; ++o) off_want[(size_t)o + 1] = off_want[(size_t)o] + n_want[(size_t)o];
    for (int64_t k = 0; k < n; ++k) pos[(size_t)k] = (int32_t)(std::lower_bound(want.begin(), want.end(), idx[(size_t)k]) - want.begin());
    // 2. every owner learns how many of its canvases each rank wants: all-gather of one row of counts per rank
    int64_t* cnt_send = reinterpret_cast<int64_t*>(m_small);
    int64_t* cnt_all = cnt_send + W;
    std::vector<int64_t> M((size_t)W * W);
    auto counts = [&](const std::vector<int64_t>& row, std::vector<int64_t>& col, const char* what) {
        hip(hipMemcpyAsync(cnt_send, row.data(), sizeof(int64_t) * (size_t)W, hipMemcpyHostToDevice, s), what);
        nccl(ncclAllGather(cnt_send, cnt_all, (size_t)W, ncclInt64, m_comm, s), what);
        hip(hipMemcpyAsync(M.data(), cnt_all, sizeof(int64_t) * (size_t)W * W, hipMemcpyDeviceToHost, s), what);
        hip(hipStreamSynchronize(s), what);
        for (int r = 0; r < W; ++r) col[(size_t)r] = M[(size_t)r * W + m_rank];  // what rank r addressed to me
    };
    std::vector<int64_t> n_req((size_t)W), off_req((size_t)W + 1, 0);
    counts(n_want, n_req, "all-gather(request counts)");
    for (int r = 0; r < W; ++r) off_req[(size_t)r + 1] = off_req[(size_t)r] + n_req[(size_t)r];
    const size_t tot_req = (size_t)off_req[(size_t)W], tot_want = want.size();
    // 3. ... and which: the local indices travel to their owners
    int32_t* ids_dev = static_cast<int32_t*>(scratch(0, sizeof(int32_t) * std::max<size_t>(tot_want, 1)));
    int32_t* req_dev = static_cast<int32_t*>(scratch(1, sizeof(int32_t) * std::max<size_t>(tot_req, 1)));
    hip(hipMemcpyAsync(ids_dev, local_ids.data(), sizeof(int32_t) * tot_want, hipMemcpyHostToDevice, s), "hipMemcpyAsync(ids)");
    nccl(ncclGroupStart(), "ncclGroupStart");
    for (int peer = 0; peer < W; ++peer)
    {
        if (n_want[(size_t)peer])
            nccl(ncclSend(ids_dev + off_want[(size_t)peer], (size_t)n_want[(size_t)peer], ncclInt32, peer, m_comm, s), "ncclSend(ids)");
        if (n_req[(size_t)peer])
            nccl(ncclRecv(req_dev + off_req[(size_t)peer], (size_t)n_req[(size_t)peer], ncclInt32, peer, m_comm, s), "ncclRecv(ids)");
    }
    nccl(ncclGroupEnd(), "ncclGroupEnd");
    // 4. per requester: the tables of its canvases and every distinct tile they name, once
    int32_t* tabs_send = static_cast<int32_t*>(scratch(2, sizeof(int32_t) * std::max<size_t>(tot_req * T, 1)));
    std::vector<int64_t> n_send((size_t)W, 0), off_send((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r)
    {
        if (n_req[(size_t)r])
            check(slam_mcl_pack_tiles_count(m_handle, req_dev + off_req[(size_t)r], n_req[(size_t)r], &n_send[(size_t)r]), "slam_mcl_pack_tiles_count");
        off_send[(size_t)r + 1] = off_send[(size_t)r] + n_send[(size_t)r];
        if (!n_req[(size_t)r]) continue;
        // (the payload buffer grows while the requesters are served: what is already packed moves along)
        const size_t need = (size_t)off_send[(size_t)r + 1] * SLAM_TILE_BYTES;
        if (need > m_scr_bytes[3])
        {
            void* bigger = nullptr;
            const size_t cap = need + need / 2 + SLAM_TILE_BYTES;
            hip(hipMalloc(&bigger, cap), "hipMalloc(tiles)");
            if (off_send[(size_t)r])
                hip(hipMemcpyAsync(bigger, m_scr[3], (size_t)off_send[(size_t)r] * SLAM_TILE_BYTES, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
            hip(hipStreamSynchronize(s), "hipStreamSynchronize");
            if (m_scr[3]) hip(hipFree(m_scr[3]), "hipFree");
            m_scr[3] = bigger;
            m_scr_bytes[3] = cap;
        }
        check(slam_mcl_pack_tiles(m_handle, req_dev + off_req[(size_t)r], n_req[(size_t)r], tabs_send + (size_t)off_req[(size_t)r] * T,
                                  static_cast<uint8_t*>(m_scr[3]) + (size_t)off_send[(size_t)r] * SLAM_TILE_BYTES, n_send[(size_t)r]),
              "slam_mcl_pack_tiles");
    }
    uint8_t* tiles_send = static_cast<uint8_t*>(m_scr[3]);
    // 5. how many tiles arrive from each owner
    std::vector<int64_t> n_recv((size_t)W), off_recv((size_t)W + 1, 0);
    counts(n_send, n_recv, "all-gather(tile counts)");
    for (int o = 0; o < W; ++o) off_recv[(size_t)o + 1] = off_recv[(size_t)o] + n_recv[(size_t)o];
    const size_t tot_recv = (size_t)off_recv[(size_t)W];
    if (tot_recv > 2147483647u) throw std::runtime_error("ShardedMCL: more than 2^31 tiles arrive");
    // 6. tables and tiles travel: one send and one receive per peer and kind, all links at once
    int32_t* tabs_recv = static_cast<int32_t*>(scratch(4, sizeof(int32_t) * std::max<size_t>(tot_want * T, 1)));
    uint8_t* tiles_recv = static_cast<uint8_t*>(scratch(5, std::max<size_t>(tot_recv * SLAM_TILE_BYTES, 1)));
    nccl(ncclGroupStart(), "ncclGroupStart");
    for (int peer = 0; peer < W; ++peer)
    {
        const size_t p = (size_t)peer;
        if (n_req[p]) nccl(ncclSend(tabs_send + (size_t)off_req[p] * T, (size_t)n_req[p] * T, ncclInt32, peer, m_comm, s), "ncclSend(tables)");
        if (n_want[p]) nccl(ncclRecv(tabs_recv + (size_t)off_want[p] * T, (size_t)n_want[p] * T, ncclInt32, peer, m_comm, s), "ncclRecv(tables)");
        if (n_send[p])
            nccl(ncclSend(tiles_send + (size_t)off_send[p] * SLAM_TILE_BYTES, (size_t)n_send[p] * SLAM_TILE_BYTES, ncclUint8, peer, m_comm, s), "ncclSend(tiles)");
        if (n_recv[p])
            nccl(ncclRecv(tiles_recv + (size_t)off_recv[p] * SLAM_TILE_BYTES, (size_t)n_recv[p] * SLAM_TILE_BYTES, ncclUint8, peer, m_comm, s), "ncclRecv(tiles)");
    }
    nccl(ncclGroupEnd(), "ncclGroupEnd");
    // 7. table entries count from 1 within their owner's payload: shift by the owner's first tile; then every slot takes
    //    the table of its source and the census settles who shares what
    for (int o = 0; o < W; ++o)
        if (n_want[(size_t)o] && off_recv[(size_t)o])
            check(slam_mcl_shift_tables(m_handle, tabs_recv + (size_t)off_want[(size_t)o] * T, n_want[(size_t)o], (int32_t)off_recv[(size_t)o]), "slam_mcl_shift_tables");
    int32_t* pos_dev = static_cast<int32_t*>(scratch(6, sizeof(int32_t) * (size_t)n));
    hip(hipMemcpyAsync(pos_dev, pos.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s), "hipMemcpyAsync(pos)");
    check(slam_mcl_shard_import_tiles(m_handle, tabs_recv, (int64_t)tot_want, tiles_recv, (int64_t)tot_recv, pos_dev), "slam_mcl_shard_import_tiles");
    hip(hipStreamSynchronize(s), "hipStreamSynchronize");  // the host vectors above are sources of queued copies
}

void ShardedMCL::synchronize() { check(slam_mcl_synchronize(m_handle), "slam_mcl_synchronize"); }

void ShardedMCL::get_local(std::vector<double>& x, std::vector<double>& y, std::vector<double>& theta, std::vector<double>& weight)
{
    x.resize(m_n), y.resize(m_n), theta.resize(m_n), weight.resize(m_n);
    check(slam_mcl_get_poses(m_handle, x.data(), y.data(), theta.data(), 0, m_n), "slam_mcl_get_poses");
    check(slam_mcl_get_weights(m_handle, weight.data(), 0, m_n), "slam_mcl_get_weights");
}

void ShardedMCL::get_local_map(int64_t local_slot, std::vector<unsigned char>& canvas)
{
    canvas.resize((size_t)m_rows * m_cols);
    check(slam_mcl_get_map_u8(m_handle, local_slot, canvas.data()), "slam_mcl_get_map_u8");
}

// one all-gather of 64 B per rank, then the library's host arithmetic in rank order: the same bits on every rank
void ShardedMCL::summarize(double avg[3], Best& b)
{
    hipStream_t s = S(m_stream);
    double mine[SLAM_SHARD_SUMMARY_DOUBLES];
    check(slam_mcl_shard_summary(m_handle, mine), "slam_mcl_shard_summary");
    std::vector<double> all((size_t)SLAM_SHARD_SUMMARY_DOUBLES * m_world);
    double* send = m_small + (size_t)m_world * m_world + m_world;  // behind the count matrices
    double* recv = send + SLAM_SHARD_SUMMARY_DOUBLES;
    hip(hipMemcpyAsync(send, mine, sizeof(mine), hipMemcpyHostToDevice, s), "hipMemcpyAsync(summary)");
    nccl(ncclAllGather(send, recv, SLAM_SHARD_SUMMARY_DOUBLES, ncclDouble, m_comm, s), "ncclAllGather(summary)");
    hip(hipMemcpyAsync(all.data(), recv, sizeof(double) * all.size(), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(summary)");
    hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    int32_t shard = 0;
    check(slam_shard_summary_combine(all.data(), m_world, avg, &b.weight, &b.global_slot, &shard, &b.local_slot), "slam_shard_summary_combine");
    b.rank = shard;
}

Pose ShardedMCL::average_pose()
{
    double avg[3];
    Best b{};
    summarize(avg, b);
    return {avg[0], avg[1], avg[2]};
}

ShardedMCL::Best ShardedMCL::best(std::vector<unsigned char>* canvas)
{
    hipStream_t s = S(m_stream);
    double avg[3];
    Best b{};
    summarize(avg, b);
    double pose[3] = {0, 0, 0};
    if (b.rank == m_rank) check(slam_mcl_get_poses(m_handle, &pose[0], &pose[1], &pose[2], b.local_slot, 1), "slam_mcl_get_poses");
    double* buf = m_small + (size_t)m_world * m_world + m_world;
    hip(hipMemcpyAsync(buf, pose, sizeof(pose), hipMemcpyHostToDevice, s), "hipMemcpyAsync(best pose)");
    nccl(ncclBroadcast(buf, buf, 3, ncclDouble, b.rank, m_comm, s), "ncclBroadcast(best pose)");
    hip(hipMemcpyAsync(pose, buf, sizeof(pose), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(best pose)");
    hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    b.pose = {pose[0], pose[1], pose[2]};
    if (canvas)
    {
        if (!m_slam) throw std::logic_error("ShardedMCL::best: canvases exist in SLAM mode only");
        const size_t bytes = (size_t)m_rows * m_cols;
        canvas->resize(bytes);
        if (b.rank == m_rank) check(slam_mcl_get_map_u8(m_handle, b.local_slot, canvas->data()), "slam_mcl_get_map_u8");
        unsigned char* dev = static_cast<unsigned char*>(scratch(7, bytes));
        hip(hipMemcpyAsync(dev, canvas->data(), bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync(best canvas)");
        nccl(ncclBroadcast(dev, dev, bytes, ncclUint8, b.rank, m_comm, s), "ncclBroadcast(best canvas)");
        hip(hipMemcpyAsync(canvas->data(), dev, bytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(best canvas)");
        hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    }
    return b;
}
}  // namespace slam
