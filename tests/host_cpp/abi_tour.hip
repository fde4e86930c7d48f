// abi_tour — calls every entry point of include/slam_hip.h the way a C/C++ host would, in both modes, with
// shards emulated on one device (the collectives replaced by device copies), and checks that sharded and
// unsharded runs agree.  Built with the library's HOST code under ASan + UBSan (make san): its job is to
// catch host-side memory errors and undefined behaviour in the C ABI layer; tests/test_gpu_parity.py owns
// the numerics.  Prints "abi_tour ok" and exits 0.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/slam_hip.h"

#define OK(e)                                                                                  \
    do {                                                                                       \
        int rc_ = (e);                                                                         \
        if (rc_ != SLAM_OK) { printf("line %d: rc %d: %s\n", __LINE__, rc_, slam_last_error()); exit(1); } \
    } while (0)
#define EXPECT(e, code)                                                                        \
    do {                                                                                       \
        int rc_ = (e);                                                                         \
        if (rc_ != (code)) { printf("line %d: rc %d, expected %d: %s\n", __LINE__, rc_, (code), slam_last_error()); exit(1); } \
    } while (0)
#define HIP(e)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (e);                                                                   \
        if (e_ != hipSuccess) { printf("line %d: %s\n", __LINE__, hipGetErrorString(e_)); exit(1); } \
    } while (0)
#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) { printf("line %d: check failed: %s\n", __LINE__, #c); exit(1); } \
    } while (0)

static slam_mcl_params base(int mode, int64_t n, int rows, int cols)
{
    slam_mcl_params p;
    std::memset(&p, 0, sizeof p);
    p.struct_size = (int32_t)sizeof p;
    p.mode = mode;
    p.n_particles = n;
    p.rows = rows;
    p.cols = cols;
    p.seed = 0xA81;
    return p;
}

static std::vector<uint8_t> room(int rows, int cols)
{
    std::vector<uint8_t> g((size_t)rows * cols, 255);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            if (i < 3 || j < 3 || i >= rows - 3 || j >= cols - 3 || (i > 90 && i < 110 && j > 60 && j < 200)) g[(size_t)i * cols + j] = 0;
    return g;
}

static const double ODOM[3] = {0.02, 2.5, 0.02}, ALPHAS[4] = {5e-4, 5e-4, 1e-2, 1e-2}, OFFSET[3] = {0.0, 6.0, 0.0};

static void scan(int B, std::vector<double>& ang, std::vector<double>& dist, double zmax)
{
    ang.resize(B);
    dist.resize(B);
    for (int b = 0; b < B; ++b)
    {
        ang[b] = -M_PI + 2 * M_PI * b / B;
        dist[b] = (b % 7 == 0) ? zmax : 20.0 + 3.0 * (b % 11);
    }
}

static std::vector<int32_t> indices(slam_mcl* h, int64_t n)
{
    std::vector<int32_t> v(n);
    OK(slam_mcl_get_last_indices(h, v.data()));
    return v;
}

static void localization()
{
    const int rows = 200, cols = 300, B = 45;
    const int64_t N = 6000;
    const std::vector<uint8_t> g = room(rows, cols);
    std::vector<double> ang, dist;
    scan(B, ang, dist, 120.0);
    const double mean[3] = {150.0, 60.0, 0.3}, sigma[3] = {6.0, 6.0, 0.2};
    // one handle, full update with the sort, every read-back
    slam_mcl* one = nullptr;
    slam_mcl_params p = base(SLAM_MODE_LOCALIZATION, N, rows, cols);
    OK(slam_mcl_create(&p, &one));
    EXPECT(slam_mcl_update(one, ang.data(), dist.data(), B, 5.0, 120.0, OFFSET), SLAM_ERR_STATE);  // no map yet
    OK(slam_mcl_set_map_u8(one, g.data(), rows, cols));
    OK(slam_mcl_seed_gaussian(one, mean, sigma));
    OK(slam_mcl_profile(one, 1));
    OK(slam_mcl_record_hits(one, 1));
    for (int s = 0; s < 3; ++s)
    {
        OK(slam_mcl_predict(one, ODOM, ALPHAS));
        OK(slam_mcl_update(one, ang.data(), dist.data(), B, 5.0, 120.0, OFFSET));
    }
    std::vector<double> x(N), y(N), th(N), w(N), prefix(N), hits(2 * N * B);
    std::vector<int32_t> perm(N), cells(N * B);
    double sum = 0, avg[3], ms[5];
    int64_t launches[5], stats[3];
    OK(slam_mcl_get_poses(one, x.data(), y.data(), th.data(), 0, N));
    OK(slam_mcl_get_weights(one, w.data(), 0, N));
    OK(slam_mcl_get_last_order(one, perm.data()));
    OK(slam_mcl_get_prefix(one, prefix.data(), &sum));
    OK(slam_mcl_get_hits(one, hits.data(), cells.data()));
    OK(slam_mcl_average_pose(one, avg));
    OK(slam_mcl_get_kernel_ms(one, ms, launches, 1));
    CHECK(std::is_sorted(w.begin(), w.end(), [](double a, double b) { return a > b; }) && sum > 0 && launches[1] == 3);
    EXPECT(slam_mcl_get_poses(one, x.data(), y.data(), th.data(), 1, N), SLAM_ERR_ARG);
    const double nan_odom[3] = {0.0, NAN, 0.0};
    EXPECT(slam_mcl_predict(one, nan_odom, ALPHAS), SLAM_ERR_ARG);
    EXPECT(slam_mcl_update(one, ang.data(), dist.data(), 0, 5.0, 120.0, OFFSET), SLAM_ERR_ARG);
    const double vel[2] = {1.0, 0.1}, a6[6] = {1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3};
    OK(slam_mcl_predict_velocity(one, vel, 0.1, a6));
    OK(slam_mcl_seed_uniform_free(one));
    OK(slam_mcl_update_weights(one, ang.data(), dist.data(), B, 5.0, 120.0, OFFSET));
    OK(slam_mcl_resample(one));
    OK(slam_mcl_synchronize(one));
    slam_mcl_destroy(one);
    // other kernel shapes and options on a small filter
    for (int variant = 0; variant < 4; ++variant)
    {
        slam_mcl_params q = base(SLAM_MODE_LOCALIZATION, 130, rows, cols);
        q.raycaster = variant == 0 ? SLAM_RAYCAST_WALK : variant == 1 ? SLAM_RAYCAST_FAST_BEAM_LANES : SLAM_RAYCAST_FAST_PARTICLE_LANES;
        q.prefix_order = variant == 1 ? SLAM_PREFIX_REFERENCE : SLAM_PREFIX_BLOCKED;
        q.resampler = variant == 2 ? SLAM_RESAMPLE_SYSTEMATIC : SLAM_RESAMPLE_MULTINOMIAL;
        q.weight_domain = variant == 3 ? SLAM_WEIGHT_LOG : SLAM_WEIGHT_LINEAR;
        q.noise = variant == 1 ? SLAM_NOISE_INJECTED : SLAM_NOISE_PHILOX;
        slam_mcl* h = nullptr;
        OK(slam_mcl_create(&q, &h));
        OK(slam_mcl_set_map_u8(h, g.data(), rows, cols));
        OK(slam_mcl_seed_gaussian(h, mean, sigma));
        if (variant == 1)
        {
            std::vector<double> z(3 * 130, 0.1), u(130, 0.4);
            EXPECT(slam_mcl_predict(h, ODOM, ALPHAS), SLAM_ERR_STATE);  // no normals injected yet
            OK(slam_mcl_set_noise(h, z.data(), u.data()));
        }
        OK(slam_mcl_predict(h, ODOM, ALPHAS));
        OK(slam_mcl_update(h, ang.data(), dist.data(), B, 5.0, 120.0, OFFSET));
        if (variant == 0)
        {
            OK(slam_mcl_get_stats(h, stats));
            OK(slam_mcl_collect_stats(h, 0));
            EXPECT(slam_mcl_get_stats(h, stats), SLAM_ERR_STATE);  // the counters are off
            OK(slam_mcl_collect_stats(h, 1));
            int32_t n_planes = 0;
            int64_t plane_stride = 0;
            OK(slam_mcl_get_clearance(h, nullptr, 0, &n_planes, &plane_stride));
            std::vector<uint8_t> planes((size_t)n_planes * (size_t)plane_stride);
            OK(slam_mcl_get_clearance(h, planes.data(), (int64_t)planes.size(), &n_planes, &plane_stride));
            EXPECT(slam_mcl_get_clearance(h, planes.data(), 1, &n_planes, &plane_stride), SLAM_ERR_ARG);
        }
        slam_mcl_destroy(h);
    }
    // two shards against one unsorted handle: whole and split exchange
    for (int split = 0; split < 2; ++split)
    {
        slam_mcl_params q = base(SLAM_MODE_LOCALIZATION, N, rows, cols);
        q.sort = SLAM_SORT_NONE;
        slam_mcl* ref = nullptr;
        OK(slam_mcl_create(&q, &ref));
        OK(slam_mcl_set_map_u8(ref, g.data(), rows, cols));
        OK(slam_mcl_seed_gaussian(ref, mean, sigma));
        slam_mcl* sh[2] = {nullptr, nullptr};
        const int64_t n = N / 2;
        for (int r = 0; r < 2; ++r)
        {
            slam_mcl_params s = q;
            s.n_particles = n;
            s.global_first = r * n;
            s.global_n = N;
            OK(slam_mcl_create(&s, &sh[r]));
            OK(slam_mcl_set_map_u8(sh[r], g.data(), rows, cols));
            OK(slam_mcl_seed_gaussian(sh[r], mean, sigma));
        }
        double *gathered, *gp, *gw;
        HIP(hipMalloc(&gathered, sizeof(double) * 4 * N));
        HIP(hipMalloc(&gp, sizeof(double) * 3 * N));
        HIP(hipMalloc(&gw, sizeof(double) * N));
        for (int s = 0; s < 2; ++s)
        {
            OK(slam_mcl_predict(ref, ODOM, ALPHAS));
            OK(slam_mcl_update(ref, ang.data(), dist.data(), B, 5.0, 120.0, OFFSET));
            for (int r = 0; r < 2; ++r)
            {
                OK(slam_mcl_predict(sh[r], ODOM, ALPHAS));
                OK(slam_mcl_update_weights(sh[r], ang.data(), dist.data(), B, 5.0, 120.0, OFFSET));
                slam_mcl_device_state st;
                OK(slam_mcl_device_state_get(sh[r], &st));
                CHECK(st.n_particles == n && st.global_n == N);
                HIP(hipMemcpy(gathered + 4 * n * r, st.state_dev, sizeof(double) * 4 * n, hipMemcpyDeviceToDevice));
                HIP(hipMemcpy(gp + 3 * n * r, st.state_dev, sizeof(double) * 3 * n, hipMemcpyDeviceToDevice));
                HIP(hipMemcpy(gw + n * r, st.state_dev + 3 * n, sizeof(double) * n, hipMemcpyDeviceToDevice));
            }
            for (int r = 0; r < 2; ++r)
                if (split)
                    OK(slam_mcl_shard_resample_split(sh[r], gp, gw, 2));
                else
                    OK(slam_mcl_shard_resample(sh[r], gathered, 2));
            std::vector<int32_t> all = indices(sh[0], n), second = indices(sh[1], n);
            all.insert(all.end(), second.begin(), second.end());
            CHECK(all == indices(ref, N));
        }
        EXPECT(slam_mcl_shard_resample(sh[0], nullptr, 2), SLAM_ERR_ARG);
        EXPECT(slam_mcl_resample(sh[0]), SLAM_ERR_STATE);  // a shard cannot resample alone
        HIP(hipFree(gathered));
        HIP(hipFree(gp));
        HIP(hipFree(gw));
        slam_mcl_destroy(ref);
        slam_mcl_destroy(sh[0]);
        slam_mcl_destroy(sh[1]);
    }
}

// wire: 0 = whole canvases (pack_maps / shard_clone_maps), 1 = tables + tiles (tiled handles only)
static void grid_slam(int storage, int wire)
{
    const int rows = 150, cols = 200, B = 30;
    const int64_t N = 40, n = N / 2;
    const size_t cells = (size_t)rows * cols;
    std::vector<double> ang, dist;
    scan(B, ang, dist, 80.0);
    slam_mcl_params q = base(SLAM_MODE_SLAM, N, rows, cols);
    q.sort = SLAM_SORT_NONE;
    q.map_storage = storage;
    slam_mcl* ref = nullptr;
    OK(slam_mcl_create(&q, &ref));
    slam_mcl* sh[2] = {nullptr, nullptr};
    for (int r = 0; r < 2; ++r)
    {
        slam_mcl_params s = q;
        s.n_particles = n;
        s.global_first = r * n;
        s.global_n = N;
        OK(slam_mcl_create(&s, &sh[r]));
    }
    slam_mcl_device_maps dm;
    OK(slam_mcl_device_maps_get(sh[0], &dm));
    const size_t stride = (size_t)dm.map_stride;
    const int T = SLAM_TILES_PER_CANVAS(rows, cols);
    for (int step = 0; step < 3; ++step)
    {
        OK(slam_mcl_predict(ref, ODOM, ALPHAS));
        OK(slam_mcl_update(ref, ang.data(), dist.data(), B, 5.0, 80.0, OFFSET));
        double* gathered;
        HIP(hipMalloc(&gathered, sizeof(double) * 4 * N));
        for (int r = 0; r < 2; ++r)
        {
            OK(slam_mcl_predict(sh[r], ODOM, ALPHAS));
            OK(slam_mcl_update_weights(sh[r], ang.data(), dist.data(), B, 5.0, 80.0, OFFSET));
            slam_mcl_device_state st;
            OK(slam_mcl_device_state_get(sh[r], &st));
            HIP(hipMemcpy(gathered + 4 * n * r, st.state_dev, sizeof(double) * 4 * n, hipMemcpyDeviceToDevice));
        }
        std::vector<std::vector<int32_t>> idx(2);
        for (int r = 0; r < 2; ++r)
        {
            OK(slam_mcl_shard_select(sh[r], gathered, 2));
            EXPECT(slam_mcl_update_weights(sh[r], ang.data(), dist.data(), B, 5.0, 80.0, OFFSET), SLAM_ERR_STATE);  // canvases pending
            idx[r] = indices(sh[r], n);
        }
        // what the all-to-alls deliver: for requester r the distinct wanted sources, packed by their owners
        struct Arrived { uint8_t* pool = nullptr; int32_t* tabs = nullptr; uint8_t* tiles = nullptr; int64_t n_tiles = 0; std::vector<int32_t> want; };
        Arrived got[2];
        for (int r = 0; r < 2; ++r)
        {
            std::vector<int32_t>& want = got[r].want;
            want = idx[r];
            std::sort(want.begin(), want.end());
            want.erase(std::unique(want.begin(), want.end()), want.end());
            const int64_t c = (int64_t)want.size();
            if (wire == 0) HIP(hipMalloc(&got[r].pool, stride * c));
            if (wire == 1) HIP(hipMalloc(&got[r].tabs, sizeof(int32_t) * T * c));
            std::vector<std::vector<uint8_t>> tile_parts;
            int64_t row = 0, tile_base = 0;
            for (int o = 0; o < 2; ++o)
            {
                std::vector<int32_t> local;
                for (int32_t s : want)
                    if (s / n == o) local.push_back((int32_t)(s - o * n));
                if (local.empty()) continue;
                int32_t* local_dev;
                HIP(hipMalloc(&local_dev, sizeof(int32_t) * local.size()));
                HIP(hipMemcpy(local_dev, local.data(), sizeof(int32_t) * local.size(), hipMemcpyHostToDevice));
                if (wire == 0)
                    OK(slam_mcl_pack_maps(sh[o], local_dev, (int64_t)local.size(), got[r].pool + stride * row));
                else
                {
                    int64_t nt = 0;
                    OK(slam_mcl_pack_tiles_count(sh[o], local_dev, (int64_t)local.size(), &nt));
                    uint8_t* tiles_dev;
                    HIP(hipMalloc(&tiles_dev, (size_t)std::max<int64_t>(nt, 1) * SLAM_TILE_BYTES));
                    OK(slam_mcl_pack_tiles(sh[o], local_dev, (int64_t)local.size(), got[r].tabs + T * row, tiles_dev, nt));
                    // the receiver's shift of this owner's table entries by the tiles that arrived before
                    std::vector<int32_t> tb((size_t)T * local.size());
                    HIP(hipMemcpy(tb.data(), got[r].tabs + T * row, sizeof(int32_t) * tb.size(), hipMemcpyDeviceToHost));
                    for (int32_t& e : tb)
                        if (e > 0) e += (int32_t)tile_base;
                    HIP(hipMemcpy(got[r].tabs + T * row, tb.data(), sizeof(int32_t) * tb.size(), hipMemcpyHostToDevice));
                    tile_parts.emplace_back((size_t)nt * SLAM_TILE_BYTES);
                    if (nt) HIP(hipMemcpy(tile_parts.back().data(), tiles_dev, tile_parts.back().size(), hipMemcpyDeviceToHost));
                    tile_base += nt;
                    HIP(hipFree(tiles_dev));
                }
                row += (int64_t)local.size();
                HIP(hipFree(local_dev));
            }
            if (wire == 1)
            {
                got[r].n_tiles = tile_base;
                HIP(hipMalloc(&got[r].tiles, (size_t)std::max<int64_t>(tile_base, 1) * SLAM_TILE_BYTES));
                size_t at = 0;
                for (auto& part : tile_parts)
                {
                    if (!part.empty()) HIP(hipMemcpy(got[r].tiles + at, part.data(), part.size(), hipMemcpyHostToDevice));
                    at += part.size();
                }
            }
        }
        for (int r = 0; r < 2; ++r)
        {
            std::vector<int32_t> pos(n);
            for (int64_t k = 0; k < n; ++k)
                pos[k] = (int32_t)(std::lower_bound(got[r].want.begin(), got[r].want.end(), idx[r][k]) - got[r].want.begin());
            int32_t* pos_dev;
            HIP(hipMalloc(&pos_dev, sizeof(int32_t) * n));
            HIP(hipMemcpy(pos_dev, pos.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
            if (wire == 0)
                OK(slam_mcl_shard_clone_maps(sh[r], got[r].pool, (int64_t)got[r].want.size(), pos_dev));
            else
                OK(slam_mcl_shard_import_tiles(sh[r], got[r].tabs, (int64_t)got[r].want.size(), got[r].tiles, got[r].n_tiles, pos_dev));
            HIP(hipFree(pos_dev));
            if (got[r].pool) HIP(hipFree(got[r].pool));
            if (got[r].tabs) HIP(hipFree(got[r].tabs));
            if (got[r].tiles) HIP(hipFree(got[r].tiles));
        }
        HIP(hipFree(gathered));
        // same selection and canvases as the unsharded filter
        std::vector<int32_t> all = idx[0];
        all.insert(all.end(), idx[1].begin(), idx[1].end());
        CHECK(all == indices(ref, N));
        std::vector<uint8_t> a(cells), b(cells);
        for (int64_t k = 0; k < N; k += 7)
        {
            OK(slam_mcl_get_map_u8(ref, k, a.data()));
            OK(slam_mcl_get_map_u8(sh[k / n], k % n, b.data()));
            CHECK(a == b);
        }
    }
    // canvases set from the host, and the error paths of the index arrays
    std::vector<uint8_t> custom(2 * cells, 200);
    OK(slam_mcl_set_maps_u8(ref, custom.data(), 3, 2));
    std::vector<uint8_t> back(cells);
    OK(slam_mcl_get_map_u8(ref, 4, back.data()));
    CHECK(back[17] == 200);
    EXPECT(slam_mcl_set_maps_u8(ref, custom.data(), N - 1, 2), SLAM_ERR_ARG);
    int32_t bad_host[2] = {0, (int32_t)n + 5}, *bad_dev;
    uint8_t* out_dev;
    HIP(hipMalloc(&bad_dev, sizeof bad_host));
    HIP(hipMalloc(&out_dev, stride * 2));
    HIP(hipMemcpy(bad_dev, bad_host, sizeof bad_host, hipMemcpyHostToDevice));
    EXPECT(slam_mcl_pack_maps(sh[0], bad_dev, 2, out_dev), SLAM_ERR_ARG);  // skipped and reported, not a fault
    EXPECT(slam_mcl_shard_clone_maps(sh[0], out_dev, 2, bad_dev), SLAM_ERR_STATE);  // no select pending
    if (storage == SLAM_MAPS_TILED)
    {
        int64_t ts[3];
        OK(slam_mcl_get_tile_stats(ref, ts));
        CHECK(ts[0] > 0 && ts[1] > 0);
    }
    else
    {
        int64_t ts[3];
        EXPECT(slam_mcl_get_tile_stats(ref, ts), SLAM_ERR_STATE);
        int64_t nt;
        EXPECT(slam_mcl_pack_tiles_count(sh[0], bad_dev, 1, &nt), SLAM_ERR_STATE);
    }
    HIP(hipFree(bad_dev));
    HIP(hipFree(out_dev));
    slam_mcl_destroy(ref);
    slam_mcl_destroy(sh[0]);
    slam_mcl_destroy(sh[1]);
}

static void one_off_raycasts()
{
    const int rows = 200, cols = 300, R = 36;
    const std::vector<uint8_t> g = room(rows, cols);
    std::vector<uint8_t> g01(g.size());
    for (size_t k = 0; k < g.size(); ++k) g01[k] = g[k] ? 1 : 0;
    const double poses[6] = {150.0, 60.0, 0.3, 40.0, 150.0, -1.0};
    std::vector<double> hits(2 * 2 * R);
    OK(slam_fake_lidar_scan(0, g01.data(), rows, cols, poses, 2, 0.0, 2 * M_PI, 200.0, R, hits.data()));
    std::vector<double> h2(4);
    OK(slam_raycast_u8(0, g.data(), rows, cols, poses, 2, 200.0 * 200.0, 0.5, h2.data()));
    OK(slam_raycast_batch(0, 2, g01.data(), rows, cols, poses, 2, 200.0 * 200.0, 0.5, h2.data()));
    EXPECT(slam_raycast_batch(0, 7, g01.data(), rows, cols, poses, 2, 200.0 * 200.0, 0.5, h2.data()), SLAM_ERR_ARG);
}

int main()
{
    CHECK(slam_version() >= 1);
    slam_mcl* h = nullptr;
    slam_mcl_params bad = base(SLAM_MODE_SLAM, 0, 10, 10);
    EXPECT(slam_mcl_create(&bad, &h), SLAM_ERR_ARG);
    EXPECT(slam_mcl_create(nullptr, &h), SLAM_ERR_ARG);
    slam_mcl_destroy(nullptr);
    localization();
    grid_slam(SLAM_MAPS_DENSE, 0);
    grid_slam(SLAM_MAPS_TILED, 0);
    grid_slam(SLAM_MAPS_TILED, 1);
    // a tile pool that runs dry is reported
    slam_mcl_params p = base(SLAM_MODE_SLAM, 16, 300, 300);
    p.map_storage = SLAM_MAPS_TILED;
    p.tile_pool_tiles = 20;
    OK(slam_mcl_create(&p, &h));
    std::vector<double> ang, dist;
    scan(40, ang, dist, 200.0);
    EXPECT(slam_mcl_update(h, ang.data(), dist.data(), 40, 5.0, 200.0, OFFSET), SLAM_ERR_NOMEM);
    slam_mcl_destroy(h);
    one_off_raycasts();
    printf("abi_tour ok\n");
    return 0;
}
