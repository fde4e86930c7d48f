// shard_rccl — the sharded localization step driven from C++ with RCCL called directly
// (slam::ShardedMCL).  One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK from the environment; the
// RCCL id travels through the file named by SLAM_NCCL_ID_FILE (rank 0 writes it).  On the one-GPU test
// box it runs as world 1 (the collective still goes through RCCL); a node with 8 GPUs runs
//   for r in 0..7: RANK=r LOCAL_RANK=r WORLD_SIZE=8 SLAM_NCCL_ID_FILE=/tmp/id ./shard_rccl map.raw rows cols n_local steps [split]
// Prints this rank's particles as hex words ("P x y theta w"), which tests/test_host_cpp.py compares
// with one unsharded handle, then the collective read-out ("AVG x y theta", "BEST weight slot rank x y theta").
// SLAM mode (per-particle canvases, the grid_slam loop of BASELINE configs[3]):
//   ./shard_rccl slam rows cols n_local steps [pool_tiles]
// additionally prints "M <slot> <hash>" per local canvas and "BESTMAP <hash>".
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "slam/sharded_mcl.h"

static int env_int(const char* k, int d) { return std::getenv(k) ? std::atoi(std::getenv(k)) : d; }
static unsigned long long hash_bytes(const std::vector<unsigned char>& v)
{
    unsigned long long h = 0;  // position-weighted sum mod 2^64 (one numpy expression on the test's side)
    for (size_t i = 0; i < v.size(); ++i) h += (unsigned long long)(v[i] + 1u) * ((unsigned long long)i * 2654435761ull + 12345ull);
    return h;
}
static void print_particles(slam::ShardedMCL& f, long n_local)
{
    std::vector<double> x, y, th, w;
    f.get_local(x, y, th, w);
    for (long k = 0; k < n_local; ++k)
    {
        unsigned long long b[4];
        std::memcpy(&b[0], &x[k], 8), std::memcpy(&b[1], &y[k], 8), std::memcpy(&b[2], &th[k], 8), std::memcpy(&b[3], &w[k], 8);
        std::printf("P %016llx %016llx %016llx %016llx\n", b[0], b[1], b[2], b[3]);
    }
}
static void print_read_out(slam::ShardedMCL& f, bool canvas)
{
    const slam::Pose a = f.average_pose();
    std::printf("AVG %.17g %.17g %.17g\n", a.x, a.y, a.theta);
    std::vector<unsigned char> m;
    const slam::ShardedMCL::Best b = f.best(canvas ? &m : nullptr);
    std::printf("BEST %.17g %lld %d %.17g %.17g %.17g\n", b.weight, (long long)b.global_slot, b.rank, b.pose.x, b.pose.y, b.pose.theta);
    if (canvas) std::printf("BESTMAP %016llx\n", hash_bytes(m));
}

int main(int argc, char** argv)
{
    if (argc < 6)
    {
        std::fprintf(stderr, "usage: %s map.raw(uint8) rows cols n_local steps [split]\n", argv[0]);
        return 2;
    }
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
    const long n_local = std::atol(argv[4]);
    const int steps = std::atoi(argv[5]);
    const bool split = argc > 6 && std::strcmp(argv[6], "split") == 0;
    const bool slam_mode = std::strcmp(argv[1], "slam") == 0;
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
    std::vector<unsigned char> map((size_t)rows * cols);
    if (!slam_mode)
    {
        std::ifstream in(argv[1], std::ios::binary);
        in.read(reinterpret_cast<char*>(map.data()), (std::streamsize)map.size());
        if (!in) return 3;
    }
    if (hipSetDevice(device) != hipSuccess) return 4;
    ncclUniqueId id;
    const char* idfile = std::getenv("SLAM_NCCL_ID_FILE");
    if (rank == 0)
    {
        if (ncclGetUniqueId(&id) != ncclSuccess) return 5;
        if (world > 1)
        {
            if (!idfile) return 6;
            std::ofstream out(std::string(idfile) + ".tmp", std::ios::binary);
            out.write(reinterpret_cast<const char*>(&id), sizeof(id));
            out.close();
            std::rename((std::string(idfile) + ".tmp").c_str(), idfile);
        }
    }
    else
    {
        if (!idfile) return 6;
        for (int tries = 0;; ++tries)
        {
            std::ifstream in(idfile, std::ios::binary);
            if (in.read(reinterpret_cast<char*>(&id), sizeof(id))) break;
            if (tries > 600) return 7;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) return 8;
    if (slam_mode)
    {
        // the app's loop (apps/grid_slam.cpp:100-168): every particle starts at the canvas centre; forward odometry,
        // a fixed fan of 24 beams whose ranges change from step to step, offset {0, 6, 0}
        slam::ShardedMCL::SlamMode sm;
        sm.tile_pool_tiles = argc > 6 ? std::atol(argv[6]) : 0;
        slam::ShardedMCL f(comm, rank, world, n_local, rows, cols, sm, device, 0x5EED0004ull);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; ++k)
        {
            std::vector<std::tuple<double, double>> scan;
            for (int b = 0; b < 24; ++b) scan.emplace_back(b * (2 * M_PI / 24) - M_PI, 12.0 + 2.5 * ((b + 3 * k) % 11));
            f.predict({0.0, 1.5, 0.01}, {0.001, 0.001, 0.001, 0.001});
            f.update(scan, 2.0, 40.0, {0, 6, 0});
        }
        f.synchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        print_particles(f, n_local);
        std::vector<unsigned char> m;
        for (long k = 0; k < n_local; ++k)
        {
            f.get_local_map(k, m);
            std::printf("M %ld %016llx\n", k, hash_bytes(m));
        }
        print_read_out(f, true);
        std::printf("RANK %d of %d: %d grid-SLAM steps of %ld particles x 24 beams on %dx%d canvases, %.3f ms per step\n", rank, world,
                    steps, n_local, rows, cols, ms / steps);
    }
    else
    {
        slam::ShardedMCL f(comm, rank, world, n_local, rows, cols, map.data(), device, 0x5EED0001ull, split);
        f.seed_gaussian({520.0, 280.0, 0.4}, {25.0, 25.0, 0.2});
        // a fixed fan of beams (angle, range): the same scan on every rank
        std::vector<std::tuple<double, double>> scan;
        for (int b = 0; b < 90; ++b) scan.emplace_back(b * (2 * M_PI / 90) - M_PI, 40.0 + 3.0 * (b % 17));
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; ++k)
        {
            f.predict({2.5, 0.02, 0.02}, {0.0005, 0.0005, 0.01, 0.01});
            f.update(scan, 5.0, 500.0, {0, 30, 0});
        }
        f.synchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        print_particles(f, n_local);
        print_read_out(f, false);
        std::printf("RANK %d of %d: %d steps of %ld particles x 90 beams, %.3f ms per step (%s exchange)\n", rank, world, steps, n_local,
                    ms / steps, split ? "split" : "single");
    }
    ncclCommDestroy(comm);
    return 0;
}
