// shard_rccl — the sharded localization step driven from C++ with RCCL called directly
// (slam::ShardedMCL).  One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK from the environment; the
// RCCL id travels through the file named by SLAM_NCCL_ID_FILE (rank 0 writes it).  On the one-GPU test
// box it runs as world 1 (the collective still goes through RCCL); a node with 8 GPUs runs
//   for r in 0..7: RANK=r LOCAL_RANK=r WORLD_SIZE=8 SLAM_NCCL_ID_FILE=/tmp/id ./shard_rccl map.raw rows cols n_local steps [split]
// Prints this rank's particles as hex words ("P x y theta w"), which tests/test_host_cpp.py compares
// with one unsharded handle.
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
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
    std::vector<unsigned char> map((size_t)rows * cols);
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
        std::vector<double> x, y, th, w;
        f.get_local(x, y, th, w);
        for (long k = 0; k < n_local; ++k)
        {
            unsigned long long b[4];
            std::memcpy(&b[0], &x[k], 8), std::memcpy(&b[1], &y[k], 8), std::memcpy(&b[2], &th[k], 8), std::memcpy(&b[3], &w[k], 8);
            std::printf("P %016llx %016llx %016llx %016llx\n", b[0], b[1], b[2], b[3]);
        }
        std::printf("RANK %d of %d: %d steps of %ld particles x 90 beams, %.3f ms per step (%s exchange)\n", rank, world, steps, n_local,
                    ms / steps, split ? "split" : "single");
    }
    ncclCommDestroy(comm);
    return 0;
}
