// CPU study (not product): how often does a wavefront of k_weights_L<0> (64 consecutive beams of one particle in lock step)
// run the re-synchronisation block of ray_fast_L0's jump loop (a binade edge or |coordinate| < 4 in the way)?
// Build: g++ -std=gnu++17 -O2 -ffp-contract=off -march=x86-64-v3 -shared -fPIC -o tools/bin/libslowpath.so tools/slowpath_sim.cpp
#include <cstdint>
#include <vector>
static thread_local uint64_t g_mask_lo, g_mask_hi;
static thread_local int g_events;
#define SLAM_RAY_TRACE_RESYNC(r) do { unsigned k_ = (r); if (k_ < 64) g_mask_lo |= 1ull << k_; else g_mask_hi |= 1ull << (k_ < 127 ? k_ - 64 : 63); ++g_events; } while (0)
#include "../slam_amd/csrc/clearance.h"
#include "../slam_amd/csrc/ray.h"
extern "C" __attribute__((visibility("default")))
void sp_pass(const uint8_t* planes, int rows, int cols, long n, const double* px, const double* py, const double* pth, int B,
             const double* angles, const double* z, double zmax, double stddev, double sd, double stheta, double srot,
             int* reads_out, double* resync_iters_per_group /* n x groups */, int* resync_events /* n x B */)
{
    const double zmax2 = zmax * zmax;
    int k_safe = (int)__builtin_floor(zmax) - 1;
    if (k_safe < 0) k_safe = 0;
    const size_t stride = slam_cone_plane_stride(rows, cols);
    const int G = (B + 63) / 64;
    for (long p = 0; p < n; ++p)
    {
        double s0, c0;
        dm_sincos(pth[p] + stheta, &s0, &c0);
        const double sx = px[p] + c0 * sd, sy = py[p] + s0 * sd;
        for (int g = 0; g < G; ++g)
        {
            uint64_t lo = 0, hi = 0;
            for (int b = g * 64; b < B && b < g * 64 + 64; ++b)
            {
                double s, c;
                dm_sincos(pth[p] + angles[b] + srot, &s, &c);
                slam_ray_result r;
                unsigned reads = 0;
                g_mask_lo = g_mask_hi = 0; g_events = 0;
                ray_fast_L(planes, (unsigned)stride, cols + 4, rows, cols, sx, sy, c, s, zmax2, k_safe, &r, &reads, slam_sample_limit(z[b], stddev));
                reads_out[p * B + b] = (int)reads;
                resync_events[p * B + b] = g_events;
                lo |= g_mask_lo; hi |= g_mask_hi;
            }
            resync_iters_per_group[p * G + g] = __builtin_popcountll(lo) + __builtin_popcountll(hi);
        }
    }
}
