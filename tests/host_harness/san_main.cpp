// TEST ONLY: the host build of slam_amd/csrc/ray.h under AddressSanitizer + UBSan.  The padded
// clearance grid is a std::vector of exactly (rows+4) x (cols+4) bytes, so any lookup of the fast hit
// search outside its two-cell frame (which on the GPU would be a memory fault) aborts here.
#include <cstdio>
#include <random>

#include "ray_harness.cpp"

int main()
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long rays = 0, bad_total = 0;
    const int shapes[][2] = {{97, 131}, {64, 64}, {5, 300}, {300, 7}, {1, 1}, {33, 2}};
    for (const auto& sh : shapes)
    {
        const int rows = sh[0], cols = sh[1];
        for (double dens : {0.0, 0.02, 0.3})
        {
            std::vector<uint8_t> map((size_t)rows * cols), E((size_t)rows * cols);
            for (auto& c : map) c = U(rng) < dens ? 0 : 255;
            hh_build_clearance(map.data(), rows, cols, E.data());
            for (double zmax : {0.4, 1.0, 17.3, 500.0})
            {
                const long n = 4000;
                std::vector<double> sx(n), sy(n), th(n);
                for (long r = 0; r < n; ++r)
                {
                    const double spread = (r % 4 == 0) ? 40.0 : 0.0;  // a quarter of the rays start outside the grid
                    sx[r] = -spread + U(rng) * (cols + 2 * spread);
                    sy[r] = -spread + U(rng) * (rows + 2 * spread);
                    th[r] = (r % 7 == 0) ? (int)(U(rng) * 8) * 0.7853981633974483 : (U(rng) * 14.0 - 7.0);
                    if (r % 11 == 0)
                    {  // on cell corners
                        sx[r] = (double)(long)sx[r];
                        sy[r] = (double)(long)sy[r];
                    }
                }
                long first;
                unsigned long long rf, rw;
                bad_total += hh_rays_compare(map.data(), E.data(), rows, cols, n, sx.data(), sy.data(), th.data(), zmax, &first,
                                             &rf, &rw, nullptr);
                rays += n;
            }
        }
    }
    // exact jumps, forced ties included
    long adv_bad = 0;
    for (int t = 0; t < 3000; ++t)
    {
        const double th = U(rng) * 6.283185307179586;
        double dx = __builtin_cos(th), dy = __builtin_sin(th);
        if (t % 3 == 0) dx = __builtin_ldexp((double)(2 * (long)(U(rng) * (1 << 20)) + 1), 10 - 53 - 0) * 2048.0;
        const int jumps[5] = {1, 7, 64, 3, 31};
        long js;
        adv_bad += hh_advance_check(1024.0 + U(rng) * 1000.0, 3.0 + U(rng) * 4000.0, dx <= 1.0 ? dx : 0.5, dy, 400, jumps, 5, &js);
    }
    std::printf("rays %ld mismatches %ld advance_mismatches %ld\n", rays, bad_total, adv_bad);
    return (bad_total || adv_bad) ? 1 : 0;
}
