// ray.h — per-ray device functions of the particle-filter step (host-compilable for unit tests).
//
// Reproduces, bit for bit, the sample sequence of slam/raycast.cpp:143-223 (raycast_mapping):
//   x_k = fl(x_{k-1} + dx), y_k = fl(y_{k-1} + dy)   (positions are ACCUMULATED, never x0 + k*dx)
//   cell_k = ((int)(rows - y_k - 1), (int)x_k)       (slam/util.h:30-33, truncation)
// Two implementations:
//   ray_walk<>   the literal cell-by-cell loop (SLAM mode writes; localization mode counts cells);
//   ray_fast_L() localization mode only: the same result with far fewer map reads, using
//                (a) exact m-step jumps of the accumulated position and (b) a clearance grid.
//
// (a) Exact jumps.  While x stays inside one binade (fixed exponent, ulp u) every x_k is a multiple
// of u and fl(x_k + dx) = x_k + inc with a constant inc = dx rounded to the u grid.  A tie
// (dx an odd multiple of u/2) breaks constancy only for the first add inside the binade: after
// it x/u is even and round-half-even then yields a constant increment as well.  So after two
// real adds inside a binade, inc = x_k - x_{k-1} (exact) and x_{k+m} = fma(m, inc, x_k) exactly,
// for as long as the exact sums stay inside the binade (axis_rem() bounds that count with margin).
// Binade crossings and the first two steps of a ray use real adds.
//
// (b) Clearance grid E (uint8 per cell, built by the library when the map is set): E = 0 for an
// occupied cell (< 128), otherwise min(255, Chebyshev distance to the nearest occupied or
// out-of-grid cell) >= 1.  From a sample inside a cell with E >= 1, all cells within Chebyshev
// distance E-1 are free and in bounds, and a step moves at most max(|dx|,|dy|) per axis, so the
// next floor((E-1)/max(|dx|,|dy|)) samples can neither hit nor leave the grid and need no read.
// Distance-limit tests are only needed near k = zmax and are done by the literal loop (the tail).
#pragma once
#include <stdint.h>

#include "detmath.h"

#if defined(__HIP_DEVICE_COMPILE__)
DM_HD int slam_f2i(double v) { return (int)v; }  // v_cvt_i32_f64: truncates, saturates
#else
// x86 cvttsd2si semantics for out-of-range input (the reference's host behaviour), without UB.
DM_HD int slam_f2i(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1); }
#endif

DM_HD void slam_cell_of(int rows, double x, double y, int* i, int* j)
{
    *i = slam_f2i(((double)rows - y) - 1.0);
    *j = slam_f2i(x);
}

DM_HD bool slam_in_bounds(int rows, int cols, int i, int j)
{
    return (unsigned)i < (unsigned)rows && (unsigned)j < (unsigned)cols;
}

struct slam_ray_result
{
    double hx, hy;  // hit position or (-1,-1)
    int hcell;      // i*cols + j of the hit cell or -1
};

struct slam_ray_stats
{
    unsigned long long cells_read, cells_written, steps;
};

// ------------------------------------------------------------------------------------------------
// Literal walk (slam/raycast.cpp:163-220).  WRITE: apply the free / occupied tables to the map.
// max_iter bounds the loop where the reference would spin forever (|coordinate| >= 2^31).
template <bool WRITE, bool STATS>
DM_HD void ray_walk(uint8_t* map, int rows, int cols, double sx, double sy, double dx, double dy, double z2,
                    double zmax2, const uint8_t* lut_free, const uint8_t* lut_occ, int max_iter, slam_ray_result* res,
                    slam_ray_stats* st)
{
    double x = sx, y = sy;
    int i, j;
    slam_cell_of(rows, x, y, &i, &j);
    int pi = i, pj = j;
    res->hx = -1.0;
    res->hy = -1.0;
    res->hcell = -1;
    bool drew = false, found = false;
    for (int it = 0; it < max_iter && (!drew || !found); ++it)
    {
        x += dx;
        y += dy;
        if (STATS) st->steps++;
        slam_cell_of(rows, x, y, &i, &j);
        if (i == pi && j == pj) continue;
        if (!slam_in_bounds(rows, cols, i, j)) break;
        const size_t off = (size_t)i * (size_t)cols + (size_t)j;
        const uint8_t c = map[off];
        if (STATS) st->cells_read++;
        if (!found && c < 128)
        {
            res->hx = x;
            res->hy = y;
            res->hcell = i * cols + j;
            found = true;
        }
        const double ex = x - sx, ey = y - sy;
        const double d2 = ex * ex + ey * ey;
        if (d2 < z2)
        {
            if (WRITE)
            {
                map[off] = lut_free[c];
                if (STATS) st->cells_written++;
            }
        }
        else
        {
            if (!drew && z2 != zmax2)
            {
                if (WRITE)
                {
                    map[off] = lut_occ[c];
                    if (STATS) st->cells_written++;
                }
            }
            drew = true;
        }
        if (d2 >= zmax2) break;
        pi = i;
        pj = j;
    }
}

// ------------------------------------------------------------------------------------------------
// Exact-jump machinery.
struct slam_axis
{
    double v;    // current accumulated coordinate
    double inc;  // steady-state increment (valid while rem > 0)
    int rem;     // jumps that are still provably exact
    int cnt;     // consecutive real adds that stayed inside the current binade
};

DM_HD int slam_axis_rem(double v, double inc)
{
    if (inc == 0.0) return 1 << 20;  // fl(v + d) == v forever
    const uint64_t b = dm_bits(v);
    const int e = (int)((b >> 52) & 0x7ff);
    const double ai = dm_fabs(inc);
    if (e == 0 || e == 0x7ff || ai < 1e-30) return 0;
    const double a = dm_fabs(v);
    const double lo = dm_from_bits((uint64_t)e << 52);
    const bool grow = (inc > 0.0) == ((b >> 63) == 0);
    const double num = grow ? (lo + lo) - a : a - lo;  // exact
    float q = (float)num / (float)ai;
    q = q < 1.0e6f ? q : 1.0e6f;
    const int r = (int)(q * 0.999f) - 2;  // margin: float rounding + one full step from the edge
    return r < 0 ? 0 : r;
}

DM_HD void slam_axis_real_step(slam_axis* s, double d)
{
    const double nv = s->v + d;
    const bool same = (dm_bits(nv) >> 52) == (dm_bits(s->v) >> 52);
    s->cnt = same ? s->cnt + 1 : 0;
    s->inc = nv - s->v;  // exact when same binade
    s->v = nv;
    s->rem = (s->cnt >= 2) ? slam_axis_rem(nv, s->inc) : 0;
}

DM_HD void slam_axis_step(slam_axis* s, double d, int t)
{
    if (s->rem >= t)
    {
        s->v = dm_fma((double)t, s->inc, s->v);  // exact: multiple of ulp inside the binade
        s->rem -= t;
    }
    else
        slam_axis_real_step(s, d);  // only reached with t == 1
}

DM_HD void slam_advance(slam_axis* X, slam_axis* Y, double dx, double dy, int m)
{
    while (m > 0)
    {
        int r = X->rem < Y->rem ? X->rem : Y->rem;
        if (r < 1) r = 1;
        const int t = m < r ? m : r;
        slam_axis_step(X, dx, t);
        slam_axis_step(Y, dy, t);
        m -= t;
    }
}

// Localization-mode hit search: identical result to ray_walk<false> on the uint8 map the
// clearance grid `E` was built from.  k_safe = max(0, floor(zmax) - 1): for samples k <= k_safe
// the distance test of slam/raycast.cpp:216 cannot fire (k^2 <= zmax^2 - 2 zmax + 1).
DM_HD void ray_fast_L(const uint8_t* E, int rows, int cols, double sx, double sy, double dx, double dy, double zmax2,
                      int k_safe, slam_ray_result* res, unsigned* reads)
{
    res->hx = -1.0;
    res->hy = -1.0;
    res->hcell = -1;
    slam_axis X = {sx, 0.0, 0, 0}, Y = {sy, 0.0, 0, 0};
    int ci0, cj0;
    slam_cell_of(rows, sx, sy, &ci0, &cj0);
    // accumulated rounding stays far below the distance margin only for moderate coordinates;
    // anything farther out starts (and stays) outside every supported grid: literal loop only.
    if (!(dm_fabs(sx) < 1048576.0 && dm_fabs(sy) < 1048576.0)) k_safe = 0;
    const float adx = (float)dm_fabs(dx), ady = (float)dm_fabs(dy);
    const float scale = 0.9999f / (adx > ady ? adx : ady);
    int k = 0;
    int i, j;
    for (;;)
    {
        slam_cell_of(rows, X.v, Y.v, &i, &j);
        const bool same0 = (i == ci0) && (j == cj0);  // still inside the start cell: never tested
        const bool inb = slam_in_bounds(rows, cols, i, j);
        if (!inb && !same0) return;  // slam/raycast.cpp:176
        unsigned e = 0;
        if (inb)
        {
            e = E[(size_t)i * (size_t)cols + (size_t)j];
            if (reads) ++*reads;
        }
        if (e == 0 && !same0)
        {  // slam/raycast.cpp:183-189
            res->hx = X.v;
            res->hy = Y.v;
            res->hcell = i * cols + j;
            return;
        }
        if (k >= k_safe) break;
        int m = 1;
        if (e > 1) m += (int)((float)(e - 1) * scale);
        if (m > k_safe - k) m = k_safe - k;
        slam_advance(&X, &Y, dx, dy, m);
        k += m;
    }
    // tail: the literal loop from sample k_safe on (dedupe against the previous sample's cell,
    // bounds, hit, distance limit).
    double x = X.v, y = Y.v;
    int pi = i, pj = j;
    for (int it = 0; it < 16; ++it)
    {
        x += dx;
        y += dy;
        slam_cell_of(rows, x, y, &i, &j);
        if (i == pi && j == pj) continue;
        if (!slam_in_bounds(rows, cols, i, j)) return;
        if (reads) ++*reads;
        if (E[(size_t)i * (size_t)cols + (size_t)j] == 0)
        {
            res->hx = x;
            res->hy = y;
            res->hcell = i * cols + j;
            return;
        }
        const double ex = x - sx, ey = y - sy;
        if (ex * ex + ey * ey >= zmax2) return;
        pi = i;
        pj = j;
    }
}

// ------------------------------------------------------------------------------------------------
// Beam likelihood, slam/raycast.cpp:233-241 + slam/util.cpp:9-19, returning log(w) for the
// accumulation of slam/mcl.cpp:73.  norm = 1/(stddev*sqrt(2 pi)), z = sqrt(z2), zmax = sqrt(zmax2).
DM_HD double slam_beam_logw(const slam_ray_result* r, double sx, double sy, double z, double zmax, double stddev,
                            double norm)
{
    double xarg;
    if (r->hx == -1.0)
        xarg = z - zmax;
    else
    {
        const double ex = r->hx - sx, ey = r->hy - sy;
        xarg = dm_sqrt(ex * ex + ey * ey) - z;
    }
    double p = 0.0;
    if (!(dm_fabs(xarg) > 4.0 * stddev))
    {
        const double q = xarg / stddev;
        p = norm * dm_exp(-0.5 * (q * q));
    }
    return dm_log(p + 0.1);
}
