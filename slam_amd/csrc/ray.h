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
// of u and fl(x_k + dx) = x_k + inc with the constant inc = dx rounded to the u grid (unless dx is
// an exact tie, an odd multiple of u/2).  Hence x_{k+m} = fma(m, inc, x_k) exactly, for as long as
// the exact sums stay inside the binade; slam_axis_setup() derives inc and that count (with
// margin).  The add that crosses a binade edge, the few adds inside the margin before it, ties and
// |x| < 4 use real adds.
//
// (b) Clearance.  Per cell and per octant of the ray's direction a uint8 C (clearance.h, built by the
// library when the map is set, stored with a two-cell frame of zeros): C = 0 for an occupied cell
// (< 128), otherwise >= 1 and such that every cell the ray can visit within its next
// floor((C-1)/max(|dx|,|dy|)) samples is free and in bounds, so those samples need no read.  (Round 1
// used one Chebyshev distance for all directions; a plane per octant only looks ahead, which about
// halves the reads.)  Distance-limit tests are only needed near k = zmax and are done by the literal
// loop (the tail).
#pragma once
#include <stdint.h>

#include "detmath.h"

#if defined(__HIP_DEVICE_COMPILE__)
DM_HD int slam_f2i(double v) { return (int)v; }  // v_cvt_i32_f64: truncates, saturates
#else
// x86 cvttsd2si semantics for out-of-range input (the reference's host behaviour), without UB.
DM_HD int slam_f2i(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1); }
#endif

// approximate 1/x (1 ulp on the device): used only where a safety factor absorbs the error
#if defined(__HIP_DEVICE_COMPILE__)
DM_HD float slam_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
#else
DM_HD float slam_rcpf(float x) { return 1.0f / x; }
#endif

// i * pitch for -2 <= i < 2^23 and pitch < 2^23 (a full-rate 24-bit multiply on the device)
#if defined(__HIP_DEVICE_COMPILE__)
DM_HD int slam_mul24(int a, int b) { return __mul24(a, b); }
DM_HD unsigned slam_mul24(unsigned a, unsigned b) { return __umul24(a, b); }
#else
DM_HD int slam_mul24(int a, int b) { return a * b; }
DM_HD unsigned slam_mul24(unsigned a, unsigned b) { return a * b; }
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
//
// One axis of the accumulated position.  While v stays inside one binade [2^E, 2^(E+1)) with ulp u,
// fl(v + d) = v + inc for the constant inc = d rounded to the u grid, PROVIDED d is not an exact
// tie (an odd multiple of u/2; then round-half-even makes the increment depend on the parity of
// v/u).  slam_axis_setup computes inc with the add-and-subtract-a-big-constant rounding trick and
// the number `rem` of further adds that provably stay inside the binade (with margin).
// A tie d = (j + 1/2) u: round-half-even makes fl(v + d) an EVEN multiple of u whatever v was, so
// from an even v/u on the increment is constant again, the even one of j*u and (j+1)*u, and that is
// the value the rounding trick yields (C/u is even).  Only an odd v/u needs one real add first.
// |v| < 4, zeros, subnormals and non-finite values get rem = 0: such a stretch is walked with real
// adds, which are always right.  (With 64 rays per wavefront and an 11-bit residual, one wavefront
// in thirty holds a tie: walking those sample by sample cost more than all other rays together.)
struct slam_axis
{
    double v;    // accumulated coordinate of the current sample
    double inc;  // constant increment inside the current binade (valid while rem > 0)
    int rem;     // adds that can still be replaced by fma(m, inc, v)
};

// The part of slam_axis_setup that depends on the coordinate alone.  Every beam of a particle starts from the same
// sensor position, so the particle-lane kernel prepares the two axes once per particle and runs only
// slam_axis_setup_pre per beam.
struct slam_axis_pre
{
    double v;        // the coordinate
    float nup, ndn;  // exact distance to the upper / lower edge of v's binade (in magnitude), rounded to float
    int e;           // biased exponent of v
    bool in_range;   // 4 <= |v| < 2^53
    bool odd;        // v / ulp(v) is odd
    bool neg;
};

DM_HD void slam_axis_prepare(slam_axis_pre* p, double v)
{
    const uint64_t b = dm_bits(v);
    const int e = (int)((b >> 52) & 0x7ff);
    const double av = dm_fabs(v);
    const double lo = dm_from_bits((uint64_t)e << 52);
    p->v = v;
    p->nup = (float)((lo + lo) - av);  // exact differences
    p->ndn = (float)(av - lo);
    DM_TIE(p->nup, v);  // (kept as the two floats: the compiler would otherwise keep the doubles and convert per ray)
    DM_TIE(p->ndn, v);
    p->e = e;
    p->in_range = (unsigned)(e - 0x401) <= (unsigned)(0x433 - 0x401);
    p->odd = (b & 1) != 0;
    p->neg = (b >> 63) != 0;
}

DM_HD void slam_axis_setup_pre(slam_axis* a, const slam_axis_pre* p, double d)
{
    // straight-line code (every lane of a wavefront runs it once per ray): all quantities are
    // computed, the validity conditions select at the end
    int e = p->e;
    DM_TIE(e, d);  // C and half_ulp are rebuilt per ray (two instructions each) instead of living in four register pairs
    // C = 1.5 * 2^E: d + C lies in [2^E, 2^(E+1)) for |d| <= 1, so fl(d + C) is d rounded to the u grid
    const double C = dm_from_bits(((uint64_t)e << 52) | 0x0008000000000000ull);
    const double r = (d + C) - C;                                              // exact difference
    const double t = dm_fabs(d - r);                                           // exact residual, <= u/2
    const double half_ulp = dm_from_bits((uint64_t)(unsigned)(e - 53) << 52);  // u/2 = 2^(E-53)
    const bool grow = (r > 0.0) != p->neg;
    const float num = grow ? p->nup : p->ndn;          // distance to the binade edge the coordinate moves towards
    float q = num * slam_rcpf((float)dm_fabs(r));      // |r| >= u >= 2^-50: no float underflow
    q = q > 0.0f ? q : 0.0f;                           // (also NaN -> 0)
    q = q < 1.0e6f ? q : 1.0e6f;
    // adds that land strictly inside the binade: ceil(num/|r|) - 1; the factor covers the float
    // rounding of q (relative 3e-7) and makes an exact integer quotient round down
    int rem = (int)(q * 0.99999f);
    if (r == 0.0) rem = 1 << 20;  // fl(v + d) == v for every add
    // |v| < 4 or |v| >= 2^53 (or inf/nan), |d| > 1, a tie from an odd v/u (one real add makes it
    // even): real adds only
    const bool ok = p->in_range & (dm_fabs(d) <= 1.0) & !((t == half_ulp) & p->odd);
    a->v = p->v;
    a->inc = ok ? r : 0.0;
    a->rem = ok ? rem : 0;
}

DM_HD void slam_axis_setup(slam_axis* a, double d)
{
    slam_axis_pre p;
    slam_axis_prepare(&p, a->v);
    slam_axis_setup_pre(a, &p, d);
}

// One real add; re-derives inc/rem when the add left the binade or no constant increment was known.
DM_HD void slam_axis_real_add(slam_axis* a, double d)
{
    const double nv = a->v + d;
    const bool crossed = (dm_bits(nv) >> 52) != (dm_bits(a->v) >> 52);
    a->v = nv;
    if (crossed || a->rem == 0)
        slam_axis_setup(a, d);
    else
        a->rem -= 1;
}

// Advance both axes by m samples (m >= 1), exactly as m real adds would.
DM_HD void slam_advance(slam_axis* X, slam_axis* Y, double dx, double dy, int m)
{
    for (;;)
    {
        const int r = X->rem < Y->rem ? X->rem : Y->rem;
        if (m <= r)
        {  // hot path: one fma per axis; v + m*inc is a multiple of ulp inside the binade -> exact
            const double tf = (double)m;
            X->v = dm_fma(tf, X->inc, X->v);
            Y->v = dm_fma(tf, Y->inc, Y->v);
            X->rem -= m;
            Y->rem -= m;
            return;
        }
        if (r > 0)
        {
            const double tf = (double)r;
            X->v = dm_fma(tf, X->inc, X->v);
            Y->v = dm_fma(tf, Y->inc, Y->v);
            X->rem -= r;
            Y->rem -= r;
            m -= r;
        }
        slam_axis_real_add(X, dx);
        slam_axis_real_add(Y, dy);
        if (--m == 0) return;
    }
}

// Localization-mode hit search: identical result to ray_walk<false> on the uint8 map the
// clearance grid was built from.
//
// Ep is the PADDED clearance plane of the ray's octant: (rows+4) x pitch bytes, pitch = cols+4, cell
// (i,j) at Ep[(i+2)*pitch + (j+2)], the two-cell frame filled with 0 (and counted as occupied when the
// planes are built).  A jump chosen from clearance C lands in the cone's column a <= C, and the cone
// (two cells wider than the ray can get) meets the frame before the ray does, so a landing sample is
// either in bounds or in the first ring outside; the frame lets the hot loop read it without a bounds
// test, and "0" means "hit or left the grid" (told apart afterwards).
// k_safe = max(0, floor(zmax) - 1): for samples k <= k_safe the distance test of
// slam/raycast.cpp:216 cannot fire (k^2 <= zmax^2 - 2 zmax + 1).
//
// The start cell of a ray (the same for every beam of a particle) and its clearance in the plane of the
// ray's octant (0 when the origin is outside the grid).
DM_HD void slam_ray_origin(const uint8_t* Ep, unsigned plane_off, int pitch, int rows, int cols, double sx, double sy, int* i0,
                           int* j0, unsigned* e0, unsigned* reads)
{
    const int i = slam_f2i(((double)rows - sy) - 1.0), j = slam_f2i(sx);
    unsigned e = 0;
    if (slam_in_bounds(rows, cols, i, j))
    {
        e = Ep[(unsigned)(slam_mul24(i, pitch) + j + 2 * pitch + 2) + plane_off];
        if (reads) ++*reads;
    }
    *i0 = i;
    *j0 = j;
    *e0 = e;
}

// Ep = the first plane, plane_off = byte offset of the ray's plane: every read is Ep[32-bit offset], which on the
// device is one load with a scalar base and a 32-bit lane offset (the planes together stay below 4 GiB, see
// slam_cone_planes_fit).
// PX, PY: the start position, prepared (slam_axis_prepare): the same for every beam of a particle.
// k_lim: search limit in samples (>= k_safe: none).  The caller that only needs the beam likelihood passes
// floor(z + 4 sigma) + 2: a hit at a later sample lies farther than z + 4 sigma from the origin (|(dx,dy)| = 1 to the
// last bit and the accumulated rounding is below 1e-6 cells), where slam/util.cpp:14-19 clamps the density to 0
// whatever the distance, and with k_lim < k_safe a ray without any hit is clamped as well (zmax - z > 4 sigma).  When
// the limit ends the search the result is (-2, -2), cell -2, which slam_beam_logw maps to the clamped value; hits
// within the limit and rays that leave the grid are reported as always.
#define SLAM_NO_SAMPLE_LIMIT 0x7fffffff
// study hook (tools/slowpath_sim.cpp counts how often the re-synchronisation block of the jump loop runs): nothing in the library
#ifndef SLAM_RAY_TRACE_RESYNC
#define SLAM_RAY_TRACE_RESYNC(reads_so_far) ((void)0)
#endif
DM_HD void ray_fast_L0(const uint8_t* Ep, unsigned plane_off, int pitch, int rows, int cols, const slam_axis_pre* PX,
                       const slam_axis_pre* PY, double dx, double dy, double zmax2, int k_safe, int k_lim, int i0, int j0,
                       unsigned e0, slam_ray_result* res, unsigned* reads)
{
    const double sx = PX->v, sy = PY->v;
    // the increments of the start position first: they do not depend on e0, whose load is still in flight on the device
    slam_axis X, Y;
    slam_axis_setup_pre(&X, PX, dx);
    slam_axis_setup_pre(&Y, PY, dy);
    const float adx = (float)dm_fabs(dx), ady = (float)dm_fabs(dy);
    // samples to advance = 1 + floor((e-1) * 0.9999 / max(|dx|,|dy|)); the 1e-4 margin covers the
    // float roundings (2^-23 relative on a value < 400), the single fused multiply-add makes
    // host and device agree
    const float sc = 0.9999f * slam_rcpf(adx > ady ? adx : ady);
    const float sc1 = 1.0f - sc;
    res->hx = -1.0;
    res->hy = -1.0;
    res->hcell = -1;
    const unsigned org = plane_off + 2u * (unsigned)pitch + 2u;  // cell (0,0) (offsets wrap modulo 2^32: frame cells lie before it)
    const double rows_d = (double)rows;
    // accumulated rounding stays far below the distance margin only for moderate coordinates;
    // anything farther out starts (and stays) outside every supported grid: literal loop only.
    // (the literal loop then runs the whole range instead of only its last samples: a ray that starts just
    // outside a 2^20-cell side and heads inward is walked as far as the reference walks it)
    const bool far_out = !(dm_fabs(sx) < 1048576.0 && dm_fabs(sy) < 1048576.0);
    const bool limited = !far_out && k_lim < k_safe;
    int left = far_out ? 0 : (limited ? k_lim : k_safe);
    const int tail = far_out ? k_safe + 16 : 16;
    double x = sx, y = sy;
    // sample 0 (the start cell) is never tested (slam/raycast.cpp:153-157); its clearance still
    // tells how far the first jump may go.
    int i = i0, j = j0;
    unsigned e = e0;
    if (e == 0 && left > 0)
    {
        // start cell occupied or outside the grid: samples still inside it are skipped by the
        // reference's same-cell test; walk until the ray leaves it
        const int ci0 = i, cj0 = j;
        do
        {
            x += dx;
            y += dy;
            --left;
            i = slam_f2i((rows_d - y) - 1.0);
            j = slam_f2i(x);
        } while (i == ci0 && j == cj0 && left > 0);
        if (i != ci0 || j != cj0)
        {
            if (!slam_in_bounds(rows, cols, i, j)) return;  // slam/raycast.cpp:176
            e = Ep[(unsigned)(slam_mul24(i, pitch) + j) + org];
            if (reads) ++*reads;
            if (e == 0)
            {  // slam/raycast.cpp:183-189
                res->hx = x;
                res->hy = y;
                res->hcell = i * cols + j;
                return;
            }
        }
        X.v = x;
        Y.v = y;
        slam_axis_setup(&X, dx);
        slam_axis_setup(&Y, dy);
    }
    if (left > 0)
    {
        // (x, y) is an in-bounds sample with clearance e >= 1
        int budget = X.rem < Y.rem ? X.rem : Y.rem;  // samples the fma path may take before a re-synchronisation
        if (budget > left) budget = left;
        int budget0 = budget;
        double incx = X.inc, incy = Y.inc;
        for (;;)
        {
            // advance m samples (any m' <= m would be legal too): one fma per axis ...
            const int m = (int)__builtin_fmaf((float)e, sc, sc1);
            int t = m < budget ? m : budget;
            budget -= t;
            double tf = (double)t;
            x = dm_fma(tf, incx, x);  // exact, see slam_axis
            y = dm_fma(tf, incy, y);
            bool last = false;
            if (__builtin_expect(t < m, 0))
            {
                // ... unless a binade edge, |coordinate| < 4 or the end of the distance-safe range
                // is in the way: one real add per axis, then jump on
                X.v = x;
                Y.v = y;
                X.rem -= budget0;
                Y.rem -= budget0;
                left -= budget0;
                if (left > 0)
                {
                    SLAM_RAY_TRACE_RESYNC(reads ? *reads : 0u);
                    slam_axis_real_add(&X, dx);
                    slam_axis_real_add(&Y, dy);
                    left -= 1;
                    budget = X.rem < Y.rem ? X.rem : Y.rem;
                    if (budget > left) budget = left;
                    budget0 = budget;
                    t = m - t - 1;
                    t = t < budget ? t : budget;
                    budget -= t;
                    tf = (double)t;
                    incx = X.inc;
                    incy = Y.inc;
                    x = dm_fma(tf, incx, X.v);
                    y = dm_fma(tf, incy, Y.v);
                }
                else
                {
                    budget0 = 0;
                    last = true;  // this sample is k_safe (or the limit): test it, then continue with the literal loop
                }
            }
            i = slam_f2i((rows_d - y) - 1.0);
            j = slam_f2i(x);
#ifdef SLAM_RAY_DEBUG
            if (i < -2 || i > rows + 1 || j < -2 || j > cols + 1) __builtin_trap();  // frame guarantee violated
#endif
            e = Ep[(unsigned)(slam_mul24(i, pitch) + j) + org];
            if (reads) ++*reads;
            if (e == 0 || last) break;
        }
        if (e == 0)
        {
            if (slam_in_bounds(rows, cols, i, j))
            {  // slam/raycast.cpp:183-189
                res->hx = x;
                res->hy = y;
                res->hcell = i * cols + j;
            }
            return;  // otherwise slam/raycast.cpp:176
        }
    }
    if (limited)
    {  // no hit within k_lim samples and still inside the grid: whatever follows is clamped
        res->hx = -2.0;
        res->hy = -2.0;
        res->hcell = -2;
        return;
    }
    // the literal loop from sample k_safe on (dedupe against the previous sample's cell, bounds,
    // hit, distance limit)
    int pi = i, pj = j;
    for (int it = 0; it < tail; ++it)
    {
        x += dx;
        y += dy;
        i = slam_f2i((rows_d - y) - 1.0);
        j = slam_f2i(x);
        if (i == pi && j == pj) continue;
        if (!slam_in_bounds(rows, cols, i, j)) return;
        if (reads) ++*reads;
        if (Ep[(unsigned)(slam_mul24(i, pitch) + j) + org] == 0)
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

// Octant of a direction, the index of the cone-clearance plane its ray reads (clearance.h).
DM_HD int slam_octant(double dx, double dy)
{
    return (dx < 0.0 ? 1 : 0) | (dy < 0.0 ? 2 : 0) | (dm_fabs(dx) < dm_fabs(dy) ? 4 : 0);
}

// planes: the 8 cone-clearance planes, plane_stride bytes apart (0: one Chebyshev-clearance plane serves every octant).
DM_HD void ray_fast_L(const uint8_t* planes, unsigned plane_stride, int pitch, int rows, int cols, double sx, double sy, double dx,
                      double dy, double zmax2, int k_safe, slam_ray_result* res, unsigned* reads, int k_lim = SLAM_NO_SAMPLE_LIMIT)
{
    const unsigned plane_off = (unsigned)slam_octant(dx, dy) * plane_stride;
    int i0, j0;
    unsigned e0;
    slam_ray_origin(planes, plane_off, pitch, rows, cols, sx, sy, &i0, &j0, &e0, reads);
    slam_axis_pre PX, PY;
    slam_axis_prepare(&PX, sx);
    slam_axis_prepare(&PY, sy);
    ray_fast_L0(planes, plane_off, pitch, rows, cols, &PX, &PY, dx, dy, zmax2, k_safe, k_lim, i0, j0, e0, res, reads);
}

// The search limit of a beam of squared range z*z (see ray_fast_L0).
DM_HD int slam_sample_limit(double z, double stddev)
{
    const double lim = z + 4.0 * stddev;
    return lim < 1.0e9 ? slam_f2i(lim) + 2 : SLAM_NO_SAMPLE_LIMIT;  // (NaN: no limit)
}

// ------------------------------------------------------------------------------------------------
// Beam likelihood, slam/raycast.cpp:233-241 + slam/util.cpp:9-19, returning log(w) for the
// accumulation of slam/mcl.cpp:73.  norm = 1/(stddev*sqrt(2 pi)), z = sqrt(z2), zmax = sqrt(zmax2).
// log_eps = dm_log(0.0 + 0.1): the value of every beam whose error exceeds 4 sigma (pdf clamp).
// clamp = 4.0 * stddev, computed once by the caller (the host for the kernels: a scalar argument instead of a register pair).
// inv_stddev: 0, or RN(1 / stddev) for a stddev in [1e-100, 1e100]: the quotient xarg / stddev of slam/util.cpp:11 is then
// formed from the reciprocal with two correction steps (5 multiply-adds instead of the 12-instruction division sequence).
// The result is the correctly rounded quotient, i.e. the same bits: q0 = RN(a y) is within 2 ulp of a/b (y = RN(1/b));
// q1 = RN(q0 + RN(a - b q0) y) is within 1 ulp (faithful), for which r1 = a - b q1 is exact and RN(q1 + r1 y) = RN(a/b)
// (Markstein's theorem, which needs exactly: y the correctly rounded reciprocal, q1 faithful, no over/underflow:
// |a| is 0 or at least 1e-17 here and |a / b| <= 4).
DM_HD double slam_beam_logw(const slam_ray_result* r, double sx, double sy, double z, double zmax, double stddev, double clamp,
                            double norm, double log_eps, double inv_stddev = 0.0)
{
    double xarg;
    if (r->hx == -2.0) return log_eps;  // the search stopped at its limit: beyond the 4 sigma clamp (ray_fast_L0)
    if (r->hx == -1.0)
        xarg = z - zmax;
    else
    {
        const double ex = r->hx - sx, ey = r->hy - sy;
        xarg = dm_sqrt(ex * ex + ey * ey) - z;
    }
    if (dm_fabs(xarg) > clamp) return log_eps;  // slam/util.cpp:14-19
    double q;
    if (inv_stddev != 0.0)
    {
        const double q0 = xarg * inv_stddev;
        const double q1 = dm_fma(dm_fma(-q0, stddev, xarg), inv_stddev, q0);
        q = dm_fma(dm_fma(-q1, stddev, xarg), inv_stddev, q1);
    }
    else
        q = xarg / stddev;
    const double p = norm * dm_exp(-0.5 * (q * q));
    return dm_log(p + 0.1);
}

// The same likelihood in two steps, for a kernel whose rays are mostly clamped (a wide cloud: k_weights_L<0>).
// slam_beam_d2: the squared distance origin -> hit as slam_beam_logw forms it (-1: no hit, -2: the search ended at its limit).
// slam_beam_clamped: true only if slam_beam_logw is CERTAIN to return log_eps (slam/util.cpp:14-19), decided without the
// square root: with t = z +- clamp, d2 beyond t^2 by a relative 1e-7 puts sqrt(d2) beyond t by 5e-8 t, far more than the
// roundings of the square root and of the subtraction (2^-52 each) can move xarg across the clamp.  Everything else (the
// few rays within 4 sigma of their measured range, and a sliver of 1e-7 around it) goes through slam_beam_logw_d2,
// which is slam_beam_logw from d2 on: same statements, same bits.
DM_HD double slam_beam_d2(const slam_ray_result* r, double sx, double sy)
{
    if (r->hx == -2.0 || r->hx == -1.0) return r->hx;
    const double ex = r->hx - sx, ey = r->hy - sy;
    return ex * ex + ey * ey;
}

DM_HD bool slam_beam_clamped(double d2, double z, double zmax, double clamp)
{
    if (d2 == -2.0) return true;
    if (d2 == -1.0) return dm_fabs(z - zmax) > clamp;  // exact: the very test of slam_beam_logw
    const double hi = z + clamp, lo = z - clamp;
    if (d2 > (hi * hi) * 1.0000001) return true;
    return lo > 0.0 && d2 < (lo * lo) * 0.9999999;
}

DM_HD double slam_beam_logw_d2(double d2, double z, double zmax, double stddev, double clamp, double norm, double log_eps,
                               double inv_stddev = 0.0)
{
    double xarg;
    if (d2 == -2.0) return log_eps;
    if (d2 == -1.0)
        xarg = z - zmax;
    else
        xarg = dm_sqrt(d2) - z;
    if (dm_fabs(xarg) > clamp) return log_eps;  // slam/util.cpp:14-19
    double q;
    if (inv_stddev != 0.0)
    {
        const double q0 = xarg * inv_stddev;
        const double q1 = dm_fma(dm_fma(-q0, stddev, xarg), inv_stddev, q0);
        q = dm_fma(dm_fma(-q1, stddev, xarg), inv_stddev, q1);
    }
    else
        q = xarg / stddev;
    const double p = norm * dm_exp(-0.5 * (q * q));
    return dm_log(p + 0.1);
}
