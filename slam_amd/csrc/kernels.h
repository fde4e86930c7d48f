// kernels.h — gfx950 kernels of the particle-filter step (included by slam_hip.hip only).
//
// Kernel groups (SURVEY.md §2 "what the build writes instead"):
//   K1  k_predict            slam/motion.cpp:9-32 + slam/util.cpp:45-51, one thread per particle
//   K2a k_weights_L          localization mode: one wavefront per particle, lanes over beams
//   K2b k_weights_S2 (slam_s2.h, the default), k_weights_S_wave, k_weights_S   SLAM mode: slam/raycast.cpp:143-223 with map updates, beams in order
//   K3  k_scan_chunks / k_downsweep / k_prefix_seq / k_select / k_gather   slam/mcl.cpp:157-203
//   K4  sort keys + gather (rocPRIM radix sort drives the permutation)     slam/mcl.cpp:232-239
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detmath.h"
#include "noise.h"
#include "ray.h"

namespace slamk
{
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// value of `v` in lane `l` for a wave-uniform l: a scalar read instead of an LDS permute
// 16-byte store that does not stay in the L2 (bulk copies whose lines are not read again soon: keeping them evicts
// the lines the walks re-read; on the fused tile copies of k_weights_S2 this took the kernel from 24.5 to 17.5 ms)
typedef unsigned int slam_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_streaming(uint4* to, const uint4& v)
{
    __builtin_nontemporal_store(*reinterpret_cast<const slam_u32x4*>(&v), reinterpret_cast<slam_u32x4*>(to));
}
__device__ __forceinline__ int bcast_i32(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ double bcast_f64(double v, int l) { return readlane_f64(v, __builtin_amdgcn_readfirstlane(l)); }
// value of `v` one lane below (lane 0 keeps its own): a DPP wave shift instead of an LDS permute
__device__ __forceinline__ int lane_below_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }

// ------------------------------------------------------------------------------------------------
struct PredictArgs
{
    int64_t n;
    double *x, *y, *th;
    double r1, t, r2;
    double s1, st, s2;  // the three sigmas of slam/motion.cpp:15-25 (uniform over particles)
    uint64_t seed, step, global_first;
    const double* injected;  // 3*n unit normals or nullptr
};

__global__ void __launch_bounds__(256) k_predict(PredictArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    double z[3];
    if (a.injected)
    {
        z[0] = a.injected[3 * p];
        z[1] = a.injected[3 * p + 1];
        z[2] = a.injected[3 * p + 2];
    }
    else
        slam_motion_normals(a.seed, a.step, a.global_first + (uint64_t)p, z);
    const double rotation_1 = a.r1 - z[0] * a.s1;
    const double translation = a.t - z[1] * a.st;
    const double rotation_2 = a.r2 - z[2] * a.s2;
    const double th = a.th[p];
    double s, c;
    dm_sincos(th + rotation_1, &s, &c);
    const double nx = a.x[p] + translation * c;
    const double ny = a.y[p] + translation * s;
    const double twopi = dm_from_bits(0x401921FB54442D18ull);
    double angle = dm_fmod_pos(th + rotation_1 + rotation_2, twopi);
    if (angle > dm_from_bits(0x400921FB54442D18ull)) angle -= twopi;
    a.x[p] = nx;
    a.y[p] = ny;
    a.th[p] = angle;
}

// slam/motion.cpp:34-56 sample_motion_model_velocity for every particle (no caller in the reference;
// same noise seam and streams as k_predict: draws v, w, gamma in that order).
struct PredictVelArgs
{
    int64_t n;
    double *x, *y, *th;
    double v, w, dt;
    double sv, sw, sg;  // the three sigmas of slam/motion.cpp:40-45
    uint64_t seed, step, global_first;
    const double* injected;
};

__global__ void __launch_bounds__(256) k_predict_velocity(PredictVelArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    double z[3];
    if (a.injected)
    {
        z[0] = a.injected[3 * p];
        z[1] = a.injected[3 * p + 1];
        z[2] = a.injected[3 * p + 2];
    }
    else
        slam_motion_normals(a.seed, a.step, a.global_first + (uint64_t)p, z);
    const double v = a.v + z[0] * a.sv;
    const double w = a.w + z[1] * a.sw;
    const double gamma = z[2] * a.sg;
    const double th = a.th[p];
    double s0, c0, s1, c1;
    dm_sincos(th, &s0, &c0);
    dm_sincos(th + (w + 1e-6) * a.dt, &s1, &c1);
    const double q = v / (w == 0 ? 1 : w);
    const double nx = a.x[p] - q * (s0 + s1);
    const double ny = a.y[p] + q * (c0 - c1);
    const double twopi = dm_from_bits(0x401921FB54442D18ull);
    double angle = dm_fmod_pos(th + w * a.dt + gamma * a.dt, twopi);
    if (angle > dm_from_bits(0x400921FB54442D18ull)) angle -= twopi;
    a.x[p] = nx;
    a.y[p] = ny;
    a.th[p] = angle;
}

// ------------------------------------------------------------------------------------------------
struct WeightArgs
{
    int64_t n;
    const double *x, *y, *th;
    uint8_t* grid;         // L/FAST: the 8 cone-clearance planes; L/WALK: raw map; S: per-particle maps
    size_t map_stride;     // S mode: bytes between particle maps; L/FAST: bytes between clearance planes (8 planes < 4 GiB; 0: one plane)
    int rows, cols, B;
    const double *angles, *z2, *z;  // per beam: angle, dist^2, sqrt(dist^2)
    double zmax2, zmax, stddev, norm, log_eps;
    double clamp4;                  // 4 * stddev (slam/util.cpp:14-19)
    double inv_stddev;              // RN(1 / stddev), or 0: divide (slam_beam_logw)
    double sd, stheta, srot;        // scanner displacement (slam/mcl.cpp:80-86) and rotation
    int k_safe, max_iter;
    double* w;
    double* hit_xy;   // optional n*B*2
    int32_t* hit_cell;  // optional n*B
    unsigned long long* stats;  // optional 3 counters
    const uint8_t* luts;        // 512 bytes: free table then occupied table
    // particle-lane decomposition (k_rays_pl / k_sum_lw)
    double* lw;                 // scratch: per (particle block, beam, lane) log-likelihoods
    int64_t p_first;            // first particle of this launch
    int64_t npblocks;           // 64-particle blocks in this launch
    int nseg, seg_beams;        // beams are cut into nseg segments of seg_beams
    const int32_t* order;       // optional: slot -> particle (pose-coherent order, see k_order_keys)
    int log_domain;             // SLAM_WEIGHT_LOG: store the sum of logs instead of its exp
    int clamp_limit;            // localization hit search: stop a ray once it is beyond z + 4 sigma (ray_fast_L0's k_lim); off when hits are recorded
    double defer_edge;          // deferred canvases: how far the sensor must be from the two-wide cells of row 0 / column 0 (slam_defer_safe)
};

// "Weights now, canvases later" (TiledDefer) rests on two rays of a scan sharing no cell beyond kstar samples, which holds
// for cells one unit wide.  The casts of slam/raycast.cpp:171 truncate toward zero, so x in (-1, 1) is column 0 and
// rows - y - 1 in (-1, 1) is row 0: those cells are two units wide and two rays can meet in one of them twice as far out.
// A particle whose sensor is at least edge = 2 sqrt(2) / sin(gap) + 2 away from both bands (-1, 0) has no ray that comes
// within 2 sqrt(2) of a point another ray has in such a band; the others take the ordinary pass right away.
__device__ __forceinline__ bool slam_defer_safe(double sx, double sy, double rows_d, double edge)
{
    const double u = sx, v = (rows_d - sy) - 1.0;
    const double du = u >= 0.0 ? u : (u <= -1.0 ? -1.0 - u : 0.0);
    const double dv = v >= 0.0 ? v : (v <= -1.0 ? -1.0 - v : 0.0);
    return du >= edge && dv >= edge;  // (false for a NaN pose)
}

// K2a.  VARIANT 0: ray_fast_L on the clearance grid.  VARIANT 1: literal walk on the raw map
// (counts cells).  One wavefront = one particle; lane l takes beams l, l+64, ...; the per-beam
// log-likelihoods are then added in beam order (slam/mcl.cpp:69-74) so the weight has the
// reference's rounding.
//
// The in-order sum.  VARIANT 0 serves wide clouds, where nearly every ray ends farther than 4 sigma from its measured
// range and the likelihood is the clamped constant log(0 + 0.1) (slam/util.cpp:14-19).  A lane decides "certainly clamped"
// from squared distances (slam_beam_clamped: no square root, quotient, exp or log, which in lock step all 64 lanes
// paid whenever ONE needed them: 62 % of the groups at 1.5 % of the rays) and otherwise appends (beam, squared hit
// distance) to a short list in LDS, in beam order.  kwl_flush evaluates the listed beams together (slam_beam_logw_d2:
// the same statements, the same bits) and then adds, every lane the same sum: the constant for each beam between two
// listed ones, the evaluated value for a listed one.  It runs when the list could overflow in the next group and at
// the end, so a converged cloud (every beam listed) is as correct as a wide one (a dozen of 720).
// Round 3: this list replaces a row of B doubles per wavefront (28.8 KB per workgroup at 720 beams: 5 wavefronts per
// SIMD at most).  A pass over a wide cloud is bound by each wavefront's chain of dependent clearance reads: with LDS
// padded to force 1..5 wavefronts per SIMD the 16384^2 shard took 864 / 465 / 330 / 263 / 223 ms, while removing a
// tenth of the instructions changed nothing.  MIN_WAVES asks the compiler for the register budget of that many.
enum { KWL_LIST = 128 };  // list entries per wavefront (at least 64: one group's worth)
struct KwlList
{
    double val[4][KWL_LIST];
    int beam[4][KWL_LIST];
};

// Adds the beams [b_done, upto) to acc in beam order; the n listed ones (beam numbers ascending, all in that range) hold
// their squared hit distance (EVAL) or their final log-likelihood.  Wave-uniform control flow; every lane adds the same.
template <bool EVAL>
__device__ __forceinline__ double kwl_flush(const WeightArgs& a, double* val, int* beam, int n, int b_done, int upto, double acc, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the list was written by other lanes)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (EVAL && n > 0)
    {
        for (int k0 = 0; k0 < n; k0 += 64)
            if (k0 + lane < n)
            {
                const int b = beam[k0 + lane];
                val[k0 + lane] = slam_beam_logw_d2(val[k0 + lane], a.z[b], a.zmax, a.stddev, a.clamp4, a.norm, a.log_eps, a.inv_stddev);
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int b = b_done;
    for (int k = 0; k <= n; ++k)
    {
        const int nb = k < n ? __builtin_amdgcn_readfirstlane(beam[k]) : upto;
        for (; b < nb; ++b) acc += a.log_eps;  // slam/mcl.cpp:69-74, a beam beyond the 4 sigma clamp
        if (k < n)
        {
            acc += val[k];
            ++b;
        }
    }
    return acc;
}

template <int VARIANT, int MIN_WAVES = 1>
__global__ void __launch_bounds__(256, MIN_WAVES) k_weights_L(WeightArgs a)
{
    __shared__ KwlList kwl;
    double* val = kwl.val[threadIdx.x >> 6];
    int* beam = kwl.beam[threadIdx.x >> 6];
    int n_list = 0, b_done = 0;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.n) return;  // whole wavefront
    // (a.order: the particles in place order, so that the wavefronts in flight work on one band of a large map)
    const int64_t p = a.order ? (int64_t)a.order[slot] : slot;
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    double acc = 0.0;
    slam_ray_stats st = {0, 0, 0};
    // the same for every beam of the particle (as in k_rays_pl): the prepared start coordinates and the start cell
    slam_axis_pre PX, PY;
    int i0 = 0, j0 = 0;
    if (VARIANT == 0)
    {
        slam_axis_prepare(&PX, sx);
        slam_axis_prepare(&PY, sy);
        slam_cell_of(a.rows, sx, sy, &i0, &j0);
    }
    const bool origin_in = slam_in_bounds(a.rows, a.cols, i0, j0);
    const unsigned off0 = (unsigned)(slam_mul24(i0, a.cols + 4) + j0 + 2 * (a.cols + 4) + 2);
    for (int b0 = 0;; b0 += 64)
    {
        // (one instance of kwl_flush in the code: the last trip of this loop only adds what is listed)
        const bool last_trip = b0 >= a.B;
        if (last_trip || n_list + 64 > KWL_LIST)
        {  // the next group might not fit: add what is listed (everything before b0) now
            const int upto = last_trip ? a.B : b0;
            acc = kwl_flush<VARIANT == 0>(a, val, beam, n_list, b_done, upto, acc, lane);
            n_list = 0;
            b_done = upto;
        }
        if (last_trip) break;
        const int b = b0 + lane;
        double lw = 0.0;
        bool listed = false;
        if (b < a.B)
        {
            const double theta = pth + a.angles[b] + a.srot;
            double s, c;
            dm_sincos(theta, &s, &c);
            slam_ray_result r;
            if (VARIANT == 0)
            {
                const unsigned plane_off = (unsigned)slam_octant(c, s) * (unsigned)a.map_stride;
                const unsigned e0 = origin_in ? a.grid[off0 + plane_off] : 0u;
                ray_fast_L0(a.grid, plane_off, a.cols + 4, a.rows, a.cols, &PX, &PY, c, s, a.zmax2, a.k_safe,
                            a.clamp_limit ? slam_sample_limit(a.z[b], a.stddev) : SLAM_NO_SAMPLE_LIMIT, i0, j0, e0, &r, nullptr);
                lw = slam_beam_d2(&r, sx, sy);
                listed = !slam_beam_clamped(lw, a.z[b], a.zmax, a.clamp4);
            }
            else
            {
                ray_walk<false, true>(a.grid, a.rows, a.cols, sx, sy, c, s, a.z2[b], a.zmax2, nullptr, nullptr,
                                      a.max_iter, &r, &st);
                // (the literal kernel lists every beam with its final value: kwl_flush<false> only adds)
                lw = slam_beam_logw(&r, sx, sy, a.z[b], a.zmax, a.stddev, a.clamp4, a.norm, a.log_eps, 0.0);
                listed = true;
            }
            if (a.hit_xy)
            {
                const size_t o = (size_t)p * a.B + b;
                a.hit_xy[2 * o] = r.hx;
                a.hit_xy[2 * o + 1] = r.hy;
                a.hit_cell[o] = r.hcell;
            }
        }
        const uint64_t m = __ballot(listed);
        if (listed)
        {
            const int k = n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            val[k] = lw;
            beam[k] = b;
        }
        n_list += __builtin_popcountll(m);
    }
    if (lane == 0) a.w[p] = a.log_domain ? acc : dm_exp(acc);
    if (VARIANT == 1 && a.stats)
    {
        atomicAdd(&a.stats[0], st.cells_read);
        atomicAdd(&a.stats[1], st.cells_written);
        atomicAdd(&a.stats[2], st.steps);
    }
}

// K2a, particle-lane form (large N).  Lane = particle, so the 64 rays a wavefront walks together
// share the beam index: they are near-parallel rays from nearby origins (after a resample+sort,
// often copies of one particle), which keeps the data-dependent trip count of ray_fast_L uniform
// across lanes and makes the scan entries wave-uniform (scalar loads).  A work item is
// (64-particle block, beam segment); the per-beam log-likelihoods go to a scratch slab
// lw[block][beam][lane] (coalesced 512-byte stores) and k_sum_lw adds them per particle in beam
// order, reproducing the sequential rounding of slam/mcl.cpp:69-74.
// DIRECT (launches with one segment, i.e. enough particle blocks to fill the chip without cutting the beams): the lane
// adds its particle's log-likelihoods itself, in beam order, and stores the weight: no slab, no k_sum_lw.
// RECORD: also store every ray's hit position and cell (tests and tools; a variant of its own so that the production
// kernels carry neither the branch nor the registers of the store addresses).
template <int MIN_WAVES, bool DIRECT = false, bool RECORD = false>
__global__ void __launch_bounds__(256, MIN_WAVES) k_rays_pl(WeightArgs a)
{
    const int lane = threadIdx.x & 63;
    // (wave-uniform, and told so: beam index, scan entries and loop control then live in scalar registers)
    // (workgroups go to the 8 XCDs round-robin.  Dealing each XCD a contiguous eighth of the segment-major items instead,
    // so that one XCD walks one span of beam angles = one or two clearance planes, was measured: 0.79 vs 0.68 ms, the
    // spans differ too much in work and the slowest XCD sets the time)
    const int64_t item = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // segment-major: the four wavefronts of a workgroup walk the same beams for neighbouring
    // (pose-sorted) particle blocks, i.e. the same map cells at about the same time
    const int seg = (int)(item / a.npblocks);
    if (seg >= a.nseg) return;  // whole wavefront
    const int64_t pb = item - (int64_t)seg * a.npblocks;
    int64_t p = a.p_first + pb * 64 + lane;
    const bool real = p < a.n;
    if (p >= a.n) p = a.n - 1;  // padding lanes repeat the last particle; their values are never summed
    if (a.order) p = a.order[p];
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    // start cell (the same for every beam of the particle); its clearance is read per beam from the plane of the beam's octant
    const int i0 = slam_f2i(((double)a.rows - sy) - 1.0), j0 = slam_f2i(sx);
    const bool inb0 = slam_in_bounds(a.rows, a.cols, i0, j0);
    const unsigned off0 = inb0 ? (unsigned)(slam_mul24(i0, a.cols + 4) + j0 + 2 * (a.cols + 4) + 2) : 0u;  // (frame corner: 0)
    // what the exact-jump set-up needs of the start position, once per particle instead of once per ray
    slam_axis_pre PX, PY;
    slam_axis_prepare(&PX, sx);
    slam_axis_prepare(&PY, sy);
    double* slab = DIRECT ? nullptr : a.lw + (size_t)pb * (size_t)a.B * 64 + lane;
    double acc = 0.0;
    const int b0 = seg * a.seg_beams;
    const int b1 = (b0 + a.seg_beams) < a.B ? (b0 + a.seg_beams) : a.B;
    // the scan entries are the same for every lane and are not written during the launch: read through the scalar cache
    // (constant address space), which frees the vector memory pipeline and four registers per lane
    typedef const double __attribute__((address_space(4))) * scan_ptr;
    const scan_ptr angles_c = (scan_ptr)(uintptr_t)a.angles, z_c = (scan_ptr)(uintptr_t)a.z;
    for (int b = b0; b < b1; ++b)
    {
        const double theta = pth + angles_c[b] + a.srot;
        const double z = z_c[b];
        double s, c;
        dm_sincos(theta, &s, &c);
        slam_ray_result r;
        const unsigned plane_off = (unsigned)slam_octant(c, s) * (unsigned)a.map_stride;
        const int k_lim = a.clamp_limit ? slam_sample_limit(z, a.stddev) : SLAM_NO_SAMPLE_LIMIT;
        ray_fast_L0(a.grid, plane_off, a.cols + 4, a.rows, a.cols, &PX, &PY, c, s, a.zmax2, a.k_safe, k_lim, i0, j0,
                    a.grid[off0 + plane_off], &r, nullptr);
        const double lw = slam_beam_logw(&r, sx, sy, z, a.zmax, a.stddev, a.clamp4, a.norm, a.log_eps, a.inv_stddev);
        if (DIRECT)
            acc += lw;  // slam/mcl.cpp:69-74
        else
            __builtin_nontemporal_store(lw, &slab[(size_t)b * 64]);  // read once, by k_sum_lw: not to stay in the L2 that holds the planes
        if (RECORD)
        {
            const size_t o = (size_t)p * a.B + b;
            a.hit_xy[2 * o] = r.hx;
            a.hit_xy[2 * o + 1] = r.hy;
            a.hit_cell[o] = r.hcell;
        }
    }
    if (DIRECT && real) a.w[p] = a.log_domain ? acc : dm_exp(acc);
}

// The cloud's extent from 256 of its particles (called by the 256 threads of one workgroup): bounding box in cells into
// box[4 * which .. +3], extent of the headings in 1/65536 turns into box[8 + which]; the box also to pinned host memory.
// 256 particles spread evenly over the index range (after a sorted resample the FIRST 256 are the heaviest particles,
// clustered however wide the cloud is; a caller that sets poses in place order has one corner there).
__device__ __forceinline__ void cloud_extent_sample(int64_t n, const double* x, const double* y, const double* th, int* box, int which, int* box_host)
{
    __shared__ int part[4][4];
    __shared__ float hpart[4][4];
    const int64_t p = n > 256 ? (int64_t)(((unsigned long long)threadIdx.x * (unsigned long long)n) >> 8) : (int64_t)threadIdx.x;
    int lx = 0x7fffffff, ly = 0x7fffffff, hx = -0x7fffffff - 1, hy = -0x7fffffff - 1;
    // headings as fractions of a turn, twice: as they are and half a turn on (a cloud across the seam of one is compact in the other)
    float a0 = 2.0f, a1 = -1.0f, b0 = 2.0f, b1 = -1.0f;
    if (p < n)
    {
        lx = hx = slam_f2i(x[p]);
        ly = hy = slam_f2i(y[p]);
        const float t = (float)(th[p] * 0.15915494309189535);
        if (t > -64.0f && t < 64.0f)
        {
            a0 = a1 = t - __builtin_floorf(t);
            b0 = b1 = (t + 0.5f) - __builtin_floorf(t + 0.5f);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        lx = min(lx, __shfl_xor(lx, o));
        ly = min(ly, __shfl_xor(ly, o));
        hx = max(hx, __shfl_xor(hx, o));
        hy = max(hy, __shfl_xor(hy, o));
        a0 = fminf(a0, __shfl_xor(a0, o));
        a1 = fmaxf(a1, __shfl_xor(a1, o));
        b0 = fminf(b0, __shfl_xor(b0, o));
        b1 = fmaxf(b1, __shfl_xor(b1, o));
    }
    if ((threadIdx.x & 63) == 0)
    {
        int* q = part[threadIdx.x >> 6];
        q[0] = lx, q[1] = ly, q[2] = hx, q[3] = hy;
        float* g = hpart[threadIdx.x >> 6];
        g[0] = a0, g[1] = a1, g[2] = b0, g[3] = b1;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int* cur = box + 4 * which;
        cur[0] = min(min(part[0][0], part[1][0]), min(part[2][0], part[3][0]));
        cur[1] = min(min(part[0][1], part[1][1]), min(part[2][1], part[3][1]));
        cur[2] = max(max(part[0][2], part[1][2]), max(part[2][2], part[3][2]));
        cur[3] = max(max(part[0][3], part[1][3]), max(part[2][3], part[3][3]));
        const float ea = fmaxf(fmaxf(hpart[0][1], hpart[1][1]), fmaxf(hpart[2][1], hpart[3][1])) - fminf(fminf(hpart[0][0], hpart[1][0]), fminf(hpart[2][0], hpart[3][0]));
        const float eb = fmaxf(fmaxf(hpart[0][3], hpart[1][3]), fmaxf(hpart[2][3], hpart[3][3])) - fminf(fminf(hpart[0][2], hpart[1][2]), fminf(hpart[2][2], hpart[3][2]));
        const float e = fminf(ea, eb);
        box[8 + which] = e >= 0.0f ? (int)(fminf(e, 1.0f) * 65536.0f) : 65536;  // (no usable heading: a full turn)
        if (box_host) *reinterpret_cast<int4*>(box_host) = make_int4(cur[0], cur[1], cur[2], cur[3]);  // one 16-byte store: never torn
    }
}

// The cloud's extent into box[which] (see k_order_keys): launched before the first
// weighting pass after the poses were set or seeded, so that even that pass is ordered for the cloud it meets.
// box_host: the same four numbers in pinned host memory (a store over PCIe, nobody waits for it): the host reads them
// some passes later to choose the kernel SHAPE for the cloud (do_weights).
__global__ void __launch_bounds__(256) k_order_box(int64_t n, const double* x, const double* y, const double* th, int* box, int which, int* box_host)
{
    cloud_extent_sample(n, x, y, th, box, which, box_host);
}

// Pose-coherent order for k_rays_pl.  The weight of a particle does not depend on which lane
// computes it, so the particles are dealt to wavefronts by a key of (heading bucket, 4-cell tile):
// the 64 rays a wavefront walks together then start from the same few cells with headings within
// 1/1024 turn, take nearly the same number of lookups and touch the same cache lines.  Keys wrap
// (11 bits per axis): far-apart particles that alias only cost coherence, never correctness.
__global__ void __launch_bounds__(256) k_order_keys(int64_t n, const double* x, const double* y, const double* th,
                                                    int mode, int key_bits, uint32_t* keys, int32_t* ident,
                                                    unsigned long long* stats, const double* scan_src, double* scan_dst,
                                                    int scan_n, int* box, int pass_no, int wide_mode, int* box_host, float scan_range)
{
    // side job of the last workgroup: the new scan from its pinned host slot to the device (read by the kernels that
    // follow on the stream; saves the copy kernel and its launch gap)
    // (eight independent loads per thread in flight: the reads cross PCIe, one round trip instead of eight)
    if (blockIdx.x == gridDim.x - 1)
        for (int base = 0; base < scan_n; base += 8 * (int)blockDim.x)
        {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
                v[u] = i < scan_n ? __builtin_nontemporal_load(scan_src + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
                if (i < scan_n) scan_dst[i] = v[u];
            }
        }
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // The cloud's extent chooses the order (wide_mode != 0): `box` holds two bounding boxes {xmin, ymin, xmax, ymax} in
    // cells; this pass reads the one the pass before it wrote and workgroup 0 writes the other from 256 particles spread over the index
    // range (a sample; no atomics: 1563 wavefronts updating
    // one cache line cost 110 us; no launch, no host round trip; the choice lags the cloud by one step, which an
    // ordering heuristic may).  Heading first is best while the part of the planes the cloud can reach stays in the
    // caches; a cloud spread over more than 2^22 cells (eight planes: 32 MB = all L2) is ordered by place first (Morton
    // order of 64-cell tiles, the heading inside a tile: the wavefronts in flight then work on one part of the map, and a
    // tile holds enough particles for the heading to order them: 2 M uniform particles on the maze 40.2 -> 23.6 ms).
    if (wide_mode)
    {
        const int* prev = box + 4 * ((pass_no + 1) & 1);
        const int bx0 = prev[0], by0 = prev[1], bx1 = prev[2], by1 = prev[3];
        if (bx1 >= bx0 && by1 >= by0 && ((double)bx1 - bx0 + 1.0) * ((double)by1 - by0 + 1.0) > 4194304.0) mode = wide_mode;
        if (blockIdx.x == 0) cloud_extent_sample(n, x, y, th, box, pass_no & 1, box_host);
    }
    if (p >= n) return;
    if (p < 3) stats[p] = 0;  // the pass's counters start at zero (saves a memset launch)
    // mode = kind | (tile_shift + 1) << 4 | heading_bits << 8 (0 fields: 4-cell tiles, 10 heading bits:
    // best for a converged cloud, 4 % over 2-cell tiles / 8 bits; a wide cloud prefers coarser headings)
    const int kind = mode & 15;
    const int tshift = ((mode >> 4) & 15) ? ((mode >> 4) & 15) - 1 : 2;
    int tbits = ((mode >> 8) & 15) ? ((mode >> 8) & 15) : 10;
    // Heading buckets as wide as the cloud wants them (kind 3 with no bucket count given, scan_range > 0; round 3).  The 64
    // particles of a wavefront should differ as little in heading as in place: with buckets d (radians) wide, 64 consecutive
    // particles of a bucket lie within L = sqrt(64 A H / (n d)) cells (A: the cloud's area, H: its heading extent), and their
    // rays end d R and L apart at the scan's mean range R; equal at d^3 = 64 A H / (n R^2).  That gives the 10 bits a
    // converged cloud was tuned to, 6 for the sigma-25 cloud of the bench (1.00 -> 0.87 ms; 10 bits there, 6 bits on the
    // converged cloud: +13 % and +10 %), 8-9 for the Gaussian cloud of the 8192^2 shard.  From the extent the pass before left.
    if (kind == 3 && ((mode >> 8) & 15) == 0 && scan_range > 0.0f && wide_mode)
    {
        const int w = (pass_no + 1) & 1;
        const int* prev = box + 4 * w;
        if (prev[2] >= prev[0] && prev[3] >= prev[1])
        {
            const float A = ((float)prev[2] - (float)prev[0] + 1.0f) * ((float)prev[3] - (float)prev[1] + 1.0f);
            const float H = fmaxf((float)box[8 + w], 1.0f) * (1.0f / 65536.0f);  // turns
            const float R = fmaxf(scan_range, 1.0f);
            // d in turns: d^3 = 64 A H / (n R^2 (2 pi)^2)
            const float d3 = 64.0f * A * H / ((float)n * R * R * 39.478418f);
            const float bits = -__builtin_log2f(fmaxf(d3, 1.0e-30f)) * (1.0f / 3.0f);
            const int bi = (int)(bits + 0.5f);
            tbits = bi < 4 ? 4 : (bi > 10 ? 10 : bi);
        }
    }
    const int tshift_y = ((mode >> 12) & 15) ? ((mode >> 12) & 15) - 1 : tshift;  // (tiles may be wider than tall: a cache line runs along x)
    const int nb = (key_bits - tbits) >> 1;  // tile bits per axis; the key has tbits + 2 nb <= key_bits bits
    const unsigned tmask = (1u << nb) - 1u;
    const unsigned tx = ((unsigned)slam_f2i(x[p]) >> tshift) & tmask;
    const unsigned ty = ((unsigned)slam_f2i(y[p]) >> tshift_y) & tmask;
    // heading in turns; 1/(2 pi) to double precision is plenty for a bucket index
    const unsigned tb = (unsigned)slam_f2i((th[p] * 0.15915494309189535 + 4.0) * (double)(1u << tbits)) & ((1u << tbits) - 1u);
    unsigned k;
    if (kind == 2)
        k = (ty << (nb + tbits)) | (tx << tbits) | tb;
    else if (kind == 3 || kind == 4)
    {  // Morton order of the tiles below the heading bucket (3) or above it (4: place first, for clouds wider than the caches)
        unsigned a = tx, b = ty;
        a = (a | (a << 8)) & 0x00ff00ffu;
        a = (a | (a << 4)) & 0x0f0f0f0fu;
        a = (a | (a << 2)) & 0x33333333u;
        a = (a | (a << 1)) & 0x55555555u;
        b = (b | (b << 8)) & 0x00ff00ffu;
        b = (b | (b << 4)) & 0x0f0f0f0fu;
        b = (b | (b << 2)) & 0x33333333u;
        b = (b | (b << 1)) & 0x55555555u;
        k = kind == 3 ? (tb << (2 * nb)) | (a | (b << 1)) : ((a | (b << 1)) << tbits) | tb;
    }
    else
        k = (tb << (2 * nb)) | (ty << nb) | tx;
    keys[p] = k;
    ident[p] = (int32_t)p;
}

// weight = exp(sum_b lw[b]) with the sum taken in beam order (slam/mcl.cpp:67-75).
__global__ void __launch_bounds__(256) k_sum_lw(WeightArgs a)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // particle within this launch
    int64_t p = a.p_first + q;
    if (q >= a.npblocks * 64 || p >= a.n) return;
    if (a.order) p = a.order[p];
    const double* slab = a.lw + (size_t)(q >> 6) * (size_t)a.B * 64 + (q & 63);
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= a.B; b += 8)
    {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&slab[(size_t)(b + u) * 64]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < a.B; ++b) acc += slab[(size_t)b * 64];
    a.w[p] = a.log_domain ? acc : dm_exp(acc);
}

// K2b, first form: one thread per particle, beams and cells strictly in the reference's order.
__global__ void __launch_bounds__(64) k_weights_S(WeightArgs a)
{
    __shared__ uint8_t luts[512];
    for (int t = threadIdx.x; t < 512; t += blockDim.x) luts[t] = a.luts[t];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    uint8_t* map = a.grid + a.map_stride * (size_t)p;
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    double acc = 0.0;
    slam_ray_stats st = {0, 0, 0};
    for (int b = 0; b < a.B; ++b)
    {
        const double theta = pth + a.angles[b] + a.srot;
        double s, c;
        dm_sincos(theta, &s, &c);
        slam_ray_result r;
        ray_walk<true, true>(map, a.rows, a.cols, sx, sy, c, s, a.z2[b], a.zmax2, luts, luts + 256, a.max_iter, &r, &st);
        acc += slam_beam_logw(&r, sx, sy, a.z[b], a.zmax, a.stddev, a.clamp4, a.norm, a.log_eps);
        if (a.hit_xy)
        {
            const size_t o = (size_t)p * a.B + b;
            a.hit_xy[2 * o] = r.hx;
            a.hit_xy[2 * o + 1] = r.hy;
            a.hit_cell[o] = r.hcell;
        }
    }
    a.w[p] = a.log_domain ? acc : dm_exp(acc);
    if (a.stats)
    {
        atomicAdd(&a.stats[0], st.cells_read);
        atomicAdd(&a.stats[1], st.cells_written);
        atomicAdd(&a.stats[2], st.steps);
    }
}

// K2b, wave form (small and medium N).  One wavefront per particle; beams strictly in order
// (slam/mcl.cpp:69-74: later beams see earlier beams' writes); inside one ray the 64 lanes take 64
// consecutive samples.  Sample positions are exact (slam_advance from a wave-uniform base), the
// loop-carried decisions of slam/raycast.cpp:163-220 (first hit, first cell at or beyond z, stop at
// zmax / leaving the grid / drew && foundHit) are resolved with ballots, and only the cells the
// reference would have processed are written.  Cells of one ray are distinct (coordinates move
// monotonically), so reading every lane's cell before any write sees exactly what the sequential
// loop sees.
struct WaveRay
{
    double hx, hy;
    int hcell;
    unsigned n_read, n_written, n_steps;
};

__device__ __forceinline__ uint64_t lanes_upto(int l) { return l >= 63 ? ~0ull : ((1ull << (l + 1)) - 1ull); }

// Storage of one particle's canvas behind ray_walk_wave.  DenseMap: rows*cols bytes (slam::Particle::map
// as the reference holds it).  TiledMap: copy-on-write 64x64 tiles out of a shared pool (SURVEY.md §8f
// row 3): the table of a particle holds one tile id per 64x64 block; bit TILE_OWN says the tile is
// referenced by this table alone and may be written in place, any other tile (tile 0 = the blank
// canvas of 128 included) is copied to a fresh tile before its first write.  Reference counts are
// not kept while the kernel runs: the resample step rebuilds them from the tables (k_tile_*).
enum { TILE_SHIFT = 6, TILE_MASK = 63, TILE_BYTES = 4096, TILE_OWN = 0x40000000 };

// Cells inside a tile are stored as 8x8 micro-blocks of 64 contiguous bytes: one memory sector then
// holds an 8x8 patch of the canvas and a ray crosses about one sector per 8 cells whatever its
// direction (row-major rows would give a steep ray a new sector for every cell).
__device__ __forceinline__ int tile_cell(int i, int j)
{
    return (((i & TILE_MASK) >> 3) << 9) | (((j & TILE_MASK) >> 3) << 6) | ((i & 7) << 3) | (j & 7);
}

struct DenseMap
{
    uint8_t* base;
    int cols;
    __device__ __forceinline__ uint8_t load(int i, int j) const { return base[(size_t)i * (size_t)cols + (size_t)j]; }
    // called by every lane of the wavefront; lanes with wr store
    __device__ __forceinline__ void store(bool wr, int i, int j, uint8_t v, int)
    {
        if (wr) base[(size_t)i * (size_t)cols + (size_t)j] = v;
    }
};

struct TileArgs
{
    uint8_t* pool;            // cap tiles of TILE_BYTES
    int32_t* tab;             // n particles x tiles_per_map entries
    int tiles_per_map, tcols; // tiles per canvas, tiles per canvas row
    const int32_t* freelist;  // ids of the free tiles (as of the last collection)
    int* ctrl;                // [0] allocation cursor into freelist, [1] number of free tiles, [2] error flag
    int lds_entries;          // > 0: the kernel was launched with 4 * lds_entries * 4 bytes of dynamic LDS for the tables
    // copies fused into k_weights_S2<true> (null: k_tile_copy has made them): particle p copies the tiles of table
    // slots plan_t[pos], pos in [plan_rng[2p], plan_rng[2p] + plan_rng[2p + 1]), into the free tiles freelist[pos]
    const int32_t* plan_rng;
    const int32_t* plan_t;
};

struct TiledMap
{
    uint8_t* pool;
    int32_t* tab;   // this particle's table in HBM
    int32_t* ltab;  // the same table staged in LDS for the lookups (or == tab when it does not fit)
    int tcols;
    const int32_t* freelist;
    int* ctrl;
    __device__ __forceinline__ uint8_t load(int i, int j) const
    {
        const int id = ltab[(i >> TILE_SHIFT) * tcols + (j >> TILE_SHIFT)] & ~TILE_OWN;
        return pool[(size_t)id * TILE_BYTES + (size_t)tile_cell(i, j)];
    }
    __device__ __forceinline__ void store(bool wr, int i, int j, uint8_t v, int lane)
    {
        const int t = wr ? (i >> TILE_SHIFT) * tcols + (j >> TILE_SHIFT) : -1;
        int e = wr ? ltab[t] : TILE_OWN;
        bool need = wr && !(e & TILE_OWN);
        uint64_t m = __ballot(need);
        const bool any_cow = m != 0;
        while (m)
        {
            // (rare once k_tile_prepare has run) one shared tile at a time: allocate, copy with the
            // whole wavefront, repoint the table
            const int l = __builtin_ctzll(m);
            const int tl = __shfl(t, l, 64);
            const int old = __shfl(e, l, 64) & ~TILE_OWN;
            int nid = -1;
            if (lane == 0)
            {
                const int pos = atomicAdd(&ctrl[0], 1);
                if (pos < ctrl[1])
                    nid = freelist[pos];
                else
                    ctrl[2] = 1;  // pool exhausted: the write is dropped, the host reports SLAM_ERR_NOMEM
            }
            nid = __builtin_amdgcn_readfirstlane(nid);
            if (nid >= 0)
            {
                const uint4* from = reinterpret_cast<const uint4*>(pool + (size_t)old * TILE_BYTES);
                uint4* to = reinterpret_cast<uint4*>(pool + (size_t)nid * TILE_BYTES);
#pragma unroll
                for (int q = 0; q < TILE_BYTES / 16 / 64; ++q) to[q * 64 + lane] = from[q * 64 + lane];
                if (lane == 0)
                {
                    tab[tl] = nid | TILE_OWN;
                    ltab[tl] = nid | TILE_OWN;
                }
            }
            if (t == tl)
            {
                if (nid >= 0)
                    e = nid | TILE_OWN;
                else
                    wr = false;
                need = false;
            }
            m = __ballot(need);
        }
        if (any_cow)
        {  // the copies (and the table) before the cell stores and the next pass's loads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (wr)
            pool[(size_t)(e & ~TILE_OWN) * TILE_BYTES + (size_t)tile_cell(i, j)] = v;
    }
};

// One pass = 64 consecutive samples of a ray: positions, cells and the cells' bytes (load issued).
struct WavePass
{
    slam_axis X, Y;
    int i, j;
    int last_exist;
    bool isnew, inb;
    uint8_t c;
};

template <class MAP>
__device__ __forceinline__ void wave_pass_prepare(const MAP& map, int rows, int cols, const slam_axis& Xb, const slam_axis& Yb,
                                                  int pi, int pj, double dx, double dy, int k0, int max_iter, int lane,
                                                  WavePass* ps)
{
    ps->X = Xb;
    ps->Y = Yb;
    slam_advance(&ps->X, &ps->Y, dx, dy, lane + 1);  // sample k0 + lane + 1, exact
    slam_cell_of(rows, ps->X.v, ps->Y.v, &ps->i, &ps->j);
    int qi = lane_below_i32(ps->i), qj = lane_below_i32(ps->j);
    if (lane == 0)
    {
        qi = pi;
        qj = pj;
    }
    ps->last_exist = (max_iter - k0 - 1) < 63 ? (max_iter - k0 - 1) : 63;
    ps->isnew = (lane <= ps->last_exist) && !(ps->i == qi && ps->j == qj);  // slam/raycast.cpp:171-174
    ps->inb = slam_in_bounds(rows, cols, ps->i, ps->j);
    // unconditional load (cell (0,0) stands in for samples that have no cell to read): a load under a
    // divergent branch would have to be waited for at the join, which would serialise the two passes
    const bool rd = ps->isnew && ps->inb;
    ps->c = map.load(rd ? ps->i : 0, rd ? ps->j : 0);
}

// The cells of the NEXT pass are read before the current pass is resolved (its loads overlap the
// current pass's latency): a ray's cells are distinct, so those bytes cannot be changed by the
// current pass's writes, and if the ray ends in the current pass they are simply not used.
template <class MAP>
__device__ __forceinline__ void ray_walk_wave(MAP map, int rows, int cols, double sx, double sy, double dx, double dy, double z2,
                              double zmax2, const uint8_t* lut_free, const uint8_t* lut_occ, int max_iter, int lane,
                              WaveRay* out)
{
    slam_axis Xb = {sx, 0.0, 0}, Yb = {sy, 0.0, 0};
    slam_axis_setup(&Xb, dx);
    slam_axis_setup(&Yb, dy);
    int pi, pj;
    slam_cell_of(rows, sx, sy, &pi, &pj);
    bool drew = false, found = false;
    out->hx = -1.0;
    out->hy = -1.0;
    out->hcell = -1;
    out->n_read = out->n_written = 0;
    out->n_steps = (unsigned)max_iter;
    const bool draw_occ = (z2 != zmax2);  // slam/raycast.cpp:205
    if (max_iter <= 0) return;
    WavePass cur;
    wave_pass_prepare(map, rows, cols, Xb, Yb, pi, pj, dx, dy, 0, max_iter, lane, &cur);
    for (int k0 = 0;; k0 += 64)
    {
        const bool more = cur.last_exist == 63 && k0 + 64 < max_iter;
        WavePass nxt = cur;
        if (more)
        {
            Xb.v = bcast_f64(cur.X.v, 63);
            Xb.inc = bcast_f64(cur.X.inc, 63);
            Xb.rem = bcast_i32(cur.X.rem, 63);
            Yb.v = bcast_f64(cur.Y.v, 63);
            Yb.inc = bcast_f64(cur.Y.inc, 63);
            Yb.rem = bcast_i32(cur.Y.rem, 63);
            wave_pass_prepare(map, rows, cols, Xb, Yb, bcast_i32(cur.i, 63), bcast_i32(cur.j, 63), dx, dy, k0 + 64,
                              max_iter, lane, &nxt);
        }
        const int i = cur.i, j = cur.j, last_exist = cur.last_exist;
        const bool isnew = cur.isnew, inb = cur.inb;
        const uint8_t c = cur.c;
        const double ex = cur.X.v - sx, ey = cur.Y.v - sy;
        const double d2 = ex * ex + ey * ey;
        const bool far = !(d2 < z2);
        const uint64_t m_new = __ballot(isnew);
        const uint64_t m_oob = __ballot(isnew && !inb);
        const uint64_t m_occ = __ballot(isnew && inb && c < 128);
        const uint64_t m_far = __ballot(isnew && far);
        const uint64_t m_vfar = __ballot(isnew && d2 >= zmax2);
        const int f_oob = m_oob ? __builtin_ctzll(m_oob) : 64;                  // break before processing (:176)
        const int f_hit = found ? -1 : (m_occ ? __builtin_ctzll(m_occ) : 64);   // :183-189
        const int f_far = drew ? -1 : (m_far ? __builtin_ctzll(m_far) : 64);    // :203-214
        const int f_vfar = m_vfar ? __builtin_ctzll(m_vfar) : 64;               // break after processing (:216)
        const int both = f_hit > f_far ? f_hit : f_far;                         // loop condition fails after this lane (:163)
        const int stop_incl = both < f_vfar ? both : f_vfar;
        const int term = stop_incl < f_oob ? stop_incl : f_oob;                 // lane whose step ends the ray
        int lim = stop_incl < (f_oob - 1) ? stop_incl : (f_oob - 1);            // last processed lane
        if (lim > last_exist) lim = last_exist;
        if (!found && f_hit <= lim)
        {
            out->hx = bcast_f64(cur.X.v, f_hit);
            out->hy = bcast_f64(cur.Y.v, f_hit);
            out->hcell = bcast_i32(i, f_hit) * cols + bcast_i32(j, f_hit);
            found = true;
        }
        const bool proc = isnew && lane <= lim;
        const bool wr = proc && (!far || (lane == f_far && draw_occ));
        map.store(wr, i, j, far ? lut_occ[c] : lut_free[c], lane);
        if (!drew && f_far <= lim) drew = true;
        if (lim >= 0)
        {
            out->n_read += (unsigned)__builtin_popcountll(m_new & lanes_upto(lim));
            out->n_written += (unsigned)__builtin_popcountll(__ballot(wr));
        }
        if (term <= last_exist)
        {
            out->n_steps = (unsigned)(k0 + term + 1);
            return;
        }
        if (!more) return;  // max_iter reached (the reference would spin here)
        cur = nxt;
    }
}

template <bool TILED>
__global__ void __launch_bounds__(256) k_weights_S_wave(WeightArgs a, TileArgs ta)
{
    __shared__ uint8_t luts[512];
    extern __shared__ int32_t lds_tabs[];
    for (int t = threadIdx.x; t < 512; t += blockDim.x) luts[t] = a.luts[t];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= a.n) return;  // whole wavefront
    DenseMap dmap = {a.grid + a.map_stride * (size_t)p, a.cols};
    TiledMap tmap = {ta.pool, nullptr, nullptr, ta.tcols, ta.freelist, ta.ctrl};
    if (TILED)
    {
        tmap.tab = ta.tab + (size_t)p * ta.tiles_per_map;
        tmap.ltab = tmap.tab;
        if (ta.lds_entries > 0)
        {  // this wavefront's table, staged in LDS (only this wavefront reads or writes it)
            tmap.ltab = lds_tabs + (threadIdx.x >> 6) * ta.lds_entries;
            for (int t = lane; t < ta.tiles_per_map; t += 64) tmap.ltab[t] = tmap.tab[t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    double acc = 0.0;
    unsigned long long n_read = 0, n_written = 0, n_steps = 0;
    for (int b0 = 0; b0 < a.B; b0 += 64)
    {
        // directions of up to 64 beams at once (lane = beam) ...
        const int b = b0 + lane;
        double mydx = 0.0, mydy = 0.0, myz2 = 0.0;
        if (b < a.B)
        {
            dm_sincos(pth + a.angles[b] + a.srot, &mydy, &mydx);
            myz2 = a.z2[b];
        }
        slam_ray_result mine = {-1.0, -1.0, -1};
        const int cnt = (a.B - b0) < 64 ? (a.B - b0) : 64;
        // ... then the rays one after the other, 64 samples of one ray per pass
        for (int q = 0; q < cnt; ++q)
        {
            WaveRay wr;
            if (TILED)
                ray_walk_wave(tmap, a.rows, a.cols, sx, sy, bcast_f64(mydx, q), bcast_f64(mydy, q), bcast_f64(myz2, q),
                              a.zmax2, luts, luts + 256, a.max_iter, lane, &wr);
            else
                ray_walk_wave(dmap, a.rows, a.cols, sx, sy, bcast_f64(mydx, q), bcast_f64(mydy, q), bcast_f64(myz2, q),
                              a.zmax2, luts, luts + 256, a.max_iter, lane, &wr);
            if (lane == q)
            {
                mine.hx = wr.hx;
                mine.hy = wr.hy;
                mine.hcell = wr.hcell;
            }
            n_read += wr.n_read;
            n_written += wr.n_written;
            n_steps += wr.n_steps;
            // the next ray of this wavefront must see these stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // likelihoods of the batch (lane = beam), added in beam order
        double lw = 0.0;
        if (b < a.B)
        {
            lw = slam_beam_logw(&mine, sx, sy, a.z[b], a.zmax, a.stddev, a.clamp4, a.norm, a.log_eps);
            if (a.hit_xy)
            {
                const size_t o = (size_t)p * a.B + b;
                a.hit_xy[2 * o] = mine.hx;
                a.hit_xy[2 * o + 1] = mine.hy;
                a.hit_cell[o] = mine.hcell;
            }
        }
        for (int l = 0; l < cnt; ++l) acc += readlane_f64(lw, l);
    }
    if (lane == 0)
    {
        a.w[p] = a.log_domain ? acc : dm_exp(acc);
        if (a.stats)
        {
            atomicAdd(&a.stats[0], n_read);
            atomicAdd(&a.stats[1], n_written);
            atomicAdd(&a.stats[2], n_steps);
        }
    }
}

// Particle state is stored packed, [x | y | theta | weight], n doubles each.  A "gathered" buffer
// holds the packed states of all shards back to back ([shard][4][n_loc]), which is exactly what an
// all-gather of the per-shard buffers produces.  Split form (wbase != nullptr): the poses and the
// weights were gathered separately, base = [shard][3][n_loc], wbase = [shard][n_loc].
struct StateView
{
    const double* base;
    int64_t n_loc;  // particles per shard
    const double* wbase;
};
__device__ __forceinline__ double sv_get(const StateView& v, int q, int64_t i)
{
    if (v.wbase && q == 3) return v.wbase[i];
    const int64_t r = i / v.n_loc;
    const int64_t o = i - r * v.n_loc;
    return v.base[(r * (v.wbase ? 3 : 4) + q) * v.n_loc + o];
}

// SLAM_WEIGHT_LOG: maximum of the weight column of all shards (max is exact, so any reduction order
// gives the same value), then wlin[i] = exp(lw[i] - max).
__global__ void __launch_bounds__(256) k_max_partials(const double* a, StateView view, int64_t n, double* part)
{
    __shared__ double sh[256];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // -inf for the padding: every real value is finite or -inf itself
    double v = i < n ? (a ? a[i] : sv_get(view, 3, i)) : -__builtin_inf();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d)
        {
            const double o = sh[threadIdx.x + d];
            if (o > sh[threadIdx.x]) sh[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(256) k_exp_shift(StateView view, int64_t n, const double* max1, double* wlin)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    wlin[i] = dm_exp(sv_get(view, 3, i) - max1[0]);
}

// ------------------------------------------------------------------------------------------------
// K3.  Canonical blocked running sum: radix-64 hierarchy of Hillis-Steele wavefront scans.
// s[i] = inclusive scan inside the 64-chunk of i; tot[c] = last element of chunk c.
// Input: a[i], or (a == nullptr) the weight column of a gathered state view.
__global__ void __launch_bounds__(256) k_scan_chunks(const double* a, StateView view, int64_t n, double* s, double* tot)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t i = c * 64 + lane;
    if (c * 64 >= n) return;
    double v = i < n ? (a ? a[i] : sv_get(view, 3, i)) : 0.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const double t = __shfl_up(v, d, 64);
        if (lane >= d) v = t + v;
    }
    if (i < n) s[i] = v;
    const int64_t last = (c * 64 + 63 < n - 1) ? c * 64 + 63 : n - 1;
    if (i == last) tot[c] = v;
}

// P[j] = (inclusive running sum of all earlier chunks) + s[j]
__global__ void __launch_bounds__(256) k_downsweep(const double* s, const double* Pnext, int64_t n, double* P)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t c = j >> 6;
    P[j] = c ? Pnext[c - 1] + s[j] : s[j];
}

// Levels 1 .. L-1 of the hierarchy (everything above the 64-chunks of the weights themselves) are small:
// one workgroup runs their scans and downsweeps back to back instead of one launch each.  Same
// operations in the same order as k_scan_chunks / k_downsweep.
struct UpperLevels
{
    int L;  // levels in total (<= 8); level 0 is handled by the caller
    int64_t n[8];
    double *s[8], *tot[8], *P[8];
};
__global__ void __launch_bounds__(1024) k_scan_upper(UpperLevels u)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int l = 1; l < u.L; ++l)
    {
        const double* a = u.tot[l - 1];
        const int64_t n = u.n[l];
        for (int64_t c = wave; c * 64 < n; c += 16)
        {
            const int64_t i = c * 64 + lane;
            double v = i < n ? a[i] : 0.0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
            {
                const double t = __shfl_up(v, d, 64);
                if (lane >= d) v = t + v;
            }
            if (i < n) u.s[l][i] = v;
            const int64_t last = (c * 64 + 63 < n - 1) ? c * 64 + 63 : n - 1;
            if (i == last) u.tot[l][c] = v;
        }
        __syncthreads();
    }
    for (int l = u.L - 2; l >= 1; --l)
    {
        for (int64_t j = threadIdx.x; j < u.n[l]; j += 1024)
        {
            const int64_t c = j >> 6;
            u.P[l][j] = c ? u.P[l + 1][c - 1] + u.s[l][j] : u.s[l][j];
        }
        __syncthreads();
    }
}

// The reference's order (slam/mcl.cpp:170-176): strictly sequential, one thread.
__global__ void k_prefix_seq(const double* w, int64_t n, double* P)
{
    if (blockIdx.x || threadIdx.x) return;
    double acc = w[0];
    P[0] = acc;
    for (int64_t i = 1; i < n; ++i)
    {
        acc = acc + w[i];
        P[i] = acc;
    }
}

struct SelectArgs
{
    const double* P;        // running sums over n_total entries
    int64_t n_total;
    int64_t slot_first;     // global index of this shard's first output slot
    int64_t n_slots;
    const double* injected; // n_slots canonical uniforms or nullptr
    uint64_t seed, step;
    int resampler;
    int32_t* idx;           // n_slots selected global indices
    StateView view;         // sort keys wanted (keys != nullptr): weight of the particle selected for slot k,
    double* keys;           // payload k
    int32_t* slots;
    // k_select<true> only: P is not materialised; s0 = scans inside the 64-chunks of the weights, tot0 = the n1 chunk totals
    const double *s0, *tot0;
    int n1;
};
enum { SELECT_FUSED_MAX_N1 = 4096 };

// slam/mcl.cpp:179-203: cutoff = canonical * sum; bisection in which equality returns mid.
// FUSED (three-level hierarchies, 4096 < n_total <= 262144): the running sums are never written out.  Every workgroup
// rebuilds the upper levels from the chunk totals in LDS (the operations of k_scan_upper, in its order: at most 4096
// totals, a microsecond), and a probe forms P[mid] = P1[mid / 64 - 1] + s0[mid] as k_downsweep would have: the same
// doubles, two launches fewer per step (k_scan_upper, k_downsweep).  slam_mcl_get_prefix completes the array on demand.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_select(SelectArgs a)
{
    __shared__ double P1[FUSED ? SELECT_FUSED_MAX_N1 : 1];
    __shared__ double tot1[FUSED ? 64 : 1];
    if (FUSED)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int n1 = a.n1;
        // (four chunks of a wavefront side by side: the scan of one chunk is a chain of six cross-lane steps)
        for (int c0 = wave * 4; c0 * 64 < n1; c0 += 16)
        {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const int i = (c0 + u) * 64 + lane;
                v[u] = i < n1 ? a.tot0[i] : 0.0;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const double t = __shfl_up(v[u], d, 64);
                    if (lane >= d) v[u] = t + v[u];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const int c = c0 + u, i = c * 64 + lane;
                if (i < n1) P1[i] = v[u];  // s1 for now
                const int last = (c * 64 + 63 < n1 - 1) ? c * 64 + 63 : n1 - 1;
                if (i == last) tot1[c] = v[u];
            }
        }
        __syncthreads();
        const int n2 = (n1 + 63) >> 6;  // <= 64: one chunk, its scan is P2
        double v2 = 0.0;
        if (wave == 0)
        {
            v2 = lane < n2 ? tot1[lane] : 0.0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
            {
                const double t = __shfl_up(v2, d, 64);
                if (lane >= d) v2 = t + v2;
            }
        }
        __syncthreads();
        if (wave == 0 && lane < n2) tot1[lane] = v2;
        __syncthreads();
        for (int j = threadIdx.x; j < n1; j += 256)
        {
            const int c = j >> 6;
            if (c) P1[j] = tot1[c - 1] + P1[j];
        }
        __syncthreads();
    }
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_slots) return;
    auto P = [&](int m) -> double {
        if (!FUSED) return a.P[m];
        const int c = m >> 6;
        return c ? P1[c - 1] + a.s0[m] : a.s0[m];
    };
    const double sum = P((int)a.n_total - 1);
    double cutoff;
    if (a.resampler == 0)
    {
        const double canon = a.injected ? a.injected[k] : slam_resample_uniform(a.seed, a.step, (uint64_t)(a.slot_first + k));
        cutoff = canon * sum;
    }
    else
    {
        const double canon = a.injected ? a.injected[0] : slam_resample_uniform(a.seed, a.step, 0);
        cutoff = ((double)(a.slot_first + k) + canon) * (sum / (double)a.n_total);
    }
    int i = 0;
    int j = (int)a.n_total - 1;
    while (i < j)
    {
        const int mid = i + (j - i) / 2;
        const double c = P(mid);
        if (c == cutoff)
        {
            i = mid;
            j = mid;
        }
        else if (c < cutoff)
            i = mid + 1;
        else
            j = mid;
    }
    a.idx[k] = i;
    if (a.keys)
    {
        a.keys[k] = sv_get(a.view, 3, i);
        a.slots[k] = (int32_t)k;
    }
}

// src[p] = idx[perm[p]] (perm == nullptr: identity); new state p = old state src[p]
__global__ void __launch_bounds__(256) k_gather(StateView view, const int32_t* idx, const int32_t* perm, int64_t n,
                                                double* dst /* packed [4][n] */, int32_t* src)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t s = idx[perm ? perm[p] : p];
    const int64_t r = s / view.n_loc;
    const int64_t o = s - r * view.n_loc;
    const double* from = view.base + r * (view.wbase ? 3 : 4) * view.n_loc + o;
    dst[p] = from[0];
    dst[n + p] = from[view.n_loc];
    dst[2 * n + p] = from[2 * view.n_loc];
    dst[3 * n + p] = view.wbase ? view.wbase[s] : from[3 * view.n_loc];
    src[p] = s;
}

// The same for ONE output position, as the epilogue of the weight sort's last kernel (sortk::k_ss_buckets): position p of
// the sorted order holds slot `slot` of the selection.
struct GatherEpilogue
{
    StateView view;
    const int32_t* idx;
    int64_t n;
    double* dst;
    int32_t* src;
    __device__ __forceinline__ void operator()(int p, int32_t slot) const
    {
        const int32_t s = idx[slot];
        const int64_t r = s / view.n_loc;
        const int64_t o = s - r * view.n_loc;
        const double* from = view.base + r * (view.wbase ? 3 : 4) * view.n_loc + o;
        dst[p] = from[0];
        dst[n + p] = from[view.n_loc];
        dst[2 * n + p] = from[2 * view.n_loc];
        dst[3 * n + p] = view.wbase ? view.wbase[s] : from[3 * view.n_loc];
        src[p] = s;
    }
};

// slam/mcl.cpp:147-155: deep copy of each selected particle's map.  grid = (particles, chunks).
// Also packs maps for the sharded exchange (dst slot p <- src map src[p]).
// An index outside [0, n_src) (a caller's device array) is skipped and reported through *bad.
__global__ void __launch_bounds__(256) k_clone_maps(const uint8_t* maps, uint8_t* maps2, const int32_t* src,
                                                    size_t stride, int64_t n_src, int* bad)
{
    const size_t p = blockIdx.x;
    if ((uint64_t)(int64_t)src[p] >= (uint64_t)n_src)
    {
        *bad = 1;
        return;
    }
    const uint4* from = reinterpret_cast<const uint4*>(maps + stride * (size_t)src[p]);
    uint4* to = reinterpret_cast<uint4*>(maps2 + stride * p);
    const size_t n16 = stride / 16;
    for (size_t t = (size_t)blockIdx.y * blockDim.x + threadIdx.x; t < n16; t += (size_t)gridDim.y * blockDim.x)
        store_streaming(&to[t], from[t]);
}

// ---- tiled canvases: table gather at resample, collection of unreferenced tiles, import / export ----
// Resample (slam/mcl.cpp:147-155 without the deep copy): new table p = old table src[p]
// (k_tab_gather; src == nullptr keeps the tables), then an exact, atomic-free census of the tiles:
//   k_tab_gather   every entry stamps its tile with the collection's number (`live`) and with its own
//                  entry index (who[id]; whichever entry wins the race)
//   k_tile_shared  every entry that did not win stamps the tile as shared
//   k_tile_own     an entry owns its tile iff it won and nobody stamped the tile shared
// Plain racing stores only: a reference COUNT needs read-modify-writes on the few hot tiles that
// thousands of tables share (1.6 ms in the first version), and a rule of thumb (ownership survives
// only in a sole child) leaves exclusive tiles unowned, which costs copies and pool space.
// (Letting one of several children inherit the ownership is NOT valid: it would write in place a
// tile its siblings still reference outside their own sensor range.  Letting the LAST sibling that
// leaves a tile keep it IS valid, with a per-tile count of the siblings that have copied it; it was
// built and measured in round 2: the weights of a grid-SLAM filter are so peaked that a family has
// ~20 members, so it saved 4.5 % of the copies, and it was taken out again.)
__global__ void __launch_bounds__(256) k_tab_gather(const int32_t* tab, int32_t* tab2, const int32_t* src, int T, int* ref,
                                                    int* who, int live)
{
    const size_t p = blockIdx.x;
    const int32_t* from = tab + (size_t)(src ? src[p] : (int32_t)p) * T;
    int32_t* to = tab2 + p * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x)
    {
        const int id = from[t] & ~TILE_OWN;
        to[t] = id;
        if (id)
        {
            ref[id] = live;
            who[id] = (int)(p * T + t);
        }
    }
}

// (four entries per thread, one 16-byte load: a thread per entry spent its time launching 700 000 workgroups)
__global__ void __launch_bounds__(256) k_tile_shared(const int32_t* tab, int64_t total, const int* who, int* shared, int live)
{
    const int64_t k0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (k0 >= total) return;
    int id[4] = {0, 0, 0, 0};
    if (k0 + 4 <= total)
    {
        const int4 v = *reinterpret_cast<const int4*>(tab + k0);
        id[0] = v.x, id[1] = v.y, id[2] = v.z, id[3] = v.w;
    }
    else
        for (int u = 0; k0 + u < total; ++u) id[u] = tab[k0 + u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (id[u] && who[id[u]] != (int)(k0 + u)) shared[id[u]] = live;
}

__global__ void __launch_bounds__(256) k_tile_own(int32_t* tab, int64_t total, const int* who, const int* shared, int live)
{
    const int64_t k0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (k0 >= total) return;
    int id[4] = {0, 0, 0, 0};
    if (k0 + 4 <= total)
    {
        const int4 v = *reinterpret_cast<const int4*>(tab + k0);
        id[0] = v.x, id[1] = v.y, id[2] = v.z, id[3] = v.w;
    }
    else
        for (int u = 0; k0 + u < total; ++u) id[u] = tab[k0 + u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (id[u] && who[id[u]] == (int)(k0 + u) && shared[id[u]] != live) tab[k0 + u] = id[u] | TILE_OWN;
}

// free list = every tile but the blank one whose mark is not `live` (order irrelevant).  8192 ids per
// block and ONE atomic per block on the count: it is a single address, and a pool is mostly free.
__global__ void __launch_bounds__(256) k_tile_free(const int* ref, int live, int cap, int32_t* freelist, int* ctrl)
{
    __shared__ int cnt_s, base_s;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    const int id0 = blockIdx.x * 8192 + threadIdx.x;
    uint32_t mask = 0;
    int mine = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q)
    {
        const int id = id0 + q * 256;
        if (id >= 1 && id < cap && ref[id] != live)
        {
            mask |= 1u << q;
            ++mine;
        }
    }
    int at = mine ? atomicAdd(&cnt_s, mine) : 0;
    __syncthreads();
    if (threadIdx.x == 0) base_s = cnt_s ? atomicAdd(&ctrl[1], cnt_s) : 0;
    __syncthreads();
    at += base_s;
    for (int q = 0; q < 32; ++q)
        if (mask & (1u << q)) freelist[at++] = id0 + q * 256;
}

struct SectorTable
{
    float r[256];
};
// out[j * 256 + k] = max of the sectors k .. k + 2^j - 1 (cyclic), j = 0..5: the farthest write over any run of up to
// 64 sectors is then two reads (k_tile_plan looked at up to 44 sectors per tile, one dependent read each)
enum { SECTOR_LEVELS = 6 };
__global__ void __launch_bounds__(256) k_set_sectors(SectorTable t, float* out)
{
    const int k = threadIdx.x;
    float v = t.r[k];
    out[k] = v;
    for (int j = 1; j < SECTOR_LEVELS; ++j)
    {
        __syncthreads();
        v = fmaxf(v, out[(j - 1) * 256 + ((k + (1 << (j - 1))) & 255)]);
        out[j * 256 + k] = v;
    }
}

// Before a weighting pass: every tile a particle may write (within `range` cells of its sensor) that
// it does not own yet is copied to a fresh tile, instead of one at a time inside k_weights_S_wave.
// k_tile_plan lists those tiles and takes one range of the free list per block of 64 particles
// (the allocation cursor is one address).  k_tile_copy copies the listed tiles and repoints the tables.  Conservative in both directions is
// fine: a tile copied needlessly costs 4 KiB, a tile missed is copied by the kernel.
struct TilePlan
{
    int32_t *wl_p, *wl_t;  // work list, indexed like the free list: particle and tile slot (wl_p < 0: nothing to do)
    int32_t* rng;          // per particle: first position and number of its entries (they are consecutive)
};

// `sector_r[k]`, k = 0..255: how far the scan can write in the directions of sector k of the scan frame
// (max over its beams of min(z, zmax) + 2, dilated by one sector): a tile farther away than that in
// all the sectors it spans is not copied (a wall at 100 cells stops the writes there although the
// sensor's range is 500).  Approximate on purpose (float, margins): a miss only moves the copy
// into the walk kernel.
enum { TILE_PLAN_PW = 4 };  // particles per wavefront of k_tile_plan (16 wavefronts per block: 64 particles per atomic)

struct TilePlanCtx
{
    double sc, sr, range, heading;
    bool sane;
    int ti_first, tj_first, side, trows, tcols;
    const int32_t* tab_p;
    const float* sector_r;
    // does the particle need a private copy of the q-th tile of its square?  (*t = its table slot)
    __device__ __forceinline__ bool need(int q, int* t) const
    {
        const int ti = ti_first + q / side, tj = tj_first + q % side;
        *t = ti * tcols + tj;
        if (!(q < side * side && sane && ti >= 0 && tj >= 0 && ti < trows && tj < tcols)) return false;
        const double r0 = (double)(ti << TILE_SHIFT), c0 = (double)(tj << TILE_SHIFT);
        const double dr = sr < r0 ? r0 - sr : (sr > r0 + 64.0 ? sr - (r0 + 64.0) : 0.0);
        const double dc = sc < c0 ? c0 - sc : (sc > c0 + 64.0 ? sc - (c0 + 64.0) : 0.0);
        const double dmin2 = dr * dr + dc * dc;
        if (!(dmin2 <= range * range) || (tab_p[*t] & TILE_OWN)) return false;
        const float dxw = (float)(c0 + 32.0 - sc), dyw = (float)(sr - (r0 + 32.0));  // to the tile's centre, y up
        const float dcen = sqrtf(dxw * dxw + dyw * dyw);
        if (dcen > 100.0f)
        {
            const float inv = 40.743665f;  // 256 / (2 pi)
            const float alpha = atan2f(dyw, dxw) - (float)heading;
            const float half = asinf(46.0f / dcen) * inv + 1.5f;  // half-diagonal of a tile, plus margin, in sectors
            const int k0 = (int)floorf(alpha * inv - half), k1 = (int)floorf(alpha * inv + half);
            const int len = k1 - k0 + 1;  // 4 .. 44 (half <= asin(0.46) * inv + 1.5)
            const int j = len >= 32 ? 5 : 31 - __builtin_clz(len);
            const float rmax = fmaxf(sector_r[j * 256 + ((k0 + 128) & 255)], sector_r[j * 256 + ((k1 - (1 << j) + 1 + 128) & 255)]);
            return dmin2 <= (double)rmax * (double)rmax;
        }
        return true;
    }
};

// One wavefront per TILE_PLAN_PW particles, 16 wavefronts per block; the tiles are evaluated twice (count, then
// list) so that a whole block takes ONE range of the free list: the allocation cursor is a single
// address, and one atomic per particle on it cost more than everything else in this kernel.
// sub_list / sub_count: plan for the particles sub_list[0 .. *sub_count) only (the survivors of a resample whose weighting
// pass left the canvases untouched, see TiledDefer in slam_s2.h); nullptr: all n particles.
__global__ void __launch_bounds__(1024) k_tile_plan(int64_t n, const double* x, const double* y, const double* th, double sd,
                                                    double stheta, double srot, double range, const float* sector_r, int rows,
                                                    int cols, int side, TileArgs ta, TilePlan pl, const int32_t* sub_list,
                                                    const int* sub_count)
{
    if (sub_list) n = *sub_count;
    __shared__ int cnt_s, base_s;
    __shared__ uint64_t votes_s[16][TILE_PLAN_PW][16];  // the votes of the counting pass (side <= 31: at most 16 x 64 tiles)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    const int64_t p_first = ((int64_t)blockIdx.x * 16 + wave) * TILE_PLAN_PW;
    int wave_total = 0;
    int wave_off = 0;
    for (int pass = 0; pass < 2; ++pass)
    {
        int running = 0;
        for (int k = 0; k < TILE_PLAN_PW; ++k)
        {
            if (p_first + k >= n) break;  // whole wavefront
            const int64_t p = sub_list ? (int64_t)sub_list[p_first + k] : p_first + k;
            double s, c;
            dm_sincos(th[p] + stheta, &s, &c);
            TilePlanCtx cx;
            cx.sc = x[p] + c * sd;                           // sensor column
            cx.sr = ((double)rows - (y[p] + s * sd)) - 1.0;  // sensor row
            cx.sane = dm_fabs(cx.sc) < 1.0e9 && dm_fabs(cx.sr) < 1.0e9;
            cx.range = range;
            cx.heading = th[p] + srot;
            cx.side = side;
            cx.ti_first = (slam_f2i(cx.sr) >> TILE_SHIFT) - (side >> 1);
            cx.tj_first = (slam_f2i(cx.sc) >> TILE_SHIFT) - (side >> 1);
            cx.trows = ta.tiles_per_map / ta.tcols;
            cx.tcols = ta.tcols;
            cx.tab_p = ta.tab + (size_t)p * ta.tiles_per_map;
            cx.sector_r = sector_r;
            const int run0 = running;
            for (int q0 = 0; q0 < side * side; q0 += 64)
            {
                int t;
                bool need;
                uint64_t vote;
                if (pass == 0)
                {
                    need = cx.need(q0 + lane, &t);
                    vote = __ballot(need);
                    if (lane == 0) votes_s[wave][k][q0 >> 6] = vote;
                }
                else
                {
                    vote = votes_s[wave][k][q0 >> 6];
                    need = __builtin_amdgcn_inverse_ballot_w64(vote);
                    t = (cx.ti_first + (q0 + lane) / side) * cx.tcols + cx.tj_first + (q0 + lane) % side;
                }
                if (pass == 1 && need)
                {
                    const int pos = base_s + wave_off + running +
                                    (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0));
                    if (pos < ta.ctrl[1])
                    {
                        pl.wl_p[pos] = (int32_t)p;
                        pl.wl_t[pos] = t;
                    }
                    else
                        ta.ctrl[2] = 1;  // pool exhausted
                }
                running += __builtin_popcountll(vote);
            }
            if (pass == 1 && lane == 0)
            {
                pl.rng[2 * p] = base_s + wave_off + run0;
                pl.rng[2 * p + 1] = running - run0;
            }
        }
        if (pass == 0)
        {
            wave_total = running;
            if (lane == 0 && wave_total) wave_off = atomicAdd(&cnt_s, wave_total);
            wave_off = __builtin_amdgcn_readfirstlane(wave_off);
            __syncthreads();
            if (threadIdx.x == 0) base_s = cnt_s ? atomicAdd(&ta.ctrl[0], cnt_s) : 0;
            __syncthreads();
        }
    }
}

// The distinct sources of a selection (slam/mcl.cpp:179-203 picks with replacement): every selected particle is stamped
// with this resample's number, then the stamped particles are listed, one counter atomic per wavefront.  The order of
// the list is whatever the atomics make it: it only decides which free tile a survivor's copy lands in.
__global__ void __launch_bounds__(256) k_mark_sources(const int32_t* idx, int64_t n_slots, int64_t first_local, int64_t n_local, int gen,
                                                      int* stamp)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    const int64_t s = (int64_t)idx[k] - first_local;
    if (s >= 0 && s < n_local) stamp[s] = gen;
}
// (skip: particles whose canvases are up to date already, see k_defer_split)
__global__ void __launch_bounds__(256) k_list_sources(const int* stamp, int64_t n_local, int gen, int32_t* list, int* count, const uint8_t* skip)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool is = p < n_local && stamp[p] == gen && !(skip && skip[p]);
    const uint64_t vote = __ballot(is);
    if (!vote) return;
    int base = 0;
    const int leader = __ffsll((long long)vote) - 1;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(count, __popcll(vote));
    base = __shfl(base, leader);
    if (is) list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0))] = (int32_t)p;
}

// The particles of a deferred pass that cannot defer (slam_defer_safe): flagged, and listed for the ordinary pass that
// follows at once.
__global__ void __launch_bounds__(256) k_defer_split(WeightArgs a, uint8_t* flag, int32_t* list, int* count)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is = false;
    if (p < a.n)
    {
        double s0, c0;
        dm_sincos(a.th[p] + a.stheta, &s0, &c0);
        is = !slam_defer_safe(a.x[p] + c0 * a.sd, a.y[p] + s0 * a.sd, (double)a.rows, a.defer_edge);
        flag[p] = is ? 1 : 0;
    }
    const uint64_t vote = __ballot(is);
    if (!vote) return;
    int base = 0;
    const int leader = __ffsll((long long)vote) - 1;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(count, __popcll(vote));
    base = __shfl(base, leader);
    if (is) list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0))] = (int32_t)p;
}

// first = allocation cursor before k_tile_plan (ctrl[3]); block b takes free-list positions first + b,
// first + b + gridDim.x, ... (the grid is bounded: how many tiles the plan listed is known on the device only,
// and a grid sized for the worst case spends milliseconds launching empty workgroups)
__global__ void __launch_bounds__(256) k_tile_copy(TileArgs ta, TilePlan pl)
{
    const int end = ta.ctrl[0] < ta.ctrl[1] ? ta.ctrl[0] : ta.ctrl[1];
    for (int64_t pos = (int64_t)ta.ctrl[3] + blockIdx.x; pos < end; pos += gridDim.x)
    {
        const int p = pl.wl_p[pos];
        if (p < 0) continue;
        const int t = pl.wl_t[pos];
        int32_t* tab_p = ta.tab + (size_t)p * ta.tiles_per_map;
        const int old = tab_p[t] & ~TILE_OWN;
        const int nid = ta.freelist[pos];
        const uint4* from = reinterpret_cast<const uint4*>(ta.pool + (size_t)old * TILE_BYTES);
        uint4* to = reinterpret_cast<uint4*>(ta.pool + (size_t)nid * TILE_BYTES);
        store_streaming(&to[threadIdx.x], from[threadIdx.x]);
        __syncthreads();  // (the table entry is read above by every thread before thread 0 repoints it)
        if (threadIdx.x == 0)
        {
            tab_p[t] = nid | TILE_OWN;
            pl.wl_p[pos] = -1;
        }
    }
}

__global__ void k_tile_mark(int* ctrl) { ctrl[3] = ctrl[0] < ctrl[1] ? ctrl[0] : ctrl[1]; }

// dense rows x cols canvas of one particle out of its tiles
__global__ void __launch_bounds__(256) k_untile(const uint8_t* pool, const int32_t* tab_p, int tcols, int rows, int cols,
                                                uint8_t* out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (int64_t)rows * cols) return;
    const int i = (int)(k / cols), j = (int)(k - (int64_t)i * cols);
    const int id = tab_p[(i >> TILE_SHIFT) * tcols + (j >> TILE_SHIFT)] & ~TILE_OWN;
    out[k] = pool[(size_t)id * TILE_BYTES + (size_t)tile_cell(i, j)];
}

// Sparse form of the sharded canvas exchange (slam_mcl_pack_tiles / slam_mcl_shard_import_tiles): a
// requested canvas travels as its table, renumbered into the payload, and every distinct tile the
// requested tables name travels once.
// k_pack_mark: one thread per (requested canvas, table entry); the first thread to stamp a tile with
// this pack's number gives it the next payload position (one atomic per wavefront on the counter).
__global__ void __launch_bounds__(256) k_pack_mark(const int32_t* tab, int T, const int32_t* local_idx, int64_t total, int gen,
                                                   int* stamp, int32_t* slot, int32_t* list, int* counter, int64_t n_src, int* bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int id = 0;
    if (k < total)
    {
        const int64_t c = k / T;
        if ((uint64_t)(int64_t)local_idx[c] < (uint64_t)n_src)
            id = tab[(size_t)local_idx[c] * T + (size_t)(k - c * T)] & ~TILE_OWN;
        else
            *bad = 1;  // a canvas that does not exist travels as a blank one
    }
    const bool fresh = id != 0 && atomicExch(&stamp[id], gen) != gen;
    const uint64_t vote = __ballot(fresh);
    if (!vote) return;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0));
    int base = 0;
    const int leader = __ffsll((long long)vote) - 1;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, __popcll(vote));
    base = __shfl(base, leader);
    if (fresh)
    {
        slot[id] = base + lane;
        list[base + lane] = id;
    }
}

__global__ void __launch_bounds__(256) k_pack_tabs(const int32_t* tab, int T, const int32_t* local_idx, int64_t total,
                                                   const int32_t* slot, int32_t* out, int64_t n_src)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int64_t c = k / T;
    const bool ok = (uint64_t)(int64_t)local_idx[c] < (uint64_t)n_src;  // (k_pack_mark has reported it)
    const int id = ok ? tab[(size_t)local_idx[c] * T + (size_t)(k - c * T)] & ~TILE_OWN : 0;
    out[k] = id ? slot[id] + 1 : 0;
}

// one block per payload tile, 16 bytes per thread; ids == nullptr: tile q itself
__global__ void __launch_bounds__(256) k_tiles_copy(const uint8_t* from, const int32_t* from_ids, uint8_t* to, const int32_t* to_ids)
{
    const size_t q = blockIdx.x;
    const uint4* f = reinterpret_cast<const uint4*>(from + (from_ids ? (size_t)from_ids[q] : q) * TILE_BYTES);
    uint4* t = reinterpret_cast<uint4*>(to + (to_ids ? (size_t)to_ids[q] : q) * TILE_BYTES);
    t[threadIdx.x] = f[threadIdx.x];
}

// received tables of one owner: entries count from 1 within that owner's payload; shift them by the position of the
// owner's first tile in the receiver's concatenated payload (0 = the blank tile stays 0)
__global__ void __launch_bounds__(256) k_shift_tabs(int32_t* tabs, int64_t total, int32_t add)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int32_t e = tabs[k];
    if (e > 0) tabs[k] = e + add;
}

// slot p takes row pos[p] of the received tables; entry e > 0 = payload tile e - 1, now pool tile ids[e - 1]
__global__ void __launch_bounds__(256) k_import_tabs(const int32_t* tabs, const int32_t* pos, int T, int64_t total, int n_tiles,
                                                     const int32_t* ids, int32_t* tab, int64_t n_tabs, int* bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int64_t p = k / T;
    const bool row_ok = (uint64_t)(int64_t)pos[p] < (uint64_t)n_tabs;
    const int e = row_ok ? tabs[(size_t)pos[p] * T + (size_t)(k - p * T)] : -1;
    if (e < 0 || e > n_tiles)
    {
        *bad = 1;  // a slot without a table, or a table that names a tile outside the payload
        tab[k] = 0;
        return;
    }
    tab[k] = e ? ids[e - 1] : 0;
}

// Sharded exchange of tiled canvases travels as dense canvases (slam_amd/dist.py): export of the
// requested particles, grid (cells / 256, count) ...
__global__ void __launch_bounds__(256) k_untile_many(const uint8_t* pool, const int32_t* tab, int T, int tcols, int rows, int cols,
                                                     const int32_t* local_idx, size_t stride, uint8_t* out, int64_t n_src, int* bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (int64_t)rows * cols) return;
    if ((uint64_t)(int64_t)local_idx[blockIdx.y] >= (uint64_t)n_src)
    {
        *bad = 1;
        return;
    }
    const int32_t* tab_p = tab + (size_t)local_idx[blockIdx.y] * T;
    const int i = (int)(k / cols), j = (int)(k - (int64_t)i * cols);
    const int id = tab_p[(i >> TILE_SHIFT) * tcols + (j >> TILE_SHIFT)] & ~TILE_OWN;
    out[stride * blockIdx.y + k] = pool[(size_t)id * TILE_BYTES + (size_t)tile_cell(i, j)];
}

// ... and import of slot p's canvas = dense canvas pos[p] of the received pool, grid (tiles, slots): a tile
// that is blank (all 128) maps to tile 0, any other becomes a fresh tile (ownership is settled by
// the census that follows).
__global__ void __launch_bounds__(256) k_tile_import_many(const uint8_t* dense_pool, const int32_t* pos, size_t stride, int rows,
                                                          int cols, int32_t* tab, int T, int tcols, uint8_t* pool,
                                                          const int32_t* freelist, int* ctrl, int64_t n_pool, int* bad)
{
    __shared__ int nid_s, any_s;
    const int t = blockIdx.x;
    const size_t p = blockIdx.y;
    if ((uint64_t)(int64_t)pos[p] >= (uint64_t)n_pool)
    {  // whole workgroup: the slot keeps a blank canvas
        if (threadIdx.x == 0)
        {
            tab[p * T + t] = 0;
            *bad = 1;
        }
        return;
    }
    const uint8_t* dense = dense_pool + stride * (size_t)pos[p];
    const int ti = t / tcols, tj = t - ti * tcols;
    if (threadIdx.x == 0) any_s = 0;
    __syncthreads();
    uint8_t mine[TILE_BYTES / 256];
    bool nonblank = false;
#pragma unroll
    for (int u = 0; u < TILE_BYTES / 256; ++u)
    {
        const int q = threadIdx.x + u * 256;
        const int i = (ti << TILE_SHIFT) + (q >> TILE_SHIFT), j = (tj << TILE_SHIFT) + (q & TILE_MASK);
        mine[u] = (i < rows && j < cols) ? dense[(size_t)i * cols + j] : (uint8_t)128;
        nonblank |= mine[u] != 128;
    }
    if (nonblank) any_s = 1;
    __syncthreads();
    if (!any_s)
    {
        if (threadIdx.x == 0) tab[p * T + t] = 0;
        return;
    }
    if (threadIdx.x == 0)
    {
        const int pos_f = atomicAdd(&ctrl[0], 1);
        nid_s = pos_f < ctrl[1] ? freelist[pos_f] : -1;
        if (nid_s < 0) ctrl[2] = 1;
    }
    __syncthreads();
    const int nid = nid_s;
    if (nid < 0)
    {
        if (threadIdx.x == 0) tab[p * T + t] = 0;
        return;
    }
    uint8_t* to = pool + (size_t)nid * TILE_BYTES;
#pragma unroll
    for (int u = 0; u < TILE_BYTES / 256; ++u)
    {
        const int q = threadIdx.x + u * 256;
        to[tile_cell(q >> TILE_SHIFT, q & TILE_MASK)] = mine[u];
    }
    if (threadIdx.x == 0) tab[p * T + t] = nid;
}

// one block per tile of ONE particle: a fresh owned tile filled from the dense canvas (128 beyond its edges)
__global__ void __launch_bounds__(256) k_tile_import(const uint8_t* dense, int rows, int cols, int32_t* tab_p, int tcols,
                                                     uint8_t* pool, const int32_t* freelist, int* ctrl)
{
    __shared__ int nid_s;
    const int t = blockIdx.x;
    if (threadIdx.x == 0)
    {
        const int pos = atomicAdd(&ctrl[0], 1);
        nid_s = pos < ctrl[1] ? freelist[pos] : -1;
        if (nid_s < 0) ctrl[2] = 1;
    }
    __syncthreads();
    const int nid = nid_s;
    if (nid < 0) return;
    const int ti = t / tcols, tj = t - ti * tcols;
    uint8_t* to = pool + (size_t)nid * TILE_BYTES;
    for (int q = threadIdx.x; q < TILE_BYTES; q += blockDim.x)
    {
        const int i = (ti << TILE_SHIFT) + (q >> TILE_SHIFT), j = (tj << TILE_SHIFT) + (q & TILE_MASK);
        to[tile_cell(i, j)] = (i < rows && j < cols) ? dense[(size_t)i * cols + j] : (uint8_t)128;
    }
    if (threadIdx.x == 0) tab_p[t] = nid | TILE_OWN;
}

__global__ void __launch_bounds__(256) k_fill_poses(int64_t n, double* x, double* y, double* th, double* w, double vx,
                                                    double vy, double vth, double vw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    x[p] = vx;
    y[p] = vy;
    th[p] = vth;
    w[p] = vw;
}

// Gaussian seeding around a pose (localization mode; no reference counterpart).
__global__ void __launch_bounds__(256) k_seed_gaussian(int64_t n, double* x, double* y, double* th, double* w, double mx,
                                                       double my, double mth, double sxy_x, double sxy_y, double sth,
                                                       uint64_t seed, uint64_t global_first, double vw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double a, b, c, d;
    slam_normal_pair(seed, SLAM_STREAM_SEED, 0, global_first + (uint64_t)p, 0, &a, &b);
    slam_normal_pair(seed, SLAM_STREAM_SEED, 0, global_first + (uint64_t)p, 1, &c, &d);
    x[p] = mx + a * sxy_x;
    y[p] = my + b * sxy_y;
    th[p] = mth + c * sth;
    w[p] = vw;
}

__global__ void __launch_bounds__(256) k_seed_uniform_free(int64_t n, double* x, double* y, double* th, double* w,
                                                           const uint8_t* map, int rows, int cols, uint64_t seed,
                                                           uint64_t global_first, double vw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    slam_seed_uniform_free(seed, global_first + (uint64_t)p, map, rows, cols, &x[p], &y[p], &th[p]);
    w[p] = vw;
}

// slam/util.cpp:66-85 average_pose: per-block partial sums (fixed tree), finished on the host.
__global__ void __launch_bounds__(256) k_pose_partials(int64_t n, const double* x, const double* y, const double* th,
                                                       double* partial /* 4 per block */)
{
    __shared__ double sh[4][256];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double vx = 0, vy = 0, vc = 0, vs = 0;
    if (p < n)
    {
        vx = x[p];
        vy = y[p];
        dm_sincos(th[p], &vs, &vc);
    }
    sh[0][threadIdx.x] = vx;
    sh[1][threadIdx.x] = vy;
    sh[2][threadIdx.x] = vc;
    sh[3][threadIdx.x] = vs;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int q = 0; q < 4; ++q) partial[4 * blockIdx.x + q] = sh[q][0];
}

// The heaviest particle of a shard: per block the largest weight and the lowest slot that holds it — the front of the
// order (weight descending, slot ascending) that slam/mcl.cpp:236-238 establishes and apps/grid_slam.cpp:112 reads.
// best[2 * block] = weight, best[2 * block + 1] = slot (as a double: exact below 2^53).  NaN weights never win.
__global__ void __launch_bounds__(256) k_best_partials(int64_t n, const double* w, double* best /* 2 per block */)
{
    __shared__ double sw[256];
    __shared__ int64_t si[256];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    sw[threadIdx.x] = p < n ? w[p] : -__builtin_huge_val();
    si[threadIdx.x] = p < n ? p : INT64_MAX;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d)
        {
            const double a = sw[threadIdx.x], b = sw[threadIdx.x + d];
            const int64_t ia = si[threadIdx.x], ib = si[threadIdx.x + d];
            if (b > a || (b == a && ib < ia) || (a != a && b == b))
            {
                sw[threadIdx.x] = b;
                si[threadIdx.x] = ib;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        best[2 * blockIdx.x] = sw[0];
        best[2 * blockIdx.x + 1] = (double)si[0];
    }
}

// Batched read-only raycasts (slam/raycast.cpp:53-141), one thread per ray.
// KIND 1: unsigned char map, tests distance -> bounds -> (<128).  KIND 2: "int" map given as
// uint8 0/1, tests bounds -> distance -> (==0)  (FakeLidar's raycast<int>).
template <int KIND>
__global__ void __launch_bounds__(256) k_raycast_ro(const uint8_t* map, int rows, int cols, int64_t n, const double* px,
                                                    const double* py, const double* pth, double max2, double step,
                                                    int max_iter, double* hits)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s, c;
    dm_sincos(pth[r], &s, &c);
    const double dx = step * c, dy = step * s;
    const double ox = px[r], oy = py[r];
    double x = ox, y = oy;
    int i, j;
    slam_cell_of(rows, x, y, &i, &j);
    int pi = i, pj = j;
    double hx = -1.0, hy = -1.0;
    // Eight samples at a time: their cells are known before any byte arrives, so the eight loads are in flight together
    // (speculative: a sample past the end of the ray reads a valid byte that is never looked at); the tests then run
    // in sample order exactly as the loop of slam/raycast.cpp:24-50 does.  A thread alone with its ray is bound by the
    // latency of these loads: a scan call of 90-1080 rays through the map handle 190 -> 152 us (tools/scan_time.py; what
    // is left is the call itself: two small copies and a synchronisation).
    enum { U = 8 };
    bool done = false;
    for (int it = 0; it < max_iter && !done; it += U)
    {
        double xs[U], ys[U];
        int is[U], js[U];
        uint8_t vs[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            x += dx;
            y += dy;
            xs[u] = x;
            ys[u] = y;
            slam_cell_of(rows, x, y, &is[u], &js[u]);
            const bool inb = slam_in_bounds(rows, cols, is[u], js[u]);
            vs[u] = map[inb ? (size_t)is[u] * cols + js[u] : (size_t)0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (done || it + u >= max_iter) continue;
            i = is[u];
            j = js[u];
            if (i == pi && j == pj) continue;
            const double ex = xs[u] - ox, ey = ys[u] - oy;
            const bool far = (ex * ex + ey * ey) >= max2;
            const bool inb = slam_in_bounds(rows, cols, i, j);
            if (far || !inb)
            {  // both orders of the two tests end the ray without a hit
                done = true;
                continue;
            }
            const uint8_t v = vs[u];
            if (KIND == 1 ? (v < 128) : (v == 0))
            {
                hx = xs[u];
                hy = ys[u];
                done = true;
                continue;
            }
            pi = i;
            pj = j;
        }
    }
    hits[2 * r] = hx;
    hits[2 * r + 1] = hy;
}


// slam::raycast_mapping (slam/raycast.cpp:143-223) for n rays applied IN ORDER to one canvas: the
// single-particle form behind the shim's raycast_mapping / measurement_model_beam.  One lane walks
// (the rays of one canvas depend on each other through it); hits[2k..] = hit position or (-1, -1).
__global__ void __launch_bounds__(64) k_raycast_mapping_seq(uint8_t* map, int rows, int cols, int64_t n, const double* px,
                                                            const double* py, const double* pth, const double* z2, double zmax2,
                                                            double step, int max_iter, const uint8_t* luts, double* hits)
{
    if (blockIdx.x || threadIdx.x) return;
    for (int64_t r = 0; r < n; ++r)
    {
        double s, c;
        dm_sincos(pth[r], &s, &c);
        slam_ray_result res;
        ray_walk<true, false>(map, rows, cols, px[r], py[r], step * c, step * s, z2[r], zmax2, luts, luts + 256, max_iter, &res,
                              nullptr);
        hits[2 * r] = res.hx;
        hits[2 * r + 1] = res.hy;
    }
}
}  // namespace slamk

#include "slam_s2.h"
