// slam_s2.h — K2b, two-phase form: the SLAM-mode weighting kernel (included by kernels.h).
//
// Reproduces slam/raycast.cpp:143-223 (raycast_mapping with its occupancy update) for every beam of
// a particle IN ORDER (slam/mcl.cpp:69-74: later beams see earlier beams' writes), then
// slam/raycast.cpp:225-242 and the beam-order sum of slam/mcl.cpp:67-75.  One wavefront = one particle.
//
// Phase A, lane = beam (64 beams at once), map-independent: every lane walks its own ray with the
// literal x += dx, y += dy of slam/raycast.cpp:165-166 and records, per sample, whether the row and
// whether the column of the cell changed (2 bits per sample, 64 samples = one uint4 per beam in
// LDS).  Cell indices move monotonically along a ray and by at most one per sample, so the cell of
// sample k is the start cell plus the number of set bits up to k.  (Rounding can make a row index
// move by two in one sample when |dy| = 1 to the last bit; phase A counts its bits against the real
// displacement and a wavefront that sees a difference walks that batch of beams with the literal
// loop instead: s2_literal.)  The distance classes of slam/raycast.cpp:192-216 are thresholds on the
// sample index, because d2_k = fl(fl(ex^2) + fl(ey^2)) is non-decreasing in k (x_k, y_k move
// monotonically away from the origin and every rounding is monotone): k_far = first k with
// !(d2_k < z2), k_vfar = first k with d2_k >= zmax2, found exactly by latching the position every 16
// samples and re-walking the last 16.
// Phase B, beams strictly in order, lane = sample, 64 samples per pass, integer work only: cell from
// the bit counts (v_mbcnt), tile lookup, byte load; the loop-carried decisions of
// slam/raycast.cpp:163-220 are scalar mask arithmetic.  The geometry of a whole ray is known before
// its first byte arrives, so the loads of several passes are in flight together (a ray's cells are
// distinct: none of them can be changed by the ray's own writes).
// Afterwards, lane = beam again: the hit positions are re-walked (exact accumulated doubles) and
// the beam likelihoods are added in beam order.
#pragma once

namespace slamk
{
// passes whose loads are in flight together: 1.  (4 until late in round 2: the loads of a ray's passes up to its measured
// range issued before the first was waited for.  With the other wavefronts of the SIMD hiding that wait, the smaller
// code wins: 125k shard 22.0 -> 21.5 ms, 1000 particles 0.569 -> 0.543 ms per step, three alternating runs each.)
enum { S2_NP = 1 };

// copy-on-write tiles with the table staged as a WINDOW of W x W tiles around the sensor origin
// (every cell a ray can reach lies within max_iter cells of the start cell)
struct TiledWin
{
    enum { DEFER = 0 };
    uint8_t* pool;
    int32_t* tab;  // this particle's table in HBM
    int32_t* win;  // LDS window: entry (a, b) = table entry of tile (ti0 + a, tj0 + b), 0 outside the canvas
    int ti0, tj0, W, tcols;
    const int32_t* freelist;
    int* ctrl;
    struct Cow
    {
        uint64_t wmask;
        int e;
        uint8_t* ptr;
    };
    __device__ __forceinline__ int slot(int i, int j) const { return ((i >> TILE_SHIFT) - ti0) * W + ((j >> TILE_SHIFT) - tj0); }
    __device__ __forceinline__ uint8_t* cell_ptr(int e, int i, int j) const
    {
        return pool + (((size_t)(unsigned)(e & ~TILE_OWN) << 12) | (size_t)(unsigned)tile_cell(i, j));
    }
    // table entry of the tile of cell (i, j); lanes without a cell to read (ok == false) take entry 0 of the
    // window: with it cell_ptr is a valid address whatever i and j are
    __device__ __forceinline__ int entry(bool ok, int i, int j) const { return win[ok ? slot(i, j) : 0]; }
    __device__ __forceinline__ bool owned(int e) const { return (e & TILE_OWN) != 0; }
    // (rare once k_tile_plan has run) the lanes of `need` are about to write tiles their table does not own:
    // one tile at a time is copied by the whole wavefront and the table repointed; e and ptr of the lanes
    // that write it are updated.  A lane whose tile could not be allocated leaves the write mask.
    __device__ __noinline__ Cow cow(uint64_t need, uint64_t wmask, int i, int j, int e, uint8_t* ptr, int lane) const
    {
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(need);
        const int t = mine ? slot(i, j) : -1;
        uint64_t m = need;
        while (m)
        {
            const int l = __builtin_ctzll(m);
            const int tl = bcast_i32(t, l);
            const int old = bcast_i32(e, l) & ~TILE_OWN;
            const int tg = bcast_i32((i >> TILE_SHIFT) * tcols + (j >> TILE_SHIFT), l);
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
                    tab[tg] = nid | TILE_OWN;
                    win[tl] = nid | TILE_OWN;
                }
            }
            const bool same = t == tl;
            if (same && nid >= 0)
            {
                e = nid | TILE_OWN;
                ptr = cell_ptr(e, i, j);
            }
            const uint64_t served = __ballot(same);
            if (nid < 0) wmask &= ~served;
            m &= ~served;
        }
        // the copies (and the table) before the cell stores and the next loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return Cow{wmask, e, ptr};
    }
    // literal fallback only: one cell, every lane the same (ks: the sample's number, for TiledDefer)
    __device__ __forceinline__ uint8_t load1(int i, int j, int) const { return *cell_ptr(win[slot(i, j)], i, j); }
    __device__ __forceinline__ void store1(bool wr, int i, int j, uint8_t v, int lane, int) const
    {
        uint64_t w = __ballot(wr);
        const int e = win[slot(i, j)];
        uint8_t* ptr = cell_ptr(e, i, j);
        const uint64_t need = w & __ballot(!owned(e));
        if (need)
        {
            const Cow r = cow(need, w, i, j, e, ptr, lane);
            w = r.wmask;
            ptr = r.ptr;
        }
        if (__builtin_amdgcn_inverse_ballot_w64(w)) *ptr = v;
    }
};

// Weights now, canvas later (round 3).  At the sizes that matter nearly every particle dies at the resample that follows
// its weighting pass (bench shard: 125 000 particles, 2-9 % distinct ancestors), yet each first copied the ~45 tiles its
// scan writes: 46 GB per step for canvases nobody keeps.  This map writes nothing to memory: it reads the particle's tiles
// as they are and keeps the pass's own writes only where a LATER ray of the same scan can read them, a square window of
// LDS around the sensor.  That is exact: two rays from one origin whose directions differ by at least `gap` pass more than
// sqrt(2) apart (no common cell) beyond sqrt(2) / sin(gap) cells from the origin, so every cell that two rays of the scan
// touch is touched at samples <= kstar by both, where kstar = that range plus margin, computed by the host from the scan's
// smallest angular gap; all samples <= kstar read and write the window, all later ones read memory and drop their writes.
// The canvases of the particles the resample SELECTS are then brought up to date by running the ordinary kernel (with its
// copy-on-write copies) on those particles alone, before their tables are handed to their children: same statements on
// the same inputs, same canvases.
struct TiledDefer
{
    enum { DEFER = 1 };
    const uint8_t* pool;
    const int32_t* win;  // LDS window of the table (read only), as in TiledWin
    int ti0, tj0, W;
    uint8_t* pw;         // LDS: Wd x Wd cells, cell (i, j) at pw[(i - wi0) * Wd + (j - wj0)]
    int wi0, wj0, Wd, kstar;
    struct Cow
    {
        uint64_t wmask;
        int e;
        uint8_t* ptr;
    };
    __device__ __forceinline__ int slot(int i, int j) const { return ((i >> TILE_SHIFT) - ti0) * W + ((j >> TILE_SHIFT) - tj0); }
    __device__ __forceinline__ uint8_t* cell_ptr(int e, int i, int j) const
    {
        return const_cast<uint8_t*>(pool) + (((size_t)(unsigned)(e & ~TILE_OWN) << 12) | (size_t)(unsigned)tile_cell(i, j));
    }
    __device__ __forceinline__ int entry(bool ok, int i, int j) const { return win[ok ? slot(i, j) : 0]; }
    __device__ __forceinline__ bool owned(int) const { return true; }
    __device__ __forceinline__ Cow cow(uint64_t, uint64_t wmask, int, int, int e, uint8_t* ptr, int) const { return Cow{wmask, e, ptr}; }
    __device__ __forceinline__ int pwi(int i, int j) const { return (i - wi0) * Wd + (j - wj0); }
    __device__ __forceinline__ uint8_t load1(int i, int j, int ks) const
    {
        return ks <= kstar ? pw[pwi(i, j)] : *cell_ptr(win[slot(i, j)], i, j);
    }
    __device__ __forceinline__ void store1(bool wr, int i, int j, uint8_t v, int, int ks) const
    {
        if (wr && ks <= kstar) pw[pwi(i, j)] = v;
    }
};

struct DenseS2
{
    enum { DEFER = 0 };
    uint8_t* base;
    int cols;
    struct Cow
    {
        uint64_t wmask;
        int e;
        uint8_t* ptr;
    };
    // bit 0 of the "entry": the lane has an in-bounds cell (any other lane reads the canvas's first byte)
    __device__ __forceinline__ int entry(bool ok, int, int) const { return ok ? (TILE_OWN | 1) : TILE_OWN; }
    __device__ __forceinline__ bool owned(int) const { return true; }
    __device__ __forceinline__ uint8_t* cell_ptr(int e, int i, int j) const
    {
        return base + ((e & 1) ? (size_t)i * (size_t)cols + (size_t)j : (size_t)0);
    }
    __device__ __forceinline__ Cow cow(uint64_t, uint64_t wmask, int, int, int e, uint8_t* ptr, int) const { return Cow{wmask, e, ptr}; }
    __device__ __forceinline__ uint8_t load1(int i, int j, int) const { return *cell_ptr(1, i, j); }
    __device__ __forceinline__ void store1(bool wr, int i, int j, uint8_t v, int, int) const
    {
        if (wr) *cell_ptr(1, i, j) = v;
    }
};

// One ray with the literal loop of slam/raycast.cpp:163-220, every lane computing the same values
// (wave-uniform arguments), lane 0 storing.  Only for the batches phase A declares irregular.
template <class MAP>
__device__ __noinline__ void s2_literal(const MAP& map, int rows, int cols, double sx, double sy, double dx, double dy, double z2,
                                        double zmax2, const uint8_t* lut_free, const uint8_t* lut_occ, int max_iter, int lane,
                                        WaveRay* out)
{
    double x = sx, y = sy;
    int i, j;
    slam_cell_of(rows, x, y, &i, &j);
    int pi = i, pj = j;
    out->hx = -1.0;
    out->hy = -1.0;
    out->hcell = -1;
    out->n_read = out->n_written = 0;
    out->n_steps = 0;
    bool drew = false, found = false;
    for (int it = 0; it < max_iter && (!drew || !found); ++it)
    {
        x += dx;
        y += dy;
        out->n_steps++;
        slam_cell_of(rows, x, y, &i, &j);
        if (i == pi && j == pj) continue;
        if (!slam_in_bounds(rows, cols, i, j)) break;
        const uint8_t c = map.load1(i, j, it + 1);
        out->n_read++;
        if (!found && c < 128)
        {
            out->hx = x;
            out->hy = y;
            out->hcell = i * cols + j;
            found = true;
        }
        const double ex = x - sx, ey = y - sy;
        const double d2 = ex * ex + ey * ey;
        bool wr = false;
        uint8_t v = 0;
        if (d2 < z2)
        {
            wr = true;
            v = lut_free[c];
        }
        else
        {
            if (!drew && z2 != zmax2)
            {
                wr = true;
                v = lut_occ[c];
            }
            drew = true;
        }
        map.store1(wr && lane == 0, i, j, v, lane, it + 1);
        if (wr) out->n_written++;
        if (d2 >= zmax2) break;
        pi = i;
        pj = j;
    }
}

// lanes below f (f <= 0: none, f >= 64: all) / lanes at and above f; f wave-uniform
__device__ __forceinline__ uint64_t s2_below(int f) { return f >= 64 ? ~0ull : (f <= 0 ? 0ull : ((1ull << f) - 1ull)); }
__device__ __forceinline__ uint64_t s2_from(int f) { return ~s2_below(f); }
__device__ __forceinline__ int s2_first(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }

// Phase B of one ray.  All ray-level values are wave-uniform.  Returns the sample number of the hit
// (0: none); the hit cell follows from the hit position.
template <bool STATS, class MAP>
__device__ __forceinline__ int s2_ray(const MAP& map, const uint4* bits_q /* bits + q, stride 64 per pass */, int rows, int cols,
                                      int i0, int j0, int si, int sj, int k_far, int k_vfar, bool draw_occ, int max_iter,
                                      const uint8_t* luts, int lane, unsigned* n_read, unsigned* n_written, unsigned* n_steps,
                                      bool* ended = nullptr /* out: the ray ended within max_iter samples (k_weights_S2F passes 64) */)
{
    const int npass = (max_iter + 63) / 64;
    const int Lfar = k_far - 1, Lvfar = k_vfar - 1;  // as bit numbers: sample k is bit k - 1 of the ray
    bool drew = false, found = false, done = false;
    int base_r = 0, base_c = 0;  // row / column moves before the current pass
    unsigned steps = (unsigned)max_iter, reads = 0, writes = 0;
    int khit = 0;
    for (int g = 0; g < npass && !done;)
    {
        // The ray cannot end before sample k_far (slam/raycast.cpp:163 needs `drew`) unless it leaves the
        // grid: the first group of loads reaches a little beyond it (a ray usually ends at the first
        // occupied cell behind the measured range); further passes one at a time.
        int gn = g == 0 ? (Lfar + 24) / 64 + 1 : 1;
        if (gn > S2_NP) gn = S2_NP;
        if (gn > npass - g) gn = npass - g;
        uint64_t cand[S2_NP], oob[S2_NP];
        int ci[S2_NP], cj[S2_NP], ent[S2_NP];
        uint8_t* ptr[S2_NP];
        uint8_t cv[S2_NP];
#pragma unroll
        for (int u = 0; u < S2_NP; ++u)
        {
            if (u < gn)
            {
                const uint4 m = bits_q[(g + u) * 64];
                const uint32_t r_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.x), r_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.y);
                const uint32_t c_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.z), c_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.w);
                const uint64_t R = ((uint64_t)r_hi << 32) | r_lo, C = ((uint64_t)c_hi << 32) | c_lo;
                // moves up to and including this lane's sample (lane l = sample G0 + l + 1)
                const int er = (int)__builtin_amdgcn_mbcnt_hi(r_hi, __builtin_amdgcn_mbcnt_lo(r_lo, 0)) + (__builtin_amdgcn_inverse_ballot_w64(R) ? 1 : 0);
                const int ec = (int)__builtin_amdgcn_mbcnt_hi(c_hi, __builtin_amdgcn_mbcnt_lo(c_lo, 0)) + (__builtin_amdgcn_inverse_ballot_w64(C) ? 1 : 0);
                const int i = __mul24(base_r + er, si) + i0, j = __mul24(base_c + ec, sj) + j0;
                const uint64_t N = R | C;  // new cells (slam/raycast.cpp:171-174); phase A leaves no bits beyond max_iter
                const uint64_t INB = __ballot(slam_in_bounds(rows, cols, i, j));
                cand[u] = N & INB;
                oob[u] = N & ~INB;
                ci[u] = i;
                cj[u] = j;
                const bool have = __builtin_amdgcn_inverse_ballot_w64(cand[u]);
                ent[u] = map.entry(have, i, j);
                ptr[u] = map.cell_ptr(ent[u], i, j);
                if constexpr (MAP::DEFER)
                {
                    // samples up to kstar live in the LDS window (they may meet another ray's cells), later ones in memory
                    const bool near = (g + u) * 64 + lane + 1 <= map.kstar;
                    cv[u] = near ? map.pw[have ? map.pwi(i, j) : 0] : *ptr[u];
                }
                else
                    cv[u] = *ptr[u];
                base_r += __builtin_popcountll(R);
                base_c += __builtin_popcountll(C);
            }
        }
#pragma unroll
        for (int u = 0; u < S2_NP; ++u)
        {
            if (u < gn && !done)
            {
                const int G0 = (g + u) * 64;
                const uint64_t OCC = __ballot(cv[u] < 128) & cand[u];
                if (drew && oob[u] == 0 && G0 + 64 <= Lvfar)
                {
                    // behind the measured range, inside the grid and the sensor's range: the ray only looks for
                    // its first occupied cell (found is false here, or the ray would have ended)
                    uint64_t seen = cand[u];
                    if (OCC)
                    {
                        const int f = __builtin_ctzll(OCC);
                        khit = G0 + f + 1;
                        if (STATS) steps = (unsigned)khit;
                        seen &= s2_below(f + 1);
                        done = true;
                    }
                    if (STATS) reads += (unsigned)__builtin_popcountll(seen);
                    continue;
                }
                uint64_t W;         // lanes that write
                uint64_t WOCC = 0;  // ... the occupied table (at most the one lane of slam/raycast.cpp:203-214)
                if (G0 + 64 <= Lfar && oob[u] == 0)
                {
                    // every sample of the pass is nearer than z and in bounds: all new cells get the free update
                    if (!found && OCC)
                    {
                        khit = G0 + __builtin_ctzll(OCC) + 1;
                        found = true;
                    }
                    W = cand[u];
                    if (STATS) reads += (unsigned)__builtin_popcountll(W);
                }
                else
                {
                    const uint64_t newx = cand[u] | oob[u];
                    const int f_oob = s2_first(oob[u]);                                 // break before processing (:176)
                    const int f_hit = found ? -1 : s2_first(OCC);                       // :183-189
                    const int f_far = drew ? -1 : s2_first(newx & s2_from(Lfar - G0));  // :203-214
                    const int f_vfar = G0 + 64 <= Lvfar ? 64 : s2_first(newx & s2_from(Lvfar - G0));  // break after processing (:216)
                    const int both = f_hit > f_far ? f_hit : f_far;  // loop condition fails after this lane (:163)
                    const int stop_incl = both < f_vfar ? both : f_vfar;
                    const int term = stop_incl < f_oob ? stop_incl : f_oob;             // lane whose step ends the ray
                    const int lim = stop_incl < (f_oob - 1) ? stop_incl : (f_oob - 1);  // last processed lane
                    const uint64_t PROC = cand[u] & s2_below(lim + 1);
                    if (!found && f_hit <= lim)
                    {
                        khit = G0 + f_hit + 1;
                        found = true;
                    }
                    if (!drew && f_far <= lim)
                    {
                        drew = true;
                        if (draw_occ) WOCC = 1ull << f_far;
                    }
                    W = (PROC & s2_below(Lfar - G0)) | WOCC;
                    if (STATS) reads += (unsigned)__builtin_popcountll(PROC);
                    if (term < 64)
                    {
                        if (STATS) steps = (unsigned)(G0 + term + 1);
                        done = true;
                    }
                }
                const uint8_t v = luts[(__builtin_amdgcn_inverse_ballot_w64(WOCC) ? 256 : 0) + cv[u]];
                const uint64_t need = W & ~__ballot(map.owned(ent[u]));
                uint8_t* to = ptr[u];
                if (need)
                {
                    const typename MAP::Cow r = map.cow(need, W, ci[u], cj[u], ent[u], ptr[u], lane);
                    W = r.wmask;
                    to = r.ptr;
                }
                if constexpr (MAP::DEFER)
                {
                    // (writes beyond kstar are dropped: no ray of this scan reads those cells again, and the canvas is
                    // brought up to date after the resample, for the particles that survive it)
                    if (__builtin_amdgcn_inverse_ballot_w64(W) && G0 + lane + 1 <= map.kstar) map.pw[map.pwi(ci[u], cj[u])] = v;
                }
                else if (__builtin_amdgcn_inverse_ballot_w64(W))
                    *to = v;
                if (STATS) writes += (unsigned)__builtin_popcountll(W);
            }
        }
        g += gn;
    }
    if (STATS)
    {
        *n_read += reads;
        *n_written += writes;
        *n_steps += steps;
    }
    if (ended) *ended = done;
    return khit;
}

// LDS of one wavefront (dynamic): [512 B update tables][npass x 64 beams x 16 B geometry bits][win_w^2 x 4 B table window]
// STATS: the counters of slam_mcl_get_stats (cells read, cells written, samples) are kept: about 5 % of the kernel
#ifndef S2_WAVES
#define S2_WAVES 4
#endif
// DEFER (tiles only): weights now, canvases later: see TiledDefer; dk = kstar, dw = side of its LDS window (a multiple of 8).
// sub (the pass that brings the survivors' canvases up to date): block b works on particle sub.list[b], blocks from
// *sub.count on have nothing to do.
struct S2Subset
{
    const int32_t* list;
    const int* count;
};
template <bool TILED, bool STATS = true, bool DEFER = false>
__global__ void __launch_bounds__(64, S2_WAVES) k_weights_S2(WeightArgs a, TileArgs ta, int npass, int win_w, int force_literal,
                                                             S2Subset sub, int dk, int dw)
{
    extern __shared__ __align__(16) unsigned char s2_lds[];
    uint8_t* luts = s2_lds;                                // free table, occupied table
    uint4* bits = reinterpret_cast<uint4*>(s2_lds + 512);  // [npass][64 beams]: {R lo, R hi, C lo, C hi}
    int32_t* win = reinterpret_cast<int32_t*>(s2_lds + 512 + (size_t)npass * 1024);
    const int lane = threadIdx.x;
    if (sub.list && (int)blockIdx.x >= *sub.count) return;
    const int64_t p = sub.list ? (int64_t)sub.list[blockIdx.x] : (int64_t)blockIdx.x;
    for (int t = lane; t < 512; t += 64) luts[t] = a.luts[t];
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    const double rows_d = (double)a.rows;
    if (DEFER && !slam_defer_safe(sx, sy, rows_d, a.defer_edge)) return;  // this particle takes the ordinary pass (k_defer_split)
    int i0, j0;
    slam_cell_of(a.rows, sx, sy, &i0, &j0);
    const DenseS2 dmap = {a.grid + a.map_stride * (size_t)p, a.cols};
    TiledWin tmap = {ta.pool, nullptr, win, 0, 0, win_w, ta.tcols, ta.freelist, ta.ctrl};
    TiledDefer fmap = {ta.pool, win, 0, 0, win_w, s2_lds + 512 + (size_t)npass * 1024 + (size_t)win_w * win_w * 4, 0, 0, dw, dk};
    if (TILED)
    {
        tmap.tab = ta.tab + (size_t)p * ta.tiles_per_map;
        if (!DEFER && ta.plan_rng)
        {
            // The deferred deep copy of slam/mcl.cpp:147-155, made here instead of by k_tile_copy: the tiles k_tile_plan
            // listed for this particle are copied by its own wavefront, four tiles (16 loads of 16 bytes per lane) in
            // flight at a time: no launch and no work list of its own.  The copy phases (HBM) and the walk phases
            // (instruction issue) of different wavefronts overlap little: 2-3 % on the step.  Measured and dropped:
            // starting the first generation of wavefronts spread over 200-900 us (another 0-4 %, not reproducible), and
            // reading the next ray's cells ahead of time to warm the L2 (slower at 1000 and at 125 000 particles:
            // the wavefront is bound by its own instruction chain, not by the latency of its loads).
            const int start =ta.plan_rng[2 * p], cnt_all = ta.plan_rng[2 * p + 1];
            const int nfree = ta.ctrl[1];  // positions beyond the free list were not planned (pool exhausted: flagged there)
            for (int c0 = 0; c0 < cnt_all; c0 += 64)
            {
                const int pos = start + c0 + lane;
                const bool valid = c0 + lane < cnt_all && pos < nfree;
                int t = 0, nid = 0, old = 0;
                if (valid)
                {
                    t = ta.plan_t[pos];
                    nid = ta.freelist[pos];
                    old = tmap.tab[t] & ~TILE_OWN;
                }
                const int m = __builtin_popcountll(__ballot(valid));  // the valid lanes are the first m
                int u = 0;
                for (; u + 4 <= m; u += 4)
                {
                    uint4 v[4][TILE_BYTES / 16 / 64];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const uint4* from = reinterpret_cast<const uint4*>(ta.pool + (size_t)(unsigned)bcast_i32(old, u + k) * TILE_BYTES);
#pragma unroll
                        for (int q = 0; q < TILE_BYTES / 16 / 64; ++q) v[k][q] = from[q * 64 + lane];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        uint4* to = reinterpret_cast<uint4*>(ta.pool + (size_t)(unsigned)bcast_i32(nid, u + k) * TILE_BYTES);
#pragma unroll
                        for (int q = 0; q < TILE_BYTES / 16 / 64; ++q) store_streaming(&to[q * 64 + lane], v[k][q]);
                    }
                }
                for (; u < m; ++u)
                {
                    const uint4* from = reinterpret_cast<const uint4*>(ta.pool + (size_t)(unsigned)bcast_i32(old, u) * TILE_BYTES);
                    uint4* to = reinterpret_cast<uint4*>(ta.pool + (size_t)(unsigned)bcast_i32(nid, u) * TILE_BYTES);
                    uint4 v[TILE_BYTES / 16 / 64];
#pragma unroll
                    for (int q = 0; q < TILE_BYTES / 16 / 64; ++q) v[q] = from[q * 64 + lane];
#pragma unroll
                    for (int q = 0; q < TILE_BYTES / 16 / 64; ++q) store_streaming(&to[q * 64 + lane], v[q]);
                }
                if (valid) tmap.tab[t] = nid | TILE_OWN;
            }
            // the copies and the table before the window and the cell loads below
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // window origin: every reachable cell lies within max_iter cells of the start cell (a start far
        // outside the canvas reaches nothing: clamped only to keep the arithmetic in range)
        const int reach = npass * 64;
        const int ci = i0 < -(1 << 24) ? -(1 << 24) : (i0 > (1 << 24) ? (1 << 24) : i0);
        const int cj = j0 < -(1 << 24) ? -(1 << 24) : (j0 > (1 << 24) ? (1 << 24) : j0);
        tmap.ti0 = (ci - reach) >> TILE_SHIFT;
        tmap.tj0 = (cj - reach) >> TILE_SHIFT;
        const int trows = ta.tiles_per_map / ta.tcols;
        for (int t = lane; t < win_w * win_w; t += 64)
        {
            const int ti = tmap.ti0 + t / win_w, tj = tmap.tj0 + t % win_w;
            win[t] = (ti >= 0 && tj >= 0 && ti < trows && tj < ta.tcols) ? tmap.tab[ti * ta.tcols + tj] : 0;
        }
        if (DEFER)
        {
            // the canvas around the sensor into LDS, eight cells (one row of an 8x8 micro-block) per load; the origin is
            // a multiple of 8 in both directions and every cell of the samples 1..kstar lies inside (host: dw)
            fmap.ti0 = tmap.ti0;
            fmap.tj0 = tmap.tj0;
            fmap.wi0 = (ci - dk - 2) & ~7;
            fmap.wj0 = (cj - dk - 2) & ~7;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int per_row = dw >> 3;
            for (int t = lane; t < dw * per_row; t += 64)
            {
                const int r = t / per_row, i = fmap.wi0 + r, j = fmap.wj0 + ((t - r * per_row) << 3);
                uint64_t v = 0;
                if (i >= 0 && i < a.rows && j >= 0 && j < a.cols)  // (j is a multiple of 8: the tile of (i, j) holds (i, j + 7) as well)
                    v = *reinterpret_cast<const uint64_t*>(fmap.cell_ptr(win[fmap.slot(i, j)], i, j));
                *reinterpret_cast<uint64_t*>(fmap.pw + (size_t)r * dw + ((t - r * per_row) << 3)) = v;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int max_iter = a.max_iter;
    double acc = 0.0;
    unsigned long long n_read = 0, n_written = 0, n_steps = 0;
    for (int b0 = 0; b0 < a.B; b0 += 64)
    {
        const int b = b0 + lane;
        const int cnt = (a.B - b0) < 64 ? (a.B - b0) : 64;
        double dx = 0.0, dy = 0.0, z2 = 0.0;
        if (b < a.B)
        {
            dm_sincos(pth + a.angles[b] + a.srot, &dy, &dx);
            z2 = a.z2[b];
        }
        // ---- phase A: lane = beam ----
        int k_far, k_vfar, si, sj;
        bool regular;
        double xl, yl;  // position at sample kl, the last multiple of 16 nearer than z: hits are usually behind it
        int kl;
        {
            double x = sx, y = sy;
            int pi = i0, pj = j0;
            unsigned cnt_r = 0, cnt_c = 0;
            double xf = sx, yf = sy, xv = sx, yv = sy;  // last latch with d2 < z2 / d2 < zmax2
            int kf = 0, kv = 0;
            for (int ps = 0; ps < npass; ++ps)
            {
                uint32_t w[4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    // the low bits of the 32 row / column indices, one funnel shift per sample (sample s ends at bit s); a
                    // cell index moves by at most one per sample, so "changed" = "parity changed"
                    uint32_t par_r = 0, par_c = 0;
                    const uint32_t before_r = (uint32_t)pi & 1u, before_c = (uint32_t)pj & 1u;
#pragma unroll
                    for (int s = 0; s < 32; ++s)
                    {
                        x += dx;
                        y += dy;
                        const int ni = slam_f2i((rows_d - y) - 1.0), nj = slam_f2i(x);
                        par_r = __builtin_amdgcn_alignbit((uint32_t)ni, par_r, 1);
                        par_c = __builtin_amdgcn_alignbit((uint32_t)nj, par_c, 1);
                        pi = ni;
                        pj = nj;
                        if ((s & 15) == 15)
                        {
                            const double ex = x - sx, ey = y - sy;
                            const double d2 = ex * ex + ey * ey;
                            const int k = ps * 64 + h * 32 + s + 1;
                            if (d2 < z2)
                            {
                                xf = x;
                                yf = y;
                                kf = k;
                            }
                            if (d2 < a.zmax2)
                            {
                                xv = x;
                                yv = y;
                                kv = k;
                            }
                        }
                    }
                    const uint32_t rw = par_r ^ ((par_r << 1) | before_r), cw = par_c ^ ((par_c << 1) | before_c);
                    w[h] = rw;
                    w[2 + h] = cw;
                    cnt_r += (unsigned)__builtin_popcount(rw);
                    cnt_c += (unsigned)__builtin_popcount(cw);
                }
                if (ps == npass - 1 && max_iter < npass * 64)
                {  // samples beyond max_iter do not exist (they were walked for the check below only)
                    const uint64_t keep = s2_below(max_iter - ps * 64);
                    w[0] &= (uint32_t)keep;
                    w[1] &= (uint32_t)(keep >> 32);
                    w[2] &= (uint32_t)keep;
                    w[3] &= (uint32_t)(keep >> 32);
                }
                bits[ps * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // the recorded bits must add up to the real displacement (see the header comment)
            const int di = pi - i0, dj = pj - j0;
            regular = (unsigned)(di < 0 ? -di : di) == cnt_r && (unsigned)(dj < 0 ? -dj : dj) == cnt_c;
            si = di > 0 ? 1 : (di < 0 ? -1 : 0);
            sj = dj > 0 ? 1 : (dj < 0 ? -1 : 0);
            // thresholds: the first sample at or beyond each distance lies within 16 samples of its latch
            const int none = npass * 64 + 1;
            k_far = none;
            k_vfar = none;
            xl = xf;
            yl = yf;
            kl = kf;
#pragma unroll 4
            for (int s = 1; s <= 16; ++s)
            {
                xf += dx;
                yf += dy;
                xv += dx;
                yv += dy;
                const double ex = xf - sx, ey = yf - sy;
                if (!(ex * ex + ey * ey < z2) && k_far == none) k_far = kf + s;
                const double fx = xv - sx, fy = yv - sy;
                if (!(fx * fx + fy * fy < a.zmax2) && k_vfar == none) k_vfar = kv + s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool literal = force_literal || __ballot(!regular && b < a.B) != 0;
        // ---- phase B: beams in order, lane = sample ----
        int my_khit = 0;
        slam_ray_result mine = {-1.0, -1.0, -1};
        for (int q = 0; q < cnt; ++q)
        {
            if (literal)
            {
                WaveRay wr;
                if (TILED && DEFER)
                    s2_literal(fmap, a.rows, a.cols, sx, sy, bcast_f64(dx, q), bcast_f64(dy, q), bcast_f64(z2, q), a.zmax2, luts,
                               luts + 256, max_iter, lane, &wr);
                else if (TILED)
                    s2_literal(tmap, a.rows, a.cols, sx, sy, bcast_f64(dx, q), bcast_f64(dy, q), bcast_f64(z2, q), a.zmax2, luts,
                               luts + 256, max_iter, lane, &wr);
                else
                    s2_literal(dmap, a.rows, a.cols, sx, sy, bcast_f64(dx, q), bcast_f64(dy, q), bcast_f64(z2, q), a.zmax2, luts,
                               luts + 256, max_iter, lane, &wr);
                if (lane == q)
                {
                    mine.hx = wr.hx;
                    mine.hy = wr.hy;
                    mine.hcell = wr.hcell;
                }
                n_read += wr.n_read;
                n_written += wr.n_written;
                n_steps += wr.n_steps;
            }
            else
            {
                unsigned r = 0, w = 0, st = 0;
                const bool draw_occ = bcast_f64(z2, q) != a.zmax2;  // slam/raycast.cpp:205
                int khit;
                if (TILED && DEFER)
                    khit = s2_ray<STATS>(fmap, bits + q, a.rows, a.cols, i0, j0, bcast_i32(si, q), bcast_i32(sj, q), bcast_i32(k_far, q),
                                  bcast_i32(k_vfar, q), draw_occ, max_iter, luts, lane, &r, &w, &st);
                else if (TILED)
                    khit = s2_ray<STATS>(tmap, bits + q, a.rows, a.cols, i0, j0, bcast_i32(si, q), bcast_i32(sj, q), bcast_i32(k_far, q),
                                  bcast_i32(k_vfar, q), draw_occ, max_iter, luts, lane, &r, &w, &st);
                else
                    khit = s2_ray<STATS>(dmap, bits + q, a.rows, a.cols, i0, j0, bcast_i32(si, q), bcast_i32(sj, q), bcast_i32(k_far, q),
                                  bcast_i32(k_vfar, q), draw_occ, max_iter, luts, lane, &r, &w, &st);
                if (lane == q) my_khit = khit;
                n_read += r;
                n_written += w;
                n_steps += st;
            }
            // the next ray of this wavefront must see these stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // ---- lane = beam again: hit positions (exact accumulated doubles), likelihoods in beam order ----
        if (!literal)
        {
            // re-walk to the hit sample from the latch behind it (or from the origin)
            const bool from_latch = my_khit >= kl;
            double hx = from_latch ? xl : sx, hy = from_latch ? yl : sy;
            const int todo = my_khit - (from_latch ? kl : 0);
            for (int k = 1; __ballot(k <= todo) != 0; ++k)
                if (k <= todo)
                {
                    hx += dx;
                    hy += dy;
                }
            if (my_khit > 0)
            {
                mine.hx = hx;
                mine.hy = hy;
                int hi, hj;
                slam_cell_of(a.rows, hx, hy, &hi, &hj);
                mine.hcell = hi * a.cols + hj;  // slam/raycast.cpp:183-189: the cell of the hit sample
            }
        }
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
        if (STATS && a.stats)
        {
            atomicAdd(&a.stats[0], n_read);
            atomicAdd(&a.stats[1], n_written);
            atomicAdd(&a.stats[2], n_steps);
        }
    }
}

// ---- weights only, fast (round 3): the first half of a deferred update (TiledDefer) without the counters -----------------
// With the canvases left alone, the rays of a scan depend on each other only through the LDS window, i.e. through their
// first kstar < 64 samples.  So only the FIRST pass of 64 samples of every ray runs in beam order (s2_ray on one pass: the
// window is read and written there exactly as the full kernel would); whatever is left of a ray after that is a search
// for its first occupied cell that no other ray can influence, and 64 rays run it side by side, lane = beam, with the
// literal x += dx; y += dy of slam/raycast.cpp:165-166 and eight samples' cells in flight per round trip to memory:
//   new cell (slam/raycast.cpp:171-174) -> left the grid: no hit (:176) -> below 128: the hit (:183-189) -> at or beyond
//   max_dist: no hit (:216).
// (The loop of :163 can end earlier only once a hit has been found, and the free / occupied updates are the canvas's
// business, which the second half of the deferred update settles for the survivors.)  The geometry phase shrinks to
// one pass as well.  Same hits, same weights, bit for bit, as k_weights_S2<true, false, true>.
// Measured (125 000 x 90): that kernel 16.7 ms; this one 11.8 ms at first, 8.1 ms with the searches of two batches
// merged and the end-of-beam test left to the last samples.  Tried and dropped: ordering only the first kstar samples, with
// a routine of its own that knows nothing but the LDS window (the search grows by 41 samples per ray: step +1.2 ms at first,
// +1.9 ms against the merged loop: lane = sample costs 4 instructions per sample, the search 30 at the occupancy of its
// lanes); searching with lane = sample over the geometry bits of all passes, one ray or four side by side (fewer vector
// instructions, 38 000 against 41 000 per particle, but 26 000 scalar ones: 13.2 ms).
// The ordered first pass of one ray in k_weights_S2F (samples 1 .. 64, m = their geometry bits): s2_ray's general case
// for a first pass (found = drew = false on entry) on a TiledDefer map, and nothing else of s2_ray.
__device__ __forceinline__ int s2f_first(const TiledDefer& map, uint4 m, int rows, int cols, int i0, int j0, int si, int sj, int k_far, int k_vfar,
                                         bool draw_occ, const uint8_t* luts, int lane, bool* ended)
{
    const int Lfar = k_far - 1, Lvfar = k_vfar - 1;
    const uint32_t r_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.x), r_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.y);
    const uint32_t c_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.z), c_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.w);
    const uint64_t R = ((uint64_t)r_hi << 32) | r_lo, C = ((uint64_t)c_hi << 32) | c_lo;
    const int er = (int)__builtin_amdgcn_mbcnt_hi(r_hi, __builtin_amdgcn_mbcnt_lo(r_lo, 0)) + (__builtin_amdgcn_inverse_ballot_w64(R) ? 1 : 0);
    const int ec = (int)__builtin_amdgcn_mbcnt_hi(c_hi, __builtin_amdgcn_mbcnt_lo(c_lo, 0)) + (__builtin_amdgcn_inverse_ballot_w64(C) ? 1 : 0);
    const int i = __mul24(er, si) + i0, j = __mul24(ec, sj) + j0;
    const uint64_t N = R | C;
    const uint64_t INB = __ballot(slam_in_bounds(rows, cols, i, j));
    const uint64_t cand = N & INB, oob = N & ~INB;
    const bool have = __builtin_amdgcn_inverse_ballot_w64(cand);
    const bool near = lane + 1 <= map.kstar;  // in the LDS window (may meet another ray's cells); later samples in memory
    const int at = have && near ? map.pwi(i, j) : 0;
    const uint8_t far_v = *map.cell_ptr(map.entry(have && !near, i, j), i, j);
    const uint8_t cv = near ? map.pw[at] : far_v;
    const uint64_t OCC = __ballot(cv < 128) & cand;
    const int f_oob = s2_first(oob);                                        // break before processing (slam/raycast.cpp:176)
    const int f_hit = s2_first(OCC);                                        // :183-189
    const int f_far = s2_first(N & s2_from(Lfar));                          // :203-214
    const int f_vfar = 64 <= Lvfar ? 64 : s2_first(N & s2_from(Lvfar));    // break after processing (:216)
    const int both = f_hit > f_far ? f_hit : f_far;                         // loop condition fails after this lane (:163)
    const int stop_incl = both < f_vfar ? both : f_vfar;
    const int term = stop_incl < f_oob ? stop_incl : f_oob;                 // lane whose step ends the ray
    const int lim = stop_incl < (f_oob - 1) ? stop_incl : (f_oob - 1);      // last processed lane
    const uint64_t PROC = cand & s2_below(lim + 1);
    const int khit = f_hit <= lim ? f_hit + 1 : 0;
    const uint64_t WOCC = (f_far <= lim && draw_occ) ? 1ull << f_far : 0ull;
    const uint64_t W = (PROC & s2_below(Lfar)) | WOCC;
    *ended = term < 64;
    // (writes beyond kstar are dropped: no ray of this scan reads those cells again)
    const uint8_t v = luts[(__builtin_amdgcn_inverse_ballot_w64(WOCC) ? 256 : 0) + cv];
    if (__builtin_amdgcn_inverse_ballot_w64(W) && near) map.pw[at] = v;
    return khit;
}

#ifndef S2F_WIDE
#define S2F_WIDE 16
#endif
enum { S2F_U = 8, S2F_W = S2F_WIDE };  // samples per round trip: the loop with / without the end-of-beam test
__global__ void __launch_bounds__(64, S2_WAVES) k_weights_S2F(WeightArgs a, TileArgs ta, int win_w, int force_literal, int dk, int dw)
{
    extern __shared__ __align__(16) unsigned char s2_lds[];
    uint8_t* luts = s2_lds;                                // free table, occupied table
    uint4* bits = reinterpret_cast<uint4*>(s2_lds + 512);  // [64 beams]: the first 64 samples
    int32_t* win = reinterpret_cast<int32_t*>(s2_lds + 512 + 1024);
    const int lane = threadIdx.x;
    const int64_t p = blockIdx.x;
    for (int t = lane; t < 512; t += 64) luts[t] = a.luts[t];
    const double px = a.x[p], py = a.y[p], pth = a.th[p];
    double s0, c0;
    dm_sincos(pth + a.stheta, &s0, &c0);
    const double sx = px + c0 * a.sd;
    const double sy = py + s0 * a.sd;
    const double rows_d = (double)a.rows;
    if (!slam_defer_safe(sx, sy, rows_d, a.defer_edge)) return;  // this particle takes the ordinary pass (k_defer_split)
    int i0, j0;
    slam_cell_of(a.rows, sx, sy, &i0, &j0);
    const int32_t* tab = ta.tab + (size_t)p * ta.tiles_per_map;
    TiledDefer fmap = {ta.pool, win, 0, 0, win_w, s2_lds + 512 + 1024 + (size_t)win_w * win_w * 4, 0, 0, dw, dk};
    const int max_iter = a.max_iter;
    const int reach = ((max_iter + 63) / 64) * 64;
    {
        const int ci = i0 < -(1 << 24) ? -(1 << 24) : (i0 > (1 << 24) ? (1 << 24) : i0);
        const int cj = j0 < -(1 << 24) ? -(1 << 24) : (j0 > (1 << 24) ? (1 << 24) : j0);
        fmap.ti0 = (ci - reach) >> TILE_SHIFT;
        fmap.tj0 = (cj - reach) >> TILE_SHIFT;
        const int trows = ta.tiles_per_map / ta.tcols;
        for (int t = lane; t < win_w * win_w; t += 64)
        {
            const int ti = fmap.ti0 + t / win_w, tj = fmap.tj0 + t % win_w;
            win[t] = (ti >= 0 && tj >= 0 && ti < trows && tj < ta.tcols) ? tab[ti * ta.tcols + tj] : 0;
        }
        fmap.wi0 = (ci - dk - 2) & ~7;
        fmap.wj0 = (cj - dk - 2) & ~7;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int per_row = dw >> 3;
        for (int t = lane; t < dw * per_row; t += 64)
        {
            const int r = t / per_row, i = fmap.wi0 + r, j = fmap.wj0 + ((t - r * per_row) << 3);
            uint64_t v = 0;
            if (i >= 0 && i < a.rows && j >= 0 && j < a.cols)
                v = *reinterpret_cast<const uint64_t*>(fmap.cell_ptr(win[fmap.slot(i, j)], i, j));
            *reinterpret_cast<uint64_t*>(fmap.pw + (size_t)r * dw + ((t - r * per_row) << 3)) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int first_n = max_iter < 64 ? max_iter : 64;  // samples of the ordered first pass
    // What a batch of 64 beams leaves behind once its ordered first pass is over: the hits found there (mine), and for the
    // rays still searching (go) their position, step and cell at sample 64.
    struct Head
    {
        bool go;
        double x, y, dx, dy;
        int pi, pj;
        slam_ray_result mine;
    };
    auto head = [&](int b0) -> Head {
        const int b = b0 + lane;
        const int cnt = (a.B - b0) < 64 ? (a.B - b0) : 64;
        double dx = 0.0, dy = 0.0, z2 = 0.0;
        if (b < a.B)
        {
            dm_sincos(pth + a.angles[b] + a.srot, &dy, &dx);
            z2 = a.z2[b];
        }
        // ---- geometry of the first 64 samples, lane = beam (as phase A of k_weights_S2, one pass) ----
        int k_far, k_vfar, si, sj, kl;
        bool regular;
        double xl, yl, x64, y64;
        int ci64, cj64;
        {
            double x = sx, y = sy;
            int pi = i0, pj = j0;
            double xf = sx, yf = sy, xv = sx, yv = sy;
            int kf = 0, kv = 0;
            uint32_t w[4];
            unsigned cnt_r = 0, cnt_c = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
            {
                uint32_t par_r = 0, par_c = 0;
                const uint32_t before_r = (uint32_t)pi & 1u, before_c = (uint32_t)pj & 1u;
#pragma unroll
                for (int s = 0; s < 32; ++s)
                {
                    x += dx;
                    y += dy;
                    const int ni = slam_f2i((rows_d - y) - 1.0), nj = slam_f2i(x);
                    par_r = __builtin_amdgcn_alignbit((uint32_t)ni, par_r, 1);
                    par_c = __builtin_amdgcn_alignbit((uint32_t)nj, par_c, 1);
                    pi = ni;
                    pj = nj;
                    if ((s & 15) == 15)
                    {
                        const double ex = x - sx, ey = y - sy;
                        const double d2 = ex * ex + ey * ey;
                        const int k = h * 32 + s + 1;
                        if (d2 < z2)
                        {
                            xf = x;
                            yf = y;
                            kf = k;
                        }
                        if (d2 < a.zmax2)
                        {
                            xv = x;
                            yv = y;
                            kv = k;
                        }
                    }
                }
                const uint32_t rw = par_r ^ ((par_r << 1) | before_r), cw = par_c ^ ((par_c << 1) | before_c);
                w[h] = rw;
                w[2 + h] = cw;
                cnt_r += (unsigned)__builtin_popcount(rw);
                cnt_c += (unsigned)__builtin_popcount(cw);
            }
            if (max_iter < 64)
            {  // samples beyond max_iter do not exist
                const uint64_t keep = s2_below(max_iter);
                w[0] &= (uint32_t)keep;
                w[1] &= (uint32_t)(keep >> 32);
                w[2] &= (uint32_t)keep;
                w[3] &= (uint32_t)(keep >> 32);
            }
            bits[lane] = make_uint4(w[0], w[1], w[2], w[3]);
            const int di = pi - i0, dj = pj - j0;
            regular = (unsigned)(di < 0 ? -di : di) == cnt_r && (unsigned)(dj < 0 ? -dj : dj) == cnt_c;
            si = di > 0 ? 1 : (di < 0 ? -1 : 0);
            sj = dj > 0 ? 1 : (dj < 0 ? -1 : 0);
            x64 = x;
            y64 = y;
            ci64 = pi;
            cj64 = pj;
            const int none = 65;
            k_far = none;
            k_vfar = none;
            xl = xf;
            yl = yf;
            kl = kf;
#pragma unroll 4
            for (int s = 1; s <= 16; ++s)
            {
                xf += dx;
                yf += dy;
                xv += dx;
                yv += dy;
                const double ex = xf - sx, ey = yf - sy;
                if (!(ex * ex + ey * ey < z2) && k_far == none && kf + s <= 64) k_far = kf + s;
                const double fx = xv - sx, fy = yv - sy;
                if (!(fx * fx + fy * fy < a.zmax2) && k_vfar == none && kv + s <= 64) k_vfar = kv + s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool literal = force_literal || __ballot(!regular && b < a.B) != 0;
        // ---- the first pass of every ray, beams in order, lane = sample ----
        int my_khit = 0;
        bool my_ended = true;
        slam_ray_result mine = {-1.0, -1.0, -1};
        for (int q = 0; q < cnt; ++q)
        {
            if (literal)
            {
                WaveRay wr;
                s2_literal(fmap, a.rows, a.cols, sx, sy, bcast_f64(dx, q), bcast_f64(dy, q), bcast_f64(z2, q), a.zmax2, luts, luts + 256,
                           max_iter, lane, &wr);
                if (lane == q)
                {
                    mine.hx = wr.hx;
                    mine.hy = wr.hy;
                    mine.hcell = wr.hcell;
                }
            }
            else
            {
                bool ended;
                const bool draw_occ = bcast_f64(z2, q) != a.zmax2;  // slam/raycast.cpp:205
                const int khit = s2f_first(fmap, bits[q], a.rows, a.cols, i0, j0, bcast_i32(si, q), bcast_i32(sj, q), bcast_i32(k_far, q),
                                           bcast_i32(k_vfar, q), draw_occ, luts, lane, &ended);
                if (lane == q)
                {
                    my_khit = khit;
                    my_ended = ended;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // ---- lane = beam: hits of the first pass re-walked from their latch; rays still searching go on from sample 65 ----
        if (!literal)
        {
            const bool from_latch = my_khit >= kl;
            double hx = from_latch ? xl : sx, hy = from_latch ? yl : sy;
            const int todo = my_khit - (from_latch ? kl : 0);
            for (int k = 1; __ballot(k <= todo) != 0; ++k)
                if (k <= todo)
                {
                    hx += dx;
                    hy += dy;
                }
            if (my_khit > 0)
            {
                mine.hx = hx;
                mine.hy = hy;
                int hi, hj;
                slam_cell_of(a.rows, hx, hy, &hi, &hj);
                mine.hcell = hi * a.cols + hj;
            }
        }
        Head h;
        h.go = !literal && b < a.B && my_khit == 0 && !my_ended && max_iter > 64;
        h.x = x64;
        h.y = y64;
        h.dx = dx;
        h.dy = dy;
        h.pi = ci64;
        h.pj = cj64;
        h.mine = mine;
        return h;
    };
    // The search of up to 64 rays from sample 65 on, lane = ray, eight samples' cells in flight (see above).
    // (The end of the beam, slam/raycast.cpp:216, is tested only from sample k_safe = floor(max_dist) - 1 on: a sample k is
    // k steps of length 1 +- 1e-15 from the sensor, a cell of margin; the localization kernels' k_safe, ray.h.)
    auto far_walk = [&](bool go, double x, double y, const double dx, const double dy, int pi, int pj, slam_ray_result& res) {
        int k0 = 64;
        // (a hit is noted as the chunk's start position and the sample's index in the chunk, and its position formed after
        //  the loop by the same additions: four selects per chunk instead of five per sample, no position arrays)
        double cx = x, cy = y;
        int hit_u = -1;
        for (; k0 + S2F_W <= a.k_safe && k0 + S2F_W <= max_iter && __ballot(go) != 0; k0 += S2F_W)
        {
            int is[S2F_W], js[S2F_W];
            uint8_t cv[S2F_W];
            const double x0 = x, y0 = y;
#pragma unroll
            for (int u = 0; u < S2F_W; ++u)
            {
                x += dx;
                y += dy;
                is[u] = slam_f2i((rows_d - y) - 1.0);
                js[u] = slam_f2i(x);
                const bool ok = go && slam_in_bounds(a.rows, a.cols, is[u], js[u]);
                cv[u] = *fmap.cell_ptr(fmap.entry(ok, is[u], js[u]), ok ? is[u] : 0, ok ? js[u] : 0);
            }
            int hu = -1;
#pragma unroll
            for (int u = 0; u < S2F_W; ++u)
            {
                if (go && (is[u] != pi || js[u] != pj))
                {
                    if (!slam_in_bounds(a.rows, a.cols, is[u], js[u]))
                        go = false;  // :176
                    else if (cv[u] < 128)
                    {  // :183-189
                        hu = u;
                        go = false;
                    }
                    else
                    {
                        pi = is[u];
                        pj = js[u];
                    }
                }
            }
            if (hu >= 0)
            {
                cx = x0;
                cy = y0;
                hit_u = hu;
            }
        }
        for (int t = 0; __ballot(t <= hit_u) != 0; ++t)
            if (t <= hit_u)
            {
                cx += dx;
                cy += dy;
            }
        if (hit_u >= 0)
        {
            res.hx = cx;
            res.hy = cy;
            int hi, hj;
            slam_cell_of(a.rows, cx, cy, &hi, &hj);
            res.hcell = hi * a.cols + hj;
        }
        for (; __ballot(go) != 0; k0 += S2F_U)
        {
            double xs[S2F_U], ys[S2F_U];
            int is[S2F_U], js[S2F_U];
            uint8_t cv[S2F_U];
#pragma unroll
            for (int u = 0; u < S2F_U; ++u)
            {
                x += dx;
                y += dy;
                xs[u] = x;
                ys[u] = y;
                is[u] = slam_f2i((rows_d - y) - 1.0);
                js[u] = slam_f2i(x);
                const bool ok = go && k0 + u + 1 <= max_iter && slam_in_bounds(a.rows, a.cols, is[u], js[u]);
                cv[u] = *fmap.cell_ptr(fmap.entry(ok, is[u], js[u]), ok ? is[u] : 0, ok ? js[u] : 0);
            }
#pragma unroll
            for (int u = 0; u < S2F_U; ++u)
            {
                if (go)
                {
                    if (k0 + u + 1 > max_iter)
                        go = false;  // slam/raycast.cpp:163 with the sample bound of the walk
                    else if (is[u] != pi || js[u] != pj)
                    {
                        if (!slam_in_bounds(a.rows, a.cols, is[u], js[u]))
                            go = false;  // :176
                        else if (cv[u] < 128)
                        {  // :183-189
                            res.hx = xs[u];
                            res.hy = ys[u];
                            res.hcell = is[u] * a.cols + js[u];
                            go = false;
                        }
                        else
                        {
                            const double ex = xs[u] - sx, ey = ys[u] - sy;
                            if (ex * ex + ey * ey >= a.zmax2) go = false;  // :216
                            pi = is[u];
                            pj = js[u];
                        }
                    }
                }
            }
        }
    };
    auto weigh = [&](int b0, const slam_ray_result& mine) -> double {
        const int b = b0 + lane;
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
        return lw;
    };
    // Two batches at a time.  The searches are independent of everything (nothing is written), so those of the second
    // batch need not wait for the first batch's: its searching rays move into the lanes whose own ray of the first batch has
    // ended already (about half of them: walls nearer than 64 cells), and ONE loop serves both batches; its length is the
    // longest ray's, where two loops cost the longest ray of each batch (90 beams: 64 + 26, the second loop ran 26 lanes).
    double acc = 0.0;
    for (int b0 = 0; b0 < a.B; b0 += 128)
    {
        Head h0 = head(b0);
        const bool two = b0 + 64 < a.B;
        Head h1;
        h1.go = false;
        h1.x = h1.y = h1.dx = h1.dy = 0.0;
        h1.pi = h1.pj = 0;
        h1.mine.hx = h1.mine.hy = -1.0;
        h1.mine.hcell = -1;
        if (two) h1 = head(b0 + 64);
        const uint64_t G0 = __ballot(h0.go), G1 = __ballot(h1.go);
        const uint64_t lower = s2_below(lane);
        const int rank1 = __builtin_popcountll(G1 & lower);
        int moved = 0, src = lane;
        bool took = false;
        double wx = h0.x, wy = h0.y, wdx = h0.dx, wdy = h0.dy;
        int wpi = h0.pi, wpj = h0.pj;
        if (G1)
        {
            const uint64_t F = ~G0;
            const int nfree = __builtin_popcountll(F), n1 = __builtin_popcountll(G1);
            moved = nfree < n1 ? nfree : n1;
            int* src_of = reinterpret_cast<int*>(bits);  // (the geometry bits have served: both first passes are over)
            if (h1.go) src_of[rank1] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int rf = __builtin_popcountll(F & lower);
            took = !h0.go && rf < moved;
            if (took) src = src_of[rf];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double tx = __shfl(h1.x, src), ty = __shfl(h1.y, src), tdx = __shfl(h1.dx, src), tdy = __shfl(h1.dy, src);
            const int tpi = __shfl(h1.pi, src), tpj = __shfl(h1.pj, src);
            if (took)
            {
                wx = tx;
                wy = ty;
                wdx = tdx;
                wdy = tdy;
                wpi = tpi;
                wpj = tpj;
            }
        }
        slam_ray_result res = {-1.0, -1.0, -1};
        far_walk(h0.go || took, wx, wy, wdx, wdy, wpi, wpj, res);
        if (h0.go) h0.mine = res;
        if (G1)
        {
            // results of the rays that moved go back to their beams' lanes: {hx or +inf for "no hit", hy} (16 bytes per ray)
            double* slot = reinterpret_cast<double*>(bits);
            if (took)
            {
                slot[2 * src] = res.hcell >= 0 ? res.hx : __builtin_inf();
                slot[2 * src + 1] = res.hy;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (h1.go && rank1 < moved)
            {
                const double hx = slot[2 * lane], hy = slot[2 * lane + 1];
                if (hx != __builtin_inf())
                {
                    h1.mine.hx = hx;
                    h1.mine.hy = hy;
                    int hi, hj;
                    slam_cell_of(a.rows, hx, hy, &hi, &hj);
                    h1.mine.hcell = hi * a.cols + hj;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // (more searching rays in the second batch than free lanes: the rest in a loop of their own)
            const bool left = h1.go && rank1 >= moved;
            if (__ballot(left) != 0)
            {
                slam_ray_result r2 = {-1.0, -1.0, -1};
                far_walk(left, h1.x, h1.y, h1.dx, h1.dy, h1.pi, h1.pj, r2);
                if (left) h1.mine = r2;
            }
        }
        const double lw0 = weigh(b0, h0.mine);
        const int cnt0 = (a.B - b0) < 64 ? (a.B - b0) : 64;
        for (int l = 0; l < cnt0; ++l) acc += readlane_f64(lw0, l);
        if (two)
        {
            const double lw1 = weigh(b0 + 64, h1.mine);
            const int cnt1 = (a.B - b0 - 64) < 64 ? (a.B - b0 - 64) : 64;
            for (int l = 0; l < cnt1; ++l) acc += readlane_f64(lw1, l);
        }
    }
    if (lane == 0) a.w[p] = a.log_domain ? acc : dm_exp(acc);
}
}  // namespace slamk
