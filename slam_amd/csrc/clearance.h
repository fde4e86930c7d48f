// clearance.h — host-side construction of the clearance grid read by ray_fast_L (see ray.h (b)).
//
// E[c] = 0 when the map cell is occupied (< 128, the hit test of slam/raycast.cpp:183-184);
// otherwise min(255, Chebyshev distance from c to the nearest occupied cell or to the nearest
// cell outside the grid) — always >= 1.  The two-pass 8-neighbour chamfer with unit weights is
// exact for the Chebyshev metric.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <vector>

inline void slam_build_clearance(const uint8_t* map, int rows, int cols, uint8_t* E)
{
    const size_t n = (size_t)rows * (size_t)cols;
    std::vector<uint16_t> d(n);
    const uint16_t BIG = 60000;
    // forward pass: neighbours (i-1,j-1) (i-1,j) (i-1,j+1) (i,j-1); outside the grid counts as 0
    for (int i = 0; i < rows; ++i)
    {
        uint16_t* row = d.data() + (size_t)i * cols;
        const uint16_t* up = i ? row - cols : nullptr;
        const uint8_t* m = map + (size_t)i * cols;
        for (int j = 0; j < cols; ++j)
        {
            if (m[j] < 128)
            {
                row[j] = 0;
                continue;
            }
            uint16_t v = BIG;
            if (!up || j == 0 || j == cols - 1)
                v = 1;  // touches the outside
            else
            {
                v = std::min<uint16_t>(v, up[j - 1] + 1);
                v = std::min<uint16_t>(v, up[j] + 1);
                v = std::min<uint16_t>(v, up[j + 1] + 1);
                v = std::min<uint16_t>(v, row[j - 1] + 1);
            }
            row[j] = v;
        }
    }
    // backward pass: neighbours (i+1,j+1) (i+1,j) (i+1,j-1) (i,j+1)
    for (int i = rows - 1; i >= 0; --i)
    {
        uint16_t* row = d.data() + (size_t)i * cols;
        const uint16_t* dn = (i + 1 < rows) ? row + cols : nullptr;
        for (int j = cols - 1; j >= 0; --j)
        {
            uint16_t v = row[j];
            if (v == 0) continue;
            if (!dn || j == 0 || j == cols - 1)
                v = 1;
            else
            {
                v = std::min<uint16_t>(v, dn[j + 1] + 1);
                v = std::min<uint16_t>(v, dn[j] + 1);
                v = std::min<uint16_t>(v, dn[j - 1] + 1);
                v = std::min<uint16_t>(v, row[j + 1] + 1);
            }
            row[j] = v;
        }
    }
    for (size_t c = 0; c < n; ++c) E[c] = (uint8_t)std::min<uint16_t>(d[c], 255);
}

// Padded layout read by the kernels: (rows+4) x (cols+4) bytes, cell (i,j) at (i+2)*(cols+4)+(j+2),
// frame = 0 ("occupied or outside": the hot loop stops there and the caller tells the two apart).
inline void slam_build_clearance_padded(const uint8_t* map, int rows, int cols, uint8_t* Ep)
{
    std::vector<uint8_t> E((size_t)rows * cols);
    slam_build_clearance(map, rows, cols, E.data());
    const size_t pitch = (size_t)cols + 4;
    std::fill(Ep, Ep + (size_t)(rows + 4) * pitch, (uint8_t)0);
    for (int i = 0; i < rows; ++i) std::copy(E.begin() + (size_t)i * cols, E.begin() + (size_t)(i + 1) * cols, Ep + (size_t)(i + 2) * pitch + 2);
}

// ------------------------------------------------------------------------------------------------
// Cone clearance, one plane per octant of the ray's direction (what the localization kernels read
// since round 2; the Chebyshev grid above remains for tools and as the reference in the tests).
//
// Octant of a direction (dx, dy), y up:  bit 0 = dx < 0, bit 1 = dy < 0, bit 2 = |dx| < |dy|.
// Call the octant's major axis the one with the larger |component| (a counts cells along it, in the
// ray's sense) and b the cells along the other axis, again in the ray's sense.  A ray that starts
// anywhere inside cell c and has taken m samples is in a cell with 0 <= a <= ceil(m * |d_major|) and
// 0 <= b <= a + 1 (|d_minor| <= |d_major|; both coordinates move monotonically); with the rounding of
// the accumulated positions (far below 1e-6 of a cell) b <= a + 2 holds rigorously.  Plane value:
//   0                      the cell is occupied (< 128) or lies outside the grid (frame);
//   max(1, min(255, C))    otherwise, C = the smallest a such that a cell (a, b), 0 <= b <= a + 2, of the
//                          cone is occupied or outside: every cell of the cone with a < C is free.
// The search may therefore skip floor((C - 1) * 0.9999 / |d_major|) samples and test the next one:
// the same rule as for the Chebyshev clearance e (where C = e would be the nearest obstacle in ANY
// direction), with clearances that see only what lies ahead: a ray along a corridor jumps its length.
// O(1) per cell: C(c) = min(H(c), H(c + minor), H(c + 2 minor), 1 + C(c + major + minor)), H = distance
// along the major axis to the next occupied cell of the same line (the cone is the union of its first
// three lines and of itself shifted by one diagonal step).
inline void slam_cone_steps(int oct, int* mi, int* mj, int* ni, int* nj)
{
    const int sx = (oct & 1) ? -1 : 1, sy = (oct & 2) ? -1 : 1;  // image rows go down when y goes up
    if (!(oct & 4))
    {
        *mi = 0, *mj = sx, *ni = -sy, *nj = 0;
    }
    else
    {
        *mi = -sy, *mj = 0, *ni = 0, *nj = sx;
    }
}

// occ: R x P bytes, nonzero = occupied or frame.  out: R x P.
inline void slam_build_cone_plane(const uint8_t* occ, int R, int P, int oct, uint8_t* out)
{
    int mi, mj, ni, nj;
    slam_cone_steps(oct, &mi, &mj, &ni, &nj);
    std::vector<uint8_t> H((size_t)R * P);
    auto inc = [](uint8_t v) { return (uint8_t)(v < 255 ? v + 1 : 255); };
    if (mi == 0)
    {
        for (int i = 0; i < R; ++i)
        {
            const uint8_t* o = occ + (size_t)i * P;
            uint8_t* h = H.data() + (size_t)i * P;
            uint8_t prev = 0;  // beyond the array: occupied
            if (mj > 0)
                for (int j = P - 1; j >= 0; --j) h[j] = prev = o[j] ? 0 : inc(prev);
            else
                for (int j = 0; j < P; ++j) h[j] = prev = o[j] ? 0 : inc(prev);
        }
    }
    else
    {
        for (int k = 0; k < R; ++k)
        {
            const int i = mi > 0 ? R - 1 - k : k;
            const uint8_t* o = occ + (size_t)i * P;
            uint8_t* h = H.data() + (size_t)i * P;
            const uint8_t* hn = (i + mi >= 0 && i + mi < R) ? H.data() + (size_t)(i + mi) * P : nullptr;
            for (int j = 0; j < P; ++j) h[j] = o[j] ? 0 : inc(hn ? hn[j] : 0);
        }
    }
    const int di = mi + ni, dj = mj + nj;  // di != 0: one of the two axes is the row axis
    std::vector<uint8_t> cprev(P, 0), ccur(P, 0);
    for (int k = 0; k < R; ++k)
    {
        const int i = di > 0 ? R - 1 - k : k;
        const bool have_prev = k > 0;  // row i + di has been done (k == 0: it lies outside: all 0)
        const uint8_t* o = occ + (size_t)i * P;
        uint8_t* dst = out + (size_t)i * P;
        for (int j = 0; j < P; ++j)
        {
            if (o[j])
            {
                ccur[j] = 0;
                dst[j] = 0;
                continue;
            }
            uint8_t v = H[(size_t)i * P + j];
            for (int t = 1; t <= 2; ++t)
            {
                const int ii = i + t * ni, jj = j + t * nj;
                const uint8_t h = (ii >= 0 && ii < R && jj >= 0 && jj < P) ? H[(size_t)ii * P + jj] : 0;
                v = std::min(v, h);
            }
            const int jn = j + dj;
            const uint8_t c = (have_prev && jn >= 0 && jn < P) ? cprev[jn] : 0;
            v = std::min(v, inc(c));
            ccur[j] = v;
            dst[j] = std::max<uint8_t>(v, 1);
        }
        cprev.swap(ccur);
    }
}

// The 8 planes in the padded layout the kernels read: plane k at planes + k * plane_stride, each
// (rows+4) x (cols+4) bytes, cell (i,j) at (i+2)*(cols+4)+(j+2), frame = 0.
inline size_t slam_cone_plane_stride(int rows, int cols) { return ((size_t)(rows + 4) * (size_t)(cols + 4) + 255) / 256 * 256; }
// The kernels address the planes with 32-bit offsets.  A map too large for eight planes below 4 GiB (more than ~23 000
// cells a side) gets ONE plane of Chebyshev clearance, valid for every octant (plane stride 0): more lookups per ray,
// the same results.
inline bool slam_cone_planes_fit(int rows, int cols) { return 8 * slam_cone_plane_stride(rows, cols) < ((size_t)1 << 32); }
inline void slam_build_cone_clearance_padded(const uint8_t* map, int rows, int cols, uint8_t* planes)
{
    const int R = rows + 4, P = cols + 4;
    const size_t stride = slam_cone_plane_stride(rows, cols);
    std::vector<uint8_t> occ((size_t)R * P, 1);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) occ[(size_t)(i + 2) * P + (j + 2)] = map[(size_t)i * cols + j] < 128 ? 1 : 0;
    std::vector<std::thread> pool;
    for (int oct = 0; oct < 8; ++oct) pool.emplace_back([&, oct] { slam_build_cone_plane(occ.data(), R, P, oct, planes + oct * stride); });
    for (auto& t : pool) t.join();
}
