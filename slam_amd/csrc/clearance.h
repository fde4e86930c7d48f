// clearance.h — host-side construction of the clearance grid read by ray_fast_L (see ray.h (b)).
//
// E[c] = 0 when the map cell is occupied (< 128, the hit test of slam/raycast.cpp:183-184);
// otherwise min(255, Chebyshev distance from c to the nearest occupied cell or to the nearest
// cell outside the grid) — always >= 1.  The two-pass 8-neighbour chamfer with unit weights is
// exact for the Chebyshev metric.
#pragma once
#include <stdint.h>

#include <algorithm>
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
