// cone_gpu.h — the eight cone-clearance planes of clearance.h, built on the device from the resident raw map
// (slam_mcl_set_map_u8: the host construction takes 3.3 s for a 16384 x 16384 map on 8 threads and then uploads
// 2.1 GB of planes; here only the raw map crosses PCIe).  Same definition, same values as slam_build_cone_plane
// (compared plane by plane in tests/test_gpu_parity.py):
//   occ   R x P bytes (R = rows + 4, P = cols + 4): 1 = occupied (< 128) or frame
//   H     distance along the octant's major axis to the next occupied cell of the same line, saturated at 255;
//         four variants (major axis = column or row index, either sense), each shared by two octants
//   C     C(c) = min(H(c), H(c + minor), H(c + 2 minor), 1 + C(c + major + minor)): a recurrence along diagonals,
//         one thread per diagonal chain, all eight octants in one launch
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conek
{
__global__ void __launch_bounds__(256) k_occ(const uint8_t* map, int rows, int cols, uint8_t* occ)
{
    const int P = cols + 4;
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t)(rows + 4) * P) return;
    const int i = (int)(k / P) - 2, j = (int)(k % P) - 2;
    occ[k] = (i < 0 || j < 0 || i >= rows || j >= cols) ? 1 : (map[(size_t)i * cols + j] < 128 ? 1 : 0);
}

// major axis = column index (sense mj = +1 or -1): one wavefront per row, 64 cells per trip from the far end of the
// row; inside a chunk the distance to the next occupied cell is a bit scan of the ballot, beyond it the chunk's
// neighbour carries on (beyond the array: occupied)
__global__ void __launch_bounds__(256) k_H_row(const uint8_t* occ, int R, int P, int mj, uint8_t* H)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const uint8_t* o = occ + (size_t)row * P;
    uint8_t* h = H + (size_t)row * P;
    const int nchunks = (P + 63) / 64;
    int carry = 0;
    for (int c = 0; c < nchunks; ++c)
    {
        const int q = mj > 0 ? nchunks - 1 - c : c;
        const int j = q * 64 + lane;
        const bool valid = j < P;
        const uint64_t M = __ballot(valid ? o[j] != 0 : true);
        int d;
        if (mj > 0)
        {
            const uint64_t m = M >> lane;
            d = m ? __builtin_ctzll(m) : (64 - lane) + carry;
        }
        else
        {
            const uint64_t m = M << (63 - lane);
            d = m ? __builtin_clzll(m) : (lane + 1) + carry;
        }
        d = d > 255 ? 255 : d;
        if (valid) h[j] = (uint8_t)d;
        carry = __builtin_amdgcn_readlane(d, mj > 0 ? 0 : 63);
    }
}

// major axis = row index (sense mi = +1 or -1): one thread per column, rows in turn (coalesced across the threads)
__global__ void __launch_bounds__(256) k_H_col(const uint8_t* occ, int R, int P, int mi, uint8_t* H)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    int prev = 0;
    for (int k = 0; k < R; ++k)
    {
        const int i = mi > 0 ? R - 1 - k : k;
        const int v = occ[(size_t)i * P + j] ? 0 : (prev < 255 ? prev + 1 : 255);
        H[(size_t)i * P + j] = (uint8_t)v;
        prev = v;
    }
}

struct ConeArgs
{
    const uint8_t* occ;
    const uint8_t* H[8];  // per octant: the H variant of its major axis
    uint8_t* planes;
    size_t plane_stride;
    int R, P;
    int ni[8], nj[8], di[8], dj[8];
};

// blockIdx.y = octant; thread t follows the chain of cells (row of step k, column t -/+ k): cell k takes the value of
// cell k - 1 of the same chain (which lies one diagonal step ahead of it); a chain outside the array carries 0
__global__ void __launch_bounds__(256) k_C(ConeArgs a)
{
    const int oct = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int R = a.R, P = a.P;
    if (t >= P + R - 1) return;
    const uint8_t* H = a.H[oct];
    uint8_t* out = a.planes + (size_t)oct * a.plane_stride;
    const int ni = a.ni[oct], nj = a.nj[oct], di = a.di[oct], dj = a.dj[oct];
    int prev = 0;
    for (int k = 0; k < R; ++k)
    {
        const int i = di > 0 ? R - 1 - k : k;
        const int j = dj > 0 ? t - k : t - (R - 1) + k;
        if (j < 0 || j >= P)
        {
            prev = 0;
            continue;
        }
        const size_t c = (size_t)i * P + j;
        if (a.occ[c])
        {
            prev = 0;
            out[c] = 0;
            continue;
        }
        int v = H[c];
#pragma unroll
        for (int s = 1; s <= 2; ++s)
        {
            const int ii = i + s * ni, jj = j + s * nj;
            const int hh = (ii >= 0 && ii < R && jj >= 0 && jj < P) ? H[(size_t)ii * P + jj] : 0;
            v = hh < v ? hh : v;
        }
        const int up = prev < 255 ? prev + 1 : 255;
        v = up < v ? up : v;
        prev = v;
        out[c] = (uint8_t)(v < 1 ? 1 : v);
    }
}
}  // namespace conek

// map_dev: rows x cols raw map on the device; planes_dev: 8 planes of plane_stride bytes (slam_cone_plane_stride).
// Allocates 5 x (rows + 4)(cols + 4) bytes of scratch for the duration of the call and synchronises the stream.
inline hipError_t slam_build_cone_planes_device(const uint8_t* map_dev, int rows, int cols, uint8_t* planes_dev, size_t plane_stride,
                                                hipStream_t stream)
{
    const int R = rows + 4, P = cols + 4;
    const size_t cells = (size_t)R * P;
    uint8_t* tmp = nullptr;
    hipError_t e = hipMalloc(&tmp, 5 * cells);
    if (e != hipSuccess) return e;
    uint8_t* occ = tmp;
    uint8_t* Hv[4] = {tmp + cells, tmp + 2 * cells, tmp + 3 * cells, tmp + 4 * cells};  // row +1, row -1, col +1, col -1
    hipLaunchKernelGGL(conek::k_occ, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, map_dev, rows, cols, occ);
    hipLaunchKernelGGL(conek::k_H_row, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, occ, R, P, 1, Hv[0]);
    hipLaunchKernelGGL(conek::k_H_row, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, occ, R, P, -1, Hv[1]);
    hipLaunchKernelGGL(conek::k_H_col, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, occ, R, P, 1, Hv[2]);
    hipLaunchKernelGGL(conek::k_H_col, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, occ, R, P, -1, Hv[3]);
    conek::ConeArgs a;
    a.occ = occ;
    a.planes = planes_dev;
    a.plane_stride = plane_stride;
    a.R = R;
    a.P = P;
    for (int oct = 0; oct < 8; ++oct)
    {
        int mi, mj, ni, nj;
        slam_cone_steps(oct, &mi, &mj, &ni, &nj);
        a.H[oct] = mi == 0 ? (mj > 0 ? Hv[0] : Hv[1]) : (mi > 0 ? Hv[2] : Hv[3]);
        a.ni[oct] = ni;
        a.nj[oct] = nj;
        a.di[oct] = mi + ni;
        a.dj[oct] = mj + nj;
    }
    hipLaunchKernelGGL(conek::k_C, dim3((unsigned)((P + R - 1 + 255) / 256), 8), dim3(256), 0, stream, a);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(stream);
    hipFree(tmp);
    return e != hipSuccess ? e : e2;
}
