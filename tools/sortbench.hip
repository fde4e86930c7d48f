// sortbench.hip — standalone timing of sorts of ~100k (key, index) pairs on gfx950: rocPRIM's device sort
// against a run sort (rocprim::block_radix_sort, 4096 pairs per workgroup) followed by a placement kernel
// in several variants.  Nothing else consumes the outputs, so experimental variants are safe here.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sortbench tools/sortbench.hip && /tmp/sortbench [n]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(e)                                                                     \
    do {                                                                             \
        hipError_t err_ = (e);                                                       \
        if (err_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); exit(1); } \
    } while (0)

enum { TILE = 4096 };
typedef uint64_t K;

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_runs(const K* in, int n, K* rk, int* rv)
{
    constexpr int IPT = TILE / BLOCK;
    using sorter = rocprim::block_radix_sort<K, BLOCK, IPT, int>;
    __shared__ typename sorter::storage_type st;
    K k[IPT];
    int v[IPT];
    const int base = blockIdx.x * TILE + threadIdx.x * IPT;
#pragma unroll
    for (int i = 0; i < IPT; ++i)
    {
        const int e = base + i;
        k[i] = e < n ? in[e] : ~(K)0;
        v[i] = e;
    }
    sorter().sort(k, v, st);
#pragma unroll
    for (int i = 0; i < IPT; ++i)
    {
        rk[base + i] = k[i];
        rv[base + i] = v[i];
    }
}

// MODE 0: full (per-lane searches in memory, 4 runs in flight), 1: searches but coalesced write of the
// rank (no scatter), 2: no searches, scatter to a pseudo-random place, 3: nothing but load + coalesced store
template <int MODE>
__global__ void __launch_bounds__(256) k_place(const K* rk, const int* rv, int n, int n_runs, int* out)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = g / TILE, q = g - r * TILE;
    if (r >= n_runs) return;
    const int left = n - r * TILE;
    if (q >= (left < TILE ? left : TILE)) return;
    const K key = rk[g];
    int rank = q;
    if (MODE <= 1)
    {
        for (int j0 = 0; j0 < n_runs - 1; j0 += 4)
        {
            const K* p[4];
            int len[4], cnt[4];
            bool ties[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const int j = j0 + u, rp = j < r ? j : j + 1;
                const bool ok = j < n_runs - 1;
                const int l = ok ? n - rp * TILE : 0;
                p[u] = rk + (size_t)(ok ? rp : r) * TILE;
                len[u] = l < TILE ? l : TILE;
                ties[u] = rp < r;
                cnt[u] = 0;
            }
            for (int s = TILE; s >= 1; s >>= 1)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const int t = cnt[u] + s;
                    const K kk = p[u][(t <= len[u] ? t : 1) - 1];
                    if (t <= len[u] && (ties[u] ? kk <= key : kk < key)) cnt[u] = t;
                }
            rank += cnt[0] + cnt[1] + cnt[2] + cnt[3];
        }
    }
    if (MODE == 0) out[rank] = rv[g];
    if (MODE == 1) out[g] = rank + rv[g];
    if (MODE == 2) out[(int)(((unsigned)g * 2654435761u) % (unsigned)n)] = rv[g];
    if (MODE == 3) out[g] = rv[g] + (int)key;
}

// Wave-cooperative placement: lane u finds the window of run u that brackets the wavefront's 64 keys,
// the windows are copied to LDS with coalesced loads and searched there.
// STAGE 1: windows only, 2: + copy to LDS, 3: everything, 4: windows + loads only, 5: as 3 without fences.
// Measured (100k keys with many duplicates, as weights after a resample have): windows 6 us, loads +3.5 us,
// and the rest of the 49 us is the fallback for windows longer than WIN (runs of equal keys make them
// long; they are searched per lane in memory, one run after the other).  Chunked LDS searches of long
// windows are the next thing to try.
enum { WIN = 192 };
template <int STAGE>
__global__ void __launch_bounds__(256) k_place_lds(const K* rk_all, const int* rv, int n, int n_runs, int* out)
{
    __shared__ K lds_win[4][8][WIN];
    K(*win)[WIN] = lds_win[threadIdx.x >> 6];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int r = g / TILE, q = g - r * TILE;
    if (r >= n_runs) return;
    const int left = n - r * TILE;
    const int my_len = left < TILE ? left : TILE;
    const int q0 = q - lane;
    if (q0 >= my_len) return;
    const int n_valid = my_len - q0 < 64 ? my_len - q0 : 64;
    const bool valid = lane < n_valid;
    const K key = rk_all[valid ? g : g - lane + n_valid - 1];
    const K kmin = __shfl(key, 0, 64), kmax = __shfl(key, n_valid - 1, 64);
    int lo = 0, hi = 0;
    if (lane < n_runs && lane != r)
    {
        const int l = n - lane * TILE;
        const int len = l < TILE ? l : TILE;
        const K* rk = rk_all + (size_t)lane * TILE;
        const bool ties = lane < r;
        for (int s = TILE; s >= 1; s >>= 1)
        {
            const int ta = lo + s, tb = hi + s;
            const K ka = rk[(ta <= len ? ta : 1) - 1], kb = rk[(tb <= len ? tb : 1) - 1];
            if (ta <= len && (ties ? ka <= kmin : ka < kmin)) lo = ta;
            if (tb <= len && (ties ? kb <= kmax : kb < kmax)) hi = tb;
        }
    }
    int rank = q;
    if (STAGE == 1)
    {
        if (valid) out[g] = lo + hi;
        return;
    }
    for (int j0 = 0; j0 < n_runs - 1; j0 += 8)
    {
        int wlo[8], w[8];
        bool ties[8];
        const K* rk[8];
        int maxw = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            const int j = j0 + u, rp = j < r ? j : j + 1;
            const bool ok = j < n_runs - 1;
            wlo[u] = __shfl(lo, ok ? rp : r, 64);
            w[u] = __shfl(hi, ok ? rp : r, 64) - wlo[u];
            ties[u] = rp < r;
            rk[u] = rk_all + (size_t)(ok ? rp : r) * TILE + wlo[u];
            const int wc = w[u] <= WIN ? w[u] : 0;
            maxw = wc > maxw ? wc : maxw;
        }
        K tmp[8][WIN / 64];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < WIN / 64; ++c)
            {
                // clamped index, no select: lanes past the window re-read its last element
                const int last = (w[u] <= WIN && w[u] > 0) ? w[u] - 1 : 0;
                const int e = c * 64 + lane;
                tmp[u][c] = rk[u][e < last ? e : last];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // all window loads have landed before the first LDS store is issued
        if (STAGE == 4)
        {  // loads only
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < WIN / 64; ++c) rank += (int)(tmp[u][c] & 1);
            continue;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < WIN / 64; ++c) win[u][c * 64 + lane] = tmp[u][c];
        if (STAGE != 5)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (STAGE >= 3)
        {
            int steps = 0;
            while ((1 << steps) <= maxw) ++steps;
            for (int s = steps ? 1 << (steps - 1) : 0; s >= 1; s >>= 1)
            {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    const int t = cnt[u] + s;
                    const bool in = t <= w[u] && w[u] <= WIN;
                    const K kk = win[u][in ? t - 1 : 0];
                    if (in && (ties[u] ? kk <= key : kk < key)) cnt[u] = t;
                }
            }
        }
        else
        {
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt[u] = (int)(win[u][lane] & 1);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            if (w[u] > WIN)
            {
                int st = 0, c = 0;
                while ((1 << st) <= w[u]) ++st;
                for (int s = 1 << (st - 1); s >= 1; s >>= 1)
                {
                    const int t = c + s;
                    const K kk = rk[u][(t <= w[u] ? t : 1) - 1];
                    if (t <= w[u] && (ties[u] ? kk <= key : kk < key)) c = t;
                }
                cnt[u] = c;
            }
            rank += wlo[u] + cnt[u];
        }
        if (STAGE != 5)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (STAGE == 3)
    {
        if (valid) out[rank] = rv[g];
    }
    else if (valid)
        out[g] = rank;
}

// Chunked form: windows of any length go through LDS in chunks of WIN elements, eight runs per pass,
// no per-lane searches in memory at all.
__global__ void __launch_bounds__(256) k_place_chunks(const K* rk_all, const int* rv, int n, int n_runs, int* out)
{
    __shared__ K lds_win[4][8][WIN];
    K(*win)[WIN] = lds_win[threadIdx.x >> 6];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int r = g / TILE, q = g - r * TILE;
    if (r >= n_runs) return;
    const int left = n - r * TILE;
    const int my_len = left < TILE ? left : TILE;
    const int q0 = q - lane;
    if (q0 >= my_len) return;
    const int n_valid = my_len - q0 < 64 ? my_len - q0 : 64;
    const bool valid = lane < n_valid;
    const K key = rk_all[valid ? g : g - lane + n_valid - 1];
    const K kmin = __shfl(key, 0, 64), kmax = __shfl(key, n_valid - 1, 64);
    int lo = 0, hi = 0;
    if (lane < n_runs && lane != r)
    {
        const int l = n - lane * TILE;
        const int len = l < TILE ? l : TILE;
        const K* rk = rk_all + (size_t)lane * TILE;
        const bool ties = lane < r;
        for (int s = TILE; s >= 1; s >>= 1)
        {
            const int ta = lo + s, tb = hi + s;
            const K ka = rk[(ta <= len ? ta : 1) - 1], kb = rk[(tb <= len ? tb : 1) - 1];
            if (ta <= len && (ties ? ka <= kmin : ka < kmin)) lo = ta;
            if (tb <= len && (ties ? kb <= kmax : kb < kmax)) hi = tb;
        }
    }
    int rank = q;
    for (int j0 = 0; j0 < n_runs - 1; j0 += 8)
    {
        int w[8];
        bool ties[8];
        const K* rk[8];
        int maxw = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            const int j = j0 + u, rp = j < r ? j : j + 1;
            const bool ok = j < n_runs - 1;
            const int wlo = __shfl(lo, ok ? rp : r, 64);
            w[u] = __shfl(hi, ok ? rp : r, 64) - wlo;
            ties[u] = rp < r;
            rk[u] = rk_all + (size_t)(ok ? rp : r) * TILE + wlo;
            rank += wlo;
            maxw = w[u] > maxw ? w[u] : maxw;
        }
        for (int base = 0; base < maxw; base += WIN)
        {
            K tmp[8][WIN / 64];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < WIN / 64; ++c)
                {
                    const int e = base + c * 64 + lane;
                    const int last = w[u] > 0 ? w[u] - 1 : 0;
                    tmp[u][c] = rk[u][e < last ? e : last];
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < WIN / 64; ++c) win[u][c * 64 + lane] = tmp[u][c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 128; s >= 1; s >>= 1)  // WIN = 192 < 256
            {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    const int len = w[u] - base < WIN ? w[u] - base : WIN;  // elements of this chunk (<= 0: none)
                    const int t = cnt[u] + s;
                    const bool in = t <= len;
                    const K kk = win[u][in ? t - 1 : 0];
                    if (in && (ties[u] ? kk <= key : kk < key)) cnt[u] = t;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += cnt[u];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (valid) out[rank] = rv[g];
}

template <class F>
float time_us(F f, int reps = 50)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / reps;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    const int runs = (n + TILE - 1) / TILE;
    std::mt19937_64 rng(7);
    std::vector<K> keys(n);
    // weights after a resample: a few thousand distinct values with multiplicities
    std::vector<K> distinct(5000);
    for (auto& d : distinct) d = rng() >> 8;
    for (auto& k : keys) k = distinct[(rng() % 5000) * (rng() % 5000) / 5000];
    K *d_in, *d_rk, *d_k2;
    int *d_rv, *d_out, *d_id, *d_out2;
    CHECK(hipMalloc(&d_in, sizeof(K) * n));
    CHECK(hipMalloc(&d_k2, sizeof(K) * n));
    CHECK(hipMalloc(&d_rk, sizeof(K) * runs * TILE));
    CHECK(hipMalloc(&d_rv, sizeof(int) * runs * TILE));
    CHECK(hipMalloc(&d_out, sizeof(int) * n));
    CHECK(hipMalloc(&d_out2, sizeof(int) * n));
    CHECK(hipMalloc(&d_id, sizeof(int) * n));
    std::vector<int> ident(n);
    std::iota(ident.begin(), ident.end(), 0);
    CHECK(hipMemcpy(d_in, keys.data(), sizeof(K) * n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_id, ident.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs(nullptr, bytes, d_in, d_k2, d_id, d_out2, (size_t)n, 0, 64, 0));
    void* tmp;
    CHECK(hipMalloc(&tmp, bytes));
    printf("n = %d, %d runs\n", n, runs);
    printf("rocprim radix_sort_pairs (64-bit keys)   %8.2f us\n",
           time_us([&] { rocprim::radix_sort_pairs(tmp, bytes, d_in, d_k2, d_id, d_out2, (size_t)n, 0, 64, 0); }));
    {
        using namespace rocprim;
        using c1 = radix_sort_config<default_config, merge_sort_config<512, 512, 8>, default_config>;
        using c2 = radix_sort_config<default_config, merge_sort_config<512, 1024, 8>, default_config>;
        using c3 = radix_sort_config<default_config, merge_sort_config<512, 256, 16>, default_config>;
        using c4 = radix_sort_config<default_config, merge_sort_config<512, 512, 8, 128, 256, 8, 0>, default_config>;
        using c5 = radix_sort_config<default_config, merge_sort_config<256, 256, 4, 128, 128, 4, 0>, default_config>;
        size_t b1 = 0;
        void* t1 = nullptr;
#define TRY_CFG(cfg, label)                                                                                         \
        b1 = 0;                                                                                                     \
        CHECK((radix_sort_pairs<cfg>(nullptr, b1, d_in, d_k2, d_id, d_out2, (size_t)n, 0, 64, 0)));                \
        CHECK(hipMalloc(&t1, b1 ? b1 : 16));                                                                        \
        printf(label "  %8.2f us\n", time_us([&] { radix_sort_pairs<cfg>(t1, b1, d_in, d_k2, d_id, d_out2, (size_t)n, 0, 64, 0); })); \
        CHECK(hipFree(t1));
        TRY_CFG(c1, "rocprim merge sort, block sort 512x8, oddeven  ")
        TRY_CFG(c2, "rocprim merge sort, block sort 1024x8, oddeven ")
        TRY_CFG(c3, "rocprim merge sort, block sort 256x16, oddeven ")
        TRY_CFG(c4, "rocprim merge sort, block sort 512x8, mergepath")
        TRY_CFG(c5, "rocprim merge sort, block sort 256x4, mergepath")
    }
    printf("k_runs<256>                              %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_runs<256>, dim3(runs), dim3(256), 0, 0, d_in, n, d_rk, d_rv); }));
    printf("k_runs<1024>                             %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_runs<1024>, dim3(runs), dim3(1024), 0, 0, d_in, n, d_rk, d_rv); }));
    const dim3 pg((runs * TILE + 255) / 256);
    printf("k_place full                             %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place<0>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    // verify against the host
    std::vector<int> got(n), ref(n);
    CHECK(hipMemcpy(got.data(), d_out, sizeof(int) * n, hipMemcpyDeviceToHost));
    std::iota(ref.begin(), ref.end(), 0);
    std::stable_sort(ref.begin(), ref.end(), [&](int a, int b) { return keys[a] < keys[b]; });
    printf("permutation %s\n", got == ref ? "equals std::stable_sort" : "DIFFERS");
    printf("k_place searches, coalesced store        %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place<1>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place no searches, scattered store     %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place<2>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place load + coalesced store only      %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place<3>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place_lds windows only                 %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_lds<1>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place_lds windows + copy to LDS        %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_lds<2>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place_lds windows + loads only         %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_lds<4>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place_lds windows + LDS, no fences     %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_lds<5>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    printf("k_place_lds full                         %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_lds<3>, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    CHECK(hipMemcpy(got.data(), d_out, sizeof(int) * n, hipMemcpyDeviceToHost));
    printf("permutation %s\n", got == ref ? "equals std::stable_sort" : "DIFFERS");
    CHECK(hipMemset(d_out, 0xff, sizeof(int) * n));
    printf("k_place_chunks                           %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_place_chunks, pg, dim3(256), 0, 0, d_rk, d_rv, n, runs, d_out); }));
    CHECK(hipMemcpy(got.data(), d_out, sizeof(int) * n, hipMemcpyDeviceToHost));
    printf("permutation %s\n", got == ref ? "equals std::stable_sort" : "DIFFERS");
    return 0;
}
