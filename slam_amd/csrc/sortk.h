// sortk.h — stable sort of 16k..256k (key, value) pairs in five short launches (sample sort).
//
// Why: the filter sorts twice per step (slam/mcl.cpp:232-239: particles by weight, descending; and the pose-coherent order
// of the particle-lane ray kernel).  At 100 000 pairs rocPRIM runs a merge sort of ten dependent launches (block sort +
// seven merge passes, 55-70 us: every pass is launch-latency bound on 1.2 MB of data).  Here:
//   1. k_ss_sample   4096 samples from hashed positions (one per stratum of n/4096 elements) are ranked by counting, 16 per
//                    workgroup; every 4th in rank is a splitter (1023);
//   2. k_ss_classify every element finds its bucket (10-step search among the splitters in LDS), 1024 elements per
//                    workgroup, bucket counts per workgroup;
//   3. k_ss_scan     one workgroup per bucket: exclusive scan of its counts over the workgroups of step 2, and its total;
//   4. k_ss_scatter  every element goes to its bucket, in input order inside (bucket, workgroup): stable;
//   5. k_ss_buckets  one workgroup per two buckets (about n/1024 elements each) ranks their elements by counting in LDS
//                    and writes the values out.  Step 5 can be left out ("coarse"): the result is then ordered between buckets only,
//                    and in input order inside a bucket.
// Networks of compare-exchanges in LDS were measured first (256 buckets of ~400 elements, bitonic): 670 cycles per stage,
// 43 us for step 5 and 15 us for step 1; counting costs n_b comparisons per element, which many small buckets make cheap.
// The order is total and defined without reference to the algorithm: by the 64-bit sort key ascending, then by input
// position ascending, so the output equals that of any stable sort (rocPRIM's included), bit for bit; a bucket that does
// not fit LDS (adversarial input; the samples' positions are hashed) is ranked from memory, slowly but correctly.
// Keys: uint32 ascending (order keys), or double descending through the usual radix transform (rocPRIM's order for
// radix_sort_pairs_desc, -0.0 below +0.0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sortk
{
constexpr int BUCKETS = 1024;
constexpr int SAMPLES = 4096;  // (2^12: k_ss_sample shifts by 12)
constexpr int ITEMS = 1024;     // elements per workgroup in steps 2 and 4 (256 threads x 4)
constexpr int64_t MIN_N = 4097, MAX_N = 262144;  // below: one workgroup ranks everything (small_sort_pairs); above: radix_sort_pairs

struct Scratch
{
    uint64_t* split_key;  // [BUCKETS]
    int32_t* split_pos;   // [BUCKETS]
    int32_t* counts;      // [BUCKETS * nblocks], then row-wise exclusive scans in place
    int32_t* btot;        // [BUCKETS] elements per bucket
    uint16_t* bucket;     // [n]
    uint64_t* key2;       // [n] keys in bucket order
    int32_t* pos2;        // [n] input positions in bucket order
};
__host__ inline size_t scratch_bytes(int64_t n)
{
    const size_t nb = (size_t)((n + ITEMS - 1) / ITEMS);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(8 * (size_t)n) + up(8 * BUCKETS) + up(4 * (size_t)n) + up(4 * BUCKETS) + up(4 * BUCKETS * nb) + up(4 * BUCKETS) + up(2 * (size_t)n);
}
__host__ inline Scratch carve(void* base, int64_t n)
{
    const size_t nb = (size_t)((n + ITEMS - 1) / ITEMS);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint8_t* p = static_cast<uint8_t*>(base);
    Scratch s;
    s.key2 = reinterpret_cast<uint64_t*>(p);
    p += up(8 * (size_t)n);
    s.split_key = reinterpret_cast<uint64_t*>(p);
    p += up(8 * BUCKETS);
    s.pos2 = reinterpret_cast<int32_t*>(p);
    p += up(4 * (size_t)n);
    s.split_pos = reinterpret_cast<int32_t*>(p);
    p += up(4 * BUCKETS);
    s.counts = reinterpret_cast<int32_t*>(p);
    p += up(4 * BUCKETS * nb);
    s.btot = reinterpret_cast<int32_t*>(p);
    p += up(4 * BUCKETS);
    s.bucket = reinterpret_cast<uint16_t*>(p);
    return s;
}

// the sort key: ascending order of it is the wanted order of the input keys
__device__ __forceinline__ uint64_t to_sort_key(uint32_t k) { return k; }
__device__ __forceinline__ uint64_t to_sort_key(double w)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, w);
    const uint64_t asc = b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);  // ascending in the value
    return ~asc;                                                           // descending
}

__device__ __forceinline__ bool less(uint64_t ka, int32_t pa, uint64_t kb, int32_t pb) { return (ka < kb) | ((ka == kb) & (pa < pb)); }

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// 256 workgroups of 256 threads (one wavefront per SIMD of the chip): each gathers all the samples into LDS and ranks 16 of
// them, 16 threads per sample.
template <typename K>
__global__ void __launch_bounds__(256) k_ss_sample(const K* keys, int n, Scratch s)
{
    __shared__ uint64_t sk[SAMPLES];
    __shared__ int32_t sp[SAMPLES];
    const int t = threadIdx.x;
    // one sample from each of 4096 equal strata, at a hashed offset inside it (positions differ: the strata are disjoint)
#pragma unroll
    for (int u = 0; u < SAMPLES / 256; ++u)
    {
        // (straight-line: the 16 loads of a thread are in flight together)
        const int q = u * 256 + t;
        // stratum q = [q n / 4096, (q + 1) n / 4096): together they cover every position (an unsampled remainder at the end
        // would fall into ONE bucket whenever most keys are equal, e.g. the children of a dominant particle)
        const uint32_t lo = (uint32_t)(((uint64_t)q * (uint32_t)n) >> 12), hi = (uint32_t)(((uint64_t)(q + 1) * (uint32_t)n) >> 12);
        const int32_t pos = (int32_t)(lo + __umulhi(hash32((uint32_t)q * 2654435761u + (uint32_t)n), hi - lo));
        sk[q] = to_sort_key(keys[pos]);
        sp[q] = pos;
    }
    __syncthreads();
    const int mine = blockIdx.x * 16 + (t >> 4), part = t & 15;
    const uint64_t k = sk[mine];
    const int32_t p = sp[mine];
    int r = 0;
#pragma unroll 8
    for (int u = 0; u < SAMPLES / 16; ++u)
    {
        const int q = u * 16 + part;  // (the 16 threads of a sample read 16 consecutive entries)
        r += less(sk[q], sp[q], k, p) ? 1 : 0;
    }
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 4);
    r += __shfl_xor(r, 8);
    // every 4th of the samples in rank is a splitter: ranks 3, 7, ..., 4091
    if (part == 0 && (r & 3) == 3 && r < SAMPLES - 1)
    {
        s.split_key[r >> 2] = k;
        s.split_pos[r >> 2] = p;
    }
}

// number of splitters <= (k, pos): the bucket, 0..BUCKETS-1 (BUCKETS-1 splitters, the last entry is never tested)
__device__ __forceinline__ int find_bucket(const uint64_t* sk, const int32_t* sp, uint64_t k, int32_t pos)
{
    int b = 0;
#pragma unroll
    for (int step = BUCKETS / 2; step >= 1; step >>= 1)
    {
        const int m = b + step - 1;  // splitter index tested
        if (!less(k, pos, sk[m], sp[m])) b += step;
    }
    return b;
}

template <typename K>
__global__ void __launch_bounds__(256) k_ss_classify(const K* keys, int n, int nblocks, Scratch s)
{
    __shared__ uint64_t sk[BUCKETS];
    __shared__ int32_t sp[BUCKETS];
    __shared__ int hist[BUCKETS];
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < BUCKETS / 256; ++u)
    {
        const int q = u * 256 + t;
        sk[q] = q < BUCKETS - 1 ? s.split_key[q] : ~0ull;
        sp[q] = q < BUCKETS - 1 ? s.split_pos[q] : 0x7fffffff;
        hist[q] = 0;
    }
    __syncthreads();
    // the four searches of a thread run side by side (each step is one LDS round trip)
    constexpr int PER = ITEMS / 256;
    uint64_t k[PER];
    int b[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j)
    {
        const int i = blockIdx.x * ITEMS + j * 256 + t;
        k[j] = i < n ? to_sort_key(keys[i]) : ~0ull;
        b[j] = 0;
    }
#pragma unroll
    for (int step = BUCKETS / 2; step >= 1; step >>= 1)
#pragma unroll
        for (int j = 0; j < PER; ++j)
        {
            const int m = b[j] + step - 1;
            if (!less(k[j], blockIdx.x * ITEMS + j * 256 + t, sk[m], sp[m])) b[j] += step;
        }
#pragma unroll
    for (int j = 0; j < PER; ++j)
    {
        const int i = blockIdx.x * ITEMS + j * 256 + t;
        if (i < n)
        {
            s.bucket[i] = (uint16_t)b[j];
            atomicAdd(&hist[b[j]], 1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BUCKETS / 256; ++u) s.counts[(u * 256 + t) * nblocks + blockIdx.x] = hist[u * 256 + t];
}

// Row b of counts (bucket b, one entry per workgroup of step 2) becomes its exclusive scan; btot[b] = the row's sum.
// (one workgroup per bucket: nblocks <= 256 entries, one per thread.  The offsets of the buckets themselves, an exclusive
// scan of btot, are formed by every workgroup of steps 4 and 5 for itself: cheaper than one more launch)
__global__ void __launch_bounds__(256) k_ss_scan(int nblocks, Scratch s)
{
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int c = t < nblocks ? s.counts[blockIdx.x * nblocks + t] : 0;
    part[t] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1)
    {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t < nblocks) s.counts[blockIdx.x * nblocks + t] = part[t] - c;
    if (t == 255) s.btot[blockIdx.x] = part[255];
}

// off[b] = first output position of bucket b, b = 0..BUCKETS (off: BUCKETS + 1 ints in LDS, tmp: 256 ints; 256 threads)
__device__ __forceinline__ void bucket_offsets(const Scratch& s, int* off, int* tmp, int t)
{
    constexpr int PER = BUCKETS / 256;
    int c[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u)
    {
        c[u] = s.btot[t * PER + u];
        sum += c[u];
    }
    tmp[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1)
    {
        const int v = t >= d ? tmp[t - d] : 0;
        __syncthreads();
        tmp[t] += v;
        __syncthreads();
    }
    int run = tmp[t] - sum;
#pragma unroll
    for (int u = 0; u < PER; ++u)
    {
        off[t * PER + u] = run;
        run += c[u];
    }
    if (t == 255) off[BUCKETS] = run;
    __syncthreads();
}

// COARSE: write the values straight to their bucket (input order inside a bucket) — no step 5.
template <typename K, typename V, bool COARSE>
__global__ void __launch_bounds__(256) k_ss_scatter(const K* keys, const V* vals, int n, int nblocks, Scratch s, V* out_vals)
{
    __shared__ int run[BUCKETS];
    __shared__ int off[BUCKETS + 1];
    __shared__ int tmp[256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    bucket_offsets(s, off, tmp, t);
#pragma unroll
    for (int u = 0; u < BUCKETS / 256; ++u) run[u * 256 + t] = off[u * 256 + t] + s.counts[(u * 256 + t) * nblocks + blockIdx.x];
    __syncthreads();
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < ITEMS / 256; ++j)
    {
        const int i = blockIdx.x * ITEMS + j * 256 + t;
        const bool valid = i < n;
        const int b = valid ? s.bucket[i] : 0;
        // lanes of this wavefront with the same bucket
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 10; ++bit)
        {
            const uint64_t bal = __ballot((b >> bit) & 1);
            same &= ((b >> bit) & 1) ? bal : ~bal;
        }
        const int before = __popcll(same & lt), cnt = __popcll(same);
        int base = 0;
        for (int w = 0; w < 4; ++w)
        {
            if (wave == w && valid && before == 0)
            {
                base = run[b];
                run[b] = base + cnt;
            }
            __syncthreads();
        }
        base = __shfl(base, valid ? (int)__builtin_ctzll(same) : lane);
        if (valid)
        {
            const int dst = base + before;
            if (COARSE)
                out_vals[dst] = vals[i];
            else
            {
                s.key2[dst] = to_sort_key(keys[i]);
                s.pos2[dst] = i;
            }
        }
    }
}

// Rank of element e of a bucket [lo, hi) = number of its entries below (key, position of e).  The scatter is stable, so
// inside a bucket the positions ascend with the index: an entry u with the same key is below e exactly when u < e, and
// "below" becomes ONE 64-bit comparison per entry, k[u] < key + (u < e), with no position read (key + 1 cannot wrap: the
// all-ones sort key is a NaN weight, which is ranked the long way); eight independent entries per trip.
__device__ __forceinline__ int rank_in_bucket(const uint64_t* k, const int32_t* p, int lo, int hi, int e, uint64_t key, int32_t pos)
{
    int r = 0, u = lo;
    if (key == ~0ull)
    {
        for (; u < hi; ++u) r += less(k[u], p[u], key, pos) ? 1 : 0;
        return r;
    }
    const uint64_t key1 = key + 1;
    for (; u + 8 <= hi; u += 8)
    {
        uint64_t kk[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) kk[v] = k[u + v];
#pragma unroll
        for (int v = 0; v < 8; ++v) r += kk[v] < ((u + v < e) ? key1 : key) ? 1 : 0;
    }
    for (; u < hi; ++u) r += k[u] < ((u < e) ? key1 : key) ? 1 : 0;
    return r;
}

// Two consecutive buckets per workgroup: their elements are one contiguous range, loaded into LDS once; every thread
// ranks its element inside its own bucket (about 200 elements per workgroup: one per thread; the largest of 1024 buckets
// holds 3-4 times the mean and sets the time: one wavefront per bucket took 58 us, this form 26 us, of which 5 us are
// everything but the comparisons: 64-bit integer compares and the mask logic of the composite order are what is left).
constexpr int GROUP = 1;
constexpr int GROUP_CAP = 2048;  // elements of one group ranked in LDS (24 KB)
// epi(position, value): whatever the caller wants done with every pair as it reaches its final position (the filter
// gathers the particle states there: one launch fewer per step).
struct NoEpilogue
{
    template <typename V>
    __device__ __forceinline__ void operator()(int, V) const
    {
    }
};
template <typename V, typename EPI>
__global__ void __launch_bounds__(256) k_ss_buckets(const V* vals, Scratch s, V* out_vals, int lds_cap /* <= GROUP_CAP */, EPI epi)
{
    __shared__ uint64_t sk[GROUP_CAP];
    __shared__ int32_t sp[GROUP_CAP];
    __shared__ int off[BUCKETS + 1];
    __shared__ int tmp[256];
    const int t = threadIdx.x;
    bucket_offsets(s, off, tmp, t);
    const int b0 = blockIdx.x * GROUP;
    const int first = off[b0], total = off[b0 + GROUP] - first;
    if (total <= 0) return;
    const bool in_lds = total <= lds_cap;
    if (in_lds)
    {
        for (int e = t; e < total; e += 256)
        {
            sk[e] = s.key2[first + e];
            sp[e] = s.pos2[first + e];
        }
        __syncthreads();
    }
    for (int e = t; e < total; e += 256)
    {
        // the element's own bucket: [lo, hi) relative to the group
        int g = 0;
#pragma unroll
        for (int u = 1; u < GROUP; ++u) g += (first + e >= off[b0 + u]) ? 1 : 0;
        const int lo = off[b0 + g] - first, hi = off[b0 + g + 1] - first;
        int r = 0;
        if (in_lds)
        {
            const uint64_t k = sk[e];
            const int32_t p = sp[e];
            r = rank_in_bucket(sk, sp, lo, hi, e, k, p);
            const V v = vals[p];
            out_vals[first + lo + r] = v;
            epi(first + lo + r, v);
        }
        else
        {
            // does not fit: the same ranking from memory (only skewed samples lead here)
            const uint64_t k = s.key2[first + e];
            const int32_t p = s.pos2[first + e];
            r = rank_in_bucket(s.key2 + first, s.pos2 + first, lo, hi, e, k, p);
            const V v = vals[p];
            out_vals[first + lo + r] = v;
            epi(first + lo + r, v);
        }
    }
}

// Enqueues the sort on `stream`.  scratch: scratch_bytes(n) bytes, 256-byte aligned.  coarse: steps 1-4 only.
template <typename K, typename V, typename EPI = NoEpilogue>
__host__ inline void sort_pairs(void* scratch, const K* keys, const V* vals, V* out_vals, int64_t n, bool coarse, hipStream_t stream,
                                int lds_cap = GROUP_CAP /* tests: a smaller value sends groups of buckets to the ranking from memory */,
                                EPI epi = EPI() /* full sorts only */)
{
    const Scratch s = carve(scratch, n);
    const int nblocks = (int)((n + ITEMS - 1) / ITEMS);
    hipLaunchKernelGGL((k_ss_sample<K>), dim3(SAMPLES / 16), dim3(256), 0, stream, keys, (int)n, s);
    hipLaunchKernelGGL((k_ss_classify<K>), dim3(nblocks), dim3(256), 0, stream, keys, (int)n, nblocks, s);
    hipLaunchKernelGGL(k_ss_scan, dim3(BUCKETS), dim3(256), 0, stream, nblocks, s);
    if (coarse)
        hipLaunchKernelGGL((k_ss_scatter<K, V, true>), dim3(nblocks), dim3(256), 0, stream, keys, vals, (int)n, nblocks, s, out_vals);
    else
    {
        hipLaunchKernelGGL((k_ss_scatter<K, V, false>), dim3(nblocks), dim3(256), 0, stream, keys, vals, (int)n, nblocks, s, out_vals);
        hipLaunchKernelGGL((k_ss_buckets<V, EPI>), dim3(BUCKETS / GROUP), dim3(256), 0, stream, vals, s, out_vals,
                           lds_cap < GROUP_CAP ? lds_cap : GROUP_CAP, epi);
    }
}

// ---- the sizes the sample sort does not serve (round 3: no library sort is left in the filter step) --------------------------
// n <= SMALL_MAX: ONE workgroup holds every key in LDS and ranks each element by counting the elements below it in the total
// order (sort key, input position): n^2 / 1024 comparisons per thread, 2 us at 1000 pairs, one launch.
constexpr int SMALL_MAX = 4096;
template <typename K, typename V>
__global__ void __launch_bounds__(1024) k_small_sort(const K* keys, const V* vals, int n, V* out_vals)
{
    __shared__ uint64_t sk[SMALL_MAX];
    const int t = threadIdx.x;
    for (int e = t; e < n; e += 1024) sk[e] = to_sort_key(keys[e]);
    __syncthreads();
    for (int e = t; e < n; e += 1024)
    {
        const uint64_t k = sk[e];
        int r = 0;
        // entries before e are below it when their key is <= k, entries behind it when their key is < k
        for (int u = 0; u < e; ++u) r += sk[u] <= k ? 1 : 0;
        for (int u = e + 1; u < n; ++u) r += sk[u] < k ? 1 : 0;
        out_vals[r] = vals[e];
    }
}
template <typename K, typename V>
__host__ inline void small_sort_pairs(const K* keys, const V* vals, V* out_vals, int64_t n, hipStream_t stream)
{
    hipLaunchKernelGGL((k_small_sort<K, V>), dim3(1), dim3(1024), 0, stream, keys, vals, (int)n, out_vals);
}

// n > MAX_N: least-significant-digit radix sort, 8 bits per pass, three launches per pass: digit counts per workgroup of
// RX_ITEMS elements; one workgroup per digit turns its row of counts into exclusive offsets; every element goes to
// (first position of its digit) + (elements of that digit in earlier workgroups) + (its rank among this workgroup's elements
// of that digit, in input order): stable, so after the last pass the order is (sort key, input position), the same total
// order as the sample sort's and as any stable sort's.  Keys travel as sort keys (uint32 order keys as they are, weights
// through to_sort_key once, in the first pass); the values ping-pong between one scratch array and out_vals.
constexpr int RX_ITEMS = 2048;
template <typename SK>
__device__ __forceinline__ SK rx_key(const uint32_t* keys, int64_t i) { return (SK)keys[i]; }
template <typename SK>
__device__ __forceinline__ SK rx_key(const uint64_t* keys, int64_t i) { return (SK)keys[i]; }
template <typename SK>
__device__ __forceinline__ SK rx_key(const double* keys, int64_t i) { return (SK)to_sort_key(keys[i]); }

template <typename K, typename SK>
__global__ void __launch_bounds__(256) k_rx_hist(const K* keys, int64_t n, int shift, int nblocks, int32_t* counts)
{
    __shared__ int hist[256];
    const int t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RX_ITEMS / 256; ++j)
    {
        const int64_t i = (int64_t)blockIdx.x * RX_ITEMS + j * 256 + t;
        if (i < n) atomicAdd(&hist[(int)((rx_key<SK>(keys, i) >> shift) & 255)], 1);
    }
    __syncthreads();
    counts[(size_t)t * nblocks + blockIdx.x] = hist[t];
}

// row d of counts (digit d, one entry per workgroup) becomes its exclusive scan; dtot[d] = the row's sum
__global__ void __launch_bounds__(256) k_rx_scan(int nblocks, int32_t* counts, int32_t* dtot)
{
    __shared__ int part[256];
    const int t = threadIdx.x;
    int32_t* row = counts + (size_t)blockIdx.x * nblocks;
    int carry = 0;
    for (int base = 0; base < nblocks; base += 256)
    {
        const int c = base + t < nblocks ? row[base + t] : 0;
        part[t] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1)
        {
            const int v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        if (base + t < nblocks) row[base + t] = carry + part[t] - c;
        carry += part[255];
        __syncthreads();
    }
    if (t == 0) dtot[blockIdx.x] = carry;
}

template <typename K, typename SK, typename V>
__global__ void __launch_bounds__(256) k_rx_scatter(const K* keys, const V* vals, int64_t n, int shift, int nblocks, const int32_t* counts,
                                                    const int32_t* dtot, SK* keys_out /* nullptr: the last pass */, V* vals_out)
{
    __shared__ int run[256];
    __shared__ int tmp[256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // first position of digit t = sum of the totals of the digits below it
    const int mine = dtot[t];
    tmp[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1)
    {
        const int v = t >= d ? tmp[t - d] : 0;
        __syncthreads();
        tmp[t] += v;
        __syncthreads();
    }
    run[t] = tmp[t] - mine + counts[(size_t)t * nblocks + blockIdx.x];
    __syncthreads();
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll 1
    for (int j = 0; j < RX_ITEMS / 256; ++j)
    {
        const int64_t i = (int64_t)blockIdx.x * RX_ITEMS + j * 256 + t;
        const bool valid = i < n;
        const SK k = valid ? rx_key<SK>(keys, i) : (SK)0;
        const int b = (int)((k >> shift) & 255);
        uint64_t same = __ballot(valid);  // lanes of this wavefront with the same digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit)
        {
            const uint64_t bal = __ballot((b >> bit) & 1);
            same &= ((b >> bit) & 1) ? bal : ~bal;
        }
        const int before = __popcll(same & lt), cnt = __popcll(same);
        int base = 0;
        for (int w = 0; w < 4; ++w)
        {  // the wavefronts of the workgroup in turn: input order
            if (wave == w && valid && before == 0)
            {
                base = run[b];
                run[b] = base + cnt;
            }
            __syncthreads();
        }
        base = __shfl(base, valid ? (int)__builtin_ctzll(same) : lane);
        if (valid)
        {
            const int64_t dst = (int64_t)base + before;
            if (keys_out) keys_out[dst] = k;
            vals_out[dst] = vals[i];
        }
    }
}

template <typename SK>
__host__ inline size_t radix_scratch_bytes(int64_t n)
{
    const size_t nb = (size_t)((n + RX_ITEMS - 1) / RX_ITEMS);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return 2 * up(sizeof(SK) * (size_t)n) + up(4 * (size_t)n) + up(4 * 256 * nb) + up(4 * 256);
}

// key_bits: significant low bits of the sort key (32-bit order keys: what the caller packed; weights: 64).  n < 2^31.
template <typename K, typename SK, typename V>
__host__ inline void radix_sort_pairs(void* scratch, const K* keys, const V* vals, V* out_vals, int64_t n, int key_bits, hipStream_t stream)
{
    const int nblocks = (int)((n + RX_ITEMS - 1) / RX_ITEMS);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint8_t* p = static_cast<uint8_t*>(scratch);
    SK* kbuf[2];
    kbuf[0] = reinterpret_cast<SK*>(p);
    p += up(sizeof(SK) * (size_t)n);
    kbuf[1] = reinterpret_cast<SK*>(p);
    p += up(sizeof(SK) * (size_t)n);
    V* vbuf = reinterpret_cast<V*>(p);
    p += up(4 * (size_t)n);
    int32_t* counts = reinterpret_cast<int32_t*>(p);
    p += up(4 * 256 * (size_t)nblocks);
    int32_t* dtot = reinterpret_cast<int32_t*>(p);
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    if (passes & 1) ++passes;  // an even number of passes: the values end in out_vals (a pass over zero digits is the identity)
    for (int ps = 0; ps < passes; ++ps)
    {
        const int shift = 8 * ps;
        const bool last = ps == passes - 1;
        const V* vin = ps == 0 ? vals : ((ps & 1) ? vbuf : out_vals);
        V* vout = (ps & 1) ? out_vals : vbuf;
        SK* kout = last ? nullptr : kbuf[ps & 1];
        if (ps == 0)
        {
            hipLaunchKernelGGL((k_rx_hist<K, SK>), dim3(nblocks), dim3(256), 0, stream, keys, n, shift, nblocks, counts);
            hipLaunchKernelGGL(k_rx_scan, dim3(256), dim3(256), 0, stream, nblocks, counts, dtot);
            hipLaunchKernelGGL((k_rx_scatter<K, SK, V>), dim3(nblocks), dim3(256), 0, stream, keys, vin, n, shift, nblocks, counts, dtot, kout, vout);
        }
        else
        {
            const SK* kin = kbuf[(ps - 1) & 1];
            hipLaunchKernelGGL((k_rx_hist<SK, SK>), dim3(nblocks), dim3(256), 0, stream, kin, n, shift, nblocks, counts);
            hipLaunchKernelGGL(k_rx_scan, dim3(256), dim3(256), 0, stream, nblocks, counts, dtot);
            hipLaunchKernelGGL((k_rx_scatter<SK, SK, V>), dim3(nblocks), dim3(256), 0, stream, kin, vin, n, shift, nblocks, counts, dtot, kout, vout);
        }
    }
}
}  // namespace sortk
