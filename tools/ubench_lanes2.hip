// ubench_lanes2.hip — follow-up to ubench_lanes.hip: half of the wavefronts of every SIMD run v_add_f64 with all lanes, the
// other half with one lane.  If a one-lane instruction took 20 cycles of the SIMD whatever its neighbours do, the mix would
// take (4.8 + 20) / 2 per instruction; if the penalty only exists while the whole SIMD runs narrow instructions, about 4.8.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define ITER 2000
__global__ void __launch_bounds__(256) k(double* out, double a0, unsigned long long mask_even, unsigned long long mask_odd)
{
    double a = a0 + threadIdx.x, b = a0 * 1.5, c = a0 * 0.25;
    const unsigned long long mask = (((blockIdx.x >> 3) + (blockIdx.x >> 11) + (threadIdx.x >> 6)) & 1) ? mask_odd : mask_even;  // both kinds on every SIMD (workgroups go round the 8 XCDs, wavefront i of a workgroup to SIMD i)
    if ((mask >> (threadIdx.x & 63)) & 1)
        for (int it = 0; it < ITER; ++it) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
}
int main()
{
    double* out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 8 * 256);
    const unsigned long long F = ~0ull, O = 1ull, N = 0ull;
    const unsigned long long cases[][2] = {{F, F}, {O, O}, {F, O}, {F, N}, {O, N}};
    const char* names[] = {"all waves full", "all waves one lane", "half full, half one lane", "half full, half idle", "half one lane, half idle"};
    for (int w : {4, 8})
        for (int ci = 0; ci < 5; ++ci)
        {
            const int blocks = 256 * w;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.25, cases[ci][0], cases[ci][1]);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.25, cases[ci][0], cases[ci][1]);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%d waves per SIMD, %-26s: %.3f ms\n", w, names[ci], ms);
        }
    return 0;
}
