// ubench_lanes.hip — issue cost of a VALU instruction against the number (and place) of active lanes (gfx950).
// Found with it: a v_add_f64 with at most a few active lanes costs 20 cycles of the SIMD, four times what it costs with 16..64
// active lanes: one-lane f64 loops (in-order sums) should run on 16 lanes that all compute the same thing.
//   hipcc --offload-arch=gfx950 -O3 -o tools/libs/ubench_lanes tools/ubench_lanes.hip && tools/libs/ubench_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define ITER 2000
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double a0, unsigned long long mask)
{
    double a = a0 + threadIdx.x, b = a0 * 1.5, c = a0 * 0.25;
    int p = threadIdx.x, q = 3, r = 5;
    float fa = (float)a0, fb = fa + 1, fc = fa + 2;
    if ((mask >> (threadIdx.x & 63)) & 1)
    {
        for (int it = 0; it < ITER; ++it)
        {
            if (OP == 0) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));) }
            if (OP == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1" : "+v"(a), "+v"(b), "+v"(c));) }
            if (OP == 2) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(p), "+v"(q), "+v"(r));) }
            if (OP == 3) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1" : "+v"(fa), "+v"(fb), "+v"(fc));) }
            if (OP == 4) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + p + q + r + fa + fb + fc;
}
template <int OP>
void run(const char* name, double* out)
{
    const int w = 4, blocks = 256 * w;
    const unsigned long long masks[] = {~0ull, 0xffffull, 0xfffull, 0xffull, 0x1ffull, 0xfull, 0x3ull, 0x1ull, 0x0001000100010001ull, 0x8000000000000000ull, 0x00000000ffff0000ull, 0x0101010101010101ull};
    for (unsigned long long m : masks)
    {
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.25, m);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.25, m);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s exec=%016llx (%2d lanes): %.2f cycles per instruction per SIMD @2.4GHz (4 waves per SIMD)\n", name, m, __builtin_popcountll(m),
               ms * 1e-3 * 2.4e9 / (16.0 * ITER * w));
    }
}
int main()
{
    double* out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 8 * 256);
    run<0>("v_add_f64", out);
    run<1>("v_fma_f64", out);
    run<4>("v_mul_f64", out);
    run<2>("v_add_u32", out);
    run<3>("v_fma_f32", out);
    return 0;
}
