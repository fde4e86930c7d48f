// ubench.hip — issue cost of the VALU instructions the ray kernels use (gfx950).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITER 2000

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, long long* cyc, double a0, int i0)
{
    double a = a0 + threadIdx.x, b = a0 * 1.5, c = a0 * 0.25, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5;
    int p = i0 + threadIdx.x, q = i0 + 3, r = i0 + 5, s = i0 + 7;
    float fa = (float)a0, fb = fa + 1, fc = fa + 2, fd = fa + 3;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it)
    {
        if (OP == 0) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %1, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %1, %3" : "=v"(p), "=v"(q) : "v"(a), "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_cvt_f64_i32 %0, %2\n v_cvt_f64_i32 %1, %3" : "=v"(a), "=v"(b) : "v"(p), "v"(q));) }
        if (OP == 4) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %1" : "+v"(a), "+v"(b), "+v"(c));) }
        if (OP == 5) { REP8(asm volatile("v_rcp_f64 %0, %2\n v_rcp_f64 %1, %3" : "=v"(a), "=v"(c) : "v"(b), "v"(d));) }
        if (OP == 6) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(p), "+v"(q), "+v"(r));) }
        if (OP == 7) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %3, %3, %1, %2" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
        if (OP == 8) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0" : : "v"(a), "v"(b) : "vcc");) }
        if (OP == 9) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1" : "+v"(fa), "+v"(fb), "+v"(fc));) }
        if (OP == 10) { REP8(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3" : "=v"(fa), "=v"(fb) : "v"(a), "v"(b));) }
        if (OP == 11) { REP8(asm volatile("v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %2, %1" : "+v"(a), "+v"(b) : "v"(p));) }
        if (OP == 12) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc" : "+v"(p), "+v"(q), "+v"(r) : : "vcc");) }
        if (OP == 13) { REP8(asm volatile("v_rsq_f64 %0, %2\n v_rsq_f64 %1, %3" : "=v"(a), "=v"(c) : "v"(b), "v"(d));) }
        if (OP == 14) { REP8(asm volatile("v_div_scale_f64 %0, vcc, %2, %2, %3\n v_div_fmas_f64 %1, %2, %3, %3" : "=v"(a), "=v"(c) : "v"(b), "v"(d) : "vcc");) }
        if (OP == 15) { REP8(asm volatile("v_trunc_f64 %0, %2\n v_trunc_f64 %1, %3" : "=v"(a), "=v"(c) : "v"(b), "v"(d));) }
        if (OP == 16) { REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5" : : "v"(p), "v"(q) : "s20", "s21");) }
        if (OP == 17) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1" : "+v"(p), "+v"(q), "+v"(r));) }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + p + q + r + s + fa + fb + fc + fd;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    double* out;
    long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25, 3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25, 3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double ninst = 16.0 * ITER;
    // clock64 ticks at 100 MHz (s_memrealtime) on gfx9?  report wall-based cycles at 2.4 GHz too
    printf("%-22s waves/SIMD=%d  wall %.3f ms -> %.2f cyc/inst/wave-slot @2.4GHz (=%.2f cyc per inst per SIMD)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ninst, ms * 1e-3 * 2.4e9 / (ninst * waves_per_simd));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w : {1, 4})
    {
        run<0>("v_add_f64", w);
        run<1>("v_fma_f64", w);
        run<4>("v_mul_f64", w);
        run<2>("v_cvt_i32_f64", w);
        run<3>("v_cvt_f64_i32", w);
        run<10>("v_cvt_f32_f64", w);
        run<15>("v_trunc_f64", w);
        run<5>("v_rcp_f64", w);
        run<13>("v_rsq_f64", w);
        run<14>("v_div_scale/fmas_f64", w);
        run<8>("v_cmp_lt_f64", w);
        run<6>("v_add_u32", w);
        run<7>("v_mad_u32_u24", w);
        run<17>("v_mul_lo_u32", w);
        run<11>("v_lshrrev_b64", w);
        run<12>("v_cndmask_b32", w);
        run<9>("v_fma_f32", w);
        run<16>("v_readlane_b32", w);
    }
    return 0;
}
