// detmath_devcheck.hip — dm_exp / dm_log / dm_sincos / dm_sqrt on the device against the same functions on the host, bit for bit.
// Built by `make -C tests/host_cpp detmath_devcheck` (build()); run by tests/test_host_cpp.py on the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../slam_amd/csrc/detmath.h"
__global__ void k(const double* x, const double* y, double* e, double* l, double* s, double* c, double* q, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e[i] = dm_exp(x[i]);
    l[i] = dm_log(dm_fabs(x[i]) + 1e-300);
    dm_sincos(x[i], &s[i], &c[i]);
    q[i] = dm_sqrt(y[i]);
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), y(n), e(n), l(n), s(n), c(n), q(n);
    srand48(7);
    for (int i = 0; i < n; ++i)
    {
        const double u = drand48();
        x[i] = i % 4 == 0 ? -8.0 * u : (i % 4 == 1 ? (u - 0.5) * 1400 : (i % 4 == 2 ? 0.1 + 0.2 * u : (u - 0.5) * 4));
        // for the table-driven sincos: next to the multiples of pi/128 (its re-reduction), its hand-over at 16384, beyond
        if (i % 16 == 5) x[i] = (double)(i % 4096 - 2048) * 0.02454369260617026 * (1.0 + (u - 0.5) * (i % 32 == 5 ? 1e-15 : 1e-9));
        if (i % 16 == 9) x[i] = (u - 0.5) * (i % 32 == 9 ? 32800.0 : 4.2e6);
        if (i % 16 == 13) x[i] = (u - 0.5) * 1e-12;
        // for the square root (the compiler's expansion on the device against the host's instruction): squares and their neighbours,
        // the whole exponent range, the edge below which the expansion rescales, specials
        const double v = drand48();
        const double sp[12] = {0.0, -0.0, __builtin_inf(), __builtin_nan(""), -1.0, 5e-324, 0x1p-767, 0x1.fffffffffffffp-767, 0x1p-766, 0x1.fffffffffffffp1023, 1.0, 0x1p-1022};
        y[i] = i % 4 == 0 ? dm_from_bits(dm_bits((v * 3000.0) * (v * 3000.0)) + (uint64_t)(i % 12 / 4) - 1) : (i % 4 == 1 ? v * 1e10 : (i % 4 == 2 ? __builtin_ldexp(1.0 + u, (int)(v * 2090.0) - 1070) : sp[(i / 4) % 12]));
    }
    double *dx, *dy, *de, *dl, *ds, *dc, *dq;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dy, n * 8); (void)hipMalloc(&dq, n * 8); (void)hipMalloc(&de, n * 8); (void)hipMalloc(&dl, n * 8); (void)hipMalloc(&ds, n * 8); (void)hipMalloc(&dc, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, de, dl, ds, dc, dq, n);
    (void)hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
    long be = 0, bl = 0, bs = 0, bq = 0;
    for (int i = 0; i < n; ++i)
    {
        double hs, hc;
        dm_sincos(x[i], &hs, &hc);
        const double he = dm_exp(x[i]), hl = dm_log(dm_fabs(x[i]) + 1e-300);
        if (dm_bits(he) != dm_bits(e[i]) && !(he != he && e[i] != e[i])) { if (!be) printf("exp(%a): host %a device %a\n", x[i], he, e[i]); ++be; }
        if (dm_bits(hl) != dm_bits(l[i])) { if (!bl) printf("log(%a): host %a device %a\n", dm_fabs(x[i]) + 1e-300, hl, l[i]); ++bl; }
        if (dm_bits(hs) != dm_bits(s[i]) || dm_bits(hc) != dm_bits(c[i])) ++bs;
        const double hq = dm_sqrt(y[i]);
        if (dm_bits(hq) != dm_bits(q[i]) && !(hq != hq && q[i] != q[i])) { if (!bq) printf("sqrt(%a): host %a device %a\n", y[i], hq, q[i]); ++bq; }
    }
    printf("%d arguments: exp differs %ld, log differs %ld, sincos differs %ld, sqrt differs %ld\n", n, be, bl, bs, bq);
    return (be || bl || bs || bq) ? 1 : 0;
}
