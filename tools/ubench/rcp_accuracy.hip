// Accuracy of v_rcp_f64 (and after one / two Newton steps) on gfx950: how many refinement steps does xt_rcp need?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = __builtin_fma(-v, r, 1.0);
    r = __builtin_fma(r, e, r);
    r1[i] = r;
    e = __builtin_fma(-v, r, 1.0);
    r = __builtin_fma(r, e, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 22;
    double *hx = (double*)malloc(n * 8), *h0 = (double*)malloc(n * 8), *h1 = (double*)malloc(n * 8), *h2 = (double*)malloc(n * 8);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 200) - 100);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)hx[i];
        m0 = fmax(m0, fabs((double)((h0[i] - t) / t)));
        m1 = fmax(m1, fabs((double)((h1[i] - t) / t)));
        m2 = fmax(m2, fabs((double)((h2[i] - t) / t)));
    }
    printf("max relative error of v_rcp_f64: raw %.3e, one Newton step %.3e, two %.3e (2^-53 = %.3e)\n", m0, m1, m2, ldexp(1.0, -53));
    return 0;
}
