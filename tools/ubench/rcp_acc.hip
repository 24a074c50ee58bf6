// Accuracy of v_rcp_f64 and of 1 / 2 Newton refinements on gfx950.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
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
    const int n = 1 << 20;
    double *hx = (double*)malloc(n * 8), *h0 = (double*)malloc(n * 8), *h1 = (double*)malloc(n * 8), *h2 = (double*)malloc(n * 8);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = exp((rand() / (double)RAND_MAX - 0.5) * 60.0) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)hx[i];
        double e0 = fabs((double)((h0[i] - t) / t)), e1 = fabs((double)((h1[i] - t) / t)), e2 = fabs((double)((h2[i] - t) / t));
        if (e0 > m0) m0 = e0; if (e1 > m1) m1 = e1; if (e2 > m2) m2 = e2;
    }
    printf("max rel err: v_rcp_f64 %.3e (2^%.1f)   +1 Newton %.3e (2^%.1f)   +2 Newton %.3e (2^%.1f)\n", m0, log2(m0), m1, log2(m1), m2, log2(m2));
    return 0;
}
