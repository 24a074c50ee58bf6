// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the fp64 / int VALU instructions used by the
// track-likelihood kernels on gfx950.  4 waves per SIMD, 8 independent chains per wave, s_memtime-free: wall clock only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define ITER 32768
#define CH 8

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double seed, int n, double c0 = 0.5, double c1 = 0.25)
{
    double x[CH];
    int y[CH];
    for (int c = 0; c < CH; ++c) {
        x[c] = seed + threadIdx.x * 1e-3 + c;
        y[c] = threadIdx.x + c;
    }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) x[c] = __builtin_fma(x[c], 1.0000001, 1e-9);
            if (OP == 1) x[c] = x[c] * 1.0000001;
            if (OP == 2) x[c] = x[c] + 1e-9;
            if (OP == 3) x[c] = __builtin_amdgcn_ldexp(x[c], y[c] & 1);
            if (OP == 4) x[c] = __builtin_amdgcn_frexp_mant(x[c]) + 1.0;
            if (OP == 5) y[c] += __builtin_amdgcn_frexp_exp(x[c] + y[c]);
            if (OP == 6) x[c] = __builtin_rint(x[c]) + 0.3;
            if (OP == 7) y[c] += (int)x[c];
            if (OP == 8) x[c] = __builtin_amdgcn_rcp(x[c]) + 1.0;
            if (OP == 9) x[c] = fmax(x[c], (double)y[c]);
            if (OP == 10) y[c] = (x[c] < (double)i) ? y[c] + 1 : y[c];
            if (OP == 11) y[c] = y[c] * 3 + 1;             // v_mul_lo/mad u32
            if (OP == 12) y[c] = (y[c] + i) ^ c;           // add + xor
            if (OP == 13) y[c] = __shfl_xor(y[c], 1, 64);  // ds_bpermute / dpp
            if (OP == 14) x[c] = sqrt(x[c]) + 1.0;
            if (OP == 15) { asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(x[c])); }
            if (OP == 16) x[c] = __builtin_fma(x[c], 0.999, c0);               // fma with an SGPR-pair constant + literal
            if (OP == 17) x[c] = __builtin_fma(x[c], c1, c0);                  // two SGPR-pair constants
            if (OP == 18) x[c] = __builtin_fma(x[c], x[(c + 1) % CH], 0.5);    // inline constant
            if (OP == 19) y[c] = y[c] > i ? y[c] : x[c] > 1.0;                 // cmp/cndmask mix
            if (OP == 20) x[c] = __builtin_fma(x[c], x[(c + 1) % CH], x[(c + 2) % CH]);  // 3 VGPR-pair sources
        }
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += x[c] + y[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, double* d, int extra_per_iter)
{
    const int blocks = 256 * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1.5, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 1.5, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = 4 waves * ITER * CH * (1 + extra)
    const double inst = 4.0 * ITER * CH;
    const double cyc = ms * 1e-3 * 2.4e9 / inst;
    printf("%-28s %8.3f ms  %6.2f cycles per (op%s) at 2.4 GHz nominal\n", name, ms, cyc, extra_per_iter ? " + helper" : "");
    return cyc;
}

int main()
{
    double* d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(double));
    run<0>("v_fma_f64", d, 0);
    run<1>("v_mul_f64", d, 0);
    run<2>("v_add_f64", d, 0);
    run<15>("v_add_f64 (asm)", d, 0);
    run<3>("v_ldexp_f64 (+and)", d, 1);
    run<4>("v_frexp_mant_f64 (+add_f64)", d, 1);
    run<5>("v_frexp_exp_i32_f64 (+2)", d, 1);
    run<6>("v_rndne_f64 (+add_f64)", d, 1);
    run<7>("v_cvt_i32_f64 (+add_u32)", d, 1);
    run<8>("v_rcp_f64 (+add_f64)", d, 1);
    run<9>("v_max_f64 (+cvt_f64_i32)", d, 1);
    run<10>("v_cmp_lt_f64+cndmask(+cvt)", d, 1);
    run<11>("v_mad_u32", d, 0);
    run<12>("v_add_u32+v_xor", d, 1);
    run<13>("shfl_xor 1 (dpp/bpermute)", d, 0);
    run<14>("sqrt f64 (+add)", d, 1);
    run<16>("v_fma_f64 v, literal, sgpr", d, 0);
    run<17>("v_fma_f64 v, sgpr, sgpr", d, 0);
    run<18>("v_fma_f64 v, v, inline", d, 0);
    run<20>("v_fma_f64 v, v, v", d, 0);
    for (int rep = 0; rep < 3; ++rep) run<0>("v_fma_f64 (repeat, warm)", d, 0);
    return 0;
}
