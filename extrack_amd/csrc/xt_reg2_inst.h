// Instantiation table of the register-resident 2-state kernels (xt_reg2.h) for ONE frame_len: included by extrack_reg2_f{4..7}.hip with
// XT_R2_F defined, so that the four frame lengths compile side by side (every instance unrolls the whole step loop: ~4 s each).
#include "xt_host.h"

#include "xt_reg2.h"

#ifndef XT_R2_LL_WAVES
#define XT_R2_LL_WAVES 4  // waves per SIMD asked of the register allocator for the likelihood-only kernels: 124 VGPRs, no scratch; its own choice (131 VGPRs, 3 waves) ran 2.5 % slower, 5 waves 0.3 % slower (same-box A/B, C2)
#endif
template <int F, int D, int K>
__global__ void __launch_bounds__(64 * XT_F2_WAVES, XT_R2_LL_WAVES) xt_ll_r2_kernel(XtKernelArgs a)
{
    DevCtx cx;
    XtGradArgs ga;
    ga.dblob = nullptr;
    ga.gpartials = nullptr;
    xt_r2_body<F, D, K, 0>(a, ga, cx);
    xt_fused_total(a);
}

// Waves per SIMD asked of the register allocator.  8 directions x 2 sequences x (1 + D + K) doubles of tangents alone are 128 VGPRs; measured
// on C2 (MI355X): 4 directions 12.6 ms at 3 waves (168 VGPRs, spills in the step loop) against 10.4 ms at 2 waves (no spill) - a spill costs
// more than the lost occupancy, the step is issue-bound.
XT_HD constexpr int xt_r2_waves(int NP, int D, int K) { return (NP * (1 + D + K) * 4 > 40) ? 2 : 3; }

template <int F, int D, int K, int NP, int WV, int VAR = 0>
__global__ void __launch_bounds__(64 * XT_F2_WAVES, WV) xt_grad_r2_kernel(XtKernelArgs a, XtGradArgs ga)
{
    DevCtx cx;
    xt_r2_body<F, D, K, NP, VAR>(a, ga, cx);
}

template <int F, int D, int K>
static const void* r2_grad_np(int NP)
{
    switch (NP) {
#define XT_R2_NP(N) \
    case N: return (const void*)xt_grad_r2_kernel<F, D, K, N, xt_r2_waves(N, D, K)>;
        XT_R2_NP(1)
        XT_R2_NP(2)
        XT_R2_NP(3)
        XT_R2_NP(4)
        XT_R2_NP(5)
        XT_R2_NP(6)
        XT_R2_NP(7)
        XT_R2_NP(8)
#undef XT_R2_NP
    }
    return nullptr;
}

// Kernel address for (dims, loc.-error dims, directions per pass) at frame_len XT_R2_F; NP = 0: the likelihood-only kernel.
#define XT_R2_CAT_(a, b) a##b
#define XT_R2_CAT(a, b) XT_R2_CAT_(a, b)
const void* XT_R2_CAT(xt_r2_kernel_f, XT_R2_F)(int D, int K, int NP)
{
#define XT_R2_DK(DD, KK)                                                       \
    if (D == DD && K == KK) {                                                  \
        if (NP == 0) return (const void*)xt_ll_r2_kernel<XT_R2_F, DD, KK>;    \
        return r2_grad_np<XT_R2_F, DD, KK>(NP);                                \
    }
    XT_R2_DK(1, 1)
    XT_R2_DK(2, 1)
    XT_R2_DK(2, 2)
    XT_R2_DK(3, 1)
    XT_R2_DK(3, 3)
#undef XT_R2_DK
    return nullptr;
}
