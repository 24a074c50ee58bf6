// libextrack_hip.so, translation unit: likelihood + gradient kernels with the sequence state and its tangents in registers and the LDS as
// the exchange medium (xt_gradr.h), for models with 2, 3 or 4 members per group.  Launched from extrack_grad.hip (extrack_loglik_grad).
#include "xt_host.h"

#include "xt_gradr.h"

template <int G_, int D, int K, int NPC>
__global__ void __launch_bounds__(256, 2) xt_gradr_kernel(XtKernelArgs a, XtGradArgs ga)
{
    DevCtx cx;
    xt_gradr_body<G_, D, K, NPC>(a, ga, cx);
}

template <int G_, int NPC>
static const void* gradr_dk(int D, int K)
{
    if (D == 1 && K == 1) return (const void*)xt_gradr_kernel<G_, 1, 1, NPC>;
    if (D == 2 && K == 1) return (const void*)xt_gradr_kernel<G_, 2, 1, NPC>;
    if (D == 2 && K == 2) return (const void*)xt_gradr_kernel<G_, 2, 2, NPC>;
    if (D == 3 && K == 1) return (const void*)xt_gradr_kernel<G_, 3, 1, NPC>;
    if (D == 3 && K == 3) return (const void*)xt_gradr_kernel<G_, 3, 3, NPC>;
    return nullptr;
}

// Kernel address for (members per group, dims, loc.-error dims, directions per pass: 3 or 4); nullptr: not built.
const void* xt_gradr_kernel_ptr(int G, int D, int K, int NPC)
{
    if (NPC == 4) {
        if (G == 2) return gradr_dk<2, 4>(D, K);
        if (G == 3) return gradr_dk<3, 4>(D, K);
        if (G == 4) return gradr_dk<4, 4>(D, K);
    } else if (NPC == 3) {
        if (G == 2) return gradr_dk<2, 3>(D, K);
        if (G == 3) return gradr_dk<3, 3>(D, K);
        if (G == 4) return gradr_dk<4, 3>(D, K);
    }
    return nullptr;
}
