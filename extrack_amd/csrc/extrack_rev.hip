// libextrack_hip.so, translation unit: likelihood + gradient by reverse-mode differentiation (xt_rev.h), for models with 2, 3 or 4
// members per group.  Launched from extrack_grad.hip (extrack_loglik_grad).
#include "xt_host.h"

#include "xt_rev.h"

#ifndef XT_REV_WAVES
#define XT_REV_WAVES 2
#endif
template <int G_, int D, int K, int NBUF>
__global__ void __launch_bounds__(256, XT_REV_WAVES) xt_rev_kernel(XtKernelArgs a, XtRevArgs ra)
{
    DevCtx cx;
    xt_rev_body<G_, D, K, NBUF>(a, ra, cx);
}

template <int G_, int NBUF>
static const void* rev_dk(int D, int K)
{
    if (D == 1 && K == 1) return (const void*)xt_rev_kernel<G_, 1, 1, NBUF>;
    if (D == 2 && K == 1) return (const void*)xt_rev_kernel<G_, 2, 1, NBUF>;
    if (D == 2 && K == 2) return (const void*)xt_rev_kernel<G_, 2, 2, NBUF>;
    if (D == 3 && K == 1) return (const void*)xt_rev_kernel<G_, 3, 1, NBUF>;
    if (D == 3 && K == 3) return (const void*)xt_rev_kernel<G_, 3, 3, NBUF>;
    return nullptr;
}

// Kernel address for (members per group, dims, loc.-error dims, exchange buffers per track: 1 | 2); nullptr: not built.
const void* xt_rev_kernel_ptr(int G, int D, int K, int nbuf)
{
    if (nbuf == 2) {
        if (G == 2) return rev_dk<2, 2>(D, K);
        if (G == 3) return rev_dk<3, 2>(D, K);
        if (G == 4) return rev_dk<4, 2>(D, K);
    } else if (nbuf == 1) {
        if (G == 2) return rev_dk<2, 1>(D, K);
        if (G == 3) return rev_dk<3, 1>(D, K);
        if (G == 4) return rev_dk<4, 1>(D, K);
    }
    return nullptr;
}

// d sum LL / d theta_i = <adjoint of the model blob, tangent block of direction i>: one workgroup per direction, fixed-order sum.
__global__ void __launch_bounds__(64) xt_rev_project_kernel(const double* __restrict__ adj, const double* __restrict__ dblob, int TB, double* __restrict__ out)
{
    __shared__ double sh[64];
    const double* tb = dblob + (size_t)blockIdx.x * TB;
    double s = 0.0;
    for (int i = threadIdx.x; i < TB; i += 64) s = __builtin_fma(adj[i], tb[i], s);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
void xt_rev_project(hipStream_t st, const double* adj, const double* dblob, int TB, int n_dir, double* out)
{
    hipLaunchKernelGGL(xt_rev_project_kernel, dim3(n_dir), dim3(64), 0, st, adj, dblob, TB, out);
}
