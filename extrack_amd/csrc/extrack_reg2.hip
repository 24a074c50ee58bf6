// libextrack_hip.so: lookup of the register-resident 2-state kernels (xt_reg2.h), instantiated per frame_len in extrack_reg2_f{4..7}.hip.
// The launch sites live in extrack_hip.hip (extrack_loglik*) and extrack_grad.hip (extrack_loglik_grad).
const void* xt_r2_kernel_f4(int D, int K, int NP);
const void* xt_r2_kernel_f5(int D, int K, int NP);
const void* xt_r2_kernel_f6(int D, int K, int NP);
const void* xt_r2_kernel_f7(int D, int K, int NP);

// Kernel address for (frame_len, dims, loc.-error dims, directions per pass); NP = 0: the likelihood-only kernel.  nullptr: not built.
const void* xt_r2_kernel(int F, int D, int K, int NP)
{
    if (F == 4) return xt_r2_kernel_f4(D, K, NP);
    if (F == 5) return xt_r2_kernel_f5(D, K, NP);
    if (F == 6) return xt_r2_kernel_f6(D, K, NP);
    if (F == 7) return xt_r2_kernel_f7(D, K, NP);
    return nullptr;
}
