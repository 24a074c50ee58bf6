// TEST INFRASTRUCTURE ONLY: the register-resident 2-state body (extrack_amd/csrc/xt_reg2.h) on CPU threads; its own translation
// unit because every (frame_len, dims, directions) instance unrolls the whole step loop.
#include "emul_ctx.h"
#include "../../extrack_amd/csrc/xt_reg2.h"

// ---- NP directions per pass are compile-time: 0 (likelihood only), 3 and 8 are instantiated here; the caller pads with zero directions
template <int F, int D, int K, int NP>
static void emul_r2_run(const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, size_t lds_bytes)
{
    th_emul_blocks(nblocks, 64 * XT_F2_WAVES, lds_bytes / 8 + 8, [&](HostCtx& cx) { xt_r2_body<F, D, K, NP>(a, ga, cx); });
}
template <int F, int D, int K>
static bool emul_r2_np(int NP, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, size_t lds_bytes)
{
    switch (NP) {
        case 0: emul_r2_run<F, D, K, 0>(a, ga, nblocks, lds_bytes); return true;
        case 3: emul_r2_run<F, D, K, 3>(a, ga, nblocks, lds_bytes); return true;
        case 8: emul_r2_run<F, D, K, 8>(a, ga, nblocks, lds_bytes); return true;
    }
    return false;
}
template <int F>
static bool emul_r2_dk(int D, int K, int NP, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, size_t lds_bytes)
{
    if (D == 1 && K == 1) return emul_r2_np<F, 1, 1>(NP, a, ga, nblocks, lds_bytes);
    if (D == 2 && K == 1) return emul_r2_np<F, 2, 1>(NP, a, ga, nblocks, lds_bytes);
    if (D == 2 && K == 2) return emul_r2_np<F, 2, 2>(NP, a, ga, nblocks, lds_bytes);
    if (D == 3 && K == 1) return emul_r2_np<F, 3, 1>(NP, a, ga, nblocks, lds_bytes);
    if (D == 3 && K == 3) return emul_r2_np<F, 3, 3>(NP, a, ga, nblocks, lds_bytes);
    return false;
}
bool emul_r2(int F, int D, int K, int KS, int NP, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks)
{
    const size_t lds = (size_t)xt_r2_block_bytes(NP + (NP ? ga.NU : 0), D, KS, 64 >> (F - 1));
    if (F == 4) return emul_r2_dk<4>(D, K, NP, a, ga, nblocks, lds);
    if (F == 5) return emul_r2_dk<5>(D, K, NP, a, ga, nblocks, lds);
    if (F == 6) return emul_r2_dk<6>(D, K, NP, a, ga, nblocks, lds);
    if (F == 7) return emul_r2_dk<7>(D, K, NP, a, ga, nblocks, lds);
    return false;
}

