// TEST INFRASTRUCTURE ONLY: the register-resident gradient body of extrack_amd/csrc/xt_gradr.h on CPU threads.
#include "emul_ctx.h"
#include "../../extrack_amd/csrc/xt_gradr.h"

template <int G_, int D, int K, int NPC>
static void run_gradr(const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, int threads, size_t lds_doubles)
{
    th_emul_blocks(nblocks, threads, lds_doubles + 8, [&](HostCtx& cx) { xt_gradr_body<G_, D, K, NPC>(a, ga, cx); });
}
template <int G_, int NPC>
static bool gradr_dk(int D, int K, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, int threads, size_t ldsd)
{
    if (D == 1 && K == 1) return run_gradr<G_, 1, 1, NPC>(a, ga, nblocks, threads, ldsd), true;
    if (D == 2 && K == 1) return run_gradr<G_, 2, 1, NPC>(a, ga, nblocks, threads, ldsd), true;
    if (D == 2 && K == 2) return run_gradr<G_, 2, 2, NPC>(a, ga, nblocks, threads, ldsd), true;
    if (D == 3 && K == 1) return run_gradr<G_, 3, 1, NPC>(a, ga, nblocks, threads, ldsd), true;
    if (D == 3 && K == 3) return run_gradr<G_, 3, 3, NPC>(a, ga, nblocks, threads, ldsd), true;
    return false;
}
// NPC = 3 or 4 directions per pass (compile-time register arrays), G = 2, 3, 4 members per group
bool emul_gradr(int G, int D, int K, int NPC, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, int threads, size_t lds_doubles)
{
    if (NPC == 4) {
        if (G == 2) return gradr_dk<2, 4>(D, K, a, ga, nblocks, threads, lds_doubles);
        if (G == 3) return gradr_dk<3, 4>(D, K, a, ga, nblocks, threads, lds_doubles);
        if (G == 4) return gradr_dk<4, 4>(D, K, a, ga, nblocks, threads, lds_doubles);
    } else if (NPC == 3) {
        if (G == 2) return gradr_dk<2, 3>(D, K, a, ga, nblocks, threads, lds_doubles);
        if (G == 3) return gradr_dk<3, 3>(D, K, a, ga, nblocks, threads, lds_doubles);
        if (G == 4) return gradr_dk<4, 3>(D, K, a, ga, nblocks, threads, lds_doubles);
    }
    return false;
}
