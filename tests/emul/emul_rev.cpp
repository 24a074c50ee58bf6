// TEST INFRASTRUCTURE ONLY: the reverse-mode gradient body of extrack_amd/csrc/xt_rev.h on CPU threads.
#include "emul_ctx.h"
#include "../../extrack_amd/csrc/xt_rev.h"

template <int G_, int D, int K>
static void run_rev(const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t lds_doubles)
{
    th_emul_blocks(nblocks, threads, lds_doubles + 8, [&](HostCtx& cx) { xt_rev_body<G_, D, K>(a, ra, cx); });
}
template <int G_>
static bool rev_dk(int D, int K, const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t ldsd)
{
    if (D == 1 && K == 1) return run_rev<G_, 1, 1>(a, ra, nblocks, threads, ldsd), true;
    if (D == 2 && K == 1) return run_rev<G_, 2, 1>(a, ra, nblocks, threads, ldsd), true;
    if (D == 2 && K == 2) return run_rev<G_, 2, 2>(a, ra, nblocks, threads, ldsd), true;
    if (D == 3 && K == 1) return run_rev<G_, 3, 1>(a, ra, nblocks, threads, ldsd), true;
    if (D == 3 && K == 3) return run_rev<G_, 3, 3>(a, ra, nblocks, threads, ldsd), true;
    return false;
}
bool emul_rev(int G, int D, int K, const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t lds_doubles)
{
    if (G == 2) return rev_dk<2>(D, K, a, ra, nblocks, threads, lds_doubles);
    if (G == 3) return rev_dk<3>(D, K, a, ra, nblocks, threads, lds_doubles);
    if (G == 4) return rev_dk<4>(D, K, a, ra, nblocks, threads, lds_doubles);
    return false;
}
