// TEST INFRASTRUCTURE ONLY: the reverse-mode gradient body of extrack_amd/csrc/xt_rev.h on CPU threads.
#include "emul_ctx.h"
#include "../../extrack_amd/csrc/xt_rev.h"

template <int G_, int D, int K, int NBUF>
static void run_rev(const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t lds_doubles)
{
    th_emul_blocks(nblocks, threads, lds_doubles + 8, [&](HostCtx& cx) { xt_rev_body<G_, D, K, NBUF>(a, ra, cx); });
}
template <int G_, int NBUF>
static bool rev_dk(int D, int K, const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t ldsd)
{
    if (D == 1 && K == 1) return run_rev<G_, 1, 1, NBUF>(a, ra, nblocks, threads, ldsd), true;
    if (D == 2 && K == 1) return run_rev<G_, 2, 1, NBUF>(a, ra, nblocks, threads, ldsd), true;
    if (D == 2 && K == 2) return run_rev<G_, 2, 2, NBUF>(a, ra, nblocks, threads, ldsd), true;
    if (D == 3 && K == 1) return run_rev<G_, 3, 1, NBUF>(a, ra, nblocks, threads, ldsd), true;
    if (D == 3 && K == 3) return run_rev<G_, 3, 3, NBUF>(a, ra, nblocks, threads, ldsd), true;
    return false;
}
// nbuf: exchange buffers per track (2: one barrier per step, 1: two barriers per step)
bool emul_rev(int G, int D, int K, int nbuf, const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t lds_doubles)
{
    if (nbuf == 2) {
        if (G == 2) return rev_dk<2, 2>(D, K, a, ra, nblocks, threads, lds_doubles);
        if (G == 3) return rev_dk<3, 2>(D, K, a, ra, nblocks, threads, lds_doubles);
        if (G == 4) return rev_dk<4, 2>(D, K, a, ra, nblocks, threads, lds_doubles);
    } else if (nbuf == 1) {
        if (G == 2) return rev_dk<2, 1>(D, K, a, ra, nblocks, threads, lds_doubles);
        if (G == 3) return rev_dk<3, 1>(D, K, a, ra, nblocks, threads, lds_doubles);
        if (G == 4) return rev_dk<4, 1>(D, K, a, ra, nblocks, threads, lds_doubles);
    }
    return false;
}
