// Host side of P_Cs_inter_bound_stats(return_matrix=True): the reference returns the log-probability of EVERY sequence of states still
// distinguished at the last position, LP[N, nB], and the digit matrix cur_Bs (extrack/tracking.py:300-318).  The kernels never hold
// that matrix - they reduce it in place - but the general kernel (xt_kernel.h) can write the log-weight of every
// (stored sequence, new digits) pair of its last position; this header maps that raw output to the reference's column order
// (digit c of column i = (i / S^c) % S, c = 0 the newest state; get_all_Bs, tracking.py:746-757) and adds the leaving / bleaching
// term of isBL tracks, which the reference expands into nb_substeps more digits (tracking.py:282-299).  Plain C++, no HIP: shared
// with the CPU-thread emulator of the tests.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

#include "xt_tables.h"

// digits of a sequence at the last position of a track of L positions before the isBL expansion (the reference fuses down to
// frame_len digits after every position but the last, tracking.py:253-277)
static inline int xt_seq_digits(int L, int NS, int F)
{
    int n = NS + 1;
    for (int t = 2; t < L; ++t) {
        n += NS;
        if (t < L - 1 && n > F) n = F;
    }
    return n;
}
static inline int64_t xt_ipow(int S, int n)
{
    int64_t r = 1;
    for (int i = 0; i < n; ++i) r *= S;
    return r;
}
// columns of the reference's LP for a track of L positions
static inline int64_t xt_seq_columns(int S, int L, int NS, int F, int isBL) { return xt_ipow(S, xt_seq_digits(L, NS, F) + (isBL ? NS : 0)); }

// raw [N][E][G] (kernel order: stored sequence = circular digit slots, xt_kernel.h; new digits q, first substep in the lowest digit)
// -> lp [N][nB] in the reference's order.  The raw launch ran WITHOUT the leaving term (isBL = 0 tables); it is added here.
static inline void xt_seq_reorder(const XtConfig& c, const XtModelHost& m, int64_t N, int L, int isBL, const double* raw, double* lp)
{
    const int S = c.S, NS = c.NS, F = c.F, G = c.G;
    const int n = xt_seq_digits(L, NS, F);
    const int64_t nb0 = xt_ipow(S, n), nb = isBL ? nb0 * G : nb0;
    const int tl = L - 1;
    const int h = (1 + (tl - 1) * NS) % F;  // first slot the (virtual) step of the last position would overwrite: holds the oldest digits
    // column i0 of the pre-expansion matrix -> (entry, q)
    std::vector<int64_t> src((size_t)nb0);
    for (int64_t i0 = 0; i0 < nb0; ++i0) {
        int64_t r = i0;
        int q = 0, entry = 0;
        for (int k = 0; k < n; ++k) {
            const int dg = (int)(r % S);
            r /= S;
            if (k < NS) {
                q += dg * c.pw[NS - 1 - k];  // newest state = last substep = highest digit of q
            } else {
                const int a = k - NS;        // age rank among the stored digits, 0 = newest
                entry += dg * c.pw[((h + F - 1 - a) % F + F) % F];
            }
        }
        src[(size_t)i0] = (int64_t)entry * G + q;
    }
    // leaving / bleaching term over the NS extra digits e (newest) + the newest digit of the sequence
    std::vector<double> LLe;
    if (isBL) {
        LLe.assign((size_t)G * S, 0.0);  // index e + G * newest
        for (int j = 0; j < G * S; ++j) {
            int dig[8], r = j;
            for (int cdx = 0; cdx <= NS; ++cdx) {
                dig[cdx] = r % S;
                r /= S;
            }
            double lt = 0.0;
            for (int cdx = 0; cdx < NS; ++cdx) lt += log(m.TrMat[dig[cdx + 1] * S + dig[cdx]]);
            const double ps = m.p_stay[dig[0]];  // indexed by the raw newest state (reference quirk, tracking.py:297)
            LLe[(size_t)j] = log(m.pBL + (1.0 - ps) - m.pBL * (1.0 - ps)) + lt;
        }
    }
    const int64_t EG = (int64_t)c.E * G;
    for (int64_t t = 0; t < N; ++t) {
        const double* rw = raw + t * EG;
        double* o = lp + t * nb;
        if (!isBL) {
            for (int64_t i0 = 0; i0 < nb0; ++i0) o[i0] = rw[src[(size_t)i0]];
        } else {
            for (int64_t i0 = 0; i0 < nb0; ++i0) {
                const double v = rw[src[(size_t)i0]];
                const int newest = (int)(i0 % S);
                for (int e = 0; e < G; ++e) o[i0 * G + e] = v + LLe[(size_t)(e + G * newest)];
            }
        }
    }
}
