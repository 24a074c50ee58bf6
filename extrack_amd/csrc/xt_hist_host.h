// Host-side model tables of the histogram kernel (plain C++): layout in xt_hist.h.
#pragma once
#include <math.h>

#include <vector>

#include "xt_hist.h"
#include "xt_tables.h"

static inline void xt_hist_build_blob(const XtModelHost& m, std::vector<double>& blob)
{
    const int S = m.S;
    blob.assign((size_t)xt_hist_blob_doubles(S), 0.0);
    for (int k = 0; k < 3; ++k) {
        const double s = m.locerr[k < m.locerr_dims ? k : 0];
        blob[k] = s * s;
    }
    blob[3] = m.slope;
    blob[4] = m.offset;
    const double qq0 = m.pBL + (1.0 - m.p_stay[0]) - m.pBL * (1.0 - m.p_stay[0]);
    for (int s = 0; s < S; ++s) {
        blob[8 + s] = log(m.Fs[s]);
        blob[16 + s] = log(m.p_stay[s] * (1.0 - m.pBL));                             // histograms.py:129
        const double qq = m.pBL + (1.0 - m.p_stay[s]) - m.pBL * (1.0 - m.p_stay[s]);  // histograms.py:229
        blob[24 + s] = log(qq + (S - 1) * qq0);  // the extra end-of-track state summed out: p_stay[0] whenever it differs (argmax quirk)
        for (int t = 0; t < S; ++t) {
            blob[32 + s * S + t] = log(m.TrMat[s * S + t]);
            blob[32 + S * S + s * S + t] = (m.ds[s] * m.ds[s] + m.ds[t] * m.ds[t]) / 2.0;
        }
    }
}
