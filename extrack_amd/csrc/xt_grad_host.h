// Host-side table construction for the gradient kernels (plain C++, no HIP): the tangent of every model table of
// xt_build_blob (xt_tables.h), obtained by running the same construction in dual numbers.
#pragma once
#include <vector>

#include "../../include/extrack_hip.h"
#include "xt_grad.h"
#include "xt_tables.h"

struct XtDual {
    double v, d;
    XtDual(double v_ = 0.0, double d_ = 0.0) : v(v_), d(d_) {}
};
static inline XtDual operator+(XtDual a, XtDual b) { return XtDual(a.v + b.v, a.d + b.d); }
static inline XtDual operator-(XtDual a, XtDual b) { return XtDual(a.v - b.v, a.d - b.d); }
static inline XtDual operator*(XtDual a, XtDual b) { return XtDual(a.v * b.v, a.d * b.v + a.v * b.d); }
static inline XtDual operator/(XtDual a, double s) { return XtDual(a.v / s, a.d / s); }
static inline double xt_dlog(XtDual a) { return a.v != 0.0 ? a.d / a.v : 0.0; }

// One direction's tangent block (layout: xt_grad.h).  `m` is the primal model, `t` its tangent.
static void xt_build_tangent_block(const XtModelHost& m, const extrack_model_tangent& t, const XtConfig& c, int locerr_mode, double* tb)
{
    const int S = c.S, NS = c.NS, G = c.G;
    const int TBn = xt_grad_tb_doubles(S, G);
    for (int i = 0; i < TBn; ++i) tb[i] = 0.0;
    if (locerr_mode == 0)
        for (int k = 0; k < 3; ++k) {
            const int kk = k < m.locerr_dims ? k : 0;
            tb[k] = 2.0 * m.locerr[kk] * t.locerr[kk];
        }
    tb[3] = t.slope;
    tb[4] = t.offset;
    for (int s = 0; s < S; ++s) tb[8 + s] = m.Fs[s] != 0.0 ? t.Fs[s] / m.Fs[s] : 0.0;
    const XtDual pBL(m.pBL, t.pBL), one(1.0, 0.0);
    std::vector<XtDual> T((size_t)S * S), ds2(S), pst(G);
    for (int i = 0; i < S * S; ++i) T[i] = XtDual(m.TrMat[i], t.TrMat[i]);
    for (int s = 0; s < S; ++s) ds2[s] = XtDual(m.ds[s] * m.ds[s], t.ds2[s]);
    for (int r = 0; r < G; ++r) pst[r] = XtDual(m.p_stay[r], t.p_stay[r]);
    // Eend = T^NS qq,  qq[s] = pBL + (1 - p_stay[s]) - pBL (1 - p_stay[s])   (p_stay indexed by the raw state: tracking.py:297)
    std::vector<XtDual> v(S), w(S);
    for (int s = 0; s < S; ++s) {
        const XtDual q1 = one - pst[s];
        v[s] = pBL + q1 - pBL * q1;
    }
    for (int it = 0; it < NS; ++it) {
        for (int i = 0; i < S; ++i) {
            XtDual acc;
            for (int j = 0; j < S; ++j) acc = acc + T[i * S + j] * v[j];
            w[i] = acc;
        }
        v = w;
    }
    double* TAB = tb + XT_BLOB_HDR;
    const size_t SG = (size_t)S * G;
    for (int prev = 0; prev < S; ++prev)
        for (int q = 0; q < G; ++q) {
            int chain[8];
            chain[0] = prev;
            int r = q;
            for (int j = 1; j <= NS; ++j) {
                chain[j] = r % S;
                r /= S;
            }
            XtDual tp = one, d2;
            for (int j = 0; j < NS; ++j) {
                tp = tp * T[chain[j] * S + chain[j + 1]];
                d2 = d2 + (ds2[chain[j]] + ds2[chain[j + 1]]) / 2.0;
            }
            d2 = d2 / (double)NS;
            int rref = 0;
            for (int cc = 0; cc < NS; ++cc) rref += chain[NS - cc] * c.pw[cc];
            const XtDual stay = pst[rref] * (one - pBL);
            const XtDual ee = v[chain[NS]];
            const size_t o = (size_t)prev * G + q;
            TAB[0 * SG + o] = xt_dlog(tp);
            TAB[1 * SG + o] = xt_dlog(tp * stay);
            TAB[2 * SG + o] = xt_dlog(tp * ee);
            TAB[3 * SG + o] = xt_dlog(tp * stay * ee);
            TAB[4 * SG + o] = d2.d;
        }
}

// The same for the model tables of the THRESHOLD-FUSION kernels (xt_th_build_blob, xt_tables.h): second table index r in the reference's
// digit order (digit c of r = c-th newest sub-state), stay term p_stay[r], end term indexed by the newest sub-state.  Layout of the block
// as above (XT_BLOB_HDR + XT_NTAB * S * G doubles): the frozen-plan gradient kernel (xt_thgrad.h) returns the adjoint in that layout.
static void xt_th_build_tangent_block(const XtModelHost& m, const extrack_model_tangent& t, int locerr_mode, double* tb)
{
    const int S = m.S, NS = m.NS;
    int G = 1;
    for (int i = 0; i < NS; ++i) G *= S;
    const int TBn = xt_grad_tb_doubles(S, G);
    for (int i = 0; i < TBn; ++i) tb[i] = 0.0;
    if (locerr_mode == 0)
        for (int k = 0; k < 3; ++k) {
            const int kk = k < m.locerr_dims ? k : 0;
            tb[k] = 2.0 * m.locerr[kk] * t.locerr[kk];
        }
    tb[3] = t.slope;
    tb[4] = t.offset;
    for (int s = 0; s < S; ++s) tb[8 + s] = m.Fs[s] != 0.0 ? t.Fs[s] / m.Fs[s] : 0.0;
    const XtDual pBL(m.pBL, t.pBL), one(1.0, 0.0);
    std::vector<XtDual> T((size_t)S * S), ds2(S), pst(G);
    for (int i = 0; i < S * S; ++i) T[i] = XtDual(m.TrMat[i], t.TrMat[i]);
    for (int s = 0; s < S; ++s) ds2[s] = XtDual(m.ds[s] * m.ds[s], t.ds2[s]);
    for (int r = 0; r < G; ++r) pst[r] = XtDual(m.p_stay[r], t.p_stay[r]);
    std::vector<XtDual> v(S), w(S);
    for (int s = 0; s < S; ++s) {
        const XtDual q1 = one - pst[s];  // raw newest state as index (reference quirk, tracking.py:624)
        v[s] = pBL + q1 - pBL * q1;
    }
    for (int it = 0; it < NS; ++it) {
        for (int i = 0; i < S; ++i) {
            XtDual acc;
            for (int j = 0; j < S; ++j) acc = acc + T[i * S + j] * v[j];
            w[i] = acc;
        }
        v = w;
    }
    double* TAB = tb + XT_BLOB_HDR;
    const size_t SG = (size_t)S * G;
    for (int prev = 0; prev < S; ++prev)
        for (int r = 0; r < G; ++r) {
            int dig[8], rr = r;
            for (int c = 0; c < NS; ++c) {
                dig[c] = rr % S;
                rr /= S;
            }
            dig[NS] = prev;
            XtDual tp = one, d2;
            for (int c = 0; c < NS; ++c) {
                tp = tp * T[dig[c + 1] * S + dig[c]];
                d2 = d2 + (ds2[dig[c]] + ds2[dig[c + 1]]) / 2.0;
            }
            d2 = d2 / (double)NS;
            const XtDual stay = pst[r] * (one - pBL);
            const XtDual ee = v[dig[0]];
            const size_t o = (size_t)prev * G + r;
            TAB[0 * SG + o] = xt_dlog(tp);
            TAB[1 * SG + o] = xt_dlog(tp * stay);
            TAB[2 * SG + o] = xt_dlog(tp * ee);
            TAB[3 * SG + o] = xt_dlog(tp * stay * ee);
            TAB[4 * SG + o] = d2.d;
        }
}


// Register-resident 2-state kernels (xt_reg2.h): is this direction "uniform", i.e. does every weight factor of a step change by the SAME
// relative amount for all sequences (no change of the localisation / diffusion variances, equal d log of the initial fractions, constant
// d log T and d log (T * stay) tables)?  Then its tangent needs no per-step work: rz is the same number for every sequence, dm = du = 0.
// Typical case: the bleaching probability pBL.  tb: the direction's tangent block (xt_build_tangent_block) of a 2-state, 1-substep model.
static inline bool xt_r2_uniform_direction(const double* tb)
{
    // "equal": to 1e-13 relative (the entries are quotients of products that differ in their rounding, not in their value)
    auto same = [](double a, double b) { return fabs(a - b) <= 1e-13 * (fabs(a) > fabs(b) ? fabs(a) : fabs(b)); };
    for (int k = 0; k < 5; ++k)
        if (tb[k] != 0.0) return false;          // d l2, d slope, d offset
    if (!same(tb[8], tb[9])) return false;       // d log Fs
    const double* T = tb + XT_BLOB_HDR;
    for (int v = 0; v < 2; ++v)
        for (int i = 1; i < 4; ++i)
            if (!same(T[v * 4 + i], T[v * 4])) return false;  // d log T, d log (T * stay): constant over [prev][q]
    for (int i = 0; i < 4; ++i)
        if (T[4 * 4 + i] != 0.0) return false;   // d d2
    return true;
}
