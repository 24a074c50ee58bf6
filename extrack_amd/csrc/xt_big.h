// Fixed-window track likelihood / state posteriors for models whose sequence state does not fit a workgroup (round 4): more than 1024
// groups per track or more than 160 KB of LDS - 5 states at the reference's default frame_len 6, 6 states at frame_len 5 - 6, 2 states at
// frame_len 11 - 15, posteriors with 7 / 8 states.  Same quantity as xt_kernel.h (reference: extrack/tracking.py:109-318
// P_Cs_inter_bound_stats + tracking_0.py:440-458 Proba_Cs; the reference has no size limit, it is just slow: ~1 track/s at 4^10 sequences).
//
// Organisation: ONE LANE PER TRACK.  A track's S^F sequences {zm, ze, m[D], u[K]} live in a per-wavefront region of global memory laid out
// [sequence][field][lane], so every access of a wavefront is one coalesced 512-byte row; a lane walks its groups serially with the same
// fuse -> expand -> integrate step and the same circular digit slots (base_tab / off_tab) as the LDS kernels - groups own disjoint entries,
// so the update is in place and needs no barrier at all.  The kernel streams its state through L2 / HBM once per position (HBM-bound by
// construction: 2 x S^F x 36 B per track and step); it exists so that these models RUN - a dataset of 1e4 tracks of a 5-state model takes
// tens of milliseconds where the LDS kernels refuse and the reference needs hours.
#pragma once
#include "xt_kernel.h"

struct XtBigArgs {
    double* ws;         // scratch: ws_stride doubles per wavefront of the launch
    int64_t ws_stride;
};
// doubles of one wavefront's region: E entries x (1 + D + K) doubles + E ints, 64 lanes each
XT_HD int64_t xt_big_ws_doubles(int E, int D, int K) { return (int64_t)E * 64 * (1 + D + K) + ((int64_t)E * 64 + 1) / 2; }

template <int D, int K, bool PREDS, class Ctx>
XT_HD void xt_big_body(const XtKernelArgs& a, const XtBigArgs& ba, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    const int S = a.S, G = a.G, E = a.E, NG = a.NG, L = b.L, F = a.F;
    const int tid = cx.tid(), nt = cx.nthreads(), x = tid & 63, wv = tid >> 6, NW = nt >> 6;
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += nt) smem[i] = xt_blob_ptr(a)[i];
    cx.sync();
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;
    double* red = smem + ((ntab + 1) & ~1);  // [nt] per-thread sums of LL (block partial)

    double* wsw = ba.ws + ((int64_t)cx.block() * NW + wv) * ba.ws_stride;
    double* zm = wsw;                                   // [E][64]
    double* mm = zm + (int64_t)E * 64;                  // [D][E][64]
    double* uu = mm + (int64_t)D * E * 64;              // [K][E][64]
    int* ze = (int*)(uu + (int64_t)K * E * 64);         // [E][64]
    auto at = [&](int64_t i) XT_INL { return i * 64 + x; };

    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int64_t nbatch = (b.N + 63) / 64;
    double my_ll = 0.0;
    for (int64_t batch = (int64_t)lb * NW + wv; batch < nbatch; batch += (int64_t)nb * NW) {
        const int64_t trk0 = batch * 64 + x;
        const bool act = trk0 < b.N;
        const int64_t trk = act ? trk0 : b.N - 1;  // idle lanes shadow the last track, they write no result
        const double* c = b.tracks + trk * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + trk * (int64_t)L * a.KS : nullptr;
        bool bad = false;  // a NaN position / error poisons the track, as in the reference
        auto load_pos = [&](int p, double* cv) XT_INL {
            for (int d = 0; d < D; ++d) {
                cv[d] = c[p * D + d];
                bad = bad || cv[d] != cv[d];
            }
        };
        auto load_l2 = [&](int p, double* l2) XT_INL {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k) {
                    double s = sg[p * a.KS + (a.KS == 1 ? 0 : k)];
                    bad = bad || s != s;
                    if (a.locerr_mode == 2) {
                        s = xt_fma(s, hdr[3], hdr[4]);
                        s = s < 1e-6 ? 1e-6 : s;
                    }
                    l2[k] = s * s;
                }
            }
        };
        // ---- position 0: one digit (initial state) in slot 0, everything else zero weight
        {
            double c0[D], l20[K];
            load_pos(0, c0);
            load_l2(0, l20);
            for (int il = 0; il < E; ++il) {
                const bool live = il < S;
                zm[at(il)] = live ? hdr[8 + il] : 0.0;
                ze[at(il)] = live ? 0 : XT_EMIN;
                for (int d = 0; d < D; ++d) mm[at((int64_t)d * E + il)] = c0[d];
                for (int k = 0; k < K; ++k) uu[at((int64_t)k * E + il)] = l20[k];
            }
        }
        // ---- positions 1 .. L-2: fuse the group, expand by the new digits, integrate position t
        for (int t = 1; t <= L - 2; ++t) {
            const int ph = (t - 1) % a.P;
            const bool do_pred = PREDS && t >= F;
            const bool stay = t >= stay_from;
            const int32_t* off = a.off_tab + ph * G;
            double ct[D], l2t[K];
            load_pos(t, ct);
            load_l2(t, l2t);
            XtAcc pa[XT_MAX_STATES];  // posterior of the digit about to be fused away (G == S when predicting)
            if (do_pred)
                for (int s = 0; s < S; ++s) pa[s].clear();
            for (int g = 0; g < NG; ++g) {
                const int prev = g / a.prev_div;
                const double* TTl = TAB + ((stay ? 1 : 0) * S + prev) * G;
                const double* TD2 = TAB + (4 * S + prev) * G;
                const int base = a.base_tab[ph * NG + g];
                int emax = XT_EMIN;
                for (int q = 0; q < G; ++q) {
                    const int e = ze[at(base + off[q])];
                    emax = e > emax ? e : emax;
                }
                double W = 0.0, mb[D], ub[K];
                for (int d = 0; d < D; ++d) mb[d] = 0.0;
                for (int k = 0; k < K; ++k) ub[k] = 0.0;
                for (int q = 0; q < G; ++q) {
                    const int idx = base + off[q];
                    const double aq = xt_ldexp(zm[at(idx)], ze[at(idx)] - emax);
                    W += aq;
                    for (int d = 0; d < D; ++d) mb[d] = xt_fma(aq, mm[at((int64_t)d * E + idx)], mb[d]);
                    for (int k = 0; k < K; ++k) ub[k] = xt_fma(aq, uu[at((int64_t)k * E + idx)], ub[k]);
                }
                if (do_pred) {
                    // weighted by the predictive density of position t (tracking.py:255-271; the reference's missing 1/2 on the log term)
                    for (int Q = 0; Q < G; ++Q) {
                        const int idx = base + off[Q];
                        const double zq = zm[at(idx)];
                        if (zq == 0.0) continue;
                        double dq[D], uq[K], dsq = 0.0;
                        for (int d = 0; d < D; ++d) {
                            dq[d] = ct[d] - mm[at((int64_t)d * E + idx)];
                            dsq = xt_fma(dq[d], dq[d], dsq);
                        }
                        for (int k = 0; k < K; ++k) uq[k] = uu[at((int64_t)k * E + idx)];
                        for (int q = 0; q < G; ++q) {
                            double quad, gf;
                            if (K == 1) {
                                const double r = xt_rcp_fast(TD2[q] + uq[0] + l2t[0]);
                                quad = 0.5 * dsq * r;
                                gf = r;
                                for (int d = 1; d < D; ++d) gf *= r;
                            } else {
                                quad = 0.0;
                                gf = 1.0;
                                for (int d = 0; d < D; ++d) {
                                    const double r = xt_rcp_fast(TD2[q] + uq[d] + l2t[d]);
                                    quad = xt_fma(0.5 * dq[d] * dq[d], r, quad);
                                    gf *= r;
                                }
                            }
                            double p;
                            int j, n;
                            xt_exp_tab_fast(-quad, p, j, n);
                            pa[Q].add(zq * TTl[q] * (gf * T64[j]) * p, ze[at(idx)] + n);
                        }
                    }
                }
                const double rW = W > 0.0 ? xt_rcp(W) : 0.0;
                for (int d = 0; d < D; ++d) mb[d] *= rW;
                for (int k = 0; k < K; ++k) ub[k] *= rW;
                const double Wm = xt_frexp_mant(W);
                const int We = W > 0.0 ? emax + xt_frexp_exp(W) : XT_EMIN;
                double dm[D], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dm[d] = ct[d] - mb[d];
                    dsq = xt_fma(dm[d], dm[d], dsq);
                }
                for (int q = 0; q < G; ++q) {
                    const int idx = base + off[q];
                    const double d2 = TD2[q];
                    double quad, gf, tt[K];
                    if (K == 1) {
                        const double s2 = d2 + ub[0];
                        const double r = xt_rcp(l2t[0] + s2);
                        tt[0] = s2 * r;
                        quad = 0.5 * dsq * r;
                        gf = xt_pow_half<D>(r);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        for (int d = 0; d < D; ++d) {
                            const double s2 = d2 + ub[d];
                            const double r = xt_rcp(l2t[d] + s2);
                            tt[d] = s2 * r;
                            quad = xt_fma(0.5 * dm[d] * dm[d], r, quad);
                            gf *= r;
                        }
                        gf = sqrt(gf);
                    }
                    double p;
                    int j, n;
                    xt_exp_tab(-quad, p, j, n);
                    const int en = We + n;
                    zm[at(idx)] = (Wm * TTl[q]) * (gf * T64[j]) * p;
                    ze[at(idx)] = en > XT_EMIN ? en : XT_EMIN;
                    for (int d = 0; d < D; ++d) mm[at((int64_t)d * E + idx)] = xt_fma(dm[d], tt[K == 1 ? 0 : d], mb[d]);
                    for (int k = 0; k < K; ++k) uu[at((int64_t)k * E + idx)] = l2t[k] * tt[k];
                }
            }
            if (do_pred && act) {
                int pem = XT_EMIN;
                for (int s = 0; s < S; ++s) pem = pa[s].e > pem ? pa[s].e : pem;
                double v[XT_MAX_STATES], tots = 0.0;
                for (int s = 0; s < S; ++s) {
                    v[s] = pa[s].m != 0.0 ? xt_ldexp(pa[s].m, pa[s].e - pem) : 0.0;
                    tots += v[s];
                }
                for (int s = 0; s < S; ++s) b.preds_out[(trk * L + (t - F)) * S + s] = v[s] / tots;
            }
        }
        // ---- last position (+ leaving / bleaching term): pure reduction over (old entry Q, new digits q)
        const int tl = L - 1;
        XtAcc tot;
        tot.clear();
        XtAcc facc[PREDS ? (16 * XT_MAX_STATES) : 1];  // posterior columns 0 .. F of the final read-out, [column][state]
        if (PREDS)
            for (int i = 0; i < (F + 1) * S; ++i) facc[i].clear();
        {
            const int ph = (tl - 1) % a.P;
            const int32_t* off = a.off_tab + ph * G;
            const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
            double cl[D], l2l[K];
            load_pos(tl, cl);
            load_l2(tl, l2l);
            for (int g = 0; g < NG; ++g) {
                const int prev = g / a.prev_div;
                const double* TF = TAB + (vfin * S + prev) * G;
                const double* TD2 = TAB + (4 * S + prev) * G;
                const int base = a.base_tab[ph * NG + g];
                XtAcc tg;  // this group's total: weights the digits of g in columns 1 .. F-1
                tg.clear();
                for (int Q = 0; Q < G; ++Q) {
                    const int idx = base + off[Q];
                    const double zq = zm[at(idx)];
                    if (zq == 0.0) continue;
                    const int eq = ze[at(idx)];
                    double dq[D], uq[K], dsq = 0.0;
                    for (int d = 0; d < D; ++d) {
                        dq[d] = cl[d] - mm[at((int64_t)d * E + idx)];
                        dsq = xt_fma(dq[d], dq[d], dsq);
                    }
                    for (int k = 0; k < K; ++k) uq[k] = uu[at((int64_t)k * E + idx)];
                    for (int q = 0; q < G; ++q) {
                        double quad, gf;
                        if (K == 1) {
                            const double r = xt_rcp(TD2[q] + uq[0] + l2l[0]);
                            quad = 0.5 * dsq * r;
                            gf = xt_pow_half<D>(r);
                        } else {
                            quad = 0.0;
                            gf = 1.0;
                            for (int d = 0; d < D; ++d) {
                                const double r = xt_rcp(TD2[q] + uq[d] + l2l[d]);
                                quad = xt_fma(0.5 * dq[d] * dq[d], r, quad);
                                gf *= r;
                            }
                            gf = sqrt(gf);
                        }
                        double p;
                        int j, n;
                        xt_exp_tab(-quad, p, j, n);
                        const double wm = zq * TF[q] * (gf * T64[j]) * p;
                        const int we = eq + n;
                        tg.add(wm, we);
                        if (PREDS) {
                            facc[0 * S + q].add(wm, we);                    // column 0: the newest digit
                            if (L - 1 >= F) facc[F * S + Q].add(wm, we);    // column F: the digit in the fused slot
                        }
                    }
                }
                tot.add(tg.m, tg.e);
                if (PREDS && tg.m != 0.0)
                    for (int j = 1; j <= F - 1 && j <= L - 1; ++j) facc[j * S + (g / a.pw[F - j - 1]) % S].add(tg.m, tg.e);  // columns 1 .. F-1: digits of g
            }
        }
        const double ll = bad ? NAN : log(tot.m) + (double)tot.e * XT_LN2 + b.ll_const;
        if (act) {
            if (b.ll_out) b.ll_out[trk] = ll;
            my_ll += ll;
            if (PREDS) {
                const int ncol = (L - 1 < F ? L - 1 : F) + 1;
                for (int j = 0; j < ncol; ++j) {
                    int em = XT_EMIN;
                    for (int s = 0; s < S; ++s) em = facc[j * S + s].e > em ? facc[j * S + s].e : em;
                    double v[XT_MAX_STATES], tots = 0.0;
                    for (int s = 0; s < S; ++s) {
                        v[s] = facc[j * S + s].m != 0.0 ? xt_ldexp(facc[j * S + s].m, facc[j * S + s].e - em) : 0.0;
                        tots += v[s];
                    }
                    for (int s = 0; s < S; ++s) b.preds_out[(trk * L + (L - 1 - j)) * S + s] = v[s] / tots;
                }
                if (bad)
                    for (int i = 0; i < L * S; ++i) b.preds_out[trk * L * S + i] = NAN;
            }
        }
    }
    // ---- block partial: fixed-order sum over the block's lanes
    red[tid] = my_ll;
    cx.sync();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < nt; ++i) s += red[i];
        a.partials[cx.block()] = s;
    }
}
