// Log-likelihood AND its exact gradient in one pass: forward-mode differentiation carried alongside the fixed-window recursion
// of xt_kernel.h.
//
// What it replaces: the reference fits by lmfit.minimize(cum_Proba_Cs, ...) (extrack/tracking.py:1371); BFGS differentiates
// the objective by finite differences, i.e. nvar + 1 full evaluations per iteration (and the differences of a sum over 1e6
// tracks carry rounding noise of ~1e-10 relative, which is what stops the optimiser).  Here one launch returns
// sum(LL) and d sum(LL) / d theta_i for NP directions theta_i of the MODEL (tangents of ds^2, Fs, TrMat, LocErr, slope/offset,
// pBL, p_stay given by the caller; the chain rule through extract_params and the bounds transform is host work).
//
// The derivative is that of the function the kernels compute - window fusion (moment matching) included - not an
// EM-style expected-count approximation: every quantity of the recursion carries its tangent,
//     z  (weight)   ->  rz = d log z           (relative: needs no extended range)
//     m  (mean)     ->  dm[D]
//     u  (variance) ->  du[K]
// merge (fuse_tracks_general, tracking.py:361-423):   W = sum w_j,  R = sum a_j rz_j  (a_j = w_j / W)
//     m_bar = sum a_j m_j          d m_bar = sum a_j (dm_j + (m_j - m_bar) rz_j)
//     u_bar likewise
// expand + integrate (log_integrale_dif, tracking.py:76-98), per new digit q, den = l2 + d2_q + u_bar, r = 1/den, tt = (d2_q+u_bar) r:
//     d den = d l2 + d d2_q + d u_bar,   d tt = r (d s2 - tt d den)
//     rz'  = R + dlogT[q] - (D/2) r d den - d quad,     quad = |c - m_bar|^2 r / 2
//     dm'  = d m_bar (1 - tt) + (c - m_bar) d tt,       du' = d l2 tt + l2 d tt
// last position: LL = log sum_{Q,q} w_Qq,  dLL = sum w_Qq (rz_Q + dlogTF[q] - (D/2) r d den - d quad) / sum w_Qq.
//
// Mapping: PJ threads per group (adjacent lanes of one wavefront); the tangents of a track's sequences live in LDS next to the
// primal state ((1 + D + K) doubles per sequence and direction) - that, not the arithmetic, limits how many tracks a CU holds, so
// the directions of a group are dealt over PJ lanes: every lane repeats the (cheap) primal merge of its group from broadcast LDS
// reads, carries the directions j, j + PJ, ..., and lane j == 0 stores the new primal state after the tangents of the step have
// been updated in place.  Per-track and per-block sums are taken in a fixed order (no floating-point atomics):
// the result is bit-reproducible for a given launch geometry.
#pragma once
#include "xt_kernel.h"

struct XtGradArgs {
    const double* dblob;   // [NP][TB] tangent tables of direction i (layout below), global memory
    double* gpartials;     // [nblocks][NP + 1]: per-block sums of LL and of dLL/dtheta_i
    int32_t NP;            // directions
    int32_t TB;            // doubles per direction: XT_BLOB_HDR + XT_NTAB * S * G
    int32_t tan_lds;       // 1: the tangent tables are copied to LDS (small models), 0: read from global memory
    int32_t PJ;            // threads per group: thread (g, j) carries the directions j, j + PJ, ... (power of two, adjacent lanes)
    int32_t NU;            // register-resident 2-state kernels (xt_reg2.h): "uniform" directions handled at the last position only
    const double* udblob;  // [NU][TB] their tangent tables
};
// Tangent table block of one direction (TB doubles):
//   [0..2] d l2 (global localisation error),  [3] d slope,  [4] d offset,  [8 + s] d log Fs[s]
//   XT_BLOB_HDR + (v * S + prev) * G + q :  v = 0..3  d log of table v (T, T*stay, T*Eend, T*stay*Eend);  v = 4  d d2 (absolute)
XT_HD int xt_grad_tb_doubles(int S, int G) { return XT_BLOB_HDR + XT_NTAB * S * G; }
// LDS doubles per track: two primal regions (read / write, swapped every step: no lane ever overwrites state that a lane of
// another direction still reads) + NP tangent planes
XT_HD int xt_grad_region_doubles(int EP, int D, int K, int NP) { return 2 * xt_region_doubles(EP, D, K) + NP * EP * (1 + D + K); }
// per track slot: block accumulators bacc[NP + 1], column sums csum[NP + 1], per-thread partials gth[NP + 1][NG]
XT_HD int xt_grad_acc_doubles(int NP, int NG) { return 2 * (NP + 2) + (NP + 1) * NG; }

template <int G_, int D, int K, class Ctx>
XT_HD void xt_grad_body(const XtKernelArgs& a, const XtGradArgs& ga, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    constexpr int GM = G_ ? G_ : 1;
    const int G = G_ ? G_ : a.G;
    const int S = a.S, E = a.E, EP = a.EP, NG = a.NG, L = b.L, NP = ga.NP, TB = ga.TB;
    const int tid = cx.tid();
    double* smem = cx.smem();

    // ---- model tables (+ tangent tables of small models) -> LDS
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += cx.nthreads()) smem[i] = a.blob[i];
    const int tan0 = (ntab + 1) & ~1;
    if (ga.tan_lds)
        for (int i = tid; i < NP * TB; i += cx.nthreads()) smem[tan0 + i] = ga.dblob[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;
    const double* DT = ga.tan_lds ? smem + tan0 : ga.dblob;  // [NP][TB]
    const int reg0 = tan0 + (ga.tan_lds ? ((NP * TB + 1) & ~1) : 0);

    const int PJ = ga.PJ, NT = NG * PJ;  // threads per track
    const int slot = tid / NT;
    const int rr_ = tid - slot * NT;      // thread index inside the track
    const int g = rr_ / PJ;
    const int j = rr_ - g * PJ;
    const bool tvalid = slot < a.TPB;
    const int rdoubles = xt_grad_region_doubles(EP, D, K, NP);
    double* reg = smem + reg0 + (tvalid ? slot : 0) * rdoubles;
    const int pdoubles = xt_region_doubles(EP, D, K);
    int* red_e = (int*)(reg + EP * (1 + D + K)) + ((EP + 1) & ~1);  // [0] final-reduce exponent, [1] NaN-input flag (in region 0)
    double* tan = reg + 2 * pdoubles;  // [NP][(1 + D + K)][EP]: rz, dm[D], du[K]
    const int tstride = (1 + D + K) * EP;
    const int adoubles = xt_grad_acc_doubles(NP, NG);
    double* bacc = smem + reg0 + a.TPB * rdoubles + (tvalid ? slot : 0) * adoubles;  // [NP + 1] (+ pad)
    double* csum = bacc + NP + 2;                                                     // [NP + 1] (+ pad)
    double* gth = csum + NP + 2;                                                      // [NP + 1][NG]
    double* spos = smem + reg0 + a.TPB * (rdoubles + adoubles) + (tvalid ? slot : 0) * xt_stage_doubles(D);
    double* ssig = spos + XT_STAGE * D;

    const int prev = g / a.prev_div;
    const double* T0 = TAB + (0 * S + prev) * G;
    const double* T1 = TAB + (1 * S + prev) * G;
    const double* TD2 = TAB + (4 * S + prev) * G;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int toff = XT_BLOB_HDR + prev * G;  // + v * S * G + q inside a direction's block
    const int SG = S * G;

    if (tvalid)
        for (int i = rr_; i < NP + 1; i += NT) bacc[i] = 0.0;
    if (tvalid && rr_ == 0) red_e[1] = 0;
    cx.sync();

    const int64_t nbatch = (b.N + a.TPB - 1) / a.TPB;
    for (int64_t batch = lb; batch < nbatch; batch += nb) {
        const int64_t trk = batch * a.TPB + slot;
        const bool act = tvalid && trk < b.N;
        const double* c = b.tracks + (act ? trk : 0) * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + (act ? trk : 0) * (int64_t)L * a.KS : nullptr;

        auto stage = [&](int p0) {
            if (act) {
                for (int i = rr_; i < XT_STAGE * D; i += NT)
                    if (p0 + i / D < L) {
                        const double v = c[p0 * D + i];
                        spos[i] = v;
                        if (v != v) red_e[1] = 1;
                    }
                if (sg)
                    for (int i = rr_; i < XT_STAGE * a.KS; i += NT)
                        if (p0 + i / a.KS < L) {
                            const double v = sg[p0 * a.KS + i];
                            ssig[i] = v;
                            if (v != v) red_e[1] = 1;
                        }
            }
            cx.sync();
        };
        // l2[k] of position pos; sc[k] = per-peak chain factor: d l2[k] = sc[k] * (sraw[k] * d slope + d offset) (mode 2)
        auto load_l2 = [&](int pos, double* l2, double* sc, double* sraw) {
            for (int k = 0; k < K; ++k) {
                sc[k] = 0.0;
                sraw[k] = 0.0;
            }
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k) {
                    const double s0 = ssig[(pos & (XT_STAGE - 1)) * a.KS + (a.KS == 1 ? 0 : k)];
                    double s = s0;
                    if (a.locerr_mode == 2) {
                        s = xt_fma(s0, hdr[3], hdr[4]);
                        const bool clipped = s < 1e-6;
                        s = clipped ? 1e-6 : s;
                        sc[k] = clipped ? 0.0 : 2.0 * s;
                        sraw[k] = s0;
                    }
                    l2[k] = s * s;
                }
            }
        };
        auto dl2_of = [&](const double* dtb, const double* sc, const double* sraw, double* dl2) {
            if (a.locerr_mode == 0)
                for (int k = 0; k < K; ++k) dl2[k] = dtb[k];
            else
                for (int k = 0; k < K; ++k) dl2[k] = sc[k] * xt_fma(sraw[k], dtb[3], dtb[4]);
        };

        // primal state: read from (zm, mm, uu, ze), the step writes (zmN, mmN, uuN, zeN); swapped after every step
        double* zm = reg;
        double* mm = zm + EP;
        double* uu = mm + D * EP;
        int* ze = (int*)(uu + K * EP);
        double* zmN = reg + pdoubles;
        double* mmN = zmN + EP;
        double* uuN = mmN + D * EP;
        int* zeN = (int*)(uuN + K * EP);
        stage(0);
        // ---- position 0
        if (act) {
            double l20[K], sc0[K], sr0[K], c0[D];
            load_l2(0, l20, sc0, sr0);
            for (int d = 0; d < D; ++d) c0[d] = spos[d];
            for (int il = rr_; il < E; il += NT) {
                const bool live = il < S;
                const int i = xt_skew(il, a.skew);
                zm[i] = live ? hdr[8 + il] : 0.0;
                ze[i] = live ? 0 : XT_EMIN;
                for (int d = 0; d < D; ++d) mm[d * EP + i] = c0[d];
                for (int k = 0; k < K; ++k) uu[k * EP + i] = l20[k];
                for (int p = 0; p < NP; ++p) {
                    const double* dtb = DT + p * TB;
                    double* tp = tan + p * tstride;
                    double dl2[K];
                    dl2_of(dtb, sc0, sr0, dl2);
                    tp[i] = live ? dtb[8 + il] : 0.0;
                    for (int d = 0; d < D; ++d) tp[(1 + d) * EP + i] = 0.0;
                    for (int k = 0; k < K; ++k) tp[(1 + D + k) * EP + i] = dl2[k];
                }
            }
            if (rr_ == 0) red_e[0] = XT_EMIN;
        }
        cx.sync();

        // ---- positions 1 .. L-2
        for (int t = 1; t <= L - 2; ++t) {
            if ((t & (XT_STAGE - 1)) == 0) stage(t);
            const int ph = (t - 1) % a.P;
            if (act) {
                const int base = a.base_tab[ph * NG + g];
                const int32_t* off = a.off_tab + ph * G;
                double ct[D], l2t[K], sct[K], srt[K];
                for (int d = 0; d < D; ++d) ct[d] = spos[(t & (XT_STAGE - 1)) * D + d];
                load_l2(t, l2t, sct, srt);
                const bool stay = t >= stay_from;
                const double* TTl = stay ? T1 : T0;
                const int tv = (stay ? 1 : 0) * SG + toff;

                // primal merge
                int emax = XT_EMIN;
                for (int q = 0; q < G; ++q) {
                    const int e = ze[xt_skew(base + off[q], a.skew)];
                    emax = e > emax ? e : emax;
                }
                double W = 0.0, mb[D], ub[K];
                double aj[GM], mj[GM][D], uj[GM][K];  // members' normalised weights / means / variances (compile-time G only)
                for (int d = 0; d < D; ++d) mb[d] = 0.0;
                for (int k = 0; k < K; ++k) ub[k] = 0.0;
                for (int q = 0; q < G; ++q) {
                    const int idx = xt_skew(base + off[q], a.skew);
                    const double aq = xt_ldexp(zm[idx], ze[idx] - emax);
                    W += aq;
                    if (G_) aj[G_ ? q : 0] = aq;
                    for (int d = 0; d < D; ++d) {
                        const double v = mm[d * EP + idx];
                        mb[d] = xt_fma(aq, v, mb[d]);
                        if (G_) mj[G_ ? q : 0][d] = v;
                    }
                    for (int k = 0; k < K; ++k) {
                        const double v = uu[k * EP + idx];
                        ub[k] = xt_fma(aq, v, ub[k]);
                        if (G_) uj[G_ ? q : 0][k] = v;
                    }
                }
                const double rW = W > 0.0 ? xt_rcp(W) : 0.0;
                for (int d = 0; d < D; ++d) mb[d] *= rW;
                for (int k = 0; k < K; ++k) ub[k] *= rW;
                if (G_)
                    for (int q = 0; q < GM; ++q) aj[q] *= rW;
                double dm[D], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dm[d] = ct[d] - mb[d];
                    dsq = xt_fma(dm[d], dm[d], dsq);
                }
                // per-new-digit primal quantities the tangents need (registers for compile-time G)
                double rq[GM][K], tq[GM][K];
                auto light = [&](int q, double* r, double* tt) {
                    const double d2 = TD2[q];
                    for (int k = 0; k < K; ++k) {
                        const double s2 = d2 + ub[k];
                        r[k] = xt_rcp(l2t[k] + s2);
                        tt[k] = s2 * r[k];
                    }
                };
                if (G_)
                    for (int q = 0; q < GM; ++q) light(q, rq[q], tq[q]);

                // tangents of this lane's directions (reads the OLD primal members, writes the new tangents in place)
                for (int p = j; p < NP; p += PJ) {
                    const double* dtb = DT + p * TB;
                    double* tp = tan + p * tstride;
                    double R = 0.0, dmb[D], dub[K];
                    for (int d = 0; d < D; ++d) dmb[d] = 0.0;
                    for (int k = 0; k < K; ++k) dub[k] = 0.0;
                    for (int q = 0; q < G; ++q) {
                        const int idx = xt_skew(base + off[q], a.skew);
                        double aq;
                        if (G_)
                            aq = aj[G_ ? q : 0];
                        else
                            aq = xt_ldexp(zm[idx], ze[idx] - emax) * rW;
                        const double rzq = tp[idx];
                        R = xt_fma(aq, rzq, R);
                        for (int d = 0; d < D; ++d) {
                            const double mv = G_ ? mj[G_ ? q : 0][d] : mm[d * EP + idx];
                            dmb[d] = xt_fma(aq, xt_fma(mv - mb[d], rzq, tp[(1 + d) * EP + idx]), dmb[d]);
                        }
                        for (int k = 0; k < K; ++k) {
                            const double uv = G_ ? uj[G_ ? q : 0][k] : uu[k * EP + idx];
                            dub[k] = xt_fma(aq, xt_fma(uv - ub[k], rzq, tp[(1 + D + k) * EP + idx]), dub[k]);
                        }
                    }
                    double dl2[K];
                    dl2_of(dtb, sct, srt, dl2);
                    double ddsq = 0.0;  // d |c - m_bar|^2 (K == 1)
                    for (int d = 0; d < D; ++d) ddsq = xt_fma(-2.0 * dm[d], dmb[d], ddsq);
                    for (int q = 0; q < G; ++q) {
                        const int idx = xt_skew(base + off[q], a.skew);
                        double rl[K], tl[K];
                        const double* r = rl;
                        const double* tt = tl;
                        if (G_) {
                            r = rq[G_ ? q : 0];
                            tt = tq[G_ ? q : 0];
                        } else {
                            light(q, rl, tl);
                        }
                        const double dd2 = dtb[4 * SG + toff + q];
                        double rz = R + dtb[tv + q];
                        double dtt[K];
                        if (K == 1) {
                            const double ds2 = dd2 + dub[0], dden = dl2[0] + ds2;
                            dtt[0] = r[0] * (ds2 - tt[0] * dden);
                            // quad = dsq r / 2:  d quad = (ddsq r - dsq r^2 dden) / 2
                            rz -= 0.5 * r[0] * (D * dden + ddsq - dsq * r[0] * dden);
                        } else {
                            for (int d = 0; d < D; ++d) {
                                const double ds2 = dd2 + dub[d], dden = dl2[d] + ds2;
                                dtt[d] = r[d] * (ds2 - tt[d] * dden);
                                rz -= 0.5 * r[d] * (dden + (-2.0 * dm[d] * dmb[d]) - dm[d] * dm[d] * r[d] * dden);
                            }
                        }
                        tp[idx] = W > 0.0 ? rz : 0.0;
                        for (int d = 0; d < D; ++d) {
                            const int kk = K == 1 ? 0 : d;
                            tp[(1 + d) * EP + idx] = xt_fma(dm[d], dtt[kk], dmb[d] * (1.0 - tt[kk]));
                        }
                        for (int k = 0; k < K; ++k) tp[(1 + D + k) * EP + idx] = xt_fma(l2t[k], dtt[k], dl2[k] * tt[k]);
                    }
                }

                // primal update (as xt_track_body), by lane j == 0 of the group
                const double Wm = xt_frexp_mant(W);
                const int We = W > 0.0 ? emax + xt_frexp_exp(W) : XT_EMIN;
                for (int q = 0; q < (j == 0 ? G : 0); ++q) {
                    const int idx = xt_skew(base + off[q], a.skew);
                    double rl[K], tl[K];
                    const double* r = rl;
                    const double* tt = tl;
                    if (G_) {
                        r = rq[G_ ? q : 0];
                        tt = tq[G_ ? q : 0];
                    } else {
                        light(q, rl, tl);
                    }
                    double quad, gf;
                    if (K == 1) {
                        quad = 0.5 * dsq * r[0];
                        gf = xt_pow_half<D>(r[0]);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        for (int d = 0; d < D; ++d) {
                            quad = xt_fma(0.5 * dm[d] * dm[d], r[d], quad);
                            gf *= r[d];
                        }
                        gf = sqrt(gf);
                    }
                    double pp;
                    int jt, n;
                    xt_exp_tab(-quad, pp, jt, n);
                    const int en = We + n;
                    zmN[idx] = (Wm * TTl[q]) * (gf * T64[jt]) * pp;
                    zeN[idx] = en > XT_EMIN ? en : XT_EMIN;
                    for (int d = 0; d < D; ++d) mmN[d * EP + idx] = xt_fma(dm[d], tt[K == 1 ? 0 : d], mb[d]);
                    for (int k = 0; k < K; ++k) uuN[k * EP + idx] = l2t[k] * tt[k];
                }
            }
            cx.sync();
            {
                double* t0 = zm; zm = zmN; zmN = t0;
                t0 = mm; mm = mmN; mmN = t0;
                t0 = uu; uu = uuN; uuN = t0;
                int* t1 = ze; ze = zeN; zeN = t1;
            }
        }

        // ---- last position (+ leaving/bleaching term).  Pass 1: extended-range total of every thread -> common exponent fe.
        // Pass 2: the same weights on the 2^fe scale (plain doubles) and, per direction, sum w * d log w.
        if (((L - 1) & (XT_STAGE - 1)) == 0) stage(L - 1);
        const int tl = L - 1;
        const int phl = (tl - 1) % a.P;
        const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
        const double* TF = TAB + (vfin * S + prev) * G;
        const int tvf = vfin * SG + toff;
        double cl[D], l2l[K], scl[K], srl[K];
        for (int d = 0; d < D; ++d) cl[d] = 0.0;
        // weight of (old member Q, new digits q) as mantissa/exponent; rel != nullptr: also d log w for direction (dtb, tp)
        auto pair_term = [&](int idx, int q, const double* dq, double dsq, double zq, int eq, double& wm, int& we, const double* dtb,
                             const double* dl2, double rzQ, const double* dmQ, const double* duQ, double& rel) {
            double quad, gf;
            if (K == 1) {
                const double r = xt_rcp(TD2[q] + uu[idx] + l2l[0]);
                quad = 0.5 * dsq * r;
                gf = xt_pow_half<D>(r);
                if (dtb) {
                    const double dden = dtb[4 * SG + toff + q] + duQ[0] + dl2[0];
                    double ddsq = 0.0;
                    for (int d = 0; d < D; ++d) ddsq = xt_fma(-2.0 * dq[d], dmQ[d], ddsq);
                    rel = rzQ + dtb[tvf + q] - 0.5 * r * (D * dden + ddsq - dsq * r * dden);
                }
            } else {
                quad = 0.0;
                gf = 1.0;
                if (dtb) rel = rzQ + dtb[tvf + q];
                for (int d = 0; d < D; ++d) {
                    const double r = xt_rcp(TD2[q] + uu[d * EP + idx] + l2l[d]);
                    quad = xt_fma(0.5 * dq[d] * dq[d], r, quad);
                    gf *= r;
                    if (dtb) {
                        const double dden = dtb[4 * SG + toff + q] + duQ[d] + dl2[d];
                        rel -= 0.5 * r * (dden - 2.0 * dq[d] * dmQ[d] - dq[d] * dq[d] * r * dden);
                    }
                }
                gf = sqrt(gf);
            }
            double pp;
            int j, n;
            xt_exp_tab(-quad, pp, j, n);
            wm = zq * TF[q] * (gf * T64[j]) * pp;
            we = eq + n;
        };
        XtAcc tot;
        tot.clear();
        if (act) {
            const int base = a.base_tab[phl * NG + g];
            const int32_t* off = a.off_tab + phl * G;
            for (int d = 0; d < D; ++d) cl[d] = spos[(tl & (XT_STAGE - 1)) * D + d];
            load_l2(tl, l2l, scl, srl);
            for (int Q = 0; Q < G; ++Q) {
                const int idx = xt_skew(base + off[Q], a.skew);
                const double zq = zm[idx];
                if (zq == 0.0) continue;
                double dq[D], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - mm[d * EP + idx];
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                for (int q = 0; q < G; ++q) {
                    double wm, rel;
                    int we;
                    pair_term(idx, q, dq, dsq, zq, ze[idx], wm, we, nullptr, nullptr, 0.0, nullptr, nullptr, rel);
                    tot.add(wm, we);
                }
            }
            if (tot.m != 0.0 && j == 0) cx.atomic_max_i32(&red_e[0], tot.e);
        }
        cx.sync();
        if (act) {
            const int fe = red_e[0];
            const int base = a.base_tab[phl * NG + g];
            const int32_t* off = a.off_tab + phl * G;
            if (j == 0) gth[g] = tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0;  // column 0: weight total of this group
            for (int p = j; p < NP; p += PJ) {
                const double* dtb = DT + p * TB;
                const double* tp = tan + p * tstride;
                double dl2[K];
                dl2_of(dtb, scl, srl, dl2);
                double acc = 0.0;
                for (int Q = 0; Q < G; ++Q) {
                    const int idx = xt_skew(base + off[Q], a.skew);
                    const double zq = zm[idx];
                    if (zq == 0.0) continue;
                    double dq[D], dsq = 0.0, dmQ[D], duQ[K];
                    for (int d = 0; d < D; ++d) {
                        dq[d] = cl[d] - mm[d * EP + idx];
                        dsq = xt_fma(dq[d], dq[d], dsq);
                        dmQ[d] = tp[(1 + d) * EP + idx];
                    }
                    for (int k = 0; k < K; ++k) duQ[k] = tp[(1 + D + k) * EP + idx];
                    const double rzQ = tp[idx];
                    for (int q = 0; q < G; ++q) {
                        double wm, rel = 0.0;
                        int we;
                        pair_term(idx, q, dq, dsq, zq, ze[idx], wm, we, dtb, dl2, rzQ, dmQ, duQ, rel);
                        acc = xt_fma(xt_ldexp(wm, we - fe), rel, acc);
                    }
                }
                gth[(p + 1) * NG + g] = acc;
            }
        }
        cx.sync();
        // fixed-order sums over the track's NG groups, one column per thread (0: weight total, 1 + p: direction p)
        if (act)
            for (int col = rr_; col < NP + 1; col += NT) {
                double s2 = 0.0;
                for (int i = 0; i < NG; ++i) s2 += gth[col * NG + i];
                csum[col] = s2;
            }
        cx.sync();
        if (act) {
            const bool poisoned = red_e[1] != 0;
            const double sw = csum[0];
            const int fe = red_e[0];
            for (int col = rr_; col < NP + 1; col += NT) {
                if (col == 0) {
                    const double ll = poisoned ? NAN : log(sw) + (double)fe * XT_LN2 + b.ll_const;
                    if (b.ll_out) b.ll_out[trk] = ll;
                    bacc[0] += ll;
                } else {
                    bacc[col] += poisoned ? NAN : csum[col] / sw;
                }
            }
        }
        cx.sync();
        if (act && rr_ == 0) red_e[1] = 0;
    }

    // ---- block partials: fixed-order sum over the block's track slots, one column per thread
    cx.sync();
    double* bacc0 = smem + reg0 + a.TPB * rdoubles;
    for (int col = tid; col < NP + 1; col += cx.nthreads()) {
        double s = 0.0;
        for (int i = 0; i < a.TPB; ++i) s += bacc0[i * adoubles + col];
        ga.gpartials[(int64_t)cx.block() * (NP + 1) + col] = s;
    }
}
