// Log-likelihood AND its exact gradient for models with G = S^ns members per group known at compile time (3- and 4-state models with
// one substep; the 2-state ones have their own kernels in xt_reg2.h): the sequence state and its tangents live in REGISTERS, the LDS
// is only the exchange medium between two steps.
//
// Same mathematics as xt_grad.h (forward-mode tangents rz = d log z, dm, du carried alongside the fixed-window recursion of
// extrack/tracking.py:109-318; replaces the finite differences of lmfit's BFGS around cum_Proba_Cs, tracking.py:1371).  What differs:
//   * xt_grad.h keeps the tangents of a track's S^F sequences in LDS next to the primal state: (1 + D + K) doubles per sequence and
//     direction - 303 KB for 3 states, frame_len 6 and 13 directions.  That caps the tracks a CU works on (LDS capacity), forces passes of
//     4 directions with ONE track per CU, and PJ lanes per group that all repeat the primal merge: 1.9 s per C3 evaluation at frame_len 6
//     against 0.4 s for the 14 evaluations of a finite-difference gradient (r02 / r03 measurements).
//   * here ONE lane owns a group: its G members {z, e, m[D], u[K]} and, per direction, {rz, dm[D], du[K]} x G are register arrays
//     (NPC <= 6 directions per pass: 6 x 3 x 4 doubles = 144 VGPRs for 3 states).  A step merges, expands and integrates in
//     registers; the new sequences then go through ONE LDS exchange buffer per round - round 0 the primal state, round 1 + p direction p
//     - written at the circular-slot address of the sequence (xt_tables.h) and read back as the members of the lane's NEXT group.  Two
//     buffers alternate, so a round costs one workgroup barrier.  LDS per track: 2 x S^F x (1 + D + K) doubles whatever the number of
//     directions (47 KB for 3 states, frame_len 6): two tracks per CU, which is also what the registers allow.
#pragma once
#include "xt_grad.h"

// exchange buffer of one track: (1 + D + K) planes of EP doubles (struct of arrays, skewed like xt_kernel.h: conflict-free for the
// stride-S^h group addressing) + EP exponents (used by the primal round only)
XT_HD int xt_gradr_xbuf_doubles(int EP, int D, int K) { return EP * (1 + D + K) + (EP + 1) / 2 + 1; }
// per track slot: two exchange buffers + block accumulators bacc[NP + 2], column sums csum[NP + 2] + 2 ints
XT_HD int xt_gradr_track_doubles(int EP, int D, int K, int NP) { return 2 * xt_gradr_xbuf_doubles(EP, D, K) + 2 * (NP + 2) + 2; }
// fixed part: model tables, tangent tables of the pass, the digit-slot tables (ints)
XT_HD int xt_gradr_fixed_doubles(int S, int G, int NP, int P, int NG)
{
    return ((xt_tab_doubles(S, G) + 1) & ~1) + ((NP * xt_grad_tb_doubles(S, G) + 1) & ~1) + (P * NG + 1) / 2 + (P * G + 1) / 2 + 2;
}
XT_HD size_t xt_gradr_lds_bytes(int S, int G, int E, int EP, int NG, int P, int D, int K, int NP, int tpb)
{
    (void)E;
    return ((size_t)xt_gradr_fixed_doubles(S, G, NP, P, NG) + (size_t)tpb * ((size_t)xt_gradr_track_doubles(EP, D, K, NP) + xt_stage_doubles(D))) * sizeof(double);
}

template <int G_, int D, int K, int NPC, class Ctx>
XT_HD void xt_gradr_body(const XtKernelArgs& a, const XtGradArgs& ga, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    constexpr int G = G_, TC = 1 + D + K;
    const int S = a.S, E = a.E, EP = a.EP, NG = a.NG, L = b.L, NP = ga.NP, TB = ga.TB, P = a.P;
    const int tid = cx.tid();
    double* smem = cx.smem();

    // ---- model tables, tangent tables and digit-slot tables -> LDS
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += cx.nthreads()) smem[i] = a.blob[i];
    const int tan0 = (ntab + 1) & ~1;
    for (int i = tid; i < NP * TB; i += cx.nthreads()) smem[tan0 + i] = ga.dblob[i];
    const int it0 = tan0 + ((NP * TB + 1) & ~1);
    int* bt = (int*)(smem + it0);                            // [P][NG] entry index of the group's q = 0 member per phase
    int* ot = (int*)(smem + it0 + (P * NG + 1) / 2);         // [P][G]  entry offset of member q per phase
    for (int i = tid; i < P * NG; i += cx.nthreads()) bt[i] = a.base_tab[i];
    for (int i = tid; i < P * G; i += cx.nthreads()) ot[i] = a.off_tab[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;
    const double* DT = smem + tan0;  // [NP][TB]
    const int reg0 = xt_gradr_fixed_doubles(S, G, NP, P, NG);

    const int slot = tid / NG;
    const int g = tid - slot * NG;
    const bool tvalid = slot < a.TPB;
    const int xdoubles = xt_gradr_xbuf_doubles(EP, D, K);
    const int tdoubles = xt_gradr_track_doubles(EP, D, K, NP);
    double* tr0 = smem + reg0 + (tvalid ? slot : 0) * tdoubles;
    double* X[2] = {tr0, tr0 + xdoubles};
    double* bacc = tr0 + 2 * xdoubles;   // [NP + 1] (+ pad)
    double* csum = bacc + NP + 2;        // [NP + 1] (+ pad)
    int* red_e = (int*)(csum + NP + 2);  // [0] final-reduce exponent, [1] NaN-input flag
    double* gth = X[0];                  // per-thread partials [NP + 1][NG] of the final reduction (the exchange buffers are idle then)
    double* spos = smem + reg0 + a.TPB * tdoubles + (tvalid ? slot : 0) * xt_stage_doubles(D);
    double* ssig = spos + XT_STAGE * D;
    auto XZ = [&](int r) XT_INL { return (int*)(X[r] + TC * EP); };  // exponent plane of buffer r

    const int prev = g / a.prev_div;
    const double* T0 = TAB + (0 * S + prev) * G;
    const double* T1 = TAB + (1 * S + prev) * G;
    const double* TD2 = TAB + (4 * S + prev) * G;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int toff = XT_BLOB_HDR + prev * G;  // + v * S * G + q inside a direction's block
    const int SG = S * G;

    if (tvalid)
        for (int i = g; i < NP + 1; i += NG) bacc[i] = 0.0;
    if (tvalid && g == 0) red_e[1] = 0;
    cx.sync();

    const int64_t nbatch = (b.N + a.TPB - 1) / a.TPB;
    for (int64_t batch = lb; batch < nbatch; batch += nb) {
        const int64_t trk = batch * a.TPB + slot;
        const bool act = tvalid && trk < b.N;
        const double* c = b.tracks + (act ? trk : 0) * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + (act ? trk : 0) * (int64_t)L * a.KS : nullptr;

        auto stage = [&](int p0) XT_INL {
            if (act) {
                for (int i = g; i < XT_STAGE * D; i += NG)
                    if (p0 + i / D < L) {
                        const double v = c[p0 * D + i];
                        spos[i] = v;
                        if (v != v) red_e[1] = 1;
                    }
                if (sg)
                    for (int i = g; i < XT_STAGE * a.KS; i += NG)
                        if (p0 + i / a.KS < L) {
                            const double v = sg[p0 * a.KS + i];
                            ssig[i] = v;
                            if (v != v) red_e[1] = 1;
                        }
            }
            cx.sync();
        };
        // l2[k] of position pos; sc[k] = per-peak chain factor: d l2[k] = sc[k] * (sraw[k] * d slope + d offset) (mode 2)
        auto load_l2 = [&](int pos, double* l2, double* sc, double* sraw) XT_INL {
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                sc[k] = 0.0;
                sraw[k] = 0.0;
            }
            if (a.locerr_mode == 0) {
                XT_UNROLL
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                XT_UNROLL
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
        auto dl2_of = [&](const double* dtb, const double* sc, const double* sraw, double* dl2) XT_INL {
            if (a.locerr_mode == 0) {
                XT_UNROLL
                for (int k = 0; k < K; ++k) dl2[k] = dtb[k];
            } else {
                XT_UNROLL
                for (int k = 0; k < K; ++k) dl2[k] = sc[k] * xt_fma(sraw[k], dtb[3], dtb[4]);
            }
        };

        // ---- the lane's group: G members (primal) and their tangents along the NPC directions of this pass, in registers
        double zm[G], mm[G][D], uu[G][K];
        int ze[G];
        double trz[NPC][G], tdm[NPC][G][D], tdu[NPC][G][K];

        stage(0);
        // ---- position 0: the members of the group of phase 0 (entry index il: the initial state in slot 0, everything else zero weight)
        // (lanes without a track - the last, partial batch - walk track 0 of the bucket: uniform control flow in the step loop, no stores)
        {
            double l20[K], sc0[K], sr0[K], c0[D];
            load_l2(0, l20, sc0, sr0);
            XT_UNROLL
            for (int d = 0; d < D; ++d) c0[d] = spos[d];
            const int base = bt[g];
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                const int il = base + ot[Q];
                const bool live = il < S;
                zm[Q] = live ? hdr[8 + (live ? il : 0)] : 0.0;
                ze[Q] = live ? 0 : XT_EMIN;
                XT_UNROLL
                for (int d = 0; d < D; ++d) mm[Q][d] = c0[d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) uu[Q][k] = l20[k];
                XT_UNROLL
                for (int p = 0; p < NPC; ++p) {
                    const double* dtb = DT + (p < NP ? p : 0) * TB;
                    double dl2[K];
                    dl2_of(dtb, sc0, sr0, dl2);
                    trz[p][Q] = (live && p < NP) ? dtb[8 + (live ? il : 0)] : 0.0;
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) tdm[p][Q][d] = 0.0;
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) tdu[p][Q][k] = p < NP ? dl2[k] : 0.0;
                }
            }
            if (act && g == 0) red_e[0] = XT_EMIN;
        }
        cx.sync();

        // ---- positions 1 .. L-2
        for (int t = 1; t <= L - 2; ++t) {
            if ((t & (XT_STAGE - 1)) == 0) stage(t);
            const int ph = (t - 1) % P, phn = t % P;
            int widx[G], ridx[G];  // LDS entry (skewed) of output q of this step / of member Q of the next step
            double ct[D], l2t[K], sct[K], srt[K];
            // shared primal factors of the tangent updates
            double aj[G], mjc[G][D], ujc[G][K], dn[D], rq[G][K], tq[G][K], Aq[G][K];
            bool liveW = false;
            const bool stay = t >= stay_from;
            const int tv = (stay ? 1 : 0) * SG + toff;
            {
                const int base = bt[ph * NG + g], baseN = bt[phn * NG + g];
                XT_UNROLL
                for (int q = 0; q < G; ++q) {
                    widx[q] = xt_skew(base + ot[ph * G + q], a.skew);
                    ridx[q] = xt_skew(baseN + ot[phn * G + q], a.skew);
                }
                XT_UNROLL
                for (int d = 0; d < D; ++d) ct[d] = spos[(t & (XT_STAGE - 1)) * D + d];
                load_l2(t, l2t, sct, srt);
                const double* TTl = stay ? T1 : T0;

                // primal merge
                int emax = XT_EMIN;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) emax = ze[Q] > emax ? ze[Q] : emax;
                double W = 0.0, mb[D], ub[K];
                XT_UNROLL
                for (int d = 0; d < D; ++d) mb[d] = 0.0;
                XT_UNROLL
                for (int k = 0; k < K; ++k) ub[k] = 0.0;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) {
                    aj[Q] = xt_ldexp(zm[Q], ze[Q] - emax);
                    W += aj[Q];
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) mb[d] = xt_fma(aj[Q], mm[Q][d], mb[d]);
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) ub[k] = xt_fma(aj[Q], uu[Q][k], ub[k]);
                }
                liveW = W > 0.0;
                const double rW = liveW ? xt_rcp(W) : 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) mb[d] *= rW;
                XT_UNROLL
                for (int k = 0; k < K; ++k) ub[k] *= rW;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) {
                    aj[Q] *= rW;
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) mjc[Q][d] = mm[Q][d] - mb[d];
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) ujc[Q][k] = uu[Q][k] - ub[k];
                }
                double dsq = 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    dn[d] = ct[d] - mb[d];
                    dsq = xt_fma(dn[d], dn[d], dsq);
                }
                const double Wm = xt_frexp_mant(W);
                const int We = liveW ? emax + xt_frexp_exp(W) : XT_EMIN;
                // round 0: the new primal sequences (as xt_track_body) -> exchange buffer 0
                int* xz = XZ(0);
                XT_UNROLL
                for (int q = 0; q < G; ++q) {
                    const double d2 = TD2[q];
                    double quad, gf;
                    if (K == 1) {
                        const double s2 = d2 + ub[0];
                        rq[q][0] = xt_rcp(l2t[0] + s2);
                        tq[q][0] = s2 * rq[q][0];
                        quad = 0.5 * dsq * rq[q][0];
                        gf = xt_pow_half<D>(rq[q][0]);
                        Aq[q][0] = -0.5 * rq[q][0] * xt_fma(-dsq, rq[q][0], (double)D);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) {
                            const double s2 = d2 + ub[d];
                            rq[q][d] = xt_rcp(l2t[d] + s2);
                            tq[q][d] = s2 * rq[q][d];
                            quad = xt_fma(0.5 * dn[d] * dn[d], rq[q][d], quad);
                            gf *= rq[q][d];
                            Aq[q][d] = -0.5 * rq[q][d] * xt_fma(-(dn[d] * dn[d]), rq[q][d], 1.0);
                        }
                        gf = sqrt(gf);
                    }
                    double pp;
                    int jt, n;
                    xt_exp_tab(-quad, pp, jt, n);
                    const int en = We + n;
                    X[0][widx[q]] = (Wm * TTl[q]) * (gf * T64[jt]) * pp;
                    xz[widx[q]] = en > XT_EMIN ? en : XT_EMIN;
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) X[0][(1 + d) * EP + widx[q]] = xt_fma(dn[d], tq[q][K == 1 ? 0 : d], mb[d]);
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) X[0][(1 + D + k) * EP + widx[q]] = l2t[k] * tq[q][k];
                }
            }
            cx.sync();
            {
                const int* xz = XZ(0);
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) {
                    zm[Q] = X[0][ridx[Q]];
                    ze[Q] = xz[ridx[Q]];
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) mm[Q][d] = X[0][(1 + d) * EP + ridx[Q]];
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) uu[Q][k] = X[0][(1 + D + k) * EP + ridx[Q]];
                }
            }
            // rounds 1 + p: direction p's new tangents -> buffer (1 + p) & 1 -> the next group's member tangents (one barrier per round:
            // a buffer is rewritten two rounds later, after every lane has passed the barrier in between)
            XT_UNROLL
            for (int p = 0; p < NPC; ++p) {
                if (p >= NP) break;
                xt_sched_fence();
                double* xb = X[(1 + p) & 1];
                {
                    const double* dtb = DT + p * TB;
                    double R = 0.0, dmb[D], dub[K];
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) dmb[d] = 0.0;
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) dub[k] = 0.0;
                    XT_UNROLL
                    for (int Q = 0; Q < G; ++Q) {
                        const double rzq = trz[p][Q];
                        R = xt_fma(aj[Q], rzq, R);
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) dmb[d] = xt_fma(aj[Q], xt_fma(mjc[Q][d], rzq, tdm[p][Q][d]), dmb[d]);
                        XT_UNROLL
                        for (int k = 0; k < K; ++k) dub[k] = xt_fma(aj[Q], xt_fma(ujc[Q][k], rzq, tdu[p][Q][k]), dub[k]);
                    }
                    double dl2[K], hd[K];  // hd: -1/2 d |c - m_bar|^2 (per dim when K == D)
                    dl2_of(dtb, sct, srt, dl2);
                    if (K == 1) {
                        hd[0] = 0.0;
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) hd[0] = xt_fma(dn[d], dmb[d], hd[0]);
                    } else {
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) hd[d] = dn[d] * dmb[d];
                    }
                    XT_UNROLL
                    for (int q = 0; q < G; ++q) {
                        const double dd2 = dtb[4 * SG + toff + q];
                        double rz = R + dtb[tv + q], dtt[K];
                        XT_UNROLL
                        for (int k = 0; k < K; ++k) {
                            const double ds2 = dd2 + dub[k], dden = dl2[k] + ds2;
                            dtt[k] = rq[q][k] * xt_fma(-tq[q][k], dden, ds2);
                            rz = xt_fma(Aq[q][k], dden, rz);
                            rz = xt_fma(rq[q][k], hd[k], rz);
                            xb[(1 + D + k) * EP + widx[q]] = xt_fma(l2t[k], dtt[k], dl2[k] * tq[q][k]);
                        }
                        xb[widx[q]] = liveW ? rz : 0.0;
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) {
                            const int kk = K == 1 ? 0 : d;
                            xb[(1 + d) * EP + widx[q]] = xt_fma(dn[d], dtt[kk], xt_fma(-tq[q][kk], dmb[d], dmb[d]));
                        }
                    }
                }
                cx.sync();
                {
                    XT_UNROLL
                    for (int Q = 0; Q < G; ++Q) {
                        trz[p][Q] = xb[ridx[Q]];
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) tdm[p][Q][d] = xb[(1 + d) * EP + ridx[Q]];
                        XT_UNROLL
                        for (int k = 0; k < K; ++k) tdu[p][Q][k] = xb[(1 + D + k) * EP + ridx[Q]];
                    }
                }
            }
            // the first write of the NEXT step goes to buffer 0; the last reads of this step were from buffer NP & 1: when that is
            // buffer 0 a barrier has to separate them
            if ((NP & 1) == 0) cx.sync();
        }

        // ---- last position (+ leaving / bleaching term).  Pass 1: extended-range total of every thread -> common exponent fe.
        // Pass 2: the same weights on the 2^fe scale (plain doubles) and, per direction, sum w * d log w.
        if (((L - 1) & (XT_STAGE - 1)) == 0) stage(L - 1);
        const int tl = L - 1;
        const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
        const double* TF = TAB + (vfin * S + prev) * G;
        const int tvf = vfin * SG + toff;
        double cl[D], l2l[K], scl[K], srl[K];
        double wmP[G][G], rP[G][G][K], dqP[G][D], dsqP[G];
        int weP[G][G];
        XtAcc tot;
        tot.clear();
        if (act) {
            XT_UNROLL
            for (int d = 0; d < D; ++d) cl[d] = spos[(tl & (XT_STAGE - 1)) * D + d];
            load_l2(tl, l2l, scl, srl);
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                dsqP[Q] = 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    dqP[Q][d] = cl[d] - mm[Q][d];
                    dsqP[Q] = xt_fma(dqP[Q][d], dqP[Q][d], dsqP[Q]);
                }
                XT_UNROLL
                for (int q = 0; q < G; ++q) {
                    double quad, gf;
                    if (K == 1) {
                        const double r = xt_rcp(TD2[q] + uu[Q][0] + l2l[0]);
                        rP[Q][q][0] = r;
                        quad = 0.5 * dsqP[Q] * r;
                        gf = xt_pow_half<D>(r);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) {
                            const double r = xt_rcp(TD2[q] + uu[Q][d] + l2l[d]);
                            rP[Q][q][d] = r;
                            quad = xt_fma(0.5 * dqP[Q][d] * dqP[Q][d], r, quad);
                            gf *= r;
                        }
                        gf = sqrt(gf);
                    }
                    double pp;
                    int jt, n;
                    xt_exp_tab(-quad, pp, jt, n);
                    wmP[Q][q] = zm[Q] * TF[q] * (gf * T64[jt]) * pp;
                    weP[Q][q] = ze[Q] + n;
                    tot.add(wmP[Q][q], weP[Q][q]);
                }
            }
            if (tot.m != 0.0) cx.atomic_max_i32(&red_e[0], tot.e);
        }
        cx.sync();  // every lane has its members in registers: the exchange buffers are free for the reduction scratch
        if (act) {
            const int fe = red_e[0];
            gth[g] = tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0;  // column 0: weight total of this group
            XT_UNROLL
            for (int p = 0; p < NPC; ++p) {
                if (p >= NP) break;
                const double* dtb = DT + p * TB;
                double dl2[K];
                dl2_of(dtb, scl, srl, dl2);
                double acc = 0.0;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) {
                    XT_UNROLL
                    for (int q = 0; q < G; ++q) {
                        double rel = trz[p][Q] + dtb[tvf + q];
                        const double dd2 = dtb[4 * SG + toff + q];
                        if (K == 1) {
                            const double r = rP[Q][q][0];
                            const double dden = dd2 + tdu[p][Q][0] + dl2[0];
                            double ddsq = 0.0;
                            XT_UNROLL
                            for (int d = 0; d < D; ++d) ddsq = xt_fma(-2.0 * dqP[Q][d], tdm[p][Q][d], ddsq);
                            rel -= 0.5 * r * (D * dden + ddsq - dsqP[Q] * r * dden);
                        } else {
                            XT_UNROLL
                            for (int d = 0; d < D; ++d) {
                                const double r = rP[Q][q][d];
                                const double dden = dd2 + tdu[p][Q][d] + dl2[d];
                                rel -= 0.5 * r * (dden - 2.0 * dqP[Q][d] * tdm[p][Q][d] - dqP[Q][d] * dqP[Q][d] * r * dden);
                            }
                        }
                        const double wsc = wmP[Q][q] != 0.0 ? xt_ldexp(wmP[Q][q], weP[Q][q] - fe) : 0.0;
                        acc = xt_fma(wsc, wsc != 0.0 ? rel : 0.0, acc);
                    }
                }
                gth[(p + 1) * NG + g] = acc;
            }
        }
        cx.sync();
        // fixed-order sums over the track's NG groups, one column per thread (0: weight total, 1 + p: direction p)
        if (act)
            for (int col = g; col < NP + 1; col += NG) {
                double s2 = 0.0;
                for (int i = 0; i < NG; ++i) s2 += gth[col * NG + i];
                csum[col] = s2;
            }
        cx.sync();
        if (act) {
            const bool poisoned = red_e[1] != 0;
            const double sw = csum[0];
            const int fe = red_e[0];
            for (int col = g; col < NP + 1; col += NG) {
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
        if (act && g == 0) red_e[1] = 0;
    }

    // ---- block partials: fixed-order sum over the block's track slots, one column per thread
    cx.sync();
    double* bacc0 = smem + reg0 + 2 * xdoubles;
    for (int col = tid; col < NP + 1; col += cx.nthreads()) {
        double s = 0.0;
        for (int i = 0; i < a.TPB; ++i) s += bacc0[i * tdoubles + col];
        ga.gpartials[(int64_t)cx.block() * (NP + 1) + col] = s;
    }
}
