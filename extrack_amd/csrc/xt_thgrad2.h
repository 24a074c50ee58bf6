// Threshold-fusion log-likelihood AND its exact gradient at a FROZEN plan, second mapping (round 4): one lane per (state sequence, track).
//
// Same mathematics and the same reference objective as xt_thgrad.h (extrack/tracking.py:1371 lmfit.minimize(cum_Proba_Cs) -> :991 -> :769
// Proba_Cs -> :427-650 P_Cs_inter_bound_stats_th with the merge groups of fuse_tracks_th, :652-743, held fixed), one forward and one backward
// sweep whatever the number of parameters.  xt_thgrad.h gives a track to ONE lane and keeps every sequence of every step in global memory:
// it is bound by HBM bandwidth (197 GB per evaluation of 1e6 three-state tracks, VALU busy 0.08).  Here a workgroup serves tiles of TT
// tracks of one chunk and the lanes of a track share its sequences:
//   * live state of a step (integrated parents; adjoints of the merged sequences) in LDS, [field][sequence][track];
//   * the forward sweep fuses "merge the members of a group" with "integrate the next position" in registers (as the apply kernel of
//     xt_th.h does) and writes ONE record per (step, merged sequence, track) to the log - the only per-step global traffic;
//   * the backward sweep gives every parent sequence of a step to a lane: re-integration from the log, the adjoint gathered from the
//     groups that took its S^nb_substeps expansions (inverse of the plan's member lists, built per step in LDS), straight back through its own
//     integration - the per-step temporaries of xt_thgrad.h (re-integrated parents, two adjoint buffers per wavefront in global memory) are gone;
//   * table adjoints without per-lane accumulator rows: every (parent, new digits) child leaves its two contributions in LDS, then lane
//     (table entry, track) sums the children that used its entry into REGISTERS it keeps for the whole launch - deterministic, no atomics.
// Output per workgroup: {sum LL, adjoint of the model blob}, reduced and projected on the directions by xt_grad_reduce / xt_rev_project.
#pragma once
#include "xt_thgrad.h"

// per tile: forward 2 state buffers | backward 2 adjoint buffers + the children's contributions (the groups' merged state is read from the log:
// staging it in LDS cost a pass, a barrier and 26 % of the tile's LDS)
XT_HD int64_t xt_thg2_tile_doubles(int capP, int TT, int D, int K, int G)
{
    const int fwd = 2 * (2 + D + K), bwd = 2 * (1 + D + K) + 2 * G;
    return (int64_t)(fwd > bwd ? fwd : bwd) * capP * TT;
}
// plan staging of one step (bytes): member words u32[capE], inverse map u16[capE], group starts u16[capP + 1], newest states u8[2][capP]
XT_HD int64_t xt_thg2_plan_doubles(int capP, int capE) { return (4 * (int64_t)capE + 2 * (int64_t)capE + 2 * ((int64_t)capP + 2) + 2 * (int64_t)capP + 15) / 8 + 1; }
XT_HD int64_t xt_thg2_lds_doubles(int S, int G, int capP, int capE, int TT, int D, int K, int nthreads, int TB)
{
    return ((xt_tab_doubles(S, G) + 1) & ~1) + xt_thg2_tile_doubles(capP, TT, D, K, G) + xt_thg2_plan_doubles(capP, capE) + 2 * (int64_t)nthreads + TB + 4;
}
// scratch of one workgroup: the log (merged state of every step of the tile it is working on) + the seed's parking area
XT_HD int64_t xt_thg2_ws_doubles(int capP, int Lmax, int TT, int D, int K, int G)
{
    return (int64_t)Lmax * capP * TT * (2 + D + K) + (int64_t)capP * TT * (1 + D + K + 2 * G);
}

// adjoint record of one sequence of one track in LDS: [a = d LL / d log z | mb[D] | ub[K]] as planes [field][sequence][track]
template <int D, int K>
struct XtThgAdjT {
    double* base;
    int plane;
    XT_HD double& a(int g, int x, int TT) const { return base[g * TT + x]; }
    XT_HD double& mb(int d, int g, int x, int TT) const { return base[(int64_t)plane * (1 + d) + g * TT + x]; }
    XT_HD double& ub(int k, int g, int x, int TT) const { return base[(int64_t)plane * (1 + D + k) + g * TT + x]; }
};

// NE: table entries (of S * G) a lane accumulates: S * G <= NE * (threads / TT)
template <int D, int K, int NE, class Ctx>
XT_HD void xt_thg2_body(const XtThArgs& a, const XtThGradArgs& ga, Ctx& cx)
{
    typedef XtThView<D, K, false> View;
    constexpr int R = 2 + D + K, RA = 1 + D + K;
    const int S = a.S, G = a.G, capE = a.capE, KS = a.KS, SG = S * G, TT = a.TT, capP = ga.capP;
    const int gch = cx.block() / a.bpc, sub = cx.block() - gch * a.bpc;
    int ch;
    const XtThBucket bk = xt_th_bind(a, gch, ch);
    const int L = bk.L;
    const int tid = cx.tid(), nt = cx.nthreads(), x = tid & (TT - 1), q = tid >> a.logTT, LPT = nt >> a.logTT;
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += nt) smem[i] = a.blob[i];  // per-chunk blobs (per-track time steps) are not served by this kernel
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* TD2 = TAB + 4 * SG;
    const double* T64 = TAB + XT_NTAB * SG;
    const int plane = capP * TT;
    double* tile = smem + ((ntab + 1) & ~1);
    // forward: two buffers of integrated sequences; backward: two adjoint buffers, the merged state of the step's groups, the children's contributions
    View Ya, Yb;
    Ya.base = tile;
    Yb.base = tile + (int64_t)R * plane;
    Ya.plane = Yb.plane = plane;
    double* adj0 = tile;
    double* adj1 = tile + (int64_t)RA * plane;
    double* C0 = tile + 2 * (int64_t)RA * plane;  // [p * G + r][x]: the child's share of d LL / d log T[o]
    double* C1 = C0 + (int64_t)G * plane;      //                 ... of d LL / d d2[o]
    uint32_t* mpkL = (uint32_t*)(tile + xt_thg2_tile_doubles(capP, TT, D, K, G));
    uint16_t* gidxL = (uint16_t*)(mpkL + capE);
    uint16_t* gstL = gidxL + capE;
    uint8_t* gnewL = (uint8_t*)(gstL + capP + 2);  // [2][capP]
    double* red = (double*)(mpkL) + xt_thg2_plan_doubles(capP, capE);  // [nt] + int [nt] + out [1 + TB]
    int* redi = (int*)(red + nt);
    double* out = red + nt + (nt + 1) / 2 + 1;
    for (int i = tid; i < 1 + ga.TB; i += nt) out[i] = 0.0;
    double* LOG = ga.ws + (int64_t)cx.block() * ga.ws_stride;
    auto logv = [&](int t) XT_INL {
        View v;
        v.base = LOG + (int64_t)t * R * plane;
        v.plane = plane;
        return v;
    };
    auto adjv = [&](double* b) XT_INL {
        XtThgAdjT<D, K> v;
        v.base = b;
        v.plane = plane;
        return v;
    };
    cx.sync();

    const int64_t c0 = (int64_t)ch * a.chunk;
    const int n = (int)((bk.N - c0) < a.chunk ? (bk.N - c0) : a.chunk);
    const int ntile = (n + TT - 1) >> a.logTT;
    const uint32_t* mpk_g = bk.mpack + (int64_t)ch * L * capE;
    const uint16_t* gst_g = bk.gstart + (int64_t)ch * L * (capE + 1);
    const uint8_t* gnew_g = bk.gnew + (int64_t)ch * L * capE;
    const int32_t* hdr_g = bk.hdr + (int64_t)ch * L * 2;
    const int tl = L - 1;
    const bool stay_l = tl >= 2 && tl >= a.min_len;
    const int vF = (bk.isBL ? 2 : 0) + (stay_l ? 1 : 0);
    const double* TF = TAB + vF * SG;

    // per-lane accumulators kept over the whole launch
    double my_ll = 0.0, l2acc[K], slacc = 0.0, ofacc = 0.0, accFs = 0.0;
    double accT0[NE], accT1[NE], accTF[NE], accD2[NE];
    for (int k = 0; k < K; ++k) l2acc[k] = 0.0;
    XT_UNROLL
    for (int j = 0; j < NE; ++j) accT0[j] = accT1[j] = accTF[j] = accD2[j] = 0.0;

    for (int tile_i = sub; tile_i < ntile; tile_i += a.bpc) {
        const int64_t first = c0 + ((int64_t)tile_i << a.logTT);
        const int nx = (int)((c0 + n - first) < TT ? (c0 + n - first) : TT);
        const bool act = x < nx;
        const int64_t trk = first + (act ? x : nx - 1);  // idle lanes shadow the tile's last track, their seed is zero
        const double keep = act ? 1.0 : 0.0;
        const double* tp = bk.tracks + trk * L * D;
        const double* sp = a.locerr_mode ? bk.sigma + trk * L * KS : nullptr;
        auto load_pos = [&](int p, double* c) XT_INL {
            for (int d = 0; d < D; ++d) c[d] = tp[p * D + d];
        };
        auto load_l2 = [&](int p, double* l2) XT_INL {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k) l2[k] = xt_th_l2_from_sigma(sp[p * KS + (KS == 1 ? 0 : k)], a.locerr_mode, hdr);
            }
        };
        auto l2_back = [&](int p, int k, double v) XT_INL {  // adjoint of the localisation variance used at position p -> the parameter behind it
            if (a.locerr_mode == 0) {
                l2acc[k] += v;
            } else if (a.locerr_mode == 2) {
                const double sr = sp[p * KS + (KS == 1 ? 0 : k)];
                const double s1 = xt_fma(sr, hdr[3], hdr[4]);
                if (!(s1 < 1e-6)) {  // not clipped (tracking.py:928-930)
                    slacc = xt_fma(v, 2.0 * s1 * sr, slacc);
                    ofacc = xt_fma(v, 2.0 * s1, ofacc);
                }
            }
        };
        auto n_groups = [&](int t) XT_INL { return t >= 1 ? (int)hdr_g[t * 2 + 1] : S; };  // sequences of X_t
        // the children's contributions, summed per table entry into this lane's registers: lane (entry e = q + j * LPT, track x) takes the
        // children (p, r = e % G) of the parents whose newest state is e / G
        auto reduce_tables = [&](int nP, const uint8_t* nw, int variant) XT_INL {
            XT_UNROLL
            for (int j = 0; j < NE; ++j) {
                const int e = q + j * LPT;
                if (e < SG) {
                    const int s = e / G, r = e - s * G;
                    double st = 0.0, sd = 0.0;
                    for (int p = 0; p < nP; ++p)
                        if (nw[p] == s) {
                            st += C0[(p * G + r) * TT + x];
                            sd += C1[(p * G + r) * TT + x];
                        }
                    if (variant == 0) accT0[j] += st;
                    else if (variant == 1) accT1[j] += st;
                    else accTF[j] += st;
                    accD2[j] += sd;
                }
            }
        };

        // ================= forward sweep: X_1 ... X_{L-2} into the log, Y_t = X_t with position t integrated in LDS =================
        {
            double c[D], l2[K];
            load_pos(0, c);
            load_l2(0, l2);
            for (int p = q; p < S; p += LPT) {  // X_0 is used as it is (weight Fs, mean = the first position, variance = its error)
                const int idx = p * TT + x;
                Ya.zm(idx) = hdr[8 + p];
                Ya.ze(idx) = 0;
                for (int d = 0; d < D; ++d) Ya.m(d, idx) = c[d];
                for (int k = 0; k < K; ++k) Ya.u(k, idx) = l2[k];
            }
        }
        View src = Ya, dst = Yb;
        int nPar = S;
        for (int t = 1; t <= L - 2; ++t) {
            const int nE = hdr_g[t * 2], nG = hdr_g[t * 2 + 1];
            for (int i = tid; i < nE; i += nt) mpkL[i] = mpk_g[(int64_t)t * capE + i];
            for (int i = tid; i <= nG; i += nt) gstL[i] = gst_g[(int64_t)t * (capE + 1) + i];
            cx.sync();
            const bool stay = t >= 2 && t >= a.min_len;
            const double* TTl = TAB + (stay ? 1 : 0) * SG;
            const View lg = logv(t);
            double c[D], l2[K];
            load_pos(t, c);
            load_l2(t, l2);
            for (int g = q; g < nG; g += LPT) {
                double W, M[D], U[K];
                int E;
                xt_th_gather_regs<D, K>(src, TT, x, mpkL, (int)gstL[g], (int)gstL[g + 1], TTl, TD2, W, E, M, U);
                const int idx = g * TT + x;
                lg.zm(idx) = W;
                lg.ze(idx) = E;
                for (int d = 0; d < D; ++d) lg.m(d, idx) = M[d];
                for (int k = 0; k < K; ++k) lg.u(k, idx) = U[k];
                xt_th_integrate_store<D, K>(W, E, M, U, c, l2, T64, dst, idx);
            }
            cx.sync();
            const View tmp = src;
            src = dst;
            dst = tmp;
            nPar = nG;
        }
        // ---- last position (+ leaving / bleaching term, tracking.py:611-633): the lanes of a track add up their parents' terms
        uint8_t* nwF = gnewL;  // newest state of the final parents
        for (int p = tid; p < nPar; p += nt) nwF[p] = (uint8_t)(L >= 3 ? gnew_g[(int64_t)(L - 2) * capE + p] : p);
        double cl[D], l2l[K];
        load_pos(tl, cl);
        load_l2(tl, l2l);
        cx.sync();
        const View fin = src;
        XtAcc tot;
        tot.clear();
        for (int g = q; g < nPar; g += LPT) {
            const int idx = g * TT + x;
            const double zq = fin.zm(idx);
            const int eq = fin.ze(idx);
            const int o = (int)nwF[g] * G;
            double dq[D], uq[K], dsq = 0.0;
            for (int d = 0; d < D; ++d) {
                dq[d] = cl[d] - fin.m(d, idx);
                dsq = xt_fma(dq[d], dq[d], dsq);
            }
            for (int k = 0; k < K; ++k) uq[k] = fin.u(k, idx);
            for (int r = 0; r < G; ++r) {
                double quad, gf;
                if (K == 1) {
                    const double rr = xt_rcp(TD2[o + r] + uq[0] + l2l[0]);
                    quad = 0.5 * dsq * rr;
                    gf = xt_pow_half<D>(rr);
                } else {
                    quad = 0.0;
                    gf = 1.0;
                    for (int d = 0; d < D; ++d) {
                        const double rr = xt_rcp(TD2[o + r] + uq[d] + l2l[d]);
                        quad = xt_fma(0.5 * dq[d] * dq[d], rr, quad);
                        gf *= rr;
                    }
                    gf = sqrt(gf);
                }
                double p;
                int j, n2;
                xt_exp_tab(-quad, p, j, n2);
                tot.add(zq * TF[o + r] * (gf * T64[j]) * p, eq + n2);
            }
        }
        red[tid] = tot.m;
        redi[tid] = tot.e;
        cx.sync();
        tot.clear();
        for (int qq = 0; qq < LPT; ++qq) tot.add(red[qq * TT + x], redi[qq * TT + x]);  // fixed order: every lane of the track gets the same total
        const double ll = log(tot.m) + (double)tot.e * XT_LN2 + bk.ll_const;  // a NaN position / error poisons the track (LL and gradient)
        if (act && q == 0) {
            if (bk.ll_out) bk.ll_out[first + x] = ll;
            my_ll += ll;
        }

        // ================= backward sweep =================
        auto integrate_regs = [&](double& z, int& e, double* m, double* u, const double* c, const double* l2) XT_INL {
            double dm[D], dsq = 0.0;
            for (int d = 0; d < D; ++d) {
                dm[d] = c[d] - m[d];
                dsq = xt_fma(dm[d], dm[d], dsq);
            }
            double quad, gf, tt[K];
            if (K == 1) {
                const double r = xt_rcp(l2[0] + u[0]);
                tt[0] = u[0] * r;
                quad = 0.5 * dsq * r;
                gf = xt_pow_half<D>(r);
            } else {
                quad = 0.0;
                gf = 1.0;
                for (int d = 0; d < D; ++d) {
                    const double r = xt_rcp(l2[d] + u[d]);
                    tt[d] = u[d] * r;
                    quad = xt_fma(0.5 * dm[d] * dm[d], r, quad);
                    gf *= r;
                }
                gf = sqrt(gf);
            }
            double p;
            int jj, n2;
            xt_exp_tab(-quad, p, jj, n2);
            const double z0 = z;
            z = z0 * (gf * T64[jj]) * p;
            const int en = e + n2;
            e = (z0 != 0.0 && en > XT_EMIN) ? en : XT_EMIN;
            for (int d = 0; d < D; ++d) m[d] = xt_fma(dm[d], tt[K == 1 ? 0 : d], m[d]);
            for (int k = 0; k < K; ++k) u[k] = l2[k] * tt[k];
        };
        // the same integration backwards: adjoint (ay, mby, uby) of the integrated sequence -> adjoint of the sequence before it (m, u: its
        // mean / variance), stored as entry g of `dst`; the localisation variance of the position gets its share
        auto integrate_back = [&](int pos, const double* m, const double* u, const double* c, const double* l2, double ay, const double* mby,
                                  const double* uby, const XtThgAdjT<D, K>& dsta, int g) XT_INL {
            double dm[D], r[K], mdm[K], dsqk[K];
            for (int k = 0; k < K; ++k) {
                r[k] = xt_rcp(l2[k] + u[k]);
                mdm[k] = 0.0;
                dsqk[k] = 0.0;
            }
            for (int d = 0; d < D; ++d) {
                const int kd = K == 1 ? 0 : d;
                dm[d] = c[d] - m[d];
                mdm[kd] = xt_fma(mby[d], dm[d], mdm[kd]);
                dsqk[kd] = xt_fma(dm[d], dm[d], dsqk[kd]);
                dsta.mb(d, g, x, TT) = xt_fma(ay * dm[d], r[kd], mby[d] * (1.0 - u[kd] * r[kd]));
            }
            dsta.a(g, x, TT) = ay;
            for (int k = 0; k < K; ++k) {
                const double r2 = r[k] * r[k];
                const double gk = ay * r[k] * xt_fma(0.5 * dsqk[k], r[k], K == 1 ? -0.5 * D : -0.5);  // d log z' / d (l2 + u)
                dsta.ub(k, g, x, TT) = gk + mdm[k] * l2[k] * r2 + uby[k] * l2[k] * l2[k] * r2;
                l2_back(pos, k, gk - mdm[k] * u[k] * r2 + uby[k] * u[k] * u[k] * r2);
            }
        };
        // ---- seed: d LL / d (every term of the last position's sum) = term / Z -> adjoint of Y_{L-2}, taken straight through the integration of
        // position L - 2 to the adjoint of X_{L-2} (L >= 3); for two-position tracks it is the adjoint of X_0
        const double rZ = keep * xt_rcp(tot.m);
        XtThgAdjT<D, K> aCur = adjv(adj1), aNxt = adjv(adj0);
        // (adj0 / adj1 / LT / C alias the forward buffers: `fin` = src is Ya or Yb; the seed reads fin and writes aCur + C, so it works out of
        // registers first: every lane reads all it needs of fin, THEN the tile synchronises, THEN the adjoints and contributions are written)
        {
            const View Xl = logv(L >= 3 ? L - 2 : 0);
            double c2[D], l22[K];
            if (L >= 3) {
                load_pos(L - 2, c2);
                load_l2(L - 2, l22);
            }
            // pass 1 (reads fin), results parked behind the log: [g][RA + 2 * G][x]
            double* park = LOG + (int64_t)(a.buckets ? a.Lmax : a.L) * R * plane;
            const int PS = RA + 2 * G;
            for (int g = q; g < nPar; g += LPT) {
                const int idx = g * TT + x;
                const double zq = fin.zm(idx);
                const int eq = fin.ze(idx);
                const int o = (int)nwF[g] * G;
                double dq[D], uq[K], dsq = 0.0, dsqk[K];
                for (int k = 0; k < K; ++k) dsqk[k] = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - fin.m(d, idx);
                    dsq = xt_fma(dq[d], dq[d], dsq);
                    dsqk[K == 1 ? 0 : d] += dq[d] * dq[d];
                }
                for (int k = 0; k < K; ++k) uq[k] = fin.u(k, idx);
                double ag = 0.0, mbg[D], ubg[K];
                for (int d = 0; d < D; ++d) mbg[d] = 0.0;
                for (int k = 0; k < K; ++k) ubg[k] = 0.0;
                double* pk = park + ((int64_t)g * PS) * TT + x;
                for (int r = 0; r < G; ++r) {
                    double quad, gf, rr[K];
                    if (K == 1) {
                        rr[0] = xt_rcp(TD2[o + r] + uq[0] + l2l[0]);
                        quad = 0.5 * dsq * rr[0];
                        gf = xt_pow_half<D>(rr[0]);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        for (int d = 0; d < D; ++d) {
                            rr[d] = xt_rcp(TD2[o + r] + uq[d] + l2l[d]);
                            quad = xt_fma(0.5 * dq[d] * dq[d], rr[d], quad);
                            gf *= rr[d];
                        }
                        gf = sqrt(gf);
                    }
                    double p;
                    int jj, n2;
                    xt_exp_tab(-quad, p, jj, n2);
                    const double term = zq * TF[o + r] * (gf * T64[jj]) * p;
                    const double f = (term == 0.0) ? 0.0 : xt_ldexp(term * rZ, eq + n2 - tot.e);
                    ag += f;
                    double hs = 0.0;
                    for (int k = 0; k < K; ++k) {
                        // d log term / d v_k,  v_k = u_k + d2 + l2_k:  -(D or 1) / (2 v) + dsq_k / (2 v^2)
                        const double h = f * rr[k] * xt_fma(0.5 * dsqk[k], rr[k], K == 1 ? -0.5 * D : -0.5);
                        ubg[k] += h;
                        hs += h;
                        l2_back(tl, k, h);
                    }
                    pk[(int64_t)(RA + r) * TT] = f;
                    pk[(int64_t)(RA + G + r) * TT] = hs;
                    for (int d = 0; d < D; ++d) mbg[d] = xt_fma(f * dq[d], rr[K == 1 ? 0 : d], mbg[d]);
                }
                pk[0] = ag;
                for (int d = 0; d < D; ++d) pk[(int64_t)(1 + d) * TT] = mbg[d];
                for (int k = 0; k < K; ++k) pk[(int64_t)(1 + D + k) * TT] = ubg[k];
            }
            cx.sync();  // fin is dead: its LDS becomes adjoints + contributions
            for (int g = q; g < nPar; g += LPT) {
                const double* pk = park + ((int64_t)g * PS) * TT + x;  // written by this very lane
                const double ag = pk[0];
                double mbg[D], ubg[K];
                for (int d = 0; d < D; ++d) mbg[d] = pk[(int64_t)(1 + d) * TT];
                for (int k = 0; k < K; ++k) ubg[k] = pk[(int64_t)(1 + D + k) * TT];
                for (int r = 0; r < G; ++r) {
                    C0[(g * G + r) * TT + x] = pk[(int64_t)(RA + r) * TT];
                    C1[(g * G + r) * TT + x] = pk[(int64_t)(RA + G + r) * TT];
                }
                if (L >= 3) {
                    const int idx = g * TT + x;
                    double mx[D], ux[K];
                    for (int d = 0; d < D; ++d) mx[d] = Xl.m(d, idx);
                    for (int k = 0; k < K; ++k) ux[k] = Xl.u(k, idx);
                    integrate_back(L - 2, mx, ux, c2, l22, ag, mbg, ubg, aCur, g);
                } else {
                    accFs += ag;  // lane q == g (LPT >= S)
                    for (int k = 0; k < K; ++k) l2_back(0, k, ubg[k]);
                }
            }
            cx.sync();
            reduce_tables(nPar, nwF, 2);
        }
        // ---- s = L-2 ... 1: aCur = adjoint of X_s (its groups).  Every parent p of step s (a sequence of X_{s-1}) is re-integrated in
        // registers, gathers its adjoint from the G groups that took its expansions and is taken straight back through its own integration
        int nwsel = 1;
        for (int s2 = L - 2; s2 >= 1; --s2) {
            const int nE = hdr_g[s2 * 2], nG = n_groups(s2), nPp = n_groups(s2 - 1);
            uint8_t* nw = gnewL + nwsel * capP;
            nwsel ^= 1;
            // staging (the table sums of the previous step may still be reading C and the OTHER newest-state buffer): inverse map, parents'
            // newest states
            for (int g2 = tid; g2 < nG; g2 += nt) {
                const int k0 = gst_g[(int64_t)s2 * (capE + 1) + g2], k1 = gst_g[(int64_t)s2 * (capE + 1) + g2 + 1];
                for (int kk = k0; kk < k1; ++kk) {
                    const uint32_t pk = mpk_g[(int64_t)s2 * capE + kk];
                    gidxL[(int)(pk >> 16) * G + (int)((pk & 0xffffu) % (uint32_t)G)] = (uint16_t)g2;
                }
            }
            (void)nE;
            for (int p = tid; p < nPp; p += nt) nw[p] = (uint8_t)(s2 >= 2 ? gnew_g[(int64_t)(s2 - 1) * capE + p] : p);
            cx.sync();
            const bool stay = s2 >= 2 && s2 >= a.min_len;
            const int vT = stay ? 1 : 0;
            const double* TTl = TAB + vT * SG;
            const View Xp = logv(s2 - 1), LT = logv(s2);  // parents / groups of the step, straight from the log
            double c[D], l2[K];
            load_pos(s2 - 1, c);
            load_l2(s2 - 1, l2);
            for (int p = q; p < nPp; p += LPT) {
                const int pi = p * TT + x;
                double mx[D], ux[K], my[D], uy[K];
                double zy;
                int ey;
                if (s2 >= 2) {
                    zy = Xp.zm(pi);
                    ey = Xp.ze(pi);
                    for (int d = 0; d < D; ++d) my[d] = mx[d] = Xp.m(d, pi);
                    for (int k = 0; k < K; ++k) uy[k] = ux[k] = Xp.u(k, pi);
                    integrate_regs(zy, ey, my, uy, c, l2);
                } else {  // X_0 is not logged
                    zy = hdr[8 + p];
                    ey = 0;
                    for (int d = 0; d < D; ++d) my[d] = mx[d] = c[d];
                    for (int k = 0; k < K; ++k) uy[k] = ux[k] = l2[k];
                }
                const int o0 = (int)nw[p] * G;
                double ap = 0.0, mbp[D], ubp[K];
                for (int d = 0; d < D; ++d) mbp[d] = 0.0;
                for (int k = 0; k < K; ++k) ubp[k] = 0.0;
                for (int r = 0; r < G; ++r) {
                    const int g2 = gidxL[p * G + r], gi = g2 * TT + x, o = o0 + r;
                    const double Wm = LT.zm(gi);
                    double ac = 0.0, ad = 0.0;
                    if (Wm != 0.0) {  // a dead group has no share in the likelihood
                        const double al = xt_ldexp(zy * TTl[o] * xt_rcp(Wm), ey - LT.ze(gi));  // this member's share of its group's weight
                        double cj = aCur.a(g2, x, TT), ubs = 0.0;
                        for (int d = 0; d < D; ++d) {
                            const double Mb = aCur.mb(d, g2, x, TT);
                            cj = xt_fma(Mb, my[d] - LT.m(d, gi), cj);
                            mbp[d] = xt_fma(al, Mb, mbp[d]);
                        }
                        for (int k = 0; k < K; ++k) {
                            const double Ub = aCur.ub(k, g2, x, TT);
                            cj = xt_fma(Ub, uy[k] + TD2[o] - LT.u(k, gi), cj);
                            ubp[k] = xt_fma(al, Ub, ubp[k]);
                            ubs += Ub;
                        }
                        ac = al * cj;
                        ad = al * ubs;
                        ap += ac;
                    }
                    C0[(p * G + r) * TT + x] = ac;
                    C1[(p * G + r) * TT + x] = ad;
                }
                if (s2 >= 2) {
                    integrate_back(s2 - 1, mx, ux, c, l2, ap, mbp, ubp, aNxt, p);
                } else {  // X_0: weight = Fs, mean = first position, variance = l2 of the first position
                    accFs += ap;  // lane q == p (LPT >= S)
                    for (int k = 0; k < K; ++k) l2_back(0, k, ubp[k]);
                }
            }
            cx.sync();
            reduce_tables(nPp, nw, vT);
            const XtThgAdjT<D, K> tmp = aCur;
            aCur = aNxt;
            aNxt = tmp;
        }
        cx.sync();  // the next tile's forward sweep reuses the LDS the table sums read
    }

    // ---- per-workgroup output: {sum LL, adjoint of every blob entry}, lanes summed in a fixed order
    auto all_lanes = [&](double v, int oidx) XT_INL {
        cx.sync();
        red[tid] = v;
        cx.sync();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < nt; ++i) s += red[i];
            out[oidx] += s;
        }
    };
    all_lanes(my_ll, 0);
    for (int k = 0; k < K; ++k) all_lanes(l2acc[k], 1 + k);
    all_lanes(slacc, 1 + 3);
    all_lanes(ofacc, 1 + 4);
    auto per_entry = [&](double v, int j, int obase, int nent) XT_INL {  // lanes (q, x): entry e = q + j * LPT < nent summed over the tracks x
        cx.sync();
        red[tid] = v;
        cx.sync();
        const int e = tid + j * LPT;
        if (tid < LPT && e < nent) {
            double s = 0.0;
            for (int xx = 0; xx < TT; ++xx) s += red[tid * TT + xx];
            out[obase + e] += s;
        }
    };
    per_entry(accFs, 0, 1 + 8, S);
    XT_UNROLL
    for (int j = 0; j < NE; ++j) {
        per_entry(accT0[j], j, 1 + XT_BLOB_HDR + 0 * SG, SG);
        per_entry(accT1[j], j, 1 + XT_BLOB_HDR + 1 * SG, SG);
        per_entry(accTF[j], j, 1 + XT_BLOB_HDR + vF * SG, SG);
        per_entry(accD2[j], j, 1 + XT_BLOB_HDR + 4 * SG, SG);
    }
    cx.sync();
    double* outw = ga.gpartials + (int64_t)cx.block() * (1 + ga.TB);
    for (int e = tid; e < 1 + ga.TB; e += nt) outw[e] = out[e];
}
