// Threshold-fusion log-likelihood AND its exact gradient at a FROZEN plan (round 4): the kernel body shared by the HIP kernel
// (extrack_thgrad.hip) and by the CPU-thread emulator used in tests (tests/emul).
//
// What it differentiates (reference, relative to /root/reference/): the objective extrack.tracking.param_fitting minimises in v1.6.3,
//   extrack/tracking.py:1371 lmfit.minimize(cum_Proba_Cs) -> :991 cum_Proba_Cs -> :769 Proba_Cs -> :427-650 P_Cs_inter_bound_stats_th,
// whose merge groups (fuse_tracks_th, :652-743) are decided from the pilot tracks of a chunk.  With the groups of an evaluation frozen
// (the chunk's plan, written by xt_th_plan_body) the value is a smooth function of the model tables; this kernel returns that value and
// the adjoint of every table entry by one forward and one backward sweep, whatever the number of parameters.  lmfit's BFGS differences the
// objective instead: nvar + 1 evaluations per iteration of a function that is only piecewise smooth.
//
// How it is organised for CDNA4:
//   * ONE LANE PER TRACK, a wavefront = 64 tracks of ONE chunk.  Every plan / table index is wave-uniform (scalar loads through the
//     constant address space, as in the wave-uniform apply kernel); the vector unit works on per-track state only, there is no cross-lane
//     traffic and no barrier inside the sweeps.
//   * State lives in a per-wavefront region of global memory laid out [step][sequence][field][lane]: every access of a wavefront is one
//     coalesced 512-byte row.  The forward sweep keeps the MERGED state of every step (the log: weight as mantissa + exponent, mean,
//     variance), the backward sweep re-integrates a step's parents from the log of the step before (one exp per parent and step) and
//     carries the adjoints (d LL / d log z, d LL / d m, d LL / d u) from the last position to the first.
//   * The adjoints of the model tables (log T of the four table variants, d2, log Fs; localisation variance / slope / offset in registers)
//     are accumulated per lane in LDS rows [entry][lane] (conflict-free, deterministic), summed over the lanes at the end of the kernel.
//   Output per wavefront: {sum LL, adjoint of the model blob}; xt_grad_reduce sums the wavefronts in a fixed order and xt_rev_project
//   contracts the blob adjoint with the directions' tangent blocks (extrack_grad.hip).
#pragma once
#include "xt_th.h"

struct XtThGradArgs {
    double* ws;          // scratch, ws_stride doubles per wavefront of the launch
    int64_t ws_stride;
    double* gpartials;   // [wavefronts of the launch][1 + TB]: {sum LL, adjoint of blob[0 .. TB)}
    int32_t TB;          // XT_BLOB_HDR + XT_NTAB * S * G
    int32_t capP;        // sequences per step the scratch is laid out for: max(S, largest group count of the launch's chunks)
    int32_t rows_global; // 1: the table-adjoint rows live at the end of the wavefront's scratch region instead of in LDS (more wavefronts per CU)
};

// LDS rows of one wavefront: the blob entries that can receive an adjoint, compacted (l2[0..2], slope, offset | log Fs[S] | tables) + sum LL
XT_HD int xt_thg_rows(int S, int G) { return 5 + S + XT_NTAB * S * G + 1; }
XT_HD int xt_thg_row_of_blob(int i, int S)  // -1: the slot has no adjoint
{
    if (i < 5) return i;
    if (i >= 8 && i < 8 + S) return 5 + (i - 8);
    if (i >= XT_BLOB_HDR) return 5 + S + (i - XT_BLOB_HDR);
    return -1;
}
XT_HD int64_t xt_thg_lds_doubles(int S, int G, int waves, int capE, bool rows_global = false)
{
    // tables | accumulator rows per wavefront | per wavefront: group index of every expanded sequence of the step being walked back (u16)
    return ((xt_tab_doubles(S, G) + 1) & ~1) + (rows_global ? 0 : (int64_t)waves * xt_thg_rows(S, G) * 64) + (int64_t)waves * (((capE + 3) & ~3) / 4);
}
// scratch of one wavefront: the log (Lmax - 1 steps) + the re-integrated parents + two adjoint buffers
XT_HD int64_t xt_thg_ws_doubles(int capP, int Lmax, int D, int K, int rows = 0)
{
    return (int64_t)Lmax * xt_th_buf_doubles(capP * 64, D, K) + 2 * (int64_t)capP * 64 * (1 + D + K) + (int64_t)rows * 64;
}

// adjoint record of one sequence of one track: [a = d LL / d log z | mb[D] | ub[K]] as rows of 64 lanes
template <int D, int K>
struct XtThgAdj {
    double* base;
    XT_HD double& a(int g, int x) const { return base[((int64_t)g * (1 + D + K)) * 64 + x]; }
    XT_HD double& mb(int d, int g, int x) const { return base[((int64_t)g * (1 + D + K) + 1 + d) * 64 + x]; }
    XT_HD double& ub(int k, int g, int x) const { return base[((int64_t)g * (1 + D + K) + 1 + D + k) * 64 + x]; }
};

// One merge group of one track gathered straight from the LOG of the step before: every member's parent is integrated with position `c`
// in registers on the way (tracking.py:76-98), so the integrated sequences Y never go through memory - the kernel is bound by the traffic of its
// global state, the G-fold repeated exponential is free (VALU busy 0.08).  ONE pass over the members with a running exponent: the sums are
// rescaled by exact powers of two, so the result is the one xt_th_gather_regs computes from a stored Y.
template <int D, int K, class V>
XT_HD void xt_thg_load_integrated(const V& X, int idx, const double* c, const double* l2, const double* T64, double& z, int& e, double* m, double* u)
{
    z = X.zm(idx);
    e = X.ze(idx);
    for (int d = 0; d < D; ++d) m[d] = X.m(d, idx);
    for (int k = 0; k < K; ++k) u[k] = X.u(k, idx);
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
    int j, n;
    xt_exp_tab(-quad, p, j, n);
    const double z0 = z;
    z = z0 * (gf * T64[j]) * p;
    const int en = e + n;
    e = (z0 != 0.0 && en > XT_EMIN) ? en : XT_EMIN;
    for (int d = 0; d < D; ++d) m[d] = xt_fma(dm[d], tt[K == 1 ? 0 : d], m[d]);
    for (int k = 0; k < K; ++k) u[k] = l2[k] * tt[k];
}
template <int D, int K, class V, class MemP, class TabP>
XT_HD void xt_thg_gather_int(const V& X, int xoff, MemP members, int k0, int k1, TabP TT, TabP TD2, const double* c, const double* l2,
                             const double* T64, double& W, int& E, double* M, double* U)
{
    if (k1 - k0 == 1) {
        const uint32_t pk = members[k0];
        const int o = (int)(pk & 0xffffu);
        double z;
        xt_thg_load_integrated<D, K>(X, (int)(pk >> 16) * 64 + xoff, c, l2, T64, z, E, M, U);
        W = z * TT[o];
        for (int k = 0; k < K; ++k) U[k] += TD2[o];
    } else {
        E = XT_EMIN;
        W = 0.0;
        for (int d = 0; d < D; ++d) M[d] = 0.0;
        for (int k = 0; k < K; ++k) U[k] = 0.0;
        for (int kk = k0; kk < k1; ++kk) {
            const uint32_t pk = members[kk];
            const int o = (int)(pk & 0xffffu);
            double z, m[D], u[K];
            int e;
            xt_thg_load_integrated<D, K>(X, (int)(pk >> 16) * 64 + xoff, c, l2, T64, z, e, m, u);
            if (e > E) {  // a larger exponent: what has been summed moves to the new scale (exact; from XT_EMIN the zero sums stay zero)
                const int sh = E - e;
                W = xt_ldexp(W, sh);
                for (int d = 0; d < D; ++d) M[d] = xt_ldexp(M[d], sh);
                for (int k = 0; k < K; ++k) U[k] = xt_ldexp(U[k], sh);
                E = e;
            }
            const double av = xt_ldexp(z * TT[o], e - E);  // hugely negative shift saturates to 0
            W += av;
            for (int d = 0; d < D; ++d) M[d] = xt_fma(av, m[d], M[d]);
            for (int k = 0; k < K; ++k) U[k] = xt_fma(av, TD2[o] + u[k], U[k]);
        }
        const double rW = (W == 0.0) ? 0.0 : xt_rcp(W);
        for (int d = 0; d < D; ++d) M[d] *= rW;
        for (int k = 0; k < K; ++k) U[k] *= rW;
    }
    const bool live = W != 0.0;
    E = live ? E + xt_frexp_exp(W) : XT_EMIN;
    W = xt_frexp_mant(W);
}

// RG: the table-adjoint rows live in the wavefront's scratch region (global memory) instead of LDS - a compile-time choice so that the
// LDS variant keeps LDS-typed accesses (a run-time select makes them flat: measured +20 %)
template <int D, int K, bool RG, class Ctx>
XT_HD void xt_thg_body(const XtThArgs& a, const XtThGradArgs& ga, Ctx& cx)
{
    typedef XtThView<D, K, false> View;
    const int S = a.S, G = a.G, capE = a.capE, KS = a.KS, SG = S * G;
    const int Lmax = a.buckets ? a.Lmax : a.L;
    const int gch = cx.block() / a.bpc, sub = cx.block() - gch * a.bpc;
    int ch;
    const XtThBucket bk = xt_th_bind(a, gch, ch);
    const int L = bk.L;
    const int tid = cx.tid(), nt = cx.nthreads(), x = tid & 63, wv = tid >> 6, NW = nt >> 6;
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    const double* blob_c = a.blob;  // per-chunk blobs (per-track time steps) are not served by this kernel
    for (int i = tid; i < ntab; i += nt) smem[i] = blob_c[i];
    const double* hdr = smem;
    const double* TABl = smem + XT_BLOB_HDR;
    const double* T64 = TABl + XT_NTAB * SG;
    const int NR = xt_thg_rows(S, G);
    const int capP = ga.capP;
    const int64_t bufd = xt_th_buf_doubles(capP * 64, D, K);
    double* wsw = ga.ws + ((int64_t)cx.block() * NW + wv) * ga.ws_stride;
    // this wavefront's accumulators [NR][64]: LDS, or the tail of its scratch region
    double* rows = RG ? wsw + (int64_t)Lmax * bufd + 2 * (int64_t)capP * 64 * (1 + D + K) : smem + ((ntab + 1) & ~1) + (int64_t)wv * NR * 64;
    for (int r = 0; r < NR; ++r) rows[r * 64 + x] = 0.0;
    const int rFs = 5, rT = 5 + S, rD2 = 5 + S + 4 * SG, rLL = NR - 1;
    // wave-uniform reads: the plan and the expansion tables through the constant address space (scalar loads on the device)
    typedef XtCPtr<true, double> CD;
    typedef XtCPtr<true, uint32_t> CU32;
    typedef XtCPtr<true, uint16_t> CU16;
    typedef XtCPtr<true, uint8_t> CU8;
    typedef XtCPtr<true, int32_t> CI32;
    const typename CD::type TAB = CD::make(blob_c + XT_BLOB_HDR);
    const typename CD::type TD2 = TAB + 4 * SG;
    cx.sync();

    auto logv = [&](int t) XT_INL {
        View v;
        v.base = wsw + (int64_t)t * bufd;
        v.plane = capP * 64;
        return v;
    };
    View Y;
    Y.base = wsw + (int64_t)(Lmax - 1) * bufd;
    Y.plane = capP * 64;
    XtThgAdj<D, K> adjA, adjB;
    adjA.base = wsw + (int64_t)Lmax * bufd;
    adjB.base = adjA.base + (int64_t)capP * 64 * (1 + D + K);

    const int64_t c0 = (int64_t)ch * a.chunk;
    const int n = (int)((bk.N - c0) < a.chunk ? (bk.N - c0) : a.chunk);
    const int ntile = (n + 63) >> 6;
    const typename CU32::type mpk_g = CU32::make(bk.mpack + (int64_t)ch * L * capE);
    const typename CU16::type gst_g = CU16::make(bk.gstart + (int64_t)ch * L * (capE + 1));
    const typename CU8::type gnew_g = CU8::make(bk.gnew + (int64_t)ch * L * capE);
    const typename CI32::type hdr_g = CI32::make(bk.hdr + (int64_t)ch * L * 2);

    double my_ll = 0.0;
    double l2acc[K], slacc = 0.0, ofacc = 0.0;  // adjoints of the global localisation variance / of slope and offset (affine per-peak errors)
    for (int k = 0; k < K; ++k) l2acc[k] = 0.0;

    for (int tile = sub * NW + wv; tile < ntile; tile += a.bpc * NW) {
        const int64_t first = c0 + ((int64_t)tile << 6);
        const int nx = (int)((c0 + n - first) < 64 ? (c0 + n - first) : 64);
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
        // adjoint of the localisation variance used at position p -> the parameter behind it
        auto l2_back = [&](int p, int k, double v) XT_INL {
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
        // Y_t = X_t with position t integrated (t >= 1); X_0 is used as it is
        auto integrate_all = [&](int t, int np) XT_INL {
            const View X = logv(t);
            double c[D], l2[K];
            load_pos(t, c);
            load_l2(t, l2);
            for (int g = 0; g < np; ++g) {
                const int idx = g * 64 + x;
                double m[D], u[K];
                for (int d = 0; d < D; ++d) m[d] = X.m(d, idx);
                for (int k = 0; k < K; ++k) u[k] = X.u(k, idx);
                xt_th_integrate_store<D, K>(X.zm(idx), X.ze(idx), m, u, c, l2, T64, Y, idx);
            }
        };

        // ================= forward sweep: X_0 ... X_{L-2} into the log =================
        {
            const View X0 = logv(0);
            double c[D], l2[K];
            load_pos(0, c);
            load_l2(0, l2);
            for (int g = 0; g < S; ++g) {
                const int idx = g * 64 + x;
                X0.zm(idx) = hdr[8 + g];
                X0.ze(idx) = 0;
                for (int d = 0; d < D; ++d) X0.m(d, idx) = c[d];
                for (int k = 0; k < K; ++k) X0.u(k, idx) = l2[k];
            }
        }
        int nPar = S;
        for (int t = 1; t <= L - 2; ++t) {
            const View dst = logv(t);
            const int nG = hdr_g[t * 2 + 1];
            const typename CU32::type mem = mpk_g + (int64_t)t * capE;
            const typename CU16::type gst = gst_g + (int64_t)t * (capE + 1);
            const bool stay = t >= 2 && t >= a.min_len;
            const typename CD::type TTl = TAB + (stay ? 1 : 0) * SG;
            if (t >= 2) {  // X_t from X_{t-1}: position t - 1 integrated into every member's parent on the way (Y_{t-1} is never stored)
                const View src = logv(t - 1);
                double c[D], l2[K];
                load_pos(t - 1, c);
                load_l2(t - 1, l2);
                for (int g2 = 0; g2 < nG; ++g2) {
                    double W, M[D], U[K];
                    int E;
                    xt_thg_gather_int<D, K>(src, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TD2, c, l2, T64, W, E, M, U);
                    const int di = g2 * 64 + x;
                    dst.zm(di) = W;
                    dst.ze(di) = E;
                    for (int d = 0; d < D; ++d) dst.m(d, di) = M[d];
                    for (int k = 0; k < K; ++k) dst.u(k, di) = U[k];
                }
            } else {  // X_1 from X_0, which is used as it is
                for (int g2 = 0; g2 < nG; ++g2) xt_th_gather<D, K>(logv(0), 64, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TD2, dst, g2 * 64 + x);
            }
            nPar = nG;
        }
        // ---- last position (+ leaving / bleaching term, tracking.py:611-633)
        const int tl = L - 1;
        if (L >= 3) integrate_all(L - 2, nPar);
        const View fin = L >= 3 ? Y : logv(0);
        const bool stay_l = tl >= 2 && tl >= a.min_len;
        const int vF = (bk.isBL ? 2 : 0) + (stay_l ? 1 : 0);
        const typename CD::type TF = TAB + vF * SG;
        double cl[D], l2l[K];
        load_pos(tl, cl);
        load_l2(tl, l2l);
        XtAcc tot;
        tot.clear();
        for (int g = 0; g < nPar; ++g) {
            const int idx = g * 64 + x;
            const double zq = fin.zm(idx);
            const int eq = fin.ze(idx);
            const int o = (L >= 3 ? (int)gnew_g[(int64_t)(L - 2) * capE + g] : g) * G;
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
        const double ll = log(tot.m) + (double)tot.e * XT_LN2 + bk.ll_const;  // a NaN position / error poisons the track (LL and gradient)
        if (act) {
            if (bk.ll_out) bk.ll_out[first + x] = ll;
            my_ll += ll;
        }

        // ================= backward sweep =================
        // Gaussian integration of position `pos` into one sequence, in registers (xt_th_integrate_store without the store)
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
                                  const double* uby, const XtThgAdj<D, K>& dst, int g) XT_INL {
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
                // m' = m + dm tt, log z' has -dm^2 r / 2:  d/dm = a dm r + mb' (1 - tt)
                dst.mb(d, g, x) = xt_fma(ay * dm[d], r[kd], mby[d] * (1.0 - u[kd] * r[kd]));
            }
            dst.a(g, x) = ay;
            for (int k = 0; k < K; ++k) {
                const double r2 = r[k] * r[k];
                const double gk = ay * r[k] * xt_fma(0.5 * dsqk[k], r[k], K == 1 ? -0.5 * D : -0.5);  // d log z' / d (l2 + u)
                dst.ub(k, g, x) = gk + mdm[k] * l2[k] * r2 + uby[k] * l2[k] * l2[k] * r2;
                l2_back(pos, k, gk - mdm[k] * u[k] * r2 + uby[k] * u[k] * u[k] * r2);
            }
        };
        // ---- seed: d LL / d (every term of the last position's sum) = term / Z -> adjoint of Y_{L-2}, taken straight through the integration of
        // position L - 2 to the adjoint of X_{L-2} (L >= 3); for two-position tracks it is the adjoint of X_0
        const double rZ = keep * xt_rcp(tot.m);
        {
            const View Xl = logv(L >= 3 ? L - 2 : 0);
            double c2[D], l22[K];
            if (L >= 3) {
                load_pos(L - 2, c2);
                load_l2(L - 2, l22);
            }
            for (int g = 0; g < nPar; ++g) {
                const int idx = g * 64 + x;
                const double zq = fin.zm(idx);
                const int eq = fin.ze(idx);
                const int o = (L >= 3 ? (int)gnew_g[(int64_t)(L - 2) * capE + g] : g) * G;
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
                    rows[(rT + vF * SG + o + r) * 64 + x] += f;
                    double hs = 0.0;
                    for (int k = 0; k < K; ++k) {
                        // d log term / d v_k,  v_k = u_k + d2 + l2_k:  -(D or 1) / (2 v) + dsq_k / (2 v^2)
                        const double h = f * rr[k] * xt_fma(0.5 * dsqk[k], rr[k], K == 1 ? -0.5 * D : -0.5);
                        ubg[k] += h;
                        hs += h;
                        l2_back(tl, k, h);
                    }
                    rows[(rD2 + o + r) * 64 + x] += hs;
                    for (int d = 0; d < D; ++d) mbg[d] = xt_fma(f * dq[d], rr[K == 1 ? 0 : d], mbg[d]);
                }
                if (L >= 3) {
                    double mx[D], ux[K];
                    for (int d = 0; d < D; ++d) mx[d] = Xl.m(d, idx);
                    for (int k = 0; k < K; ++k) ux[k] = Xl.u(k, idx);
                    integrate_back(L - 2, mx, ux, c2, l22, ag, mbg, ubg, adjB, g);
                } else {
                    rows[(rFs + g) * 64 + x] += ag;
                    for (int k = 0; k < K; ++k) l2_back(0, k, ubg[k]);
                }
            }
        }
        // ---- s = L-2 ... 1: adjB = adjoint of X_s (its groups).  Every parent p of step s (a sequence of X_{s-1}) is re-integrated in
        // registers, gathers its adjoint from the G groups that took its expansions (the inverse of the plan's member lists, built per step
        // in LDS), and is taken straight back through its own integration: one pass per step, the adjoint of Y_{s-1} is never stored
        uint16_t* gidx = (uint16_t*)(smem + ((ntab + 1) & ~1) + (RG ? 0 : (int64_t)NW * NR * 64)) + (int64_t)wv * ((capE + 3) & ~3);
        XtThgAdj<D, K> aCur = adjB, aNxt = adjA;
        for (int s2 = L - 2; s2 >= 1; --s2) {
            const int nG = n_groups(s2), nPp = n_groups(s2 - 1);
            const typename CU32::type mem = mpk_g + (int64_t)s2 * capE;
            const typename CU16::type gst = gst_g + (int64_t)s2 * (capE + 1);
            cx.wave_sync();  // the previous step's reads of the map are done
            for (int g2 = x; g2 < nG; g2 += 64)
                for (int kk = gst[g2]; kk < (int)gst[g2 + 1]; ++kk) {
                    const uint32_t pk = mem[kk];
                    gidx[(int)(pk >> 16) * G + (int)((pk & 0xffffu) % (uint32_t)G)] = (uint16_t)g2;
                }
            cx.wave_sync();
            const bool stay = s2 >= 2 && s2 >= a.min_len;
            const int vT = stay ? 1 : 0;
            const typename CD::type TTl = TAB + vT * SG;
            const View Xs = logv(s2), Xp = logv(s2 - 1);
            double c[D], l2[K];
            if (s2 >= 2) {
                load_pos(s2 - 1, c);
                load_l2(s2 - 1, l2);
            }
            for (int p = 0; p < nPp; ++p) {
                const int pi = p * 64 + x;
                double mx[D], ux[K], my[D], uy[K];
                double zy = Xp.zm(pi);
                int ey = Xp.ze(pi);
                for (int d = 0; d < D; ++d) my[d] = mx[d] = Xp.m(d, pi);
                for (int k = 0; k < K; ++k) uy[k] = ux[k] = Xp.u(k, pi);
                if (s2 >= 2) integrate_regs(zy, ey, my, uy, c, l2);
                const int o0 = (s2 >= 2 ? (int)gnew_g[(int64_t)(s2 - 1) * capE + p] : p) * G;
                double ap = 0.0, mbp[D], ubp[K];
                for (int d = 0; d < D; ++d) mbp[d] = 0.0;
                for (int k = 0; k < K; ++k) ubp[k] = 0.0;
                for (int r = 0; r < G; ++r) {
                    const int g2 = gidx[p * G + r], gi = g2 * 64 + x, o = o0 + r;
                    const double Wm = Xs.zm(gi);
                    if (Wm == 0.0) continue;  // a dead group has no share in the likelihood
                    const double al = xt_ldexp(zy * TTl[o] * xt_rcp(Wm), ey - Xs.ze(gi));  // this member's share of its group's weight
                    double cj = aCur.a(g2, x), ubs = 0.0;
                    for (int d = 0; d < D; ++d) {
                        const double Mb = aCur.mb(d, g2, x);
                        cj = xt_fma(Mb, my[d] - Xs.m(d, gi), cj);
                        mbp[d] = xt_fma(al, Mb, mbp[d]);
                    }
                    for (int k = 0; k < K; ++k) {
                        const double Ub = aCur.ub(k, g2, x);
                        cj = xt_fma(Ub, uy[k] + TD2[o] - Xs.u(k, gi), cj);
                        ubp[k] = xt_fma(al, Ub, ubp[k]);
                        ubs += Ub;
                    }
                    const double ac = al * cj;
                    ap += ac;
                    rows[(rT + vT * SG + o) * 64 + x] += ac;
                    rows[(rD2 + o) * 64 + x] += al * ubs;
                }
                if (s2 >= 2) {
                    integrate_back(s2 - 1, mx, ux, c, l2, ap, mbp, ubp, aNxt, p);
                } else {  // X_0: weight = Fs, mean = first position, variance = l2 of the first position
                    rows[(rFs + p) * 64 + x] += ap;
                    for (int k = 0; k < K; ++k) l2_back(0, k, ubp[k]);
                }
            }
            const XtThgAdj<D, K> tmp = aCur;
            aCur = aNxt;
            aNxt = tmp;
        }
    }

    // ---- per-wavefront output: lanes summed in a fixed order
    for (int k = 0; k < K; ++k) rows[k * 64 + x] = l2acc[k];
    rows[3 * 64 + x] = slacc;
    rows[4 * 64 + x] = ofacc;
    rows[rLL * 64 + x] = my_ll;
    cx.sync();
    double* outw = ga.gpartials + ((int64_t)cx.block() * NW + wv) * (1 + ga.TB);
    for (int e = x; e < 1 + ga.TB; e += 64) {
        const int r = e == 0 ? rLL : xt_thg_row_of_blob(e - 1, S);
        double s = 0.0;
        if (r >= 0)
            for (int l = 0; l < 64; ++l) s += rows[r * 64 + l];
        outw[e] = s;
    }
}
