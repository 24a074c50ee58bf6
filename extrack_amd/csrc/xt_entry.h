// Entry-parallel variant of the recursion for models with SEVERAL substeps per frame (ns >= 2, BASELINE configs[4]):
// a group then has G = S^ns members (up to 64) but a track has only S^(F-ns) groups, so "one thread per group"
// (xt_kernel.h) leaves most lanes idle and loops serially over the members.  Here ONE THREAD OWNS ONE SEQUENCE:
//   * lanes [g*GP, g*GP + G) of a wave are the members of group g (GP = G rounded up to a power of two <= 64), so the
//     moment-matching merge (max exponent, W, sum z*m, sum z*u) is a log2(GP)-step all-reduce over those lanes - DPP lane
//     permutations inside a row of 16, one ds_swizzle / v_readlane step beyond (DevCtx::group_sum_f64): no LDS storage, no atomics;
//   * after the merge every member expands/integrates ITS OWN new digit combination q and writes its own sequence in
//     place; the only cross-thread hazard is the re-grouping of the next step (one workgroup barrier per position);
//   * the last position enumerates all (old sequence, new digits) pairs: each thread loops over the G digit combinations
//     of its own old sequence (that S^(F+ns)-term sum is inherent to the model, tracking.py:282-306).
// Mathematics, tables and LDS layout are those of xt_kernel.h (log-likelihood only; posteriors need ns == 1).
#pragma once
#include "xt_kernel.h"

// threads reserved per track: NG*GP rounded up to a multiple of 64, or to a power of two when it is below 64
XT_HD int xt_entry_tptp(int NG, int GP)
{
    const int t = NG * GP;
    if (t >= 64) return (t + 63) / 64 * 64;
    int p = 4;
    while (p < t) p <<= 1;
    return p;
}

template <int GP, int D, int K, class Ctx>
XT_HD void xt_entry_body(const XtKernelArgs& a, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    const int G = a.G, S = a.S, E = a.E, EP = a.EP, NG = a.NG, L = b.L;
    const int tid = cx.tid();
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += cx.nthreads()) smem[i] = xt_blob_ptr(a)[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;

    const int TPT = NG * GP;               // threads that own a (group, member) pair
    const int TPTP = xt_entry_tptp(NG, GP);  // threads reserved per track: TPT padded so tracks never straddle a reduction
    const int RW = TPTP < 64 ? TPTP : 64;    // lanes of the track-level butterfly
    const int slot = tid / TPTP;
    const int r = tid - slot * TPTP;
    const int g = r / GP;
    const int q = r - g * GP;
    const bool tvalid = slot < a.TPB;
    const bool qvalid = q < G && r < TPT;
    const int rdoubles = xt_region_doubles(EP, D, K);
    double* reg = smem + ((ntab + 1) & ~1) + (tvalid ? slot : 0) * rdoubles;
    double* zm = reg;
    double* mm = zm + EP;
    double* uu = mm + D * EP;
    int* ze = (int*)(uu + K * EP);
    int* red_e = ze + ((EP + 1) & ~1);
    double* wsum = smem + ((ntab + 1) & ~1) + a.TPB * rdoubles + (tvalid ? slot : 0) * 16;  // per-wave partials of the final sum

    const int prev = (r < TPT ? g : 0) / a.prev_div;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int qq = qvalid ? q : 0;
    cx.sync();
    const double t0q = TAB[(0 * S + prev) * G + qq], t1q = TAB[(1 * S + prev) * G + qq], d2q = TAB[(4 * S + prev) * G + qq];

    double block_ll = 0.0;
    const int64_t nbatch = (b.N + a.TPB - 1) / a.TPB;
    for (int64_t batch = lb; batch < nbatch; batch += nb) {
        const int64_t trk = batch * a.TPB + slot;
        const bool act = tvalid && trk < b.N;
        const double* c = b.tracks + (act ? trk : 0) * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + (act ? trk : 0) * (int64_t)L * a.KS : nullptr;
        auto load_l2 = [&](int pos, double* l2) {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k) {
                    double s = sg[pos * a.KS + (a.KS == 1 ? 0 : k)];
                    if (a.locerr_mode == 2) {
                        s = xt_fma(s, hdr[3], hdr[4]);
                        s = s < 1e-6 ? 1e-6 : s;
                    }
                    l2[k] = s * s;
                }
            }
        };

        // ---- position 0
        if (act) {
            double l20[K], c0[D];
            load_l2(0, l20);
            for (int d = 0; d < D; ++d) c0[d] = c[d];
            for (int il = r; il < E; il += TPTP) {
                const bool live = il < S;
                const int i = xt_skew(il, a.skew);
                zm[i] = live ? hdr[8 + il] : 0.0;
                ze[i] = live ? 0 : XT_EMIN;
                for (int d = 0; d < D; ++d) mm[d * EP + i] = c0[d];
                for (int k = 0; k < K; ++k) uu[k * EP + i] = l20[k];
            }
            if (r == 0) red_e[0] = XT_EMIN;
        }
        cx.sync();

        bool bad = false;  // NaN position / sigma seen by this thread
        // ---- positions 1 .. L-2
        for (int t = 1; t <= L - 2; ++t) {
            const int ph = (t - 1) % a.P;
            int idx = 0;
            double z = 0.0, mq[D], uq[K];
            int e = XT_EMIN;
            for (int d = 0; d < D; ++d) mq[d] = 0.0;
            for (int k = 0; k < K; ++k) uq[k] = 0.0;
            double ct[D], l2t[K];
            for (int d = 0; d < D; ++d) ct[d] = 0.0;
            for (int k = 0; k < K; ++k) l2t[k] = 1.0;
            if (act) {
                for (int d = 0; d < D; ++d) {
                    ct[d] = c[t * D + d];
                    bad = bad || ct[d] != ct[d];
                }
                load_l2(t, l2t);
                for (int k = 0; k < K; ++k) bad = bad || l2t[k] != l2t[k];
                if (qvalid) {
                    idx = xt_skew(a.base_tab[ph * NG + g] + a.off_tab[ph * G + q], a.skew);
                    z = zm[idx];
                    e = ze[idx];
                    for (int d = 0; d < D; ++d) mq[d] = mm[d * EP + idx];
                    for (int k = 0; k < K; ++k) uq[k] = uu[k * EP + idx];
                }
            }
            // group merge: butterfly over the GP lanes of the group (every lane of the wave takes part)
            const int emax = cx.template group_max_i32<GP>(e);
            const double aq = xt_ldexp(z, e - emax);
            double W = cx.template group_sum_f64<GP>(aq), M[D], U[K];
            for (int d = 0; d < D; ++d) M[d] = cx.template group_sum_f64<GP>(aq * mq[d]);
            for (int k = 0; k < K; ++k) U[k] = cx.template group_sum_f64<GP>(aq * uq[k]);
            if (act && qvalid) {
                const double rW = W > 0.0 ? xt_rcp(W) : 0.0;
                for (int d = 0; d < D; ++d) M[d] *= rW;
                for (int k = 0; k < K; ++k) U[k] *= rW;
                const double Wm = xt_frexp_mant(W);
                const int We = W > 0.0 ? emax + xt_frexp_exp(W) : XT_EMIN;
                double dm[D], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dm[d] = ct[d] - M[d];
                    dsq = xt_fma(dm[d], dm[d], dsq);
                }
                double quad, gf, tt[K];
                if (K == 1) {
                    const double s2 = d2q + U[0];
                    const double rr = xt_rcp(l2t[0] + s2);
                    tt[0] = s2 * rr;
                    quad = 0.5 * dsq * rr;
                    gf = xt_pow_half<D>(rr);
                } else {
                    quad = 0.0;
                    gf = 1.0;
                    for (int d = 0; d < D; ++d) {
                        const double s2 = d2q + U[d];
                        const double rr = xt_rcp(l2t[d] + s2);
                        tt[d] = s2 * rr;
                        quad = xt_fma(0.5 * dm[d] * dm[d], rr, quad);
                        gf *= rr;
                    }
                    gf = sqrt(gf);
                }
                double p;
                int j, n;
                xt_exp_tab(-quad, p, j, n);
                const int en = We + n;
                zm[idx] = (Wm * (t >= stay_from ? t1q : t0q)) * (gf * T64[j]) * p;
                ze[idx] = en > XT_EMIN ? en : XT_EMIN;
                for (int d = 0; d < D; ++d) mm[d * EP + idx] = xt_fma(dm[d], tt[K == 1 ? 0 : d], M[d]);
                for (int k = 0; k < K; ++k) uu[k * EP + idx] = l2t[k] * tt[k];
            }
            cx.sync();
        }

        // ---- last position: thread = old sequence Q (its own), loop over the G new digit combinations
        XtAcc tot;
        tot.clear();
        if (act && qvalid) {
            const int tl = L - 1;
            const int ph = (tl - 1) % a.P;
            const int idx = xt_skew(a.base_tab[ph * NG + g] + a.off_tab[ph * G + q], a.skew);
            const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
            const double* TF = TAB + (vfin * S + prev) * G;
            const double* TD2 = TAB + (4 * S + prev) * G;
            double cl[D], l2l[K];
            for (int d = 0; d < D; ++d) {
                cl[d] = c[tl * D + d];
                bad = bad || cl[d] != cl[d] || c[d] != c[d];
            }
            load_l2(tl, l2l);
            for (int k = 0; k < K; ++k) bad = bad || l2l[k] != l2l[k];
            const double zq = zm[idx];
            if (zq != 0.0) {
                const int eq = ze[idx];
                double dq[D], uq[K], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - mm[d * EP + idx];
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                for (int k = 0; k < K; ++k) uq[k] = uu[k * EP + idx];
                for (int j = 0; j < G; ++j) {
                    double quad, gf;
                    if (K == 1) {
                        const double rr = xt_rcp(TD2[j] + uq[0] + l2l[0]);
                        quad = 0.5 * dsq * rr;
                        gf = xt_pow_half<D>(rr);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        for (int d = 0; d < D; ++d) {
                            const double rr = xt_rcp(TD2[j] + uq[d] + l2l[d]);
                            quad = xt_fma(0.5 * dq[d] * dq[d], rr, quad);
                            gf *= rr;
                        }
                        gf = sqrt(gf);
                    }
                    double p;
                    int jj, n;
                    xt_exp_tab(-quad, p, jj, n);
                    tot.add(zq * TF[j] * (gf * T64[jj]) * p, eq + n);
                }
            }
        }
        if (bad) tot.add(NAN, 0);  // NaN input -> NaN likelihood, as in the reference
        // track-level reduction: butterfly over each wave, then the (<= 16) per-wave partials in fixed order
        int fe = tot.m != 0.0 ? tot.e : XT_EMIN;
        for (int m = 1; m < RW; m <<= 1) {
            const int o = cx.shfl_xor_i32(fe, m);
            fe = o > fe ? o : fe;
        }
        if (act && (r & (RW - 1)) == 0 && fe > XT_EMIN) cx.atomic_max_i32(&red_e[0], fe);
        cx.sync();
        const int fE = act ? red_e[0] : XT_EMIN;
        double part = (act && tot.m != 0.0) ? xt_ldexp(tot.m, tot.e - fE) : 0.0;
        for (int m = 1; m < RW; m <<= 1) part += cx.shfl_xor_f64(part, m);
        if (act && (r & (RW - 1)) == 0) wsum[r / RW] = part;
        cx.sync();
        if (act && r == 0) {
            double sum = 0.0;
            for (int i = 0; i < TPTP / RW; ++i) sum += wsum[i];
            const double ll = log(sum) + (double)fE * XT_LN2 + b.ll_const;
            if (b.ll_out) b.ll_out[trk] = ll;
            block_ll += ll;
        }
        cx.sync();
    }

    cx.sync();
    if (tvalid && r == 0) smem[slot] = block_ll;
    cx.sync();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < a.TPB; ++i) s += smem[i];
        a.partials[cx.block()] = s;
    }
}

// geometry helpers ---------------------------------------------------------------------------------------------------------
static inline int xt_entry_gp(int G)
{
    int gp = 4;
    while (gp < G) gp <<= 1;
    return gp;
}
static inline bool xt_use_entry(int NS, int G, int NG, bool preds)
{
    return NS >= 2 && G <= 64 && xt_entry_tptp(NG, xt_entry_gp(G)) <= 1024 && !preds;
}
// tracks per block / threads per block / LDS bytes
static inline void xt_entry_geometry(int S, int G, int E, int NG, int D, int K, int& tpb, int& threads, size_t& lds)
{
    const int tptp = xt_entry_tptp(NG, xt_entry_gp(G));
    const size_t fixed = (size_t)((xt_tab_doubles(S, G) + 1) & ~1) * 8;
    const size_t per_track = (size_t)(xt_region_doubles(xt_padded_entries(E, (S & (S - 1)) == 0), D, K) + 16) * 8;
    int by_threads = tptp >= 256 ? 1 : 256 / tptp;
    int by_lds = 64 * 1024 > fixed + per_track ? (int)((64 * 1024 - fixed) / per_track) : 1;
    tpb = by_threads < by_lds ? by_threads : by_lds;
    if (tpb < 1) tpb = 1;
    threads = (tpb * tptp + 63) / 64 * 64;
    lds = fixed + (size_t)tpb * per_track;
}
