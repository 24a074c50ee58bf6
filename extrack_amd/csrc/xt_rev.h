// Log-likelihood AND its exact gradient by REVERSE-mode differentiation of the fixed-window recursion: one forward sweep over the
// positions of a track that logs the merged state of every group, one backward sweep that carries the adjoint of every live sequence.
// The cost does not depend on the number of free parameters.
//
// What it replaces: the finite differences lmfit's BFGS takes around cum_Proba_Cs (extrack/tracking.py:1371: nvar + 1 evaluations per
// gradient), and - for models with 3 or 4 states - the forward-mode kernels of xt_grad.h / xt_gradr.h, whose work grows with the number
// of directions (13 for a 3-state model: 4 passes that each repeat the primal recursion).
//
// Same mathematics as xt_grad.h (fuse -> expand -> integrate of extrack/tracking.py:109-318 with moment-matching fusion, tracking.py:
// 361-423, and the Gaussian integration of tracking.py:76-98).  A step of group g (one lane):
//     merge    a_Q = z_Q / W,  W = sum_Q z_Q,  m_bar = sum a_Q m_Q,  u_bar = sum a_Q u_Q                       (members Q < G)
//     expand   den_qk = l2_k + d2_q + u_bar_k,  r = 1 / den,  tq = (d2_q + u_bar_k) r,  dn = c_t - m_bar          (new digits q < G)
//              log z'_q = log W + log T_q + log gf(r) - quad,   m'_q = m_bar + dn tq,   u'_q = l2 tq
// Adjoints (lam = d LL / d log z, mu = d LL / d m, nu = d LL / d u) of the outputs q give those of the members Q:
//     adj tq_k  = sum_{d of k} mu'_d dn_d + nu'_k l2_k            adj den_k = lam' A_k - adj tq_k tq_k r_k     (A = d log z' / d den)
//     U_k = sum_q (adj den + adj tq r)_qk                          adj dn_d = sum_q (-lam' dn_d r + mu'_d tq)   M_d = sum_q mu'_d - adj dn_d
//     lam_Q = a_Q (sum_q lam'_q + M . (m_Q - m_bar) + U . (u_Q - u_bar)),   mu_Q = a_Q M,   nu_Q = a_Q U
// and the adjoints of the MODEL TABLES accumulate on the way: log T[v][prev][q] += lam'_q, d2[prev][q] += sum_k (adj den + adj tq r),
// l2_k += adj den_k + nu'_k tq_k (global error; through the clip(s * slope + offset) chain for per-peak errors), log Fs at position 0.
// The kernel returns sum LL and the adjoint of every entry of the model blob (layout of a tangent block, xt_grad.h); the host - a small
// kernel - contracts it with the tangent blocks of the directions: d sum LL / d theta_i = <adjoint, d blob / d theta_i>.
//
// Mapping (as xt_gradr.h): one lane owns one group, members and adjoints are register arrays, the LDS is the exchange medium - the
// outputs (s, q) of lane s are the members (g, Q) of lane g = s / G + q * NG / G, Q = s % G: written at s + q * NG, read at g * G + Q
// (a plain shift register; nothing else of the kernel depends on where a sequence is stored, so the circular digit slots of
// xt_kernel.h are not needed).  The adjoints travel the same way backwards.  What the backward sweep needs of the forward one is the
// MERGED state of every group and step, (W, m_bar, u_bar): (1 + D + K) doubles + one exponent per lane and step, logged to a per-slot
// region of global memory that the workgroup re-reads a few microseconds later (it stays in L2 / Infinity Cache); the members of a group
// are re-expanded from the logs of their G sender groups.
#pragma once
#include "xt_grad.h"

struct XtRevArgs {
    double* gpartials;   // [nblocks][1 + TB]: per-block sum of LL, then the adjoint of every blob entry
    double* log;         // [nblocks * TPB][log_stride] merged-state log of the track a slot is working on
    int64_t log_stride;  // doubles per track slot: (max track length - 2) * xt_rev_step_doubles
    int32_t TB;          // xt_grad_tb_doubles(S, G)
    int32_t pad_;
};
// log of one step of one track: planes Wm, m_bar[D], u_bar[K] of NG doubles, then NG exponents
XT_HD int xt_rev_step_doubles(int NG, int D, int K) { return (1 + D + K) * NG + (NG + 1) / 2; }
XT_HD int xt_rev_xbuf_doubles(int EP, int D, int K) { return EP * (1 + D + K) + (EP + 1) / 2 + 1; }
#define XT_REV_PART 16  // partial sums of the per-track total (two-level fixed-order sum)
// per track slot: two exchange buffers, the partial sums, 2 ints
// nbuf exchange buffers: 2 = one barrier per step; 1 = two barriers per step and half the LDS (taken when that lets a second workgroup on the CU:
// 4 states, 5e5 x 60, frame_len 4: 136 -> 106 ms; where two workgroups fit anyway the second barrier costs 3 %)
XT_HD int xt_rev_track_doubles(int EP, int D, int K, int nbuf) { return nbuf * xt_rev_xbuf_doubles(EP, D, K) + XT_REV_PART + 2; }
// per-lane accumulators: log T (v = 0, 1, final), d2, l2[K], slope, offset, log Fs
XT_HD constexpr int xt_rev_nacc(int G, int K) { return 4 * G + K + 3; }
XT_HD size_t xt_rev_lds_bytes(int S, int G, int EP, int D, int K, int tpb, int threads, int nbuf)
{
    const size_t fixed = (size_t)((xt_tab_doubles(S, G) + 1) & ~1);
    size_t per = (size_t)tpb * ((size_t)xt_rev_track_doubles(EP, D, K, nbuf) + xt_stage_doubles(D));
    const size_t red = (size_t)(xt_rev_nacc(G, K) + 1) * threads;  // block-level reduction of the accumulators (aliases the slots)
    if (red > per) per = red;
    return (fixed + per) * sizeof(double);
}
// the shift-register exchange needs whole sender groups: S^(F - NS) divisible by S^NS
XT_HD bool xt_rev_supported(int G, int NG) { return G >= 2 && G <= 4 && NG >= G && NG % G == 0 && NG <= 256; }

template <int G_, int D, int K, int NBUF, class Ctx>
XT_HD void xt_rev_body(const XtKernelArgs& a, const XtRevArgs& ra, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    constexpr int G = G_, TC = 1 + D + K, NACC = xt_rev_nacc(G_, K);
    const int S = a.S, EP = a.EP, NG = a.NG, L = b.L, TB = ra.TB, NT = cx.nthreads();
    const int tid = cx.tid();
    double* smem = cx.smem();

    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += NT) smem[i] = a.blob[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;
    const int reg0 = (ntab + 1) & ~1;

    const int slot = tid / NG;
    const int g = tid - slot * NG;
    const bool tvalid = slot < a.TPB;
    const int xdoubles = xt_rev_xbuf_doubles(EP, D, K);
    const int tdoubles = xt_rev_track_doubles(EP, D, K, NBUF);
    double* tr0 = smem + reg0 + (tvalid ? slot : 0) * tdoubles;
    double* X[2] = {tr0, tr0 + (NBUF - 1) * xdoubles};
    double* part = tr0 + NBUF * xdoubles;  // [XT_REV_PART]
    int* red_e = (int*)(part + XT_REV_PART);      // [0] exponent of the track total, [1] NaN-input flag
    double* spos = smem + reg0 + a.TPB * tdoubles + (tvalid ? slot : 0) * xt_stage_doubles(D);
    double* ssig = spos + XT_STAGE * D;
    auto XZ = [&](int r) XT_INL { return (int*)(X[r] + TC * EP); };

    // ---- the lane's place in the shift register
    const int NGG = NG / G;                       // member Q of group g was output q_s = g / NGG of group sbase + Q
    const int q_s = g / NGG, sbase = G * (g - q_s * NGG);
    const int prev = g / a.prev_div;
    const int SG = S * G;
    const double* T0 = TAB + (0 * S + prev) * G;
    const double* T1 = TAB + (1 * S + prev) * G;
    const double* TD2 = TAB + (4 * S + prev) * G;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const bool init_lane = g - prev * a.prev_div == 0;  // position 0: the only live sequence of state s is member 0 of group s * prev_div
    int widx[G], ridx[G], sprev[G];
    XT_UNROLL
    for (int q = 0; q < G; ++q) {
        widx[q] = xt_skew(g + q * NG, a.skew);    // as output q of this lane / as adjoint of output q
        ridx[q] = xt_skew(g * G + q, a.skew);     // as member q of this lane
        sprev[q] = (sbase + q) / a.prev_div;
    }
    const int sdoubles = xt_rev_step_doubles(NG, D, K);
    double* LOG = ra.log + ((int64_t)cx.block() * a.TPB + (tvalid ? slot : 0)) * ra.log_stride;

    // ---- accumulators of the block (registers)
    double acc[NACC], ll_acc = 0.0;
    XT_UNROLL
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    constexpr int A_T = 0, A_TF = 2 * G, A_D2 = 3 * G, A_L2 = 4 * G, A_SL = 4 * G + K, A_OF = 4 * G + K + 1, A_F = 4 * G + K + 2;

    if (tvalid && g == 0) red_e[1] = 0;
    cx.sync();

    const int64_t nbatch = (b.N + a.TPB - 1) / a.TPB;
    for (int64_t batch = lb; batch < nbatch; batch += nb) {
        const int64_t trk = batch * a.TPB + slot;
        const bool act = tvalid && trk < b.N;
        const double* c = b.tracks + (act ? trk : 0) * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + (act ? trk : 0) * (int64_t)L * a.KS : nullptr;
        int cur_blk = -1;

        // positions [p0, p0 + XT_STAGE) of the track -> LDS (barrier before: nobody still reads the block being replaced)
        auto ensure = [&](int pos) XT_INL {
            if ((pos >> 5) == cur_blk) return;
            cur_blk = pos >> 5;
            const int p0 = cur_blk << 5;
            cx.sync();
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
        auto load_c = [&](int pos, double* ct) XT_INL {
            XT_UNROLL
            for (int d = 0; d < D; ++d) ct[d] = spos[(pos & (XT_STAGE - 1)) * D + d];
        };
        // l2[k] of position pos; sc[k], sraw[k]: d l2[k] = sc[k] * (sraw[k] * d slope + d offset) (mode 2)
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
        // adjoint of l2[k] of a position -> the parameters behind it
        auto put_l2 = [&](const double* al2, const double* sc, const double* sraw) XT_INL {
            if (a.locerr_mode == 0) {
                XT_UNROLL
                for (int k = 0; k < K; ++k) acc[A_L2 + k] += act ? al2[k] : 0.0;
            } else {
                XT_UNROLL
                for (int k = 0; k < K; ++k) {
                    const double v = al2[k] * sc[k];  // (a lane without a track holds stale, possibly non-finite values: select, do not multiply)
                    acc[A_SL] += act ? v * sraw[k] : 0.0;
                    acc[A_OF] += act ? v : 0.0;
                }
            }
        };
        // one new sequence: merged state (Wm 2^We, mb, ub) of a group, table entries (Tq, d2) of the new digits, position c, error l2
        auto expand = [&](double Wm, int We, const double* mb, const double* ub, double Tq, double d2, const double* ct, const double* l2,
                          double& zo, int& eo, double* mo, double* uo) XT_INL {
            double dn[D], quad = 0.0, gf = 1.0, dsq = 0.0, tq[K];
            XT_UNROLL
            for (int d = 0; d < D; ++d) {
                dn[d] = ct[d] - mb[d];
                dsq = xt_fma(dn[d], dn[d], dsq);
            }
            if (K == 1) {
                const double s2 = d2 + ub[0];
                const double r = xt_rcp(l2[0] + s2);
                tq[0] = s2 * r;
                quad = 0.5 * dsq * r;
                gf = xt_pow_half<D>(r);
            } else {
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    const double s2 = d2 + ub[d];
                    const double r = xt_rcp(l2[d] + s2);
                    tq[d] = s2 * r;
                    quad = xt_fma(0.5 * dn[d] * dn[d], r, quad);
                    gf *= r;
                }
                gf = sqrt(gf);
            }
            double pp;
            int jt, n;
            xt_exp_tab(-quad, pp, jt, n);
            const int en = We + n;
            zo = (Wm * Tq) * (gf * T64[jt]) * pp;
            eo = en > XT_EMIN ? en : XT_EMIN;
            XT_UNROLL
            for (int d = 0; d < D; ++d) mo[d] = xt_fma(dn[d], tq[K == 1 ? 0 : d], mb[d]);
            XT_UNROLL
            for (int k = 0; k < K; ++k) uo[k] = l2[k] * tq[k];
        };
        // the members of the lane's group at step 1: the initial state in member 0 of the groups s * prev_div, zero weight elsewhere
        auto init_members = [&](const double* c0, const double* l20, double* zm, int* ze, double (*mm)[D], double (*uu)[K]) XT_INL {
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                const bool live = Q == 0 && init_lane;
                zm[Q] = live ? hdr[8 + (live ? prev : 0)] : 0.0;
                ze[Q] = live ? 0 : XT_EMIN;
                XT_UNROLL
                for (int d = 0; d < D; ++d) mm[Q][d] = c0[d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) uu[Q][k] = l20[k];
            }
        };
        // merge of the members: normalised weights aj, merged mean / variance, total weight Wm 2^We
        auto merge = [&](const double* zm, const int* ze, const double (*mm)[D], const double (*uu)[K], double* aj, double* mb, double* ub,
                         double& Wm, int& We) XT_INL {
            int emax = XT_EMIN;
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) emax = ze[Q] > emax ? ze[Q] : emax;
            double W = 0.0;
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
            const bool liveW = W > 0.0;
            const double rW = liveW ? xt_rcp(W) : 0.0;
            XT_UNROLL
            for (int d = 0; d < D; ++d) mb[d] *= rW;
            XT_UNROLL
            for (int k = 0; k < K; ++k) ub[k] *= rW;
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) aj[Q] *= rW;
            Wm = xt_frexp_mant(W);
            We = liveW ? emax + xt_frexp_exp(W) : XT_EMIN;
        };

        double zm[G], mm[G][D], uu[G][K];
        int ze[G];
        ensure(0);
        {
            double c0[D], l20[K], sc0[K], sr0[K];
            load_c(0, c0);
            load_l2(0, l20, sc0, sr0);
            init_members(c0, l20, zm, ze, mm, uu);
            if (act && g == 0) red_e[0] = XT_EMIN;
        }

        // =========================== forward sweep: positions 1 .. L-2, merged state of every step -> log
        for (int t = 1; t <= L - 2; ++t) {
            ensure(t);
            double ct[D], l2t[K], sct[K], srt[K], aj[G], mb[D], ub[K], Wm;
            int We;
            load_c(t, ct);
            load_l2(t, l2t, sct, srt);
            merge(zm, ze, mm, uu, aj, mb, ub, Wm, We);
#if defined(XT_REV_DIAG) && (XT_REV_DIAG & 2)
            if (act && t == 1) {
#else
            if (act) {
#endif
                double* lg = LOG + (int64_t)(t - 1) * sdoubles;
                lg[g] = Wm;
                XT_UNROLL
                for (int d = 0; d < D; ++d) lg[(1 + d) * NG + g] = mb[d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) lg[(1 + D + k) * NG + g] = ub[k];
                ((int*)(lg + TC * NG))[g] = We;
            }
            const double* TTl = t >= stay_from ? T1 : T0;
            double* xb = X[t & 1];
            int* xz = XZ(t & 1);
            XT_UNROLL
            for (int q = 0; q < G; ++q) {
                double zo, mo[D], uo[K];
                int eo;
                expand(Wm, We, mb, ub, TTl[q], TD2[q], ct, l2t, zo, eo, mo, uo);
                xb[widx[q]] = zo;
                xz[widx[q]] = eo;
                XT_UNROLL
                for (int d = 0; d < D; ++d) xb[(1 + d) * EP + widx[q]] = mo[d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) xb[(1 + D + k) * EP + widx[q]] = uo[k];
            }
            cx.sync();
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                zm[Q] = xb[ridx[Q]];
                ze[Q] = xz[ridx[Q]];
                XT_UNROLL
                for (int d = 0; d < D; ++d) mm[Q][d] = xb[(1 + d) * EP + ridx[Q]];
                XT_UNROLL
                for (int k = 0; k < K; ++k) uu[Q][k] = xb[(1 + D + k) * EP + ridx[Q]];
            }
            if (NBUF == 1) cx.sync();
        }

        // =========================== last position (+ leaving / bleaching term): LL of the track and the seeds of the backward sweep
        const int tl = L - 1;
        const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
        const double* TF = TAB + (vfin * S + prev) * G;
        double lam[G], mu[G][D], nu[G][K];  // adjoints of the members
        {
            ensure(tl);
            double cl[D], l2l[K], scl[K], srl[K];
            double wmP[G][G], rP[G][G][K], dqP[G][D], dsqP[G];
            int weP[G][G];
            XtAcc tot;
            tot.clear();
            load_c(tl, cl);
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
            if (act && tot.m != 0.0) cx.atomic_max_i32(&red_e[0], tot.e);
            cx.sync();
            // two-level fixed-order sum of the groups' totals on the common scale 2^fe: gth -> XT_REV_PART partial sums -> every lane adds them up
            double* gth = X[(L - 1) & 1];
            const int fe = red_e[0];
            if (act) gth[g] = tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0;
            cx.sync();
            if (act && g < XT_REV_PART) {
                double s2 = 0.0;
                for (int i = g; i < NG; i += XT_REV_PART) s2 += gth[i];
                part[g] = s2;
            }
            cx.sync();
            double sw = 0.0;
            for (int i = 0; i < XT_REV_PART && i < NG; ++i) sw += part[i];
            const bool poisoned = red_e[1] != 0;
            if (act && g == 0) {
                const double ll = poisoned ? NAN : log(sw) + (double)fe * XT_LN2 + b.ll_const;
                if (b.ll_out) b.ll_out[trk] = ll;
                ll_acc += ll;
            }
            const double rs = poisoned ? NAN : 1.0 / sw;
            double al2[K];
            XT_UNROLL
            for (int k = 0; k < K; ++k) al2[k] = 0.0;
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                lam[Q] = 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) mu[Q][d] = 0.0;
                XT_UNROLL
                for (int k = 0; k < K; ++k) nu[Q][k] = 0.0;
                XT_UNROLL
                for (int q = 0; q < G; ++q) {
                    const double wsc = wmP[Q][q] != 0.0 ? xt_ldexp(wmP[Q][q], weP[Q][q] - fe) : 0.0;
                    const double pi = wsc != 0.0 ? wsc * rs : 0.0;
                    lam[Q] += pi;
                    acc[A_TF + q] += act ? pi : 0.0;
                    double sB = 0.0;
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) {
                        const double r = rP[Q][q][k];
                        const double B = K == 1 ? -0.5 * r * xt_fma(-dsqP[Q], r, (double)D) : -0.5 * r * xt_fma(-(dqP[Q][k] * dqP[Q][k]), r, 1.0);
                        const double pB = pi * B;
                        nu[Q][k] += pB;
                        al2[k] += pB;
                        sB += pB;
                    }
                    acc[A_D2 + q] += act ? sB : 0.0;
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) mu[Q][d] = xt_fma(pi * dqP[Q][d], rP[Q][q][K == 1 ? 0 : d], mu[Q][d]);
                }
            }
            put_l2(al2, scl, srl);
        }

        // =========================== backward sweep: positions L-2 .. 1
        double ctn[D], l2n[K], scn[K], srn[K];  // position t of the coming iteration
        if (L >= 3) {
            ensure(L - 2);
            load_c(L - 2, ctn);
            load_l2(L - 2, l2n, scn, srn);
        }
        for (int t = L - 2; t >= 1; --t) {
            // the member adjoints (lam, mu, nu) of step t + 1 go back to the lanes that produced those members: adjoints of the outputs of step t
            double* xb = X[t & 1];
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                xb[ridx[Q]] = lam[Q];
                XT_UNROLL
                for (int d = 0; d < D; ++d) xb[(1 + d) * EP + ridx[Q]] = mu[Q][d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) xb[(1 + D + k) * EP + ridx[Q]] = nu[Q][k];
            }
            // meanwhile: the members of step t, re-expanded from the logs of the sender groups at step t - 1
            double ct[D], l2t[K], sct[K], srt[K];
            XT_UNROLL
            for (int d = 0; d < D; ++d) ct[d] = ctn[d];
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                l2t[k] = l2n[k];
                sct[k] = scn[k];
                srt[k] = srn[k];
            }
            ensure(t - 1);
            load_c(t - 1, ctn);
            load_l2(t - 1, l2n, scn, srn);
            if (t == 1) {
                init_members(ctn, l2n, zm, ze, mm, uu);
            } else {
#if defined(XT_REV_DIAG) && (XT_REV_DIAG & 1)
                const double* lg = LOG;
#else
                const double* lg = LOG + (int64_t)(t - 2) * sdoubles;
#endif
                const double* TTs = TAB + ((t - 1 >= stay_from ? 1 : 0) * S) * G;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) {
                    const int s = sbase + Q;
                    double mbs[D], ubs[K];
                    const double Wms = lg[s];
                    const int Wes = ((const int*)(lg + TC * NG))[s];
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) mbs[d] = lg[(1 + d) * NG + s];
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) ubs[k] = lg[(1 + D + k) * NG + s];
                    expand(Wms, Wes, mbs, ubs, TTs[sprev[Q] * G + q_s], TAB[(4 * S + sprev[Q]) * G + q_s], ctn, l2n, zm[Q], ze[Q], mm[Q], uu[Q]);
                }
            }
            double aj[G], mb[D], ub[K], Wm;
            int We;
            merge(zm, ze, mm, uu, aj, mb, ub, Wm, We);
            double dn[D], dsq = 0.0;
            XT_UNROLL
            for (int d = 0; d < D; ++d) {
                dn[d] = ct[d] - mb[d];
                dsq = xt_fma(dn[d], dn[d], dsq);
            }
            cx.sync();
            // adjoints of the outputs of step t -> of its members, and of the tables on the way
            const bool st = t >= stay_from;
            double Lam = 0.0, U[K], adn[D], Ms[D], al2[K];
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                U[k] = 0.0;
                al2[k] = 0.0;
            }
            XT_UNROLL
            for (int d = 0; d < D; ++d) {
                adn[d] = 0.0;
                Ms[d] = 0.0;
            }
            XT_UNROLL
            for (int q = 0; q < G; ++q) {
                const double lq = xb[widx[q]];
                double mq[D], nq[K], r[K], tq[K], atq[K];
                XT_UNROLL
                for (int d = 0; d < D; ++d) mq[d] = xb[(1 + d) * EP + widx[q]];
                XT_UNROLL
                for (int k = 0; k < K; ++k) nq[k] = xb[(1 + D + k) * EP + widx[q]];
                const double d2 = TD2[q];
                XT_UNROLL
                for (int k = 0; k < K; ++k) {
                    const double s2 = d2 + ub[k];
                    r[k] = xt_rcp(l2t[k] + s2);
                    tq[k] = s2 * r[k];
                    atq[k] = nq[k] * l2t[k];
                }
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    const int kk = K == 1 ? 0 : d;
                    atq[kk] = xt_fma(mq[d], dn[d], atq[kk]);
                    adn[d] = xt_fma(mq[d], tq[kk], xt_fma(-lq * dn[d], r[kk], adn[d]));
                    Ms[d] += mq[d];
                }
                double sd = 0.0;
                XT_UNROLL
                for (int k = 0; k < K; ++k) {
                    const double A = K == 1 ? -0.5 * r[k] * xt_fma(-dsq, r[k], (double)D) : -0.5 * r[k] * xt_fma(-(dn[k] * dn[k]), r[k], 1.0);
                    const double aden = xt_fma(lq, A, -atq[k] * tq[k] * r[k]);
                    const double as2 = xt_fma(atq[k], r[k], aden);
                    U[k] += as2;
                    sd += as2;
                    al2[k] += xt_fma(nq[k], tq[k], aden);
                }
                Lam += lq;
                acc[A_T + q] += (act && !st) ? lq : 0.0;
                acc[A_T + G + q] += (act && st) ? lq : 0.0;
                acc[A_D2 + q] += act ? sd : 0.0;
            }
            put_l2(al2, sct, srt);
            double M[D];
            XT_UNROLL
            for (int d = 0; d < D; ++d) M[d] = Ms[d] - adn[d];
            XT_UNROLL
            for (int Q = 0; Q < G; ++Q) {
                double v = Lam;
                XT_UNROLL
                for (int d = 0; d < D; ++d) v = xt_fma(M[d], mm[Q][d] - mb[d], v);
                XT_UNROLL
                for (int k = 0; k < K; ++k) v = xt_fma(U[k], uu[Q][k] - ub[k], v);
                lam[Q] = aj[Q] * v;
                XT_UNROLL
                for (int d = 0; d < D; ++d) mu[Q][d] = aj[Q] * M[d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) nu[Q][k] = aj[Q] * U[k];
            }
            if (NBUF == 1) cx.sync();
        }
        // ---- position 0: initial fractions and the localisation error of the first position
        if (L >= 3) {
            double al2[K];
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                al2[k] = 0.0;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) al2[k] += nu[Q][k];
            }
            put_l2(al2, scn, srn);
            acc[A_F] += (act && init_lane) ? lam[0] : 0.0;
        } else {
            // two positions: the members of the last position ARE the initial state
            double c0[D], l20[K], sc0[K], sr0[K], al2[K];
            ensure(0);
            load_c(0, c0);
            load_l2(0, l20, sc0, sr0);
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                al2[k] = 0.0;
                XT_UNROLL
                for (int Q = 0; Q < G; ++Q) al2[k] += nu[Q][k];
            }
            put_l2(al2, sc0, sr0);
            acc[A_F] += (act && init_lane) ? lam[0] : 0.0;
        }
        cx.sync();
        if (act && g == 0) red_e[1] = 0;
    }

    // ---- block sums in a fixed order: every lane's accumulators -> LDS, one output column per thread
    cx.sync();
    double* racc = smem + reg0;  // [NACC + 1][NT] (the track slots are idle now)
    XT_UNROLL
    for (int i = 0; i < NACC; ++i) racc[i * NT + tid] = tvalid ? acc[i] : 0.0;
    racc[NACC * NT + tid] = tvalid ? ll_acc : 0.0;
    cx.sync();
    const int vfin_b = (b.isBL ? 2 : 0) + (L - 1 >= stay_from ? 1 : 0);
    const int nlanes = a.TPB * NG;
    for (int col = tid; col < 1 + TB; col += NT) {
        double s = 0.0;
        if (col == 0) {
            for (int i = 0; i < nlanes; ++i) s += racc[NACC * NT + i];
        } else {
            const int idx = col - 1;
            int row = -1, row2 = -1, pv = -1, lane_g = -1;  // pv >= 0: lanes whose prev digit is pv; lane_g >= 0: that group only
            if (idx < XT_BLOB_HDR) {
                if (idx < K && a.locerr_mode == 0) row = A_L2 + idx;
                else if (idx == 3) row = A_SL;
                else if (idx == 4) row = A_OF;
                else if (idx >= 8 && idx < 8 + S) {
                    row = A_F;
                    lane_g = (idx - 8) * a.prev_div;
                }
            } else if (idx < XT_BLOB_HDR + XT_NTAB * SG) {
                const int j = idx - XT_BLOB_HDR, v = j / SG, pq = j - v * SG;
                pv = pq / G;
                const int q = pq - pv * G;
                if (v < 2) row = A_T + v * G + q;
                if (v == vfin_b) row2 = A_TF + q;
                if (v == 4) row = A_D2 + q;
            }
            if (row >= 0 || row2 >= 0)
                for (int i = 0; i < nlanes; ++i) {
                    const int gi = i % NG;
                    if (pv >= 0 && gi / a.prev_div != pv) continue;
                    if (lane_g >= 0 && gi != lane_g) continue;
                    if (row >= 0) s += racc[row * NT + i];
                    if (row2 >= 0) s += racc[row2 * NT + i];
                }
        }
        ra.gpartials[(int64_t)cx.block() * (1 + TB) + col] = s;
    }
}
