// Register-resident path of the track-likelihood recursion for TWO-STATE models with one substep (S = 2, ns = 1,
// frame_len 4..7): log-likelihood (NP = 0) and log-likelihood + exact gradient along NP model directions (NP > 0).
//
// Reference: extrack/tracking.py:109-318 (P_Cs_inter_bound_stats), :76-98 (log_integrale_dif), :361-423 (fuse_tracks_general);
// the gradient replaces the finite differences lmfit's BFGS takes around cum_Proba_Cs (tracking.py:1371).  Same mathematics as
// xt_fast2.h (primal) and xt_grad.h (tangents: rz = d log z, dm, du); what differs is where the state lives:
//
//   * a lane owns one GROUP = the two sequences that differ only in their OLDEST state digit, held in VGPRs
//     {z, e, m[D], u[K]} x 2 (+ per direction {rz, dm[D], du[K]} x 2).  The other F - 1 digits of the lane's sequences are lane-id
//     bits.  A step merges the two members (the oldest digit is fused away), expands by the new digit q in {0, 1} and integrates
//     the position: the lane now holds two sequences that differ in their NEWEST digit;
//   * then ONE lane exchange over the lane bit that holds the oldest remaining digit brings that digit into the registers and
//     parks the newest one in the lane bit (a 2 x 2 transpose between the two lanes of a pair).  The exchanged bit rotates through
//     the F - 1 group bits, so the step loop is unrolled over F - 1 phases with compile-time exchange instructions:
//       lane bits 0, 1, 3  -> one DPP move per dword (quad_perm / row_ror:8): the lane computes the output it KEEPS (new digit =
//                             its own bit) and the one it SENDS (the other digit) - no selects;
//       lane bit 2         -> two DPP moves (row_shl:4 / row_shr:4 under bank masks); only frame_len 7 uses it;
//       lane bits 4, 5     -> v_permlane16_swap / v_permlane32_swap (gfx950): the transpose itself, outputs in natural order;
//   * nothing of the recursion goes through LDS: xt_fast2.h's step pays 10 ds_read + 10 ds_write per wave-step (a
//     ds_write_b64 occupies the CU's LDS pipe for ~6 cycles), and tangents in LDS (xt_grad.h) are LDS-bandwidth bound: 8 doubles
//     per direction and group there and back ~ 64 LDS cycles against ~40 fp64 operations.  LDS only holds the model tables
//     (read per step at lane-dependent [prev][q] addresses), the staged positions and the per-slot accumulators.
//
// Digit bookkeeping: step t (position t) exchanges group bit GB[(t-1) mod (F-1)]; the newest old digit ("prev" of the transition)
// sits in GB[(t-2) mod (F-1)].  Position 0 puts the initial state into GB[F-2]; dummy digits (time < 0) are zero-weight.
#pragma once
#include "xt_fast2.h"
#include "xt_grad.h"


template <int F>
struct XtR2Geom {
    static constexpr int NGB = F - 1;     // group bits
    static constexpr int NG = 1 << NGB;   // lanes per track
    static constexpr int TPW = 64 / NG;   // tracks per wave
};
// lane bit of group bit i (i = exchange order).  frame_len 7 needs all six lane bits; shorter windows skip bit 2 (two DPP moves).
XT_HD constexpr int xt_r2_gbit(int F, int i) { return F == 7 ? i : (i < 2 ? i : i + 1); }
XT_HD constexpr int xt_r2_gmask(int F) { return F == 7 ? 63 : (F == 6 ? 0x3b : (F == 5 ? 0x1b : 0x0b)); }
template <int F>
XT_HD int xt_r2_slot(int lane)
{
    if (F == 7) return 0;
    int ts = (lane >> 2) & 1;
    if (F == 5) ts |= ((lane >> 5) & 1) << 1;
    if (F == 4) ts |= ((lane >> 4) & 3) << 1;
    return ts;
}

// LDS map (bytes): [model blob 1 KiB (as xt_fast2.h: tables, T64, NaN flags)] [1024-entry exp table 8 KiB] [tangent blocks NP x TB] [staged positions] [accumulators]
#define XT_R2_TAN0 XT_F2_TAB_BYTES
#define XT_R2_TB 36  // xt_grad_tb_doubles(2, 2)
#define XT_R2_MAXU 4  // uniform directions served by one launch (on top of its NP full ones)
XT_HD int xt_r2_pos0(int NP) { return XT_R2_TAN0 + NP * XT_R2_TB * 8; }
XT_HD int xt_r2_acc0(int NP, int D, int KS, int tpw) { return xt_r2_pos0(NP) + XT_F2_WAVES * tpw * XT_F2_CHUNK * (D + KS) * 8; }
XT_HD int xt_r2_block_bytes(int NP, int D, int KS, int tpw) { return xt_r2_acc0(NP, D, KS, tpw) + XT_F2_WAVES * 8 * (NP + 3) * 8; }
#define XT_R2_TAB0 (XT_BLOB_HDR * 8)  // byte address of table v = 0; table v at + v * 32, entry [prev][q] at + (prev * 2 + q) * 8

static inline bool xt_use_reg2(int S, int NS, int F) { return S == 2 && NS == 1 && F >= 4 && F <= 7; }

template <int D, int K, int NP>
struct XtR2Lane {
    double z[2];
    int e[2];
    double m[2][D], u[2][K];
    double rz[NP ? NP : 1][2], dm[NP ? NP : 1][2][D], du[NP ? NP : 1][2][K];
};

// One recursion step at compile-time phase H (the loop is unrolled over the F - 1 phases: every exchange instruction and lane bit is an
// immediate).  tab: byte address of the transition table in use (T or T * stay).  A variant with a run-time phase - one step body, the
// exchange picked by a switch - was measured 1.8x SLOWER with 7 directions (16.4 -> 43.8 ms on C2: the switches cut the step into
// basic blocks the scheduler cannot overlap).
template <int F, int D, int K, int NP, int H, bool ZF, bool LAZY, int VAR, class Ctx>
XT_HD void xt_r2_step(Ctx& cx, char* lds, XtR2Lane<D, K, NP>& s, int tab, const double* c, const double* l2)
{
    constexpr int NGB = F - 1;
    constexpr int XB = xt_r2_gbit(F, H), PB = xt_r2_gbit(F, (H + NGB - 1) % NGB);
    const int lane = xt_opaque(cx.lane());  // keeps the per-phase constants (prev, qa, table offsets) out of the loop-invariant registers
    const int prev = (lane >> PB) & 1;
    const int qa = Ctx::template pair_natural<XB>() ? 0 : (lane >> XB) & 1;
    const int io[2] = {(prev * 2 + qa) * 8, (prev * 2 + (qa ^ 1)) * 8};  // [prev][q] byte offsets of the two outputs
    double TT[2], TD2[2];
    XT_UNROLL
    for (int q = 0; q < 2; ++q) {
        TT[q] = xt_at<double>(lds, tab + io[q]);
        TD2[q] = xt_at<double>(lds, XT_R2_TAB0 + 4 * 32 + io[q]);
    }

    // ---- primal: merge, expand, integrate (xt_f2_step with the members in registers)
    const int e0 = s.e[0], e1 = s.e[1];
    const int emax = e0 > e1 ? e0 : e1;
    const double w0 = xt_ldexp(s.z[0], e0 - emax), w1 = xt_ldexp(s.z[1], e1 - emax);
    const double W = w0 + w1;
    double M[D], U[K];
    XT_UNROLL
    for (int d = 0; d < D; ++d) M[d] = xt_fma(w1, s.m[1][d], w0 * s.m[0][d]);
    XT_UNROLL
    for (int k = 0; k < K; ++k) U[k] = xt_fma(w1, s.u[1][k], w0 * s.u[0][k]);
    const bool live = ZF ? true : W > 0.0;
    const double Ws = ZF ? W : (live ? W : 1.0);
    constexpr bool RN = !LAZY || (H % XT_F2_RENORM) == 0;
    const double Wm = RN ? xt_frexp_mant(W) : W;
    const int We = live ? (RN ? emax + xt_frexp_exp(W) : emax) : XT_EMIN;

    double Dq[2][K];
    XT_UNROLL
    for (int q = 0; q < 2; ++q)
        XT_UNROLL
        for (int k = 0; k < K; ++k) Dq[q][k] = xt_fma(Ws, l2[k] + TD2[q], U[k]);
    double rD[2][K], rW;
    if (K == 1) {
        const double d01 = Dq[0][0] * Dq[1][0];
        const double R = xt_rcp(Ws * d01);
        const double RW = R * Ws;
        rW = R * d01;
        rD[0][0] = RW * Dq[1][0];
        rD[1][0] = RW * Dq[0][0];
    } else {
        double f[2 * K + 1], pre[2 * K + 2], suf[2 * K + 2];
        f[0] = Ws;
        XT_UNROLL
        for (int q = 0; q < 2; ++q)
            XT_UNROLL
            for (int k = 0; k < K; ++k) f[1 + q * K + k] = Dq[q][k];
        pre[0] = 1.0;
        XT_UNROLL
        for (int i = 0; i < 2 * K + 1; ++i) pre[i + 1] = pre[i] * f[i];
        suf[2 * K + 1] = 1.0;
        XT_UNROLL
        for (int i = 2 * K; i >= 0; --i) suf[i] = suf[i + 1] * f[i];
        const double R = xt_rcp(pre[2 * K + 1]);
        rW = R * suf[1];
        XT_UNROLL
        for (int q = 0; q < 2; ++q)
            XT_UNROLL
            for (int k = 0; k < K; ++k) rD[q][k] = R * pre[1 + q * K + k] * suf[2 + q * K + k];
    }
    double dmW[D], dsqW = 0.0;
    XT_UNROLL
    for (int d = 0; d < D; ++d) {
        dmW[d] = xt_fma(c[d], Ws, -M[d]);
        if (K == 1) dsqW = xt_fma(dmW[d], dmW[d], dsqW);
    }
    double x[2], gf[2], tt[2][K];
    const double A = -0.5 * rW * dsqW;
    XT_UNROLL
    for (int q = 0; q < 2; ++q) {
        if (K == 1) {
            x[q] = A * rD[q][0];
            tt[q][0] = xt_fma(Ws, TD2[q], U[0]) * rD[q][0];
            gf[q] = xt_pow_half<D>(Ws * rD[q][0]);
        } else {
            double xx = 0.0, gg = 1.0;
            XT_UNROLL
            for (int d = 0; d < D; ++d) {
                xx = xt_fma(dmW[d] * dmW[d], rD[q][d], xx);
                tt[q][d] = xt_fma(Ws, TD2[q], U[d]) * rD[q][d];
                gg *= Ws * rD[q][d];
            }
            x[q] = xx * (-0.5 * rW);
            gf[q] = sqrt(gg);
        }
    }
    double p[2];
    int j[2], n[2];
    xt_exp_tab_x2(x[0], x[1], p[0], p[1], j[0], j[1], n[0], n[1]);
    double nz[2], nm[2][D], nu[2][K];
    int ne[2];
    XT_UNROLL
    for (int q = 0; q < 2; ++q) {
        int en = We + n[q];
        const double tj = xt_at<double>(lds, XT_F2_EXPB_OFF + j[q] * 8);
        double zn = (Wm * TT[q]) * (gf[q] * tj) * p[q];
        if (!LAZY) {
            en += xt_frexp_exp(zn);
            zn = xt_frexp_mant(zn);
        }
        nz[q] = zn;
        ne[q] = en > XT_EMIN ? en : XT_EMIN;
        XT_UNROLL
        for (int d = 0; d < D; ++d) nm[q][d] = xt_fma(dmW[d], tt[q][K == 1 ? 0 : d], M[d]) * rW;
        XT_UNROLL
        for (int k = 0; k < K; ++k) nu[q][k] = l2[k] * tt[q][k];
    }

    // ---- tangents (formulas: header of xt_grad.h), all on normalised quantities; first the primal factors they share (kept few:
    // with 7-8 directions x 16 VGPRs of tangent state the step must fit the rest into ~100 registers)
    double a0 = 0.0, cm[D], cu[K], dn[D], Aq[2][K], rq[2][K];
    if (NP > 0) {
        a0 = w0 * rW;  // a0 = a1 = 0 for an all-zero group (rW = 1 then)
        const double aa = a0 * (w1 * rW);
        XT_UNROLL
        for (int d = 0; d < D; ++d) {
            cm[d] = aa * (s.m[0][d] - s.m[1][d]);
            dn[d] = dmW[d] * rW;  // c - m_bar
        }
        XT_UNROLL
        for (int k = 0; k < K; ++k) cu[k] = aa * (s.u[0][k] - s.u[1][k]);
        XT_UNROLL
        for (int q = 0; q < 2; ++q)
            XT_UNROLL
            for (int k = 0; k < K; ++k) {
                rq[q][k] = Ws * rD[q][k];  // 1 / den
                if (K == 1)
                    Aq[q][0] = -0.5 * rq[q][0] * xt_fma(-(dsqW * rW * rW), rq[q][0], (double)D);
                else
                    Aq[q][k] = -0.5 * rq[q][k] * xt_fma(-(dn[k] * dn[k]), rq[q][k], 1.0);
            }
    }
    // ---- the lane exchange: the oldest remaining digit comes into the registers, the new digit goes to lane bit XB (primal state
    // first: that frees the registers of the new state before the tangent passes)
    cx.template pair_exchange<XB>(nz[0], nz[1]);
    cx.template pair_exchange_i32<XB>(ne[0], ne[1]);
    XT_UNROLL
    for (int d = 0; d < D; ++d) cx.template pair_exchange<XB>(nm[0][d], nm[1][d]);
    XT_UNROLL
    for (int k = 0; k < K; ++k) cx.template pair_exchange<XB>(nu[0][k], nu[1][k]);
    XT_UNROLL
    for (int q = 0; q < 2; ++q) {
        s.z[q] = nz[q];
        s.e[q] = ne[q];
        XT_UNROLL
        for (int d = 0; d < D; ++d) s.m[q][d] = nm[q][d];
        XT_UNROLL
        for (int k = 0; k < K; ++k) s.u[q][k] = nu[q][k];
    }
    if (NP > 0) {
        XT_UNROLL
        for (int pp = 0; pp < NP; ++pp) {
            if (!(VAR & 1)) xt_sched_fence();  // one direction at a time: interleaved directions multiply the live temporaries
            const int tb = XT_R2_TAN0 + pp * XT_R2_TB * 8;
            double dl[K];
            XT_UNROLL
            for (int k = 0; k < K; ++k) dl[k] = xt_at<double>(lds, tb + k * 8);
            const double del = s.rz[pp][0] - s.rz[pp][1];
            const double R = xt_fma(a0, del, s.rz[pp][1]);
            double dmb[D], dub[K];
            XT_UNROLL
            for (int d = 0; d < D; ++d) dmb[d] = xt_fma(cm[d], del, xt_fma(a0, s.dm[pp][0][d] - s.dm[pp][1][d], s.dm[pp][1][d]));
            XT_UNROLL
            for (int k = 0; k < K; ++k) dub[k] = xt_fma(cu[k], del, xt_fma(a0, s.du[pp][0][k] - s.du[pp][1][k], s.du[pp][1][k]));
            double hd[K];  // -1/2 d |c - m_bar|^2 = sum_d (c - m_bar)_d d m_bar_d (per dim when K == D)
            if (K == 1) {
                hd[0] = 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) hd[0] = xt_fma(dn[d], dmb[d], hd[0]);
            } else {
                XT_UNROLL
                for (int d = 0; d < D; ++d) hd[d] = dn[d] * dmb[d];
            }
            double trz[2], tdm[2][D], tdu[2][K];
            XT_UNROLL
            for (int q = 0; q < 2; ++q) {
                const double dlT = xt_at<double>(lds, tb + tab + io[q]);
                const double dd2 = xt_at<double>(lds, tb + XT_R2_TAB0 + 4 * 32 + io[q]);
                double rzn = R + dlT, dtt[K];
                XT_UNROLL
                for (int k = 0; k < K; ++k) {
                    const double ds2 = dd2 + dub[k], dden = dl[k] + ds2;
                    dtt[k] = rq[q][k] * xt_fma(-tt[q][k], dden, ds2);
                    rzn = xt_fma(Aq[q][k], dden, rzn);
                    rzn = xt_fma(rq[q][k], hd[k], rzn);  // - r/2 * d|c - m_bar|^2
                    tdu[q][k] = xt_fma(l2[k], dtt[k], dl[k] * tt[q][k]);
                }
                trz[q] = ZF ? rzn : (live ? rzn : 0.0);
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    const int kk = K == 1 ? 0 : d;
                    tdm[q][d] = xt_fma(dn[d], dtt[kk], xt_fma(-tt[q][kk], dmb[d], dmb[d]));  // d m_bar (1 - tt) + (c - m_bar) d tt
                }
            }
            cx.template pair_exchange<XB>(trz[0], trz[1]);
            XT_UNROLL
            for (int d = 0; d < D; ++d) cx.template pair_exchange<XB>(tdm[0][d], tdm[1][d]);
            XT_UNROLL
            for (int k = 0; k < K; ++k) cx.template pair_exchange<XB>(tdu[0][k], tdu[1][k]);
            XT_UNROLL
            for (int q = 0; q < 2; ++q) {
                s.rz[pp][q] = trz[q];
                XT_UNROLL
                for (int d = 0; d < D; ++d) s.dm[pp][q][d] = tdm[q][d];
                XT_UNROLL
                for (int k = 0; k < K; ++k) s.du[pp][q][k] = tdu[q][k];
            }
        }
    }
}

// all-reduce over the lanes of a track (the F - 1 group bits)
template <int F, class Ctx>
XT_HD double xt_r2_gsum(Ctx& cx, double v)
{
    constexpr int NGB = F - 1;
    v += cx.template xor_f64<xt_r2_gbit(F, 0)>(v);
    v += cx.template xor_f64<xt_r2_gbit(F, 1)>(v);
    v += cx.template xor_f64<xt_r2_gbit(F, 2)>(v);
    if (NGB > 3) v += cx.template xor_f64<xt_r2_gbit(F, NGB > 3 ? 3 : 0)>(v);
    if (NGB > 4) v += cx.template xor_f64<xt_r2_gbit(F, NGB > 4 ? 4 : 0)>(v);
    if (NGB > 5) v += cx.template xor_f64<xt_r2_gbit(F, NGB > 5 ? 5 : 0)>(v);
    return v;
}
template <int F, class Ctx>
XT_HD int xt_r2_gmax(Ctx& cx, int v)
{
    constexpr int NGB = F - 1;
    auto mx = [](int a, int b) XT_INL { return a > b ? a : b; };
    v = mx(v, cx.template xor_i32<xt_r2_gbit(F, 0)>(v));
    v = mx(v, cx.template xor_i32<xt_r2_gbit(F, 1)>(v));
    v = mx(v, cx.template xor_i32<xt_r2_gbit(F, 2)>(v));
    if (NGB > 3) v = mx(v, cx.template xor_i32<xt_r2_gbit(F, NGB > 3 ? 3 : 0)>(v));
    if (NGB > 4) v = mx(v, cx.template xor_i32<xt_r2_gbit(F, NGB > 4 ? 4 : 0)>(v));
    if (NGB > 5) v = mx(v, cx.template xor_i32<xt_r2_gbit(F, NGB > 5 ? 5 : 0)>(v));
    return v;
}

// NP == 0: log-likelihood only, per-block sums to a.partials (ga unused).  NP > 0: ga.gpartials[block][1 + NP] = {sum LL, sum dLL/dtheta_p}.
template <int F, int D, int K, int NP, int VAR = 0, class Ctx>
XT_HD void xt_r2_body(const XtKernelArgs& a, const XtGradArgs& ga, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    typedef XtR2Geom<F> Gm;
    constexpr int NGB = Gm::NGB, TPW = Gm::TPW;
    constexpr int GMASK = xt_r2_gmask(F);
    const int lane = cx.lane();
    const int wib = cx.wave_in_block();
    const int nwb = cx.waves_per_block();
    const int L = b.L;
    const int KS = a.locerr_mode ? a.KS : 0;
    double* smem = cx.smem();
    char* lds = (char*)smem;

    const int ntab = xt_tab_doubles(2, 2);
    for (int i = cx.tid(); i < ntab; i += cx.nthreads()) smem[i] = xt_blob_ptr(a)[i];
    if (NP > 0)
        for (int i = cx.tid(); i < NP * XT_R2_TB; i += cx.nthreads()) xt_at<double>(lds, XT_R2_TAN0 + i * 8) = ga.dblob[i];
    // "uniform" directions (ga.NU of them, tangent blocks after the NP full ones): d log of every weight factor of a step is the same for
    // all sequences (e.g. the bleaching probability pBL) - then rz stays equal over the sequences, dm = du = 0, and only the last
    // position's factor table tells the sequences apart: no per-step work at all (see the read-out below)
    const int NU = NP > 0 ? ga.NU : 0, NPT = NP + NU;
    for (int i = cx.tid(); i < NU * XT_R2_TB; i += cx.nthreads()) xt_at<double>(lds, XT_R2_TAN0 + (NP * XT_R2_TB + i) * 8) = ga.udblob[i];
    cx.sync();
    xt_f2_check_lds_base(lds);
    xt_f2_build_exp_table(cx, lds, XT_F2_EXPB_OFF, (const double*)(lds + XT_F2_T64_OFF));  // the 1024-entry table of xt_exp_tab_x2 from the blob's 64 entries
    cx.sync();
    const double* hdr = smem;

    const int ts = xt_r2_slot<F>(lane);
    const bool leader = (lane & GMASK) == 0;
    const int tlast = L - 1;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int vfin = (b.isBL ? 2 : 0) + (tlast >= stay_from ? 1 : 0);
    double l2g[K];
    XT_UNROLL
    for (int k = 0; k < K; ++k) l2g[k] = hdr[k];
    const bool well_scaled = a.well_scaled != 0;

    double* pos = (double*)(lds + xt_r2_pos0(NPT)) + wib * TPW * XT_F2_CHUNK * (D + KS);  // [TPW][CHUNK][D]
    double* sig = pos + TPW * XT_F2_CHUNK * D;                                           // [TPW][CHUNK][KS]
    // per (wave, track slot): NP == 0 {mantissa, exponent, count} of the running likelihood product; NP > 0 {sum LL, sum dLL_p}
    double* accp = (double*)(lds + xt_r2_acc0(NPT, D, KS, TPW)) + (wib * 8 + ts) * (NPT + 3);
    if (leader) {
        if (NP == 0) {
            accp[0] = 1.0;
            accp[1] = 0.0;
            accp[2] = 0.0;
        } else {
            for (int i = 0; i < NPT + 1; ++i) accp[i] = 0.0;
        }
    }

    const int64_t nbatch = (b.N + TPW - 1) / TPW;
    const int64_t W0 = (int64_t)lb * nwb + wib, NW = (int64_t)nb * nwb;
    for (int64_t batch = W0; batch < nbatch; batch += NW) {
        const int64_t trk = batch * TPW + ts;
        const bool act = trk < b.N;

        auto stage = [&](int p0) XT_INL {
            cx.wave_sync();  // the reads of the previous chunk are done
            for (int i = lane; i < TPW * XT_F2_CHUNK * D; i += 64) {
                const int t_ = i / (XT_F2_CHUNK * D), r = i - t_ * (XT_F2_CHUNK * D);
                const int64_t tk = batch * TPW + t_;
                const int64_t tkc = tk < b.N ? tk : b.N - 1;
                const int pp = p0 + r / D;
                if (pp < L) {
                    const double v = b.tracks[(tkc * L + p0) * D + r];
                    pos[i] = v;
                    if (v != v) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + t_) * 4) = 1;
                }
            }
            if (KS)
                for (int i = lane; i < TPW * XT_F2_CHUNK * KS; i += 64) {
                    const int t_ = i / (XT_F2_CHUNK * KS), r = i - t_ * (XT_F2_CHUNK * KS);
                    const int64_t tk = batch * TPW + t_;
                    const int64_t tkc = tk < b.N ? tk : b.N - 1;
                    const int pp = p0 + r / KS;
                    if (pp < L) {
                        const double v = b.sigma[(tkc * L + p0) * KS + r];
                        sig[i] = v;
                        if (v != v) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + t_) * 4) = 1;
                    }
                }
            cx.wave_sync();
        };
        auto getpos = [&](int t, double* c, double* l2) XT_INL {
            const int r = t & (XT_F2_CHUNK - 1);
            XT_UNROLL
            for (int d = 0; d < D; ++d) c[d] = pos[(ts * XT_F2_CHUNK + r) * D + d];
            if (KS == 0) {
                XT_UNROLL
                for (int k = 0; k < K; ++k) l2[k] = xt_at<double>(lds, k * 8);  // broadcast read of the blob header (two fewer live VGPRs per k than a register copy)
            } else {
                XT_UNROLL
                for (int k = 0; k < K; ++k) {
                    double sg = sig[(ts * XT_F2_CHUNK + r) * KS + (KS == 1 ? 0 : k)];
                    if (a.locerr_mode == 2) {
                        sg = xt_fma(sg, hdr[3], hdr[4]);
                        sg = sg < 1e-6 ? 1e-6 : sg;
                    }
                    l2[k] = sg * sg;
                }
            }
        };

        XtR2Lane<D, K, NP> s;
#define XT_R2_PHASE(H, ZF_, LAZY_)                                                                     \
    if (NGB > (H) && t <= tend2 && ph == (H)) {                                                        \
        double c[D], l2[K];                                                                            \
        getpos(t, c, l2);                                                                              \
        xt_r2_step<F, D, K, NP, ((H) < NGB ? (H) : 0), ZF_, LAZY_, VAR>(cx, lds, s, XT_R2_TABSEL, c, l2);   \
        ++t;                                                                                           \
        ph = (H) + 1 == NGB ? 0 : (H) + 1;                                                             \
    }
#define XT_R2_PHASES(ZF_, LAZY_) \
    XT_R2_PHASE(0, ZF_, LAZY_)   \
    XT_R2_PHASE(1, ZF_, LAZY_)   \
    XT_R2_PHASE(2, ZF_, LAZY_)   \
    XT_R2_PHASE(3, ZF_, LAZY_)   \
    XT_R2_PHASE(4, ZF_, LAZY_)   \
    XT_R2_PHASE(5, ZF_, LAZY_)
#define XT_R2_TABSEL tab
        auto run_steps = [&](int& t, int tend, int tab) XT_INL {
            int ph = (t - 1) % NGB;
            if (!well_scaled) {
                const int tend2 = tend;
                while (t <= tend2) { XT_R2_PHASES(false, false) }
                return;
            }
            {   // steps t < F merge a dummy digit (zero-weight member)
                const int tend2 = tend < F - 1 ? tend : F - 1;
                while (t <= tend2) { XT_R2_PHASES(false, true) }
            }
            const int tend2 = tend;
            while (t <= tend2) { XT_R2_PHASES(true, true) }
        };
#undef XT_R2_TABSEL
#define XT_R2_TABSEL (XT_R2_TAB0 + (t >= stay_from ? 32 : 0))
        // gradient kernels: ONE variant of the unrolled loop per launch (guarded arithmetic: zero weights handled; LAZY only skips the
        // normalisation of the stored mantissas of well-scaled models) and the transition table picked per step - the three variants x
        // two tables of the likelihood-only loop would be 170 KB of code with 7 directions
        auto run_steps_g = [&](int& t, int tend) XT_INL {
            int ph = (t - 1) % NGB;
            const int tend2 = tend;
            if (well_scaled) {
                while (t <= tend2) { XT_R2_PHASES(false, true) }
            } else {
                while (t <= tend2) { XT_R2_PHASES(false, false) }
            }
        };
#undef XT_R2_TABSEL
#undef XT_R2_PHASES
#undef XT_R2_PHASE

        XtAcc tot;
        tot.clear();
        double gacc[NP ? NP : 1];  // sum over this lane's (Q, q) pairs of w * d log w, on the 2^fe scale
        double uacc[XT_R2_MAXU];   // the same for the uniform directions (d log w = the direction's constant + d log of the last factor)
        int fe = XT_EMIN;
        int t = 1;
        if (lane < 8) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + lane) * 4) = 0;
        for (int p0 = 0; p0 < L; p0 += XT_F2_CHUNK) {
            stage(p0);
            if (p0 == 0) {
                // ---- position 0: the initial state is the newest digit (group bit NGB - 1); dummy digits set -> zero weight
                constexpr int NBIT = xt_r2_gbit(F, NGB - 1);
                double c0[D], l20[K];
                getpos(0, c0, l20);
                const bool live0 = (lane & (GMASK & ~(1 << NBIT))) == 0;
                const int s0 = (lane >> NBIT) & 1;
                s.z[0] = live0 ? hdr[8 + s0] : 0.0;
                s.e[0] = live0 ? 0 : XT_EMIN;
                s.z[1] = 0.0;
                s.e[1] = XT_EMIN;
                XT_UNROLL
                for (int q = 0; q < 2; ++q) {
                    XT_UNROLL
                    for (int d = 0; d < D; ++d) s.m[q][d] = c0[d];
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) s.u[q][k] = l20[k];
                }
                XT_UNROLL
                for (int pp = 0; pp < NP; ++pp) {
                    const int tb = XT_R2_TAN0 + pp * XT_R2_TB * 8;
                    s.rz[pp][0] = live0 ? xt_at<double>(lds, tb + (8 + s0) * 8) : 0.0;
                    s.rz[pp][1] = 0.0;
                    XT_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) s.dm[pp][q][d] = 0.0;
                        XT_UNROLL
                        for (int k = 0; k < K; ++k) s.du[pp][q][k] = xt_at<double>(lds, tb + k * 8);
                    }
                }
            }
            const int tend = (L - 2 < p0 + XT_F2_CHUNK - 1) ? L - 2 : p0 + XT_F2_CHUNK - 1;
            if (NP == 0) {
                run_steps(t, tend < stay_from - 1 ? tend : stay_from - 1, XT_R2_TAB0);
                run_steps(t, tend, XT_R2_TAB0 + 32);
            } else {
                run_steps_g(t, tend);
            }
            if (tlast < p0 || tlast >= p0 + XT_F2_CHUNK) continue;
            // ---- last position (+ leaving / bleaching factor): reduction over (member Q, new digit q)
            double cl[D], l2l[K];
            getpos(tlast, cl, l2l);
            const int hp = (tlast - 2 + NGB) % NGB;  // group bit of the newest digit
            const int pbit = (F == 7 || hp < 2) ? hp : hp + 1;
            const int prev = (lane >> pbit) & 1;
            double wm[4], rr[4][K], dq[2][D], dsqQ[2];
            int we[4];
            XT_UNROLL
            for (int Q = 0; Q < 2; ++Q) {
                dsqQ[Q] = 0.0;
                XT_UNROLL
                for (int d = 0; d < D; ++d) {
                    dq[Q][d] = cl[d] - s.m[Q][d];
                    dsqQ[Q] = xt_fma(dq[Q][d], dq[Q][d], dsqQ[Q]);
                }
                double x[2], gf[2];
                XT_UNROLL
                for (int q = 0; q < 2; ++q) {
                    const double d2 = xt_at<double>(lds, XT_R2_TAB0 + 4 * 32 + (prev * 2 + q) * 8);
                    if (K == 1) {
                        const double r = xt_rcp(d2 + s.u[Q][0] + l2l[0]);
                        rr[Q * 2 + q][0] = r;
                        x[q] = -0.5 * dsqQ[Q] * r;
                        gf[q] = xt_pow_half<D>(r);
                    } else {
                        double xx = 0.0, gg = 1.0;
                        XT_UNROLL
                        for (int d = 0; d < D; ++d) {
                            const double r = xt_rcp(d2 + s.u[Q][d] + l2l[d]);
                            rr[Q * 2 + q][d] = r;
                            xx = xt_fma(-0.5 * dq[Q][d] * dq[Q][d], r, xx);
                            gg *= r;
                        }
                        x[q] = xx;
                        gf[q] = sqrt(gg);
                    }
                }
                double p[2];
                int j[2], n[2];
                xt_exp_tab_x2(x[0], x[1], p[0], p[1], j[0], j[1], n[0], n[1]);
                XT_UNROLL
                for (int q = 0; q < 2; ++q) {
                    const double tf = xt_at<double>(lds, XT_R2_TAB0 + vfin * 32 + (prev * 2 + q) * 8);
                    wm[Q * 2 + q] = s.z[Q] * tf * gf[q] * xt_at<double>(lds, XT_F2_EXPB_OFF + j[q] * 8) * p[q];
                    we[Q * 2 + q] = s.e[Q] + n[q];
                    tot.add(wm[Q * 2 + q], we[Q * 2 + q]);
                }
            }
            fe = xt_r2_gmax<F>(cx, tot.m != 0.0 ? tot.e : XT_EMIN);
            if (NP > 0) {
                double ws[4];
                XT_UNROLL
                for (int i = 0; i < 4; ++i) ws[i] = wm[i] != 0.0 ? xt_ldexp(wm[i], we[i] - fe) : 0.0;
                XT_UNROLL
                for (int pp = 0; pp < NP; ++pp) {
                    const int tb = XT_R2_TAN0 + pp * XT_R2_TB * 8;
                    double dl[K];
                    XT_UNROLL
                    for (int k = 0; k < K; ++k) dl[k] = xt_at<double>(lds, tb + k * 8);
                    double acc = 0.0;
                    XT_UNROLL
                    for (int Q = 0; Q < 2; ++Q) {
                        XT_UNROLL
                        for (int q = 0; q < 2; ++q) {
                            const double dlT = xt_at<double>(lds, tb + XT_R2_TAB0 + vfin * 32 + (prev * 2 + q) * 8);
                            const double dd2 = xt_at<double>(lds, tb + XT_R2_TAB0 + 4 * 32 + (prev * 2 + q) * 8);
                            double rel = s.rz[pp][Q] + dlT;
                            if (K == 1) {
                                const double r = rr[Q * 2 + q][0];
                                const double dden = dd2 + s.du[pp][Q][0] + dl[0];
                                double ddsq = 0.0;
                                XT_UNROLL
                                for (int d = 0; d < D; ++d) ddsq = xt_fma(-2.0 * dq[Q][d], s.dm[pp][Q][d], ddsq);
                                rel -= 0.5 * r * (D * dden + ddsq - dsqQ[Q] * r * dden);
                            } else {
                                XT_UNROLL
                                for (int d = 0; d < D; ++d) {
                                    const double r = rr[Q * 2 + q][d];
                                    const double dden = dd2 + s.du[pp][Q][d] + dl[d];
                                    rel -= 0.5 * r * (dden - 2.0 * dq[Q][d] * s.dm[pp][Q][d] - dq[Q][d] * dq[Q][d] * r * dden);
                                }
                            }
                            acc = xt_fma(ws[Q * 2 + q], ws[Q * 2 + q] != 0.0 ? rel : 0.0, acc);
                        }
                    }
                    gacc[pp] = acc;
                }
                XT_UNROLL
                for (int u = 0; u < XT_R2_MAXU; ++u) {  // uniform directions: sum over the pairs of w * d log(last factor)
                    double acc = 0.0;
                    if (u < NU) {
                        const int tb = XT_R2_TAN0 + (NP + u) * XT_R2_TB * 8;
                        XT_UNROLL
                        for (int i = 0; i < 4; ++i) acc = xt_fma(ws[i], xt_at<double>(lds, tb + XT_R2_TAB0 + vfin * 32 + (prev * 2 + (i & 1)) * 8), acc);
                    }
                    uacc[u] = acc;
                }
            }
        }
        // reduce over the track's lanes (every lane ends with the same values)
        double sum = xt_r2_gsum<F>(cx, tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0);
        const bool poisoned = xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + ts) * 4) != 0;
        if (poisoned) sum = NAN;  // NaN input -> NaN likelihood, as in the reference
        if (NP == 0) {
            if (act && leader) {
                if (b.ll_out) b.ll_out[trk] = log(sum) + (double)fe * XT_LN2 + b.ll_const;
                const double pm = accp[0] * xt_frexp_mant(sum);
                double acce = accp[1] + (double)(fe + xt_frexp_exp(sum) + xt_frexp_exp(pm));
                if (sum == 0.0) acce = -INFINITY;
                accp[0] = xt_frexp_mant(pm);
                accp[1] = acce;
                accp[2] += 1.0;
            }
        } else {
            const double rs = 1.0 / sum;
            double gsum[NP ? NP : 1];
            XT_UNROLL
            for (int pp = 0; pp < NP; ++pp) gsum[pp] = xt_r2_gsum<F>(cx, gacc[pp]);
            double usum[XT_R2_MAXU];
            XT_UNROLL
            for (int u = 0; u < XT_R2_MAXU; ++u) usum[u] = u < NU ? xt_r2_gsum<F>(cx, uacc[u]) : 0.0;
            if (act && leader) {
                const double ll = log(sum) + (double)fe * XT_LN2 + b.ll_const;
                if (b.ll_out) b.ll_out[trk] = ll;
                accp[0] += ll;
                XT_UNROLL
                for (int pp = 0; pp < NP; ++pp) accp[1 + pp] += gsum[pp] * rs;
                // uniform directions: d log of the initial fraction + one constant per step (without / with the stay-in-FOV factor)
                const int nsteps = L - 2 > 0 ? L - 2 : 0;
                const int n1 = nsteps >= stay_from ? nsteps - stay_from + 1 : 0, n0 = nsteps - n1;
                XT_UNROLL
                for (int u = 0; u < XT_R2_MAXU; ++u)
                    if (u < NU) {
                        const int tb = XT_R2_TAN0 + (NP + u) * XT_R2_TB * 8;
                        const double rzu = xt_at<double>(lds, tb + 8 * 8) + n0 * xt_at<double>(lds, tb + XT_R2_TAB0) + n1 * xt_at<double>(lds, tb + XT_R2_TAB0 + 32);
                        accp[1 + NP + u] += usum[u] * rs + rzu;
                    }
            }
        }
        cx.wave_sync();
    }

    // ---- per-slot sums -> block partials (fixed order)
    cx.sync();
    if (NP == 0) {
        if (leader) smem[wib * TPW + ts] = accp[2] > 0.0 ? log(accp[0]) + accp[1] * XT_LN2 + accp[2] * b.ll_const : 0.0;
        cx.sync();
        if (cx.tid() == 0) {
            double sacc = 0.0;
            for (int i = 0; i < nwb * TPW; ++i) sacc += smem[i];
            a.partials[cx.block()] = sacc;
        }
    } else {
        const double* acc0 = (const double*)(lds + xt_r2_acc0(NPT, D, KS, TPW));
        for (int col = cx.tid(); col < NPT + 1; col += cx.nthreads()) {
            double sacc = 0.0;
            for (int w = 0; w < nwb; ++w)
                for (int i = 0; i < TPW; ++i) sacc += acc0[(w * 8 + i) * (NPT + 3) + col];
            ga.gpartials[(int64_t)cx.block() * (NPT + 1) + col] = sacc;
        }
    }
}
