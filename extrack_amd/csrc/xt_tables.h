// Host-side table construction for the track-likelihood kernels (plain C++, no HIP).
//
// Replaces, as precomputed tables, what the reference recomputes with numpy for every chunk of
// every evaluation:
//   get_all_Bs           extrack/tracking.py:746-757  -> implicit digit arithmetic (base/off tables)
//   get_Ts_from_Bs       extrack/tracking.py:759-767  -> T tables (linear domain)
//   step variance block  extrack/tracking.py:174-180  -> d2 table
//   Lp_stay / end term   extrack/tracking.py:192,282-299 -> stay factor and Eend (isBL) tables
#pragma once
#include <algorithm>
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "xt_kernel.h"

struct XtConfig {
    int S = 0, NS = 0, F = 0, G = 0, E = 0, NG = 0, P = 0, prev_div = 1, EP = 0, skew = 0;
    int pw[16] = {0};
    std::vector<int32_t> base_tab;  // [P][NG]
    std::vector<int32_t> off_tab;   // [P][G]
};

static inline int xt_gcd(int a, int b) { return b ? xt_gcd(b, a % b) : a; }

// Returns "" on success, else a message.
static inline std::string xt_build_config(int S, int NS, int F, XtConfig& c)
{
    if (S < 2 || S > XT_MAX_STATES) return "n_states must be in [2, 8]";
    if (NS < 1 || NS > 4) return "nb_substeps must be in [1, 4]";
    if (F <= NS) return "frame_len must be at least nb_substeps + 1";
    if (F > 15) return "frame_len too large";
    double e = pow((double)S, F);
    // up to 8192 sequences the state of a track lives in LDS / registers; beyond that (5 states at frame_len 6, 2 states at frame_len 14, ...)
    // in a per-wavefront region of global memory (xt_big.h), up to 2^20 sequences per track
    if (e > 1048576.0) return "n_states^frame_len exceeds 2^20 sequences per track: lower frame_len (the window), or use the threshold-fusion kernel (fusion='threshold'), whose live-sequence count adapts to the data";
    c.S = S;
    c.NS = NS;
    c.F = F;
    c.pw[0] = 1;
    for (int i = 1; i < 16; ++i) c.pw[i] = (i <= F) ? c.pw[i - 1] * S : 0;
    c.G = c.pw[NS];
    c.E = c.pw[F];
    c.NG = c.E / c.G;
    c.skew = (S & (S - 1)) == 0 ? 1 : 0;  // power-of-two S: skewed storage (xt_kernel.h)
    c.EP = xt_padded_entries(c.E, c.skew);
    c.P = F / xt_gcd(F, NS);
    c.prev_div = c.pw[F - NS - 1];
    c.base_tab.assign((size_t)c.P * c.NG, 0);
    c.off_tab.assign((size_t)c.P * c.G, 0);
    for (int ph = 0; ph < c.P; ++ph) {
        const int h = (1 + ph * NS) % F;  // first slot overwritten at step t = ph + 1 (+ k*P)
        for (int q = 0; q < c.G; ++q) {
            int off = 0, r = q;
            for (int j = 0; j < NS; ++j) {
                off += (r % S) * c.pw[(h + j) % F];
                r /= S;
            }
            c.off_tab[(size_t)ph * c.G + q] = off;
        }
        for (int g = 0; g < c.NG; ++g) {
            int base = 0, r = g;
            for (int i = 0; i < F - NS; ++i) {
                base += (r % S) * c.pw[(h + NS + i) % F];
                r /= S;
            }
            c.base_tab[(size_t)ph * c.NG + g] = base;
        }
    }
    return "";
}

struct XtModelHost {
    int S, NS;
    int locerr_dims;       // K for the global mode
    double locerr[3];      // std per dim (global mode)
    double slope, offset;  // affine per-peak mode
    double pBL;
    const double* ds;      // [S]
    const double* Fs;      // [S]
    const double* TrMat;   // [S*S] row-major P(i->j) per substep
    const double* p_stay;  // [S^NS], index r: digit c = (r / S^c) % S, c = 0 newest
};

// blob: XT_BLOB_HDR + XT_NTAB*S*G doubles (layout in xt_kernel.h)
static inline void xt_build_blob(const XtModelHost& m, const XtConfig& c, std::vector<double>& blob)
{
    const int S = c.S, NS = c.NS, G = c.G;
    blob.assign((size_t)xt_tab_doubles(S, G), 0.0);
    for (int j = 0; j < 64; ++j) blob[(size_t)XT_BLOB_HDR + (size_t)XT_NTAB * S * G + j] = exp2((double)j / 64.0);
    for (int k = 0; k < 3; ++k) {
        const double s = m.locerr[k < m.locerr_dims ? k : 0];
        blob[k] = s * s;
    }
    blob[3] = m.slope;
    blob[4] = m.offset;
    for (int s = 0; s < S; ++s) blob[8 + s] = m.Fs[s];
    // Eend[a] = sum over ns further transitions from a of prod(T) * qq[last]; qq uses p_stay indexed by the RAW
    // newest state value (reference quirk, tracking.py:297)
    std::vector<double> qq(S), Eend(S, 0.0);
    for (int s = 0; s < S; ++s) {
        const double ps = m.p_stay[s];
        qq[s] = m.pBL + (1.0 - ps) - m.pBL * (1.0 - ps);
    }
    {
        std::vector<double> v = qq, w(S);
        for (int it = 0; it < NS; ++it) {  // v <- T v
            for (int i = 0; i < S; ++i) {
                double acc = 0.0;
                for (int j = 0; j < S; ++j) acc += m.TrMat[i * S + j] * v[j];
                w[i] = acc;
            }
            v = w;
        }
        Eend = v;
    }
    double* TAB = blob.data() + XT_BLOB_HDR;
    const size_t SG = (size_t)S * G;
    for (int prev = 0; prev < S; ++prev) {
        for (int q = 0; q < G; ++q) {
            int chain[8];
            chain[0] = prev;
            int r = q;
            for (int j = 1; j <= NS; ++j) {
                chain[j] = r % S;
                r /= S;
            }
            double tp = 1.0, d2 = 0.0;
            for (int j = 0; j < NS; ++j) {
                tp *= m.TrMat[chain[j] * S + chain[j + 1]];
                d2 += (m.ds[chain[j]] * m.ds[chain[j]] + m.ds[chain[j + 1]] * m.ds[chain[j + 1]]) / 2.0;
            }
            d2 /= NS;
            int rref = 0;  // reference index of the new digits: digit c = a_{NS-c}
            for (int cc = 0; cc < NS; ++cc) rref += chain[NS - cc] * c.pw[cc];
            const double stay = m.p_stay[rref] * (1.0 - m.pBL);
            const double ee = Eend[chain[NS]];
            const size_t o = (size_t)prev * G + q;
            TAB[0 * SG + o] = tp;
            TAB[1 * SG + o] = tp * stay;
            TAB[2 * SG + o] = tp * ee;
            TAB[3 * SG + o] = tp * stay * ee;
            TAB[4 * SG + o] = d2;
        }
    }
    // "Well scaled" model: every initial fraction and transition (x stay) weight in [1e-20, 1], localisation variance + displacement
    // variance in [1e-12, 1e4].  Then (a) no sequence weight
    // is ever exactly zero once the window is populated and (b) a mantissa left un-normalised for XT_F2_RENORM steps stays within
    // [1e-80, 1e55], far from where the step's products (W^3 den^2) would leave the fp64 range: the 2-state fast path then drops its
    // zero handling and re-normalises lazily; any other model takes the fully guarded steps.
    // The launcher (xt_model_well_scaled) combines slot 5 (fractions / weights in range) and the displacement-variance range in slots
    // 6, 7 with the range of the localisation variance - global, or from the per-peak errors seen at upload - into the kernel argument.
    bool ok = true;
    for (int s = 0; s < S; ++s) ok = ok && m.Fs[s] >= 1e-20 && m.Fs[s] <= 1.0;
    for (size_t i = 0; i < 2 * SG; ++i) ok = ok && TAB[i] >= 1e-20 && TAB[i] <= 1.0;
    double d2min = 1e300, d2max = 0.0;
    for (size_t i = 0; i < SG; ++i) {
        d2min = std::min(d2min, TAB[4 * SG + i]);
        d2max = std::max(d2max, TAB[4 * SG + i]);
    }
    blob[5] = ok ? 1.0 : 0.0;
    blob[6] = d2min;
    blob[7] = d2max;
}

// Well-scaled test of a launch (see xt_build_blob): l2lo / l2hi = range of the localisation variance over everything the launch reads.
static inline bool xt_model_well_scaled(const std::vector<double>& blob, double l2lo, double l2hi)
{
    return blob[5] != 0.0 && l2lo == l2lo && l2hi == l2hi && l2lo + blob[6] >= 1e-12 && 2.0 * l2hi + blob[7] <= 1e4;
}

// Model tables of the threshold-fusion kernels (xt_th.h): same five [prev][.] tables and header as xt_build_blob, but the
// second index is r in the REFERENCE's digit order (digit c of r = c-th newest sub-state, tracking.py:548-551), which is
// how the expanded sequence index j = parent * S^ns + r of P_Cs_inter_bound_stats_th is laid out.
static inline std::string xt_th_build_blob(const XtModelHost& m, std::vector<double>& blob, int& G_out)
{
    const int S = m.S, NS = m.NS;
    if (S < 2 || S > XT_MAX_STATES) return "n_states must be in [2, 8]";
    if (NS < 1 || NS > 4) return "nb_substeps must be in [1, 4]";
    int G = 1;
    for (int i = 0; i < NS; ++i) G *= S;
    G_out = G;
    blob.assign((size_t)xt_tab_doubles(S, G), 0.0);
    for (int j = 0; j < 64; ++j) blob[(size_t)XT_BLOB_HDR + (size_t)XT_NTAB * S * G + j] = exp2((double)j / 64.0);
    for (int k = 0; k < 3; ++k) {
        const double s = m.locerr[k < m.locerr_dims ? k : 0];
        blob[k] = s * s;
    }
    blob[3] = m.slope;
    blob[4] = m.offset;
    for (int s = 0; s < S; ++s) blob[8 + s] = m.Fs[s];
    std::vector<double> v(S), w(S);
    for (int s = 0; s < S; ++s) {
        const double ps = m.p_stay[s];  // raw newest state as index (reference quirk, tracking.py:624)
        v[s] = m.pBL + (1.0 - ps) - m.pBL * (1.0 - ps);
    }
    for (int it = 0; it < NS; ++it) {
        for (int i = 0; i < S; ++i) {
            double acc = 0.0;
            for (int j = 0; j < S; ++j) acc += m.TrMat[i * S + j] * v[j];
            w[i] = acc;
        }
        v = w;
    }
    double* TAB = blob.data() + XT_BLOB_HDR;
    const size_t SG = (size_t)S * G;
    for (int prev = 0; prev < S; ++prev)
        for (int r = 0; r < G; ++r) {
            int dig[8], rr = r;
            for (int c = 0; c < NS; ++c) {
                dig[c] = rr % S;  // c-th newest
                rr /= S;
            }
            dig[NS] = prev;
            double tp = 1.0, d2 = 0.0;
            for (int c = 0; c < NS; ++c) {
                tp *= m.TrMat[dig[c + 1] * S + dig[c]];
                d2 += (m.ds[dig[c]] * m.ds[dig[c]] + m.ds[dig[c + 1]] * m.ds[dig[c + 1]]) / 2.0;
            }
            d2 /= NS;
            const double stay = m.p_stay[r] * (1.0 - m.pBL);
            const double ee = v[dig[0]];
            const size_t o = (size_t)prev * G + r;
            TAB[0 * SG + o] = tp;
            TAB[1 * SG + o] = tp * stay;
            TAB[2 * SG + o] = tp * ee;
            TAB[3 * SG + o] = tp * stay * ee;
            TAB[4 * SG + o] = d2;
        }
    return "";
}

static inline void xt_fill_args_from_config(const XtConfig& c, XtKernelArgs& a)
{
    a.S = c.S;
    a.NS = c.NS;
    a.F = c.F;
    a.G = c.G;
    a.E = c.E;
    a.NG = c.NG;
    a.EP = c.EP;
    a.skew = c.skew;
    a.P = c.P;
    a.prev_div = c.prev_div;
    for (int i = 0; i < 16; ++i) a.pw[i] = c.pw[i];
}

static inline size_t xt_lds_bytes(const XtConfig& c, int D, int K, int tpb)
{
    size_t d = (size_t)((xt_tab_doubles(c.S, c.G) + 1) & ~1) + (size_t)tpb * xt_region_doubles(c.EP, D, K);
    d += (size_t)tpb * (xt_pred_doubles(c.S, c.F) + xt_stage_doubles(D));  // posterior accumulators + staged positions
    return d * sizeof(double);
}

// Launch geometry: one thread per group, as many tracks per block as fit 256 threads and a 64 KiB LDS
// budget (so that several blocks share a CU); a single track may take up to the whole 160 KiB.
static inline void xt_geometry(const XtConfig& c, int D, int K, int& tpb, int& threads)
{
    const size_t per_track = xt_lds_bytes(c, D, K, 1) - xt_lds_bytes(c, D, K, 0);
    const size_t fixed = xt_lds_bytes(c, D, K, 0);
    const size_t budget = 64 * 1024;
    int by_threads = c.NG >= 256 ? 1 : 256 / c.NG;
    int by_lds = budget > fixed + per_track ? (int)((budget - fixed) / per_track) : 1;
    tpb = by_threads < by_lds ? by_threads : by_lds;
    if (tpb < 1) tpb = 1;
    threads = (tpb * c.NG + 63) / 64 * 64;
}
