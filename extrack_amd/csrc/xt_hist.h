// State-duration histograms (extrack/histograms.py:26-286 P_segment_len, driven by len_hist :294-373) on the GPU.
//
// What the reference does per track: enumerate state sequences WITH their full history, one more state per position; after a
// position, when more than max_nb_states sequences are alive, keep the max_nb_states most probable ones (ranked including the
// predictive density of the next position); at the end weight every surviving sequence by its probability and histogram the
// lengths of its runs of equal states.  Unlike the likelihood kernels nothing is merged here: the sequences keep their identity.
//
// Mapping: one workgroup per track at a time.  The <= K surviving sequences ("parents") live as struct-of-arrays
// {LP, LL, m[D], s2[K], history bits} in LDS (or, for very large K, in a global workspace); a position is processed as
//   A  per parent:     Gaussian integration of the position (the part that does not depend on the new state): LC, m', l2 s2/den
//   B  per candidate:  (parent, new state) -> LP', LL' and the ranking key (log-probability incl. the next position)
//   C  bitonic sort of (key, candidate) pairs, descending                                   [only when more than K candidates]
//   D  gather the K best candidates into the other parent buffer; the history gets the new state appended.
// The reference's quirk of carrying over the LL terms of the LAST K entries of the ranking (histograms.py:201) is reproduced in D.
// Ranking: by the log-probability itself, equal values by candidate index.  The reference ranks exp(log-probability) with
// np.argsort (unstable): identical as long as no two candidates tie - exact ties (models with symmetric rates produce them) and
// values that underflow in the reference's exp() (below about -745) are ordered by numpy's sort implementation there, which is
// not reproducible across machines either; through the LL quirk the order of tied entries changes the result at the 1e-4 level.
// The last position is streamed: the candidates' final weights go through a block-wide max / sum, every candidate then adds its
// normalised weight to the histogram bins of its runs (LDS fp64 atomics).  Log domain throughout, like the reference.
#pragma once
#include "xt_math.h"

#define XT_HIST_MAXW 64  // history words (64 bits each) per sequence, a run-time count: len * bits_per_state <= 4096 (2 states: 4096 positions, 3 - 4: 2048, 5 - 8: 1365)
#define XT_HIST_EPT 8   // register path of the ranking sort: at most this many candidates per thread (2048 candidates at 256 threads)

struct XtHistArgs {
    const double* tracks;  // [N][L][D]
    const double* sigma;   // [N][L][KS] or nullptr
    const double* blob;    // model tables, layout below
    double* partials;      // [nblocks][(L - 1) * S] per-block histograms
    double* ws;            // parent buffers in global memory (par_lds == 0): ws_stride doubles per block
    int64_t ws_stride;
    int64_t N;
    int32_t L, S, KS, locerr_mode, isBL, min_l;
    int32_t K;     // max_nb_states
    int32_t PC;    // parent capacity: max(K, S * S)
    int32_t NC;    // candidate capacity: power of two >= PC * S
    int32_t HW;    // history words per sequence
    int32_t bits;  // bits per state
    int32_t par_lds;
};
// blob (doubles): [0..2] l2, [3] slope, [4] offset, [8 + s] log F[s], [16 + s] Lp_stay[s], [24 + s] log Q[s] (end-of-track factor),
//                 [32 + a * S + b] log T[a][b], [32 + S*S + a * S + b] d2[a][b] = (ds[a]^2 + ds[b]^2) / 2
XT_HD int xt_hist_blob_doubles(int S) { return 32 + 2 * S * S; }
XT_HD int xt_hist_parent_doubles(int PC, int D, int K, int HW) { return PC * (2 + D + K + HW); }
XT_HD int xt_hist_tmp_doubles(int PC, int D, int K) { return PC * (1 + D + K); }
// LDS doubles: blob + staged track + candidate arrays (key: NC doubles; idx: NC ints - a candidate's LP / LL are recomputed from its parent
// when it is gathered, which keeps the footprint of the default max_nb_states = 500 below 80 KiB: two workgroups per CU) + per-parent
// scratch + histogram + reduction scratch (+ the two parent buffers when they fit)
XT_HD size_t xt_hist_lds_doubles(int S, int L, int D, int K, int KS, int PC, int NC, int HW, int nthreads, bool par_lds)
{
    size_t d = (size_t)((xt_hist_blob_doubles(S) + 1) & ~1) + (size_t)L * (D + KS) + (size_t)NC + (size_t)NC / 2 + xt_hist_tmp_doubles(PC, D, K) +
               (size_t)(L - 1) * S + nthreads + 8;
    if (par_lds) d += 2 * (size_t)xt_hist_parent_doubles(PC, D, K, HW);
    return d;
}

// Ranking sort of the candidates, descending by key, equal keys by candidate index ascending (a total order: the result does not depend on
// the sorting network): bitonic network over NS2 = EPT * nthreads (key, index) pairs with the elements in REGISTERS, thread t owning the
// positions t * EPT .. t * EPT + EPT - 1.  Stages with a partner distance below EPT compare inside a thread, those below 64 * EPT trade the
// elements between the lanes of a wavefront (DPP / permlane moves: no LDS round trip, no barrier), only the few stages with a larger distance go
// through the LDS arrays.  EPT is a compile-time constant and every comparison is branch-free: the first version (run-time EPT, `||`
// comparisons) compiled to one exec-mask branch per element and ran no faster than the all-LDS form (650 cycles per stage, r02).
template <int EPT, class Ctx>
XT_HD void xt_hist_sort_regs(Ctx& cx, double* key, int* idx, int NS2)
{
    const int tid = cx.tid(), nt = cx.nthreads();
    const int NP2 = NS2 >> 1;
    constexpr int LOG_EPT = EPT == 1 ? 0 : (EPT == 2 ? 1 : (EPT == 4 ? 2 : 3));
    double kk[EPT];
    int ii[EPT];
    XT_UNROLL
    for (int b = 0; b < EPT; ++b) {
        kk[b] = key[tid * EPT + b];
        ii[b] = idx[tid * EPT + b];
    }
    for (int k2 = 2; k2 <= NS2; k2 <<= 1) {
        int j2 = k2 >> 1;
        if (j2 >= 64 * EPT) {  // partner in another wavefront: through LDS
            XT_UNROLL
            for (int b = 0; b < EPT; ++b) {
                key[tid * EPT + b] = kk[b];
                idx[tid * EPT + b] = ii[b];
            }
            cx.sync();
            for (; j2 >= 64 * EPT; j2 >>= 1) {
                for (int pp = tid; pp < NP2; pp += nt) {
                    const int t = ((pp & ~(j2 - 1)) << 1) | (pp & (j2 - 1)), u = t + j2;
                    const bool desc = (t & k2) == 0;
                    const double ka = key[t], kb = key[u];
                    const int ia = idx[t], ib = idx[u];
                    const bool a_first = (ka > kb) | ((ka == kb) & (ia < ib));
                    if (desc != a_first) {
                        key[t] = kb;
                        key[u] = ka;
                        idx[t] = ib;
                        idx[u] = ia;
                    }
                }
                cx.sync();
            }
            XT_UNROLL
            for (int b = 0; b < EPT; ++b) {
                kk[b] = key[tid * EPT + b];
                ii[b] = idx[tid * EPT + b];
            }
        }
        // partner in another lane of the wavefront (lane ^ 2^MB): DPP / permlane moves with compile-time lane masks (Ctx::xor_*)
#define XT_HIST_LANE_STAGE(MB)                                                                        \
    if ((EPT << (MB)) <= (k2 >> 1)) {                                                                 \
        constexpr int j2s = EPT << (MB);                                                              \
        double ok[EPT];                                                                               \
        int oi[EPT];                                                                                  \
        XT_UNROLL                                                                                     \
        for (int b = 0; b < EPT; ++b) {                                                               \
            ok[b] = cx.template xor_f64<(MB)>(kk[b]);                                                 \
            oi[b] = cx.template xor_i32<(MB)>(ii[b]);                                                 \
        }                                                                                             \
        XT_UNROLL                                                                                     \
        for (int b = 0; b < EPT; ++b) {                                                               \
            const int pidx = tid * EPT + b;                                                           \
            const bool self_first = (kk[b] > ok[b]) | ((kk[b] == ok[b]) & (ii[b] < oi[b]));           \
            const bool want_first = ((pidx & j2s) == 0) == ((pidx & k2) == 0);                        \
            const bool take = self_first != want_first;                                               \
            kk[b] = take ? ok[b] : kk[b];                                                             \
            ii[b] = take ? oi[b] : ii[b];                                                             \
        }                                                                                             \
    }
        XT_HIST_LANE_STAGE(5)
        XT_HIST_LANE_STAGE(4)
        XT_HIST_LANE_STAGE(3)
        XT_HIST_LANE_STAGE(2)
        XT_HIST_LANE_STAGE(1)
        XT_HIST_LANE_STAGE(0)
#undef XT_HIST_LANE_STAGE
        XT_UNROLL
        for (int jj = EPT / 2; jj > 0; jj >>= 1) {  // both elements in this thread (compile-time register indices)
            if (jj > (k2 >> 1)) continue;
            XT_UNROLL
            for (int b = 0; b < EPT; ++b)
                if ((b & jj) == 0) {
                    const int u = b | jj;
                    const bool desc = ((tid * EPT + b) & k2) == 0;
                    const bool a_first = (kk[b] > kk[u]) | ((kk[b] == kk[u]) & (ii[b] < ii[u]));
                    const bool swap = desc != a_first;
                    const double tk = swap ? kk[u] : kk[b];
                    kk[u] = swap ? kk[b] : kk[u];
                    kk[b] = tk;
                    const int ti = swap ? ii[u] : ii[b];
                    ii[u] = swap ? ii[b] : ii[u];
                    ii[b] = ti;
                }
        }
    }
    XT_UNROLL
    for (int b = 0; b < EPT; ++b) {
        key[tid * EPT + b] = kk[b];
        idx[tid * EPT + b] = ii[b];
    }
}

template <int D, int K, class Ctx>
XT_HD void xt_hist_body(const XtHistArgs& a, Ctx& cx)
{
    const int tid = cx.tid(), nt = cx.nthreads();
    const int S = a.S, L = a.L, PC = a.PC, NC = a.NC, HW = a.HW, bits = a.bits, KS = a.locerr_mode ? a.KS : 0;
    const uint64_t smask = (1ull << bits) - 1ull;
    double* smem = cx.smem();
    const int nblob = xt_hist_blob_doubles(S);
    for (int i = tid; i < nblob; i += nt) smem[i] = a.blob[i];
    const double* hdr = smem;
    const double* logF = smem + 8;
    const double* Lpst = smem + 16;
    const double* logQ = smem + 24;
    const double* logT = smem + 32;
    const double* d2t = smem + 32 + S * S;
    double* w = smem + ((nblob + 1) & ~1);
    double* spos = w;
    w += (size_t)L * D;
    double* ssig = w;
    w += (size_t)L * KS;
    double* key = w;
    w += NC;
    int* idx = (int*)w;
    w += NC / 2;
    double* tLC = w;               // per-parent scratch: LC, m'[D], l2 s2 / den [K]
    double* tM = tLC + PC;
    double* tS = tM + (size_t)D * PC;
    w += xt_hist_tmp_doubles(PC, D, K);
    double* hacc = w;
    w += (size_t)(L - 1) * S;
    double* red = w;
    w += nt;
    int* flag = (int*)w;
    w += 8;
    const int pdoubles = xt_hist_parent_doubles(PC, D, K, HW);
    double* parA = a.par_lds ? w : a.ws + (size_t)cx.block() * a.ws_stride;
    double* parB = parA + pdoubles;
    const int nbins = (L - 1) * S;
    for (int i = tid; i < nbins; i += nt) hacc[i] = 0.0;
    cx.sync();

    auto l2_at = [&](int pos, double* l2) {
        if (a.locerr_mode == 0) {
            for (int k = 0; k < K; ++k) l2[k] = hdr[k];
        } else {
            for (int k = 0; k < K; ++k) {
                double s = ssig[pos * KS + (KS == 1 ? 0 : k)];
                if (a.locerr_mode == 2) {
                    s = xt_fma(s, hdr[3], hdr[4]);
                    s = s < 1e-6 ? 1e-6 : s;
                }
                l2[k] = s * s;
            }
        }
    };
    // log N(c; m, v) summed over the dims (a K == 1 variance serves all dims)
    auto gauss = [&](const double* c, const double* m, const double* v) {
        double q = 0.0, lg = 0.0;
        if (K == 1) {
            for (int d = 0; d < D; ++d) q = xt_fma(c[d] - m[d], c[d] - m[d], q);
            return -0.5 * D * log(2.0 * M_PI * v[0]) - q / (2.0 * v[0]);
        }
        for (int d = 0; d < D; ++d) {
            lg += log(2.0 * M_PI * v[d]);
            q += (c[d] - m[d]) * (c[d] - m[d]) / (2.0 * v[d]);
        }
        return -0.5 * lg - q;
    };
    auto block_max = [&](double v) {
        red[tid] = v;
        cx.sync();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) red[tid] = red[tid] > red[tid + s] ? red[tid] : red[tid + s];
            cx.sync();
        }
        const double o = red[0];
        cx.sync();
        return o;
    };
    auto block_sum = [&](double v) {
        red[tid] = v;
        cx.sync();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            cx.sync();
        }
        const double o = red[0];
        cx.sync();
        return o;
    };

    for (int64_t trk = cx.block(); trk < a.N; trk += cx.nblocks()) {
        // ---- the whole track -> LDS
        if (tid == 0) flag[0] = 0;
        cx.sync();
        for (int i = tid; i < L * D; i += nt) {
            const double v = a.tracks[trk * L * D + i];
            spos[i] = v;
            if (v != v) flag[0] = 1;
        }
        for (int i = tid; i < L * KS; i += nt) {
            const double v = a.sigma[trk * L * KS + i];
            ssig[i] = v;
            if (v != v) flag[0] = 1;
        }
        cx.sync();
        if (flag[0]) {  // NaN input: the track's histogram contribution is NaN, as in the reference
            for (int i = tid; i < nbins; i += nt) hacc[i] = NAN;
            cx.sync();
            continue;
        }
        double* cur = parA;
        double* nxt = parB;
        auto P_LP = [&](double* p) { return p; };
        auto P_LL = [&](double* p) { return p + PC; };
        auto P_M = [&](double* p) { return p + 2 * (size_t)PC; };
        auto P_S = [&](double* p) { return p + (2 + D) * (size_t)PC; };
        auto P_H = [&](double* p) { return (uint64_t*)(p + (2 + D + K) * (size_t)PC); };

        // ---- first position (histograms.py:93-138): S^2 sequences of two states, index i = old * S + new
        int n = S * S;
        {
            double l20[K];
            l2_at(0, l20);
            for (int i = tid; i < n; i += nt) {
                const int nw = i % S, od = i / S;
                P_LP(cur)[i] = logT[od * S + nw] + logF[od];
                P_LL(cur)[i] = 1 >= a.min_l ? Lpst[nw] : 0.0;
                for (int d = 0; d < D; ++d) P_M(cur)[(size_t)d * PC + i] = spos[d];
                for (int k = 0; k < K; ++k) P_S(cur)[(size_t)k * PC + i] = l20[k] + d2t[od * S + nw];
                uint64_t* h = P_H(cur);
                h[i] = ((uint64_t)od << bits) | (uint64_t)nw;
                for (int wd = 1; wd < HW; ++wd) h[(size_t)wd * PC + i] = 0ull;
            }
        }
        cx.sync();

        // ---- positions 1 .. L-2 with pruning (the step that injects position L-2 is the streamed final step below)
        for (int c = 2; c <= L - 2; ++c) {
            const int p = c - 1;
            double lp[K], ln[K], cp[D], cn[D];
            l2_at(p, lp);
            l2_at(p + 1, ln);
            for (int d = 0; d < D; ++d) {
                cp[d] = spos[p * D + d];
                cn[d] = spos[(p + 1) * D + d];
            }
            // A: per parent
            for (int i = tid; i < n; i += nt) {
                double m[D], s2[K], den[K];
                for (int d = 0; d < D; ++d) m[d] = P_M(cur)[(size_t)d * PC + i];
                for (int k = 0; k < K; ++k) {
                    s2[k] = P_S(cur)[(size_t)k * PC + i];
                    den[k] = lp[k] + s2[k];
                }
                tLC[i] = gauss(cp, m, den);
                for (int d = 0; d < D; ++d) {
                    const int k = K == 1 ? 0 : d;
                    tM[(size_t)d * PC + i] = (m[d] * lp[k] + cp[d] * s2[k]) / den[k];
                }
                for (int k = 0; k < K; ++k) tS[(size_t)k * PC + i] = lp[k] * s2[k] / den[k];
            }
            cx.sync();
            // B: per candidate j = i * S + r
            const int nc = n * S;
            const bool prune = nc > a.K;
            if (prune)
                for (int j = tid; j < nc; j += nt) {
                    const int i = j / S, r = j - i * S;
                    const int prev = (int)(P_H(cur)[i] & smask);
                    double m[D], v[K];
                    for (int d = 0; d < D; ++d) m[d] = tM[(size_t)d * PC + i];
                    for (int k = 0; k < K; ++k) v[k] = d2t[prev * S + r] + tS[(size_t)k * PC + i] + ln[k];
                    key[j] = P_LP(cur)[i] + logT[prev * S + r] + tLC[i] + gauss(cn, m, v);
                    idx[j] = j;
                }
            if (prune)
                for (int j = nc + tid; j < NC; j += nt) {
                    key[j] = -INFINITY;
                    idx[j] = j;
                }
            cx.sync();
            // C: bitonic sort, descending by key (power-of-two size covering the candidates)
            int nnew = nc;
            if (prune) {
                int NS2 = 1;
                while (NS2 < nc) NS2 <<= 1;
                // One compare-exchange per thread and pair (pair p of a stage with partner distance j2: elements t and t + j2, t = p with a
                // zero bit inserted at j2).  Stages with j2 <= 64 stay inside a 128-element block that ONE wavefront owns (64 consecutive
                // pairs), so they need no workgroup barrier - LDS operations of a wavefront execute in order; only the few stages with a
                // larger distance (and the hand-over to them) synchronise the workgroup: 9 barriers instead of 55 for 1024 candidates.
                const int NP2 = NS2 >> 1;
                auto cmpx = [&](int pp, int j2, int k2) {
                    const int t = ((pp & ~(j2 - 1)) << 1) | (pp & (j2 - 1)), u = t + j2;
                    const bool desc = (t & k2) == 0;
                    const double ka = key[t], kb = key[u];
                    const int ia = idx[t], ib = idx[u];
                    // order: key descending, equal keys by candidate index ascending (a total order: the result does not depend on
                    // the sorting network)
                    const bool a_first = ka > kb || (ka == kb && ia < ib);
                    if (desc ? !a_first : a_first) {
                        key[t] = kb;
                        key[u] = ka;
                        idx[t] = ib;
                        idx[u] = ia;
                    }
                };
                const int ept = NS2 / nt;  // elements per thread of the register path (thread t owns positions t * ept .. t * ept + ept - 1)
                if (ept >= 1 && ept <= XT_HIST_EPT && (nt & 63) == 0) {
                    if (ept == 1) xt_hist_sort_regs<1>(cx, key, idx, NS2);
                    else if (ept == 2) xt_hist_sort_regs<2>(cx, key, idx, NS2);
                    else if (ept == 4) xt_hist_sort_regs<4>(cx, key, idx, NS2);
                    else xt_hist_sort_regs<8>(cx, key, idx, NS2);
                } else {
                    for (int k2 = 2; k2 <= NS2; k2 <<= 1) {
                        int j2 = k2 >> 1;
                        for (; j2 > 64; j2 >>= 1) {
                            for (int pp = tid; pp < NP2; pp += nt) cmpx(pp, j2, k2);
                            cx.sync();
                        }
                        for (int p0 = 0; p0 < NP2; p0 += nt)
                            for (int jj = j2; jj > 0; jj >>= 1) {
                                if (p0 + tid < NP2) cmpx(p0 + tid, jj, k2);
                                cx.wave_sync();
                            }
                        if (k2 >= 128) cx.sync();
                    }
                }
                cx.sync();
                nnew = a.K;
            }
            // D: gather into the other buffer
            for (int t = tid; t < nnew; t += nt) {
                const int j = prune ? idx[t] : t;
                const int jl = prune ? idx[nc - a.K + t] : t;  // reference quirk: LL of the LAST K entries of the ranking
                const int i = j / S, r = j - i * S;
                const int prev = (int)(P_H(cur)[i] & smask);
                const int il = jl / S, rl = jl - il * S;
                P_LP(nxt)[t] = P_LP(cur)[i] + logT[prev * S + r] + tLC[i];
                P_LL(nxt)[t] = P_LL(cur)[il] + (c >= a.min_l ? Lpst[rl] : 0.0);
                for (int d = 0; d < D; ++d) P_M(nxt)[(size_t)d * PC + t] = tM[(size_t)d * PC + i];
                for (int k = 0; k < K; ++k) P_S(nxt)[(size_t)k * PC + t] = d2t[prev * S + r] + tS[(size_t)k * PC + i];
                uint64_t carry = (uint64_t)r;
                for (int wd = 0; wd < HW; ++wd) {
                    const uint64_t hv = P_H(cur)[(size_t)wd * PC + i];
                    P_H(nxt)[(size_t)wd * PC + t] = (hv << bits) | carry;
                    carry = hv >> (64 - bits);
                }
            }
            cx.sync();
            double* tb = cur;
            cur = nxt;
            nxt = tb;
            n = nnew;
        }

        // ---- final step, streamed: (L > 2) expansion by the state at position L-2 + its integration, then the last position
        const bool expand = L > 2;
        const int nc = expand ? n * S : n;
        double ll[K], lq[K], cq[D], cl[D];
        l2_at(L - 1, ll);
        for (int d = 0; d < D; ++d) cl[d] = spos[(L - 1) * D + d];
        if (expand) {
            l2_at(L - 2, lq);
            for (int d = 0; d < D; ++d) cq[d] = spos[(L - 2) * D + d];
            for (int i = tid; i < n; i += nt) {
                double m[D], s2[K], den[K];
                for (int d = 0; d < D; ++d) m[d] = P_M(cur)[(size_t)d * PC + i];
                for (int k = 0; k < K; ++k) {
                    s2[k] = P_S(cur)[(size_t)k * PC + i];
                    den[k] = lq[k] + s2[k];
                }
                tLC[i] = gauss(cq, m, den);
                for (int d = 0; d < D; ++d) {
                    const int k = K == 1 ? 0 : d;
                    tM[(size_t)d * PC + i] = (m[d] * lq[k] + cq[d] * s2[k]) / den[k];
                }
                for (int k = 0; k < K; ++k) tS[(size_t)k * PC + i] = lq[k] * s2[k] / den[k];
            }
            cx.sync();
        }
        double mymax = -INFINITY;
        for (int j = tid; j < nc; j += nt) {
            const int i = expand ? j / S : j, r = expand ? j - i * S : 0;
            const uint64_t h0 = P_H(cur)[i];
            const int prev = (int)(h0 & smask);
            double W, m[D], v[K];
            if (expand) {
                for (int d = 0; d < D; ++d) m[d] = tM[(size_t)d * PC + i];
                for (int k = 0; k < K; ++k) v[k] = d2t[prev * S + r] + tS[(size_t)k * PC + i] + ll[k];
                W = P_LP(cur)[i] + logT[prev * S + r] + tLC[i] + P_LL(cur)[i] + (L - 1 >= a.min_l ? Lpst[r] : 0.0);
            } else {
                for (int d = 0; d < D; ++d) m[d] = P_M(cur)[(size_t)d * PC + i];
                for (int k = 0; k < K; ++k) v[k] = P_S(cur)[(size_t)k * PC + i] + ll[k];
                W = P_LP(cur)[i] + P_LL(cur)[i];
            }
            const int newest = expand ? r : prev;
            W += gauss(cl, m, v) + (a.isBL ? logQ[newest] : 0.0);
            key[j] = W;
            mymax = W > mymax ? W : mymax;
        }
        const double wmax = block_max(mymax);
        double mysum = 0.0;
        for (int j = tid; j < nc; j += nt) {
            const double e = exp(key[j] - wmax);
            key[j] = e;
            mysum += e;
        }
        const double wsum = block_sum(mysum);
        const double rsum = 1.0 / wsum;
        // runs of equal states along the full history (newest first); only runs shorter than the track are counted
        for (int j = tid; j < nc; j += nt) {
            const double Pn = key[j] * rsum;
            if (!(Pn > 0.0)) continue;
            const int i = expand ? j / S : j, r = expand ? j - i * S : 0;
            int state = -1, run = 0, pos = 0;  // pos: digits consumed
            if (expand) {
                state = r;
                run = 1;
                pos = 1;
            }
            const int ndig = expand ? L - 1 : L;  // digits stored in the parent's history
            for (int q = 0; q < ndig; ++q) {
                const int bitpos = q * bits;
                const int wd = bitpos >> 6, sh = bitpos & 63;
                uint64_t hv = P_H(cur)[(size_t)wd * PC + i] >> sh;
                if (sh + bits > 64) hv |= P_H(cur)[(size_t)(wd + 1) * PC + i] << (64 - sh);
                const int dg = (int)(hv & smask);
                if (dg == state) {
                    ++run;
                } else {
                    if (state >= 0) cx.atomic_add_f64(&hacc[(run - 1) * S + state], Pn);
                    state = dg;
                    run = 1;
                }
                ++pos;
            }
            if (run < L) cx.atomic_add_f64(&hacc[(run - 1) * S + state], Pn);
        }
        cx.sync();
    }
    cx.sync();
    for (int i = tid; i < nbins; i += nt) a.partials[(size_t)cx.block() * nbins + i] = hacc[i];
}
