// Track-likelihood / state-posterior recursion: the kernel body shared by the HIP kernels
// (extrack_hip.hip) and by the CPU-thread emulator used in tests (tests/emul).
//
// What it computes (reference: extrack/tracking.py:109-318 P_Cs_inter_bound_stats, fixed window,
// + extrack/tracking_0.py:440-458 Proba_Cs; exact statement in SURVEY.md Appendix A):
//   per track: LL = log sum_{state sequences} P(track, sequence | model)   with sequences older than
//   frame_len states merged by moment matching, and (PREDS) the per-position state posteriors.
//
// How it is organised for CDNA4 (this is NOT the reference's data flow):
//   * A track's S^F live sequences sit in LDS as struct-of-arrays {zm, ze, m[D], u[K]} holding the
//     state AFTER the Gaussian integration of a position: weight z = zm*2^ze (linear domain,
//     extended range), mean m, and u = l2*s2/(l2+s2) (the variance before the next step's
//     diffusion term d2 is added).
//   * Sequence index = F base-S digit "slots" used as a circular buffer: the ns new state digits of
//     a step overwrite the slots of the ns oldest digits, which are exactly the digits the reference
//     fuses away (tracking.py:253-277).  A "group" = the S^ns sequences that differ only in those
//     slots.  Because the expansion terms (transition probability, stay probability, d2) depend only
//     on (newest old digit, new digits) - constant inside a group - the reference's expand -> integrate
//     -> fuse becomes fuse -> expand -> integrate: ONE moment-matching merge per group, then S^ns
//     cheap per-new-digit updates.  One thread owns one group: read S^ns entries, merge, write S^ns
//     entries in place; one workgroup barrier per track position.
//   * Unused slots during warm-up (track shorter than the window) hold zero-weight entries, so the
//     same code path covers warm-up, steady state and short tracks.
//   * The last position and the bleaching/leaving term (isBL) are a pure reduction:
//     sum_{Q,q} z_Q * T[prev][q] * Eend[q_newest] * N(c_last; m_Q, s2_Qq + l2).
#pragma once
#include "xt_math.h"

#define XT_MAX_STATES 8
#define XT_MAX_DIMS 3
#define XT_BLOB_HDR 16  // doubles: [0..2] l2 (global loc. error^2 per dim), [3] slope, [4] offset, [8..15] Fs
#define XT_NTAB 5       // tables [v][prev][q]: 0 T, 1 T*stay, 2 T*Eend, 3 T*stay*Eend, 4 d2

#define XT_MAX_BUCKETS 64  // length buckets served by one launch
#define XT_INLINE_BLOB 176 // doubles: model blobs up to this size are passed inside the kernel arguments (covers <= 4 states, nb_substeps 1)

// One length bucket as seen by a launch that serves several buckets at once: the blocks [blk_end[i-1], blk_end[i]) of the
// grid work on bucket i (XtKernelArgs::blk_end), striding over its track batches.
struct XtBucketDesc {
    const double* tracks;  // [N][L][D]
    const double* sigma;   // [N][L][KS] or nullptr
    double* ll_out;        // [N] or nullptr
    double* preds_out;     // [N][L][S] or nullptr
    int64_t N;
    int32_t L, isBL;
    double ll_const;       // -(L-1)*D/2*log(2*pi)
    double* seq_out = nullptr;  // [N][E][G] log-weight of every (stored sequence, new digits) at the last position, or nullptr (xt_seqmat.h)
};

struct XtKernelArgs {
    const XtBucketDesc* desc; // device array [ndesc], or nullptr: single bucket described by the fields below
    int32_t ndesc;
    int32_t blk_end[XT_MAX_BUCKETS];  // exclusive prefix of blocks per bucket
    const double* tracks;     // [N][L][D] device
    const double* sigma;      // [N][L][KS] per-peak localisation error (std) or nullptr
    const double* blob;       // model blob, XT_BLOB_HDR + XT_NTAB*S*G doubles
    const int32_t* base_tab;  // [P][NG] entry index of the group's q=0 member per phase
    const int32_t* off_tab;   // [P][G]  entry offset of member q per phase
    double* ll_out;           // [N] per-track log-likelihood or nullptr
    double* partials;         // [nblocks] per-block sum of LL
    double* preds_out;        // [N][L][S] (PREDS kernels)
    double* seq_out;          // [N][E][G] per-sequence log-weights at the last position (general kernel only) or nullptr
    int64_t N;
    int32_t L, S, NS, F, G, E, NG, P;
    int32_t EP;               // padded sequence-array length (E, or E + E/32 + E/1024 + 1 with the bank-conflict skew)
    int32_t skew;             // 1: storage index = i + (i >> 5) + (i >> 10)  (power-of-two S: kills the 4^h / 2^h stride conflicts)
    int32_t TPB;              // tracks processed concurrently by one block
    int32_t isBL, min_len;
    int32_t locerr_mode;      // 0 global (blob[0..2]), 1 per-peak sigma, 2 per-peak affine clip(s*slope+offset,1e-6)
    int32_t KS;               // last dim of sigma (1 or D)
    int32_t well_scaled;      // 2-state fast path: lazy re-normalisation + zero-free steps are sound for this launch (xt_model_well_scaled)
    int32_t prev_div;         // S^(F-NS-1): prev digit of group g = g / prev_div
    int32_t pw[16];           // S^i
    double ll_const;          // -(L-1)*D/2*log(2*pi)
    // Fused total (round 4): when `done` is set the LAST block to finish (device counter) sums the per-block partials in a fixed order and
    // writes the evaluation's total to total_out (device) and, when given, total_host (pinned host memory mapped into the device): no
    // separate reduction launch, no device-to-host copy.  The counter is left at zero for the next launch.
    unsigned int* done;
    double* total_out;
    double* total_host;
    // Small model blobs travel in the kernel arguments instead of through a host-to-device copy (one dispatch less per evaluation):
    // blob == nullptr -> the blob is blob_inline (read from the kernarg segment, see xt_blob_ptr).
    double blob_inline[XT_INLINE_BLOB];
};

// The model blob of a launch: device memory, or the copy inside the kernel arguments themselves.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const double* xt_blob_ptr(const XtKernelArgs& a)
{
    if (a.blob) return a.blob;
    // the kernel's (only) argument is the XtKernelArgs struct at offset 0 of the kernarg segment; reading it through the segment pointer
    // keeps a per-lane indexed load from turning the by-value struct into a scratch copy
    return (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(XtKernelArgs, blob_inline));
}
#else
inline const double* xt_blob_ptr(const XtKernelArgs& a) { return a.blob ? a.blob : a.blob_inline; }
#endif

// LDS footprint in doubles.  Layout: [tables][per-track regions x TPB][pred accumulators x TPB]
XT_HD int xt_tab_doubles(int S, int G) { return XT_BLOB_HDR + XT_NTAB * S * G + 64; }  // + T64[j] = 2^(j/64)
XT_HD int xt_region_doubles(int E, int D, int K) { return E * (1 + D + K) + (E + 1) / 2 + 2; }
XT_HD int xt_pred_doubles(int S, int F) { return 2 * (S + 1) + (F + 1) * S + 2; }
#define XT_STAGE 32  // positions of a track staged in LDS per refill (coalesced loads instead of a dependent global load per step)
XT_HD int xt_stage_doubles(int D) { return XT_STAGE * (D + XT_MAX_DIMS); }
// Storage index of logical sequence i.  For power-of-two S consecutive groups are 2^k / 4^k sequences apart, an up to
// 12-way LDS bank conflict on the 8-byte arrays; the additive skew brings it to <= 2-way (tools/lds_conflicts_general.py).
XT_HD int xt_skew(int i, int skew) { return skew ? i + (i >> 5) + (i >> 10) : i; }
XT_HD int xt_padded_entries(int E, int skew) { return skew ? E + (E >> 5) + (E >> 10) + 1 : E; }

// Resolves which bucket this block serves: returns its descriptor by value (registers), the block's index inside the
// bucket (lb) and the number of blocks serving the bucket (nb).  The kernel arguments stay in the constant kernarg segment.
XT_HD XtBucketDesc xt_bind_bucket(const XtKernelArgs& a, int block, int nblocks, int& lb, int& nb)
{
    XtBucketDesc d;
    if (a.desc == nullptr) {
        d.tracks = a.tracks;
        d.sigma = a.sigma;
        d.ll_out = a.ll_out;
        d.preds_out = a.preds_out;
        d.seq_out = a.seq_out;
        d.N = a.N;
        d.L = a.L;
        d.isBL = a.isBL;
        d.ll_const = a.ll_const;
        lb = block;
        nb = nblocks;
        return d;
    }
    int i = 0, lo = 0, hi = a.blk_end[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < XT_MAX_BUCKETS - 1; ++j) {
        const bool past = j < a.ndesc - 1 && block >= a.blk_end[j];
        i = past ? j + 1 : i;
        lo = past ? a.blk_end[j] : lo;
        hi = past ? a.blk_end[j + 1] : hi;
    }
    d = a.desc[i];
    lb = block - lo;
    nb = hi - lo;
    return d;
}

template <int G_, int D, int K, bool PREDS, class Ctx>
XT_HD void xt_track_body(const XtKernelArgs& a, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    const int G = G_ ? G_ : a.G;
    const int S = a.S, E = a.E, EP = a.EP, NG = a.NG, L = b.L, F = a.F;
    const int tid = cx.tid();
    double* smem = cx.smem();

    // ---- model tables -> LDS
    const int ntab = xt_tab_doubles(S, G);
    for (int i = tid; i < ntab; i += cx.nthreads()) smem[i] = xt_blob_ptr(a)[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;

    const int slot = tid / NG;
    const int g = tid - slot * NG;
    const bool tvalid = slot < a.TPB;
    const int rdoubles = xt_region_doubles(EP, D, K);
    double* reg = smem + ((ntab + 1) & ~1) + (tvalid ? slot : 0) * rdoubles;
    double* zm = reg;
    double* mm = zm + EP;
    double* uu = mm + D * EP;
    int* ze = (int*)(uu + K * EP);
    int* red_e = ze + ((EP + 1) & ~1);  // [2] ints: final-reduce exponent, spare
    double* pbase = smem + ((ntab + 1) & ~1) + a.TPB * rdoubles + (tvalid ? slot : 0) * xt_pred_doubles(S, F);
    // PREDS accumulators: pe[2] (ints, in one double), pacc[2][S], facc[F+1][S]
    int* pe = (int*)pbase;
    double* pacc = pbase + 2;
    double* facc = pacc + 2 * S;
    double* spos = smem + ((ntab + 1) & ~1) + a.TPB * (rdoubles + xt_pred_doubles(S, F)) + (tvalid ? slot : 0) * xt_stage_doubles(D);
    double* ssig = spos + XT_STAGE * D;

    const int prev = g / a.prev_div;
    const double* T0 = TAB + (0 * S + prev) * G;
    const double* T1 = TAB + (1 * S + prev) * G;
    const double* TD2 = TAB + (4 * S + prev) * G;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;

    double block_ll = 0.0;  // meaningful in thread g == 0 of each slot
    // Posterior sums: when a track's groups are an aligned power-of-two range of lanes (or whole wavefronts) the per-thread terms are
    // summed inside that range first (DevCtx::group_sum_f64) and ONE thread per range issues the LDS atomic - hundreds of threads
    // hammering S addresses serialise in the LDS.  Every thread of the workgroup takes part in the lane exchange (idle slots add zeros).
    const bool lane_sums = PREDS && (NG & (NG - 1)) == 0 && NG >= 2;
    const int gl = NG >= 64 ? 64 : NG;
    auto lane_sum = [&](double v) -> double {
        switch (gl) {
            case 64: return cx.template group_sum_f64<64>(v);
            case 32: return cx.template group_sum_f64<32>(v);
            case 16: return cx.template group_sum_f64<16>(v);
            case 8: return cx.template group_sum_f64<8>(v);
            case 4: return cx.template group_sum_f64<4>(v);
            default: return cx.template group_sum_f64<2>(v);
        }
    };
    const int64_t nbatch = (b.N + a.TPB - 1) / a.TPB;
    if (tvalid && g == 0) red_e[1] = 0;
    cx.sync();
    // compile-time group size: the table rows of this thread's (constant) newest old digit live in registers
    double T0r[G_ ? G_ : 1], T1r[G_ ? G_ : 1], D2r[G_ ? G_ : 1];
    if (G_)
        for (int q = 0; q < (G_ ? G_ : 1); ++q) {
            T0r[q] = T0[q];
            T1r[q] = T1[q];
            D2r[q] = TD2[q];
        }

    for (int64_t batch = lb; batch < nbatch; batch += nb) {
        const int64_t trk = batch * a.TPB + slot;
        const bool act = tvalid && trk < b.N;
        const double* c = b.tracks + (act ? trk : 0) * (int64_t)L * D;
        const double* sg = b.sigma ? b.sigma + (act ? trk : 0) * (int64_t)L * a.KS : nullptr;

        // positions [p0, p0 + XT_STAGE) of this track -> LDS, by the track's own threads (coalesced along the track)
        auto stage = [&](int p0) {
            if (act) {
                for (int i = g; i < XT_STAGE * D; i += NG)
                    if (p0 + i / D < L) {
                        const double v = c[p0 * D + i];
                        spos[i] = v;
                        if (v != v) red_e[1] = 1;  // NaN input: the track's results become NaN, as in the reference
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
        auto load_l2 = [&](int pos, double* l2) {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k) {
                    double s = ssig[(pos & (XT_STAGE - 1)) * a.KS + (a.KS == 1 ? 0 : k)];
                    if (a.locerr_mode == 2) {
                        s = xt_fma(s, hdr[3], hdr[4]);
                        s = s < 1e-6 ? 1e-6 : s;
                    }
                    l2[k] = s * s;
                }
            }
        };

        stage(0);
        // ---- position 0: one digit (initial state) in slot 0, everything else zero weight
        if (act) {
            double l20[K], c0[D];
            load_l2(0, l20);
            for (int d = 0; d < D; ++d) c0[d] = spos[d];
            for (int il = g; il < E; il += NG) {
                const bool live = il < S;
                const int i = xt_skew(il, a.skew);
                zm[i] = live ? hdr[8 + il] : 0.0;
                ze[i] = live ? 0 : XT_EMIN;
                for (int d = 0; d < D; ++d) mm[d * EP + i] = c0[d];
                for (int k = 0; k < K; ++k) uu[k * EP + i] = l20[k];
            }
            if (g == 0) {
                red_e[0] = XT_EMIN;
                if (PREDS) {
                    pe[0] = pe[1] = XT_EMIN;
                    for (int i = 0; i < 2 * S + (F + 1) * S; ++i) pacc[i] = 0.0;
                }
            }
        }
        cx.sync();

        // ---- positions 1 .. L-2: fuse the group, expand by the new digits, integrate position t
        for (int t = 1; t <= L - 2; ++t) {
            if ((t & (XT_STAGE - 1)) == 0) stage(t);
            const int ph = (t - 1) % a.P;
            const bool do_pred = PREDS && t >= F;
            const int par = t & 1;
            XtAcc pq[PREDS ? (G_ ? G_ : 1) : 1];
            if (act) {
                const int base = a.base_tab[ph * NG + g];
                const int32_t* off = a.off_tab + ph * G;
                double ct[D], l2t[K];
                for (int d = 0; d < D; ++d) ct[d] = spos[(t & (XT_STAGE - 1)) * D + d];
                load_l2(t, l2t);
                const bool stay = t >= stay_from;
                const double* TTl = stay ? T1 : T0;
                auto TT = [&](int q) { return G_ ? (stay ? T1r[G_ ? q : 0] : T0r[G_ ? q : 0]) : TTl[q]; };
                auto TDD = [&](int q) { return G_ ? D2r[G_ ? q : 0] : TD2[q]; };

                int emax = XT_EMIN;
                for (int q = 0; q < G; ++q) {
                    const int e = ze[xt_skew(base + off[q], a.skew)];
                    emax = e > emax ? e : emax;
                }
                double W = 0.0, mb[D], ub[K];
                for (int d = 0; d < D; ++d) mb[d] = 0.0;
                for (int k = 0; k < K; ++k) ub[k] = 0.0;
                for (int q = 0; q < G; ++q) {
                    const int idx = xt_skew(base + off[q], a.skew);
                    const double aq = xt_ldexp(zm[idx], ze[idx] - emax);
                    W += aq;
                    for (int d = 0; d < D; ++d) mb[d] = xt_fma(aq, mm[d * EP + idx], mb[d]);
                    for (int k = 0; k < K; ++k) ub[k] = xt_fma(aq, uu[k * EP + idx], ub[k]);
                }

                if (do_pred) {
                    // posterior of the digit about to be fused away, weighted by the predictive density of
                    // position t (tracking.py:255-271; note the reference's missing 1/2 on the log term)
                    int pemax = XT_EMIN;
                    for (int Q = 0; Q < G; ++Q) {
                        const int idx = xt_skew(base + off[Q], a.skew);
                        pq[Q].clear();
                        const double zq = zm[idx];
                        if (zq != 0.0) {
                            double dq[D], uq[K], dsq = 0.0;
                            for (int d = 0; d < D; ++d) {
                                dq[d] = ct[d] - mm[d * EP + idx];
                                dsq = xt_fma(dq[d], dq[d], dsq);
                            }
                            for (int k = 0; k < K; ++k) uq[k] = uu[k * EP + idx];
                            for (int q = 0; q < G; ++q) {
                                double quad, gf;
                                // (S^2 of these per group and step: reciprocal with one Newton step and a degree-4 exponential, ~1e-13 on
                                // a weight that ends in a posterior compared at 1e-9)
                                if (K == 1) {
                                    const double r = xt_rcp_fast(TDD(q) + uq[0] + l2t[0]);
                                    quad = 0.5 * dsq * r;
                                    gf = r;
                                    for (int d = 1; d < D; ++d) gf *= r;
                                } else {
                                    quad = 0.0;
                                    gf = 1.0;
                                    for (int d = 0; d < D; ++d) {
                                        const double r = xt_rcp_fast(TDD(q) + uq[d] + l2t[d]);
                                        quad = xt_fma(0.5 * dq[d] * dq[d], r, quad);
                                        gf *= r;
                                    }
                                }
                                double p;
                                int j, n;
                                xt_exp_tab_fast(-quad, p, j, n);
                                pq[Q].add(zq * TT(q) * (gf * T64[j]) * p, ze[idx] + n);
                            }
                        }
                        pemax = pq[Q].e > pemax ? pq[Q].e : pemax;
                    }
                    if (pemax > XT_EMIN) cx.atomic_max_i32(&pe[par], pemax);
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
                    const int idx = xt_skew(base + off[q], a.skew);
                    const double d2 = TDD(q);
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
                    zm[idx] = (Wm * TT(q)) * (gf * T64[j]) * p;  // Wm == 0 for an all-zero group, whose We is XT_EMIN
                    ze[idx] = en > XT_EMIN ? en : XT_EMIN;
                    for (int d = 0; d < D; ++d) mm[d * EP + idx] = xt_fma(dm[d], tt[K == 1 ? 0 : d], mb[d]);
                    for (int k = 0; k < K; ++k) uu[k * EP + idx] = l2t[k] * tt[k];
                }
            }
            cx.sync();
            if (PREDS && do_pred) {
                const int pem = act ? pe[par] : 0;
                if (lane_sums) {
                    for (int Q = 0; Q < G; ++Q) {
                        const double sQ = lane_sum((act && pq[Q].m != 0.0) ? xt_ldexp(pq[Q].m, pq[Q].e - pem) : 0.0);
                        if (act && (g & (gl - 1)) == 0 && sQ != 0.0) cx.atomic_add_f64(&pacc[par * S + Q], sQ);
                    }
                } else if (act) {
                    for (int Q = 0; Q < G; ++Q)
                        if (pq[Q].m != 0.0) cx.atomic_add_f64(&pacc[par * S + Q], xt_ldexp(pq[Q].m, pq[Q].e - pem));
                }
                if (act) {
                    if (g < S) pacc[(par ^ 1) * S + g] = 0.0;
                    if (g == 0) pe[par ^ 1] = XT_EMIN;
                }
                cx.sync();
                if (act && g < S) {
                    double tot = 0.0;
                    for (int s = 0; s < S; ++s) tot += pacc[par * S + s];
                    b.preds_out[(trk * L + (t - F)) * S + g] = pacc[par * S + g] / tot;
                }
            }
        }

        // ---- last position (+ leaving/bleaching term): pure reduction over (old entry Q, new digits q)
        if (((L - 1) & (XT_STAGE - 1)) == 0) stage(L - 1);
        XtAcc tot;
        tot.clear();
        XtAcc accQ[PREDS ? (G_ ? G_ : 1) : 1], accq[PREDS ? (G_ ? G_ : 1) : 1];
        if (act) {
            const int tl = L - 1;
            const int ph = (tl - 1) % a.P;
            const int base = a.base_tab[ph * NG + g];
            const int32_t* off = a.off_tab + ph * G;
            const int vfin = (b.isBL ? 2 : 0) + (tl >= stay_from ? 1 : 0);
            const double* TF = TAB + (vfin * S + prev) * G;
            double cl[D], l2l[K];
            for (int d = 0; d < D; ++d) cl[d] = spos[(tl & (XT_STAGE - 1)) * D + d];
            load_l2(tl, l2l);
            if (PREDS)
                for (int q = 0; q < G; ++q) {
                    accQ[q].clear();
                    accq[q].clear();
                }
            for (int Q = 0; Q < G; ++Q) {
                const int idx = xt_skew(base + off[Q], a.skew);
                const double zq = zm[idx];
                if (zq == 0.0) {
                    if (b.seq_out)
                        for (int q = 0; q < G; ++q) b.seq_out[((int64_t)trk * E + base + off[Q]) * G + q] = -INFINITY;
                    continue;
                }
                const int eq = ze[idx];
                double dq[D], uq[K], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - mm[d * EP + idx];
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                for (int k = 0; k < K; ++k) uq[k] = uu[k * EP + idx];
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
                    tot.add(wm, we);
                    if (b.seq_out) b.seq_out[((int64_t)trk * E + base + off[Q]) * G + q] = wm > 0.0 ? log(wm) + (double)we * XT_LN2 + b.ll_const : (wm == 0.0 ? -INFINITY : NAN);
                    if (PREDS) {
                        accQ[Q].add(wm, we);
                        accq[q].add(wm, we);
                    }
                }
            }
            if (tot.m != 0.0) cx.atomic_max_i32(&red_e[0], tot.e);
        }
        cx.sync();  // all reads of the state are done: zm can be reused as reduction scratch
        int fe = XT_EMIN;
        if (act) fe = red_e[0];
        if (lane_sums) {
            // column 0 = newest digit q; column F = fused slot digit Q: the same S addresses for every thread of the track
            for (int q = 0; q < G; ++q) {
                const double s0 = lane_sum((act && accq[q].m != 0.0) ? xt_ldexp(accq[q].m, accq[q].e - fe) : 0.0);
                const double sF = lane_sum((act && L - 1 >= F && accQ[q].m != 0.0) ? xt_ldexp(accQ[q].m, accQ[q].e - fe) : 0.0);
                if (act && (g & (gl - 1)) == 0) {
                    if (s0 != 0.0) cx.atomic_add_f64(&facc[0 * S + q], s0);
                    if (sF != 0.0) cx.atomic_add_f64(&facc[F * S + q], sF);
                }
            }
        }
        if (act) {
            zm[g] = tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0;
            if (PREDS) {
                const double al = zm[g];
                // columns 1..F-1 = digits of g
                if (!lane_sums)
                    for (int q = 0; q < G; ++q) {
                        if (accq[q].m != 0.0) cx.atomic_add_f64(&facc[0 * S + q], xt_ldexp(accq[q].m, accq[q].e - fe));
                        if (L - 1 >= F && accQ[q].m != 0.0) cx.atomic_add_f64(&facc[F * S + q], xt_ldexp(accQ[q].m, accQ[q].e - fe));
                    }
                if (al != 0.0)
                    for (int j = 1; j <= F - 1 && j <= L - 1; ++j) {
                        const int dig = (g / a.pw[F - j - 1]) % S;
                        cx.atomic_add_f64(&facc[j * S + dig], al);
                    }
            }
        }
        cx.sync();
        const bool poisoned = act && red_e[1] != 0;
        if (act && g == 0) {
            double sum = 0.0;
            for (int i = 0; i < NG; ++i) sum += zm[i];
            const double ll = poisoned ? NAN : log(sum) + (double)fe * XT_LN2 + b.ll_const;
            if (b.ll_out) b.ll_out[trk] = ll;
            block_ll += ll;
        }
        if (PREDS && act) {
            const int ncol = (L - 1 < F ? L - 1 : F) + 1;
            for (int i = g; i < ncol * S; i += NG) {
                const int j = i / S;
                double tots = 0.0;
                for (int s = 0; s < S; ++s) tots += facc[j * S + s];
                b.preds_out[(trk * L + (L - 1 - j)) * S + (i - j * S)] = facc[i] / tots;
            }
            if (poisoned)
                for (int i = g; i < L * S; i += NG) b.preds_out[trk * L * S + i] = NAN;
        }
        cx.sync();  // scratch (zm) and accumulators are re-initialised by the next batch
        if (act && g == 0) red_e[1] = 0;
    }

    // ---- block partial: fixed-order sum over the block's track slots
    cx.sync();
    if (tvalid && g == 0) smem[slot] = block_ll;
    cx.sync();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < a.TPB; ++i) s += smem[i];
        a.partials[cx.block()] = s;
    }
}
