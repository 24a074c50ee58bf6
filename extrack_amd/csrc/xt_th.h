// Threshold-fusion track likelihood: the kernel bodies shared by the HIP kernels (extrack_hip.hip) and by the
// CPU-thread emulator used in tests (tests/emul).
//
// What it computes (reference, relative to /root/reference/):
//   extrack/tracking.py:427-650  P_Cs_inter_bound_stats_th   (the variant param_fitting / predict_Bs call in v1.6.3)
//   extrack/tracking.py:652-743  fuse_tracks_th              (greedy, data-dependent grouping of state sequences)
//   extrack/tracking.py:769-787  Proba_Cs                    (per-track log-sum)
// The reference decides WHICH sequences to merge at a step from the first 30 tracks of a chunk ("pilot" tracks,
// tracking.py:678-679) and applies that decision to every track of the chunk (2000 tracks in cum_Proba_Cs,
// tracking.py:1043).  The result is therefore a function of (track, chunk), not of the track alone.
//
// How it is organised for CDNA4 (NOT the reference's data flow):
//   * PLAN kernel (xt_th_plan_body<D,K,false>): one workgroup per chunk walks the recursion for the <= 30 pilot tracks and runs
//     the greedy grouping on the device; output = the "plan" of the chunk: for every step the member lists of the merge groups
//     (CSR: members[] sorted by group, gstart[]), members pre-resolved to (parent sequence, table offset) words.  All
//     (pivot, candidate) tests of a step are evaluated in one parallel phase into a bit matrix (counts via wave ballots), the
//     greedy scan itself is bit arithmetic.  The state-history bookkeeping the reference uses for its "same last frame_len
//     states" rule (the `cat` array, averaged with np.mean at every merge, tracking.py:729) is carried in fp64 with numpy's
//     summation order, so that argmax ties resolve as they do in the reference.
//   * APPLY kernel (xt_th_apply_body<D,K,UNI,SINGLE>): every track of the chunk follows the plan.  Sequence weights are
//     linear-domain extended-range numbers (zm * 2^ze) as in xt_kernel.h.  The Gaussian integration of a position depends only
//     on the PARENT sequence (mean, variance), not on the new state digits, so it is done once per parent (one exp, one rcp)
//     and the expansion by the S^ns new digits is folded into the merge: a group's new weight/mean/variance is a gather over
//     its members (parent, new digits) of table lookups and FMAs.  The expanded S^ns-fold array of the reference never
//     exists.  State lives in LDS as one record per (sequence, track) with TT tracks of ONE chunk per workgroup (same plan ->
//     uniform control flow); with TT = 64 a wavefront is the 64 tracks for one sequence, every plan / table index is
//     wave-uniform and read with scalar loads, and the merge is fused with the next integration in registers.
//   * POSTERIORS (xt_th_plan_body<D,K,true>): predict_Bs cuts a bucket into chunks of nb_max tracks; the (at most 30) first
//     tracks of a chunk are the pilots and the plan kernel itself serves them: forward pass with per-track histories truncated to
//     frame_len, the merge weights recorded, then one backward pass over the merge tree reads the posteriors out.  Tracks
//     beyond the pilots (nb_max > 30) replay the recorded plan with their own merge weights, 30 at a time in the pilots' slots.
//   * One launch of each kernel serves all length buckets of a dataset through a table of bucket descriptors (XtThBucket).
#pragma once
#include <type_traits>

#include "xt_kernel.h"

#define XT_TH_PILOT 30  // tracking.py:678-679
// Phase timers of the plan kernel (development aid: build with -DXT_TH_PROFILE, workgroup 0 prints its cycle counts per phase)
#if defined(XT_TH_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define XT_TH_TICK(i)                                   \
    do {                                                \
        const long long now__ = wall_clock64();         \
        prof__[i] += now__ - last__;                    \
        last__ = now__;                                 \
    } while (0)
#define XT_TH_PROF_DECL long long prof__[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last__ = wall_clock64()
#define XT_TH_PROF_DUMP(tag)                                                                                                    \
    if (cx.block() == 0 && tid == 0)                                                                                            \
    printf("th plan phases (100 MHz ticks) %s: integrate %lld  stds+stage %lld  pairs %lld  greedy %lld  members %lld  merge %lld  keys %lld  other %lld\n", \
           tag, prof__[0], prof__[1], prof__[2], prof__[3], prof__[4], prof__[5], prof__[6], prof__[7])
#else
#define XT_TH_TICK(i) \
    do {              \
    } while (0)
#define XT_TH_PROF_DECL \
    do {                \
    } while (0)
#define XT_TH_PROF_DUMP(tag) \
    do {                     \
    } while (0)
#endif
#define XT_TH_STAGE 8   // positions staged in LDS per refill (apply kernel)
#define XT_TH_MAXCAP 8192       // expanded sequences per step whose plan arrays (10 B each) stay in LDS: prediction / refinement modes stop here
#define XT_TH_MAXCAP_FIT 32768  // likelihood (fit) mode: beyond XT_TH_MAXCAP the per-step plan arrays move to the global workspace (XtThArgs::plan_glb)
#define XT_TH_CMAT_WORDS 2048  // LDS budget (32-bit words) of the pivot -> candidate compatibility bit matrix
#define XT_TH_GPW 4      // single-buffer apply kernel: merge groups per wavefront held in registers

// Per-bucket part of the arguments: a launch may serve all length buckets of a dataset at once (XtThArgs::buckets), the
// global chunk index of a workgroup is then resolved to (bucket, chunk of that bucket) through XtThArgs::chunk_end.
struct XtThBucket {
    const double* tracks;  // [N][L][D]
    const double* sigma;   // [N][L][KS] or nullptr
    const double* dt;      // [N][L] per-track time steps (tracking.py:979-982) or nullptr: then the model's ds already contain dt
    double* ll_out;        // [N] or nullptr
    double* preds_out;     // [N][L][S] (prediction kernel)
    int64_t N;
    int32_t L, isBL;
    double ll_const;       // -(L-1)*D/2*log(2*pi)
    uint16_t* members;     // plan arrays of this bucket, see XtThArgs
    uint32_t* mpack;
    uint16_t* gstart;
    uint8_t* gnew;
    int32_t* hdr;
    int32_t* status;
    double* seq_out;       // [N][seq_stride] log-probability of every (final parent, new digits) sequence at the last position, WITHOUT the
    int32_t seq_stride;    // leaving term (extrack_sequence_matrix_th), or nullptr
};

struct XtThArgs {
    const double* tracks;  // [N][L][D]
    const double* sigma;   // [N][L][KS] or nullptr
    const double* blob;    // model tables, [prev][r] with r in the reference's digit order (digit 0 = newest sub-state)
    int64_t blob_stride;   // 0: one blob for every chunk; else doubles between the blobs of consecutive (global) chunks - with per-track
                           // time steps the field-of-view table, hence the stay / end-of-track tables, belongs to the chunk (tracking.py:507-511)
    const double* dt;      // single-bucket form of XtThBucket::dt
    double* ll_out;        // [N] or nullptr
    double* partials;      // [grid] per-block LL sums (apply kernel)
    int64_t N;
    int32_t L, S, NS, G, F;
    int32_t isBL, min_len, locerr_mode, KS;
    int32_t chunk, nchunks;
    int32_t capE;          // plan capacity: expanded sequences per step
    int32_t max_nb;        // max_nb_states (tracking.py:601-602: threshold *= 1.2 while exceeded)
    double threshold;
    double ll_const;       // -(L-1)*D/2*log(2*pi)
    uint16_t* members;     // [nchunks][L][capE]  expanded index j of every member, sorted by group (diagnostics)
    uint32_t* mpack;       // [nchunks][L][capE]  the same members as (parent << 16 | newest(parent) * G + r): what the apply kernel reads
    uint16_t* gstart;      // [nchunks][L][capE + 1]
    uint8_t* gnew;         // [nchunks][L][capE]  newest state of every group after the merge
    int32_t* hdr;          // [nchunks][L][2]: nE (expanded sequences at step t), nG (groups after the merge, 0 if none)
    int32_t* status;       // [nchunks][4]: overflow flag, max nE, max nG, sum over merged steps of nE
    double* ws;            // plan-kernel workspace in global memory, ws_stride doubles per workgroup (ws_lds == 0)
    int64_t ws_stride;
    int32_t pcap;          // pilot-track capacity of the workspace: min(30, chunk)
    double* preds_out;     // [N][L][S] state posteriors (prediction kernel)
    int32_t ws_lds;        // 1: the pilot-track state lives in LDS (capacities learned from the previous evaluation)
    int32_t wsP, wsE;      // workspace capacities: parent sequences / expanded sequences per pilot track
    int32_t cmat_words;        // LDS words reserved for the compatibility bit matrix (0: XT_TH_CMAT_WORDS)
    int32_t plan_bs;           // grouping with more than 64 sequences: batches of wavefronts x max(plan_bs, 1) pivot rows with the greedy scan in
                               // between; < 0: all rows of a step in one batch
    int32_t pair_lanes_max_p;  // pilot counts up to this use one lane per (pivot, candidate) pair in the grouping, more use ballots
    int32_t stP, stE;      // global workspace only: capacities of the LDS staging copy of the pilots' means / stds that the
                           // grouping reads (0: none); steps with more sequences read the workspace directly
    int32_t plan_glb;      // 1 (global workspace, fit mode only): the per-step plan arrays (member words, newest states, member / group-start
                           // lists: 10 B per expanded sequence) live at the head of the workgroup's workspace slice instead of LDS - capE > XT_TH_MAXCAP
    double* seq_out;            // single-bucket form of XtThBucket::seq_out / seq_stride
    int32_t seq_stride;
    const XtThBucket* buckets;  // device array [nbuckets], or nullptr: the single bucket described by the fields above
    const int32_t* chunk_end;   // device array [nbuckets]: exclusive prefix sum of the buckets' chunk counts
    int32_t nbuckets;
    int32_t Lmax;               // longest track length of the launch (LDS sizing of the apply kernel)
    int32_t TT, logTT;     // apply kernel: tracks per workgroup tile (power of two)
    int32_t capG;          // apply kernel: parent-sequence capacity of the LDS buffers
    int32_t bpc;           // apply kernel: workgroups per chunk (a workgroup serves tiles of ONE chunk)
    int32_t plan_cap;      // apply kernel: members of ALL merged steps kept in LDS (0: the plan is streamed step by step; < 0: read from global memory)
    // Position refinement (extrack/refined_localization.py:48-204 get_LC_Km_Ks): the prediction-mode plan kernel run on ONE chunk
    // (the whole bucket) records, after every position, each track's surviving sequences instead of reading out posteriors.
    int32_t refine;        // 1: record mode; every workgroup repeats the pilot pass (same plan) and serves its share of the other tracks
    int32_t rf_cap;        // sequences recorded per (entry, track)
    double* rf_out;        // [L - 1][rf_cap][2 + D][rows]: log-weight, mean[D], std   (nullptr: nothing is recorded - capacity probe)
    uint8_t* rf_new;       // [L - 1][rf_cap] newest state of every recorded sequence (shared by the tracks)
    int32_t* rf_cnt;       // [L - 1] recorded sequences per entry
    int64_t rf_row0, rf_rows;  // this launch records the tracks [rf_row0, rf_row0 + rf_rows) of the bucket (row blocks bound the record memory);
                               // rf_out is then [L - 1][rf_cap][2 + D][rf_rows], rows relative to rf_row0
};

// Pointer to read-only data that is addressed with wave-uniform indices: on the device it lives in the constant address
// space, which lets the compiler use scalar loads (s_load) instead of one vector load per lane.
template <bool C, class T>
struct XtCPtr {
    typedef const T* type;
    static XT_HD type make(const T* p) { return p; }
};
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
struct XtCPtr<true, T> {
    typedef const __attribute__((address_space(4))) T* type;
    static XT_HD type make(const T* p) { return (type)p; }
};
#endif

XT_HD int xt_popc64(unsigned long long v) { return __builtin_popcountll(v); }

// New-state entry the reference writes into a sequence's state HISTORY when sequence q (index at that expansion level) is
// created: it takes the index from np.arange(n, dtype='int8') (tracking.py:542), i.e. wrapped to [-128, 127], modulo S.
// Equal to q % S whenever S is a power of two or q < 128 - otherwise it differs from the sequence's actual new state, and
// since the history decides the merge classes (and is what predict_Bs returns) the wrap is reproduced.
XT_HD int xt_th_cat_digit(int q, int S)
{
    const int w = (int)(int8_t)(q & 0xff);
    const int d = w % S;
    return d < 0 ? d + S : d;
}

// Resolves a global chunk index to its bucket (by value) and the chunk index inside that bucket.
XT_HD XtThBucket xt_th_bind(const XtThArgs& a, int gch, int& lc)
{
    XtThBucket k;
    if (a.buckets == nullptr) {
        k.tracks = a.tracks;
        k.sigma = a.sigma;
        k.dt = a.dt;
        k.ll_out = a.ll_out;
        k.preds_out = a.preds_out;
        k.N = a.N;
        k.L = a.L;
        k.isBL = a.isBL;
        k.ll_const = a.ll_const;
        k.members = a.members;
        k.mpack = a.mpack;
        k.gstart = a.gstart;
        k.gnew = a.gnew;
        k.hdr = a.hdr;
        k.status = a.status;
        k.seq_out = a.seq_out;
        k.seq_stride = a.seq_stride;
        lc = gch;
        return k;
    }
    // the table is read through the constant address space: the chunk index is workgroup-uniform, so the descriptor (and every
    // pointer in it) lands in scalar registers and stays provably uniform for the scalar-load paths of the kernels
    const typename XtCPtr<true, int32_t>::type cend = XtCPtr<true, int32_t>::make(a.chunk_end);
    const typename XtCPtr<true, XtThBucket>::type tab = XtCPtr<true, XtThBucket>::make(a.buckets);
    int lo = 0, hi = a.nbuckets - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (gch >= cend[mid])
            lo = mid + 1;
        else
            hi = mid;
    }
    lc = gch - (lo ? cend[lo - 1] : 0);
    k.tracks = tab[lo].tracks;
    k.sigma = tab[lo].sigma;
    k.dt = tab[lo].dt;
    k.ll_out = tab[lo].ll_out;
    k.preds_out = tab[lo].preds_out;
    k.N = tab[lo].N;
    k.L = tab[lo].L;
    k.isBL = tab[lo].isBL;
    k.ll_const = tab[lo].ll_const;
    k.members = tab[lo].members;
    k.mpack = tab[lo].mpack;
    k.gstart = tab[lo].gstart;
    k.gnew = tab[lo].gnew;
    k.hdr = tab[lo].hdr;
    k.status = tab[lo].status;
    k.seq_out = tab[lo].seq_out;
    k.seq_stride = tab[lo].seq_stride;
    return k;
}

// a / b < thr with the outcome of the correctly rounded IEEE division (what numpy computes, tracking.py:691-694): the
// hardware reciprocal (rel. error < 1e-7) decides everything that is not within 1e-5 of the threshold, the division proper
// is only executed for the rare borderline value.  NaN / zero denominators fall through to the exact expression.
XT_HD bool xt_div_lt(double a, double b, double thr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = a * __builtin_amdgcn_rcp(b);
    if (fabs(q - thr) > 1e-5 * thr) return q < thr;
#endif
    return a / b < thr;
}

// Hardware reciprocal (no Newton step): callers allow for its error and re-do borderline cases exactly.
XT_HD double xt_rcp_raw(double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(b);
#else
    return 1.0 / b;
#endif
}

// View of a parent-sequence state buffer {zm, m[D], u[K], ze} per entry.  u = variance: after a merge s2 incl. the diffusion
// term, after an integration l2*s2/(l2+s2).
//   AOS = false (plan kernel, global or LDS workspace): field planes of `plane` entries.
//   AOS = true  (apply kernel, LDS): one (D+K+2)-double record per entry, so the fields sit at immediate offsets of ONE
//   address computation; a record stride of 10 dwords (D=2, K=1) is bank-conflict free for 32 consecutive lanes.
template <int D, int K, bool AOS>
struct XtThView {
    double* base;
    int plane;
    static constexpr int ES = D + K + 2;
    XT_HD double& zm(int i) const { return AOS ? base[i * ES] : base[i]; }
    XT_HD double& m(int d, int i) const { return AOS ? base[i * ES + 1 + d] : base[(int64_t)plane * (1 + d) + i]; }
    XT_HD double& u(int k, int i) const { return AOS ? base[i * ES + 1 + D + k] : base[(int64_t)plane * (1 + D + k) + i]; }
    XT_HD int& ze(int i) const { return AOS ? ((int*)base)[(i * ES + 1 + D + K) * 2] : ((int*)(base + (int64_t)plane * (1 + D + K)))[i]; }
};

XT_HD int xt_th_hm(int F, int NS) { return F + NS; }
XT_HD int64_t xt_th_buf_doubles(int plane, int D, int K) { return (int64_t)plane * (2 + D + K); }
XT_HD int64_t xt_th_cmat_doubles(int wsE) { return ((int64_t)wsE * ((wsE + 32) / 32 + 1) + 1) / 2 + 1; }
// Workspace of one workgroup of the plan kernel.  State part (LDS or global): pilot sequences, stds, histories, keys,
// bit matrix (+ prediction mode: member / final weights, sequence masses).  History part (prediction mode, always global):
// what the backward pass reads - members, group starts and counts of every merge step, the members' weights per track.
XT_HD int64_t xt_th_hist_doubles(int wsE, int pcap, bool preds, int L)
{
    if (!preds) return 0;
    return 2 * (((int64_t)L * wsE + 1) / 2) + ((int64_t)L * (wsE + 1) + 3) / 4 + L + (int64_t)pcap * L * wsE + 4;
}
XT_HD int64_t xt_th_ws_doubles(int wsP, int wsE, int D, int K, int F, int NS, int S, int pcap = XT_TH_PILOT, bool preds = false)
{
    const int HM = xt_th_hm(F, NS);
    const int NC = preds ? pcap : 1;
    int64_t n = 2 * xt_th_buf_doubles(pcap * wsP, D, K) + (int64_t)K * pcap * wsE + 2 * (int64_t)NC * wsP * HM * S + 2 * (int64_t)NC * wsP +
                xt_th_cmat_doubles(wsE) + (wsE + 63) / 64 + 1 + 8;
    if (preds) n += (int64_t)pcap * wsE + ((int64_t)pcap * wsE + 1) / 2 + 2 * (int64_t)pcap * wsP + 8;
    return n;
}
XT_HD int64_t xt_th_plan_glb_doubles(int capE) { return (10 * (int64_t)capE + 2 + 7) / 8 + 2; }
XT_HD int xt_th_plan_lds_doubles(int S, int G, int capE, int D, int K, int cmat_words = XT_TH_CMAT_WORDS, bool plan_glb = false)
{
    // tables | per-track scalars | wave counts | compatibility bit matrix + grouped flags | bytes: mpk u32[capE], newest[2][capE],
    // mem u16[capE], gst u16[capE + 1] (plan_glb: those in the global workspace)
    (void)D;
    (void)K;
    const int bytes = plan_glb ? 0 : 4 * capE + 2 * capE + 2 * capE + 2 * (capE + 1);
    return ((xt_tab_doubles(S, G) + 1) & ~1) + XT_TH_PILOT + 8 + (cmat_words + 1) / 2 + (capE + 63) / 64 + 1 + (bytes + 7) / 8 + 2;
}
XT_HD int xt_th_apply_lds_doubles(int S, int G, int capG, int TT, int D, int K, int KS, int L, int plan_cap, bool uni, bool single = false)
{
    const int plane = capG * TT;
    const int capEl = capG * G;
    // plan region: member words + group starts (+ per-step offsets) for all steps (plan_cap > 0) or for one step;
    // none in the wave-uniform mode (scalar loads from global memory)
    // plan_cap < 0 ("direct"): the general variants read the member lists straight from global memory - for models whose largest step
    // (n_states^(nb_substeps + 1) parents x n_states^nb_substeps digits: 16 384 members for 4 states x 3 substeps) would fill the LDS on its own
    const int pm = (uni || plan_cap < 0) ? 0 : (plan_cap > 0 ? plan_cap : capEl);
    const int pg = (uni || plan_cap < 0) ? 0 : (plan_cap > 0 ? plan_cap + L : capEl + 1);
    const int bytes = 4 * pm + 2 * pg + 16 * L + capG + 4 * TT + 16;
    return ((xt_tab_doubles(S, G) + 1) & ~1) + (single ? 1 : 2) * (int)xt_th_buf_doubles(plane, D, K) + XT_TH_STAGE * (D + KS) * (TT + 1) + TT +
           (bytes + 7) / 8 + 2;
}

template <class V>
XT_HD void xt_th_carve(double*& w, V& b, int plane, int D, int K)
{
    b.base = w;
    b.plane = plane;
    w += xt_th_buf_doubles(plane, D, K);
}

// Gaussian integration of one position into one parent sequence, in place (tracking.py:76-98 log_integrale_dif without
// the diffusion term, which depends on the new digits and is added in the gather).
template <int D, int K, class V>
XT_HD void xt_th_integrate(const V& b, int idx, const double* c, const double* l2, const double* T64)
{
    const double z = b.zm(idx);
    double dm[D], dsq = 0.0;
    for (int d = 0; d < D; ++d) {
        dm[d] = c[d] - b.m(d, idx);
        dsq = xt_fma(dm[d], dm[d], dsq);
    }
    double quad, gf, tt[K];
    if (K == 1) {
        const double s2 = b.u(0, idx);
        const double r = xt_rcp(l2[0] + s2);
        tt[0] = s2 * r;
        quad = 0.5 * dsq * r;
        gf = xt_pow_half<D>(r);
    } else {
        quad = 0.0;
        gf = 1.0;
        for (int d = 0; d < D; ++d) {
            const double s2 = b.u(d, idx);
            const double r = xt_rcp(l2[d] + s2);
            tt[d] = s2 * r;
            quad = xt_fma(0.5 * dm[d] * dm[d], r, quad);
            gf *= r;
        }
        gf = sqrt(gf);
    }
    double p;
    int j, n;
    xt_exp_tab(-quad, p, j, n);
    b.zm(idx) = z * (gf * T64[j]) * p;
    const int en = b.ze(idx) + n;
    b.ze(idx) = (z != 0.0 && en > XT_EMIN) ? en : XT_EMIN;
    for (int d = 0; d < D; ++d) b.m(d, idx) = xt_fma(dm[d], tt[K == 1 ? 0 : d], b.m(d, idx));
    for (int k = 0; k < K; ++k) b.u(k, idx) = l2[k] * tt[k];
}

// One merge group of one track: gather over the members (parent g, table offset o = newest(g) * G + r, packed as
// g << 16 | o) of the integrated parents in `src`; weight = parent weight * T[o], variance = parent variance + d2[o]
// (tracking.py:548-600 expansion + tracking.py:703-741 softmax-weighted merge, in the linear domain).  Two passes: the
// largest exponent of the group, then a branch-free accumulation (zero-weight parents carry exponent XT_EMIN and
// scale to exactly 0).
template <int D, int K, class V, class MemP, class TabP>
XT_HD void xt_th_gather_regs(const V& src, int gs, int xoff, MemP members, int k0, int k1, TabP TT, TabP TD2, double& W, int& E, double* M,
                             double* U, const double d2s = 1.0)  // d2s: this track's time step at this position (1 with a fixed dt)
{
    if (k1 - k0 == 1) {
        const uint32_t pk = members[k0];
        const int o = (int)(pk & 0xffffu), idx = (int)(pk >> 16) * gs + xoff;
        W = src.zm(idx) * TT[o];
        E = src.ze(idx);
        for (int d = 0; d < D; ++d) M[d] = src.m(d, idx);
        for (int k = 0; k < K; ++k) U[k] = xt_fma(TD2[o], d2s, src.u(k, idx));
    } else {
        E = XT_EMIN;
        for (int kk = k0; kk < k1; ++kk) {
            const int e = src.ze((int)(members[kk] >> 16) * gs + xoff);
            E = e > E ? e : E;
        }
        W = 0.0;
        for (int d = 0; d < D; ++d) M[d] = 0.0;
        for (int k = 0; k < K; ++k) U[k] = 0.0;
        for (int kk = k0; kk < k1; ++kk) {
            const uint32_t pk = members[kk];
            const int o = (int)(pk & 0xffffu), idx = (int)(pk >> 16) * gs + xoff;
            const double av = xt_ldexp(src.zm(idx) * TT[o], src.ze(idx) - E);  // hugely negative shift saturates to 0
            W += av;
            for (int d = 0; d < D; ++d) M[d] = xt_fma(av, src.m(d, idx), M[d]);
            for (int k = 0; k < K; ++k) U[k] = xt_fma(av, xt_fma(TD2[o], d2s, src.u(k, idx)), U[k]);
        }
        const double rW = (W == 0.0) ? 0.0 : xt_rcp(W);
        for (int d = 0; d < D; ++d) M[d] *= rW;
        for (int k = 0; k < K; ++k) U[k] *= rW;
    }
    const bool live = W != 0.0;
    E = live ? E + xt_frexp_exp(W) : XT_EMIN;
    W = xt_frexp_mant(W);
}

template <int D, int K, class V, class V2, class MemP, class TabP>
XT_HD void xt_th_gather(const V& src, int gs, int xoff, MemP members, int k0, int k1, TabP TT, TabP TD2, const V2& dst, int didx,
                        const double d2s = 1.0)
{
    double W, M[D], U[K];
    int E;
    xt_th_gather_regs<D, K>(src, gs, xoff, members, k0, k1, TT, TD2, W, E, M, U, d2s);
    dst.zm(didx) = W;
    dst.ze(didx) = E;
    for (int d = 0; d < D; ++d) dst.m(d, didx) = M[d];
    for (int k = 0; k < K; ++k) dst.u(k, didx) = U[k];
}

// Gaussian integration of one position into a merged sequence held in registers; result stored as entry didx of dst.
template <int D, int K, class V>
XT_HD void xt_th_integrate_store(double z, int e, const double* m, const double* s2v, const double* c, const double* l2, const double* T64,
                                 const V& dst, int didx)
{
    double dm[D], dsq = 0.0;
    for (int d = 0; d < D; ++d) {
        dm[d] = c[d] - m[d];
        dsq = xt_fma(dm[d], dm[d], dsq);
    }
    double quad, gf, tt[K];
    if (K == 1) {
        const double r = xt_rcp(l2[0] + s2v[0]);
        tt[0] = s2v[0] * r;
        quad = 0.5 * dsq * r;
        gf = xt_pow_half<D>(r);
    } else {
        quad = 0.0;
        gf = 1.0;
        for (int d = 0; d < D; ++d) {
            const double r = xt_rcp(l2[d] + s2v[d]);
            tt[d] = s2v[d] * r;
            quad = xt_fma(0.5 * dm[d] * dm[d], r, quad);
            gf *= r;
        }
        gf = sqrt(gf);
    }
    double p;
    int j, n;
    xt_exp_tab(-quad, p, j, n);
    dst.zm(didx) = z * (gf * T64[j]) * p;
    const int en = e + n;
    dst.ze(didx) = (z != 0.0 && en > XT_EMIN) ? en : XT_EMIN;
    for (int d = 0; d < D; ++d) dst.m(d, didx) = xt_fma(dm[d], tt[K == 1 ? 0 : d], m[d]);
    for (int k = 0; k < K; ++k) dst.u(k, didx) = l2[k] * tt[k];
}

XT_HD double xt_th_l2_from_sigma(double s, int mode, const double* hdr)
{
    if (mode == 2) {
        s = xt_fma(s, hdr[3], hdr[4]);  // tracking.py:928-930
        s = s < 1e-6 ? 1e-6 : s;
    }
    return s * s;
}

// ------------------------------------------------------------------------------------------------------------------
// PLAN kernel body: one workgroup per chunk.
// ------------------------------------------------------------------------------------------------------------------
// WS: where the pilot-track state lives - 1 = LDS, 0 = global workspace, -1 = decided at run time (a.ws_lds).  With a compile-time
// WS the state pointers have a known address space (LDS loads / stores instead of flat ones in the LDS case).
template <int D, int K, bool PREDS, int WS = -1, class Ctx>
XT_HD void xt_th_plan_body(const XtThArgs& a, Ctx& cx)
{
    const bool ws_lds = WS < 0 ? (a.ws_lds != 0) : (WS == 1);
    const int S = a.S, G = a.G, NS = a.NS, F = a.F, L = a.L, capE = a.capE;
    const int tid = cx.tid(), nt = cx.nthreads();
    const int HM = xt_th_hm(F, NS);                 // history entries kept per sequence
    const int PC = a.pcap;                          // pilot capacity: min(30, chunk)
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    if (a.blob_stride == 0)
        for (int i = tid; i < ntab; i += nt) smem[i] = a.blob[i];
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;
    const double* T64 = TAB + XT_NTAB * S * G;
    const double* TD2 = TAB + 4 * S * G;
    double* pm = smem + ((ntab + 1) & ~1);
    int* wcnt = (int*)(pm + XT_TH_PILOT);  // pm: per-track totals of the final sequence weights (prediction mode)
    uint32_t* cmatL = (uint32_t*)(wcnt + 16);
    const int cmw = a.cmat_words > 0 ? a.cmat_words : XT_TH_CMAT_WORDS;  // LDS words reserved for the bit matrix
    uint32_t* gbitsL = cmatL + cmw + (cmw & 1);
    double* wh = a.ws + (int64_t)cx.block() * a.ws_stride;  // [plan arrays (plan_glb)] history part (prediction mode), then the state part unless it is in LDS
    // per-step plan arrays: LDS, or (compile-time global-workspace variant of the fit mode only, so that every other variant keeps
    // LDS-typed pointers) the head of the workgroup's workspace slice
    const bool plan_glb = WS == 0 && !PREDS && a.plan_glb != 0;
    uint32_t* mpk = plan_glb ? (uint32_t*)wh : gbitsL + 2 * ((capE + 63) / 64) + 2;
    uint8_t* newA = (uint8_t*)(mpk + capE);
    uint8_t* newB = newA + capE;
    uint16_t* mem = (uint16_t*)(newB + capE);
    uint16_t* gst = mem + capE;
    if (plan_glb) wh += xt_th_plan_glb_doubles(capE);

    // pilot-track state: LDS when the learned capacities fit (a.ws_lds), else this workgroup's slice of the global workspace
    // (then the grouping works on an LDS copy of the two arrays it reads over and over: pilots' means and stds)
    const int wsP = a.wsP, wsE = a.wsE, stP = a.stP, stE = a.stE;
    // staging copy: after the LDS-resident state when that is in LDS too (then it only serves to give the compiler LDS-typed
    // addresses instead of flat ones for the hot pair loop)
    double* stM = smem + xt_th_plan_lds_doubles(S, G, capE, D, K, cmw, plan_glb) +
                  (ws_lds ? xt_th_ws_doubles(wsP, wsE, D, K, F, NS, S, a.pcap, PREDS) : 0);  // [PC][stP][D]
    double* stS = stM + (int64_t)a.pcap * stP * D;                  // [PC][stE][K]
    double* w = ws_lds ? smem + xt_th_plan_lds_doubles(S, G, capE, D, K, cmw, plan_glb) : wh + xt_th_hist_doubles(wsE, a.pcap, PREDS, L);
    const int plane = PC * wsE;  // sE plane
    typedef XtThView<D, K, false> View;
    View A, B;
    xt_th_carve(w, A, PC * wsP, D, K);
    xt_th_carve(w, B, PC * wsP, D, K);
    double* sE = w;
    w += (int64_t)K * plane;
    // state history ("cat", tracking.py:515-520): one per chunk in fit mode, one per track when predicting
    const int64_t cstride = (int64_t)wsP * HM * S;  // per pilot
    const int NC = PREDS ? PC : 1;
    double* catA = w;
    w += NC * cstride;
    double* catB = w;
    w += NC * cstride;
    unsigned long long* keyA = (unsigned long long*)w;
    w += (int64_t)NC * wsP;
    unsigned long long* keyB = (unsigned long long*)w;
    w += (int64_t)NC * wsP;
    uint32_t* cmatG = (uint32_t*)w;  // [wsE][ceil(wsE / 32)] pivot -> candidate compatibility bits
    w += xt_th_cmat_doubles(wsE);
    uint32_t* gbitsG = (uint32_t*)w;  // [ceil(wsE / 32)] grouped flags of the greedy scan
    w += (wsE + 63) / 64 + 1;
    double* wgt = w;  // PREDS: normalised member weights / final sequence weights [PC][wsE]
    int* wexp = (int*)(wgt + (PREDS ? plane : 0));
    // PREDS: what the backward pass needs - per merge step the members (parent << 16 | new state), group starts, counts
    // (shared by the chunk) and the members' normalised merge weights per track; two vectors of sequence masses per track
    double* beta = wgt + (PREDS ? plane + (plane + 1) / 2 : 0);  // [PC][2][wsP]
    uint32_t* hmem = (uint32_t*)wh;          // [L][wsE]  (parent << 16 | history entry of the new state)
    wh += PREDS ? ((int64_t)L * wsE + 1) / 2 : 0;
    uint32_t* hmpk = (uint32_t*)wh;          // [L][wsE]  (parent << 16 | table offset): the plan itself, replayed by tracks beyond the pilots
    wh += PREDS ? ((int64_t)L * wsE + 1) / 2 : 0;
    uint16_t* hgst = (uint16_t*)wh;          // [L][wsE + 1]
    wh += PREDS ? ((int64_t)L * (wsE + 1) + 3) / 4 : 0;
    int* hn = (int*)wh;                      // [L][2]
    wh += PREDS ? L : 0;
    double* hw = wh;                         // [PC][L][wsE]
    const int Fk = F - NS;  // parent history entries inside the frame_len window of an expanded sequence
    int pwS[8];
    pwS[0] = 1;
    for (int i = 1; i < 8; ++i) pwS[i] = pwS[i - 1] * S;

    const bool RF = PREDS && a.refine != 0;
    for (int gch = RF ? 0 : cx.block(); gch < a.nchunks; gch += RF ? a.nchunks : cx.nblocks()) {
        int ch;  // chunk index inside its bucket
        const XtThBucket bk = xt_th_bind(a, gch, ch);
        const int L = bk.L;
        const int64_t c0 = (int64_t)ch * a.chunk;
        const int n = (int)((bk.N - c0) < a.chunk ? (bk.N - c0) : a.chunk);
        const int P = n < XT_TH_PILOT ? n : XT_TH_PILOT;
        uint16_t* mem_g = bk.members + (int64_t)ch * L * capE;
        uint32_t* mpk_g = bk.mpack + (int64_t)ch * L * capE;
        uint16_t* gst_g = bk.gstart + (int64_t)ch * L * (capE + 1);
        uint8_t* gnew_g = bk.gnew + (int64_t)ch * L * capE;
        int32_t* hdr_g = bk.hdr + (int64_t)ch * L * 2;
        cx.sync();  // tables loaded / previous chunk done
        if (a.blob_stride != 0) {  // per-track time steps: the chunk's own tables
            for (int i = tid; i < ntab; i += nt) smem[i] = a.blob[(int64_t)gch * a.blob_stride + i];
            cx.sync();
        }
        // time step of track x of the chunk that scales the diffusion term added at step t (tracking.py:494-499, 548-551: column
        // len - current_step of the unreversed array)
        auto dtf = [&](int x, int t) -> double { return bk.dt ? bk.dt[(c0 + x) * L + (L - t)] : 1.0; };

        // refinement record of sequence q of entry `ent` (0 .. L-2) of chunk track xg: log-weight (constants dropped), mean, std
        auto rf_put = [&](int ent, int64_t xg, int q, double zm, int ze, const double* mv, double var0) {
            // layout [entry][sequence][field][track]: the combine kernel reads a field of one sequence for 64 neighbouring tracks at once
            double* o = a.rf_out + (((int64_t)ent * a.rf_cap + q) * (2 + D)) * a.rf_rows + (xg - a.rf_row0);
            o[0] = zm > 0.0 ? log(zm) + (double)ze * XT_LN2 : -INFINITY;
            for (int d = 0; d < D; ++d) o[(int64_t)(1 + d) * a.rf_rows] = mv[d];
            o[(int64_t)(1 + D) * a.rf_rows] = sqrt(var0);
        };
        auto load_l2 = [&](int x, int pos, double* l2) {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k)
                    l2[k] = xt_th_l2_from_sigma(bk.sigma[((c0 + x) * L + pos) * a.KS + (a.KS == 1 ? 0 : k)], a.locerr_mode, hdr);
            }
        };

        // ---- position 0: S parents (the oldest state a), history = [a]
        for (int i = tid; i < P * S; i += nt) {
            const int x = i / S, s = i - x * S, idx = x * wsP + s;
            double l2[K];
            load_l2(x, 0, l2);
            A.zm(idx) = hdr[8 + s];
            A.ze(idx) = 0;
            for (int d = 0; d < D; ++d) A.m(d, idx) = bk.tracks[((c0 + x) * L + 0) * D + d];
            for (int k = 0; k < K; ++k) A.u(k, idx) = l2[k];
        }
        for (int i = tid; i < (PREDS ? P : 1) * S * S; i += nt) {
            const int x = i / (S * S), q = i - x * (S * S);
            catA[x * cstride + (q / S) * HM * S + (q % S)] = (q / S == q % S) ? 1.0 : 0.0;
        }
        for (int i = tid; i < S; i += nt) newA[i] = (uint8_t)i;
        int nPar = S, Hc = 1, maxE = 0, maxG = S, overflow = 0, nfuse = 0, sumE = 0;
        // np.mean(flags) > 0.8 over the P*K flags of a (pivot, candidate) pair <=> count >= cmin, with cmin found with the very same
        // double arithmetic (count / (P*K) > 0.8) once per chunk instead of two divisions per pair
        int cmin = P * K + 1;
        for (int cnt = P * K; cnt >= 0; --cnt)
            if ((double)cnt / (double)(P * K) > 0.8) cmin = cnt;
        XT_TH_PROF_DECL;
        double thr = a.threshold;
        uint8_t *nwA = newA, *nwB = newB;
        double *ctA = catA, *ctB = catB;
        unsigned long long *kyA = keyA, *kyB = keyB;
        View bA = A, bB = B;
        cx.sync();

        for (int t = 1; t <= L - 1; ++t) {
            if (t >= 2) {
                const int pos = t - 1;
                for (int i = tid; i < P * nPar; i += nt) {
                    const int x = i / nPar, g = i - x * nPar;
                    double c[D], l2[K];
                    for (int d = 0; d < D; ++d) c[d] = bk.tracks[((c0 + x) * L + pos) * D + d];
                    load_l2(x, pos, l2);
                    xt_th_integrate<D, K>(bA, x * wsP + g, c, l2, T64);
                }
                cx.sync();
                XT_TH_TICK(0);
            }
            const int nE = nPar * G;
            if (nE > capE || nE > wsE) {
                overflow = 1;
                maxE = nE;
                break;
            }
            if (t >= 2 && nE > a.max_nb) thr = thr * 1.2;  // tracking.py:601-602
            maxE = nE > maxE ? nE : maxE;
            int nG = 0;
            if (t < L - 1) {
                const int He = Hc + NS;
                if (t == 1) {
                    // the initial S^(ns+1) sequences are not merged (tracking.py:479-534): identity plan
                    for (int i = tid; i < nE; i += nt) mem[i] = (uint16_t)i;
                    for (int i = tid; i <= nE; i += nt) gst[i] = (uint16_t)i;
                    nG = nE;
                } else {
                    // ---- greedy grouping on the pilot tracks (tracking.py:652-701), in two phases:
                    // (1) all (pivot b, candidate j > b with the same newest state) pairs in parallel -> bit matrix
                    //     "j may join the group opened by b"; candidates of b are b + S, b + 2S, ...  A wavefront evaluates
                    //     two pairs at a time, 32 pilot slots each; the per-pair counts of the reference's
                    //     np.mean(...) > 0.8 tests come from wave ballots.
                    // (2) the greedy scan itself (lowest ungrouped index opens a group and takes every compatible,
                    //     still ungrouped candidate) is then pure bit arithmetic, done serially by one thread.
                    const bool useA = He > F;
                    // row b of the matrix: bit c <-> candidate b + cstep (c + 1); in LDS when it fits the reserved words (the serial
                    // scan below is latency bound), else in the workspace
                    // candidates of pivot b = later sequences of the same history class (newest history entry).  Normally that is
                    // b + S, b + 2S, ...; with the reference's int8 wrap (see xt_th_cat_digit) the classes are irregular beyond 127
                    // sequences when S is not a power of two: then every later sequence is a candidate and the class is tested.
                    const bool wrapS = (S & (S - 1)) != 0 && nE > 128;
                    const int cstep = wrapS ? 1 : S;
                    const int NWD = (nE / cstep + 32) >> 5;
                    const bool cml = nE * NWD <= cmw;
                    uint32_t* cmat = cml ? cmatL : cmatG;
                    uint32_t* gbits = cml ? gbitsL : gbitsG;
                    for (int i = tid; i < P * nE; i += nt) {
                        const int x = i / nE, jj = i - x * nE, g = jj / G, r = jj - g * G;
                        for (int k = 0; k < K; ++k)
                            sE[k * plane + x * wsE + jj] = sqrt(xt_fma(TD2[(int)nwA[g] * G + r], dtf(x, t), bA.u(k, x * wsP + g)));
                    }
                    for (int i = tid; i < nE * NWD; i += nt) cmat[i] = 0u;
                    const bool staged = stP > 0 && nPar <= stP && nE <= stE;
                    if (staged)
                        for (int i = tid; i < P * nPar * D; i += nt) {
                            const int x = i / (nPar * D), q = i - x * (nPar * D), g = q / D, d = q - g * D;
                            stM[(x * stP + g) * D + d] = bA.m(d, x * wsP + g);
                        }
                    cx.sync();
                    if (staged) {
                        for (int i = tid; i < P * nE * K; i += nt) {
                            const int x = i / (nE * K), q = i - x * (nE * K), jj = q / K, k = q - jj * K;
                            stS[(x * stE + jj) * K + k] = sE[k * plane + x * wsE + jj];
                        }
                        cx.sync();
                    }
                    XT_TH_TICK(1);
                    auto Mv = [&](int d, int x, int g) -> double { return staged ? stM[(x * stP + g) * D + d] : bA.m(d, x * wsP + g); };
                    auto Sv = [&](int k, int x, int jj) -> double { return staged ? stS[(x * stE + jj) * K + k] : sE[k * plane + x * wsE + jj]; };
                    // same history class / same new-state history entries of two expanded sequences
                    // (WRAP is a compile-time flag: the regular case must not pay for the wrapped-index arithmetic)
                    auto same_class = [&](auto WRAP, int jj, int bb) -> bool {
                        if (!decltype(WRAP)::value) return true;
                        return xt_th_cat_digit(jj, S) == xt_th_cat_digit(bb, S);
                    };
                    auto same_digits = [&](auto WRAP, int jj, int bb, int rj, int rb) -> bool {
                        if (!decltype(WRAP)::value) return rj == rb;
                        bool eq = true;
                        for (int c = 0; c < NS; ++c) eq = eq && xt_th_cat_digit(jj / pwS[c], S) == xt_th_cat_digit(bb / pwS[c], S);
                        return eq;
                    };
                    // Wrapped classes: the candidates of a pivot are every third (S-th) later sequence only inside a block of 128 - instead of
                    // testing the class of EVERY later sequence the rows walk per-class lists, clist[first(c) + k] = k-th sequence of class c as
                    // (sequence | parent << 16).  The list lives in the member-word array, which is free until this step's members are known.
                    // class_count(jend, c0, c1): sequences j < jend whose class is in [c0, c1)  (the class advances by one per sequence inside
                    // a block of 128 sequences and jumps between blocks, see xt_th_cat_digit)
                    auto class_count = [&](int jend, int c0, int c1) -> int {
                        int k = 0;
                        for (int t0 = 0; t0 < jend; t0 += 128) {
                            const int n = (jend - t0) < 128 ? (jend - t0) : 128, f = xt_th_cat_digit(t0, S);
                            for (int c = c0; c < c1; ++c) {
                                const int r = (c - f + S) % S;
                                k += n > r ? (n - r + S - 1) / S : 0;
                            }
                        }
                        return k;
                    };
                    const bool clisted = wrapS && P > a.pair_lanes_max_p;
                    uint32_t* clist = mpk;
                    // One pivot row, lanes = pilot tracks (two candidates at a time, 32 pilot slots each; the counts of the reference's
                    // np.mean(...) > 0.8 tests come from wave ballots).  Written for latency: every load of a candidate pair is
                    // unconditional (slots beyond the pilots / the last odd candidate read valid addresses and are masked out of the ballots)
                    // and issued before the arithmetic; a / b < thr is decided by the hardware reciprocal, the correctly rounded divisions run
                    // in a wave-uniform branch taken only when some lane is within 1e-5 of the threshold (or not finite).  STG: the pilots'
                    // means / stds are read from their LDS staging copy (compile-time, so that the loads are LDS loads, not flat ones).
                    auto row_ballot = [&](const int b, auto WRAP, auto STG) {
                        constexpr bool ST = decltype(STG)::value;
                        auto Mq = [&](int d, int x, int g) -> double { return ST ? stM[(x * stP + g) * D + d] : bA.m(d, x * wsP + g); };
                        auto Sq = [&](int k, int x, int jj) -> double { return ST ? stS[(x * stE + jj) * K + k] : sE[k * plane + x * wsE + jj]; };
                        const int lane = cx.lane(), half = lane >> 5, x = lane & 31;
                        const unsigned long long hmask = half ? 0xffffffff00000000ull : 0x00000000ffffffffull;
                        const bool xl = x < P;
                        const int xs = xl ? x : 0;
                        const int gb = b / G, rb = b - gb * G;
                        const double tol = 1e-5 * thr, invD = 1.0 / (double)D, invK = 1.0 / (double)K;
                        double pmv[D], psv[K];
                        for (int d = 0; d < D; ++d) pmv[d] = Mq(d, xs, gb);
                        for (int k = 0; k < K; ++k) psv[k] = Sq(k, xs, b);
                        const unsigned long long keyb = (!PREDS && useA) ? kyA[gb] : 0ull;
                        auto test_pair = [&](const bool valid, const int jj, const int gj, const int rj, const int ci) {
                            const int jv = valid ? jj : b, gv = valid ? gj : gb;
                            double mj[D], sj[K];
                            for (int d = 0; d < D; ++d) mj[d] = Mq(d, xs, gv);
                            for (int k = 0; k < K; ++k) sj[k] = Sq(k, xs, jv);
                            bool same_hist = valid && useA && same_digits(WRAP, jj, b, rj, rb);
                            if (PREDS) {
                                if (same_hist)  // predicting: on every pilot track (mean > 0.999, tracking.py:686)
                                    for (int xx = 0; xx < P; ++xx) same_hist = same_hist && kyA[xx * wsP + gj] == kyA[xx * wsP + gb];
                            } else if (useA) {
                                same_hist = same_hist && kyA[gv] == keyb;
                            }
                            const bool live = valid && xl && !same_hist;
                            double dmn = 0.0, dsd = 0.0;
                            for (int d = 0; d < D; ++d) dmn += fabs(mj[d] - pmv[d]);
                            for (int k = 0; k < K; ++k) dsd += fabs(sj[k] - psv[k]);
                            int cm = 0, cs = 0;
                            for (int k = 0; k < K; ++k) {
                                const double ri = xt_rcp_raw(sj[k]);
                                const double q1 = (dmn * invD) * ri, q2 = (dsd * invK) * ri;
                                bool t1 = q1 < thr, t2 = q2 < thr;
                                const bool border = !(fabs(q1 - thr) > tol) || !(fabs(q2 - thr) > tol);  // incl. NaN
                                if (cx.ballot(live && border) != 0ull) {
                                    if (border) {
                                        t1 = (dmn / (double)D) / sj[k] < thr;
                                        t2 = (dsd / (double)K) / sj[k] < thr;
                                    }
                                }
                                const unsigned long long bm = cx.ballot(live && t1);
                                const unsigned long long bs = cx.ballot(live && t2);
                                cm += xt_popc64(bm & hmask);
                                cs += xt_popc64(bs & hmask);
                            }
                            const bool flag = valid && (same_hist || (cm >= cmin && cs >= cmin));
                            if (x == 0 && flag) cx.atomic_or_u32(&cmat[b * NWD + (ci >> 5)], 1u << (ci & 31));
                        };
                        if (decltype(WRAP)::value) {
                            // wrapped classes: walk the class list (cstep == 1: bit = distance to the pivot - 1)
                            const int c = xt_th_cat_digit(b, S);
                            const int cb = class_count(nE, 0, c), kb = class_count(b, c, c + 1), nc = class_count(nE, c, c + 1);
                            for (int kq = kb + 1; kq < nc; kq += 2) {
                                const bool valid = kq + half < nc;
                                const uint32_t ent = clist[cb + (valid ? kq + half : kb)];
                                const int jj = (int)(ent & 0xffffu), gj = (int)(ent >> 16);
                                test_pair(valid, jj, gj, jj - gj * G, jj - b - 1);
                            }
                        } else {
                            const int dg = (2 * cstep) / G, dr = (2 * cstep) - dg * G;  // (parent, new digits) advance of a candidate per iteration
                            int gj = (b + cstep + half * cstep) / G, rj = (b + cstep + half * cstep) - gj * G, ci = half;
                            for (int j0 = b + cstep; j0 < nE; j0 += 2 * cstep, gj += dg, rj += dr, ci += 2) {
                                if (rj >= G) {
                                    rj -= G;
                                    ++gj;
                                }
                                const int jj = j0 + half * cstep;
                                test_pair(jj < nE, jj, gj, rj, ci);
                            }
                        }
                    };
                    auto compute_row = [&](const int b, auto WRAP) {
                    if (P <= a.pair_lanes_max_p) {
                        // few pilot tracks (predict_Bs with nb_max <= 4): one lane per (pivot, candidate) pair, pilots in a loop
                        {
                            const int gb = b / G, rb = b - gb * G;
                            for (int jj = b + cstep * (1 + cx.lane()); jj < nE; jj += 64 * cstep) {
                                if (!same_class(WRAP, jj, b)) continue;
                                const int gj = jj / G, rj = jj - gj * G;
                                bool same_hist = useA && same_digits(WRAP, jj, b, rj, rb);
                                if (same_hist)
                                    for (int xx = 0; xx < (PREDS ? P : 1); ++xx) same_hist = same_hist && kyA[xx * wsP + gj] == kyA[xx * wsP + gb];
                                bool flag = same_hist;
                                if (!same_hist) {
                                    int cm = 0, cs = 0;
                                    for (int x = 0; x < P; ++x) {
                                        double dmn = 0.0, dsd = 0.0, sj[K];
                                        for (int d = 0; d < D; ++d) dmn += fabs(Mv(d, x, gj) - Mv(d, x, gb));
                                        dmn = dmn / (double)D;
                                        for (int k = 0; k < K; ++k) {
                                            sj[k] = Sv(k, x, jj);
                                            dsd += fabs(sj[k] - Sv(k, x, b));
                                        }
                                        dsd = dsd / (double)K;
                                        for (int k = 0; k < K; ++k) {
                                            cm += xt_div_lt(dmn, sj[k], thr) ? 1 : 0;
                                            cs += xt_div_lt(dsd, sj[k], thr) ? 1 : 0;
                                        }
                                    }
                                    flag = cm >= cmin && cs >= cmin;
                                }
                                const int ci = (jj - b) / cstep - 1;
                                if (flag) cx.atomic_or_u32(&cmat[b * NWD + (ci >> 5)], 1u << (ci & 31));
                            }
                        }
                    } else {
                        if (staged)
                            row_ballot(b, WRAP, std::true_type());
                        else
                            row_ballot(b, WRAP, std::false_type());
                    }
                    };
                    int mpos = 0, ng = 0;  // used by the first wavefront (uniform)
                    for (int i = tid; i < ((nE + 31) >> 5); i += nt) gbits[i] = 0u;
                    if (clisted)
                        for (int j = tid; j < nE; j += nt) {
                            const int c = xt_th_cat_digit(j, S);
                            clist[class_count(nE, 0, c) + class_count(j, c, c + 1)] = (uint32_t)j | ((uint32_t)(j / G) << 16);
                        }
                    cx.sync();
                    XT_TH_TICK(7);
                    const int NWv = cx.waves_per_block();
                    // Rows are computed in batches of one pivot candidate per wavefront, the greedy scan advancing batch by batch: a
                    // sequence that an earlier batch has already put into a group never becomes a pivot, so its row - more than half of
                    // all rows - is never computed.  With few sequences (<= 64) and several wavefronts everything is one batch (the
                    // barriers of more batches would cost more than the skipped rows save); row b then has ~(nE - b) / S candidates and
                    // the rows are dealt in snake order (w, 2 NWv - 1 - w, 2 NWv + w, ...) so that every wavefront gets the same work.
                    const bool one_batch = NWv > 1 && (nE <= 64 || a.plan_bs < 0);
                    const int BS = one_batch ? nE : NWv * (a.plan_bs > 0 ? a.plan_bs : 1);
                    for (int b0 = 0; b0 < nE; b0 += BS) {
                        for (int it = 0, bw = b0 + cx.wave_in_block(); bw < b0 + BS && bw < nE;
                             ++it, bw = b0 + (it >> 1) * 2 * NWv + (one_batch && (it & 1) ? 2 * NWv - 1 - cx.wave_in_block() : cx.wave_in_block() + (it & 1) * NWv))
                            if (!((gbits[bw >> 5] >> (bw & 31)) & 1u)) {
                                if (wrapS)
                                    compute_row(bw, std::true_type());
                                else
                                    compute_row(bw, std::false_type());
                            }
                        cx.sync();
                        XT_TH_TICK(2);
                        if (cx.wave_in_block() == 0) {
                            // The greedy scan, by the first wavefront: rows in order (inherently serial), the candidates of a pivot in
                            // parallel - lane i looks at candidate bit c0 + i, the members are appended in candidate order (ranks from the
                            // ballot).  mpos / ng are wave-uniform.  The grouped flags of the word the row index is in are kept in a
                            // register between pivots, so a row that is already grouped costs no LDS round trip.
                            const int bend = b0 + BS < nE ? b0 + BS : nE;
                            const int lane = cx.lane();
                            const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                            uint32_t gcur = 0u;
                            for (int b = b0; b < bend; ++b) {
                                if ((b & 31) == 0 || b == b0) {
                                    gcur = gbits[b >> 5];
                                    cx.wave_sync();  // (CPU emulation: every lane has its copy before lane 0 touches the word)
                                }
                                if ((gcur >> (b & 31)) & 1u) continue;
                                if (lane == 0) {
                                    gst[ng] = (uint16_t)mpos;
                                    mem[mpos] = (uint16_t)b;  // the pivot itself
                                    gbits[b >> 5] = gcur | (1u << (b & 31));
                                }
                                ++ng;
                                ++mpos;
                                cx.wave_sync();
                                const int nc = (nE - 1 - b) / cstep;  // candidates b + cstep, ..., b + nc cstep
                                for (int c0 = 0; c0 < nc; c0 += 64) {
                                    const int ci = c0 + lane;
                                    const int jj = b + cstep * (ci + 1);
                                    bool take = false;
                                    if (ci < nc) take = ((cmat[b * NWD + (ci >> 5)] >> (ci & 31)) & 1u) && !((gbits[jj >> 5] >> (jj & 31)) & 1u);
                                    const unsigned long long tk = cx.ballot(take);
                                    if (take) {
                                        mem[mpos + xt_popc64(tk & below)] = (uint16_t)jj;
                                        cx.atomic_or_u32(&gbits[jj >> 5], 1u << (jj & 31));
                                    }
                                    mpos += xt_popc64(tk);
                                }
                                cx.wave_sync();
                                gcur = gbits[b >> 5];
                                cx.wave_sync();  // (CPU emulation: every lane has its copy before lane 0 touches the word again)
                            }
                        }
                        cx.sync();
                        XT_TH_TICK(3);
                    }
                    if (tid == 0) {
                        gst[ng] = (uint16_t)mpos;
                        wcnt[0] = ng;
                    }
                    cx.sync();
                    nG = wcnt[0];
                    XT_TH_TICK(3);
                }
                cx.sync();
                if (nG > wsP) {
                    overflow = 1;
                    maxG = nG;
                    break;
                }
                // members as (parent, table offset) words: what the gathers (here and in the apply kernel) consume
                for (int i = tid; i < nE; i += nt) {
                    const int j = mem[i], g = j / G, r = j - g * G;
                    mpk[i] = ((uint32_t)g << 16) | (uint32_t)((int)nwA[g] * G + r);
                }
                sumE += nE;
                cx.sync();
                XT_TH_TICK(4);
                // ---- merge: pilots' states, the shared state history, newest state of each group; publish the plan
                const bool stay = t >= 2 && t >= a.min_len;
                const double* TTl = TAB + (stay ? 1 : 0) * S * G;
                for (int i = tid; i < P * nG; i += nt) {
                    const int x = i / nG, g2 = i - x * nG;
                    xt_th_gather<D, K>(bA, 1, x * wsP, mpk, (int)gst[g2], (int)gst[g2 + 1], TTl, TD2, bB, x * wsP + g2, dtf(x, t));
                    if (RF && a.rf_out && cx.block() == 0 && a.rf_row0 == 0 && g2 < a.rf_cap) {
                        double mv[D];
                        for (int d = 0; d < D; ++d) mv[d] = bB.m(d, x * wsP + g2);
                        rf_put(t - 1, c0 + x, g2, bB.zm(x * wsP + g2), bB.ze(x * wsP + g2), mv, bB.u(0, x * wsP + g2));
                    }
                }
                if (RF && a.rf_out && cx.block() == 0) {
                    for (int i = tid; i < nG && i < a.rf_cap; i += nt) a.rf_new[(int64_t)(t - 1) * a.rf_cap + i] = (uint8_t)((int)mem[gst[i]] % S);
                    if (tid == 0) a.rf_cnt[t - 1] = nG;
                }
                // fit mode keeps frame_len history entries (tracking.py:699-701); when predicting the reference keeps all of
                // them, but only the first frame_len are ever looked at before the final read-out, which is done here by a
                // backward pass over the stored merge weights instead of carrying L entries per sequence through every merge
                const int Hn = (t == 1) ? He : (He < F ? He : F);
                const int Pc = nfuse == 0 ? 1 : P;                          // rows of the reference's cat array (tracking.py:726-729)
                if (PREDS) {
                    // per-track histories, merged with the tracks' own softmax weights (tracking.py:731-736)
                    for (int i = tid; i < P * nG; i += nt) {
                        const int x = i / nG, g2 = i - x * nG;
                        const int k0 = gst[g2], k1 = gst[g2 + 1];
                        int E = XT_EMIN;
                        for (int kk = k0; kk < k1; ++kk) {
                            const int e = bA.ze(x * wsP + (int)(mpk[kk] >> 16));
                            E = e > E ? e : E;
                        }
                        double W = 0.0;
                        for (int kk = k0; kk < k1; ++kk) {
                            const int idx = x * wsP + (int)(mpk[kk] >> 16);
                            const double av = xt_ldexp(bA.zm(idx) * TTl[mpk[kk] & 0xffffu], bA.ze(idx) - E);
                            wgt[x * wsE + kk] = av;
                            W += av;
                        }
                        for (int kk = k0; kk < k1; ++kk) {
                            const double wn = (k1 - k0 == 1) ? 1.0 : wgt[x * wsE + kk] / W;
                            wgt[x * wsE + kk] = wn;
                            hw[((int64_t)x * L + t) * wsE + kk] = wn;
                        }
                    }
                    for (int i = tid; i < nE; i += nt) {
                        const int jj = mem[i], g = jj / G;
                        // history entry, not the state (the initial sequences, t == 1, come without the int8 index)
                        hmem[(int64_t)t * wsE + i] = ((uint32_t)g << 16) | (uint32_t)(t == 1 ? jj % S : xt_th_cat_digit(jj, S));
                        hmpk[(int64_t)t * wsE + i] = mpk[i];
                    }
                    for (int i = tid; i <= nG; i += nt) hgst[(int64_t)t * (wsE + 1) + i] = gst[i];
                    if (tid == 0) {
                        hn[t * 2] = nE;
                        hn[t * 2 + 1] = nG;
                    }
                    cx.sync();
                    for (int i = tid; i < P * nG * Hn * S; i += nt) {
                        const int x = i / (nG * Hn * S), q = i - x * (nG * Hn * S);
                        const int g2 = q / (Hn * S), hs = q - g2 * (Hn * S), h = hs / S, s2 = hs - h * S;
                        const int k0 = gst[g2], k1 = gst[g2 + 1];
                        auto val = [&](int jj) -> double {
                            const int g = jj / G;
                            if (h < NS) return ((t == 1 ? (jj / pwS[h]) % S : xt_th_cat_digit(jj / pwS[h], S)) == s2) ? 1.0 : 0.0;
                            return ctA[x * cstride + (g * HM + (h - NS)) * S + s2];
                        };
                        double o;
                        if (k1 - k0 == 1) {
                            o = val(mem[k0]);
                        } else {
                            o = 0.0;
                            for (int kk = k0; kk < k1; ++kk) o = xt_fma(wgt[x * wsE + kk], val(mem[kk]), o);
                        }
                        ctB[x * cstride + (g2 * HM + h) * S + s2] = o;
                    }
                }
                for (int i = tid; i < (PREDS ? 0 : nG * Hn * S); i += nt) {
                    const int g2 = i / (Hn * S), hs = i - g2 * (Hn * S), h = hs / S, s = hs - h * S;
                    const int k0 = gst[g2], k1 = gst[g2 + 1];
                    auto val = [&](int j) -> double {
                        const int g = j / G;
                        if (h < NS) return ((t == 1 ? (j / pwS[h]) % S : xt_th_cat_digit(j / pwS[h], S)) == s) ? 1.0 : 0.0;
                        return ctA[(g * HM + (h - NS)) * S + s];
                    };
                    double o;
                    if (k1 - k0 == 1) {
                        o = val(mem[k0]);
                    } else {
                        // np.mean over (pilot rows x members) of identical rows: sequential, member-major (see tests)
                        double acc = 0.0;
                        for (int kk = k0; kk < k1; ++kk) {
                            const double v = val(mem[kk]);
                            for (int rep = 0; rep < Pc; ++rep) acc = acc + v;
                        }
                        o = acc / (double)(Pc * (k1 - k0));
                    }
                    ctB[(g2 * HM + h) * S + s] = o;
                }
                for (int i = tid; i < nG; i += nt) {
                    const uint8_t nw = (uint8_t)((int)mem[gst[i]] % S);
                    nwB[i] = nw;
                    if (!PREDS) gnew_g[(int64_t)t * capE + i] = nw;
                }
                if (!PREDS) {
                    for (int i = tid; i < nE; i += nt) {
                        mem_g[(int64_t)t * capE + i] = mem[i];
                        mpk_g[(int64_t)t * capE + i] = mpk[i];
                    }
                    for (int i = tid; i <= nG; i += nt) gst_g[(int64_t)t * (capE + 1) + i] = gst[i];
                }
                cx.sync();
                XT_TH_TICK(5);
                // history keys of the new parents: argmax over states of the first Fk entries
                for (int i = tid; i < (PREDS ? P : 1) * nG; i += nt) {
                    const int x = i / nG, g2 = i - x * nG;
                    const double* cb = ctB + x * cstride + (int64_t)g2 * HM * S;
                    unsigned long long key = 0;
                    for (int h = 0; h < Fk && h < Hn; ++h) {
                        int best = 0;
                        double bv = cb[h * S];
                        for (int s = 1; s < S; ++s) {
                            const double v = cb[h * S + s];
                            if (v > bv) {
                                bv = v;
                                best = s;
                            }
                        }
                        key |= (unsigned long long)best << (3 * h);
                    }
                    kyB[x * wsP + g2] = key;
                }
                if (t >= 2) ++nfuse;
                {
                    View tb = bA;
                    bA = bB;
                    bB = tb;
                    uint8_t* tn = nwA;
                    nwA = nwB;
                    nwB = tn;
                    double* tc = ctA;
                    ctA = ctB;
                    ctB = tc;
                    unsigned long long* tk = kyA;
                    kyA = kyB;
                    kyB = tk;
                }
                nPar = nG;
                Hc = Hn;
                maxG = nG > maxG ? nG : maxG;
                cx.sync();
                XT_TH_TICK(6);
            }
            if (tid == 0 && !PREDS) {
                hdr_g[t * 2] = nE;
                hdr_g[t * 2 + 1] = nG;
            }
        }
        // tracks xb .. xb + cnt - 1 of the chunk sit in slots 0 .. cnt - 1 of the state `vA` (nParF parents)
        // last entry of the refinement record: the expanded, unfused sequences (parent g, new state r) after the last integration
        auto rf_last = [&](const int xb, const int cnt, const View& vA, const int nParF, const bool shared) {
            cx.sync();
            const int tl = L - 1, nE = nParF * G;
            for (int i = tid; i < cnt * nE; i += nt) {
                const int x = i / nE, jj = i - x * nE, g = jj / G, r = jj - g * G, idx = x * wsP + g, o = (int)nwA[g] * G + r;
                if (jj >= a.rf_cap || (shared && a.rf_row0 != 0)) continue;  // the pilots' rows belong to the first row block
                double mv[D];
                for (int d = 0; d < D; ++d) mv[d] = vA.m(d, idx);
                rf_put(tl - 1, c0 + xb + x, jj, vA.zm(idx) * TAB[o], vA.ze(idx), mv, xt_fma(TD2[o], dtf(xb + x, tl), vA.u(0, idx)));
            }
            if (shared) {
                for (int i = tid; i < nE && i < a.rf_cap; i += nt) a.rf_new[(int64_t)(tl - 1) * a.rf_cap + i] = (uint8_t)(i % G % S);
                if (tid == 0) a.rf_cnt[tl - 1] = nE;
            }
            cx.sync();
        };
        auto finish = [&](const int xb, const int cnt, const View& vA, const int nParF) {
            // ---- posteriors (tracking.py:611-648): weights of the final sequences (parent g, new state r) at the last
            // position, then the weighted mean of their state histories; history index 0 = last position
            cx.sync();
            const int tl = L - 1, nE = nParF * G;
            const bool stay = tl >= 2 && tl >= a.min_len;
            const double* TF = TAB + ((bk.isBL ? 2 : 0) + (stay ? 1 : 0)) * S * G;
            for (int i = tid; i < cnt * nE; i += nt) {
                const int x = i / nE, jj = i - x * nE, g = jj / G, r = jj - g * G, idx = x * wsP + g, o = (int)nwA[g] * G + r;
                double cl[D], l2l[K], dq[D], dsq = 0.0;
                load_l2(xb + x, tl, l2l);
                for (int d = 0; d < D; ++d) {
                    cl[d] = bk.tracks[((c0 + xb + x) * L + tl) * D + d];
                    dq[d] = cl[d] - vA.m(d, idx);
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                double quad, gf;
                if (K == 1) {
                    const double rr = xt_rcp(xt_fma(TD2[o], dtf(xb + x, tl), vA.u(0, idx)) + l2l[0]);
                    quad = 0.5 * dsq * rr;
                    gf = xt_pow_half<D>(rr);
                } else {
                    quad = 0.0;
                    gf = 1.0;
                    for (int d = 0; d < D; ++d) {
                        const double rr = xt_rcp(xt_fma(TD2[o], dtf(xb + x, tl), vA.u(d, idx)) + l2l[d]);
                        quad = xt_fma(0.5 * dq[d] * dq[d], rr, quad);
                        gf *= rr;
                    }
                    gf = sqrt(gf);
                }
                double p;
                int j6, n2;
                xt_exp_tab(-quad, p, j6, n2);
                const double wm = vA.zm(idx) * TF[o] * (gf * T64[j6]) * p;
                wgt[x * wsE + jj] = wm;
                wexp[x * wsE + jj] = (wm != 0.0) ? vA.ze(idx) + n2 : XT_EMIN;
            }
            cx.sync();
            for (int x = tid; x < cnt; x += nt) {  // per track: common exponent, normalisation
                int E = XT_EMIN;
                for (int jj = 0; jj < nE; ++jj) E = wexp[x * wsE + jj] > E ? wexp[x * wsE + jj] : E;
                double tot = 0.0;
                for (int jj = 0; jj < nE; ++jj) {
                    const double v = xt_ldexp(wgt[x * wsE + jj], wexp[x * wsE + jj] - E);
                    wgt[x * wsE + jj] = v;
                    tot += v;
                }
                pm[x] = tot;
            }
            cx.sync();
            // backward pass (one thread per track, fixed summation order): the mass a final sequence carries flows back
            // through the merge tree; what passes through a member with new state r at merge step t is the posterior mass
            // of state r at position t
            for (int x = tid; x < cnt; x += nt) {
                double* post = bk.preds_out + (c0 + xb + x) * (int64_t)L * S;
                for (int i = 0; i < L * S; ++i) post[i] = 0.0;
                double* bc = beta + (int64_t)x * 2 * wsP;
                double* bp = bc + wsP;
                const double rt = 1.0 / pm[x];
                for (int g = 0; g < nParF; ++g) bc[g] = 0.0;
                for (int jj = 0; jj < nE; ++jj) {
                    const int g = jj / G;
                    const double om = wgt[x * wsE + jj] * rt;
                    post[(int64_t)(L - 1) * S + (L == 2 ? jj % S : xt_th_cat_digit(jj, S))] += om;
                    bc[g] += om;
                }
                for (int t = L - 2; t >= 1; --t) {
                    const int nGt = hn[t * 2 + 1], nPt = t > 1 ? hn[(t - 1) * 2 + 1] : S;
                    for (int g = 0; g < nPt; ++g) bp[g] = 0.0;
                    for (int g2 = 0; g2 < nGt; ++g2) {
                        const double bg = bc[g2];
                        const int k0 = hgst[(int64_t)t * (wsE + 1) + g2], k1 = hgst[(int64_t)t * (wsE + 1) + g2 + 1];
                        for (int kk = k0; kk < k1; ++kk) {
                            const uint32_t pk = hmem[(int64_t)t * wsE + kk];
                            const double cc = bg * hw[((int64_t)x * L + t) * wsE + kk];
                            post[(int64_t)t * S + (int)(pk & 0xffffu)] += cc;
                            bp[pk >> 16] += cc;
                        }
                    }
                    double* tb = bc;
                    bc = bp;
                    bp = tb;
                }
                for (int s2 = 0; s2 < S; ++s2) post[s2] = bc[s2];
            }
            cx.sync();
        };
        if (PREDS && !overflow) {
            if (!RF)
                finish(0, P, bA, nPar);
            else if (a.rf_out && cx.block() == 0)
                rf_last(0, P, bA, nPar, true);
            // ---- tracks beyond the pilots (predict_Bs with nb_max > 30, tracking.py:856-868): they take no part in the merge
            // decisions (fuse_tracks_th looks at the first 30 tracks only, tracking.py:676-691) but are merged with THEIR OWN
            // weights (tracking.py:703-741): replay the recorded plan batch by batch in the pilots' slots
            const int64_t rf_first = RF ? (a.rf_row0 > P ? a.rf_row0 : (int64_t)P) : 0;
            const int nrep = RF ? (int)((a.rf_row0 + a.rf_rows) < n ? (a.rf_row0 + a.rf_rows) : n) : n;  // replay bound of this launch
            for (int xb = RF ? (int)rf_first + PC * cx.block() : P; xb < nrep && (!RF || a.rf_out); xb += RF ? PC * cx.nblocks() : PC) {
                const int cnt = (nrep - xb) < PC ? (nrep - xb) : PC;
                cx.sync();
                for (int i = tid; i < cnt * S; i += nt) {
                    const int x = i / S, s2 = i - x * S, idx = x * wsP + s2;
                    double l2[K];
                    load_l2(xb + x, 0, l2);
                    A.zm(idx) = hdr[8 + s2];
                    A.ze(idx) = 0;
                    for (int d = 0; d < D; ++d) A.m(d, idx) = bk.tracks[((c0 + xb + x) * L + 0) * D + d];
                    for (int k = 0; k < K; ++k) A.u(k, idx) = l2[k];
                }
                View fA = A, fB = B;
                int fPar = S;
                cx.sync();
                for (int t = 1; t <= L - 1; ++t) {
                    if (t >= 2) {
                        const int pos = t - 1;
                        for (int i = tid; i < cnt * fPar; i += nt) {
                            const int x = i / fPar, g = i - x * fPar;
                            double c[D], l2[K];
                            for (int d = 0; d < D; ++d) c[d] = bk.tracks[((c0 + xb + x) * L + pos) * D + d];
                            load_l2(xb + x, pos, l2);
                            xt_th_integrate<D, K>(fA, x * wsP + g, c, l2, T64);
                        }
                        cx.sync();
                    }
                    if (t < L - 1) {
                        const int nEt = hn[t * 2], nGt = hn[t * 2 + 1];
                        for (int i = tid; i < nEt; i += nt) mpk[i] = hmpk[(int64_t)t * wsE + i];
                        for (int i = tid; i <= nGt; i += nt) gst[i] = hgst[(int64_t)t * (wsE + 1) + i];
                        cx.sync();
                        const bool stay = t >= 2 && t >= a.min_len;
                        const double* TTl = TAB + (stay ? 1 : 0) * S * G;
                        for (int i = tid; i < cnt * nGt; i += nt) {
                            const int x = i / nGt, g2 = i - x * nGt;
                            const int k0 = gst[g2], k1 = gst[g2 + 1];
                            xt_th_gather<D, K>(fA, 1, x * wsP, mpk, k0, k1, TTl, TD2, fB, x * wsP + g2, dtf(xb + x, t));
                            int E = XT_EMIN;
                            for (int kk = k0; kk < k1; ++kk) {
                                const int e = fA.ze(x * wsP + (int)(mpk[kk] >> 16));
                                E = e > E ? e : E;
                            }
                            double W = 0.0;
                            for (int kk = k0; kk < k1; ++kk) {
                                const int idx = x * wsP + (int)(mpk[kk] >> 16);
                                const double av = xt_ldexp(fA.zm(idx) * TTl[mpk[kk] & 0xffffu], fA.ze(idx) - E);
                                hw[((int64_t)x * L + t) * wsE + kk] = av;
                                W += av;
                            }
                            for (int kk = k0; kk < k1; ++kk)
                                hw[((int64_t)x * L + t) * wsE + kk] = (k1 - k0 == 1) ? 1.0 : hw[((int64_t)x * L + t) * wsE + kk] / W;
                            if (RF && g2 < a.rf_cap) {
                                double mv[D];
                                for (int d = 0; d < D; ++d) mv[d] = fB.m(d, x * wsP + g2);
                                rf_put(t - 1, c0 + xb + x, g2, fB.zm(x * wsP + g2), fB.ze(x * wsP + g2), mv, fB.u(0, x * wsP + g2));
                            }
                        }
                        cx.sync();
                        View tb = fA;
                        fA = fB;
                        fB = tb;
                        fPar = nGt;
                    }
                }
                if (!RF)
                    finish(xb, cnt, fA, fPar);
                else
                    rf_last(xb, cnt, fA, fPar, false);
            }
        }
        XT_TH_PROF_DUMP(PREDS ? "posteriors" : "fit");
        if (tid == 0 && (!RF || cx.block() == 0)) {
            bk.status[ch * 4 + 0] = overflow;
            bk.status[ch * 4 + 1] = maxE;
            bk.status[ch * 4 + 2] = maxG;
            bk.status[ch * 4 + 3] = sumE;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// APPLY kernel body: a workgroup serves tiles of TT tracks of ONE chunk.
//   UNI = true : TT == 64, a wavefront = the 64 tracks of the tile for ONE parent / group at a time, so every plan and
//                table index is wave-uniform: the plan (global memory) and the model tables are read with scalar loads,
//                the vector unit only touches the per-track state records in LDS.
//                SINGLE = true: ONE state buffer (half the LDS, for models with more live sequences): every wavefront merges
//                its (at most XT_TH_GPW) groups into registers, a barrier, then the integrated results overwrite the buffer.
//   UNI = false: TT < 64 (many live sequences: fewer tracks fit the LDS); a wavefront spans several groups, the chunk's
//                plan is staged in LDS (all merged steps when they fit, else step by step).
// ------------------------------------------------------------------------------------------------------------------
// DT = true: per-track time steps (XtThBucket::dt) - a separate instantiation, so that the fixed-dt kernels keep their register
// budget (the two extra VGPRs of the time step cost the wave-uniform variant one wave per SIMD, i.e. one of its two workgroups per CU).
// SEQ = true: also writes the per-sequence matrix of the last position (XtThBucket::seq_out, extrack_sequence_matrix_th) - a separate
// instantiation of the general variant only: the log() of that branch, compiled into every variant, cost the wave-uniform kernel 10 VGPRs
// and one wave per SIMD (2 states x 30: apply kernel 2.08 -> 3.25 ms, measured round 4) although the branch is never taken in a fit.
template <int D, int K, bool UNI, bool SINGLE, bool DT, bool SEQ = false, class Ctx>
XT_HD void xt_th_apply_body(const XtThArgs& a, Ctx& cx)
{
    typedef XtThView<D, K, true> View;
    const int S = a.S, G = a.G, capE = a.capE, TT = UNI ? 64 : a.TT, capG = a.capG, KS = a.KS;
    const int Lmax = a.buckets ? a.Lmax : a.L;
    const int gch = cx.block() / a.bpc, sub = cx.block() - gch * a.bpc;
    int ch;  // chunk index inside its bucket
    const XtThBucket bk = xt_th_bind(a, gch, ch);
    const int L = bk.L;
    const int logTT = UNI ? 6 : a.logTT;
    const int tid = cx.tid(), nt = cx.nthreads();
    const int TP = TT + 1;  // padded row of the position stage
    double* smem = cx.smem();
    const int ntab = xt_tab_doubles(S, G);
    const double* blob_c = a.blob + (int64_t)gch * a.blob_stride;  // the chunk's tables (the same for all chunks unless dt is per track)
    for (int i = tid; i < ntab; i += nt) smem[i] = blob_c[i];
    const double* hdr = smem;
    const double* TABl = smem + XT_BLOB_HDR;
    const double* T64 = TABl + XT_NTAB * S * G;
    // expansion tables: scalar loads straight from the blob when the index is wave-uniform, LDS otherwise
    typedef XtCPtr<UNI, double> CD;
    typedef XtCPtr<UNI, uint32_t> CU32;
    typedef XtCPtr<UNI, uint16_t> CU16;
    typedef XtCPtr<UNI, uint8_t> CU8;
    typedef XtCPtr<UNI, int32_t> CI32;
    const typename CD::type TAB = CD::make(UNI ? blob_c + XT_BLOB_HDR : TABl);
    const typename CD::type TD2 = TAB + 4 * S * G;
    double* w = smem + ((ntab + 1) & ~1);
    const int plane = capG * TT;
    const int capEl = capG * G;
    View bA, bB;
    xt_th_carve(w, bA, plane, D, K);
    if (SINGLE)
        bB = bA;  // the tail then parks its per-parent terms in the records it has just consumed
    else
        xt_th_carve(w, bB, plane, D, K);
    double* spos = w;
    w += XT_TH_STAGE * D * TP;
    double* ssig = w;
    w += XT_TH_STAGE * (a.locerr_mode ? KS : 0) * TP;  // no sigma stage with a global localisation error
    double* red = w;
    w += TT;
    const bool resident = !UNI && a.plan_cap > 0;
    const bool direct = !UNI && a.plan_cap < 0;  // member lists read from global memory (no LDS copy)
    const int pmcap = (UNI || direct) ? 0 : (resident ? a.plan_cap : capEl);
    const int pgcap = (UNI || direct) ? 0 : (resident ? a.plan_cap + Lmax : capEl + 1);
    int* nanflag = (int*)w;
    int* pstep = nanflag + TT;               // [L][4]: member offset, gstart offset, nE, nG
    uint32_t* pmem = (uint32_t*)(pstep + 4 * Lmax);
    uint16_t* pgst = (uint16_t*)(pmem + pmcap);
    uint8_t* nfin = (uint8_t*)(pgst + pgcap + (pgcap & 1));  // newest state of the parents the last position sees

    const int64_t c0 = (int64_t)ch * a.chunk;
    const int n = (int)((bk.N - c0) < a.chunk ? (bk.N - c0) : a.chunk);
    const int ntile = (n + TT - 1) >> logTT;
    const uint32_t* mpk_g = bk.mpack + (int64_t)ch * L * capE;
    const uint16_t* gst_g = bk.gstart + (int64_t)ch * L * (capE + 1);
    const uint8_t* gnew_g = bk.gnew + (int64_t)ch * L * capE;
    const int32_t* hdr_g = bk.hdr + (int64_t)ch * L * 2;
    const typename CI32::type hdr_u = CI32::make(hdr_g);
    const typename CU8::type gnew_u = CU8::make(gnew_g);
    const int x = tid & (TT - 1);  // TT is a power of two <= nthreads
    const int g0 = UNI ? cx.uniform(tid >> 6) : (tid >> logTT);
    const int gstep = nt >> logTT;
    double my_ll = 0.0;  // meaningful in threads with g0 == 0

    // ---- the chunk's plan -> LDS (all merged steps, or only the step table when it is streamed / read by scalar loads)
    if (tid == 0) {
        int om = 0, og = 0;
        for (int t = 1; t <= L - 2; ++t) {
            const int nE = hdr_g[t * 2], nG = hdr_g[t * 2 + 1];
            pstep[t * 4 + 0] = resident ? om : 0;
            pstep[t * 4 + 1] = resident ? og : 0;
            pstep[t * 4 + 2] = nE;
            pstep[t * 4 + 3] = nG;
            om += nE;
            og += nG + 1;
        }
    }
    if (!UNI) {
        if (L >= 3) {
            const int nGl = hdr_g[(L - 2) * 2 + 1];
            for (int i = tid; i < nGl; i += nt) nfin[i] = gnew_g[(int64_t)(L - 2) * capE + i];
        } else {
            for (int i = tid; i < S; i += nt) nfin[i] = (uint8_t)i;
        }
    }
    cx.sync();
    if (resident)
        for (int t = 1; t <= L - 2; ++t) {
            const int om = pstep[t * 4], og = pstep[t * 4 + 1], nE = pstep[t * 4 + 2], nG = pstep[t * 4 + 3];
            for (int i = tid; i < nE; i += nt) pmem[om + i] = mpk_g[(int64_t)t * capE + i];
            for (int i = tid; i <= nG; i += nt) pgst[og + i] = gst_g[(int64_t)t * (capE + 1) + i];
        }

    for (int tile = sub; tile < ntile; tile += a.bpc) {
        const int64_t first = c0 + ((int64_t)tile << logTT);
        const int nx = (int)((c0 + n - first) < TT ? (c0 + n - first) : TT);
        const bool act = x < nx;
        cx.sync();  // tables + plan loaded / previous tile's reads done
        if (tid < TT) nanflag[tid] = 0;
        cx.sync();

        auto stage = [&](int p0) {
            const int np = (L - p0) < XT_TH_STAGE ? (L - p0) : XT_TH_STAGE;  // positions to stage
            for (int i = tid; i < TT * XT_TH_STAGE * D; i += nt) {
                const int xx = i / (XT_TH_STAGE * D), o = i - xx * (XT_TH_STAGE * D);
                if (xx < nx && o < np * D) {
                    const double v = bk.tracks[((first + xx) * L + p0) * D + o];
                    spos[o * TP + xx] = v;
                    if (v != v) nanflag[xx] = 1;  // NaN input: the track's result becomes NaN, as in the reference
                }
            }
            if (a.locerr_mode != 0)
                for (int i = tid; i < TT * XT_TH_STAGE * KS; i += nt) {
                    const int xx = i / (XT_TH_STAGE * KS), o = i - xx * (XT_TH_STAGE * KS);
                    if (xx < nx && o < np * KS) {
                        const double v = bk.sigma[((first + xx) * L + p0) * KS + o];
                        ssig[o * TP + xx] = v;
                        if (v != v) nanflag[xx] = 1;
                    }
                }
            cx.sync();
        };
        auto load_l2 = [&](int pos, double* l2) {
            if (a.locerr_mode == 0) {
                for (int k = 0; k < K; ++k) l2[k] = hdr[k];
            } else {
                for (int k = 0; k < K; ++k)
                    l2[k] = xt_th_l2_from_sigma(ssig[((pos & (XT_TH_STAGE - 1)) * KS + (KS == 1 ? 0 : k)) * TP + x], a.locerr_mode, hdr);
            }
        };

        // this track's time step scaling the diffusion term added at step t (1 with a fixed dt: the tables then contain it)
        auto dt_at = [&](int t) -> double { return (DT && act) ? bk.dt[(first + x) * L + (L - t)] : 1.0; };
        stage(0);
        // ---- position 0: S parents
        if (act) {
            double l2[K];
            load_l2(0, l2);
            for (int g = g0; g < S; g += gstep) {
                const int idx = g * TT + x;
                bA.zm(idx) = hdr[8 + g];
                bA.ze(idx) = 0;
                for (int d = 0; d < D; ++d) bA.m(d, idx) = spos[d * TP + x];
                for (int k = 0; k < K; ++k) bA.u(k, idx) = l2[k];
            }
        }
        int nPar = S;
        View cur = bA, nxt = bB;
        cx.sync();

        if (UNI) {
            // merge step t fused with the integration of position t: the merged sequence never leaves the registers, one
            // workgroup barrier per position
            for (int t = 1; t <= L - 2; ++t) {
                if ((t & (XT_TH_STAGE - 1)) == 0) stage(t);
                const int nG = hdr_u[t * 2 + 1];
                const typename CU32::type mem = CU32::make(mpk_g + (int64_t)t * capE);
                const typename CU16::type gst = CU16::make(gst_g + (int64_t)t * (capE + 1));
                const bool stay = t >= 2 && t >= a.min_len;
                const double* TTl = TABl + (stay ? 1 : 0) * S * G;
                double c[D], l2[K];
                for (int d = 0; d < D; ++d) c[d] = spos[((t & (XT_TH_STAGE - 1)) * D + d) * TP + x];
                load_l2(t, l2);
                const double dtv = dt_at(t);
                if (SINGLE) {
                    // half the LDS: every wavefront first merges its (at most XT_TH_GPW) groups into registers, and the
                    // integrated results overwrite the buffer only after all wavefronts have finished reading it
                    double Wq[XT_TH_GPW], Mq[XT_TH_GPW][D], Uq[XT_TH_GPW][K];
                    int Eq[XT_TH_GPW];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (int q = 0; q < XT_TH_GPW; ++q) {
                        const int g2 = g0 + q * gstep;
                        if (g2 < nG)
                            xt_th_gather_regs<D, K>(cur, TT, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TABl + 4 * S * G, Wq[q], Eq[q], Mq[q],
                                                    Uq[q], dtv);
                    }
                    cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (int q = 0; q < XT_TH_GPW; ++q) {
                        const int g2 = g0 + q * gstep;
                        if (g2 < nG) xt_th_integrate_store<D, K>(Wq[q], Eq[q], Mq[q], Uq[q], c, l2, T64, cur, g2 * TT + x);
                    }
                    cx.sync();
                    nPar = nG;
                    continue;
                }
                for (int g2 = g0; g2 < nG; g2 += gstep) {
                    double W, M[D], U[K];
                    int E;
                    xt_th_gather_regs<D, K>(cur, TT, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TABl + 4 * S * G, W, E, M, U, dtv);
                    xt_th_integrate_store<D, K>(W, E, M, U, c, l2, T64, nxt, g2 * TT + x);
                }
                cx.sync();
                View tb = cur;
                cur = nxt;
                nxt = tb;
                nPar = nG;
            }
        } else if (SINGLE) {
            // many live sequences (more than 64): one state buffer, TT < 64 tracks per tile so that it fits the LDS; a thread
            // merges its (at most XT_TH_GPW) groups into registers, barrier, integrates the next position and overwrites the
            // buffer.  With TT = 32 a wavefront spans only two groups, so the member loops of a wavefront rarely differ much.
            for (int t = 1; t <= L - 2; ++t) {
                if ((t & (XT_TH_STAGE - 1)) == 0) stage(t);
                const int nE = pstep[t * 4 + 2], nG = pstep[t * 4 + 3];
                const uint32_t* mem = direct ? mpk_g + (int64_t)t * capE : pmem + pstep[t * 4];
                const uint16_t* gst = direct ? gst_g + (int64_t)t * (capE + 1) : pgst + pstep[t * 4 + 1];
                if (!resident && !direct) {
                    for (int i = tid; i < nE; i += nt) pmem[i] = mpk_g[(int64_t)t * capE + i];
                    for (int i = tid; i <= nG; i += nt) pgst[i] = gst_g[(int64_t)t * (capE + 1) + i];
                    cx.sync();
                }
                const bool stay = t >= 2 && t >= a.min_len;
                const double* TTl = TABl + (stay ? 1 : 0) * S * G;
                double c[D], l2[K];
                for (int d = 0; d < D; ++d) c[d] = spos[((t & (XT_TH_STAGE - 1)) * D + d) * TP + x];
                load_l2(t, l2);
                const double dtv = dt_at(t);
                double Wq[XT_TH_GPW], Mq[XT_TH_GPW][D], Uq[XT_TH_GPW][K];
                int Eq[XT_TH_GPW];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int q = 0; q < XT_TH_GPW; ++q) {
                    const int g2 = g0 + q * gstep;
                    if (act && g2 < nG)
                        xt_th_gather_regs<D, K>(cur, TT, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TABl + 4 * S * G, Wq[q], Eq[q], Mq[q], Uq[q], dtv);
                }
                cx.sync();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int q = 0; q < XT_TH_GPW; ++q) {
                    const int g2 = g0 + q * gstep;
                    if (act && g2 < nG) xt_th_integrate_store<D, K>(Wq[q], Eq[q], Mq[q], Uq[q], c, l2, T64, cur, g2 * TT + x);
                }
                cx.sync();
                nPar = nG;
            }
        } else {
        for (int t = 1; t <= L - 1; ++t) {
            if (t >= 2) {
                const int pos = t - 1;
                if ((pos & (XT_TH_STAGE - 1)) == 0) stage(pos);
                if (act) {
                    double c[D], l2[K];
                    for (int d = 0; d < D; ++d) c[d] = spos[((pos & (XT_TH_STAGE - 1)) * D + d) * TP + x];
                    load_l2(pos, l2);
                    for (int g = g0; g < nPar; g += gstep) xt_th_integrate<D, K>(cur, g * TT + x, c, l2, T64);
                }
            }
            if (t < L - 1) {
                const int nE = UNI ? hdr_u[t * 2] : pstep[t * 4 + 2];
                const int nG = UNI ? hdr_u[t * 2 + 1] : pstep[t * 4 + 3];
                const typename CU32::type mem = CU32::make((UNI || direct) ? mpk_g + (int64_t)t * capE : pmem + pstep[t * 4]);
                const typename CU16::type gst = CU16::make((UNI || direct) ? gst_g + (int64_t)t * (capE + 1) : pgst + pstep[t * 4 + 1]);
                if (!UNI && !resident && !direct) {
                    for (int i = tid; i < nE; i += nt) pmem[i] = mpk_g[(int64_t)t * capE + i];
                    for (int i = tid; i <= nG; i += nt) pgst[i] = gst_g[(int64_t)t * (capE + 1) + i];
                }
                cx.sync();
                const bool stay = t >= 2 && t >= a.min_len;
                const typename CD::type TTl = TAB + (stay ? 1 : 0) * S * G;
                const double dtv = dt_at(t);
                if (act || UNI)  // UNI: keep the control flow wave-uniform (inactive lanes work on slot garbage, never stored out)
                    for (int g2 = g0; g2 < nG; g2 += gstep)
                        xt_th_gather<D, K>(cur, TT, x, mem, (int)gst[g2], (int)gst[g2 + 1], TTl, TD2, nxt, g2 * TT + x, dtv);
                cx.sync();
                View tb = cur;
                cur = nxt;
                nxt = tb;
                nPar = nG;
                (void)nE;
            }
        }
        cx.sync();
        }

        // ---- last position (+ leaving/bleaching term, tracking.py:611-633): reduction over (parent, new digits)
        const int tl = L - 1;
        if ((tl & (XT_TH_STAGE - 1)) == 0) stage(tl);
        if (act) {
            const bool stay = tl >= 2 && tl >= a.min_len;
            const typename CD::type TF = TAB + ((bk.isBL ? 2 : 0) + (stay ? 1 : 0)) * S * G;
            double cl[D], l2l[K];
            for (int d = 0; d < D; ++d) cl[d] = spos[((tl & (XT_TH_STAGE - 1)) * D + d) * TP + x];
            load_l2(tl, l2l);
            const double dtl = dt_at(tl);
            for (int g = g0; g < nPar; g += gstep) {
                const int idx = g * TT + x;
                XtAcc acc;
                acc.clear();
                const double zq = cur.zm(idx);
                const int eq = cur.ze(idx);
                const int nw = UNI ? (L >= 3 ? (int)gnew_u[(int64_t)(L - 2) * capE + g] : g) : (int)nfin[g];
                const int o = nw * G;
                double dq[D], uq[K], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - cur.m(d, idx);
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                for (int k = 0; k < K; ++k) uq[k] = cur.u(k, idx);
                for (int r = 0; r < G; ++r) {
                    double quad, gf;
                    if (K == 1) {
                        const double rr = xt_rcp(xt_fma(TD2[o + r], dtl, uq[0]) + l2l[0]);
                        quad = 0.5 * dsq * rr;
                        gf = xt_pow_half<D>(rr);
                    } else {
                        quad = 0.0;
                        gf = 1.0;
                        for (int d = 0; d < D; ++d) {
                            const double rr = xt_rcp(xt_fma(TD2[o + r], dtl, uq[d]) + l2l[d]);
                            quad = xt_fma(0.5 * dq[d] * dq[d], rr, quad);
                            gf *= rr;
                        }
                        gf = sqrt(gf);
                    }
                    double p;
                    int j, n2;
                    xt_exp_tab(-quad, p, j, n2);
                    acc.add(zq * TF[o + r] * (gf * T64[j]) * p, eq + n2);
                    if (SEQ && bk.seq_out) {  // the reference's per-sequence matrix (tracking.py:632-650), before its leaving term: the caller expands that
                        const double wq = zq * TAB[(stay ? 1 : 0) * S * G + o + r] * (gf * T64[j]) * p;
                        bk.seq_out[(first + x) * (int64_t)bk.seq_stride + g * G + r] =
                            nanflag[x] ? NAN : (wq > 0.0 ? log(wq) + (double)(eq + n2) * XT_LN2 + bk.ll_const : (wq == 0.0 ? -INFINITY : NAN));
                    }
                }
                nxt.zm(idx) = acc.m;
                nxt.ze(idx) = acc.e;
            }
        }
        cx.sync();
        if (act && g0 == 0) {
            XtAcc tot;
            tot.clear();
            for (int g = 0; g < nPar; ++g) tot.add(nxt.zm(g * TT + x), nxt.ze(g * TT + x));
            const double ll = nanflag[x] ? NAN : log(tot.m) + (double)tot.e * XT_LN2 + bk.ll_const;
            if (bk.ll_out) bk.ll_out[first + x] = ll;
            my_ll += ll;
        }
    }

    // ---- block partial: fixed-order sum over the tile's track slots
    cx.sync();
    if (g0 == 0) red[x] = my_ll;
    cx.sync();
    if (tid == 0) {
        double s2 = 0.0;
        for (int i = 0; i < TT; ++i) s2 += red[i];
        a.partials[cx.block()] = s2;
    }
}
