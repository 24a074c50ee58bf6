// TEST INFRASTRUCTURE ONLY: runs the kernel body of extrack_amd/csrc/xt_kernel.h on CPU threads
// (one std::thread per GPU thread, a pthread barrier for __syncthreads) so that the index logic and
// the extended-range arithmetic can be checked against the oracle without a GPU.  Never shipped,
// never reachable from the product package.
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "emul_ctx.h"
#include "../../extrack_amd/csrc/xt_dispatch.h"
#include "../../extrack_amd/csrc/xt_entry.h"
#include "../../extrack_amd/csrc/xt_fast2.h"
#include "../../extrack_amd/csrc/xt_grad.h"
#include "../../extrack_amd/csrc/xt_grad_host.h"
#include "../../extrack_amd/csrc/xt_hist.h"
#include "../../extrack_amd/csrc/xt_hist_host.h"
#include "../../extrack_amd/csrc/xt_reg2.h"
#include "../../extrack_amd/csrc/xt_gradr.h"
#include "../../extrack_amd/csrc/xt_rev.h"
#include "../../extrack_amd/csrc/xt_seqmat.h"
#include "../../extrack_amd/csrc/xt_tables.h"
#include "../../extrack_amd/csrc/xt_th.h"
#include "../../extrack_amd/csrc/xt_thgrad.h"
#include "../../extrack_amd/csrc/xt_thgrad2.h"
#include "../../extrack_amd/csrc/xt_big.h"

struct EmulLauncher {
    XtKernelArgs a;
    int threads, nblocks;
    size_t lds_bytes;
    template <int F, int D, int K>
    bool run_f2()
    {
        const int nw = threads / 64;
        for (int b = 0; b < nblocks; ++b) {
            std::vector<double> smem(lds_bytes / 8 + 16, 0.0);
            pthread_barrier_t bar;
            pthread_barrier_init(&bar, nullptr, threads);
            std::vector<pthread_barrier_t> wb(nw);
            std::vector<std::vector<double>> ws(nw, std::vector<double>(64, 0.0));
            for (auto& x : wb) pthread_barrier_init(&x, nullptr, 64);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t]() {
                    HostCtx cx{t, threads, b, nblocks, smem.data(), &bar};
                    cx.wbar_ = &wb[t >> 6];
                    cx.wscr_ = ws[t >> 6].data();
                    xt_ll_s2_body<F, D, K>(a, cx);
                });
            for (auto& x : th) x.join();
            pthread_barrier_destroy(&bar);
            for (auto& x : wb) pthread_barrier_destroy(&x);
        }
        return true;
    }

    template <int GP, int D, int K>
    bool run_entry()
    {
        const int nw = threads / 64;
        for (int b = 0; b < nblocks; ++b) {
            std::vector<double> smem(lds_bytes / 8 + 16, 0.0);
            pthread_barrier_t bar;
            pthread_barrier_init(&bar, nullptr, threads);
            std::vector<pthread_barrier_t> wb(nw);
            std::vector<std::vector<double>> ws(nw, std::vector<double>(64, 0.0));
            for (auto& x : wb) pthread_barrier_init(&x, nullptr, 64);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t]() {
                    HostCtx cx{t, threads, b, nblocks, smem.data(), &bar};
                    cx.wbar_ = &wb[t >> 6];
                    cx.wscr_ = ws[t >> 6].data();
                    xt_entry_body<GP, D, K>(a, cx);
                });
            for (auto& x : th) x.join();
            pthread_barrier_destroy(&bar);
            for (auto& x : wb) pthread_barrier_destroy(&x);
        }
        return true;
    }

    template <int G_, int D, int K, bool PREDS>
    bool run()
    {
        const int nw = (threads + 63) / 64;
        if (threads % 64) return false;  // the posterior read-out exchanges values between the lanes of a wavefront
        for (int b = 0; b < nblocks; ++b) {
            std::vector<double> smem(lds_bytes / 8 + 16, 0.0);
            pthread_barrier_t bar;
            pthread_barrier_init(&bar, nullptr, threads);
            std::vector<pthread_barrier_t> wb(nw);
            std::vector<std::vector<double>> ws(nw, std::vector<double>(64, 0.0));
            for (auto& x : wb) pthread_barrier_init(&x, nullptr, 64);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t]() {
                    HostCtx cx{t, threads, b, nblocks, smem.data(), &bar};
                    cx.wbar_ = &wb[t >> 6];
                    cx.wscr_ = ws[t >> 6].data();
                    xt_track_body<G_, D, K, PREDS>(a, cx);
                });
            for (auto& x : th) x.join();
            pthread_barrier_destroy(&bar);
            for (auto& x : wb) pthread_barrier_destroy(&x);
        }
        return true;
    }
};

bool emul_r2(int F, int D, int K, int KS, int NP, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks);  // emul_r2.cpp
bool emul_gradr(int G, int D, int K, int NPC, const XtKernelArgs& a, const XtGradArgs& ga, int nblocks, int threads, size_t lds_doubles);  // emul_gradr.cpp
bool emul_rev(int G, int D, int K, int nbuf, const XtKernelArgs& a, const XtRevArgs& ra, int nblocks, int threads, size_t lds_doubles);  // emul_rev.cpp

// per-sequence matrix (xt_seqmat.h): raw output buffer [N][E][G] of the NEXT xt_emul_run call, which then runs the general kernel body
static double* g_seq_raw = nullptr;
extern "C" void xt_emul_set_seq_out(double* raw) { g_seq_raw = raw; }
extern "C" long long xt_emul_seq_columns(int S, int L, int NS, int F, int isBL) { return xt_seq_columns(S, L, NS, F, isBL); }
extern "C" int xt_emul_seq_reorder(int S, int NS, int F, long long N, int L, int isBL, double pBL, const double* TrMat, const double* p_stay,
                                   const double* raw, double* lp)
{
    XtConfig cfg;
    if (!xt_build_config(S, NS, F, cfg).empty()) return -1;
    XtModelHost m{S, NS, 1, {0, 0, 0}, 0.0, 0.0, pBL, nullptr, nullptr, TrMat, p_stay};
    xt_seq_reorder(cfg, m, N, L, isBL, raw, lp);
    return 0;
}

extern "C" int xt_emul_run(const double* tracks, const double* sigma, long long N, int L, int D, int KS, int S, int NS, int F,
                           int isBL, int min_len, int locerr_mode, int locerr_dims, const double* locerr, double slope,
                           double offset, double pBL, const double* ds, const double* Fs, const double* TrMat,
                           const double* p_stay, int preds, int nblocks, double* ll_out, double* preds_out, double* total,
                           int* info /* [4]: tpb, threads, lds_bytes, E */)
{
    XtConfig cfg;
    std::string err = xt_build_config(S, NS, F, cfg);
    if (!err.empty()) return -1;
    XtModelHost m{S, NS, locerr_dims, {0, 0, 0}, slope, offset, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr ? locerr[k < locerr_dims ? k : 0] : 0.0;
    std::vector<double> blob;
    xt_build_blob(m, cfg, blob);
    const int K = locerr_mode == 0 ? locerr_dims : KS;
    EmulLauncher l;
    memset(&l.a, 0, sizeof(l.a));
    xt_fill_args_from_config(cfg, l.a);
    int tpb, threads;
    xt_geometry(cfg, D, K, tpb, threads);
    const bool big = threads > 1024 || getenv("XT_EMUL_BIG") != nullptr;  // one lane per track, state in (emulated) global memory: xt_big.h
    std::vector<double> partials(nblocks, 0.0);
    l.a.tracks = tracks;
    l.a.sigma = locerr_mode ? sigma : nullptr;
    l.a.blob = blob.data();
    l.a.base_tab = cfg.base_tab.data();
    l.a.off_tab = cfg.off_tab.data();
    l.a.ll_out = ll_out;
    l.a.partials = partials.data();
    l.a.preds_out = preds_out;
    l.a.N = N;
    l.a.L = L;
    l.a.TPB = tpb;
    l.a.isBL = isBL;
    l.a.min_len = min_len;
    l.a.locerr_mode = locerr_mode;
    l.a.KS = KS;
    {   // the launcher's scaling decision for the 2-state fast path (extrack_hip.hip: xt_launch_group)
        double lo = INFINITY, hi = -INFINITY;
        if (locerr_mode == 0) {
            for (int k = 0; k < locerr_dims && k < 3; ++k) {
                lo = std::min(lo, m.locerr[k] * m.locerr[k]);
                hi = std::max(hi, m.locerr[k] * m.locerr[k]);
            }
        } else {
            for (long long i = 0; i < N * L * KS; ++i) {
                double v = sigma[i];
                if (locerr_mode == 2) v = std::max(v * slope + offset, 1e-6);
                if (v == v) {
                    lo = std::min(lo, v * v);
                    hi = std::max(hi, v * v);
                }
            }
        }
        l.a.well_scaled = xt_model_well_scaled(blob, lo, hi) ? 1 : 0;
        if (getenv("XT_EMUL_GUARDED")) l.a.well_scaled = 0;
    }
    l.a.ll_const = -(double)(L - 1) * D * 0.5 * XT_LOG2PI;
    l.threads = threads;
    l.nblocks = nblocks;
    l.lds_bytes = xt_lds_bytes(cfg, D, K, tpb);
    if (info) {
        info[0] = tpb;
        info[1] = threads;
        info[2] = (int)l.lds_bytes;
        info[3] = cfg.E;
    }
    double* const seq_raw = g_seq_raw;
    g_seq_raw = nullptr;
    l.a.seq_out = seq_raw;
    if (big && !seq_raw) {
        const int NW = 2;
        XtBigArgs ba;
        ba.ws_stride = xt_big_ws_doubles(cfg.E, D, K);
        std::vector<double> ws((size_t)ba.ws_stride * nblocks * NW, xt_emul_poison() ? NAN : 0.0);
        ba.ws = ws.data();
        const size_t ldsd = (size_t)((xt_tab_doubles(S, cfg.G) + 1) & ~1) + 64 * NW;
        if (info) {
            info[0] = 64 * NW;
            info[1] = 64 * NW;
            info[2] = (int)(ldsd * 8);
        }
#define XT_BIG_RUN(DD, KK)                                                                                                \
    th_emul_blocks(nblocks, 64 * NW, ldsd, [&](HostCtx& cx) {                                                              \
        if (preds) xt_big_body<DD, KK, true>(l.a, ba, cx);                                                                 \
        else xt_big_body<DD, KK, false>(l.a, ba, cx);                                                                      \
    })
        if (D == 1 && K == 1) XT_BIG_RUN(1, 1);
        else if (D == 2 && K == 1) XT_BIG_RUN(2, 1);
        else if (D == 2 && K == 2) XT_BIG_RUN(2, 2);
        else if (D == 3 && K == 1) XT_BIG_RUN(3, 1);
        else if (D == 3 && K == 3) XT_BIG_RUN(3, 3);
        else return -3;
#undef XT_BIG_RUN
    } else if (big) {
        return -2;
    } else if (seq_raw) {
        if (!xt_dispatch(cfg.G, D, K, preds != 0, l)) return -3;
    } else if (xt_use_reg2(S, NS, F) && !preds && getenv("XT_EMUL_REG2")) {
        XtGradArgs ga;
        memset(&ga, 0, sizeof(ga));
        if (info) {
            info[0] = 64 >> (F - 1);
            info[1] = 64 * XT_F2_WAVES;
            info[2] = xt_r2_block_bytes(0, D, locerr_mode ? KS : 0, 64 >> (F - 1));
        }
        if (!emul_r2(F, D, K, locerr_mode ? KS : 0, 0, l.a, ga, nblocks)) return -3;
    } else if (xt_use_fast2(S, NS, F, preds != 0) && !getenv("XT_EMUL_GENERIC")) {
        l.threads = 64 * XT_F2_WAVES;
        l.a.TPB = 0;
        l.lds_bytes = (size_t)xt_f2_block_bytes(D, K, locerr_mode ? KS : 0, 64 >> (F - 1));
        if (info) {
            info[0] = 64 >> (F - 1);
            info[1] = l.threads;
            info[2] = (int)l.lds_bytes;
        }
        if (!xt_dispatch_f2(F, D, K, l)) return -3;
    } else if (xt_use_entry(NS, cfg.G, cfg.NG, preds != 0) && !getenv("XT_EMUL_GENERIC")) {
        int tpb2, thr2;
        size_t lds2;
        xt_entry_geometry(S, cfg.G, cfg.E, cfg.NG, D, K, tpb2, thr2, lds2);
        l.threads = thr2;
        l.a.TPB = tpb2;
        l.lds_bytes = lds2;
        if (info) {
            info[0] = tpb2;
            info[1] = thr2;
            info[2] = (int)lds2;
        }
        if (!xt_dispatch_entry(xt_entry_gp(cfg.G), D, K, l)) return -3;
    } else if (!xt_dispatch(cfg.G, D, K, preds != 0, l)) return -3;
    double s = 0.0;
    for (double p : partials) s += p;
    if (total) *total = s;
    return 0;
}


// Several length buckets served by ONE emulated launch through the bucket-descriptor table (the product's launch mode).
extern "C" int xt_emul_run_multi(int nbuckets, const double** tracks, const long long* Ns, const int* Ls, int D, int S, int NS, int F,
                                 int max_len, int min_len, int locerr_dims, const double* locerr, double pBL, const double* ds,
                                 const double* Fs, const double* TrMat, const double* p_stay, const int* blocks_per_bucket,
                                 double** ll_out, double* total)
{
    if (nbuckets < 1 || nbuckets > XT_MAX_BUCKETS) return -4;
    XtConfig cfg;
    if (!xt_build_config(S, NS, F, cfg).empty()) return -1;
    XtModelHost m{S, NS, locerr_dims, {0, 0, 0}, 0.0, 0.0, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr[k < locerr_dims ? k : 0];
    std::vector<double> blob;
    xt_build_blob(m, cfg, blob);
    const int K = locerr_dims;
    EmulLauncher l;
    memset(&l.a, 0, sizeof(l.a));
    xt_fill_args_from_config(cfg, l.a);
    std::vector<XtBucketDesc> descs(nbuckets);
    int acc = 0;
    for (int i = 0; i < nbuckets; ++i) {
        descs[i] = XtBucketDesc{tracks[i], nullptr, ll_out ? ll_out[i] : nullptr, nullptr, Ns[i], Ls[i], Ls[i] != max_len ? 1 : 0,
                                -(double)(Ls[i] - 1) * D * 0.5 * XT_LOG2PI};
        acc += blocks_per_bucket[i];
        l.a.blk_end[i] = acc;
    }
    l.a.desc = descs.data();
    l.a.ndesc = nbuckets;
    l.nblocks = acc;
    std::vector<double> partials(acc, 0.0);
    l.a.blob = blob.data();
    l.a.base_tab = cfg.base_tab.data();
    l.a.off_tab = cfg.off_tab.data();
    l.a.partials = partials.data();
    l.a.min_len = min_len;
    l.a.locerr_mode = 0;
    l.a.KS = 1;
    {
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < locerr_dims && k < 3; ++k) {
            lo = std::min(lo, m.locerr[k] * m.locerr[k]);
            hi = std::max(hi, m.locerr[k] * m.locerr[k]);
        }
        l.a.well_scaled = xt_model_well_scaled(blob, lo, hi) ? 1 : 0;
    }
    bool ok;
    if (xt_use_fast2(S, NS, F, false)) {
        l.threads = 64 * XT_F2_WAVES;
        l.lds_bytes = (size_t)xt_f2_block_bytes(D, K, 0, 64 >> (F - 1));
        ok = xt_dispatch_f2(F, D, K, l);
    } else if (xt_use_entry(NS, cfg.G, cfg.NG, false)) {
        int tpb2, thr2;
        size_t lds2;
        xt_entry_geometry(S, cfg.G, cfg.E, cfg.NG, D, K, tpb2, thr2, lds2);
        l.threads = thr2;
        l.a.TPB = tpb2;
        l.lds_bytes = lds2;
        ok = xt_dispatch_entry(xt_entry_gp(cfg.G), D, K, l);
    } else {
        int tpb, threads;
        xt_geometry(cfg, D, K, tpb, threads);
        l.threads = threads;
        l.a.TPB = tpb;
        l.lds_bytes = xt_lds_bytes(cfg, D, K, tpb);
        ok = xt_dispatch(cfg.G, D, K, false, l);
    }
    if (!ok) return -3;
    double s = 0.0;
    for (double p : partials) s += p;
    if (total) *total = s;
    return 0;
}


// Per-track time steps for the next xt_emul_th_run / xt_emul_th_predict call: dt [N][L] and one p_stay table [G] per chunk.
static const double* g_th_dt = nullptr;
static const double* g_th_pstay_chunks = nullptr;
extern "C" void xt_emul_th_set_dt(const double* dt, const double* p_stay_chunks)
{
    g_th_dt = dt;
    g_th_pstay_chunks = p_stay_chunks;
}

// Frozen-plan gradient (xt_thgrad.h) for the next xt_emul_th_run call: after the plan + apply bodies the gradient body follows the same plan.
// tangents: n_dir rows of [locerr(3), slope, offset, pBL, ds2(S), Fs(S), TrMat(S*S), p_stay(G)]; out: [1 + n_dir] = {sum LL, d sum LL / d theta_i};
// ll_out: per-track LL of the gradient body (or nullptr); waves: wavefronts per emulated workgroup.
static int g_thg_ndir = -1, g_thg_waves = 1;
static const double* g_thg_tangents = nullptr;
static double* g_thg_out = nullptr;
static double* g_thg_ll = nullptr;
extern "C" void xt_emul_th_set_grad(int n_dir, const double* tangents, double* out, double* ll_out, int waves)
{
    g_thg_ndir = n_dir;
    g_thg_tangents = tangents;
    g_thg_out = out;
    g_thg_ll = ll_out;
    g_thg_waves = waves > 0 ? waves : 1;
}

// blobs of all chunks, one after the other (p_stay differs per chunk); returns the stride in doubles
static int64_t th_chunk_blobs(const XtModelHost& m, int nchunks, int G, std::vector<double>& blobs)
{
    int64_t stride = 0;
    for (int c = 0; c < nchunks; ++c) {
        XtModelHost mc = m;
        mc.p_stay = g_th_pstay_chunks + (size_t)c * G;
        std::vector<double> b;
        int G2 = 0;
        xt_th_build_blob(mc, b, G2);
        if (c == 0) {
            stride = (int64_t)b.size();
            blobs.assign((size_t)stride * nchunks, 0.0);
        }
        memcpy(blobs.data() + (size_t)c * stride, b.data(), b.size() * sizeof(double));
    }
    return stride;
}

extern "C" int xt_emul_th_run(const double* tracks, const double* sigma, long long N, int L, int D, int KS, int S, int NS, int F, int isBL,
                              int min_len, int locerr_mode, int locerr_dims, const double* locerr, double slope, double offset, double pBL,
                              const double* ds, const double* Fs, const double* TrMat, const double* p_stay, double threshold, int max_nb,
                              int chunk, int capE, int TT, int apply_threads, int nblocks, double* ll_out, double* total,
                              int* hdr_out /* [nchunks][L][2] */, unsigned short* members_out /* [nchunks][L][capE] */,
                              unsigned short* gstart_out /* [nchunks][L][capE+1] */, int* status_out /* [nchunks][4] */)
{
    XtModelHost m{S, NS, locerr_dims, {0, 0, 0}, slope, offset, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr ? locerr[k < locerr_dims ? k : 0] : 0.0;
    std::vector<double> blob;
    int G = 0;
    if (!xt_th_build_blob(m, blob, G).empty()) return -1;
    const int K = locerr_mode == 0 ? locerr_dims : KS;
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.tracks = tracks;
    a.sigma = locerr_mode ? sigma : nullptr;
    a.blob = blob.data();
    a.ll_out = ll_out;
    a.N = N;
    a.L = L;
    a.S = S;
    a.NS = NS;
    a.G = G;
    a.F = F;
    a.isBL = isBL;
    a.min_len = min_len;
    a.locerr_mode = locerr_mode;
    a.KS = KS ? KS : 1;
    a.chunk = chunk;
    a.nchunks = (int)((N + chunk - 1) / chunk);
    a.capE = capE;
    a.max_nb = max_nb;
    a.threshold = threshold;
    a.ll_const = -(double)(L - 1) * D * 0.5 * XT_LOG2PI;
    std::vector<double> chunk_blobs;
    if (g_th_dt) {
        a.dt = g_th_dt;
        a.blob_stride = th_chunk_blobs(m, a.nchunks, G, chunk_blobs);
        a.blob = chunk_blobs.data();
        g_th_dt = nullptr;
    }
    std::vector<uint16_t> mem((size_t)a.nchunks * L * capE, 0), gst((size_t)a.nchunks * L * (capE + 1), 0);
    std::vector<uint32_t> mpack((size_t)a.nchunks * L * capE, 0);
    std::vector<uint8_t> gnew((size_t)a.nchunks * L * capE, 0);
    a.mpack = mpack.data();
    a.gnew = gnew.data();
    std::vector<int32_t> hdr((size_t)a.nchunks * L * 2, 0), status((size_t)a.nchunks * 4, 0);
    a.members = mem.data();
    a.gstart = gst.data();
    a.hdr = hdr.data();
    a.status = status.data();
    const int plan_blocks = (nblocks < 0 ? -nblocks : nblocks) < a.nchunks ? (nblocks < 0 ? -nblocks : nblocks) : a.nchunks;
    a.wsP = a.wsE = capE;
    a.ws_lds = 0;
    if (getenv("XT_EMUL_TH_WSP")) {  // exercise the LDS-resident workspace with explicit capacities
        a.wsP = atoi(getenv("XT_EMUL_TH_WSP"));
        a.wsE = atoi(getenv("XT_EMUL_TH_WSE"));
        a.ws_lds = 1;
    }
    a.pcap = chunk < XT_TH_PILOT ? chunk : XT_TH_PILOT;
    a.pair_lanes_max_p = getenv("XT_EMUL_TH_PAIR_LANES") ? atoi(getenv("XT_EMUL_TH_PAIR_LANES")) : 4;
    // per-step plan arrays in the global workspace (what the library does beyond XT_TH_MAXCAP expanded sequences; XT_EMUL_TH_PLAN_GLB forces it)
    a.plan_glb = (!a.ws_lds && (capE > XT_TH_MAXCAP || getenv("XT_EMUL_TH_PLAN_GLB"))) ? 1 : 0;
    a.ws_stride = xt_th_ws_doubles(a.wsP, a.wsE, D, K, F, NS, S, a.pcap) + (a.plan_glb ? xt_th_plan_glb_doubles(capE) : 0);
    std::vector<double> ws((size_t)a.ws_stride * plan_blocks, 0.0);
    a.ws = ws.data();
    if (getenv("XT_EMUL_TH_STP")) {  // exercise the LDS staging copy used with the global workspace
        a.stP = atoi(getenv("XT_EMUL_TH_STP"));
        a.stE = atoi(getenv("XT_EMUL_TH_STE"));
    }
    const size_t plan_lds = xt_th_plan_lds_doubles(S, G, capE, D, K, XT_TH_CMAT_WORDS, a.plan_glb != 0) +
                            (a.ws_lds ? (size_t)a.ws_stride : 0) + (size_t)a.pcap * (a.stP * D + a.stE * K) + 8;
    const int plan_threads = apply_threads;  // same block size for both kernels in the emulation
#define TH_RUN(BODY, NB, NT, LDS)                                                                             \
    do {                                                                                                      \
        if (D == 1 && K == 1) th_emul_blocks(NB, NT, LDS, [&](HostCtx& cx) { BODY<1, 1, false>(a, cx); });           \
        else if (D == 2 && K == 1) th_emul_blocks(NB, NT, LDS, [&](HostCtx& cx) { BODY<2, 1, false>(a, cx); });      \
        else if (D == 2 && K == 2) th_emul_blocks(NB, NT, LDS, [&](HostCtx& cx) { BODY<2, 2, false>(a, cx); });      \
        else if (D == 3 && K == 1) th_emul_blocks(NB, NT, LDS, [&](HostCtx& cx) { BODY<3, 1, false>(a, cx); });      \
        else if (D == 3 && K == 3) th_emul_blocks(NB, NT, LDS, [&](HostCtx& cx) { BODY<3, 3, false>(a, cx); });      \
        else return -3;                                                                                       \
    } while (0)
    if (a.plan_glb) {  // the compile-time global-workspace variant (the only one that honours plan_glb)
        if (D == 1 && K == 1) th_emul_blocks(plan_blocks, plan_threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<1, 1, false, 0>(a, cx); });
        else if (D == 2 && K == 1) th_emul_blocks(plan_blocks, plan_threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<2, 1, false, 0>(a, cx); });
        else if (D == 2 && K == 2) th_emul_blocks(plan_blocks, plan_threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<2, 2, false, 0>(a, cx); });
        else if (D == 3 && K == 1) th_emul_blocks(plan_blocks, plan_threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<3, 1, false, 0>(a, cx); });
        else if (D == 3 && K == 3) th_emul_blocks(plan_blocks, plan_threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<3, 3, false, 0>(a, cx); });
        else return -3;
    } else {
        TH_RUN(xt_th_plan_body, plan_blocks, plan_threads, plan_lds);
    }
    if (hdr_out) memcpy(hdr_out, hdr.data(), hdr.size() * sizeof(int32_t));
    if (members_out) memcpy(members_out, mem.data(), mem.size() * sizeof(uint16_t));
    if (gstart_out) memcpy(gstart_out, gst.data(), gst.size() * sizeof(uint16_t));
    if (status_out) memcpy(status_out, status.data(), status.size() * sizeof(int32_t));
    int maxG = 0, sumE = 0;
    for (int c = 0; c < a.nchunks; ++c) {
        if (status[(size_t)c * 4]) return -5;  // plan capacity overflow
        maxG = status[(size_t)c * 4 + 2] > maxG ? status[(size_t)c * 4 + 2] : maxG;
        sumE = status[(size_t)c * 4 + 3] > sumE ? status[(size_t)c * 4 + 3] : sumE;
    }
    a.capG = maxG;
    a.TT = TT;
    a.logTT = 0;
    while ((1 << a.logTT) < TT) ++a.logTT;
    // nblocks doubles as "workgroups per chunk"; a negative value asks for the streamed-plan mode
    a.bpc = nblocks < 0 ? -nblocks : nblocks;
    a.plan_cap = nblocks < 0 ? (getenv("XT_EMUL_TH_DIRECT") ? -1 : 0) : (sumE > 0 ? sumE : 1);  // streamed through LDS, or (direct) read from global memory
    const int grid = a.nchunks * a.bpc;
    std::vector<double> partials(grid, 0.0);
    a.partials = partials.data();
    const bool uni = TT == 64;
    const bool single_buf = getenv("XT_EMUL_TH_SINGLE") && (uni ? (apply_threads / 64) : (apply_threads / TT)) * XT_TH_GPW >= maxG;
    const size_t apply_lds = xt_th_apply_lds_doubles(S, G, maxG, TT, D, K, a.locerr_mode ? a.KS : 0, L, a.plan_cap, uni, single_buf);
#define TH_APPLY_SGL(DD, KK) th_emul_blocks(grid, apply_threads, apply_lds, [&](HostCtx& cx) { (a.dt ? xt_th_apply_body<DD, KK, true, true, true>(a, cx) : xt_th_apply_body<DD, KK, true, true, false>(a, cx)); })
#define TH_APPLY_UNI(DD, KK) th_emul_blocks(grid, apply_threads, apply_lds, [&](HostCtx& cx) { (a.dt ? xt_th_apply_body<DD, KK, true, false, true>(a, cx) : xt_th_apply_body<DD, KK, true, false, false>(a, cx)); })
#define TH_APPLY_GEN(DD, KK) th_emul_blocks(grid, apply_threads, apply_lds, [&](HostCtx& cx) { (a.dt ? xt_th_apply_body<DD, KK, false, false, true>(a, cx) : xt_th_apply_body<DD, KK, false, false, false>(a, cx)); })
#define TH_APPLY_GSG(DD, KK) th_emul_blocks(grid, apply_threads, apply_lds, [&](HostCtx& cx) { (a.dt ? xt_th_apply_body<DD, KK, false, true, true>(a, cx) : xt_th_apply_body<DD, KK, false, true, false>(a, cx)); })
#define TH_APPLY(DD, KK)                            \
    do {                                            \
        if (single_buf && uni) TH_APPLY_SGL(DD, KK); \
        else if (single_buf) TH_APPLY_GSG(DD, KK);  \
        else if (uni) TH_APPLY_UNI(DD, KK);         \
        else TH_APPLY_GEN(DD, KK);                  \
    } while (0)
    if (D == 1 && K == 1) TH_APPLY(1, 1);
    else if (D == 2 && K == 1) TH_APPLY(2, 1);
    else if (D == 2 && K == 2) TH_APPLY(2, 2);
    else if (D == 3 && K == 1) TH_APPLY(3, 1);
    else if (D == 3 && K == 3) TH_APPLY(3, 3);
    else return -3;
#undef TH_RUN
    double s = 0.0;
    for (double p : partials) s += p;
    if (total) *total = s;
    if (g_thg_ndir >= 0) {  // the frozen-plan gradient body on the plan just made
        const int n_dir = g_thg_ndir;
        g_thg_ndir = -1;
        if (a.dt) return -7;
        const int TB = xt_grad_tb_doubles(S, G), row = 6 + 2 * S + S * S + G, NW = g_thg_waves;
        std::vector<double> dblob((size_t)(n_dir > 0 ? n_dir : 1) * TB, 0.0);
        for (int i = 0; i < n_dir; ++i) {
            const double* r = g_thg_tangents + (size_t)i * row;
            extrack_model_tangent t;
            for (int k = 0; k < 3; ++k) t.locerr[k] = r[k];
            t.slope = r[3];
            t.offset = r[4];
            t.pBL = r[5];
            t.ds2 = r + 6;
            t.Fs = r + 6 + S;
            t.TrMat = r + 6 + 2 * S;
            t.p_stay = r + 6 + 2 * S + S * S;
            xt_th_build_tangent_block(m, t, locerr_mode, dblob.data() + (size_t)i * TB);
        }
        XtThGradArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.TB = TB;
        ga.capP = maxG > S ? maxG : S;
        ga.rows_global = getenv("XT_EMUL_THG_ROWS_GLOBAL") ? 1 : 0;
        ga.ws_stride = xt_thg_ws_doubles(ga.capP, L, D, K, ga.rows_global ? xt_thg_rows(S, G) : 0);
        std::vector<double> gws((size_t)ga.ws_stride * grid * NW, xt_emul_poison() ? NAN : 0.0);
        std::vector<double> gp((size_t)grid * NW * (1 + TB), 0.0);
        ga.ws = gws.data();
        ga.gpartials = gp.data();
        a.ll_out = g_thg_ll;
        int nrows = grid * NW;
        const int TT2 = getenv("XT_EMUL_THG2") ? atoi(getenv("XT_EMUL_THG2")) : 0;  // tracks per tile of the (sequence, track)-lane body (xt_thgrad2.h)
        if (TT2 > 0) {
            a.TT = TT2;
            a.logTT = 0;
            while ((1 << a.logTT) < TT2) ++a.logTT;
            const int threads2 = 64 * NW, LPT = threads2 / TT2;
            if (LPT < S || S * G > 4 * LPT) return -8;
            ga.ws_stride = xt_thg2_ws_doubles(ga.capP, L, TT2, D, K, G);
            gws.assign((size_t)ga.ws_stride * grid, xt_emul_poison() ? NAN : 0.0);
            ga.ws = gws.data();
            const size_t glds2 = (size_t)xt_thg2_lds_doubles(S, G, ga.capP, capE, TT2, D, K, threads2, TB);
#define TH_GRAD2(DD, KK) th_emul_blocks(grid, threads2, glds2, [&](HostCtx& cx) { if (S * G <= LPT) xt_thg2_body<DD, KK, 1>(a, ga, cx); else xt_thg2_body<DD, KK, 4>(a, ga, cx); })
            if (D == 1 && K == 1) TH_GRAD2(1, 1);
            else if (D == 2 && K == 1) TH_GRAD2(2, 1);
            else if (D == 2 && K == 2) TH_GRAD2(2, 2);
            else if (D == 3 && K == 1) TH_GRAD2(3, 1);
            else if (D == 3 && K == 3) TH_GRAD2(3, 3);
            else return -3;
#undef TH_GRAD2
            nrows = grid;
        } else {
        const size_t glds = (size_t)xt_thg_lds_doubles(S, G, NW, capE, ga.rows_global != 0);
#define TH_GRAD(DD, KK) th_emul_blocks(grid, 64 * NW, glds, [&](HostCtx& cx) { if (ga.rows_global) xt_thg_body<DD, KK, true>(a, ga, cx); else xt_thg_body<DD, KK, false>(a, ga, cx); })
        if (D == 1 && K == 1) TH_GRAD(1, 1);
        else if (D == 2 && K == 1) TH_GRAD(2, 1);
        else if (D == 2 && K == 2) TH_GRAD(2, 2);
        else if (D == 3 && K == 1) TH_GRAD(3, 1);
        else if (D == 3 && K == 3) TH_GRAD(3, 3);
        else return -3;
#undef TH_GRAD
        }
        std::vector<double> adj(TB, 0.0);
        g_thg_out[0] = 0.0;
        for (int w = 0; w < nrows; ++w) {
            g_thg_out[0] += gp[(size_t)w * (1 + TB)];
            for (int c = 0; c < TB; ++c) adj[c] += gp[(size_t)w * (1 + TB) + 1 + c];
        }
        for (int i = 0; i < n_dir; ++i) {
            double s2 = 0.0;
            for (int c = 0; c < TB; ++c) s2 += adj[c] * dblob[(size_t)i * TB + c];
            g_thg_out[1 + i] = s2;
        }
    }
    return 0;
}


// ---- threshold-fusion posteriors: the plan body in prediction mode (the first <= 30 tracks of a chunk are the pilots).
extern "C" int xt_emul_th_predict(const double* tracks, const double* sigma, long long N, int L, int D, int KS, int S, int F, int isBL,
                                  int min_len, int locerr_mode, int locerr_dims, const double* locerr, double slope, double offset, double pBL,
                                  const double* ds, const double* Fs, const double* TrMat, const double* p_stay, double threshold, int max_nb,
                                  int chunk, int capE, int threads, int nblocks, double* preds_out, int* status_out)
{
    XtModelHost m{S, 1, locerr_dims, {0, 0, 0}, slope, offset, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr ? locerr[k < locerr_dims ? k : 0] : 0.0;
    std::vector<double> blob;
    int G = 0;
    if (!xt_th_build_blob(m, blob, G).empty()) return -1;
    const int K = locerr_mode == 0 ? locerr_dims : KS;
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.tracks = tracks;
    a.sigma = locerr_mode ? sigma : nullptr;
    a.blob = blob.data();
    a.preds_out = preds_out;
    a.N = N;
    a.L = L;
    a.S = S;
    a.NS = 1;
    a.G = G;
    a.F = F;
    a.isBL = isBL;
    a.min_len = min_len;
    a.locerr_mode = locerr_mode;
    a.KS = KS ? KS : 1;
    a.chunk = chunk;
    a.nchunks = (int)((N + chunk - 1) / chunk);
    a.capE = capE;
    a.max_nb = max_nb;
    a.threshold = threshold;
    a.pcap = chunk < XT_TH_PILOT ? chunk : XT_TH_PILOT;
    a.pair_lanes_max_p = 4;
    a.wsP = a.wsE = capE;
    a.ws_lds = 0;
    std::vector<double> chunk_blobs;
    if (g_th_dt) {
        a.dt = g_th_dt;
        a.blob_stride = th_chunk_blobs(m, a.nchunks, G, chunk_blobs);
        a.blob = chunk_blobs.data();
        g_th_dt = nullptr;
    }
    std::vector<int32_t> status((size_t)a.nchunks * 4, 0);
    a.status = status.data();
    const int blocks = nblocks < a.nchunks ? nblocks : a.nchunks;
    if (getenv("XT_EMUL_TH_WSP")) {  // LDS-resident state with explicit capacities
        a.wsP = atoi(getenv("XT_EMUL_TH_WSP"));
        a.wsE = atoi(getenv("XT_EMUL_TH_WSE"));
        a.ws_lds = 1;
    }
    const int64_t state = xt_th_ws_doubles(a.wsP, a.wsE, D, K, F, 1, S, a.pcap, true);
    a.ws_stride = xt_th_hist_doubles(a.wsE, a.pcap, true, L) + (a.ws_lds ? 0 : state);
    std::vector<double> ws((size_t)a.ws_stride * blocks, 0.0);
    a.ws = ws.data();
    const size_t lds = xt_th_plan_lds_doubles(S, G, capE, D, K) + (a.ws_lds ? (size_t)state : 0);
    if (D == 1 && K == 1) th_emul_blocks(blocks, threads, lds, [&](HostCtx& cx) { xt_th_plan_body<1, 1, true>(a, cx); });
    else if (D == 2 && K == 1) th_emul_blocks(blocks, threads, lds, [&](HostCtx& cx) { xt_th_plan_body<2, 1, true>(a, cx); });
    else if (D == 2 && K == 2) th_emul_blocks(blocks, threads, lds, [&](HostCtx& cx) { xt_th_plan_body<2, 2, true>(a, cx); });
    else if (D == 3 && K == 1) th_emul_blocks(blocks, threads, lds, [&](HostCtx& cx) { xt_th_plan_body<3, 1, true>(a, cx); });
    else if (D == 3 && K == 3) th_emul_blocks(blocks, threads, lds, [&](HostCtx& cx) { xt_th_plan_body<3, 3, true>(a, cx); });
    else return -3;
    if (status_out) memcpy(status_out, status.data(), status.size() * sizeof(int32_t));
    for (int c = 0; c < a.nchunks; ++c)
        if (status[(size_t)c * 4]) return -5;
    return 0;
}


// ---- threshold-fusion evaluation of several buckets through the bucket-descriptor table (ONE emulated plan launch and ONE
// apply launch for all of them, the product's launch mode).  Global scalar localisation error only.
extern "C" int xt_emul_th_run_multi(int nbuckets, const double** tracks, const long long* Ns, const int* Ls, int D, int S, int NS, int F,
                                    int max_len, int min_len, int locerr_dims, const double* locerr, double pBL, const double* ds,
                                    const double* Fs, const double* TrMat, const double* p_stay, double threshold, int max_nb, int chunk,
                                    int capE, int TT, int threads, int bpc, double** ll_out, double* total)
{
    XtModelHost m{S, NS, locerr_dims, {0, 0, 0}, 0.0, 0.0, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr[k < locerr_dims ? k : 0];
    std::vector<double> blob;
    int G = 0;
    if (!xt_th_build_blob(m, blob, G).empty()) return -1;
    const int K = locerr_dims;
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.blob = blob.data();
    a.S = S;
    a.NS = NS;
    a.G = G;
    a.F = F;
    a.min_len = min_len;
    a.KS = 1;
    a.chunk = chunk;
    a.capE = capE;
    a.max_nb = max_nb;
    a.threshold = threshold;
    a.pcap = chunk < XT_TH_PILOT ? chunk : XT_TH_PILOT;
    a.pair_lanes_max_p = 4;
    a.wsP = a.wsE = capE;
    a.nbuckets = nbuckets;
    std::vector<XtThBucket> desc(nbuckets);
    std::vector<int32_t> cend(nbuckets);
    std::vector<std::vector<uint16_t>> mem(nbuckets), gst(nbuckets);
    std::vector<std::vector<uint32_t>> mpk(nbuckets);
    std::vector<std::vector<uint8_t>> gnw(nbuckets);
    std::vector<std::vector<int32_t>> hdr(nbuckets);
    int total_chunks = 0, Lmax = 0;
    for (int i = 0; i < nbuckets; ++i) {
        total_chunks += (int)((Ns[i] + chunk - 1) / chunk);
        cend[i] = total_chunks;
        Lmax = Ls[i] > Lmax ? Ls[i] : Lmax;
    }
    std::vector<int32_t> status((size_t)total_chunks * 4, 0);
    for (int i = 0; i < nbuckets; ++i) {
        const size_t nch = (size_t)(cend[i] - (i ? cend[i - 1] : 0));
        mem[i].assign(nch * Ls[i] * capE, 0);
        mpk[i].assign(nch * Ls[i] * capE, 0);
        gnw[i].assign(nch * Ls[i] * capE, 0);
        gst[i].assign(nch * Ls[i] * (capE + 1), 0);
        hdr[i].assign(nch * Ls[i] * 2, 0);
        desc[i] = XtThBucket{tracks[i], nullptr, nullptr, ll_out ? ll_out[i] : nullptr, nullptr, Ns[i], Ls[i], Ls[i] != max_len ? 1 : 0,
                             -(double)(Ls[i] - 1) * D * 0.5 * XT_LOG2PI, mem[i].data(), mpk[i].data(), gst[i].data(), gnw[i].data(),
                             hdr[i].data(), status.data() + (size_t)(i ? cend[i - 1] : 0) * 4};
    }
    a.buckets = desc.data();
    a.chunk_end = cend.data();
    a.nchunks = total_chunks;
    a.Lmax = a.L = Lmax;
    const int plan_blocks = 3 < total_chunks ? 3 : total_chunks;
    a.ws_stride = xt_th_ws_doubles(a.wsP, a.wsE, D, K, F, NS, S, a.pcap);
    std::vector<double> ws((size_t)a.ws_stride * plan_blocks, 0.0);
    a.ws = ws.data();
    const size_t plan_lds = xt_th_plan_lds_doubles(S, G, capE, D, K);
    if (D == 2 && K == 1) th_emul_blocks(plan_blocks, threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<2, 1, false>(a, cx); });
    else if (D == 3 && K == 1) th_emul_blocks(plan_blocks, threads, plan_lds, [&](HostCtx& cx) { xt_th_plan_body<3, 1, false>(a, cx); });
    else return -3;
    int maxG = 0, sumE = 0;
    for (int c = 0; c < total_chunks; ++c) {
        if (status[(size_t)c * 4]) return -5;
        maxG = status[(size_t)c * 4 + 2] > maxG ? status[(size_t)c * 4 + 2] : maxG;
        sumE = status[(size_t)c * 4 + 3] > sumE ? status[(size_t)c * 4 + 3] : sumE;
    }
    a.capG = maxG;
    a.TT = TT;
    a.logTT = 0;
    while ((1 << a.logTT) < TT) ++a.logTT;
    a.bpc = bpc;
    a.plan_cap = sumE > 0 ? sumE : 1;
    const int grid = total_chunks * bpc;
    std::vector<double> partials(grid, 0.0);
    a.partials = partials.data();
    const bool uni = TT == 64;
    const size_t lds = xt_th_apply_lds_doubles(S, G, maxG, TT, D, K, 0, Lmax, a.plan_cap, uni, false);
    if (D == 2 && K == 1) {
        if (uni) th_emul_blocks(grid, threads, lds, [&](HostCtx& cx) { xt_th_apply_body<2, 1, true, false, false>(a, cx); });
        else th_emul_blocks(grid, threads, lds, [&](HostCtx& cx) { xt_th_apply_body<2, 1, false, false, false>(a, cx); });
    } else {
        if (uni) th_emul_blocks(grid, threads, lds, [&](HostCtx& cx) { xt_th_apply_body<3, 1, true, false, false>(a, cx); });
        else th_emul_blocks(grid, threads, lds, [&](HostCtx& cx) { xt_th_apply_body<3, 1, false, false, false>(a, cx); });
    }
    double sacc = 0.0;
    for (double p : partials) sacc += p;
    if (total) *total = sacc;
    return 0;
}


// ---- likelihood + gradient body (xt_grad.h): one bucket, all directions in one pass ------------------------------------------
struct EmulGradLauncher {
    XtKernelArgs a;
    XtGradArgs ga;
    int threads, nblocks;
    size_t lds_doubles;
    template <int G_, int D, int K>
    bool run()
    {
        for (int b = 0; b < nblocks; ++b) {
            std::vector<double> smem(lds_doubles + 16, 0.0);
            pthread_barrier_t bar;
            pthread_barrier_init(&bar, nullptr, threads);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t]() {
                    HostCtx cx{t, threads, b, nblocks, smem.data(), &bar};
                    xt_grad_body<G_, D, K>(a, ga, cx);
                });
            for (auto& x : th) x.join();
            pthread_barrier_destroy(&bar);
        }
        return true;
    }
};

template <int GG, class L>
static bool emul_grad_dispatch_dk(int D, int K, L& l)
{
    if (D == 1 && K == 1) return l.template run<GG, 1, 1>();
    if (D == 2 && K == 1) return l.template run<GG, 2, 1>();
    if (D == 2 && K == 2) return l.template run<GG, 2, 2>();
    if (D == 3 && K == 1) return l.template run<GG, 3, 1>();
    if (D == 3 && K == 3) return l.template run<GG, 3, 3>();
    return false;
}

// tangents: n_dir rows of [locerr(3), slope, offset, pBL, ds2(S), Fs(S), TrMat(S*S), p_stay(G)]
extern "C" int xt_emul_grad(const double* tracks, const double* sigma, long long N, int L, int D, int KS, int S, int NS, int F, int isBL,
                            int min_len, int locerr_mode, int locerr_dims, const double* locerr, double slope, double offset, double pBL,
                            const double* ds, const double* Fs, const double* TrMat, const double* p_stay, int n_dir, const double* tangents,
                            int nblocks, int tpb, int tan_lds, int generic_g, int PJ, double* ll_out, double* out /* [1 + n_dir] */)
{
    XtConfig cfg;
    if (!xt_build_config(S, NS, F, cfg).empty()) return -1;
    XtModelHost m{S, NS, locerr_dims, {0, 0, 0}, slope, offset, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr ? locerr[k < locerr_dims ? k : 0] : 0.0;
    std::vector<double> blob;
    xt_build_blob(m, cfg, blob);
    const int K = locerr_mode == 0 ? locerr_dims : KS;
    const int TB = xt_grad_tb_doubles(S, cfg.G);
    const int row = 6 + 2 * S + S * S + cfg.G;
    std::vector<double> dblob((size_t)(n_dir > 0 ? n_dir : 1) * TB, 0.0);
    for (int i = 0; i < n_dir; ++i) {
        const double* r = tangents + (size_t)i * row;
        extrack_model_tangent t;
        for (int k = 0; k < 3; ++k) t.locerr[k] = r[k];
        t.slope = r[3];
        t.offset = r[4];
        t.pBL = r[5];
        t.ds2 = r + 6;
        t.Fs = r + 6 + S;
        t.TrMat = r + 6 + 2 * S;
        t.p_stay = r + 6 + 2 * S + S * S;
        xt_build_tangent_block(m, t, cfg, locerr_mode, dblob.data() + (size_t)i * TB);
    }
    EmulGradLauncher l;
    memset(&l.a, 0, sizeof(l.a));
    xt_fill_args_from_config(cfg, l.a);
    const int threads = (tpb * cfg.NG * PJ + 63) / 64 * 64;
    if (threads > 1024) return -2;
    std::vector<double> gp((size_t)nblocks * (n_dir + 1), 0.0);
    l.a.tracks = tracks;
    l.a.sigma = locerr_mode ? sigma : nullptr;
    l.a.blob = blob.data();
    l.a.base_tab = cfg.base_tab.data();
    l.a.off_tab = cfg.off_tab.data();
    l.a.ll_out = ll_out;
    l.a.N = N;
    l.a.L = L;
    l.a.TPB = tpb;
    l.a.isBL = isBL;
    l.a.min_len = min_len;
    l.a.locerr_mode = locerr_mode;
    l.a.KS = KS;
    l.a.ll_const = -(double)(L - 1) * D * 0.5 * XT_LOG2PI;
    if (generic_g) l.a.G = cfg.G;
    l.ga.dblob = dblob.data();
    l.ga.gpartials = gp.data();
    l.ga.NP = n_dir;
    l.ga.TB = TB;
    l.ga.tan_lds = tan_lds;
    l.ga.PJ = PJ;
    l.threads = threads;
    l.nblocks = nblocks;
    size_t d = (size_t)((xt_tab_doubles(S, cfg.G) + 1) & ~1);
    if (tan_lds) d += (size_t)((n_dir * TB + 1) & ~1);
    d += (size_t)tpb * ((size_t)xt_grad_region_doubles(cfg.EP, D, K, n_dir) + xt_grad_acc_doubles(n_dir, cfg.NG) + xt_stage_doubles(D));
    l.lds_doubles = d;
    if (generic_g == 5) {  // xt_rev.h: reverse mode - adjoint of the model blob, contracted with the tangent blocks here (the product: a small kernel)
        if (!xt_rev_supported(cfg.G, cfg.NG)) return -5;
        const int tpbr = std::max(1, 256 / cfg.NG), thr = (tpbr * cfg.NG + 63) / 64 * 64;
        l.a.TPB = tpbr;
        XtRevArgs ra;
        memset(&ra, 0, sizeof(ra));
        std::vector<double> gp2((size_t)nblocks * (1 + TB), 0.0);
        ra.log_stride = (int64_t)std::max(L - 2, 1) * xt_rev_step_doubles(cfg.NG, D, K);
        std::vector<double> logbuf((size_t)nblocks * tpbr * ra.log_stride, xt_emul_poison() ? NAN : 0.0);
        ra.gpartials = gp2.data();
        ra.log = logbuf.data();
        ra.TB = TB;
        const int nbuf = getenv("XT_EMUL_REV_NBUF") ? atoi(getenv("XT_EMUL_REV_NBUF")) : 2;  // the launcher picks 1 where two buffers leave no room for a second workgroup
        const size_t ldsd = xt_rev_lds_bytes(S, cfg.G, cfg.EP, D, K, tpbr, thr, nbuf) / 8;
        if (!emul_rev(cfg.G, D, K, nbuf, l.a, ra, nblocks, thr, ldsd)) return -3;
        std::vector<double> adj(TB, 0.0);
        for (int b = 0; b < nblocks; ++b) {
            out[0] += gp2[(size_t)b * (1 + TB)];
            for (int c = 0; c < TB; ++c) adj[c] += gp2[(size_t)b * (1 + TB) + 1 + c];
        }
        for (int i = 0; i < n_dir; ++i) {
            double s2 = 0.0;
            for (int c = 0; c < TB; ++c) s2 += adj[c] * dblob[(size_t)i * TB + c];
            out[1 + i] = s2;
        }
        return 0;
    }
    if (generic_g == 3 || generic_g == 4) {  // xt_gradr.h: state and tangents in registers, LDS exchange; generic_g = 3 / 4: that many directions per pass
        if (cfg.G < 2 || cfg.G > 4 || cfg.NG > 256) return -5;
        const int NPC = generic_g;
        const int tpbr = std::max(1, 256 / cfg.NG), thr = (tpbr * cfg.NG + 63) / 64 * 64;
        l.a.TPB = tpbr;
        for (int p0 = 0; p0 < n_dir; p0 += NPC) {
            const int NPp = std::min(NPC, n_dir - p0);
            std::vector<double> gp2((size_t)nblocks * (NPp + 1), 0.0);
            l.ga.dblob = dblob.data() + (size_t)p0 * TB;
            l.ga.gpartials = gp2.data();
            l.ga.NP = NPp;
            const size_t ldsd = xt_gradr_lds_bytes(S, cfg.G, cfg.E, cfg.EP, cfg.NG, cfg.P, D, K, NPp, tpbr) / 8;
            if (!emul_gradr(cfg.G, D, K, NPC, l.a, l.ga, nblocks, thr, ldsd)) return -3;
            for (int b = 0; b < nblocks; ++b) {
                if (p0 == 0) out[0] += gp2[(size_t)b * (NPp + 1)];
                for (int c = 0; c < NPp; ++c) out[1 + p0 + c] += gp2[(size_t)b * (NPp + 1) + 1 + c];
            }
        }
        return 0;
    }
    if (generic_g == 2) {  // register-resident 2-state path, passes of <= 8 directions
        if (!xt_use_reg2(S, NS, F) || locerr_mode != 0) return -5;
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < locerr_dims && k < 3; ++k) {
            lo = std::min(lo, m.locerr[k] * m.locerr[k]);
            hi = std::max(hi, m.locerr[k] * m.locerr[k]);
        }
        l.a.well_scaled = (xt_model_well_scaled(blob, lo, hi) && !getenv("XT_EMUL_GUARDED")) ? 1 : 0;
        if (n_dir == 0) {
            std::vector<double> part(nblocks, 0.0);
            l.a.partials = part.data();
            l.ga.NP = 0;
            if (!emul_r2(F, D, K, 0, 0, l.a, l.ga, nblocks)) return -3;
            for (int b = 0; b < nblocks; ++b) out[0] += part[b];
        }
        // like the product's launcher (extrack_grad.hip): uniform directions (xt_r2_uniform_direction) ride along with the first pass
        std::vector<int> full, uni;
        for (int i = 0; i < n_dir; ++i)
            ((int)uni.size() < XT_R2_MAXU && xt_r2_uniform_direction(dblob.data() + (size_t)i * TB) ? uni : full).push_back(i);
        if (full.empty() && !uni.empty()) {
            full.push_back(uni.back());
            uni.pop_back();
        }
        std::vector<double> ublob(std::max<size_t>(1, uni.size()) * TB, 0.0);
        for (size_t i = 0; i < uni.size(); ++i) memcpy(ublob.data() + i * TB, dblob.data() + (size_t)uni[i] * TB, TB * sizeof(double));
        const int NF = (int)full.size();
        for (int p0 = 0; p0 < NF;) {  // passes of 3 or 8 directions (what emul_r2.cpp instantiates), padded with zero directions
            const int NPp = NF - p0 > 3 ? 8 : 3, real = std::min(NPp, NF - p0), NU = p0 == 0 ? (int)uni.size() : 0, NPT = NPp + NU;
            std::vector<double> gp2((size_t)nblocks * (NPT + 1), 0.0), pad((size_t)NPp * TB, 0.0);
            for (int i = 0; i < real; ++i) memcpy(pad.data() + (size_t)i * TB, dblob.data() + (size_t)full[p0 + i] * TB, (size_t)TB * sizeof(double));
            l.ga.dblob = pad.data();
            l.ga.udblob = ublob.data();
            l.ga.NU = NU;
            l.ga.gpartials = gp2.data();
            l.ga.NP = NPp;
            if (!emul_r2(F, D, K, 0, NPp, l.a, l.ga, nblocks)) return -3;
            for (int b = 0; b < nblocks; ++b) {
                if (p0 == 0) out[0] += gp2[(size_t)b * (NPT + 1)];
                for (int c = 0; c < real; ++c) out[1 + full[p0 + c]] += gp2[(size_t)b * (NPT + 1) + 1 + c];
                for (int c = 0; c < NU; ++c) out[1 + uni[c]] += gp2[(size_t)b * (NPT + 1) + 1 + NPp + c];
            }
            p0 += real;
        }
        return 0;
    }
    bool ok;
    if (generic_g || cfg.G > 4) ok = emul_grad_dispatch_dk<0>(D, K, l);
    else if (cfg.G == 2) ok = emul_grad_dispatch_dk<2>(D, K, l);
    else if (cfg.G == 3) ok = emul_grad_dispatch_dk<3>(D, K, l);
    else ok = emul_grad_dispatch_dk<4>(D, K, l);
    if (!ok) return -3;
    for (int c = 0; c < n_dir + 1; ++c) {
        double s2 = 0.0;
        for (int b = 0; b < nblocks; ++b) s2 += gp[(size_t)b * (n_dir + 1) + c];
        out[c] = s2;
    }
    return 0;
}


// ---- state-duration histogram body (xt_hist.h): one bucket ---------------------------------------------------------------------
template <int D, int K>
static void emul_hist_run(const XtHistArgs& a, int nblocks, int threads, size_t lds_doubles)
{
    th_emul_blocks(nblocks, threads, lds_doubles, [&](HostCtx& cx) { xt_hist_body<D, K>(a, cx); });  // with per-wavefront barriers
}

extern "C" int xt_emul_hist(const double* tracks, const double* sigma, long long N, int L, int D, int KS, int S, int isBL, int min_l,
                            int locerr_mode, int locerr_dims, const double* locerr, double slope, double offset, double pBL, const double* ds,
                            const double* Fs, const double* TrMat, const double* p_stay, int max_nb_states, int nblocks, int threads, int par_lds,
                            double* hist /* [(L-1)][S] */)
{
    XtModelHost m{S, 1, locerr_dims, {0, 0, 0}, slope, offset, pBL, ds, Fs, TrMat, p_stay};
    for (int k = 0; k < 3; ++k) m.locerr[k] = locerr ? locerr[k < locerr_dims ? k : 0] : 0.0;
    std::vector<double> blob;
    xt_hist_build_blob(m, blob);
    const int K = locerr_mode == 0 ? locerr_dims : KS;
    XtHistArgs a;
    memset(&a, 0, sizeof(a));
    a.bits = S <= 2 ? 1 : (S <= 4 ? 2 : 3);
    a.HW = (L * a.bits + 63) / 64;
    if (a.HW > XT_HIST_MAXW) return -2;
    a.K = max_nb_states;
    a.PC = max_nb_states > S * S ? max_nb_states : S * S;
    a.NC = 1;
    while (a.NC < a.PC * S) a.NC <<= 1;
    a.tracks = tracks;
    a.sigma = locerr_mode ? sigma : nullptr;
    a.blob = blob.data();
    a.N = N;
    a.L = L;
    a.S = S;
    a.KS = KS;
    a.locerr_mode = locerr_mode;
    a.isBL = isBL;
    a.min_l = min_l;
    a.par_lds = par_lds;
    const int nbins = (L - 1) * S;
    std::vector<double> partials((size_t)nblocks * nbins, 0.0);
    a.partials = partials.data();
    a.ws_stride = 2 * (int64_t)xt_hist_parent_doubles(a.PC, D, K, a.HW);
    std::vector<double> ws(par_lds ? 1 : (size_t)a.ws_stride * nblocks, 0.0);
    a.ws = ws.data();
    const size_t ldsd = xt_hist_lds_doubles(S, L, D, K, locerr_mode ? KS : 0, a.PC, a.NC, a.HW, threads, par_lds != 0);
    if (D == 1 && K == 1) emul_hist_run<1, 1>(a, nblocks, threads, ldsd);
    else if (D == 2 && K == 1) emul_hist_run<2, 1>(a, nblocks, threads, ldsd);
    else if (D == 2 && K == 2) emul_hist_run<2, 2>(a, nblocks, threads, ldsd);
    else if (D == 3 && K == 1) emul_hist_run<3, 1>(a, nblocks, threads, ldsd);
    else if (D == 3 && K == 3) emul_hist_run<3, 3>(a, nblocks, threads, ldsd);
    else return -3;
    for (int i = 0; i < nbins; ++i) {
        double s2 = 0.0;
        for (int b = 0; b < nblocks; ++b) s2 += partials[(size_t)b * nbins + i];
        hist[i] = s2;
    }
    return 0;
}
