// libextrack_hip.so, translation unit: log-likelihood + exact gradient of the THRESHOLD-FUSION objective at the frozen plan of the
// evaluation (xt_thgrad.h) behind extrack_loglik_th_grad.  Replaces the finite-difference loop that lmfit's BFGS runs around
// cum_Proba_Cs -> P_Cs_inter_bound_stats_th (extrack/tracking.py:1371 -> :991 -> :427-743).
#include "xt_host.h"

#include "xt_grad_host.h"
#include "xt_thgrad.h"
#include "xt_thgrad2.h"

// one lane per track; state in a per-wavefront region of global memory (coalesced rows, served by L2 / Infinity Cache): the kernel is bound
// by memory latency - what counts is the number of wavefronts a CU holds (registers: XT_THG_WAVES per SIMD; LDS: the accumulator rows)
#ifndef XT_THG_WAVES
#define XT_THG_WAVES 3
#endif
template <int D, int K, bool RG>
__global__ void __launch_bounds__(256, RG ? XT_THG_WAVES : 1) xt_thg_kernel(XtThArgs a, XtThGradArgs ga)  // LDS rows: the LDS bounds the wavefronts, the allocator is left alone (2 states: 23.9 ms unbounded, 27.9 ms at 3 waves)
{
    DevCtx cx;
    xt_thg_body<D, K, RG>(a, ga, cx);
}
// second mapping (xt_thgrad2.h): lanes = (sequence, track), live state in LDS, one log record per (step, merged sequence, track)
#ifndef XT_THG2_WAVES
#define XT_THG2_WAVES 2
#endif
template <int D, int K, int NE>
__global__ void __launch_bounds__(256, XT_THG2_WAVES) xt_thg2_kernel(XtThArgs a, XtThGradArgs ga)
{
    DevCtx cx;
    xt_thg2_body<D, K, NE>(a, ga, cx);
}
template <int D, int K>
static const void* xt_thg2_kernel_ptr(int ne) { return ne <= 1 ? (const void*)xt_thg2_kernel<D, K, 1> : (const void*)xt_thg2_kernel<D, K, 4>; }

template <int D, int K>
static const void* xt_thg_kernel_ptr(bool rg) { return rg ? (const void*)xt_thg_kernel<D, K, true> : (const void*)xt_thg_kernel<D, K, false>; }

static int xt_thg_reserve(extrack_ctx* ctx, double** buf, size_t* cap, size_t n)
{
    if (n <= *cap) return EXTRACK_OK;
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    XT_HIP(ctx, hipMalloc(buf, n * sizeof(double)));
    *cap = n;
    return EXTRACK_OK;
}

// Enqueues plan + gradient kernels of one evaluation; d_out (device, 1 + n_dir doubles) receives {sum LL, d sum LL / d theta_i}.  The plan
// stage reads the chunks' sequence counts back (as extrack_loglik_th does); everything after it is stream-ordered.
static int xt_th_grad_enqueue(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk, int32_t n_dir,
                              const extrack_model_tangent* tangents, double* d_out)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    for (int i = 0; i < n_dir; ++i)
        if (!tangents[i].ds2 || !tangents[i].Fs || !tangents[i].TrMat || !tangents[i].p_stay) return xt_fail(ctx, EXTRACK_E_INVALID, "null tangent field");
    const int S = m->n_states, NS = m->nb_substeps;
    if (S < 2 || S > XT_MAX_STATES || NS < 1 || NS > 4) return xt_fail(ctx, EXTRACK_E_INVALID, "n_states must be in [2, 8], nb_substeps in [1, 4]");
    int G = 1;
    for (int i = 0; i < NS; ++i) G *= S;
    const int TB = xt_grad_tb_doubles(S, G);
    // one wavefront's accumulator rows must leave room for at least two workgroups per CU
    if ((size_t)xt_thg_lds_doubles(S, G, 1, XT_TH_MAXCAP) * sizeof(double) > 96 * 1024)
        return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "frozen-plan gradient: n_states^(nb_substeps+1) table adjoints do not fit the LDS (use finite differences)");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    // tangent blocks of the directions (threshold-fusion table layout) -> device
    XtModelHost mh;
    xt_model_host(m, mh);
    const size_t ndbl = (size_t)std::max(n_dir, 1) * TB;
    if (ctx->dblob_busy) {
        XT_HIP(ctx, hipEventSynchronize(ctx->ev_dblob));
        ctx->dblob_busy = false;
    }
    if (ndbl > ctx->h_dblob_cap) {
        if (ctx->h_dblob) (void)hipHostFree(ctx->h_dblob);
        ctx->h_dblob = nullptr;
        ctx->h_dblob_cap = 0;
        XT_HIP(ctx, hipHostMalloc((void**)&ctx->h_dblob, ndbl * sizeof(double)));
        ctx->h_dblob_cap = ndbl;
    }
    if (!ctx->ev_dblob) XT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_dblob, hipEventDisableTiming));
    for (int i = 0; i < n_dir; ++i) xt_th_build_tangent_block(mh, tangents[i], m->locerr_mode, ctx->h_dblob + (size_t)i * TB);
    if ((rc = xt_thg_reserve(ctx, &ctx->d_dblob, &ctx->dblob_cap, ndbl))) return rc;
    XT_HIP(ctx, hipMemcpyAsync(ctx->d_dblob, ctx->h_dblob, ndbl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    XT_HIP(ctx, hipEventRecord(ctx->ev_dblob, ctx->stream));
    ctx->dblob_busy = true;
    if ((rc = xt_thg_reserve(ctx, &ctx->d_revadj, &ctx->revadj_cap, (size_t)TB))) return rc;
    if (!ctx->evg0) XT_HIP(ctx, hipEventCreate(&ctx->evg0));
    if (!ctx->evg1) XT_HIP(ctx, hipEventCreate(&ctx->evg1));
    XT_HIP(ctx, hipEventRecord(ctx->evg0, ctx->stream));

    // per-wavefront output rows of all launch groups, one after the other
    size_t rows = 0;
    const XtThAfterPlan cb = [&](XtThArgs& a, int D, int K, int maxG, int Lmax) -> int {
        XtThGradArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.TB = TB;
        ga.capP = std::max(maxG, S);
        // ---- the (sequence, track)-lane kernel (xt_thgrad2.h) for models with few live sequences.  Measured (MI355X, 1e6 tracks): 2 states x 30
        // positions, 12 live sequences: 15.0 ms against 23.9 ms of the one-lane-per-track kernel below (256 threads, 64 tracks per tile, 4 lanes
        // per track; 128:32 16.6, 256:32 18.0, 64:16 20.6 ms); 3 states, lengths 5 - 50, ~33 live sequences: 52.8 ms at best (128 threads, 8
        // tracks) against 51 ms - the tile's LDS (14 doubles per sequence and track) leaves 6 wavefronts per CU, the kernel below wins or ties.
        // EXTRACK_THG_KERNEL=1 | 2 forces one, EXTRACK_THG2_THREADS / _TT / _LDS_KB set the tile
        {
            int want = ga.capP <= 16 ? 2 : 1;
            if (const char* ev = getenv("EXTRACK_THG_KERNEL")) want = atoi(ev);
            int NT = 256;
            if (const char* ev = getenv("EXTRACK_THG2_THREADS")) NT = atoi(ev) == 64 ? 64 : (atoi(ev) == 128 ? 128 : 256);
            int lpt = 4;
            while (lpt < S) lpt *= 2;  // few lanes per track (every lane walks several sequences): what the sweep above favours
            int TT = std::max(2, std::min(64, NT / lpt));
            if (const char* ev = getenv("EXTRACK_THG2_TT")) TT = std::max(2, std::min(64, atoi(ev)));
            while (TT & (TT - 1)) TT &= TT - 1;
            auto lds2 = [&](int tt) { return (size_t)xt_thg2_lds_doubles(S, G, ga.capP, a.capE, tt, D, K, NT, TB) * sizeof(double); };
            size_t lds_cap = 80 * 1024;  // two workgroups per CU
            if (const char* ev = getenv("EXTRACK_THG2_LDS_KB")) lds_cap = (size_t)std::max(16, std::min(160, atoi(ev))) * 1024;
            while (TT > 2 && lds2(TT) > lds_cap) TT >>= 1;
            const int LPT = NT / TT;
            if (want == 2 && lds2(TT) <= lds_cap && LPT >= S && S * G <= 4 * LPT && a.capE <= 4096) {
                const size_t lds = lds2(TT);
                a.TT = TT;
                a.logTT = 0;
                while ((1 << a.logTT) < TT) ++a.logTT;
                ga.ws_stride = xt_thg2_ws_doubles(ga.capP, Lmax, TT, D, K, G);
                const int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)XT_THG2_WAVES * 256 / NT, (160 * 1024) / lds));
                const int64_t tpc = ((int64_t)a.chunk + TT - 1) / TT;  // tiles per chunk
                int64_t target = (int64_t)ctx->n_cu * blocks_per_cu * 2;
                int64_t bpc = std::max<int64_t>(1, std::min<int64_t>((target + a.nchunks - 1) / a.nchunks, tpc));
                a.bpc = (int32_t)bpc;
                const int grid = (int)(a.nchunks * bpc);
                int rc2;
                if ((rc2 = xt_thg_reserve(ctx, &ctx->d_revlog, &ctx->revlog_cap, (size_t)grid * (size_t)ga.ws_stride))) return rc2;
                const size_t need = (rows + (size_t)grid) * (size_t)(1 + TB);
                if (need > ctx->gpartials_cap) {
                    double* nw = nullptr;
                    XT_HIP(ctx, hipMalloc(&nw, need * 2 * sizeof(double)));
                    if (ctx->d_gpartials) {
                        if (rows) XT_HIP(ctx, hipMemcpyAsync(nw, ctx->d_gpartials, rows * (size_t)(1 + TB) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        (void)hipFree(ctx->d_gpartials);
                    }
                    ctx->d_gpartials = nw;
                    ctx->gpartials_cap = need * 2;
                }
                ga.ws = ctx->d_revlog;
                ga.gpartials = ctx->d_gpartials + rows * (size_t)(1 + TB);
                const int ne = S * G <= LPT ? 1 : 4;
                const void* kp = nullptr;
                if (D == 1 && K == 1) kp = xt_thg2_kernel_ptr<1, 1>(ne);
                else if (D == 2 && K == 1) kp = xt_thg2_kernel_ptr<2, 1>(ne);
                else if (D == 2 && K == 2) kp = xt_thg2_kernel_ptr<2, 2>(ne);
                else if (D == 3 && K == 1) kp = xt_thg2_kernel_ptr<3, 1>(ne);
                else if (D == 3 && K == 3) kp = xt_thg2_kernel_ptr<3, 3>(ne);
                else return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "frozen-plan gradient: track / error dimensionality not built");
                if (lds > 64 * 1024) XT_HIP(ctx, hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                void* kargs[2] = {(void*)&a, (void*)&ga};
                XT_HIP(ctx, hipLaunchKernel(kp, dim3(grid), dim3(NT), kargs, lds, ctx->stream));
                XT_HIP(ctx, hipGetLastError());
                rows += (size_t)grid;
                ctx->launch_info[0] = grid;
                ctx->launch_info[1] = NT;
                ctx->launch_info[2] = (int32_t)lds;
                ctx->launch_info[3] = TT;
                ctx->launch_info[4] = blocks_per_cu;
                ctx->launch_info[5] = ctx->n_cu;
                if (getenv("EXTRACK_TH_DEBUG"))
                    fprintf(stderr, "[thgrad2] chunks %d maxG %d Lmax %d | threads %d TT %d lanes/track %d lds %zu bpc %d grid %d log/block %.2f MB total %.1f MB\n", a.nchunks, maxG, Lmax,
                            NT, TT, LPT, lds, a.bpc, grid, ga.ws_stride * 8.0 / 1048576.0, grid * ga.ws_stride * 8.0 / 1048576.0);
                return EXTRACK_OK;
            }
        }
        // wavefronts per workgroup: the accumulator rows of a wavefront decide how many wavefronts a CU holds
        const size_t row_bytes = (size_t)xt_thg_rows(S, G) * 64 * sizeof(double);
        int rows_global = row_bytes > 16 * 1024 ? 1 : 0;
        if (const char* ev = getenv("EXTRACK_THG_ROWS_GLOBAL")) rows_global = atoi(ev) != 0;
        ga.rows_global = rows_global;
        ga.ws_stride = xt_thg_ws_doubles(ga.capP, Lmax, D, K, rows_global ? xt_thg_rows(S, G) : 0);
        // measured (MI355X, r04): rows in scratch for 3 states 56 -> 48 ms per 1e6 mixed-length tracks (5 -> 12 wavefronts per CU); for 2 states the
        // LDS rows (14 KB per wavefront, 10 wavefronts per CU) are faster (23.9 vs 29 ms per 1e6 x 30)
        int NW = rows_global ? 2 : (row_bytes <= 8 * 1024 ? 4 : (row_bytes <= 16 * 1024 ? 2 : 1));
        if (const char* ev = getenv("EXTRACK_THG_NW")) NW = std::max(1, std::min(4, atoi(ev)));
        const size_t lds = (size_t)xt_thg_lds_doubles(S, G, NW, a.capE, rows_global != 0) * sizeof(double);
        const int threads = 64 * NW;
        int blocks_per_cu = (int)std::min<size_t>(16, (160 * 1024) / lds);
        blocks_per_cu = std::max(1, std::min(blocks_per_cu, 32 / NW));  // at most 8 wavefronts per SIMD
        const int64_t tpc = ((int64_t)a.chunk + 63) / 64;                 // tiles per chunk
        // scratch budget: every wavefront of the launch owns a region (EXTRACK_THG_WS_MB, default 48 GiB of the 288)
        size_t budget_mb = 48 * 1024;
        if (const char* ev = getenv("EXTRACK_THG_WS_MB")) budget_mb = (size_t)std::max(64, atoi(ev));
        const int64_t max_waves = std::max<int64_t>(1, (int64_t)((budget_mb << 20) / ((size_t)ga.ws_stride * sizeof(double))));
        int64_t target = (int64_t)ctx->n_cu * blocks_per_cu * 2;
        target = std::min(target, std::max<int64_t>(1, max_waves / NW));
        int64_t bpc = std::max<int64_t>(1, std::min<int64_t>((target + a.nchunks - 1) / a.nchunks, (tpc + NW - 1) / NW));
        if ((int64_t)a.nchunks * bpc * NW > max_waves && a.nchunks > max_waves / NW)
            return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "frozen-plan gradient: the state log of one wavefront per chunk exceeds the scratch budget (EXTRACK_THG_WS_MB)");
        a.bpc = (int32_t)bpc;
        const int grid = (int)(a.nchunks * bpc);
        const size_t nw_total = (size_t)grid * NW;
        int rc2;
        if ((rc2 = xt_thg_reserve(ctx, &ctx->d_revlog, &ctx->revlog_cap, nw_total * (size_t)ga.ws_stride))) return rc2;
        // d_gpartials may still hold the rows of an earlier group of this evaluation: grow by copy
        const size_t need = (rows + nw_total) * (size_t)(1 + TB);
        if (need > ctx->gpartials_cap) {
            double* nw = nullptr;
            XT_HIP(ctx, hipMalloc(&nw, need * 2 * sizeof(double)));
            if (ctx->d_gpartials) {
                if (rows) XT_HIP(ctx, hipMemcpyAsync(nw, ctx->d_gpartials, rows * (size_t)(1 + TB) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                (void)hipFree(ctx->d_gpartials);
            }
            ctx->d_gpartials = nw;
            ctx->gpartials_cap = need * 2;
        }
        ga.ws = ctx->d_revlog;
        ga.gpartials = ctx->d_gpartials + rows * (size_t)(1 + TB);
        const void* kp = nullptr;
        if (D == 1 && K == 1) kp = xt_thg_kernel_ptr<1, 1>(rows_global != 0);
        else if (D == 2 && K == 1) kp = xt_thg_kernel_ptr<2, 1>(rows_global != 0);
        else if (D == 2 && K == 2) kp = xt_thg_kernel_ptr<2, 2>(rows_global != 0);
        else if (D == 3 && K == 1) kp = xt_thg_kernel_ptr<3, 1>(rows_global != 0);
        else if (D == 3 && K == 3) kp = xt_thg_kernel_ptr<3, 3>(rows_global != 0);
        else return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "frozen-plan gradient: track / error dimensionality not built");
        if (lds > 64 * 1024) XT_HIP(ctx, hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        void* kargs[2] = {(void*)&a, (void*)&ga};
        XT_HIP(ctx, hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream));
        XT_HIP(ctx, hipGetLastError());
        rows += nw_total;
        ctx->launch_info[0] = grid;
        ctx->launch_info[1] = threads;
        ctx->launch_info[2] = (int32_t)lds;
        ctx->launch_info[3] = 64;
        ctx->launch_info[4] = blocks_per_cu;
        ctx->launch_info[5] = ctx->n_cu;
        if (getenv("EXTRACK_TH_DEBUG"))
            fprintf(stderr, "[thgrad] chunks %d maxG %d Lmax %d | NW %d lds %zu bpc %d grid %d ws/wave %.2f MB total %.1f MB\n", a.nchunks, maxG, Lmax, NW, lds,
                    a.bpc, grid, ga.ws_stride * 8.0 / 1048576.0, nw_total * ga.ws_stride * 8.0 / 1048576.0);
        return EXTRACK_OK;
    };
    if ((rc = xt_th_plan_groups(ctx, m, threshold, max_nb_states, chunk, cb))) return rc;
    // fixed-order sums over the wavefronts: column 0 = sum LL, columns 1 .. TB = adjoint of the model blob; then < adjoint, tangent block >
    xt_grad_reduce_launch(ctx->stream, ctx->d_gpartials, (int)rows, TB + 1, d_out, ctx->d_revadj);
    XT_HIP(ctx, hipGetLastError());
    if (n_dir > 0) {
        xt_rev_project(ctx->stream, ctx->d_revadj, ctx->d_dblob, TB, n_dir, d_out + 1);
        XT_HIP(ctx, hipGetLastError());
    }
    XT_HIP(ctx, hipEventRecord(ctx->evg1, ctx->stream));
    ctx->grad_timed = true;
    return EXTRACK_OK;
}

extern "C" int extrack_loglik_th_grad_async(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk,
                                            int32_t n_dir, const extrack_model_tangent* tangents, double* d_out)
{
    if (!ctx || !d_out || n_dir < 0 || (n_dir > 0 && !tangents)) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    return xt_th_grad_enqueue(ctx, m, threshold, max_nb_states, chunk, n_dir, tangents, d_out);
}

extern "C" int extrack_loglik_th_grad(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk,
                                      int32_t n_dir, const extrack_model_tangent* tangents, double* total_ll, double* grad)
{
    if (!ctx || !total_ll || n_dir < 0 || (n_dir > 0 && (!tangents || !grad))) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_thg_reserve(ctx, &ctx->d_gout, &ctx->gout_cap, (size_t)n_dir + 1);
    if (rc) return rc;
    if ((rc = xt_th_grad_enqueue(ctx, m, threshold, max_nb_states, chunk, n_dir, tangents, ctx->d_gout))) return rc;
    std::vector<double> host((size_t)n_dir + 1);
    XT_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_gout, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_ll = host[0];
    for (int i = 0; i < n_dir; ++i) grad[i] = host[1 + i];
    return EXTRACK_OK;
}
