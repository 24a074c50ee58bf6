// libextrack_hip.so, translation unit 3: state-duration histograms (xt_hist.h) behind extrack_segment_len_hist.
#include "xt_host.h"

#include "xt_hist.h"
#include "xt_hist_host.h"

// Waves per SIMD asked of the register allocator (256-thread workgroups): unbounded the kernel takes 136 VGPRs = 3 workgroups per CU; measured r03
// (1e5 x 30, max_nb_states 500 / 120): 3 -> 68.2 / 24.8 ms, 4 -> 62.0 / 21.2, 5 -> 61.4 / 19.9, 6 -> 74.0 / 21.8.
#ifndef XT_HIST_WAVES
#define XT_HIST_WAVES 5
#endif
template <int D, int K, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT == 256 ? XT_HIST_WAVES : 1)) xt_hist_kernel(XtHistArgs a)
{
    DevCtx cx;
    xt_hist_body<D, K>(a, cx);
}

// Sum of the per-block histograms in a fixed order: one thread per bin.
__global__ void __launch_bounds__(256) xt_hist_reduce(const double* __restrict__ partials, int nblocks, int nbins, double* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nbins) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partials[(size_t)b * nbins + i];
    out[i] = s;
}

template <int D, int K>
static hipError_t xt_hist_launch(extrack_ctx* ctx, const XtHistArgs& a, int grid, size_t lds, int threads)
{
    const void* kp = threads > 256 ? (const void*)xt_hist_kernel<D, K, 512> : (const void*)xt_hist_kernel<D, K, 256>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    void* kargs[1] = {(void*)&a};
    hipError_t e = hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream);
    return e != hipSuccess ? e : hipGetLastError();
}

extern "C" int extrack_segment_len_hist(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, int32_t max_nb_states, double* hist)
{
    if (!ctx || !hist) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    if (m->nb_substeps != 1) return xt_fail(ctx, EXTRACK_E_INVALID, "state-duration histograms are defined for nb_substeps == 1");
    if (max_nb_states < 1) return xt_fail(ctx, EXTRACK_E_INVALID, "max_nb_states must be >= 1");
    const int S = m->n_states;
    if (S < 2 || S > XT_MAX_STATES) return xt_fail(ctx, EXTRACK_E_INVALID, "n_states must be in [2, 8]");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtBucket& b = ctx->buckets[bucket_id];
    const int D = b.D, L = b.L;
    int K;
    if (m->locerr_mode == 0) {
        K = m->locerr_dims;
        if (K != 1 && K != D) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1 or the track dimensionality");
    } else {
        if (!b.d_sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "per-peak localisation error mode but the bucket has no sigma");
        K = b.KS;
    }
    XtHistArgs a;
    memset(&a, 0, sizeof(a));
    a.bits = S <= 2 ? 1 : (S <= 4 ? 2 : 3);
    a.HW = (L * a.bits + 63) / 64;
    if (a.HW > XT_HIST_MAXW) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "track too long for the state-history words (len * bits per state <= 4096)");
    a.K = max_nb_states;
    a.PC = std::max(max_nb_states, S * S);
    if ((int64_t)a.PC * S > 16384) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "max_nb_states * n_states > 16384 is not built");
    a.NC = 1;
    while (a.NC < a.PC * S) a.NC <<= 1;
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    xt_hist_build_blob(mh, blob);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    a.tracks = b.d_tracks;
    a.sigma = m->locerr_mode ? b.d_sigma : nullptr;
    a.blob = ctx->d_blob;
    a.N = b.N;
    a.L = L;
    a.S = S;
    a.KS = b.KS ? b.KS : 1;
    a.locerr_mode = m->locerr_mode;
    a.isBL = (L != m->max_len) ? 1 : 0;
    a.min_l = m->min_len;
    const int KSl = m->locerr_mode ? a.KS : 0;
    // workgroup size 256; 512 threads (2 candidates per thread at max_nb_states 500) were measured SLOWER: 104 vs 79 ms per 1e5 x 30 - twice the
    // stages cross wavefronts (LDS + barrier) and a CU holds the same two tracks (EXTRACK_HIST_THREADS=512 keeps the variant reachable)
    int threads = 256;
    if (const char* ev = getenv("EXTRACK_HIST_THREADS")) threads = atoi(ev) == 512 ? 512 : 256;
    size_t lds = xt_hist_lds_doubles(S, L, D, K, KSl, a.PC, a.NC, a.HW, threads, true) * sizeof(double);
    // parent arrays (the surviving sequences of the previous position) in LDS, or in a per-workgroup region of global memory when that
    // lets more workgroups share a CU: at max_nb_states 500 the LDS copy leaves room for 2 (78 KB each), the global one for 5 - measured
    // r03, 1e5 x 30: 78.3 -> 68.1 ms (then 61.4 ms with the register bound below); at 120 the LDS copy already allows 8 and is the faster one (24.5 vs 26.3 ms)
    const size_t lds_g = xt_hist_lds_doubles(S, L, D, K, KSl, a.PC, a.NC, a.HW, threads, false) * sizeof(double);
    auto per_cu_of = [](size_t bytes) { return (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(bytes, 1)); };
    a.par_lds = (lds <= 150 * 1024 && std::min(XT_HIST_WAVES, per_cu_of(lds_g)) <= std::min(XT_HIST_WAVES, per_cu_of(lds))) ? 1 : 0;  // XT_HIST_WAVES workgroups per CU is what the registers allow
    if (const char* ev = getenv("EXTRACK_HIST_PAR_LDS")) a.par_lds = (lds <= 150 * 1024 && atoi(ev) != 0) ? 1 : 0;
    if (!a.par_lds) lds = lds_g;
    if (lds > 160 * 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "candidate arrays do not fit the 160 KiB LDS of a CU: lower max_nb_states");
    const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / lds)));
    const int grid = (int)std::min<int64_t>(b.N, (int64_t)ctx->n_cu * per_cu * 2);
    a.ws_stride = 2 * (int64_t)xt_hist_parent_doubles(a.PC, D, K, a.HW);
    if (!a.par_lds) {
        const size_t need = (size_t)a.ws_stride * grid * sizeof(double);
        if (need > ctx->th_ws_cap) {
            XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_th_ws) (void)hipFree(ctx->d_th_ws);
            ctx->d_th_ws = nullptr;
            ctx->th_ws_cap = 0;
            XT_HIP(ctx, hipMalloc(&ctx->d_th_ws, need));
            ctx->th_ws_cap = need;
        }
        a.ws = ctx->d_th_ws;
    }
    const int nbins = (L - 1) * S;
    if ((rc = xt_reserve_partials(ctx, (size_t)grid * nbins + nbins))) return rc;
    a.partials = ctx->d_partials;
    XT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipError_t e;
    if (D == 1 && K == 1) e = xt_hist_launch<1, 1>(ctx, a, grid, lds, threads);
    else if (D == 2 && K == 1) e = xt_hist_launch<2, 1>(ctx, a, grid, lds, threads);
    else if (D == 2 && K == 2) e = xt_hist_launch<2, 2>(ctx, a, grid, lds, threads);
    else if (D == 3 && K == 1) e = xt_hist_launch<3, 1>(ctx, a, grid, lds, threads);
    else if (D == 3 && K == 3) e = xt_hist_launch<3, 3>(ctx, a, grid, lds, threads);
    else return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "histogram kernel variant not built");
    if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("histogram kernel launch: ") + hipGetErrorString(e));
    XT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    double* d_out = ctx->d_partials + (size_t)grid * nbins;
    hipLaunchKernelGGL(xt_hist_reduce, dim3((nbins + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_partials, grid, nbins, d_out);
    XT_HIP(ctx, hipGetLastError());
    XT_HIP(ctx, hipMemcpyAsync(hist, d_out, (size_t)nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->launch_info[0] = grid;
    ctx->launch_info[1] = threads;
    ctx->launch_info[2] = (int32_t)lds;
    ctx->launch_info[3] = 1;
    ctx->launch_info[4] = per_cu;
    ctx->launch_info[5] = ctx->n_cu;
    return EXTRACK_OK;
}
