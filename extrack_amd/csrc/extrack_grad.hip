// libextrack_hip.so, translation unit 2: log-likelihood + exact gradient (xt_grad.h) behind extrack_loglik_grad.
// Replaces the finite-difference loop that lmfit's BFGS runs around cum_Proba_Cs (extrack/tracking.py:1371).
#include "xt_host.h"

#include "xt_grad.h"
#include "xt_grad_host.h"
#include "xt_reg2.h"
#include "xt_gradr.h"
#include "xt_rev.h"

// Waves per SIMD the register allocator is asked to allow.  Measured on C2 (1e6 x 30, 7 directions, PJ = 4): 2 -> 63 ms,
// 3 -> 53 ms (168 VGPRs, 108 B of scratch per lane), 4 -> 60 ms (128 VGPRs, 272 B of scratch).  Three members per group (C3, 13 directions): 3 waves (232 B of
// scratch) 78.8 ms, 2 waves (248 VGPRs, no scratch) 91.5 ms - the spills are not what bounds this kernel (r03).
#ifndef XT_GRAD_WAVES
#define XT_GRAD_WAVES 3
#endif
template <int G_, int D, int K, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT == 256 ? XT_GRAD_WAVES : 1)) xt_grad_kernel(XtKernelArgs a, XtGradArgs ga)
{
    DevCtx cx;
    xt_grad_body<G_, D, K>(a, ga, cx);
}

// Column sums of the per-block partials [nrows][ncol] in a fixed order: one workgroup per column.  Column 0 (sum LL) goes to ll_dst
// (nullptr: dropped - every pass recomputes it, only the first one reports it), column 1 + i to out[dst.idx[i]] (the launch may have
// served the directions in another order than the caller's).
struct XtGradDst {
    int32_t idx[16];
};
__global__ void __launch_bounds__(256) xt_grad_reduce(const double* __restrict__ partials, int nrows, int ncol, double* __restrict__ ll_dst,
                                                      double* __restrict__ out, XtGradDst dst)
{
    __shared__ double sh[256];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nrows; i += 256) s += partials[(int64_t)i * ncol + col];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (col == 0) {
            if (ll_dst) *ll_dst = sh[0];
        } else {
            out[col - 1 < 16 ? dst.idx[col - 1] : col - 1] = sh[0];
        }
    }
}
// out[i] += add[i]: the launch groups after the first leave {sum LL, gradient} in a scratch row that is added to the evaluation's result
__global__ void __launch_bounds__(64) xt_grad_add_kernel(double* __restrict__ out, const double* __restrict__ add, int n)
{
    for (int i = threadIdx.x; i < n; i += 64) out[i] += add[i];
}
static XtGradDst xt_grad_dst_identity(int base)
{
    XtGradDst d;
    for (int i = 0; i < 16; ++i) d.idx[i] = base + i;
    return d;
}

void xt_grad_reduce_launch(hipStream_t st, const double* partials, int nrows, int ncol, double* ll_dst, double* out)
{
    XtGradDst d;
    for (int i = 0; i < 16; ++i) d.idx[i] = i;
    hipLaunchKernelGGL(xt_grad_reduce, dim3(ncol), dim3(256), 0, st, partials, nrows, ncol, ll_dst, out, d);
}

struct GradLauncher {
    extrack_ctx* ctx;
    XtKernelArgs a;
    XtGradArgs ga;
    int threads = 0;
    size_t lds = 0;
    int grid = 0;
    hipError_t herr = hipSuccess;

    template <int G_, int D, int K>
    bool run()
    {
        if (threads <= 256) return launch(xt_grad_kernel<G_, D, K, 256>);
        return launch(xt_grad_kernel<G_, D, K, 1024>);
    }
    template <class KernT>
    bool launch(KernT kern)
    {
        if (lds > 64 * 1024) {
            herr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (herr != hipSuccess) return true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, a, ga);
        herr = hipGetLastError();
        return true;
    }
};

template <int GG, class L>
static bool xt_grad_dispatch_dk(int D, int K, L& l)
{
    if (D == 1 && K == 1) return l.template run<GG, 1, 1>();
    if (D == 2 && K == 1) return l.template run<GG, 2, 1>();
    if (D == 2 && K == 2) return l.template run<GG, 2, 2>();
    if (D == 3 && K == 1) return l.template run<GG, 3, 1>();
    if (D == 3 && K == 3) return l.template run<GG, 3, 3>();
    return false;
}

template <class L>
static bool xt_grad_dispatch(int G, int D, int K, L& l)
{
    if (G == 2) return xt_grad_dispatch_dk<2>(D, K, l);
    if (G == 3) return xt_grad_dispatch_dk<3>(D, K, l);
    if (G == 4) return xt_grad_dispatch_dk<4>(D, K, l);
    return xt_grad_dispatch_dk<0>(D, K, l);
}

// LDS bytes of a block of tpb tracks with NP directions
static size_t xt_grad_lds_bytes(const XtConfig& c, int D, int K, int NP, int tpb, bool tan_lds)
{
    size_t d = (size_t)((xt_tab_doubles(c.S, c.G) + 1) & ~1);
    if (tan_lds) d += (size_t)((NP * xt_grad_tb_doubles(c.S, c.G) + 1) & ~1);
    d += (size_t)tpb * ((size_t)xt_grad_region_doubles(c.EP, D, K, NP) + xt_grad_acc_doubles(NP, c.NG) + xt_stage_doubles(D));
    return d * sizeof(double);
}

static int xt_grad_reserve(extrack_ctx* ctx, double** buf, size_t* cap, size_t n)
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

// Enqueues the kernels of one likelihood + gradient evaluation on the context's stream; d_out (device, 1 + n_dir doubles) receives
// {sum LL, d sum LL / d theta_i}.  Nothing waits for the device: passes and launch groups accumulate in stream order.
static int xt_grad_enqueue(extrack_ctx* ctx, const extrack_model* m, int32_t n_dir, const extrack_model_tangent* tangents, double* d_out)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (ctx->buckets.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, "no bucket uploaded");
    for (int i = 0; i < n_dir; ++i)
        if (!tangents[i].ds2 || !tangents[i].Fs || !tangents[i].TrMat || !tangents[i].p_stay)
            return xt_fail(ctx, EXTRACK_E_INVALID, "null tangent field");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = xt_prepare_config(ctx, m))) return rc;
    const XtConfig& c = ctx->cfg;
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    xt_build_blob(mh, c, blob);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    const int TB = xt_grad_tb_doubles(c.S, c.G);
    // tangent tables: built in a pinned staging buffer the asynchronous copy reads from; an event guards its reuse by the next call
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
    for (int i = 0; i < n_dir; ++i) xt_build_tangent_block(mh, tangents[i], c, m->locerr_mode, ctx->h_dblob + (size_t)i * TB);
    if ((rc = xt_grad_reserve(ctx, &ctx->d_dblob, &ctx->dblob_cap, ndbl))) return rc;
    XT_HIP(ctx, hipMemcpyAsync(ctx->d_dblob, ctx->h_dblob, ndbl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    XT_HIP(ctx, hipEventRecord(ctx->ev_dblob, ctx->stream));
    ctx->dblob_busy = true;

    // launch groups: buckets with the same (dims, sigma dims), longest first
    std::vector<XtBucket*> order;
    for (auto& b : ctx->buckets) order.push_back(&b);
    std::stable_sort(order.begin(), order.end(), [](const XtBucket* x, const XtBucket* y) {
        if (x->D != y->D) return x->D < y->D;
        if (x->KS != y->KS) return x->KS < y->KS;
        return x->L > y->L;
    });
    std::vector<std::vector<XtBucket*>> groups;
    for (XtBucket* b : order) {
        if (groups.empty() || groups.back().size() >= XT_MAX_BUCKETS || groups.back()[0]->D != b->D || groups.back()[0]->KS != b->KS)
            groups.emplace_back();
        groups.back().push_back(b);
    }
    if (order.size() > (size_t)XT_DESC_CAP / 2) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "too many buckets");

    // more than one launch group (more than XT_MAX_BUCKETS track lengths, or buckets of different layouts): the groups after the first write
    // {sum LL, gradient} to a scratch row that is added to d_out in stream order
    double* const d_out_final = d_out;
    if (groups.size() > 1 && (rc = xt_grad_reserve(ctx, &ctx->d_gtmp, &ctx->gtmp_cap, (size_t)n_dir + 1))) return rc;
    int group_index = 0;
    size_t doff = xt_desc_base(ctx);
    // per-block partial sums of every pass of the evaluation, one after the other (a pass has at most 32 blocks per CU)
    size_t poff = 0;
    if ((rc = xt_grad_reserve(ctx, &ctx->d_gpartials, &ctx->gpartials_cap, ((size_t)ctx->n_cu * 32 * 4 + XT_MAX_BUCKETS) * ((size_t)n_dir + 16)))) return rc;
    if (!ctx->evg0) XT_HIP(ctx, hipEventCreate(&ctx->evg0));
    if (!ctx->evg1) XT_HIP(ctx, hipEventCreate(&ctx->evg1));
    XT_HIP(ctx, hipEventRecord(ctx->evg0, ctx->stream));
    for (auto& g : groups) {
        double* const d_out = group_index == 0 ? d_out_final : ctx->d_gtmp;  // (shadows the parameter inside the loop)
        poff = 0;  // the previous group's reductions precede this group's kernels in the stream: the partial-sum rows are free again
        auto group_done = [&]() {
            if (group_index > 0) hipLaunchKernelGGL(xt_grad_add_kernel, dim3(1), dim3(64), 0, ctx->stream, d_out_final, ctx->d_gtmp, n_dir + 1);
            ++group_index;
        };
        const XtBucket& b0 = *g[0];
        const int D = b0.D;
        int K;
        if (m->locerr_mode == 0) {
            K = m->locerr_dims;
            if (K != 1 && K != D) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1 or the track dimensionality");
        } else {
            if (!b0.d_sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "per-peak localisation error mode but the bucket has no sigma");
            K = b0.KS;
        }
        // directions per pass: as many as keep one track's state within the LDS of a CU (all of them for the usual models)
        int npass_dir = std::max(n_dir, 1);
        while (npass_dir > 1 && (npass_dir > 16 || xt_grad_lds_bytes(c, D, K, npass_dir, 1, false) > 150 * 1024)) npass_dir = (npass_dir + 1) / 2;
        // bucket descriptors of this group (shared by its passes)
        std::vector<XtBucketDesc> descs;
        for (XtBucket* b : g) {
            XtBucketDesc d;
            d.tracks = b->d_tracks;
            d.sigma = m->locerr_mode ? b->d_sigma : nullptr;
            d.ll_out = nullptr;
            d.preds_out = nullptr;
            d.N = b->N;
            d.L = b->L;
            d.isBL = (b->L != m->max_len) ? 1 : 0;  // tracking.py:1037-1040
            d.ll_const = -(double)(b->L - 1) * D * 0.5 * XT_LOG2PI;
            descs.push_back(d);
        }
        ctx->desc_shadow.clear();  // this path writes the device table itself: the likelihood launcher's shadow of it no longer holds
        memcpy(ctx->h_desc + doff, descs.data(), descs.size() * sizeof(XtBucketDesc));
        XT_HIP(ctx, hipMemcpyAsync(ctx->d_desc + doff, ctx->h_desc + doff, descs.size() * sizeof(XtBucketDesc), hipMemcpyHostToDevice, ctx->stream));
        XT_HIP(ctx, hipEventRecord(ctx->ev_blob[(ctx->blob_turn - 1u) & 1u], ctx->stream));
        // ---- two-state models: register-resident kernels (xt_reg2.h), <= 8 directions per pass, tangents in VGPRs
        const bool r2 = ctx->grad_reg2 == 1 && xt_use_reg2(c.S, c.NS, c.F) && m->locerr_mode == 0 && n_dir > 0 && xt_r2_kernel(c.F, D, K, 1) != nullptr;
        // ---- reverse mode (xt_rev.h): one forward + one backward sweep whatever the number of directions; the adjoint of the model blob
        // is contracted with the tangent blocks by a small kernel.  3 / 4 members per group by default (r03: C3, 13 directions)
        {
            const int tpb = std::max(1, 256 / c.NG), threads = (tpb * c.NG + 63) / 64 * 64;
            // one exchange buffer (two barriers per step) where two do not leave room for a second workgroup on the CU
            const size_t lds2 = xt_rev_lds_bytes(c.S, c.G, c.EP, D, K, tpb, threads, 2), lds1 = xt_rev_lds_bytes(c.S, c.G, c.EP, D, K, tpb, threads, 1);
            const int nbuf = (2 * lds2 > 160 * 1024 && 2 * lds1 <= 160 * 1024) ? 1 : 2;
            const size_t lds = nbuf == 1 ? lds1 : lds2;
            const void* kp = n_dir > 0 && xt_rev_supported(c.G, c.NG) ? xt_rev_kernel_ptr(c.G, D, K, nbuf) : nullptr;
            // the merged-state logs (one region per track slot of every block) must fit the budget with at least one block per two CUs:
            // very long tracks go to the forward-mode kernels instead
            int Lmax0 = 2;
            for (auto& d : descs) Lmax0 = std::max(Lmax0, (int)d.L);
            const size_t slot_doubles = (size_t)tpb * std::max(Lmax0 - 2, 1) * xt_rev_step_doubles(c.NG, D, K);
            const size_t max_blocks = (ctx->rev_log_mb << 20) / (slot_doubles * sizeof(double));
            const bool use_rev = kp && lds <= 160 * 1024 && max_blocks >= (size_t)ctx->n_cu / 2 &&
                                 (ctx->grad_rev == 2 || (ctx->grad_rev == 1 && ctx->grad_reg2 == 1 && !r2));
            if (use_rev) {
                if (lds > 64 * 1024) XT_HIP(ctx, hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                auto key = std::make_pair(kp, std::make_pair(threads, lds));
                auto it = ctx->occ_cache.find(key);
                if (it == ctx->occ_cache.end()) {
                    int o = 0;
                    XT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kp, threads, lds));
                    it = ctx->occ_cache.emplace(key, o < 1 ? 1 : o).first;
                }
                const int occ = it->second;
                XtKernelArgs a;
                memset(&a, 0, sizeof(a));
                xt_fill_args_from_config(c, a);
                XtRevArgs ra;
                memset(&ra, 0, sizeof(ra));
                const double target = std::min((double)occ * ctx->n_cu * ctx->rev_oversub, (double)max_blocks);
                double wsum = 0.0;
                int Lmax = 2;
                std::vector<int64_t> nbatch(descs.size());
                for (size_t i = 0; i < descs.size(); ++i) {
                    nbatch[i] = (descs[i].N + tpb - 1) / tpb;
                    wsum += (double)nbatch[i] * (descs[i].L - 1);
                    Lmax = std::max(Lmax, (int)descs[i].L);
                }
                int64_t acc = 0;
                for (size_t i = 0; i < descs.size(); ++i) {
                    int64_t n = (int64_t)ceil(target * ((double)nbatch[i] * (descs[i].L - 1)) / wsum);
                    n = n < 1 ? 1 : (n > nbatch[i] ? nbatch[i] : n);
                    acc += n;
                    a.blk_end[i] = (int32_t)acc;
                }
                const int grid = (int)acc;
                ra.TB = TB;
                ra.log_stride = (int64_t)std::max(Lmax - 2, 1) * xt_rev_step_doubles(c.NG, D, K);
                if ((rc = xt_grad_reserve(ctx, &ctx->d_revlog, &ctx->revlog_cap, (size_t)grid * tpb * (size_t)ra.log_stride))) return rc;
                if ((rc = xt_grad_reserve(ctx, &ctx->d_revadj, &ctx->revadj_cap, (size_t)TB))) return rc;
                if ((rc = xt_grad_reserve(ctx, &ctx->d_gpartials, &ctx->gpartials_cap, poff + (size_t)grid * (TB + 1)))) return rc;
                a.desc = ctx->d_desc + doff;
                a.ndesc = (int32_t)descs.size();
                a.blob = ctx->d_blob;
                a.TPB = tpb;
                a.min_len = m->min_len;
                a.locerr_mode = m->locerr_mode;
                a.KS = b0.KS ? b0.KS : 1;
                ra.gpartials = ctx->d_gpartials + poff;
                ra.log = ctx->d_revlog;
                void* kargs[2] = {(void*)&a, (void*)&ra};
                XT_HIP(ctx, hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream));
                hipLaunchKernelGGL(xt_grad_reduce, dim3(TB + 1), dim3(256), 0, ctx->stream, ctx->d_gpartials + poff, grid, TB + 1, d_out, ctx->d_revadj,
                                   xt_grad_dst_identity(0));
                XT_HIP(ctx, hipGetLastError());
                xt_rev_project(ctx->stream, ctx->d_revadj, ctx->d_dblob, TB, n_dir, d_out + 1);
                XT_HIP(ctx, hipGetLastError());
                poff += (size_t)grid * (TB + 1);
                ctx->launch_info[0] = grid;
                ctx->launch_info[1] = threads;
                ctx->launch_info[2] = (int32_t)lds;
                ctx->launch_info[3] = tpb;
                ctx->launch_info[4] = occ;
                ctx->launch_info[5] = ctx->n_cu;
                doff += g.size();
                group_done();
                continue;
            }
        }
        if (r2) {
            const int tpw = 64 >> (c.F - 1), tpb = tpw * XT_F2_WAVES, threads = 64 * XT_F2_WAVES;
            // "uniform" directions (xt_r2_uniform_direction, e.g. pBL) cost no per-step work: they ride along with the first pass
            std::vector<int> full, uni;
            for (int i = 0; i < n_dir; ++i)
                ((int)uni.size() < XT_R2_MAXU && xt_r2_uniform_direction(ctx->h_dblob + (size_t)i * TB) ? uni : full).push_back(i);
            if (full.empty()) {
                full.push_back(uni.back());
                uni.pop_back();
            }
            // device copy of the tangent blocks in launch order: full directions first, then the uniform ones
            const int NF = (int)full.size(), NUn = (int)uni.size();
            if ((rc = xt_grad_reserve(ctx, &ctx->d_dblob2, &ctx->dblob2_cap, (size_t)n_dir * TB))) return rc;
            for (int i = 0; i < n_dir; ++i) {
                const int src = i < NF ? full[i] : uni[i - NF];
                XT_HIP(ctx, hipMemcpyAsync(ctx->d_dblob2 + (size_t)i * TB, ctx->d_dblob + (size_t)src * TB, (size_t)TB * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            }
            static const int r2_maxnp = [] {
                const char* e = getenv("EXTRACK_R2_MAXNP");
                const int v = e ? atoi(e) : 8;
                return v >= 1 && v <= 8 ? v : 8;
            }();
            const int npass = (NF + r2_maxnp - 1) / r2_maxnp, per = (NF + npass - 1) / npass;
            double lo = INFINITY, hi = -INFINITY;
            for (int k = 0; k < m->locerr_dims && k < 3; ++k) {
                lo = std::min(lo, m->locerr[k] * m->locerr[k]);
                hi = std::max(hi, m->locerr[k] * m->locerr[k]);
            }
            for (int p0 = 0; p0 < NF; p0 += per) {
                const int NP = std::min(per, NF - p0);
                const int NU = p0 == 0 ? NUn : 0, NPT = NP + NU;
                const void* kp = xt_r2_kernel(c.F, D, K, NP);
                if (!kp) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "gradient kernel variant not built");
                XtKernelArgs a;
                memset(&a, 0, sizeof(a));
                xt_fill_args_from_config(c, a);
                XtGradArgs ga;
                memset(&ga, 0, sizeof(ga));
                const size_t lds = (size_t)xt_r2_block_bytes(NPT, D, 0, tpw);
                auto key = std::make_pair(kp, std::make_pair(threads, lds));
                auto it = ctx->occ_cache.find(key);
                if (it == ctx->occ_cache.end()) {
                    int o = 0;
                    XT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kp, threads, lds));
                    it = ctx->occ_cache.emplace(key, o < 1 ? 1 : o).first;
                }
                const int occ = it->second;
                const double target = (double)occ * ctx->n_cu * ctx->oversub;
                double wsum = 0.0;
                std::vector<int64_t> nbatch(descs.size());
                for (size_t i = 0; i < descs.size(); ++i) {
                    nbatch[i] = (descs[i].N + tpb - 1) / tpb;
                    wsum += (double)nbatch[i] * (descs[i].L - 1);
                }
                int64_t acc = 0;
                for (size_t i = 0; i < descs.size(); ++i) {
                    int64_t n = (int64_t)ceil(target * ((double)nbatch[i] * (descs[i].L - 1)) / wsum);
                    n = n < 1 ? 1 : (n > nbatch[i] ? nbatch[i] : n);
                    acc += n;
                    a.blk_end[i] = (int32_t)acc;
                }
                const int grid = (int)acc;
                if (poff + (size_t)grid * (NPT + 1) > ctx->gpartials_cap) return xt_fail(ctx, EXTRACK_E_HIP, "gradient partial-sum buffer too small");
                a.desc = ctx->d_desc + doff;
                a.ndesc = (int32_t)descs.size();
                a.blob = ctx->d_blob;
                a.TPB = tpb;
                a.min_len = m->min_len;
                a.locerr_mode = 0;
                a.KS = 1;
                a.well_scaled = xt_model_well_scaled(blob, lo, hi) ? 1 : 0;
                ga.dblob = ctx->d_dblob2 + (size_t)p0 * TB;
                ga.gpartials = ctx->d_gpartials + poff;
                ga.NP = NP;
                ga.TB = TB;
                ga.NU = NU;
                ga.udblob = ctx->d_dblob2 + (size_t)NF * TB;
                void* kargs[2] = {(void*)&a, (void*)&ga};
                XT_HIP(ctx, hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream));
                XtGradDst dst = xt_grad_dst_identity(0);  // column 1 + i of this launch -> the caller's direction index
                for (int i = 0; i < NP; ++i) dst.idx[i] = full[p0 + i];
                for (int i = 0; i < NU; ++i) dst.idx[NP + i] = uni[i];
                hipLaunchKernelGGL(xt_grad_reduce, dim3(NPT + 1), dim3(256), 0, ctx->stream, ctx->d_gpartials + poff, grid, NPT + 1,
                                   p0 == 0 ? d_out : nullptr, d_out + 1, dst);
                XT_HIP(ctx, hipGetLastError());
                poff += (size_t)grid * (NPT + 1);
                ctx->launch_info[0] = grid;
                ctx->launch_info[1] = threads;
                ctx->launch_info[2] = (int32_t)lds;
                ctx->launch_info[3] = tpb;
                ctx->launch_info[4] = occ;
                ctx->launch_info[5] = ctx->n_cu;
            }
            doff += g.size();
            group_done();
            continue;
        }
        // ---- 2 - 4 members per group, <= 256 groups per track: state and tangents in registers, LDS as the exchange medium (xt_gradr.h)
        // Measured against the LDS-resident kernel below (r03): C3 (3 states, 13 directions) frame_len 6 601 ms vs 1 960 ms, frame_len 4 63 vs 79 ms;
        // C2-type data through the general kernels (2 states with per-peak errors; in the launcher's order reg2 -> rev -> gradr -> lds the reverse-mode
        // kernels above now take those models first) frame_len 6 43.8 vs 52.9 ms, frame_len 4 16.1 vs 16.4 ms.
        if (ctx->grad_reg2 && n_dir > 0 && c.G >= 2 && c.G <= 4 && c.NG <= 256 && xt_gradr_kernel_ptr(c.G, D, K, 4) != nullptr) {
            const int tpb = std::max(1, 256 / c.NG), threads = (tpb * c.NG + 63) / 64 * 64;
            // 4 directions per pass: with 6 the register allocator spills inside the step loop (3 states: 917 GB of scratch traffic per C3
            // launch, r03 PMC) and the pass count saved does not pay for it; 3 per pass when that needs no more passes
            int NPC = ctx->gradr_npc ? ctx->gradr_npc : 4;
            const int npass = (n_dir + NPC - 1) / NPC, per = (n_dir + npass - 1) / npass;
            if (per <= 3 && !ctx->gradr_npc) NPC = 3;
            const void* kp = xt_gradr_kernel_ptr(c.G, D, K, NPC);
            const size_t lds = xt_gradr_lds_bytes(c.S, c.G, c.E, c.EP, c.NG, c.P, D, K, per, tpb);
            if (kp && lds <= 160 * 1024) {
                if (lds > 64 * 1024) XT_HIP(ctx, hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                auto key = std::make_pair(kp, std::make_pair(threads, lds));
                auto it = ctx->occ_cache.find(key);
                if (it == ctx->occ_cache.end()) {
                    int o = 0;
                    XT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kp, threads, lds));
                    it = ctx->occ_cache.emplace(key, o < 1 ? 1 : o).first;
                }
                const int occ = it->second;
                for (int p0 = 0; p0 < n_dir; p0 += per) {
                    const int NP = std::min(per, n_dir - p0);
                    XtKernelArgs a;
                    memset(&a, 0, sizeof(a));
                    xt_fill_args_from_config(c, a);
                    XtGradArgs ga;
                    memset(&ga, 0, sizeof(ga));
                    const double target = (double)occ * ctx->n_cu * 4;
                    double wsum = 0.0;
                    std::vector<int64_t> nbatch(descs.size());
                    for (size_t i = 0; i < descs.size(); ++i) {
                        nbatch[i] = (descs[i].N + tpb - 1) / tpb;
                        wsum += (double)nbatch[i] * (descs[i].L - 1);
                    }
                    int64_t acc = 0;
                    for (size_t i = 0; i < descs.size(); ++i) {
                        int64_t n = (int64_t)ceil(target * ((double)nbatch[i] * (descs[i].L - 1)) / wsum);
                        n = n < 1 ? 1 : (n > nbatch[i] ? nbatch[i] : n);
                        acc += n;
                        a.blk_end[i] = (int32_t)acc;
                    }
                    const int grid = (int)acc;
                    if (poff + (size_t)grid * (NP + 1) > ctx->gpartials_cap) return xt_fail(ctx, EXTRACK_E_HIP, "gradient partial-sum buffer too small");
                    a.desc = ctx->d_desc + doff;
                    a.ndesc = (int32_t)descs.size();
                    a.blob = ctx->d_blob;
                    a.base_tab = ctx->d_base_tab;
                    a.off_tab = ctx->d_off_tab;
                    a.TPB = tpb;
                    a.min_len = m->min_len;
                    a.locerr_mode = m->locerr_mode;
                    a.KS = b0.KS ? b0.KS : 1;
                    ga.dblob = ctx->d_dblob + (size_t)p0 * TB;
                    ga.gpartials = ctx->d_gpartials + poff;
                    ga.NP = NP;
                    ga.TB = TB;
                    void* kargs[2] = {(void*)&a, (void*)&ga};
                    XT_HIP(ctx, hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream));
                    hipLaunchKernelGGL(xt_grad_reduce, dim3(NP + 1), dim3(256), 0, ctx->stream, ctx->d_gpartials + poff, grid, NP + 1,
                                       p0 == 0 ? d_out : nullptr, d_out + 1, xt_grad_dst_identity(p0));
                    XT_HIP(ctx, hipGetLastError());
                    poff += (size_t)grid * (NP + 1);
                    ctx->launch_info[0] = grid;
                    ctx->launch_info[1] = threads;
                    ctx->launch_info[2] = (int32_t)lds;
                    ctx->launch_info[3] = tpb;
                    ctx->launch_info[4] = occ;
                    ctx->launch_info[5] = ctx->n_cu;
                }
                doff += g.size();
                group_done();
                continue;
            }
        }
        // ---- everything else: the LDS-resident kernel (xt_grad.h)
        if (xt_grad_lds_bytes(c, D, K, std::min(npass_dir, std::max(n_dir, 0)), 1, false) > 160 * 1024)
            return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "sequence state with one tangent direction does not fit the 160 KiB LDS of a CU");
        for (int p0 = 0; p0 < std::max(n_dir, 1); p0 += npass_dir) {
            const int NP = n_dir == 0 ? 0 : std::min(npass_dir, n_dir - p0);
            GradLauncher l;
            l.ctx = ctx;
            memset(&l.a, 0, sizeof(l.a));
            xt_fill_args_from_config(c, l.a);
            const bool tan_lds = (size_t)NP * TB * 8 <= 16 * 1024;
            const size_t per_track = xt_grad_lds_bytes(c, D, K, NP, 1, tan_lds) - xt_grad_lds_bytes(c, D, K, NP, 0, tan_lds);
            const size_t fixed = xt_grad_lds_bytes(c, D, K, NP, 0, tan_lds);
            const size_t budget = 64 * 1024;
            // PJ lanes per group: about two directions per lane, as long as a track's threads fit a workgroup
            int PJ = 1;
            while (PJ < 8 && PJ * 2 <= NP && NP > 2 * PJ - 1 && c.NG * PJ * 2 <= 1024) PJ *= 2;
            if (const char* ev = getenv("EXTRACK_GRAD_PJ")) {
                const int v = atoi(ev);
                if ((v == 1 || v == 2 || v == 4 || v == 8) && c.NG * v <= 1024) PJ = v;
            }
            const int NT = c.NG * PJ;
            const int by_threads = NT >= 256 ? 1 : 256 / NT;
            const int by_lds = budget > fixed + per_track ? (int)((budget - fixed) / per_track) : 1;
            int tpb = std::max(1, std::min(by_threads, by_lds));
            const int threads = (tpb * NT + 63) / 64 * 64;
            if (threads > 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "n_states^(frame_len-nb_substeps) > 1024 groups per track is not built");
            l.threads = threads;
            l.lds = xt_grad_lds_bytes(c, D, K, NP, tpb, tan_lds);
            if (l.lds > 160 * 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "sequence state does not fit the 160 KiB LDS of a CU");
            // grid: blocks per bucket in proportion to its work, CUs oversubscribed (as the likelihood launcher does)
            const int occ = std::max(1, std::min((int)((160 * 1024) / l.lds), 2048 / threads));
            const double target = (double)occ * ctx->n_cu * 4;
            double wsum = 0.0;
            std::vector<int64_t> nbatch(descs.size());
            for (size_t i = 0; i < descs.size(); ++i) {
                nbatch[i] = (descs[i].N + tpb - 1) / tpb;
                wsum += (double)nbatch[i] * (descs[i].L - 1);
            }
            int64_t acc = 0;
            for (size_t i = 0; i < descs.size(); ++i) {
                int64_t n = (int64_t)ceil(target * ((double)nbatch[i] * (descs[i].L - 1)) / wsum);
                n = n < 1 ? 1 : (n > nbatch[i] ? nbatch[i] : n);
                acc += n;
                l.a.blk_end[i] = (int32_t)acc;
            }
            l.grid = (int)acc;
            if (poff + (size_t)l.grid * (NP + 1) > ctx->gpartials_cap) return xt_fail(ctx, EXTRACK_E_HIP, "gradient partial-sum buffer too small");
            l.a.desc = ctx->d_desc + doff;
            l.a.ndesc = (int32_t)descs.size();
            l.a.blob = ctx->d_blob;
            l.a.base_tab = ctx->d_base_tab;
            l.a.off_tab = ctx->d_off_tab;
            l.a.TPB = tpb;
            l.a.min_len = m->min_len;
            l.a.locerr_mode = m->locerr_mode;
            l.a.KS = b0.KS ? b0.KS : 1;
            l.ga.dblob = ctx->d_dblob + (size_t)p0 * TB;
            l.ga.gpartials = ctx->d_gpartials + poff;
            l.ga.NP = NP;
            l.ga.TB = TB;
            l.ga.tan_lds = tan_lds ? 1 : 0;
            l.ga.PJ = PJ;
            if (!xt_grad_dispatch(c.G, D, K, l)) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "gradient kernel variant not built");
            if (l.herr != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("gradient kernel launch: ") + hipGetErrorString(l.herr));
            if (NP > 16) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "more than 16 directions per pass");
            hipLaunchKernelGGL(xt_grad_reduce, dim3(NP + 1), dim3(256), 0, ctx->stream, ctx->d_gpartials + poff, l.grid, NP + 1,
                               p0 == 0 ? d_out : nullptr, d_out + 1, xt_grad_dst_identity(p0));
            XT_HIP(ctx, hipGetLastError());
            poff += (size_t)l.grid * (NP + 1);
            ctx->launch_info[0] = l.grid;
            ctx->launch_info[1] = threads;
            ctx->launch_info[2] = (int32_t)l.lds;
            ctx->launch_info[3] = tpb;
            ctx->launch_info[4] = occ;
            ctx->launch_info[5] = ctx->n_cu;
        }
        doff += g.size();
        group_done();
    }
    XT_HIP(ctx, hipGetLastError());
    XT_HIP(ctx, hipEventRecord(ctx->evg1, ctx->stream));
    ctx->grad_timed = true;
    return EXTRACK_OK;
}

extern "C" int extrack_loglik_grad_async(extrack_ctx* ctx, const extrack_model* m, int32_t n_dir, const extrack_model_tangent* tangents,
                                         double* d_out)
{
    if (!ctx || !d_out || n_dir < 0 || (n_dir > 0 && !tangents)) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    if (n_dir == 0) return extrack_loglik_async(ctx, m, d_out);
    return xt_grad_enqueue(ctx, m, n_dir, tangents, d_out);
}

extern "C" int extrack_loglik_grad(extrack_ctx* ctx, const extrack_model* m, int32_t n_dir, const extrack_model_tangent* tangents,
                                   double* total_ll, double* grad)
{
    if (!ctx || !total_ll || n_dir < 0 || (n_dir > 0 && (!tangents || !grad))) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    if (n_dir == 0) return extrack_loglik(ctx, m, total_ll, nullptr);  // no direction: the plain likelihood kernels
    int rc = xt_grad_reserve(ctx, &ctx->d_gout, &ctx->gout_cap, (size_t)n_dir + 1);
    if (rc) return rc;
    if ((rc = xt_grad_enqueue(ctx, m, n_dir, tangents, ctx->d_gout))) return rc;
    std::vector<double> host((size_t)n_dir + 1);
    XT_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_gout, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_ll = host[0];
    for (int i = 0; i < n_dir; ++i) grad[i] = host[1 + i];
    return EXTRACK_OK;
}

extern "C" int extrack_last_grad_ms(extrack_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return EXTRACK_E_INVALID;
    if (ctx->grad_timed) {
        XT_HIP(ctx, hipEventSynchronize(ctx->evg1));
        XT_HIP(ctx, hipEventElapsedTime(&ctx->grad_ms, ctx->evg0, ctx->evg1));
        ctx->grad_timed = false;
    }
    *ms = ctx->grad_ms;
    return EXTRACK_OK;
}
