// libextrack_hip.so - HIP kernels (gfx950) + C ABI for ExTrack's track-likelihood hot path.
// See include/extrack_hip.h for the contract and xt_kernel.h for the algorithm/data layout.
#include "xt_host.h"

#include "xt_dispatch.h"
#include "xt_seqmat.h"
#include "xt_entry.h"
#include "xt_fast2.h"
#include "xt_reg2.h"
#include "xt_big.h"

// Waves per SIMD the register allocator is asked to allow (workgroups of 256 threads).  Likelihood kernels: 4, except 4 members per group
// (4 states: 66 ms unbounded against 74 ms at 3 on the 5e5 x 60 set, frame_len 5).  Posterior kernels (226 VGPRs unbounded = 2 waves):
// 3 waves (168 VGPRs, 27-60 spilled dwords) - measured r03: C5 (4 states, 5e5 x 60, frame_len 5) 318 -> 265 ms, 4 waves 365 ms;
// 2 states 1e6 x 30 frame_len 6 18.3 -> 15.2 ms; 3 states 2e5 x 30 frame_len 6 49.6 -> 44.2 ms.  3-state likelihood at 5 waves: frame_len 6 unchanged,
// frame_len 4 3.06 -> 3.94 ms (kept at 4).  Entry-parallel kernel (94 VGPRs unbounded = 5 waves): C5 (nb_substeps 3) 48.6 ms, 6 waves 46.9, 7 waves 46.2.
#ifndef XT_ENTRY_WAVES
#define XT_ENTRY_WAVES 7
#endif
#ifndef XT_LL_WAVES
#define XT_LL_WAVES 4
#endif
#ifndef XT_G4_WAVES
#define XT_G4_WAVES 1
#endif
#ifndef XT_PREDS_WAVES
#define XT_PREDS_WAVES 3
#endif
template <int G_, int D, int K, bool PREDS, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT == 256 && !PREDS ? (G_ != 4 ? XT_LL_WAVES : XT_G4_WAVES) : (MAXT == 256 && PREDS ? XT_PREDS_WAVES : 1))) xt_track_kernel(XtKernelArgs a)
{
    DevCtx cx;
    xt_track_body<G_, D, K, PREDS>(a, cx);
    if (!PREDS) xt_fused_total(a);
}

template <int F, int D, int K>
__global__ void __launch_bounds__(64 * XT_F2_WAVES) xt_ll_s2_kernel(XtKernelArgs a)
{
    DevCtx cx;
    xt_ll_s2_body<F, D, K>(a, cx);
    xt_fused_total(a);
}

template <int GP, int D, int K, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT == 256 ? XT_ENTRY_WAVES : 1)) xt_entry_kernel(XtKernelArgs a)
{
    DevCtx cx;
    xt_entry_body<GP, D, K>(a, cx);
    xt_fused_total(a);
}

// Models whose sequence state does not fit a workgroup: one lane per track, state in global memory (xt_big.h)
template <int D, int K, bool PREDS>
__global__ void __launch_bounds__(256) xt_big_kernel(XtKernelArgs a, XtBigArgs ba)
{
    DevCtx cx;
    xt_big_body<D, K, PREDS>(a, ba, cx);
    if (!PREDS) xt_fused_total(a);
}
template <int D, int K>
static const void* xt_big_kernel_ptr(bool preds) { return preds ? (const void*)xt_big_kernel<D, K, true> : (const void*)xt_big_kernel<D, K, false>; }
static const void* xt_big_kernel_dk(int D, int K, bool preds)
{
    if (D == 1 && K == 1) return xt_big_kernel_ptr<1, 1>(preds);
    if (D == 2 && K == 1) return xt_big_kernel_ptr<2, 1>(preds);
    if (D == 2 && K == 2) return xt_big_kernel_ptr<2, 2>(preds);
    if (D == 3 && K == 1) return xt_big_kernel_ptr<3, 1>(preds);
    if (D == 3 && K == 3) return xt_big_kernel_ptr<3, 3>(preds);
    return nullptr;
}

// Posterior / recording mode (PREDS) is launched with 64 or 256 threads per chunk: bounded by 256 threads, PW waves per SIMD asked of the register
// allocator (the fit-mode plan walks with up to 1024 threads: 128 VGPRs).  Measured r03 (kernel ms; 2 states 2e5 x 30 | 4 states 5e4 x 60, nb_max 1):
// 3 waves 77.6 | 433, 4 waves 99.0 | 354, 5 waves 93.3 | 382, 6 waves 87.5 | 365, 8 waves 97.2 | 446 -> 3 for two states (168 VGPRs, no spills), else 4.
static inline int xt_th_pred_waves(int S) { return S == 2 ? 3 : 4; }
template <int D, int K, bool PREDS, int WS = -1, int PW = 4>
__global__ void __launch_bounds__(PREDS ? 256 : 1024, PREDS ? PW : 1) xt_th_plan_kernel(XtThArgs a)
{
    DevCtx cx;
    xt_th_plan_body<D, K, PREDS, WS>(a, cx);
}

// The wave-uniform two-buffer variant runs workgroups of up to 16 wavefronts, two per CU when the LDS allows: 24 wavefronts per CU need 6 per
// SIMD, i.e. at most 80 VGPRs - the allocator takes 77.  Round 4 lost that twice (2 states x 30, 1e6 tracks, evaluation 2.65 -> 3.8 ms): a
// branch with a log() compiled into every variant (now the SEQ instantiation) and two more non-inline constants in the exponential (reverted).
// Asking for 6 waves per SIMD through the launch bound instead (79 VGPRs, no spill) changes the schedule: 2.92 ms - no bound here.
template <int D, int K, bool UNI, bool SINGLE, bool DT, bool SEQ = false>
__global__ void __launch_bounds__(1024) xt_th_apply_kernel(XtThArgs a)
{
    DevCtx cx;
    xt_th_apply_body<D, K, UNI, SINGLE, DT, SEQ>(a, cx);
}

// Fixed-order reduction of the per-block partial sums (deterministic for a given launch geometry).
__global__ void __launch_bounds__(256) xt_reduce_partials(const double* __restrict__ partials, int n, double* __restrict__ out)
{
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

static std::string g_create_err;

int xt_fail(extrack_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" int extrack_abi_version(void) { return EXTRACK_ABI_VERSION; }

extern "C" const char* extrack_last_error(const extrack_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int extrack_create(int device_id, extrack_ctx** out)
{
    if (!out) return EXTRACK_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device: ") + hipGetErrorString(e);
        return EXTRACK_E_NODEVICE;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_err = "device id out of range";
        return EXTRACK_E_INVALID;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        g_create_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return EXTRACK_E_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return EXTRACK_E_NODEVICE;
    }
    extrack_ctx* c = new extrack_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    if (const char* ev = getenv("EXTRACK_OVERSUB")) {
        int v = atoi(ev);
        if (v >= 1 && v <= 64) {
            c->oversub = v;
            c->oversub_forced = true;
        }
    }
    if (const char* ev = getenv("EXTRACK_LL_PATH")) c->ll_reg2 = strcmp(ev, "reg2") == 0 ? 1 : (strcmp(ev, "lds") == 0 ? 0 : c->ll_reg2);
    if (const char* ev = getenv("EXTRACK_GRAD_PATH")) c->grad_reg2 = strcmp(ev, "lds") == 0 ? 0 : (strcmp(ev, "gradr") == 0 ? 2 : 1);
    if (const char* ev = getenv("EXTRACK_GRAD_PATH")) c->grad_rev = strcmp(ev, "rev") == 0 ? 2 : (strcmp(ev, "auto") == 0 ? 1 : 0);
    if (const char* ev = getenv("EXTRACK_REV_OVERSUB")) c->rev_oversub = std::max(1, atoi(ev));
    if (const char* ev = getenv("EXTRACK_REV_LOG_MB")) c->rev_log_mb = (size_t)std::max(1, atoi(ev));
    if (const char* ev = getenv("EXTRACK_GRADR_NPC")) c->gradr_npc = atoi(ev) == 4 ? 4 : (atoi(ev) == 3 ? 3 : 0);
    if (const char* ev = getenv("EXTRACK_TH_TT")) {
        int v = atoi(ev);
        if (v >= 1 && v <= 256 && (v & (v - 1)) == 0) c->th_force_tt = v;
    }
    if (const char* ev = getenv("EXTRACK_TH_THREADS")) {
        int v = atoi(ev);
        if (v >= 64 && v <= 1024 && v % 64 == 0) c->th_force_threads = v;
    }
    if (const char* ev = getenv("EXTRACK_TH_PLAN_THREADS")) {
        int v = atoi(ev);
        if (v >= 64 && v <= 1024 && v % 64 == 0) {
            c->th_plan_threads = v;
            c->th_plan_threads_forced = true;
        }
    }
    if (const char* ev = getenv("EXTRACK_TH_PLAN_BS")) c->th_plan_bs = atoi(ev);
    if (const char* ev = getenv("EXTRACK_TH_NO_SPLIT")) c->th_no_split = atoi(ev) != 0;
    if (const char* ev = getenv("EXTRACK_TH_SPLIT_PCT")) {
        int hi = 0, lo = 0;
        const int n = sscanf(ev, "%d,%d", &hi, &lo);
        if (n >= 1 && hi > 0 && hi < 100) {
            c->th_split_pct[0] = hi;
            c->th_split_pct[1] = (n == 2 && lo > 0 && lo < hi) ? lo : 0;
        }
    }
    if (const char* ev = getenv("EXTRACK_TH_STAGE_LDS")) c->th_stage_in_lds_mode = atoi(ev) != 0;
    if (const char* ev = getenv("EXTRACK_TH_NO_GEN_SINGLE")) c->th_no_gen_single = atoi(ev) != 0;
    if (const char* ev = getenv("EXTRACK_TH_PAIR_LANES")) c->th_pair_lanes = atoi(ev);
    if (const char* ev = getenv("EXTRACK_TH_SINGLE")) c->th_force_single = atoi(ev) != 0;
    if (const char* ev = getenv("EXTRACK_TH_OVERSUB")) {
        int v = atoi(ev);
        if (v >= 1 && v <= 64) c->th_oversub = v;
    }
#define XT_CREATE(call)                                                             \
    if ((e = (call)) != hipSuccess) {                                               \
        g_create_err = std::string(#call) + ": " + hipGetErrorString(e);            \
        delete c;                                                                   \
        return EXTRACK_E_HIP;                                                       \
    }
    XT_CREATE(hipSetDevice(device_id));
    XT_CREATE(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    XT_CREATE(hipEventCreate(&c->ev0));
    XT_CREATE(hipEventCreate(&c->ev1));
    XT_CREATE(hipEventCreateWithFlags(&c->ev_blob[0], hipEventDisableTiming));
    XT_CREATE(hipEventCreateWithFlags(&c->ev_blob[1], hipEventDisableTiming));
    XT_CREATE(hipMalloc(&c->d_total, sizeof(double)));
    XT_CREATE(hipHostMalloc(&c->h_total, sizeof(double), hipHostMallocMapped));
    XT_CREATE(hipHostGetDevicePointer((void**)&c->h_total_dev, c->h_total, 0));
    XT_CREATE(hipMalloc(&c->d_done, sizeof(unsigned int)));
    XT_CREATE(hipMemset(c->d_done, 0, sizeof(unsigned int)));
    if (const char* ev = getenv("EXTRACK_NO_FUSED")) c->no_fused = atoi(ev) != 0;
    XT_CREATE(hipMalloc(&c->d_desc, XT_DESC_CAP * sizeof(XtBucketDesc)));
    XT_CREATE(hipHostMalloc(&c->h_desc, XT_DESC_CAP * sizeof(XtBucketDesc), hipHostMallocDefault));
#undef XT_CREATE
    *out = c;
    return EXTRACK_OK;
}

static void xt_free_bucket(XtBucket& b)
{
    if (b.owned) {
        if (b.d_tracks) (void)hipFree((void*)b.d_tracks);
        if (b.d_sigma) (void)hipFree((void*)b.d_sigma);
    }
    if (b.d_ll) (void)hipFree(b.d_ll);
    if (b.d_dt) (void)hipFree(b.d_dt);
    if (b.th_members) (void)hipFree(b.th_members);
    if (b.th_mpack) (void)hipFree(b.th_mpack);
    if (b.th_gnew) (void)hipFree(b.th_gnew);
    if (b.th_gstart) (void)hipFree(b.th_gstart);
    if (b.th_hdr) (void)hipFree(b.th_hdr);
    if (b.th_status) (void)hipFree(b.th_status);
    b = XtBucket();
}

extern "C" int extrack_clear_buckets(extrack_ctx* ctx)
{
    if (!ctx) return EXTRACK_E_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->buckets) xt_free_bucket(b);
    ctx->buckets.clear();
    return EXTRACK_OK;
}

extern "C" void extrack_destroy(extrack_ctx* ctx)
{
    if (!ctx) return;
    extrack_clear_buckets(ctx);
    if (ctx->d_base_tab) (void)hipFree(ctx->d_base_tab);
    if (ctx->d_off_tab) (void)hipFree(ctx->d_off_tab);
    for (int i = 0; i < extrack_ctx::TH_SLOTS; ++i) {
        if (i == ctx->th_cur_slot) continue;  // the current set lives in the fields freed below
        extrack_ctx::ThSlot& sl = ctx->th_slot[i];
        if (sl.d_ws) (void)hipFree(sl.d_ws);
        if (sl.h_status) (void)hipHostFree(sl.h_status);
        if (sl.d_status) (void)hipFree(sl.d_status);
        if (sl.d_desc) (void)hipFree(sl.d_desc);
        if (sl.d_cend) (void)hipFree(sl.d_cend);
    }
    for (int i = 0; i < extrack_ctx::TH_SLOTS; ++i)
        if (ctx->th_streams[i]) (void)hipStreamDestroy(ctx->th_streams[i]);
    for (int i = 0; i < extrack_ctx::TH_SLOTS + 1; ++i)
        if (ctx->th_ev[i]) (void)hipEventDestroy(ctx->th_ev[i]);
    if (ctx->d_th_ws) (void)hipFree(ctx->d_th_ws);
    if (ctx->h_th_status) (void)hipHostFree(ctx->h_th_status);
    if (ctx->d_th_status) (void)hipFree(ctx->d_th_status);
    if (ctx->d_th_desc) (void)hipFree(ctx->d_th_desc);
    if (ctx->d_th_cend) (void)hipFree(ctx->d_th_cend);
    for (int i = 0; i < 2; ++i) {
        if (ctx->d_blob_s[i]) (void)hipFree(ctx->d_blob_s[i]);
        if (ctx->h_blob_s[i]) (void)hipHostFree(ctx->h_blob_s[i]);
        if (ctx->ev_blob[i]) (void)hipEventDestroy(ctx->ev_blob[i]);
    }
    if (ctx->d_preds) (void)hipFree(ctx->d_preds);
    if (ctx->d_dblob) (void)hipFree(ctx->d_dblob);
    if (ctx->h_dblob) (void)hipHostFree(ctx->h_dblob);
    if (ctx->d_dblob2) (void)hipFree(ctx->d_dblob2);
    if (ctx->ev_dblob) (void)hipEventDestroy(ctx->ev_dblob);
    if (ctx->d_gout) (void)hipFree(ctx->d_gout);
    if (ctx->d_gtmp) (void)hipFree(ctx->d_gtmp);
    if (ctx->d_revlog) (void)hipFree(ctx->d_revlog);
    if (ctx->d_revadj) (void)hipFree(ctx->d_revadj);
    for (int i = 0; i < extrack_ctx::RF_SLOTS; ++i)
        if (ctx->rf_buf[i]) (void)hipFree(ctx->rf_buf[i]);
    if (ctx->evg0) (void)hipEventDestroy(ctx->evg0);
    if (ctx->evg1) (void)hipEventDestroy(ctx->evg1);
    if (ctx->d_th_blobs) (void)hipFree(ctx->d_th_blobs);
    if (ctx->d_gpartials) (void)hipFree(ctx->d_gpartials);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_total) (void)hipFree(ctx->d_total);
    if (ctx->d_big_ws) (void)hipFree(ctx->d_big_ws);
    if (ctx->d_done) (void)hipFree(ctx->d_done);
    if (ctx->h_total) (void)hipHostFree(ctx->h_total);
    if (ctx->d_desc) (void)hipFree(ctx->d_desc);
    if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int extrack_set_stream(extrack_ctx* ctx, void* hip_stream)
{
    if (!ctx) return EXTRACK_E_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return EXTRACK_OK;
}

extern "C" int extrack_bucket_count(const extrack_ctx* ctx) { return ctx ? (int)ctx->buckets.size() : EXTRACK_E_INVALID; }

static int xt_check_bucket_shape(extrack_ctx* ctx, int64_t n, int32_t len, int32_t dims, const void* sigma, int32_t sigma_dims)
{
    if (n <= 0) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket must hold at least one track");
    if (len < 2) return xt_fail(ctx, EXTRACK_E_INVALID, "minimal track length = 2");  // tracking.py:149-150
    if (dims < 1 || dims > XT_MAX_DIMS) return xt_fail(ctx, EXTRACK_E_INVALID, "dims must be 1, 2 or 3");
    if (sigma && sigma_dims != 1 && sigma_dims != dims)
        return xt_fail(ctx, EXTRACK_E_INVALID, "sigma_dims must be 1 or dims");  // tracking.py:138-143
    return EXTRACK_OK;
}

extern "C" int extrack_upload_bucket(extrack_ctx* ctx, const double* tracks, int64_t n, int32_t len, int32_t dims,
                                     const double* sigma, int32_t sigma_dims, int32_t* bucket_id_out)
{
    if (!ctx || !tracks) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_check_bucket_shape(ctx, n, len, dims, sigma, sigma_dims);
    if (rc) return rc;
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtBucket b;
    b.owned = true;
    b.N = n;
    b.L = len;
    b.D = dims;
    b.KS = sigma ? sigma_dims : 0;
    if (sigma) {  // range of the per-peak errors: lets the 2-state fast path prove its scaling bounds (NaN entries poison their track anyway)
        double lo = INFINITY, hi = -INFINITY;
        const size_t ns = (size_t)n * len * sigma_dims;
        for (size_t i = 0; i < ns; ++i) {
            const double v = sigma[i];
            if (v == v) {
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
        }
        b.sig_min = lo;
        b.sig_max = hi;
    }
    const size_t tb = (size_t)n * len * dims * sizeof(double);
    double* dt = nullptr;
    XT_HIP(ctx, hipMalloc(&dt, tb));
    b.d_tracks = dt;
    hipError_t e = hipMemcpyAsync(dt, tracks, tb, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && sigma) {
        const size_t sb = (size_t)n * len * sigma_dims * sizeof(double);
        double* dsg = nullptr;
        e = hipMalloc(&dsg, sb);
        b.d_sigma = dsg;
        if (e == hipSuccess) e = hipMemcpyAsync(dsg, sigma, sb, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the caller may free its host buffers on return
    if (e != hipSuccess) {
        xt_free_bucket(b);
        return xt_fail(ctx, EXTRACK_E_HIP, std::string("bucket upload: ") + hipGetErrorString(e));
    }
    ctx->buckets.push_back(b);
    if (bucket_id_out) *bucket_id_out = (int32_t)ctx->buckets.size() - 1;
    return EXTRACK_OK;
}

extern "C" int extrack_attach_bucket(extrack_ctx* ctx, const double* d_tracks, int64_t n, int32_t len, int32_t dims,
                                     const double* d_sigma, int32_t sigma_dims, int32_t* bucket_id_out)
{
    if (!ctx || !d_tracks) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_check_bucket_shape(ctx, n, len, dims, d_sigma, sigma_dims);
    if (rc) return rc;
    XtBucket b;
    b.owned = false;
    b.d_tracks = d_tracks;
    b.d_sigma = d_sigma;
    b.N = n;
    b.L = len;
    b.D = dims;
    b.KS = d_sigma ? sigma_dims : 0;
    ctx->buckets.push_back(b);
    if (bucket_id_out) *bucket_id_out = (int32_t)ctx->buckets.size() - 1;
    return EXTRACK_OK;
}

extern "C" int extrack_set_bucket_dt(extrack_ctx* ctx, int32_t bucket_id, const double* dt)
{
    if (!ctx) return EXTRACK_E_INVALID;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtBucket& b = ctx->buckets[bucket_id];
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (b.d_dt) (void)hipFree(b.d_dt);
    b.d_dt = nullptr;
    if (!dt) return EXTRACK_OK;
    const size_t nb = (size_t)b.N * b.L * sizeof(double);
    XT_HIP(ctx, hipMalloc(&b.d_dt, nb));
    XT_HIP(ctx, hipMemcpy(b.d_dt, dt, nb, hipMemcpyHostToDevice));
    return EXTRACK_OK;
}

int xt_validate_model(extrack_ctx* ctx, const extrack_model* m)
{
    if (!m || !m->ds || !m->Fs || !m->TrMat || !m->p_stay) return xt_fail(ctx, EXTRACK_E_INVALID, "null model field");
    if (m->locerr_mode < 0 || m->locerr_mode > 2) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_mode must be 0, 1 or 2");
    if (m->locerr_mode == 0 && (m->locerr_dims < 1 || m->locerr_dims > 3))
        return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1..3");
    if (m->min_len < 1 || m->max_len < 2) return xt_fail(ctx, EXTRACK_E_INVALID, "min_len must be >= 1 and max_len >= 2");
    return EXTRACK_OK;
}

void xt_model_host(const extrack_model* m, XtModelHost& mh)
{
    mh.S = m->n_states;
    mh.NS = m->nb_substeps;
    mh.locerr_dims = m->locerr_mode == 0 ? m->locerr_dims : 1;
    for (int k = 0; k < 3; ++k) mh.locerr[k] = m->locerr[k];
    mh.slope = m->slope;
    mh.offset = m->offset;
    mh.pBL = m->pBL;
    mh.ds = m->ds;
    mh.Fs = m->Fs;
    mh.TrMat = m->TrMat;
    mh.p_stay = m->p_stay;
}

// Ships a model blob through one of the two pinned staging slots to its device slot (stream-ordered) and makes that slot
// the current one (ctx->d_blob).  The host only waits for the copy issued two evaluations ago.
int xt_upload_blob(extrack_ctx* ctx, const std::vector<double>& blob)
{
    if (blob.size() > ctx->blob_cap) {
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 2; ++i) {
            if (ctx->d_blob_s[i]) (void)hipFree(ctx->d_blob_s[i]);
            if (ctx->h_blob_s[i]) (void)hipHostFree(ctx->h_blob_s[i]);
            ctx->d_blob_s[i] = ctx->h_blob_s[i] = nullptr;
            ctx->blob_busy[i] = false;
        }
        ctx->blob_cap = 0;
        for (int i = 0; i < 2; ++i) {
            XT_HIP(ctx, hipMalloc(&ctx->d_blob_s[i], blob.size() * sizeof(double)));
            XT_HIP(ctx, hipHostMalloc(&ctx->h_blob_s[i], blob.size() * sizeof(double), hipHostMallocDefault));
        }
        ctx->blob_cap = blob.size();
    }
    const int slot = (int)(ctx->blob_turn++ & 1u);
    if (ctx->blob_busy[slot]) XT_HIP(ctx, hipEventSynchronize(ctx->ev_blob[slot]));  // the slot's previous copy has left the host buffer
    memcpy(ctx->h_blob_s[slot], blob.data(), blob.size() * sizeof(double));
    XT_HIP(ctx, hipMemcpyAsync(ctx->d_blob_s[slot], ctx->h_blob_s[slot], blob.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    XT_HIP(ctx, hipEventRecord(ctx->ev_blob[slot], ctx->stream));
    ctx->blob_busy[slot] = true;
    ctx->d_blob = ctx->d_blob_s[slot];
    return EXTRACK_OK;
}

// The bucket-descriptor staging area is used in two halves that alternate with the blob slots (same guard events).
size_t xt_desc_base(const extrack_ctx* ctx) { return (size_t)((ctx->blob_turn - 1u) & 1u) * (XT_DESC_CAP / 2); }

// Posterior output buffer of at least `bytes` bytes (kept for the next call).
static int xt_reserve_preds(extrack_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->preds_cap) return EXTRACK_OK;
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_preds) (void)hipFree(ctx->d_preds);
    ctx->d_preds = nullptr;
    ctx->preds_cap = 0;
    XT_HIP(ctx, hipMalloc(&ctx->d_preds, bytes));
    ctx->preds_cap = bytes;
    return EXTRACK_OK;
}

// (Re)builds the digit-slot tables when (S, ns, F) changes.
int xt_prepare_config(extrack_ctx* ctx, const extrack_model* m)
{
    if (ctx->cfg.S != m->n_states || ctx->cfg.NS != m->nb_substeps || ctx->cfg.F != m->frame_len || !ctx->d_base_tab) {
        XtConfig c;
        std::string err = xt_build_config(m->n_states, m->nb_substeps, m->frame_len, c);
        if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_base_tab) (void)hipFree(ctx->d_base_tab);
        if (ctx->d_off_tab) (void)hipFree(ctx->d_off_tab);
        ctx->d_base_tab = ctx->d_off_tab = nullptr;
        XT_HIP(ctx, hipMalloc(&ctx->d_base_tab, c.base_tab.size() * sizeof(int32_t)));
        XT_HIP(ctx, hipMalloc(&ctx->d_off_tab, c.off_tab.size() * sizeof(int32_t)));
        XT_HIP(ctx, hipMemcpy(ctx->d_base_tab, c.base_tab.data(), c.base_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        XT_HIP(ctx, hipMemcpy(ctx->d_off_tab, c.off_tab.data(), c.off_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        ctx->cfg = c;
    }
    return EXTRACK_OK;
}

// Digit-slot tables + model blob of one evaluation.
static int xt_prepare(extrack_ctx* ctx, const extrack_model* m)
{
    int rc = xt_prepare_config(ctx, m);
    if (rc) return rc;
    XtModelHost mh;
    xt_model_host(m, mh);
    xt_build_blob(mh, ctx->cfg, ctx->blob_host);  // kept on the host: the launcher reads the scaling slots of the header
    // a small blob rides in the kernel arguments (XtKernelArgs::blob_inline): no staging copy, one dispatch less per evaluation
    ctx->blob_inline = !ctx->no_fused && ctx->blob_host.size() <= (size_t)XT_INLINE_BLOB;
    if (ctx->blob_inline) return EXTRACK_OK;
    return xt_upload_blob(ctx, ctx->blob_host);
}

struct DevLauncher {
    extrack_ctx* ctx;
    XtKernelArgs a;
    int threads;
    size_t lds;
    int tracks_per_block = 1;
    // buckets served by this launch (<= XT_MAX_BUCKETS), their descriptors are written at ctx->h_desc[desc_off ...]
    std::vector<XtBucketDesc> descs;
    size_t desc_off = 0;
    int grid = 0, occ = 0;
    hipError_t herr = hipSuccess;
    void* extra_arg = nullptr;  // second kernel argument (xt_big_kernel: its scratch description) or nullptr
    double max_blocks = 0.0;    // > 0: upper bound of the grid (scratch budget of the launch)

    template <int G_, int D, int K, bool PREDS>
    bool run()
    {
        if (threads <= 256) return launch(xt_track_kernel<G_, D, K, PREDS, 256>);
        return launch(xt_track_kernel<G_, D, K, PREDS, 1024>);
    }

    template <int GP, int D, int K>
    bool run_entry()
    {
        if (threads <= 256) return launch(xt_entry_kernel<GP, D, K, 256>);
        return launch(xt_entry_kernel<GP, D, K, 1024>);
    }

    template <int F, int D, int K>
    bool run_f2()
    {
        if (ctx->ll_reg2) {  // register-resident variant (xt_reg2.h)
            const void* kp = xt_r2_kernel(F, D, K, 0);
            return kp ? launch_ptr(kp) : false;
        }
        return launch(xt_ll_s2_kernel<F, D, K>);
    }

    template <class KernT>
    bool launch(KernT kern)
    {
        return launch_ptr((const void*)kern);
    }

    bool launch_ptr(const void* kp)
    {
        auto key = std::make_pair(kp, std::make_pair(threads, lds));
        auto it = ctx->occ_cache.find(key);
        if (it == ctx->occ_cache.end()) {
            if (lds > 64 * 1024) {
                herr = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (herr != hipSuccess) return true;
            }
            int o = 0;
            herr = hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kp, threads, lds);
            if (herr != hipSuccess) return true;
            it = ctx->occ_cache.emplace(key, o < 1 ? 1 : o).first;
        }
        occ = it->second;
        // Split the grid over the buckets in proportion to their work (track batches x positions).  The CUs are
        // oversubscribed: waves of equal work do NOT progress equally (VALU issue is arbitrated by age), so a static
        // one-wave-set-per-CU split ends in an under-occupied tail; with several block generations per CU the hardware
        // dispatcher backfills as blocks retire.
        const int nb = (int)descs.size();
        double target = (double)occ * ctx->n_cu * ctx->oversub;
        std::vector<int64_t> nbatch(nb);
        double wsum = 0.0;
        int64_t nbsum = 0;
        for (int i = 0; i < nb; ++i) {
            nbatch[i] = (descs[i].N + tracks_per_block - 1) / tracks_per_block;
            wsum += (double)nbatch[i] * (descs[i].L - 1);
            nbsum += nbatch[i];
        }
        // small launches: a block should still walk >= 4 batches (its fixed costs - tables, staging set-up, final reduction - are about
        // one batch's worth), but never fewer blocks than fill the chip once.  125 000 x 30 (the 8-way shard of the headline dataset):
        // 8 generations 0.399 ms, 4 generations 0.389 ms, 1 generation 0.423 ms (r04, same box)
        if (!ctx->oversub_forced) target = std::max((double)occ * ctx->n_cu, std::min(target, (double)nbsum / 4.0));
        if (max_blocks > 0.0) target = std::max((double)nb, std::min(target, max_blocks));
        int64_t acc = 0;
        for (int i = 0; i < nb; ++i) {
            int64_t n = (int64_t)ceil(target * ((double)nbatch[i] * (descs[i].L - 1)) / wsum);
            n = n < 1 ? 1 : (n > nbatch[i] ? nbatch[i] : n);
            acc += n;
            a.blk_end[i] = (int32_t)acc;
        }
        grid = (int)acc;
        // the descriptors only change when the buckets (or the per-track / posterior outputs) do: keep a host shadow of the device table
        // and skip the copy when it already holds them (one dispatch less per evaluation in a fit)
        bool same = !ctx->no_fused && ctx->desc_shadow.size() >= desc_off + (size_t)nb &&
                    memcmp(ctx->desc_shadow.data() + desc_off, descs.data(), nb * sizeof(XtBucketDesc)) == 0;
        if (!same) {
            if (ctx->blob_inline) {  // no blob slot guards this half of the staging area: wait for whatever still reads it
                herr = hipStreamSynchronize(ctx->stream);
                if (herr != hipSuccess) return true;
            }
            memcpy(ctx->h_desc + desc_off, descs.data(), nb * sizeof(XtBucketDesc));
            herr = hipMemcpyAsync(ctx->d_desc + desc_off, ctx->h_desc + desc_off, nb * sizeof(XtBucketDesc), hipMemcpyHostToDevice, ctx->stream);
            if (herr != hipSuccess) return true;
            if (!ctx->blob_inline) {
                // the staging half is reusable once this copy is done too: move the slot's guard event behind it
                herr = hipEventRecord(ctx->ev_blob[(ctx->blob_turn - 1u) & 1u], ctx->stream);
                if (herr != hipSuccess) return true;
            }
            if (ctx->desc_shadow.size() < desc_off + (size_t)nb) ctx->desc_shadow.resize(desc_off + (size_t)nb);
            memcpy(ctx->desc_shadow.data() + desc_off, descs.data(), nb * sizeof(XtBucketDesc));
        }
        a.desc = ctx->d_desc + desc_off;
        a.ndesc = nb;
        void* kargs[2] = {(void*)&a, extra_arg};
        herr = hipLaunchKernel(kp, dim3(grid), dim3(threads), kargs, lds, ctx->stream);
        if (herr == hipSuccess) herr = hipGetLastError();
        return true;
    }
};

int xt_reserve_partials(extrack_ctx* ctx, size_t n)
{
    if (n <= ctx->partials_cap) return EXTRACK_OK;
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    ctx->d_partials = nullptr;
    ctx->partials_cap = 0;
    XT_HIP(ctx, hipMalloc(&ctx->d_partials, n * sizeof(double)));
    ctx->partials_cap = n;
    return EXTRACK_OK;
}

// Launches ONE kernel for a set of buckets that share (dims, sigma dims): partial sums go to d_partials[poff .. poff+grid).
struct XtFuseTotal {
    double* d_total;      // device word that receives the evaluation's total
    double* h_total_dev;  // + the pinned host word (device view) or nullptr
};

static int xt_launch_group(extrack_ctx* ctx, const extrack_model* m, const std::vector<XtBucket*>& bks, bool preds, bool per_track,
                           double* d_preds, size_t poff, size_t desc_off, int* grid_out, double* d_seq = nullptr,
                           const XtFuseTotal* fuse = nullptr)
{
    const XtConfig& c = ctx->cfg;
    const XtBucket& b0 = *bks[0];
    const int D = b0.D;
    int K;
    if (m->locerr_mode == 0) {
        K = m->locerr_dims;
        if (K != 1 && K != D) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1 or the track dimensionality");
    } else {
        if (!b0.d_sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "per-peak localisation error mode but the bucket has no sigma");
        K = b0.KS;
    }
    DevLauncher l;
    l.ctx = ctx;
    memset(&l.a, 0, sizeof(l.a));
    xt_fill_args_from_config(c, l.a);
    int tpb, threads;
    // d_seq (extrack_sequence_matrix): the general kernel writes the log-weight of every sequence of the last position, without the leaving term
    const bool fast2 = !d_seq && xt_use_fast2(c.S, c.NS, c.F, preds);
    const bool entry = !d_seq && !fast2 && xt_use_entry(c.NS, c.G, c.NG, preds);
    if (entry) {
        xt_entry_geometry(c.S, c.G, c.E, c.NG, D, K, tpb, threads, l.lds);
    } else if (fast2) {
        const int tpw = 64 >> (c.F - 1);
        tpb = tpw * XT_F2_WAVES;
        threads = 64 * XT_F2_WAVES;
        l.lds = ctx->ll_reg2 ? (size_t)xt_r2_block_bytes(0, D, m->locerr_mode ? b0.KS : 0, tpw)
                             : (size_t)xt_f2_block_bytes(D, K, m->locerr_mode ? b0.KS : 0, tpw);
    } else {
        xt_geometry(c, D, K, tpb, threads);
        l.lds = xt_lds_bytes(c, D, K, tpb);
    }
    // the sequence state of a track does not fit a workgroup (more than 1024 groups, more than the CU's LDS, posteriors beyond the built group
    // sizes): one lane per track with the state in global memory (xt_big.h)
    bool big = l.lds > 160 * 1024 || (!fast2 && !entry && (threads > 1024 || (preds && c.G > 6)));
    if (const char* ev = getenv("EXTRACK_FORCE_BIG")) big = big || (atoi(ev) != 0 && !d_seq);
    XtBigArgs bargs;
    if (big) {
        if (d_seq) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "sequence matrix: n_states^frame_len sequences per track do not fit a workgroup");
        if (preds && c.F > 15) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "posteriors: frame_len > 15");
        const int NW = 4;
        threads = 64 * NW;
        tpb = threads;
        l.lds = (size_t)(((xt_tab_doubles(c.S, c.G) + 1) & ~1) + threads) * sizeof(double);
        bargs.ws_stride = xt_big_ws_doubles(c.E, D, K);
        size_t budget_mb = 32 * 1024;
        if (const char* ev = getenv("EXTRACK_BIG_WS_MB")) budget_mb = (size_t)std::max(16, atoi(ev));
        const size_t per_block = (size_t)bargs.ws_stride * sizeof(double) * NW;
        const size_t maxb = std::max<size_t>(1, (budget_mb << 20) / per_block);
        if (maxb < bks.size()) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "n_states^frame_len sequences per track: the state of one workgroup per bucket exceeds the scratch budget (EXTRACK_BIG_WS_MB)");
        l.max_blocks = (double)std::min<size_t>(maxb, (size_t)ctx->n_cu * 8);
        const size_t need = (size_t)l.max_blocks * NW * (size_t)bargs.ws_stride + 64;
        if (need > ctx->big_ws_cap) {
            XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_big_ws) (void)hipFree(ctx->d_big_ws);
            ctx->d_big_ws = nullptr;
            ctx->big_ws_cap = 0;
            XT_HIP(ctx, hipMalloc(&ctx->d_big_ws, need * sizeof(double)));
            ctx->big_ws_cap = need;
        }
        bargs.ws = ctx->d_big_ws;
        l.extra_arg = (void*)&bargs;
    }
    l.threads = threads;
    l.tracks_per_block = tpb;
    if (fast2) {
        // range of the localisation variance over the launch -> may the fast path drop its guards (xt_build_blob)?
        double lo = INFINITY, hi = -INFINITY;
        if (m->locerr_mode == 0) {
            for (int k = 0; k < m->locerr_dims && k < 3; ++k) {
                lo = std::min(lo, m->locerr[k] * m->locerr[k]);
                hi = std::max(hi, m->locerr[k] * m->locerr[k]);
            }
        } else {
            for (XtBucket* b : bks) {
                double s0 = b->sig_min, s1 = b->sig_max;  // NaN when the bucket was attached by device pointer: not provable
                if (m->locerr_mode == 2) {
                    const double a0 = s0 * m->slope + m->offset, a1 = s1 * m->slope + m->offset;
                    s0 = std::max(std::min(a0, a1), 1e-6);
                    s1 = std::max(std::max(a0, a1), 1e-6);
                    if (a0 != a0 || a1 != a1) s0 = s1 = NAN;
                }
                lo = (s0 == s0) ? std::min(lo, s0 * s0) : NAN;
                hi = (s1 == s1) ? std::max(hi, s1 * s1) : NAN;
                if (lo != lo || hi != hi) break;
            }
        }
        l.a.well_scaled = (ctx->blob_host.size() > 8 && xt_model_well_scaled(ctx->blob_host, lo, hi)) ? 1 : 0;
    }
    l.desc_off = desc_off;
    for (XtBucket* b : bks) {
        XtBucketDesc d;
        d.tracks = b->d_tracks;
        d.sigma = m->locerr_mode ? b->d_sigma : nullptr;
        d.ll_out = per_track ? b->d_ll : nullptr;
        d.preds_out = d_preds;
        d.N = b->N;
        d.L = b->L;
        d.isBL = (!d_seq && b->L != m->max_len) ? 1 : 0;  // tracking.py:1037-1040
        d.ll_const = -(double)(b->L - 1) * D * 0.5 * XT_LOG2PI;
        d.seq_out = d_seq;
        l.descs.push_back(d);
    }
    if (ctx->blob_inline) {
        l.a.blob = nullptr;
        memcpy(l.a.blob_inline, ctx->blob_host.data(), ctx->blob_host.size() * sizeof(double));
    } else {
        l.a.blob = ctx->d_blob;
    }
    if (fuse) {
        l.a.done = ctx->d_done;
        l.a.total_out = fuse->d_total;
        l.a.total_host = fuse->h_total_dev;
    }
    l.a.base_tab = ctx->d_base_tab;
    l.a.off_tab = ctx->d_off_tab;
    l.a.partials = ctx->d_partials + poff;
    l.a.TPB = tpb;
    l.a.min_len = m->min_len;
    l.a.locerr_mode = m->locerr_mode;
    l.a.KS = b0.KS ? b0.KS : 1;
    bool ok;
    if (big) {
        const void* kp = xt_big_kernel_dk(D, K, preds);
        ok = kp != nullptr && l.launch_ptr(kp);
    } else {
        ok = fast2 ? xt_dispatch_f2(c.F, D, K, l) : (entry ? xt_dispatch_entry(xt_entry_gp(c.G), D, K, l) : xt_dispatch(c.G, D, K, preds, l));
    }
    if (!ok)
        return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, preds ? "posteriors are built for n_states <= 6" : "kernel variant not built");
    if (l.herr != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("kernel launch: ") + hipGetErrorString(l.herr));
    ctx->launch_info[0] = l.grid;
    ctx->launch_info[1] = threads;
    ctx->launch_info[2] = (int32_t)l.lds;
    ctx->launch_info[3] = tpb;
    ctx->launch_info[4] = l.occ;
    ctx->launch_info[5] = ctx->n_cu;
    *grid_out = l.grid;
    return EXTRACK_OK;
}

// Upper bound of the blocks of one launch (= partial-sum slots to reserve).
size_t xt_max_grid(const extrack_ctx* ctx) { return (size_t)ctx->n_cu * 8 * ctx->oversub + XT_MAX_BUCKETS; }

static int xt_loglik_enqueue(extrack_ctx* ctx, const extrack_model* m, double* d_total, bool per_track, bool to_host = false)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (ctx->buckets.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, "no bucket uploaded");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = xt_prepare(ctx, m))) return rc;
    // launch groups: buckets with the same (dims, sigma dims), longest first, at most XT_MAX_BUCKETS per launch
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
    if ((rc = xt_reserve_partials(ctx, groups.size() * xt_max_grid(ctx)))) return rc;
    if (per_track)
        for (auto& b : ctx->buckets)
            if (!b.d_ll) XT_HIP(ctx, hipMalloc(&b.d_ll, (size_t)b.N * sizeof(double)));
    size_t poff = 0, doff = ctx->blob_inline ? 0 : xt_desc_base(ctx);
    // one launch group (the usual case): the likelihood kernel itself leaves the total in d_total (and in the pinned host word for
    // the synchronous entry point) - ONE dispatch per evaluation instead of five (blob copy, descriptor copy, kernel, reduction, read-back)
    const bool fused = !ctx->no_fused && groups.size() == 1;
    XtFuseTotal fz = {d_total, to_host ? ctx->h_total_dev : nullptr};
    ctx->fused_host = fused && to_host;
    XT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (auto& g : groups) {
        int grid = 0;
        if ((rc = xt_launch_group(ctx, m, g, false, per_track, nullptr, poff, doff, &grid, nullptr, fused ? &fz : nullptr))) return rc;
        poff += (size_t)grid;
        doff += g.size();
    }
    XT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    if (!fused) {
        hipLaunchKernelGGL(xt_reduce_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, (int)poff, d_total);
        XT_HIP(ctx, hipGetLastError());
    }
    return EXTRACK_OK;
}

extern "C" int extrack_loglik_async(extrack_ctx* ctx, const extrack_model* model, double* d_total_ll)
{
    if (!ctx) return EXTRACK_E_INVALID;
    return xt_loglik_enqueue(ctx, model, d_total_ll ? d_total_ll : ctx->d_total, false);
}

extern "C" int extrack_loglik(extrack_ctx* ctx, const extrack_model* model, double* total_ll, double* per_track)
{
    if (!ctx || !total_ll) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_loglik_enqueue(ctx, model, ctx->d_total, per_track != nullptr, true);
    if (rc) return rc;
    if (!ctx->fused_host) XT_HIP(ctx, hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (per_track) {
        size_t o = 0;
        for (auto& b : ctx->buckets) {
            XT_HIP(ctx, hipMemcpyAsync(per_track + o, b.d_ll, (size_t)b.N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            o += (size_t)b.N;
        }
    }
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_ll = *ctx->h_total;
    return EXTRACK_OK;
}

extern "C" int extrack_predict(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double* preds)
{
    if (!ctx || !preds) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    if (m->nb_substeps != 1) return xt_fail(ctx, EXTRACK_E_INVALID, "state predictions require nb_substeps == 1");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = xt_prepare(ctx, m))) return rc;
    if ((rc = xt_reserve_partials(ctx, xt_max_grid(ctx)))) return rc;
    XtBucket& b = ctx->buckets[bucket_id];
    const size_t nb = (size_t)b.N * b.L * m->n_states * sizeof(double);
    if ((rc = xt_reserve_preds(ctx, nb))) return rc;
    double* d_preds = ctx->d_preds;
    int grid = 0;
    hipError_t e = hipEventRecord(ctx->ev0, ctx->stream);
    std::vector<XtBucket*> one(1, &b);
    rc = xt_launch_group(ctx, m, one, true, false, d_preds, 0, xt_desc_base(ctx), &grid);
    if (rc == EXTRACK_OK) {
        if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
        ctx->timed = true;
        if (e == hipSuccess) e = hipMemcpyAsync(preds, d_preds, nb, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = xt_fail(ctx, EXTRACK_E_HIP, std::string("predict: ") + hipGetErrorString(e));
    }
    return rc;
}

// Per-sequence log-probabilities of one bucket in the reference's layout (P_Cs_inter_bound_stats' first return value,
// extrack/tracking.py:300-318): raw kernel output -> host -> column order of the reference (xt_seqmat.h).  For small inputs:
// N * S^(frame_len + nb_substeps) doubles go through host memory.
extern "C" int64_t extrack_sequence_columns(int32_t n_states, int32_t len, int32_t nb_substeps, int32_t frame_len, int32_t isBL)
{
    if (n_states < 2 || len < 2 || nb_substeps < 1 || frame_len <= nb_substeps) return -1;
    return xt_seq_columns(n_states, len, nb_substeps, frame_len, isBL);
}

extern "C" int extrack_sequence_matrix(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double* lp, int64_t n_cols)
{
    if (!ctx || !lp) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = xt_prepare(ctx, m))) return rc;
    if ((rc = xt_reserve_partials(ctx, xt_max_grid(ctx)))) return rc;
    XtBucket& b = ctx->buckets[bucket_id];
    const XtConfig& c = ctx->cfg;
    const int isBL = (b.L != m->max_len) ? 1 : 0;
    if (n_cols != xt_seq_columns(c.S, b.L, c.NS, c.F, isBL)) return xt_fail(ctx, EXTRACK_E_INVALID, "sequence matrix: n_cols must be extrack_sequence_columns(...)");
    const size_t nraw = (size_t)b.N * c.E * c.G;
    if (nraw > ((size_t)1 << 28)) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "sequence matrix: more than 2^28 entries = 2 GiB through host memory (it exists for small inputs; the likelihood needs no matrix: split the tracks)");
    if ((rc = xt_reserve_preds(ctx, nraw * sizeof(double)))) return rc;
    int grid = 0;
    std::vector<XtBucket*> one(1, &b);
    if ((rc = xt_launch_group(ctx, m, one, false, false, nullptr, 0, xt_desc_base(ctx), &grid, ctx->d_preds))) return rc;
    std::vector<double> raw(nraw);
    XT_HIP(ctx, hipMemcpyAsync(raw.data(), ctx->d_preds, nraw * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    XtModelHost mh;
    xt_model_host(m, mh);
    xt_seq_reorder(c, mh, b.N, b.L, isBL, raw.data(), lp);
    return EXTRACK_OK;
}

// ------------------------------------------------------------------------------------------------
// threshold-fusion variant (xt_th.h): plan kernel + apply kernel per bucket
// ------------------------------------------------------------------------------------------------
template <class KernT>
static hipError_t xt_th_set_lds(extrack_ctx* /*ctx*/, KernT kern, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int D, int K>
static hipError_t xt_th_launch_plan(extrack_ctx* ctx, const XtThArgs& a, int grid, int threads, size_t lds, hipStream_t stream)
{
    // pilot-track state in LDS / in the global workspace: two instantiations, so that the state pointers have a known address space
    if (a.ws_lds) {
        hipError_t e = xt_th_set_lds(ctx, xt_th_plan_kernel<D, K, false, 1>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((xt_th_plan_kernel<D, K, false, 1>), dim3(grid), dim3(threads), lds, stream, a);
    } else {
        hipError_t e = xt_th_set_lds(ctx, xt_th_plan_kernel<D, K, false, 0>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((xt_th_plan_kernel<D, K, false, 0>), dim3(grid), dim3(threads), lds, stream, a);
    }
    return hipGetLastError();
}

template <int D, int K, bool UNI, bool SINGLE, bool DT, bool SEQ = false>
static hipError_t xt_th_launch_apply_vd(extrack_ctx* ctx, const XtThArgs& a, int grid, int threads, size_t lds, hipStream_t stream)
{
    hipError_t e = xt_th_set_lds(ctx, xt_th_apply_kernel<D, K, UNI, SINGLE, DT, SEQ>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((xt_th_apply_kernel<D, K, UNI, SINGLE, DT, SEQ>), dim3(grid), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

template <int D, int K, bool UNI, bool SINGLE>
static hipError_t xt_th_launch_apply_v(extrack_ctx* ctx, const XtThArgs& a, int grid, int threads, size_t lds, hipStream_t stream)
{
    if (a.blob_stride != 0) return xt_th_launch_apply_vd<D, K, UNI, SINGLE, true>(ctx, a, grid, threads, lds, stream);  // per-track time steps
    return xt_th_launch_apply_vd<D, K, UNI, SINGLE, false>(ctx, a, grid, threads, lds, stream);
}

// mode 0: general (fewer than 64 tracks per tile), 1: wave-uniform, two state buffers, 2: wave-uniform, one state buffer,
// 3: general with one state buffer (more than 64 live sequences), 4: general + the per-sequence matrix of the last position
template <int D, int K>
static hipError_t xt_th_launch_apply(extrack_ctx* ctx, const XtThArgs& a, int grid, int threads, size_t lds, int mode, hipStream_t stream)
{
    if (mode == 4)
        return a.blob_stride != 0 ? xt_th_launch_apply_vd<D, K, false, false, true, true>(ctx, a, grid, threads, lds, stream)
                                  : xt_th_launch_apply_vd<D, K, false, false, false, true>(ctx, a, grid, threads, lds, stream);
    if (mode == 3) return xt_th_launch_apply_v<D, K, false, true>(ctx, a, grid, threads, lds, stream);
    if (mode == 2) return xt_th_launch_apply_v<D, K, true, true>(ctx, a, grid, threads, lds, stream);
    if (mode == 1) return xt_th_launch_apply_v<D, K, true, false>(ctx, a, grid, threads, lds, stream);
    return xt_th_launch_apply_v<D, K, false, false>(ctx, a, grid, threads, lds, stream);
}

// Grows the partial-sum array to n entries, keeping what earlier launches of this evaluation wrote.
static int xt_grow_partials(extrack_ctx* ctx, size_t n)
{
    if (n <= ctx->partials_cap) return EXTRACK_OK;
    const size_t cap = std::max(n, ctx->partials_cap * 2);
    double* nw = nullptr;
    XT_HIP(ctx, hipMalloc(&nw, cap * sizeof(double)));
    if (ctx->d_partials) {
        XT_HIP(ctx, hipMemcpyAsync(nw, ctx->d_partials, ctx->partials_cap * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_partials);
    }
    ctx->d_partials = nw;
    ctx->partials_cap = cap;
    return EXTRACK_OK;
}

static int xt_th_reserve_plan(extrack_ctx* ctx, XtBucket& b, int chunk, int capE)
{
    const int64_t nchunks = (b.N + chunk - 1) / chunk;
    if (b.th_members && b.th_capE == capE && b.th_chunk == chunk && b.th_nchunks == nchunks) return EXTRACK_OK;
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (b.th_members) (void)hipFree(b.th_members);
    if (b.th_mpack) (void)hipFree(b.th_mpack);
    if (b.th_gnew) (void)hipFree(b.th_gnew);
    if (b.th_gstart) (void)hipFree(b.th_gstart);
    if (b.th_hdr) (void)hipFree(b.th_hdr);
    if (b.th_status) (void)hipFree(b.th_status);
    b.th_members = b.th_gstart = nullptr;
    b.th_mpack = nullptr;
    b.th_gnew = nullptr;
    b.th_hdr = b.th_status = nullptr;
    XT_HIP(ctx, hipMalloc(&b.th_members, (size_t)nchunks * b.L * capE * sizeof(uint16_t)));
    XT_HIP(ctx, hipMalloc(&b.th_mpack, (size_t)nchunks * b.L * capE * sizeof(uint32_t)));
    XT_HIP(ctx, hipMalloc(&b.th_gnew, (size_t)nchunks * b.L * capE));
    XT_HIP(ctx, hipMalloc(&b.th_gstart, (size_t)nchunks * b.L * (capE + 1) * sizeof(uint16_t)));
    XT_HIP(ctx, hipMalloc(&b.th_hdr, (size_t)nchunks * b.L * 2 * sizeof(int32_t)));
    XT_HIP(ctx, hipMalloc(&b.th_status, (size_t)nchunks * 4 * sizeof(int32_t)));
    b.th_capE = capE;
    b.th_chunk = chunk;
    b.th_nchunks = nchunks;
    b.th_maxG = -1;
    return EXTRACK_OK;
}

// Device-side copy of a small host array (bucket descriptors, chunk prefix): grows on demand.
static int xt_th_upload_small(extrack_ctx* ctx, void** d_buf, size_t* cap, const void* src, size_t bytes)
{
    if (bytes > *cap) {
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (*d_buf) (void)hipFree(*d_buf);
        *d_buf = nullptr;
        *cap = 0;
        XT_HIP(ctx, hipMalloc(d_buf, bytes * 2));
        *cap = bytes * 2;
    }
    XT_HIP(ctx, hipMemcpyAsync(*d_buf, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return EXTRACK_OK;
}

// Per-track time steps (XtBucket::d_dt): the field-of-view table - hence the stay / end-of-track tables - belongs to the chunk
// (tracking.py:507-511: median over the chunk's tracks of the first column of ds).  model->p_stay then holds one table of G entries
// per chunk; `tables[c]` is the table index of the c-th chunk of this launch.  Builds and uploads one blob per chunk.
static int xt_th_chunk_blobs(extrack_ctx* ctx, const extrack_model* m, const std::vector<int64_t>& tables, int G, int64_t* stride_out)
{
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> all, one;
    int64_t stride = 0;
    for (size_t c = 0; c < tables.size(); ++c) {
        mh.p_stay = m->p_stay + (size_t)tables[c] * G;
        int G2 = 0;
        xt_th_build_blob(mh, one, G2);
        if (c == 0) {
            stride = (int64_t)one.size();
            all.assign((size_t)stride * tables.size(), 0.0);
        }
        memcpy(all.data() + c * (size_t)stride, one.data(), one.size() * sizeof(double));
    }
    if (all.size() > ctx->th_blobs_cap) {
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_th_blobs) (void)hipFree(ctx->d_th_blobs);
        ctx->d_th_blobs = nullptr;
        ctx->th_blobs_cap = 0;
        XT_HIP(ctx, hipMalloc(&ctx->d_th_blobs, all.size() * sizeof(double)));
        ctx->th_blobs_cap = all.size();
    }
    {
        hipError_t e = hipMemcpy(ctx->d_th_blobs, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            char msg[256];
            snprintf(msg, sizeof(msg), "chunk blobs upload (%zu tables, stride %lld, capacity %zu doubles, dst %p): %s", tables.size(), (long long)stride,
                     ctx->th_blobs_cap, (void*)ctx->d_th_blobs, hipGetErrorString(e));
            return xt_fail(ctx, EXTRACK_E_HIP, msg);
        }
    }
    *stride_out = stride;
    return EXTRACK_OK;
}

// One launch group of a threshold-fusion evaluation: all buckets that share (dims, sigma dims) are served by ONE plan launch
// and ONE apply launch through a device table of bucket descriptors (a real dataset has one bucket per track length; the plan
// kernel of a single small bucket could not fill the GPU and its latency would add up bucket after bucket).
// Several sets of per-launch buffers (chunk status, bucket descriptors, chunk prefix, plan workspace) and side streams, for evaluations
// that run several launch groups concurrently: xt_th_use_slot parks the current set and makes set j (and its stream) the current one.
static void xt_th_use_slot(extrack_ctx* ctx, int j)
{
    extrack_ctx::ThSlot& cur = ctx->th_slot[ctx->th_cur_slot];
    cur.h_status = ctx->h_th_status;
    cur.d_status = ctx->d_th_status;
    cur.status_cap = ctx->th_status_cap;
    cur.d_desc = ctx->d_th_desc;
    cur.desc_cap = ctx->th_desc_cap;
    cur.d_cend = ctx->d_th_cend;
    cur.cend_cap = ctx->th_cend_cap;
    cur.d_ws = ctx->d_th_ws;
    cur.ws_cap = ctx->th_ws_cap;
    const extrack_ctx::ThSlot& nx = ctx->th_slot[j];
    ctx->h_th_status = nx.h_status;
    ctx->d_th_status = nx.d_status;
    ctx->th_status_cap = nx.status_cap;
    ctx->d_th_desc = nx.d_desc;
    ctx->th_desc_cap = nx.desc_cap;
    ctx->d_th_cend = nx.d_cend;
    ctx->th_cend_cap = nx.cend_cap;
    ctx->d_th_ws = nx.d_ws;
    ctx->th_ws_cap = nx.ws_cap;
    ctx->th_cur_slot = j;
    ctx->stream = ctx->th_streams[j];
}
static int xt_th_split_streams(extrack_ctx* ctx)
{
    if (ctx->th_streams[0]) return EXTRACK_OK;
    for (int i = 0; i < extrack_ctx::TH_SLOTS; ++i) XT_HIP(ctx, hipStreamCreateWithFlags(&ctx->th_streams[i], hipStreamNonBlocking));
    for (int i = 0; i < extrack_ctx::TH_SLOTS + 1; ++i) XT_HIP(ctx, hipEventCreateWithFlags(&ctx->th_ev[i], hipEventDisableTiming));
    return EXTRACK_OK;
}

// `between`: called once, after the plan kernel of this group has been launched and before the host waits for it - the caller uses it to
// run ANOTHER group (on another stream, with the other set of launch buffers) while this group's plan - a serial walk over the positions
// of its longest chunk - is in flight.
static int xt_th_run_group(extrack_ctx* ctx, const extrack_model* m, const std::vector<XtBucket*>& bks, double threshold, int32_t max_nb_states,
                           int32_t chunk, int G, bool per_track, size_t& poff, const std::vector<int64_t>* chunk_base,
                           const std::function<int()>* between = nullptr, const XtThAfterPlan* after_plan = nullptr)
{
    const int S = m->n_states, NS = m->nb_substeps, F = m->frame_len;
    const XtBucket& b0 = *bks[0];
    const int D = b0.D;
    int K;
    if (m->locerr_mode == 0) {
        K = m->locerr_dims;
    } else {
        for (XtBucket* b : bks)
            if (!b->d_sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "per-peak localisation error mode but the bucket has no sigma");
        K = b0.KS;
    }
    if (!(K == 1 || (K == D && D > 1))) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1 or the track dimensionality");
    const int nbk = (int)bks.size();
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.blob = ctx->d_blob;
    a.S = S;
    a.NS = NS;
    a.G = G;
    a.F = F;
    a.min_len = m->min_len;
    a.locerr_mode = m->locerr_mode;
    a.KS = b0.KS ? b0.KS : 1;
    a.chunk = chunk;
    a.max_nb = max_nb_states;
    a.threshold = threshold;
    a.pcap = std::min(chunk, XT_TH_PILOT);
    a.pair_lanes_max_p = ctx->th_pair_lanes;
    a.plan_bs = ctx->th_plan_bs;
    a.nbuckets = nbk;
    std::vector<int32_t> chunk_end(nbk);
    int64_t total = 0;
    int Lmax = 0;
    for (int i = 0; i < nbk; ++i) {
        total += (bks[i]->N + chunk - 1) / chunk;
        if (total > (int64_t)1 << 30) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "too many chunks");
        chunk_end[i] = (int32_t)total;
        Lmax = std::max(Lmax, bks[i]->L);
    }
    a.nchunks = (int32_t)total;
    a.Lmax = Lmax;
    a.L = Lmax;
    if (chunk_base) {  // per-track time steps: one blob per chunk, in this launch's chunk order
        std::vector<int64_t> tables;
        for (int i = 0; i < nbk; ++i) {
            const int64_t base = (*chunk_base)[bks[i] - &ctx->buckets[0]];
            for (int64_t c = 0; c < (bks[i]->N + chunk - 1) / chunk; ++c) tables.push_back(base + c);
        }
        int64_t stride = 0;
        int rcb = xt_th_chunk_blobs(ctx, m, tables, G, &stride);
        if (rcb) return rcb;
        a.blob = ctx->d_th_blobs;
        a.blob_stride = stride;
    }
    // status of every chunk of the group: one device array, one pinned host copy
    if ((size_t)total * 4 > ctx->th_status_cap) {
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_th_status) (void)hipHostFree(ctx->h_th_status);
        if (ctx->d_th_status) (void)hipFree(ctx->d_th_status);
        ctx->h_th_status = ctx->d_th_status = nullptr;
        ctx->th_status_cap = 0;
        XT_HIP(ctx, hipHostMalloc(&ctx->h_th_status, (size_t)total * 8 * sizeof(int32_t), hipHostMallocDefault));
        XT_HIP(ctx, hipMalloc(&ctx->d_th_status, (size_t)total * 8 * sizeof(int32_t)));
        ctx->th_status_cap = (size_t)total * 8;
    }
    std::vector<XtThBucket> desc(nbk);
    hipError_t e = hipSuccess;
    int rc, maxG = 0, sumE = 0;
    bool force_global = false;
    if (ctx->th_frozen) {
        // frozen plan: no plan kernel, no read-back - the buckets still hold the plan of the last planning evaluation
        bool ok = !chunk_base;
        for (int i = 0; ok && i < nbk; ++i)
            ok = bks[i]->th_members && bks[i]->th_maxG >= 0 && bks[i]->th_capE == bks[0]->th_capE && bks[i]->th_chunk == chunk;
        if (!ok) return xt_fail(ctx, EXTRACK_E_INVALID, "frozen plan: no plan of a previous evaluation with this chunk size for these buckets (evaluate once unfrozen first; per-track time steps are not served)");
        a.capE = bks[0]->th_capE;
        for (int i = 0; i < nbk; ++i) {
            maxG = std::max(maxG, bks[i]->th_maxG);
            sumE = std::max(sumE, bks[i]->th_sumE);
        }
        for (int i = 0; i < nbk; ++i) {
            XtBucket& b = *bks[i];
            XtThBucket& k = desc[i];
            k.tracks = b.d_tracks;
            k.sigma = m->locerr_mode ? b.d_sigma : nullptr;
            k.dt = nullptr;
            k.ll_out = per_track ? b.d_ll : nullptr;
            k.preds_out = nullptr;
            k.N = b.N;
            k.L = b.L;
            k.isBL = (b.L != m->max_len) ? 1 : 0;
            k.ll_const = -(double)(b.L - 1) * D * 0.5 * XT_LOG2PI;
            k.members = b.th_members;
            k.mpack = b.th_mpack;
            k.gstart = b.th_gstart;
            k.gnew = b.th_gnew;
            k.hdr = b.th_hdr;
            k.status = ctx->d_th_status + (size_t)(i ? chunk_end[i - 1] : 0) * 4;
            k.seq_out = b.d_seqth;
            k.seq_stride = b.seqth_stride;
        }
        if ((rc = xt_th_upload_small(ctx, (void**)&ctx->d_th_desc, &ctx->th_desc_cap, desc.data(), desc.size() * sizeof(XtThBucket)))) return rc;
        if ((rc = xt_th_upload_small(ctx, (void**)&ctx->d_th_cend, &ctx->th_cend_cap, chunk_end.data(), chunk_end.size() * sizeof(int32_t)))) return rc;
        a.buckets = ctx->d_th_desc;
        a.chunk_end = ctx->d_th_cend;
        if (between) {
            const std::function<int()>* f = between;
            between = nullptr;
            if ((rc = (*f)())) return rc;
        }
    }
    for (; !ctx->th_frozen;) {  // plan, growing the capacity on overflow
        int capE = ctx->th_capE;
        while (capE < S * G) capE *= 2;
        ctx->th_capE = capE;
        a.capE = capE;
        for (int i = 0; i < nbk; ++i) {
            XtBucket& b = *bks[i];
            if ((rc = xt_th_reserve_plan(ctx, b, chunk, capE))) return rc;
            XtThBucket& k = desc[i];
            k.tracks = b.d_tracks;
            k.sigma = m->locerr_mode ? b.d_sigma : nullptr;
            k.dt = chunk_base ? b.d_dt : nullptr;
            k.ll_out = per_track ? b.d_ll : nullptr;
            k.preds_out = nullptr;
            k.N = b.N;
            k.L = b.L;
            k.isBL = (b.L != m->max_len) ? 1 : 0;  // tracking.py:1037-1040
            k.ll_const = -(double)(b.L - 1) * D * 0.5 * XT_LOG2PI;
            k.members = b.th_members;
            k.mpack = b.th_mpack;
            k.gstart = b.th_gstart;
            k.gnew = b.th_gnew;
            k.hdr = b.th_hdr;
            k.status = ctx->d_th_status + (size_t)(i ? chunk_end[i - 1] : 0) * 4;
            k.seq_out = b.d_seqth;
            k.seq_stride = b.seqth_stride;
        }
        if ((rc = xt_th_upload_small(ctx, (void**)&ctx->d_th_desc, &ctx->th_desc_cap, desc.data(), desc.size() * sizeof(XtThBucket)))) return rc;
        if ((rc = xt_th_upload_small(ctx, (void**)&ctx->d_th_cend, &ctx->th_cend_cap, chunk_end.data(), chunk_end.size() * sizeof(int32_t)))) return rc;
        a.buckets = ctx->d_th_desc;
        a.chunk_end = ctx->d_th_cend;
        int grid = (int)std::min<int64_t>(a.nchunks, (int64_t)ctx->n_cu * 2);
        // pilot-track state: in LDS when the sequence counts of the previous evaluation (+25 %) fit 64 KiB, else in a global
        // workspace sized for the full plan capacity
        size_t lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, K) * sizeof(double);
        bool lds_mode = false;
        a.wsP = a.wsE = capE;
        if (ctx->th_learnE > 0 && !force_global) {
            // ODD per-pilot strides: the pair tests read the pilots' means / stds with lanes = pilot tracks, i.e. at a stride of wsP / wsE
            // doubles - an even stride put the 32 lanes on 16 ... 1 bank pairs (r02 PMC: 59 - 71 % of the plan kernel's LDS cycles were conflicts)
            const int wp = std::min(capE, std::max(S * G, ctx->th_learnP)) | 1, we = std::min(capE, std::max(S * G, ctx->th_learnE)) | 1;
            const size_t need = lds + (size_t)xt_th_ws_doubles(wp, we, D, K, F, NS, S, a.pcap) * sizeof(double);
            if (need <= 64 * 1024) {
                lds_mode = true;
                lds = need;
                a.wsP = wp;
                a.wsE = we;
            }
        }
        a.ws_lds = lds_mode ? 1 : 0;
        // more expanded sequences per step than the LDS holds plan arrays for (4 states x 3 substeps: 4^4 x 4^3 = 16 384 at the second position):
        // the per-step plan arrays move to the global workspace too
        a.plan_glb = (!lds_mode && capE > XT_TH_MAXCAP) ? 1 : 0;
        if (a.plan_glb) lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, K, XT_TH_CMAT_WORDS, true) * sizeof(double);
        a.stP = a.stE = 0;
        if (lds_mode && ctx->th_stage_in_lds_mode) {
            const size_t st = (size_t)a.pcap * ((size_t)a.wsP * D + (size_t)a.wsE * K) * sizeof(double);
            if (lds + st <= 80 * 1024) {
                a.stP = a.wsP;
                a.stE = a.wsE;
                lds += st;
            }
        }
        if (!lds_mode && ctx->th_learnE > 0 && !force_global) {
            // LDS copy of what the grouping reads (pilot means, stds), sized by the previous evaluation's sequence counts
            const int sp = std::min(capE, ctx->th_learnP) | 1, se = std::min(capE, ctx->th_learnE) | 1;
            const size_t st = (size_t)a.pcap * ((size_t)sp * D + (size_t)se * K) * sizeof(double);
            if (lds + st <= 120 * 1024) {
                a.stP = sp;
                a.stE = se;
                lds += st;
            }
        }
        a.ws_stride = xt_th_ws_doubles(a.wsP, a.wsE, D, K, F, NS, S, a.pcap) + (a.plan_glb ? xt_th_plan_glb_doubles(capE) : 0);
        if (!lds_mode) {
            // the compatibility bit matrix of a workgroup grows with capE^2 (32 MiB at 16 384): fewer workgroups in flight keep the workspace below ~24 GiB
            const size_t per_wg = (size_t)a.ws_stride * sizeof(double);
            grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)grid, ((size_t)24 << 30) / per_wg));
            const size_t need = (size_t)a.ws_stride * grid * sizeof(double);
            if (need > ctx->th_ws_cap) {
                XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->d_th_ws) (void)hipFree(ctx->d_th_ws);
                ctx->d_th_ws = nullptr;
                ctx->th_ws_cap = 0;
                XT_HIP(ctx, hipMalloc(&ctx->d_th_ws, need));
                ctx->th_ws_cap = need;
            }
        }
        a.ws = ctx->d_th_ws;
        if (lds > 160 * 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "plan tables do not fit the 160 KiB LDS of a CU");
        // workgroup size: a chunk's plan is a serial walk over its positions; with many live sequences (more than 64 expanded per
        // step at the previous evaluation) the grouping's pair tests dominate a step and are shared by all wavefronts
        const int plan_threads = ctx->th_plan_threads_forced ? ctx->th_plan_threads : (ctx->th_learnE > 64 + 64 / 4 + 2 ? 1024 : ctx->th_plan_threads);
#define XT_TH_PLAN_CALL(...) xt_th_launch_plan<__VA_ARGS__>(ctx, a, grid, plan_threads, lds, ctx->stream)
        if (D == 1 && K == 1) e = XT_TH_PLAN_CALL(1, 1);
        else if (D == 2 && K == 1) e = XT_TH_PLAN_CALL(2, 1);
        else if (D == 2 && K == 2) e = XT_TH_PLAN_CALL(2, 2);
        else if (D == 3 && K == 1) e = XT_TH_PLAN_CALL(3, 1);
        else e = XT_TH_PLAN_CALL(3, 3);
#undef XT_TH_PLAN_CALL
        if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("plan kernel launch: ") + hipGetErrorString(e));
        XT_HIP(ctx, hipMemcpyAsync(ctx->h_th_status, ctx->d_th_status, (size_t)a.nchunks * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (between) {
            const std::function<int()>* f = between;
            between = nullptr;
            if ((rc = (*f)())) return rc;
        }
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int over = 0, maxE = 0;
        maxG = sumE = 0;
        for (int c = 0; c < a.nchunks; ++c) {
            over |= ctx->h_th_status[(size_t)c * 4];
            maxE = std::max(maxE, ctx->h_th_status[(size_t)c * 4 + 1]);
            maxG = std::max(maxG, ctx->h_th_status[(size_t)c * 4 + 2]);
            sumE = std::max(sumE, ctx->h_th_status[(size_t)c * 4 + 3]);
        }
        for (int i = 0; i < nbk; ++i) {  // per bucket: what a later evaluation with this plan frozen needs to size its launches
            XtBucket& b = *bks[i];
            b.th_maxG = over ? -1 : 0;
            b.th_sumE = 0;
            for (int c = (i ? chunk_end[i - 1] : 0); !over && c < chunk_end[i]; ++c) {
                b.th_maxG = std::max(b.th_maxG, ctx->h_th_status[(size_t)c * 4 + 2]);
                b.th_sumE = std::max(b.th_sumE, ctx->h_th_status[(size_t)c * 4 + 3]);
            }
        }
        if (!over) {
            const int lp = maxG + maxG / 4 + 2, le = maxE + maxE / 4 + 2;
            ctx->th_learnP = ctx->th_split_active ? std::max(ctx->th_learnP_split, lp) : lp;
            ctx->th_learnE = ctx->th_split_active ? std::max(ctx->th_learnE_split, le) : le;
            ctx->th_learnP_split = ctx->th_learnP;
            ctx->th_learnE_split = ctx->th_learnE;
            break;
        }
        if (lds_mode) {  // the learned LDS capacities were too small for these parameters: redo with the global workspace
            force_global = true;
            continue;
        }
        int ncap = capE;
        while (ncap < std::max(maxE, maxG)) ncap *= 2;
        if (ncap == capE) ncap *= 2;
        if (ncap > XT_TH_MAXCAP_FIT)
            return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "more than 32768 live state sequences per step (threshold fusion expands every sequence by n_states^nb_substeps before it merges): raise threshold, lower max_nb_states or nb_substeps - or use the fixed-window kernel (fusion='window' / extrack_loglik), which serves this model");
        ctx->th_capE = ncap;
    }
    if (after_plan) return (*after_plan)(a, D, K, maxG, Lmax);  // the plan is all the caller wanted (frozen-plan gradient, extrack_thgrad.hip)
    // apply geometry: a workgroup serves tiles of TT tracks of one chunk and keeps that chunk's plan in LDS when it is small
    // enough (always, for the usual 2-3 state models); TT = as many tracks as keep the tile within ~48 KiB of LDS
    a.capG = maxG;
    a.plan_cap = (size_t)sumE * 6 + 2 * (size_t)Lmax <= 24 * 1024 ? std::max(sumE, 1) : 0;
    // a step whose member list alone would take more than 32 KiB of LDS (4 states x 3 substeps: 16 384 members at the second position): the
    // general variants read the lists from global memory instead, which leaves the LDS to the state of more tracks per tile
    if (a.plan_cap == 0 && (size_t)maxG * G * 6 > 32 * 1024 && !getenv("EXTRACK_TH_NO_DIRECT")) a.plan_cap = -1;
    auto lds_of = [&](int tt, bool single = false) {
        return (size_t)xt_th_apply_lds_doubles(S, G, maxG, tt, D, K, a.locerr_mode ? a.KS : 0, Lmax, a.plan_cap, tt == 64, single) * 8;
    };
    // 64 tracks per tile (wave-uniform scalar path): two state buffers when two such workgroups fit a CU's LDS, one buffer
    // (merged sequences wait in registers) while at most XT_TH_GPW groups fall to a wavefront; else fewer tracks
    int TT = 64;
    int single_buf = 0;
    if (ctx->th_force_tt > 0) TT = ctx->th_force_tt;
    else if (chunk < 48 || lds_of(64) > 76 * 1024) {
        if (chunk >= 48 && maxG <= 16 * XT_TH_GPW && lds_of(64, true) <= 160 * 1024) {
            single_buf = 1;
        } else {
            // more live sequences than the wave-uniform variants hold: the largest tile whose single state buffer fits the
            // LDS and whose groups fit XT_TH_GPW per thread of a 1024-thread workgroup; else the two-buffer general variant
            TT = 0;
            if (chunk >= 48 && !ctx->th_no_gen_single)
                for (int tt = 32; tt >= 8; tt >>= 1)
                    if (lds_of(tt, true) <= 150 * 1024 && maxG <= (1024 / tt) * XT_TH_GPW) {
                        TT = tt;
                        single_buf = 1;
                        break;
                    }
            if (!TT) {
                TT = 32;
                while (TT > 1 && (TT > chunk * 2 || lds_of(TT) > 48 * 1024)) TT >>= 1;
            }
        }
    }
    if (TT == 64 && ctx->th_force_single && maxG <= 16 * XT_TH_GPW) single_buf = 1;
    bool want_seq = false;  // extrack_sequence_matrix_th: only the general two-buffer variant writes the per-sequence matrix
    for (int i = 0; i < nbk; ++i) want_seq = want_seq || bks[i]->d_seqth != nullptr;
    if (want_seq) {
        single_buf = 0;
        TT = 32;
        while (TT > 1 && (TT > chunk * 2 || lds_of(TT) > 48 * 1024)) TT >>= 1;
    }
    while (TT > 1 && lds_of(TT, single_buf) > 160 * 1024) TT >>= 1;
    if (TT != 64 && single_buf && maxG > (1024 / TT) * XT_TH_GPW) single_buf = 0;
    const bool uni = TT == 64;
    const size_t lds = lds_of(TT, single_buf);
    if (lds > 160 * 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "live state sequences do not fit the 160 KiB LDS of a CU");
    a.TT = TT;
    a.logTT = 0;
    while ((1 << a.logTT) < TT) ++a.logTT;
    int threads = (maxG * TT + 63) / 64 * 64;
    threads = threads > 256 ? 256 : threads;
    threads = threads < TT ? TT : threads;
    if (uni) threads = 64 * std::max(4, std::min(16, maxG));  // one wavefront per live parent sequence of the 64-track tile
    if (!uni && single_buf) threads = 1024;
    int force_threads = ctx->th_force_threads;
    if (!uni && single_buf) force_threads = 0;
    if (uni && single_buf && force_threads > 0 && (force_threads / 64) * XT_TH_GPW < maxG) force_threads = 0;
    if (force_threads > 0 && force_threads % TT == 0) threads = force_threads;
    const int64_t tpc = (chunk + TT - 1) / TT;
    int blocks_per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
    blocks_per_cu = std::max(1, std::min(blocks_per_cu, 2048 / threads));
    // several length buckets in one launch: chunks differ in cost by the ratio of their track lengths, so cut them finer
    const int64_t target = (int64_t)ctx->n_cu * blocks_per_cu * ctx->th_oversub * (nbk > 1 ? 2 : 1);
    int64_t bpc = (target + a.nchunks - 1) / a.nchunks;
    bpc = std::max<int64_t>(1, std::min<int64_t>(bpc, tpc));
    a.bpc = (int32_t)bpc;
    const int grid = (int)(a.nchunks * bpc);
    if ((rc = xt_grow_partials(ctx, poff + (size_t)grid))) return rc;
    a.partials = ctx->d_partials + poff;
#define XT_TH_APPLY_CALL(...) xt_th_launch_apply<__VA_ARGS__>(ctx, a, grid, threads, lds, want_seq ? 4 : (uni ? (single_buf ? 2 : 1) : (single_buf ? 3 : 0)), ctx->stream)
    if (D == 1 && K == 1) e = XT_TH_APPLY_CALL(1, 1);
    else if (D == 2 && K == 1) e = XT_TH_APPLY_CALL(2, 1);
    else if (D == 2 && K == 2) e = XT_TH_APPLY_CALL(2, 2);
    else if (D == 3 && K == 1) e = XT_TH_APPLY_CALL(3, 1);
    else e = XT_TH_APPLY_CALL(3, 3);
#undef XT_TH_APPLY_CALL
    if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("apply kernel launch: ") + hipGetErrorString(e));
    poff += (size_t)grid;
    if (getenv("EXTRACK_TH_DEBUG"))
        fprintf(stderr, "[th] chunks %d  plan: lds_mode %d wsP %d wsE %d stP %d | maxG %d sumE %d plan_cap %d | apply: uni %d single %d TT %d threads %d lds %zu bpc %d grid %d\n",
                a.nchunks, a.ws_lds, a.wsP, a.wsE, a.stP, maxG, sumE, a.plan_cap, (int)uni, single_buf, TT, threads, lds, a.bpc, grid);
    ctx->launch_info[0] = grid;
    ctx->launch_info[1] = threads;
    ctx->launch_info[2] = (int32_t)lds;
    ctx->launch_info[3] = TT;
    ctx->launch_info[4] = blocks_per_cu;
    ctx->launch_info[5] = ctx->n_cu;
    return EXTRACK_OK;
}

// Enqueues one threshold-fusion evaluation; the scalar ends up in d_total (device).  The plan kernel's status words are read back
// between the plan and the apply launch (the apply geometry depends on the live-sequence counts), everything after that is
// stream-ordered.
static int xt_loglik_th_enqueue(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk,
                                double* d_total, bool per_track)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (ctx->buckets.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, "no bucket uploaded");
    if (chunk < 1) return xt_fail(ctx, EXTRACK_E_INVALID, "chunk must be >= 1");
    if (!(threshold >= 0.0)) return xt_fail(ctx, EXTRACK_E_INVALID, "threshold must be >= 0");
    if (m->frame_len <= m->nb_substeps || m->frame_len > 15) return xt_fail(ctx, EXTRACK_E_INVALID, "frame_len must be in (nb_substeps, 15]");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    int G = 0;
    std::string err = xt_th_build_blob(mh, blob, G);
    if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    if (m->n_states * G > XT_TH_MAXCAP) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "n_states^(nb_substeps+1) exceeds the plan capacity");
    // per-track time steps: every bucket carries a dt array and the model one p_stay table per chunk (buckets in id order)
    bool dt_mode = false;
    std::vector<int64_t> chunk_base(ctx->buckets.size(), 0);
    {
        size_t ndt = 0;
        int64_t acc = 0;
        for (size_t i = 0; i < ctx->buckets.size(); ++i) {
            ndt += ctx->buckets[i].d_dt ? 1 : 0;
            chunk_base[i] = acc;
            acc += (ctx->buckets[i].N + chunk - 1) / chunk;
        }
        dt_mode = ndt > 0;
        if (dt_mode && ndt != ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "per-track time steps were set for some buckets only");
        if (dt_mode && (int64_t)m->n_p_stay != acc)
            return xt_fail(ctx, EXTRACK_E_INVALID, "per-track time steps: model->n_p_stay must be the number of chunks (one p_stay table per chunk)");
        if (!dt_mode && m->n_p_stay > 1) return xt_fail(ctx, EXTRACK_E_INVALID, "several p_stay tables but no per-track time steps");
    }
    if (per_track)
        for (auto& b : ctx->buckets)
            if (!b.d_ll) XT_HIP(ctx, hipMalloc(&b.d_ll, (size_t)b.N * sizeof(double)));
    // launch groups: buckets with the same (dims, sigma dims)
    std::vector<XtBucket*> order;
    for (auto& b : ctx->buckets) order.push_back(&b);
    // longest tracks first inside a launch group: a chunk's plan is a serial walk over its positions, so the long chunks are the
    // critical path of the plan kernel and must not be the last ones to start
    std::stable_sort(order.begin(), order.end(), [](const XtBucket* x, const XtBucket* y) {
        if (x->D != y->D) return x->D < y->D;
        if (x->KS != y->KS) return x->KS < y->KS;
        return x->L > y->L;
    });
    size_t poff = 0;
    XT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (size_t i = 0; i < order.size();) {
        size_t jn = i;
        std::vector<XtBucket*> grp;
        while (jn < order.size() && order[jn]->D == order[i]->D && order[jn]->KS == order[i]->KS) grp.push_back(order[jn++]);
        // Large multi-bucket group in steady state (capacities learned, buffers allocated): the long buckets' plan - whose critical path is
        // the serial walk over the longest chunk, during which most of the chip idles - runs on one stream while the short buckets are
        // planned AND applied on a second one.
        int64_t gchunks = 0;
        for (XtBucket* b : grp) gchunks += (b->N + chunk - 1) / chunk;
        // segments by track length (the group is sorted longest first)
        std::vector<std::vector<XtBucket*>> seg;
        {
            size_t k0 = 0;
            for (int t = 0; t < 2 && ctx->th_split_pct[t] > 0; ++t) {
                size_t k1 = k0;
                while (k1 < grp.size() && grp[k1]->L * 100 > grp[0]->L * ctx->th_split_pct[t]) ++k1;
                if (k1 > k0) seg.emplace_back(grp.begin() + k0, grp.begin() + k1);
                k0 = k1;
            }
            if (k0 < grp.size()) seg.emplace_back(grp.begin() + k0, grp.end());
        }
        const bool split = !ctx->th_frozen && !ctx->th_no_split && !dt_mode && grp.size() >= 4 && seg.size() >= 2 && gchunks >= ctx->n_cu && ctx->th_learnE > 0;
        if (!split) {
            if ((rc = xt_th_run_group(ctx, m, grp, threshold, max_nb_states, chunk, G, per_track, poff, dt_mode ? &chunk_base : nullptr))) return rc;
        } else {
            if ((rc = xt_th_split_streams(ctx))) return rc;
            const int nseg = (int)seg.size();
            // partial sums of all apply launches: reserved up front (a reallocation while another stream's kernel writes would be fatal)
            if ((rc = xt_grow_partials(ctx, poff + (size_t)gchunks + (size_t)nseg * ((size_t)ctx->n_cu * 8 * ctx->th_oversub * 2 + 64)))) return rc;
            hipStream_t main_stream = ctx->stream;
            XT_HIP(ctx, hipEventRecord(ctx->th_ev[extrack_ctx::TH_SLOTS], main_stream));
            for (int j = 0; j < nseg; ++j) XT_HIP(ctx, hipStreamWaitEvent(ctx->th_streams[j], ctx->th_ev[extrack_ctx::TH_SLOTS], 0));
            ctx->th_split_active = true;
            ctx->th_learnP_split = ctx->th_learnE_split = 0;
            // segment j on stream j with buffer set j; while its plan is in flight, segment j + 1 (and so on) is planned and applied
            std::function<int(int)> run_seg = [&](int j) -> int {
                xt_th_use_slot(ctx, j);
                const std::function<int()> next = [&, j]() -> int {
                    const int r2 = run_seg(j + 1);
                    xt_th_use_slot(ctx, j);
                    return r2;
                };
                return xt_th_run_group(ctx, m, seg[j], threshold, max_nb_states, chunk, G, per_track, poff, nullptr, j + 1 < nseg ? &next : nullptr);
            };
            rc = run_seg(0);
            xt_th_use_slot(ctx, 0);
            ctx->stream = main_stream;
            ctx->th_split_active = false;
            // join (also after a failure: nothing may be left running on the side streams)
            for (int j = 0; j < nseg; ++j) {
                (void)hipEventRecord(ctx->th_ev[j], ctx->th_streams[j]);
                (void)hipStreamWaitEvent(main_stream, ctx->th_ev[j], 0);
            }
            if (rc) {
                for (int j = 0; j < nseg; ++j) (void)hipStreamSynchronize(ctx->th_streams[j]);
                return rc;
            }
        }
        i = jn;
    }
    XT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    hipLaunchKernelGGL(xt_reduce_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, (int)poff, d_total);
    XT_HIP(ctx, hipGetLastError());
    return EXTRACK_OK;
}

// The plan stage alone, for every launch group (buckets sharing dims / sigma dims) of the uploaded dataset: validates like
// xt_loglik_th_enqueue, uploads the threshold-fusion blob (ctx->d_blob), runs the plan kernel (capacity growth included) and hands the
// group's arguments to `cb` (extrack_thgrad.hip launches the frozen-plan gradient kernel there).  One stream, no concurrent groups.
int xt_th_plan_groups(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk, const XtThAfterPlan& cb)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (ctx->buckets.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, "no bucket uploaded");
    if (chunk < 1) return xt_fail(ctx, EXTRACK_E_INVALID, "chunk must be >= 1");
    if (!(threshold >= 0.0)) return xt_fail(ctx, EXTRACK_E_INVALID, "threshold must be >= 0");
    if (m->frame_len <= m->nb_substeps || m->frame_len > 15) return xt_fail(ctx, EXTRACK_E_INVALID, "frame_len must be in (nb_substeps, 15]");
    for (auto& b : ctx->buckets)
        if (b.d_dt) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "per-track time steps are not served by the frozen-plan gradient");
    if (m->n_p_stay > 1) return xt_fail(ctx, EXTRACK_E_INVALID, "several p_stay tables but no per-track time steps");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    int G = 0;
    std::string err = xt_th_build_blob(mh, blob, G);
    if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    if (m->n_states * G > XT_TH_MAXCAP) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "n_states^(nb_substeps+1) exceeds the plan capacity");
    std::vector<XtBucket*> order;
    for (auto& b : ctx->buckets) order.push_back(&b);
    std::stable_sort(order.begin(), order.end(), [](const XtBucket* x, const XtBucket* y) {
        if (x->D != y->D) return x->D < y->D;
        if (x->KS != y->KS) return x->KS < y->KS;
        return x->L > y->L;
    });
    size_t poff = 0;
    for (size_t i = 0; i < order.size();) {
        size_t jn = i;
        std::vector<XtBucket*> grp;
        while (jn < order.size() && order[jn]->D == order[i]->D && order[jn]->KS == order[i]->KS) grp.push_back(order[jn++]);
        if ((rc = xt_th_run_group(ctx, m, grp, threshold, max_nb_states, chunk, G, false, poff, nullptr, nullptr, &cb))) return rc;
        i = jn;
    }
    return EXTRACK_OK;
}

extern "C" int extrack_th_freeze_plan(extrack_ctx* ctx, int32_t on)
{
    if (!ctx) return EXTRACK_E_INVALID;
    ctx->th_frozen = on != 0;
    return EXTRACK_OK;
}

// Per-sequence log-probabilities of the threshold-fusion kernel for ONE bucket taken as one chunk (what P_Cs_inter_bound_stats_th returns first,
// extrack/tracking.py:650, before the caller's log-sum): lp host [n][n_cols] with n_cols = (sequences alive after the last merge) x
// n_states^nb_substeps, column (g, r) = g * n_states^nb_substeps + r in the reference's order; WITHOUT the leaving / bleaching term of isBL
// tracks (a further expansion by n_states^nb_substeps that the caller adds: its factors depend on the model only).  First call with lp ==
// nullptr to get *n_cols_out.  For small inputs: n * n_cols doubles cross the host.
extern "C" int extrack_sequence_matrix_th(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double threshold, int32_t max_nb_states,
                                          double* lp, int64_t n_cols_cap, int64_t* n_cols_out)
{
    if (!ctx || !n_cols_out) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    XtBucket& b = ctx->buckets[bucket_id];
    if (b.d_dt) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "sequence matrix: per-track time steps are not served");
    if (!(threshold >= 0.0)) return xt_fail(ctx, EXTRACK_E_INVALID, "threshold must be >= 0");
    if (m->frame_len <= m->nb_substeps || m->frame_len > 15) return xt_fail(ctx, EXTRACK_E_INVALID, "frame_len must be in (nb_substeps, 15]");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    int G = 0;
    std::string err = xt_th_build_blob(mh, blob, G);
    if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    const int32_t chunk = (int32_t)std::min<int64_t>(b.N, (int64_t)1 << 30);  // the whole bucket is one chunk: its first 30 tracks decide the merges
    std::vector<XtBucket*> one(1, &b);
    const bool was_frozen = ctx->th_frozen;
    ctx->th_frozen = false;
    size_t poff = 0;
    if ((rc = xt_grow_partials(ctx, 64))) return rc;
    b.d_seqth = nullptr;
    rc = xt_th_run_group(ctx, m, one, threshold, max_nb_states, chunk, G, false, poff, nullptr);  // plans (and evaluates once)
    if (rc) {
        ctx->th_frozen = was_frozen;
        return rc;
    }
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // sequences alive after the last merge step (L - 2); two-position tracks have no merge: the S initial ones
    int32_t hd[2] = {0, m->n_states};
    if (b.L >= 3) XT_HIP(ctx, hipMemcpy(hd, b.th_hdr + (size_t)(b.L - 2) * 2, sizeof(hd), hipMemcpyDeviceToHost));
    const int64_t ncols = (int64_t)hd[1] * G;
    *n_cols_out = ncols;
    if (!lp) {
        ctx->th_frozen = was_frozen;
        return EXTRACK_OK;
    }
    if (n_cols_cap < ncols) {
        ctx->th_frozen = was_frozen;
        return xt_fail(ctx, EXTRACK_E_INVALID, "sequence matrix: output capacity too small");
    }
    const size_t nbytes = (size_t)b.N * (size_t)ncols * sizeof(double);
    if ((rc = xt_reserve_preds(ctx, nbytes))) {
        ctx->th_frozen = was_frozen;
        return rc;
    }
    b.d_seqth = ctx->d_preds;
    b.seqth_stride = (int)ncols;
    ctx->th_frozen = true;  // the plan just made, followed once more with the per-sequence output switched on
    poff = 0;
    rc = xt_th_run_group(ctx, m, one, threshold, max_nb_states, chunk, G, false, poff, nullptr);
    b.d_seqth = nullptr;
    b.seqth_stride = 0;
    ctx->th_frozen = was_frozen;
    if (rc) return rc;
    XT_HIP(ctx, hipMemcpyAsync(lp, ctx->d_preds, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EXTRACK_OK;
}

extern "C" int extrack_loglik_th_async(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk,
                                       double* d_total_ll)
{
    if (!ctx) return EXTRACK_E_INVALID;
    return xt_loglik_th_enqueue(ctx, m, threshold, max_nb_states, chunk, d_total_ll ? d_total_ll : ctx->d_total, false);
}

extern "C" int extrack_loglik_th(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk,
                                 double* total_ll, double* per_track)
{
    if (!ctx || !total_ll) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_loglik_th_enqueue(ctx, m, threshold, max_nb_states, chunk, ctx->d_total, per_track != nullptr);
    if (rc) return rc;
    XT_HIP(ctx, hipMemcpyAsync(ctx->h_total, ctx->d_total, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (per_track) {
        size_t o = 0;
        for (auto& b : ctx->buckets) {
            XT_HIP(ctx, hipMemcpyAsync(per_track + o, b.d_ll, (size_t)b.N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            o += (size_t)b.N;
        }
    }
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_ll = *ctx->h_total;
    return EXTRACK_OK;
}

template <int D, int K>
static hipError_t xt_th_launch_predict(extrack_ctx* ctx, const XtThArgs& a, int grid, int threads, size_t lds)
{
    if (xt_th_pred_waves(a.S) == 3) {
        hipError_t e = xt_th_set_lds(ctx, xt_th_plan_kernel<D, K, true, -1, 3>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((xt_th_plan_kernel<D, K, true, -1, 3>), dim3(grid), dim3(threads), lds, ctx->stream, a);
        return hipGetLastError();
    }
    hipError_t e = xt_th_set_lds(ctx, xt_th_plan_kernel<D, K, true, -1, 4>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((xt_th_plan_kernel<D, K, true, -1, 4>), dim3(grid), dim3(threads), lds, ctx->stream, a);
    return hipGetLastError();
}

extern "C" int extrack_predict_th(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double threshold, int32_t max_nb_states,
                                  int32_t nb_max, double* preds)
{
    if (!ctx || !preds) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    if (m->nb_substeps != 1) return xt_fail(ctx, EXTRACK_E_INVALID, "state predictions require nb_substeps == 1");
    if (nb_max < 1) return xt_fail(ctx, EXTRACK_E_INVALID, "nb_max must be >= 1");
    if (!(threshold >= 0.0)) return xt_fail(ctx, EXTRACK_E_INVALID, "threshold must be >= 0");
    if (m->frame_len <= 1 || m->frame_len > 15) return xt_fail(ctx, EXTRACK_E_INVALID, "frame_len must be in (1, 15]");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XtModelHost mh;
    xt_model_host(m, mh);
    std::vector<double> blob;
    int G = 0;
    std::string err = xt_th_build_blob(mh, blob, G);
    if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
    if ((rc = xt_upload_blob(ctx, blob))) return rc;
    XtBucket& b = ctx->buckets[bucket_id];
    const int S = m->n_states, F = m->frame_len, D = b.D;
    int K;
    if (m->locerr_mode == 0) {
        K = m->locerr_dims;
    } else {
        if (!b.d_sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "per-peak localisation error mode but the bucket has no sigma");
        K = b.KS;
    }
    if (!((K == 1) || (K == D && D > 1))) return xt_fail(ctx, EXTRACK_E_INVALID, "locerr_dims must be 1 or the track dimensionality");
    const size_t nbytes = (size_t)b.N * b.L * S * sizeof(double);
    if ((rc = xt_reserve_preds(ctx, nbytes))) return rc;
    double* d_preds = ctx->d_preds;
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.tracks = b.d_tracks;
    a.sigma = m->locerr_mode ? b.d_sigma : nullptr;
    a.blob = ctx->d_blob;
    if (b.d_dt) {  // per-track time steps: one p_stay table (one blob) per chunk of nb_max tracks; model->p_stay covers ALL buckets
        const int64_t nch = (b.N + nb_max - 1) / nb_max;
        int64_t base = 0, total = 0;
        for (size_t i = 0; i < ctx->buckets.size(); ++i) {
            if ((int)i == bucket_id) base = total;
            total += (ctx->buckets[i].N + nb_max - 1) / nb_max;
        }
        if ((int64_t)m->n_p_stay != total)
            return xt_fail(ctx, EXTRACK_E_INVALID, "per-track time steps: model->n_p_stay must be the number of chunks (one p_stay table per chunk of nb_max tracks, buckets in id order)");
        std::vector<int64_t> tables((size_t)nch);
        for (int64_t c = 0; c < nch; ++c) tables[(size_t)c] = base + c;
        int64_t stride = 0;
        if ((rc = xt_th_chunk_blobs(ctx, m, tables, G, &stride))) return rc;
        a.blob = ctx->d_th_blobs;
        a.blob_stride = stride;
        a.dt = b.d_dt;
    } else if (m->n_p_stay > 1) {
        return xt_fail(ctx, EXTRACK_E_INVALID, "several p_stay tables but no per-track time steps");
    }
    a.preds_out = d_preds;
    a.N = b.N;
    a.L = b.L;
    a.S = S;
    a.NS = 1;
    a.G = G;
    a.F = F;
    a.isBL = (b.L != m->max_len) ? 1 : 0;
    a.min_len = m->min_len;
    a.locerr_mode = m->locerr_mode;
    a.KS = b.KS ? b.KS : 1;
    a.chunk = nb_max;
    a.nchunks = (int32_t)((b.N + nb_max - 1) / nb_max);
    a.max_nb = max_nb_states;
    a.threshold = threshold;
    a.pcap = std::min(nb_max, XT_TH_PILOT);  // slots of per-track state: the pilots, then the other tracks of the chunk 30 at a time
    a.pair_lanes_max_p = ctx->th_pair_lanes;
    int32_t* d_status = nullptr;
    hipError_t e = hipMalloc(&d_status, (size_t)a.nchunks * 4 * sizeof(int32_t));
    if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("predict_th: ") + hipGetErrorString(e));
    a.status = d_status;
    rc = EXTRACK_OK;
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    // Pass 0 (probe): the first chunks with the state in the global workspace -> live-sequence counts of this model.
    // Pass 1: everything with the state in LDS, capacities = 1.5 x the probe's maxima (when that fits ~40 KiB per workgroup).
    // Pass 2 (only after an overflow of pass 1, or when LDS does not fit): everything with the global workspace.
    const int probe_chunks = 512;
    int pass = a.nchunks <= probe_chunks ? 2 : 0, learnP = 0, learnE = 0;
    const int32_t all_chunks = a.nchunks;
    for (;;) {
        int capE = ctx->th_capE;
        while (capE < S * G) capE *= 2;
        ctx->th_capE = capE;
        a.capE = a.wsP = a.wsE = capE;
        a.ws_lds = 0;
        a.nchunks = pass == 0 ? std::min(all_chunks, probe_chunks) : all_chunks;
        a.cmat_words = 0;
        size_t lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, K) * sizeof(double);
        if (pass == 1) {
            const int wp = std::min(capE, std::max(S * G, learnP)) | 1, we = std::min(capE, std::max(S * G, learnE)) | 1;  // odd strides: see the fit-mode launcher
            // bit matrix: only what the probed sequence counts need (a workgroup is one wavefront here: LDS decides how many
            // tracks a CU works on at a time)
            const int cst = (S & (S - 1)) == 0 ? S : 1;
            a.cmat_words = std::min(XT_TH_CMAT_WORDS, std::max(64, we * ((we / cst + 32) >> 5)));
            lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, K, a.cmat_words) * sizeof(double);
            const size_t need = lds + (size_t)xt_th_ws_doubles(wp, we, D, K, F, 1, S, a.pcap, true) * sizeof(double);
            if (need <= 40 * 1024) {
                a.ws_lds = 1;
                a.wsP = wp;
                a.wsE = we;
                lds = need;
            } else {
                pass = 2;
                a.cmat_words = 0;
                lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, K) * sizeof(double);
            }
        }
        const int threads = nb_max <= 2 ? 64 : 256;
        const int grid = (int)std::min<int64_t>(a.nchunks, (int64_t)ctx->n_cu * (threads == 64 ? 4 : 1) * xt_th_pred_waves(S));
        a.ws_stride = xt_th_hist_doubles(a.wsE, a.pcap, true, b.L) + (a.ws_lds ? 0 : xt_th_ws_doubles(a.wsP, a.wsE, D, K, F, 1, S, a.pcap, true));
        const size_t need = (size_t)a.ws_stride * grid * sizeof(double);
        if (need > ctx->th_ws_cap) {
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->d_th_ws) (void)hipFree(ctx->d_th_ws);
            ctx->d_th_ws = nullptr;
            ctx->th_ws_cap = 0;
            if ((e = hipMalloc(&ctx->d_th_ws, need)) != hipSuccess) {
                rc = xt_fail(ctx, EXTRACK_E_HIP, std::string("predict_th workspace: ") + hipGetErrorString(e));
                break;
            }
            ctx->th_ws_cap = need;
        }
        a.ws = ctx->d_th_ws;
        if (lds > 160 * 1024) {
            rc = xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "plan tables do not fit the 160 KiB LDS of a CU");
            break;
        }
        if (D == 1 && K == 1) e = xt_th_launch_predict<1, 1>(ctx, a, grid, threads, lds);
        else if (D == 2 && K == 1) e = xt_th_launch_predict<2, 1>(ctx, a, grid, threads, lds);
        else if (D == 2 && K == 2) e = xt_th_launch_predict<2, 2>(ctx, a, grid, threads, lds);
        else if (D == 3 && K == 1) e = xt_th_launch_predict<3, 1>(ctx, a, grid, threads, lds);
        else e = xt_th_launch_predict<3, 3>(ctx, a, grid, threads, lds);
        if (e == hipSuccess) {
            ctx->th_status_host.resize((size_t)a.nchunks * 4);
            e = hipMemcpyAsync(ctx->th_status_host.data(), d_status, (size_t)a.nchunks * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            rc = xt_fail(ctx, EXTRACK_E_HIP, std::string("predict_th: ") + hipGetErrorString(e));
            break;
        }
        int over = 0, maxE = 0, maxG = 0;
        for (int c = 0; c < a.nchunks; ++c) {
            over |= ctx->th_status_host[(size_t)c * 4];
            maxE = std::max(maxE, ctx->th_status_host[(size_t)c * 4 + 1]);
            maxG = std::max(maxG, ctx->th_status_host[(size_t)c * 4 + 2]);
        }
        if (over && a.ws_lds) {  // the probe's capacities were too small for some track: global workspace for all
            pass = 2;
            continue;
        }
        if (over) {
            int ncap = capE;
            while (ncap < std::max(maxE, maxG)) ncap *= 2;
            if (ncap == capE) ncap *= 2;
            if (ncap > XT_TH_MAXCAP) {
                rc = xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "more than 8192 live state sequences per step (threshold fusion expands every sequence by n_states^nb_substeps before it merges): raise threshold, lower max_nb_states or nb_substeps - or use the fixed-window kernel (fusion='window' / extrack_loglik), which serves this model");
                break;
            }
            ctx->th_capE = ncap;
            continue;
        }
        if (pass == 0) {
            learnP = maxG + maxG / 2 + 2;
            learnE = maxE + maxE / 2 + 2;
            pass = 1;
            continue;
        }
        break;
    }
    a.nchunks = all_chunks;
    if (rc == EXTRACK_OK) {
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        ctx->timed = true;
        e = hipMemcpyAsync(preds, d_preds, nbytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = xt_fail(ctx, EXTRACK_E_HIP, std::string("predict_th: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_status);
    return rc;
}


// ------------------------------------------------------------------------------------------------
// position refinement (extrack/refined_localization.py:207-338): two recording passes of the prediction-mode plan kernel
// (xt_th.h, refine mode) + the combination of the "future" and "past" predictions of every position
// ------------------------------------------------------------------------------------------------
struct XtRefineArgs {
    const double* tracks;  // [N][L][D] original time order (rows of this launch)
    const double* fut;     // records of the pass over the time-reversed track: entry e = state after positions L-1 .. L-1-e
    const double* past;    // records of the pass over the track as it is:      entry e = state after positions 0 .. e
    const uint8_t* fut_new;
    const uint8_t* past_new;
    const int32_t* fut_cnt;
    const int32_t* past_cnt;
    double* mu_out;        // [N][L][D]
    double* sig_out;       // [N][L]
    int64_t N;             // rows of this launch (the records are [L - 1][cap][2 + D][N]: a wavefront reads 64 neighbouring tracks' values of a field)
    int32_t L, S, cap_f, cap_p;  // sequences recorded per (entry, track) by the two passes
    double l2;             // squared localisation error (global), or
    const double* sigma;   // per-peak localisation errors [N][L] of these rows (nullptr: the global one)
    double logF[XT_MAX_STATES];
    // the mixture itself (get_pos_PDF's return values, refined_localization.py:298), xt_refine_components only: component j of position k at
    // row comp_off[k] + j of means [.][N][D], stds [.][N], logw [.][N]
    const int64_t* comp_off;
    double* comp_mean;
    double* comp_std;
    double* comp_logw;
};

// One thread per (track, position): softmax-weighted mean of the pair means / root mean of the pair variances
// (refined_localization.py:222-298 get_pos_PDF + :329-337).  ONE sweep over the pairs with a running maximum of the log-weights (the sums
// are rescaled when it grows): every record is read once.  Adjacent threads serve adjacent tracks of the same position, so a wavefront
// walks 64 neighbouring record rows.
template <int D>
__global__ void __launch_bounds__(256) xt_refine_combine(XtRefineArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N * a.L) return;
    const int k = (int)(i / a.N);
    const int64_t x = i - (int64_t)k * a.N;
    const int L = a.L, R = 2 + D;
    double c[D];
    for (int d = 0; d < D; ++d) c[d] = a.tracks[(x * L + k) * D + d];
    // this position's own localisation variance (get_pos_PDF, refined_localization.py:222, 271, 289).  Per-peak errors: the in-place update
    // of the last record inside get_LC_Km_Ks (:186-193) takes the error of index len - 1 of the array it was given - for position 0 (pass
    // over the unreversed array) that is the LAST position's error, reproduced as it is; for position len - 1 it is its own
    const double l2k = a.sigma ? a.sigma[x * L + k] * a.sigma[x * L + k] : a.l2;
    const double l2q = a.sigma ? a.sigma[x * L + (L - 1)] * a.sigma[x * L + (L - 1)] : a.l2;
    double wmax = -INFINITY, sw = 0.0, smu[D], ssg = 0.0;
    for (int d = 0; d < D; ++d) smu[d] = 0.0;
    auto add = [&](double w, const double* mu, double var) {
        if (w > wmax) {  // rescale what has been summed to the new maximum (exp(-inf) = 0 the first time, when the sums are 0 anyway)
            const double sc = exp(wmax - w);
            sw *= sc;
            ssg *= sc;
            for (int d = 0; d < D; ++d) smu[d] *= sc;
            wmax = w;
        }
        const double p = exp(w - wmax);
        sw += p;
        for (int d = 0; d < D; ++d) smu[d] += p * mu[d];
        ssg += p * var;
    };
    // record field f of sequence q of entry e of this thread's track: rec[((e * cap + q) * R + f) * N + x]
    if (k == 0 || k == L - 1) {
        // end positions: one pass only; the reference's last record already carries the density of this position (and the
        // initial fractions for position 0) through its in-place update (refined_localization.py:188-193), and get_pos_PDF adds the
        // overlap term once more
        const int cap = k == 0 ? a.cap_f : a.cap_p;
        const double* rec = (k == 0 ? a.fut : a.past) + ((int64_t)(L - 2) * cap * R) * a.N + x;
        const uint8_t* nw = (k == 0 ? a.fut_new : a.past_new) + (int64_t)(L - 2) * cap;
        const int n = (k == 0 ? a.fut_cnt : a.past_cnt)[L - 2];
        for (int q = 0; q < n; ++q) {
            const double* r = rec + (int64_t)q * R * a.N;
            const double lp = r[0], sd = r[(int64_t)(1 + D) * a.N], v = sd * sd + l2k, vq = sd * sd + l2q;
            double dsq = 0.0, mu[D];
            for (int d = 0; d < D; ++d) {
                const double m = r[(int64_t)(1 + d) * a.N];
                dsq += (c[d] - m) * (c[d] - m);
                mu[d] = (m * l2k + c[d] * sd * sd) / v;
            }
            const double lk = -0.5 * D * log(2.0 * M_PI * v) - dsq / (2.0 * v);       // get_pos_PDF's overlap term
            const double lkq = -0.5 * D * log(2.0 * M_PI * vq) - dsq / (2.0 * vq);    // the in-place update of the last record
            add(lp + lk + lkq + (k == 0 ? a.logF[nw[q]] : 0.0), mu, l2k * sd * sd / v);
        }
    } else {
        const double* rf = a.fut + ((int64_t)(L - 2 - k) * a.cap_f * R) * a.N + x;
        const double* rp = a.past + ((int64_t)(k - 1) * a.cap_p * R) * a.N + x;
        const uint8_t* nf = a.fut_new + (int64_t)(L - 2 - k) * a.cap_f;
        const uint8_t* np_ = a.past_new + (int64_t)(k - 1) * a.cap_p;
        const int n1 = a.fut_cnt[L - 2 - k], n2 = a.past_cnt[k - 1];
        for (int q1 = 0; q1 < n1; ++q1) {
            const double* r1 = rf + (int64_t)q1 * R * a.N;
            const double lp1 = r1[0], s1 = r1[(int64_t)(1 + D) * a.N];
            const double v12 = s1 * s1 + l2k, vA = s1 * s1 * l2k / v12;
            double muA[D], d1 = 0.0;
            for (int d = 0; d < D; ++d) {
                const double m1 = r1[(int64_t)(1 + d) * a.N];
                muA[d] = (m1 * l2k + c[d] * s1 * s1) / v12;
                d1 += (m1 - c[d]) * (m1 - c[d]);
            }
            const double lk1 = -0.5 * D * log(2.0 * M_PI * v12) - d1 / (2.0 * v12);
            for (int q2 = 0; q2 < n2; ++q2) {
                if (np_[q2] != nf[q1]) continue;  // pairs that agree on the state at this position
                const double* r2 = rp + (int64_t)q2 * R * a.N;
                const double s3 = r2[(int64_t)(1 + D) * a.N], v3 = vA + s3 * s3;
                double d2 = 0.0, mu[D];
                for (int d = 0; d < D; ++d) {
                    const double m3 = r2[(int64_t)(1 + d) * a.N];
                    d2 += (muA[d] - m3) * (muA[d] - m3);
                    mu[d] = (muA[d] * s3 * s3 + m3 * vA) / v3;
                }
                add(lp1 + r2[0] + lk1 - 0.5 * D * log(2.0 * M_PI * v3) - d2 / (2.0 * v3), mu, vA * s3 * s3 / v3);
            }
        }
    }
    for (int d = 0; d < D; ++d) a.mu_out[(x * L + k) * D + d] = smu[d] / sw;
    a.sig_out[x * L + k] = sqrt(ssg / sw);
}

// The Gaussian mixture of every position as the reference returns it from get_pos_PDF (refined_localization.py:207-298): same pair walk as
// xt_refine_combine, in the reference's component order - end positions: the sequences of the pass's last record; positions between: for
// every state s, (sequences from the future whose state at this position is s) x (sequences from the past with state s), the former outer.
// For inspection of small inputs (every component of every track goes through HBM); the refinement proper never materialises them.
template <int D>
__global__ void __launch_bounds__(256) xt_refine_components(XtRefineArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N * a.L) return;
    const int k = (int)(i / a.N);
    const int64_t x = i - (int64_t)k * a.N;
    const int L = a.L, R = 2 + D;
    double c[D];
    for (int d = 0; d < D; ++d) c[d] = a.tracks[(x * L + k) * D + d];
    const double l2k = a.sigma ? a.sigma[x * L + k] * a.sigma[x * L + k] : a.l2;
    const double l2q = a.sigma ? a.sigma[x * L + (L - 1)] * a.sigma[x * L + (L - 1)] : a.l2;
    int64_t row = a.comp_off[k];
    // the recording kernel keeps the -dims/2 log(2 pi) of every integration step out of the weights (the likelihood kernels add them once per
    // track): the records this position combines went through len - 2 (end positions) or len - 3 steps together
    const double wconst = -0.5 * D * log(2.0 * M_PI) * (double)((k == 0 || k == L - 1) ? L - 2 : L - 3);
    auto put = [&](double w, const double* mu, double var) {
        for (int d = 0; d < D; ++d) a.comp_mean[(row * a.N + x) * D + d] = mu[d];
        a.comp_std[row * a.N + x] = sqrt(var);
        a.comp_logw[row * a.N + x] = w + wconst;
        ++row;
    };
    if (k == 0 || k == L - 1) {
        const int cap = k == 0 ? a.cap_f : a.cap_p;
        const double* rec = (k == 0 ? a.fut : a.past) + ((int64_t)(L - 2) * cap * R) * a.N + x;
        const uint8_t* nw = (k == 0 ? a.fut_new : a.past_new) + (int64_t)(L - 2) * cap;
        const int n = (k == 0 ? a.fut_cnt : a.past_cnt)[L - 2];
        for (int q = 0; q < n; ++q) {
            const double* r = rec + (int64_t)q * R * a.N;
            const double lp = r[0], sd = r[(int64_t)(1 + D) * a.N], v = sd * sd + l2k, vq = sd * sd + l2q;
            double dsq = 0.0, mu[D];
            for (int d = 0; d < D; ++d) {
                const double m = r[(int64_t)(1 + d) * a.N];
                dsq += (c[d] - m) * (c[d] - m);
                mu[d] = (m * l2k + c[d] * sd * sd) / v;
            }
            const double lk = -0.5 * D * log(2.0 * M_PI * v) - dsq / (2.0 * v);
            const double lkq = -0.5 * D * log(2.0 * M_PI * vq) - dsq / (2.0 * vq);
            // the pass from the past runs with neutral initial fractions 1 / S (refined_localization.py:216): a constant the read-out cancels
            put(lp + lk + lkq + (k == 0 ? a.logF[nw[q]] : -log((double)a.S)), mu, l2k * sd * sd / v);
        }
    } else {
        const double* rf = a.fut + ((int64_t)(L - 2 - k) * a.cap_f * R) * a.N + x;
        const double* rp = a.past + ((int64_t)(k - 1) * a.cap_p * R) * a.N + x;
        const uint8_t* nf = a.fut_new + (int64_t)(L - 2 - k) * a.cap_f;
        const uint8_t* np_ = a.past_new + (int64_t)(k - 1) * a.cap_p;
        const int n1 = a.fut_cnt[L - 2 - k], n2 = a.past_cnt[k - 1];
        for (int st = 0; st < a.S; ++st)
            for (int q1 = 0; q1 < n1; ++q1) {
                if (nf[q1] != st) continue;
                const double* r1 = rf + (int64_t)q1 * R * a.N;
                const double lp1 = r1[0], s1 = r1[(int64_t)(1 + D) * a.N];
                const double v12 = s1 * s1 + l2k, vA = s1 * s1 * l2k / v12;
                double muA[D], d1 = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double m1 = r1[(int64_t)(1 + d) * a.N];
                    muA[d] = (m1 * l2k + c[d] * s1 * s1) / v12;
                    d1 += (m1 - c[d]) * (m1 - c[d]);
                }
                const double lk1 = -0.5 * D * log(2.0 * M_PI * v12) - d1 / (2.0 * v12);
                for (int q2 = 0; q2 < n2; ++q2) {
                    if (np_[q2] != st) continue;
                    const double* r2 = rp + (int64_t)q2 * R * a.N;
                    const double s3 = r2[(int64_t)(1 + D) * a.N], v3 = vA + s3 * s3;
                    double d2 = 0.0, mu[D];
                    for (int d = 0; d < D; ++d) {
                        const double m3 = r2[(int64_t)(1 + d) * a.N];
                        d2 += (muA[d] - m3) * (muA[d] - m3);
                        mu[d] = (muA[d] * s3 * s3 + m3 * vA) / v3;
                    }
                    put(lp1 + r2[0] + lk1 - 0.5 * D * log(2.0 * M_PI * v3) - d2 / (2.0 * v3), mu, vA * s3 * s3 / v3);
                }
            }
    }
}

// Time-reversed copy of a bucket [N][L][D] on the device (the pass "from the future" walks the track backwards).
__global__ void __launch_bounds__(256) xt_reverse_tracks(const double* __restrict__ src, double* __restrict__ dst, int64_t N, int L, int D)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * L * D) return;
    const int64_t x = i / ((int64_t)L * D);
    const int r = (int)(i - x * L * D), p = r / D, d = r - p * D;
    dst[i] = src[(x * L + (L - 1 - p)) * D + d];
}

// Grow-only device buffers of the refinement path, kept in the context between calls: a hipMalloc / hipFree pair per record array and
// call cost more than the kernels (r02: 0.25 s wall for 30 ms of kernels on 1e5 x 30).
static int xt_rf_reserve(extrack_ctx* ctx, int slot, size_t bytes)
{
    if (bytes <= ctx->rf_cap_bytes[slot]) return EXTRACK_OK;
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->rf_buf[slot]) (void)hipFree(ctx->rf_buf[slot]);
    ctx->rf_buf[slot] = nullptr;
    ctx->rf_cap_bytes[slot] = 0;
    hipError_t e = hipMalloc(&ctx->rf_buf[slot], bytes);
    if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("refinement buffer (") + std::to_string(bytes >> 20) + " MiB): " + hipGetErrorString(e));
    ctx->rf_cap_bytes[slot] = bytes;
    return EXTRACK_OK;
}
enum { XT_RF_REV = 0, XT_RF_REC0, XT_RF_REC1, XT_RF_NEW0, XT_RF_NEW1, XT_RF_CNT0, XT_RF_CNT1, XT_RF_MU, XT_RF_SIG, XT_RF_STATUS };

// One launch of the recording kernel over bucket `d_tracks` ([N][L][D] on the device).  rows == 0: capacity probe on the pilot tracks
// (nothing recorded; *cap_out = sequences to record per entry); else: the tracks [row0, row0 + rows) are recorded into a.rf_out.
static int xt_refine_launch(extrack_ctx* ctx, const extrack_model* m, const double* d_tracks, const double* d_sigma, int64_t N, int L, int D, double threshold,
                            int32_t max_nb_states, int64_t row0, int64_t rows, int rf_cap, double* d_rec, uint8_t* d_new, int32_t* d_cnt, int* cap_out)
{
    const int S = m->n_states, F = m->frame_len, G = S;
    XtThArgs a;
    memset(&a, 0, sizeof(a));
    a.tracks = d_tracks;
    a.blob = ctx->d_blob;
    a.L = L;
    a.S = S;
    a.NS = 1;
    a.G = G;
    a.F = F;
    a.isBL = 0;
    a.min_len = L + 2;  // no field-of-view / bleaching factors in the recorded weights
    a.sigma = d_sigma;  // per-peak errors [N][L][1], read at the SAME index as the position of d_tracks (see extrack_refine_positions)
    a.locerr_mode = d_sigma ? 1 : 0;
    a.KS = 1;
    a.max_nb = max_nb_states;
    a.threshold = threshold;
    a.pcap = (int)std::min<int64_t>(N, XT_TH_PILOT);
    a.pair_lanes_max_p = ctx->th_pair_lanes;
    a.refine = 1;
    int rc = xt_rf_reserve(ctx, XT_RF_STATUS, 4 * sizeof(int32_t));
    if (rc) return rc;
    int32_t* d_status = (int32_t*)ctx->rf_buf[XT_RF_STATUS];
    a.status = d_status;
    const bool probe = rows == 0;
    if (!probe) {
        a.rf_cap = rf_cap;
        a.rf_out = d_rec;
        a.rf_new = d_new;
        a.rf_cnt = d_cnt;
        a.rf_row0 = row0;
        a.rf_rows = rows;
    }
    hipError_t e = hipSuccess;
    for (;;) {
        int capE = ctx->th_capE;
        while (capE < S * G) capE *= 2;
        ctx->th_capE = capE;
        a.capE = a.wsP = a.wsE = capE;
        a.ws_lds = 0;
        a.N = probe ? std::min<int64_t>(N, XT_TH_PILOT) : N;
        a.chunk = (int32_t)std::min<int64_t>(a.N, (int64_t)1 << 30);
        a.nchunks = 1;
        a.cmat_words = 0;
        const int64_t first = std::max<int64_t>(row0, a.pcap), last = std::min<int64_t>(N, row0 + rows);
        const int64_t nbatch = (!probe && last > first) ? (last - first + a.pcap - 1) / a.pcap : 0;
        const int grid = probe ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(nbatch, (int64_t)ctx->n_cu * xt_th_pred_waves(S)));
        const size_t lds = (size_t)xt_th_plan_lds_doubles(S, G, capE, D, 1) * sizeof(double);
        a.ws_stride = xt_th_hist_doubles(a.wsE, a.pcap, true, L) + xt_th_ws_doubles(a.wsP, a.wsE, D, 1, F, 1, S, a.pcap, true);
        const size_t need = (size_t)a.ws_stride * grid * sizeof(double);
        if (need > ctx->th_ws_cap) {
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->d_th_ws) (void)hipFree(ctx->d_th_ws);
            ctx->d_th_ws = nullptr;
            ctx->th_ws_cap = 0;
            if ((e = hipMalloc(&ctx->d_th_ws, need)) != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("refinement workspace: ") + hipGetErrorString(e));
            ctx->th_ws_cap = need;
        }
        a.ws = ctx->d_th_ws;
        if (lds > 160 * 1024) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "plan tables do not fit the 160 KiB LDS of a CU");
        if (D == 1) e = xt_th_launch_predict<1, 1>(ctx, a, grid, 256, lds);
        else if (D == 2) e = xt_th_launch_predict<2, 1>(ctx, a, grid, 256, lds);
        else e = xt_th_launch_predict<3, 1>(ctx, a, grid, 256, lds);
        int32_t st[4] = {0, 0, 0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(st, d_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("refinement pass: ") + hipGetErrorString(e));
        if (st[0]) {  // capacity overflow: grow and repeat
            int ncap = capE;
            while (ncap < std::max(st[1], st[2])) ncap *= 2;
            if (ncap == capE) ncap *= 2;
            if (ncap > XT_TH_MAXCAP)
                return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "more than 8192 live state sequences per step (threshold fusion expands every sequence by n_states^nb_substeps before it merges): raise threshold, lower max_nb_states or nb_substeps - or use the fixed-window kernel (fusion='window' / extrack_loglik), which serves this model");
            ctx->th_capE = ncap;
            continue;
        }
        if (cap_out) *cap_out = std::max(std::max(st[1], st[2]), S * G);
        return EXTRACK_OK;
    }
}

// Request for the mixture components (extrack_refine_pos_pdf); nullptr: the refined positions only.
struct XtPdfOut {
    int32_t* counts;   // [L] components per position (always filled)
    int64_t capacity;  // rows the three arrays below hold
    double* means;     // nullptr: counts only
    double* stds;
    double* logw;
};

static int xt_refine_run(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double threshold, int32_t max_nb_states, double* mu, double* sigma,
                         const XtPdfOut* pdf)
{
    int rc = xt_validate_model(ctx, m);
    if (rc) return rc;
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    if (m->nb_substeps != 1) return xt_fail(ctx, EXTRACK_E_INVALID, "position refinement is defined for nb_substeps == 1");
    if (m->locerr_mode == 2 || (m->locerr_mode == 0 && m->locerr_dims != 1))
        return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "position refinement takes one global localisation error or per-peak errors [n][len][1] (what the reference's reshapes carry through)");
    if (!(threshold >= 0.0)) return xt_fail(ctx, EXTRACK_E_INVALID, "threshold must be >= 0");
    if (m->frame_len <= 1 || m->frame_len > 15) return xt_fail(ctx, EXTRACK_E_INVALID, "frame_len must be in (1, 15]");
    XtBucket& b = ctx->buckets[bucket_id];
    const int S = m->n_states, L = b.L, D = b.D, R = 2 + D;
    if (L < 2) return xt_fail(ctx, EXTRACK_E_INVALID, "position refinement needs tracks of at least 2 positions");
    if (m->locerr_mode == 1 && (!b.d_sigma || b.KS != 1))
        return xt_fail(ctx, EXTRACK_E_INVALID, "position refinement with per-peak errors needs the bucket's sigma [n][len][1]");
    // Per-peak errors (refined_localization.py:59-70): get_LC_Km_Ks reverses the error array but walks an UNREVERSED track from its end, so the
    // k-th position it injects meets the error of index k counted from the START of the array it was given - the same array in both passes
    // (:211, :216).  Here both passes walk their track from index 0, the pass "from the future" on the time-reversed copy: handing BOTH the
    // bucket's sigma as it is reproduces exactly that pairing (mirrored errors in the pass from the future, the right ones from the past).
    const double* d_sig_in = m->locerr_mode == 1 ? b.d_sigma : nullptr;
    XT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nel = (size_t)b.N * L * D;
    // time-reversed copy of the bucket for the pass "from the future", made on the device
    if ((rc = xt_rf_reserve(ctx, XT_RF_REV, nel * sizeof(double)))) return rc;
    double* d_rev = (double*)ctx->rf_buf[XT_RF_REV];
    hipLaunchKernelGGL(xt_reverse_tracks, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, ctx->stream, b.d_tracks, d_rev, b.N, L, D);
    XT_HIP(ctx, hipGetLastError());
    // pass 0: from the future (reversed track, the matrix as given, refined_localization.py:211); pass 1: from the past (track as it is,
    // transposed matrix, :213-216).  No initial fractions in the recorded weights (:93).
    std::vector<double> ones(S, 1.0), Tt((size_t)S * S);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) Tt[(size_t)i * S + j] = m->TrMat[(size_t)j * S + i];
    std::vector<double> blobs[2];
    for (int pass = 0; pass < 2; ++pass) {
        XtModelHost mh;
        xt_model_host(m, mh);
        mh.Fs = ones.data();
        mh.TrMat = pass == 0 ? m->TrMat : Tt.data();
        int G = 0;
        std::string err = xt_th_build_blob(mh, blobs[pass], G);
        if (!err.empty()) return xt_fail(ctx, EXTRACK_E_INVALID, err);
    }
    const double* src[2] = {d_rev, b.d_tracks};
    // capacity probes on the pilot tracks: sequences to record per entry of either pass
    int cap[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        if ((rc = xt_upload_blob(ctx, blobs[pass]))) return rc;
        if ((rc = xt_refine_launch(ctx, m, src[pass], d_sig_in, b.N, L, D, threshold, max_nb_states, 0, 0, 0, nullptr, nullptr, nullptr, &cap[pass]))) return rc;
    }
    // row blocks: both passes' records of a block stay within the memory budget (EXTRACK_REFINE_BUDGET_MB, default 16 GiB of the 288 GB);
    // the merge plan only depends on the pilot tracks, which every launch re-walks, so the blocks are independent
    size_t budget = (size_t)16 << 30;
    if (const char* ev = getenv("EXTRACK_REFINE_BUDGET_MB")) {
        const long v = atol(ev);
        if (v >= 1) budget = (size_t)v << 20;
    }
    const size_t per_row = (size_t)(L - 1) * (size_t)(cap[0] + cap[1]) * R * sizeof(double);
    int64_t RB = (int64_t)std::max<size_t>(XT_TH_PILOT, budget / per_row);
    RB = std::min<int64_t>(RB, b.N);
    if (pdf && RB < b.N)
        return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "the mixture components are returned for buckets whose records fit ONE row block (EXTRACK_REFINE_BUDGET_MB, default 16 GiB): pass fewer tracks");
    for (int pass = 0; pass < 2; ++pass) {
        if ((rc = xt_rf_reserve(ctx, XT_RF_REC0 + pass, (size_t)(L - 1) * (size_t)RB * cap[pass] * R * sizeof(double)))) return rc;
        if ((rc = xt_rf_reserve(ctx, XT_RF_NEW0 + pass, (size_t)(L - 1) * cap[pass]))) return rc;
        if ((rc = xt_rf_reserve(ctx, XT_RF_CNT0 + pass, (size_t)(L - 1) * sizeof(int32_t)))) return rc;
    }
    if ((rc = xt_rf_reserve(ctx, XT_RF_MU, nel * sizeof(double)))) return rc;
    if ((rc = xt_rf_reserve(ctx, XT_RF_SIG, (size_t)b.N * L * sizeof(double)))) return rc;
    double* d_mu = (double*)ctx->rf_buf[XT_RF_MU];
    double* d_sig = (double*)ctx->rf_buf[XT_RF_SIG];
    XT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int64_t row0 = 0; row0 < b.N; row0 += RB) {
        const int64_t rows = std::min<int64_t>(RB, b.N - row0);
        for (int pass = 0; pass < 2; ++pass) {
            if ((rc = xt_upload_blob(ctx, blobs[pass]))) return rc;
            if ((rc = xt_refine_launch(ctx, m, src[pass], d_sig_in, b.N, L, D, threshold, max_nb_states, row0, rows, cap[pass], (double*)ctx->rf_buf[XT_RF_REC0 + pass],
                                       (uint8_t*)ctx->rf_buf[XT_RF_NEW0 + pass], (int32_t*)ctx->rf_buf[XT_RF_CNT0 + pass], nullptr)))
                return rc;
        }
        XtRefineArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.tracks = b.d_tracks + (size_t)row0 * L * D;
        ra.fut = (const double*)ctx->rf_buf[XT_RF_REC0];
        ra.past = (const double*)ctx->rf_buf[XT_RF_REC1];
        ra.fut_new = (const uint8_t*)ctx->rf_buf[XT_RF_NEW0];
        ra.past_new = (const uint8_t*)ctx->rf_buf[XT_RF_NEW1];
        ra.fut_cnt = (const int32_t*)ctx->rf_buf[XT_RF_CNT0];
        ra.past_cnt = (const int32_t*)ctx->rf_buf[XT_RF_CNT1];
        ra.mu_out = d_mu + (size_t)row0 * L * D;
        ra.sig_out = d_sig + (size_t)row0 * L;
        ra.N = rows;
        ra.L = L;
        ra.S = S;
        ra.cap_f = cap[0];
        ra.cap_p = cap[1];
        ra.l2 = m->locerr[0] * m->locerr[0];
        ra.sigma = d_sig_in ? d_sig_in + (size_t)row0 * L : nullptr;
        for (int s2 = 0; s2 < S; ++s2) ra.logF[s2] = log(m->Fs[s2]);
        const int grid = (int)(((int64_t)rows * L + 255) / 256);
        if (!pdf) {
            if (D == 1) hipLaunchKernelGGL(xt_refine_combine<1>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            else if (D == 2) hipLaunchKernelGGL(xt_refine_combine<2>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            else hipLaunchKernelGGL(xt_refine_combine<3>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            XT_HIP(ctx, hipGetLastError());
            continue;
        }
        // ---- mixture components (one row block): count them on the host from the passes' plans, then one thread per (track, position)
        std::vector<int32_t> cnt[2];
        std::vector<uint8_t> nw[2];
        for (int pass = 0; pass < 2; ++pass) {
            cnt[pass].resize(L - 1);
            nw[pass].resize((size_t)(L - 1) * cap[pass]);
            XT_HIP(ctx, hipMemcpyAsync(cnt[pass].data(), ctx->rf_buf[XT_RF_CNT0 + pass], (size_t)(L - 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            XT_HIP(ctx, hipMemcpyAsync(nw[pass].data(), ctx->rf_buf[XT_RF_NEW0 + pass], nw[pass].size(), hipMemcpyDeviceToHost, ctx->stream));
        }
        XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int64_t> off(L + 1, 0);
        for (int k = 0; k < L; ++k) {
            int64_t n = 0;
            if (k == 0 || k == L - 1) {
                n = cnt[k == 0 ? 0 : 1][L - 2];
            } else {
                for (int st = 0; st < S; ++st) {
                    int64_t n1 = 0, n2 = 0;
                    for (int q = 0; q < cnt[0][L - 2 - k]; ++q) n1 += nw[0][(size_t)(L - 2 - k) * cap[0] + q] == st;
                    for (int q = 0; q < cnt[1][k - 1]; ++q) n2 += nw[1][(size_t)(k - 1) * cap[1] + q] == st;
                    n += n1 * n2;
                }
            }
            if (n > INT32_MAX) return xt_fail(ctx, EXTRACK_E_UNSUPPORTED, "too many mixture components at one position");
            pdf->counts[k] = (int32_t)n;
            off[k + 1] = off[k] + n;
        }
        if (!pdf->means) continue;
        if (off[L] > pdf->capacity) return xt_fail(ctx, EXTRACK_E_INVALID, "mixture component arrays too small (sum of the counts of a counts-only call)");
        const size_t rows_c = (size_t)off[L] * (size_t)rows;
        double* d_comp = nullptr;
        int64_t* d_off = nullptr;
        hipError_t e = hipMalloc(&d_comp, std::max<size_t>(rows_c, 1) * (D + 2) * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&d_off, (size_t)(L + 1) * sizeof(int64_t));
        if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), (size_t)(L + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            ra.comp_off = d_off;
            ra.comp_mean = d_comp;
            ra.comp_std = d_comp + rows_c * D;
            ra.comp_logw = d_comp + rows_c * (D + 1);
            if (D == 1) hipLaunchKernelGGL(xt_refine_components<1>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            else if (D == 2) hipLaunchKernelGGL(xt_refine_components<2>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            else hipLaunchKernelGGL(xt_refine_components<3>, dim3(grid), dim3(256), 0, ctx->stream, ra);
            e = hipGetLastError();
        }
        if (e == hipSuccess && rows_c) e = hipMemcpyAsync(pdf->means, ra.comp_mean, rows_c * D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rows_c) e = hipMemcpyAsync(pdf->stds, ra.comp_std, rows_c * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rows_c) e = hipMemcpyAsync(pdf->logw, ra.comp_logw, rows_c * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_comp);
        (void)hipFree(d_off);
        if (e != hipSuccess) return xt_fail(ctx, EXTRACK_E_HIP, std::string("mixture components: ") + hipGetErrorString(e));
    }
    XT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    if (!pdf) {
        XT_HIP(ctx, hipMemcpyAsync(mu, d_mu, nel * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        XT_HIP(ctx, hipMemcpyAsync(sigma, d_sig, (size_t)b.N * L * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EXTRACK_OK;
}

extern "C" int extrack_refine_positions(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double threshold, int32_t max_nb_states,
                                        double* mu, double* sigma)
{
    if (!ctx || !mu || !sigma) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    return xt_refine_run(ctx, m, bucket_id, threshold, max_nb_states, mu, sigma, nullptr);
}

extern "C" int extrack_refine_pos_pdf(extrack_ctx* ctx, const extrack_model* m, int32_t bucket_id, double threshold, int32_t max_nb_states,
                                      int32_t* counts, int64_t capacity, double* means, double* stds, double* logw)
{
    if (!ctx || !counts || capacity < 0 || (means && (!stds || !logw))) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    XtPdfOut pdf = {counts, capacity, means, stds, logw};
    return xt_refine_run(ctx, m, bucket_id, threshold, max_nb_states, nullptr, nullptr, &pdf);
}

extern "C" int extrack_th_plan_step(extrack_ctx* ctx, int32_t bucket_id, int64_t chunk_index, int32_t t, int32_t* n_expanded,
                                    int32_t* n_groups, uint16_t* members, uint16_t* gstart, int32_t cap)
{
    if (!ctx || !n_expanded || !n_groups) return xt_fail(ctx, EXTRACK_E_INVALID, "null argument");
    if (bucket_id < 0 || bucket_id >= (int)ctx->buckets.size()) return xt_fail(ctx, EXTRACK_E_INVALID, "bucket id out of range");
    XtBucket& b = ctx->buckets[bucket_id];
    if (!b.th_members) return xt_fail(ctx, EXTRACK_E_INVALID, "no threshold-fusion evaluation has run on this bucket");
    if (chunk_index < 0 || chunk_index >= b.th_nchunks || t < 1 || t > b.L - 1) return xt_fail(ctx, EXTRACK_E_INVALID, "chunk or step out of range");
    XT_HIP(ctx, hipSetDevice(ctx->device));
    XT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int32_t h[2];
    XT_HIP(ctx, hipMemcpy(h, b.th_hdr + ((size_t)chunk_index * b.L + t) * 2, sizeof(h), hipMemcpyDeviceToHost));
    *n_expanded = h[0];
    *n_groups = h[1];
    if (members && gstart && h[1] > 0) {
        if (cap < h[0] || cap < h[1] + 1) return xt_fail(ctx, EXTRACK_E_INVALID, "output capacity too small");
        XT_HIP(ctx, hipMemcpy(members, b.th_members + ((size_t)chunk_index * b.L + t) * b.th_capE, (size_t)h[0] * sizeof(uint16_t), hipMemcpyDeviceToHost));
        XT_HIP(ctx, hipMemcpy(gstart, b.th_gstart + ((size_t)chunk_index * b.L + t) * (b.th_capE + 1), (size_t)(h[1] + 1) * sizeof(uint16_t),
                              hipMemcpyDeviceToHost));
    }
    return EXTRACK_OK;
}

extern "C" int extrack_last_kernel_ms(extrack_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return EXTRACK_E_INVALID;
    if (!ctx->timed) return xt_fail(ctx, EXTRACK_E_INVALID, "no timed launch yet");
    XT_HIP(ctx, hipEventSynchronize(ctx->ev1));
    XT_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return EXTRACK_OK;
}

extern "C" int extrack_last_launch_info(const extrack_ctx* ctx, int32_t info[6])
{
    if (!ctx || !info) return EXTRACK_E_INVALID;
    for (int i = 0; i < 6; ++i) info[i] = ctx->launch_info[i];
    return EXTRACK_OK;
}

extern "C" int extrack_p_stay_table(const double* ds, int32_t S, int32_t ns, const double* cell_dims, int32_t n_cell, double* out)
{
    if (!ds || !out || S < 1 || ns < 1 || (n_cell > 0 && !cell_dims)) return EXTRACK_E_INVALID;
    int G = 1;
    for (int i = 0; i < ns; ++i) G *= S;
    for (int r = 0; r < G; ++r) {
        double sub = 0.0;
        int rr = r;
        for (int c = 0; c < ns; ++c) {
            sub += ds[rr % S] * ds[rr % S];
            rr /= S;
        }
        const double sd = sqrt(sub / ns) + 1e-200;
        double p = 1.0;
        for (int j = 0; j < n_cell; ++j) {
            const double cl = cell_dims[j];
            const double x0 = cl / 2000.0, x1 = cl - cl / 2000.0;
            double acc = 0.0;
            for (int i = 0; i < 1000; ++i) {
                const double x = x0 + (x1 - x0) * (double)i / 999.0;
                // Phi(z) = erfc(-z / sqrt(2)) / 2
                acc += 0.5 * erfc(-((cl - x) / sd) * M_SQRT1_2) - 0.5 * erfc((x / sd) * M_SQRT1_2);
            }
            p *= acc / 1000.0;
        }
        out[r] = p;
    }
    return EXTRACK_OK;
}
