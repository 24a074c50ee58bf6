// Host-side state of libextrack_hip.so shared by its translation units (extrack_hip.hip: likelihood / posterior / threshold-fusion
// entry points; extrack_grad.hip: likelihood + gradient; extrack_hist.hip: state-duration histograms) and the device-side
// execution context the kernel bodies are written against.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/extrack_hip.h"
#include "xt_kernel.h"
#include "xt_tables.h"
#include "xt_th.h"

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------
extern __shared__ double xt_smem[];

struct DevCtx {
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ int block() const { return blockIdx.x; }
    __device__ __forceinline__ int nblocks() const { return gridDim.x; }
    __device__ __forceinline__ double* smem() const { return xt_smem; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    // promise that v is the same in every lane of the wavefront (moves it to an SGPR: scalar loads, scalar address math)
    __device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ int wave_in_block() const { return threadIdx.x >> 6; }
    __device__ __forceinline__ int waves_per_block() const { return blockDim.x >> 6; }
    // LDS operations of one wavefront execute in order; only the compiler must be kept from moving
    // LDS accesses across the point where other lanes' data is consumed.
    __device__ __forceinline__ void wave_sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ unsigned long long ballot(bool flag) { return __ballot(flag); }
    // number of lanes below this one with flag set; total = lanes of the wave with flag set
    __device__ __forceinline__ int wave_rank(bool flag, int& total)
    {
        const unsigned long long b = __ballot(flag);
        total = __popcll(b);
        return __popcll(b & ((1ull << (threadIdx.x & 63)) - 1ull));
    }
    // All-reduce over aligned groups of GP lanes (power of two) without the LDS crossbar where the hardware allows it: DPP lane
    // permutations inside a row of 16 lanes (quad swaps, then the mirrored half-row / row: any pairing of the two halves works for an
    // all-reduce and both lanes of a pair compute the same sum), ds_swizzle for the neighbouring row, v_readlane for the wave halves.
    template <int CTRL>
    __device__ static __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
    template <int CTRL>
    __device__ static __forceinline__ double dpp_f64(double v)
    {
        return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
    }
    template <int GP>
    __device__ __forceinline__ double group_sum_f64(double v)
    {
        if (GP >= 2) v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
        if (GP >= 4) v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
        if (GP >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror
        if (GP >= 16) v += dpp_f64<0x140>(v); // row_mirror
        if (GP >= 32)
            v += __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F));
        if (GP >= 64) {
            const double lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
            const double hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
            v = lo + hi;
        }
        return v;
    }
    template <int GP>
    __device__ __forceinline__ int group_max_i32(int v)
    {
        auto mx = [](int a, int b) { return a > b ? a : b; };
        if (GP >= 2) v = mx(v, dpp_i32<0xB1>(v));
        if (GP >= 4) v = mx(v, dpp_i32<0x4E>(v));
        if (GP >= 8) v = mx(v, dpp_i32<0x141>(v));
        if (GP >= 16) v = mx(v, dpp_i32<0x140>(v));
        if (GP >= 32) v = mx(v, __builtin_amdgcn_ds_swizzle(v, 0x401F));
        if (GP >= 64) v = mx(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
        return v;
    }
    // ---- lane-pair primitives of the register-resident 2-state path (xt_reg2.h).  BIT = lane-id bit of the pair (partner = lane ^ 2^BIT).
    // xor_*: the partner's value.  Bits 0, 1, 3: one DPP move per dword; bit 2: row_shl:4 / row_shr:4 under complementary bank masks;
    // bit 4: ds_swizzle (xor 16 inside 32 lanes); bit 5: v_permlane32_swap of two copies.
    template <int BIT>
    __device__ static __forceinline__ int xor_i32(int v)
    {
        if (BIT == 0) return dpp_i32<0xB1>(v);   // quad_perm [1,0,3,2]
        if (BIT == 1) return dpp_i32<0x4E>(v);   // quad_perm [2,3,0,1]
        if (BIT == 2) {
            const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);  // quads 0, 2 read lane + 4
            return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xA, false);          // quads 1, 3 read lane - 4
        }
        if (BIT == 3) return dpp_i32<0x128>(v);  // row_ror:8
        if (BIT == 4) return __builtin_amdgcn_ds_swizzle(v, 0x401F);
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
    }
    template <int BIT>
    __device__ static __forceinline__ double xor_f64(double v)
    {
        return __hiloint2double(xor_i32<BIT>(__double2hiint(v)), xor_i32<BIT>(__double2loint(v)));
    }
    // pair_exchange: the 2 x 2 transpose of a step.  Before: a, b = the lane's two new sequences (new digit qa for a, 1 - qa for b); after:
    // a, b = the two sequences of the lane's next group (newest digit = the lane's bit BIT).  "Natural" bits (4, 5): qa = 0 and the
    // permlane swap instructions transpose directly; the other bits: qa = the lane's own bit, a stays and b is traded with the partner.
    template <int BIT>
    __device__ static constexpr bool pair_natural() { return BIT >= 4; }
    template <int BIT>
    __device__ static __forceinline__ void pair_exchange_i32(int& a, int& b)
    {
        if (BIT == 4) {
            const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
            a = (int)r[0];
            b = (int)r[1];
        } else if (BIT == 5) {
            const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
            a = (int)r[0];
            b = (int)r[1];
        } else {
            b = xor_i32<BIT>(b);
        }
    }
    template <int BIT>
    __device__ static __forceinline__ void pair_exchange(double& a, double& b)
    {
        int ah = __double2hiint(a), al = __double2loint(a), bh = __double2hiint(b), bl = __double2loint(b);
        pair_exchange_i32<BIT>(ah, bh);
        pair_exchange_i32<BIT>(al, bl);
        a = __hiloint2double(ah, al);
        b = __hiloint2double(bh, bl);
    }
    __device__ __forceinline__ int shfl_xor_i32(int v, int m) { return __shfl_xor(v, m, 64); }
    __device__ __forceinline__ double shfl_xor_f64(double v, int m) { return __shfl_xor(v, m, 64); }
    __device__ __forceinline__ void atomic_max_i32(int* p, int v)
    {
        __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void atomic_or_u32(uint32_t* p, uint32_t v)
    {
        __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void atomic_add_f64(double* p, double v)
    {
        __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// MAXT: 256 for the common one-wave-set-per-few-tracks geometry (lets the allocator use up to 256
// VGPRs at 2 waves/SIMD if it needs them), 1024 when one track's groups need more than 256 threads.
// Fused total of a likelihood launch (XtKernelArgs::done): called by every likelihood kernel after its body.  The block's partial sum
// is in a.partials[blockIdx.x] (written by one of its threads); a device-scope counter finds the LAST block of the grid to get here, and
// that block sums all partials in a fixed order (thread t takes the partials t, t + NT, ... sequentially; lanes by the DPP all-reduce;
// wavefronts sequentially) - deterministic for a given launch geometry, like the separate xt_reduce_partials launch it replaces.
// Visibility across the per-XCD L2s (MI355X_MICROARCH.md, "Correctness boundaries"): producer = workgroup barrier, then one thread
// re-stores the partial with an agent-scope (sc1) store, drains it and increments the counter; consumer = agent-scope acquire fence after
// the increment that saw the last count, and every partial is read with an agent-scope (sc1) load.
__device__ __forceinline__ void xt_fused_total(const XtKernelArgs& a)
{
    if (!a.done) return;  // kernel argument: uniform over the grid
    __syncthreads();      // the body is done with the LDS and this block's partial sum has been stored (HIP's barrier drains vmcnt first)
    int* flag = (int*)(xt_smem + 16);
    if (threadIdx.x == 0) {
        // hand-off without an L2 write-back per block (an agent-scope release fence costs one): the block's partial is re-stored with an
        // agent-scope (sc1, write-through) store and drained (vmcnt(0)) before the counter increment - the guide's "every store of the
        // handed-off bytes sc1 and drained" form; the consumer reads with sc1 loads
        const double mine = a.partials[blockIdx.x];
        __hip_atomic_store(a.partials + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store is acknowledged
        const unsigned int prev = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev + 1u == gridDim.x;
        if (last) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const int n = (int)gridDim.x, nt = (int)blockDim.x;
    double s = 0.0;
    // eight loads in flight per thread (a dependent add after every single load would cost one memory round trip per partial)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * nt) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * nt;
            v[k] = i < n ? __hip_atomic_load(a.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    DevCtx cx;
    s = cx.group_sum_f64<64>(s);
    __syncthreads();  // every thread has read the flag
    if ((threadIdx.x & 63) == 0) xt_smem[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) tot += xt_smem[w];
        *a.total_out = tot;
        if (a.total_host) __hip_atomic_store(a.total_host, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch (stream-ordered)
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct XtBucket {
    const double* d_tracks = nullptr;
    const double* d_sigma = nullptr;
    double* d_dt = nullptr;  // [N][L] per-track time steps (extrack_set_bucket_dt), owned
    double sig_min = NAN, sig_max = NAN;  // range of the per-peak localisation errors (host uploads only; NaN: unknown)
    bool owned = false;
    int64_t N = 0;
    int L = 0, D = 0, KS = 0;
    double* d_ll = nullptr;  // per-track output, allocated on first request
    // threshold-fusion plan of the last extrack_loglik_th call (xt_th.h)
    uint16_t* th_members = nullptr;
    uint32_t* th_mpack = nullptr;
    uint8_t* th_gnew = nullptr;
    uint16_t* th_gstart = nullptr;
    int32_t* th_hdr = nullptr;
    int32_t* th_status = nullptr;
    int th_capE = 0, th_chunk = 0;
    int64_t th_nchunks = 0;
    double* d_seqth = nullptr;  // transient (extrack_sequence_matrix_th): per-sequence output of the apply kernel, not owned by the bucket
    int seqth_stride = 0;
    int th_maxG = -1, th_sumE = 0;  // of that plan: largest group count of a step / largest sum of expanded sequences over the steps, over the bucket's chunks (-1: no valid plan)
};

struct extrack_ctx {
    int device = 0;
    int n_cu = 0;
    int oversub = 8;  // block generations per CU (EXTRACK_OVERSUB overrides; tuning knob)
    int ll_reg2 = 1;  // 2-state likelihood: 1 = register-resident kernel (xt_reg2.h), 0 = LDS-resident (xt_fast2.h); EXTRACK_LL_PATH=reg2|lds
    int grad_reg2 = 1;  // gradient kernels: 1 = register-resident where built (xt_reg2.h for 2 states, else xt_gradr.h), 0 = the LDS-resident xt_grad.h
                        // only, 2 = xt_gradr.h before xt_reg2.h (tests); EXTRACK_GRAD_PATH = reg2 | lds | gradr
    int grad_rev = 1;   // reverse-mode kernels (xt_rev.h) for 3 / 4 members per group: 1 = where they win, 0 = never, 2 = wherever built; EXTRACK_GRAD_PATH = rev
    int rev_oversub = 16;  // block generations per CU of the reverse-mode launch (each block owns a log region: fewer blocks, smaller cache footprint)
    size_t rev_log_mb = 16384;   // budget of those log regions (EXTRACK_REV_LOG_MB): the launch uses fewer blocks to stay within it
    double* d_revlog = nullptr;  // merged-state logs of the reverse-mode kernels
    size_t revlog_cap = 0;       // doubles
    double* d_revadj = nullptr;  // adjoint of the model blob [TB]
    size_t revadj_cap = 0;
    int gradr_npc = 0;  // directions per pass of the xt_gradr.h kernels (0: chosen by the launcher; EXTRACK_GRADR_NPC = 3 | 4 also forces these kernels for small models)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::vector<XtBucket> buckets;
    XtConfig cfg;
    int32_t* d_base_tab = nullptr;
    int32_t* d_off_tab = nullptr;
    // model blob: two (pinned host, device) staging slots used alternately, each guarded by an event recorded after its
    // host->device copy, so that an evaluation never waits for the previous one (extrack_loglik_async stays asynchronous)
    double* d_blob = nullptr;       // slot in use by the evaluation being enqueued
    double* d_blob_s[2] = {nullptr, nullptr};
    double* h_blob_s[2] = {nullptr, nullptr};  // pinned
    hipEvent_t ev_blob[2] = {nullptr, nullptr};
    bool blob_busy[2] = {false, false};
    size_t blob_cap = 0;
    unsigned blob_turn = 0;
    double* d_preds = nullptr;      // posterior output buffer, kept between extrack_predict / extrack_predict_th calls
    size_t preds_cap = 0;
    double* d_th_blobs = nullptr;   // threshold-fusion path with per-track time steps: one table blob per chunk
    size_t th_blobs_cap = 0;        // doubles
    double* d_dblob = nullptr;      // gradient path: tangent tables [n_dir][TB]
    size_t dblob_cap = 0;           // doubles
    double* d_dblob2 = nullptr;     // the same blocks in launch order (2-state kernels: full directions first, then the uniform ones)
    size_t dblob2_cap = 0;
    double* h_dblob = nullptr;      // pinned staging of the tangent tables (read by an asynchronous copy)
    size_t h_dblob_cap = 0;
    hipEvent_t ev_dblob = nullptr;  // recorded after that copy: the next evaluation waits for it before refilling the staging buffer
    bool dblob_busy = false;
    double* d_gpartials = nullptr;  // gradient path: per-block partial sums [grid][NP + 1] + the reduced row
    size_t gpartials_cap = 0;       // doubles
    float grad_ms = 0.f;            // device time of the gradient kernels of the last extrack_loglik_grad call
    hipEvent_t evg0 = nullptr, evg1 = nullptr;  // bracket the kernels of the last gradient evaluation
    bool grad_timed = false;        // ... whose time has not been read yet
    double* d_gout = nullptr;       // gradient path: {sum LL, gradient} of the synchronous entry point
    size_t gout_cap = 0;
    double* d_gtmp = nullptr;       // ... of the launch groups after the first (added to the result in stream order)
    size_t gtmp_cap = 0;
    double* d_partials = nullptr;
    size_t partials_cap = 0;
    static constexpr int RF_SLOTS = 10;   // position refinement: grow-only device buffers kept between calls (extrack_hip.hip: XT_RF_*)
    void* rf_buf[RF_SLOTS] = {nullptr};
    size_t rf_cap_bytes[RF_SLOTS] = {0};
    double* d_total = nullptr;
    double* h_total = nullptr;  // pinned
    double* h_total_dev = nullptr;   // the same word as the device sees it (the fused total is written there by the kernel itself)
    unsigned int* d_done = nullptr;  // block counter of the fused total (zero between launches)
    bool blob_inline = false;        // the model blob of the current evaluation travels in the kernel arguments (xt_prepare)
    bool fused_host = false;         // the last xt_loglik_enqueue left the total in h_total (no device-to-host copy needed)
    bool oversub_forced = false;     // EXTRACK_OVERSUB given: no adaptive block count for small launches
    bool no_fused = false;           // EXTRACK_NO_FUSED=1: separate reduction launch + copies, as before round 4 (A/B)
    std::vector<XtBucketDesc> desc_shadow;  // what d_desc[0 .. size) holds: unchanged descriptors are not copied again
    XtBucketDesc* d_desc = nullptr;  // [XT_DESC_CAP] bucket descriptors of the launches of one evaluation
    XtBucketDesc* h_desc = nullptr;  // pinned staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int32_t launch_info[6] = {0, 0, 0, 0, 0, 0};
    std::map<std::pair<const void*, std::pair<int, size_t>>, int> occ_cache;
    double* d_th_ws = nullptr;  // plan-kernel workspace
    size_t th_ws_cap = 0;
    int th_capE = 128;          // plan capacity (expanded sequences per step); grows on overflow
    int th_learnP = 0, th_learnE = 0;  // live parent / expanded sequence counts seen by the last plan (+ headroom): LDS workspace sizing
    std::vector<int32_t> th_status_host;
    int32_t* h_th_status = nullptr;  // pinned: plan status of every chunk of a launch group
    int32_t* d_th_status = nullptr;
    size_t th_status_cap = 0;        // ints
    XtThBucket* d_th_desc = nullptr;  // bucket descriptors of a launch group
    size_t th_desc_cap = 0;
    int32_t* d_th_cend = nullptr;     // chunk prefix of a launch group
    size_t th_cend_cap = 0;
    // more sets of the per-launch buffers above + side streams: several launch groups of one evaluation in flight (xt_th_use_slot)
    struct ThSlot {
        int32_t* h_status = nullptr;
        int32_t* d_status = nullptr;
        size_t status_cap = 0;
        XtThBucket* d_desc = nullptr;
        size_t desc_cap = 0;
        int32_t* d_cend = nullptr;
        size_t cend_cap = 0;
        double* d_ws = nullptr;
        size_t ws_cap = 0;
    };
    static constexpr int TH_SLOTS = 3;
    ThSlot th_slot[TH_SLOTS];   // parked sets; the current one lives in the fields above
    int th_cur_slot = 0;
    hipStream_t th_streams[TH_SLOTS] = {nullptr, nullptr, nullptr};
    hipEvent_t th_ev[TH_SLOTS + 1] = {nullptr, nullptr, nullptr, nullptr};
    bool th_split_active = false, th_no_split = false;  // EXTRACK_TH_NO_SPLIT=1: never run several launch groups concurrently
    int th_learnP_split = 0, th_learnE_split = 0;
    // buckets longer than th_split_pct[0] % of the longest track length form the first group, longer than th_split_pct[1] % the second
    // (0: no third group), the rest the last (EXTRACK_TH_SPLIT_PCT="hi,lo")
    int th_split_pct[2] = {50, 25};
    float th_plan_ms = 0.f;
    int th_force_single = 0;
    int th_pair_lanes = 4;  // EXTRACK_TH_PAIR_LANES
    int th_stage_in_lds_mode = 0;  // EXTRACK_TH_STAGE_LDS: LDS-typed copy of the pilot means/stds also when the state is in LDS (measured: no gain)
    int th_no_gen_single = 0;  // EXTRACK_TH_NO_GEN_SINGLE: never use the one-buffer general apply variant
    int th_plan_bs = 0;         // plan kernel, > 64 sequences: pivot rows per batch = wavefronts x max(n, 1); < 0: one batch (EXTRACK_TH_PLAN_BS)
    // Frozen plan (extrack_th_freeze_plan): threshold-fusion evaluations skip the plan kernel and follow the plan the last planning
    // evaluation left in the buckets; per launch group (keyed by its first bucket and size) the sequence counts that size the apply / gradient launch
    double* d_big_ws = nullptr;  // per-wavefront sequence state of the global-memory kernel for big models (xt_big.h)
    size_t big_ws_cap = 0;       // doubles
    bool th_frozen = false;  // the per-bucket sequence counts that size the apply / gradient launch: XtBucket::th_maxG, th_sumE
    std::vector<double> blob_host;  // model tables of the current fixed-window evaluation (xt_prepare)
    bool th_plan_threads_forced = false;
    int th_plan_threads = 512;  // workgroup size of the plan kernel (EXTRACK_TH_PLAN_THREADS)
    int th_force_tt = 0, th_force_threads = 0, th_oversub = 2;  // tuning knobs (EXTRACK_TH_TT / _THREADS / _OVERSUB)
    std::string err;
};

static const int XT_DESC_CAP = 4096;  // bucket descriptors per evaluation (buckets beyond 64 per launch group are chunked)


#define XT_HIP(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return EXTRACK_E_HIP;                                                               \
        }                                                                                       \
    } while (0)

// shared host helpers (defined in extrack_hip.hip)
int xt_fail(extrack_ctx* ctx, int code, const std::string& msg);
int xt_validate_model(extrack_ctx* ctx, const extrack_model* m);
void xt_model_host(const extrack_model* m, XtModelHost& mh);
int xt_upload_blob(extrack_ctx* ctx, const std::vector<double>& blob);   // -> ctx->d_blob (double-buffered staging)
int xt_prepare_config(extrack_ctx* ctx, const extrack_model* m);         // digit-slot tables of (S, ns, F) -> ctx->cfg, d_base_tab, d_off_tab
int xt_reserve_partials(extrack_ctx* ctx, size_t n);
size_t xt_desc_base(const extrack_ctx* ctx);
size_t xt_max_grid(const extrack_ctx* ctx);
__global__ void xt_reduce_partials(const double* __restrict__ partials, int n, double* __restrict__ out);
const void* xt_r2_kernel(int F, int D, int K, int NP);  // extrack_reg2.hip: register-resident 2-state kernels, nullptr = not built
const void* xt_rev_kernel_ptr(int G, int D, int K, int nbuf);  // extrack_rev.hip: reverse-mode gradient kernels (xt_rev.h), 1 | 2 exchange buffers
// threshold-fusion plan stage for other translation units (extrack_hip.hip): `cb` gets, per launch group, the kernel arguments with the plan
// made (a.buckets / a.chunk_end on the device), the track / error dimensionality, the largest group count of its chunks and the longest length
typedef std::function<int(XtThArgs& a, int D, int K, int maxG, int Lmax)> XtThAfterPlan;
int xt_th_plan_groups(extrack_ctx* ctx, const extrack_model* m, double threshold, int32_t max_nb_states, int32_t chunk, const XtThAfterPlan& cb);
// column sums of per-block partials [nrows][ncol] (extrack_grad.hip): column 0 -> *ll_dst, column 1 + i -> out[i]
void xt_grad_reduce_launch(hipStream_t st, const double* partials, int nrows, int ncol, double* ll_dst, double* out);
void xt_rev_project(hipStream_t st, const double* adj, const double* dblob, int TB, int n_dir, double* out);  // out[i] = <adj, dblob[i]>
const void* xt_gradr_kernel_ptr(int G, int D, int K, int NPC);  // extrack_gradr.hip: register-resident gradient kernels (xt_gradr.h), NPC = 3 | 4
