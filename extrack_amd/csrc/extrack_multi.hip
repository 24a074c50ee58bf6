// libextrack_hip.so, translation unit: ONE PROCESS, SEVERAL GPUs - the piece of the boundary a host without torch.distributed needs
// (option B).  The reference's only parallelism is multiprocessing.Pool.map over track chunks with the per-chunk results
// concatenated and summed (extrack/tracking.py:1061-1069); here every device keeps a contiguous row range of every bucket resident, an
// evaluation enqueues the likelihood kernel on every device's own stream and ends with ONE all-reduce of the scalar over RCCL (xGMI) -
// ncclCommInitAll over the devices of the node, ncclAllReduce(sum, double, count = 1) per evaluation inside a group call.
// RCCL is loaded at run time (dlopen: the library stays loadable on hosts without it, and a process that already carries PyTorch's copy
// gets that one); when it is not available - or the same device is listed twice, which RCCL refuses and the single-GPU tests use - the
// per-device totals are summed on the host from the pinned words the kernels write: same value, no collective.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the functions are resolved with dlsym, the library is not a link-time dependency

#include "xt_host.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (h) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd;
    }
};
Rccl g_rccl;
}  // namespace

struct extrack_multi {
    std::vector<extrack_ctx*> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;   // empty: host-side sum
    std::vector<double*> d_tot;     // one device word per rank: local sum in, all-reduced sum out
    std::vector<double*> h_tot;     // pinned read-back per rank
    std::string err;
    int64_t n_tracks = 0;
};

static std::string g_multi_err;
static int xm_fail(extrack_multi* m, int code, const std::string& s)
{
    if (m) m->err = s;
    else g_multi_err = s;
    return code;
}

extern "C" const char* extrack_multi_last_error(const extrack_multi* m) { return m ? m->err.c_str() : g_multi_err.c_str(); }

extern "C" void extrack_multi_destroy(extrack_multi* m)
{
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        if (i < m->comm.size() && m->comm[i]) (void)g_rccl.CommDestroy(m->comm[i]);
        if (i < m->d_tot.size() && m->d_tot[i]) (void)hipFree(m->d_tot[i]);
        if (i < m->h_tot.size() && m->h_tot[i]) (void)hipHostFree(m->h_tot[i]);
        if (m->ctx[i]) extrack_destroy(m->ctx[i]);
    }
    delete m;
}

extern "C" int extrack_multi_create(int32_t n_devices, const int32_t* device_ids, int32_t use_rccl, extrack_multi** out)
{
    if (!out || n_devices < 1 || !device_ids) return xm_fail(nullptr, EXTRACK_E_INVALID, "extrack_multi_create: null argument / no device");
    *out = nullptr;
    extrack_multi* m = new extrack_multi();
    bool dup = false;
    for (int i = 0; i < n_devices; ++i) {
        for (int j = 0; j < i; ++j) dup = dup || device_ids[j] == device_ids[i];
        extrack_ctx* c = nullptr;
        const int rc = extrack_create(device_ids[i], &c);
        if (rc) {
            g_multi_err = std::string("extrack_multi_create: device ") + std::to_string(device_ids[i]) + ": " + extrack_last_error(nullptr);
            extrack_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
        m->dev.push_back(device_ids[i]);
        double *d = nullptr, *h = nullptr;
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipMalloc(&d, 2 * sizeof(double)) != hipSuccess ||
            hipHostMalloc(&h, 2 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            g_multi_err = "extrack_multi_create: buffer allocation failed";
            if (d) (void)hipFree(d);
            extrack_multi_destroy(m);
            return EXTRACK_E_HIP;
        }
        m->d_tot.push_back(d);
        m->h_tot.push_back(h);
    }
    if (use_rccl && n_devices > 1 && !dup) {
        if (!g_rccl.load()) {
            if (use_rccl > 1) {  // 2: RCCL demanded
                g_multi_err = "extrack_multi_create: librccl.so could not be loaded";
                extrack_multi_destroy(m);
                return EXTRACK_E_UNSUPPORTED;
            }
        } else {
            m->comm.assign((size_t)n_devices, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, m->dev.data());
            if (r != ncclSuccess) {
                g_multi_err = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
                m->comm.clear();
                extrack_multi_destroy(m);
                return EXTRACK_E_HIP;
            }
        }
    }
    *out = m;
    return EXTRACK_OK;
}

extern "C" int32_t extrack_multi_device_count(const extrack_multi* m) { return m ? (int32_t)m->ctx.size() : EXTRACK_E_INVALID; }
extern "C" int32_t extrack_multi_uses_rccl(const extrack_multi* m) { return m ? (m->comm.empty() ? 0 : 1) : EXTRACK_E_INVALID; }
extern "C" extrack_ctx* extrack_multi_context(extrack_multi* m, int32_t i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

// Rows [start, stop) of a bucket of n tracks owned by rank r of w: contiguous, balanced to one row (extrack_amd.distributed.shard_range)
static void xm_range(int64_t n, int r, int w, int64_t& a, int64_t& z)
{
    const int64_t base = n / w, rem = n % w;
    a = (int64_t)r * base + std::min<int64_t>(r, rem);
    z = a + base + (r < rem ? 1 : 0);
}

extern "C" int extrack_multi_upload_bucket(extrack_multi* m, const double* tracks, int64_t n, int32_t len, int32_t dims, const double* sigma,
                                           int32_t sigma_dims)
{
    if (!m || !tracks || n < 0 || len < 2 || dims < 1) return xm_fail(m, EXTRACK_E_INVALID, "extrack_multi_upload_bucket: bad argument");
    const int w = (int)m->ctx.size();
    for (int r = 0; r < w; ++r) {
        int64_t a, z;
        xm_range(n, r, w, a, z);
        if (z <= a) continue;  // a rank may own no row of a small bucket: it contributes nothing for it
        int32_t id = -1;
        const int rc = extrack_upload_bucket(m->ctx[r], tracks + (size_t)a * len * dims, z - a, len, dims,
                                             sigma ? sigma + (size_t)a * len * sigma_dims : nullptr, sigma_dims, &id);
        if (rc) return xm_fail(m, rc, std::string("device ") + std::to_string(m->dev[r]) + ": " + extrack_last_error(m->ctx[r]));
    }
    m->n_tracks += n;
    return EXTRACK_OK;
}

extern "C" int extrack_multi_clear_buckets(extrack_multi* m)
{
    if (!m) return EXTRACK_E_INVALID;
    for (auto* c : m->ctx) extrack_clear_buckets(c);
    m->n_tracks = 0;
    return EXTRACK_OK;
}

// One evaluation of sum(LL) over every bucket on every device: model->min_len / max_len are the dataset-global ones, as for a shard of
// the multi-process path.  Every device's kernel is enqueued first (all GPUs work concurrently), then the collective, then ONE read-back.
extern "C" int extrack_multi_loglik(extrack_multi* m, const extrack_model* model, double* total_ll)
{
    if (!m || !model || !total_ll) return xm_fail(m, EXTRACK_E_INVALID, "extrack_multi_loglik: null argument");
    const int w = (int)m->ctx.size();
    std::vector<char> has(w, 0);
    for (int r = 0; r < w; ++r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess) return xm_fail(m, EXTRACK_E_HIP, "hipSetDevice");
        has[r] = extrack_bucket_count(m->ctx[r]) > 0;
        if (has[r]) {
            const int rc = extrack_loglik_async(m->ctx[r], model, m->d_tot[r]);
            if (rc) return xm_fail(m, rc, std::string("device ") + std::to_string(m->dev[r]) + ": " + extrack_last_error(m->ctx[r]));
        } else if (hipMemsetAsync(m->d_tot[r], 0, sizeof(double), m->ctx[r]->stream) != hipSuccess) {
            return xm_fail(m, EXTRACK_E_HIP, "hipMemsetAsync");
        }
    }
    if (!m->comm.empty()) {
        ncclResult_t r0 = g_rccl.GroupStart();
        for (int r = 0; r < w && r0 == ncclSuccess; ++r) r0 = g_rccl.AllReduce(m->d_tot[r], m->d_tot[r], 1, ncclDouble, ncclSum, m->comm[r], m->ctx[r]->stream);
        const ncclResult_t r1 = g_rccl.GroupEnd();
        if (r0 != ncclSuccess || r1 != ncclSuccess) return xm_fail(m, EXTRACK_E_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r0 ? r0 : r1) : "error"));
        if (hipSetDevice(m->dev[0]) != hipSuccess || hipMemcpyAsync(m->h_tot[0], m->d_tot[0], sizeof(double), hipMemcpyDeviceToHost, m->ctx[0]->stream) != hipSuccess)
            return xm_fail(m, EXTRACK_E_HIP, "read-back of the reduced sum");
        for (int r = 0; r < w; ++r)  // every rank's collective has completed before the buffers are reused
            if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamSynchronize(m->ctx[r]->stream) != hipSuccess) return xm_fail(m, EXTRACK_E_HIP, "hipStreamSynchronize");
        *total_ll = *m->h_tot[0];
        return EXTRACK_OK;
    }
    // no communicator (one device, RCCL absent, or the same device listed twice): fixed-order host sum of the per-device totals
    for (int r = 0; r < w; ++r)
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipMemcpyAsync(m->h_tot[r], m->d_tot[r], sizeof(double), hipMemcpyDeviceToHost, m->ctx[r]->stream) != hipSuccess)
            return xm_fail(m, EXTRACK_E_HIP, "read-back of a device total");
    double s = 0.0;
    for (int r = 0; r < w; ++r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamSynchronize(m->ctx[r]->stream) != hipSuccess) return xm_fail(m, EXTRACK_E_HIP, "hipStreamSynchronize");
        s += *m->h_tot[r];
    }
    *total_ll = s;
    return EXTRACK_OK;
}
