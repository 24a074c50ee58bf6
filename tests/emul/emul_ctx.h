// TEST INFRASTRUCTURE ONLY: the CPU-thread execution context the kernel bodies are compiled against in tests/emul (one std::thread per
// GPU thread, pthread barriers for __syncthreads and for the lock-step of a wavefront).  Shared by emul.cpp and emul_r2.cpp.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

static inline bool xt_emul_poison()
{
    const char* e = getenv("XT_EMUL_POISON");
    return !(e && e[0] == '0');
}

struct HostCtx {
    int tid_, nthreads_, block_, nblocks_;
    double* smem_;
    pthread_barrier_t* bar_;
    pthread_barrier_t* wbar_ = nullptr;  // this wave's barrier
    double* wscr_ = nullptr;             // this wave's 64-entry shuffle scratch
    int lane() const { return tid_ & 63; }
    int uniform(int v) const { return v; }
    int wave_in_block() const { return tid_ >> 6; }
    int waves_per_block() const { return nthreads_ >> 6; }
    void wave_sync() { pthread_barrier_wait(wbar_); }
    double shfl_xor_f64(double v, int m)
    {
        wscr_[lane()] = v;
        pthread_barrier_wait(wbar_);
        double o = wscr_[lane() ^ m];
        pthread_barrier_wait(wbar_);
        return o;
    }
    int shfl_xor_i32(int v, int m) { return (int)shfl_xor_f64((double)v, m); }
    // lane-pair primitives of the register-resident 2-state path (xt_reg2.h; device versions: DevCtx in xt_host.h)
    template <int BIT>
    static constexpr bool pair_natural() { return BIT >= 4; }
    template <int BIT>
    double xor_f64(double v) { return shfl_xor_f64(v, 1 << BIT); }
    template <int BIT>
    int xor_i32(int v) { return shfl_xor_i32(v, 1 << BIT); }
    template <int BIT>
    void pair_exchange(double& a, double& b)
    {
        if (pair_natural<BIT>()) {  // what v_permlane16/32_swap do: lane bit 0 keeps a and gets the partner's a; bit 1 keeps b, gets the partner's b
            const double pa = xor_f64<BIT>(a), pb = xor_f64<BIT>(b);
            if (((lane() >> BIT) & 1) == 0) b = pa;
            else a = pb;
        } else {
            b = xor_f64<BIT>(b);
        }
    }
    template <int BIT>
    void pair_exchange_i32(int& a, int& b)
    {
        double da = a, db = b;
        pair_exchange<BIT>(da, db);
        a = (int)da;
        b = (int)db;
    }
    template <int GP>
    double group_sum_f64(double v)
    {
        for (int m = 1; m < GP; m <<= 1) v += shfl_xor_f64(v, m);
        return v;
    }
    template <int GP>
    int group_max_i32(int v)
    {
        for (int m = 1; m < GP; m <<= 1) {
            const int o = shfl_xor_i32(v, m);
            v = o > v ? o : v;
        }
        return v;
    }
    unsigned long long ballot(bool flag)
    {
        wscr_[lane()] = flag ? 1.0 : 0.0;
        pthread_barrier_wait(wbar_);
        unsigned long long m = 0;
        for (int i = 0; i < 64; ++i)
            if (wscr_[i] != 0.0) m |= 1ull << i;
        pthread_barrier_wait(wbar_);
        return m;
    }
    int wave_rank(bool flag, int& total)
    {
        wscr_[lane()] = flag ? 1.0 : 0.0;
        pthread_barrier_wait(wbar_);
        int r = 0, t = 0;
        for (int i = 0; i < 64; ++i) {
            const int f = wscr_[i] != 0.0;
            t += f;
            r += (i < lane()) ? f : 0;
        }
        pthread_barrier_wait(wbar_);
        total = t;
        return r;
    }
    int tid() const { return tid_; }
    int nthreads() const { return nthreads_; }
    int block() const { return block_; }
    int nblocks() const { return nblocks_; }
    double* smem() const { return smem_; }
    void sync() { pthread_barrier_wait(bar_); }
    void atomic_max_i32(int* p, int v)
    {
        int old = __atomic_load_n(p, __ATOMIC_RELAXED);
        while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
        }
    }
    void atomic_or_u32(uint32_t* p, uint32_t v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
    void atomic_add_f64(double* p, double v)
    {
        uint64_t* pi = (uint64_t*)p;
        uint64_t old = __atomic_load_n(pi, __ATOMIC_RELAXED);
        for (;;) {
            double d;
            memcpy(&d, &old, 8);
            d += v;
            uint64_t nw;
            memcpy(&nw, &d, 8);
            if (__atomic_compare_exchange_n(pi, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
        }
    }
};

// One emulated launch: nblocks workgroups of `threads` threads, block by block, each with per-wavefront barriers.
template <class Body>
static void th_emul_blocks(int nblocks, int threads, size_t lds_doubles, Body body)
{
    const int nw = threads / 64;
    for (int b = 0; b < nblocks; ++b) {
        std::vector<double> smem(lds_doubles + 16, xt_emul_poison() ? NAN : 0.0);  // the device does not clear LDS: start from NaN (XT_EMUL_POISON=0: from zeros)
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
                body(cx);
            });
        for (auto& x : th) x.join();
        pthread_barrier_destroy(&bar);
        for (auto& x : wb) pthread_barrier_destroy(&x);
    }
}

