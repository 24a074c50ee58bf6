// Wave-synchronous fast path of the track-likelihood recursion for TWO-STATE models with one
// substep (S = 2, ns = 1: BASELINE configs[0], [1], [3]) and frame_len F <= 7, log-likelihood only.
//
// Same mathematics as xt_kernel.h (fuse -> expand -> integrate with extended-range linear weights);
// what differs is the mapping onto the machine:
//   * a track's 2^(F-1) groups are lanes of ONE wavefront (64 / 2^(F-1) tracks per wave), so a step
//     needs no workgroup barrier - LDS operations of a wave complete in order;
//   * the step loop is unrolled over the F phases of the circular digit buffer, so every LDS address
//     is a loop-invariant register (no per-step index arithmetic, no index tables);
//   * sequence storage is XOR-swizzled (index ^= 31 when bit 5 is set) so that the 32 lanes of an
//     LDS access group hit 32 distinct 8-byte bank pairs in every phase (v1: 53 % of LDS cycles
//     were bank conflicts);
//   * the track's positions are staged through LDS in 32-position chunks with coalesced loads (one
//     HBM read of the track, nothing else), and read back as a broadcast;
//   * the three reciprocals of a step (1/W, 1/den_0, 1/den_1) come from ONE v_rcp_f64 of the
//     product; means are carried as M/W implicitly until the single multiply that stores them;
//   * per-track log() is replaced by a running product of the (mantissa, exponent) likelihoods -
//     one log per track slot at the end of the kernel (unless per-track output is requested).
#pragma once
#include "xt_kernel.h"

#define XT_F2_CHUNK 32  // positions staged per refill
#ifndef XT_F2_RENORM
#define XT_F2_RENORM 3  // phases between two re-normalisations of the merged weight (1: every step)
#endif
#define XT_F2_WAVES 4   // waves per block (independent of each other; they only share the table copy)

static inline bool xt_use_fast2(int S, int NS, int F, bool preds) { return S == 2 && NS == 1 && F >= 4 && F <= 7 && !preds; }

template <int F>
struct XtF2Geom {
    static constexpr int E = 1 << F;           // sequences per track
    static constexpr int NG = E / 2;           // groups (= lanes) per track
    static constexpr int TPW = 64 / NG;        // tracks per wave
    static constexpr int EW = E * TPW;         // sequences per wave (always 128)
};

// GF(2)-linear storage swizzle s = B.w of the wave-level sequence index w = ts * 2^F + idx (7 bits), found by
// tools/swizzle_search.py: for every phase and both group members, each 32-lane group of a wave hits 32
// distinct 8-byte bank pairs (ds_read_b64/b32 banking) and each 16-lane group 16 distinct 8-byte units mod 16
// (ds_write_b64 banking) - zero LDS bank conflicts in the step loop.  Column j = image of bit j of w.
template <int F>
XT_HD int xt_f2_swz(int w)
{
    const int c4[7] = {0x5, 0x12, 0x14, 0x2c, 0x4f, 0x41, 0x15};   // F = 4 and 5
    const int c6[7] = {0x15, 0x36, 0x67, 0x1c, 0x2, 0x1a, 0x16};   // F = 6
    const int c7[7] = {0x65, 0x72, 0x6, 0x69, 0x1c, 0x4, 0x5f};    // F = 7
    int s = 0;
    for (int j = 0; j < 7; ++j) {
        const int col = F == 6 ? c6[j] : (F == 7 ? c7[j] : c4[j]);
        s ^= ((w >> j) & 1) ? col : 0;
    }
    return s;
}

// exp(x), x <= 0, for the two register-resident 2-state kernels (this file and xt_reg2.h): x = (1024 e + j) ln2/1024 + r, |r| <= ln2/2048,
// exp(x) = 2^e * TB[j] * P3(r) with a 1024-entry table TB[j] = 2^(j/1024) that every workgroup builds in its LDS from the blob's 64-entry
// table (xt_f2_build_exp_table).  These kernels are bound by fp64 VALU issue (VALU busy 1.00): the 8 KiB of LDS buy two FMAs per exponential,
// four per step - a timing experiment with a shortened polynomial measured 3.00 -> 2.86 ms on the headline workload (round 4); bank
// conflicts of the lookups do not matter there (a conflict-free 32-entry table changed nothing, xt_math.h).  P3 = 1 + r (1 + r (c2 + r / 6)),
// c2 = 1/2 + a^2 / 24 (the r^4 term folded in, a = ln2 / 2048): |rel err| < 1.5e-16 in exact arithmetic.
// n = 1024 x / ln2 exceeds 32 bits at the clamp (XT_TCLAMP): the exponent e = n >> 10 is cut out of the 64-bit pattern of the magic-number
// sum (one v_alignbit), j is the low 10 bits.  Two independent evaluations are interleaved instruction by instruction (a wave alone can only
// issue a DEPENDENT fp64 FMA every few issue slots).  Returns p (without the table factor), the table index j and the exponent e.
#define XT_F2_XCLAMP XT_TCLAMP
#define XT_F2_EXP_ENTRIES 1024
#define XT_F2_EXP_BYTES (XT_F2_EXP_ENTRIES * 8)
#define XT_F2_EXP_SCALE 1477.3197218702985       // 1024 / ln2
#define XT_F2_EXP_HI (-6.769015308236703e-04)    // -ln2 / 1024, high part (1/16 of xt_exp_tab's: exact) ...
#define XT_F2_EXP_LO (-1.2691901263564344e-11)   // ... and the rest
#define XT_F2_EXP_C2 0.5000000047728719          // 1/2 + a^2 / 24, a = ln2 / 2048
XT_HD void xt_f2_exp_bits(double t, int& j, int& e)
{
    union {
        double d;
        unsigned long long u;
    } v;
    v.d = t;  // 1.5 * 2^52 + n: the mantissa field holds 2^51 + n in two's complement
    j = (int)(v.u & (XT_F2_EXP_ENTRIES - 1));
    e = (int)(unsigned int)(v.u >> 10);  // bits 41 .. 10: n >> 10 for |n| < 2^41 (the 2^51 and the exponent field lie above)
}
XT_HD void xt_exp_tab_x2(double x0, double x1, double& p0, double& p1, int& j0, int& j1, int& e0, int& e1)
{
    x0 = x0 > XT_F2_XCLAMP ? x0 : XT_F2_XCLAMP;
    x1 = x1 > XT_F2_XCLAMP ? x1 : XT_F2_XCLAMP;
    const double t0 = xt_fma(x0, XT_F2_EXP_SCALE, XT_MAGIC);  // integer part in the low bits of the mantissa (xt_math.h)
    const double t1 = xt_fma(x1, XT_F2_EXP_SCALE, XT_MAGIC);
    const double k0 = t0 - XT_MAGIC;
    const double k1 = t1 - XT_MAGIC;
    double r0 = xt_fma(k0, XT_F2_EXP_HI, x0);
    double r1 = xt_fma(k1, XT_F2_EXP_HI, x1);
    r0 = xt_fma(k0, XT_F2_EXP_LO, r0);
    r1 = xt_fma(k1, XT_F2_EXP_LO, r1);
    double q0 = 1.66666666666666666667e-01, q1 = 1.66666666666666666667e-01;
#define XT_H2(C)              \
    q0 = xt_fma(q0, r0, (C)); \
    q1 = xt_fma(q1, r1, (C));
    XT_H2(XT_F2_EXP_C2)
    XT_H2(1.0)
    XT_H2(1.0)
#undef XT_H2
    p0 = q0;
    p1 = q1;
    xt_f2_exp_bits(t0, j0, e0);
    xt_f2_exp_bits(t1, j1, e1);
}
// TB[i] = 2^(i / 1024) = T64[i >> 4] * exp((i & 15) ln2 / 1024) at byte offset `off` of the workgroup's LDS; T64: the blob's table (already
// staged).  The second factor: Taylor to s^7 (s <= 0.0102: truncation far below 1e-18).  Call between two workgroup barriers.
template <class Ctx>
XT_HD void xt_f2_build_exp_table(Ctx& cx, char* lds, int off, const double* T64)
{
    for (int i = cx.tid(); i < XT_F2_EXP_ENTRIES; i += cx.nthreads()) {
        const double sv = (double)(i & 15) * (XT_LN2 / 1024.0);
        double q = 1.0 / 5040.0;
        q = xt_fma(q, sv, 1.0 / 720.0);
        q = xt_fma(q, sv, 1.0 / 120.0);
        q = xt_fma(q, sv, 1.0 / 24.0);
        q = xt_fma(q, sv, 1.0 / 6.0);
        q = xt_fma(q, sv, 0.5);
        q = xt_fma(q, sv, 1.0);
        q = xt_fma(q, sv, 1.0);
        *(double*)(lds + off + i * 8) = T64[i >> 4] * q;
    }
}

template <int F, int D, int K>
struct XtF2State {
    // absolute LDS byte addresses (of the zm element) of this lane's two sequences per phase, packed a0 | a1 << 16
    int s01[F];
};

// Fixed LDS map of the fast path (bytes).  One array per field over all waves of the block, so that a field is reached
// from the zm address by a compile-time offset - the exponents ze too: they sit in 8-byte slots (upper half unused), which
// saves the address arithmetic a packed int array would need in every step.
#define XT_F2_EXPB_OFF 1024                                    /* the 1024-entry exp table of xt_exp_tab_x2, built per workgroup */
#define XT_F2_TAB_BYTES (1024 + XT_F2_EXP_BYTES)               /* model blob (tables 288 B + the blob's 64-entry exp table 512 B, NaN flags) + that table */
#define XT_F2_T64_OFF ((XT_BLOB_HDR + XT_NTAB * 4) * 8)   /* the blob's T64 table (xt_tables.h) */
#define XT_F2_NAN_OFF 832                                      /* int[XT_F2_WAVES][8]: track has a NaN position / sigma */
#define XT_F2_ARR (XT_F2_WAVES * 128 * 8)                      /* bytes of one double field for all waves */
#define XT_F2_ZM0 XT_F2_TAB_BYTES
XT_HD int xt_f2_ze0(int D, int K) { return XT_F2_ZM0 + (1 + D + K) * XT_F2_ARR; }
XT_HD int xt_f2_pos0(int D, int K) { return xt_f2_ze0(D, K) + XT_F2_ARR; }
XT_HD int xt_f2_acc0(int D, int K, int KS, int tpw) { return xt_f2_pos0(D, K) + XT_F2_WAVES * tpw * XT_F2_CHUNK * (D + KS) * 8; }
XT_HD int xt_f2_block_bytes(int D, int K, int KS, int tpw) { return xt_f2_acc0(D, K, KS, tpw) + XT_F2_WAVES * 8 * 3 * 8; }

// Keeps the compiler from hoisting the (loop-invariant) unpacked addresses of all F phases out of the
// step loop, which costs ~50 VGPRs and halves the occupancy.
XT_HD int xt_opaque(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// LDS access by byte offset.  On the device the offset IS the LDS address: the kernels' only LDS object is the dynamic array (xt_host.h), which
// starts at address 0 (checked once per workgroup by xt_f2_check_lds_base), and the address is formed from the integer - written as
// `lds + byte_off` the compiler keeps an add of the array's (zero) base in front of every data-dependent access: three 32-bit VALU
// instructions per step of the headline kernel.
template <class T>
XT_HD T& xt_at(char* lds, int byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lds;
    return *(__attribute__((address_space(3))) T*)(unsigned int)byte_off;
#else
    return *(T*)(lds + byte_off);
#endif
}
XT_HD void xt_f2_check_lds_base(char* lds)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) char*)lds != 0u) __builtin_trap();
#else
    (void)lds;
#endif
}

// One recursion step at compile-time phase H.  c: position, l2: localisation variance(s), TT: transition row in use.
// ZF ("zero free"): every sequence of the window has a positive weight (well-scaled model, all slots populated): the zero-weight special
// case (compare + three selects per step) is compiled out.  LAZY: the merged weight is re-normalised in every XT_F2_RENORM-th phase only
// (well-scaled models, see xt_build_blob); otherwise in every step.
template <int F, int D, int K, int H, bool ZF = false, bool LAZY = false>
XT_HD void xt_f2_step(char* lds, const XtF2State<F, D, K>& st, const double* c, const double* l2, const double* TT, const double* TD2)
{
    constexpr int ZEO = (1 + D + K) * XT_F2_ARR;  // ze address = a + ZEO
    const int pk = xt_opaque(st.s01[H]);
    const int a0 = pk & 0xffff, a1 = (int)((unsigned)pk >> 16);
    const double z0 = xt_at<double>(lds, a0), z1 = xt_at<double>(lds, a1);
    const int e0 = xt_at<int>(lds, a0 + ZEO), e1 = xt_at<int>(lds, a1 + ZEO);
    const int emax = e0 > e1 ? e0 : e1;
    const double w0 = xt_ldexp(z0, e0 - emax), w1 = xt_ldexp(z1, e1 - emax);
    const double W = w0 + w1;
    double M[D], U[K];
    for (int d = 0; d < D; ++d)
        M[d] = xt_fma(w1, xt_at<double>(lds, a1 + (1 + d) * XT_F2_ARR), w0 * xt_at<double>(lds, a0 + (1 + d) * XT_F2_ARR));
    for (int k = 0; k < K; ++k)
        U[k] = xt_fma(w1, xt_at<double>(lds, a1 + (1 + D + k) * XT_F2_ARR), w0 * xt_at<double>(lds, a0 + (1 + D + k) * XT_F2_ARR));
    const bool live = ZF ? true : W > 0.0;
    const double Ws = ZF ? W : (live ? W : 1.0);
    // LAZY: the merged weight is re-normalised (mantissa in [0.5, 1), exponent into ze) in every XT_F2_RENORM-th phase only; every
    // expression of the step is homogeneous in W, and for a well-scaled model the drift between two normalisations stays far inside
    // the fp64 range (bounds in xt_build_blob).
    constexpr bool RN = !LAZY || (H % XT_F2_RENORM) == 0;
    const double Wm = RN ? xt_frexp_mant(W) : W;  // 0 when W == 0
    const int We = live ? (RN ? emax + xt_frexp_exp(W) : emax) : XT_EMIN;

    // Dq[k] = W * den_q[k] = W*(l2 + d2_q) + U
    double Dq[2][K];
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < K; ++k) Dq[q][k] = xt_fma(Ws, l2[k] + TD2[q], U[k]);
    // one reciprocal for 1/W and all 1/Dq
    double rD[2][K], rW;
    if (K == 1) {
        const double d01 = Dq[0][0] * Dq[1][0];
        const double R = xt_rcp(Ws * d01);
        const double RW = R * Ws;
        rW = R * d01;
        rD[0][0] = RW * Dq[1][0];
        rD[1][0] = RW * Dq[0][0];
    } else {
        // prefix/suffix products over the 2K+1 factors {Ws, Dq[0][*], Dq[1][*]}
        double f[2 * K + 1], pre[2 * K + 2], suf[2 * K + 2];
        f[0] = Ws;
        for (int q = 0; q < 2; ++q)
            for (int k = 0; k < K; ++k) f[1 + q * K + k] = Dq[q][k];
        pre[0] = 1.0;
        for (int i = 0; i < 2 * K + 1; ++i) pre[i + 1] = pre[i] * f[i];
        suf[2 * K + 1] = 1.0;
        for (int i = 2 * K; i >= 0; --i) suf[i] = suf[i + 1] * f[i];
        const double R = xt_rcp(pre[2 * K + 1]);
        rW = R * suf[1];
        for (int q = 0; q < 2; ++q)
            for (int k = 0; k < K; ++k) rD[q][k] = R * pre[1 + q * K + k] * suf[2 + q * K + k];
    }
    double dmW[D], dsqW = 0.0;
    for (int d = 0; d < D; ++d) {
        dmW[d] = xt_fma(c[d], Ws, -M[d]);
        if (K == 1) dsqW = xt_fma(dmW[d], dmW[d], dsqW);
    }
    double x[2], gf[2], tt[2][K];
    const double A = -0.5 * rW * dsqW;
    for (int q = 0; q < 2; ++q) {
        if (K == 1) {
            x[q] = A * rD[q][0];
            tt[q][0] = xt_fma(Ws, TD2[q], U[0]) * rD[q][0];
            gf[q] = xt_pow_half<D>(Ws * rD[q][0]);
        } else {
            double xx = 0.0, gg = 1.0;
            for (int d = 0; d < D; ++d) {
                xx = xt_fma(dmW[d] * dmW[d], rD[q][d], xx);
                tt[q][d] = xt_fma(Ws, TD2[q], U[d]) * rD[q][d];
                gg *= Ws * rD[q][d];
            }
            x[q] = xx * (-0.5 * rW);
            gf[q] = sqrt(gg);
        }
    }
    double p[2];
    int j[2], n[2];
    xt_exp_tab_x2(x[0], x[1], p[0], p[1], j[0], j[1], n[0], n[1]);
    for (int q = 0; q < 2; ++q) {
        const int aq = q ? a1 : a0;
        int en = We + n[q];
        const double tj = xt_at<double>(lds, XT_F2_EXPB_OFF + j[q] * 8);
        double zn = (Wm * TT[q]) * (gf[q] * tj) * p[q];
        if (!LAZY) {
            // guarded steps (models outside the well-scaled bounds, e.g. a transition probability of 1e-200): the stored mantissa is
            // normalised too, so that the next step's W^3-sized products cannot leave the fp64 range
            en += xt_frexp_exp(zn);
            zn = xt_frexp_mant(zn);
        }
        xt_at<double>(lds, aq) = zn;
        xt_at<int>(lds, aq + ZEO) = en > XT_EMIN ? en : XT_EMIN;
        for (int d = 0; d < D; ++d) xt_at<double>(lds, aq + (1 + d) * XT_F2_ARR) = xt_fma(dmW[d], tt[q][K == 1 ? 0 : d], M[d]) * rW;
        for (int k = 0; k < K; ++k) xt_at<double>(lds, aq + (1 + D + k) * XT_F2_ARR) = l2[k] * tt[q][k];
    }
}

template <int F, int D, int K, class Ctx>
XT_HD void xt_ll_s2_body(const XtKernelArgs& a, Ctx& cx)
{
    int lb, nb;
    const XtBucketDesc b = xt_bind_bucket(a, cx.block(), cx.nblocks(), lb, nb);
    typedef XtF2Geom<F> Gm;
    constexpr int E = Gm::E, NG = Gm::NG, TPW = Gm::TPW;
    const int lane = cx.lane();
    const int wib = cx.wave_in_block();
    const int nwb = cx.waves_per_block();
    const int L = b.L;
    const int KS = a.locerr_mode ? a.KS : 0;
    double* smem = cx.smem();
    char* lds = (char*)smem;

    // block-shared model tables + the 2^(j/64) table of the exponential
    const int ntab = xt_tab_doubles(2, 2);
    for (int i = cx.tid(); i < ntab; i += cx.nthreads()) smem[i] = xt_blob_ptr(a)[i];
    cx.sync();
    xt_f2_check_lds_base(lds);
    xt_f2_build_exp_table(cx, lds, XT_F2_EXPB_OFF, (const double*)(lds + XT_F2_T64_OFF));
    cx.sync();
    const double* hdr = smem;
    const double* TAB = smem + XT_BLOB_HDR;

    const int ts = lane / NG;        // track slot inside the wave
    const int g = lane - ts * NG;    // group of that track
    const int prev = g >> (F - 2 >= 0 ? F - 2 : 0);  // top digit of g (F >= 2)
    double T0[2], T1[2], TD2[2];
    const int tlast = L - 1;
    const int stay_from = a.min_len > 2 ? a.min_len : 2;
    const int vfin = (b.isBL ? 2 : 0) + (tlast >= stay_from ? 1 : 0);
    for (int q = 0; q < 2; ++q) {
        T0[q] = TAB[(0 * 2 + prev) * 2 + q];
        T1[q] = TAB[(1 * 2 + prev) * 2 + q];
        TD2[q] = TAB[(4 * 2 + prev) * 2 + q];
    }
    double l2g[K];
    for (int k = 0; k < K; ++k) l2g[k] = hdr[k];
    const bool well_scaled = a.well_scaled != 0;  // decided per launch on the host (xt_model_well_scaled)

    // LDS map (bytes): [tables 1 KiB][zm][m x D][u x K] (each XT_F2_WAVES x 128 doubles) [ze: XT_F2_WAVES x 128 ints in 8-byte slots][pos][sig]
    constexpr int ZEO = (1 + D + K) * XT_F2_ARR;
    const int wave0 = XT_F2_ZM0 + wib * 128 * 8;  // byte address of this wave's zm[0]
    double* pos = (double*)(lds + xt_f2_pos0(D, K)) + wib * TPW * XT_F2_CHUNK * (D + KS);  // [TPW][CHUNK][D]
    double* sig = pos + TPW * XT_F2_CHUNK * D;                                             // [TPW][CHUNK][KS]

    XtF2State<F, D, K> st;
    for (int h = 0; h < F; ++h) {
        const int base = ((g << (h + 1)) | (g >> (F - 1 - h))) & (E - 1);  // g's digit i -> slot (h+1+i) mod F
        const int b0 = wave0 + xt_f2_swz<F>(ts * E + base) * 8, b1 = wave0 + xt_f2_swz<F>(ts * E + (base | (1 << h))) * 8;
        st.s01[h] = b0 | (b1 << 16);
    }

    // running product of the tracks' likelihoods per track slot {mantissa, exponent, count}: kept in LDS (owned by the
    // slot's lane g == 0) rather than in six VGPRs that would be live across the whole kernel
    double* accp = (double*)(lds + xt_f2_acc0(D, K, KS, TPW)) + (wib * 8 + ts) * 3;
    if (g == 0) {
        accp[0] = 1.0;
        accp[1] = 0.0;
        accp[2] = 0.0;
    }

    const int64_t nbatch = (b.N + TPW - 1) / TPW;
    const int64_t W0 = (int64_t)lb * nwb + wib, NW = (int64_t)nb * nwb;
    for (int64_t batch = W0; batch < nbatch; batch += NW) {
        const int64_t trk = batch * TPW + ts;
        const bool act = trk < b.N;

        auto stage = [&](int p0) {  // positions [p0, p0 + CHUNK) of the wave's TPW tracks -> LDS (coalesced along the track)
            for (int i = lane; i < TPW * XT_F2_CHUNK * D; i += 64) {
                const int t_ = i / (XT_F2_CHUNK * D), r = i - t_ * (XT_F2_CHUNK * D);
                const int64_t tk = batch * TPW + t_;
                const int64_t tkc = tk < b.N ? tk : b.N - 1;
                const int pp = p0 + r / D;
                if (pp < L) {
                    const double v = b.tracks[(tkc * L + p0) * D + r];
                    pos[i] = v;
                    if (v != v) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + t_) * 4) = 1;
                }
            }
            if (KS)
                for (int i = lane; i < TPW * XT_F2_CHUNK * KS; i += 64) {
                    const int t_ = i / (XT_F2_CHUNK * KS), r = i - t_ * (XT_F2_CHUNK * KS);
                    const int64_t tk = batch * TPW + t_;
                    const int64_t tkc = tk < b.N ? tk : b.N - 1;
                    const int pp = p0 + r / KS;
                    if (pp < L) {
                        const double v = b.sigma[(tkc * L + p0) * KS + r];
                        sig[i] = v;
                        if (v != v) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + t_) * 4) = 1;
                    }
                }
            cx.wave_sync();
        };
        auto getpos = [&](int t, double* c, double* l2) {
            const int r = t & (XT_F2_CHUNK - 1);
            for (int d = 0; d < D; ++d) c[d] = pos[(ts * XT_F2_CHUNK + r) * D + d];
            if (KS == 0) {
                for (int k = 0; k < K; ++k) l2[k] = l2g[k];
            } else {
                for (int k = 0; k < K; ++k) {
                    double s = sig[(ts * XT_F2_CHUNK + r) * KS + (KS == 1 ? 0 : k)];
                    if (a.locerr_mode == 2) {
                        s = xt_fma(s, hdr[3], hdr[4]);
                        s = s < 1e-6 ? 1e-6 : s;
                    }
                    l2[k] = s * s;
                }
            }
        };
        // steps t .. tend with transition row TT (the F phases of the circular digit buffer unrolled: every LDS
        // address is a register, phase h = t mod F)
#define XT_F2_PHASE(H, ZF_, LAZY_)                                                     \
    if (F > (H) && t <= tend2 && ph == (H)) {                                          \
        double c[D], l2[K];                                                            \
        getpos(t, c, l2);                                                              \
        xt_f2_step<F, D, K, ((H) < F ? (H) : 0), ZF_, LAZY_>(lds, st, c, l2, TT, TD2); \
        cx.wave_sync();                                                                \
        ++t;                                                                           \
        ph = (H) + 1 == F ? 0 : (H) + 1;                                               \
    }
#define XT_F2_PHASES(ZF_, LAZY_)  \
    XT_F2_PHASE(1, ZF_, LAZY_)    \
    XT_F2_PHASE(2, ZF_, LAZY_)    \
    XT_F2_PHASE(3, ZF_, LAZY_)    \
    XT_F2_PHASE(4, ZF_, LAZY_)    \
    XT_F2_PHASE(5, ZF_, LAZY_)    \
    XT_F2_PHASE(6, ZF_, LAZY_)    \
    XT_F2_PHASE(0, ZF_, LAZY_)
        auto run_steps = [&](int& t, int tend, const double* TT) {
            int ph = t % F;  // phase of the circular digit buffer, advanced with t (no modulo in the step loop)
            if (!well_scaled) {  // fully guarded steps: zero weights handled, re-normalisation in every step
                const int tend2 = tend;
                while (t <= tend2) { XT_F2_PHASES(false, false) }
                return;
            }
            {   // steps t < F may meet slots that are not populated yet (zero weight)
                const int tend2 = tend < F - 1 ? tend : F - 1;
                while (t <= tend2) { XT_F2_PHASES(false, true) }
            }
            const int tend2 = tend;
            while (t <= tend2) { XT_F2_PHASES(true, true) }
        };
#undef XT_F2_PHASES
#undef XT_F2_PHASE

        XtAcc tot;
        tot.clear();
        int t = 1;
        if (lane < 8) xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + lane) * 4) = 0;
        cx.wave_sync();
        for (int p0 = 0; p0 < L; p0 += XT_F2_CHUNK) {  // one staged chunk of positions at a time
            stage(p0);
            if (p0 == 0) {
                // ---- position 0: the initial-state digit in slot 0, every other sequence has zero weight
                double c0[D], l20[K];
                getpos(0, c0, l20);
                for (int q = 0; q < 2; ++q) {
                    const int w = ts * E + g * 2 + q;  // any bijection lanes x {0,1} -> the track's E sequences
                    const int idx = w - ts * E;
                    const int aq = wave0 + xt_f2_swz<F>(w) * 8;
                    xt_at<double>(lds, aq) = idx < 2 ? hdr[8 + idx] : 0.0;  // initial fractions F0, F1
                    xt_at<int>(lds, aq + ZEO) = idx < 2 ? 0 : XT_EMIN;
                    for (int d = 0; d < D; ++d) xt_at<double>(lds, aq + (1 + d) * XT_F2_ARR) = c0[d];
                    for (int k = 0; k < K; ++k) xt_at<double>(lds, aq + (1 + D + k) * XT_F2_ARR) = l20[k];
                }
                cx.wave_sync();
            }
            // ---- positions 1 .. L-2 that lie in this chunk: without the stay-in-FOV factor before step stay_from, with it after
            const int tend = (L - 2 < p0 + XT_F2_CHUNK - 1) ? L - 2 : p0 + XT_F2_CHUNK - 1;
            run_steps(t, tend < stay_from - 1 ? tend : stay_from - 1, T0);
            run_steps(t, tend, T1);
            if (tlast < p0 || tlast >= p0 + XT_F2_CHUNK) continue;
            // ---- last position (+ leaving/bleaching factor folded into TFIN): reduction over (Q, q)
            double cl[D], l2l[K];
            getpos(tlast, cl, l2l);
            const int h = tlast % F;
            // phase index is runtime here: pick the packed addresses with a small select chain (once per track)
            int pk = st.s01[0];
            for (int hh = 1; hh < F; ++hh) pk = h == hh ? st.s01[hh] : pk;
            const int a0 = pk & 0xffff, a1 = (int)((unsigned)pk >> 16);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1  // cold code: keep its register footprint below the step loop's so that it does not set the kernel's VGPR count
#endif
            for (int Q = 0; Q < 2; ++Q) {
                const int aq = Q ? a1 : a0;
                const double zq = xt_at<double>(lds, aq);
                const int eq = xt_at<int>(lds, aq + ZEO);
                double dq[D], dsq = 0.0;
                for (int d = 0; d < D; ++d) {
                    dq[d] = cl[d] - xt_at<double>(lds, aq + (1 + d) * XT_F2_ARR);
                    dsq = xt_fma(dq[d], dq[d], dsq);
                }
                double x[2], gf[2];
                for (int q = 0; q < 2; ++q) {
                    if (K == 1) {
                        const double r = xt_rcp(TD2[q] + xt_at<double>(lds, aq + (1 + D) * XT_F2_ARR) + l2l[0]);
                        x[q] = -0.5 * dsq * r;
                        gf[q] = xt_pow_half<D>(r);
                    } else {
                        double xx = 0.0, gg = 1.0;
                        for (int d = 0; d < D; ++d) {
                            const double r = xt_rcp(TD2[q] + xt_at<double>(lds, aq + (1 + D + d) * XT_F2_ARR) + l2l[d]);
                            xx = xt_fma(-0.5 * dq[d] * dq[d], r, xx);
                            gg *= r;
                        }
                        x[q] = xx;
                        gf[q] = sqrt(gg);
                    }
                }
                double p[2];
                int j[2], n[2];
                xt_exp_tab_x2(x[0], x[1], p[0], p[1], j[0], j[1], n[0], n[1]);
                for (int q = 0; q < 2; ++q)
                    tot.add(zq * TAB[(vfin * 2 + prev) * 2 + q] * gf[q] * xt_at<double>(lds, XT_F2_EXPB_OFF + j[q] * 8) * p[q], eq + n[q]);
            }
        }
        // reduce over the track's NG lanes (all lanes end with the same values)
        const int fe = cx.template group_max_i32<NG>(tot.m != 0.0 ? tot.e : XT_EMIN);
        double sum = cx.template group_sum_f64<NG>(tot.m != 0.0 ? xt_ldexp(tot.m, tot.e - fe) : 0.0);
        if (xt_at<int>(lds, XT_F2_NAN_OFF + (wib * 8 + ts) * 4)) sum = NAN;  // NaN input -> NaN likelihood, as in the reference
        if (act && g == 0) {
            if (b.ll_out) b.ll_out[trk] = log(sum) + (double)fe * XT_LN2 + b.ll_const;
            const double pm = accp[0] * xt_frexp_mant(sum);
            double acce = accp[1] + (double)(fe + xt_frexp_exp(sum) + xt_frexp_exp(pm));
            if (sum == 0.0) acce = -INFINITY;  // zero likelihood: log = -inf, as the reference would produce
            accp[0] = xt_frexp_mant(pm);
            accp[1] = acce;
            accp[2] += 1.0;
        }
        cx.wave_sync();
    }

    // ---- per-slot log-likelihood sums -> block partial (fixed order)
    cx.sync();
    if (g == 0) smem[wib * TPW + ts] = accp[2] > 0.0 ? log(accp[0]) + accp[1] * XT_LN2 + accp[2] * b.ll_const : 0.0;
    cx.sync();
    if (cx.tid() == 0) {
        double s = 0.0;
        for (int i = 0; i < nwb * TPW; ++i) s += smem[i];
        a.partials[cx.block()] = s;
    }
}
