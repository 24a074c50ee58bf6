// Compile-time instantiation table shared by the HIP launcher and the CPU-thread emulator.
// G_ = S^ns members per group (0 = runtime-G streaming path), D = spatial dims, K = 1 (one
// localisation-error variance for all dims) or D (one per dim), PREDS = state posteriors.
#pragma once

template <int GG, int DD, bool PREDS, class L>
static inline bool xt_dispatch_k(int K, L& l)
{
    if (K == 1) return l.template run<GG, DD, 1, PREDS>();
    if (K == DD && DD > 1) return l.template run<GG, DD, DD, PREDS>();
    return false;
}

template <int GG, bool PREDS, class L>
static inline bool xt_dispatch_d(int D, int K, L& l)
{
    if (D == 1) return xt_dispatch_k<GG, 1, PREDS>(K, l);
    if (D == 2) return xt_dispatch_k<GG, 2, PREDS>(K, l);
    if (D == 3) return xt_dispatch_k<GG, 3, PREDS>(K, l);
    return false;
}

template <class L>
static inline bool xt_dispatch(int G, int D, int K, bool preds, L& l)
{
    if (preds) {  // posteriors need compile-time G (nb_substeps == 1, so G == n_states)
        if (G == 2) return xt_dispatch_d<2, true>(D, K, l);
        if (G == 3) return xt_dispatch_d<3, true>(D, K, l);
        if (G == 4) return xt_dispatch_d<4, true>(D, K, l);
        if (G == 5) return xt_dispatch_d<5, true>(D, K, l);
        if (G == 6) return xt_dispatch_d<6, true>(D, K, l);
        return false;
    }
    if (G == 2) return xt_dispatch_d<2, false>(D, K, l);
    if (G == 3) return xt_dispatch_d<3, false>(D, K, l);
    if (G == 4) return xt_dispatch_d<4, false>(D, K, l);
    return xt_dispatch_d<0, false>(D, K, l);
}

// ---- two-state wave-synchronous fast path (xt_fast2.h): run<F, D, K>()
template <int FF, class L>
static inline bool xt_dispatch_f2_dk(int D, int K, L& l)
{
    if (D == 1 && K == 1) return l.template run_f2<FF, 1, 1>();
    if (D == 2 && K == 1) return l.template run_f2<FF, 2, 1>();
    if (D == 2 && K == 2) return l.template run_f2<FF, 2, 2>();
    if (D == 3 && K == 1) return l.template run_f2<FF, 3, 1>();
    if (D == 3 && K == 3) return l.template run_f2<FF, 3, 3>();
    return false;
}

template <class L>
static inline bool xt_dispatch_f2(int F, int D, int K, L& l)
{
    if (F == 4) return xt_dispatch_f2_dk<4>(D, K, l);
    if (F == 5) return xt_dispatch_f2_dk<5>(D, K, l);
    if (F == 6) return xt_dispatch_f2_dk<6>(D, K, l);
    if (F == 7) return xt_dispatch_f2_dk<7>(D, K, l);
    return false;
}

// ---- entry-parallel path for ns >= 2 (xt_entry.h): run_entry<GP, D, K>()
template <int GP, class L>
static inline bool xt_dispatch_entry_dk(int D, int K, L& l)
{
    if (D == 1 && K == 1) return l.template run_entry<GP, 1, 1>();
    if (D == 2 && K == 1) return l.template run_entry<GP, 2, 1>();
    if (D == 2 && K == 2) return l.template run_entry<GP, 2, 2>();
    if (D == 3 && K == 1) return l.template run_entry<GP, 3, 1>();
    if (D == 3 && K == 3) return l.template run_entry<GP, 3, 3>();
    return false;
}

template <class L>
static inline bool xt_dispatch_entry(int GP, int D, int K, L& l)
{
    if (GP == 4) return xt_dispatch_entry_dk<4>(D, K, l);
    if (GP == 8) return xt_dispatch_entry_dk<8>(D, K, l);
    if (GP == 16) return xt_dispatch_entry_dk<16>(D, K, l);
    if (GP == 32) return xt_dispatch_entry_dk<32>(D, K, l);
    if (GP == 64) return xt_dispatch_entry_dk<64>(D, K, l);
    return false;
}
