// exact_generic.hpp -- exact_site_wave: the reference's fp64 pipeline for ONE site of a sparse-label document, run by
// a whole wavefront in the dense group layout with the layout known only at run time.
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// The sparse kernels (one lane per ALLOWED topic) decide the draw from unnormalised fp64 prefix sums with a 2^-40
// margin.  For the ~1e-11 of the sites where that margin is not met the topic must come from the reference's own
// pipeline -- prob = lab*a*(num/den); prob /= np.sum(prob) in numpy's pairwise order (LabeledLDA.py:113-118), then
// the keyed draw of oracle/llda_oracle.py:draw_keyed in device-position order -- which is defined on the DENSE group
// layout of the problem's K topics (G = 8..64 lanes x T slots).  Here the 64 lanes of the wave
// play the G lanes of that layout for one document: lane g holds the T scores of its slots (exact zeros except at
// the document's allowed positions -- adding an exact zero never changes a partial sum, so the association order of
// the survivors is the reference's), and the sum / normalisation / prefix / Hillis-Steele scan / selection below are
// the statements of group_sum<>, cold_tiers<> (exact tier) and draw_position<> with G and T as run-time values.
// Rare path: rolled loops, selects instead of dynamic register indexing, no LDS.
struct ExactLayout {
    int G, T;                 // lanes of the dense group, slots per lane
    int last_leaf, tail, tail_row, n_rounds, xor_tree;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];
};

__device__ __forceinline__ double readlane_var_f64(double x, int l)       // l wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}

// Wave-collective: call with all 64 lanes active and wave-uniform (base, A, u).  Sparse lane base + i (i < A) holds
// the exact score `w_mine` and the device position `pos_mine` of the document's i-th allowed topic (in draw order:
// ascending (lane, slot) of the dense layout).
// Returns the chosen device position (the same value in every lane) or -1 when no topic has a positive probability.
__device__ __noinline__ int exact_site_wave(double w_mine, int pos_mine, int base, int A, double u, int G, int T,
                                            int last_leaf, int tail, int tail_row, int n_rounds, int xor_tree,
                                            const uint32_t *rounds_pk, int lane)
{
    ExactLayout L;
    L.G = G; L.T = T; L.last_leaf = last_leaf; L.tail = tail; L.tail_row = tail_row; L.n_rounds = n_rounds; L.xor_tree = xor_tree;
#pragma unroll
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) L.rounds_pk[r] = (rounds_pk && r < n_rounds) ? rounds_pk[r] : 0u;
    // (rolled loops over a per-lane array the compiler keeps in scratch: this path runs ~1e-11 per site and must not
    // cost the sparse kernels, which call it, their registers -- they run 8 waves per SIMD without it)
    base = __builtin_amdgcn_readfirstlane(base);
    A = __builtin_amdgcn_readfirstlane(A);
    double w[16];
#pragma unroll 1
    for (int s = 0; s < 16; ++s) w[s] = 0.0;
#pragma unroll 1
    for (int i = 0; i < A; ++i) {                               // scatter the allowed topics into the dense layout
        const double wi = readlane_var_f64(w_mine, base + i);
        const int pi = __builtin_amdgcn_readlane(pos_mine, base + i);
        int g, s;
        lane_slot_of_rt(G, T, pi, g, s);
        if (lane == g) w[s] = wi;
    }
    // np.sum(prob): group_sum<> / group_sum_tail<> with run-time G, T
    const int leaf = lane >> 3;
    double acc = 0.0, tv = 0.0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        const double v = w[s];
        if (L.tail != 0 && s == L.tail_row && leaf == L.last_leaf) tv = v;
        else acc = acc + v;
    }
    acc = acc + dpp_f64<DPP_XOR1>(acc);
    acc = acc + dpp_f64<DPP_XOR2>(acc);
    acc = acc + dpp_f64<DPP_HALF_MIRROR>(acc);
#pragma unroll 1
    for (int t = 0; t < L.tail; ++t) {
        const double o = __shfl(tv, L.last_leaf * 8 + t, 64);
        if (leaf == L.last_leaf) acc = acc + o;
    }
    if (G > 8) {
        if (L.xor_tree) {
            acc = acc + dpp_f64<DPP_ROW_ROR + 8>(acc);
            if (G > 16) acc = acc + xor16_f64(acc, lane);
            if (G > 32) acc = acc + xor32_f64(acc, lane);
        } else {
#pragma unroll 1
            for (int r = 0; r < L.n_rounds; ++r) {
                const int partner = (L.rounds_pk[r] >> (4 * (leaf & 7))) & 15;
                const double o = __shfl(acc, partner * 8 + (lane & 7), 64);
                if (partner != leaf) acc = acc + o;
            }
            acc = __shfl(acc, 0, 64);
        }
    }
    const double S = readlane_var_f64(acc, 0);                  // every lane of the group holds the same sum
    const double y = 1.0 / S;
    // prob /= S; per-lane prefix (kept in w); Hillis-Steele scan of the lane totals over the G lanes
    double run = 0.0;
    uint32_t pm = 0;
#pragma unroll 1
    for (int k = 0; k < T; ++k) {
        const double wk = w[k];
        const double p = div_by(wk, S, y);
        run = (k == 0) ? p : run + p;
        pm |= (wk > 0.0 ? 1u : 0u) << k;
        w[k] = run;                                             // q[k]
    }
    double X = run;
#pragma unroll 1
    for (int d = 1; d < G; d *= 2) {
        const double up = __shfl_up(X, d, 64);
        X = (lane >= d) ? up + X : X;
    }
    const double tot = readlane_var_f64(X, G - 1);
    const double prev = __shfl_up(X, 1, 64);
    const double tg = u * tot - (lane ? prev : 0.0);
    uint32_t fm = 0;
#pragma unroll 1
    for (int k = 0; k < T; ++k) fm |= (((pm >> k) & 1u) && w[k] > tg) ? (1u << k) : 0u;
    const bool mine = lane < G;
    const uint64_t gf = __ballot(mine && fm != 0);
    const uint64_t gp = __ballot(mine && pm != 0);
    if (gp == 0 || !(S > 0.0)) return -1;
    const bool hit = gf != 0;
    const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)gp);
    const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(pm | 1u));
    return pos_of_rt(G, T, sl, __builtin_amdgcn_readlane(my, sl));
}

}  // namespace
