// llda_gibbs.hip -- collapsed-Gibbs sweep for Labeled LDA / CascadeLDA on MI355X (gfx950, wave64).
//
// Hot path replaced: LabeledLDA.training_iteration (/root/reference/LabeledLDA.py:101-125) ==
// SubLDA.training_iteration (/root/reference/CascadeLDA.py:397-421).  C ABI: include/llda_gibbs.h.
//
// Execution model (DESIGN.md sections 3-4):
//   * group layout: one lane GROUP of G = 8..64 lanes per document (64/G documents per wavefront), T topic
//     slots per lane; topic k lives at (lane, slot) chosen so that numpy's pairwise summation order (np.sum at
//     LabeledLDA.py:117) is lane-local: one accumulator chain = one lane walking its slots, the 8 accumulators
//     of a 128-topic leaf = 8 neighbouring lanes;
//   * the n_kw row of the current word is one contiguous KP*4-byte read (word-major layout), prefetched one
//     site ahead; count updates are int32 atomics into n_kw_delta and, through an LDS accumulator per workgroup,
//     into n_k_delta.  Snapshot semantics + integer atomics => bit-deterministic;
//   * the draw is TIERED: an fp32 decision with a proven margin, an fp64 decision, and the
//     reference's fp64 pipeline bit for bit (IEEE division, numpy-ordered sum, keyed Philox4x32-10 draw) for
//     the sites the cheaper tiers cannot decide -- the chosen topic is always the exact pipeline's.
// No MFMA (gather/scan, not a contraction).  FMA contraction is OFF: the reference rounds after every ufunc.
//
// Contents
//   1. helpers: Philox, row loads, one-hot updates, exact division, DPP / permlane cross-lane moves
//   2. numpy-ordered group sum, keyed categorical draw (exact), tier-0 (fp32) decision, cold tiers (fp64)
//   3. llda_sweep_exact_kernel   general kernel, every site through the exact pipeline
//      llda_sweep_kernel         tiered kernel, per-document state in LDS (the one that runs in practice)
//      llda_sweep_sparse_kernel  one lane per ALLOWED topic for sparse label sets; hands undecided documents
//                                to llda_sweep_kernel (resume list)
//   4. llda_loglik_kernel, llda_foldin_kernel (test-time sampler), llda_readout_phi / _theta (thinning
//      read-outs), llda_apply_delta, llda_count_init, self test
//   5. host side: layout (llda_layout_init), dispatch, C entry points
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

#include "llda_gibbs.h"

#pragma clang fp contract(off)

namespace {

thread_local int g_last_hip_error = 0;

#define LLDA_MAX_LIVE 64   // most allowed topics per document the sparse kernel handles

struct KParams {
    const int64_t *doc_off;
    const int32_t *doc_order;
    const int32_t *word;
    const int32_t *freq;
    int32_t *z;
    const uint16_t *lab_mask;
    int32_t *n_dk;
    const int32_t *n_kw;
    int32_t *n_kw_delta;
    const int32_t *n_k;
    int32_t *n_k_delta;
    int32_t *status;
    int64_t D;
    int64_t doc_base;
    double alpha, beta, vbeta;
    uint32_t key0, key1, sweep, stream_id;
    int32_t dpg;
    int32_t last_leaf;      // index of the last (tail-carrying) leaf
    int32_t tail, tail_row;
    int32_t n_rounds;
    int32_t xor_tree;       // leaves combine as p^1, p^2, p^4 (balanced recursion, no padded leaves)
    double margin_rel;      // tier-1 decision margin relative to the total score (2^-40; debug: wider / inf)
    float margin0_rel;      // tier-0 (fp32) margin (2^-16; >= 1 disables tier 0)
    // sparse-label path: per document the device positions of its allowed topics, ascending
    const int64_t *live_off;
    const int32_t *live_pos;
    int32_t KP;             // row length (the sparse kernel is not templated on the layout)
    // hand-over from the sparse kernel to the dense tiered kernel: documents whose draw the sparse kernel
    // could not decide within its margin continue there from the recorded site
    int32_t *resume;        // [cap][2 + LLDA_MAX_LIVE]: doc, site, n_dk delta of the live topics so far
    int32_t *resume_count;  // [1]
    int32_t resume_cap;
    int32_t resume_mode;    // 1: this launch of the dense kernel walks the resume list instead of all documents
    // commit log (both NULL: n_kw_delta atomics): one word per site at its word-major position
    const int32_t *csc_pos;
    uint32_t *commit_log;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];   // 4 bits per leaf: partner leaf
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Counter (c0..c3), key (k0,k1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                              uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// T contiguous int32 starting at p (p is 4*T-byte aligned when T is a multiple of 4).
template <int T>
__device__ __forceinline__ void load_row(const int32_t *__restrict__ p, int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int4 v = reinterpret_cast<const int4 *>(p)[i];
            x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
        }
    } else if constexpr (T == 2) {
        const int2 v = *reinterpret_cast<const int2 *>(p);
        x[0] = v.x; x[1] = v.y;
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = p[i];
    }
}

template <int T>
__device__ __forceinline__ void store_row(int32_t *__restrict__ p, const int (&x)[T])
{
    if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; ++i)
            reinterpret_cast<int4 *>(p)[i] = make_int4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
    } else if constexpr (T == 2) {
        *reinterpret_cast<int2 *>(p) = make_int2(x[0], x[1]);
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) p[i] = x[i];
    }
}

// One-hot slot updates without compares (hipcc turns "(bit) * f" back into v_cmp + v_cndmask and
// spills the masks): m = v_bfe_i32(onehot, S, 1) is 0 or -1, value += m * g via v_mad_i32_i24
// (|g| < 2^23: g is a word frequency inside one document).
template <int S>
__device__ __forceinline__ int onehot_bit(uint32_t oh)
{
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(oh), "n"(S));
    return m;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a[S] += m_S * g and b[S] += m_S * g for every slot S (m_S = 0 / -1)
template <int T, int S = 0>
__device__ __forceinline__ void onehot_add2(int (&a)[T], int (&b)[T], uint32_t oh, int g)
{
    if constexpr (S < T) {
        const int m = onehot_bit<S>(oh);
        a[S] = mad_i24(m, g, a[S]);
        b[S] = mad_i24(m, g, b[S]);
        onehot_add2<T, S + 1>(a, b, oh, g);
    }
}
template <int T, int S = 0>
__device__ __forceinline__ void onehot_add1(int (&a)[T], uint32_t oh, int g)
{
    if constexpr (S < T) {
        a[S] = mad_i24(onehot_bit<S>(oh), g, a[S]);
        onehot_add1<T, S + 1>(a, oh, g);
    }
}

template <int T, int S = 0>
__device__ __forceinline__ void scores(double (&w)[T], const int (&ndk)[T], const int (&nkb)[T], const int (&x)[T],
                                       uint32_t mask, double alpha, double beta, double vbeta)
{
    if constexpr (S < T) {
        const double a = (double)ndk[S] + alpha;
        const double num_b = (double)x[S] + beta;
        const double den_b = (double)(nkb[S] + ndk[S]) + vbeta;
        const double ws = a * (num_b / den_b);
        const long long m = (long long)onehot_bit<S>(mask);          // 0 or -1, sign-extended
        w[S] = __longlong_as_double(__double_as_longlong(ws) & m);
        scores<T, S + 1>(w, ndk, nkb, x, mask, alpha, beta, vbeta);
    }
}

// a / b given y = RN(1/b) (IEEE), correctly rounded: q0 = RN(a y); two exact-residual corrections.
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
}

// ---------------------------------------------------------------------------------------------
// Cross-lane moves of doubles without an LDS round trip (DPP / permlane), gfx950.
// ---------------------------------------------------------------------------------------------
// DPP move; lanes whose source lane is outside the row (row_shr) or the wave (wave_shr) receive 0.0.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane j <- lane 7-j of its 8-lane half (an "xor 4" once quads are uniform)
constexpr int DPP_ROW_SHR = 0x110;      // + n
constexpr int DPP_ROW_ROR = 0x120;      // + n
constexpr int DPP_WAVE_SHR1 = 0x138;

// v_permlane16_swap vdst, src: odd 16-lane rows of vdst <-> even rows of src.  With both operands x:
// r[0] = [R0,R0,R2,R2] (odd rows see the row below), r[1] = [R1,R1,R3,R3] (even rows see the row above).
__device__ __forceinline__ void rows_swapped(double x, double &below, double &above)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    below = __hiloint2double((int)hi[0], (int)lo[0]);
    above = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double xor16_f64(double x, int lane)
{
    double below, above;
    rows_swapped(x, below, above);
    return (lane & 16) ? below : above;
}
// v_permlane32_swap vdst, src: upper half of vdst <-> lower half of src.  r[0] = [lo,lo], r[1] = [hi,hi].
__device__ __forceinline__ double xor32_f64(double x, int lane)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return (lane & 32) ? __hiloint2double((int)hi[0], (int)lo[0]) : __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double readlane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}
// value of the last lane of the caller's group
template <int G>
__device__ __forceinline__ double bcast_last(double x, int lane)
{
    if constexpr (G == 64) {
        return readlane_f64(x, 63);
    } else if constexpr (G == 32) {
        const double a = readlane_f64(x, 31), b = readlane_f64(x, 63);
        return (lane & 32) ? b : a;
    } else if constexpr (G == 16) {
        const double a = readlane_f64(x, 15), b = readlane_f64(x, 31), c = readlane_f64(x, 47), d = readlane_f64(x, 63);
        const double ab = (lane & 16) ? b : a, cd = (lane & 16) ? d : c;
        return (lane & 32) ? cd : ab;
    } else {
        return __shfl(x, G - 1, G);
    }
}
// one Hillis-Steele step of the inclusive scan over the G lanes of a group: X[g] = X[g-D] + X[g], g >= D
template <int G, int D>
__device__ __forceinline__ double scan_step(double X, int lig)
{
    if constexpr (G == 64) {
        const double y = __shfl_up(X, D, G);
        return (lig >= D) ? y + X : X;
    } else if constexpr (G == 32) {
        if constexpr (D < 16) {
            const double y = dpp_f64<DPP_ROW_ROR + D>(X);      // lane i <- lane (i-D) mod 16 of its row
            double below, above;
            rows_swapped(y, below, above);                     // odd rows: the same rotation of the row below
            const double src = ((lig & 15) >= D) ? y : ((lig >= 16) ? below : 0.0);
            return src + X;
        } else {
            double below, above;
            rows_swapped(X, below, above);
            return ((lig >= 16) ? below : 0.0) + X;
        }
    } else {
        const double y = dpp_f64<DPP_ROW_SHR + D>(X);          // 0.0 shifted in at the row start
        if constexpr (G == 16) return y + X;
        else return ((lig >= D) ? y : 0.0) + X;                // 8-lane groups share a row
    }
}
template <int G, int D = 1>
__device__ __forceinline__ double group_scan(double X, int lig)
{
    if constexpr (D < G) return group_scan<G, D * 2>(scan_step<G, D>(X, lig), lig);
    else return X;
}

// value held by lane `src` (same for the whole group, but a run-time value) in every lane of the group.
// 8- and 16-lane groups: OR-butterfly over DPP moves (no LDS round trip); wider groups: ds_bpermute.
template <int G>
__device__ __forceinline__ int group_pick(int v, int src, int lig)
{
    if constexpr (G <= 16) {
        int x = (lig == src) ? v : 0;
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR1, 0xF, 0xF, false);
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR2, 0xF, 0xF, false);
        x |= __builtin_amdgcn_update_dpp(0, x, DPP_HALF_MIRROR, 0xF, 0xF, false);
        if constexpr (G == 16) x |= __builtin_amdgcn_update_dpp(0, x, DPP_ROW_ROR + 8, 0xF, 0xF, false);
        return x;
    } else {
        return __shfl(v, src, G);
    }
}

// Sum of the group's K scores in numpy's pairwise order.  Every lane of the group returns S.
//   chain : per-lane sequential sum over its slots (one of numpy's 8 accumulators)
//   xor butterfly 1,2,4 : ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))   (fp add is commutative)
//   tail  : n % 8 leftovers of the last leaf, added sequentially
//   leaves: combined along numpy's recursion tree by the partner schedule
// cross-lane part of the sum: acc = this lane's chain, tv = this lane's tail element
template <int G, bool HAS_TAIL>
__device__ __forceinline__ double group_sum_tail(double acc, double tv, const KParams &P, int lig, int lane)
{
    const int leaf = lig >> 3;
    acc = acc + dpp_f64<DPP_XOR1>(acc);
    acc = acc + dpp_f64<DPP_XOR2>(acc);
    acc = acc + dpp_f64<DPP_HALF_MIRROR>(acc);
    if (HAS_TAIL) {
        for (int t = 0; t < P.tail; ++t) {
            const double o = __shfl(tv, P.last_leaf * 8 + t, G);
            if (leaf == P.last_leaf) acc = acc + o;
        }
    }
    if constexpr (G > 8) {
        if (P.xor_tree) {
            // balanced recursion (leaf p pairs with p^1, then p^2, p^4): lane xor 8 / 16 / 32
            acc = acc + dpp_f64<DPP_ROW_ROR + 8>(acc);
            if constexpr (G > 16) acc = acc + xor16_f64(acc, lane);
            if constexpr (G > 32) acc = acc + xor32_f64(acc, lane);
        } else {
#pragma unroll
            for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) {
                if (r < P.n_rounds) {
                    const int partner = (P.rounds_pk[r] >> (4 * leaf)) & 15;
                    const double o = __shfl(acc, partner * 8 + (lig & 7), G);
                    if (partner != leaf) acc = acc + o;
                }
            }
            acc = __shfl(acc, 0, G);
        }
    }
    return acc;
}

template <int G, int T, bool HAS_TAIL>
__device__ __forceinline__ double group_sum(const double (&w)[T], const KParams &P, int lig, int lane)
{
    const int leaf = lig >> 3;
    double acc = 0.0, tv = 0.0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
        if (HAS_TAIL && s == P.tail_row && leaf == P.last_leaf) tv = w[s];
        else acc = acc + w[s];
    }
    return group_sum_tail<G, HAS_TAIL>(acc, tv, P, lig, lane);
}

// Keyed categorical draw over the group's K probabilities p (device order, oracle/llda_oracle.py
// draw_keyed): q = per-lane prefix over the slots, X = Hillis-Steele scan of the lane totals,
// t = u * X[G-1]; result = first position with p > 0 and q > t - X[lane-1], else the last position with
// p > 0; -1 if there is none (or !valid).  FAST: "p > 0" is read off the label mask.
template <int G, int T, bool FAST>
__device__ __forceinline__ int draw_position(const double (&w)[T], double u, uint32_t mask, bool valid, int lig, int lane)
{
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    double q[T];
    q[0] = w[0];
#pragma unroll
    for (int s = 1; s < T; ++s) q[s] = q[s - 1] + w[s];
    const double X = group_scan<G>(q[T - 1], lig);
    const double tot = bcast_last<G>(X, lane);
    const double t = u * tot;
    const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
    const double tg = t - (lig ? prev : 0.0);
    uint32_t fm = 0, pm = 0;
    if (FAST) {
        // q is non-decreasing along the slots, so {s : q[s] > tg} is the suffix starting at
        // cnt = #{s : q[s] <= tg}; positive-probability slots are the label-mask bits.
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < T; ++s) cnt += (q[s] <= tg) ? 1 : 0;
        pm = mask;
        fm = mask & (0xFFFFu << cnt);
    } else {
#pragma unroll
        for (int s = 0; s < T; ++s) {
            const bool pos = w[s] > 0.0;
            pm |= (pos ? 1u : 0u) << s;
            fm |= ((pos && q[s] > tg) ? 1u : 0u) << s;
        }
    }
    const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
    const uint64_t gp = (__ballot(pm != 0) >> gbase) & gmask;
    int zn = -1;
    if (gp != 0 && valid) {
        const bool hit = gf != 0;
        const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)gp);
        const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(pm | 1u));
        const int ss = __shfl(my, sl, G);
        zn = sl * T + ss;
    }
    return zn;
}

// ---------------------------------------------------------------------------------------------
// Two-tier draw (FAST kernels).  The integer state only depends on WHICH topic the draw picks, i.e. on
// the signs of  E[g][s] = q[g][s] - (t - X[g-1])  in the exact fp64 pipeline above.  Tier 1 evaluates
// the same comparison from unnormalised, cheaply rounded scores
//     w~ = a * (num_b * RN(1/den_b)),   Q~ = prefix(w~),   X~ = scan,   T~ = u * X~[G-1] - X~[g-1]
// (no division, no pairwise sum, no normalisation).  Relative to the total every quantity differs from
// its exact counterpart by at most a few hundred units of 2^-53 (DESIGN.md section 4.3 derives
// |E~ - E| <= 2^-44 of the total), so whenever every |Q~ - T~| exceeds 2^-40 of the total the signs -- and
// with them the chosen position -- are those of the exact pipeline.  Otherwise (probability ~1e-9 per
// site) the group falls back to the exact tier.  Returns false when the group must fall back.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Tier 0: the same decision in fp32.  Every quantity is within 103 * 2^-24 (< 2^-17.3) of the total of its
// real-number value, the exact pipeline within 2^-44; with a margin of 2^-16 of
// the total a "sure" fp32 decision therefore has the signs of the exact pipeline.  About 1.6 % of the
// sites (K = 512) are "unsure" and go on to tier 1.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}

template <int T, bool DENSE, int S = 0>
__device__ __forceinline__ void prefix_scores_f32(float (&qw)[T], const int (&x)[T], const float (*s_pa)[256], int tid,
                                                  uint32_t mask, float beta)
{
    if constexpr (S < T) {
        float ws = ((float)x[S] + beta) * s_pa[S][tid];          // num_b * fl32(a / den_b)
        if constexpr (!DENSE) ws = __int_as_float(__float_as_int(ws) & onehot_bit<S>(mask));
        if constexpr (S == 0) qw[0] = ws;
        else qw[S] = qw[S - 1] + ws;
        prefix_scores_f32<T, DENSE, S + 1>(qw, x, s_pa, tid, mask, beta);
    }
}

// inclusive scan over the G lanes of a group (any association order will do here)
template <int G>
__device__ __forceinline__ float group_scan_f32(float X, int lig)
{
    if constexpr (G == 8) {
        float y;
        y = dpp_f32<DPP_ROW_SHR + 1>(X); X += (lig >= 1) ? y : 0.0f;
        y = dpp_f32<DPP_ROW_SHR + 2>(X); X += (lig >= 2) ? y : 0.0f;
        y = dpp_f32<DPP_ROW_SHR + 4>(X); X += (lig >= 4) ? y : 0.0f;
    } else {
        X += dpp_f32<DPP_ROW_SHR + 1>(X);
        X += dpp_f32<DPP_ROW_SHR + 2>(X);
        X += dpp_f32<DPP_ROW_SHR + 4>(X);
        X += dpp_f32<DPP_ROW_SHR + 8>(X);
        if constexpr (G >= 32) X += dpp_f32<0x142, 0xA>(X);     // row_bcast:15 into rows 1 and 3
        if constexpr (G == 64) X += dpp_f32<0x143, 0xC>(X);     // row_bcast:31 into rows 2 and 3
    }
    return X;
}

template <int G>
__device__ __forceinline__ float bcast_last_f32(float x, int lane)
{
    if constexpr (G == 64) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    } else if constexpr (G == 32) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
        return (lane & 32) ? b : a;
    } else if constexpr (G == 16) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 15));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 47));
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
        const float ab = (lane & 16) ? b : a, cd = (lane & 16) ? d : c;
        return (lane & 32) ? cd : ab;
    } else {
        return __int_as_float(group_pick<G>(__float_as_int(x), G - 1, lane & (G - 1)));
    }
}

template <int G, int T>
__device__ __forceinline__ bool draw_fast_f32(const float (&qw)[T], float u, uint32_t mask, uint64_t gp,
                                              float margin_rel, int lig, int lane, int &zn)
{
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    const float X = group_scan_f32<G>(qw[T - 1], lig);
    const float tot = bcast_last_f32<G>(X, lane);
    const float prev = dpp_f32<DPP_WAVE_SHR1>(X);
    const float tg = u * tot - (lig ? prev : 0.0f);
    const float margin = tot * margin_rel;
    const float lo = tg - margin, hi = tg + margin;
    int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
        cnt_lo += (qw[s] <= lo) ? 1 : 0;
        cnt_hi += (qw[s] <= hi) ? 1 : 0;
    }
    const bool unsure = (cnt_lo != cnt_hi) || !(tot > 0.0f) || !(margin < tot) || !(tot < 3.0e38f);
    if (((__ballot(unsure) >> gbase) & gmask) != 0) return false;
    const uint32_t fm = mask & (0xFFFFu << cnt_lo);
    const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
    const bool hit = gf != 0;
    const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)(gp | 1ull));
    const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
    zn = sl * T + group_pick<G>(my, sl, lig);
    return true;
}

// Cold tiers of the FAST kernels, out of line: they run for the ~1.6 % of the sites
// tier 0 is unsure about and work on scratch copies, so the hot loop's register allocation never sees them.
//   tier 1: the decision from unnormalised fp64 prefix sums, margin 2^-40 of the total;
//   exact : the reference's fp64 pipeline bit for bit -- scores, numpy-ordered sum, p = fl(w/S) through
//           ONE IEEE reciprocal y = RN(1/S) and two residual corrections per slot (Markstein: with y
//           correctly rounded and q1 faithful, q2 = RN(q1 + (w - S q1) y) is the correctly rounded
//           quotient; llda_selftest_div checks it against the hardware division), keyed draw.
// Returns the chosen device position or -1.
template <int G, int T, bool HAS_TAIL, bool DENSE>
__device__ __noinline__ int cold_tiers(const int (*s_ndk)[256], const int *x, const int (*s_nkc)[256], int tid,
                                       uint32_t mask, double u, int lig, int lane, const KParams *P)
{
    // Written as rolled loops over scratch arrays on purpose: few registers, so that this rarely taken
    // function does not dictate the kernel's register allocation (occupancy of the hot loop).
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    const double alpha = P->alpha, beta = P->beta, vbeta = P->vbeta;
    const uint32_t lmask = DENSE ? 0xFFFFu : mask;
    double w[T];
    if (lig == 0 && P->status) atomicAdd(P->status + 1, 1);      // statistics: sites tier 0 was unsure about
    // ---- tier 1: unnormalised fp64 prefix sums, margin 2^-40 of the total ----
    {
        double run = 0.0;
#pragma unroll 4
        for (int s = 0; s < T; ++s) {
            // 1/den to within 2^-50: hardware estimate + two Newton steps (tier 1 only needs a few 2^-53)
            const double den = (double)s_nkc[s][tid] + vbeta;
            double y = __builtin_amdgcn_rcp(den);
            y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
            y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
            const double ws = ((double)s_ndk[s][tid] + alpha) * (((double)x[s] + beta) * y);
            run = run + (((lmask >> s) & 1u) ? ws : 0.0);
            w[s] = run;
        }
        const double X = group_scan<G>(run, lig);
        const double tot = bcast_last<G>(X, lane);
        const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
        const double tg = u * tot - (lig ? prev : 0.0);
        const double margin = tot * P->margin_rel;
        int cnt_lo = 0, cnt_hi = 0;
#pragma unroll 4
        for (int s = 0; s < T; ++s) {
            cnt_lo += (w[s] <= tg - margin) ? 1 : 0;
            cnt_hi += (w[s] <= tg + margin) ? 1 : 0;
        }
        const bool unsure = (cnt_lo != cnt_hi) || !(tot > 0.0) || !(margin < tot);
        if (((__ballot(unsure) >> gbase) & gmask) == 0) {
            const uint32_t fm = mask & (0xFFFFu << cnt_lo);
            const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
            const uint64_t gp = (__ballot(mask != 0) >> gbase) & gmask;
            const bool hit = gf != 0;
            const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)(gp | 1ull));
            const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
            return sl * T + __shfl(my, sl, G);
        }
    }
    // ---- exact tier: the reference's fp64 pipeline, bit for bit ----
    if (lig == 0 && P->status) { atomicOr(P->status, 2); atomicAdd(P->status + 2, 1); }   // the exact tier ran
    const int leaf = lig >> 3;
    double acc = 0.0, tv = 0.0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        // prob = lab * a * (num_b / den_b)   (LabeledLDA.py:113-116)
        const double ws = ((double)s_ndk[s][tid] + alpha) * (((double)x[s] + beta) / ((double)s_nkc[s][tid] + vbeta));
        const double v = ((lmask >> s) & 1u) ? ws : 0.0;
        w[s] = v;
        if (HAS_TAIL && s == P->tail_row && leaf == P->last_leaf) tv = v;
        else acc = acc + v;
    }
    const double S = group_sum_tail<G, HAS_TAIL>(acc, tv, *P, lig, lane);      // np.sum(prob), LabeledLDA.py:117
    const double y = 1.0 / S;
    // prob /= np.sum(prob); keyed draw: per-lane prefix, Hillis-Steele scan, first slot with p > 0 and q > t - X[g-1]
    double run = 0.0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) {
        run = (s == 0) ? div_by(w[s], S, y) : run + div_by(w[s], S, y);
        w[s] = run;
    }
    const double X = group_scan<G>(run, lig);
    const double tot = bcast_last<G>(X, lane);
    const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
    const double tg = u * tot - (lig ? prev : 0.0);
    int cnt = 0;
#pragma unroll 1
    for (int s = 0; s < T; ++s) cnt += (w[s] <= tg) ? 1 : 0;
    const uint32_t fm = mask & (0xFFFFu << cnt);
    const uint64_t gf = (__ballot(fm != 0) >> gbase) & gmask;
    const uint64_t gp = (__ballot(mask != 0) >> gbase) & gmask;
    if (gp == 0 || !(S > 0.0)) return -1;
    const bool hit = gf != 0;
    const int sl = hit ? (int)__ffsll((unsigned long long)gf) - 1 : 63 - (int)__clzll((unsigned long long)gp);
    const int my = hit ? (int)__ffs((int)(fm | 0x10000u)) - 1 : 31 - (int)__clz((int)(mask | 1u));
    return sl * T + __shfl(my, sl, G);
}

// store the new assignment of a site and move its count in n_kw_delta (int32 atomics, no return value)
// (c = the site's position in the commit log when there is one; v, f are only needed without a log)
__device__ __forceinline__ void commit_site(const KParams &P, int64_t i, int v, int f, int zo, int zn, int c, int KP)
{
    P.z[i] = zn;
    if (P.commit_log) {
        P.commit_log[c] = (uint32_t)zo | ((uint32_t)zn << 16);
    } else if (zn != zo) {
        int32_t *row = P.n_kw_delta + (int64_t)v * KP;
        atomicAdd(row + zo, -f);
        atomicAdd(row + zn, f);
    }
}

// ---------------------------------------------------------------------------------------------
// The sweep kernel
// ---------------------------------------------------------------------------------------------
// Per-site pieces shared by the two sweep kernels ------------------------------------------------
// random bits of site n: one Philox block serves sites 2b and 2b+1; the G lanes of the group compute G
// consecutive blocks at once (every 2G sites, or at the first site of a resumed document) and hand them out
template <int G>
__device__ __forceinline__ void site_random_bits(const KParams &P, int n, bool first, uint32_t gdoc, int lig,
                                                 uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3,
                                                 uint32_t &ra, uint32_t &rb)
{
    if (first || (n & (2 * G - 1)) == 0) {
        r0 = (uint32_t)((n >> 1) & ~(G - 1)) + (uint32_t)lig; r1 = gdoc; r2 = P.stream_id; r3 = P.sweep;
        philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
    }
    const int holder = (n >> 1) & (G - 1);
    ra = (uint32_t)group_pick<G>((int)((n & 1) ? r2 : r0), holder, lig);
    rb = (uint32_t)group_pick<G>((int)((n & 1) ? r3 : r1), holder, lig);
}

// the 53-bit keyed uniform u = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 (every operation exact)
__device__ __forceinline__ double uniform53(uint32_t ra, uint32_t rb)
{
    return ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
}

template <int G>
__device__ __forceinline__ double site_uniform(const KParams &P, int n, bool first, uint32_t gdoc, int lig,
                                               uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3)
{
    uint32_t ra, rb;
    site_random_bits<G>(P, n, first, gdoc, lig, r0, r1, r2, r3, ra, rb);
    return uniform53(ra, rb);
}

// ---------------------------------------------------------------------------------------------
// Sweep kernel, general form: every site through the reference's fp64 pipeline, inline.  Used when the
// tiered kernel's preconditions do not hold (alpha or beta < 1e-6, V*beta >= 2^40).
// ---------------------------------------------------------------------------------------------
template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_sweep_exact_kernel(const KParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;              // lane groups (documents in flight) per workgroup
    __shared__ int s_nk[KP];                  // workgroup accumulator of the n_k changes
    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    __syncthreads();
    const int lane = tid & 63;
    const int lig = tid & (G - 1);            // lane in group
    const int grp = tid / G;

    for (int it = 0; it < P.dpg; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * GPB + grp;
        if (idx >= P.D) break;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[idx] : idx;
        const int64_t s0 = P.doc_off[d];
        const int len = (int)(P.doc_off[d + 1] - s0);
        if (len <= 0) continue;

        int ndk[T], nkb[T];           // nkb = n_k(sweep start) - n_dk(sweep start): n_k seen by the
        int32_t *ndk_row = P.n_dk + d * KP + lig * T;   // document is nkb + ndk at any time
        load_row<T>(ndk_row, ndk);
        load_row<T>(P.n_k + lig * T, nkb);
#pragma unroll
        for (int s = 0; s < T; ++s) nkb[s] -= ndk[s];
        const uint32_t mask = P.lab_mask[d * G + lig];
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);

        // memory pipeline: see llda_sweep_kernel
        int v_c = P.word[s0], f_c = P.freq[s0], zo_c = P.z[s0], c_c = P.csc_pos ? P.csc_pos[s0] : 0;
        const int64_t i1 = s0 + (len > 1 ? 1 : 0);
        int v_1 = P.word[i1], f_1 = P.freq[i1], zo_1 = P.z[i1], c_1 = P.csc_pos ? P.csc_pos[i1] : 0;
        int xn[T];
        load_row<T>(P.n_kw + (int64_t)v_c * KP + lig * T, xn);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        int64_t pend_i = -1;
        int pend_v = 0, pend_f = 0, pend_zo = 0, pend_zn = 0, pend_c = 0;

        for (int n = 0; n < len; ++n) {
            const int v = v_c, f = f_c, zo = zo_c, c = c_c;
            int x[T];
#pragma unroll
            for (int s = 0; s < T; ++s) x[s] = xn[s];
            if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);
            load_row<T>(P.n_kw + (int64_t)v_1 * KP + lig * T, xn);
            v_c = v_1; f_c = f_1; zo_c = zo_1; c_c = c_1;
            {
                const int64_t i2 = s0 + (n + 2 < len ? n + 2 : len - 1);
                v_1 = P.word[i2]; f_1 = P.freq[i2]; zo_1 = P.z[i2];
                if (P.csc_pos) c_1 = P.csc_pos[i2];
            }
            const double u = site_uniform<G>(P, n, n == 0, gdoc, lig, r0, r1, r2, r3);

            // remove the site (LabeledLDA.py:109-111)
            {
                const int lo = zo / T, so = zo - lo * T;
                onehot_add2<T>(ndk, x, (lig == lo) ? (1u << so) : 0u, f);
            }
            // scores (LabeledLDA.py:113-116), np.sum, prob /= sum, keyed draw
            double w[T];
            scores<T>(w, ndk, nkb, x, mask, P.alpha, P.beta, P.vbeta);
            const double S = group_sum<G, T, HAS_TAIL>(w, P, lig, lane);
            const double y = 1.0 / S;
#pragma unroll
            for (int s = 0; s < T; ++s) w[s] = div_by(w[s], S, y);
            int zn = draw_position<G, T, false>(w, u, mask, S > 0.0, lig, lane);
            if (zn < 0) {
                zn = zo;
                if (lig == 0 && P.status) atomicOr(P.status, 1);    // no topic with positive probability
            }
            // add the site back (LabeledLDA.py:121-125)
            {
                const int ln = zn / T, sn = zn - ln * T;
                onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);
            }
            pend_i = s0 + n; pend_v = v; pend_f = f; pend_zo = zo; pend_zn = zn; pend_c = c;
        }
        if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);

        int old[T];
        load_row<T>(ndk_row, old);
#pragma unroll
        for (int s = 0; s < T; ++s) {
            const int dl = ndk[s] - old[s];
            if (dl) atomicAdd(&s_nk[lig * T + s], dl);
        }
        store_row<T>(ndk_row, ndk);
    }
    __syncthreads();
    for (int i = tid; i < KP; i += 256) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

// ---------------------------------------------------------------------------------------------
// Sweep kernel, tiered form (the one that runs in practice).  Preconditions, host-checked in llda_sweep:
// alpha, beta >= 1e-6 (every label-allowed topic has a strictly positive probability, so the "p > 0" tests
// of the draw can be read off the label mask) and V*beta < 2^40.  DENSE (K == KP): every document allows
// every topic and the label mask is not applied.
// The document's n_dk row, the n_k it sees and an fp32 reciprocal of n_k + V*beta live in LDS as
// [slot][thread] arrays: conflict-free, and the owning lane updates ONE dynamically indexed slot per
// change (VGPR arrays would need a 16-deep select chain per update).
// ---------------------------------------------------------------------------------------------
#ifndef LLDA_MARGIN0
#define LLDA_MARGIN0 0x1p-16f   // tier-0 (fp32) decision margin relative to the total score
#endif
#ifndef LLDA_WAVES
#define LLDA_WAVES 3          // waves per SIMD the register allocator must leave room for
#endif

// tier-0 factor of one topic: fl32(a * y) with a = fl32(n_dk + alpha), y = v_rcp_f32(fl32(n_k + V*beta)).
// n_dk and n_k of a topic always change together, so the product is cached as ONE float per slot.
__device__ __forceinline__ float tier0_factor(int ndk, int nk, float alpha32, float vbeta32)
{
    return ((float)ndk + alpha32) * __builtin_amdgcn_rcpf((float)nk + vbeta32);
}

// a topic count of this document changes by df: n_dk, the n_k the document sees, and the cached tier-0 factor
__device__ __forceinline__ void count_update(int (*s_ndk)[256], int (*s_nkc)[256], float (*s_pa)[256], int slot,
                                             int tid, float alpha32, float vbeta32, int df)
{
    const int nd = s_ndk[slot][tid] + df, nk = s_nkc[slot][tid] + df;
    s_ndk[slot][tid] = nd;
    s_nkc[slot][tid] = nk;
    s_pa[slot][tid] = tier0_factor(nd, nk, alpha32, vbeta32);
}

template <int G, int T, bool HAS_TAIL, bool DENSE>
__global__ void __launch_bounds__(256, LLDA_WAVES) llda_sweep_kernel(const KParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;              // lane groups (documents in flight) per workgroup
    __shared__ int s_nk[KP];                  // workgroup accumulator of the n_k changes
    __shared__ int s_ndk[T][256];             // n_dk row of the document
    __shared__ int s_nkc[T][256];             // n_k as the document sees it
    __shared__ float s_pa[T][256];            // tier-0 factor fl32((n_dk + alpha) / (n_k + V*beta))

    const int tid = threadIdx.x;
    for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    __syncthreads();

    const int lane = tid & 63;
    const int lig = tid & (G - 1);            // lane in group
    const int grp = tid / G;
    const float vbeta32 = (float)P.vbeta, alpha32 = (float)P.alpha, beta32 = (float)P.beta;

    const int n_resume = P.resume_mode ? min(*P.resume_count, P.resume_cap) : 0;
    for (int it = 0; P.resume_mode || it < P.dpg; ++it) {
        int64_t d;
        int n0 = 0;                           // first site to sample (resume mode: where the sparse kernel stopped)
        const int32_t *rec = nullptr;
        if (P.resume_mode) {
            const int64_t idx = ((int64_t)it * gridDim.x + blockIdx.x) * GPB + grp;
            if (idx >= n_resume) break;
            rec = P.resume + idx * (2 + LLDA_MAX_LIVE);
            d = rec[0];
            n0 = rec[1];
        } else {
            const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * GPB + grp;
            if (idx >= P.D) break;
            d = P.doc_order ? (int64_t)P.doc_order[idx] : idx;
        }
        const int64_t s0 = P.doc_off[d];
        const int len = (int)(P.doc_off[d + 1] - s0);
        if (len <= n0) continue;

        int32_t *ndk_row = P.n_dk + d * KP + lig * T;
        {
            int r[T], k[T];
            load_row<T>(ndk_row, r);
            load_row<T>(P.n_k + lig * T, k);
#pragma unroll
            for (int s = 0; s < T; ++s) {
                s_ndk[s][tid] = r[s];
                s_nkc[s][tid] = k[s];                              // sweep-start n_k
                s_pa[s][tid] = tier0_factor(r[s], k[s], alpha32, vbeta32);
            }
        }
        if (rec) {
            // resumed document: the n_k it sees already moved by its own earlier sites (n_dk row holds them)
            const int64_t l0 = P.live_off[d];
            const int A = (int)(P.live_off[d + 1] - l0);
            for (int j = 0; j < A; ++j) {
                const int pos = P.live_pos[l0 + j], dl = rec[2 + j];
                if (dl != 0 && lig == pos / T) {
                    const int nk = s_nkc[pos % T][tid] + dl;
                    s_nkc[pos % T][tid] = nk;
                    s_pa[pos % T][tid] = tier0_factor(s_ndk[pos % T][tid], nk, alpha32, vbeta32);
                }
            }
        }
        const uint32_t mask = P.lab_mask[d * G + lig];
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);
        const uint64_t gp_doc = (__ballot(mask != 0) >> (lane & ~(G - 1))) & ((G == 64) ? ~0ull : ((1ull << G) - 1ull));   // lanes with an allowed topic

        // Software pipeline of the memory operations: at the top of iteration n the registers hold the
        // scalars (word, freq, z) of site n, the row of site n is in flight (xn) and so are the scalars of
        // site n+1.  Right after the single s_waitcnt vmcnt(0) of the iteration (first use of xn) the body
        // issues, in this order: the z store + two n_kw_delta atomics of site n-1, the row of site n+1,
        // the scalars of site n+2 -- so nothing the next wait covers is younger than one full site.
        int v_c = P.word[s0 + n0], f_c = P.freq[s0 + n0], zo_c = P.z[s0 + n0], c_c = P.csc_pos ? P.csc_pos[s0 + n0] : 0;
        const int64_t i1 = s0 + (n0 + 1 < len ? n0 + 1 : n0);
        int v_1 = P.word[i1], f_1 = P.freq[i1], zo_1 = P.z[i1], c_1 = P.csc_pos ? P.csc_pos[i1] : 0;
        int xn[T];
        load_row<T>(P.n_kw + (int64_t)v_c * KP + lig * T, xn);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        int64_t pend_i = -1;
        int pend_v = 0, pend_f = 0, pend_zo = 0, pend_zn = 0, pend_c = 0;
        {   // site 0 leaves its topic (LabeledLDA.py:109-111); later sites do so at the end of the loop body
            const int lo = zo_c / T;
            if (lig == lo) count_update(s_ndk, s_nkc, s_pa, zo_c - lo * T, tid, alpha32, vbeta32, -f_c);
        }

        for (int n = n0; n < len; ++n) {
            const int v = v_c, f = f_c, zo = zo_c, c = c_c;
            int x[T];
#pragma unroll
            for (int s = 0; s < T; ++s) x[s] = xn[s];
#ifndef ABL_NOCOMMIT
            if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);
#endif
#ifndef ABL_NOLOAD
            load_row<T>(P.n_kw + (int64_t)v_1 * KP + lig * T, xn);        // row of site n+1 (clamped)
#else
#pragma unroll
            for (int s = 0; s < T; ++s) xn[s] = (v_1 + s) & 7;            // ablation: no n_kw traffic
#endif
            v_c = v_1; f_c = f_1; zo_c = zo_1; c_c = c_1;
            {
                const int64_t i2 = s0 + (n + 2 < len ? n + 2 : len - 1);  // scalars of site n+2 (clamped)
                v_1 = P.word[i2]; f_1 = P.freq[i2]; zo_1 = P.z[i2];
                if (P.csc_pos) c_1 = P.csc_pos[i2];
            }
            uint32_t ra, rb;
            site_random_bits<G>(P, n, n == n0, gdoc, lig, r0, r1, r2, r3, ra, rb);

            // the site's own count leaves the fetched n_kw row (n_dk / n_k were updated already)
            {
                const int lo = zo / T, so = zo - lo * T;
                onehot_add1<T>(x, (lig == lo) ? (1u << so) : 0u, f);       // m = -1 at the slot: += (-1) * f
            }

            // tiered draw
            int zn = -1;
            bool decided = false;
            if (P.margin0_rel < 1.0f) {           // tier 0: fp32
                float qf[T];
                prefix_scores_f32<T, DENSE>(qf, x, s_pa, tid, mask, beta32);
                // fp32 image of the uniform: the top 27 bits (within 2^-24 relative + 2^-27 absolute of u)
                const float u32 = (float)(ra >> 5) * 0x1p-27f;
                decided = draw_fast_f32<G, T>(qf, u32, mask, gp_doc, P.margin0_rel, lig, lane, zn);
            }
            if (!decided) {
                int x_c[T];
#pragma unroll
                for (int s = 0; s < T; ++s) x_c[s] = x[s];
                zn = cold_tiers<G, T, HAS_TAIL, DENSE>(s_ndk, x_c, s_nkc, tid, mask, uniform53(ra, rb), lig, lane, &P);
            }
            if (zn < 0) {
                zn = zo;
                if (lig == 0 && P.status) atomicOr(P.status, 1);    // no topic with positive probability
            }

            // add the site back (LabeledLDA.py:121-125), and take the NEXT site out of its topic already (its
            // scalars are in registers): the LDS state is final long before the next site's scores read it.
            // Both updates usually belong to different lanes and are done in ONE masked pass; a second pass runs
            // only for groups where the same lane owns both.
            {
                const int ln = zn / T;
                const bool more = n + 1 < len;
                const int lo2 = more ? zo_c / T : -1;
                const bool own_new = lig == ln, own_old = lig == lo2;
                if (own_new || own_old)
                    count_update(s_ndk, s_nkc, s_pa, own_new ? zn - ln * T : zo_c - lo2 * T, tid, alpha32, vbeta32,
                                 own_new ? f : -f_c);
                if (own_new && own_old) count_update(s_ndk, s_nkc, s_pa, zo_c - lo2 * T, tid, alpha32, vbeta32, -f_c);
            }
            pend_i = s0 + n; pend_v = v; pend_f = f; pend_zo = zo; pend_zn = zn; pend_c = c;
        }
        if (lig == 0 && pend_i >= 0) commit_site(P, pend_i, pend_v, pend_f, pend_zo, pend_zn, pend_c, KP);

        // document done: fold its n_dk change into the workgroup's n_k accumulator, store the row
        int old[T], cur[T];
        load_row<T>(ndk_row, old);
#pragma unroll
        for (int s = 0; s < T; ++s) {
            cur[s] = s_ndk[s][tid];
            const int dl = cur[s] - old[s];
            if (dl) atomicAdd(&s_nk[lig * T + s], dl);
        }
        store_row<T>(ndk_row, cur);
    }

    __syncthreads();
    for (int i = tid; i < KP; i += 256) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

// ---------------------------------------------------------------------------------------------
// Sweep kernel for sparse label sets (Labeled LDA proper: a handful of allowed topics out of hundreds,
// e.g. 4.6 of 392 on the abstracts corpus).  One lane per ALLOWED topic, GS = 8..64 lanes per document
// (64/GS documents per wavefront); per site each lane gathers its single n_kw entry, so the traffic is
// 4*A + 32 bytes instead of a 4*KP-byte row.  All per-topic state (n_dk, the n_k the document sees, the
// reciprocal of n_k + V*beta) is a scalar register of the owning lane.
// The draw is the tier-1 decision of DESIGN.md section 4.3 restricted to the live topics: inclusive scan of
// the unnormalised fp64 scores in device-position order, first lane with Q > u*total, sure when every
// |Q - u*total| exceeds 2^-40 of the total.  A document with an unsure site (probability ~1e-11 per site)
// is handed to the dense tiered kernel, which continues from that site (resume list).
// Preconditions as for llda_sweep_kernel (alpha, beta >= 1e-6, V*beta < 2^40).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton(double den)
{
    double y = __builtin_amdgcn_rcp(den);                    // hardware estimate, then two Newton steps:
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);    // within a few 2^-53 of 1/den
    return __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
}

// inclusive scan over the GS lanes of a group, any association order (DPP within 16-lane rows + row carries)
template <int GS>
__device__ __forceinline__ double scan_any_f64(double X, int lig)
{
    if constexpr (GS == 8) {
        double y;
        y = dpp_f64<DPP_ROW_SHR + 1>(X); X = X + ((lig >= 1) ? y : 0.0);
        y = dpp_f64<DPP_ROW_SHR + 2>(X); X = X + ((lig >= 2) ? y : 0.0);
        y = dpp_f64<DPP_ROW_SHR + 4>(X); X = X + ((lig >= 4) ? y : 0.0);
        return X;
    } else {
        X = X + dpp_f64<DPP_ROW_SHR + 1>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 2>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 4>(X);
        X = X + dpp_f64<DPP_ROW_SHR + 8>(X);
        if constexpr (GS >= 32) {           // carry the totals of the 16-lane rows upwards, row by row
            const double c1 = __shfl(X, 15, GS);
            X = X + ((lig >= 16 && lig < 32) ? c1 : 0.0);
        }
        if constexpr (GS == 64) {
            const double c2 = __shfl(X, 31, GS);
            X = X + ((lig >= 32 && lig < 48) ? c2 : 0.0);
            const double c3 = __shfl(X, 47, GS);
            X = X + ((lig >= 48) ? c3 : 0.0);
        }
        return X;
    }
}

// value of lane J (J < 8, compile time) of the caller's GS-lane group, in every lane of the group
template <int GS, int J>
__device__ __forceinline__ int bcast_lane(int v, int lig)
{
    if constexpr (GS <= 16) {
        constexpr int QP = (J & 3) * 0x55;                                  // quad_perm [j,j,j,j]
        const int q = __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, false);
        int r;
        if constexpr ((J >> 2) == 0) {                                      // source quad is the lower one of its 8
            const int up = __builtin_amdgcn_update_dpp(0, q, DPP_ROW_SHR + 4, 0xF, 0xF, false);
            r = (lig & 4) ? up : q;
        } else {
            const int dn = __builtin_amdgcn_update_dpp(0, q, 0x100 + 4, 0xF, 0xF, false);   // row_shl:4
            r = (lig & 4) ? q : dn;
        }
        if constexpr (GS == 16) {                                           // upper 8 lanes take it from the lower 8
            const int up8 = __builtin_amdgcn_update_dpp(0, r, DPP_ROW_SHR + 8, 0xF, 0xF, false);
            r = (lig & 8) ? up8 : r;
        }
        return r;
    } else {
        return __shfl(v, J, GS);
    }
}

// sum over the GS lanes of a group in every lane (any association order)
template <int GS>
__device__ __forceinline__ double allsum_any_f64(double x, int lane)
{
    x = x + dpp_f64<DPP_XOR1>(x);
    x = x + dpp_f64<DPP_XOR2>(x);
    x = x + dpp_f64<DPP_HALF_MIRROR>(x);
    if constexpr (GS >= 16) x = x + dpp_f64<DPP_ROW_ROR + 8>(x);
    if constexpr (GS >= 32) x = x + xor16_f64(x, lane);
    if constexpr (GS == 64) x = x + xor32_f64(x, lane);
    return x;
}

// OR over the GS lanes of a group in every lane
template <int GS>
__device__ __forceinline__ int allor_i32(int x, int lane)
{
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR1, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_XOR2, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, DPP_HALF_MIRROR, 0xF, 0xF, false);
    if constexpr (GS >= 16) x |= __builtin_amdgcn_update_dpp(0, x, DPP_ROW_ROR + 8, 0xF, 0xF, false);
    if constexpr (GS >= 32) x |= __shfl_xor(x, 16, GS);
    if constexpr (GS == 64) x |= __shfl_xor(x, 32, GS);
    return x;
}

// One site of the sparse kernel (J = index inside the current batch of 8 sites).  Returns false when the
// draw cannot be decided within the margin (the document is then handed to the dense kernel).
template <int GS, int J>
__device__ __forceinline__ bool sparse_site(const KParams &P, int nb, int sv, int sf, int sz, int su_lo, int su_hi,
                                            const int (&xg)[8], bool live, int pos, int A, int &ndk, int &nk, double &y,
                                            int &my_zn, int lig, int lane, int gbase, uint64_t gmask)
{
    if (J >= nb) return true;
    const int f = bcast_lane<GS, J>(sf, lig), zo = bcast_lane<GS, J>(sz, lig);
    const double u = __hiloint2double(bcast_lane<GS, J>(su_hi, lig), bcast_lane<GS, J>(su_lo, lig));
    if (pos == zo) { ndk -= f; nk -= f; y = rcp_newton((double)nk + P.vbeta); }     // LabeledLDA.py:109-111
    const int x = xg[J] - ((pos == zo) ? f : 0);
    const double w = live ? ((double)ndk + P.alpha) * (((double)x + P.beta) * y) : 0.0;
    const double Q = scan_any_f64<GS>(w, lig);
    const double tot = allsum_any_f64<GS>(w, lane);
    const double t = u * tot, margin = tot * P.margin_rel;
    const bool unsure = (live && !(fabs(Q - t) > margin)) || !(tot > 0.0) || !(margin < tot);
    if (((__ballot(unsure) >> gbase) & gmask) != 0) {
        if (pos == zo) { ndk += f; nk += f; }               // undo: the dense kernel starts at this site
        return false;
    }
    const uint64_t gf = (__ballot(live && Q > t) >> gbase) & gmask;
    const int sel = gf ? (int)__ffsll((unsigned long long)gf) - 1 : A - 1;          // none: last allowed topic
    const int zn = allor_i32<GS>((lig == sel) ? pos : 0, lane);
    if (pos == zn) { ndk += f; nk += f; y = rcp_newton((double)nk + P.vbeta); }     // LabeledLDA.py:121-125
    if (lig == J) my_zn = zn;
    (void)sv;
    return true;
}

template <int GS>
__global__ void __launch_bounds__(256) llda_sweep_sparse_kernel(const KParams P)
{
    constexpr int GPB = 256 / GS;
    __shared__ int s_nk[LLDA_MAX_K];          // workgroup accumulator of the n_k changes
    const int tid = threadIdx.x;
    const int KP = P.KP;
    for (int i = tid; i < KP; i += 256) s_nk[i] = 0;
    __syncthreads();
    const int lane = tid & 63;
    const int lig = tid & (GS - 1);
    const int grp = tid / GS;
    const int gbase = lane & ~(GS - 1);
    const uint64_t gmask = (GS == 64) ? ~0ull : ((1ull << GS) - 1ull);

    for (int it = 0; it < P.dpg; ++it) {
        const int64_t idx = ((int64_t)blockIdx.x * P.dpg + it) * GPB + grp;
        if (idx >= P.D) break;
        const int64_t d = P.doc_order ? (int64_t)P.doc_order[idx] : idx;
        const int64_t s0 = P.doc_off[d];
        const int len = (int)(P.doc_off[d + 1] - s0);
        if (len <= 0) continue;
        const int64_t l0 = P.live_off[d];
        const int A = (int)(P.live_off[d + 1] - l0);
        const bool live = lig < A;
        const int pos = live ? P.live_pos[l0 + lig] : -1;
        int32_t *ndk_p = P.n_dk + d * KP + (live ? pos : 0);
        int ndk = live ? *ndk_p : 0;
        const int ndk0 = ndk;
        int nk = live ? P.n_k[pos] : 0;
        double y = rcp_newton((double)nk + P.vbeta);
        const uint32_t gdoc = (uint32_t)(d + P.doc_base);
        int stop_at = -1;

        // Sites are processed in batches of 8 so that memory latency is paid once per batch: lane j < 8 of
        // the group loads the scalars of site n0+j and draws its uniform, every lane gathers its own topic's
        // n_kw entry for all 8 words, then 8 sites run back to back on registers / DPP only, and lane j
        // commits site n0+j (z store + two atomics) while the next batch loads.
        for (int n0 = 0; n0 < len && stop_at < 0; n0 += 8) {
            const int nb = min(8, len - n0);
            const int jj = lig & 7;
            const int64_t si = s0 + n0 + (jj < nb ? jj : nb - 1);
            const int sv = P.word[si], sf = P.freq[si], sz = P.z[si], sc = P.csc_pos ? P.csc_pos[si] : 0;
            int su_lo, su_hi;
            {   // keyed uniform of site n0+jj: Philox block (site >> 1), words (0,1) / (2,3) by parity
                const int n = n0 + jj;
                uint32_t c0 = (uint32_t)(n >> 1), c1 = gdoc, c2 = P.stream_id, c3 = P.sweep;
                philox4x32_10(c0, c1, c2, c3, P.key0, P.key1);
                const uint32_t ra = (n & 1) ? c2 : c0, rb = (n & 1) ? c3 : c1;
                const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
                su_lo = __double2loint(u); su_hi = __double2hiint(u);
            }
            // (the broadcasts must run in ALL lanes: a DPP read from a lane that is masked off returns 0)
            const int w0 = bcast_lane<GS, 0>(sv, lig), w1 = bcast_lane<GS, 1>(sv, lig), w2 = bcast_lane<GS, 2>(sv, lig),
                      w3 = bcast_lane<GS, 3>(sv, lig), w4 = bcast_lane<GS, 4>(sv, lig), w5 = bcast_lane<GS, 5>(sv, lig),
                      w6 = bcast_lane<GS, 6>(sv, lig), w7 = bcast_lane<GS, 7>(sv, lig);
            int xg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (live) {
                const int32_t *col = P.n_kw + pos;
                xg[0] = col[(int64_t)w0 * KP]; xg[1] = col[(int64_t)w1 * KP]; xg[2] = col[(int64_t)w2 * KP];
                xg[3] = col[(int64_t)w3 * KP]; xg[4] = col[(int64_t)w4 * KP]; xg[5] = col[(int64_t)w5 * KP];
                xg[6] = col[(int64_t)w6 * KP]; xg[7] = col[(int64_t)w7 * KP];
            }

            int my_zn = sz;
            bool ok = true;
            int done = 0;                     // sites of this batch that were decided
#define LLDA_SPARSE_SITE(J)                                                                                    \
            if (ok) {                                                                                          \
                ok = sparse_site<GS, J>(P, nb, sv, sf, sz, su_lo, su_hi, xg, live, pos, A, ndk, nk, y, my_zn, \
                                        lig, lane, gbase, gmask);                                              \
                if (ok && J < nb) done = J + 1;                                                                \
            }
            LLDA_SPARSE_SITE(0) LLDA_SPARSE_SITE(1) LLDA_SPARSE_SITE(2) LLDA_SPARSE_SITE(3)
            LLDA_SPARSE_SITE(4) LLDA_SPARSE_SITE(5) LLDA_SPARSE_SITE(6) LLDA_SPARSE_SITE(7)
#undef LLDA_SPARSE_SITE
            if (!ok) stop_at = n0 + done;
            // commit the decided sites of the batch: lane j handles site n0+j
            if (lig < 8 && lig < done) commit_site(P, s0 + n0 + lig, sv, sf, sz, my_zn, sc, KP);
        }

        if (stop_at >= 0) {
            // hand the document over to the dense kernel: record (doc, site, n_dk deltas so far)
            int slot = 0;
            if (lig == 0) slot = atomicAdd(P.resume_count, 1);
            slot = __shfl(slot, 0, GS);
            if (slot < P.resume_cap) {
                int32_t *rec = P.resume + (int64_t)slot * (2 + LLDA_MAX_LIVE);
                if (lig == 0) { rec[0] = (int32_t)d; rec[1] = stop_at; }
                if (live) rec[2 + lig] = ndk - ndk0;
            } else if (lig == 0 && P.status) {
                atomicOr(P.status, 4);                  // resume list overflow (cannot happen with production margins)
            }
        }
        if (live) {
            *ndk_p = ndk;
            if (ndk != ndk0) atomicAdd(&s_nk[pos], ndk - ndk0);
        }
    }
    __syncthreads();
    for (int i = tid; i < KP; i += 256) {
        const int dl = s_nk[i];
        if (dl) atomicAdd(P.n_k_delta + i, dl);
    }
}

// ---------------------------------------------------------------------------------------------
// log-likelihood read-out (LabeledLDA.py:231-239, 256-265), same group layout
// ---------------------------------------------------------------------------------------------
struct LParams {
    const int64_t *doc_off;
    const int32_t *word;
    const uint16_t *lab_mask;
    const int32_t *n_dk;
    const int32_t *n_kw;
    const int32_t *n_k;
    double *out_doc;
    int64_t D;
    double alpha, beta, vbeta;
};

template <int G>
__device__ __forceinline__ double group_allsum(double x)
{
#pragma unroll
    for (int d = 1; d < G; d <<= 1) x = x + __shfl_xor(x, d, G);
    return x;
}

template <int G, int T>
__global__ void __launch_bounds__(256) llda_loglik_kernel(const LParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lig = tid & (G - 1);
    const int64_t d = (int64_t)blockIdx.x * GPB + tid / G;
    if (d >= P.D) return;
    int ndk[T], nk[T];
    load_row<T>(P.n_dk + d * KP + lig * T, ndk);
    load_row<T>(P.n_k + lig * T, nk);
    const uint32_t mask = P.lab_mask[d * G + lig];
    double th[T], rden[T], rs = 0.0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
        th[s] = (double)ndk[s] + (((mask >> s) & 1u) ? P.alpha : 0.0);     // n_d_k + labs*alpha
        rs = rs + th[s];
        rden[s] = (double)nk[s] + P.vbeta;
    }
    rs = group_allsum<G>(rs);
#pragma unroll
    for (int s = 0; s < T; ++s) th[s] = th[s] / rs;
    double acc = 0.0;
    for (int64_t i = P.doc_off[d]; i < P.doc_off[d + 1]; ++i) {
        int x[T];
        load_row<T>(P.n_kw + (int64_t)P.word[i] * KP + lig * T, x);
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < T; ++s) dot = dot + th[s] * (((double)x[s] + P.beta) / rden[s]);
        dot = group_allsum<G>(dot);
        acc = acc - log(dot);
    }
    if (lig == 0) P.out_doc[d] = acc;
}


// ---------------------------------------------------------------------------------------------
// Test-time fold-in sampler: LabeledLDA.prep4test / run_test (LabeledLDA.py:155-212).
// One lane group per held-out document; topic-word loadings ph_hat are fixed, only the document's n_dk
// moves.  phn = ph_hat with every word column normalised (prep4test, LabeledLDA.py:162-167, done by
// the host); both matrices are word-major in device order: (V, KP) doubles.
// ---------------------------------------------------------------------------------------------
struct FParams {
    const int64_t *doc_off;
    const int32_t *word;
    const int32_t *init_idx; // row of `phn` holding the initial probabilities of every site
    const int32_t *freq;
    int32_t *z;              // [S] out: final assignments (device positions)
    const double *ph;        // [V*KP]
    const double *phn;       // [V*KP]
    int32_t *n_dk;           // [D*KP] out: final counts
    double *th;              // [D*KP] out: thinned average of n_dk / sum(n_dk)
    const uint8_t *slot_valid; // [KP] 1 for slots that hold a topic (0 in the padding)
    int32_t *status;
    int64_t D;
    int64_t doc_base;
    double alpha, beta;
    double c_init, c_loop;   // the reference's "while prob.sum() > 1: prob /= c" constants
    uint32_t key0, key1, stream_id;
    int32_t iters, thinning;
    int32_t beta_fallback;   // CascadeLDA.cascade_test: prob.sum() == 0 -> prob = num_a * (b + beta)
    int32_t avg_mode;        // 0: (s-1)/s*avg + (1/s)*cur   1: m*avg + (1-m)*cur with m = (s-1)/s
    int32_t last_leaf, tail, tail_row, n_rounds, xor_tree;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];
};

template <int T>
__device__ __forceinline__ void load_row_f64(const double *__restrict__ p, double (&x)[T])
{
    if constexpr (T % 2 == 0) {
#pragma unroll
        for (int i = 0; i < T / 2; ++i) {
            const double2 v = reinterpret_cast<const double2 *>(p)[i];
            x[2 * i] = v.x; x[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) x[i] = p[i];
    }
}

// `while prob.sum() > 1: prob /= c`  (LabeledLDA.py:170-171, 192-193); c_rcp = RN(1/c)
template <int G, int T, bool HAS_TAIL>
__device__ __forceinline__ void shrink_to_one(double (&p)[T], double c, double c_rcp, const KParams &K, int lig, int lane)
{
    for (int guard = 0; guard < (1 << 28); ++guard) {   // the reference loops until the sum is <= 1
        const double s = group_sum<G, T, HAS_TAIL>(p, K, lig, lane);
        if (!(s > 1.0)) break;                     // group-uniform: every lane holds the same s
#pragma unroll
        for (int k = 0; k < T; ++k) p[k] = div_by(p[k], c, c_rcp);
    }
}

template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_foldin_kernel(const FParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lig = tid & (G - 1);
    const int64_t d = (int64_t)blockIdx.x * GPB + tid / G;
    if (d >= P.D) return;
    KParams K;                                     // the summation schedule group_sum() reads
    K.last_leaf = P.last_leaf; K.tail = P.tail; K.tail_row = P.tail_row; K.n_rounds = P.n_rounds;
    K.xor_tree = P.xor_tree;
#pragma unroll
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) K.rounds_pk[r] = P.rounds_pk[r];

    const int64_t s0 = P.doc_off[d];
    const int len = (int)(P.doc_off[d + 1] - s0);
    const uint32_t gdoc = (uint32_t)(d + P.doc_base);
    int ndk[T];
    double avg[T];
#pragma unroll
    for (int s = 0; s < T; ++s) { ndk[s] = 0; avg[s] = 0.0; }
    int ntot = 0;
    const double c0 = P.c_init, c0r = 1.0 / c0, c1 = P.c_loop, c1r = 1.0 / c1;

    // sweep = -1: prep4test (initial assignments from the normalised loadings), then `iters` sweeps
    for (int sweep = -1; sweep < P.iters; ++sweep) {
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int n = 0; n < len; ++n) {
            const int v = P.word[s0 + n], f = P.freq[s0 + n];
            if ((n & (2 * G - 1)) == 0) {
                r0 = (uint32_t)(n >> 1) + (uint32_t)lig; r1 = gdoc; r2 = P.stream_id; r3 = (uint32_t)sweep;
                philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
            }
            const int holder = (n >> 1) & (G - 1);
            const uint32_t ra = (uint32_t)__shfl((int)((n & 1) ? r2 : r0), holder, G);
            const uint32_t rb = (uint32_t)__shfl((int)((n & 1) ? r3 : r1), holder, G);
            const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);

            double w[T];
            int zo = -1;
            if (sweep < 0) {
                load_row_f64<T>(P.phn + (int64_t)P.init_idx[s0 + n] * KP + lig * T, w);
                shrink_to_one<G, T, HAS_TAIL>(w, c0, c0r, K, lig, lane);
                ntot += f;
            } else {
                zo = P.z[s0 + n];
                {
                    const int lo = zo / T, so = zo - lo * T;
                    onehot_add1<T>(ndk, (lig == lo) ? (1u << so) : 0u, f);       // n_dk[z] -= f
                }
                double b[T];
                load_row_f64<T>(P.ph + (int64_t)v * KP + lig * T, b);
#pragma unroll
                for (int s = 0; s < T; ++s) w[s] = ((double)ndk[s] + P.alpha) * b[s];   // num_a * b
                double S = group_sum<G, T, HAS_TAIL>(w, K, lig, lane);
                if (P.beta_fallback && S == 0.0) {     // 0/0 raises in the reference (CascadeLDA.py:225-230)
#pragma unroll
                    for (int s = 0; s < T; ++s) {
                        const bool real = P.slot_valid[lig * T + s] != 0;
                        w[s] = real ? ((double)ndk[s] + P.alpha) * (b[s] + P.beta) : 0.0;
                    }
                    S = group_sum<G, T, HAS_TAIL>(w, K, lig, lane);
                }
                const double y = 1.0 / S;
#pragma unroll
                for (int s = 0; s < T; ++s) w[s] = div_by(w[s], S, y);                  // prob /= prob.sum()
                shrink_to_one<G, T, HAS_TAIL>(w, c1, c1r, K, lig, lane);
            }
            int zn = draw_position<G, T, false>(w, u, 0u, true, lig, lane);
            if (zn < 0) {                         // all-zero / NaN probabilities: the reference would raise
                zn = zo < 0 ? 0 : zo;
                if (lig == 0 && P.status) atomicOr(P.status, 1);
            }
            {
                const int ln = zn / T, sn = zn - ln * T;
                onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);          // n_dk[new_z] += f
            }
            if (lig == 0) P.z[s0 + n] = zn;
        }
        // thinned running average of the document-topic state (LabeledLDA.py:199-211)
        if (sweep >= 0 && (sweep + 1) % P.thinning == 0) {
            const int s2 = (sweep + 1) / P.thinning;
            const double tot = (double)ntot;
            if (s2 == 1) {
#pragma unroll
                for (int s = 0; s < T; ++s) avg[s] = (double)ndk[s] / tot;
            } else if (P.avg_mode == 0) {          // LabeledLDA.py:204-209, CascadeLDA.py:240-246
                const double f_old = (double)(s2 - 1) / (double)s2, f_new = 1.0 / (double)s2;
#pragma unroll
                for (int s = 0; s < T; ++s) {
                    const double old_part = f_old * avg[s];
                    const double new_part = f_new * ((double)ndk[s] / tot);
                    avg[s] = old_part + new_part;
                }
            } else {                               // CascadeLDA.run_test, CascadeLDA.py:337-341
                const double m = (double)(s2 - 1) / (double)s2, m1 = 1.0 - m;
#pragma unroll
                for (int s = 0; s < T; ++s) {
                    const double old_part = m * avg[s];
                    const double new_part = m1 * ((double)ndk[s] / tot);
                    avg[s] = old_part + new_part;
                }
            }
        }
    }
    store_row<T>(P.n_dk + d * KP + lig * T, ndk);
#pragma unroll
    for (int s = 0; s < T; ++s) P.th[d * KP + lig * T + s] = avg[s];
}

// ---------------------------------------------------------------------------------------------
// Fold of the commit log into word-major counts (llda_commit_log, include/llda_gibbs.h): one wavefront per
// item (a run of log entries of one word), a KP-entry histogram per wavefront in LDS.  The histogram is
// flushed either by walking the item's entries again (short items: each touched topic is claimed with an LDS
// exchange) or by scanning all KP entries (long items).
// ---------------------------------------------------------------------------------------------
struct CParams {
    const int64_t *item_begin;
    const int32_t *item_len, *item_word;
    int64_t n_items;
    const uint32_t *log;
    const int32_t *freq;
    int32_t *target, *n_k, *n_k_delta;
    int32_t KP;
};

__device__ __forceinline__ void add_count(int32_t *p, int a, bool shared_row)
{
    if (shared_row) atomicAdd(p, a);
    else *p += a;
}

__global__ void __launch_bounds__(256) llda_commit_log_kernel(const CParams P)
{
    extern __shared__ int s_hist[];               // [4][KP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int KP = P.KP;
    if (blockIdx.x == 0 && P.n_k)
        for (int p = tid; p < KP; p += 256) {
            P.n_k[p] += P.n_k_delta[p];
            P.n_k_delta[p] = 0;
        }
    int *hist = s_hist + w * KP;
    for (int p = lane; p < KP; p += 64) hist[p] = 0;
    const int64_t item = (int64_t)blockIdx.x * 4 + w;
    if (item >= P.n_items) return;
    const int64_t b = P.item_begin[item];
    const int len = P.item_len[item];
    const int wv = P.item_word[item];
    const bool shared_row = wv < 0;
    int32_t *row = P.target + (int64_t)(wv & 0x7fffffff) * KP;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (int j = lane; j < len; j += 64) {
        const uint32_t e = P.log[b + j];
        const int zo = (int)(e & 0xFFFFu), zn = (int)(e >> 16);
        if (zo != zn) {
            const int f = P.freq[b + j];
            atomicAdd(&hist[zo], -f);
            atomicAdd(&hist[zn], f);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (len <= KP) {
        for (int j = lane; j < len; j += 64) {
            const uint32_t e = P.log[b + j];
            const int zo = (int)(e & 0xFFFFu), zn = (int)(e >> 16);
            if (zo != zn) {
                const int a = atomicExch(&hist[zo], 0), c = atomicExch(&hist[zn], 0);
                if (a) add_count(row + zo, a, shared_row);
                if (c) add_count(row + zn, c, shared_row);
            }
        }
    } else {
        for (int p = lane; p < KP; p += 64) {
            const int a = hist[p];
            if (a) add_count(row + p, a, shared_row);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Thinning read-outs (LabeledLDA.py:131-153, 231-239; CascadeLDA.py:394-395, 423-434): phi / theta of
// the current counts and their running means, written in the reference's (K, V) / (D, K) layout.
// ---------------------------------------------------------------------------------------------
struct RParams {
    const int32_t *n_kw, *n_k, *n_dk;
    const double *den;
    const uint16_t *lab_mask;
    double *out;
    int32_t *flags;
    int64_t V, D;
    int32_t K, KP, T, mode;
    double alpha, beta, vbeta, keep, share;
    int32_t leaf_start[LLDA_MAX_LEAVES], leaf_len[LLDA_MAX_LEAVES];
    int32_t last_leaf, tail, tail_row, n_rounds, xor_tree;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];
};

// topic held by a device position, -1 for padding (inverse of llda_layout.topic_pos)
__device__ __forceinline__ int topic_of_position(const RParams &P, int pos)
{
    const int g = pos / P.T, slot = pos - g * P.T;
    const int leaf = g >> 3, rel = (g & 7) + 8 * slot;
    return rel < P.leaf_len[leaf] ? P.leaf_start[leaf] + rel : -1;
}

__device__ __forceinline__ double running_mean(const RParams &P, double old, double cur)
{
    if (P.mode == 0) return cur;
    const double a = P.keep * old, b = P.share * cur;      // two roundings, then the sum (no FMA)
    return a + b;
}

// One workgroup per 64 words: 64 x 64 (word, position) tiles of n_kw go through LDS so that both the
// word-major reads and the topic-major writes are contiguous.
__global__ void __launch_bounds__(256) llda_readout_phi_kernel(const RParams P)
{
    __shared__ int s_tile[64][65];
    __shared__ int s_seen[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    if (tid < 64) s_seen[tid] = 0;
    int bad = 0, seen = 0;
    const int64_t v = v0 + lane;
    for (int c0 = 0; c0 < P.KP; c0 += 64) {
        __syncthreads();
        for (int r = w; r < 64; r += 4)
            if (v0 + r < P.V && c0 + lane < P.KP) s_tile[r][lane] = P.n_kw[(v0 + r) * P.KP + c0 + lane];
        __syncthreads();
        for (int j = w; j < 64 && c0 + j < P.KP; j += 4) {
            const int k = topic_of_position(P, c0 + j);
            if (k < 0 || v >= P.V) continue;
            const double den = P.den ? P.den[c0 + j] : (double)P.n_k[c0 + j] + P.vbeta;
            const double cur = ((double)s_tile[lane][j] + P.beta) / den;
            double *o = P.out + (int64_t)k * P.V + v;
            const double val = running_mean(P, P.mode ? *o : 0.0, cur);
            *o = val;
            if (val < 0.0) bad |= LLDA_READOUT_NEGATIVE;
            if (val != val) bad |= LLDA_READOUT_NAN;
            if (val != 0.0) seen = 1;
        }
    }
    if (seen) atomicOr(&s_seen[lane], 1);
    __syncthreads();
    if (tid < 64 && v0 + tid < P.V && !s_seen[tid]) bad |= LLDA_READOUT_NO_LOAD;
    if (bad && P.flags) atomicOr(P.flags, bad);
}

template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_readout_theta_kernel(const RParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lig = tid & (G - 1);
    const int64_t d = (int64_t)blockIdx.x * GPB + tid / G;
    if (d >= P.D) return;
    KParams K;                                     // the summation schedule group_sum() reads
    K.last_leaf = P.last_leaf; K.tail = P.tail; K.tail_row = P.tail_row; K.n_rounds = P.n_rounds;
    K.xor_tree = P.xor_tree;
#pragma unroll
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) K.rounds_pk[r] = P.rounds_pk[r];
    int ndk[T];
    load_row<T>(P.n_dk + d * KP + lig * T, ndk);
    const uint32_t mask = P.lab_mask[d * G + lig];
    double num[T];
#pragma unroll
    for (int s = 0; s < T; ++s) num[s] = (double)ndk[s] + (((mask >> s) & 1u) ? P.alpha : 0.0);   // n_d_k + labs*alpha
    const double rs = group_sum<G, T, HAS_TAIL>(num, K, lig, lane);                              // np.sum, axis 1
#pragma unroll
    for (int s = 0; s < T; ++s) {
        const int k = topic_of_position(P, lig * T + s);
        if (k < 0) continue;
        double *o = P.out + d * P.K + k;
        *o = running_mean(P, P.mode ? *o : 0.0, num[s] / rs);
    }
}

// ---------------------------------------------------------------------------------------------
// self test: div_by (reciprocal + two corrections) against the hardware IEEE division
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) llda_selftest_div_kernel(uint64_t seed, int iters, unsigned long long *bad)
{
    uint32_t mism = 0;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        uint32_t c0 = tid, c1 = (uint32_t)i, c2 = 0x5e1f7e57u, c3 = 0;
        philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
        // b: a sum-like positive double with a random 52-bit significand and exponent in [-20, 40];
        // a: anything from 0 to 2^60 times smaller than b up to a few times b
        const uint64_t mb = ((uint64_t)(c0 & 0xFFFFFu) << 32) | c1;
        const uint64_t ma = ((uint64_t)(c2 & 0xFFFFFu) << 32) | c3;
        const int eb = (int)((c0 >> 20) % 61) - 20;
        const int ea = eb + 2 - (int)((c2 >> 20) % 64);
        double b = __longlong_as_double((long long)(((uint64_t)(1023 + eb) << 52) | mb));
        double a = __longlong_as_double((long long)(((uint64_t)(1023 + ea) << 52) | ma));
        if ((i & 7) == 7) {            // integer-valued operands, the shape of the count terms
            b = (double)(c0 >> 4) + 1000.0 * 1.0000000000000002;
            a = (double)(c2 >> 12) * 0.1;
        }
        if ((i & 63) == 63) a = 0.0;
        const double y = 1.0 / b;
        if (div_by(a, b, y) != a / b) ++mism;
    }
    if (mism) atomicAdd(bad, (unsigned long long)mism);
}

// ---------------------------------------------------------------------------------------------
// helpers: fold deltas, build counts from assignments
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) llda_apply_delta_kernel(int32_t *__restrict__ counts,
                                                               int32_t *__restrict__ delta, int64_t n4,
                                                               int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int4 *c4 = reinterpret_cast<int4 *>(counts);
    int4 *d4 = reinterpret_cast<int4 *>(delta);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        int4 c = c4[i];
        const int4 d = d4[i];
        if (d.x | d.y | d.z | d.w) {
            c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w;
            c4[i] = c;
            d4[i] = make_int4(0, 0, 0, 0);
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        counts[i] += delta[i];
        delta[i] = 0;
    }
}

__global__ void __launch_bounds__(256) llda_count_init_kernel(const int64_t *__restrict__ doc_off,
                                                              const int32_t *__restrict__ word,
                                                              const int32_t *__restrict__ freq,
                                                              const int32_t *__restrict__ z, int64_t D, int KP,
                                                              int32_t *n_dk, int32_t *n_kw, int32_t *n_k)
{
    // one wavefront per document at a time, lanes stride over its sites.  The document's n_dk row and the
    // workgroup's share of n_k are histograms in LDS (the row is then written with plain stores, n_k with KP
    // atomics per workgroup); only n_kw takes one global atomic per site.
    extern __shared__ int s_init[];               // [KP] n_k of the workgroup, then [4][KP] per-wavefront rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int *s_nk = s_init, *hist = s_init + (1 + w) * KP;
    for (int p = tid; p < KP; p += 256) s_nk[p] = 0;
    for (int p = lane; p < KP; p += 64) hist[p] = 0;
    __syncthreads();
    for (int64_t d = (int64_t)blockIdx.x * 4 + w; d < D; d += (int64_t)gridDim.x * 4) {
        for (int64_t i = doc_off[d] + lane; i < doc_off[d + 1]; i += 64) {
            const int f = freq[i], p = z[i];
            atomicAdd(&hist[p], f);
            atomicAdd(n_kw + (int64_t)word[i] * KP + p, f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (int p = lane; p < KP; p += 64) {
            const int h = hist[p];
            if (h) {
                hist[p] = 0;
                n_dk[d * KP + p] += h;
                atomicAdd(&s_nk[p], h);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    __syncthreads();
    for (int p = tid; p < KP; p += 256) {
        const int h = s_nk[p];
        if (h) atomicAdd(n_k + p, h);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void add_leaves(llda_layout *L, int n, int start)
{
    if (n <= 128) {
        if (L->n_leaves < LLDA_MAX_LEAVES) {
            L->leaf_start[L->n_leaves] = start;
            L->leaf_len[L->n_leaves] = n;
        }
        L->n_leaves++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    add_leaves(L, n2, start);
    add_leaves(L, n - n2, start + n2);
}

// numpy's recursion over leaf ranges [first, first+count): returns depth, fills the schedule
int schedule(llda_layout *L, int n, int *next_leaf, int *first_out, int *count_out)
{
    if (n <= 128) {
        *first_out = (*next_leaf)++;
        *count_out = 1;
        return 0;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int lf, lc, rf, rc;
    const int dl = schedule(L, n2, next_leaf, &lf, &lc);
    const int dr = schedule(L, n - n2, next_leaf, &rf, &rc);
    const int d = (dl > dr ? dl : dr);      // this node combines in round d
    if (d < LLDA_MAX_ROUNDS) {
        for (int a = lf; a < lf + lc; ++a) L->rounds[d][a] = rf;
        for (int b = rf; b < rf + rc; ++b) L->rounds[d][b] = lf;
    }
    if (d + 1 > L->n_rounds) L->n_rounds = d + 1;
    *first_out = lf;
    *count_out = lc + rc;
    return d + 1;
}

int hip_fail(hipError_t e)
{
    g_last_hip_error = (int)e;
    return LLDA_E_HIP;
}

template <int G, int T>
int launch_sweep(const KParams &P, bool has_tail, bool fast, bool dense, int64_t blocks, hipStream_t st)
{
    const dim3 grid((unsigned)blocks), block(256);
    if (!fast) {
        if (has_tail) hipLaunchKernelGGL((llda_sweep_exact_kernel<G, T, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((llda_sweep_exact_kernel<G, T, false>), grid, block, 0, st, P);
    } else if (dense) {
        hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, true>), grid, block, 0, st, P);
    } else if (has_tail) {
        hipLaunchKernelGGL((llda_sweep_kernel<G, T, true, false>), grid, block, 0, st, P);
    } else {
        hipLaunchKernelGGL((llda_sweep_kernel<G, T, false, false>), grid, block, 0, st, P);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_sweep_T(int T, const KParams &P, bool has_tail, bool fast, bool dense, int64_t blocks, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_sweep<8, 1>(P, has_tail, fast, dense, blocks, st);
        case 2: return launch_sweep<8, 2>(P, has_tail, fast, dense, blocks, st);
        case 4: return launch_sweep<8, 4>(P, has_tail, fast, dense, blocks, st);
        case 8: return launch_sweep<8, 8>(P, has_tail, fast, dense, blocks, st);
        }
    }
    switch (T) {
    case 12: return launch_sweep<G, 12>(P, has_tail, fast, dense, blocks, st);
    case 16: return launch_sweep<G, 16>(P, has_tail, fast, dense, blocks, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_loglik(const LParams &P, hipStream_t st)
{
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    hipLaunchKernelGGL((llda_loglik_kernel<G, T>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_loglik_T(int T, const LParams &P, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_loglik<8, 1>(P, st);
        case 2: return launch_loglik<8, 2>(P, st);
        case 4: return launch_loglik<8, 4>(P, st);
        case 8: return launch_loglik<8, 8>(P, st);
        }
    }
    switch (T) {
    case 12: return launch_loglik<G, 12>(P, st);
    case 16: return launch_loglik<G, 16>(P, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_foldin(const FParams &P, bool has_tail, hipStream_t st)
{
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    if (has_tail) hipLaunchKernelGGL((llda_foldin_kernel<G, T, true>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((llda_foldin_kernel<G, T, false>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_foldin_T(int T, const FParams &P, bool has_tail, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_foldin<8, 1>(P, has_tail, st);
        case 2: return launch_foldin<8, 2>(P, has_tail, st);
        case 4: return launch_foldin<8, 4>(P, has_tail, st);
        case 8: return launch_foldin<8, 8>(P, has_tail, st);
        }
    }
    switch (T) {
    case 12: return launch_foldin<G, 12>(P, has_tail, st);
    case 16: return launch_foldin<G, 16>(P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

template <int G, int T>
int launch_theta(const RParams &P, bool has_tail, hipStream_t st)
{
    const int64_t blocks = (P.D + (256 / G) - 1) / (256 / G);
    if (has_tail) hipLaunchKernelGGL((llda_readout_theta_kernel<G, T, true>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((llda_readout_theta_kernel<G, T, false>), dim3((unsigned)blocks), dim3(256), 0, st, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

template <int G>
int dispatch_theta_T(int T, const RParams &P, bool has_tail, hipStream_t st)
{
    if constexpr (G == 8) {
        switch (T) {
        case 1: return launch_theta<8, 1>(P, has_tail, st);
        case 2: return launch_theta<8, 2>(P, has_tail, st);
        case 4: return launch_theta<8, 4>(P, has_tail, st);
        case 8: return launch_theta<8, 8>(P, has_tail, st);
        }
    }
    switch (T) {
    case 12: return launch_theta<G, 12>(P, has_tail, st);
    case 16: return launch_theta<G, 16>(P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

// the summation schedule shared by llda_sweep and llda_foldin
void fill_schedule(const llda_layout &L, int32_t &last_leaf, int32_t &tail, int32_t &tail_row, int32_t &n_rounds,
                   int32_t &xor_tree, uint32_t (&rounds_pk)[LLDA_MAX_ROUNDS])
{
    last_leaf = L.n_leaves - 1; tail = L.tail; tail_row = L.tail_row; n_rounds = L.n_rounds;
    const int P2 = L.G / 8;
    int xt = (L.n_leaves == P2) ? 1 : 0;
    for (int r = 0; (1 << r) < P2 && xt; ++r)
        for (int p = 0; p < P2; ++p)
            if (L.rounds[r][p] != (p ^ (1 << r))) xt = 0;
    xor_tree = xt;
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) {
        uint32_t pk = 0;
        for (int p = 0; p < LLDA_MAX_LEAVES; ++p) pk |= (uint32_t)L.rounds[r][p] << (4 * p);
        rounds_pk[r] = pk;
    }
}

}  // namespace

extern "C" {

int llda_abi_version(void) { return LLDA_ABI_VERSION; }

int llda_last_hip_error(void) { return g_last_hip_error; }

const char *llda_strerror(int code)
{
    switch (code) {
    case LLDA_OK: return "ok";
    case LLDA_E_BAD_K: return "K outside 1..1024 or more than 8 pairwise leaves";
    case LLDA_E_BAD_ARG: return "bad argument";
    case LLDA_E_HIP: return "HIP runtime error";
    case LLDA_E_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
    }
}

int llda_layout_init(int32_t K, llda_layout *L)
{
    if (!L) return LLDA_E_BAD_ARG;
    if (K < 1 || K > LLDA_MAX_K) return LLDA_E_BAD_K;
    memset(L, 0, sizeof *L);
    L->K = K;
    add_leaves(L, K, 0);
    if (L->n_leaves > LLDA_MAX_LEAVES) return LLDA_E_BAD_K;
    int P = 1;
    while (P < L->n_leaves) P *= 2;
    L->G = 8 * P;
    int t = 0;
    for (int p = 0; p < L->n_leaves; ++p) {
        const int r = (L->leaf_len[p] + 7) / 8;
        if (r > t) t = r;
    }
    if (t > 2) t = (t + 3) / 4 * 4;
    L->T = t;
    L->KP = L->G * t;
    L->tail = L->leaf_len[L->n_leaves - 1] % 8;
    L->tail_row = L->leaf_len[L->n_leaves - 1] / 8;
    for (int i = 0; i < LLDA_MAX_K; ++i) L->pos_topic[i] = -1;
    for (int p = 0; p < L->n_leaves; ++p)
        for (int rel = 0; rel < L->leaf_len[p]; ++rel) {
            const int pos = (8 * p + (rel & 7)) * t + (rel >> 3);
            L->topic_pos[L->leaf_start[p] + rel] = pos;
            L->pos_topic[pos] = L->leaf_start[p] + rel;
        }
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r)
        for (int p = 0; p < LLDA_MAX_LEAVES; ++p) L->rounds[r][p] = p;
    int next = 0, first, count;
    schedule(L, K, &next, &first, &count);
    if (L->n_rounds > LLDA_MAX_ROUNDS) return LLDA_E_BAD_K;
    return LLDA_OK;
}

int llda_sweep(const llda_sweep_args *a, void *stream)
{
    if (!a || a->D < 0 || a->V < 1) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(a->K, &L);
    if (rc) return rc;
    if (a->D == 0) return LLDA_OK;            // an empty shard: nothing to do, array pointers may be NULL
    const bool logged = a->csc_pos && a->commit_log;
    if ((a->csc_pos != nullptr) != (a->commit_log != nullptr)) return LLDA_E_BAD_ARG;
    if (!a->doc_off || !a->word || !a->freq || !a->z || !a->lab_mask || !a->n_dk || !a->n_kw ||
        (!a->n_kw_delta && !logged) || !a->n_k || !a->n_k_delta)
        return LLDA_E_BAD_ARG;

    KParams P;
    memset(&P, 0, sizeof P);
    P.doc_off = a->doc_off; P.doc_order = a->doc_order; P.word = a->word; P.freq = a->freq; P.z = a->z;
    P.lab_mask = a->lab_mask; P.n_dk = a->n_dk; P.n_kw = a->n_kw; P.n_kw_delta = a->n_kw_delta;
    P.n_k = a->n_k; P.n_k_delta = a->n_k_delta; P.status = a->status;
    P.csc_pos = a->csc_pos; P.commit_log = a->commit_log;
    P.D = a->D; P.doc_base = a->doc_base;
    P.alpha = a->alpha; P.beta = a->beta;
    P.vbeta = (double)a->V * a->beta;                       // V * beta evaluated first (LabeledLDA.py:115)
    P.key0 = (uint32_t)a->seed; P.key1 = (uint32_t)(a->seed >> 32);
    P.sweep = a->sweep; P.stream_id = a->stream_id;
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
    P.KP = L.KP;
    const bool has_tail = L.tail != 0;
    // with alpha, beta >= 1e-6 and int32 counts no label-allowed score can underflow to zero ...
    // ... and with V*beta < 2^40 the fp32 / fp64 reciprocals of n_k + V*beta stay in range
    const bool fast = a->alpha >= 1e-6 && a->beta >= 1e-6 && P.vbeta < 1099511627776.0;
    // all-ones label masks and no padded slots: the mask need not be applied at all
    const bool dense = fast && a->dense_mask != 0 && L.K == L.KP;
    // debug_margin: 0 = production margins; n > 0 = 2^-n (wider: more fallbacks); -1 = always exact tier;
    // -2 = no fp32 tier
    P.margin_rel = a->debug_margin == 0 || a->debug_margin == -2 ? 0x1p-40 : (a->debug_margin > 0 ? ldexp(1.0, -a->debug_margin) : 2.0);
    P.margin0_rel = a->debug_margin == 0 ? (float)LLDA_MARGIN0 : (a->debug_margin > 0 && a->debug_margin < 16 ? ldexpf(1.0f, -a->debug_margin) : 2.0f);
    hipStream_t st = (hipStream_t)stream;

    // sparse label sets: one lane per allowed topic, undecided documents continue in the dense kernel
    const bool sparse = fast && !dense && a->live_off && a->live_pos && a->resume && a->resume_count &&
                        a->resume_cap > 0 && a->live_max >= 1 && a->live_max <= LLDA_MAX_LIVE;
    const int G = sparse ? (a->live_max <= 8 ? 8 : a->live_max <= 16 ? 16 : a->live_max <= 32 ? 32 : 64) : L.G;
    const int gpb = 256 / G;
    int dpg = a->docs_per_group;
    // auto: one pass of documents per workgroup.  Workgroups of equal-length documents finish in lock step, so
    // the tail of the launch idles for up to one workgroup's run time: the shorter the workgroup the better
    // (synth2: 3540 M sites/s at 1, 3341 at 4, 3205 at 6 documents per lane group)
    if (dpg < 1) dpg = 1;
    P.dpg = dpg;
    const int64_t per_block = (int64_t)gpb * dpg;
    const int64_t blocks = (a->D + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;

    if (sparse) {
        P.live_off = a->live_off; P.live_pos = a->live_pos;
        P.resume = a->resume; P.resume_count = a->resume_count; P.resume_cap = a->resume_cap;
        hipError_t e = hipMemsetAsync(a->resume_count, 0, sizeof(int32_t), st);
        if (e != hipSuccess) return hip_fail(e);
        if (a->debug_margin < 0) P.margin_rel = 2.0;       // test hook: every document is handed to the dense kernel
        const dim3 grid((unsigned)blocks), block(256);
        switch (G) {
        case 8: hipLaunchKernelGGL(llda_sweep_sparse_kernel<8>, grid, block, 0, st, P); break;
        case 16: hipLaunchKernelGGL(llda_sweep_sparse_kernel<16>, grid, block, 0, st, P); break;
        case 32: hipLaunchKernelGGL(llda_sweep_sparse_kernel<32>, grid, block, 0, st, P); break;
        default: hipLaunchKernelGGL(llda_sweep_sparse_kernel<64>, grid, block, 0, st, P); break;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e);
        // second launch: the dense tiered kernel walks the (almost always empty) resume list
        P.resume_mode = 1;
        P.margin_rel = a->debug_margin == 0 || a->debug_margin == -2 ? 0x1p-40 : (a->debug_margin > 0 ? ldexp(1.0, -a->debug_margin) : 2.0);
        int64_t rblocks = ((int64_t)a->resume_cap + (256 / L.G) - 1) / (256 / L.G);
        if (rblocks > 256) rblocks = 256;
        switch (L.G) {
        case 8: return dispatch_sweep_T<8>(L.T, P, has_tail, true, false, rblocks, st);
        case 16: return dispatch_sweep_T<16>(L.T, P, has_tail, true, false, rblocks, st);
        case 32: return dispatch_sweep_T<32>(L.T, P, has_tail, true, false, rblocks, st);
        case 64: return dispatch_sweep_T<64>(L.T, P, has_tail, true, false, rblocks, st);
        }
        return LLDA_E_BAD_K;
    }
    switch (L.G) {
    case 8: return dispatch_sweep_T<8>(L.T, P, has_tail, fast, dense, blocks, st);
    case 16: return dispatch_sweep_T<16>(L.T, P, has_tail, fast, dense, blocks, st);
    case 32: return dispatch_sweep_T<32>(L.T, P, has_tail, fast, dense, blocks, st);
    case 64: return dispatch_sweep_T<64>(L.T, P, has_tail, fast, dense, blocks, st);
    }
    return LLDA_E_BAD_K;
}

int llda_commit_log(const int64_t *item_begin, const int32_t *item_len, const int32_t *item_word, int64_t n_items,
                    const uint32_t *commit_log, const int32_t *freq_csc, int32_t K, int32_t *target, int32_t *n_k,
                    int32_t *n_k_delta, void *stream)
{
    if (n_items < 0 || (n_k != nullptr) != (n_k_delta != nullptr)) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(K, &L);
    if (rc) return rc;
    if (n_items > 0 && (!item_begin || !item_len || !item_word || !commit_log || !freq_csc || !target)) return LLDA_E_BAD_ARG;
    if (n_items == 0 && !n_k) return LLDA_OK;
    CParams P;
    P.item_begin = item_begin; P.item_len = item_len; P.item_word = item_word; P.n_items = n_items;
    P.log = commit_log; P.freq = freq_csc; P.target = target; P.n_k = n_k; P.n_k_delta = n_k_delta; P.KP = L.KP;
    int64_t blocks = (n_items + 3) / 4;
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    hipLaunchKernelGGL(llda_commit_log_kernel, dim3((unsigned)blocks), dim3(256), 4 * L.KP * sizeof(int), (hipStream_t)stream, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_apply_delta(int32_t *counts, int32_t *delta, int64_t n, void *stream)
{
    if (!counts || !delta || n < 0) return LLDA_E_BAD_ARG;
    if (n == 0) return LLDA_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(counts) | reinterpret_cast<uintptr_t>(delta)) & 15) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(llda_apply_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       counts, delta, n4, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_count_init(const int64_t *doc_off, const int32_t *word, const int32_t *freq, const int32_t *z,
                    int64_t D, int32_t K, int32_t *n_dk, int32_t *n_kw, int32_t *n_k, void *stream)
{
    if (D < 0) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(K, &L);
    if (rc) return rc;
    if (D == 0) return LLDA_OK;
    if (!doc_off || !word || !freq || !z || !n_dk || !n_kw || !n_k) return LLDA_E_BAD_ARG;
    int64_t blocks = (D + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(llda_count_init_kernel, dim3((unsigned)blocks), dim3(256), 5 * L.KP * sizeof(int),
                       (hipStream_t)stream, doc_off, word, freq, z, D, L.KP, n_dk, n_kw, n_k);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_loglik(const int64_t *doc_off, const int32_t *word, const uint16_t *lab_mask, const int32_t *n_dk,
                const int32_t *n_kw, const int32_t *n_k, int64_t D, int64_t V, int32_t K, double alpha,
                double beta, double *out_doc, void *stream)
{
    if (D < 0 || V < 1) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(K, &L);
    if (rc) return rc;
    if (D == 0) return LLDA_OK;
    if (!doc_off || !word || !lab_mask || !n_dk || !n_kw || !n_k || !out_doc) return LLDA_E_BAD_ARG;
    LParams P;
    P.doc_off = doc_off; P.word = word; P.lab_mask = lab_mask; P.n_dk = n_dk; P.n_kw = n_kw; P.n_k = n_k;
    P.out_doc = out_doc; P.D = D; P.alpha = alpha; P.beta = beta; P.vbeta = (double)V * beta;
    hipStream_t st = (hipStream_t)stream;
    switch (L.G) {
    case 8: return dispatch_loglik_T<8>(L.T, P, st);
    case 16: return dispatch_loglik_T<16>(L.T, P, st);
    case 32: return dispatch_loglik_T<32>(L.T, P, st);
    case 64: return dispatch_loglik_T<64>(L.T, P, st);
    }
    return LLDA_E_BAD_K;
}

static void readout_layout(const llda_layout &L, RParams &P)
{
    P.K = L.K; P.KP = L.KP; P.T = L.T;
    for (int p = 0; p < LLDA_MAX_LEAVES; ++p) { P.leaf_start[p] = L.leaf_start[p]; P.leaf_len[p] = L.leaf_len[p]; }
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
}

int llda_readout_phi(const int32_t *n_kw, const int32_t *n_k, const double *den, int64_t V, int32_t K, double beta,
                     int32_t mode, double keep, double share, double *out, int32_t *flags, void *stream)
{
    if (V < 1 || (mode != 0 && mode != 1) || !n_kw || (!n_k && !den) || !out) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(K, &L);
    if (rc) return rc;
    RParams P;
    memset(&P, 0, sizeof P);
    readout_layout(L, P);
    P.n_kw = n_kw; P.n_k = n_k; P.den = den; P.out = out; P.flags = flags; P.V = V; P.mode = mode;
    P.beta = beta; P.vbeta = (double)V * beta; P.keep = keep; P.share = share;
    hipLaunchKernelGGL(llda_readout_phi_kernel, dim3((unsigned)((V + 63) / 64)), dim3(256), 0, (hipStream_t)stream, P);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

int llda_readout_theta(const int32_t *n_dk, const uint16_t *lab_mask, int64_t D, int32_t K, double alpha, int32_t mode,
                       double keep, double share, double *out, void *stream)
{
    if (D < 0 || (mode != 0 && mode != 1)) return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(K, &L);
    if (rc) return rc;
    if (D == 0) return LLDA_OK;
    if (!n_dk || !lab_mask || !out) return LLDA_E_BAD_ARG;
    RParams P;
    memset(&P, 0, sizeof P);
    readout_layout(L, P);
    P.n_dk = n_dk; P.lab_mask = lab_mask; P.out = out; P.D = D; P.mode = mode; P.alpha = alpha;
    P.keep = keep; P.share = share;
    hipStream_t st = (hipStream_t)stream;
    const bool has_tail = L.tail != 0;
    switch (L.G) {
    case 8: return dispatch_theta_T<8>(L.T, P, has_tail, st);
    case 16: return dispatch_theta_T<16>(L.T, P, has_tail, st);
    case 32: return dispatch_theta_T<32>(L.T, P, has_tail, st);
    case 64: return dispatch_theta_T<64>(L.T, P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

int llda_foldin(const llda_foldin_args *a, void *stream)
{
    if (!a || !a->doc_off || !a->word || !a->init_idx || !a->freq || !a->ph || !a->init_rows || !a->z || !a->n_dk ||
        !a->th || !a->slot_valid || a->D < 0 || a->iters < 0 || a->thinning < 1)
        return LLDA_E_BAD_ARG;
    llda_layout L;
    const int rc = llda_layout_init(a->K, &L);
    if (rc) return rc;
    if (a->D == 0) return LLDA_OK;
    FParams P;
    memset(&P, 0, sizeof P);
    P.doc_off = a->doc_off; P.word = a->word; P.init_idx = a->init_idx; P.freq = a->freq; P.z = a->z;
    P.ph = a->ph; P.phn = a->init_rows; P.n_dk = a->n_dk; P.th = a->th; P.slot_valid = a->slot_valid;
    P.status = a->status; P.D = a->D; P.doc_base = a->doc_base; P.alpha = a->alpha; P.beta = a->beta;
    P.c_init = a->c_init; P.c_loop = a->c_loop;
    P.key0 = (uint32_t)a->seed; P.key1 = (uint32_t)(a->seed >> 32); P.stream_id = a->stream_id;
    P.iters = a->iters; P.thinning = a->thinning; P.beta_fallback = a->beta_fallback; P.avg_mode = a->avg_mode;
    fill_schedule(L, P.last_leaf, P.tail, P.tail_row, P.n_rounds, P.xor_tree, P.rounds_pk);
    hipStream_t st = (hipStream_t)stream;
    const bool has_tail = L.tail != 0;
    switch (L.G) {
    case 8: return dispatch_foldin_T<8>(L.T, P, has_tail, st);
    case 16: return dispatch_foldin_T<16>(L.T, P, has_tail, st);
    case 32: return dispatch_foldin_T<32>(L.T, P, has_tail, st);
    case 64: return dispatch_foldin_T<64>(L.T, P, has_tail, st);
    }
    return LLDA_E_BAD_K;
}

int llda_selftest_div(uint64_t seed, int64_t n, unsigned long long *mismatches_dev, void *stream)
{
    if (!mismatches_dev || n < 1) return LLDA_E_BAD_ARG;
    const int iters = 1024;
    int64_t threads = (n + iters - 1) / iters;
    int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return LLDA_E_BAD_ARG;
    hipLaunchKernelGGL(llda_selftest_div_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       seed, iters, mismatches_dev);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LLDA_OK : hip_fail(e);
}

}  // extern "C"
