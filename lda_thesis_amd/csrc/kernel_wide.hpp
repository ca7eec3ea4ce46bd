// kernel_wide.hpp -- "wide" layouts: K with more than 8 pairwise leaves (some K in 969..1023, every K > 1024, up to
// LLDA_MAX_K): llda_sweep_wide_reg_kernel / llda_sweep_wide_kernel, llda_loglik_wide_kernel,
// llda_readout_theta_wide_kernel, llda_foldin_init_wide_kernel / llda_foldin_wide_kernel
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// The reference takes any K (LabeledLDA.py:52); numpy's pairwise sum (np.sum at LabeledLDA.py:117) then has more than
// 8 leaves and a document no longer fits the <= 64 lanes x <= 16 slots of the narrow kernels.  Here ONE wavefront walks
// one document and its 64 lanes play G = 64 * NT VIRTUAL lanes, tier by tier: virtual lane gv = 64 * tier + lane, the
// same position formula pos = ((s >> 2) * G + gv) * 4 + (s & 3) with the larger G (so every 16-byte chunk of the row is
// still read as one contiguous run per tier), the same (virtual lane, slot) draw order, and the Hillis-Steele scan of
// the keyed draw over all G virtual lanes (oracle/llda_oracle.py draw_keyed).  The exact pipeline (wide_sum, wide_div,
// wide_draw: run-time loops over NT and T, per-topic values staged in LDS in position order) is what the read-outs, the
// fold-in and the undecided sites of the sweep run; the sweep itself decides almost every site from unnormalised fp64
// prefix sums with the 2^-40 margin.  This is the GENERAL path: correct for every K, two to three
// orders of magnitude faster than a CPU core, not tuned to a roofline; none of the five BASELINE configs needs it.
struct WideLayout {
    int32_t NT, T, G, KP;              // tiers, slots per virtual lane (8 / 12 / 16), 64 * NT, G * T
    int32_t m, last_leaf, tail, tail_row;
    uint8_t comb_dst[LLDA_MAX_WIDE_LEAVES], comb_src[LLDA_MAX_WIDE_LEAVES];   // numpy's recursion tree, post-order
};

struct WParams {
    KParams k;
    WideLayout w;
};

constexpr int WIDE_MAX_TIERS = LLDA_MAX_WIDE_LEAVES / 8;

// np.sum of the KP values at wv (position order; exact zeros in masked / padded places) in numpy's pairwise order:
// per-virtual-lane chains, xor butterfly over the 8 chains of a leaf, the last leaf's tail, then the recursion tree
// over the leaf totals (lane l holds the total of leaf l; m <= 64).  Every lane returns the sum.
__device__ __forceinline__ double wide_sum(const double *wv, const WideLayout &W, int lane)
{
    const int T = W.T, G = W.G;
    double lt = 0.0;
    for (int t = 0; t < W.NT; ++t) {
        const int gv = t * 64 + lane, leaf = gv >> 3;
        const bool tail_lane = W.tail != 0 && leaf == W.last_leaf;
        double acc = 0.0, tv = 0.0;
        for (int c = 0; c < (T >> 2); ++c) {
            const double *p = wv + ((c * G + gv) << 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = p[j];
                if (tail_lane && 4 * c + j == W.tail_row) tv = v;
                else acc = acc + v;
            }
        }
        acc = acc + dpp_f64<DPP_XOR1>(acc);
        acc = acc + dpp_f64<DPP_XOR2>(acc);
        acc = acc + dpp_f64<DPP_HALF_MIRROR>(acc);
        if (W.tail != 0 && (W.last_leaf >> 3) == t) {
            for (int i = 0; i < W.tail; ++i) {
                const double o = __shfl(tv, (W.last_leaf & 7) * 8 + i, 64);
                if (leaf == W.last_leaf) acc = acc + o;
            }
        }
        const double mine = __shfl(acc, 8 * (lane & 7), 64);     // total of leaf 8t + (lane & 7)
        if ((lane >> 3) == t) lt = mine;
    }
    for (int i = 0; i + 1 < W.m; ++i) {
        const int a = W.comb_dst[i], b = W.comb_src[i];
        const double s = readlane_var_f64(lt, a) + readlane_var_f64(lt, b);
        if (lane == a) lt = s;
    }
    return readlane_var_f64(lt, 0);
}

// wv[i] = wv[i] / c for the entries this lane owns (the 32-byte chunks lane, lane + 64, ...); y = RN(1 / c)
__device__ __forceinline__ void wide_div(double *wv, const WideLayout &W, int lane, double c, double y)
{
    for (int q = lane; q < (W.KP >> 2); q += 64) {
        double *p = wv + (q << 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = div_by(p[j], c, y);
    }
}

// Hillis-Steele inclusive scan of the lane totals over the G = 64 * NT virtual lanes: x[gv] <- x[gv - d] + x[gv] for
// gv >= d, d = 1, 2, 4, ... < G (oracle/llda_oracle.py draw_keyed).  Returns x[G - 1] in every lane.
__device__ __forceinline__ double wide_scan(double (&X)[WIDE_MAX_TIERS], int NT, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const int src = (lane - d) & 63;
#pragma unroll
        for (int t = WIDE_MAX_TIERS - 1; t >= 0; --t) {          // descending: X[t - 1] is still the old value
            if (t < NT) {
                const double a = __shfl(X[t], src, 64);
                const double b = (t > 0) ? __shfl(X[t > 0 ? t - 1 : 0], src, 64) : 0.0;
                const double up = (lane >= d) ? a : b;
                if (lane >= d || t > 0) X[t] = up + X[t];
            }
        }
    }
    for (int dt = 1; dt < NT; dt <<= 1) {
#pragma unroll
        for (int t = WIDE_MAX_TIERS - 1; t >= 1; --t) {
            if (t < NT && t >= dt) {
                double lower = 0.0;
#pragma unroll
                for (int q = 0; q < WIDE_MAX_TIERS - 1; ++q)
                    if (q == t - dt) lower = X[q];
                X[t] = lower + X[t];
            }
        }
    }
    double tot = 0.0;
#pragma unroll
    for (int t = 0; t < WIDE_MAX_TIERS; ++t)
        if (t == NT - 1) tot = readlane_f64(X[t], 63);
    return tot;
}

// scan value of the virtual lane before this one (0 for virtual lane 0): Xt = this tier's values, Xlow = the tier below
__device__ __forceinline__ double wide_prev(double Xt, double Xlow, bool has_low, int lane)
{
    const double a = __shfl(Xt, (lane - 1) & 63, 64);
    const double b = has_low ? readlane_f64(Xlow, 63) : 0.0;
    return lane ? a : b;
}

// Keyed categorical draw over the KP probabilities at wv (overwritten by their per-lane prefix sums): the statements of
// draw_position<> with the G virtual lanes spread over NT tiers.  Returns the position (wave-uniform) or -1.
__device__ __forceinline__ int wide_draw(double *wv, const WideLayout &W, double u, int lane)
{
    const int T = W.T, G = W.G, NT = W.NT;
    double X[WIDE_MAX_TIERS];
    uint32_t pm[WIDE_MAX_TIERS];
#pragma unroll
    for (int t = 0; t < WIDE_MAX_TIERS; ++t) {
        X[t] = 0.0; pm[t] = 0;
        if (t < NT) {
            const int gv = t * 64 + lane;
            double run = 0.0;
            uint32_t m = 0;
            for (int c = 0; c < (T >> 2); ++c) {
                double *p = wv + ((c * G + gv) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double v = p[j];
                    run = (c == 0 && j == 0) ? v : run + v;
                    p[j] = run;
                    m |= (v > 0.0 ? 1u : 0u) << (4 * c + j);
                }
            }
            X[t] = run; pm[t] = m;
        }
    }
    const double tot = wide_scan(X, NT, lane);
    const double tt = u * tot;
    int hit = -1, last = -1;
#pragma unroll
    for (int t = 0; t < WIDE_MAX_TIERS; ++t) {
        if (t < NT) {
            const double a = __shfl(X[t], (lane - 1) & 63, 64);
            const double b = (t > 0) ? readlane_f64(X[t > 0 ? t - 1 : 0], 63) : 0.0;
            const double tg = tt - (lane ? a : b);
            const int gv = t * 64 + lane;
            uint32_t fm = 0;
            for (int c = 0; c < (T >> 2); ++c) {
                const double *p = wv + ((c * G + gv) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fm |= (((pm[t] >> (4 * c + j)) & 1u) && p[j] > tg) ? (1u << (4 * c + j)) : 0u;
            }
            const uint64_t bf = __ballot(fm != 0), bp = __ballot(pm[t] != 0);
            if (hit < 0 && bf != 0) {
                const int sl = (int)__ffsll((unsigned long long)bf) - 1;
                const int ss = __shfl((int)__ffs((int)(fm | 0x10000u)) - 1, sl, 64);
                hit = pos_of_rt(G, T, t * 64 + sl, ss);
            }
            if (bp != 0) {
                const int sl = 63 - (int)__clzll((unsigned long long)bp);
                const int ss = __shfl(31 - (int)__clz((int)(pm[t] | 1u)), sl, 64);
                last = pos_of_rt(G, T, t * 64 + sl, ss);
            }
        }
    }
    return hit >= 0 ? hit : last;
}

// ---------------------------------------------------------------------------------------------
// Sparse label sets on a wide layout (Labeled LDA proper with thousands of labels, a handful per document): the sparse
// kernel of kernel_sparse.hpp -- one lane per ALLOWED topic; its decided sites never see the layout -- with this exact tier
// for the ~1e-11 of the sites the margin cannot decide: the wavefront scatters the document's exact scores into the
// workgroup's KP doubles (zeros elsewhere: adding an exact zero never changes a partial sum) and runs wide_sum / wide_div /
// wide_draw on them.  One buffer per workgroup, taken with an LDS lock by the wavefront that needs it.
// ---------------------------------------------------------------------------------------------
struct WSParams : KParams {
    WideLayout w;
};
template <> struct sparse_is_wide<WSParams> { static constexpr bool value = true; };

// dynamic LDS of the wide sparse kernel: [KP] doubles, then the lock word
__device__ void sparse_wide_init()
{
    extern __shared__ double s_wide[];
    const WSParams *K = (const WSParams *)__builtin_amdgcn_kernarg_segment_ptr();
    *reinterpret_cast<int *>(s_wide + K->w.KP) = 0;
}

__device__ __noinline__ int exact_call(const WSParams *K, int, int, double wx, int pos, int base, int A, double u, int lane)
{
    extern __shared__ double s_wide[];
    const WideLayout &W = K->w;
    double *wv = s_wide;
    int *lock = reinterpret_cast<int *>(s_wide + W.KP);
    base = __builtin_amdgcn_readfirstlane(base);
    A = __builtin_amdgcn_readfirstlane(A);
    if (lane == 0)
        while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int q = lane; q < (W.KP >> 2); q += 64) {
        double *p = wv + (q << 2);
        p[0] = 0.0; p[1] = 0.0; p[2] = 0.0; p[3] = 0.0;
    }
    if (lane >= base && lane < base + A) wv[pos] = wx;            // (ordered after the zeros: LDS runs a wavefront's
                                                                  // instructions in order)
    const double S = wide_sum(wv, W, lane);                       // np.sum(prob)
    wide_div(wv, W, lane, S, 1.0 / S);                            // prob /= np.sum(prob)
    int r = wide_draw(wv, W, u, lane);
    if (!(S > 0.0)) r = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicExch(lock, 0);
    return r;
}

// topic held by a position of a wide row (-1: padding)
__device__ __forceinline__ int wide_topic_of(const int32_t *leaf_start, const int32_t *leaf_len, int G, int T, int pos)
{
    int gv, s;
    lane_slot_of_rt(G, T, pos, gv, s);
    const int leaf = gv >> 3, rel = (gv & 7) + 8 * s;
    return rel < leaf_len[leaf] ? leaf_start[leaf] + rel : -1;
}

// ---------------------------------------------------------------------------------------------
// Sweep, wide layouts: LabeledLDA.training_iteration (LabeledLDA.py:101-125), one wavefront per document.
// LDS per wavefront: KP doubles, the document's n_dk row and the n_k it sees (int32, position order; lane 0 applies the
// two +-f of a site).
// TIERED (alpha, beta >= 1e-6, V*beta < 2^40, as for the narrow kernels): the draw is DECIDED from unnormalised fp64 prefix
// sums with the 2^-40 margin of DESIGN.md 4.3 / 4.7 -- score = fac * (n_kw - own + beta) with fac = (n_dk + alpha) * ~1/(n_k +
// V*beta) cached per position in the KP doubles (lane 0 refreshes the two positions a site changes), lane totals in a first
// pass over the row, scan over the virtual lanes, then a second pass that recomputes the prefix and compares it with
// t -+ margin.  A site with any prefix value inside the margin band (or no hit) goes through the exact pipeline below
// -- the reference's arithmetic bit for bit -- which borrows the KP doubles; the factors are rebuilt afterwards.
// ---------------------------------------------------------------------------------------------
// Where the cached factor of a position lives in the KP doubles: slot-of-chunk major, [pos & 3][pos >> 2], so that the 64
// lanes of a pass read 64 consecutive doubles (position order -- a lane's four doubles side by side, lanes 32 bytes
// apart -- made two thirds of the LDS cycles bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.65).
__device__ __forceinline__ int fac_index(int pos, int KP4)
{
    return (pos & 3) * KP4 + (pos >> 2);
}

__device__ __forceinline__ double wide_factor(int ndk, int nkc, bool allowed, double alpha, double vbeta)
{
    return allowed ? ((double)ndk + alpha) * rcp_newton((double)nkc + vbeta) : 0.0;
}

// (dk: optional per-position change of the document against the start values at s_ndk / s_nkc, see the COMPACT kernel)
__device__ __forceinline__ void wide_factors(double *fac, const int *s_ndk, const int *s_nkc, const int16_t *dk,
                                             const uint16_t *mrow, const WideLayout &W, double alpha, double vbeta, int lane)
{
    for (int q = lane; q < (W.KP >> 2); q += 64) {
        const int c = q / W.G, gv = q - c * W.G;
        const uint32_t mask = mrow[gv];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = (q << 2) | j;
            const int d = dk ? (int)dk[pos] : 0;
            fac[j * (W.KP >> 2) + q] = wide_factor(s_ndk[pos] + d, s_nkc[pos] + d, (mask >> (4 * c + j)) & 1u, alpha, vbeta);
        }
    }
}

// the tier's decision for one site: position, or -1 when the margin cannot decide it
__device__ __forceinline__ int wide_tier(const double *fac, const int4 *xrow, const WideLayout &W, double u, int zo, int f,
                                         double beta, double margin_rel, int lane)
{
    const int T = W.T, G = W.G, NT = W.NT, KP4 = W.KP >> 2;
    double X[WIDE_MAX_TIERS];
#pragma unroll
    for (int t = 0; t < WIDE_MAX_TIERS; ++t) {
        X[t] = 0.0;
        if (t < NT) {
            const int gv = t * 64 + lane;
            double run = 0.0;
            for (int c = 0; c < (T >> 2); ++c) {
                const int q = c * G + gv;
                const int4 x4 = xrow[q];
                const double *p = fac + q;
                const int xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    run = run + p[j * KP4] * ((double)(xs[j] - ((((q << 2) | j) == zo) ? f : 0)) + beta);
            }
            X[t] = run;
        }
    }
    const double tot = wide_scan(X, NT, lane);
    const double tt = u * tot, m = margin_rel * tot;
    int hit = -1;
    bool unsure = !(tot > 0.0);
#pragma unroll
    for (int t = 0; t < WIDE_MAX_TIERS; ++t) {
        if (t < NT) {
            const double tg = tt - wide_prev(X[t], X[t > 0 ? t - 1 : 0], t > 0, lane);
            const double lo = tg - m, hi = tg + m;
            const int gv = t * 64 + lane;
            double run = 0.0;
            uint32_t hm = 0, um = 0;
            for (int c = 0; c < (T >> 2); ++c) {
                const int q = c * G + gv;
                const int4 x4 = xrow[q];
                const double *p = fac + q;
                const int xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    run = run + p[j * KP4] * ((double)(xs[j] - ((((q << 2) | j) == zo) ? f : 0)) + beta);
                    hm |= (run > hi ? 1u : 0u) << (4 * c + j);
                    um |= ((run > lo && !(run > hi)) ? 1u : 0u) << (4 * c + j);
                }
            }
            const uint64_t bh = __ballot(hm != 0);
            unsure = unsure || __ballot(um != 0) != 0;
            if (hit < 0 && bh != 0) {
                const int sl = (int)__ffsll((unsigned long long)bh) - 1;
                const int ss = __shfl((int)__ffs((int)(hm | 0x10000u)) - 1, sl, 64);
                hit = pos_of_rt(G, T, t * 64 + sl, ss);
            }
        }
    }
    return unsure ? -1 : hit;
}

// one site through the reference's fp64 pipeline (LabeledLDA.py:113-119): scores into wv, np.sum, prob /= sum, keyed draw
__device__ inline int wide_exact_site(double *wv, const int *s_ndk, const int *s_nkc, const int16_t *dk, const uint16_t *mrow,
                                      const int4 *xrow, const WideLayout &W, const KParams &K, int zo, int f, double u, int lane)
{
    const int G = W.G, T = W.T, NT = W.NT;
    for (int t = 0; t < NT; ++t) {
        const int gv = t * 64 + lane;
        const uint32_t mask = mrow[gv];
        for (int c = 0; c < (T >> 2); ++c) {
            const int q = c * G + gv;
            const int4 x4 = xrow[q];
            const int4 nd4 = reinterpret_cast<const int4 *>(s_ndk)[q];
            const int4 nk4 = reinterpret_cast<const int4 *>(s_nkc)[q];
            const int xs[4] = {x4.x, x4.y, x4.z, x4.w}, nds[4] = {nd4.x, nd4.y, nd4.z, nd4.w},
                      nks[4] = {nk4.x, nk4.y, nk4.z, nk4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pos = (q << 2) | j;
                const int d = dk ? (int)dk[pos] : 0;
                const double a = (double)(nds[j] + d) + K.alpha;
                const double num_b = (double)(xs[j] - (pos == zo ? f : 0)) + K.beta;
                const double den_b = (double)(nks[j] + d) + K.vbeta;
                const double ws = a * (num_b / den_b);               // LabeledLDA.py:113-116
                wv[pos] = ((mask >> (4 * c + j)) & 1u) ? ws : 0.0;
            }
        }
    }
    const double S = wide_sum(wv, W, lane);                          // np.sum(prob)
    wide_div(wv, W, lane, S, 1.0 / S);                               // prob /= np.sum(prob)
    int zn = wide_draw(wv, W, u, lane);
    if (zn < 0 || !(S > 0.0)) {
        zn = zo;
        if (lane == 0 && K.status) atomicOr(K.status, 1);            // no topic with positive probability
    }
    return zn;
}

template <bool TIERED>
__global__ void __launch_bounds__(64) llda_sweep_wide_kernel(const WParams P)
{
    extern __shared__ double s_wide[];
    const KParams &K = P.k;
    const WideLayout &W = P.w;
    const int lane = threadIdx.x, KP = W.KP, G = W.G;
    double *wv = s_wide;
    int *s_ndk = reinterpret_cast<int *>(wv + KP), *s_nkc = s_ndk + KP;
    int n_exact = 0;

    for (int64_t idx = blockIdx.x; idx < K.D; idx += gridDim.x) {
        const int64_t d = K.doc_order ? (int64_t)K.doc_order[idx] : idx;
        const int64_t s0 = K.doc_off[d];
        const int len = (int)(K.doc_off[d + 1] - s0);
        if (len <= 0) continue;
        int32_t *ndk_row = K.n_dk + d * KP;
        for (int q = lane; q < (KP >> 2); q += 64) {
            reinterpret_cast<int4 *>(s_ndk)[q] = reinterpret_cast<const int4 *>(ndk_row)[q];
            reinterpret_cast<int4 *>(s_nkc)[q] = reinterpret_cast<const int4 *>(K.n_k)[q];
        }
        const uint16_t *mrow = K.lab_mask + d * G;
        if (TIERED) wide_factors(wv, s_ndk, s_nkc, nullptr, mrow, W, K.alpha, K.vbeta, lane);
        const uint32_t gdoc = (uint32_t)(d + K.doc_base);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;

        for (int n = 0; n < len; ++n) {
            const int64_t i = s0 + n;
            const int v = K.word[i], f = K.freq[i], zo = K.z[i];
            const double u = site_uniform<64>(K, n, n == 0, gdoc, lane, r0, r1, r2, r3);
            if (lane == 0) {                                                 // remove the site (LabeledLDA.py:109-111)
                const int nd = s_ndk[zo] - f, nk = s_nkc[zo] - f;
                s_ndk[zo] = nd; s_nkc[zo] = nk;
                if (TIERED) wv[fac_index(zo, KP >> 2)] = wide_factor(nd, nk, wv[fac_index(zo, KP >> 2)] != 0.0, K.alpha, K.vbeta);   // (a cached factor is 0 exactly
                                                                                             // where the label mask is)
            }
            const int4 *xrow = reinterpret_cast<const int4 *>(K.n_kw + (int64_t)v * KP);
            int zn = -1;
            if (TIERED) zn = wide_tier(wv, xrow, W, u, zo, f, K.beta, K.margin_rel, lane);
            const bool exact = zn < 0;
            if (exact) {
                ++n_exact;
                zn = wide_exact_site(wv, s_ndk, s_nkc, nullptr, mrow, xrow, W, K, zo, f, u, lane);
            }
            if (lane == 0) {                                                 // add the site back (LabeledLDA.py:121-125)
                const int nd = s_ndk[zn] + f, nk = s_nkc[zn] + f;
                s_ndk[zn] = nd; s_nkc[zn] = nk;
                if (TIERED && !exact) wv[fac_index(zn, KP >> 2)] = wide_factor(nd, nk, true, K.alpha, K.vbeta);
                commit_site(K, i, v, f, zo, zn, K.csc_pos ? K.csc_pos[i] : 0, KP);
            }
            if (TIERED && exact) wide_factors(wv, s_ndk, s_nkc, nullptr, mrow, W, K.alpha, K.vbeta, lane);
        }
        for (int q = lane; q < (KP >> 2); q += 64) {
            const int4 old = reinterpret_cast<const int4 *>(ndk_row)[q];
            const int4 cur = reinterpret_cast<const int4 *>(s_ndk)[q];
            const int dl[4] = {cur.x - old.x, cur.y - old.y, cur.z - old.z, cur.w - old.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (dl[j]) atomicAdd(K.n_k_delta + ((q << 2) | j), dl[j]);
            reinterpret_cast<int4 *>(ndk_row)[q] = cur;
        }
    }
    if (TIERED && n_exact && lane == 0 && K.status) {                        // statistics, as the narrow kernels
        atomicOr(K.status, 2);
        atomicAdd(K.status + 2, n_exact);
    }
}

// ---------------------------------------------------------------------------------------------
// The tiered sweep with the row in registers (NT tiers known at compile time): the row of a site is loaded ONCE into
// registers and serves both passes of the decision, and the scalars of the next two sites run ahead.  The wide path
// runs 1 - 5 wavefronts per CU (LDS), so a wavefront owns hundreds of VGPRs; what it lacks is latency hiding (the
// LDS-only kernel above is one dependent chain word -> row -> pass 1 -> scan -> row again -> pass 2 -> next word).
// Measured (tools/abl_wide.py): K = 2 048 119 -> 217 M sites/s, K = 4 096 37 -> 64, K = 7 688 11 -> 16.  Also prefetching the
// row of site n+1 into a second register set was tried and is NOT faster (2 tiers: 210, 8 tiers: 9 -- it spills).
// ---------------------------------------------------------------------------------------------
// COMPACT (the caller vouches that no document holds 32 768 tokens or more, llda_sweep_args.max_doc_tokens): the
// document's counts are not copied to LDS at all -- they are the start values in HBM (its n_dk row, n_k: both constant
// while the document is sampled) plus ONE int16 change per position in LDS (a topic of the document changes n_dk and the
// n_k it sees by the same amount).  10 bytes of LDS per position instead of 16: 8 wavefronts per CU instead of 5 at K = 2 048,
// 4 instead of 2 at K = 4 096, 2 instead of 1 at K = 7 688 -- and this path is bound by exactly that.  Lane 0 fetches the two
// start values of the old topic one site ahead and those of the new topic after the draw (L2 hits).
template <int NT, bool COMPACT>
__global__ void __launch_bounds__(64) llda_sweep_wide_reg_kernel(const WParams P)
{
    extern __shared__ double s_wide[];
    const KParams &K = P.k;
    const WideLayout &W = P.w;
    const int lane = threadIdx.x, KP = W.KP, KP4 = W.KP >> 2, T = W.T, TC = W.T >> 2;
    constexpr int G = 64 * NT;
    double *wv = s_wide;
    int *s_ndk = reinterpret_cast<int *>(wv + KP), *s_nkc = s_ndk + KP;          // (!COMPACT)
    int16_t *s_dk = reinterpret_cast<int16_t *>(wv + KP);                        // (COMPACT)
    int n_exact = 0;

    for (int64_t idx = blockIdx.x; idx < K.D; idx += gridDim.x) {
        const int64_t d = K.doc_order ? (int64_t)K.doc_order[idx] : idx;
        const int64_t s0 = K.doc_off[d];
        const int len = (int)(K.doc_off[d + 1] - s0);
        if (len <= 0) continue;
        int32_t *ndk_row = K.n_dk + d * KP;
        for (int q = lane; q < (KP >> 2); q += 64) {
            if (COMPACT) {
                reinterpret_cast<int2 *>(s_dk)[q] = make_int2(0, 0);
            } else {
                reinterpret_cast<int4 *>(s_ndk)[q] = reinterpret_cast<const int4 *>(ndk_row)[q];
                reinterpret_cast<int4 *>(s_nkc)[q] = reinterpret_cast<const int4 *>(K.n_k)[q];
            }
        }
        // where the counts are read: start values (HBM) + change (LDS), or the LDS copies
        const int *c_ndk = COMPACT ? ndk_row : s_ndk, *c_nk = COMPACT ? K.n_k : s_nkc;
        const int16_t *c_dk = COMPACT ? s_dk : nullptr;
        const uint16_t *mrow = K.lab_mask + d * G;
        wide_factors(wv, c_ndk, c_nk, c_dk, mrow, W, K.alpha, K.vbeta, lane);
        const uint32_t gdoc = (uint32_t)(d + K.doc_base);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;

        int v_c = K.word[s0], f_c = K.freq[s0], zo_c = K.z[s0];
        const int64_t i1 = s0 + (len > 1 ? 1 : 0);
        int v_1 = K.word[i1], f_1 = K.freq[i1], zo_1 = K.z[i1];
        int c_c = K.csc_pos ? K.csc_pos[s0] : 0, c_1 = K.csc_pos ? K.csc_pos[i1] : 0;       // (commit-log positions)
        int nd0_c = 0, nk0_c = 0;                                                // COMPACT: start counts of the old topic
        if (COMPACT) { nd0_c = ndk_row[zo_c]; nk0_c = K.n_k[zo_c]; }

        for (int n = 0; n < len; ++n) {
            const int64_t i = s0 + n;
            const int v = v_c, f = f_c, zo = zo_c, cpos = c_c, nd0 = nd0_c, nk0 = nk0_c;
            int x[NT][4][4];
            {
                const int4 *row = reinterpret_cast<const int4 *>(K.n_kw + (int64_t)v * KP);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < TC) {
                            const int4 r = row[c * G + t * 64 + lane];
                            x[t][c][0] = r.x; x[t][c][1] = r.y; x[t][c][2] = r.z; x[t][c][3] = r.w;
                        }
                // scalars of site n+2
                v_c = v_1; f_c = f_1; zo_c = zo_1; c_c = c_1;
                if (COMPACT) { nd0_c = ndk_row[zo_c]; nk0_c = K.n_k[zo_c]; }     // (for site n+1; zo_c arrived a site ago)
                const int64_t i2 = s0 + (n + 2 < len ? n + 2 : len - 1);
                v_1 = K.word[i2]; f_1 = K.freq[i2]; zo_1 = K.z[i2];
                if (K.csc_pos) c_1 = K.csc_pos[i2];
            }
            const double u = site_uniform<64>(K, n, n == 0, gdoc, lane, r0, r1, r2, r3);
            if (lane == 0) {                                                 // remove the site (LabeledLDA.py:109-111)
                int nd, nk;
                if (COMPACT) {
                    const int dz = (int)s_dk[zo] - f;
                    s_dk[zo] = (int16_t)dz;
                    nd = nd0 + dz; nk = nk0 + dz;
                } else {
                    nd = s_ndk[zo] - f; nk = s_nkc[zo] - f;
                    s_ndk[zo] = nd; s_nkc[zo] = nk;
                }
                wv[fac_index(zo, KP >> 2)] = wide_factor(nd, nk, wv[fac_index(zo, KP >> 2)] != 0.0, K.alpha, K.vbeta);     // (a cached factor is 0 exactly where the
                                                                                   // label mask is: no mask load per site)
            }
            // own count out of the row: position zo belongs to lane (zo >> 2) & 63, tier ((zo >> 2) % G) >> 6, chunk (zo >> 2) / G
            {
                const int qz = zo >> 2, cz = qz / G, gz = qz - cz * G;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < TC) {
                            const bool mine = (gz == t * 64 + lane) && (cz == c);
#pragma unroll
                            for (int j = 0; j < 4; ++j) x[t][c][j] -= (mine && (zo & 3) == j) ? f : 0;
                        }
            }
            // pass 1: lane totals
            double X[WIDE_MAX_TIERS];
#pragma unroll
            for (int t = 0; t < WIDE_MAX_TIERS; ++t) X[t] = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double run = 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < TC) {
                        const double *p = wv + (c * G + t * 64 + lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) run = run + p[j * KP4] * ((double)x[t][c][j] + K.beta);
                    }
                X[t] = run;
            }
            const double tot = wide_scan(X, NT, lane);
            const double tt = u * tot, m = K.margin_rel * tot;
            int zn = -1;
            bool unsure = !(tot > 0.0);
            // pass 2: prefix against t -+ margin
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double tg = tt - wide_prev(X[t], X[t > 0 ? t - 1 : 0], t > 0, lane);
                const double lo = tg - m, hi = tg + m;
                double run = 0.0;
                uint32_t hm = 0, um = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < TC) {
                        const double *p = wv + (c * G + t * 64 + lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            run = run + p[j * KP4] * ((double)x[t][c][j] + K.beta);
                            hm |= (run > hi ? 1u : 0u) << (4 * c + j);
                            um |= ((run > lo && !(run > hi)) ? 1u : 0u) << (4 * c + j);
                        }
                    }
                const uint64_t bh = __ballot(hm != 0);
                unsure = unsure || __ballot(um != 0) != 0;
                if (zn < 0 && bh != 0) {
                    const int sl = (int)__ffsll((unsigned long long)bh) - 1;
                    const int ss = __shfl((int)__ffs((int)(hm | 0x10000u)) - 1, sl, 64);
                    zn = pos_of_rt(G, T, t * 64 + sl, ss);
                }
            }
            const bool exact = unsure || zn < 0;
            if (__builtin_expect(exact, 0)) {
                ++n_exact;
                zn = wide_exact_site(wv, c_ndk, c_nk, c_dk, mrow, reinterpret_cast<const int4 *>(K.n_kw + (int64_t)v * KP), W, K,
                                     zo, f, u, lane);
            }
            if (lane == 0) {                                                 // add the site back (LabeledLDA.py:121-125)
                int nd, nk;
                if (COMPACT) {
                    const int dz = (int)s_dk[zn] + f;
                    s_dk[zn] = (int16_t)dz;
                    nd = ndk_row[zn] + dz; nk = K.n_k[zn] + dz;
                } else {
                    nd = s_ndk[zn] + f; nk = s_nkc[zn] + f;
                    s_ndk[zn] = nd; s_nkc[zn] = nk;
                }
                if (!exact) wv[fac_index(zn, KP >> 2)] = wide_factor(nd, nk, true, K.alpha, K.vbeta);
                commit_site(K, i, v, f, zo, zn, cpos, KP);
            }
            if (__builtin_expect(exact, 0)) wide_factors(wv, c_ndk, c_nk, c_dk, mrow, W, K.alpha, K.vbeta, lane);
        }
        for (int q = lane; q < (KP >> 2); q += 64) {
            if (COMPACT) {
                const int2 pk = reinterpret_cast<const int2 *>(s_dk)[q];
                if (pk.x | pk.y) {
                    const int dl[4] = {(int)(int16_t)(pk.x & 0xFFFF), pk.x >> 16, (int)(int16_t)(pk.y & 0xFFFF), pk.y >> 16};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dl[j]) {
                            atomicAdd(K.n_k_delta + ((q << 2) | j), dl[j]);
                            ndk_row[(q << 2) | j] += dl[j];
                        }
                }
            } else {
                const int4 old = reinterpret_cast<const int4 *>(ndk_row)[q];
                const int4 cur = reinterpret_cast<const int4 *>(s_ndk)[q];
                const int dl[4] = {cur.x - old.x, cur.y - old.y, cur.z - old.z, cur.w - old.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (dl[j]) atomicAdd(K.n_k_delta + ((q << 2) | j), dl[j]);
                reinterpret_cast<int4 *>(ndk_row)[q] = cur;
            }
        }
    }
    if (n_exact && lane == 0 && K.status) {                                  // statistics, as the narrow kernels
        atomicOr(K.status, 2);
        atomicAdd(K.status + 2, n_exact);
    }
}

// ---------------------------------------------------------------------------------------------
// Tier 0 for wide layouts: the fp32 decision of the narrow kernels in front of the fp64 decision of the
// register kernel above, with the G = 64 * NT virtual lanes of one wavefront.  Same LDS state as that kernel in its COMPACT
// form (one fp64 factor (n_dk + alpha) * ~1/(n_k + V*beta) and one int16 count change per position), so the rare tiers
// need nothing rebuilt; what changes is the work per site:
//   * ONE pass in fp32: the row (registers) times the factors (rounded to fp32 as they are read) gives the T fp32 prefix
//     values of each of the lane's NT virtual lanes (packed fp32), which stay in registers; lane totals are scanned over the
//     wavefront (DPP) tier by tier with a sequential carry over the tiers; per virtual lane the count of prefix values below
//     the target and the "no value inside the margin band" test are the branch-free search of the narrow kernel
//     (~350 VALU instructions per site at K = 2 048 instead of 660 fp64 ones);
//   * the site's own count leaves the row through ONE indexed register write (zo is uniform: the index goes through M0);
//   * the row of site n + 1 is in flight while site n is decided (two register sets), scalars two sites ahead, the start
//     counts of the next site's old topic one site ahead.
// Error bound (v = 2^-24 per fp32 rounding, total score 1; as DESIGN.md 4.3 with fp64-accurate factors and the longer scan): a
// factor rounded from fp64 is within 1v; (float)x + beta32 within 3v (x >= 2^24 rounds, beta32 is rounded, the sum rounds); the
// score within 5v; a virtual lane's prefix adds <= 15v: 20v; the wavefront scan adds 6v, the carry over <= 7 tiers 7v and the add
// of the carry 1v: X within 34v, the total within 33v; t = fl(u~ * total) within 35.2v (u~: the uniform's top 27 bits rounded to
// fp32); the target fl(t - X[g-1]) within 70.2v, its margin bounds one more v: every compared
// difference is within 20v + 71.2v < 92v of its real value (the exact pipeline: 2^-44).  Margin: 112v (LLDA_MARGIN0_WIDE),
// so a "sure" decision has the signs of the exact pipeline.  An unsure site -- 2 * margin * (topics with a score above the
// margin): ~3 % at K = 2 048 -- takes the fp64 two-pass decision on the same registers and factors and, if that is unsure
// too (~1e-9), the exact pipeline; the topic is always the exact pipeline's.  Needs max_doc_tokens < 2^15 (int16 changes).
// ---------------------------------------------------------------------------------------------
// fp32 image of the fp64 factors (SLIM form of the kernel below): computed in fp64, rounded once
__device__ __forceinline__ void wide_factors32(float *fac, const int *s_ndk, const int *s_nkc, const uint16_t *mrow,
                                               const WideLayout &W, double alpha, double vbeta, int lane)
{
    for (int q = lane; q < (W.KP >> 2); q += 64) {
        const int c = q / W.G, gv = q - c * W.G;
        const uint32_t mask = mrow[gv];
        const int4 nd = reinterpret_cast<const int4 *>(s_ndk)[q], nk = reinterpret_cast<const int4 *>(s_nkc)[q];
        const int nds[4] = {nd.x, nd.y, nd.z, nd.w}, nks[4] = {nk.x, nk.y, nk.z, nk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            fac[j * (W.KP >> 2) + q] = (float)wide_factor(nds[j], nks[j], (mask >> (4 * c + j)) & 1u, alpha, vbeta);
    }
}

// The rare tiers of a site of the SLIM kernel, out of line (a real call: their registers are not part of the hot loop's
// allocation) and on a scratch row of KP doubles in HBM: fp64 factors from the start counts + the int16 changes, the fp64
// two-pass decision (the row is read again), then the exact pipeline.  Returns the position; *exact tells whether the exact
// pipeline ran.  (Parameters through the kernel-argument segment: see the note at the exact tier of the kernel below.)
__device__ __noinline__ int wide_rare_tiers(const WParams *Pk, double *scr, const int32_t *ndk_row, const int16_t *s_dk,
                                            const uint16_t *mrow, int v, int zo, int f, uint32_t ra, uint32_t rb, int lane, int *exact)
{
    const KParams &K = Pk->k;
    const WideLayout &W = Pk->w;
    const double u = uniform53(ra, rb);
    const int4 *xrow = reinterpret_cast<const int4 *>(K.n_kw + (int64_t)v * W.KP);
    wide_factors(scr, ndk_row, K.n_k, s_dk, mrow, W, K.alpha, K.vbeta, lane);
    int zn = wide_tier(scr, xrow, W, u, zo, f, K.beta, K.margin_rel, lane);
    *exact = 0;
    if (zn < 0) {
        *exact = 1;
        zn = wide_exact_site(scr, ndk_row, K.n_k, s_dk, mrow, xrow, W, K, zo, f, u, lane);
    }
    return zn;
}

#ifndef LLDA_MARGIN0_WIDE
#define LLDA_MARGIN0_WIDE (112.0f * 0x1p-24f)
#endif

// TC = slots per virtual lane / 4 (3 or 4: the only values wide layouts have).  (Registers: two rows, the prefix values and
// the fp64 tier's temporaries want more than the 168 VGPRs of three wavefronts per SIMD -- with that cap the allocator
// spilled the PREFETCHED row, i.e. waited for it on the spot; LDS allows 8 wavefronts per CU at K = 2 048 anyway.)
// SLIM (the caller passes llda_sweep_args.scratch): the factors live in LDS as fp32 only -- 6 instead of 10 bytes per position: 13
// instead of 8 wavefronts per CU at K = 2 048 -- rounded once from the fp64 value (the error bound above is unchanged); the sites
// tier 0 is unsure about take wide_rare_tiers() on a scratch row in HBM.  With a third wavefront per SIMD the row is NOT double-
// buffered (two rows + prefix values + a call do not fit 168 VGPRs: the allocator spilled the prefetched row, i.e. waited for it on
// the spot).  Measured (tools/abl_wide.py, M sites/s, SLIM vs fp64 factors in LDS): K = 1 088 841 / 838, K = 2 048 669 / 830,
// K = 3 000 384 / 337, K = 4 296 78 / 90, K = 7 688 57 / 84 -- three wavefronts without the row prefetch are no better than two
// with it, and beyond four tiers the dearer fall-back (7 % of the sites at K = 7 688) costs more than the occupancy buys.
// llda_sweep therefore takes SLIM only for three and four tiers (debug_margin -7 forces it, -6 forbids it).
template <int NT, int TC, bool SLIM>
__global__ void __launch_bounds__(64)
    __attribute__((amdgpu_waves_per_eu(SLIM ? (NT <= 2 ? 3 : NT <= 4 ? 2 : 1) : (NT <= 4 ? 2 : 1))))
llda_sweep_wide_f32_kernel(const WParams P, const float margin0_rel, double *scratch)
{
    extern __shared__ double s_wide[];
    const KParams &K = P.k;
    const WideLayout &W = P.w;
    constexpr int G = 64 * NT, T = 4 * TC;
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef typename std::conditional<SLIM, float, double>::type fac_t;
    const int lane = threadIdx.x, KP = W.KP, KP4 = W.KP >> 2;
    fac_t *wv = reinterpret_cast<fac_t *>(s_wide);              // factor of position p at (p & 3) * KP4 + (p >> 2)
    int16_t *s_dk = reinterpret_cast<int16_t *>(wv + KP);        // change of n_dk (= of the n_k the document sees), position order
    double *scr = SLIM ? scratch + (size_t)blockIdx.x * KP : nullptr;    // the rare tiers' row of doubles
    const float beta32 = (float)K.beta;
    int n_unsure = 0, n_exact = 0;

    for (int64_t idx = blockIdx.x; idx < K.D; idx += gridDim.x) {
        const int64_t d = K.doc_order ? (int64_t)K.doc_order[idx] : idx;
        const int64_t s0 = K.doc_off[d];
        const int len = (int)(K.doc_off[d + 1] - s0);
        if (len <= 0) continue;
        int32_t *ndk_row = K.n_dk + d * KP;
        const uint16_t *mrow = K.lab_mask + d * G;
        for (int q = lane; q < KP4; q += 64) reinterpret_cast<int2 *>(s_dk)[q] = make_int2(0, 0);
        if constexpr (SLIM) wide_factors32(wv, ndk_row, K.n_k, mrow, W, K.alpha, K.vbeta, lane);
        else wide_factors(wv, ndk_row, K.n_k, s_dk, mrow, W, K.alpha, K.vbeta, lane);
        uint32_t mk[NT];                                          // allowed slots of this lane's virtual lanes
#pragma unroll
        for (int t = 0; t < NT; ++t) mk[t] = mrow[t * 64 + lane];
        const uint32_t gdoc = (uint32_t)(d + K.doc_base);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;

        auto load_row = [&](const int v, v16i (&x)[NT]) {
            const int4 *row = reinterpret_cast<const int4 *>(K.n_kw + (int64_t)v * KP);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int c = 0; c < TC; ++c) {
#ifdef ABL_WIDE_NOROW                                           // ablation: no n_kw traffic
                    const int4 r = make_int4(v & 7, c, t, lane & 3);
#else
                    const int4 r = row[c * G + t * 64 + lane];
#endif
                    x[t][4 * c] = r.x; x[t][4 * c + 1] = r.y; x[t][4 * c + 2] = r.z; x[t][4 * c + 3] = r.w;
                }
        };
        // scalars of the sites n (._c) and n + 1 (._1) are in registers at the top of site n; the row of site n too
        int v_c = K.word[s0], f_c = K.freq[s0], zo_c = K.z[s0];
        const int64_t i1 = s0 + (len > 1 ? 1 : 0);
        int v_1 = K.word[i1], f_1 = K.freq[i1], zo_1 = K.z[i1];
        int c_c = K.csc_pos ? K.csc_pos[s0] : 0, c_1 = K.csc_pos ? K.csc_pos[i1] : 0;       // (commit-log positions)
        // (two row buffers need 32 * NT registers: beyond two tiers the allocator keeps them in scratch memory, so there the
        // row of a site is loaded at its top and only the scalars run ahead)
        constexpr bool PREFETCH = NT <= 2 && !SLIM;
        v16i xa[NT], xb[PREFETCH ? NT : 1];
        if constexpr (PREFETCH) load_row(__builtin_amdgcn_readfirstlane(v_c), xa);
        // a topic count of the document changes by df (lane 0): the int16 change and the cached factor; nd0, nk0 = the
        // start values of the position (HBM; fetched ahead of time where the position is known ahead of time)
        auto count_change = [&](const int pos, const int df, const int nd0, const int nk0) {
            const int dz = (int)s_dk[pos] + df;
            s_dk[pos] = (int16_t)dz;
            fac_t *pf = wv + fac_index(pos, KP4);
            // (a cached factor is 0 exactly where the label mask is: no mask load per site)
            *pf = (fac_t)wide_factor(nd0 + dz, nk0 + dz, *pf != (fac_t)0, K.alpha, K.vbeta);
        };
        if (lane == 0) count_change(zo_c, -f_c, ndk_row[zo_c], K.n_k[zo_c]);    // site 0 leaves its topic (LabeledLDA.py:109-111)

        auto site = [&](const int n, v16i (&x)[NT], v16i (&xnext)[PREFETCH ? NT : 1]) {
            const int64_t i = s0 + n;
            const int v = __builtin_amdgcn_readfirstlane(v_c), f = __builtin_amdgcn_readfirstlane(f_c),
                      zo = __builtin_amdgcn_readfirstlane(zo_c), cpos = c_c;
            uint32_t ra, rb;
            site_random_bits<64>(K, n, n == 0, gdoc, lane, r0, r1, r2, r3, ra, rb);
            // row of site n + 1, scalars of site n + 2, start counts of site n + 1's old topic (needed at the end of this site)
            if constexpr (PREFETCH) load_row(__builtin_amdgcn_readfirstlane(v_1), xnext);
            else load_row(v, x);
            v_c = v_1; f_c = f_1; zo_c = zo_1; c_c = c_1;
            const int nd0_n = ndk_row[zo_c], nk0_n = K.n_k[zo_c];
            {
                const int64_t i2 = s0 + (n + 2 < len ? n + 2 : len - 1);
                v_1 = K.word[i2]; f_1 = K.freq[i2]; zo_1 = K.z[i2];
                if (K.csc_pos) c_1 = K.csc_pos[i2];
            }
            // own count out of the row: position zo is slot 4 * chunk + (zo & 3) of virtual lane (zo >> 2) % G -- all uniform,
            // so it is ONE indexed register write in the lane that owns it
            {
                const int qz = zo >> 2, cz = qz / G, gz = qz - cz * G, tz = gz >> 6, lz = gz & 63, sz = 4 * cz + (zo & 3);
                const int fm = (lane == lz) ? f : 0;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (tz == t) {
                        v16i xv = x[t];                              // (a local tuple: the index goes through M0)
                        xv[sz & 15] -= fm;
                        x[t] = xv;
                    }
            }
            // ---- tier 0: fp32 prefix values, lane totals, scan ----
            int zn = -1;
            bool decided = false;
            // (issue priority by phase, as the quad kernels': the chains' vector work wins the arbitration against the other wavefronts'
            // one-lane count updates -- 2.381 -> 2.336 ms per sweep at K = 2 048, profiles/r06_site_loop_budget.md section 2)
            __builtin_amdgcn_s_setprio(2);
            if (margin0_rel < 1.0f) {
                float q[NT][T], Y[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int xs[T];
                    float pa[T];
#pragma unroll
                    for (int c = 0; c < TC; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) pa[4 * c + j] = (float)wv[j * KP4 + c * G + t * 64 + lane];
#pragma unroll
                    for (int s = 0; s < T; ++s) xs[s] = x[t][s];
                    prefix_scores_f32<T, true>(q[t], xs, pa, 0xFFFFu, beta32);
                    Y[t] = group_scan_f32<64>(q[t][T - 1], lane);
                }
                float carry[NT + 1];
                carry[0] = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    carry[t + 1] = carry[t] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Y[t]), 63));
                const float tot = carry[NT];
                const float u32 = (float)(ra >> 5) * 0x1p-27f;   // fp32 image of the uniform: its top 27 bits
                const float tt = u32 * tot, margin = tot * margin0_rel;
                bool dirty = !(tot > 0.0f) || !(margin < tot) || !(tot < 3.0e38f);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float prev = dpp_f32<DPP_WAVE_SHR1>(Y[t]);
                    const float before = lane ? prev + carry[t] : carry[t];       // X of the virtual lane before this one
                    const float tg = tt - before;
                    int cnt;
                    bool clean;
                    count_sorted_f32<T>(q[t], tg - margin, tg + margin, cnt, clean);
                    dirty = dirty || !clean;
                    const uint32_t fmk = mk[t] & (0xFFFFu << cnt);
                    const uint64_t bh = __ballot(fmk != 0);
                    if (zn < 0 && bh != 0) {
                        const int sl = (int)__builtin_ctzll(bh);
                        const int ss = __builtin_amdgcn_readlane((int)__ffs((int)(fmk | 0x10000u)) - 1, sl);
                        zn = pos_of_rt(G, T, t * 64 + sl, ss);
                    }
                }
                // (a clean site always has a prefix value above its target -- the last allowed one is the total, and u < 1 --
                // so "no hit" goes the way of the unsure sites)
                decided = __ballot(dirty) == 0 && zn >= 0;
            }
            bool exact = false;
            if (__builtin_expect(!decided, 0)) {
              ++n_unsure;
              if constexpr (SLIM) {
                int ex = 0;
                zn = wide_rare_tiers((const WParams *)__builtin_amdgcn_kernarg_segment_ptr(), scr, ndk_row, s_dk, mrow, v, zo, f, ra, rb,
                                     lane, &ex);
                n_exact += ex;
              } else {
                // ---- tier 1: the fp64 two-pass decision of the register kernel on the same row and factors ----
                const double u = uniform53(ra, rb);
                double X[WIDE_MAX_TIERS];
#pragma unroll
                for (int t = 0; t < WIDE_MAX_TIERS; ++t) X[t] = 0.0;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    double run = 0.0;
#pragma unroll
                    for (int c = 0; c < TC; ++c) {
                        const double *p = wv + (c * G + t * 64 + lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) run = run + p[j * KP4] * ((double)x[t][4 * c + j] + K.beta);
                    }
                    X[t] = run;
                }
                const double tot = wide_scan(X, NT, lane);
                const double tt = u * tot, m = K.margin_rel * tot;
                zn = -1;
                bool unsure = !(tot > 0.0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double tg = tt - wide_prev(X[t], X[t > 0 ? t - 1 : 0], t > 0, lane);
                    const double lo = tg - m, hi = tg + m;
                    double run = 0.0;
                    uint32_t hm = 0, um = 0;
#pragma unroll
                    for (int c = 0; c < TC; ++c) {
                        const double *p = wv + (c * G + t * 64 + lane);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            run = run + p[j * KP4] * ((double)x[t][4 * c + j] + K.beta);
                            hm |= (run > hi ? 1u : 0u) << (4 * c + j);
                            um |= ((run > lo && !(run > hi)) ? 1u : 0u) << (4 * c + j);
                        }
                    }
                    const uint64_t bh = __ballot(hm != 0);
                    unsure = unsure || __ballot(um != 0) != 0;
                    if (zn < 0 && bh != 0) {
                        const int sl = (int)__ffsll((unsigned long long)bh) - 1;
                        const int ss = __shfl((int)__ffs((int)(hm | 0x10000u)) - 1, sl, 64);
                        zn = pos_of_rt(G, T, t * 64 + sl, ss);
                    }
                }
                exact = unsure || zn < 0;
                if (__builtin_expect(exact, 0)) {
                    ++n_exact;
                    // (the parameters through the kernel-argument segment: wide_sum indexes the layout's tables dynamically,
                    // and on the by-value copy that would put ALL of P into scratch memory -- and every pointer of the hot
                    // loop behind a scratch load)
                    const WParams *Pk = (const WParams *)__builtin_amdgcn_kernarg_segment_ptr();
                    zn = wide_exact_site(wv, ndk_row, Pk->k.n_k, s_dk, mrow,
                                         reinterpret_cast<const int4 *>(Pk->k.n_kw + (int64_t)v * KP), Pk->w, Pk->k, zo, f, u, lane);
                    wide_factors(wv, ndk_row, Pk->k.n_k, s_dk, mrow, Pk->w, Pk->k.alpha, Pk->k.vbeta, lane);   // (the exact pipeline borrowed the doubles)
                }
              }
            }
            __builtin_amdgcn_s_setprio(0);
            if (lane == 0) {
#ifdef ABL_WIDE_NOADDLOAD                                        // ablation (tools/abl_wide.py): no start-value loads behind the draw
                const int nd0_z = 0, nk0_z = 1000;
#else
                const int nd0_z = ndk_row[zn], nk0_z = K.n_k[zn];   // (issued first: everything below up to their use overlaps them)
#endif
                commit_site(K, i, v, f, zo, zn, cpos, KP);
                const bool more = n + 1 < len;
                // take the next site out of its topic already; when that is the topic just drawn the two changes must be
                // applied in order (the factor is computed from the final count)
                if (more && zo_c != zn) count_change(zo_c, -f_c, nd0_n, nk0_n);
                count_change(zn, f, nd0_z, nk0_z);               // add the site back (LabeledLDA.py:121-125)
                if (more && zo_c == zn) count_change(zo_c, -f_c, nd0_n, nk0_n);
            }
        };
        if constexpr (PREFETCH) {
            for (int n = 0;; n += 2) {
                site(n, xa, xb);
                if (n + 1 >= len) break;
                site(n + 1, xb, xa);
                if (n + 2 >= len) break;
            }
        } else {
            for (int n = 0; n < len; ++n) site(n, xa, xb);
        }
        for (int q = lane; q < KP4; q += 64) {
            const int2 pk = reinterpret_cast<const int2 *>(s_dk)[q];
            if (pk.x | pk.y) {
                const int dl[4] = {(int)(int16_t)(pk.x & 0xFFFF), pk.x >> 16, (int)(int16_t)(pk.y & 0xFFFF), pk.y >> 16};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (dl[j]) {
                        atomicAdd(K.n_k_delta + ((q << 2) | j), dl[j]);
                        ndk_row[(q << 2) | j] += dl[j];
                    }
            }
        }
    }
    if (lane == 0 && K.status) {                                 // statistics, as the narrow kernels
        if (n_unsure) atomicAdd(K.status + 1, n_unsure);
        if (n_exact) {
            atomicOr(K.status, 2);
            atomicAdd(K.status + 2, n_exact);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Read-outs, wide layouts
// ---------------------------------------------------------------------------------------------
struct WLParams {
    LParams l;
    int32_t NT, T, G, KP;
};

__device__ __forceinline__ double wave_allsum(double x)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) x = x + __shfl_xor(x, d, 64);
    return x;
}

// log-likelihood (LabeledLDA.py:256-265); as llda_loglik_kernel, theta of the document staged in LDS
__global__ void __launch_bounds__(64) llda_loglik_wide_kernel(const WLParams P)
{
    extern __shared__ double s_wide[];
    const LParams &L = P.l;
    const int lane = threadIdx.x, KP = P.KP, G = P.G;
    double *th = s_wide, *rden = s_wide + KP;
    for (int64_t d = blockIdx.x; d < L.D; d += gridDim.x) {
        double rs = 0.0;
        for (int q = lane; q < (KP >> 2); q += 64) {
            const int c = q / G, gv = q - c * G;
            const uint32_t mask = L.lab_mask[d * G + gv];
            const int4 nd4 = reinterpret_cast<const int4 *>(L.n_dk + d * KP)[q];
            const int4 nk4 = reinterpret_cast<const int4 *>(L.n_k)[q];
            const int nds[4] = {nd4.x, nd4.y, nd4.z, nd4.w}, nks[4] = {nk4.x, nk4.y, nk4.z, nk4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double t = (double)nds[j] + (((mask >> (4 * c + j)) & 1u) ? L.alpha : 0.0);
                th[(q << 2) | j] = t;
                rden[(q << 2) | j] = (double)nks[j] + L.vbeta;
                rs = rs + t;
            }
        }
        rs = wave_allsum(rs);
        double acc = 0.0;
        for (int64_t i = L.doc_off[d]; i < L.doc_off[d + 1]; ++i) {
            const int4 *xrow = reinterpret_cast<const int4 *>(L.n_kw + (int64_t)L.word[i] * KP);
            double dot = 0.0;
            for (int q = lane; q < (KP >> 2); q += 64) {
                const int4 x4 = xrow[q];
                const int xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dot = dot + (th[(q << 2) | j] / rs) * (((double)xs[j] + L.beta) / rden[(q << 2) | j]);
            }
            dot = wave_allsum(dot);
            acc = acc - log(dot);
        }
        if (lane == 0) L.out_doc[d] = acc;
    }
}

struct WRParams {
    const int32_t *n_dk;
    const uint16_t *lab_mask;
    double *out;
    int64_t D;
    int32_t K, mode;
    double alpha, keep, share;
    int32_t leaf_start[LLDA_MAX_WIDE_LEAVES], leaf_len[LLDA_MAX_WIDE_LEAVES];
    WideLayout w;
};

// get_theta and its running mean (LabeledLDA.py:236-239, 131-145): the row sum in numpy's order
__global__ void __launch_bounds__(64) llda_readout_theta_wide_kernel(const WRParams P)
{
    extern __shared__ double s_wide[];
    const WideLayout &W = P.w;
    const int lane = threadIdx.x, KP = W.KP, G = W.G, T = W.T;
    double *wv = s_wide;
    for (int64_t d = blockIdx.x; d < P.D; d += gridDim.x) {
        for (int q = lane; q < (KP >> 2); q += 64) {
            const int c = q / G, gv = q - c * G;
            const uint32_t mask = P.lab_mask[d * G + gv];
            const int4 nd4 = reinterpret_cast<const int4 *>(P.n_dk + d * KP)[q];
            const int nds[4] = {nd4.x, nd4.y, nd4.z, nd4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv[(q << 2) | j] = (double)nds[j] + (((mask >> (4 * c + j)) & 1u) ? P.alpha : 0.0);   // n_d_k + labs*alpha
        }
        const double rs = wide_sum(wv, W, lane);
        for (int q = lane; q < (KP >> 2); q += 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pos = (q << 2) | j;
                const int k = wide_topic_of(P.leaf_start, P.leaf_len, G, T, pos);
                if (k < 0) continue;
                double *o = P.out + d * P.K + k;
                const double cur = wv[pos] / rs;
                if (P.mode == 0) *o = cur;
                else {
                    const double a = P.keep * *o, b = P.share * cur;       // two roundings, then the sum (no FMA)
                    *o = a + b;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Test-time fold-in, wide layouts (LabeledLDA.prep4test / run_test, LabeledLDA.py:155-212; see kernel_foldin.hpp for the
// narrow kernels and the meaning of the arguments).  The global matrices keep the fold-in's LANE-MAJOR order (entry
// gv * T + s); the wavefront's LDS copy is in position order, which is what wide_sum / wide_draw walk.
// ---------------------------------------------------------------------------------------------
struct WFParams {
    FParams f;
    WideLayout w;
};

// `while prob.sum() > 1: prob /= c`
__device__ __forceinline__ void wide_shrink(double *wv, const WideLayout &W, int lane, double c, double c_rcp)
{
    for (int guard = 0; guard < (1 << 28); ++guard) {
        const double s = wide_sum(wv, W, lane);
        if (!(s > 1.0)) break;
        wide_div(wv, W, lane, c, c_rcp);
    }
}

// wv (position order) <- row of a lane-major matrix
__device__ __forceinline__ void wide_load_lm(double *wv, const double *row, const WideLayout &W, int lane)
{
    for (int t = 0; t < W.NT; ++t) {
        const int gv = t * 64 + lane;
        for (int s = 0; s < W.T; ++s) wv[pos_of_rt(W.G, W.T, gv, s)] = row[gv * W.T + s];
    }
}

__global__ void __launch_bounds__(64) llda_foldin_init_wide_kernel(const WFParams P)
{
    extern __shared__ double s_wide[];
    const FParams &F = P.f;
    const WideLayout &W = P.w;
    const int lane = threadIdx.x;
    double *wv = s_wide;
    for (int64_t site = blockIdx.x; site < F.n_sites; site += gridDim.x) {
        int64_t lo = 0, hi = F.D;                       // document of the site: last d with doc_off[d] <= site
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (F.doc_off[mid] <= site) lo = mid; else hi = mid;
        }
        const int64_t d = lo;
        const int n = (int)(site - F.doc_off[d]);
        const uint32_t gdoc = F.doc_ids ? (uint32_t)F.doc_ids[d] : (uint32_t)(d + F.doc_base);
        uint32_t r0 = (uint32_t)(n >> 1), r1 = gdoc, r2 = F.doc_stream ? F.doc_stream[d] : F.stream_id, r3 = 0xFFFFFFFFu;
        philox4x32_10(r0, r1, r2, r3, F.key0, F.key1);
        const uint32_t ra = (n & 1) ? r2 : r0, rb = (n & 1) ? r3 : r1;
        const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
        wide_load_lm(wv, F.phn + (int64_t)F.init_idx[site] * W.KP, W, lane);
        wide_shrink(wv, W, lane, F.c_init, 1.0 / F.c_init);
        int zn = wide_draw(wv, W, u, lane);
        if (zn < 0) {
            zn = 0;
            if (lane == 0 && F.status) atomicOr(F.status, 1);
        }
        if (lane == 0) {
            int ln, sn;
            lane_slot_of_rt(W.G, W.T, zn, ln, sn);
            F.z[site] = zn;
            atomicAdd(F.n_dk + d * W.KP + ln * W.T + sn, F.freq[site]);
        }
    }
}

// the sweeps (the initial assignments always come from llda_foldin_init_wide_kernel)
__global__ void __launch_bounds__(64) llda_foldin_wide_kernel(const WFParams P)
{
    extern __shared__ double s_wide[];
    const FParams &F = P.f;
    const WideLayout &W = P.w;
    const int lane = threadIdx.x, KP = W.KP, G = W.G, T = W.T, NT = W.NT;
    double *wv = s_wide;
    int *s_ndk = reinterpret_cast<int *>(wv + KP);              // position order
    for (int64_t d = blockIdx.x; d < F.D; d += gridDim.x) {
        const int64_t s0 = F.doc_off[d];
        const int len = (int)(F.doc_off[d + 1] - s0);
        const uint32_t gdoc = F.doc_ids ? (uint32_t)F.doc_ids[d] : (uint32_t)(d + F.doc_base);
        const uint32_t stream_id = F.doc_stream ? F.doc_stream[d] : F.stream_id;
        const double *ph = F.ph + (F.ph_base ? F.ph_base[d] : 0);
        int32_t *ndk_row = F.n_dk + d * KP;                     // lane-major
        double *th_row = F.th + d * KP;
        for (int t = 0; t < NT; ++t) {
            const int gv = t * 64 + lane;
            for (int s = 0; s < T; ++s) {
                s_ndk[pos_of_rt(G, T, gv, s)] = ndk_row[gv * T + s];
                th_row[gv * T + s] = 0.0;
            }
        }
        int ntot = 0;
        for (int n = 0; n < len; ++n) ntot += F.freq[s0 + n];
        const double c1 = F.c_loop, c1r = 1.0 / c1;

        for (int sweep = 0; sweep < F.iters; ++sweep) {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            for (int n = 0; n < len; ++n) {
                const int v = F.word[s0 + n], f = F.freq[s0 + n];
                if ((n & 127) == 0) {
                    r0 = (uint32_t)(n >> 1) + (uint32_t)lane; r1 = gdoc; r2 = stream_id; r3 = (uint32_t)sweep;
                    philox4x32_10(r0, r1, r2, r3, F.key0, F.key1);
                }
                const int holder = (n >> 1) & 63;
                const uint32_t ra = (uint32_t)__shfl((int)((n & 1) ? r2 : r0), holder, 64);
                const uint32_t rb = (uint32_t)__shfl((int)((n & 1) ? r3 : r1), holder, 64);
                const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
                const int zo = F.z[s0 + n];
                if (lane == 0) s_ndk[zo] -= f;                                   // n_dk[z] -= f
                const double *brow = ph + (int64_t)v * KP;
                for (int t = 0; t < NT; ++t) {
                    const int gv = t * 64 + lane;
                    for (int s = 0; s < T; ++s) {
                        const int pos = pos_of_rt(G, T, gv, s);
                        wv[pos] = ((double)s_ndk[pos] + F.alpha) * brow[gv * T + s];      // num_a * b
                    }
                }
                double S = wide_sum(wv, W, lane);
                if (F.beta_fallback && S == 0.0) {         // 0/0 raises in the reference (CascadeLDA.py:225-230)
                    for (int t = 0; t < NT; ++t) {
                        const int gv = t * 64 + lane;
                        for (int s = 0; s < T; ++s) {
                            const int pos = pos_of_rt(G, T, gv, s);
                            const bool real = F.slot_valid[gv * T + s] != 0;
                            wv[pos] = real ? ((double)s_ndk[pos] + F.alpha) * (brow[gv * T + s] + F.beta) : 0.0;
                        }
                    }
                    S = wide_sum(wv, W, lane);
                }
                wide_div(wv, W, lane, S, 1.0 / S);                               // prob /= prob.sum()
                wide_shrink(wv, W, lane, c1, c1r);
                int zn = wide_draw(wv, W, u, lane);
                if (zn < 0) {                              // all-zero / NaN probabilities: the reference would raise
                    zn = zo;
                    if (lane == 0 && F.status) atomicOr(F.status, 1);
                }
                if (lane == 0) {
                    s_ndk[zn] += f;                                              // n_dk[new_z] += f
                    F.z[s0 + n] = zn;
                }
            }
            // thinned running average of the document-topic state (LabeledLDA.py:199-211)
            if ((sweep + 1) % F.thinning == 0) {
                const int s2 = (sweep + 1) / F.thinning;
                const double tot = (double)ntot;
                const double f_old = F.avg_mode == 0 ? (double)(s2 - 1) / (double)s2 : (double)(s2 - 1) / (double)s2;
                const double f_new = F.avg_mode == 0 ? 1.0 / (double)s2 : 1.0 - f_old;
                for (int t = 0; t < NT; ++t) {
                    const int gv = t * 64 + lane;
                    for (int s = 0; s < T; ++s) {
                        const double cur = (double)s_ndk[pos_of_rt(G, T, gv, s)] / tot;
                        double *a = th_row + gv * T + s;
                        if (s2 == 1) *a = cur;
                        else {
                            const double old_part = f_old * *a;
                            const double new_part = f_new * cur;
                            *a = old_part + new_part;
                        }
                    }
                }
            }
        }
        for (int t = 0; t < NT; ++t) {
            const int gv = t * 64 + lane;
            for (int s = 0; s < T; ++s) ndk_row[gv * T + s] = s_ndk[pos_of_rt(G, T, gv, s)];
        }
    }
}

}  // namespace
