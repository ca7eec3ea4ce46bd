// kernel_foldin.hpp -- llda_foldin_kernel: the test-time sampler
// Part of the single translation unit llda_gibbs.hip (included in order; see the contents list there).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Test-time fold-in sampler: LabeledLDA.prep4test / run_test (LabeledLDA.py:155-212).
// One lane group per held-out document; topic-word loadings ph_hat are fixed, only the document's n_dk
// moves.  phn = ph_hat with every word column normalised (prep4test, LabeledLDA.py:162-167, done by
// the host); both matrices are word-major, (V, KP) doubles, LANE-MAJOR (entry lane*T + slot: a lane's slots are
// contiguous 16-byte pairs of doubles already); z holds positions of the sweep kernels' row layout (pos_of<G, T>).
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
    const int64_t *doc_ids;  // optional per-document RNG ids (else doc_base + d)
    double alpha, beta;
    double c_init, c_loop;   // the reference's "while prob.sum() > 1: prob /= c" constants
    uint32_t key0, key1, stream_id;
    int32_t iters, thinning;
    int32_t beta_fallback;   // CascadeLDA.cascade_test: prob.sum() == 0 -> prob = num_a * (b + beta)
    int32_t avg_mode;        // 0: (s-1)/s*avg + (1/s)*cur   1: m*avg + (1-m)*cur with m = (s-1)/s
    int32_t exact_only;      // test hook: every site through the reference's pipeline (no decided tier)
    int32_t last_leaf, tail, tail_row, n_rounds, xor_tree;
    uint32_t rounds_pk[LLDA_MAX_ROUNDS];
    int64_t n_sites;         // > 0: the initial assignments were drawn by llda_foldin_init_kernel (one lane group per SITE)
    const int64_t *ph_base;  // optional [D]: element offset of the document's loadings inside ph (several label subsets,
    const uint32_t *doc_stream;   // optional [D]: ... and its RNG stream id) -- documents of different calls in one launch
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

// `while prob.sum() > 1: prob /= c`  (LabeledLDA.py:170-171, 192-193); c_rcp = RN(1/c).
// The reference evaluates the sum before every division; CascadeLDA's prep4test rows sum to 1 - p_0 + 1/len(doc) and
// c = 1.0000005, so one site can take tens of thousands of steps, and the numpy-ordered sum (a cross-lane reduction) is
// by far the dearest part of a step.  Most of those sums can be skipped WITHOUT changing the result: every element is
// non-increasing under p <- RN(p / c) (c > 1, rounding is monotone) and a tree of floating-point additions is monotone
// in each of its non-negative leaves, so the computed sum S(k) after k steps is non-increasing in k; the reference
// stops at the first k with S(k) <= 1, hence if S(k0) > 1 it has not stopped at any k <= k0.  So: run k0 division
// steps on a copy without looking at the sum (k0 from log(S)/log(c), taken short), then compute S(k0); when it is
// still > 1 the copy is where the reference is after k0 steps and the loop goes on from there; when it is not (the
// estimate was too long -- never seen) the copy is dropped and the steps are taken one by one as the reference does.
template <int G, int T, bool HAS_TAIL>
__device__ __forceinline__ void shrink_to_one(double (&p)[T], double c, double c_rcp, const KParams &K, int lig, int lane)
{
    double s = group_sum<G, T, HAS_TAIL>(p, K, lig, lane);     // group-uniform: every lane holds the same s
    bool may_jump = true;
    int steps = 0;
    while (s > 1.0 && steps < (1 << 28)) {                     // (the reference loops until the sum is <= 1)
        if (may_jump && s > 1.0 + 128.0 * (c - 1.0)) {         // (far enough from 1 for a jump to pay: the logs are dear)
            const double est = log(s) / log(c);                // real-arithmetic distance to sum = 1, in steps
            const int k0 = (est > 96.0 && est < 2.0e8) ? (int)(est * 0.998) - 16 : 0;
            if (k0 > 0) {
                double q[T];
#pragma unroll
                for (int k = 0; k < T; ++k) q[k] = p[k];
                for (int i = 0; i < k0; ++i) {
#pragma unroll
                    for (int k = 0; k < T; ++k) q[k] = div_by(q[k], c, c_rcp);
                }
                const double sq = group_sum<G, T, HAS_TAIL>(q, K, lig, lane);
                if (sq > 1.0) {
#pragma unroll
                    for (int k = 0; k < T; ++k) p[k] = q[k];
                    s = sq; steps += k0;
                    continue;
                }
                may_jump = false;
            }
        }
#pragma unroll
        for (int k = 0; k < T; ++k) p[k] = div_by(p[k], c, c_rcp);
        s = group_sum<G, T, HAS_TAIL>(p, K, lig, lane);
        ++steps;
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
    const uint32_t gdoc = P.doc_ids ? (uint32_t)P.doc_ids[d] : (uint32_t)(d + P.doc_base);
    const uint32_t stream_id = P.doc_stream ? P.doc_stream[d] : P.stream_id;
    const double *ph = P.ph + (P.ph_base ? P.ph_base[d] : 0);
    int ndk[T];
    double avg[T];
#pragma unroll
    for (int s = 0; s < T; ++s) { ndk[s] = 0; avg[s] = 0.0; }
    int ntot = 0;
    const double c0 = P.c_init, c0r = 1.0 / c0, c1 = P.c_loop, c1r = 1.0 / c1;
    const bool pre = P.n_sites > 0;
    if (pre) {                                      // start state left by llda_foldin_init_kernel
        load_row<T>(P.n_dk + d * KP + lig * T, ndk);
        for (int n = 0; n < len; ++n) ntot += P.freq[s0 + n];
    }

    // thinned running average of the document-topic state (LabeledLDA.py:199-211)
    auto thin = [&](const int sweep) {
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
    };
    // one site of a sweep >= 0 given its scalars and its row of loadings; returns the new position
    auto sample = [&](const int n, const int sweep, const int f, const int zo, const double (&b)[T], uint32_t &r0, uint32_t &r1,
                      uint32_t &r2, uint32_t &r3) {
        if ((n & (2 * G - 1)) == 0) {
            r0 = (uint32_t)(n >> 1) + (uint32_t)lig; r1 = gdoc; r2 = stream_id; r3 = (uint32_t)sweep;
            philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
        }
        const int holder = (n >> 1) & (G - 1);
        const uint32_t ra = (uint32_t)__shfl((int)((n & 1) ? r2 : r0), holder, G);
        const uint32_t rb = (uint32_t)__shfl((int)((n & 1) ? r3 : r1), holder, G);
        const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
        {
            int lo, so;
            lane_slot_of<G, T>(zo, lo, so);
            onehot_add1<T>(ndk, (lig == lo) ? (1u << so) : 0u, f);               // n_dk[z] -= f
        }
        double w[T];
#pragma unroll
        for (int s = 0; s < T; ++s) w[s] = ((double)ndk[s] + P.alpha) * b[s];   // num_a * b
        // ---- decided tier (DESIGN.md 4.3 applied to this pipeline): the reference normalises the scores, divides them by c
        // while their sum exceeds 1 -- after `prob /= prob.sum()` the sum is within (K + 2) * 2^-53 of 1 and c - 1 >= 5e-7,
        // so that is at most ONE division -- and draws by inverse CDF; all of it is a common positive scale on every score
        // plus a few roundings (relative 2^-51 at most), so WHICH topic is drawn is decided by the signs of
        // prefix - u * total of the UNNORMALISED scores wherever those differences exceed 2^-40 of the total.  A site with
        // a difference inside that band, a zero / non-finite total or no hit takes the reference's pipeline below -- the
        // topic is the reference's either way; the per-site cost drops from ~600 dependent fp64 instructions (numpy-ordered
        // sum, IEEE divisions, second sum, scan) to ~60, and this kernel is one dependent chain per document.
        if (!P.exact_only) {
            double q[T];
            uint32_t pm = 0;
            double run = 0.0;
#pragma unroll
            for (int s = 0; s < T; ++s) {
                run = run + w[s];
                q[s] = run;
                pm |= (w[s] > 0.0 ? 1u : 0u) << s;
            }
            double X = group_scan<G>(run, lig);
            double tot = bcast_last<G>(X, lane);
            if (P.beta_fallback && tot == 0.0) {
                // every score is an exact zero (the word loads on none of these topics -- common in CascadeLDA's label subsets):
                // the reference's sum is 0 too, its 0/0 raises, and it takes prob = num_a * (b + beta) (CascadeLDA.py:225-230)
                // -- the same scores here, then the same decision
                pm = 0;
                run = 0.0;
#pragma unroll
                for (int s = 0; s < T; ++s) {
                    const bool real = P.slot_valid[lig * T + s] != 0;
                    const double ws = real ? ((double)ndk[s] + P.alpha) * (b[s] + P.beta) : 0.0;
                    run = run + ws;
                    q[s] = run;
                    pm |= (ws > 0.0 ? 1u : 0u) << s;
                }
                X = group_scan<G>(run, lig);
                tot = bcast_last<G>(X, lane);
            }
            const double prev = dpp_f64<DPP_WAVE_SHR1>(X);
            const double tg = u * tot - (lig ? prev : 0.0);
            const double margin = tot * 0x1p-40;
            int cnt_lo = 0, cnt_hi = 0;
#pragma unroll
            for (int s = 0; s < T; ++s) {
                cnt_lo += (q[s] <= tg - margin) ? 1 : 0;
                cnt_hi += (q[s] <= tg + margin) ? 1 : 0;
            }
            const uint32_t fm = pm & (0xFFFFu << cnt_lo);
            const int gbase = lane & ~(G - 1);
            const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
            const bool unsure = (cnt_lo != cnt_hi) || !(tot > 0.0) || !(margin < tot) || !(tot < 1.0e300);
            const uint64_t gu = (__ballot(unsure) >> gbase) & gmask, gf = (__ballot(fm != 0) >> gbase) & gmask;
            if (gu == 0 && gf != 0) {
                const int sl = (int)__ffsll((unsigned long long)gf) - 1;
                const int my = (int)__ffs((int)(fm | 0x10000u)) - 1;
                const int zn = pos_of<G, T>(sl, __shfl(my, sl, G));
                int ln, sn;
                lane_slot_of<G, T>(zn, ln, sn);
                onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);      // n_dk[new_z] += f
                return zn;
            }
        }
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
        int zn = draw_position<G, T, false>(w, u, 0u, true, lig, lane);
        if (zn < 0) {                         // all-zero / NaN probabilities: the reference would raise
            zn = zo < 0 ? 0 : zo;
            if (lig == 0 && P.status) atomicOr(P.status, 1);
        }
        {
            int ln, sn;
            lane_slot_of<G, T>(zn, ln, sn);
            onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);          // n_dk[new_z] += f
        }
        return zn;
    };

    if (pre) {
        // The start state was drawn by llda_foldin_init_kernel: `iters` sweeps with the memory operations software-pipelined --
        // a site's row of loadings depends on its word, so a plain loop is two dependent memory round trips per site and
        // the kernel (one short chain per document, the chip mostly idle) is bound by exactly that latency: the word and
        // frequency of the site after the next and the row + old topic of the next site are in flight while a site is
        // sampled.  The sites of a document are walked cyclically (site len of a sweep is site 0 of the next one).
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if (len > 0 && P.iters > 0) {
            int v_1 = P.word[s0], f_1 = P.freq[s0];                     // site "next"
            const int n2 = len > 1 ? 1 : 0;
            int v_2 = P.word[s0 + n2], f_2 = P.freq[s0 + n2];           // site "after next"
            double b_1[T];
            load_row_f64<T>(ph + (int64_t)v_1 * KP + lig * T, b_1);
            int zo_1 = P.z[s0];
            int nn = n2;                                                 // index of the site whose scalars are in ._2
            for (int sweep = 0; sweep < P.iters; ++sweep) {
                for (int n = 0; n < len; ++n) {
                    const int f = f_1, zo = zo_1;
                    double b[T];
#pragma unroll
                    for (int s = 0; s < T; ++s) b[s] = b_1[s];
                    // next site: its row and its old topic (written one sweep ago -- or just now when the document has one site)
                    v_1 = v_2; f_1 = f_2;
                    load_row_f64<T>(ph + (int64_t)v_1 * KP + lig * T, b_1);
                    if (len > 1) zo_1 = P.z[s0 + nn];
                    // the site after the next: word and frequency
                    nn = nn + 1 < len ? nn + 1 : 0;
                    v_2 = P.word[s0 + nn]; f_2 = P.freq[s0 + nn];
                    const int zn = sample(n, sweep, f, zo, b, r0, r1, r2, r3);
                    if (lig == 0) P.z[s0 + n] = zn;
                    if (len == 1) zo_1 = zn;
                }
                thin(sweep);
            }
        }
    } else {
    // sweep = -1: prep4test (initial assignments from the normalised loadings), then `iters` sweeps
    for (int sweep = -1; sweep < P.iters; ++sweep) {
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int n = 0; n < len; ++n) {
            const int v = P.word[s0 + n], f = P.freq[s0 + n];
            if (sweep < 0) {
                if ((n & (2 * G - 1)) == 0) {
                    r0 = (uint32_t)(n >> 1) + (uint32_t)lig; r1 = gdoc; r2 = stream_id; r3 = (uint32_t)sweep;
                    philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
                }
                const int holder = (n >> 1) & (G - 1);
                const uint32_t ra = (uint32_t)__shfl((int)((n & 1) ? r2 : r0), holder, G);
                const uint32_t rb = (uint32_t)__shfl((int)((n & 1) ? r3 : r1), holder, G);
                const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
                double w[T];
                load_row_f64<T>(P.phn + (int64_t)P.init_idx[s0 + n] * KP + lig * T, w);
                shrink_to_one<G, T, HAS_TAIL>(w, c0, c0r, K, lig, lane);
                ntot += f;
                int zn = draw_position<G, T, false>(w, u, 0u, true, lig, lane);
                if (zn < 0) {
                    zn = 0;
                    if (lig == 0 && P.status) atomicOr(P.status, 1);
                }
                int ln, sn;
                lane_slot_of<G, T>(zn, ln, sn);
                onehot_add1<T>(ndk, (lig == ln) ? (1u << sn) : 0u, -f);
                if (lig == 0) P.z[s0 + n] = zn;
            } else {
                double b[T];
                load_row_f64<T>(ph + (int64_t)v * KP + lig * T, b);
                const int zn = sample(n, sweep, f, P.z[s0 + n], b, r0, r1, r2, r3);
                if (lig == 0) P.z[s0 + n] = zn;
            }
        }
        thin(sweep);
    }
    }
    store_row<T>(P.n_dk + d * KP + lig * T, ndk);
#pragma unroll
    for (int s = 0; s < T; ++s) P.th[d * KP + lig * T + s] = avg[s];
}

// prep4test's initial assignments (LabeledLDA.py:168-175, CascadeLDA.py:199-206), one lane group per SITE: the draw of
// a site depends only on its word's row of initial probabilities and on its keyed uniform, not on the other sites,
// and the reference's `while prob.sum() > 1: prob /= c` can take tens of thousands of iterations for one site (the
// CascadeLDA rows sum to 1 - p_0 + 1/len(doc) and c = 1.0000005), so walking a document's sites one after the other
// made this phase cost more than all the sweeps.  z gets the position, n_dk the count (integer atomics).
template <int G, int T, bool HAS_TAIL>
__global__ void __launch_bounds__(256) llda_foldin_init_kernel(const FParams P)
{
    constexpr int KP = G * T;
    constexpr int GPB = 256 / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int lig = tid & (G - 1);
    const int64_t site = (int64_t)blockIdx.x * GPB + tid / G;
    if (site >= P.n_sites) return;
    KParams K;
    K.last_leaf = P.last_leaf; K.tail = P.tail; K.tail_row = P.tail_row; K.n_rounds = P.n_rounds;
    K.xor_tree = P.xor_tree;
#pragma unroll
    for (int r = 0; r < LLDA_MAX_ROUNDS; ++r) K.rounds_pk[r] = P.rounds_pk[r];
    // document of the site: last d with doc_off[d] <= site
    int64_t lo = 0, hi = P.D;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (P.doc_off[mid] <= site) lo = mid; else hi = mid;
    }
    const int64_t d = lo;
    const int n = (int)(site - P.doc_off[d]);
    const uint32_t gdoc = P.doc_ids ? (uint32_t)P.doc_ids[d] : (uint32_t)(d + P.doc_base);
    uint32_t r0 = (uint32_t)(n >> 1), r1 = gdoc, r2 = P.doc_stream ? P.doc_stream[d] : P.stream_id, r3 = 0xFFFFFFFFu;   // sweep word of prep4test: -1
    philox4x32_10(r0, r1, r2, r3, P.key0, P.key1);
    const uint32_t ra = (n & 1) ? r2 : r0, rb = (n & 1) ? r3 : r1;
    const double u = ((double)(ra >> 5) * 67108864.0 + (double)(rb >> 6)) * (1.0 / 9007199254740992.0);
    double w[T];
    load_row_f64<T>(P.phn + (int64_t)P.init_idx[site] * KP + lig * T, w);
    shrink_to_one<G, T, HAS_TAIL>(w, P.c_init, 1.0 / P.c_init, K, lig, lane);
    int zn = draw_position<G, T, false>(w, u, 0u, true, lig, lane);
    if (zn < 0) {
        zn = 0;
        if (lig == 0 && P.status) atomicOr(P.status, 1);
    }
    if (lig == 0) {
        int ln, sn;
        lane_slot_of<G, T>(zn, ln, sn);
        P.z[site] = zn;
        atomicAdd(P.n_dk + d * KP + ln * T + sn, P.freq[site]);
    }
}

}  // namespace
